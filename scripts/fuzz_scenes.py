"""The random scenes of the fuzz scripts (fuzz_modes.py: FAST vs STRICT; fuzz_oracle.py: both against the CPU oracle): cameras on the axes and
in the disk plane, the centre of the hole dead ahead (a purely radial ray: k = 0), very near and very far cameras, coarse and fine steps,
disks inside the photon sphere, odd resolutions.  scene(rng, i) draws from rng in a FIXED order: scene i of a seed is the same scene in
every script (tests/test_gpu_parity.py FUZZ_WORST names two of them by seed and index)."""
import numpy as np


def scene(rng, i):
    kind = i % 8
    r = float(np.exp(rng.uniform(np.log(1.6), np.log(400.0))))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if kind == 1: d = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1, 1])          # on an axis
    if kind == 2: d[1] = 0.0; d /= np.linalg.norm(d)                                 # in the disk plane
    cam = d * r
    look = rng.normal(size=3) * rng.uniform(0, 3)
    if kind in (1, 3): look = np.zeros(3)                                             # the hole dead ahead
    up = rng.normal(size=3)
    if kind == 1: up = np.eye(3)[(int(np.argmax(np.abs(d))) + 1) % 3]
    inner = float(rng.uniform(1.05, 8.0))
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
    if kind in (1, 3): w |= 1; h |= 1
    return dict(cam_pos=tuple(map(float, cam)), cam_lookat=tuple(map(float, look)), cam_up=tuple(map(float, up)),
                fov=float(rng.uniform(0.05, 3.0)), step_size=float(rng.choice([0.05, 0.15, 0.3, 0.5, 1.0])),
                star_intensity=float(rng.uniform(0.1, 1.0)), star_saturation=float(rng.uniform(0.0, 2.0)),
                disk_hsi=(float(rng.uniform(0, 0.999)), float(rng.uniform(0, 0.5)), float(rng.uniform(0.3, 1.2))),
                disk_opacity=float(rng.choice([0.0, 0.5, 0.95, 1.0])), disk_inner=inner, disk_outer=inner + float(rng.uniform(0.5, 30.0)),
                width=w, height=h, supersampling=bool(rng.integers(0, 2)))
