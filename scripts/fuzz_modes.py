#!/usr/bin/env python
"""Stress of FAST against STRICT on the GPU (no oracle: STRICT is the one pinned to it by the test suite): N random scenes,
including the degenerate geometries the orbital-plane reduction and the per-ray units have to survive -- cameras on the
axes and in the disk plane, the centre of the hole dead ahead (a purely radial ray: k = 0), very near and very far
cameras, coarse and fine steps, disks inside the photon sphere.  Prints one JSON summary.
Usage: fuzz_modes.py [N_SCENES [SEED [SKY]]]   SKY = small (2,000 uniform stars, default) | clustered (20,000 uniform stars + 30,000
clusters of 6..40 stars inside 0.001 rad + a dense band: about one escaping ray in twenty sums a whole cluster, so FAST's terminal
directions are compared where a star-hit SET of dozens depends on them)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 424242)
SKY = sys.argv[3] if len(sys.argv) > 3 else "small"
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_SMALL) if SKY == "small" else
                               synthetic.clustered_catalogue_bytes(n_uniform=20000, n_clusters=30000)))
tree.set_max_steps(20000)
RT, AT = 1e-4, 1e-7


from fuzz_scenes import scene as _scene  # noqa: E402  (scripts/fuzz_scenes.py: the generator, shared with fuzz_oracle.py)


def scene(i):
    return _scene(rng, i)


out = dict(scenes=0, values=0, outside_tolerance=0, fate_mismatch_scenes=0, step_mismatch_scenes=0, disk_hit_mismatch_scenes=0,
           star_hit_mismatch_scenes=0, star_hits=0, escaped=0, fast_scenes_traced_in_strict=0, worst_abs=0.0, worst_rel=0.0, worst_rel_scene=None, nonfinite_scenes=0, bad=[], sky=SKY, stars=len(tree))
for i in range(N):
    cfg = scene(i)
    tree.set_mode(_lib.BS_MODE_STRICT); a = bs.render(cfg, tree); sa = tree.stats()
    tree.set_mode(_lib.BS_MODE_FAST); b = bs.render(cfg, tree); sb = tree.stats()
    out["scenes"] += 1; out["values"] += a.size
    if not (np.isfinite(a).all() and np.isfinite(b).all()):
        out["nonfinite_scenes"] += 1
    d = np.abs(a - b)
    bad = int((d > AT + RT * np.abs(a)).sum())
    out["outside_tolerance"] += bad
    out["worst_abs"] = max(out["worst_abs"], float(np.nanmax(d)))
    m = np.abs(a) > 1e-3
    if m.any():
        rel = np.where(m, d / np.where(m, np.abs(a), 1.0), 0.0)
        worst = float(np.nanmax(rel))
        if worst > out["worst_rel"]:   # NAME the worst case: the scene, the pixel, both values -- tests/test_gpu_parity.py replays it against the oracle
            y, x, c = (int(v) for v in np.unravel_index(int(np.nanargmax(rel)), rel.shape))
            out["worst_rel"] = worst
            out["worst_rel_scene"] = dict(index=i, cfg=cfg, pixel=[y, x, c], strict=float(a[y, x, c]), fast=float(b[y, x, c]),
                                          effective_mode_of_fast=["strict", "fast"][int(sb["effective_mode"])],
                                          steps=[int(sa["steps"]), int(sb["steps"])], star_hits=[int(sa["star_hits"]), int(sb["star_hits"])])
    f = (sa["horizon"], sa["escaped"], sa["capped"]) != (sb["horizon"], sb["escaped"], sb["capped"])
    out["fate_mismatch_scenes"] += int(f)
    out["step_mismatch_scenes"] += int(sa["steps"] != sb["steps"])
    out["disk_hit_mismatch_scenes"] += int(sa["disk_hits"] != sb["disk_hits"])
    out["star_hit_mismatch_scenes"] += int(sa["star_hits"] != sb["star_hits"])  # a star on the very edge of the radius may flip: reported, not a failure
    out["star_hits"] += int(sa["star_hits"]); out["escaped"] += int(sa["escaped"])
    out["fast_scenes_traced_in_strict"] += int(sb["effective_mode"] == _lib.BS_MODE_STRICT)  # stepSize > 0.5: FAST mode falls back (bs_effective_mode)
    if (bad or f) and len(out["bad"]) < 5:
        out["bad"].append(dict(index=i, cfg=cfg, outside=bad, strict=(sa["horizon"], sa["escaped"], sa["capped"], sa["steps"]),
                               fast=(sb["horizon"], sb["escaped"], sb["capped"], sb["steps"])))
out["fast_guard"] = os.environ.get("BLACKSTAR_FAST_GUARD", "1")
print(json.dumps(out))
