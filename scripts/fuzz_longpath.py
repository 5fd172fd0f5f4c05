#!/usr/bin/env python
"""How FAST's deviation from the reference arithmetic grows with the LENGTH of the path, up to the step cap's practical range: N scenes drawn
for their expected steps per ray N0 = (|cam| + sqrt(safeDistance)) / stepSize, log-uniform in [N0_MIN, N0_MAX] (the reference's own scenes:
230; the round-5 fuzz reached 14 000), on the CLUSTERED sky (a star's weight exp(-d^2 / 2w^2) turns a terminal-direction difference e into
up to 6000 e of relative colour difference -- the amplifier that made the round-5 worst case).  Each scene is rendered by the CPU ORACLE,
by the HIP library in STRICT and in FAST.  Run with BLACKSTAR_FAST_MAX_STEPS=0 (no long-path rule: FAST's own arithmetic at every length --
how the rule's threshold was chosen) or without it (the library as shipped: frames above BS_FAST_MAX_EXPECTED_STEPS are traced in STRICT
and counted as such).
Reports, per decade-third of N0: scenes, FAST's worst relative / absolute deviation from the oracle, values outside the parity bar
|gpu - cpu| <= 1e-4 |cpu| + 1e-7, step / fate equality; STRICT against the oracle (must be exact in counters, 1e-12 in values).
Usage: fuzz_longpath.py [N_SCENES [SEED [N0_MAX [WIDTH HEIGHT]]]]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import c_oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 20261001
N0_MAX = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0e5
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (64, 36)
N0_MIN = 300.0
CAP = 1 << 20
rng = np.random.default_rng(SEED)
sky = synthetic.clustered_catalogue_bytes(n_uniform=20000, n_clusters=30000)
tree = bs.StarTree(bs.read_map(sky))
ix = c_oracle.Index(c_oracle.read_ppm(sky))
tree.set_max_steps(CAP)


def scene(rng):
    n0 = float(np.exp(rng.uniform(np.log(N0_MIN), np.log(N0_MAX))))
    h = float(rng.choice([0.5, 0.3, 0.1, 0.05, 0.02, 0.01]))
    # N0 = (r + sqrt(max(2500, 2 r^2))) / h  ->  r = N0 h / (1 + sqrt 2) once 2 r^2 > 2500, else N0 h - 50
    r = n0 * h / (1.0 + np.sqrt(2.0))
    if 2 * r * r < 2500.0:
        r = max(n0 * h - 50.0, 2.5)
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    cam = d * r
    look = rng.normal(size=3) * rng.uniform(0, 4)          # near the hole: the rays that matter pass it
    up = rng.normal(size=3)
    inner = float(rng.uniform(1.5, 6.0))
    fov = float(rng.uniform(0.02, 1.5)) * min(1.0, 40.0 / r)  # (a far camera sees the hole and its disk in a narrow cone)
    return dict(cam_pos=tuple(map(float, cam)), cam_lookat=tuple(map(float, look)), cam_up=tuple(map(float, up)), fov=max(fov, 1e-4),
                step_size=h, star_intensity=float(rng.uniform(0.2, 1.0)), star_saturation=float(rng.uniform(0.0, 2.0)),
                disk_hsi=(float(rng.uniform(0, 0.999)), float(rng.uniform(0, 0.5)), float(rng.uniform(0.3, 1.2))),
                disk_opacity=float(rng.choice([0.0, 0.95])), disk_inner=inner, disk_outer=inner + float(rng.uniform(2.0, 20.0)),
                width=W, height=H, supersampling=False)


def n0_of(cfg):
    r2 = sum(c * c for c in cfg["cam_pos"])
    return (np.sqrt(r2) + np.sqrt(max(2500.0, 2.0 * r2))) / cfg["step_size"]


EDGES = [300, 1000, 3000, 10000, 30000, 100000, 300000, 1000000]
bins = [dict(n0_from=EDGES[k], n0_to=EDGES[k + 1], scenes=0, values=0, fast_worst_rel=0.0, fast_worst_abs=0.0, fast_outside=0, fast_step_mismatch_scenes=0,
             fast_fate_mismatch_scenes=0, fast_scenes_traced_in_strict=0, strict_outside=0, strict_counter_mismatch_scenes=0, worst_scene=None)
        for k in range(len(EDGES) - 1)]
records = []
t_oracle = 0.0
for i in range(N):
    cfg = scene(rng)
    n0 = n0_of(cfg)
    t0 = time.perf_counter()
    ref, ost = c_oracle.render(cfg, ix, threads=0, max_steps=CAP)
    t_oracle += time.perf_counter() - t0
    tree.set_mode(_lib.BS_MODE_STRICT); a = bs.render(cfg, tree); sa = tree.stats()
    tree.set_mode(_lib.BS_MODE_FAST); b = bs.render(cfg, tree); sb = tree.stats()
    B = next(x for x in bins if x["n0_from"] <= n0 < x["n0_to"]) if n0 < EDGES[-1] else bins[-1]
    fin = np.isfinite(ref)
    db = np.abs(b - ref)
    da = np.abs(a - ref)
    out_b = int((~(db <= 1e-7 + 1e-4 * np.abs(ref)))[fin].sum()) + int((np.isfinite(b) != fin).sum())
    out_a = int((~(da <= 1e-14 + 1e-12 * np.abs(ref)))[fin].sum()) + int((np.isfinite(a) != fin).sum())
    m = fin & (np.abs(ref) > 1e-3)
    rel = float((db[m] / np.abs(ref[m])).max()) if m.any() else 0.0
    B["scenes"] += 1; B["values"] += int(ref.size)
    B["fast_outside"] += out_b; B["strict_outside"] += out_a
    B["fast_worst_abs"] = max(B["fast_worst_abs"], float(db[fin].max()) if fin.any() else 0.0)
    B["fast_step_mismatch_scenes"] += int(int(sb["steps"]) != int(ost["steps"]))
    B["fast_fate_mismatch_scenes"] += int(any(int(sb[k]) != int(ost[k]) for k in ("horizon", "escaped", "capped")))
    B["fast_scenes_traced_in_strict"] += int(sb["effective_mode"] == _lib.BS_MODE_STRICT)
    B["strict_counter_mismatch_scenes"] += int(any(int(sa[k]) != int(ost[k]) for k in ("steps", "horizon", "escaped", "capped", "disk_hits", "star_hits")))
    if rel > B["fast_worst_rel"]:
        B["fast_worst_rel"] = rel
        B["worst_scene"] = dict(index=i, n0=n0, mean_steps_per_ray=ost["steps"] / ost["rays"], cfg=cfg)
    records.append([round(n0, 1), round(ost["steps"] / ost["rays"], 1), rel, float(db[fin].max()) if fin.any() else 0.0, out_b,
                    int(sb["effective_mode"] == _lib.BS_MODE_STRICT)])
print(json.dumps(dict(scenes=N, seed=SEED, sky="clustered", BLACKSTAR_FAST_MAX_STEPS=os.environ.get("BLACKSTAR_FAST_MAX_STEPS", "unset (the library's rule applies)"), stars=len(tree), frame=[W, H], cap=CAP, n0_range=[N0_MIN, N0_MAX], oracle_seconds=t_oracle,
                      n0_definition="(|cam| + sqrt(max(2500, 2 |cam|^2))) / stepSize: the longest straight path through the traced volume, in steps",
                      bins=[b for b in bins if b["scenes"]],
                      per_scene_columns=["n0", "mean_steps_per_ray", "fast_worst_rel", "fast_worst_abs", "fast_outside", "traced_in_strict"],
                      per_scene=records)))
tree.close()
