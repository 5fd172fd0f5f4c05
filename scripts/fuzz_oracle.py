#!/usr/bin/env python
"""The fuzz of scripts/fuzz_modes.py with the CPU ORACLE as the reference instead of STRICT: N random scenes (scripts/fuzz_scenes.py: degenerate
geometries included), each rendered by the oracle (all host cores), by the HIP library in STRICT and in FAST.  STRICT must match the oracle
in every counter (steps, fates, disk and star hits) and in every value at the strict tolerance (1e-12 relative + 1e-14: the trajectories are
bit-identical, the colours differ by libm's last bit at most); FAST in fates and within the parity bar |gpu - cpu| <= 1e-4 |cpu| + 1e-7.
The oracle is the checker here, nothing else.  Prints one JSON summary.
Usage: fuzz_oracle.py [N_SCENES [SEED [SKY]]]   SKY = small (2,000 uniform stars, default) | clustered (20,000 + 30,000 clusters, as fuzz_modes.py)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import c_oracle  # noqa: E402
from fuzz_scenes import scene  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 31337
SKY = sys.argv[3] if len(sys.argv) > 3 else "small"
rng = np.random.default_rng(SEED)
sky = synthetic.ppm_catalogue_bytes(synthetic.N_SMALL) if SKY == "small" else synthetic.clustered_catalogue_bytes(n_uniform=20000, n_clusters=30000)
tree = bs.StarTree(bs.read_map(sky))
ix = c_oracle.Index(c_oracle.read_ppm(sky))
CAP = 20000
tree.set_max_steps(CAP)
COUNTERS = ("steps", "horizon", "escaped", "capped", "disk_hits", "star_hits")
out = dict(scenes=0, values=0, rays=0, oracle_steps=0, strict_counter_mismatch_scenes=0, strict_outside=0, strict_worst_abs=0.0, strict_bit_identical_scenes=0,
           fast_fate_mismatch_scenes=0, fast_step_mismatch_scenes=0, fast_outside=0, fast_worst_abs=0.0, fast_worst_rel=0.0, fast_worst_rel_scene=None,
           fast_scenes_traced_in_strict=0, nonfinite_pattern_mismatch_scenes=0, bad=[], sky=SKY, stars=len(tree), seed=SEED, cap=CAP)
t_oracle = 0.0
for i in range(N):
    cfg = scene(rng, i)
    t0 = time.perf_counter()
    ref, ost = c_oracle.render(cfg, ix, threads=0, max_steps=CAP)
    t_oracle += time.perf_counter() - t0
    tree.set_mode(_lib.BS_MODE_STRICT); a = bs.render(cfg, tree); sa = tree.stats()
    tree.set_mode(_lib.BS_MODE_FAST); b = bs.render(cfg, tree); sb = tree.stats()
    out["scenes"] += 1; out["values"] += ref.size; out["rays"] += int(ost["rays"]); out["oracle_steps"] += int(ost["steps"])
    fin = np.isfinite(ref)
    if not (np.array_equal(np.isfinite(a), fin) and np.array_equal(np.isfinite(b), fin)):
        out["nonfinite_pattern_mismatch_scenes"] += 1
    cm = any(int(sa[k]) != int(ost[k]) for k in COUNTERS)
    out["strict_counter_mismatch_scenes"] += int(cm)
    nonfinite_mismatch = int((np.isfinite(a) != fin).sum()) + int((np.isfinite(b) != fin).sum())   # a NaN / inf on one side only is OUTSIDE, whatever the comparison says
    da = np.abs(a - ref)[fin]
    bad_a = int((~(da <= 1e-14 + 1e-12 * np.abs(ref[fin]))).sum())   # (~(<=): a NaN from the GPU where the oracle is finite counts)
    out["strict_outside"] += bad_a
    out["strict_worst_abs"] = max(out["strict_worst_abs"], float(da.max()) if da.size else 0.0)
    out["strict_bit_identical_scenes"] += int(np.array_equal(a, ref, equal_nan=True))
    fm = (int(sb["horizon"]), int(sb["escaped"]), int(sb["capped"])) != (int(ost["horizon"]), int(ost["escaped"]), int(ost["capped"]))
    out["fast_fate_mismatch_scenes"] += int(fm)
    out["fast_step_mismatch_scenes"] += int(int(sb["steps"]) != int(ost["steps"]))
    out["fast_scenes_traced_in_strict"] += int(sb["effective_mode"] == _lib.BS_MODE_STRICT)
    db = np.abs(b - ref)
    bad_b = int((~(db[fin] <= 1e-7 + 1e-4 * np.abs(ref[fin]))).sum())
    out["fast_outside"] += bad_b
    out["fast_worst_abs"] = max(out["fast_worst_abs"], float(db[fin].max()) if fin.any() else 0.0)
    m = fin & (np.abs(ref) > 1e-3)
    if m.any():
        rel = np.where(m, db / np.where(m, np.abs(ref), 1.0), 0.0)
        w = float(rel.max())
        if w > out["fast_worst_rel"]:
            y, x, c = (int(v) for v in np.unravel_index(int(np.argmax(rel)), rel.shape))
            out["fast_worst_rel"] = w
            out["fast_worst_rel_scene"] = dict(index=i, cfg=cfg, pixel=[y, x, c], oracle=float(ref[y, x, c]), fast=float(b[y, x, c]))
    if (cm or bad_a or fm or bad_b or nonfinite_mismatch) and len(out["bad"]) < 5:
        out["bad"].append(dict(index=i, cfg=cfg, strict_counters_differ=cm, strict_outside=bad_a, fast_fates_differ=fm, fast_outside=bad_b, nonfinite_mismatch=nonfinite_mismatch,
                               oracle={k: int(ost[k]) for k in COUNTERS}, strict={k: int(sa[k]) for k in COUNTERS}))
out["oracle_seconds"] = t_oracle
out["oracle_threads"] = int(ost["threads"]) if N else 0
print(json.dumps(out))
