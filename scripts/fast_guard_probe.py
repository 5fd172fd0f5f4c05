#!/usr/bin/env python
"""Which rays carry FAST mode's largest deviations from STRICT?  Per ray: step count against the straight-line estimate
N0 = (|camera| + sqrt(safeDistance)) / stepSize, and the largest channel deviation |FAST - STRICT| / (|STRICT| + 1e-3).
Prints, per excess-step bin, the number of rays and the worst / 99.9th percentile deviation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402

tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
rng = np.random.default_rng(1)
cases = {"C3 default-aa": scenes.DEFAULT_AA, "C4 lensing-disk 4K": scenes.with_res(scenes.LENSING_DISK, 3840, 2160),
         "closeup": scenes.CLOSEUP, "wideangle-disk": scenes.WIDEANGLE_DISK, "fartheraway": scenes.FARTHERAWAY}
for name, cfg in cases.items():
    f = 2 if cfg["supersampling"] else 1
    n = 6_000_000
    ys, xs = rng.integers(0, f * cfg["height"], n), rng.integers(0, f * cfg["width"], n)
    tree.set_mode(_lib.BS_MODE_STRICT)
    a = bs.trace_rays(cfg, tree, ys, xs)
    tree.set_mode(_lib.BS_MODE_FAST)
    b = bs.trace_rays(cfg, tree, ys, xs)
    cam = np.linalg.norm(cfg["cam_pos"])
    safe = max(2500.0, 2 * cam * cam)
    n0 = (cam + np.sqrt(safe)) / cfg["step_size"]
    dev = (np.abs(b["rgba"][:, :3] - a["rgba"][:, :3]) / (np.abs(a["rgba"][:, :3]) + 1e-3)).max(axis=1)
    dstep = (a["steps"] != b["steps"]).sum()
    ex = a["steps"] - n0
    print(f"== {name}: N0 = {n0:.0f}, steps median {np.median(a['steps']):.0f} max {a['steps'].max()}, step-count differences {dstep}, worst dev {dev.max():.3e}")
    edges = [-1e9, 0, 10, 20, 30, 40, 50, 60, 80, 100, 150, 1e9]
    for lo, hi in zip(edges, edges[1:]):
        m = (ex >= lo) & (ex < hi)
        if m.sum():
            print(f"   excess steps [{lo:>6.0f}, {hi:>6.0f}): {m.sum():>8d} rays  worst {dev[m].max():.2e}  p99.9 {np.quantile(dev[m], 0.999):.2e}  escaped {int((a['fate'][m] == 1).sum())}")
