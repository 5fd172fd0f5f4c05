#!/usr/bin/env python3
"""The fixed cost of a launch, re-measured with a warm-up BY TIME (round 6): frames of scenes/default.yaml (no star map, no supersampling)
from 0.13 M to 33 M rays, each warmed for 60 ms of launches (the chip needs ~35 ms of work to bring its clocks back from idle; rounds 2-5
warmed small frames by COUNT, i.e. for a few milliseconds), then 12 launches bracketed by HIP events.  Fits t = f + rays / rate.
Usage: python scripts/launch_cost_probe.py [mode]   -> one JSON line per size, then the fit."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = bs.Config.from_file(os.path.join(root, "scenes", "default.yaml"))
tree = bs.StarTree(None, device=0)
tree.set_mode(_lib.BS_MODE_FAST if mode == "fast" else _lib.BS_MODE_STRICT)
stream = torch.cuda.current_stream()
rows = []
for w, h in ((480, 270), (640, 360), (960, 540), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (5760, 3240), (7680, 4320)):
    cfg = base.with_resolution(w, h).to_bs_config()
    img = torch.empty((h, w, 3), dtype=torch.float64, device="cuda:0")
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) < 0.060 or n < 3:
        for _ in range(4):
            bs.render_device(cfg, tree, img.data_ptr(), img.numel(), stream.cuda_stream)
        n += 4
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in ev:
        a.record(stream)
        bs.render_device(cfg, tree, img.data_ptr(), img.numel(), stream.cuda_stream)
        b.record(stream)
    torch.cuda.synchronize()
    st = tree.stats()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    rows.append((w * h, ms, (int(st["steps"]) - int(st["rays"])) or 222 * w * h))   # (an A/B build without statistics: rays x the mean step count)
    print(json.dumps({"size": f"{w}x{h}", "rays": w * h, "warm_launches": n, "ms_median_of_12": round(ms, 4), "rk4_steps": rows[-1][2],
                      "frac": round(145 * rows[-1][2] / (ms * 1e-3) / 78.6e12, 4)}), flush=True)
    del img
x = np.array([r[2] for r in rows], float)   # work = executed RK4 steps
y = np.array([r[1] for r in rows], float)
A = np.vstack([np.ones_like(x), x]).T
(f, k), *_ = np.linalg.lstsq(A, y, rcond=None)
print(json.dumps({"fit": "ms = f + steps * k", "f_ms": round(float(f), 4), "Gsteps_per_s": round(1e-6 / float(k), 2),
                  "residual_ms_max": round(float(np.abs(A @ np.array([f, k]) - y).max()), 4), "mode": mode}))
tree.close()
