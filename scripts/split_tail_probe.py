#!/usr/bin/env python3
"""Experiment: can a frame hide its own tail?  Band A (most rows, 4 wavefronts per SIMD) on one stream, band B (the last rows) on
a second, lower-priority stream from a context limited to fewer wavefronts per SIMD: B's workgroups can only enter when A's
leave.  Per-frame time (both bands done) against the single launch, every frame synchronised (no overlap between frames)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402

stars = bs.read_map(synthetic.ppm_catalogue_bytes())
tree = bs.StarTree(stars)
trees_b = {}
for bpc in (1, 2, 3, 4):
    os.environ["BLACKSTAR_BLOCKS_PER_CU"] = str(bpc)
    trees_b[bpc] = bs.StarTree(stars)
del os.environ["BLACKSTAR_BLOCKS_PER_CU"]
for t in [tree, *trees_b.values()]:
    t.set_mode(_lib.BS_MODE_FAST)
cfg = scenes.DEFAULT_AA
H, W = cfg["height"], cfg["width"]
out = torch.empty((H, W, 3), dtype=torch.float64, device="cuda:0")
ref = torch.empty_like(out)
s1 = torch.cuda.Stream(priority=-1)
s2 = torch.cuda.Stream(priority=0)
bs.render_device(cfg, tree, ref.data_ptr(), ref.numel(), s1.cuda_stream)
torch.cuda.synchronize()


def frame(split_rows, tb):
    e0, e1, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    e0.record(s1)
    if split_rows is None:
        bs.render_device(cfg, tree, out.data_ptr(), out.numel(), s1.cuda_stream)
    else:
        r1 = H - split_rows
        bs.render_rows_device(cfg, tree, 0, r1, out.data_ptr(), r1 * W * 3, s1.cuda_stream)
        bs.render_rows_device(cfg, tb, r1, H, out.data_ptr() + r1 * W * 3 * 8, split_rows * W * 3, s2.cuda_stream)
        eb.record(s2)
        s1.wait_event(eb)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def run(name, split_rows, tb=None, n=15):
    for _ in range(3):
        frame(split_rows, tb)
    ms = float(np.median([frame(split_rows, tb) for _ in range(n)]))
    ok = bool(torch.equal(out, ref))
    print(f"{name:60s} {ms:7.3f} ms   image identical: {ok}", flush=True)


run("single launch", None)
for rows in (16, 32, 54, 108, 216):
    for bpc in (1, 2, 3, 4):
        run(f"band B = last {rows:3d} rows ({100 * rows / H:4.1f} %), {bpc} wavefront(s)/SIMD, low priority", rows, trees_b[bpc])
run("single launch", None)
