#!/usr/bin/env python3
"""Per-launch fixed cost of the frame kernel: sustained back-to-back launches (image resident in HBM) of frames with different
ray counts, with and without supersampling / stars; kernel time by hipEvents around each launch, and the least-squares line
t = a + b * rays through the default-aa frames (a = what a launch costs whatever its size: start-up, and the tail after the tile
queue runs dry -- scripts/trace_timeline.py shows where it goes).  An optional first argument tags the lines (A/B of builds via
BLACKSTAR_LIB)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402

tree, empty = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes())), bs.StarTree(None)
stream = torch.cuda.current_stream()
tag = sys.argv[1] if len(sys.argv) > 1 else ""


fit = []


def run(name, cfg, t, n=20):
    t.set_mode(_lib.BS_MODE_FAST)
    out = torch.empty((cfg["height"], cfg["width"], 3), dtype=torch.float64, device="cuda:0")
    for _ in range(3):
        bs.render_device(cfg, t, out.data_ptr(), out.numel(), stream.cuda_stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(stream)
        bs.render_device(cfg, t, out.data_ptr(), out.numel(), stream.cuda_stream)
        b.record(stream)
    torch.cuda.synchronize()
    st = t.stats()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    if "default-aa" in name and "stars" in name and "no stars" not in name:
        fit.append((st["rays"], ms))
    print(f"{tag:24s} {name:50s} rays {st['rays']:9d}  {ms:7.3f} ms  {st['rays'] / ms / 1e3:7.1f} Mray/s  tiles/wave {st['rays'] / 64 / 4096:6.1f}", flush=True)


D, A = scenes.DEFAULT, scenes.DEFAULT_AA
run("640x360 noSS, stars", scenes.with_res(D, 640, 360), tree)
run("C1-size 640x480 noSS, no stars", scenes.with_res(D, 640, 480), empty)
run("1280x720 noSS, stars", scenes.with_res(D, 1280, 720), tree)
run("C2 default 1920x1080 noSS, no stars", D, empty)
run("default 1920x1080 noSS, stars", D, tree)
run("default-aa 960x540 SS (same rays as C2), stars", scenes.with_res(A, 960, 540), tree)
run("default-aa 1358x764 SS (half the rays of C3), stars", scenes.with_res(A, 1358, 764), tree)
run("C3 default-aa 1920x1080 SS, stars", A, tree)
run("default-aa 2716x1528 SS (2x rays of C3), stars", scenes.with_res(A, 2716, 1528), tree)
run("C4 lensing-disk 3840x2160 SS, stars", scenes.with_res(scenes.LENSING_DISK, 3840, 2160), tree, n=8)
run("closeup 1280x960 noSS, stars", scenes.CLOSEUP, tree)
x, y = np.array([f[0] for f in fit], float), np.array([f[1] for f in fit], float)
b, a = np.polyfit(x, y, 1)
print(f"{tag:24s} default-aa frames: t = {a * 1e3:.0f} us + rays / {1 / b / 1e3:.0f} Mray/s  (fixed cost = {100 * a / y[[i for i, f in enumerate(fit) if f[0] == 8294400][0]]:.1f} % of the C3 frame)")
