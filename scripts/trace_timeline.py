#!/usr/bin/env python3
"""Timeline of the persistent wavefronts of trace_frame_kernel (probe build: make -C blackstar_amd/csrc OUT=../libblackstar_probe.so
EXTRA=-DBS_TRACE_PROBE; run with BLACKSTAR_LIB=blackstar_amd/libblackstar_probe.so).  Where does the launch's fixed cost go:
entry skew, the stagger, the tail after the tile queue runs dry?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402

L = _lib.lib()   # (a probe build, through BLACKSTAR_LIB: it exports bs_debug_trace_probe itself)
if not hasattr(L, "bs_debug_trace_probe"):
    raise SystemExit("not a probe build: set BLACKSTAR_LIB to a library built with -DBS_TRACE_PROBE")
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
tree.set_mode(_lib.BS_MODE_FAST)
stream = torch.cuda.current_stream()
W = 8
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = {"c3": scenes.DEFAULT_AA, "c2": scenes.DEFAULT, "c4": scenes.with_res(scenes.LENSING_DISK, 3840, 2160)}[which]
out = torch.empty((cfg["height"], cfg["width"], 3), dtype=torch.float64, device="cuda:0")
for _ in range(6):  # sustained clocks; the probe buffer holds the last launch
    bs.render_device(cfg, tree, out.data_ptr(), out.numel(), stream.cuda_stream)
torch.cuda.synchronize()
st = tree.stats()
buf = np.zeros(W * 8192, np.uint64)
assert L.bs_debug_trace_probe(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
q = buf.reshape(-1, W)
q = q[q[:, 0] != 0]
t0 = q[:, 0].min()
us = lambda a: (a.astype(np.int64) - np.int64(t0)) / 100.0  # 100 MHz wall clock
entry, first, last, exit_ = us(q[:, 0]), us(q[:, 1]), us(q[:, 2]), us(q[:, 3])
tiles = q[:, 4].astype(np.int64)
hw = q[:, 5].astype(np.int64)
simd = ((q[:, 6].astype(np.int64) & 15) << 16) | (hw & 0xFFF0)  # XCC | SE/SH/CU/PIPE/SIMD bits of HW_ID
pc = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 10, 50, 90, 100))
print(f"{which}: kernel_ms (hipEvents) {st['kernel_ms']:.3f}; {len(q)} wavefronts, {tiles.sum()} tiles; span entry..exit {exit_.max():.1f} us")
print(f"percentiles 0/10/50/90/100 [us since the first wavefront's entry]")
print(f"  entry            {pc(entry)}")
print(f"  first tile start {pc(first)}")
print(f"  last tile end    {pc(last)}")
print(f"  exit             {pc(exit_)}")
print(f"  tiles per wave   {pc(tiles)}   mean tile time {np.mean((last - first) / np.maximum(tiles, 1)):.1f} us")
dry = np.sort(last)
print(f"tail: first wavefront out of work at {dry[0]:.1f} us, median {np.median(last):.1f}, last at {dry[-1]:.1f}: window {dry[-1] - dry[0]:.1f} us "
      f"= {100 * (dry[-1] - dry[0]) / exit_.max():.1f} % of the span; mean idle per wavefront at the end {np.mean(dry[-1] - last):.1f} us")
groups = {}
for s, l in zip(simd, last):
    groups.setdefault(int(s), []).append(l)
fin = np.array([max(v) for v in groups.values()])
cnt = np.array([len(v) for v in groups.values()])
print(f"{len(groups)} SIMDs seen, wavefronts per SIMD min/max {cnt.min()}/{cnt.max()}; SIMD finish time (its last wavefront) {pc(fin)}; "
      f"mean SIMD idle at the end {np.mean(fin.max() - fin):.1f} us")
grid = np.linspace(dry[0] - 20, dry[-1], 12)
print("wavefronts still tracing at t: " + "  ".join(f"{t:.0f}us:{int((last > t).sum())}" for t in grid))
ls = us(q[:, 7])
li = (q[:, 6].astype(np.int64) >> 8)
print(f"last tile of each wavefront: start {pc(ls)}; duration {pc(last - ls)} us; iterations {pc(li)}")
order = np.argsort(last)[-8:]
print("the 8 wavefronts that finished last: " + "; ".join(f"end {last[i]:.0f} dur {last[i] - ls[i]:.0f} it {li[i]} tiles {tiles[i]}" for i in order))
per_simd_tiles = {}
for sid, n in zip(simd, tiles):
    per_simd_tiles[int(sid)] = per_simd_tiles.get(int(sid), 0) + int(n)
pt = np.array(list(per_simd_tiles.values()))
print(f"tiles per SIMD {pc(pt)}")
slot = (np.arange(len(q)) // 4) // 256  # workgroup index / #CU: the residency slot (render.cpp: blocks_per_slot)
for k in range(int(slot.max()) + 1):
    m = slot == k
    print(f"  slot {k}: {m.sum()} wavefronts, tiles per wave {pc(tiles[m])}, last tile end {pc(last[m])}")
late = first - entry
print(f"entry -> first tile (stagger + queue pop): {pc(late)}")
