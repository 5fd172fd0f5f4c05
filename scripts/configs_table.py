#!/usr/bin/env python3
"""One line per BASELINE config that runs on one GPU (C2..C5; C1 is the CPU plumbing case), both arithmetic modes:
kernel time (hipEvents, bs_stats), wall time of bs_render into a page-locked buffer (zero copy) and of bs_render_rgb8
(render -> bloom -> sRGB8, RGB8 into a page-locked buffer), steps per ray, lane efficiency.  Then the animation batch
(24 frames of C5 through bs_render_batch into page-locked buffers).  Output: JSON lines (profiles/rNN_configs_table.jsonl)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402  (scene dictionaries only: nothing of the oracle's arithmetic runs here)

stars = bs.read_map(synthetic.ppm_catalogue_bytes())
tree, empty = bs.StarTree(stars), bs.StarTree(None)


def run(name, cfg, t, mode, n):
    t.set_mode(mode)
    h, w = cfg["height"], cfg["width"]
    out = bs.alloc_image(t, h, w)
    out8 = bs.alloc_image(t, h, w, 3, np.uint8)
    km, wall, wall8 = [], [], []
    for i in range(n + 1):
        bs.render(cfg, t, out=out)
        st = t.stats()
        t0 = time.perf_counter()
        _rgb8(cfg, t, out8)
        t1 = time.perf_counter()
        if i:  # the first pass allocates the context's scratch
            km.append(st["kernel_ms"]); wall.append(st["wall_ms"]); wall8.append((t1 - t0) * 1e3)
    px, rays = w * h, st["rays"]
    k, wl, w8 = (float(np.median(x)) for x in (km, wall, wall8))
    print(json.dumps({"cfg": name, "mode": "fast" if mode == _lib.BS_MODE_FAST else "strict", "kernel_ms": round(k, 4), "bs_render_pinned_ms": round(wl, 4),
                      "bs_render_rgb8_pinned_ms": round(w8, 4), "Mpixel_s_kernel": round(px / k / 1e3, 1), "Mray_s_kernel": round(rays / k / 1e3, 1),
                      "Mpixel_s_bs_render": round(px / wl / 1e3, 1), "Mpixel_s_bs_render_rgb8": round(px / w8 / 1e3, 1),
                      "steps_per_ray": round(st["steps"] / rays, 3), "lane_eff": round(st["steps"] / (64 * st["wave_iters"]), 5), "capped": st["capped"]}), flush=True)


def _rgb8(cfg, t, out8):
    L = _lib.lib()
    c = _lib.make_config(cfg)
    _lib.check(L.bs_render_rgb8(t.handle, c, 0.15, 25, out8.ctypes.data, out8.size), "bs_render_rgb8")


for mode in (_lib.BS_MODE_FAST, _lib.BS_MODE_STRICT):
    run("C2 default.yaml 1920x1080, no starmap", scenes.DEFAULT, empty, mode, 8)
    run("C3 default-aa.yaml 1920x1080 4xSS + stars", scenes.DEFAULT_AA, tree, mode, 8)
    run("C4 lensing-disk.yaml 3840x2160 4xSS + stars", scenes.with_res(scenes.LENSING_DISK, 3840, 2160), tree, mode, 4)
    run("C5 default-ani.yaml frame 300/600 1920x1080 4xSS + stars", scenes.ani_frame(300, 600), tree, mode, 8)

tree.set_mode(_lib.BS_MODE_FAST)
cfgs = [scenes.ani_frame(i, 600) for i in range(0, 600, 25)]
outs = [bs.alloc_image(tree, 1080, 1920) for _ in cfgs]
bs.render_batch(cfgs[:2], [tree], outs=outs[:2])
t0 = time.perf_counter()
bs.render_batch(cfgs, [tree], outs=outs)
dt = time.perf_counter() - t0
print(json.dumps({"cfg": "C5 batch: 24 frames (every 25th of 600) through bs_render_batch, page-locked buffers, one GPU", "mode": "fast",
                  "ms_per_frame": round(dt * 1e3 / len(cfgs), 4), "Mpixel_s": round(len(cfgs) * 1920 * 1080 / dt / 1e6, 1)}), flush=True)

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
anim = bs.Animation.from_file(os.path.join(root, "animations", "default-ani.yaml"))
anim.nFrames = 600
frames = bs.generate_frames(anim)[::25]
outs8 = [bs.alloc_image(tree, 1080, 1920, dtype=np.uint8) for _ in frames]
bs.render_rgb8_batch(frames[:4], [tree], outs=outs8[:4])  # (the first batch of a context makes its streams and images: ~20 ms once)
t0 = time.perf_counter()
for c, o in zip(frames, outs8):
    bs.render_rgb8(c, tree, out=o)
dt1 = time.perf_counter() - t0
t0 = time.perf_counter()
bs.render_rgb8_batch(frames, [tree], outs=outs8)
dt2 = time.perf_counter() - t0
print(json.dumps({"cfg": "C5 doRender on the device: 24 frames (every 25th of 600), render -> bloom -> sRGB8, RGB8 into page-locked buffers, one GPU", "mode": "fast",
                  "bs_render_rgb8_frame_by_frame_ms": round(dt1 * 1e3 / len(frames), 4), "bs_render_rgb8_batch_ms_per_frame": round(dt2 * 1e3 / len(frames), 4),
                  "Mpixel_s_batch": round(len(frames) * 1920 * 1080 / dt2 / 1e6, 1)}), flush=True)
