#!/usr/bin/env python
"""bs_render_png_files against bs_render_png_batch on the same frames (BASELINE configs[2]): per-frame time at several batch lengths, so
that what the file form costs per FRAME (nothing, if its writer keeps up) separates from what it costs per CALL (ring set-up, the writer's
start and join, the last file's write, which nothing can hide).  Prints one JSON line per batch length."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import synthetic  # noqa: E402

lengths = [int(x) for x in sys.argv[1:]] or [20, 60, 200]
cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
tree = bs.StarTree(bs.read_map(synthetic.catalogue_bytes("synthetic")))
d = tempfile.mkdtemp(prefix="bs_files_probe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    W, H = cfg.scene.resolution
    ring = [bs.alloc_png(tree, H, W) for _ in range(4)]
    warm = [cfg] * 64
    bs.render_png_batch(warm, [tree], outs=[ring[i % 4] for i in range(64)])       # the partition trial of this shape ends here
    bs.render_png_files(warm, [tree], [os.path.join(d, f"w{i % 16}.png") for i in range(64)])
    for n in lengths:
        frames = [cfg] * n
        outs = [ring[i % 4] for i in range(n)]
        paths = [os.path.join(d, f"f{i % 32}.png") for i in range(n)]
        tb, tf, st = [], [], None
        for rep in range(4):
            t0 = time.perf_counter()
            bs.render_png_batch(frames, [tree], outs=outs)
            t1 = time.perf_counter()
            bs.render_png_files(frames, [tree], paths)
            t2 = time.perf_counter()
            if rep:
                tb.append((t1 - t0) * 1e3)
                tf.append((t2 - t1) * 1e3)
                st = bs.files_stats(tree)
        b, f = min(tb), min(tf)
        print(json.dumps({"frames": n, "png_batch_ms_per_frame": b / n, "png_files_ms_per_frame": f / n, "files_over_batch": f / b,
                          "extra_ms_per_call": f - b, "writer_busy_frac": st["writer_busy_frac"], "buffer_wait_ms": st["buffer_wait_ms"],
                          "library_wall_ms": st["wall_ms"], "python_wall_ms": f}), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
    tree.close()
