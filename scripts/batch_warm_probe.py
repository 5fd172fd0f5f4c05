"""Why is `bench.py --form batch` (bs_render_batch right after start-up) slower per frame than the same leg after the resident loop?
Times the same 20-frame bs_render_batch call several times in a row from a cold start, with and without GPU work in front."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import synthetic  # noqa: E402

cfg_obj = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
cfg = cfg_obj.to_bs_config()
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
H, W = cfg["height"], cfg["width"]
ring = [bs.alloc_image(tree, H, W) for _ in range(4)]
ring8 = [bs.alloc_image(tree, H, W, dtype=np.uint8) for _ in range(4)]
n = 20
for rep in range(6):
    t0 = time.perf_counter()
    bs.render_batch([cfg] * n, [tree], outs=[ring[i % 4] for i in range(n)])
    t1 = time.perf_counter()
    bs.render_rgb8_batch([cfg_obj] * n, [tree], outs=[ring8[i % 4] for i in range(n)])
    t2 = time.perf_counter()
    print(f"rep {rep}: bs_render_batch {(t1 - t0) / n * 1e3:.3f} ms/frame, bs_render_rgb8_batch {(t2 - t1) / n * 1e3:.3f} ms/frame", flush=True)
    if rep == 2:
        time.sleep(0.5)
        print("   (0.5 s idle)")
one = bs.alloc_image(tree, H, W)
for rep in range(3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        bs.render(cfg, tree, out=one)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"bs_render x{n}: first {ts[0]:.3f} median {np.median(ts):.3f} last {ts[-1]:.3f} ms")
