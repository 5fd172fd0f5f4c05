"""Frames per second of bs_render_device when consecutive frames alternate between 1, 2, 3, 4 streams (launches in flight),
images resident in HBM.  Two in flight is what bs_render_batch does per context; is a third worth anything?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import synthetic  # noqa: E402

cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "default-aa.yaml")).to_bs_config()
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
H, W = cfg["height"], cfg["width"]
N = 60
for rep in range(2):
    for ns in (1, 2, 3, 4, 2, 3):
        lanes = [(torch.empty((H, W, 3), dtype=torch.float64, device="cuda"), torch.cuda.Stream()) for _ in range(ns)]
        for k in range(2 * ns + 8):
            o, s = lanes[k % ns]
            bs.render_device(cfg, tree, o.data_ptr(), o.numel(), s.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(N):
            o, s = lanes[k % ns]
            bs.render_device(cfg, tree, o.data_ptr(), o.numel(), s.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / N * 1e3
        print(f"{ns} stream(s): {ms:.3f} ms per frame, {W * H / ms / 1e3:.1f} Mpixel/s", flush=True)
