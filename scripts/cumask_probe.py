"""Are stream CU masks honoured on this chip, and what do the trace kernel and the bloom cost on a subset of the CUs?
(hipExtStreamCreateWithCUMask; groundwork for running the post stage of bs_render_rgb8_batch on its own few CUs.)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402

torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


cfg_obj = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
cfg = cfg_obj.to_bs_config()
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
H, W = cfg["height"], cfg["width"]
out = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
img = torch.rand((H, W, 3), dtype=torch.float64, device="cuda")
blo = torch.empty_like(img)
L = _lib.lib()
masks = {"all 256": set(range(256)), "first 240": set(range(240)), "last 240": set(range(16, 256)), "first 128": set(range(128)),
         "even bits (128)": set(range(0, 256, 2)), "bits 0-15": set(range(16)), "bits 0-7": set(range(8)),
         "one bit in 16 (16)": set(range(0, 256, 16)), "two bits in 32 (16)": set(b for b in range(256) if b % 32 < 2),
         "all but one in 16 (240)": set(b for b in range(256) if b % 16 != 0)}
for name, bits in masks.items():
    s = masked_stream(bits)
    for _ in range(2):
        bs.render_device(cfg, tree, out.data_ptr(), out.numel(), s.value)
    hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        bs.render_device(cfg, tree, out.data_ptr(), out.numel(), s.value)
    hip.hipStreamSynchronize(s)
    t_trace = (time.perf_counter() - t0) / n * 1e3
    for _ in range(2):
        _lib.check(L.bs_bloom_device(tree.handle, img.data_ptr(), blo.data_ptr(), W, H, C.c_double(0.15), 25, s), "bloom")
    hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.check(L.bs_bloom_device(tree.handle, img.data_ptr(), blo.data_ptr(), W, H, C.c_double(0.15), 25, s), "bloom")
    hip.hipStreamSynchronize(s)
    t_bloom = (time.perf_counter() - t0) / n * 1e3
    print(f"mask {name:26s} ({len(bits):3d} CUs): trace {t_trace:7.3f} ms   bloom {t_bloom:7.3f} ms   (trace x CUs/256 = {t_trace * len(bits) / 256:.3f})", flush=True)
    hip.hipStreamDestroy(s)
# concurrency: trace on 240 CUs while the bloom runs on the other 16
st = masked_stream(set(range(16, 256)))
sp = masked_stream(set(range(16)))
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(6):
        bs.render_device(cfg, tree, out.data_ptr(), out.numel(), st.value)
        _lib.check(L.bs_bloom_device(tree.handle, img.data_ptr(), blo.data_ptr(), W, H, C.c_double(0.15), 25, sp), "bloom")
    hip.hipStreamSynchronize(st); hip.hipStreamSynchronize(sp)
    print(f"concurrent: 6 x (trace on bits 16-255 || bloom on bits 0-15): {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms per pair", flush=True)
