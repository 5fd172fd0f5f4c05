"""Cost of the post stage (bs_bloom_device = six sweeps + combine) on M CUs of the chip, sweeps planned for M CUs, for the frame
sizes / radii that matter -- the table the partition heuristic of bs_render_rgb8_batch is fitted to."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib  # noqa: E402

torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]


def masked_stream(m):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if 32 * w + b < m) for w in range(8)])
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return s


L = _lib.lib()
for (w, h, div) in ((1920, 1080, 25), (3840, 2160, 25), (1920, 1080, 10), (1280, 720, 25), (640, 360, 25), (1920, 1080, 60)):
    img = torch.rand((h, w, 3), dtype=torch.float64, device="cuda")
    out = torch.empty_like(img)
    row = []
    for m in (8, 16, 24, 32, 256):
        os.environ["BLACKSTAR_BLOOM_PLAN_CUS"] = str(m)
        tree = bs.StarTree(None)
        s = masked_stream(m)
        f = lambda: _lib.check(L.bs_bloom_device(tree.handle, img.data_ptr(), out.data_ptr(), w, h, C.c_double(0.15), div, s), "bloom")
        for _ in range(2):
            f()
        hip.hipStreamSynchronize(s)
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        hip.hipStreamSynchronize(s)
        row.append(f"M={m}: {(time.perf_counter() - t0) / 5 * 1e3:.3f}")
        tree.close()
    print(f"{w}x{h} r={w // div}: bloom ms  " + "   ".join(row), flush=True)
