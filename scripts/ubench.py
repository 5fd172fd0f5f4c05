#!/usr/bin/env python
"""FP64 VALU roofline probe on the box: measured issue rates and dependent latencies of v_fma/mul/add/rsq/rcp_f64
(SURVEY 8d asks for a v_fma_f64 microbenchmark because the local guide lists only the FP32 vector peak)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib  # noqa: E402

t = bs.StarTree(None)
L = _lib.lib()
res = {}


def run(kind, blocks, iters):
    ms, gi = C.c_double(), C.c_double()
    best = 0.0
    for _ in range(3):
        _lib.check(_lib.debug_lib().bs_debug_ubench(t.handle, kind, blocks, iters, C.byref(ms), C.byref(gi)), "ubench")
        best = max(best, gi.value / ms.value * 1e3)
    return best  # 1e9 lane-instructions per second


for kind, name in enumerate(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64")):
    r = run(kind, 256 * 8, 20000 if kind < 3 else 5000)
    res[name] = {"Ginstr_per_s": r, "TFLOPs_if_fma": r * 2 / 1e3, "cycles_per_wave_instr_at_2.4GHz": 256 * 4 * 64 * 2.4 / r}
# is there a cheaper seed than v_rsq_f64?  (kinds 9-12 of debug_kernels.hip; the library counts 32 instructions per trip, they run 32 / 32 / 96 / 128)
for kind, name, per_trip, n_seq in ((9, "v_rsq_f32", 32, 1), (10, "v_cvt_f32_f64+v_cvt_f64_f32", 32, 2), (11, "v_cvt_f32_f64+v_rsq_f32+v_cvt_f64_f32", 96, 3),
                                    (12, "v_rsq_f64+3*v_fma_f64", 128, 4), (13, "v_mfma_f64_4x4x4_4b", 32, 1), (14, "v_mfma_f64_4x4x4_4b+3*v_fma_f64", 128, 4)):
    r = run(kind, 256 * 8, 5000) * per_trip / 32
    res[name] = {"Ginstr_per_s": r, "cycles_per_wave_instr_at_2.4GHz": 256 * 4 * 64 * 2.4 / r, "cycles_per_sequence_at_2.4GHz": n_seq * 256 * 4 * 64 * 2.4 / r}
# dependent-chain latency: N waves per SIMD (N blocks of 256 threads per CU), c chains per lane
lat = {}
for kind, name in ((5, "fma_1chain"), (6, "fma_2chains"), (7, "fma_4chains"), (8, "rsq_1chain")):
    for wps in (1, 2, 4, 8):
        r = run(kind, 256 * wps, 4000)
        # per SIMD: wps waves each issuing instr; ns per wave-instruction = wps / (r*1e9/64/1024)
        lat[f"{name}_{wps}wave_per_simd_ns_per_wave_instr"] = wps / (r * 1e9 / 64 / 1024) * 1e9
res["dependent_latency"] = lat
print(json.dumps(res, indent=1))
