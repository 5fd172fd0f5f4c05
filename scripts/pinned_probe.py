"""What hipPointerGetAttributes / hipMemGetAddressRange report for the kinds of host memory a caller may hand to bs_render
(diagnostic for csrc/context.cpp:device_alias_of_pinned).  Run on the GPU box: python scripts/pinned_probe.py"""
import ctypes as C

import numpy as np
import torch

torch.cuda.init()
hip = C.CDLL("libamdhip64.so")


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]


hip.hipPointerGetAttributes.argtypes = [C.POINTER(Attr), C.c_void_p]
hip.hipMemGetAddressRange.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
hip.hipPointerGetAttribute.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
RANGE_START_ADDR, RANGE_SIZE = 11, 12  # hipPointer_attribute (driver_types.h)


def show(what, p):
    a = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(a), p)
    line = f"{what:46s} ptr {p:#x} attr rc {rc}"
    if rc == 0:
        line += f" type {a.type} dev {a.device} devptr {a.devicePointer or 0:#x} hostptr {a.hostPointer or 0:#x} flags {a.allocationFlags:#x}"
        for name, q in (("devptr", a.devicePointer), ("ptr", p)):
            if q:
                base, size = C.c_void_p(), C.c_size_t()
                r2 = hip.hipMemGetAddressRange(C.byref(base), C.byref(size), q)
                line += f" | range({name}) rc {r2} base {base.value or 0:#x} size {size.value}"
        st, sz = C.c_void_p(), C.c_size_t()
        ra = hip.hipPointerGetAttribute(C.byref(st), RANGE_START_ADDR, p)
        rb = hip.hipPointerGetAttribute(C.byref(sz), RANGE_SIZE, p)
        line += f" | attr RANGE_START rc {ra} {st.value or 0:#x} RANGE_SIZE rc {rb} {sz.value}"
        d = C.c_void_p()
        r3 = hip.hipHostGetDevicePointer(C.byref(d), p, 0)
        line += f" | hostGetDevPtr rc {r3} {d.value or 0:#x}"
    hip.hipGetLastError()
    print(line, flush=True)


p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), 1 << 22, 0x1) == 0
show("hipHostMalloc(portable) base", p.value)
show("hipHostMalloc(portable) interior +1 MiB", p.value + (1 << 20))
tp = torch.empty(1 << 19, dtype=torch.float64).pin_memory()
show("torch pin_memory", tp.data_ptr())
raw = np.zeros((8 << 20) + 4096, np.uint8)
base = (raw.ctypes.data + 4095) // 4096 * 4096
show("pageable numpy", base)
for flags in (0, 0x1, 0x2, 0x3):
    assert hip.hipHostRegister(base, 2 << 20, flags) == 0
    show(f"hipHostRegister(flags={flags}) base", base)
    show(f"hipHostRegister(flags={flags}) interior +1 MiB", base + (1 << 20))
    show(f"hipHostRegister(flags={flags}) one past the end", base + (2 << 20))
    hip.hipHostUnregister(base)
assert hip.hipHostRegister(base, 1 << 20, 0) == 0 and hip.hipHostRegister(base + (2 << 20), 1 << 20, 0) == 0
show("two ranges with a hole: in range A", base + 4096)
show("two ranges with a hole: in the hole", base + (1 << 20) + 4096)
show("two ranges with a hole: in range B", base + (2 << 20) + 4096)

import time
a = Attr()
n = 12150
t0 = time.perf_counter()
for k in range(n):
    hip.hipPointerGetAttributes(C.byref(a), base + (k % 256) * 4096)
print(f"hipPointerGetAttributes: {(time.perf_counter() - t0) / n * 1e6:.2f} us per call through ctypes ({n} calls = one per page of a 1080p f64 frame)")
