"""ctypes binding for oracle/libblackstar_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libblackstar_oracle.so")


class OrcConfig(C.Structure):
    _fields_ = [("cam_pos", C.c_double * 3), ("cam_lookat", C.c_double * 3), ("cam_up", C.c_double * 3), ("fov", C.c_double),
                ("step_size", C.c_double), ("star_intensity", C.c_double), ("star_saturation", C.c_double),
                ("disk_hsi", C.c_double * 3), ("disk_opacity", C.c_double), ("disk_inner", C.c_double), ("disk_outer", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32), ("supersampling", C.c_int32), ("_pad", C.c_int32)]


STAR_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("hue", "<f8"), ("sat", "<f8"), ("mag", "<i4"), ("_pad", "<i4")])
RECORD_DTYPE = np.dtype([("vel", "<f8", 3), ("pos", "<f8", 3), ("rgba", "<f8", 4), ("steps", "<i4"), ("fate", "<i4"),
                         ("disk_hits", "<i4"), ("star_hits", "<i4")])


class OrcStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "steps", "capped", "horizon", "escaped", "disk_hits", "star_hits")] + \
               [("seconds", C.c_double), ("threads", C.c_int32), ("_pad", C.c_int32)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "blackstar_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libblackstar_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_index_create.restype = C.c_void_p
        L.orc_index_create.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_index_destroy.argtypes = [C.c_void_p]
        L.orc_render.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(OrcStats)]
        L.orc_trace_rays.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_generate_ray.argtypes = [C.POINTER(OrcConfig), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_rk4.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hsi_to_rgb.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_star_lookup.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_star_lookup_brute.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_supersample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_read_ppm.restype = C.c_long
        L.orc_read_ppm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_bloom.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_srgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def make_config(d: dict) -> OrcConfig:
    c = OrcConfig()
    for k in ("cam_pos", "cam_lookat", "cam_up", "disk_hsi"):
        for i in range(3):
            getattr(c, k)[i] = float(d[k][i])
    for k in ("fov", "step_size", "star_intensity", "star_saturation", "disk_opacity", "disk_inner", "disk_outer"):
        setattr(c, k, float(d[k]))
    c.width, c.height, c.supersampling = int(d["width"]), int(d["height"]), int(bool(d["supersampling"]))
    return c


def stars_struct(stars6: np.ndarray) -> np.ndarray:
    """(n,6) float array x,y,z,hue,sat,mag -> structured array in orc_star/bs_star layout."""
    s = np.zeros(len(stars6), STAR_DTYPE)
    if len(stars6):
        for i, k in enumerate(("x", "y", "z", "hue", "sat")):
            s[k] = stars6[:, i]
        s["mag"] = stars6[:, 5].astype(np.int32)
    return s


class Index:
    def __init__(self, stars: np.ndarray | None):
        """stars: structured STAR_DTYPE array (or None/empty for 'no starmap')."""
        self.stars = np.ascontiguousarray(stars if stars is not None else np.zeros(0, STAR_DTYPE))
        self.h = lib().orc_index_create(self.stars.ctypes.data, len(self.stars))
        if not self.h:
            raise MemoryError("orc_index_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_destroy(self.h)
            self.h = None


def render(cfg: dict, index: Index, threads: int = 0, max_steps: int = 100000):
    c = make_config(cfg)
    out = np.empty((c.height, c.width, 3), np.float64)
    st = OrcStats()
    rc = lib().orc_render(C.byref(c), index.h, out.ctypes.data, out.size, threads, max_steps, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render rc={rc}")
    stats = {k: getattr(st, k) for k, _ in OrcStats._fields_ if k != "_pad"}
    return out, stats


def trace_rays(cfg: dict, index: Index, ys, xs, max_steps: int = 100000) -> np.ndarray:
    c = make_config(cfg)
    yx = np.ascontiguousarray(np.stack([np.asarray(ys), np.asarray(xs)], axis=1).astype(np.int32))
    rec = np.zeros(len(yx), RECORD_DTYPE)
    rc = lib().orc_trace_rays(C.byref(c), index.h, yx.ctypes.data, len(yx), max_steps, rec.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_trace_rays rc={rc}")
    return rec


def generate_ray(cfg: dict, y: int, x: int):
    c = make_config(cfg)
    v = np.zeros(3); p = np.zeros(3)
    lib().orc_generate_ray(C.byref(c), y, x, v.ctypes.data, p.ctypes.data)
    return v, p


def rk4(h, h2, vel, pos):
    vel = np.ascontiguousarray(vel, np.float64); pos = np.ascontiguousarray(pos, np.float64)
    nv = np.zeros(3); npos = np.zeros(3)
    lib().orc_rk4(h, h2, vel.ctypes.data, pos.ctypes.data, nv.ctypes.data, npos.ctypes.data)
    return nv, npos


def hsi_to_rgb(h, s, i):
    o = np.zeros(3)
    lib().orc_hsi_to_rgb(h, s, i, o.ctypes.data)
    return o


def star_lookup(index: Index, intensity, saturation, vel, brute=False):
    vel = np.ascontiguousarray(vel, np.float64)
    o = np.zeros(3)
    if brute:
        n = lib().orc_star_lookup_brute(index.stars.ctypes.data, len(index.stars), intensity, saturation, vel.ctypes.data, o.ctypes.data)
    else:
        n = lib().orc_star_lookup(index.h, intensity, saturation, vel.ctypes.data, o.ctypes.data)
    return o, n


def supersample(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.float64)
    h2, w2 = img.shape[:2]
    out = np.empty((h2 // 2, w2 // 2, 3))
    lib().orc_supersample(img.ctypes.data, h2, w2, out.ctypes.data)
    return out


def read_ppm(data: bytes) -> np.ndarray:
    cap = max(0, (len(data) - 28) // 28)
    out = np.zeros(cap, STAR_DTYPE)
    buf = np.frombuffer(data, np.uint8)
    n = lib().orc_read_ppm(buf.ctypes.data if len(buf) else None, len(data), out.ctypes.data, cap)
    if n < 0:
        raise ValueError("catalogue too short")
    return out[:n]


def bloom(strength: float, divider: int, img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.float64)
    out = np.empty_like(img)
    rc = lib().orc_bloom(strength, divider, img.ctypes.data, img.shape[0], img.shape[1], out.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_bloom rc={rc}")
    return out


def srgb8(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.float64)
    out = np.empty(img.shape, np.uint8)
    lib().orc_srgb8(img.ctypes.data, out.ctypes.data, img.size)
    return out
