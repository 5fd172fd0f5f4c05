"""ctypes loader for blackstar_amd/libblackstar_gpu.so (the C ABI of include/blackstar_gpu.h).

The library is the product; this module only binds it.  There is no fallback of any kind: if the
shared object is missing, `lib()` raises, and if no HIP device is present `bs_create` fails.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BLACKSTAR_LIB") or os.path.join(_HERE, "libblackstar_gpu.so")  # BLACKSTAR_LIB: A/B builds of the same ABI

BS_MODE_STRICT, BS_MODE_FAST = 0, 1
BS_ABI_VERSION = 5  # include/blackstar_gpu.h
BS_MAX_STEPS_LIMIT = 1 << 30
BS_FAST_MAX_EXPECTED_STEPS = 2000  # include/blackstar_gpu.h
# The test hooks (include/blackstar_gpu_debug.h) live in a library of their own, next to the product it was built with; only tests,
# scripts/ and bench.py's issue-rate probe load it (debug_lib()).
DEBUG_SO_PATH = SO_PATH[:-3] + "_debug.so" if SO_PATH.endswith(".so") else SO_PATH + "_debug"


class BsConfig(C.Structure):
    """struct bs_config (include/blackstar_gpu.h)."""
    _fields_ = [("cam_pos", C.c_double * 3), ("cam_lookat", C.c_double * 3), ("cam_up", C.c_double * 3), ("fov", C.c_double),
                ("step_size", C.c_double), ("star_intensity", C.c_double), ("star_saturation", C.c_double),
                ("disk_hsi", C.c_double * 3), ("disk_opacity", C.c_double), ("disk_inner", C.c_double), ("disk_outer", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32), ("supersampling", C.c_int32), ("_pad", C.c_int32)]


class BsStats(C.Structure):
    """struct bs_stats_t."""
    _fields_ = [(k, C.c_uint64) for k in ("rays", "steps", "capped", "horizon", "escaped", "disk_hits", "star_hits", "wave_iters")] + \
               [("kernel_ms", C.c_double), ("wall_ms", C.c_double), ("effective_mode", C.c_int32), ("zero_copy", C.c_int32)]


class BsFilesStats(C.Structure):
    """struct bs_files_stats_t."""
    _fields_ = [("files", C.c_uint64), ("bytes", C.c_uint64), ("wall_ms", C.c_double), ("writer_busy_ms", C.c_double),
                ("buffer_wait_ms", C.c_double), ("ring", C.c_int32), ("writer_threads", C.c_int32), ("numa_node_gpu", C.c_int32),
                ("numa_node_buffers", C.c_int32), ("threads_bound", C.c_int32), ("_pad", C.c_int32)]


# struct bs_star / struct bs_ray_record as numpy structured dtypes (same layout as the C structs)
STAR_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("hue", "<f8"), ("sat", "<f8"), ("mag", "<i4"), ("_pad", "<i4")])
RECORD_DTYPE = np.dtype([("vel", "<f8", 3), ("pos", "<f8", 3), ("rgba", "<f8", 4), ("steps", "<i4"), ("fate", "<i4"),
                         ("disk_hits", "<i4"), ("star_hits", "<i4")])

# every symbol include/blackstar_gpu.h declares (the product ABI)
SYMBOLS = ("bs_create", "bs_destroy", "bs_device_count", "bs_render", "bs_render_device", "bs_host_alloc", "bs_host_free", "bs_render_rows",
           "bs_render_rows_device", "bs_render_split", "bs_render_batch", "bs_bloom_device", "bs_bloom", "bs_supersample", "bs_srgb8_device",
           "bs_srgb8", "bs_render_rgb8", "bs_render_rgb8_batch", "bs_png_bound", "bs_encode_png_device", "bs_encode_png", "bs_render_png",
           "bs_render_png_batch", "bs_render_png_files", "bs_star_lookup", "bs_set_mode", "bs_get_mode", "bs_effective_mode",
           "bs_set_max_steps", "bs_stats", "bs_last_error", "bs_abi_version", "bs_read_ppm", "bs_validate_config", "bs_hsi_to_rgb",
           "bs_files_stats", "bs_numa_node", "bs_host_page_node")
# every symbol include/blackstar_gpu_debug.h declares (libblackstar_gpu_debug.so)
DEBUG_SYMBOLS = ("bs_debug_abi_check", "bs_trace_rays", "bs_debug_sqrt_div", "bs_debug_set_disk_slots", "bs_debug_ubench", "bs_debug_png_phases",
                 "bs_debug_last_post_cus", "bs_debug_last_trial", "bs_debug_partition_choice", "bs_debug_pick_partition",
                 "bs_debug_forget_partitions", "bs_debug_star_grid", "bs_debug_srgb8_table")

_lib = None
_debug = None


class BlackstarError(RuntimeError):
    pass


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 under torch/lib.  Two HSA
    runtimes in one process cannot both own the device ("No HIP GPUs are available" in whichever initialises
    second), so when a torch install is present its runtime is made the process-wide one BEFORE this library's
    NEEDED libamdhip64.so.7 is resolved (same soname -> the loader reuses it).  Without torch (e.g. the Haskell
    host) the library resolves /opt/rocm's runtime through its RUNPATH.  This does not import torch."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # already loaded: soname match reuses it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise BlackstarError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(SO_PATH)
    L.bs_abi_version.restype = C.c_int
    if L.bs_abi_version() != BS_ABI_VERSION:  # a stale build of the same name: fail here, not at some later symbol lookup or struct read
        raise BlackstarError(f"{SO_PATH} has ABI version {L.bs_abi_version()}, this binding was written against {BS_ABI_VERSION}: rebuild it "
                             "(python -c 'import __graft_entry__ as g; g.build()')")
    vp, sz, dp = C.c_void_p, C.c_size_t, C.c_double
    L.bs_create.restype = vp
    L.bs_create.argtypes = [C.c_int, vp, sz]
    L.bs_destroy.restype = None
    L.bs_destroy.argtypes = [vp]
    L.bs_render.argtypes = [vp, C.POINTER(BsConfig), vp, sz]
    L.bs_render_device.argtypes = [vp, C.POINTER(BsConfig), vp, sz, vp]
    if hasattr(L, "bs_render_rows"):
        L.bs_render_rows.argtypes = [vp, C.POINTER(BsConfig), C.c_int, C.c_int, vp, sz]
        L.bs_render_rows_device.argtypes = [vp, C.POINTER(BsConfig), C.c_int, C.c_int, vp, sz, vp]
    if hasattr(L, "bs_host_alloc"):
        L.bs_host_alloc.restype = vp
        L.bs_host_alloc.argtypes = [vp, sz]
        L.bs_host_free.restype = None
        L.bs_host_free.argtypes = [vp]
    if hasattr(L, "bs_render_split"):
        L.bs_render_split.argtypes = [vp, C.c_int, C.POINTER(BsConfig), vp, sz]
    L.bs_render_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.bs_star_lookup.argtypes = [vp, dp, dp, vp, sz, vp, vp]
    L.bs_set_mode.argtypes = [vp, C.c_int]
    L.bs_get_mode.argtypes = [vp]
    L.bs_effective_mode.argtypes = [vp, C.POINTER(BsConfig)]
    L.bs_validate_config.argtypes = [C.POINTER(BsConfig)]
    L.bs_set_max_steps.argtypes = [vp, C.c_int]
    L.bs_stats.argtypes = [vp, C.POINTER(BsStats)]
    L.bs_last_error.restype = C.c_char_p
    L.bs_read_ppm.restype = C.c_long
    L.bs_read_ppm.argtypes = [vp, sz, vp, sz]
    L.bs_hsi_to_rgb.argtypes = [dp, dp, dp, vp]
    if hasattr(L, "bs_bloom_device"):
        L.bs_bloom_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, dp, C.c_int, vp]
        L.bs_bloom.argtypes = [vp, vp, vp, C.c_int, C.c_int, dp, C.c_int]
        L.bs_srgb8_device.argtypes = [vp, vp, vp, sz, vp]
        L.bs_srgb8.argtypes = [vp, vp, vp, sz]
    if hasattr(L, "bs_supersample"):
        L.bs_supersample.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    if hasattr(L, "bs_render_rgb8"):
        L.bs_render_rgb8.argtypes = [vp, C.POINTER(BsConfig), dp, C.c_int, vp, sz]
    if hasattr(L, "bs_render_rgb8_batch"):
        L.bs_render_rgb8_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.bs_png_bound.argtypes = [C.c_int, C.c_int, C.POINTER(sz)]
    L.bs_encode_png_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, sz, vp, vp]
    L.bs_encode_png.argtypes = [vp, vp, C.c_int, C.c_int, vp, sz, C.POINTER(sz)]
    L.bs_render_png.argtypes = [vp, C.POINTER(BsConfig), dp, C.c_int, vp, sz, C.POINTER(sz)]
    L.bs_render_png_files.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int]
    L.bs_render_png_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    L.bs_files_stats.argtypes = [vp, C.POINTER(BsFilesStats)]
    L.bs_numa_node.argtypes = [vp]
    L.bs_host_page_node.argtypes = [vp]
    _lib = L
    return L


def debug_lib() -> C.CDLL:
    """libblackstar_gpu_debug.so: the test hooks.  Loads the product first (the debug library's NEEDED libblackstar_gpu.so then resolves to
    that very object) and refuses a pair that was not built together."""
    global _debug
    if _debug is not None:
        return _debug
    lib()
    if not os.path.exists(DEBUG_SO_PATH):
        raise BlackstarError(f"{DEBUG_SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    D = C.CDLL(DEBUG_SO_PATH)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.c_double
    D.bs_debug_abi_check.restype = C.c_int
    if D.bs_debug_abi_check() != 0:
        raise BlackstarError(f"{DEBUG_SO_PATH}: {last_error()}")
    D.bs_trace_rays.argtypes = [vp, C.POINTER(BsConfig), vp, sz, vp]
    D.bs_debug_sqrt_div.argtypes = [vp, vp, vp, sz, vp, vp, C.c_int]
    D.bs_debug_set_disk_slots.argtypes = [vp, C.c_int]
    D.bs_debug_ubench.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    D.bs_debug_png_phases.argtypes = [vp, vp, C.c_int, C.c_int, vp, sz]
    D.bs_debug_last_post_cus.argtypes = [vp]
    D.bs_debug_last_trial.argtypes = [vp]
    D.bs_debug_partition_choice.argtypes = [vp, C.POINTER(BsConfig), dp, C.c_int, C.c_int, vp]
    D.bs_debug_pick_partition.argtypes = [vp, vp, C.c_int]
    D.bs_debug_forget_partitions.argtypes = [vp]
    D.bs_debug_star_grid.restype = C.c_long
    D.bs_debug_star_grid.argtypes = [vp, sz, vp, vp, sz]
    D.bs_debug_srgb8_table.argtypes = [vp]
    _debug = D
    return D


def last_error() -> str:
    return (lib().bs_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise BlackstarError(f"{what} failed (rc={rc}): {last_error()}")


def make_config(d: dict) -> BsConfig:
    c = BsConfig()
    for k in ("cam_pos", "cam_lookat", "cam_up", "disk_hsi"):
        for i in range(3):
            getattr(c, k)[i] = float(d[k][i])
    for k in ("fov", "step_size", "star_intensity", "star_saturation", "disk_opacity", "disk_inner", "disk_outer"):
        setattr(c, k, float(d[k]))
    c.width, c.height, c.supersampling = int(d["width"]), int(d["height"]), int(bool(d["supersampling"]))
    return c
