"""Host-side mirror of the reference's Raytracer module: `render` and `write_img` (src/Raytracer.hs:4).

    render :: Config -> StarTree -> Image U RGB Double        (src/Raytracer.hs:53)

Here `render(cfg, startree)` returns a float64 ndarray (height, width, 3): linear light, unclamped, already
supersample-reduced -- produced entirely by the gfx950 kernel behind bs_render.  Nothing in this module
computes pixels on the host.
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib
from .config_file import Config
from .star_map import StarTree


def _bs_config(cfg) -> _lib.BsConfig:
    return _lib.make_config(cfg.to_bs_config() if isinstance(cfg, Config) else cfg)


def alloc_image(startree: StarTree, height: int, width: int, channels: int = 3, dtype=np.float64) -> np.ndarray:
    """An image buffer in page-locked host memory (`bs_host_alloc`): the GPU's copy engine writes it directly, so
    render(cfg, tree, out=buf) delivers a frame without the runtime's staging copies.  Freed when the array (and every view
    of it) is garbage."""
    import weakref
    n = int(height) * int(width) * int(channels) * np.dtype(dtype).itemsize
    L = _lib.lib()
    p = L.bs_host_alloc(startree.handle, n)
    if not p:
        raise _lib.BlackstarError(f"bs_host_alloc failed: {_lib.last_error()}")
    raw = (C.c_ubyte * n).from_address(p)
    weakref.finalize(raw, L.bs_host_free, p)
    return np.frombuffer(raw, dtype=dtype).reshape(height, width, channels)


def render(cfg, startree: StarTree, out: np.ndarray = None) -> np.ndarray:
    """render cfg tree (src/Raytracer.hs:53).  `out`: an (h, w, 3) C-contiguous float64 array to fill (e.g. from alloc_image)."""
    c = _bs_config(cfg)
    if out is None:
        out = np.empty((c.height, c.width, 3), np.float64)
    elif out.shape != (c.height, c.width, 3) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous float64 array of shape {(c.height, c.width, 3)}")
    _lib.check(_lib.lib().bs_render(startree.handle, C.byref(c), out.ctypes.data, out.size), "bs_render")
    return out


def render_rows(cfg, startree: StarTree, row0: int, row1: int, out: np.ndarray = None) -> np.ndarray:
    """Output rows [row0, row1) of the frame `cfg` describes, (row1-row0, width, 3): one band of a frame sharded by rows
    over several GPUs.  Bands concatenated are bit-identical to render(cfg).  `out`: the band's buffer (a page-locked
    one from alloc_image is written by the kernel itself)."""
    c = _bs_config(cfg)
    if not (0 <= row0 < row1 <= c.height):
        raise ValueError(f"row band [{row0}, {row1}) outside the frame's {c.height} rows")
    if out is None:
        out = np.empty((row1 - row0, c.width, 3), np.float64)
    elif out.shape != (row1 - row0, c.width, 3) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous float64 array of shape {(row1 - row0, c.width, 3)}")
    _lib.check(_lib.lib().bs_render_rows(startree.handle, C.byref(c), row0, row1, out.ctypes.data, out.size), "bs_render_rows")
    return out


def render_rows_device(cfg, startree: StarTree, row0: int, row1: int, d_out_ptr: int, out_doubles: int, stream_ptr: int = 0) -> None:
    """render_rows with the band left in HBM at d_out_ptr ((row1-row0)*width*3 doubles)."""
    c = _bs_config(cfg)
    _lib.check(_lib.lib().bs_render_rows_device(startree.handle, C.byref(c), row0, row1, d_out_ptr, out_doubles, stream_ptr or None),
               "bs_render_rows_device")


def render_device(cfg, startree: StarTree, d_out_ptr: int, out_doubles: int, stream_ptr: int = 0) -> None:
    """Enqueue a render whose image stays in HBM (d_out_ptr = device pointer, e.g. torch tensor.data_ptr())."""
    c = _bs_config(cfg)
    _lib.check(_lib.lib().bs_render_device(startree.handle, C.byref(c), d_out_ptr, out_doubles, stream_ptr or None), "bs_render_device")


def render_rgb8(cfg: Config, startree: StarTree, out: np.ndarray = None) -> np.ndarray:
    """doRender's pipeline (app/Main.hs:105-123) on the device: render, bloom if scene.bloomStrength /= 0, sRGB + 8-bit.
    Returns (h, w, 3) uint8 -- what writeImg hands to the PNG encoder.  `out`: the buffer to fill (e.g. alloc_image(tree, h, w,
    dtype=np.uint8): page-locked, written by the last kernel directly)."""
    c = _bs_config(cfg)
    if out is None:
        out = np.empty((c.height, c.width, 3), np.uint8)
    elif out.shape != (c.height, c.width, 3) or out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous uint8 array of shape {(c.height, c.width, 3)}")
    _lib.check(_lib.lib().bs_render_rgb8(startree.handle, C.byref(c), float(cfg.scene.bloomStrength), int(cfg.scene.bloomDivider),
                                         out.ctypes.data, out.size), "bs_render_rgb8")
    return out


def png_bound(height: int, width: int) -> int:
    """Bytes an (height, width, 3) uint8 frame needs at most as a PNG file of the device encoder (`bs_png_bound`)."""
    n = C.c_size_t()
    _lib.check(_lib.lib().bs_png_bound(int(width), int(height), C.byref(n)), "bs_png_bound")
    return n.value


def alloc_png(startree: StarTree, height: int, width: int) -> np.ndarray:
    """A page-locked byte buffer big enough for the PNG file of an (height, width) frame: the encoder writes the file into it itself."""
    return alloc_image(startree, 1, png_bound(height, width), channels=1, dtype=np.uint8).reshape(-1)


def _check_png_out(out: np.ndarray) -> None:
    """A file buffer handed to C as (pointer, size): it must BE size contiguous bytes (a strided view such as buf[::2] advertises its
    size but would be written contiguously, past the view)."""
    if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or out.ndim != 1 or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
        raise ValueError("out must be a flat, C-contiguous, writeable uint8 array (e.g. from alloc_png)")


def encode_png(rgb8: np.ndarray, startree: StarTree, out: np.ndarray = None) -> memoryview:
    """The file writeImg writes (src/Raytracer.hs:30-32: massiv-io writeImage = a PNG encoder) for an (h, w, 3) uint8 image, made on
    the GPU (`bs_encode_png`): filter choice, deflate, checksums.  Returns the file's bytes (a view of `out` -- e.g. alloc_png(...) --
    if given).  Decoding it (zlib, libpng, Pillow) gives back rgb8 exactly."""
    rgb8 = np.ascontiguousarray(rgb8)
    if rgb8.ndim != 3 or rgb8.shape[2] != 3 or rgb8.dtype != np.uint8:
        raise ValueError("encode_png takes an (h, w, 3) uint8 image")
    h, w, _ = rgb8.shape
    if out is None:
        out = np.empty(png_bound(h, w), np.uint8)
    else:
        _check_png_out(out)
    n = C.c_size_t()
    _lib.check(_lib.lib().bs_encode_png(startree.handle, rgb8.ctypes.data, w, h, out.ctypes.data, out.size, C.byref(n)), "bs_encode_png")
    return memoryview(out)[:n.value]


def render_png(cfg: Config, startree: StarTree, out: np.ndarray = None) -> memoryview:
    """The whole of doRender (app/Main.hs:105-123) on the device: render, bloom if scene.bloomStrength /= 0, sRGB + 8-bit, the PNG
    file (`bs_render_png`).  Returns the file's bytes, ready for open(path, "wb").write(...)."""
    c = _bs_config(cfg)
    if out is None:
        out = np.empty(png_bound(c.height, c.width), np.uint8)
    else:
        _check_png_out(out)
    n = C.c_size_t()
    _lib.check(_lib.lib().bs_render_png(startree.handle, C.byref(c), float(cfg.scene.bloomStrength), int(cfg.scene.bloomDivider),
                                        out.ctypes.data, out.size, C.byref(n)), "bs_render_png")
    return memoryview(out)[:n.value]


def write_png(rgb8: np.ndarray, path: str, tree: StarTree = None) -> None:
    """Write an (h, w, 3) uint8 image as a PNG file.  With a StarTree the file is made on its GPU (encode_png); without one by
    zlib on the host (0.1-0.25 s per 1080p frame: the reference's way -- JuicyPixels via massiv-io)."""
    if tree is not None:
        with open(path, "wb") as f:
            f.write(encode_png(rgb8, tree))
        return
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def trace_rays(cfg, startree: StarTree, ys, xs) -> np.ndarray:
    """Test hook (libblackstar_gpu_debug.so): per-ray terminal records for traced-resolution pixels (ys, xs)."""
    c = _bs_config(cfg)
    yx = np.ascontiguousarray(np.stack([np.asarray(ys), np.asarray(xs)], axis=1).astype(np.int32))
    rec = np.zeros(len(yx), _lib.RECORD_DTYPE)
    _lib.check(_lib.debug_lib().bs_trace_rays(startree.handle, C.byref(c), yx.ctypes.data, len(yx), rec.ctypes.data), "bs_trace_rays")
    return rec


def srgb(x: np.ndarray) -> np.ndarray:
    """sRGB transfer (src/Raytracer.hs:23-27)."""
    a = 0.055
    x = np.asarray(x, np.float64)
    with np.errstate(invalid="ignore"):
        return np.where(x < 0.0031308, 12.92 * x, (1 + a) * np.power(x, 1.0 / 2.4) - a)


def to_word8(x: np.ndarray) -> np.ndarray:
    """massiv-io toWord8: clamp to [0,1], scale by 255, round half to even (recalled)."""
    return np.rint(255 * np.clip(x, 0.0, 1.0)).astype(np.uint8)


def write_img(img: np.ndarray, path: str, tree: StarTree = None) -> None:
    """writeImg (src/Raytracer.hs:29-32): sRGB transfer + 8-bit quantise (GPU, bs_srgb8), then the PNG file (on the same GPU when a
    StarTree is given)."""
    from .image_filters import srgb8
    write_png(srgb8(img, tree), path, tree)
