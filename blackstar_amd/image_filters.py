"""Host-side mirror of the reference's ImageFilters module: `bloom` and `supersample` (src/ImageFilters.hs:5).

Both run on the GPU through the C ABI (`bs_bloom`, `bs_supersample`; inside `render` the supersample is fused into
the trace kernel's epilogue); this module contains no pixel arithmetic.  The reference's functions take no context, the GPU needs one: pass
the `StarTree` you render with, or let a lazily created context on device 0 be used.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .star_map import StarTree

_default_ctx: Optional[StarTree] = None


def _ctx(tree: Optional[StarTree]) -> StarTree:
    global _default_ctx
    if tree is not None:
        return tree
    if _default_ctx is None:
        _default_ctx = StarTree(None, device=0)
    return _default_ctx


def bloom(strength: float, divider: int, img: np.ndarray, tree: Optional[StarTree] = None) -> np.ndarray:
    """bloom :: Double -> Int -> Image U RGB Double -> IO (Image U RGB Double)   (src/ImageFilters.hs:80-86)."""
    img = np.ascontiguousarray(img, np.float64)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected an (h, w, 3) RGB image")
    out = np.empty_like(img)
    h, w, _ = img.shape
    _lib.check(_lib.lib().bs_bloom(_ctx(tree).handle, img.ctypes.data, out.ctypes.data, w, h, float(strength), int(divider)), "bs_bloom")
    return out


def srgb8(img: np.ndarray, tree: Optional[StarTree] = None) -> np.ndarray:
    """The pixel map of writeImg (src/Raytracer.hs:23-32): sRGB transfer + 8-bit quantise, on the GPU."""
    img = np.ascontiguousarray(img, np.float64)
    out = np.empty(img.shape, np.uint8)
    _lib.check(_lib.lib().bs_srgb8(_ctx(tree).handle, img.ctypes.data, out.ctypes.data, img.size), "bs_srgb8")
    return out


def supersample(img: np.ndarray, tree: Optional[StarTree] = None) -> np.ndarray:
    """supersample :: Image U RGB Double -> Image U RGB Double   (src/ImageFilters.hs:88-97): 2x2 mean in the
    reference's summation order, on the GPU (`bs_supersample`).  `render` does not call this: there the reduction is
    fused into the trace kernel's epilogue."""
    img = np.ascontiguousarray(img, np.float64)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected an (h, w, 3) RGB image")
    h2, w2, _ = img.shape
    out = np.empty((h2 // 2, w2 // 2, 3), np.float64)
    _lib.check(_lib.lib().bs_supersample(_ctx(tree).handle, img.ctypes.data, out.ctypes.data, w2, h2), "bs_supersample")
    return out
