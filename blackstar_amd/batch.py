"""Batch mode: a list of scenes rendered with the same star tree (app/Main.hs:68-77), through `bs_render_batch`."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np

from . import _lib
from .config_file import Config
from .star_map import StarTree


def render_batch(cfgs: Sequence, trees: Sequence[StarTree], outs: Sequence[np.ndarray] = None) -> List[np.ndarray]:
    """Render cfgs[i] on trees[i % len(trees)] (one StarTree per GPU); per tree two frames are in flight: frame k's
    device-to-host copy and its end-of-frame tail overlap frame k+1's kernel.  Returns the (h, w, 3) float64 images in order;
    `outs` supplies the buffers to fill (reused or page-locked ones from alloc_image avoid first-touch page faults)."""
    if not trees:
        raise ValueError("need at least one StarTree")
    cs = [_lib.make_config(c.to_bs_config() if isinstance(c, Config) else c) for c in cfgs]
    n = len(cs)
    if outs is None:
        outs = [np.empty((c.height, c.width, 3), np.float64) for c in cs]
    else:
        outs = list(outs)
        if len(outs) != n or any(o.shape != (c.height, c.width, 3) or o.dtype != np.float64 or not o.flags["C_CONTIGUOUS"] for o, c in zip(outs, cs)):
            raise ValueError("outs must hold one C-contiguous float64 (h, w, 3) array per frame")
    if n == 0:
        return outs
    arr = (_lib.BsConfig * n)(*cs)
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ctxs = (C.c_void_p * len(trees))(*[t.handle for t in trees])
    _lib.check(_lib.lib().bs_render_batch(ctxs, len(trees), arr, n, ptrs), "bs_render_batch")
    return outs


def render_rgb8_batch(cfgs: Sequence[Config], trees: Sequence[StarTree], outs: Sequence[np.ndarray] = None) -> List[np.ndarray]:
    """doRender for a list of scenes (app/Main.hs:68-77 -> :105-123) with everything up to the PNG encoder on the device
    (`bs_render_rgb8_batch`): cfgs[i] -- a Config, whose scene carries its own bloomStrength / bloomDivider -- is rendered, bloomed
    and mapped to RGB8 on trees[i % len(trees)], two frames in flight per tree.  Returns the (h, w, 3) uint8 images in order,
    byte-identical to render_rgb8 frame by frame; `outs`: buffers to fill (page-locked ones from alloc_image are written in place)."""
    if not trees:
        raise ValueError("need at least one StarTree")
    if any(not isinstance(c, Config) for c in cfgs):
        raise TypeError("render_rgb8_batch takes Config objects (the scene's bloom parameters are part of the frame)")
    cs = [_lib.make_config(c.to_bs_config()) for c in cfgs]
    n = len(cs)
    if outs is None:
        outs = [np.empty((c.height, c.width, 3), np.uint8) for c in cs]
    else:
        outs = list(outs)
        if len(outs) != n or any(o.shape != (c.height, c.width, 3) or o.dtype != np.uint8 or not o.flags["C_CONTIGUOUS"] for o, c in zip(outs, cs)):
            raise ValueError("outs must hold one C-contiguous uint8 (h, w, 3) array per frame")
    if n == 0:
        return outs
    arr = (_lib.BsConfig * n)(*cs)
    strengths = (C.c_double * n)(*[float(c.scene.bloomStrength) for c in cfgs])
    dividers = (C.c_int * n)(*[int(c.scene.bloomDivider) for c in cfgs])
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ctxs = (C.c_void_p * len(trees))(*[t.handle for t in trees])
    _lib.check(_lib.lib().bs_render_rgb8_batch(ctxs, len(trees), arr, n, strengths, dividers, ptrs), "bs_render_rgb8_batch")
    return outs


def render_png_batch(cfgs: Sequence[Config], trees: Sequence[StarTree], outs: Sequence[np.ndarray] = None) -> List[memoryview]:
    """render_rgb8_batch with finished PNG files instead of pixels (`bs_render_png_batch`): the whole of doRender (app/Main.hs:105-123)
    but the write(2) on the device, two frames in flight per tree.  Returns the files' bytes in order (views of `outs` -- flat uint8
    buffers of at least png_bound(h, w) bytes each, e.g. from raytracer.alloc_png, which the encoder fills itself -- if given)."""
    from .raytracer import png_bound
    if not trees:
        raise ValueError("need at least one StarTree")
    if any(not isinstance(c, Config) for c in cfgs):
        raise TypeError("render_png_batch takes Config objects (the scene's bloom parameters are part of the frame)")
    cs = [_lib.make_config(c.to_bs_config()) for c in cfgs]
    n = len(cs)
    if outs is None:
        outs = [np.empty(png_bound(c.height, c.width), np.uint8) for c in cs]
    else:
        outs = list(outs)
        if len(outs) != n or any(o.ndim != 1 or o.dtype != np.uint8 or not o.flags["C_CONTIGUOUS"] for o in outs):
            raise ValueError("outs must hold one flat C-contiguous uint8 buffer per frame")
    if n == 0:
        return []
    arr = (_lib.BsConfig * n)(*cs)
    strengths = (C.c_double * n)(*[float(c.scene.bloomStrength) for c in cfgs])
    dividers = (C.c_int * n)(*[int(c.scene.bloomDivider) for c in cfgs])
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * n)(*[o.size for o in outs])
    sizes = (C.c_size_t * n)()
    ctxs = (C.c_void_p * len(trees))(*[t.handle for t in trees])
    _lib.check(_lib.lib().bs_render_png_batch(ctxs, len(trees), arr, n, strengths, dividers, ptrs, caps, sizes), "bs_render_png_batch")
    return [memoryview(o)[:sizes[i]] for i, o in enumerate(outs)]


def render_png_files(cfgs: Sequence[Config], trees: Sequence[StarTree], paths: Sequence[str], pipe: int = 16) -> None:
    """The reference's batch loop to the very end, in the library (`bs_render_png_files`): cfgs[i] rendered, bloomed, mapped to sRGB8 and
    PNG-encoded on trees[i % len(trees)] and written to paths[i] -- per tree one rolling pipeline of frames in flight, a ring of `pipe`
    page-locked file buffers and a native writer thread of its own (on the GPU's NUMA node); no tree waits for another.  Raises
    BlackstarError (BS_EIO) if a file cannot be written.  `files_stats(tree)` tells what each tree's writer did."""
    if not trees:
        raise ValueError("need at least one StarTree")
    if any(not isinstance(c, Config) for c in cfgs):
        raise TypeError("render_png_files takes Config objects (the scene's bloom parameters are part of the frame)")
    n = len(cfgs)
    if len(paths) != n:
        raise ValueError("one path per frame")
    if n == 0:
        return
    arr = (_lib.BsConfig * n)(*[_lib.make_config(c.to_bs_config()) for c in cfgs])
    strengths = (C.c_double * n)(*[float(c.scene.bloomStrength) for c in cfgs])
    dividers = (C.c_int * n)(*[int(c.scene.bloomDivider) for c in cfgs])
    cpaths = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    ctxs = (C.c_void_p * len(trees))(*[t.handle for t in trees])
    _lib.check(_lib.lib().bs_render_png_files(ctxs, len(trees), arr, n, strengths, dividers, cpaths, int(pipe)), "bs_render_png_files")


def files_stats(tree: StarTree) -> dict:
    """`bs_files_stats`: the host side of this tree's share of the last render_png_files call (files, bytes, wall_ms, writer_busy_ms,
    buffer_wait_ms, ring, writer_threads, numa_node_gpu, numa_node_buffers, threads_bound) plus writer_busy_frac = busy / wall."""
    st = _lib.BsFilesStats()
    _lib.check(_lib.lib().bs_files_stats(tree.handle, C.byref(st)), "bs_files_stats")
    d = {k: getattr(st, k) for k, _ in _lib.BsFilesStats._fields_ if k != "_pad"}
    d["writer_busy_frac"] = st.writer_busy_ms / st.wall_ms if st.wall_ms > 0 else 0.0
    return d


def render_split(cfg, trees: Sequence[StarTree], out: np.ndarray = None) -> np.ndarray:
    """ONE frame over several StarTrees (one per GPU): tree k renders the k-th contiguous band of rows (`bs_render_split`).
    Bit-identical to render(cfg, trees[0]).  `out`: the (h, w, 3) float64 buffer to fill; a page-locked one (alloc_image) is
    written by every GPU's kernel directly, each into its own band."""
    if not trees:
        raise ValueError("need at least one StarTree")
    c = _lib.make_config(cfg.to_bs_config() if isinstance(cfg, Config) else cfg)
    if out is None:
        out = np.empty((c.height, c.width, 3), np.float64)
    elif out.shape != (c.height, c.width, 3) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous float64 array of shape {(c.height, c.width, 3)}")
    ctxs = (C.c_void_p * len(trees))(*[t.handle for t in trees])
    _lib.check(_lib.lib().bs_render_split(ctxs, len(trees), C.byref(c), out.ctypes.data, out.size), "bs_render_split")
    return out
