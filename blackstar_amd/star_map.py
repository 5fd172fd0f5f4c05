"""Star catalogue / star tree -- host-side mirror of the reference's StarMap module (src/StarMap.hs).

Reference -> here:
  readMapFromFile  (:77-80)   -> read_map_from_file   (catalogue parse runs in the C library: bs_read_ppm)
  buildStarTree    (:90-91)   -> build_star_tree      (direction grid built + uploaded by bs_create)
  treeToByteString (:87-88)   -> tree_to_byte_string  (own flat format; the `.kdt` cereal layout is out of scope)
  readTreeFromFile (:82-85)   -> read_tree_from_file
  starLookup       (:93-115)  -> star_lookup          (device function, batched through bs_star_lookup)
A `StarTree` owns one `bs_ctx` (one HIP device); like the reference's tree it is built once and reused
for every scene (app/Main.hs:46-49).
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Optional

import numpy as np

from . import _lib
from ._lib import STAR_DTYPE, BlackstarError

_MAGIC = b"BSKD1\x00\x00\x00"


def read_map(data: bytes) -> np.ndarray:
    """readMap + starColor' (StarMap.hs:45-62): PPM catalogue bytes -> stars (STAR_DTYPE)."""
    L = _lib.lib()
    buf = np.frombuffer(data, np.uint8)
    n = L.bs_read_ppm(buf.ctypes.data if len(buf) else None, len(data), None, 0)
    if n < 0:
        raise BlackstarError("too few bytes")  # cereal: skip 28 on a short input fails
    out = np.zeros(n, STAR_DTYPE)
    L.bs_read_ppm(buf.ctypes.data, len(data), out.ctypes.data, n)
    return out


def read_map_from_file(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return read_map(f.read())


class StarTree:
    """The `StarTree` argument of render: the star set resident on one GPU as a cube-map direction grid (csrc/star_index.cpp)."""

    def __init__(self, stars: Optional[np.ndarray], device: int = 0):
        stars = np.zeros(0, STAR_DTYPE) if stars is None else np.ascontiguousarray(stars, dtype=STAR_DTYPE)
        self.stars = stars
        self.device = device
        L = _lib.lib()
        self._h = L.bs_create(device, stars.ctypes.data if len(stars) else None, len(stars))
        if not self._h:
            raise BlackstarError(f"bs_create failed: {_lib.last_error()}")

    def __len__(self) -> int:
        return len(self.stars)

    @property
    def handle(self):
        if not self._h:
            raise BlackstarError("StarTree is closed")
        return self._h

    def set_mode(self, mode: int) -> None:
        _lib.check(_lib.lib().bs_set_mode(self.handle, mode), "bs_set_mode")

    def get_mode(self) -> int:
        return int(_lib.lib().bs_get_mode(self.handle))

    def set_max_steps(self, n: int) -> None:
        _lib.check(_lib.lib().bs_set_max_steps(self.handle, n), "bs_set_max_steps")

    def stats(self) -> dict:
        st = _lib.BsStats()
        _lib.check(_lib.lib().bs_stats(self.handle, C.byref(st)), "bs_stats")
        return {k: getattr(st, k) for k, _ in _lib.BsStats._fields_}

    def numa_node(self) -> int:
        """`bs_numa_node`: the NUMA node of the host this tree's GPU hangs off (-1: the host does not say)."""
        return int(_lib.lib().bs_numa_node(self.handle))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().bs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_star_tree(stars: np.ndarray, device: int = 0) -> StarTree:
    return StarTree(stars, device)


def tree_to_byte_string(stars: np.ndarray) -> bytes:
    stars = np.ascontiguousarray(stars, dtype=STAR_DTYPE)
    return _MAGIC + struct.pack("<Q", len(stars)) + stars.tobytes()


def read_tree_from_file(path: str, device: int = 0) -> StarTree:
    """readTreeFromFile (src/StarMap.hs:82-85).  Accepts this repository's flat `.bskd` file and, best effort, the
    reference's cereal-encoded `stars.kdt` (blackstar_amd/kdt_file.py: layout recalled, unverified offline)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\x00\x00" and data[:8] != _MAGIC:
        from .kdt_file import KdtDecodeError, read_kdt
        try:
            return StarTree(read_kdt(data), device)
        except KdtDecodeError as e:
            raise BlackstarError(str(e)) from e
    if data[:8] != _MAGIC or len(data) < 16:
        raise BlackstarError("Error decoding star tree: bad magic")
    (n,) = struct.unpack("<Q", data[8:16])
    if len(data) != 16 + n * STAR_DTYPE.itemsize:
        raise BlackstarError("Error decoding star tree: truncated")
    return StarTree(np.frombuffer(data, STAR_DTYPE, count=n, offset=16).copy(), device)


def star_lookup(tree: StarTree, intensity: float, saturation: float, vel, return_hits: bool = False):
    """starLookup starmap intensity saturation vel -- vel may be (3,) or (n,3)."""
    v = np.ascontiguousarray(np.atleast_2d(np.asarray(vel, np.float64)))
    n = v.shape[0]
    rgb = np.zeros((n, 3))
    hits = np.zeros(n, np.int32)
    _lib.check(_lib.lib().bs_star_lookup(tree.handle, intensity, saturation, v.ctypes.data, n, rgb.ctypes.data, hits.ctypes.data), "bs_star_lookup")
    if np.ndim(vel) == 1:
        return (rgb[0], int(hits[0])) if return_hits else rgb[0]
    return (rgb, hits) if return_hits else rgb
