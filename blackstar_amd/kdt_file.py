"""Best-effort reader / writer for the reference's `stars.kdt` file.

`stars.kdt` is `S.encode` of `KdMap Double (V3 Double) (Int, Char)` (src/StarMap.hs:30-41, 87-88): cereal's Generic
encoding of kdt's records, with the two function fields replaced by one dummy byte each (:35-41).  Neither kdt nor
cereal is vendored in the reference and no real file exists offline, so this layout is RECALLED, NOT VERIFIED against
a real file -- treat a decode failure on a real `stars.kdt` as a bug in this module, not in the file:

    KdMap    = u8 0 (pointAsList dummy) . u8 0 (distSqr dummy) . TreeNode . i64be size          -- record field order
    TreeNode = u8 0 . TreeNode(left) . V3 (3 x f64be) . i64be mag . utf8 spectral-char . f64be axisValue . TreeNode(right)
             | u8 1                                                                              -- Empty
Only the (position, (mag, spectral)) pairs matter downstream: the GPU builds its own direction grid from them
(bs_create), so the tree shape in the file is not used.
"""
from __future__ import annotations

import struct
from typing import List, Tuple

import numpy as np

from ._lib import STAR_DTYPE

_COLOURS = {"O": (0.631, 0.39), "B": (0.628, 0.33), "A": (0.622, 0.21), "F": (0.650, 0.03), "G": (0.089, 0.09),
            "K": (0.094, 0.29), "M": (0.094, 0.56)}  # starColor, src/StarMap.hs:64-72


class KdtDecodeError(ValueError):
    pass


def read_kdt(data: bytes) -> np.ndarray:
    """Decode a `stars.kdt` image into stars (STAR_DTYPE, starColor' applied), in the tree's in-order."""
    if len(data) < 3:
        raise KdtDecodeError("Error decoding star tree: too few bytes")
    pos = 2  # the two function-field dummies
    out: List[Tuple[float, float, float, int, str]] = []
    # iterative in-order walk of the prefix-coded tree: stack of pending "after-left" continuations
    stack: List[int] = []
    state = "node"
    try:
        while True:
            if state == "node":
                tag = data[pos]; pos += 1
                if tag == 0:
                    stack.append(0)  # descend into left first
                    continue
                if tag != 1:
                    raise KdtDecodeError(f"Error decoding star tree: bad constructor tag {tag} at byte {pos - 1}")
                state = "up"
            else:  # finished a subtree
                if not stack:
                    break
                phase = stack.pop()
                if phase == 0:  # left done: read this node's payload, then the right subtree
                    x, y, z = struct.unpack_from(">ddd", data, pos); pos += 24
                    (mag,) = struct.unpack_from(">q", data, pos); pos += 8
                    b0 = data[pos]
                    n = 1 if b0 < 0x80 else 2 if b0 < 0xE0 else 3 if b0 < 0xF0 else 4
                    ch = data[pos:pos + n].decode("utf-8"); pos += n
                    pos += 8  # axisValue
                    out.append((x, y, z, mag, ch))
                    stack.append(1)
                    state = "node"
                else:
                    state = "up"
        (size,) = struct.unpack_from(">q", data, pos); pos += 8
    except (IndexError, struct.error, UnicodeDecodeError) as e:
        raise KdtDecodeError(f"Error decoding star tree: truncated or malformed ({e})") from e
    if size != len(out):
        raise KdtDecodeError(f"Error decoding star tree: size field {size} != {len(out)} nodes")
    stars = np.zeros(len(out), STAR_DTYPE)
    for i, (x, y, z, mag, ch) in enumerate(out):
        hue, sat = _COLOURS.get(ch, (0.0, 0.0))
        stars[i] = (x, y, z, hue, sat, mag, 0)
    return stars


def write_kdt(positions: np.ndarray, mags, spectral: str) -> bytes:
    """Encode stars as a balanced k-d tree in the layout above (what `generate-tree` writes; for round-trip tests)."""
    pts = np.asarray(positions, np.float64)
    order = list(range(len(pts)))
    chunks: List[bytes] = [b"\x00\x00"]

    def emit(idx: List[int], axis: int) -> None:
        if not idx:
            chunks.append(b"\x01")
            return
        idx.sort(key=lambda i: pts[i, axis])
        m = len(idx) // 2
        i = idx[m]
        chunks.append(b"\x00")
        emit(idx[:m], (axis + 1) % 3)
        chunks.append(struct.pack(">dddq", pts[i, 0], pts[i, 1], pts[i, 2], int(mags[i])) + spectral[i].encode("utf-8") +
                      struct.pack(">d", pts[i, axis]))
        emit(idx[m + 1:], (axis + 1) % 3)

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    emit(order, 0)
    chunks.append(struct.pack(">q", len(pts)))
    return b"".join(chunks)
