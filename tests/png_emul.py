"""TEST INFRASTRUCTURE: ctypes wrapper of tests/cpp/png_emul.cpp -- the device PNG encoder's phase program (blackstar_amd/csrc/
png_block.h) run lane by lane on the host -- plus a strict reader of what it (and the GPU) produce."""
import ctypes as C
import io
import os
import struct
import subprocess
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "png_emul.cpp")
HDR = os.path.join(os.path.dirname(HERE), "blackstar_amd", "csrc", "png_block.h")
SO = os.path.join(HERE, "cpp", "_build", "libpng_emul.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", SRC, "-o", SO])
        L = C.CDLL(SO)
        L.png_emul_bound.restype = C.c_uint64
        L.png_emul_bound.argtypes = [C.c_int, C.c_int]
        L.png_emul_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int, C.c_void_p, C.c_void_p]
        L.png_emul_crc.restype = C.c_uint32
        L.png_emul_crc.argtypes = [C.c_void_p, C.c_uint32]
        L.png_emul_crc_combine.restype = C.c_uint32
        L.png_emul_crc_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def bound(h, w):
    return int(lib().png_emul_bound(w, h))


def encode(img, order=0):
    """(h, w, 3) uint8 -> (file bytes, [blocks, blocks stored], filter type per row); order: 0 lanes forwards, 1 backwards, >= 2 shuffled."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    cap = bound(h, w)
    out = np.empty(cap, np.uint8)
    n = C.c_uint64()
    stats = np.zeros(2, np.uint32)
    filt = np.zeros(h, np.uint8)
    rc = lib().png_emul_encode(img.ctypes.data, w, h, out.ctypes.data, cap, C.byref(n), order, filt.ctypes.data, stats.ctypes.data)
    assert rc == 0
    return bytes(out[:n.value]), stats, filt


def check_file(data: bytes, img: np.ndarray) -> dict:
    """Everything a strict reader checks, by hand: signature, chunk order, every chunk's CRC-32 (zlib.crc32), the zlib stream (inflate
    checks Adler-32 and that the final block ends the stream), its length, the filter bytes; then Pillow (libpng-grade decoder) must
    return exactly `img`.  Returns a few facts about the file."""
    from PIL import Image
    data = bytes(data)
    h, w, _ = img.shape
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, tags, idat = 8, [], []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF, (tag, pos)
        tags.append(tag)
        if tag == b"IDAT":
            idat.append(body)
        if tag == b"IHDR":
            assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 2, 0, 0, 0)
        pos += 12 + n
    assert pos == len(data)
    assert tags[0] == b"IHDR" and tags[-1] == b"IEND" and set(tags[1:-1]) == {b"IDAT"}
    d = zlib.decompressobj()
    raw = d.decompress(b"".join(idat))
    assert d.eof and d.unused_data == b"" and len(raw) == h * (3 * w + 1)
    filt = np.frombuffer(raw, np.uint8)[::3 * w + 1]
    assert filt.max() <= 4
    dec = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    assert dec.shape == img.shape and np.array_equal(dec, img)
    return {"bytes": len(data), "idat_chunks": len(idat), "filters": np.bincount(filt, minlength=5)}
