"""Synthetic star catalogue in the PPM on-disk layout (the real PPM file is not available offline).

The reference reads the PPM catalogue with src/StarMap.hs:45-58 (28 header bytes, then 28-byte
big-endian records `f64 ra, f64 dec, u8 spectral, skip 1, i16 mag*100, skip 8`).  This module writes
that layout so the product's own catalogue reader (bs_read_ppm) is exercised by tests and bench.

Recipe: SplitMix64, seed 0x5EEDB1AC57A2, u = (next()>>11) * 2^-53, four draws per
star: z = 2*u1-1, dec = asin z, ra = 2*pi*u2 (uniform on the sphere); mag100 = 1200 - floor(700*u3^3);
spectral = "OBAFGKM?"[floor(8*u4)].
"""
from __future__ import annotations

import numpy as np

SEED = 0x5EEDB1AC57A2
N_FULL = 470_000
N_SMALL = 2_000

_MASK = (1 << 64) - 1


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of SplitMix64 seeded with `seed` (uint64 array)."""
    gold = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + np.arange(1, n + 1, dtype=np.uint64) * gold
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def ppm_catalogue_bytes(n: int = N_FULL, seed: int = SEED) -> bytes:
    u = (splitmix64(seed, 4 * n) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    u = u.reshape(n, 4)
    z = 2 * u[:, 0] - 1
    dec = np.arcsin(z)
    ra = 2 * np.pi * u[:, 1]
    mag = (1200 - np.floor(700 * u[:, 2] ** 3)).astype(np.int16)
    sp = np.frombuffer(b"OBAFGKM?", np.uint8)[np.floor(8 * u[:, 3]).astype(np.int64)]
    rec = np.zeros(n, np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]))
    rec["ra"], rec["dec"], rec["sp"], rec["mag"] = ra, dec, sp, mag
    assert rec.dtype.itemsize == 28
    return bytes(28) + rec.tobytes()
