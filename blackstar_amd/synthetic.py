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


# ---- a NON-uniform sky ------------------------------------------------------------------------------------------------
# The real PPM catalogue (src/StarMap.hs:45-58 reads it; it is not available offline) is not uniform: open clusters and
# multiple stars put many stars inside one lookup radius (0.0015 rad, src/StarMap.hs:104), and the galactic plane is an
# order of magnitude denser than the poles.  starLookup folds over however many stars inRadius returns (:104,115), so the
# kernel's lookup must too -- its ">= 6 hits" branch and its behaviour on unbalanced grid cells are exercised with this sky.
CLUSTER_RADIUS = 0.001        # rad: every member lies within this of its cluster's centre (so a query there sees them all)
BAND_HALF_WIDTH = 0.035       # rad (2 degrees) either side of a tilted great circle
BAND_POLE = (0.3, -0.5, 0.81)  # un-normalised pole of that great circle


def _u01(seed: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return ((splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53).reshape(shape)


def _records(xyz: np.ndarray, mag: np.ndarray, sp: np.ndarray) -> bytes:
    """28-byte big-endian records (no header) for unit vectors xyz: ra = atan2(y, x) in [0, 2 pi), dec = asin z."""
    xyz = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
    rec = np.zeros(len(xyz), np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]))
    rec["ra"] = np.mod(np.arctan2(xyz[:, 1], xyz[:, 0]), 2 * np.pi)
    rec["dec"] = np.arcsin(np.clip(xyz[:, 2], -1.0, 1.0))
    rec["sp"], rec["mag"] = sp, mag.astype(np.int16)
    return rec.tobytes()


def _tangent_frame(c: np.ndarray):
    """Two unit vectors orthogonal to each row of c (unit vectors) and to each other."""
    ref = np.where(np.abs(c[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    e1 = np.cross(c, ref)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    return e1, np.cross(c, e1)


def cluster_records(centres: np.ndarray, sizes: np.ndarray, seed: int) -> bytes:
    """sizes[i] stars within CLUSTER_RADIUS of centres[i] (unit vectors): uniform in the disc, faint (mag100 1150..1449:
    2^-4 .. 2^-10 of full brightness, so that the sum of a whole cluster mostly stays below starLookup's `min 1` clamp)."""
    sizes = np.asarray(sizes, np.int64)
    tot = int(sizes.sum())
    c = np.repeat(np.asarray(centres, np.float64), sizes, axis=0)
    e1, e2 = _tangent_frame(c)
    u = _u01(seed, (tot, 4))
    rho, phi = CLUSTER_RADIUS * np.sqrt(u[:, 0:1]), 2 * np.pi * u[:, 1:2]
    xyz = c + rho * (np.cos(phi) * e1 + np.sin(phi) * e2)
    mag = 1150 + np.floor(300 * u[:, 2])
    sp = np.frombuffer(b"OBAFGKM?", np.uint8)[np.floor(8 * u[:, 3]).astype(np.int64)]
    return _records(xyz, mag, sp)


def clustered_catalogue_bytes(n_uniform: int = N_FULL, seed: int = SEED, n_clusters: int = 3000, band_factor: float = 10.0) -> bytes:
    """PPM-layout catalogue = the uniform recipe above (same first n_uniform stars as ppm_catalogue_bytes)
    + n_clusters clusters of 6..40 stars inside CLUSTER_RADIUS at uniformly drawn centres
    + a band of +-BAND_HALF_WIDTH about a tilted great circle filled up to band_factor x the mean density."""
    base = ppm_catalogue_bytes(n_uniform, seed)
    u = _u01(seed + 0x100, (n_clusters, 3))
    z = 2 * u[:, 0] - 1
    ra = 2 * np.pi * u[:, 1]
    s = np.sqrt(np.maximum(0.0, 1 - z * z))
    centres = np.stack([s * np.cos(ra), s * np.sin(ra), z], axis=1)
    sizes = 6 + np.floor(35 * u[:, 2]).astype(np.int64)
    clusters = cluster_records(centres, sizes, seed + 0x200)
    area = 2 * np.pi * 2 * np.sin(BAND_HALF_WIDTH)
    n_band = int(round((band_factor - 1) * n_uniform / (4 * np.pi) * area))
    ub = _u01(seed + 0x300, (n_band, 4))
    pole = np.array(BAND_POLE) / np.linalg.norm(BAND_POLE)
    e1, e2 = _tangent_frame(pole[None, :])
    sinb = np.sin(BAND_HALF_WIDTH) * (2 * ub[:, 0:1] - 1)  # uniform in sin(latitude) = uniform in area
    lam = 2 * np.pi * ub[:, 1:2]
    xyz = np.sqrt(1 - sinb * sinb) * (np.cos(lam) * e1 + np.sin(lam) * e2) + sinb * pole[None, :]
    mag = 1200 - np.floor(700 * ub[:, 2] ** 3)
    sp = np.frombuffer(b"OBAFGKM?", np.uint8)[np.floor(8 * ub[:, 3]).astype(np.int64)]
    return base + clusters + _records(xyz, mag, sp)


def catalogue_bytes(kind: str = "synthetic") -> bytes:
    """bench.py --catalogue: 'synthetic' (the uniform BASELINE sky), 'clustered', or a path to a real PPM catalogue file."""
    if kind == "synthetic":
        return ppm_catalogue_bytes()
    if kind == "clustered":
        return clustered_catalogue_bytes()
    with open(kind, "rb") as f:
        return f.read()
