"""50-digit mpmath evaluation of the SAME discrete RK4 map (src/Raytracer.hs:113-134) and guards (:91-102).

TEST INFRASTRUCTURE ONLY.  Pins the FP64 restatements independently of any compiler: with the step size and
initial state taken as exact binary64 values, the discrete map is evaluated in 50-digit arithmetic; FP64
results must agree on step count / fate and to ~1e-11 relative on the terminal state.
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 50


def trace(vel, pos, h, safe, max_steps=100000):
    """vel, pos: binary64 initial state (generateRay output).  Returns (steps, fate, vel, pos) as mpf."""
    v = [mp.mpf(float(x)) for x in vel]
    p = [mp.mpf(float(x)) for x in pos]
    h = mp.mpf(float(h))
    safe = mp.mpf(float(safe))
    cx = [p[1] * v[2] - p[2] * v[1], p[2] * v[0] - p[0] * v[2], p[0] * v[1] - p[1] * v[0]]
    h2 = cx[0] ** 2 + cx[1] ** 2 + cx[2] ** 2

    def f(vv, pp):
        n = mp.sqrt(pp[0] ** 2 + pp[1] ** 2 + pp[2] ** 2)
        c = -mp.mpf(3) / 2 * h2 / n ** 5
        return [c * x for x in pp], list(vv)

    steps = 0
    while steps < max_steps:
        steps += 1
        r2 = p[0] ** 2 + p[1] ** 2 + p[2] ** 2
        if r2 < 1:
            return steps, 0, v, p
        if r2 > safe:
            return steps, 1, v, p
        k1v, k1p = f(v, p)
        k2v, k2p = f([a + b * h / 2 for a, b in zip(v, k1v)], [a + b * h / 2 for a, b in zip(p, k1p)])
        k3v, k3p = f([a + b * h / 2 for a, b in zip(v, k2v)], [a + b * h / 2 for a, b in zip(p, k2p)])
        k4v, k4p = f([a + b * h for a, b in zip(v, k3v)], [a + b * h for a, b in zip(p, k3p)])
        v = [a + (b + 2 * c + 2 * d + e) * h / 6 for a, b, c, d, e in zip(v, k1v, k2v, k3v, k4v)]
        p = [a + (b + 2 * c + 2 * d + e) * h / 6 for a, b, c, d, e in zip(p, k1p, k2p, k3p, k4p)]
    return steps, 2, v, p


def star_lookup(stars, intensity, saturation, vel):
    """starLookup (src/StarMap.hs:93-115) in 50-digit arithmetic for ONE direction: stars is an (n, 6) float array x, y, z, hue, sat,
    mag (binary64 values taken as exact).  Returns (rgb as mpf, number of stars within the radius, smallest |d^2 - r^2| / r^2 over
    all stars = how close the hit SET is to flipping).  Pins the colour arithmetic of lookups that sum many stars independently
    of libm, of FMA contraction and of the summation order (which only matters at the 1e-16 level this evaluation sits far below)."""
    w = mp.mpf("0.0005")
    r2 = (3 * w) ** 2
    v = [mp.mpf(float(x)) for x in vel]
    l = v[0] ** 2 + v[1] ** 2 + v[2] ** 2
    n = v if (abs(l) <= mp.mpf("1e-12") or abs(1 - l) <= mp.mpf("1e-12")) else [x / mp.sqrt(l) for x in v]
    a = mp.log(2) / 50
    acc = [mp.mpf(0)] * 3
    hits, margin = 0, mp.inf
    pi = mp.pi
    for x, y, z, hue, sat, mag in stars:
        d2 = (mp.mpf(float(x)) - n[0]) ** 2 + (mp.mpf(float(y)) - n[1]) ** 2 + (mp.mpf(float(z)) - n[2]) ** 2
        margin = min(margin, abs(d2 - r2) / r2)
        if d2 > r2:
            continue
        hits += 1
        val = mp.mpf(float(intensity)) * min(mp.mpf(1), mp.exp(a * (950 - mp.mpf(float(mag))) - d2 / (2 * w ** 2)))
        s = mp.mpf(float(saturation)) * mp.mpf(float(sat))
        h = mp.mpf(float(hue)) * 2 * pi
        is_ = val * s
        second = val - is_
        if h < 2 * pi / 3:
            r = val + is_ * mp.cos(h) / mp.cos(pi / 3 - h); b = second; g = val + 2 * is_ + b - r
        elif h < 4 * pi / 3:
            g = val + is_ * mp.cos(h - 2 * pi / 3) / mp.cos(h + pi); r = second; b = val + 2 * is_ + r - g
        else:
            b = val + is_ * mp.cos(h - 4 * pi / 3) / mp.cos(2 * pi - pi / 3 - h); g = second; r = val + 2 * is_ + g - b
        acc = [acc[0] + r, acc[1] + g, acc[2] + b]
    return [min(mp.mpf(1), c) for c in acc], hits, margin
