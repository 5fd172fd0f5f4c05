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
