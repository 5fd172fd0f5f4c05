#!/usr/bin/env python
"""Large randomised check that the device f64 sqrt / divide used by STRICT mode are correctly rounded: hipcc's lowering
(bare=0) and the scaling-free sequences inside the RK4 RHS (bare=1), 2^27 operand pairs each in the ranges the RHS sees
(r^2 in [1e-3, 1e5]; numerators 1.5 h^2 in [1e-6, 1e4], denominators r^5) plus a wide log-uniform range."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib  # noqa: E402

t = bs.StarTree(None)
L = _lib.lib()
rng = np.random.default_rng(2718)
res = {}
chunk = 1 << 24
for name, gen in (("rhs_range", lambda n: (np.exp(rng.uniform(np.log(1e-3), np.log(1e5), n)), np.exp(rng.uniform(np.log(1e-6), np.log(1e4), n)))),
                  ("wide_range", lambda n: (np.exp(rng.uniform(-300, 300, n)), np.exp(rng.uniform(-300, 300, n))))):
    bad = {0: [0, 0], 1: [0, 0]}
    total = 0
    for _ in range(8):
        a, num = gen(chunk)
        den = a ** 2.5 if name == "rhs_range" else np.exp(rng.uniform(-300, 300, chunk))
        s = np.empty(chunk); d = np.empty(chunk)
        for bare in (0, 1):
            _lib.check(_lib.debug_lib().bs_debug_sqrt_div(t.handle, a.ctypes.data, den.ctypes.data, chunk, s.ctypes.data, d.ctypes.data, bare), "sqrt_div")
            bad[bare][0] += int((s != np.sqrt(a)).sum())
            bad[bare][1] += int((d != a / den).sum())
        total += chunk
    res[name] = {"operands": total, "hipcc_lowering_mismatches_sqrt_div": bad[0], "bare_sequence_mismatches_sqrt_div": bad[1]}
print(json.dumps(res))
