#!/usr/bin/env python
"""A/B of the bloom sweep paths on the box: bs_bloom_device (three passes of H + V sweep + combine) timed with HIP events
per path (BLACKSTAR_BLOOM_PATH = dma | lds | direct), bit-exactness against the oracle checked on the way."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib  # noqa: E402
from oracle import c_oracle  # noqa: E402

tree = bs.StarTree(None)
L = _lib.lib()
rng = np.random.default_rng(0)
for (h, w, div) in ((1080, 1920, 25), (2160, 3840, 25), (720, 1280, 25), (1080, 1920, 10)):
    a = rng.uniform(0, 1.5, (h, w, 3))
    img = torch.from_numpy(a).cuda()
    out = torch.empty_like(img)
    ref = c_oracle.bloom(0.15, div, a) if h * w <= 1920 * 1080 else None
    for path in ("dma", "lds", "direct"):
        os.environ["BLACKSTAR_BLOOM_PATH"] = path
        fn = lambda: _lib.check(L.bs_bloom_device(tree.handle, img.data_ptr(), out.data_ptr(), w, h, 0.15, div, None), "bloom")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for e0, e1 in ev:
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        ok = "n/a" if ref is None else bool(np.array_equal(out.cpu().numpy(), ref))
        gb = 7 * 2 * a.nbytes / 1e9 - a.nbytes / 1e9  # six sweeps read+write, combine reads 2 writes 1
        print(f"bloom {w}x{h} r={w // div} path={path}: median {ts[len(ts) // 2] * 1e3:.1f} us, min {ts[0] * 1e3:.1f} us "
              f"({gb / (ts[len(ts) // 2] * 1e-3) / 1e3:.2f} TB/s algorithmic)  bit-exact: {ok}", flush=True)
