#!/usr/bin/env python
"""Does it pay to let the trace kernel write the frame straight into page-locked HOST memory (zero copy over PCIe) instead of
rendering to HBM and copying?  bs_render(out=pinned) [kernel + D2H, two half-frame launches] against bs_render_device with the
pinned host buffer as its output."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import scenes  # noqa: E402

tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
L = _lib.lib()
for name, cfg in (("C3 1080p ss", scenes.DEFAULT_AA), ("C2 1080p", scenes.DEFAULT), ("C4 4K ss", scenes.with_res(scenes.LENSING_DISK, 3840, 2160))):
    H, W = cfg["height"], cfg["width"]
    buf = bs.alloc_image(tree, H, W)
    ref = bs.render(cfg, tree).copy()

    def staged():
        bs.render(cfg, tree, out=buf)

    def zero_copy():
        bs.render_device(cfg, tree, buf.ctypes.data, buf.size, 0)
        tree.stats()  # waits for the render

    for fn in (staged, zero_copy):
        buf[:] = -1
        fn()
        ok = np.array_equal(buf, ref)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        st = tree.stats()
        print(f"{name}: {fn.__name__:>9}: median {np.median(ts):.3f} ms  min {min(ts):.3f} ms  kernel {st['kernel_ms']:.3f} ms  pixels equal: {ok}", flush=True)
