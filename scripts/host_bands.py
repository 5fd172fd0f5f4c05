#!/usr/bin/env python
"""PCIe-inclusive frame time of bs_render (host buffer out) against the number of sub-bands the frame is delivered in."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import synthetic
from oracle import scenes
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
for name, cfg in (("C3 default-aa 1080p 4xSS", scenes.DEFAULT_AA), ("C2 default 1080p", scenes.DEFAULT)):
    for bands in (1, 2, 4, 8):
        os.environ["BLACKSTAR_HOST_BANDS"] = str(bands)
        t = bs.StarTree(stars if "aa" in name else None)
        for kind in ("fresh pageable", "reused pageable", "pinned"):
          buf = None if kind == "fresh pageable" else (np.empty((cfg["height"], cfg["width"], 3)) if kind == "reused pageable" else bs.alloc_image(t, cfg["height"], cfg["width"]))
          for _ in range(3):
            bs.render(cfg, t, out=buf)
          w = []
          for _ in range(12):
            t0 = time.perf_counter(); bs.render(cfg, t, out=buf); w.append((time.perf_counter() - t0) * 1e3)
          st = t.stats()
          print(f"{name}: bands={bands} {kind}: wall_ms median {np.median(w):.3f} min {min(w):.3f} kernel_ms(last, first launch start to last launch end) {st['kernel_ms']:.3f}")
          del buf
        t.close()
