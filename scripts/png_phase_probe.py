#!/usr/bin/env python
"""Where a workgroup of the PNG block kernel spends its time: shader-clock stamps after every phase (bs_debug_png_phases) on the C3
frame, as the median / 90th percentile over the blocks, in shader cycles and as a share of the block."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic

PHASES = ["init", "load", "runs", "tokenize", "ll shannon", "ll rank", "ll counts", "ll assign", "ll codes", "header zeros", "header tokens",
          "cl shannon", "cl rank", "cl counts", "cl assign", "cl codes", "bitcount", "plan", "emit", "crc", "crc groups", "write"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL)))
cfg = bs.Config.from_file(os.path.join(root, "scenes", "default-aa.yaml"))
rgb8 = bs.render_rgb8(cfg, tree)
h, w, _ = rgb8.shape
nb = -(-(h * (3 * w + 1)) // 8192)
clk = np.zeros((nb, len(PHASES) + 1), np.uint64)
for _ in range(3):
    _lib.check(_lib.debug_lib().bs_debug_png_phases(tree.handle, rgb8.ctypes.data, w, h, clk.ctypes.data, clk.size), "bs_debug_png_phases")
d = np.diff(clk.astype(np.int64), axis=1)
tot = d.sum(axis=1)
rows = [{"phase": p, "median_cycles": int(np.median(d[:, i])), "p90_cycles": int(np.percentile(d[:, i], 90)),
         "share": float(np.median(d[:, i]) / np.median(tot))} for i, p in enumerate(PHASES)]
print(json.dumps({"frame": f"{w}x{h}", "blocks": nb, "block_median_cycles": int(np.median(tot)), "block_p90_cycles": int(np.percentile(tot, 90)),
                  "kernel_span_cycles": int(clk.max() - clk.min()), "phases": rows}, indent=1))
