#!/usr/bin/env python
"""Tiny driver for rocprofv3: renders the BASELINE C3 frame a few times through the C ABI (no torch)."""
import argparse
import os
import sys

os.environ.setdefault("BLACKSTAR_HOST_BANDS", "1")  # ONE launch per frame (what bs_render_device / bench.py do), so per-launch counters are per frame
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fast")
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--scene", default="default-aa.yaml")
ap.add_argument("--res", default="")
ap.add_argument("--stars", default="synthetic", help="synthetic | clustered | none | PATH of a PPM catalogue file")
a = ap.parse_args()
cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", a.scene))
if a.res:
    w, h = a.res.split("x")
    cfg = cfg.with_resolution(int(w), int(h))
tree = bs.StarTree(None if a.stars == "none" else bs.read_map(synthetic.catalogue_bytes(a.stars)))
tree.set_mode(_lib.BS_MODE_FAST if a.mode == "fast" else _lib.BS_MODE_STRICT)
for _ in range(a.frames):
    img = bs.render(cfg, tree)
    st = tree.stats()
print(a.mode, img.shape, st)
