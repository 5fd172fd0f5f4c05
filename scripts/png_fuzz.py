#!/usr/bin/env python
"""N seeded images (tests/test_png.py: fuzz_image -- runs, gradients, noise in random proportions; every tenth one larger, up to
1200 x 300) through bs_encode_png: the GPU's file must be the host emulation's, byte for byte, and every twentieth file goes through the
strict reader + Pillow.  Usage: png_fuzz.py [N [SEED]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import synthetic
from tests import png_emul
from tests.test_png import fuzz_image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260927
rng = np.random.default_rng(seed)
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_SMALL)))
t0 = time.time()
out = dict(images=0, bytes_in=0, bytes_out=0, stored_blocks=0, blocks=0, mismatches=0, decoded=0, seed=seed)
for case in range(N):
    img = fuzz_image(rng, case)
    if case % 10 == 0:
        img = np.ascontiguousarray(np.tile(img, (int(rng.integers(1, 8)), int(rng.integers(1, 3)), 1))[:300, :1200])
    want, stats, _ = png_emul.encode(img)
    got = bytes(bs.encode_png(img, tree))
    out["images"] += 1; out["bytes_in"] += img.size; out["bytes_out"] += len(got)
    out["blocks"] += int(stats[0]); out["stored_blocks"] += int(stats[1])
    if got != want:
        out["mismatches"] += 1
        print(f"MISMATCH case {case} shape {img.shape}", file=sys.stderr)
    if case % 20 == 0:
        png_emul.check_file(got, img)
        out["decoded"] += 1
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))
sys.exit(1 if out["mismatches"] else 0)
