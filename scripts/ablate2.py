import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic
from oracle import scenes
trees = {"none": bs.StarTree(None), "1023 (LDS levels only)": bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(1023))),
         "65535": bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(65535))), "470000": bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))}
for rnd in range(2):
    for name, t in trees.items():
        t.set_mode(_lib.BS_MODE_FAST); ms = []
        for _ in range(7):
            bs.render(scenes.DEFAULT_AA, t); st = t.stats(); ms.append(st["kernel_ms"])
        print(f"stars {name:24s} kernel_ms median {np.median(ms):.3f}  star_hits {st['star_hits']}")
