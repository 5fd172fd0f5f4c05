import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import synthetic
from oracle import scenes
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
full = bs.StarTree(stars); empty = bs.StarTree(None)
def run(name, cfg, t, mode, n=7):
    t.set_mode(mode); ms=[]
    for _ in range(n):
        bs.render(cfg, t); ms.append(t.stats()['kernel_ms'])
    print(f"{name:40s} mode {mode} kernel_ms median {np.median(ms):.3f} min {min(ms):.3f}")
nodisk = dict(scenes.DEFAULT_AA, disk_opacity=0.0)
for rnd in range(2):
    for mode in (1, 0):
        run('C3 stars+disk', scenes.DEFAULT_AA, full, mode)
        run('C3 no stars', scenes.DEFAULT_AA, empty, mode)
        run('C3 no disk', nodisk, full, mode)
        run('C3 no stars no disk', nodisk, empty, mode)
