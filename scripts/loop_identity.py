#!/usr/bin/env python
"""Bit-identity of two builds of the trace kernel (the assembly stepping loop -- built with -DBS_FL_SERIES=0, the series variant moves FAST by up to
4.5e-10 -- against the C++ statement of the same steps, -DBS_ASM_LOOP=0; or any two builds that should not differ in a bit):
prints one sha256 per frame over the FAST image bytes and the statistics -- the nine scene files the reference ships at 480 x 270 (with and
without supersampling), N random fuzz scenes on the clustered sky (degenerate geometries, disks inside the photon sphere, tiny disk queues
via long orbits), and the C3 frame at full size.  Run it once per library and diff the outputs:
    BLACKSTAR_LIB=$PWD/variants_w_base.so python scripts/loop_identity.py > a.txt; python scripts/loop_identity.py > b.txt; cmp a.txt b.txt
Usage: loop_identity.py [N_FUZZ [SEED]]"""
import copy, glob, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic
from fuzz_scenes import scene as _scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261001)
tree = bs.StarTree(bs.read_map(synthetic.clustered_catalogue_bytes(n_uniform=20000, n_clusters=30000)))
tree.set_max_steps(20000)
tree.set_mode(_lib.BS_MODE_FAST)
KEYS = ("rays", "steps", "capped", "horizon", "escaped", "disk_hits", "star_hits", "wave_iters", "effective_mode")


def line(name, cfg):
    img = bs.render(cfg, tree)
    st = tree.stats()
    print(name, hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()[:32], " ".join(str(int(st[k])) for k in KEYS), flush=True)


for path in sorted(glob.glob(os.path.join(ROOT, "scenes", "*.yaml"))):
    cfg = bs.Config.from_file(path)
    for ss in (False, True):
        c = copy.deepcopy(cfg.with_resolution(480, 270))
        c.scene.supersampling = ss
        line(f"{os.path.basename(path)} ss={int(ss)}", c)
for i in range(N):
    line(f"fuzz {i}", _scene(rng, i))
line("default-aa.yaml full size", bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml")))
