#!/usr/bin/env python
"""Where do FAST's largest deviations from STRICT in scripts/fuzz_modes.py come from?  Re-runs the fuzz scenes, keeps the worst
pixels, and traces their rays in both modes: step counts against the guard threshold, fates, crossings, star hits.
Usage: fuzz_worst.py [N_SCENES [SEED]]"""
import heapq, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 424242
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_modes.py")).read()
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_modes.py")}
exec(src[:src.index("out = dict(scenes=0")].replace("N = int(sys.argv[1])", "N = %d  #" % N).replace("int(sys.argv[2]) if len(sys.argv) > 2 else 424242", str(seed)), ns)
scene, tree = ns["scene"], ns["tree"]
top = []
for i in range(N):
    cfg = scene(i)
    tree.set_mode(_lib.BS_MODE_STRICT); a = bs.render(cfg, tree)
    tree.set_mode(_lib.BS_MODE_FAST); b = bs.render(cfg, tree)
    d = np.abs(a - b)
    rel = np.where(np.abs(a) > 1e-3, d / np.maximum(np.abs(a), 1e-300), 0.0)
    k = int(np.argmax(rel))
    item = (float(rel.flat[k]), i, k, cfg)
    if len(top) < 6: heapq.heappush(top, item)
    elif item[0] > top[0][0]: heapq.heapreplace(top, item)
for relv, i, k, cfg in sorted(top, reverse=True):
    y, x, c = np.unravel_index(k, (cfg["height"], cfg["width"], 3))
    f = 2 if cfg["supersampling"] else 1
    ys = np.array([f * y + dy for dy in range(f) for dx in range(f)]); xs = np.array([f * x + dx for dy in range(f) for dx in range(f)])
    tree.set_mode(_lib.BS_MODE_STRICT); ra = bs.trace_rays(cfg, tree, ys, xs)
    tree.set_mode(_lib.BS_MODE_FAST); rb = bs.trace_rays(cfg, tree, ys, xs)
    cam = float(np.linalg.norm(cfg["cam_pos"])); safe = max(2500.0, 2 * cam * cam)
    n0 = (cam + np.sqrt(safe)) / cfg["step_size"]
    print(json.dumps(dict(worst_rel=relv, scene=i, pixel=(int(y), int(x), int(c)), step_size=cfg["step_size"], cam_r=cam, n0=n0, guard=n0 + 9 / cfg["step_size"],
                          steps=ra["steps"].tolist(), fate=ra["fate"].tolist(), disk_hits=ra["disk_hits"].tolist(), star_hits=ra["star_hits"].tolist(),
                          ray_rel_dev=[float(v) for v in (np.abs(ra["rgba"] - rb["rgba"]) / (np.abs(ra["rgba"]) + 1e-3)).max(axis=1)],
                          vel_dev=[float(v) for v in np.abs(ra["vel"] - rb["vel"]).max(axis=1)], disk_inner=cfg["disk_inner"], disk_outer=cfg["disk_outer"])))
