#!/usr/bin/env python
"""A/B of the tile queue's LATE POPS at the end of a launch (TraceParams.tail_tiles, env BLACKSTAR_TAIL_TILES) -- the 14th experiment on the
fixed cost of a launch (profiles/EXPERIMENTS.md 1.2).  Variants = (library, BLACKSTAR_TAIL_TILES); each runs in its own child process
(the setting is read at bs_create), variants interleaved over ROUNDS rounds on one box; per variant and workload: hipEvent kernel time of 24
frames rendered back to back into a page-locked buffer (zero copy), the median and the minimum of the last 16 kept.
NULL RESULT, patch not kept (profiles/EXPERIMENTS.md 1.2 describes it in full: the library in the tree ignores BLACKSTAR_TAIL_TILES, so today this
script measures seven times the same thing -- it is here to show how the numbers of profiles/r05_tail_pop_ab.jsonl were taken).
Usage: tail_pop_ab.py [ROUNDS]     (child: tail_pop_ab.py --child)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKLOADS = (("C3 default-aa 1080p ss", "default-aa.yaml", None, True), ("C2 default 1080p", "default.yaml", None, False),
             ("C4 lensing 4K ss", "lensing-disk.yaml", (3840, 2160), True), ("default-aa 720p ss", "default-aa.yaml", (1280, 720), True),
             ("C4 band 1/8 (270 rows)", "lensing-disk.yaml", (3840, 2160), True))

if "--child" in sys.argv:
    import numpy as np
    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic
    stars = bs.read_map(synthetic.catalogue_bytes("synthetic"))
    trees = {True: bs.StarTree(stars), False: bs.StarTree(None)}
    out = {}
    for name, scene, res, with_stars in WORKLOADS:
        cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", scene))
        if res:
            cfg = cfg.with_resolution(*res)
        t = trees[with_stars]
        t.set_mode(_lib.BS_MODE_FAST)
        band = "band" in name
        rows = (1080, 1350) if band else (0, cfg.scene.resolution[1])
        buf = bs.alloc_image(t, rows[1] - rows[0], cfg.scene.resolution[0])
        ms = []
        for _ in range(24):
            bs.render_rows(cfg, t, rows[0], rows[1], out=buf)
            ms.append(float(t.stats()["kernel_ms"]))
        st = t.stats()
        out[name] = {"median": float(np.median(ms[8:])), "min": float(min(ms[8:])), "steps": int(st["steps"])}
    print(json.dumps(out))
    sys.exit(0)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tree_lib, patched = os.path.join(ROOT, "blackstar_amd", "libblackstar_gpu.so"), os.path.join(ROOT, "variants_wtail.so")
variants = [("the tree's library", tree_lib, None)]
if os.path.exists(patched):   # TAIL_TILES > 0: late pops in the tail; < 0: EXIT mode (slots 1..3 stop taking tiles |T| << {0,2,4} tiles before the end)
    variants += [(f"patched, TAIL_TILES={t}", patched, str(t)) for t in (os.environ.get("TAIL_SET") or "0,-1024,-2048,-4096,-8192,-16384").split(",")]
acc = {v[0]: {w[0]: [] for w in WORKLOADS} for v in variants}
for r in range(rounds):
    for label, lib, tail in variants:
        env = dict(os.environ, BLACKSTAR_LIB=lib)
        env.pop("BLACKSTAR_TAIL_TILES", None)
        if tail is not None:
            env["BLACKSTAR_TAIL_TILES"] = tail
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            print(json.dumps({"variant": label, "round": r, "error": p.stderr[-400:]}), flush=True)
            continue
        got = json.loads(p.stdout.strip().splitlines()[-1])
        for w, v in got.items():
            acc[label][w].append(v)
for label, per in acc.items():
    row = {"variant": label}
    for w, vs in per.items():
        if vs:
            row[w] = {"median_ms": round(sorted(v["median"] for v in vs)[len(vs) // 2], 4), "min_ms": round(min(v["min"] for v in vs), 4),
                      "all_medians": [round(v["median"], 4) for v in vs], "steps": vs[0]["steps"]}
    print(json.dumps(row), flush=True)
