#!/usr/bin/env python
"""Parity report (SURVEY.md 8d "Parity check"): per BASELINE config, max abs / rel error and the COUNT of channel values
outside |gpu - ref| <= 1e-4 |ref| + 1e-7.  The oracle is only the checker here.  Full-size configs whose CPU cost is too
high are checked (a) GPU STRICT vs oracle on a down-scaled copy and on 16k sampled rays at full size, (b) GPU FAST vs GPU
STRICT on every pixel at full size."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from oracle import c_oracle, scenes  # noqa: E402

RT, AT = 1e-4, 1e-7
cat = synthetic.ppm_catalogue_bytes()
tree = bs.StarTree(bs.read_map(cat)); empty = bs.StarTree(None)
ix = c_oracle.Index(c_oracle.read_ppm(cat)); ix0 = c_oracle.Index(None)


def cmp(a, ref):
    d = np.abs(a - ref)
    rel = d / np.maximum(np.abs(ref), 1e-300)
    return {"values": int(ref.size), "outside_1e-4": int((d > AT + RT * np.abs(ref)).sum()), "max_abs": float(d.max()),
            "max_rel_where_ref>1e-3": float(rel[np.abs(ref) > 1e-3].max()) if (np.abs(ref) > 1e-3).any() else 0.0,
            "bit_equal_fraction": float((a == ref).mean())}


configs = [("C1 default.yaml 640x480", scenes.with_res(scenes.DEFAULT, 640, 480), tree, ix, 1.0),
           ("C2 default.yaml 1920x1080 no stars", scenes.DEFAULT, empty, ix0, 1.0),
           ("C3 default-aa.yaml 1920x1080 4xSS", scenes.DEFAULT_AA, tree, ix, 0.5),
           ("C4 lensing-disk.yaml 3840x2160 4xSS", scenes.with_res(scenes.LENSING_DISK, 3840, 2160), tree, ix, 0.25),
           ("C5 default-ani frame 300/600 1920x1080 4xSS", scenes.ani_frame(300, 600), tree, ix, 0.5)]
rng = np.random.default_rng(1)
rows = []
for name, cfg, t, oix, scale in configs:
    small = scenes.with_res(cfg, int(cfg["width"] * scale), int(cfg["height"] * scale))
    ref, ost = c_oracle.render(small, oix, threads=0)
    row = {"config": name, "oracle_image": f"{small['width']}x{small['height']}"}
    for mode, m in (("strict", _lib.BS_MODE_STRICT), ("fast", _lib.BS_MODE_FAST)):
        t.set_mode(m)
        img = bs.render(small, t); st = t.stats()
        row[f"gpu_{mode}_vs_oracle"] = dict(cmp(img, ref), steps_equal=bool(st["steps"] == ost["steps"]),
                                            fates_equal=bool((st["horizon"], st["escaped"]) == (ost["horizon"], ost["escaped"])))
    # full size: sampled rays vs oracle (strict, bit-exact trajectories) and FAST vs STRICT on every pixel
    ss = 2 if cfg["supersampling"] else 1
    ys, xs = rng.integers(0, cfg["height"] * ss, 16384), rng.integers(0, cfg["width"] * ss, 16384)
    t.set_mode(_lib.BS_MODE_STRICT)
    rec = bs.trace_rays(cfg, t, ys, xs); orc = c_oracle.trace_rays(cfg, oix, ys, xs)
    row["full_size_16k_rays_strict"] = {k: bool(np.array_equal(rec[k], orc[k])) for k in ("steps", "fate", "disk_hits", "star_hits", "vel", "pos")}
    row["full_size_16k_rays_strict"]["rgba"] = cmp(rec["rgba"], orc["rgba"])
    s_img = bs.render(cfg, t); t.set_mode(_lib.BS_MODE_FAST); f_img = bs.render(cfg, t)
    row["full_size_fast_vs_strict"] = cmp(f_img, s_img)
    rows.append(row)
    print(json.dumps(row), flush=True)
