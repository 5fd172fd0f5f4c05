#!/usr/bin/env python3
"""Soak of bs_render_png_files' per-context pipelines, rings and writers (csrc/batch.cpp: FileRing): rounds of 1..6 contexts on this box's
device, 1..90 frames of random small shapes (one shape per round or mixed), ring sizes 1..20, and in a third of the rounds ONE path that
cannot be created.  A good round: every file is bs_render_png's bytes and every writer wrote its share.  A failing round: BS_EIO naming
the path, the files that exist are complete and correct, nothing appears after the call has returned, and every context renders again
at once.  Looks for deadlocks (run it under `timeout`), lost wake-ups, files written twice or never, buffers reused too early.
Usage: files_soak.py [SECONDS [SEED]]"""
import copy
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stars = bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_SMALL))
os.environ.setdefault("BLACKSTAR_POST_CUS", "0")   # several contexts share one device here: no CU partition (what bench.py's smoke modes do)
trees = [bs.StarTree(stars) for _ in range(6)]
for t in trees:
    t.set_mode(_lib.BS_MODE_FAST)
anim = bs.Animation.from_file(os.path.join(root, "animations", "default-ani.yaml"))
anim.nFrames = 600
frames = bs.generate_frames(anim)
base = tempfile.mkdtemp(prefix="bs_files_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t_end = time.time() + seconds
rounds = good = failing = files_checked = 0
try:
    while time.time() < t_end:
        n_ctx = int(rng.integers(1, 7))
        use = [trees[i] for i in rng.permutation(6)[:n_ctx]]
        n = int(rng.integers(1, 91))
        pipe = int(rng.integers(1, 21))
        mixed = rng.random() < 0.3
        shape = lambda: ((int(rng.integers(20, 120)) * 2, int(rng.integers(15, 70)) * 2), bool(rng.random() < 0.6),
                         0.0 if rng.random() < 0.3 else float(rng.uniform(0.05, 0.5)), int(rng.integers(5, 30)))
        one = shape()
        cfgs = []
        for _ in range(n):
            c = copy.deepcopy(frames[int(rng.integers(0, 600))])
            c.scene.resolution, c.scene.supersampling, c.scene.bloomStrength, c.scene.bloomDivider = shape() if mixed else one
            cfgs.append(c)
        d = os.path.join(base, f"r{rounds}")
        os.mkdir(d)
        paths = [os.path.join(d, f"f{i}.png") for i in range(n)]
        bad = int(rng.integers(0, n)) if rng.random() < 0.33 else -1
        if bad >= 0:
            paths[bad] = os.path.join(d, "missing", f"f{bad}.png")
        check = sorted(set(int(i) for i in rng.choice(n, size=min(n, 5), replace=False)))
        want = {i: bytes(bs.render_png(cfgs[i], trees[0])) for i in check}
        try:
            bs.render_png_files(cfgs, use, paths, pipe=pipe)
            assert bad < 0, f"round {rounds}: the unwritable path {paths[bad]} did not fail the call"
            assert all(os.path.exists(p) for p in paths)
            st = [bs.files_stats(t) for t in use]
            assert [s["files"] for s in st] == [len(range(c, n, n_ctx)) for c in range(n_ctx)], (n, n_ctx, [s["files"] for s in st])
            assert sum(s["bytes"] for s in st) == sum(os.path.getsize(p) for p in paths)
            for i in check:
                assert open(paths[i], "rb").read() == want[i], f"round {rounds}: file {i} differs"
            files_checked += len(check)
            good += 1
        except _lib.BlackstarError as e:
            assert bad >= 0 and "missing" in str(e) and "rc=-6" in str(e), f"round {rounds}: {e}"
            listing = sorted(os.listdir(d))
            sizes = [os.path.getsize(os.path.join(d, f)) for f in listing]
            assert f"f{bad}.png" not in listing
            time.sleep(0.02)
            assert sorted(os.listdir(d)) == listing and [os.path.getsize(os.path.join(d, f)) for f in listing] == sizes, "a writer outlived the call"
            for i in check:
                if os.path.exists(paths[i]) and i != bad:
                    assert open(paths[i], "rb").read() == want[i], f"round {rounds}: file {i} (before the failure) differs"
                    files_checked += 1
            k = int(rng.integers(0, n_ctx))   # every context is usable at once
            assert bytes(bs.render_png(cfgs[check[0]], use[k])) == want[check[0]]
            failing += 1
        shutil.rmtree(d)
        rounds += 1
finally:
    shutil.rmtree(base, ignore_errors=True)
    for t in trees:
        t.close()
print(f"files soak: {rounds} rounds in {seconds:.0f} s ({good} good, {failing} with an unwritable path), {files_checked} files compared byte for byte: all as expected")
