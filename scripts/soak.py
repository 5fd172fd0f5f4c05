#!/usr/bin/env python3
"""Soak of the pipelined host paths: many rounds of bs_render_batch / bs_render_rgb8_batch / bs_render_png_batch / multi-stream
bs_render_device + bs_bloom_device + bs_encode_png_device with random frame sizes, buffer kinds (pageable / page-locked) and bloom
settings, every result compared byte for byte with the frame-by-frame blocking calls (PNG files: with the bytes of bs_encode_png of
the frame's pixels, and every tenth one decoded).  Looks for ordering bugs (streams, events, shared scratch) that a single test run might miss."""
import copy
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import _lib, synthetic  # noqa: E402
from tests.ghc_pin import decode_png_rgb8  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
trees = [bs.StarTree(stars), bs.StarTree(stars)]
for t in trees:
    t.set_mode(_lib.BS_MODE_FAST)
anim = bs.Animation.from_file(os.path.join(root, "animations", "default-ani.yaml"))
anim.nFrames = 600
frames = bs.generate_frames(anim)
t_end = time.time() + seconds
rounds = checked = 0
D = _lib.debug_lib()
trials = {"ended": 0, "progressed": 0, "choices": {}}


def same_shape_round():
    """Round 4: batches of ONE shape, 8 .. 40 frames of different cameras, all outputs page-locked -- what the partition trial needs
    (csrc/batch.cpp).  A shape is kept for a few calls so that trials are continued over calls, end, and later calls run with the remembered
    choice; whichever pipeline made a frame (shared / 8 / 16 post-stage CUs, warm-up or timed segment), its bytes are the blocking call's.
    Right before some batches an enqueue-only bs_encode_png_device is left in flight on a caller's stream (its scratch must be
    its own)."""
    global checked
    w, h = int(rng.integers(40, 160)) * 2, int(rng.integers(30, 90)) * 2
    ss = bool(rng.random() < 0.7)
    strength = 0.0 if rng.random() < 0.2 else float(rng.uniform(0.05, 0.5))
    divider = int(rng.integers(5, 30))
    png = bool(rng.random() < 0.5)
    use = trees if rng.random() < 0.4 else trees[:1]
    for _call in range(int(rng.integers(1, 5))):
        n = int(rng.integers(8, 41)) * len(use)
        cfgs = []
        for _ in range(n):
            c = copy.deepcopy(frames[int(rng.integers(0, 600))])
            c.scene.resolution, c.scene.supersampling, c.scene.bloomStrength, c.scene.bloomDivider = (w, h), ss, strength, divider
            cfgs.append(c)
        idx = sorted(int(i) for i in rng.choice(n, size=min(n, 6), replace=False))   # reference renders for a few frames of the call, at random positions
        want8 = {i: bs.render_rgb8(cfgs[i], trees[0]).copy() for i in idx}
        pending = None
        if rng.random() < 0.5:   # an enqueue-only encode on a caller's stream, not waited for before the batch starts
            src = want8[idx[0]]
            st = torch.cuda.Stream()
            d8 = torch.from_numpy(src).cuda()
            dp = torch.zeros(bs.png_bound(*src.shape[:2]), dtype=torch.uint8, device="cuda:0")
            dn = torch.zeros(1, dtype=torch.int64, device="cuda:0")
            st.wait_stream(torch.cuda.current_stream())
            _lib.check(_lib.lib().bs_encode_png_device(trees[0].handle, d8.data_ptr(), src.shape[1], src.shape[0], dp.data_ptr(), dp.numel(), dn.data_ptr(),
                                                       C.c_void_p(st.cuda_stream)), "png on a stream")
            pending = (src, d8, dp, dn)
        if png:
            outs = [bs.alloc_png(use[i % len(use)], h, w) for i in range(n)]
            got = bs.render_png_batch(cfgs, use, outs=outs)
            for i in idx:
                assert bytes(got[i]) == bytes(bs.encode_png(want8[i], trees[0])), f"same-shape round: png frame {i} of {n} differs ({w}x{h})"
        else:
            outs = [bs.alloc_image(use[i % len(use)], h, w, dtype=np.uint8) for i in range(n)]
            got = bs.render_rgb8_batch(cfgs, use, outs=outs)
            for i in idx:
                assert np.array_equal(got[i], want8[i]), f"same-shape round: rgb8 frame {i} of {n} differs ({w}x{h})"
        if pending is not None:
            torch.cuda.synchronize()
            src, d8, dp, dn = pending
            assert bytes(dp[:int(dn[0].item())].cpu().numpy()) == bytes(bs.encode_png(src, trees[0])), "enqueue-only png raced with the batch"
        for t in use:
            state = D.bs_debug_last_trial(t.handle)
            trials["ended"] += state == 1
            trials["progressed"] += state == 2
            if state == 1:
                k = str(D.bs_debug_last_post_cus(t.handle))
                trials["choices"][k] = trials["choices"].get(k, 0) + 1
        checked += len(idx)


while time.time() < t_end:
    if rounds % 4 == 3:
        same_shape_round()
    n = int(rng.integers(1, 9))
    cfgs = []
    for _ in range(n):
        c = copy.deepcopy(frames[int(rng.integers(0, 600))])
        w = int(rng.integers(8, 200)) * 2
        h = int(rng.integers(8, 120)) * (2 if rng.random() < 0.7 else 1) + int(rng.random() < 0.2)
        c.scene.resolution = (w, h)
        c.scene.supersampling = bool(rng.random() < 0.6)
        c.scene.bloomStrength = 0.0 if rng.random() < 0.25 else float(rng.uniform(0.05, 0.5))
        c.scene.bloomDivider = int(rng.integers(3, 40))
        if w // c.scene.bloomDivider == 0:
            c.scene.bloomDivider = 2
        cfgs.append(c)
    use = trees if rng.random() < 0.5 else trees[:1]
    want8 = [bs.render_rgb8(c, trees[0]) for c in cfgs]
    want = [bs.render(c.to_bs_config(), trees[0]) for c in cfgs]
    outs8 = [bs.alloc_image(use[i % len(use)], *w8.shape[:2], dtype=np.uint8) if rng.random() < 0.5 else np.zeros_like(w8) for i, w8 in enumerate(want8)]
    got8 = bs.render_rgb8_batch(cfgs, use, outs=outs8)
    outs = [bs.alloc_image(use[i % len(use)], *wf.shape[:2]) if rng.random() < 0.5 else np.zeros_like(wf) for i, wf in enumerate(want)]
    got = bs.render_batch([c.to_bs_config() for c in cfgs], use, outs=outs)
    wantp = [bytes(bs.encode_png(w8, trees[0])) for w8 in want8]
    outsp = [bs.alloc_png(use[i % len(use)], *w8.shape[:2]) if rng.random() < 0.5 else np.zeros(bs.png_bound(*w8.shape[:2]), np.uint8) for i, w8 in enumerate(want8)]
    gotp = bs.render_png_batch(cfgs, use, outs=outsp)
    for i in range(n):
        assert bytes(gotp[i]) == wantp[i], f"round {rounds}: png batch frame {i} differs ({cfgs[i].scene.resolution})"
        if (checked + i) % 10 == 0:
            assert np.array_equal(decode_png_rgb8(wantp[i]), want8[i]), f"round {rounds}: png of frame {i} does not decode to it"
        assert np.array_equal(got8[i], want8[i]), f"round {rounds}: rgb8 batch frame {i} differs ({cfgs[i].scene.resolution})"
        assert np.array_equal(got[i], want[i]), f"round {rounds}: f64 batch frame {i} differs"
    # the same frames enqueued on as many streams as frames, no synchronisation in between, plus bloom on each stream
    streams = [torch.cuda.Stream() for _ in range(n)]
    dev = [torch.empty(wf.shape, dtype=torch.float64, device="cuda:0") for wf in want]
    L = _lib.lib()
    for c, d, s in zip(cfgs, dev, streams):
        bs.render_device(c.to_bs_config(), trees[0], d.data_ptr(), d.numel(), s.cuda_stream)
        if c.scene.bloomStrength != 0:
            hh, ww = d.shape[:2]
            _lib.check(L.bs_bloom_device(trees[0].handle, d.data_ptr(), d.data_ptr(), ww, hh, float(c.scene.bloomStrength), int(c.scene.bloomDivider),
                                         C.c_void_p(s.cuda_stream)), "bloom on a stream")
    # ... and the PNG encoder enqueued on each stream behind them (one scratch per context: handed from stream to stream in order)
    d8 = [torch.from_numpy(w8).cuda() for w8 in want8]
    dp = [torch.zeros(bs.png_bound(*w8.shape[:2]), dtype=torch.uint8, device="cuda:0") for w8 in want8]
    dn = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    for i, s in enumerate(streams):
        s.wait_stream(torch.cuda.current_stream())
        hh, ww = want8[i].shape[:2]
        _lib.check(L.bs_encode_png_device(trees[0].handle, d8[i].data_ptr(), ww, hh, dp[i].data_ptr(), dp[i].numel(), dn.data_ptr() + 8 * i,
                                          C.c_void_p(s.cuda_stream)), "png on a stream")
    torch.cuda.synchronize()
    for i in range(n):
        assert bytes(dp[i][:int(dn[i].item())].cpu().numpy()) == wantp[i], f"round {rounds}: multi-stream png {i} differs"
    for i, (c, d) in enumerate(zip(cfgs, dev)):
        ref = want[i] if c.scene.bloomStrength == 0 else bs.bloom(float(c.scene.bloomStrength), int(c.scene.bloomDivider), want[i], trees[0])
        assert np.array_equal(d.cpu().numpy(), ref), f"round {rounds}: multi-stream frame {i} differs"
    rounds += 1
    checked += 5 * n
print(f"soak: {rounds} rounds, {checked} frames compared, all identical, {seconds:.0f} s; partition trials ended {trials['ended']}, "
      f"continued over calls {trials['progressed']}, choices {trials['choices']}")
