"""The device PNG encoder (blackstar_amd/csrc/png_block.h + png_kernels.hip; writeImg's file format, src/Raytracer.hs:30-32).

CPU tests run the encoder's phase program lane by lane on the host (tests/cpp/png_emul.cpp: the same functions the GPU kernel calls) and
check its files with zlib, a by-hand chunk reader and Pillow.  GPU tests check that the HIP kernels produce the SAME BYTES as that
emulation and that the render -> file entry points decode to bs_render_rgb8's pixels."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from conftest import to_device, to_host  # noqa: E402

from tests import png_emul
from tests.conftest import load_golden

RNG = np.random.default_rng(20260927)
SCENES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")


def scene(name, w, h, bloom=None):
    import blackstar_amd as bs
    c = bs.Config.from_file(os.path.join(SCENES, name + ".yaml")).with_resolution(w, h)
    if bloom is not None:
        c.scene.bloomStrength = bloom
    return c


def frame_like(h, w, seed=0):
    """Something shaped like a rendered frame: a dark sky with a few blurred stars, a bright smooth disk, 8-bit quantised."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for _ in range(max(3, h * w // 4000)):
        cy, cx, s = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0.6, 2.5)
        img += np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2 * s * s))[..., None] * rng.uniform(0.2, 1.0, 3)
    r = np.hypot((y - h / 2) / (h / 5 + 1), (x - w / 2) / (w / 3 + 1))
    img += (np.clip(1.2 - np.abs(r - 1.0) * 3, 0, 1) ** 2)[..., None] * np.array([1.0, 0.8, 0.5])
    return np.rint(255 * np.clip(img, 0, 1)).astype(np.uint8)


def cases():
    yield "1x1", np.full((1, 1, 3), 200, np.uint8)
    yield "black 5x7", np.zeros((5, 7, 3), np.uint8)
    yield "white 100x3000", np.full((100, 3000, 3), 255, np.uint8)
    yield "noise 40x33 (stored blocks)", RNG.integers(0, 256, (40, 33, 3)).astype(np.uint8)
    yield "width 1", (RNG.integers(0, 2, (300, 1, 3)) * 255).astype(np.uint8)
    yield "height 1", RNG.integers(0, 4, (1, 5000, 3)).astype(np.uint8)
    yield "stream = exactly 2 blocks (4 x 1365)", frame_like(4, 1365, 1)
    yield "stream = 2 blocks + 1 byte", np.concatenate([frame_like(4, 1365, 1).reshape(-1), [7, 7, 7]]).astype(np.uint8)[:4 * 1365 * 3].reshape(4, 1365, 3)
    yield "frame-like 270x480", frame_like(270, 480, 2)
    yield "frame-like odd 37x23", frame_like(37, 23, 3)
    yield "two values, long runs", np.repeat(RNG.integers(0, 2, (64, 20, 3)) * 255, 40, axis=1).astype(np.uint8)
    runs = np.zeros(3 * 7000, np.uint8)   # runs of every length 1 .. 300 (matches are cut at 256 bytes and at groups of eight lanes), at every lane phase
    pos, length = 0, 1
    while pos + length < runs.size:
        runs[pos:pos + length] = 1 + (length % 250)
        pos += length + (length % 3 == 0)   # sometimes a gap of one byte (value 0) between runs
        length = length % 300 + 1
    yield "runs of every length", runs.reshape(1, 7000, 3)
    yield "every byte value", np.arange(256 * 3 * 4, dtype=np.uint32).astype(np.uint8).reshape(4, 256, 3)
    g = load_golden("image_c3_default_aa_96x54") if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "image_c3_default_aa_96x54.npz")) else None
    if g is not None:
        v = np.clip(g["img"], 0, 1)
        yield "golden c3 96x54 (sRGB8 on the host)", np.rint(255 * np.where(v < 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055)).astype(np.uint8)


CASES = list(cases())


@pytest.mark.parametrize("name,img", CASES, ids=[c[0] for c in CASES])
def test_emulated_encoder_writes_valid_png(name, img):
    data, stats, filt = png_emul.encode(img)
    facts = png_emul.check_file(data, img)
    assert len(data) <= png_emul.bound(*img.shape[:2])
    assert facts["idat_chunks"] == stats[0] + 2                      # one per block + the zlib header's + the final block's
    assert np.array_equal(np.bincount(filt, minlength=5), facts["filters"])
    for order in (1, 2, 3):  # lanes of a phase run concurrently on the GPU: the order they run in here must not matter
        assert png_emul.encode(img, order)[0] == data, f"lane order {order} changes the file"


def test_emulated_encoder_compresses_like_zlib_on_frames():
    """Ratio is not the point of this encoder (distance-1 matches only, a code table per 8 KiB), but it must stay in zlib's neighbourhood
    on what the renderer produces: at most 15 % above libpng + zlib level 1 and 70 % above level 6 on this two-thirds-black frame-like
    image (its worst case: zlib finds the repeats between rows; a frame the oracle rendered at 960x540 with bloom came out 13 % above
    level 6 and 5 % above level 1), and far below the pixels on a black one."""
    import io

    from PIL import Image
    img = frame_like(540, 960, 5)
    data, stats, _ = png_emul.encode(img)
    png_emul.check_file(data, img)
    ref = {}
    for level in (1, 6):
        b = io.BytesIO()
        Image.fromarray(img).save(b, format="PNG", compress_level=level)
        ref[level] = len(b.getvalue())
    assert stats[1] == 0 and len(data) < 1.15 * ref[1] and len(data) < 1.7 * ref[6], (len(data), ref)
    black = np.zeros((540, 960, 3), np.uint8)
    assert len(png_emul.encode(black)[0]) < black.size / 60


def fuzz_image(rng, case):
    """An image of random shape made of runs, gradients and noise in random proportions (run lengths 1 .. 600, so matches are cut at 256
    bytes, at groups of eight lanes, at blocks, at rows)."""
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 700))
    flat = np.empty(h * w * 3, np.uint8)
    pos = 0
    while pos < flat.size:
        n = int(min(flat.size - pos, rng.integers(1, 600 if case % 3 else 40)))
        kind = rng.integers(0, 4)
        if kind == 0:
            flat[pos:pos + n] = rng.integers(0, 256)
        elif kind == 1:
            flat[pos:pos + n] = (np.arange(n) // int(rng.integers(1, 9)) + rng.integers(0, 256)) % 256
        elif kind == 2:
            flat[pos:pos + n] = rng.integers(0, 256, n)
        else:
            flat[pos:pos + n] = rng.integers(0, 2, n) * rng.integers(1, 256)
        pos += n
    return flat.reshape(h, w, 3)


def test_emulated_encoder_seeded_fuzz():
    """80 seeded images (fuzz_image): every file passes the strict reader and Pillow, whatever order the lanes of a phase run in."""
    rng = np.random.default_rng(77)
    for case in range(80):
        img = fuzz_image(rng, case)
        data, stats, _ = png_emul.encode(img, order=case % 4)
        png_emul.check_file(data, img)
        assert len(data) <= png_emul.bound(*img.shape[:2])


def test_emulated_encoder_property_any_image_round_trips():
    """hypothesis: ANY (h, w, 3) byte image -- drawn from a small alphabet so that runs, repeats and block-long constants are common --
    becomes a file the strict reader and Pillow accept as exactly that image, within bs_png_bound."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 12).flatmap(lambda h: st.integers(1, 500).flatmap(lambda w: hnp.arrays(
        np.uint8, (h, w, 3), elements=st.sampled_from([0, 0, 0, 1, 2, 127, 128, 254, 255]) | st.integers(0, 255)))), st.integers(0, 3))
    def check(img, order):
        data, _, _ = png_emul.encode(img, order=order)
        png_emul.check_file(data, img)
        assert len(data) <= png_emul.bound(*img.shape[:2])

    check()


def test_encoder_format_is_pinned_by_a_digest():
    """The file of a fixed integer-built image, byte for byte (SHA-256): any change of the encoder's decisions (filter rule, tokeniser,
    code lengths, header coding, chunking) shows up here and has to be made on purpose -- update the digest together with
    profiles/EXPERIMENTS.md section 4.  (Validity is what the other tests check; this one pins the format.)"""
    import hashlib
    y, x = np.mgrid[0:120, 0:200]
    img = np.stack([(x * 7 + y * 13) % 251, np.where((x // 16 + y // 8) % 3 == 0, 0, (x * y) % 256), (x // 5) * 5 % 256], -1).astype(np.uint8)
    data, stats, _ = png_emul.encode(img)
    png_emul.check_file(data, img)
    assert (len(data), hashlib.sha256(data).hexdigest()) == (28955, "cb28f52a864de63a95f721be434634ca31626c5c86b0078146290cef3140dd2f")


def test_noise_falls_back_to_stored_blocks_within_bound():
    img = RNG.integers(0, 256, (128, 256, 3)).astype(np.uint8)
    data, stats, _ = png_emul.encode(img)
    png_emul.check_file(data, img)
    assert stats[1] == stats[0] and len(data) <= png_emul.bound(128, 256)


def test_crc_pieces_match_zlib():
    L = png_emul.lib()
    for n in (0, 1, 3, 4, 63, 64, 65, 4097, 8209):
        a = RNG.integers(0, 256, n).astype(np.uint8)
        assert L.png_emul_crc(a.ctypes.data, n) == zlib.crc32(a.tobytes()) & 0xFFFFFFFF
        for m in (0, 1, 7, 128, 5000):
            b = RNG.integers(0, 256, m).astype(np.uint8)
            got = L.png_emul_crc_combine(zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes()), m)
            assert got == zlib.crc32(a.tobytes() + b.tobytes()) & 0xFFFFFFFF, (n, m)


def test_python_bound_is_the_encoders():
    import blackstar_amd as bs
    for h, w in ((1, 1), (54, 96), (1080, 1920), (2160, 3840), (7, 10000)):
        assert bs.png_bound(h, w) == png_emul.bound(h, w)
    with pytest.raises(bs._lib.BlackstarError):
        bs.png_bound(0, 10)
    with pytest.raises(bs._lib.BlackstarError):
        bs.png_bound(100000, 100000)


# ---- on the GPU ------------------------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def tree(catalogue_bytes):
    import blackstar_amd as bs
    t = bs.StarTree(bs.read_map(catalogue_bytes))
    yield t
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,img", CASES, ids=[c[0] for c in CASES])
def test_gpu_encoder_writes_the_emulations_bytes(tree, name, img):
    """bs_encode_png: byte for byte the file of the host emulation (same phase functions), into pageable and into page-locked memory."""
    import blackstar_amd as bs
    want = png_emul.encode(img)[0]
    got = bytes(bs.encode_png(img, tree))
    assert got == want
    assert tree.stats()["zero_copy"] == 0
    buf = bs.alloc_png(tree, *img.shape[:2])
    assert bytes(bs.encode_png(img, tree, out=buf)) == want
    assert tree.stats()["zero_copy"] == 1
    png_emul.check_file(got, img)


@pytest.mark.gpu
def test_gpu_encoder_seeded_fuzz(tree):
    """The same 80 seeded images on the GPU: the emulation's bytes, each one."""
    import blackstar_amd as bs
    rng = np.random.default_rng(77)
    for case in range(80):
        img = fuzz_image(rng, case)
        assert bytes(bs.encode_png(img, tree)) == png_emul.encode(img)[0], (case, img.shape)


@pytest.mark.gpu
def test_gpu_encoder_full_size_frames(tree):
    """BASELINE-sized frames: 1080p and 4K frame-like images decode to themselves; the 1080p file equals the emulation's; timing printed."""
    import time

    import blackstar_amd as bs
    small = frame_like(270, 480, 7)
    for scale, emulate in ((4, True), (8, False)):
        img = np.ascontiguousarray(np.kron(small, np.ones((scale, scale, 1), np.uint8)))
        img[::3, ::5] ^= (RNG.integers(0, 4, img[::3, ::5].shape)).astype(np.uint8)   # some texture, so that not every block is runs
        buf = bs.alloc_png(tree, *img.shape[:2])
        data = bytes(bs.encode_png(img, tree, out=buf))
        t0 = time.perf_counter()
        for _ in range(5):
            bs.encode_png(img, tree, out=buf)
        dt = (time.perf_counter() - t0) / 5
        png_emul.check_file(data, img)
        if emulate:
            assert data == png_emul.encode(img)[0]
        print(f"bs_encode_png {img.shape[1]}x{img.shape[0]}: {len(data)} bytes ({img.size / len(data):.1f}x), {dt * 1e3:.2f} ms per call incl. H2D of the pixels")


@pytest.mark.gpu
def test_render_png_decodes_to_render_rgb8(tree):
    """bs_render_png = doRender to the end (app/Main.hs:105-123): its file decodes to exactly bs_render_rgb8's pixels -- with and without
    bloom, odd sizes, pageable and page-locked file buffers."""
    import io

    from PIL import Image

    import blackstar_amd as bs
    for name, w, h, strength in (("default-aa", 192, 108, 0.4), ("default", 160, 90, 0.0), ("lensing-disk", 97, 61, 0.25)):
        cfg = scene(name, w, h, strength)
        want = bs.render_rgb8(cfg, tree)
        for out in (None, bs.alloc_png(tree, h, w)):
            data = bytes(bs.render_png(cfg, tree, out=out))
            assert np.array_equal(np.array(Image.open(io.BytesIO(data)).convert("RGB")), want)
            png_emul.check_file(data, want)
            assert data == png_emul.encode(want)[0]


@pytest.mark.gpu
def test_render_png_batch_equals_frame_by_frame(tree):
    """bs_render_png_batch (two frames in flight, the encoder of one frame under the trace kernel of the next): every file is the file
    bs_render_png writes for that frame; mixed sizes, bloom on and off, pageable and page-locked buffers mixed; also on two contexts
    and with the chip partitioned on request."""
    import blackstar_amd as bs
    cfgs = []
    for k in range(7):
        name, w, h = (("default-aa", 160, 90), ("lensing-disk", 120, 68), ("default", 96, 54))[k % 3]
        c = scene(name, w, h, 0.0 if k == 2 else 0.3)
        c.camera.position = (c.camera.position[0], c.camera.position[1] + 0.05 * k, c.camera.position[2])
        cfgs.append(c)
    want = [bytes(bs.render_png(c, tree)) for c in cfgs]
    outs = [bs.alloc_png(tree, c.scene.resolution[1], c.scene.resolution[0]) if k % 2 == 0 else np.empty(bs.png_bound(c.scene.resolution[1], c.scene.resolution[0]), np.uint8)
            for k, c in enumerate(cfgs)]
    got = bs.render_png_batch(cfgs, [tree], outs=outs)
    assert [bytes(g) for g in got] == want
    pinned = [bs.alloc_png(tree, c.scene.resolution[1], c.scene.resolution[0]) for c in cfgs]
    assert [bytes(g) for g in bs.render_png_batch(cfgs, [tree], outs=pinned)] == want
    t2 = bs.StarTree(tree.stars)
    try:
        assert [bytes(g) for g in bs.render_png_batch(cfgs, [tree, t2])] == want
        os.environ["BLACKSTAR_POST_CUS"] = "16"   # read at bs_create: this context partitions the chip for every batch
        t3 = bs.StarTree(tree.stars)
        del os.environ["BLACKSTAR_POST_CUS"]
        try:
            pinned3 = [bs.alloc_png(t3, c.scene.resolution[1], c.scene.resolution[0]) for c in cfgs]
            assert [bytes(g) for g in bs.render_png_batch(cfgs, [t3], outs=pinned3)] == want
            assert bs._lib.debug_lib().bs_debug_last_post_cus(t3.handle) == 16
        finally:
            t3.close()
    finally:
        os.environ.pop("BLACKSTAR_POST_CUS", None)
        t2.close()
    assert bs.render_png_batch([], [tree]) == []
    with pytest.raises(bs._lib.BlackstarError, match="too small"):
        bs.render_png_batch(cfgs[:1], [tree], outs=[np.empty(100, np.uint8)])
    assert [bytes(g) for g in bs.render_png_batch(cfgs, [tree])] == want   # the context is usable after the refusal


@pytest.mark.gpu
def test_render_png_batch_at_full_size(tree):
    """BASELINE's C3 frame (default-aa.yaml, 1920x1080, 4x supersampled) through bs_render_png_batch: every file is bs_render_png's, decodes
    to bs_render_rgb8's pixels, whichever way the frame was made (the 36-frame calls run the partition trial: shared chip / 8 / 16 CUs for
    bloom + sRGB8 + the encoder, measured separately for files and for pixels); the price of the file over the pixels
    (bs_render_rgb8_batch) stays below 8 %."""
    import io
    import time

    from PIL import Image

    import blackstar_amd as bs
    import torch
    cfg = scene("default-aa", 1920, 1080)
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the partition sizes are those of a 256-CU device")
    from blackstar_amd import synthetic
    full = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL)))
    try:
        n = 36
        want_px = bs.render_rgb8(cfg, full)
        want = bytes(bs.render_png(cfg, full))
        assert np.array_equal(np.array(Image.open(io.BytesIO(want)).convert("RGB")), want_px)
        png = [bs.alloc_png(full, 1080, 1920) for _ in range(4)]
        pix = [bs.alloc_image(full, 1080, 1920, dtype=np.uint8) for _ in range(4)]
        times = {}
        for name, fn, bufs in (("png", bs.render_png_batch, png), ("rgb8", bs.render_rgb8_batch, pix)):
            outs = [bufs[i % 4] for i in range(n)]
            res = fn([cfg] * n, [full], outs=outs)
            if name == "png":
                assert all(bytes(r) == want for r in res[-4:])
                D = bs._lib.debug_lib()
                import ctypes as C
                ms = (C.c_double * 3)()
                choice = D.bs_debug_partition_choice(full.handle, C.byref(bs._lib.make_config(cfg.to_bs_config())), cfg.scene.bloomStrength, cfg.scene.bloomDivider, 1, ms)
                assert D.bs_debug_last_trial(full.handle) == 1 and choice == D.bs_debug_last_post_cus(full.handle) and choice in (0, 8, 16)
                print(f"trial, C3 as PNG files: shared {ms[0]:.3f}, 8 CUs {ms[1]:.3f}, 16 CUs {ms[2]:.3f} ms per frame -> {choice}")
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                fn([cfg] * n, [full], outs=outs)
                best = min(best, (time.perf_counter() - t0) / n)
            times[name] = best * 1e3
        print(f"C3 per frame: PNG files {times['png']:.3f} ms ({len(want)} bytes each), RGB8 pixels {times['rgb8']:.3f} ms")
        assert times["png"] < 1.08 * times["rgb8"]
    finally:
        full.close()


@pytest.mark.gpu
def test_render_scene_directory_is_the_references_batch_mode(tree, tmp_path):
    """app/Main.hs:64-77: every *.yaml of a directory, sorted, to <out>/<name>.png; an undecodable scene is reported and skipped; preview
    mode = prepareScene (300-px long side, no supersampling, no bloom, `prev-` prefix).  Files decode to bs_render_rgb8 of the scene."""
    import shutil

    import importlib.util

    import blackstar_amd as bs
    from tests.ghc_pin import decode_png_rgb8
    # (an EXAMPLE since round 4 -- the CLI side of blackstar is out of scope -- kept under test because it drives bs_render_png_files)
    spec = importlib.util.spec_from_file_location("render_scene_directory_example", os.path.join(os.path.dirname(SCENES), "examples", "render_scene_directory.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    render_scene_directory = mod.render_scene_directory
    src = tmp_path / "scenes"
    src.mkdir()
    for name in ("default-aa", "lensing-disk", "closeup"):
        text = open(os.path.join(SCENES, name + ".yaml")).read()
        (src / (name + ".yaml")).write_text(text)
    (src / "broken.yaml").write_text("scene: {stepSize: 'abc'}\ncamera: 3\n")
    (src / "notes.txt").write_text("not a scene")
    small = {}
    for name in ("default-aa", "lensing-disk", "closeup"):   # keep the test quick: the scenes at 1/10 of their resolution
        cfg = bs.Config.from_file(str(src / (name + ".yaml")))
        w, h = cfg.scene.resolution
        small[name] = cfg.with_resolution(max(w // 10, 8), max(h // 10, 8))
        import yaml
        d = yaml.safe_load(open(src / (name + ".yaml")))
        d["scene"]["resolution"] = list(small[name].scene.resolution)
        (src / (name + ".yaml")).write_text(yaml.safe_dump(d))
        small[name] = bs.Config.from_file(str(src / (name + ".yaml")))
    out = tmp_path / "out"
    paths = render_scene_directory(str(src), str(out), [tree])
    assert [os.path.basename(p) for p in paths] == ["closeup.png", "default-aa.png", "lensing-disk.png"] and sorted(os.listdir(out)) == sorted(os.path.basename(p) for p in paths)
    for name, cfg in small.items():
        assert np.array_equal(decode_png_rgb8(open(out / (name + ".png"), "rb").read()), bs.render_rgb8(cfg, tree)), name
    prev = render_scene_directory(str(src), str(out), [tree], preview=True)
    assert [os.path.basename(p) for p in prev] == ["prev-closeup.png", "prev-default-aa.png", "prev-lensing-disk.png"]
    img = decode_png_rgb8(open(out / "prev-default-aa.png", "rb").read())
    assert max(img.shape[:2]) == 300 and np.array_equal(img, bs.render_rgb8(bs.prepare_scene(small["default-aa"], True), tree))
    shutil.rmtree(out)


@pytest.mark.gpu
def test_render_png_files_writes_what_render_png_returns(tree, tmp_path):
    """bs_render_png_files (the batch loop incl. the write, in the library): 11 frames through a ring of 4 file buffers (the smallest: every
    buffer is reused, the pipeline has to wait for its writer), mixed sizes and bloom; every file is bs_render_png's bytes.  A path that
    cannot be written gives BS_EIO naming it, leaves the context usable, and the files before it are complete.  bs_files_stats tells what
    the context's writer did."""
    import blackstar_amd as bs
    cfgs = []
    for k in range(11):
        name, w, h = (("default-aa", 160, 90), ("lensing-disk", 120, 68), ("default", 96, 54))[k % 3]
        c = scene(name, w, h, 0.0 if k == 4 else 0.3)
        c.camera.position = (c.camera.position[0], c.camera.position[1] + 0.03 * k, c.camera.position[2])
        cfgs.append(c)
    want = [bytes(bs.render_png(c, tree)) for c in cfgs]
    paths = [str(tmp_path / f"f{k}.png") for k in range(11)]
    mine = os.sched_getaffinity(0)
    bs.render_png_files(cfgs, [tree], paths, pipe=2)
    assert os.sched_getaffinity(0) == mine                    # the caller's thread was only borrowed: its affinity is back
    assert [open(p, "rb").read() for p in paths] == want
    st = bs.files_stats(tree)
    assert st["files"] == 11 and st["bytes"] == sum(map(len, want)) and st["ring"] == 4 and st["writer_threads"] == 1
    assert 0 < st["writer_busy_ms"] <= st["wall_ms"] and 0 < st["writer_busy_frac"] <= 1 and st["buffer_wait_ms"] >= 0
    assert st["numa_node_gpu"] == tree.numa_node()
    if st["numa_node_gpu"] >= 0:    # the host says where the GPU hangs: the library's page-locked buffers are on that node, its threads on that node's CPUs
        assert st["numa_node_buffers"] == st["numa_node_gpu"] and st["threads_bound"] == 1, st
    t2 = bs.StarTree(tree.stars)
    try:
        two = [str(tmp_path / f"g{k}.png") for k in range(11)]
        bs.render_png_files(cfgs, [tree, t2], two, pipe=1)
        assert [open(p, "rb").read() for p in two] == want
        assert (bs.files_stats(tree)["files"], bs.files_stats(t2)["files"]) == (6, 5)
    finally:
        t2.close()
    bad = [str(tmp_path / f"h{k}.png") for k in range(11)]
    bad[5] = str(tmp_path / "no_such_directory" / "h5.png")
    with pytest.raises(bs._lib.BlackstarError, match="no_such_directory"):
        bs.render_png_files(cfgs, [tree], bad, pipe=2)
    assert [open(p, "rb").read() for p in bad[:5]] == want[:5]          # one writer, in frame order: everything before the failing file is on disk, whole
    assert bs.files_stats(tree)["files"] == 5
    bs.render_png_files(cfgs[:3], [tree], paths[:3])
    assert [open(p, "rb").read() for p in paths[:3]] == want[:3]
    bs.render_png_files([], [tree], [])


@pytest.mark.gpu
def test_one_of_several_writers_fails_mid_batch(tree, tmp_path):
    """Four contexts (all on this box's one device: four pipelines, four rings, four writer threads), 64 frames, and ONE file -- frame 22,
    context 2's sixth -- cannot be created.  The call returns BS_EIO naming that path; every context has stopped taking frames (far fewer
    than 64 files exist), nothing is written after the call has returned, every file that exists is complete and correct, and all four
    contexts render again at once -- the same batch with good paths gives 64 correct files, 16 from each writer."""
    import time

    import blackstar_amd as bs
    others = [bs.StarTree(tree.stars) for _ in range(3)]
    trees = [tree] + others
    try:
        for t in others:
            t.set_mode(tree.get_mode())
        cfgs = []
        for k in range(64):
            c = scene(*(("default-aa", 160, 90, 0.3) if k % 2 == 0 else ("default", 128, 72, 0.0)))
            c.camera.position = (c.camera.position[0], c.camera.position[1] + 0.01 * k, c.camera.position[2])
            cfgs.append(c)
        want = [bytes(bs.render_png(c, tree)) for c in cfgs]
        paths = [str(tmp_path / f"a{k}.png") for k in range(64)]
        paths[22] = str(tmp_path / "missing" / "a22.png")
        with pytest.raises(bs._lib.BlackstarError, match="missing/a22.png") as e:
            bs.render_png_files(cfgs, trees, paths, pipe=4)
        assert "rc=-6" in str(e.value)                                    # BS_EIO
        listing = sorted(os.listdir(tmp_path))
        sizes = {f: os.path.getsize(tmp_path / f) for f in listing}
        assert "a22.png" not in listing and len(listing) < 64
        stats = [bs.files_stats(t) for t in trees]
        assert sum(s["files"] for s in stats) == len(listing) and stats[2]["files"] == 5     # frames 2, 6, 10, 14, 18 -- then 22 failed
        assert all(s["writer_threads"] == 1 and s["ring"] == 4 for s in stats)
        time.sleep(0.3)
        assert sorted(os.listdir(tmp_path)) == listing and {f: os.path.getsize(tmp_path / f) for f in listing} == sizes, "a writer outlived the call"
        for f in listing:
            assert open(tmp_path / f, "rb").read() == want[int(f[1:-4])], f
        for k, t in enumerate(trees):                                     # nothing in flight, every context usable straight away
            assert bytes(bs.render_png(cfgs[k], t)) == want[k]
        good = [str(tmp_path / f"b{k}.png") for k in range(64)]
        bs.render_png_files(cfgs, trees, good, pipe=4)
        assert [open(p, "rb").read() for p in good] == want
        assert [bs.files_stats(t)["files"] for t in trees] == [16, 16, 16, 16]
    finally:
        for t in others:
            t.close()


@pytest.mark.gpu
def test_png_entry_points_refuse_bad_arguments(tree):
    """BS_EINVAL (-1) with a message, nothing launched, the context usable afterwards: null pointers, buffers below bs_png_bound, frames
    the encoder's 32-bit chunk offsets cannot hold, a bloom radius of 0, a bad configuration."""
    import blackstar_amd as bs
    from blackstar_amd import _lib
    L = _lib.lib()
    img = frame_like(20, 30, 1)
    cap = bs.png_bound(20, 30)
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(123)
    cfg = scene("default-aa", 30, 20)
    c = _lib.make_config(cfg.to_bs_config())
    cases = {
        "encode: null image": lambda: L.bs_encode_png(tree.handle, None, 30, 20, out.ctypes.data, cap, C.byref(n)),
        "encode: null out": lambda: L.bs_encode_png(tree.handle, img.ctypes.data, 30, 20, None, cap, C.byref(n)),
        "encode: null size": lambda: L.bs_encode_png(tree.handle, img.ctypes.data, 30, 20, out.ctypes.data, cap, None),
        "encode: small buffer": lambda: L.bs_encode_png(tree.handle, img.ctypes.data, 30, 20, out.ctypes.data, cap - 1, C.byref(n)),
        "encode: width 0": lambda: L.bs_encode_png(tree.handle, img.ctypes.data, 0, 20, out.ctypes.data, cap, C.byref(n)),
        "encode: too large": lambda: L.bs_encode_png(tree.handle, img.ctypes.data, 60000, 60000, out.ctypes.data, cap, C.byref(n)),
        "render: small buffer": lambda: L.bs_render_png(tree.handle, C.byref(c), 0.2, 5, out.ctypes.data, cap - 1, C.byref(n)),
        "render: bloom radius 0": lambda: L.bs_render_png(tree.handle, C.byref(c), 0.2, 1000, out.ctypes.data, cap, C.byref(n)),
        "render: null cfg": lambda: L.bs_render_png(tree.handle, None, 0.2, 5, out.ctypes.data, cap, C.byref(n)),
        "batch: null sizes": lambda: L.bs_render_png_batch((C.c_void_p * 1)(tree.handle), 1, C.byref(c), 1, None, None, (C.c_void_p * 1)(out.ctypes.data), (C.c_size_t * 1)(cap), None),
        "phases: small clock buffer": lambda: _lib.debug_lib().bs_debug_png_phases(tree.handle, img.ctypes.data, 30, 20, out.ctypes.data, 3),
    }
    for name, call in cases.items():
        assert call() == -1 and _lib.last_error(), name
    assert n.value == 123 and not out.any()
    c.fov = float("nan")
    assert L.bs_render_png(tree.handle, C.byref(c), 0.0, 5, out.ctypes.data, cap, C.byref(n)) == -1 and "fov" in _lib.last_error()
    assert bytes(bs.encode_png(img, tree)) == png_emul.encode(img)[0]


@pytest.mark.gpu
def test_encode_png_device_is_enqueue_only(tree):
    """bs_encode_png_device on torch tensors and a torch stream: the file and its size appear once the stream has passed."""
    import torch

    import blackstar_amd as bs
    from blackstar_amd import _lib
    img = frame_like(120, 200, 11)
    d_img = to_device(img)
    d_png = torch.zeros(bs.png_bound(120, 200), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    for _ in range(3):   # back to back on one stream, and once more on another: the context's scratch is handed over in order
        _lib.check(_lib.lib().bs_encode_png_device(tree.handle, d_img.data_ptr(), 200, 120, d_png.data_ptr(), d_png.numel(), d_n.data_ptr(), s.cuda_stream), "bs_encode_png_device")
    s.synchronize()
    want = png_emul.encode(img)[0]
    assert int(d_n.item()) == len(want) and bytes(d_png[:len(want)].cpu().numpy()) == want
    rc = _lib.lib().bs_encode_png_device(tree.handle, d_img.data_ptr(), 200, 120, d_png.data_ptr(), 1000, d_n.data_ptr(), s.cuda_stream)
    assert rc == -1 and "too small" in _lib.last_error()   # BS_EINVAL
