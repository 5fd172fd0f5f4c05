"""Reader for the output of tools/ghc_pin/Dump.hs -- the REFERENCE's own functions run on this repository's fixed inputs -- and the
comparisons the oracle / GPU tests make against it.  tests/golden/ghc/<set>/ holds a dump once somebody with GHC 8.6 / lts-13.16 has run
the kit (tools/ghc_pin/README.md); until then every test that needs it SKIPS with the reason below: the oracle stays pinned only by
independent restatements, mpmath and physics -- "parity unpinned" by the reference itself."""
import os
import struct
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMPS = os.path.join(ROOT, "tests", "golden", "ghc")
INPUTS = os.path.join(ROOT, "tools", "ghc_pin", "inputs")
SETS = ("uniform", "clustered")
SKIP_REASON = ("PARITY UNPINNED BY THE REFERENCE: tests/golden/ghc/{set}/manifest.txt is absent.  The reference is Haskell and cannot be built "
               "in this image; tools/ghc_pin (Dump.hs + inputs) produces the dump wherever GHC 8.6 / lts-13.16 exists -- see tools/ghc_pin/README.md")


def available(which, root=DUMPS):
    return os.path.exists(os.path.join(root, which, "manifest.txt"))


def decode_png_rgb8(data: bytes) -> np.ndarray:
    """8-bit RGB (or RGBA / grey) non-interlaced PNG -> (h, w, 3) uint8, without Pillow (JuicyPixels writes plain ones)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0 and ctype in (0, 2, 6), (depth, ctype, interlace)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    raw = zlib.decompress(idat)
    stride = w * ch
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int64)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int64)
        cur = np.zeros(stride, np.int64)
        if f in (0, 2):
            cur = (line + (prev if f == 2 else 0)) & 255
        else:
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    return np.repeat(img, 3, axis=2) if ch == 1 else np.ascontiguousarray(img[:, :, :3])


class Dump:
    """One output directory of ghc-pin-dump plus the input set it was run on."""

    def __init__(self, which, root=DUMPS, inputs=INPUTS):
        self.dir, self.inputs = os.path.join(root, which), os.path.join(inputs, which)
        self.scenes, self.meta = {}, {}
        with open(os.path.join(self.dir, "manifest.txt")) as f:
            for line in f:
                t = line.split()
                if t and t[0] == "scene":
                    self.scenes[t[1]] = (int(t[2]), int(t[3]), t[4] == "bloom")
                elif t:
                    self.meta[t[0]] = t[1:]

    def catalogue_bytes(self):
        with open(os.path.join(self.inputs, "catalogue.ppm"), "rb") as f:
            return f.read()

    def config(self, name):
        """The scene as THIS repository's decoder reads the same file the reference was given."""
        import blackstar_amd as bs
        return bs.Config.from_file(os.path.join(self.inputs, "scenes", name + ".yaml"))

    def image(self, name, kind):
        w, h, _ = self.scenes[name]
        return np.fromfile(os.path.join(self.dir, f"{name}.{kind}.f64"), "<f8").reshape(h, w, 3)

    def png(self, name):
        with open(os.path.join(self.dir, name + ".png"), "rb") as f:
            return decode_png_rgb8(f.read())

    def lookup(self):
        dirs = np.fromfile(os.path.join(self.inputs, "dirs.f64"), "<f8").reshape(-1, 3)
        rgb = np.fromfile(os.path.join(self.dir, "starlookup.f64"), "<f8").reshape(-1, 3)
        inten, sat = map(float, open(os.path.join(self.inputs, "lookup.txt")).read().split())
        assert len(dirs) == len(rgb)
        return dirs, rgb, inten, sat

    def assocs(self):
        return np.fromfile(os.path.join(self.dir, "assocs.f64"), "<f8").reshape(-1, 6)

    def animation_frames(self):
        """(n, 10) cameras of Animation.generateFrames on inputs/<set>/animation.yaml (position, lookAt, upVec, fov), or None."""
        fn = os.path.join(self.dir, "animation_frames.f64")
        return np.fromfile(fn, "<f8").reshape(-1, 10) if os.path.exists(fn) else None

    def kdt_bytes(self):
        with open(os.path.join(self.dir, "stars.kdt"), "rb") as f:
            return f.read()


def compare(dump, render, star_lookup, bloom, srgb8, rtol, atol, exact_bytes=True):
    """Every scene and lookup of a dump against an implementation given as four callables (the oracle's, or the GPU library's):
    render(cfg_dict) -> (h, w, 3) f64; star_lookup(intensity, saturation, dirs) -> (n, 3); bloom(strength, divider, img); srgb8(img).
    Returns a report dict; raises AssertionError on the first quantity outside tolerance."""
    rep = {"scenes": 0, "values": 0, "bit_equal": 0, "worst_rel": 0.0}

    def close(got, want, what):
        bad = np.abs(got - want) > atol + rtol * np.abs(want)
        assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values outside {rtol:g} rel + {atol:g} abs (max abs diff {np.abs(got - want).max():.3e})"
        rep["values"] += want.size
        rep["bit_equal"] += int((got == want).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(want != 0, np.abs(got - want) / np.abs(want), 0.0)
        rep["worst_rel"] = max(rep["worst_rel"], float(rel.max()))

    dirs, want, inten, sat = dump.lookup()
    close(star_lookup(inten, sat, dirs), want, "starLookup")
    for name, (w, h, bloomed) in sorted(dump.scenes.items()):
        c = dump.config(name)
        img = render(c.to_bs_config())
        close(img, dump.image(name, "render"), f"render {name}")
        final = dump.image(name, "render")
        if bloomed:
            # bloom is compared on the REFERENCE's render, so that a last-ulp difference upstream cannot hide in (or be blamed on) it
            got = bloom(float(c.scene.bloomStrength), int(c.scene.bloomDivider), dump.image(name, "render"))
            final = dump.image(name, "bloom")
            assert np.array_equal(got, final), f"bloom {name}: not bit-exact ({int((got != final).sum())} values differ)"
        png = dump.png(name)
        mine = srgb8(final)
        diff = int((png != mine).sum())
        assert diff == 0 or not exact_bytes, f"writeImg {name}: {diff} bytes differ (max {int(np.abs(png.astype(int) - mine.astype(int)).max())} LSB)"
        rep["scenes"] += 1
    return rep


def compare_animation(dump):
    """Row f3: blackstar_amd.animation.generate_frames on the same animation file against Animation.generateFrames (src/Animation.hs:45-86).
    Linear interpolation in the same operation order: equal to the last bit is the expectation, 4e-16 relative the bar."""
    import blackstar_amd as bs
    want = dump.animation_frames()
    if want is None:
        return 0
    anim = bs.Animation.from_file(os.path.join(dump.inputs, "animation.yaml"))
    bs.validate_keyframes(anim.keyframes)
    got = np.array([list(c.camera.position) + list(c.camera.lookAt) + list(c.camera.upVec) + [c.camera.fov] for c in bs.generate_frames(anim)])
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=4e-16, atol=0)
    fn = os.path.join(dump.dir, "padzero.txt")
    if os.path.exists(fn):  # padZero mis-pads index 0 (logBase 10 0); from index 1 on it is plain zero padding
        width = len(str(len(want) - 1))
        for line in open(fn):
            i, name = line.split()
            if int(i) >= 1:
                assert name == str(int(i)).zfill(width), (i, name)
            # blackstar_amd.animation.pad_zero restates padZero with its quirks (index 0: RECALLED floor (-Infinity) :: Int): every line must agree
            assert name == bs.pad_zero(len(want) - 1, int(i)), (i, name, bs.pad_zero(len(want) - 1, int(i)))
    return len(want)
