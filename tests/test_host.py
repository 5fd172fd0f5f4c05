"""CPU tests of the host logic and of the C-ABI library (load + exported symbols; no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import blackstar_amd as bs
from blackstar_amd import _lib, synthetic
from conftest import ROOT
from oracle import scenes


def test_library_loads_and_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "blackstar_gpu.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(bs_[a-z0-9_]+)\s*\(", text))
    assert {"bs_create", "bs_render", "bs_render_device", "bs_destroy", "bs_stats", "bs_last_error"} <= declared
    L = _lib.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/blackstar_gpu.h but not exported"
    assert set(_lib.SYMBOLS) <= declared
    assert L.bs_abi_version() == _lib.BS_ABI_VERSION == 5


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.BsConfig) == 19 * 8 + 4 * 4
    assert _lib.STAR_DTYPE.itemsize == 48 and _lib.RECORD_DTYPE.itemsize == 96
    assert ctypes.sizeof(_lib.BsStats) == 11 * 8


def test_no_cpu_backend():
    L = _lib.lib()
    assert not L.bs_create(-1, None, 0)
    assert b"no CPU backend" in L.bs_last_error()


def test_scene_files_resolve_to_appendix_c():
    assert len(scenes.REFERENCE_SCENES) == 9  # every scene file the reference ships
    for name, exp in scenes.REFERENCE_SCENES.items():
        c = bs.Config.from_file(os.path.join(ROOT, "scenes", name + ".yaml"))
        assert c.to_bs_config() == exp, name
        assert (c.scene.bloomStrength, c.scene.bloomDivider) == scenes.REFERENCE_BLOOM[name], name
    c = bs.Config.from_file(os.path.join(ROOT, "scenes", "default.yaml"))
    assert (c.scene.bloomStrength, c.scene.bloomDivider) == (0.15, 25)
    assert c.with_resolution(640, 480).to_bs_config()["width"] == 640


def test_config_defaults_and_errors():
    c = bs.Config.from_yaml("camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0], fov: 1}\nscene: {}\n")
    s = c.scene  # src/ConfigFile.hs:66-79
    assert (s.stepSize, s.bloomStrength, s.bloomDivider, s.starIntensity, s.starSaturation) == (0.3, 0.4, 25, 0.7, 0.7)
    assert s.diskColor == (0.16, 0.1, 0.95) and (s.diskOpacity, s.diskInner, s.diskOuter) == (0.0, 3.0, 12.0)
    assert s.resolution == (1280, 720) and s.supersampling is False and s.safeDistance == 0.0
    for bad in ("scene: {}\n", "camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0]}\nscene: {}\n",
                "camera: {position: [1,2], lookAt: [0,0,0], upVec: [0,1,0], fov: 1}\nscene: {}\n",
                "camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0], fov: 1}\nscene: 3\n", "- 1\n", ": : :"):
        with pytest.raises(bs.ConfigError):
            bs.Config.from_yaml(bad)
    # numbers the reference's decoder accepts but PyYAML's YAML 1.1 resolver leaves as strings / floats
    c3 = bs.Config.from_yaml("camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0], fov: 1e0}\n"
                             "scene: {stepSize: 1e-1, resolution: [1920.0, 1.08e3], bloomDivider: 2.5e1, diskOuter: 12}\n")
    assert c3.scene.stepSize == 0.1 and c3.scene.resolution == (1920, 1080) and c3.scene.bloomDivider == 25 and c3.camera.fov == 1.0
    # ... and what it refuses: non-integral Ints, non-numbers, QUOTED numbers (aeson yields String for them), infinite Ints
    for bad in ("scene: {resolution: [1920.5, 1080]}", "scene: {stepSize: abc}", "scene: {bloomDivider: 2.5}", 'scene: {stepSize: "0.3"}',
                "scene: {resolution: ['1920', 1080]}", 'scene: {stepSize: "1e-1"}', "scene: {resolution: [1e999, 5]}", "scene: {bloomDivider: .inf}"):
        with pytest.raises(bs.ConfigError):
            bs.Config.from_yaml("camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0], fov: 1}\n" + bad + "\n")
    c2 = bs.Config.from_yaml(c.to_yaml())  # ToJSON round trip (hue * 360 and back)
    assert c2.to_bs_config() == pytest.approx(c.to_bs_config())
    p = bs.prepare_scene(bs.Config.from_file(os.path.join(ROOT, "scenes", "default.yaml")), True)  # app/Main.hs:93-103
    assert p.scene.resolution == (300, 168) and p.scene.supersampling is False and p.scene.bloomStrength == 0


def test_quoted_fov_is_rejected_like_aeson_does():
    with pytest.raises(bs.ConfigError):
        bs.Config.from_yaml('camera: {position: [1,2,3], lookAt: [0,0,0], upVec: [0,1,0], fov: "1.5"}\nscene: {}\n')


def test_animation_frames_match_independent_restatement():
    a = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
    assert a.nFrames == 375 and len(a.keyframes) == 2
    assert a.scene.diskColor == (0.16, 0.1, 0.95)  # 'diskHSV' is not a parsed key -> default
    bs.validate_keyframes(a.keyframes)
    a.nFrames = 600
    frames = bs.generate_frames(a)
    assert len(frames) == 600
    for i in (0, 1, 299, 300, 598, 599):
        assert frames[i].to_bs_config() == scenes.ani_frame(i, 600)
    with pytest.raises(bs.ConfigError):
        bs.validate_keyframes(a.keyframes[:1])
    a.keyframes[1].time = 0.9
    with pytest.raises(bs.ConfigError):
        bs.validate_keyframes(a.keyframes)


def test_frame_file_names_follow_the_references_pad_zero():
    """Row f3's file naming (app/Animate.hs:55-56 over src/Util.hs:43-48 padZero): digits counted with floor (logBase 10 x) + 1 in Double /
    Int, which pads like zfill from index 1 on, leaves index 0 unpadded (logBase 10 0 = -Infinity: SURVEY Appendix F.7, recalled Int
    semantics) and counts one digit short where log x / log 10 falls below the integer (1000).  The default names of write_animation /
    render_animation are plainly zero-padded; reference_names=True gives these."""
    import math
    from blackstar_amd.distributed import _frame_namer
    assert [bs.pad_zero(599, i) for i in (0, 1, 9, 10, 99, 100, 599)] == ["0", "001", "009", "010", "099", "100", "599"]
    for mx in (1, 9, 10, 99, 374, 599, 998):
        for v in range(1, mx + 1, max(1, mx // 40)):
            assert bs.pad_zero(mx, v) == str(v).zfill(len(str(mx))), (mx, v)
        assert bs.pad_zero(mx, 0) == "0"
    assert bs.pad_zero(0, 0) == "0"                                   # a one-frame animation
    assert math.log(1000.0) / math.log(10.0) < 3                     # the quirk's premise holds in this libm too
    assert bs.pad_zero(1000, 5) == "005" and bs.pad_zero(9999, 1000) == "01000" and bs.pad_zero(999, 5) == "005"
    assert bs.frame_file_name("default-ani", 600, 0) == "default-ani_0.yaml" and bs.frame_file_name("default-ani", 600, 42, ".png") == "default-ani_042.png"
    ours, theirs = _frame_namer(600, "ani", False), _frame_namer(600, "ani", True)
    assert ours(0) == "ani_000.png" and theirs(0) == "ani_0.png" and all(ours(i) == theirs(i) for i in range(1, 600))
    # both namings keep the batch loop's lexicographic order (app/Main.hs:68-70 sorts the directory listing) equal to frame order
    for namer in (ours, theirs):
        names = [namer(i) for i in range(600)]
        assert sorted(names) == names


def test_animate_compatible_frame_files_round_trip(tmp_path):
    """SURVEY 7.7 "animate-compatible frame files": write_frame_files leaves what `animate --force` leaves (app/Animate.hs:47-61) -- one scene
    file per frame under the reference's names -- and reading them back (the decoder the directory mode uses) gives the frames' configs
    bit for bit: floats survive the YAML text (repr round trip), the disk hue goes out in degrees and comes back divided by 360."""
    a = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
    a.nFrames = 12
    paths = bs.write_frame_files(a, str(tmp_path / "frames"), "default-ani")
    assert [os.path.basename(p) for p in paths] == ["default-ani_0.yaml"] + [f"default-ani_{i:02d}.yaml" for i in range(1, 12)]
    assert sorted(os.listdir(tmp_path / "frames")) == sorted(os.path.basename(p) for p in paths)
    frames = bs.generate_frames(a)
    for p, want in zip(paths, frames):
        got = bs.Config.from_file(p)
        assert got.camera == want.camera, p
        assert got.to_bs_config() == want.to_bs_config(), p
        # hue x 360 / 360 must come back exactly for the value the reference's scenes use (0.16 is not representable; 360 * h is rounded once each way)
        assert got.scene.diskColor[1:] == want.scene.diskColor[1:] and abs(got.scene.diskColor[0] - want.scene.diskColor[0]) <= 2.8e-17
    a.keyframes = a.keyframes[:1]
    with pytest.raises(bs.ConfigError):
        bs.write_frame_files(a, str(tmp_path / "bad"), "x")


def test_synthetic_catalogue_layout_and_reader(catalogue_bytes, oracle):
    data = synthetic.ppm_catalogue_bytes(2000, synthetic.SEED + 1)
    assert data == catalogue_bytes  # vectorised generator == the scalar one that made the fixture
    stars = bs.read_map(data)
    assert len(stars) == 2000
    assert stars.tobytes() == oracle.read_ppm(data).tobytes()  # product reader vs oracle reader, bit for bit
    g = np.load(os.path.join(ROOT, "tests", "golden", "catalogue_2000_parsed.npz"))["stars"]
    for i, k in enumerate(("x", "y", "z", "hue", "sat")):
        assert np.array_equal(stars[k], g[:, i])
    assert np.array_equal(stars["mag"], g[:, 5].astype(np.int32))
    np.testing.assert_allclose(stars["x"] ** 2 + stars["y"] ** 2 + stars["z"] ** 2, 1.0, rtol=1e-15)
    assert (stars["mag"] >= 500).all() and (stars["mag"] <= 1200).all()
    with pytest.raises(bs.BlackstarError if hasattr(bs, "BlackstarError") else Exception):
        bs.read_map(b"too short")
    assert len(bs.read_map(bytes(28))) == 0


def test_host_hsi_matches_oracle(oracle):
    L = _lib.lib()
    out = np.zeros(3)
    for h, s, i in ((0.5, 0.1, 1.05), (0.16, 0.1, 0.95), (0.631, 0.585, 0.4), (0.0, 0.0, 0.3), (0.999, 0.5, 0.2)):
        assert L.bs_hsi_to_rgb(h, s, i, out.ctypes.data) == 0
        assert np.array_equal(out, oracle.hsi_to_rgb(h, s, i))
    assert L.bs_hsi_to_rgb(1.0, 0.1, 0.5, out.ctypes.data) == -1
    assert L.bs_hsi_to_rgb(-0.1, 0.1, 0.5, out.ctypes.data) == -1


def test_tree_serialisation_roundtrip(tmp_path, catalogue_bytes):
    stars = bs.read_map(catalogue_bytes)
    blob = bs.tree_to_byte_string(stars)
    assert blob[:5] == b"BSKD1" and len(blob) == 16 + 48 * len(stars)
    p = tmp_path / "stars.bskd"
    p.write_bytes(blob[:-1])
    with pytest.raises(Exception):
        bs.read_tree_from_file(str(p))


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    """The C++ host mirror builds with plain g++ against the C ABI; without a HIP device it must fail, not fall back."""
    import subprocess
    exe = tmp_path / "host_render"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_render.cpp"),
                           "-o", str(exe), "-L" + os.path.join(ROOT, "blackstar_amd"), "-lblackstar_gpu",
                           "-Wl,-rpath," + os.path.join(ROOT, "blackstar_amd")])
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_parity.py::test_cpp_host_mirror")
    r = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "catalogue_2000.ppm"), str(tmp_path / "o.f64")], capture_output=True, text=True)
    assert r.returncode == 1 and "bs_create" in r.stderr
    assert not (tmp_path / "o.f64").exists()


def test_kdt_file_roundtrip(catalogue_bytes):
    """best-effort `stars.kdt` layout (recalled, unverified against a real file): write -> read round trip."""
    from blackstar_amd import kdt_file
    rec = np.frombuffer(catalogue_bytes, np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]), offset=28)[:300]
    stars = bs.read_map(catalogue_bytes)[:300]
    pos = np.stack([stars["x"], stars["y"], stars["z"]], axis=1)
    blob = kdt_file.write_kdt(pos, rec["mag"].astype(int), "".join(chr(c) for c in rec["sp"]))
    back = kdt_file.read_kdt(blob)
    assert len(back) == 300
    key = lambda s: sorted(map(tuple, np.stack([s["x"], s["y"], s["z"], s["hue"], s["sat"], s["mag"].astype(float)], axis=1).tolist()))
    assert key(back) == key(stars)  # same star SET with starColor' applied (file order is the tree's in-order)
    for bad in (blob[:-3], blob[:40], b"\x00\x00\x07", b""):
        with pytest.raises(kdt_file.KdtDecodeError):
            kdt_file.read_kdt(bad)


def test_header_is_c99_and_layouts_match_the_shim(tmp_path):
    """include/blackstar_gpu.h compiled as C99 (-pedantic); struct offsets are the ones INTEGRATION.md's Haskell shim pokes."""
    import subprocess
    exe = tmp_path / "abi_check"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_check.c"), "-o", str(exe), "-L" + os.path.join(ROOT, "blackstar_amd"),
                           "-lblackstar_gpu", "-Wl,-rpath," + os.path.join(ROOT, "blackstar_amd")])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "abi ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


BAD_CONFIGS = {  # what -> (override, substring of the message).  src/Raytracer.hs:80-86 never terminates on any of these.
    "NaN camera position": (dict(cam_pos=(0.0, float("nan"), -20.0)), "camera.position"),
    "infinite lookAt": (dict(cam_lookat=(float("inf"), 0.0, 0.0)), "camera.lookAt"),
    "NaN up vector": (dict(cam_up=(0.0, 1.0, float("nan"))), "camera.upVec"),
    "NaN fov": (dict(fov=float("nan")), "camera.fov"),
    "zero stepSize": (dict(step_size=0.0), "stepSize"),
    "negative stepSize": (dict(step_size=-0.3), "stepSize"),
    "NaN stepSize": (dict(step_size=float("nan")), "stepSize"),
    "infinite stepSize": (dict(step_size=float("inf")), "stepSize"),
    "lookAt == position": (dict(cam_lookat=(0.0, 1.0, -20.0)), "lookAt equals"),
    "lookAt 1e-7 from position": (dict(cam_lookat=(1e-7, 1.0, -20.0)), "lookAt equals"),
    "bad hue": (dict(disk_hsi=(1.0, 0.1, 1.0)), "not properly scaled"),
    "zero width": (dict(width=0), "resolution"),
    "more than 2^28 pixels": (dict(width=32768, height=16384), "resolution too large"),
}


# Configurations the reference DOES render: radii enter only squared (src/Raytracer.hs:61-62), safeDistance depends on the camera
# alone (:59-60), so non-finite disk / star parameters never keep a ray from ending.  They validate, and render like the oracle's.
ODD_BUT_RENDERABLE = {
    "negative diskInner": dict(disk_inner=-1.8),
    "negative diskOuter": dict(disk_outer=-13.0),
    "both radii negative": dict(disk_inner=-1.8, disk_outer=-13.0),
    "infinite diskOuter": dict(disk_outer=float("inf")),
    "NaN diskInner": dict(disk_inner=float("nan")),
    "infinite starIntensity": dict(star_intensity=float("inf")),
    "NaN starSaturation": dict(star_saturation=float("nan")),
    "NaN diskOpacity": dict(disk_opacity=float("nan")),
    "infinite disk intensity": dict(disk_hsi=(0.5, 0.1, float("inf"))),
}


@pytest.mark.parametrize("what", sorted(ODD_BUT_RENDERABLE))
def test_validate_config_accepts_what_the_reference_renders(what):
    import ctypes as C
    L = _lib.lib()
    assert L.bs_validate_config(C.byref(_lib.make_config(dict(scenes.with_res(scenes.DEFAULT, 8, 8), **ODD_BUT_RENDERABLE[what])))) == 0, L.bs_last_error()


@pytest.mark.parametrize("what", sorted(BAD_CONFIGS))
def test_validate_config_rejects_what_the_reference_never_returns_from(what):
    """bs_validate_config is host-only: the checks every render entry point applies before any GPU work."""
    import ctypes as C
    L = _lib.lib()
    assert L.bs_validate_config(C.byref(_lib.make_config(scenes.with_res(scenes.DEFAULT, 8, 8)))) == 0
    over, msg = BAD_CONFIGS[what]
    assert L.bs_validate_config(C.byref(_lib.make_config(dict(scenes.with_res(scenes.DEFAULT, 8, 8), **over)))) == -1
    assert msg.encode() in L.bs_last_error(), L.bs_last_error()
    assert L.bs_validate_config(None) == -1
    assert L.bs_validate_config(C.byref(_lib.make_config(dict(scenes.DEFAULT, width=16384, height=16384)))) == 0  # 2^28 pixels: the largest frame accepted


def test_every_shipped_scene_and_animation_frame_validates():
    import ctypes as C
    L = _lib.lib()
    for name in scenes.REFERENCE_SCENES:
        c = bs.Config.from_file(os.path.join(ROOT, "scenes", name + ".yaml"))
        assert L.bs_validate_config(C.byref(_lib.make_config(c.to_bs_config()))) == 0, name
    anim = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
    for c in bs.generate_frames(anim)[::25]:
        assert L.bs_validate_config(C.byref(_lib.make_config(c.to_bs_config()))) == 0


def test_partition_trial_decision_rule():
    """Round 4: the CU partition of bs_render_rgb8_batch / bs_render_png_batch is decided by MEASUREMENT (csrc/batch.cpp: a trial of 8 + 3 x 8
    frames per frame shape and context), not by a model.  Host-only: the rule that turns the trial's three per-frame times into the choice
    -- the fastest, but a partition only if it beats the shared chip by more than 1.5 % (what a segment of eight frames resolves).  The times below are
    rounds 2-3's measured A/B results (profiles/r03_post_partition_ab.txt, r03_partition_large_ab.jsonl, r03_png_partition_ab.jsonl)."""
    import ctypes as C
    D = _lib.debug_lib()

    def pick(shared, m8, m16):
        ms = (C.c_double * 3)(shared, m8, m16)
        cus = (C.c_int * 3)(0, 8, 16)
        return D.bs_debug_pick_partition(ms, cus, 3)
    assert pick(4.67, 4.28, 4.45) == 8          # C3 at 1080p
    assert pick(2.25, 2.26, 2.00) == 16         # 720p: the post stage is the bottleneck on 8 CUs
    assert pick(4.72, 7.10, 4.47) == 16         # bloomDivider 10 (r = 192)
    assert pick(17.83, 18.60, 17.39) == 16      # 3840x2160
    assert pick(19.90, 19.23, 19.50) == 8       # lensing-disk at 4K: the longer trace hides the post stage on 8 CUs
    assert pick(1.33, 2.30, 4.00) == 0          # no supersampling: too cheap to trace per pixel
    assert pick(10.71, 10.59, 10.80) == 0       # C3 in STRICT: 1.1 % is inside what the trial resolves -> do nothing
    assert pick(4.73, 7.40, 4.41) == 16         # C3 as PNG files
    assert pick(4.00, 3.99, 3.98) == 0 and pick(4.00, 3.93, 4.2) == 8   # the margin: 0.5 % is not a reason, 1.75 % is
    assert pick(4.67, 0.0, 0.0) == 0 and pick(0.0, 4.2, 4.1) == 16       # segments that did not run do not count
    assert D.bs_debug_pick_partition(None, None, 3) == -1 and D.bs_debug_partition_choice(None, None, 0.1, 25, 0, None) == -1


def test_product_library_exports_the_stable_abi_only():
    """the test hooks are not in the product.  libblackstar_gpu.so exports exactly the functions include/blackstar_gpu.h
    declares -- no bs_debug_*, no bs_trace_rays -- and libblackstar_gpu_debug.so exports exactly those of include/blackstar_gpu_debug.h,
    needs the product library and finds it next to itself."""
    import re
    import subprocess

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-2:-1] == ["T"] and ln.split()[-1].startswith("bs_")}

    def declared(header):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        return set(re.findall(r"\b(bs_[a-z0-9_]+)\s*\(", text))
    _lib.lib()
    prod, dbg = exported(_lib.SO_PATH), exported(_lib.DEBUG_SO_PATH)
    assert prod == declared("blackstar_gpu.h") == set(_lib.SYMBOLS), (prod ^ declared("blackstar_gpu.h"), prod ^ set(_lib.SYMBOLS))
    assert not [s_ for s_ in prod if "debug" in s_ or s_ == "bs_trace_rays"]
    assert dbg == declared("blackstar_gpu_debug.h") - declared("blackstar_gpu.h") == set(_lib.DEBUG_SYMBOLS), dbg ^ set(_lib.DEBUG_SYMBOLS)
    dyn = subprocess.run(["readelf", "-d", _lib.DEBUG_SO_PATH], capture_output=True, text=True, check=True).stdout
    assert "libblackstar_gpu.so" in dyn and "$ORIGIN" in dyn
    assert _lib.debug_lib().bs_debug_abi_check() == 0
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert maps.count("libblackstar_gpu.so") >= 1 and len({ln.split()[-1] for ln in maps.splitlines() if ln.endswith("libblackstar_gpu.so")}) == 1


def test_stale_library_is_refused_by_its_abi_version(tmp_path):
    """A libblackstar_gpu.so of another ABI version under the same name fails at load with a message that says so (not at a later
    symbol lookup or, worse, a struct read)."""
    import subprocess
    import sys
    src = tmp_path / "stale.c"
    src.write_text("int bs_abi_version(void) { return 1; }\n")
    so = tmp_path / "libblackstar_gpu.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    code = "import blackstar_amd as bs\ntry:\n    bs._lib.lib()\n    print('LOADED')\nexcept bs._lib.BlackstarError as e:\n    print('refused', 'ABI version 1' in str(e))\n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, BLACKSTAR_LIB=str(so)))
    assert r.returncode == 0 and r.stdout.strip() == "refused True", (r.stdout, r.stderr)


def test_missing_native_library_fails_loudly():
    """No eager / CPU fallback: with the shared object absent every product entry point raises."""
    import subprocess
    import sys
    code = ("import blackstar_amd as bs, numpy as np\n"
            "for f in (lambda: bs.read_map(bytes(56)), lambda: bs.StarTree(None), lambda: bs.bloom(0.1, 2, np.zeros((4, 4, 3)))):\n"
            "    try:\n        f()\n        print('NO ERROR')\n    except bs._lib.BlackstarError as e:\n        print('raised', 'is missing' in str(e))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, BLACKSTAR_LIB="/nonexistent/libblackstar_gpu.so"))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:3] == ["raised True"] * 3, r.stdout


# ---- star direction grid (csrc/star_index.cpp), host side ---------------------------------------------------------------

G, DELTA, RADIUS = 256, 0.0039, 0.0015  # kGridG, kGridDelta, kStarRadius of csrc/bs_internal.h


def _star_grid(stars):
    L = _lib.lib()
    cs = np.zeros(6 * G * G + 2, np.uint32)
    cap = 4 * len(stars) + 8
    ent = np.zeros(cap, np.int32)
    n = _lib.debug_lib().bs_debug_star_grid(stars.ctypes.data if len(stars) else None, len(stars), cs.ctypes.data, ent.ctypes.data, cap)
    assert 0 <= n <= cap
    return cs, ent[:n]


def _cell(t):
    c = (t + 1.0) * (0.5 * G)
    return int(min(max(np.floor(c), 0), G - 1)) if c > 0 else 0


def _grid_query(cs, ent, q):
    """The kernel's addressing (star_lookup, csrc/trace_kernel.hip) replayed with numpy: candidate star indices of unit q."""
    axis = 0 if (abs(q[0]) >= abs(q[1]) and abs(q[0]) >= abs(q[2])) else (1 if abs(q[1]) >= abs(q[2]) else 2)
    m, a, b = q[axis], q[(axis + 1) % 3], q[(axis + 2) % 3]
    face = 2 * axis + (1 if m < 0 else 0)
    u, v = a / abs(m), b / abs(m)
    iu0, iu1, iv0, iv1 = _cell(u - DELTA), _cell(u + DELTA), _cell(v - DELTA), _cell(v + DELTA)
    assert iu1 - iu0 <= 1 and iv1 - iv0 <= 1  # at most 2 x 2 cells: 2 * DELTA <= cell width
    out = []
    for iv in range(iv0, iv1 + 1):
        row = (face * G + iv) * G
        out.extend(ent[cs[row + iu0]:cs[row + iu1 + 1]])
    return out


def test_star_grid_builder_and_query_geometry_on_cpu():
    """Every star within the radius of a unit query must be among the candidates of the query's own face, exactly once
    -- including at face edges/corners (border copies), for off-sphere stars, and with the margin kGridDelta."""
    rng = np.random.default_rng(3)
    stars = bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_SMALL))
    extra = np.zeros(64, _lib.STAR_DTYPE)
    s3, s2 = 1 / np.sqrt(3), 1 / np.sqrt(2)
    pts = [[sx * s3, sy * s3, sz * s3] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    pts += [[s2, s2, 0], [s2, 0, -s2], [0, -s2, s2], [-s2, s2, 1e-9], [0.9986 * s2, 0.9986 * s2, 0], [0, 0, 0], [1e-4, 0, 0], [np.nan, 0, 0]]
    extra["x"][:len(pts)], extra["y"][:len(pts)], extra["z"][:len(pts)] = np.array(pts).T
    extra["x"][len(pts):] = 1.0  # the rest: copies of (1,0,0)
    stars = np.concatenate([stars, extra])
    cs, ent = _star_grid(stars)
    assert np.all(np.diff(cs.astype(np.int64)) >= 0) and cs[-1] == len(ent)
    xyz = np.stack([stars["x"], stars["y"], stars["z"]], axis=1)
    fin = np.isfinite(xyz).all(axis=1)
    nz = fin & (np.abs(xyz).sum(axis=1) > 0)
    counts = np.bincount(ent[:cs[6 * G * G]], minlength=len(stars))
    assert np.all(counts[nz] >= 1) and np.all(counts <= 3) and np.all(counts[~fin] == 0)  # own face (+ up to 2 neighbours)
    origin = set(ent[cs[6 * G * G]:cs[6 * G * G + 1]])
    r2all = (xyz * xyz).sum(axis=1)
    with np.errstate(invalid="ignore"):
        assert origin == set(np.nonzero(fin & (r2all <= 0.0016 ** 2))[0])
    # queries: at every star, offset by up to 2 radii in a random direction, then normalised
    base = xyz[nz]
    d = rng.normal(size=base.shape); d /= np.linalg.norm(d, axis=1)[:, None]
    qs = base / np.linalg.norm(base, axis=1)[:, None] + rng.uniform(0, 2 * RADIUS, (len(base), 1)) * d
    qs /= np.linalg.norm(qs, axis=1)[:, None]
    hits = 0
    for q in qs:
        cand = _grid_query(cs, ent, q)
        assert len(cand) == len(set(cand)), "a star is listed twice within one face"
        with np.errstate(invalid="ignore"):
            inside = set(np.nonzero(((xyz - q) ** 2).sum(axis=1) <= RADIUS * RADIUS)[0])
        assert inside <= set(cand), f"query {q}: in-radius stars {inside - set(cand)} are not candidates"
        hits += len(inside)
    assert hits > len(qs) // 3


def test_star_grid_margin_bound():
    """The bound behind kGridDelta: on a face (|u|,|v| <= 1 + D) a direction change of asin(r) moves u by less than D."""
    rng = np.random.default_rng(4)
    n = 200000
    uv = rng.uniform(-1.0, 1.0, (n, 2))
    uv[: n // 4] = np.sign(uv[: n // 4]) * rng.uniform(0.99, 1.0, (n // 4, 2))  # corners
    p = np.concatenate([uv, np.ones((n, 1))], axis=1)
    p /= np.linalg.norm(p, axis=1)[:, None]
    t = rng.normal(size=(n, 3)); t -= (t * p).sum(axis=1)[:, None] * p; t /= np.linalg.norm(t, axis=1)[:, None]
    th = np.arcsin(RADIUS)
    s = np.cos(th) * p + np.sin(th) * t  # directions exactly asin(r) away
    du = np.abs(s[:, 0] / s[:, 2] - uv[:, 0]); dv = np.abs(s[:, 1] / s[:, 2] - uv[:, 1])
    assert max(du.max(), dv.max()) < DELTA * 0.97, (du.max(), dv.max())
    assert 2 * DELTA <= 2.0 / G


def test_srgb8_threshold_table_is_the_oracles_pixel_map(oracle):
    """The 255 thresholds the device compares against (bs_debug_srgb8_table, built with the host's libm): each one is exactly
    where the oracle's toWord8 . sRGB steps from k-1 to k, and the map is monotone on a dense sample in between."""
    import ctypes as C
    T = np.zeros(257)
    assert _lib.debug_lib().bs_debug_srgb8_table(T.ctypes.data_as(C.c_void_p)) == 0
    assert T[0] == -np.inf and T[256] == np.inf and (np.diff(T[1:256]) > 0).all()
    k = np.arange(1, 256)
    assert np.array_equal(oracle.srgb8(T[1:256]), k.astype(np.uint8))
    assert np.array_equal(oracle.srgb8(np.nextafter(T[1:256], -np.inf)), (k - 1).astype(np.uint8))
    x = np.sort(np.random.default_rng(5).uniform(0, 1.05, 400000))
    b = oracle.srgb8(x)
    assert (np.diff(b.astype(int)) >= 0).all()
    assert np.array_equal(b, (np.searchsorted(T[1:256], x, side="right")).astype(np.uint8))


def test_bench_cli_contract_without_a_gpu():
    """bench.py: the flags the driver passes exist, N > 1 without a launcher is accepted (self-launching), and on a box
    without a HIP device every form fails loudly instead of falling back to anything."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--launcher", "--gather", "--mode", "--workload"):
        assert flag in out.stdout
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device behaviour cannot be observed here")
    for args in (["--steps", "1", "--warmup", "0"], ["--gpus", "2", "--steps", "1", "--warmup", "0"]):
        r = subprocess.run([sys.executable, bench] + args, capture_output=True, text=True, timeout=300,
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
        assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_batch_wrappers_check_their_arguments_before_the_library():
    """The Python mirrors of the batch entry points refuse malformed calls on the host (no device needed): no trees, dict configs where
    the scene's bloom parameters are needed, buffers of the wrong shape / count, a path list of the wrong length."""
    import numpy as np

    import blackstar_amd as bs
    cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml")).with_resolution(16, 8)
    fake_tree = object()
    for fn in (bs.render_rgb8_batch, bs.render_png_batch):
        with pytest.raises(ValueError):
            fn([cfg], [])
        with pytest.raises(TypeError):
            fn([cfg.to_bs_config()], [fake_tree])
    with pytest.raises(ValueError):
        bs.render_rgb8_batch([cfg], [fake_tree], outs=[np.zeros((8, 16, 4), np.uint8)])
    with pytest.raises(ValueError):
        bs.render_png_batch([cfg, cfg], [fake_tree], outs=[np.zeros(10000, np.uint8)])
    with pytest.raises(ValueError):
        bs.render_png_batch([cfg], [fake_tree], outs=[np.zeros((10, 1000), np.uint8)])
    with pytest.raises(ValueError):
        bs.render_png_files([cfg, cfg], [fake_tree], ["only-one.png"])
    with pytest.raises(TypeError):
        bs.render_png_files([cfg.to_bs_config()], [fake_tree], ["a.png"])
    with pytest.raises(ValueError):
        bs.render_png_files([cfg], [], ["a.png"])
    # single-frame PNG wrappers: `out` goes to C as (pointer, size), so it must be `size` contiguous writeable bytes
    img = np.zeros((8, 16, 3), np.uint8)
    big = np.zeros(4 * bs.png_bound(8, 16), np.uint8)
    for bad in (big[::2], big.reshape(4, -1), big.astype(np.int8), np.zeros((2, 3)), memoryview(bytearray(10000))):
        with pytest.raises(ValueError, match="flat, C-contiguous"):
            bs.encode_png(img, fake_tree, out=bad)
        with pytest.raises(ValueError, match="flat, C-contiguous"):
            bs.render_png(cfg, fake_tree, out=bad)
    ro = np.zeros(bs.png_bound(8, 16), np.uint8)
    ro.setflags(write=False)
    with pytest.raises(ValueError, match="writeable"):
        bs.encode_png(img, fake_tree, out=ro)
    assert bs.png_bound(8, 16) == 8 * (3 * 16 + 1) + 47 + 33 + 17   # pixels + filter bytes + fixed chunks + one block's stored-header and chunk framing


def test_validate_config_property():
    """hypothesis: bs_validate_config on arbitrary bit patterns of every double and int field never crashes, and says OK only for
    configurations the kernels terminate on -- camera and stepSize finite, stepSize > 0, lookAt away from position, a positive
    resolution that fits, hue inside [0, 1) -- and for ALL of those that are otherwise well-formed (radii, opacity, star parameters are free)."""
    import ctypes as C
    import math

    from hypothesis import given, settings
    from hypothesis import strategies as st
    L = _lib.lib()
    doubles = st.one_of(st.floats(allow_nan=True, allow_infinity=True), st.sampled_from([0.0, -0.0, 0.3, 1.0, 1e-320, 1e300, -1.0, 12.0]))
    ints = st.one_of(st.integers(-2**31, 2**31 - 1), st.sampled_from([0, 1, 2, 96, 1920, 1080, 65536]))

    @settings(max_examples=300, deadline=None)
    @given(st.lists(doubles, min_size=19, max_size=19), st.lists(ints, min_size=3, max_size=3))
    def check(d, i):
        c = _lib.BsConfig()
        c.cam_pos[:] = d[0:3]; c.cam_lookat[:] = d[3:6]; c.cam_up[:] = d[6:9]
        c.fov, c.step_size, c.star_intensity, c.star_saturation = d[9:13]
        c.disk_hsi[:] = d[13:16]
        c.disk_opacity, c.disk_inner, c.disk_outer = d[16:19]
        c.width, c.height, c.supersampling = i
        rc = L.bs_validate_config(C.byref(c))
        assert rc in (0, -1)
        q = sum((a - b) * (a - b) for a, b in zip(d[0:3], d[3:6])) if all(math.isfinite(x) for x in d[0:6]) else float("nan")
        h = d[13] * 2 * math.pi
        ok = (all(math.isfinite(x) for x in d[0:11]) and d[10] > 0 and q > 1e-12 and c.width > 0 and c.height > 0 and c.width * c.height <= 1 << 28 and
              0 <= h < 2 * math.pi)
        if rc == 0:
            assert all(math.isfinite(x) for x in d[0:11]) and d[10] > 0 and q > 1e-12
            assert c.width > 0 and c.height > 0 and 0 <= d[13] < 1
        else:
            assert _lib.last_error()
        if ok and abs(q - 1e-12) > 1e-20:   # (away from the rounding of the quadrance itself)
            assert rc == 0, _lib.last_error()

    check()


def test_no_cpp_exception_can_cross_the_c_abi():
    """The header promises "never throws across the ABI".  Structural check: every entry point of the product library that has a body of
    more than one line is a function-try-block ending in bs::abi_exception (std::bad_alloc -> BS_ENOMEM, anything else -> BS_EINTERNAL,
    message in bs_last_error), and the threads the batch entry points start cannot throw out of them either."""
    import re
    csrc = os.path.join(ROOT, "blackstar_amd", "csrc")
    guarded, bare = set(), set()
    for fn in ("context.cpp", "render.cpp", "post.cpp", "batch.cpp", "host_math.cpp"):
        lines = open(os.path.join(csrc, fn)).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r'^(?:extern "C" )?(?:int|long|void \*|bs_ctx \*|void|const char \*) ?(bs_\w+)\(', ln)
            if not m or ln.rstrip().endswith(";"):
                continue
            if ln.rstrip().endswith("}"):   # one-line accessors: nothing in them can throw
                bare.add(m.group(1))
                continue
            j = next(k for k in range(i, i + 6) if lines[k] in ("{", "try {"))
            assert lines[j] == "try {", f"{fn}: {m.group(1)} is not a function-try-block"
            end = next(k for k in range(j, len(lines)) if lines[k].startswith("}"))
            assert "bs::abi_exception(\"%s\")" % m.group(1) in lines[end], f"{fn}: {m.group(1)}"
            guarded.add(m.group(1))
    assert guarded | bare == set(_lib.SYMBOLS), (guarded | bare) ^ set(_lib.SYMBOLS)
    assert bare <= {"bs_abi_version", "bs_last_error", "bs_get_mode", "bs_numa_node"}   # (one-line field reads: nothing in them can throw)
    batch = open(os.path.join(csrc, "batch.cpp")).read()
    assert batch.count("catch (const std::system_error &)") >= 2 and "th.emplace_back(body, c)" in batch   # thread creation failures are handled


def test_stepping_loops_are_what_the_roofline_counts():
    """d: `roofline.valu_issue_frac` prices a wavefront step at LOOP_VALU (bench_legs.py) issue slots.  The compiler's own assembly says what a step
    is: the FAST loop (csrc/fast_loop_asm.h) is two steps in line with no v_mov in them (plus an out-of-line block per step), the STRICT loop two blocks of one step each; and the
    Makefile's PAD (scripts/pick_pad.py) puts the FAST loop's head at the start of a 32-byte fetch window.  hipcc cross-compiles: no GPU."""
    import shutil
    import subprocess
    import sys
    import bench_legs
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")

    def blocks(*flags):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/isa_hot_blocks.py"), *flags], capture_output=True, text=True, check=True).stdout
        return [(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
                for m in re.finditer(r"^(\S+)\s+depth \S+\s+VALU\s+(\d+) \(f64\s+(\d+), of them transcendental\s+(\d+); v_mov (\d+)\)", out, re.M)]

    every = blocks()
    fast = [b for b in every if b[0] == "FAST_LOOP_IN_LINE"]  # .Lbs_loop and its join blocks added up (scripts/isa_hot_blocks.py)
    assert len(fast) == 1, fast
    lv = bench_legs.LOOP_VALU["fast"]
    assert fast[0][1:] == (2 * (lv["full_rate"] + lv["quarter_rate"]), 2 * (lv["full_rate"] + lv["quarter_rate"]), 2 * lv["quarter_rate"], 0), fast
    slow = [b for b in every if b[0].startswith(".Lbs_slow")]  # the out-of-line stage 1 (own v_rsq_f64), one per copy of the step
    assert len(slow) == 4 and all(b[1:] == (8, 8, 1, 0) for b in slow), slow  # stages 1 and 3, two copies of the step
    lv = bench_legs.LOOP_VALU["strict"]
    strict = [b for b in blocks("--strict") if b[1] == lv["full_rate"] + lv["quarter_rate"] and b[3] == lv["quarter_rate"]]
    assert len(strict) == 2, strict  # the loop is unrolled by two
    show = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/pick_pad.py"), "--show"], capture_output=True, text=True, check=True)
    assert re.search(r"PAD (\d+) puts the head at offset 0\b", show.stderr), show.stderr
    assert 0 <= int(show.stdout) <= 7


def test_division_by_the_resolution_in_three_instructions_is_ieee_division(tmp_path):
    """a3: generate_ray divides by the traced width and height (Raytracer.hs:45-46) as q = a y, r = fma(-b, q, a), fma(r, y, q) with y = 1.0 / b
    from the host (csrc/trace_device.h div_by).  That is the correctly rounded quotient whenever y is the correctly rounded reciprocal and b's
    significand is not all ones -- checked here against the CPU's own division (same IEEE fma as the device's v_fma_f64): every pixel index over
    every width up to 4096 and the large ones, and random numerators of every sign and magnitude generate_ray's second division sees."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "divby.c"
    src.write_text(r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
static double div_by(double a, double b, double y) { double q = a * y; double r = fma(-b, q, a); return fma(r, y, q); }
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t nxt(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
    long bad = 0, n = 0;
    static const int big[] = {5120, 5760, 7680, 8192, 15360, 16384, 30720, 32768, 65535, 65536};
    for (int k = 0; k < 4096 + 10; k++) {
        const int W = k < 4096 ? k + 1 : big[k - 4096];
        const double b = W, y = 1.0 / b;
        for (int x = 0; x < W; x++, n++) bad += div_by((double)x, b, y) != (double)x / b;
        for (int j = 0; j < 2000; j++, n++) {
            const double m = (double)(nxt() >> 11) * 0x1p-53 + 0.5;
            const double a = ldexp(m, (int)(nxt() % 61) - 30) * ((nxt() & 1) ? 1 : -1);
            bad += div_by(a, b, y) != a / b;
        }
    }
    printf("%ld %ld\n", n, bad);
    return 0;
}
""")
    exe = tmp_path / "divby"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"], check=True)
    n, bad = (int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert n > 15_000_000 and bad == 0, (n, bad)
