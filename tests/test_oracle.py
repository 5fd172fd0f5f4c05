"""CPU tests that pin the oracle (the reference has no tests and cannot be built here -- 'parity unpinned'):
C restatement vs golden vectors from the independent numpy restatement, vs 50-digit mpmath, vs physics and
colour known-answers."""
import math
import os

import numpy as np
import pytest

from conftest import IMAGE_GOLDENS, TRACE_GOLDENS, load_golden
from oracle import np_oracle, scenes


@pytest.mark.parametrize("name", IMAGE_GOLDENS)
def test_c_oracle_matches_golden_images(name, oracle, oracle_index, oracle_index_empty):
    g = load_golden("image_" + name)
    ix = oracle_index_empty if "nostars" in name else oracle_index
    img, st = oracle.render(g["cfg"], ix, threads=2)
    assert st["steps"] == int(g["total_steps"])
    assert [st["horizon"], st["escaped"], st["capped"]] == list(g["fate_counts"])
    assert st["disk_hits"] == int(g["disk_hits"]) and st["star_hits"] == int(g["star_hits"])
    # both are binary64 restatements of the same operation order: transcendental libm calls are the only
    # place they may differ (they do not, here), so demand near-bit equality
    np.testing.assert_allclose(img, g["img"], rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("name", TRACE_GOLDENS)
def test_c_oracle_matches_golden_traces(name, oracle, oracle_index):
    g = load_golden("trace_" + name)
    rec = oracle.trace_rays(g["cfg"], oracle_index, g["ys"], g["xs"])
    assert np.array_equal(rec["steps"], g["steps"])
    assert np.array_equal(rec["fate"], g["fate"])
    assert np.array_equal(rec["disk_hits"], g["disk_hits"])
    assert np.array_equal(rec["star_hits"], g["star_hits"])
    assert np.array_equal(rec["vel"], g["vel"]) and np.array_equal(rec["pos"], g["pos"])  # bit-exact trajectories
    np.testing.assert_allclose(rec["rgba"], g["rgba"], rtol=1e-13, atol=1e-15)
    for k in range(0, len(g["ys"]), 37):
        v, p = oracle.generate_ray(g["cfg"], int(g["ys"][k]), int(g["xs"][k]))
        assert np.array_equal(v, g["vel0"][k])


def test_mpmath_pins_discrete_map(oracle, oracle_index_empty):
    from oracle import mp_oracle
    g = load_golden("trace_c3")
    cfg = g["cfg"]
    sc = np_oracle.derive(cfg)
    pick = list(range(0, 384, 29))
    for k in pick:
        steps, fate, v, p = mp_oracle.trace(g["vel0"][k], sc["cam"], sc["h"], sc["safe"])
        assert steps == int(g["steps"][k]) and fate == int(g["fate"][k])
        for a, b in zip(list(v) + list(p), list(g["vel"][k]) + list(g["pos"][k])):
            assert abs(float(a) - b) <= 1e-11 * max(1.0, abs(b))


def _fly(oracle, b, x0=-1000.0, h=0.3, max_steps=200000):
    """Photon launched from (x0, b, 0) along +x; returns (captured, final velocity)."""
    vel, pos = np.array([1.0, 0.0, 0.0]), np.array([x0, b, 0.0])
    h2 = float(np_oracle.quadrance(np_oracle.cross(pos, vel)))
    for _ in range(max_steps):
        r2 = float(pos @ pos)
        if r2 < 1:
            return True, vel
        if r2 > 2 * x0 * x0:
            return False, vel
        vel, pos = oracle.rk4(h, h2, vel, pos)
    raise AssertionError("did not terminate")


def test_physics_capture_threshold(oracle):
    # critical impact parameter of the Schwarzschild photon sphere: b_c = 3*sqrt(3)/2 (r_s = 1)
    assert math.isclose(3 * math.sqrt(3) / 2, 2.598076, rel_tol=1e-6)
    assert _fly(oracle, 2.597)[0] is True
    assert _fly(oracle, 2.599)[0] is False


def test_physics_weak_field_deflection(oracle):
    # deflection angle -> 2 r_s / b = 2/b as b -> infinity
    for b, tol in ((50.0, 0.04), (200.0, 0.01)):
        cap, v = _fly(oracle, b)
        assert not cap
        ang = math.atan2(-v[1], v[0])
        assert abs(ang * b / 2 - 1.0) < tol


def test_physics_angular_momentum_plane(oracle):
    vel, pos = np.array([0.3, 0.1, 0.9486832980505138]), np.array([5.0, 1.0, -20.0])
    L0 = np.cross(pos, vel)
    h2 = float(L0 @ L0)
    for _ in range(150):
        vel, pos = oracle.rk4(0.3, h2, vel, pos)
    L1 = np.cross(pos, vel)
    assert np.linalg.norm(np.cross(L0, L1)) / (L0 @ L0) < 1e-9  # direction of pos x vel conserved


def test_physics_photon_sphere_orbit(oracle):
    # tangential launch at r = 1.5 with |v| = 1: circular orbit for the continuous ODE; the discrete h = 0.3
    # map stays within 1e-2 of r = 1.5 for a full revolution (2*pi*1.5/0.3 ~ 31 steps)
    vel, pos = np.array([0.0, 0.0, 1.0]), np.array([1.5, 0.0, 0.0])
    h2 = 1.5 ** 2
    rs = []
    for _ in range(31):
        vel, pos = oracle.rk4(0.3, h2, vel, pos)
        rs.append(math.sqrt(pos @ pos))
    assert max(abs(r - 1.5) for r in rs) < 1e-2


def test_colour_kats(oracle):
    # SURVEY.md 8c (5) / B.3
    np.testing.assert_allclose(oracle.hsi_to_rgb(0.5, 0.1, 1.05), (0.945, 1.1025, 1.1025), rtol=1e-14)
    np.testing.assert_allclose(oracle.hsi_to_rgb(0.16, 0.1, 0.95), (1.0009482357819488, 0.9940517642180511, 0.855), rtol=1e-14)
    np.testing.assert_allclose(oracle.hsi_to_rgb(0.0, 1.0, 1 / 3), (1.0, 0.0, 0.0), atol=1e-15)
    table = {"O": (0.631, 0.39, (0.166000, 0.298464, 0.735536)), "B": (0.628, 0.33, (0.202000, 0.320938, 0.677062)),
             "A": (0.622, 0.21, (0.274000, 0.357919, 0.568081)), "F": (0.650, 0.03, (0.382000, 0.387544, 0.430456)),
             "G": (0.089, 0.09, (0.451824, 0.402176, 0.346000)), "K": (0.094, 0.29, (0.561017, 0.412983, 0.226000)),
             "M": (0.094, 0.56, (0.710930, 0.425070, 0.064000))}
    for hue, sat, rgb in table.values():
        np.testing.assert_allclose(oracle.hsi_to_rgb(hue, 1.5 * sat, 0.4), rgb, atol=6e-7)
    assert np.array_equal(oracle.hsi_to_rgb(0.0, 0.0, 0.4), (0.4, 0.4, 0.4))
    assert np.all(np.isnan(oracle.hsi_to_rgb(1.0, 0.1, 0.5)))  # reference: error "not properly scaled"
    v = np_oracle.hsi_to_rgb(np.array([0.1, 0.4, 0.9]), 0.3, 0.7)
    for k, h in enumerate((0.1, 0.4, 0.9)):
        np.testing.assert_allclose(v[k], oracle.hsi_to_rgb(h, 0.3, 0.7), rtol=1e-15)


def test_catalogue_reader_kat(oracle):
    import struct
    rec = [(1.0, 0.5, b"G", 650), (4.0, -1.2, b"M", -146), (0.0, 0.0, b"?", 1200)]
    data = bytes(28) + b"".join(struct.pack(">ddcBh8x", ra, dec, sp, 0, mag) for ra, dec, sp, mag in rec) + b"\x01\x02\x03"
    s = oracle.read_ppm(data)
    assert len(s) == 3  # trailing partial record ignored (nBytes `div` 28)
    assert s["mag"].tolist() == [650, -146, 1200]
    assert (s["hue"][0], s["sat"][0]) == (0.089, 0.09) and (s["hue"][1], s["sat"][1]) == (0.094, 0.56) and (s["hue"][2], s["sat"][2]) == (0, 0)
    for k, (ra, dec, _, _) in enumerate(rec):
        assert s["x"][k] == math.cos(dec) * math.cos(ra) and s["y"][k] == math.cos(dec) * math.sin(ra) and s["z"][k] == math.sin(dec)
    with pytest.raises(ValueError):
        oracle.read_ppm(b"short")


def test_star_lookup_grid_vs_brute_force(oracle, oracle_stars, oracle_index):
    rng = np.random.default_rng(7)
    dirs = rng.normal(size=(3000, 3))
    # aim a third of the queries near actual stars so that hits occur
    near = oracle_stars[rng.integers(0, len(oracle_stars), 1000)]
    dirs[:1000] = np.stack([near["x"], near["y"], near["z"]], axis=1) * rng.uniform(0.5, 3, (1000, 1)) + rng.normal(scale=4e-4, size=(1000, 3))
    nhit = 0
    for d in dirs:
        a, na = oracle.star_lookup(oracle_index, 0.4, 1.5, d)
        b, nb = oracle.star_lookup(oracle_index, 0.4, 1.5, d, brute=True)
        assert na == nb and np.array_equal(a, b)
        nhit += na
    assert nhit > 500
    rgb, hits = np_oracle.star_lookup(np.load("tests/golden/catalogue_2000_parsed.npz")["stars"], 0.4, 1.5, dirs[:300])
    for k in range(300):
        a, na = oracle.star_lookup(oracle_index, 0.4, 1.5, dirs[k])
        assert na == hits[k]
        np.testing.assert_allclose(a, rgb[k], rtol=1e-14, atol=1e-16)


def test_clustered_sky_goldens(oracle, oracle_index_clustered):
    """starLookup folds over EVERY star inRadius returns (src/StarMap.hs:104,115).  The clustered-sky fixtures come from the numpy
    restatement; the C restatement must reproduce them: lookups summing up to 40+ stars, the frame that contains such pixels,
    and the per-ray records of the rays that end inside the clusters."""
    g = load_golden("lookup_clustered")
    assert g["hits"].max() >= 40 and (g["hits"] >= 6).sum() > 300
    for k in range(len(g["dirs"])):
        rgb, n = oracle.star_lookup(oracle_index_clustered, float(g["intensity"]), float(g["saturation"]), g["dirs"][k], brute=(k % 7 == 0))
        assert n == g["hits"][k], k
        np.testing.assert_allclose(rgb, g["rgb"][k], rtol=1e-13, atol=1e-16)
    gi = load_golden("image_clustered_default_aa_96x54")
    img, st = oracle.render(gi["cfg"], oracle_index_clustered, threads=2)
    assert st["steps"] == int(gi["total_steps"]) and st["star_hits"] == int(gi["star_hits"]) and st["disk_hits"] == int(gi["disk_hits"])
    np.testing.assert_allclose(img, gi["img"], rtol=1e-13, atol=1e-15)
    gt = load_golden("trace_clustered")
    rec = oracle.trace_rays(gt["cfg"], oracle_index_clustered, gt["ys"], gt["xs"])
    assert np.array_equal(rec["star_hits"], gt["star_hits"]) and rec["star_hits"].max() >= 40
    assert sorted(set(rec["star_hits"].tolist()) & {5, 6, 7, 12}) == [5, 6, 7, 12]  # either side of the kernel's 5 LDS hit slots
    assert np.array_equal(rec["vel"], gt["vel"]) and np.array_equal(rec["steps"], gt["steps"])
    np.testing.assert_allclose(rec["rgba"], gt["rgba"], rtol=1e-13, atol=1e-15)


def test_mpmath_pins_many_star_lookups(oracle, oracle_index_clustered):
    """Lookups that sum 5 .. 40+ stars, evaluated in 50-digit arithmetic from the same binary64 inputs: the hit SETS agree (no star
    within 1e-9 of the radius in these queries) and the FP64 sums -- whatever their order -- sit within 1e-13 of the exact ones."""
    from oracle import mp_oracle
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import parse_catalogue
    g = load_golden("lookup_clustered")
    stars6 = parse_catalogue(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "catalogue_clustered.ppm"), "rb").read())
    pick = [int(k) for k in np.argsort(-g["hits"])[:6]] + [int(np.nonzero(g["hits"] == h)[0][0]) for h in (5, 6, 7, 12)]
    for k in pick:
        d = g["dirs"][k]
        nd = d / np.linalg.norm(d)
        near = stars6[np.sum((stars6[:, :3] - nd) ** 2, axis=1) < 0.003 ** 2]  # a superset of the stars in reach (radius 0.0015)
        rgb, hits, margin = mp_oracle.star_lookup(near, float(g["intensity"]), float(g["saturation"]), d)
        assert margin > 1e-9 and hits == int(g["hits"][k])
        c, n = oracle.star_lookup(oracle_index_clustered, float(g["intensity"]), float(g["saturation"]), d)
        assert n == hits
        for a, b, q in zip(rgb, c, g["rgb"][k]):
            assert abs(float(a) - b) <= 1e-13 * max(1.0, abs(b)) and abs(float(a) - q) <= 1e-13 * max(1.0, abs(q))


def test_oracle_hit_list_is_never_truncated(oracle):
    """Round 2's oracle capped a lookup at 4096 stars silently; the reference has no cap.  6,000 coincident faint stars: all are hits,
    through the index and by brute force, and the colour is the full sum."""
    n = 6000
    stars = np.zeros(n, oracle.STAR_DTYPE)
    stars["x"], stars["mag"], stars["hue"], stars["sat"] = 1.0, 950 + 50 * 14, 0.094, 0.29  # val = intensity * 2^-14 each
    ix = oracle.Index(stars)
    for brute in (False, True):
        rgb, hits = oracle.star_lookup(ix, 1.0, 0.0, np.array([2.0, 0.0, 0.0]), brute=brute)
        assert hits == n
        np.testing.assert_allclose(rgb, [n * 2.0 ** -14] * 3, rtol=1e-12)


def test_clustered_synthetic_catalogue_recipe():
    """bench.py --catalogue clustered: the uniform BASELINE sky + clusters + a dense band, in the PPM on-disk layout."""
    from blackstar_amd import synthetic
    small = synthetic.clustered_catalogue_bytes(n_uniform=5000, n_clusters=40)
    assert small[:28 + 5000 * 28] == synthetic.ppm_catalogue_bytes(5000)  # the uniform part is the BASELINE recipe, unchanged
    from oracle import c_oracle
    st = c_oracle.read_ppm(small)
    xyz = np.stack([st["x"], st["y"], st["z"]], axis=1)
    np.testing.assert_allclose(np.linalg.norm(xyz, axis=1), 1.0, rtol=1e-15)
    ix = c_oracle.Index(st)
    hits = [c_oracle.star_lookup(ix, 0.4, 1.5, xyz[k])[1] for k in range(5000, 5000 + 6 * 40, 3)]
    assert min(hits) >= 2 and max(hits) >= 20  # queries at cluster members see (most of) their cluster
    pole = np.array(synthetic.BAND_POLE) / np.linalg.norm(synthetic.BAND_POLE)
    lat = np.arcsin(xyz @ pole)
    in_band = np.abs(lat) < synthetic.BAND_HALF_WIDTH
    area = 2 * np.pi * 2 * np.sin(synthetic.BAND_HALF_WIDTH)
    assert 7 < in_band.sum() / area / (5000 / (4 * np.pi)) < 13  # ~10x the mean density


def test_supersample_order(oracle):
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 2, (10, 14, 3))
    out = oracle.supersample(img)
    exp = 0.25 * (((img[0::2, 0::2] + img[1::2, 0::2]) + img[0::2, 1::2]) + img[1::2, 1::2])
    assert np.array_equal(out, exp)


def test_empty_star_set_is_black_sky(oracle, oracle_index_empty):
    cfg = scenes.with_res(scenes.DEFAULT, 32, 18)
    cfg["disk_opacity"] = 0.0
    img, st = oracle.render(cfg, oracle_index_empty)
    assert st["disk_hits"] == 0 and np.all(img == 0)  # no starmap == empty star set


def test_step_cap(oracle, oracle_index_empty):
    cfg = scenes.with_res(scenes.DEFAULT, 16, 9)
    img, st = oracle.render(cfg, oracle_index_empty, max_steps=50)
    assert st["capped"] == 16 * 9 and st["steps"] == 50 * 16 * 9


def _bloom_scalar(strength, divider, img):
    """Third, scalar pure-Python restatement of boxBlur/bloom for tiny images (src/ImageFilters.hs:28-86)."""
    h, w = len(img), len(img[0])
    r = w // divider
    nf = 1 / (2 * float(r) + 1)
    cur = [row[:] for row in img]

    def sweep(get, n):
        pix = lambda i: get(i) if 0 <= i < n else 0.0
        vals = [pix(i) for i in range(min(r, n))]
        s = vals[0]
        for v in vals[1:]:
            s = s + v
        out = []
        for x in range(n):
            s = (s + pix(x + r)) - pix(x - r)
            out.append(nf * s)
        return out

    for _ in range(3):
        tmp = [row[:] for row in cur]
        for y in range(h):
            cur[y] = sweep(lambda i, y=y: tmp[y][i], w)
        tmp = [row[:] for row in cur]
        for x in range(w):
            col = sweep(lambda i, x=x: tmp[i][x], h)
            for y in range(h):
                cur[y][x] = col[y]
    return [[img[y][x] + strength * cur[y][x] for x in range(w)] for y in range(h)]


def test_bloom_restatements_agree_and_quirks(oracle):
    rng = np.random.default_rng(21)
    img = rng.uniform(0, 1.5, (37, 53, 3))
    a = oracle.bloom(0.15, 7, img)          # r = 53 // 7 = 7
    b = np_oracle.bloom(0.15, 7, img)
    assert np.array_equal(a, b)             # same sequential running sums, bit for bit
    small = rng.uniform(0, 2, (5, 9, 3))
    for div in (2, 3, 9):                   # r = 4, 3, 1 (r > h exercises `take r` on a short column)
        got = oracle.bloom(0.4, div, small)
        for c in range(3):
            exp = _bloom_scalar(0.4, div, small[:, :, c].tolist())
            assert np.array_equal(got[:, :, c], np.array(exp))
    # hand-computed first pass of the quirk: window [x-r+1, x+r], normalised by 1/(2r+1), zero padding:
    # row [1,2,3,4], r = 2: running sums 6, 10, 9, 7 -> /5; the 1-pixel-high vertical sweep multiplies by 1/5 again
    row = np.array([[1.0, 2.0, 3.0, 4.0]])
    one_pass = np.array([6.0, 10.0, 9.0, 7.0]) / 5 / 5
    three = np.stack([row] * 3, axis=2)
    blurred3 = oracle.bloom(1.0, 2, three)[0, :, 0] - row[0]
    p = one_pass
    for _ in range(2):  # two more passes by the same hand rule
        s = [p[0] + p[1] + p[2], p[0] + p[1] + p[2] + p[3], p[1] + p[2] + p[3], p[2] + p[3]]
        p = np.array(s) / 25
    np.testing.assert_allclose(blurred3, p, rtol=1e-12)  # (out - img) cancels ~2 digits
    with pytest.raises(ValueError):
        oracle.bloom(0.1, 100, img)  # radius 0: the reference crashes


def test_srgb8_restatements_agree(oracle):
    rng = np.random.default_rng(22)
    img = np.concatenate([rng.uniform(-0.1, 1.3, 5000), [0.0, 0.0031308, 0.00313079, 1.0, 2.0, -1.0, 0.5]]).reshape(-1, 1)
    a = oracle.srgb8(img)
    b = np_oracle.srgb8(img)
    assert np.abs(a.astype(int) - b.astype(int)).max() == 0
    assert oracle.srgb8(np.array([[0.0, 1.0, 0.5, 0.0031308]])).tolist() == [[0, 255, 188, 10]]


def test_config0_default_640x480_cpu_path(oracle, oracle_index):
    """BASELINE configs[0]: scenes/default.yaml at 640x480, no supersampling, on the CPU path (the C restatement standing
    in for the reference's Haskell CPU path, which cannot be built here) against the numpy restatement's full-frame summary."""
    g = load_golden("summary_c1_default_640x480")
    img, st = oracle.render(g["cfg"], oracle_index, threads=0)
    assert st["rays"] == 640 * 480 and st["capped"] == 0
    assert st["steps"] == int(g["total_steps"]) and [st["horizon"], st["escaped"], st["capped"]] == list(g["fate_counts"])
    assert st["disk_hits"] == int(g["disk_hits"]) and st["star_hits"] == int(g["star_hits"])
    np.testing.assert_allclose(img[g["ys"], g["xs"]], g["samples"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(img.reshape(-1, 3).sum(axis=0), g["channel_sums"], rtol=1e-12)
    assert abs(st["steps"] / st["rays"] - 224.0) < 0.1  # mean 224.0 steps per ray on C1


def test_photon_ring_matches_the_reference_repositorys_example_image(oracle, oracle_index_empty):
    """The only rendered output the reference holds is example.png (README.md:4): the scenes/default.yaml camera at
    1280x720 from an unknown revision (different disk colour and bloom, real catalogue) -- no pixel golden.  But the thin
    photon ring inside the shadow depends only on generateRay (fov convention, look-at basis, aspect ratio) and on the
    geodesic integration: the oracle's ring must sit where the reference's own picture has it."""
    from conftest import ring_offsets_vs_reference_example
    img, _ = oracle.render(scenes.with_res(scenes.DEFAULT, 1280, 720), oracle_index_empty, threads=0)
    d, n = ring_offsets_vs_reference_example(img)
    print(f"ring radius, reference example.png - oracle: mean {d.mean():+.2f} px, std {d.std():.2f}, max |d| {np.abs(d).max():.2f} over {len(d)}/{n} angles")
    assert len(d) >= 0.85 * n
    assert abs(d.mean()) < 0.5 and d.std() < 0.8 and np.abs(d).max() <= 2.5
    # negative control: the comparison resolves a 2 % change of the field of view (ring radius 114 px -> about 2.3 px)
    wide = dict(scenes.with_res(scenes.DEFAULT, 1280, 720), fov=1.5 * 1.02)
    d2, _ = ring_offsets_vs_reference_example(oracle.render(wide, oracle_index_empty, threads=0)[0])
    assert len(d2) >= 0.7 * n and d2.mean() > 1.5


def _example_scene(**over):
    """The scene example.png shows: scenes/default.yaml's camera at 1280x720, supersampled, with the ConfigFile DEFAULT disk
    (diskInner 3, diskOuter 12, default colour: src/ConfigFile.hs:74-77) -- see tests/golden/make_reference_disk_edges.py."""
    return dict(dict(scenes.with_res(scenes.DEFAULT, 1280, 720, ss=True), disk_inner=3.0, disk_outer=12.0, disk_hsi=(0.16, 0.1, 0.95)), **over)


def test_disk_inner_edge_matches_the_reference_repositorys_example_image(oracle, oracle_index_empty):
    """example.png again (cf. the photon-ring test), second feature: the locus of the disk's inner edge, r = diskInner, in the
    primary image (above the shadow) and in the lensed secondary image (below it) -- about 150 angles.  It is where findColor's
    y-plane sign-change test with the interpolated r2ave first exceeds diskInner^2 (src/Raytracer.hs:96-102), reached by rays
    that pass the hole at 3 .. 6 Schwarzschild radii: a different family from the ring's.  The older revision that produced the
    picture uses another intensity law (narrower profile, B/R = 0.79), which delays ITS threshold crossing by 1-2 px: the bar
    is 'within 2.2 px on average'; a 10 % change of diskInner moves the locus by 4 px and is rejected."""
    from conftest import disk_inner_edge_offsets_vs_reference_example
    img, _ = oracle.render(_example_scene(), oracle_index_empty, threads=0)
    o = disk_inner_edge_offsets_vs_reference_example(img)
    edge, ring = o[o[:, 1] > 122], o[o[:, 1] <= 122]
    lower, upper = edge[edge[:, 0] < 180], edge[edge[:, 0] >= 180]  # lensed secondary image / primary image
    print(f"disk inner edge, reference - oracle: {len(edge)} angles, mean {edge[:, 2].mean():+.2f} px, std {edge[:, 2].std():.2f}, max |d| {np.abs(edge[:, 2]).max():.2f} "
          f"(lensed image: {len(lower)} angles {lower[:, 2].mean():+.2f}; primary: {len(upper)} angles {upper[:, 2].mean():+.2f}); "
          f"ring: {len(ring)} angles, mean {ring[:, 2].mean():+.2f}, std {ring[:, 2].std():.2f}")
    assert len(lower) >= 20 and len(upper) >= 80 and len(ring) >= 90
    assert _edge_locus_accepted(edge)
    assert abs(ring[:, 2].mean()) < 0.6 and ring[:, 2].std() < 0.6
    # negative controls: a 10 % change of diskInner either way moves the locus by about 4 px (or out of the matching window) ...
    for inner in (3.3, 2.7):
        o2 = disk_inner_edge_offsets_vs_reference_example(oracle.render(_example_scene(disk_inner=inner), oracle_index_empty, threads=0)[0])
        assert not _edge_locus_accepted(o2[o2[:, 1] > 122]), inner
    # ... and it is the ConfigFile default radius, not default.yaml's 1.8, that the picture shows
    o3 = disk_inner_edge_offsets_vs_reference_example(oracle.render(_example_scene(disk_inner=1.8, disk_outer=13.0), oracle_index_empty, threads=0)[0])
    assert not _edge_locus_accepted(o3[o3[:, 1] > 122])


def _edge_locus_accepted(edge):
    """The bar of the disk-edge pin: onsets matched at >= 120 of the 180 angles, reference - render between -0.5 and +2.2 px on
    average (the older revision's intensity law delays its threshold crossing by 1-2 px), scatter below 2 px."""
    return len(edge) >= 120 and -0.5 < edge[:, 2].mean() < 2.2 and edge[:, 2].std() < 2.0


def test_disk_intensity_law_ranks_like_the_reference_repositorys_example_image(oracle, oracle_index_empty):
    """example.png a third time -- PHOTOMETRIC, as far as the picture allows (tests/golden/make_reference_disk_colour.py): its colour is a
    hue-60 diskColor of an unknown scene and its brightness goes as I^1.7 of today's I (an older revision), so neither massiv-io's HSI
    sectors nor blend / opacity / bloom can be pinned by it.  What no monotone post-processing can hide is the ARGUMENT of the law: over
    7 000 unclipped single-crossing pixels the picture's blue channel ranks like the oracle's own I = sin(pi t^2), t = (rO - r)/(rO - rI) with
    the ConfigFile default radii 3 / 12 (Spearman rho 0.965; the best exponent in sin(pi t^p) is 2.1), and its profile peaks where the law
    does (r = 5.64).  Negative controls fail: sin(pi t) (rho 0.54, peak at 7.5), p = 1.5 and 3, t reversed, default.yaml's radii.
    Control that CANNOT fail (documented, not hidden): a pre-multiplied blend (I^2) ranks identically."""
    from conftest import disk_law_vs_reference_example

    def by_oracle(cfg, ys, xs):
        rec = oracle.trace_rays(cfg, oracle_index_empty, ys, xs)
        return rec["rgba"][:, 0], rec["disk_hits"]

    o = disk_law_vs_reference_example(by_oracle)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in o.items()})
    assert o["single_crossing"] and o["pixels"] > 6500
    assert o["max_dev_from_sin_pi_t2"] < 1e-9                     # the fixture's radii (numpy restatement) and the C oracle are the same geometry
    assert o["implementation"] > 0.96
    for control in ("sin(pi t)", "sin(pi t^1.5)", "sin(pi t^3)", "t reversed", "default.yaml radii 1.8/13"):
        assert o[control] < o["implementation"] - 0.025, control
    assert o["sin(pi t)"] < 0.6 and o["t reversed"] < 0
    assert 1.9 <= o["best_exponent"] <= 2.3
    assert abs(o["picture_peak_radius"] - o["law_peak_radius"]) < 0.3 and abs(o["picture_peak_radius"] - 7.5) > 1.5    # sin(pi t) would peak at 7.5
    assert abs(o["premultiplied I^2"] - o["implementation"]) < 1e-12   # rank statistics cannot tell a pre-multiplied blend: NOT pinned by the picture


# ---- the reference itself as the pin (tools/ghc_pin) --------------------------------------------------------------------------

def _oracle_callables(oracle, d):
    ix = oracle.Index(oracle.read_ppm(d.catalogue_bytes()))
    return dict(render=lambda cfg: oracle.render(cfg, ix, threads=0)[0],
                star_lookup=lambda i, s, dirs: np.stack([oracle.star_lookup(ix, i, s, v)[0] for v in dirs]),
                bloom=oracle.bloom, srgb8=oracle.srgb8)


@pytest.mark.parametrize("which", ["uniform", "clustered"])
def test_oracle_against_the_reference_itself(which, oracle):
    """The C restatement against the output of the REFERENCE's own render / bloom / writeImg / starLookup on this repository's
    fixed inputs (tools/ghc_pin/Dump.hs).  This is the test that would turn 'parity unpinned' into 'pinned'; it needs a dump made with
    GHC, which this image cannot produce -- so it skips, loudly, until tests/golden/ghc/ exists."""
    import ghc_pin
    if not ghc_pin.available(which):
        pytest.skip(ghc_pin.SKIP_REASON.format(set=which))
    d = ghc_pin.Dump(which)
    rep = ghc_pin.compare(d, rtol=1e-12, atol=1e-15, **_oracle_callables(oracle, d))
    print(f"{which}: {rep['scenes']} scenes, {rep['values']} values vs GHC ({' '.join(d.meta.get('compiler', []))}): "
          f"{rep['bit_equal'] / rep['values']:.2%} bit-equal, worst rel {rep['worst_rel']:.2e}")
    # SURVEY Appendix B, recalled semantics this settles: linear.normalize's shortcut, kdt inRadius '<=', massiv-io HSI->RGB, toWord8
    stars = oracle.read_ppm(d.catalogue_bytes())
    mine = sorted(map(tuple, np.stack([stars["x"], stars["y"], stars["z"], stars["mag"].astype(float), stars["hue"], stars["sat"]], axis=1).tolist()))
    assert sorted(map(tuple, d.assocs().tolist())) == mine, "KdMap.assocs (readMap + starColor') differs from the restated catalogue reader"
    from blackstar_amd import kdt_file  # row f4: the recalled cereal layout of stars.kdt against a real file
    back = kdt_file.read_kdt(d.kdt_bytes())
    got = np.stack([back["x"], back["y"], back["z"], back["mag"].astype(float), back["hue"], back["sat"]], axis=1)
    assert np.array_equal(got, d.assocs()), "stars.kdt decodes to other stars (or another order) than KdMap.assocs"
    if which == "uniform":
        assert ghc_pin.compare_animation(d) == 375  # row f3: generateFrames on animations/default-ani.yaml


def test_ghc_pin_harness_on_a_stand_in_dump(tmp_path, oracle):
    """Self-test of the comparison harness ONLY -- not a pin.  A directory in the dump's format is written from the numpy restatement
    (never into tests/golden/ghc), the C oracle is compared against it through the very code path a real dump takes, and a
    corrupted copy must be caught.  Proves the reader, the PNG decoder and the comparisons work before a real dump exists."""
    import shutil
    import ghc_pin
    import blackstar_amd as bs
    from blackstar_amd import kdt_file
    from blackstar_amd.raytracer import write_png
    which = "clustered"
    out = tmp_path / "dumps" / which
    out.mkdir(parents=True)
    inp = os.path.join(ghc_pin.INPUTS, which)
    cat = open(os.path.join(inp, "catalogue.ppm"), "rb").read()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import parse_catalogue
    stars6 = parse_catalogue(cat)
    name = "clustered_default_aa_96x54"
    cfg = bs.Config.from_file(os.path.join(inp, "scenes", name + ".yaml"))
    img, _ = np_oracle.render(cfg.to_bs_config(), stars6)
    img.astype("<f8").tofile(out / f"{name}.render.f64")
    bl = np_oracle.bloom(cfg.scene.bloomStrength, cfg.scene.bloomDivider, img)
    bl.astype("<f8").tofile(out / f"{name}.bloom.f64")
    write_png(np_oracle.srgb8(bl), str(out / f"{name}.png"))
    dirs = np.fromfile(os.path.join(inp, "dirs.f64"), "<f8").reshape(-1, 3)
    i, s = map(float, open(os.path.join(inp, "lookup.txt")).read().split())
    np_oracle.star_lookup(stars6, i, s, dirs)[0].astype("<f8").tofile(out / "starlookup.f64")
    rec = np.frombuffer(cat, np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]), offset=28)
    kdt = kdt_file.write_kdt(stars6[:, :3], rec["mag"], "".join(chr(c) for c in rec["sp"]))
    (out / "stars.kdt").write_bytes(kdt)
    back = kdt_file.read_kdt(kdt)
    np.stack([back["x"], back["y"], back["z"], back["mag"].astype(float), back["hue"], back["sat"]], axis=1).astype("<f8").tofile(out / "assocs.f64")
    (out / "manifest.txt").write_text(f"compiler STAND-IN numpy-restatement\nstars {len(stars6)} dirs {len(dirs)}\nscene {name} 96 54 bloom\nanimation 375\n")
    shutil.copyfile(os.path.join(ghc_pin.INPUTS, "uniform", "animation.yaml"), tmp_path / "animation.yaml")
    cams = [scenes.ani_frame(i, 375) for i in range(375)]  # oracle/scenes.py: an independent restatement of the two-keyframe interpolation
    np.array([list(c["cam_pos"]) + list(c["cam_lookat"]) + list(c["cam_up"]) + [c["fov"]] for c in cams]).astype("<f8").tofile(out / "animation_frames.f64")
    assert ghc_pin.available(which, root=str(tmp_path / "dumps"))
    d = ghc_pin.Dump(which, root=str(tmp_path / "dumps"))
    assert np.array_equal(d.png(name), np_oracle.srgb8(bl))  # own PNG decoder reads back what the encoder wrote
    rep = ghc_pin.compare(d, rtol=1e-12, atol=1e-15, **_oracle_callables(oracle, d))
    assert rep["scenes"] == 1 and rep["values"] == 96 * 54 * 3 + 3 * len(dirs) and rep["bit_equal"] > 0.9 * rep["values"]
    d.inputs = str(tmp_path)  # (the stand-in's animation file sits beside it)
    assert ghc_pin.compare_animation(d) == 375
    d.inputs = os.path.join(ghc_pin.INPUTS, which)
    # a single wrong value anywhere must fail the comparison
    for victim, offset in ((f"{name}.render.f64", 8 * 1234), ("starlookup.f64", 0), (f"{name}.bloom.f64", 8 * 77)):
        bad_root = tmp_path / ("bad_" + victim)
        shutil.copytree(tmp_path / "dumps", bad_root)
        raw = bytearray((bad_root / which / victim).read_bytes())
        v = np.frombuffer(bytes(raw[offset:offset + 8]), "<f8")[0]
        raw[offset:offset + 8] = np.array([v * (1 + 1e-9) + 1e-9], "<f8").tobytes()
        (bad_root / which / victim).write_bytes(bytes(raw))
        with pytest.raises(AssertionError):
            ghc_pin.compare(ghc_pin.Dump(which, root=str(bad_root)), rtol=1e-12, atol=1e-15, **_oracle_callables(oracle, d))


def test_ghc_pin_kit_is_complete_and_consistent():
    """The hand-off must not rot: every input Dump.hs reads exists, the scene files decode with this repository's own decoder, the
    runner script parses, and Dump.hs names only functions the reference's library stanza exports (blackstar.cabal:16-24 modules)."""
    import subprocess
    import ghc_pin
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ghc_pin")
    assert subprocess.run(["sh", "-n", os.path.join(root, "run.sh")]).returncode == 0
    for which, n_scenes, n_dirs in (("uniform", 11, 10000), ("clustered", 1, 1500)):
        d = os.path.join(ghc_pin.INPUTS, which)
        assert os.path.getsize(os.path.join(d, "catalogue.ppm")) % 28 == 0
        assert os.path.getsize(os.path.join(d, "dirs.f64")) == n_dirs * 24
        assert len(open(os.path.join(d, "lookup.txt")).read().split()) == 2
        names = sorted(f for f in os.listdir(os.path.join(d, "scenes")) if f.endswith(".yaml"))
        assert len(names) == n_scenes
        import blackstar_amd as bs
        for f in names:
            c = bs.Config.from_file(os.path.join(d, "scenes", f))
            assert c.scene.resolution[0] // c.scene.bloomDivider >= 1  # bloom would crash the reference otherwise (foldl1' on [])
    assert os.path.exists(os.path.join(ghc_pin.INPUTS, "uniform", "animation.yaml"))
    src = open(os.path.join(root, "Dump.hs")).read()
    for fn in ("readMapFromFile", "buildStarTree", "treeToByteString", "readTreeFromFile", "starLookup", "render cfg tree", "writeImg", "bloom (bloomStrength scn)",
               "An.generateFrames", "An.validateKeyframes", "K.assocs", "padZero"):
        assert fn in src, fn
    stanza = open(os.path.join(root, "run.sh")).read()
    for dep in ("blackstar", "cereal", "yaml", "kdt", "massiv-io"):
        assert dep in stanza
    # the FFI shim: INTEGRATION.md quotes tools/ghc_pin/RaytracerFFI.hs verbatim; it binds entry points the header declares, expects
    # the header's ABI version, and run.sh type-checks it where GHC exists
    import re
    repo = os.path.dirname(os.path.dirname(root))
    shim = open(os.path.join(root, "RaytracerFFI.hs")).read()
    quoted = [m.group(1) for m in re.finditer(r"```haskell\n(.*?)```", open(os.path.join(repo, "INTEGRATION.md")).read(), re.S)]
    assert shim in quoted
    header = open(os.path.join(repo, "include", "blackstar_gpu.h")).read()
    bound = re.findall(r'foreign import ccall (?:safe|unsafe)\s+"&?(bs_\w+)"', shim)
    assert len(bound) >= 11 and all(re.search(r"\b%s\(" % b, header) for b in bound), bound
    # the multi-GPU batch path (north_star: frames sharded over the GPUs of a node) is bound too, and every import is USED
    assert {"bs_device_count", "bs_render_png_files", "bs_render_batch", "bs_create", "bs_render", "bs_destroy"} <= set(bound)
    for hs_name in re.findall(r'foreign import ccall (?:safe|unsafe)\s+"[^"]+"\s+(\w+)', shim):
        assert len(re.findall(r"\b%s\b" % hs_name, shim)) >= 2, f"{hs_name} is imported and never used"
    # arity of each import = the number of parameters the header declares for it
    for name, sig in re.findall(r'foreign import ccall (?:safe|unsafe)\s+"(bs_\w+)"\s+\w+\s*::\s*(.*?)(?=\nforeign|\n--|\n\n)', shim, re.S):
        depth, arrows = 0, 0
        for a, b in zip(sig, sig[1:] + " "):
            depth += a == "("
            depth -= a == ")"
            arrows += depth == 0 and a == "-" and b == ">"
        decl = re.search(r"\b%s\(([^;]*?)\);" % name, header, re.S).group(1).strip()
        n_params = 0 if decl in ("", "void") else decl.count(",") + 1
        assert arrows == n_params, (name, arrows, n_params, sig)
    batch = open(os.path.join(root, "BatchMain.hs")).read()
    assert batch in quoted and "BatchMain.hs" in stanza
    exported = re.search(r"module RaytracerFFI \((.*?)\) where", shim, re.S).group(1).replace("\n", " ")
    for fn in ("withGpuTrees", "renderScenesToFiles", "renderBatch", "renderPure"):
        assert fn in exported and re.search(r"^%s ::" % fn, shim, re.M), fn
    assert "renderScenesToFiles gpus jobs" in batch
    # massiv re-exports Prelude / Control.Monad names: the shim must not import it unqualified wholesale (ambiguous `zip`, `forM_` ...)
    assert not re.search(r"^import\s+Data\.Massiv\.Array\s*(as \w+)?\s*$", shim, re.M)
    version = re.search(r"#define BS_ABI_VERSION (\d+)", header).group(1)
    assert f"this shim expects {version}" in shim and "RaytracerFFI.hs" in stanza and "-fno-code" in stanza
    verdict = os.path.join(repo, "tests", "golden", "ghc", "shim_typecheck.txt")
    if os.path.exists(verdict):
        assert open(verdict).read().strip() == "OK", "the shim did not type-check against the reference (tests/golden/ghc/shim_typecheck.log)"


def test_haskell_files_of_the_kit_use_only_names_that_are_in_scope(tmp_path):
    """No GHC here, so the first thing a compiler would say is checked by a lint (tests/hs_scope.py): every name Dump.hs, RaytracerFFI.hs and
    BatchMain.hs use is bound in the file or provided by exactly ONE of its imports, per hand-kept export tables of the modules (lts-13.16).
    Negative controls: the shim as round 4 had it (massiv imported wholesale next to Prelude: ambiguous `zip`, `forM_`), a dropped import, a
    name base does not export from `Foreign` (unsafeForeignPtrToPtr), a typo in a foreign import's name -- each is caught."""
    import re
    import hs_scope
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ghc_pin")
    shim_src = open(os.path.join(root, "RaytracerFFI.hs")).read()
    exported = set(re.sub(r"\s+", " ", re.search(r"module RaytracerFFI \((.*?)\) where", shim_src, re.S).group(1)).replace(",", " ").split())
    for name, extra in (("Dump.hs", None), ("RaytracerFFI.hs", None), ("BatchMain.hs", {"RaytracerFFI": exported})):
        unknown, ambiguous = hs_scope.check(os.path.join(root, name), extra)
        assert not unknown and not ambiguous, (name, unknown, ambiguous)

    def variant(edit):
        p = tmp_path / "V.hs"
        edited = edit(shim_src)
        assert edited != shim_src
        p.write_text(edited)
        return hs_scope.check(str(p))

    unknown, ambiguous = variant(lambda t: t.replace("import qualified Data.Massiv.Array as A", "import Data.Massiv.Array as A"))
    assert any(a.startswith("zip ") for a in ambiguous) and any(a.startswith("forM_ ") for a in ambiguous)
    assert variant(lambda t: t.replace("import Data.List (partition)\n", ""))[0] == {"partition"}
    assert variant(lambda t: t.replace("withMany withForeignPtr bufs", "withArray (map unsafeForeignPtrToPtr bufs)"))[0] == {"unsafeForeignPtrToPtr"}
    assert variant(lambda t: t.replace("c_bs_render_batch pctxs", "c_bs_render_batchh pctxs"))[0] == {"c_bs_render_batchh"}
    assert variant(lambda t: t.replace("import Control.Monad (when, forM, forM_)", "import Control.Monad (forM, forM_)"))[0] == {"when"}


def test_mirror_symmetry_pins_ray_generation_and_integration(oracle, oracle_stars):
    """A property no restatement can share a mistake about: the scene (hole + disk in y = 0) is mirror-symmetric and IEEE arithmetic is
    sign-symmetric, so with a power-of-two resolution (x'/W, y'/H exact) the scene mirrored in the disk plane -- camera, lookAt and stars
    y -> -y, upVec mirrored and negated -- renders the same frame upside down, row y = row H - y (no half-pixel offset: row 0 has no
    partner, src/Raytracer.hs:40-51), and a camera in the plane x = 0 looking at the hole renders a left-right symmetric frame.  Bit for
    bit: a wrong sign, a swapped axis or an off-by-one in generateRay / lookAt / findColor's crossing test breaks it."""
    W, H = 128, 64
    base = dict(scenes.DEFAULT_AA, width=W, height=H, supersampling=False)
    m = dict(base, cam_pos=(base["cam_pos"][0], -base["cam_pos"][1], base["cam_pos"][2]),
             cam_lookat=(base["cam_lookat"][0], -base["cam_lookat"][1], base["cam_lookat"][2]),
             cam_up=(-base["cam_up"][0], base["cam_up"][1], -base["cam_up"][2]))
    mirrored = oracle_stars.copy()
    mirrored["y"] = -mirrored["y"]
    a, sa = oracle.render(base, oracle.Index(oracle_stars), threads=0)
    b, sb = oracle.render(m, oracle.Index(mirrored), threads=0)
    assert sa["rays"] == sb["rays"] and sa["star_hits"] > 0 and sa["disk_hits"] > 0    # (row 0 of either frame has no partner: the totals need not agree)
    np.testing.assert_allclose(b[:0:-1], a[1:], rtol=1e-13, atol=0)     # (the stars of a lookup are summed in index order: the same here)
    empty = oracle.Index(None)
    a0, _ = oracle.render(base, empty, threads=0)
    b0, _ = oracle.render(m, empty, threads=0)
    assert np.array_equal(b0[:0:-1], a0[1:]) and a0[1:].any()
    assert not np.array_equal(b0[1:], a0[1:])                          # (negative control: not simply the same image)
    c, _ = oracle.render(dict(base, cam_pos=(0.0, 3.0, -20.0), cam_lookat=(0.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.0)), empty, threads=0)
    assert np.array_equal(c[:, 1:], c[:, :0:-1]) and c.any() and not np.array_equal(c[:, 1:], c[:, :-1])
