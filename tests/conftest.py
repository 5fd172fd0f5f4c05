import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A kernel that never terminates would otherwise hold the GPU box until the caller's own limit: every GPU test gets
    a hard per-test limit (pytest-timeout, thread method = the process is ended, which also ends the queue)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(240, method="thread"))


def to_host(t):
    """A device tensor as a numpy array, through PAGE-LOCKED host memory when it is large: `.cpu()` into pageable memory above 1 MiB makes
    the HIP runtime pin pages on the fly and DMA at their address -- the path on which a GPU memory fault was caught in round 4
    (profiles/EXPERIMENTS.md section 5).  The library avoids it; the tests' own torch copies should too."""
    import torch
    if t.is_cuda and t.numel() * t.element_size() > (1 << 20):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h.numpy()
    return t.cpu().numpy()


def to_device(a):
    """A numpy array as a device tensor, through page-locked host memory (see to_host)."""
    import torch
    return torch.from_numpy(a).pin_memory().cuda()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    if "cfg" in d:
        d["cfg"] = json.loads(str(d["cfg"]))
    return d


@pytest.fixture(scope="session")
def catalogue_bytes():
    with open(os.path.join(GOLDEN, "catalogue_2000.ppm"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def clustered_bytes():
    """The NON-uniform test sky (tests/golden/make_golden.py --clustered): the 2,000 stars + 48 clusters of 5..40 stars inside
    0.001 rad (at directions in which rays of the 96x54 default-aa frame leave the scene) + a band at 10x the mean density."""
    with open(os.path.join(GOLDEN, "catalogue_clustered.ppm"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def oracle_stars(oracle, catalogue_bytes):
    return oracle.read_ppm(catalogue_bytes)


@pytest.fixture(scope="session")
def oracle_index(oracle, oracle_stars):
    return oracle.Index(oracle_stars)


@pytest.fixture(scope="session")
def oracle_index_clustered(oracle, clustered_bytes):
    return oracle.Index(oracle.read_ppm(clustered_bytes))


@pytest.fixture(scope="session")
def oracle_index_empty(oracle):
    return oracle.Index(None)


C1_SUMMARY = "summary_c1_default_640x480"
IMAGE_GOLDENS = ["c2_default_96x54_nostars", "c3_default_aa_96x54", "c4_lensing_disk_96x54", "c5_ani_frame300_80x45",
                 "odd_default_aa_37x23",
                 # the reference's other six scene files (tests/golden/make_golden.py --extra)
                 "ref_closeup_96x72", "ref_fartheraway_96x54", "ref_lensing_96x72", "ref_wideangle_disk_96x54",
                 "ref_wideangle_96x51", "ref_wideangle1_96x54"]
TRACE_GOLDENS = ["c1", "c2", "c3", "c4", "c5_f0", "c5_f599",
                 "ref_closeup", "ref_fartheraway", "ref_lensing", "ref_wideangle_disk", "ref_wideangle", "ref_wideangle1"]


def ring_offsets_vs_reference_example(img):
    """Photon-ring radius of a rendered default.yaml frame (1280x720, linear RGB f64, no stars needed) against the ring in the
    reference repository's own example.png, via tests/golden/reference_example_ring.npz (see make_reference_ring.py).
    Returns (radius differences in px for the angles at which both images show a distinct ring, number of angles probed)."""
    from scipy import ndimage as ndi
    g = np.load(os.path.join(GOLDEN, "reference_example_ring.npz"))
    cx, cy, r, th, ref = float(g["cx"]), float(g["cy"]), g["r"], g["theta"], g["lum3"].astype(np.float64)
    assert img.shape == (int(g["height"]), int(g["width"]), 3)
    lum = img.sum(axis=2)
    mine = np.stack([ndi.map_coordinates(lum, [cy + r * np.sin(t), cx + r * np.cos(t)], order=1) for t in th])

    def peak(v, lo, hi):  # position and height of the highest point of v[lo:hi] above the straight line between the window's ends
        w = v[lo:hi]
        bg = np.linspace(w[:4].mean(), w[-4:].mean(), len(w))
        k = int(np.argmax(w - bg))
        return lo + k, (w - bg)[k]

    out = []
    for i in range(len(th)):
        km, am = peak(mine[i], 0, len(r))
        if am < 0.05 or km < 40 or km > len(r) - 40:  # no isolated ring at this angle (it merges with the disk image)
            continue
        kr, ar = peak(ref[i], km - 32, km + 33)       # +-8 px around my ring
        if ar < 18:                                   # the reference's ring is not distinct here either
            continue
        out.append(r[kr] - r[km])
    return np.array(out), len(th)


def disk_inner_edge_offsets_vs_reference_example(img, tau=0.15):
    """Locus of the accretion disk's INNER edge in a render of the scenes/default.yaml camera at 1280x720 with the ConfigFile
    default disk radii (diskInner 3, diskOuter 12) against the reference repository's example.png, via
    tests/golden/reference_example_disk.npz (make_reference_disk_edges.py).  Along each of 180 rays from the shadow's centre, every
    ONSET of light in `img` (>= 12 px of exact darkness, then a rise past 2.5 tau) is located by where its luminance crosses
    tau (linear light, R+G+B), and the reference by where ITS (sRGB-decoded, lightly smoothed) luminance crosses the local
    background + tau.  Onsets are the photon ring (r < 122 px) and the r = diskInner edge of the primary and of the lensed
    secondary image.  Returns an array of (angle deg, radius px of the onset in img, reference - img in px)."""
    from scipy import ndimage as ndi
    g = np.load(os.path.join(GOLDEN, "reference_example_disk.npz"))
    cx, cy, r, th = float(g["cx"]), float(g["cy"]), g["r"], g["theta"]
    assert img.shape == (int(g["height"]), int(g["width"]), 3)
    raw = g["lum3"].astype(np.float64)
    inside = raw < 60000
    v = np.clip(raw, 0, 765) / 765.0
    ref = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4) * 3.0  # linear light, R+G+B
    lum = img.sum(axis=2)
    mine = np.stack([ndi.map_coordinates(lum, [cy + r * np.sin(t), cx + r * np.cos(t)], order=1, mode="constant", cval=0.0) for t in th])
    dr = float(r[1] - r[0])
    dark, rise, reach = int(12 / dr), int(8 / dr), int(14 / dr)

    def cross(a, level, start):
        for q in range(max(start, 1), min(start + reach, len(a))):
            if a[q] >= level > a[q - 1]:
                return q - 1 + (level - a[q - 1]) / (a[q] - a[q - 1])
        return None

    out = []
    for i in range(len(th)):
        m, rf, lit = mine[i], ndi.gaussian_filter1d(ref[i], 0.75 / dr), mine[i] > 1e-9
        k = dark
        while k < len(r) - reach - 12:
            if lit[k] and not lit[k - dark:k].any() and m[k:k + rise].max() > 2.5 * tau and inside[i, k - dark:k + reach + 12].all():
                bg = float(np.median(rf[k - dark + 4:k - 8]))
                cm, cr = cross(m, tau, k - 1), cross(rf, bg + tau, k - int(6 / dr))
                if bg < 0.35 and cm is not None and cr is not None:  # a star or the disk's own glow in the dark stretch: skip
                    out.append((np.rad2deg(th[i]), r[k], (cr - cm) * dr))
                k += reach + 12
            else:
                k += 1
    return np.array(out)


def disk_law_vs_reference_example(trace_intensity):
    """The ARGUMENT of the disk's intensity law (src/Raytracer.hs:106-110: sin (pi * ((rO - r)/(rO - rI))^2)) against the reference repository's
    example.png, via tests/golden/reference_example_disk_colour.npz (make_reference_disk_colour.py): 8 000 pixels of the picture whose ray
    meets the disk exactly once.  trace_intensity(cfg, ys, xs) -> (I, disk_hits) is the implementation under test tracing THOSE pixels of the
    example's scene with a white disk of opacity 1 (one crossing: rgba = I exactly).  What is compared is monotone-invariant -- how the
    picture's blue channel (the one that does not clip) RANKS the pixels, and where along the disk its profile peaks -- because the
    picture's colour, amplitude and bloom belong to an unknown scene and an older revision.
    Returns a dict: rho of the implementation's I, rho of alternative laws on the fixture's radii, the peak radius of the picture."""
    from scipy import stats
    from oracle import scenes
    g = np.load(os.path.join(GOLDEN, "reference_example_disk_colour.npz"))
    ys, xs, rgb, r = g["ys"].astype(np.int64), g["xs"].astype(np.int64), g["rgb"].astype(np.float64), g["r"]
    rI, rO = float(g["disk_inner"]), float(g["disk_outer"])
    cfg = dict(scenes.with_res(scenes.DEFAULT, int(g["width"]), int(g["height"])), disk_inner=rI, disk_outer=rO, disk_opacity=1.0,
               disk_hsi=(0.0, 0.0, 1.0), star_intensity=0.0)
    I, hits = trace_intensity(cfg, ys, xs)
    blue = rgb[:, 2]
    ok = blue < 250                                      # unclipped
    t = (rO - r) / (rO - rI)

    def rho(v):
        return float(stats.spearmanr(np.asarray(v)[ok], blue[ok]).statistic)

    lin = np.where(blue / 255 <= 0.04045, blue / 255 / 12.92, ((blue / 255 + 0.055) / 1.055) ** 2.4)
    edges = np.arange(rI, rO + 1e-9, 0.125)
    which = np.digitize(r, edges)
    prof = np.array([np.median(lin[which == k]) if (which == k).sum() > 10 else np.nan for k in range(1, len(edges))])
    top = (edges[:-1] + 0.0625)[prof >= 0.97 * np.nanmax(prof)]
    return {"implementation": rho(I), "single_crossing": bool((np.asarray(hits) == 1).all()),
            "max_dev_from_sin_pi_t2": float(np.abs(np.asarray(I) - np.sin(np.pi * t * t)).max()),
            "sin(pi t^2)": rho(np.sin(np.pi * t * t)), "sin(pi t)": rho(np.sin(np.pi * t)), "sin(pi t^1.5)": rho(np.sin(np.pi * t ** 1.5)),
            "sin(pi t^3)": rho(np.sin(np.pi * t ** 3)), "t reversed": rho(np.sin(np.pi * (1 - t) ** 2)),
            "default.yaml radii 1.8/13": rho(np.sin(np.pi * np.clip((13.0 - r) / (13.0 - 1.8), 0, 1) ** 2)),
            "premultiplied I^2": rho(np.sin(np.pi * t * t) ** 2),
            "best_exponent": float(max(np.arange(1.0, 3.01, 0.1), key=lambda p: rho(np.sin(np.pi * t ** p)))),
            "picture_peak_radius": float(0.5 * (top.min() + top.max())), "law_peak_radius": rO - (rO - rI) / np.sqrt(2.0), "pixels": int(ok.sum())}
