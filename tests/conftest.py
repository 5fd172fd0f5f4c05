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
