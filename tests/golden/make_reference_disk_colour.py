#!/usr/bin/env python
"""Derives tests/golden/reference_example_disk_colour.npz from /root/reference/example.png (README.md:4) -- a third, PHOTOMETRIC look at
the one rendered output the reference repository holds (after the photon ring and the disk's inner edge, which are geometric).

What the picture can and cannot pin (measured by this script, printed when it runs; DESIGN.md section 4 has the numbers):
  * colour: in the disk R == G to the last bit (median raw G/R = 1.000) and B/R = 0.905 raw -- an HSI hue of exactly 60 deg with
    saturation ~0.15, a diskColor no scene file of today's repository and not the ConfigFile default (57.6 deg, 0.1) has.  The scene is
    unknown, so one free parameter (the hue) absorbs any rotation of massiv-io's HSI sectors: NO PIN of App. B.3.
  * amplitude / shape: brightness goes as I^1.7 of today's I = sin(pi t^2) in all three channels (robust fit over 7 000 unsaturated
    pixels; bloom and the unknown opacity are in it) -- the revision that made the picture squares or pre-multiplies somewhere today's
    blend (src/Raytracer.hs:34-37) does not.  NO PIN of blend / opacity / bloom strength.
  * the ARGUMENT of the intensity law: WHERE along the disk the light peaks and how pixels RANK by brightness does not depend on any
    monotone post-processing.  Over single-crossing pixels (the ray meets the disk exactly once: no lensed second image under it) the
    picture's blue channel (the one that does not clip) ranks like sin(pi t^p), t = (rO - r)/(rO - rI), with Spearman rho = 0.965 at
    p = 2 (maximum 0.967 at p = 2.1), 0.545 at p = 1 (sin(pi t)), 0.89 at 1.5, 0.93 at 3, -0.32 for t reversed, and its profile peaks at
    r = 5.63 (rO - (rO - rI)/sqrt 2 = 5.636; sin(pi t) would peak at 7.5).  THIS is the pin: src/Raytracer.hs:106-110's
    `sin (pi * ((rO - r)/(rO - rI))^2)` with the ConfigFile default radii 3 and 12.

The fixture is data only: 8 000 sampled pixel coordinates of example.png whose ray crosses the disk exactly once, the picture's raw RGB
bytes there, and each pixel's crossing radius r from the independent numpy restatement (oracle/np_oracle.py; the C oracle and the HIP
kernel are then TESTED against the picture through their own intensities at those pixels).  Run in the build container (reads
/root/reference, ~70 s for the 921 600-ray radius map); the tests read only the .npz."""
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage as ndi, stats

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_oracle as O, scenes  # noqa: E402

H, W, N, SEED = 720, 1280, 8000, 2026
cfg = dict(scenes.with_res(scenes.DEFAULT, W, H), disk_inner=3.0, disk_outer=12.0, disk_opacity=0.95)


def radius_map():
    """First disk-crossing radius and number of crossings per pixel: oracle/np_oracle.py's trace loop with the radius kept."""
    sc = O.derive(cfg)
    ys, xs = (a.ravel() for a in np.mgrid[0:H, 0:W])
    vel, pos = O.generate_rays(sc, ys, xs)
    n = len(ys)
    h2 = O.quadrance(O.cross(pos, vel))
    r1, nh, alive = np.full(n, np.nan), np.zeros(n, np.int32), np.ones(n, bool)
    while alive.any():
        idx = np.nonzero(alive)[0]
        v, p = vel[idx], pos[idx]
        nv, npos = O.rk4(sc["h"], h2[idx], v, p)
        r2, r2n, y, yn = O.quadrance(p), O.quadrance(npos), p[:, 1], npos[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            r2ave = (yn * r2 - y * r2n) / (yn - y)
        hor = r2 < 1
        esc = (~hor) & (r2 > sc["safe"])
        dsk = (~hor) & (~esc) & (O._signum(yn) != O._signum(y)) & (r2ave > sc["in2"]) & (r2ave < sc["out2"])
        j = idx[dsk]
        first = nh[j] == 0
        r1[j[first]] = np.sqrt(r2ave[dsk])[first]
        nh[j] += 1
        done = hor | esc
        cont = idx[~done]
        vel[cont], pos[cont] = nv[~done], npos[~done]
        alive[idx[done]] = False
    return r1.reshape(H, W), nh.reshape(H, W)


ref = np.asarray(Image.open("/root/reference/example.png").convert("RGB"))
r1, nh = radius_map()
single = ndi.binary_erosion(nh == 1, iterations=3)   # three pixels clear of any lensed second image
ys, xs = np.nonzero(single)
pick = np.sort(np.random.default_rng(SEED).choice(len(ys), N, replace=False))
ys, xs = ys[pick], xs[pick]
rgb, r = ref[ys, xs], r1[ys, xs]

# ---- what the picture says (printed; the tests re-derive the pinned part from the fixture) ----
f = rgb.astype(np.float64)
lin = np.where(f / 255 <= 0.04045, f / 255 / 12.92, ((f / 255 + 0.055) / 1.055) ** 2.4)
mid = (f[:, 0] > 80) & (f[:, 0] < 250)
print(f"raw G/R median {np.median(f[mid, 1] / f[mid, 0]):.4f}, raw B/R median {np.median(f[mid, 2] / f[mid, 0]):.4f}, "
      f"linear B/R median {np.median(lin[mid, 2] / lin[mid, 0]):.4f} (today's default colour: G/R 0.9931, B/R 0.8542 linear)")
t = (12.0 - r) / 9.0
ok = rgb[:, 2] < 250
for p in (1.0, 1.5, 2.0, 2.1, 3.0):
    print(f"sin(pi t^{p}): Spearman rho {stats.spearmanr(np.sin(np.pi * t ** p)[ok], f[ok, 2]).statistic:.4f}")
print(f"t reversed: {stats.spearmanr(np.sin(np.pi * (1 - t) ** 2)[ok], f[ok, 2]).statistic:.4f}")

out = os.path.join(HERE, "reference_example_disk_colour.npz")
np.savez_compressed(out, ys=ys.astype(np.int16), xs=xs.astype(np.int16), rgb=rgb.astype(np.uint8), r=r, width=W, height=H,
                    disk_inner=3.0, disk_outer=12.0,
                    source="flannelhead/blackstar example.png (README.md:4): raw RGB bytes at 8000 pixels whose ray crosses the disk exactly once "
                           "(scenes/default.yaml camera at 1280x720, ConfigFile default radii 3 / 12), r = crossing radius from oracle/np_oracle.py")
print(out, os.path.getsize(out), "bytes")
