#!/usr/bin/env python
"""Derives tests/golden/reference_example_ring.npz from the ONE rendered output the reference repository holds:
/root/reference/example.png (1280x720 RGB8, shown in its README.md:4; the scenes/default.yaml camera, produced by an
unknown revision with the real star catalogue, its own disk colour and bloom).  Not a pixel golden -- but the thin photon
ring inside the shadow is a pure function of the camera model (generateRay: fov convention, look-at basis, aspect) and of the
geodesic integration, so its position IS comparable.

The fixture is data only: the reference image's luminance sampled on a polar grid around a nominal centre (bilinear),
r = 90 .. 140 px in steps of 0.25, theta = 0 .. 356 deg in steps of 4 (uint8-rounded luminance x 3, 90 x 201 values).
Run in the build container (reads /root/reference); the tests read only the .npz."""
import os

import numpy as np
from PIL import Image
from scipy import ndimage as ndi

CX, CY = 725.0, 343.0  # nominal centre of the ring in example.png (pixels); the test fits its own
R = np.arange(90.0, 140.0 + 1e-9, 0.25)
TH = np.deg2rad(np.arange(0, 360, 4))

ref = np.asarray(Image.open("/root/reference/example.png").convert("RGB")).astype(np.float64)
lum3 = ref.sum(axis=2)  # 0 .. 765
prof = np.empty((len(TH), len(R)))
for i, th in enumerate(TH):
    prof[i] = ndi.map_coordinates(lum3, [CY + R * np.sin(th), CX + R * np.cos(th)], order=1)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_example_ring.npz")
np.savez_compressed(out, cx=CX, cy=CY, r=R, theta=TH, lum3=np.rint(prof).astype(np.uint16), width=1280, height=720,
                    source="flannelhead/blackstar example.png (README.md:4), luminance R+G+B on a polar grid")
print(out, os.path.getsize(out), "bytes")
