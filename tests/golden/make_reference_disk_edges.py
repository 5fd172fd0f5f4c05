#!/usr/bin/env python
"""Derives tests/golden/reference_example_disk.npz from /root/reference/example.png (README.md:4) -- the one rendered output the
reference repository holds: the scenes/default.yaml CAMERA at 1280x720 from an unknown earlier revision (real catalogue, its own
disk colour, intensity law and bloom: measured B/R = 0.79 against 0.854 for today's default colour, and a narrower radial
profile -- so it is no pixel golden and cannot pin colour, profile or bloom).  What it can pin is GEOMETRY: overlaying the
oracle's render with the ConfigFile DEFAULT disk radii (diskInner 3, diskOuter 12, src/ConfigFile.hs:76-77) on it, both the
primary image of the disk and its lensed secondary image coincide.  The disk's INNER edge (r = diskInner) is a sharp onset of
light in either intensity law (sin(pi t^2) rises linearly at t -> 1), so its locus -- around the shadow for the primary image,
and the inner boundary of the lensed image -- is comparable to about a pixel.

The fixture is data only: the reference image's luminance R+G+B (0..765) sampled bilinearly on a polar grid around a nominal
centre, r = 100 .. 330 px in steps of 0.5, theta = 0 .. 358 deg in steps of 2.  Run in the build container (reads
/root/reference); the tests read only the .npz."""
import os

import numpy as np
from PIL import Image
from scipy import ndimage as ndi

CX, CY = 725.0, 343.0
R = np.arange(100.0, 330.0 + 1e-9, 0.5)
TH = np.deg2rad(np.arange(0, 360, 2))

ref = np.asarray(Image.open("/root/reference/example.png").convert("RGB")).astype(np.float64)
lum3 = ref.sum(axis=2)
prof = np.stack([ndi.map_coordinates(lum3, [CY + R * np.sin(t), CX + R * np.cos(t)], order=1, mode="constant", cval=65535.0) for t in TH])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_example_disk.npz")
np.savez_compressed(out, cx=CX, cy=CY, r=R, theta=TH, lum3=np.rint(prof).astype(np.uint16), width=1280, height=720,
                    source="flannelhead/blackstar example.png (README.md:4), luminance R+G+B on a polar grid; 65535 = outside the image")
print(out, os.path.getsize(out), "bytes")
