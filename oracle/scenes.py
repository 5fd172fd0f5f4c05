"""Resolved inputs of the BASELINE configs, written out by hand from the reference's
scene files and ConfigFile defaults -- independent of the product's YAML parser, which tests compare
against these.  TEST INFRASTRUCTURE ONLY.  Keys are those of bs_config / orc_config.
"""
import copy

_DEFAULT_DISK = (0.16, 0.1, 0.95)  # src/ConfigFile.hs:74 (hue already in [0,1))

# scenes/default.yaml:1-28 (C1, C2)
DEFAULT = dict(cam_pos=(0.0, 1.0, -20.0), cam_lookat=(2.0, 0.0, 0.0), cam_up=(-0.2, 1.0, 0.0), fov=1.5,
               step_size=0.3, star_intensity=0.4, star_saturation=1.5, disk_hsi=(180.0 / 360, 0.1, 1.05),
               disk_opacity=0.95, disk_inner=1.8, disk_outer=13.0, width=1920, height=1080, supersampling=False)
# scenes/default-aa.yaml:1-16 (C3)
DEFAULT_AA = dict(DEFAULT, supersampling=True)
# scenes/lensing-disk.yaml:1-15 (C4; BASELINE overrides the resolution to 3840x2160)
LENSING_DISK = dict(cam_pos=(30.0, 0.4, 3.0), cam_lookat=(0.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.2), fov=1.0,
                    step_size=0.3, star_intensity=0.4, star_saturation=1.5, disk_hsi=_DEFAULT_DISK,
                    disk_opacity=0.95, disk_inner=3.0, disk_outer=12.0, width=1280, height=800, supersampling=True)
# animations/default-ani.yaml:7-36 scene + keyframe 0 / keyframe 1 cameras (C5); 'diskHSV' is ignored -> default colour
ANI_SCENE = dict(step_size=0.3, star_intensity=0.7, star_saturation=0.7, disk_hsi=_DEFAULT_DISK, disk_opacity=0.95,
                 disk_inner=1.8, disk_outer=13.0, width=1920, height=1080, supersampling=True)
ANI_KEY0 = dict(ANI_SCENE, cam_pos=(3.0, 3.0, -20.0), cam_lookat=(-7.0, 5.0, 0.0), cam_up=(-0.2, 1.0, 0.0), fov=1.5)
ANI_KEY1 = dict(ANI_SCENE, cam_pos=(-15.0, 1.0, -20.0), cam_lookat=(13.0, -7.0, 0.0), cam_up=(-0.2, 1.0, 0.0), fov=2.0)

# ---- the reference's other six scene files (/root/reference/scenes/*.yaml), resolved by hand with src/ConfigFile.hs:66-79's
# defaults (stepSize 0.3, diskColor HSI(0.16, 0.1, 0.95), diskInner 3, diskOuter 12, supersampling False).  They are the
# reference's own edge cases: a camera exactly in the disk plane (y = 0) with diskOpacity 0 (wideangle, wideangle1), fov 3.5
# (wideangle-disk), 2|cam|^2 = 8452 > 2500 -> camera-dependent safeDistance (fartheraway), no supersampling + 4:3 (closeup).
def _scene(pos, look, up, fov, w, h, ss, opacity, inner=3.0, outer=12.0, si=0.4, sat=1.5):
    return dict(cam_pos=tuple(map(float, pos)), cam_lookat=tuple(map(float, look)), cam_up=tuple(map(float, up)), fov=float(fov),
                step_size=0.3, star_intensity=si, star_saturation=sat, disk_hsi=_DEFAULT_DISK, disk_opacity=opacity,
                disk_inner=inner, disk_outer=outer, width=w, height=h, supersampling=ss)


CLOSEUP = _scene((10, 1, -2), (0, 0, 6), (0, 1, 0), 1.2, 1280, 960, False, 0.95, 3.0, 9.0, si=0.7, sat=0.7)   # closeup.yaml:1-13
FARTHERAWAY = _scene((-25, 1, -60), (-12, -4, 0), (0.15, 1, 0), 2.0, 1920, 1080, True, 0.95)                  # fartheraway.yaml:1-14
LENSING = _scene((30, 0.4, 3), (0, 0, 0), (0, 1, 0.2), 1.0, 1600, 1200, True, 0.0)                            # lensing.yaml:1-14
WIDEANGLE_DISK = _scene((-6, 1, -20), (-6, -4, 0), (-0.2, 1, 0), 3.5, 1920, 1080, True, 0.95, 2.5, 12.0)      # wideangle-disk.yaml:1-14
WIDEANGLE = _scene((20, 0, 0), (0, 0, 3.5), (0, 1, 0), 2.0, 1920, 1020, True, 0.0)                            # wideangle.yaml:1-12
WIDEANGLE1 = _scene((0, 0, 20), (3.5, 0, 0), (0, 1, 0), 2.0, 1920, 1080, True, 0.0)                           # wideangle1.yaml:1-12

# every scene file the reference ships, by file name
REFERENCE_SCENES = {"default": DEFAULT, "default-aa": DEFAULT_AA, "lensing-disk": LENSING_DISK, "closeup": CLOSEUP,
                    "fartheraway": FARTHERAWAY, "lensing": LENSING, "wideangle-disk": WIDEANGLE_DISK, "wideangle": WIDEANGLE,
                    "wideangle1": WIDEANGLE1}
EXTRA_SCENES = ("closeup", "fartheraway", "lensing", "wideangle-disk", "wideangle", "wideangle1")
# bloom parameters of the same files (not inputs of render; used by the bs_render_rgb8 tests): (strength, divider)
REFERENCE_BLOOM = {"default": (0.15, 25), "default-aa": (0.15, 25), "lensing-disk": (0.15, 25), "closeup": (0.7, 25),
                   "fartheraway": (0.15, 25), "lensing": (0.15, 25), "wideangle-disk": (0.15, 25), "wideangle": (0.15, 25),
                   "wideangle1": (0.15, 25)}


def with_res(cfg, w, h, ss=None):
    c = copy.deepcopy(cfg)
    c["width"], c["height"] = int(w), int(h)
    if ss is not None:
        c["supersampling"] = bool(ss)
    return c


def ani_frame(i, n):
    """Camera of frame i of n (src/Animation.hs:45-86 with the two keyframes of default-ani.yaml)."""
    t = float(i) * (1.0 / float(n - 1))
    c = copy.deepcopy(ANI_KEY0)
    if t < 1.0:
        tp = (t - 0.0) / (1.0 - 0.0)
        for k in ("cam_pos", "cam_lookat", "cam_up"):
            c[k] = tuple(a + tp * (b - a) for a, b in zip(ANI_KEY0[k], ANI_KEY1[k]))
        c["fov"] = ANI_KEY0["fov"] + tp * (ANI_KEY1["fov"] - ANI_KEY0["fov"])
    else:  # findFrames [fr] = (fr, fr{time+1}) -> t' = 0 -> last keyframe
        c = copy.deepcopy(ANI_KEY1)
    return c
