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
