"""Animation keyframes -- host-side mirror of the reference's Animation module (src/Animation.hs).

`generate_frames` restates src/Animation.hs:45-86: nFrames cameras by linear interpolation of fov,
position, lookAt and upVec between time-sorted keyframes, t_i = i * (1 / (nFrames - 1)).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any, List

import yaml

from .config_file import Camera, Config, ConfigError, Scene, load_yaml


@dataclass
class Keyframe:
    camera: Camera
    time: float

    @staticmethod
    def decode(obj: Any) -> "Keyframe":
        if not isinstance(obj, dict) or "camera" not in obj or "time" not in obj:
            raise ConfigError("keyframe: keys 'camera' and 'time' are required")
        t = obj["time"]
        if isinstance(t, bool) or not isinstance(t, (int, float)):
            raise ConfigError(f"keyframe.time: expected a number, got {t!r}")
        return Keyframe(Camera.decode(obj["camera"]), float(t))


@dataclass
class Animation:
    scene: Scene
    nFrames: int
    interpolation: str  # any string parses as Linear (src/Animation.hs:29-34)
    keyframes: List[Keyframe]

    @staticmethod
    def decode(obj: Any) -> "Animation":
        if not isinstance(obj, dict):
            raise ConfigError("animation: expected an object")
        for k in ("scene", "nFrames", "interpolation", "keyframes"):
            if k not in obj:
                raise ConfigError(f"animation: key {k!r} not present")
        if isinstance(obj["nFrames"], bool) or not isinstance(obj["nFrames"], int):
            raise ConfigError("animation.nFrames: expected Int")
        if not isinstance(obj["interpolation"], str):
            raise ConfigError("animation.interpolation: expected String")
        if not isinstance(obj["keyframes"], list):
            raise ConfigError("animation.keyframes: expected a list")
        return Animation(Scene.decode(obj["scene"]), int(obj["nFrames"]), "linear", [Keyframe.decode(k) for k in obj["keyframes"]])

    @staticmethod
    def from_file(path: str) -> "Animation":
        try:
            with open(path, "r", encoding="utf-8") as f:
                return Animation.decode(load_yaml(f.read()))
        except (OSError, yaml.YAMLError) as e:
            raise ConfigError(str(e)) from e


def validate_keyframes(frs: List[Keyframe]) -> None:
    """src/Animation.hs:38-43 (raises instead of returning Left)."""
    if len(frs) < 2:
        raise ConfigError("Must have at least two keyframes")
    if not (frs[0].time == 0 and frs[-1].time == 1):
        raise ConfigError("First keyframe must have time == 0, last time == 1")


def _interpolate(frames: List[Keyframe], t: float) -> Camera:
    # findFrames (src/Animation.hs:63-66): first adjacent pair with time fr1 <= t < time fr2; past the end -> (last, last{time+1})
    f1 = f2 = None
    for a, b in zip(frames, frames[1:]):
        if t >= a.time and t < b.time:
            f1, f2 = a, b
            break
    if f1 is None:
        f1 = frames[-1]
        f2 = Keyframe(f1.camera, f1.time + 1)
    tp = (t - f1.time) / (f2.time - f1.time)

    def lerp(a, b):  # a + t `times` (b - a)  (src/Animation.hs:86)
        return a + tp * (b - a)

    def lerp3(a, b):
        return tuple(lerp(x, y) for x, y in zip(a, b))

    c1, c2 = f1.camera, f2.camera
    return Camera(position=lerp3(c1.position, c2.position), lookAt=lerp3(c1.lookAt, c2.lookAt),
                  upVec=lerp3(c1.upVec, c2.upVec), fov=lerp(c1.fov, c2.fov))


def generate_frames(animation: Animation) -> List[Config]:
    stepsize = 1.0 / float(animation.nFrames - 1)
    frames = sorted(animation.keyframes, key=lambda k: k.time)  # sortBy (comparing time) is stable, like sorted()
    points = [float(i) * stepsize for i in range(animation.nFrames)]
    return [Config(scene=copy.deepcopy(animation.scene), camera=_interpolate(frames, p)) for p in points]


def _wrap64(v: int) -> int:
    """Int arithmetic of a 64-bit GHC: two's-complement wrap-around."""
    return (v + (1 << 63)) % (1 << 64) - (1 << 63)


def _n_digits(x: int) -> int:
    """nDigits of padZero (src/Util.hs:45): (floor . logBase 10 $ fromIntegral x) + 1 in Int.  logBase 10 x = log x / log 10 in Double,
    so 1000 has "3 digits" (2.9999999999999996) like in GHC; for x = 0 the logarithm is -Infinity and `floor :: Double -> Int` gives
    what x86-64 GHC's floorDoubleInt gives: cvttsd2si's minBound, minus one because -inf < minBound, wrapped -- RECALLED, like SURVEY
    Appendix F.7 says; tools/ghc_pin/Dump.hs writes padzero.txt and tests/ghc_pin.py compares it, index 0 included, when it arrives."""
    import math
    if x > 0:
        fl = math.floor(math.log(float(x)) / math.log(10.0))
    elif x == 0:
        fl = _wrap64(-(1 << 63) - 1)   # floor (-Infinity) :: Int
    else:
        fl = -(1 << 63)                # floor NaN :: Int (cvttsd2si's "indefinite"; NaN < n is False)
    return _wrap64(fl + 1)


def pad_zero(max_val: int, val: int) -> str:
    """src/Util.hs:43-48 padZero maxVal val: `val` left-padded with zeros to the digit count of maxVal -- as the reference computes it,
    quirks included: index 0 is NOT padded (nDigits 0 is hugely negative, the difference wraps negative, replicate of a negative count
    is empty), and a power of ten whose logBase 10 falls just short (1000) counts one digit less."""
    n_zeros = _wrap64(_n_digits(max_val) - _n_digits(val))
    return "0" * max(n_zeros, 0) + str(val)


def frame_file_name(basename: str, n_frames: int, idx: int, ext: str = ".yaml") -> str:
    """The file `animate` writes frame idx of an nFrames animation to (app/Animate.hs:55-56): basename ++ "_" ++ padZero (nFr - 1) idx
    <.> ".yaml" -- which the batch loop (app/Main.hs:68-77) renders to the same name with .png.  With the reference's padding."""
    return f"{basename}_{pad_zero(n_frames - 1, idx)}{ext}"


def write_frame_files(animation: Animation, out_dir: str, basename: str) -> List[str]:
    """What `animate` leaves behind (app/Animate.hs:47-61 with --force): one scene file per frame, `<basename>_<padZero (nFrames-1) i>.yaml`
    = Data.Yaml.encode of the frame's Config (src/ConfigFile.hs:45-46, :52-53: V3 as [x, y, z], diskColor with its hue back in degrees) --
    files the reference's own `blackstar` directory mode (app/Main.hs:68-77) and this library's render_scene_directory both read.
    Returns the paths in frame order."""
    import os
    validate_keyframes(animation.keyframes)
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for i, cfg in enumerate(generate_frames(animation)):
        path = os.path.join(out_dir, frame_file_name(basename, animation.nFrames, i))
        with open(path, "w", encoding="utf-8") as f:
            f.write(cfg.to_yaml())
        paths.append(path)
    return paths
