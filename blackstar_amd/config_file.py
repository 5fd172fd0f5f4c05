"""Scene / camera configuration -- host-side mirror of the reference's ConfigFile module.

Mirrors /root/reference/src/ConfigFile.hs:16-84: the `Config{scene, camera}` records, the YAML/JSON
decoding rules (Scene fields optional with defaults :66-79, Camera and Config fields all required
:56,:61, unknown keys ignored, V3 as [x,y,z] :40-43, HSI as [deg,s,i] with hue/360 :48-51) and the
field names, so a scene file written for the reference decodes to the same values here.

`Config.to_bs_config()` produces the POD the C ABI takes (include/blackstar_gpu.h:bs_config), i.e. the
config *as parsed* -- `render` derives squared radii, safeDistance and the doubled resolution itself,
exactly like Raytracer.render (src/Raytracer.hs:57-64).
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Any, Tuple

import yaml


class ConfigError(ValueError):
    """Decoding failure (the reference's `Left err` from decodeFileEither, app/Main.hs:85-91)."""


Vec3 = Tuple[float, float, float]


class _Loader(yaml.SafeLoader):
    """PyYAML resolves plain scalars by YAML 1.1, where a float needs a dot: `1e-1` or `5E3` stay strings.  The reference's
    decoder (Data.Yaml = libyaml + aeson's number parser) reads them as numbers.  The 1.2-style float form is therefore added as an
    IMPLICIT resolver -- it applies to plain scalars only, so an explicitly quoted scalar (`fov: "1.5"`) stays a string and is
    rejected, like aeson rejects a String where a number is expected."""


import re as _re  # noqa: E402

_Loader.add_implicit_resolver("tag:yaml.org,2002:float", _re.compile(r"^[-+]?(\.[0-9]+|[0-9]+(\.[0-9]*)?)[eE][-+]?[0-9]+$"), list("-+0123456789."))


def load_yaml(text: str) -> Any:
    return yaml.load(text, Loader=_Loader)  # noqa: S506 -- a SafeLoader subclass


def _num(v: Any, what: str) -> float:
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise ConfigError(f"{what}: expected a number, got {v!r}")
    return float(v)


def _int(v: Any, what: str) -> int:
    """An aeson `Int` field: any JSON number with an integral value (25, 25.0, 2.5e1) decodes; 25.5, .inf and 1e999 do not."""
    if isinstance(v, bool) or not isinstance(v, (int, float)) or not math.isfinite(v) or float(v) != int(v):
        raise ConfigError(f"{what}: expected an Int, got {v!r}")
    return int(v)


def _vec3(v: Any, what: str) -> Vec3:
    # instance FromJSON (V3 Double): [x, y, z] <- parseJSON  (ConfigFile.hs:40-43)
    if not isinstance(v, (list, tuple)) or len(v) != 3:
        raise ConfigError(f"{what}: expected [x, y, z], got {v!r}")
    return (_num(v[0], what), _num(v[1], what), _num(v[2], what))


@dataclass
class Camera:
    """ConfigFile.hs:34-38.  All four fields are required (generic FromJSON, :61)."""
    position: Vec3
    lookAt: Vec3
    upVec: Vec3
    fov: float

    @staticmethod
    def decode(obj: Any) -> "Camera":
        if not isinstance(obj, dict):
            raise ConfigError(f"camera: expected an object, got {type(obj).__name__}")
        for k in ("position", "lookAt", "upVec", "fov"):
            if k not in obj:
                raise ConfigError(f"camera: key {k!r} not present")
        return Camera(_vec3(obj["position"], "camera.position"), _vec3(obj["lookAt"], "camera.lookAt"),
                      _vec3(obj["upVec"], "camera.upVec"), _num(obj["fov"], "camera.fov"))

    def encode(self) -> dict:
        return {"position": list(self.position), "lookAt": list(self.lookAt), "upVec": list(self.upVec), "fov": self.fov}


@dataclass
class Scene:
    """ConfigFile.hs:20-32 with the defaults of :66-79.  diskColor is HSI with hue in [0,1)."""
    safeDistance: float = 0.0  # never read from YAML (:67); render overwrites it (Raytracer.hs:59-60)
    stepSize: float = 0.3
    bloomStrength: float = 0.4
    bloomDivider: int = 25
    starIntensity: float = 0.7
    starSaturation: float = 0.7
    diskColor: Vec3 = (0.16, 0.1, 0.95)
    diskOpacity: float = 0.0
    diskInner: float = 3.0
    diskOuter: float = 12.0
    resolution: Tuple[int, int] = (1280, 720)
    supersampling: bool = False

    @staticmethod
    def decode(obj: Any) -> "Scene":
        if not isinstance(obj, dict):  # parseJSON invalid = typeMismatch "Object" (:81)
            raise ConfigError(f"scene: expected Object, got {type(obj).__name__}")
        s = Scene()
        for k in ("stepSize", "bloomStrength", "starIntensity", "starSaturation", "diskOpacity", "diskInner", "diskOuter"):
            if obj.get(k) is not None:  # (.:?) treats an explicit null like a missing key
                setattr(s, k, _num(obj[k], f"scene.{k}"))
        if obj.get("bloomDivider") is not None:
            s.bloomDivider = _int(obj["bloomDivider"], "scene.bloomDivider")
        if obj.get("diskColor") is not None:
            x, y, z = _vec3(obj["diskColor"], "scene.diskColor")
            s.diskColor = (x / 360, y, z)  # PixelHSI (x / 360) y z  (:51)
        if obj.get("resolution") is not None:
            r = obj["resolution"]
            if not isinstance(r, (list, tuple)) or len(r) != 2:
                raise ConfigError(f"scene.resolution: expected [width, height] of Int, got {r!r}")
            s.resolution = (_int(r[0], "scene.resolution"), _int(r[1], "scene.resolution"))
        if obj.get("supersampling") is not None:
            if not isinstance(obj["supersampling"], bool):
                raise ConfigError(f"scene.supersampling: expected Bool, got {obj['supersampling']!r}")
            s.supersampling = obj["supersampling"]
        return s

    def encode(self) -> dict:
        h, s_, i = self.diskColor
        return {"safeDistance": self.safeDistance, "stepSize": self.stepSize, "bloomStrength": self.bloomStrength,
                "bloomDivider": self.bloomDivider, "starIntensity": self.starIntensity, "starSaturation": self.starSaturation,
                "diskColor": [360 * h, s_, i], "diskOpacity": self.diskOpacity, "diskInner": self.diskInner,
                "diskOuter": self.diskOuter, "resolution": list(self.resolution), "supersampling": self.supersampling}


@dataclass
class Config:
    """ConfigFile.hs:16-18.  Both keys are required (generic FromJSON, :56)."""
    scene: Scene = field(default_factory=Scene)
    camera: Camera = None  # type: ignore[assignment]

    @staticmethod
    def decode(obj: Any) -> "Config":
        if not isinstance(obj, dict):
            raise ConfigError(f"config: expected an object, got {type(obj).__name__}")
        for k in ("scene", "camera"):
            if k not in obj:
                raise ConfigError(f"config: key {k!r} not present")
        return Config(scene=Scene.decode(obj["scene"]), camera=Camera.decode(obj["camera"]))

    @staticmethod
    def from_yaml(text: str) -> "Config":
        try:
            obj = load_yaml(text)
        except yaml.YAMLError as e:
            raise ConfigError(str(e)) from e
        return Config.decode(obj)

    @staticmethod
    def from_file(path: str) -> "Config":
        try:
            with open(path, "r", encoding="utf-8") as f:
                return Config.from_yaml(f.read())
        except OSError as e:
            raise ConfigError(str(e)) from e

    def encode(self) -> dict:
        return {"scene": self.scene.encode(), "camera": self.camera.encode()}

    def to_yaml(self) -> str:
        return yaml.safe_dump(self.encode(), default_flow_style=None, sort_keys=False)

    def with_resolution(self, width: int, height: int) -> "Config":
        c = copy.deepcopy(self)
        c.scene.resolution = (int(width), int(height))
        return c

    def to_bs_config(self) -> dict:
        """Flat dict with the field names of bs_config (include/blackstar_gpu.h); config as parsed."""
        s, c = self.scene, self.camera
        return {"cam_pos": tuple(c.position), "cam_lookat": tuple(c.lookAt), "cam_up": tuple(c.upVec), "fov": c.fov,
                "step_size": s.stepSize, "star_intensity": s.starIntensity, "star_saturation": s.starSaturation,
                "disk_hsi": tuple(s.diskColor), "disk_opacity": s.diskOpacity, "disk_inner": s.diskInner,
                "disk_outer": s.diskOuter, "width": s.resolution[0], "height": s.resolution[1],
                "supersampling": bool(s.supersampling)}


def prepare_scene(cfg: Config, do_preview: bool) -> Config:
    """app/Main.hs:93-103 prepareScene: preview = 300-px long side, no supersampling, no bloom."""
    if not do_preview:
        return cfg
    c = copy.deepcopy(cfg)
    w, h = c.scene.resolution
    res = 300
    c.scene.resolution = (res, res * h // w) if w >= h else (res * w // h, res)
    c.scene.supersampling = False
    c.scene.bloomStrength = 0.0
    return c
