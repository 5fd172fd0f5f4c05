"""blackstar_amd -- MI355X-native drop-in for the hot path of flannelhead/blackstar: `Raytracer.render`.

The product is blackstar_amd/libblackstar_gpu.so (hand-written HIP for gfx950 behind the C ABI of
include/blackstar_gpu.h).  The Python modules mirror the reference's module interface for that path
(ConfigFile, StarMap, Raytracer, Animation, ImageFilters) on top of the C ABI via ctypes; they contain
no pixel arithmetic of their own and no CPU fallback.
"""
from . import _lib  # noqa: F401
from .animation import Animation, Keyframe, frame_file_name, generate_frames, pad_zero, validate_keyframes, write_frame_files  # noqa: F401
from .batch import files_stats, render_batch, render_png_batch, render_png_files, render_rgb8_batch, render_split  # noqa: F401
from .config_file import Camera, Config, ConfigError, Scene, prepare_scene  # noqa: F401
from .image_filters import bloom, srgb8, supersample  # noqa: F401
from .raytracer import alloc_image, alloc_png, encode_png, png_bound, render_png, render, render_device, render_rgb8, render_rows, render_rows_device, trace_rays, write_img, write_png  # noqa: F401
from .star_map import (StarTree, build_star_tree, read_map, read_map_from_file, read_tree_from_file,  # noqa: F401
                       star_lookup, tree_to_byte_string)

__all__ = ["Animation", "Keyframe", "generate_frames", "validate_keyframes", "pad_zero", "frame_file_name", "write_frame_files", "Camera", "Config", "ConfigError", "Scene",
           "prepare_scene", "bloom", "srgb8", "supersample", "render", "alloc_image", "render_batch", "render_png_batch", "render_png_files", "files_stats", "render_rgb8_batch", "alloc_png", "encode_png", "png_bound", "render_png", "render_split", "render_device", "render_rgb8", "render_rows", "render_rows_device", "trace_rays", "write_img", "write_png", "StarTree", "build_star_tree", "read_map",
           "read_map_from_file", "read_tree_from_file", "star_lookup", "tree_to_byte_string"]
