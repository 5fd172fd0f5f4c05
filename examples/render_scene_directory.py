#!/usr/bin/env python
"""EXAMPLE, not product: the reference's directory batch mode (app/Main.hs:64-77 over handleScene :80-91 and doRender :105-123) driven
through this library's C ABI.  The CLI side of blackstar is out of scope (it stays in Haskell and calls the
boundary); this file only shows that the batch loop needs nothing but `bs_render_png_files`.

    python examples/render_scene_directory.py SCENES_DIR OUT_DIR [--preview] [--catalogue PPM_FILE]
"""
from __future__ import annotations

import os
import sys
from typing import List, Sequence

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from blackstar_amd.batch import render_png_files  # noqa: E402
from blackstar_amd.config_file import Config  # noqa: E402
from blackstar_amd.star_map import StarTree  # noqa: E402


def render_scene_directory(in_dir: str, out_dir: str, trees: Sequence[StarTree], preview: bool = False, pipe: int = 16) -> List[str]:
    """The reference's batch mode (app/Main.hs:64-77 over handleScene :80-91 and doRender :105-123) without its terminal: every `*.yaml`
    of `in_dir`, in sorted order, decoded like `decodeFileEither`, `prepareScene`d (preview: 300-px long side, no supersampling, no
    bloom, name prefixed `prev-`), rendered / bloomed / mapped to sRGB8 / PNG-encoded on the device and written by the library
    (`bs_render_png_files`, scene i on trees[i % len(trees)]) to `<out_dir>/<scene name>.png` (existing files are overwritten: the reference's --force).
    A scene file that does not decode is reported like the reference does -- its error is printed, the others are rendered.
    Returns the paths written."""
    from blackstar_amd.config_file import ConfigError, prepare_scene
    names = sorted(f for f in os.listdir(in_dir) if os.path.splitext(f)[1] == ".yaml")
    cfgs, outs = [], []
    for f in names:
        try:
            cfg = Config.from_file(os.path.join(in_dir, f))
        except (ConfigError, OSError, ValueError) as e:
            print(f"{os.path.join(in_dir, f)}: {e}", file=sys.stderr)
            continue
        cfgs.append(prepare_scene(cfg, preview))
        outs.append(os.path.join(out_dir, ("prev-" if preview else "") + os.path.splitext(f)[0] + ".png"))
    os.makedirs(out_dir, exist_ok=True)
    render_png_files(cfgs, trees, outs, pipe=pipe)
    return outs



if __name__ == "__main__":
    import argparse

    import blackstar_amd as bs
    from blackstar_amd import synthetic
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("scenes_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--preview", action="store_true")
    ap.add_argument("--catalogue", default="synthetic")
    a = ap.parse_args()
    trees = [StarTree(bs.read_map(synthetic.catalogue_bytes(a.catalogue)), device=d) for d in range(max(1, bs._lib.lib().bs_device_count()))]
    for p in render_scene_directory(a.scenes_dir, a.out_dir, trees, preview=a.preview):
        print(p)
