#!/usr/bin/env python
"""bs_render_png_batch with the post stage (bloom + sRGB8 + PNG encoder) on its own CUs: ms per frame for BLACKSTAR_POST_CUS = 0 (shared
chip), auto (the cost model's choice, printed) and 8 / 16 / 24 forced (AB_SETTINGS=0,auto,8,12,... for another list; AB_FORM=rgb8 for bs_render_rgb8_batch), on a few frame shapes; page-locked file buffers, N frames,
best of 3 calls.  Usage: png_partition_ab.py [N_FRAMES]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stars = bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL))
rows = []
for scene, w, h, bloom in (("default-aa", 1920, 1080, 0.4), ("default-aa", 1920, 1080, 0.0), ("default-aa", 1280, 720, 0.4),
                           ("default-aa", 1280, 720, 0.0), ("default-aa", 2560, 1440, 0.4), ("lensing-disk", 1920, 1080, 0.4)):
    cfg = bs.Config.from_file(os.path.join(root, "scenes", scene + ".yaml")).with_resolution(w, h)
    cfg.scene.bloomStrength = bloom
    rec = {"scene": scene, "frame": f"{w}x{h}", "bloom": bloom, "form": os.environ.get("AB_FORM", "png")}
    for setting in os.environ.get("AB_SETTINGS", "0,auto,8,16,24").split(","):
        os.environ["BLACKSTAR_POST_CUS"] = setting
        tree = bs.StarTree(stars)
        del os.environ["BLACKSTAR_POST_CUS"]
        rgb8 = os.environ.get("AB_FORM", "png") == "rgb8"
        bufs = [bs.alloc_image(tree, h, w, dtype=np.uint8) if rgb8 else bs.alloc_png(tree, h, w) for _ in range(4)]
        outs = [bufs[i % 4] for i in range(N)]
        fn = bs.render_rgb8_batch if rgb8 else bs.render_png_batch
        fn([cfg] * N, [tree], outs=outs)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            fn([cfg] * N, [tree], outs=outs)
            best = min(best, (time.perf_counter() - t0) / N)
        rec[setting] = round(best * 1e3, 3)
        if setting == "auto":
            rec["auto_post_cus"] = _lib.debug_lib().bs_debug_last_post_cus(tree.handle)
        tree.close()
    rows.append(rec)
    print(json.dumps(rec), flush=True)
