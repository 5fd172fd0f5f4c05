#!/usr/bin/env python
"""Where does the post-stage partition stop paying?  bs_render_rgb8_batch and bs_render_png_batch on frames ABOVE 1080p (default-aa
camera, bloom 0.4, divider 25, 4x supersampled), BLACKSTAR_POST_CUS = 0 / 8 / 16 / 24 / auto, page-locked outputs, N frames, best of 3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stars = bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL))
for w, h in ((2240, 1260), (2560, 1440), (3200, 1800), (3840, 2160)):
    cfg = bs.Config.from_file(os.path.join(root, "scenes", "default-aa.yaml")).with_resolution(w, h)
    for form in ("rgb8", "png"):
        rec = {"frame": f"{w}x{h}", "form": form}
        for setting in ("0", "auto", "8", "16", "24"):
            os.environ["BLACKSTAR_POST_CUS"] = setting
            tree = bs.StarTree(stars)
            del os.environ["BLACKSTAR_POST_CUS"]
            if form == "png":
                bufs = [bs.alloc_png(tree, h, w) for _ in range(4)]
                fn = bs.render_png_batch
            else:
                bufs = [bs.alloc_image(tree, h, w, dtype=np.uint8) for _ in range(4)]
                fn = bs.render_rgb8_batch
            outs = [bufs[i % 4] for i in range(N)]
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
        print(json.dumps(rec), flush=True)
