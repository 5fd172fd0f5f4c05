#!/usr/bin/env python
"""The device PNG encoder on the C3 frame (default-aa.yaml, 1920x1080, 4x supersampled, 470k-star sky): file size against
libpng + zlib (Pillow) at levels 1 and 6 with the host time those take, the encoder's own time, and bs_render_png_batch against
bs_render_rgb8_batch per frame (page-locked outputs, 20 frames).  Usage: png_probe.py [N_FRAMES]   (run under rocprofv3 --kernel-trace
--stats for the per-kernel times)"""
import io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL)))
out = {}
for scene, w, h in (("default-aa", 1920, 1080), ("lensing-disk", 3840, 2160)):
    cfg = bs.Config.from_file(os.path.join(root, "scenes", scene + ".yaml")).with_resolution(w, h)
    rgb8 = bs.render_rgb8(cfg, tree)
    buf = bs.alloc_png(tree, h, w)
    data = bytes(bs.encode_png(rgb8, tree, out=buf))
    t0 = time.perf_counter()
    for _ in range(10):
        bs.encode_png(rgb8, tree, out=buf)
    t_enc = (time.perf_counter() - t0) / 10
    rec = {"pixels_bytes": int(rgb8.size), "file_bytes": len(data), "ratio": rgb8.size / len(data), "bs_encode_png_ms_incl_h2d": t_enc * 1e3,
           "zero_fraction": float((rgb8 == 0).mean())}
    try:
        from PIL import Image
        assert np.array_equal(np.array(Image.open(io.BytesIO(data)).convert("RGB")), rgb8)
        rec["decodes_to_the_frame"] = True
        for level in (1, 6):
            b = io.BytesIO()
            t0 = time.perf_counter()
            Image.fromarray(rgb8).save(b, format="PNG", compress_level=level)
            rec[f"pillow_level{level}"] = {"bytes": len(b.getvalue()), "host_ms": (time.perf_counter() - t0) * 1e3}
    except ImportError:
        pass
    if scene == "default-aa":
        cfgs = [cfg] * N
        pix = [bs.alloc_image(tree, h, w, dtype=np.uint8) for _ in range(4)]
        png = [bs.alloc_png(tree, h, w) for _ in range(4)]
        for name, fn, bufs in (("rgb8_batch", bs.render_rgb8_batch, pix), ("png_batch", bs.render_png_batch, png)):
            outs = [bufs[i % 4] for i in range(N)]
            fn(cfgs, [tree], outs=outs)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                res = fn(cfgs, [tree], outs=outs)
                best = min(best, (time.perf_counter() - t0) / N)
            rec[name + "_ms_per_frame"] = best * 1e3
        rec["png_batch_file_is_encode_png_file"] = bytes(res[-1]) == data
        for m in ("8", "16"):
            os.environ["BLACKSTAR_POST_CUS"] = m
            t2 = bs.StarTree(tree.stars)
            del os.environ["BLACKSTAR_POST_CUS"]
            png2 = [bs.alloc_png(t2, h, w) for _ in range(4)]
            outs = [png2[i % 4] for i in range(N)]
            bs.render_png_batch(cfgs, [t2], outs=outs)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                res = bs.render_png_batch(cfgs, [t2], outs=outs)
                best = min(best, (time.perf_counter() - t0) / N)
            rec[f"png_batch_post_cus_{m}_ms_per_frame"] = best * 1e3
            t2.close()
    out[f"{scene} {w}x{h}"] = rec
print(json.dumps(out, indent=1))
