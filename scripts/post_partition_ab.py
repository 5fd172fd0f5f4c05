"""A/B of bs_render_rgb8_batch with the chip partitioned between the trace kernels and the post stage (BLACKSTAR_POST_CUS = CUs the
bloom + sRGB8 stream owns; 0 = shared chip, the round-2 pipeline) -- ms per frame over N frames of the C3 scene, bytes compared with the
frame-by-frame bs_render_rgb8.  BLACKSTAR_POST_PLAN_CUS = the CU count the blur sweeps are planned for on that stream."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackstar_amd as bs  # noqa: E402
from blackstar_amd import synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
scene = sys.argv[2] if len(sys.argv) > 2 else "default-aa.yaml"
res = sys.argv[3] if len(sys.argv) > 3 else ""
cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", scene))
if res:
    w, h = res.split("x")
    cfg = cfg.with_resolution(int(w), int(h))
H, W = cfg.scene.resolution[1], cfg.scene.resolution[0]
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
variants = [(v, 0) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["auto", "0", "8", "16", "auto", "0"])]
if len(sys.argv) > 5 and sys.argv[5]:
    cfg.scene.bloomDivider = int(sys.argv[5])
if len(sys.argv) > 6:
    cfg.scene.bloomStrength = float(sys.argv[6])  # 0: the post stage is the sRGB8 map alone -> what the trace kernels cost on the masked streams
ref = None
for post, plan in variants:
    os.environ["BLACKSTAR_POST_CUS"] = str(post)
    os.environ["BLACKSTAR_POST_PLAN_CUS"] = str(plan)
    tree = bs.StarTree(stars)
    ring = [bs.alloc_image(tree, H, W, dtype=np.uint8) for _ in range(4)]
    outs = [ring[i % 4] for i in range(N)]
    if ref is None:
        ref = bs.render_rgb8(cfg, tree).copy()
    bs.render_rgb8_batch([cfg] * N, [tree], outs=outs)
    ok = all(np.array_equal(r, ref) for r in ring)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        bs.render_rgb8_batch([cfg] * N, [tree], outs=outs)
        ts.append((time.perf_counter() - t0) / N * 1e3)
    print(f"{scene} {W}x{H} divider {cfg.scene.bloomDivider} post_cus {post:>4s}: bs_render_rgb8_batch {min(ts):.3f} ms per frame (runs {' '.join(f'{t:.3f}' for t in ts)}), "
          f"{W * H / min(ts) / 1e3:.1f} Mpixel/s, bytes identical to bs_render_rgb8: {ok}", flush=True)
    tree.close()
