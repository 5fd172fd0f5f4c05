import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import blackstar_amd as bs
from blackstar_amd import synthetic
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
N = 30
for post in ("auto", "0", "auto", "0"):
    os.environ["BLACKSTAR_POST_CUS"] = post
    tree = bs.StarTree(stars)
    ring = [np.zeros((1080, 1920, 3), np.uint8) for _ in range(4)]  # pageable, touched
    outs = [ring[i % 4] for i in range(N)]
    bs.render_rgb8_batch([cfg] * N, [tree], outs=outs)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); bs.render_rgb8_batch([cfg] * N, [tree], outs=outs); ts.append((time.perf_counter() - t0) / N * 1e3)
    print(f"pageable outputs, post_cus {post:>4s}: {min(ts):.3f} ms per frame ({' '.join(f'{t:.3f}' for t in ts)})", flush=True)
    tree.close()
