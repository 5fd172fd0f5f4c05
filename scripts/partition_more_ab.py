import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
stars = bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL))
N = 10
for scene, w, h, mode in (("lensing-disk", 3840, 2160, "fast"), ("lensing-disk", 2560, 1440, "fast"), ("default-aa", 1920, 1080, "strict"), ("default-aa", 1280, 720, "strict"), ("closeup", 1920, 1080, "fast"), ("default", 1920, 1080, "fast")):
    cfg = bs.Config.from_file(os.path.join(root, "scenes", scene + ".yaml")).with_resolution(w, h)
    if cfg.scene.bloomStrength == 0: cfg.scene.bloomStrength = 0.4
    rec = {"scene": scene, "frame": f"{w}x{h}", "mode": mode, "ss": cfg.scene.supersampling}
    for setting in ("0", "auto", "8", "16"):
        os.environ["BLACKSTAR_POST_CUS"] = setting
        tree = bs.StarTree(stars); del os.environ["BLACKSTAR_POST_CUS"]
        tree.set_mode(_lib.BS_MODE_STRICT if mode == "strict" else _lib.BS_MODE_FAST)
        bufs = [bs.alloc_image(tree, h, w, dtype=np.uint8) for _ in range(4)]
        outs = [bufs[i % 4] for i in range(N)]
        bs.render_rgb8_batch([cfg] * N, [tree], outs=outs)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); bs.render_rgb8_batch([cfg] * N, [tree], outs=outs); best = min(best, (time.perf_counter() - t0) / N)
        rec[setting] = round(best * 1e3, 3)
        if setting == "auto": rec["auto_post_cus"] = _lib.debug_lib().bs_debug_last_post_cus(tree.handle)
        tree.close()
    print(json.dumps(rec), flush=True)
