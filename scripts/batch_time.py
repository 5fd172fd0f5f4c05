import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs
from blackstar_amd import synthetic
from oracle import scenes
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
cfgs = [scenes.ani_frame(i, 600) for i in range(0, 600, 25)]
bs.render(cfgs[0], tree)
pinned = [bs.alloc_image(tree, 1080, 1920) for _ in cfgs]
for name, outs in (("fresh pageable buffers", None), ("page-locked buffers (bs_host_alloc)", pinned), ("page-locked buffers, 2nd pass", pinned)):
    t0 = time.perf_counter(); bs.render_batch(cfgs, [tree], outs=outs); t1 = time.perf_counter()
    print(f"render_batch {len(cfgs)} frames, {name}: ms/frame {(t1 - t0) * 1e3 / len(cfgs):.3f}")
for name, outs in (("fresh pageable", None), ("page-locked", pinned)):
    t0 = time.perf_counter()
    for i, c in enumerate(cfgs):
        bs.render(c, tree, out=None if outs is None else outs[i])
    t1 = time.perf_counter()
    print(f"frame-by-frame bs_render, {name}: ms/frame {(t1 - t0) * 1e3 / len(cfgs):.3f}")
