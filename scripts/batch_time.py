import sys, time
sys.path.insert(0, '.')
import blackstar_amd as bs
from blackstar_amd import synthetic
from oracle import scenes
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
cfgs = [scenes.ani_frame(i, 600) for i in range(0, 600, 25)]
bs.render(cfgs[0], tree)
t0 = time.perf_counter(); imgs = bs.render_batch(cfgs, [tree]); t1 = time.perf_counter()
print('render_batch', len(cfgs), 'frames: ms/frame', (t1 - t0) * 1e3 / len(cfgs))
t0 = time.perf_counter()
for c in cfgs: bs.render(c, tree)
t1 = time.perf_counter()
print('frame-by-frame bs_render: ms/frame', (t1 - t0) * 1e3 / len(cfgs))
