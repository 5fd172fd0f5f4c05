import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackstar_amd as bs
from blackstar_amd import synthetic
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
cfg = bs.Config.from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scenes', 'default-aa.yaml'))
for i in range(6):
    t0 = time.perf_counter(); img = bs.render_rgb8(cfg, tree); t1 = time.perf_counter()
    print('render_rgb8 ms', (t1 - t0) * 1e3, 'trace kernel', tree.stats()['kernel_ms'])
for i in range(3):
    t0 = time.perf_counter(); img = bs.render(cfg, tree); t1 = time.perf_counter()
    print('render (f64 D2H) ms', (t1 - t0) * 1e3)
