#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5) > gpurun_out/pytest_gpu.log 2>&1
python - > gpurun_out/png.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import blackstar_amd as bs
from blackstar_amd import synthetic
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes()))
cfg = bs.Config.from_file('scenes/default-aa.yaml').with_resolution(1280, 720)
t0=time.perf_counter(); img = bs.render_rgb8(cfg, tree); print('render_rgb8 1280x720 ms', (time.perf_counter()-t0)*1e3)
bs.write_png(img, 'gpurun_out/default-aa_1280x720.png')
cfg = bs.Config.from_file('scenes/default-aa.yaml')
for _ in range(3):
    t0=time.perf_counter(); img = bs.render_rgb8(cfg, tree); print('render_rgb8 1920x1080 (render+bloom+srgb8+D2H 6.2MB) ms', (time.perf_counter()-t0)*1e3, tree.stats()['kernel_ms'])
cfg = bs.Config.from_file('scenes/lensing-disk.yaml')
bs.write_png(bs.render_rgb8(cfg, tree), 'gpurun_out/lensing-disk_1280x800.png')
PY
cat gpurun_out/pytest_gpu.log gpurun_out/png.log
