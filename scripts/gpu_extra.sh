#!/bin/bash
# Extra measurements for DESIGN.md: C2 / C4 frames, bloom + sRGB8 timings, PCIe-inclusive wall.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - > gpurun_out/extra.txt 2>&1 <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import blackstar_amd as bs
from blackstar_amd import _lib, synthetic
from oracle import scenes
stars = bs.read_map(synthetic.ppm_catalogue_bytes())
tree = bs.StarTree(stars); empty = bs.StarTree(None)
L = _lib.lib()
def run(name, cfg, t, mode, n=6):
    t.set_mode(mode)
    ms=[]; wall=[]
    for _ in range(n):
        bs.render(cfg, t); st=t.stats(); ms.append(st['kernel_ms']); wall.append(st['wall_ms'])
    rays=st['rays']; px=cfg['width']*cfg['height']
    k=float(np.median(ms)); w=float(np.median(wall))
    print(json.dumps({'cfg':name,'mode':'fast' if mode else 'strict','kernel_ms':k,'wall_ms':w,'Mpixel_s_kernel':px/k/1e3,'Mray_s':rays/k/1e3,'Mpixel_s_wall':px/w/1e3,'steps_per_ray':st['steps']/rays,'lane_eff':st['steps']/(64*st['wave_iters'])}))
for mode in (1,0):
    run('C2 default.yaml 1920x1080 no stars', scenes.DEFAULT, empty, mode)
    run('C3 default-aa.yaml 1920x1080 4xSS', scenes.DEFAULT_AA, tree, mode)
    run('C4 lensing-disk.yaml 3840x2160 4xSS', scenes.with_res(scenes.LENSING_DISK, 3840, 2160), tree, mode, n=4)
    run('C5 frame 300/600 1920x1080 4xSS', scenes.ani_frame(300,600), tree, mode)
# post kernels on device
img = torch.from_numpy(bs.render(scenes.DEFAULT_AA, tree)).cuda(); out = torch.empty_like(img); u8 = torch.empty(img.shape, dtype=torch.uint8, device='cuda')
for name, fn in (('bloom', lambda: L.bs_bloom_device(tree.handle, img.data_ptr(), out.data_ptr(), 1920, 1080, 0.15, 25, None)),
                 ('srgb8', lambda: L.bs_srgb8_device(tree.handle, out.data_ptr(), u8.data_ptr(), img.numel(), None))):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(name, '1920x1080 ms', (time.perf_counter()-t0)/5*1e3)
PY
cat gpurun_out/extra.txt
