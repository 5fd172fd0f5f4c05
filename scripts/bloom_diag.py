import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
import blackstar_amd as bs
from blackstar_amd import _lib
tree = bs.StarTree(None); L = _lib.lib()
rng = np.random.default_rng(0)
a = rng.uniform(0, 1.5, (1080, 1920, 3))
img = torch.from_numpy(a).cuda(); out = torch.empty_like(img)
fn = lambda: L.bs_bloom_device(tree.handle, img.data_ptr(), out.data_ptr(), 1920, 1080, 0.15, 25, None)
fn(); torch.cuda.synchronize()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("BLACKSTAR_LIB", "main"), f"bloom 1080p {min(ts):.3f} ms")
