#!/usr/bin/env python
"""BASELINE configs[4] (animations/default-ani.yaml: 1920x1080, 4x supersampled, bloom) END TO END, files on disk: N frames written as PNG
files (a) by write_animation -- rendered, bloomed, quantised and PNG-encoded on the GPU (bs_render_png_batch), the host only write(2)s --
and (b) the way round 2 did it and the reference does: pixels to the host (bs_render_rgb8_batch), zlib level 6 on a pool of host threads
(the reference: JuicyPixels on one).  Files go to /dev/shm (no disk in the way).  Usage: png_animation_probe.py [N_FRAMES [THREADS]]"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blackstar_amd as bs
from blackstar_amd import synthetic
from blackstar_amd.distributed import write_animation
from concurrent.futures import ThreadPoolExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
THREADS = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(16, (os.cpu_count() or 2) - 1))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tree = bs.StarTree(bs.read_map(synthetic.ppm_catalogue_bytes(synthetic.N_FULL)))
anim = bs.Animation.from_file(os.path.join(root, "animations", "default-ani.yaml"))
anim.nFrames = N
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
out = {"frames": N, "frame": "%dx%d" % tuple(anim.scene.resolution), "host_threads_for_zlib": THREADS, "host_cores": os.cpu_count()}

d = tempfile.mkdtemp(dir=base)
try:
    write_animation(anim, tree, d, basename="w")            # warm-up: contexts' buffers, page-locked sets
    shutil.rmtree(d); os.makedirs(d)
    t0 = time.perf_counter()
    paths = write_animation(anim, tree, d, basename="f")
    dt = time.perf_counter() - t0
    size = sum(os.path.getsize(p) for p in paths)
    out["gpu_png"] = {"seconds": dt, "frames_per_s": N / dt, "ms_per_frame": dt / N * 1e3, "mean_file_bytes": size // N}
finally:
    shutil.rmtree(d, ignore_errors=True)

d = tempfile.mkdtemp(dir=base)
try:
    frames = bs.generate_frames(anim)
    bufs = [bs.alloc_image(tree, 1080, 1920, dtype=np.uint8) for _ in range(32)]
    t0 = time.perf_counter()
    pending = []
    with ThreadPoolExecutor(max_workers=THREADS) as pool:
        for pos in range(0, N, 16):
            chunk = frames[pos:pos + 16]
            k = (pos // 16) & 1
            for f in pending[:-16]:
                f.result()            # the set of buffers about to be reused has been encoded
            pending = pending[-16:]
            imgs = bs.render_rgb8_batch(chunk, [tree], outs=bufs[16 * k:16 * k + len(chunk)])
            pending += [pool.submit(bs.write_png, img, os.path.join(d, f"z_{pos + j:03d}.png")) for j, img in enumerate(imgs)]
        for f in pending:
            f.result()
    dt = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(d, p)) for p in os.listdir(d))
    out["host_zlib6"] = {"seconds": dt, "frames_per_s": N / dt, "ms_per_frame": dt / N * 1e3, "mean_file_bytes": size // N}
finally:
    shutil.rmtree(d, ignore_errors=True)
out["speedup"] = out["host_zlib6"]["seconds"] / out["gpu_png"]["seconds"]
print(json.dumps(out, indent=1))
