/*
 * blackstar_gpu_debug.h -- TEST HOOKS and probes.  NOT part of the product ABI (include/blackstar_gpu.h) and not in the product library:
 * these functions live in blackstar_amd/libblackstar_gpu_debug.so, which links against libblackstar_gpu.so and is loaded by this
 * repository's tests, scripts/ and bench.py's issue-rate probe only.  A host application (the Haskell shim of INTEGRATION.md) never
 * needs it.  Both libraries are built by one `make -C blackstar_amd/csrc`; call bs_debug_abi_check() first -- it refuses a product
 * library of another build (the hooks read the context's internals).  No stability promise: hooks come and go with the tests.
 */
#ifndef BLACKSTAR_GPU_DEBUG_H
#define BLACKSTAR_GPU_DEBUG_H

#include "blackstar_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* BS_OK if the two libraries in this process belong together (same BS_ABI_VERSION, same context layout), else BS_EINTERNAL. */
int bs_debug_abi_check(void);

/* Per-ray terminal state (no reference counterpart). */
typedef struct bs_ray_record {
    double vel[3], pos[3]; /* state fed to the terminating findColor call */
    double rgba[4];        /* composited colour before dropAlpha */
    int32_t steps, fate;   /* fate: 0 horizon, 1 escaped, 2 step cap */
    int32_t disk_hits, star_hits;
} bs_ray_record;

/* Trace the given traced-resolution pixels (y,x pairs) and return per-ray records (host buffers): the same device function the frame
 * kernel inlines (csrc/trace_device.h: trace_ray), in a one-lane-per-listed-ray kernel of the debug library. */
int bs_trace_rays(bs_ctx *ctx, const bs_config *cfg, const int32_t *yx, size_t n_rays, bs_ray_record *out);

/* out_sqrt[i] = sqrt(a[i]), out_div[i] = a[i] / b[i] computed on the device (host buffers); proves the f64 sqrt / divide sequences
 * STRICT mode relies on are correctly rounded.  bare = 0: hipcc's lowering of sqrt and '/'; bare = 1: the scaling-free FMA sequences
 * used inside the RK4 right-hand side; bare = 2: the raw v_rsq_f64 / v_rcp_f64 seeds. */
int bs_debug_sqrt_div(bs_ctx *ctx, const double *a, const double *b, size_t n, double *out_sqrt, double *out_div, int bare);

/* Depth (0..4, default 4) of the per-lane LDS queue of disk crossings; a ray with more crossings takes the kernel's simple re-trace
 * path, which tests force by shrinking the queue.  CHANGES THE BEHAVIOUR OF A LIVE CONTEXT (pixels stay the same; the path does not). */
int bs_debug_set_disk_slots(bs_ctx *ctx, int slots);

/* Roofline probe: times `iters` x 32 dependent-chain FP64 VALU instructions per lane (8 independent chains) on `blocks` x 256 lanes.
 * kind 0 v_fma_f64, 1 v_mul_f64, 2 v_add_f64, 3 v_rsq_f64, 4 v_rcp_f64 (8 chains: issue rate); 5/6/7 v_fma_f64 with 1/2/4 chains,
 * 8 v_rsq_f64 with 1 chain (dependent latency at 1 wave/SIMD).  out_ginstr = lane-instructions executed / 1e9. */
int bs_debug_ubench(bs_ctx *ctx, int kind, int blocks, int iters, double *out_ms, double *out_ginstr);

/* bs_encode_png's block kernel with a shader-clock stamp taken by every workgroup before its first phase and after each of its 22 phases
 * (csrc/png_block.h): clocks[b * 23 + p], b < ceil(height * (3 width + 1) / 8192).  scripts/png_phase_probe.py makes the table. */
int bs_debug_png_phases(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned long long *clocks, size_t n_clocks);

/* ---- the CU partition of bs_render_rgb8_batch / bs_render_png_batch (csrc/batch.cpp: measured once per frame shape and context) ---- */
/* The CUs the post stage owned in this context's share of the last batch call (0 = the shared chip, -1 = no batch yet). */
int bs_debug_last_post_cus(const bs_ctx *ctx);
/* This context's share of the last batch call: 0 no trial; 1 a trial ended in it (the shape is remembered now); 2 a trial progressed
 * (segments timed, or waiting for a call long enough) without ending. */
int bs_debug_last_trial(const bs_ctx *ctx);
/* What the context has measured for frames of this shape (width, height, supersampling, bloom divider or none, pixels or PNG file, the
 * arithmetic the frame gets): the remembered post-stage CU count (0, 8, 16) and, in ms[3] (may be NULL), the per-frame times of the
 * trial's segments on the shared chip / 8 / 16 CUs; -1 if that shape has not been measured on this context. */
int bs_debug_partition_choice(const bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, int png, double ms[3]);
/* Host-only: the trial's decision rule on given per-frame times ms[i] for cus[i] post-stage CUs (cus[i] == 0: shared chip; ms[i] <= 0: not
 * run): the fastest, but the shared chip unless a partition beats it by more than 1.5 %. */
int bs_debug_pick_partition(const double *ms, const int *cus, int n);
/* Forget every remembered shape of this context (the next long-enough batch measures again). */
int bs_debug_forget_partitions(bs_ctx *ctx);

/* Host-only (no device is touched): the star index bs_create builds for buildStarTree (src/StarMap.hs:90-91), a cube-map grid of star
 * directions ("Star lookup").  Writes the 6*256*256 + 2 cell offsets to cell_start (entries of cell c are
 * [cell_start[c], cell_start[c+1]); the last cell lists the stars around the origin) and, for each entry, the index of its star in
 * `stars` to entry_star (at most `cap`).  Returns the number of entries (stars + copies in neighbouring faces), or BS_EINVAL. */
long bs_debug_star_grid(const bs_star *stars, size_t n_stars, uint32_t *cell_start, int32_t *entry_star, size_t cap);

/* Host-only: the 257 thresholds of writeImg's pixel map toWord8 . sRGB (src/Raytracer.hs:23-32) that bs_srgb8 and bs_render_rgb8 compare
 * against on the device: table[k], k = 1..255, is the smallest double whose byte is >= k (found by bisection with the host libm's pow,
 * once per process); table[0] = -inf, table[256] = +inf. */
int bs_debug_srgb8_table(double table[257]);

#ifdef __cplusplus
}
#endif
#endif /* BLACKSTAR_GPU_DEBUG_H */
