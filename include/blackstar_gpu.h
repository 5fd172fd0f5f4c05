/*
 * blackstar_gpu.h -- C ABI of the MI355X-native replacement for blackstar's Raytracer.render.
 *
 * The reference has no FFI: the hot path sits behind ONE pure Haskell function,
 *     render :: Config -> StarTree -> Image U RGB Double        (src/Raytracer.hs:53, exported :4)
 * whose only caller is app/Main.hs:109.  This header is what a `foreign import ccall` shim for that
 * function binds (INTEGRATION.md shows the Haskell side).  Plain pointers and sizes only; no C++,
 * torch or HIP types appear in any signature (streams / device pointers travel as void*).
 *
 * Library: blackstar_amd/libblackstar_gpu.so (hipcc --offload-arch=gfx950).  There is NO CPU backend in
 * this library: every entry point that renders requires a HIP device and fails (never falls back).
 *
 * Conventions: caller allocates and owns every buffer; callee never frees caller memory, never calls
 * back, never throws across the ABI.  A bs_ctx belongs to one device and is driven by one OS thread at a
 * time (calls on the same context must not overlap); different contexts may be driven from different
 * OS threads.  The *_device entry points only enqueue: renders enqueued on DIFFERENT streams of one context
 * are independent of each other (each takes its own tile-queue/statistics block from a ring of 8; the ninth
 * waits for the first to finish) and bs_stats reports the one enqueued last.  Every blocking entry point returns
 * only after all work it enqueued has finished -- also on an error return, so caller buffers may be released.
 * The calling thread's current HIP device is the same after every call as before it (the library switches to the
 * context's device for the call and back: a host application that drives other devices from that thread is not disturbed).
 * Functions returning int return 0 on success and a negative BS_E* code on failure; bs_last_error() gives a
 * thread-local message.
 */
#ifndef BLACKSTAR_GPU_H
#define BLACKSTAR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6): additions only; every struct and every version-4 signature is unchanged.  bs_render_png_files runs one rolling pipeline,
 * one ring of file buffers and one writer thread PER CONTEXT, and `pipe` is now that ring's size.  New: bs_files_stats / bs_files_stats_t,
 * bs_numa_node, bs_host_page_node.  The host threads that drive a context run on the CPUs of its GPU's NUMA node.  bs_destroy's contract
 * is written down.
 * 4 (round 4): the test hooks (bs_debug_*, bs_trace_rays, struct bs_ray_record) left this header and this library: they are declared in
 * blackstar_gpu_debug.h and live in libblackstar_gpu_debug.so; bs_debug_post_cus is gone (the CU partition is measured, not modelled);
 * bs_set_max_steps refuses values above BS_MAX_STEPS_LIMIT; bs_validate_config accepts what the reference renders (negative disk
 * radii, non-finite disk / star parameters).  Every struct and every other signature is unchanged.
 * 3 (round 3): the PNG entry points (bs_png_bound, bs_encode_png[_device], bs_render_png[_batch], bs_render_png_files), BS_EIO -- additions only, every
 * struct and every version-2 signature is unchanged.
 * 2 (round 3): bs_stats_t grew `effective_mode`; bs_effective_mode added; bs_render* validate their bs_config (BS_EINVAL for
 * non-finite fields, stepSize <= 0, negative radii, lookAt within 1e-6 of position).  1: rounds 1-2.  A binding should compare
 * bs_abi_version() with the BS_ABI_VERSION it was written against before anything else. */
#define BS_ABI_VERSION 5

enum {
    BS_OK = 0,
    BS_EINVAL = -1,  /* bad argument: null pointer, non-positive resolution, buffer too small, bad hue; or a configuration on which the
                      * reference's colorize never terminates (src/Raytracer.hs:80-86 has no iteration cap): a non-finite camera
                      * position / lookAt / upVec / fov / stepSize, stepSize <= 0, lookAt within 1e-6 of position.  Returned before any
                      * GPU work.  Everything the reference does render is accepted: negative disk radii (render squares them,
                      * src/Raytracer.hs:61-62), infinite or NaN disk and star parameters (pixels come out inf / NaN like the reference's). */
    BS_EDEVICE = -2, /* no such HIP device / HIP runtime error */
    BS_ENOMEM = -3,  /* host or device allocation failed */
    BS_ECAPPED = -4, /* never returned: rays stopped by the step cap are reported through bs_stats_t.capped only */
    BS_EINTERNAL = -5,
    BS_EIO = -6      /* bs_render_png_files: a file could not be created or written (bs_last_error names it) */
};

/* Arithmetic mode of the trace kernel (DESIGN.md "Kernels").
 * STRICT: one IEEE binary64 operation per reference operation, in the reference's order, no FMA
 *         contraction, correctly rounded sqrt and divide -> trajectories bit-identical to the CPU oracle.
 * FAST:   the same discrete RK4 map evaluated in the ray's orbital plane with FMA-accumulated stage sums and
 *         r^-5 from v_rsq_f64 + a 2nd-order series correction.  Step counts, fates and disk crossings equal
 *         STRICT's on every ray tested; pixel values agree with STRICT to 3.4e-8 absolute / 3.7e-7 relative
 *         on the BASELINE frames -- inside the 1e-4 relative bar of north_star, not bit-exact.  Over the committed fuzz runs on this
 *         library (profiles/r06_fuzz_modes_100000.json, r06_fuzz_modes_clustered_100000.json: 100 000 random scenes and 2.7e9 values
 *         each, FAST against STRICT; r06_fuzz_oracle_*.json: 5 500 scenes against the CPU oracle) the worst value is 5.1e-7 relative on a
 *         uniform sky and 3.8e-6 on a CLUSTERED one: 26x inside the bar at worst.  The clustered figure is the star lookup's doing, not
 *         the integrator's: a star's weight exp(-d^2 / (2 * 0.0005^2)) (src/StarMap.hs:99-110) turns a terminal-direction difference e into
 *         up to 6000 e of relative difference.  Three guards keep it there:
 *         (1) a ray that orbits the hole (more steps than the longest straight path plus one photon-sphere circumference; a few per
 *             million; rays grazing the photon sphere amplify any rounding difference) is re-traced with STRICT arithmetic inside the kernel;
 *         (2) a frame whose stepSize exceeds 0.5 (the RK4 step no longer resolves the field next to the hole) is traced in STRICT altogether;
 *         (3) so is (round 6) a frame whose expected steps per ray N0 = (|camera.position| + sqrt safeDistance) / stepSize exceed
 *             BS_FAST_MAX_EXPECTED_STEPS: FAST's direction difference grows with the LENGTH of the path.  Measured with this rule off, over
 *             3 560 long-path scenes against the oracle (profiles/r06_fuzz_oracle_longpath.json; clustered sky): worst value 1.8e-6 below
 *             2 000 expected steps, 1.2e-5 below 10 000, 3.6e-5 below 30 000 and 2.2e-4 -- OUTSIDE the bar -- between 30 000 and 100 000
 *             (round 5's worst case, 2.3e-5, was such a scene: stepSize 0.05 from 318 radii away, 14 000 steps; FUZZ_WORST in
 *             tests/test_gpu_parity.py replays it by name).  The reference's own scene files have N0 = 233 .. 523.
 *         (2) and (3) cost STRICT's time, 2.4x FAST's per step (C3 frame: 10.7 vs 4.4 ms at stepSize 0.3).  bs_effective_mode(ctx, cfg) tells
 *         which arithmetic a frame will get and bs_stats_t.effective_mode which one the last render got. */
#define BS_FAST_MAX_EXPECTED_STEPS 2000
enum { BS_MODE_STRICT = 0, BS_MODE_FAST = 1 };

/* Replaces the `Config` argument of render (src/ConfigFile.hs:16-38), AS PARSED: radii un-squared,
 * safeDistance absent (render derives it, src/Raytracer.hs:59-60), user resolution (render doubles it
 * when supersampling, :58).  disk_hsi[0] is hue/360 as produced by src/ConfigFile.hs:51. */
typedef struct bs_config {
    double cam_pos[3], cam_lookat[3], cam_up[3], fov; /* Camera, src/ConfigFile.hs:34-38 */
    double step_size, star_intensity, star_saturation; /* Scene, :21,:24,:25 */
    double disk_hsi[3];                                /* Scene.diskColor :26 */
    double disk_opacity, disk_inner, disk_outer;       /* :27-29 */
    int32_t width, height, supersampling;              /* :30-31 */
    int32_t _pad;
} bs_config;

/* Replaces one element of the `StarTree` argument: (V3 position, (mag*100, hue, sat)), i.e. one
 * `KdMap.assocs` entry after starColor' (src/StarMap.hs:25-26,61-62).  Position is a unit vector. */
typedef struct bs_star {
    double x, y, z, hue, sat;
    int32_t mag;
    int32_t _pad;
} bs_star;

/* Counters of the most recent render on a context (no reference counterpart; feeds the roofline). */
typedef struct bs_stats_t {
    uint64_t rays;      /* traced rays = w' * h' */
    uint64_t steps;     /* sum over rays of colorize' iterations (src/Raytracer.hs:80-86) */
    uint64_t capped;    /* rays stopped by the step cap (the reference would not terminate) */
    uint64_t horizon;   /* rays ended by r^2 < 1 (:93) */
    uint64_t escaped;   /* rays ended by r^2 > safeDistance (:94) */
    uint64_t disk_hits; /* Layer blends (:96-98) */
    uint64_t star_hits; /* stars summed by starLookup (src/StarMap.hs:104) */
    uint64_t wave_iters; /* sum over wavefronts of the iterations of their slowest lane: lane efficiency = steps / (64 * wave_iters) */
    double kernel_ms;   /* hipEvent time of the kernels of the last render */
    double wall_ms;     /* host wall time of the last bs_render call (H2D params + kernels + D2H image) */
    int32_t effective_mode; /* BS_MODE_* the last render was actually traced with (FAST frames with stepSize > 0.5 run in STRICT) */
    int32_t zero_copy;      /* 1: the last blocking bs_render / bs_render_rows / bs_render_rgb8 wrote the caller's page-locked buffer
                             * from the kernel itself; 0: it went through a device image and a copy (pageable memory, a buffer that is
                             * not provably inside ONE page-locked range, BLACKSTAR_ZERO_COPY=0) */
} bs_stats_t;

typedef struct bs_ctx bs_ctx;

/* Replaces: readTreeFromFile's result being handed to doStart once (app/Main.hs:46-49) -- the star set is
 * uploaded once and reused for every frame.  Copies `stars`, builds the star direction grid, uploads.
 * n_stars == 0 is the "no starmap" case: inRadius yields [] and escaped rays are black (src/StarMap.hs:104,115).
 * device: HIP device ordinal (>= 0).  Returns NULL on error. */
bs_ctx *bs_create(int device, const bs_star *stars, size_t n_stars);
/* Waits for everything THIS context has enqueued -- on its own streams and, through an event the library records behind every *_device
 * call (error returns included), on the caller's streams -- then frees its memory.  The library itself waits for nothing else; the HIP
 * runtime's hipFree does, though: on ROCm 7.2 it returns only when the DEVICE is idle, so a bs_destroy stalls behind another stream's
 * running kernel (measured, tests/test_gpu_parity.py: 18.9 ms behind a 19 ms frame of another context; bs_create 1.5 ms, not stalled).
 * Correctness never relies on that.  Buffers from bs_host_alloc are the caller's and survive the context.  No call on the context may be
 * in progress on another thread. */
void bs_destroy(bs_ctx *ctx);
/* Number of HIP devices this process can see (one bs_ctx per device for bs_render_batch / bs_render_split), or a negative
 * BS_E* code.  No reference counterpart (the reference has one backend: the host's cores, blackstar.cabal:47). */
int bs_device_count(void);

/* Replaces: render cfg tree (src/Raytracer.hs:53-67) at its only call site app/Main.hs:109.
 * Blocking.  Fills out_rgb[height*width*3], interleaved RGB f64, row-major (y down), linear light,
 * unclamped, already supersample-reduced -- the `Image S RGB Double` layout the Haskell shim wraps.
 * If out_rgb is page-locked memory (bs_host_alloc, or the caller's own hipHostMalloc / hipHostRegister) the kernel writes it
 * directly over PCIe -- no device image and no copy: the call costs the kernel time (C3 frame: 4.6 ms).  Into pageable memory the
 * frame is traced as two consecutive launches of half the rows each, so that the first half's way to the host overlaps the
 * second half's kernel (about 6 ms into a buffer that has been touched before, 9 ms into a fresh one: first-touch page faults).
 * Pageable caller memory is never handed to the HIP runtime in pieces above 1 MiB: it travels through the context's own page-locked
 * staging pieces and a host memcpy -- the runtime would pin the caller's pages on the fly and DMA at the caller's address, a path
 * on which a GPU memory fault (which ends the process) was observed under heavy allocate / free traffic of the host application.
 * Pixels and bs_stats are those of the whole frame either way (bs_stats_t.zero_copy says which way it went).
 * A buffer that STARTS in page-locked memory must be contained in that one page-locked range: one that runs past its end (into
 * pageable memory, or across a gap between two hipHostRegister ranges) is refused with BS_EINVAL -- the kernel's stores would fault
 * there, and the runtime's own copy refuses such a destination too.  The same holds for every host-output entry point. */
int bs_render(bs_ctx *ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles);

/* Same, but the image stays in HBM: d_out_rgb is a device pointer on the context's device, the work is
 * enqueued on hip_stream (a hipStream_t cast to void*, NULL = default stream) and the call returns without
 * synchronising.  Used by bench.py (inputs/outputs resident) and by on-device post-processing. */
int bs_render_device(bs_ctx *ctx, const bs_config *cfg, void *d_out_rgb, size_t out_doubles, void *hip_stream);

/* Page-locked host memory for images ("pinned buffers, no per-frame hipMalloc").  Any host pointer works
 * as an output buffer, but a FRESH pageable buffer costs the operating system's first-touch page faults on top of the copy
 * (measured for a 1080p f64 frame: 9.0 ms per bs_render into newly allocated memory, about 6 ms into a pageable buffer that is
 * reused, 4.5 ms into one from here; kernel 4.3 ms).  Memory from bs_host_alloc never faults and is written by the trace kernel
 * itself (zero copy) in bs_render / bs_render_rows / bs_render_split / bs_render_batch.
 * It belongs to the caller until bs_host_free (it may outlive the context; Haskell: newForeignPtr with bs_host_free as
 * finalizer).  Returns NULL on failure. */
void *bs_host_alloc(bs_ctx *ctx, size_t bytes);
void bs_host_free(void *p);

/* One frame split into horizontal bands (a single huge frame sharded by rows over several GPUs, no halo:
 * rays are independent and the 2x2 supersample never straddles an output row).  Renders output rows [row0, row1) of the
 * frame `cfg` describes into a buffer of (row1-row0)*width*3 doubles; the bands of a frame concatenated are bit-identical
 * to bs_render of the whole frame.  bs_render(ctx, cfg, ...) == bs_render_rows(ctx, cfg, 0, cfg->height, ...). */
int bs_render_rows(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, double *out_rgb, size_t out_doubles);
int bs_render_rows_device(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, void *d_out_rgb, size_t out_doubles, void *hip_stream);
/* The same from one process that holds several contexts (one per GPU): context c renders the c-th of n_ctx contiguous row
 * bands of the frame (one host thread per context) straight into its place in out_rgb (height*width*3 doubles).  The
 * result is bit-identical to bs_render on any one of the contexts. */
int bs_render_split(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles);

/* Batch mode (app/Main.hs:68-77 renders a directory of scenes sequentially with the same tree):
 * frame i is rendered by ctxs[i % n_ctx] (one context per device, frames sharded round-robin, one host thread per
 * context); outs[i] is a host buffer of cfgs[i].height*width*3 doubles.  Per context two frames are in flight (two device
 * images, two compute streams, one copy stream): frame k's device-to-host copy and its end-of-frame tail overlap frame
 * k+1's kernel.  bs_stats is not updated by batch frames.  The contexts must be distinct (BS_EINVAL otherwise: each is driven by
 * its own host thread); the same holds for bs_render_split and bs_render_rgb8_batch. */
int bs_render_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, double *const *outs);

/* ---- "next" rows: the two steps after render in app/Main.hs:113-123, kept on the device ---- */

/* Replaces: bloom strength divider img (src/ImageFilters.hs:80-86): out = img + strength * boxBlur(w `div` divider, 3 passes).
 * d_in / d_out: device pointers to height*width*3 interleaved RGB f64 (may alias); enqueued on hip_stream, no sync.
 * The intermediate sweeps live in ONE pair of scratch images per context: a call on another stream than the previous
 * user of that scratch is ordered behind it with an event (correct, but such calls do not overlap).
 * Returns BS_EINVAL when width `div` divider == 0 (the reference crashes there: foldl1' on an empty window). */
int bs_bloom_device(bs_ctx *ctx, const void *d_in, void *d_out, int width, int height, double strength, int divider, void *hip_stream);
int bs_bloom(bs_ctx *ctx, const double *in, double *out, int width, int height, double strength, int divider); /* host buffers, blocking */

/* Replaces: supersample (src/ImageFilters.hs:88-97) as a standalone call (render itself fuses it): in is height2 x width2
 * RGB f64, out is (height2 div 2) x (width2 div 2).  Host buffers, blocking. */
int bs_supersample(bs_ctx *ctx, const double *in, double *out, int width2, int height2);

/* Replaces: A.map (toWord8 . fmap sRGB) in writeImg (src/Raytracer.hs:23-32): n_values f64 channel values ->
 * n_values bytes (sRGB transfer, clamp to [0,1], *255, round half to even). */
int bs_srgb8_device(bs_ctx *ctx, const void *d_in, void *d_out_u8, size_t n_values, void *hip_stream);
int bs_srgb8(bs_ctx *ctx, const double *in, unsigned char *out, size_t n_values); /* host buffers, blocking */

/* Replaces: the body of doRender (app/Main.hs:105-123) up to the PNG encoder -- render, bloom when bloom_strength != 0,
 * then writeImg's pixel map -- with the f64 image never leaving HBM: only height*width*3 BYTES come back (6.2 MB
 * instead of 49.8 MB at 1080p).  Blocking.  out_rgb8: interleaved RGB8, row-major. */
int bs_render_rgb8(bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, unsigned char *out_rgb8, size_t out_bytes);

/* Batch mode with the whole of doRender on the device (app/Main.hs:68-77 calls doRender, :105-123, for every scene of a directory:
 * render, bloom with the scene's own bloomStrength / bloomDivider, writeImg): frame i is rendered by ctxs[i % n_ctx] (one host
 * thread per context) and leaves the GPU as height*width*3 bytes of RGB8 in outs[i].  bloom_strengths[i] == 0 (or a NULL array)
 * skips the bloom of that frame like the reference does; bloom_dividers is only read where the strength is not 0.  Per context two
 * frames are in flight on two streams: the next frame's trace kernel takes over the SIMDs this frame's last tiles leave, and this
 * frame's bloom runs on the first CUs the next trace kernel frees.  Where it MEASURES faster the chip is PARTITIONED instead: the trace
 * kernels run on streams whose CU mask leaves 8 or 16 CUs out, bloom + sRGB8 of the previous frames run on a stream that owns exactly
 * those, three frames in flight (the default-aa frame: 4.28 instead of 4.67 ms; 3840x2160: 17.4 instead of 17.8).  The decision is a
 * measurement, taken once per frame shape (size, supersampling, bloom divider, arithmetic) and context: a share of frames of one shape
 * the context has not measured yet is rendered in segments of 8 frames -- shared chip, 16, 8 post-stage CUs; 8 more on the shared chip
 * first if the context has been idle -- each segment's steady state is timed, and once all three are (32 frames in one call, or e.g. three
 * calls of 16) the context remembers the fastest (a partition only if it wins by more than 1.5 %).  Shares of fewer than 8 frames or of
 * mixed shapes use what has been remembered, else the shared chip.  Only when every outs[i] is page-locked.  Environment BLACKSTAR_POST_CUS = 0 (never) | 8 |
 * 16 | 24 | 32 (always that many) overrides.  The CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags):
 * a host application that keeps the NULL stream of the device busy from another thread during this call serialises the pipeline
 * against its own work -- correct, but the overlap is lost; BLACKSTAR_POST_CUS=0 avoids those streams altogether.  Page-locked outs[i]
 * (bs_host_alloc) are written by the last kernel itself.  PAGEABLE outs[i] are the slow form, here and in bs_render_batch: a frame is
 * delivered through the context's page-locked staging and a host memcpy (never pinned on the fly, see bs_render) BEFORE the frame after
 * next is enqueued, so during each delivery only one frame is on the GPU, not two -- use bs_host_alloc buffers for the documented overlap.
 * Blocking; byte-identical to bs_render_rgb8 frame by frame whichever way a frame was made; bs_stats is not updated. */
int bs_render_rgb8_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths,
                         const int *bloom_dividers, unsigned char *const *outs);

/* ---- writeImg's FILE on the device (src/Raytracer.hs:30-32: `writeImage path`; app/Main.hs:119-123) ----
 * The reference hands the RGB8 image to massiv-io's writeImage, i.e. JuicyPixels' PNG encoder over zlib: 0.1-0.25 s of one host core per
 * 1080p frame, 25-60x what this library needs to render it.  These entry points produce the finished PNG FILE on the device (filter
 * choice, deflate with per-block dynamic Huffman codes, CRCs, Adler-32: blackstar_amd/csrc/png_block.h), so a frame crosses PCIe as
 * about 1 MB of file and the caller only has to write(2) it.  What the format guarantees is the DECODED image: every file decodes
 * (zlib, libpng, Pillow) to exactly the RGB8 frame bs_render_rgb8 returns; its bytes differ from JuicyPixels' like any two zlib versions'.
 * Files are about 1.5 % larger than libpng's at zlib level 1 and 12 % larger than at level 6 on rendered frames (distance-1 matches
 * only), never larger than bs_png_bound. */

/* Bytes a width x height RGB8 frame needs at most as a PNG file of this encoder (about 0.2 % above the pixels): the capacity every
 * out_png / d_png buffer below must have.  BS_EINVAL for non-positive sizes and frames above 1e9 pixels / files of 4 GiB. */
int bs_png_bound(int width, int height, size_t *out_bytes);
/* Enqueue-only: d_rgb8 (height*width*3 bytes, device) -> the file in d_png (cap >= bs_png_bound; device memory or a page-locked host
 * buffer's device alias), its size in *d_file_bytes (one 8-byte-aligned uint64, same choice), both valid once the stream has passed.
 * (The encoder's scratch for this entry point and for bs_encode_png / bs_render_png is the context's own, separate from the batch
 * pipelines': a batch call may follow at once.  Two calls on different streams are ordered by an event, like bs_bloom_device.) */
int bs_encode_png_device(bs_ctx *ctx, const void *d_rgb8, int width, int height, void *d_png, size_t cap, void *d_file_bytes, void *hip_stream);
/* Host buffers, blocking: the parity hook of the encoder alone (rgb8 -> file).  A page-locked out_png is written by the GPU itself. */
int bs_encode_png(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned char *out_png, size_t cap, size_t *out_bytes);
/* Replaces: the whole of doRender (app/Main.hs:105-123) except the write(2): render, bloom when bloom_strength != 0, writeImg's pixel
 * map and file format.  Blocking; *out_bytes bytes of out_png are the file.  Decodes to bs_render_rgb8's bytes. */
int bs_render_png(bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, unsigned char *out_png, size_t cap, size_t *out_bytes);
/* bs_render_rgb8_batch with files instead of pixels: outs[i] (capacity caps[i] >= bs_png_bound of frame i) receives frame i's PNG file,
 * out_bytes[i] its size.  Two frames in flight per context, the encoder of frame k running under the trace kernel of frame k+1;
 * page-locked outs[i] are written by the encoder itself (only the file's bytes cross PCIe).  The chip is partitioned like in
 * bs_render_rgb8_batch -- measured separately for files, whose post stage also runs the encoder's kernels (the default-aa frame: 16 CUs,
 * 4.44 ms per frame against 4.77 on the shared chip and 4.31 for bs_render_rgb8_batch).  Blocking; bs_stats is not updated. */
int bs_render_png_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths,
                        const int *bloom_dividers, unsigned char *const *outs, const size_t *caps, size_t *out_bytes);

/* Replaces: the reference's batch loop to the very end (app/Main.hs:68-77 mapping doRender, :105-123, over the scenes, including writeImg's
 * write, src/Raytracer.hs:29-32): frame i is rendered, bloomed, encoded on ctxs[i % n_ctx] and WRITTEN to paths[i] (created or truncated, like
 * --force).  Per context: ONE rolling pipeline over its whole share of the frames (the same one bs_render_png_batch runs, CU partition and
 * all), a ring of `pipe` page-locked file buffers of the library's own (<= 0: 16; at least 4) and ONE writer thread that creates / writes /
 * closes the files as the frames leave the device.  A context never waits for another context's frames or files; its pipeline waits for its
 * own writer only when every buffer of the ring is still to be written (bs_files_stats_t.buffer_wait_ms).  The context's host thread, its
 * writer and -- through bs_host_alloc -- its buffers sit on the NUMA node of its GPU.  Blocking.  BS_EIO if a file cannot be created or
 * written (bs_last_error names it): every context stops taking new frames, what is in flight is drained, every writer is joined, the
 * contexts stay usable; files already written stay.  Files decode to bs_render_rgb8's pixels.  C3: within 1 % of bs_render_png_batch. */
int bs_render_png_files(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths,
                        const int *bloom_dividers, const char *const *paths, int pipe);

/* What the host side of ONE context's share of the last bs_render_png_files call did (no reference counterpart: writeImg's write is
 * sequential there, src/Raytracer.hs:29-32).  writer_busy_ms / wall_ms is the fraction of the call the context's writer spent in
 * fopen / fwrite / fclose: far below 1 = the GPU sets the pace; buffer_wait_ms > 0 = the writer did. */
typedef struct bs_files_stats_t {
    uint64_t files;             /* files this context's writer wrote */
    uint64_t bytes;             /* ... and their bytes */
    double wall_ms;             /* this context's share: first buffer prepared ... last file closed */
    double writer_busy_ms;      /* of that, the writer inside fopen / fwrite / fclose */
    double buffer_wait_ms;      /* the render pipeline waiting for a free file buffer */
    int32_t ring;               /* file buffers of the context in that call */
    int32_t writer_threads;     /* 1; 0: no thread could be started and the pipeline's own thread wrote the files */
    int32_t numa_node_gpu;      /* bs_numa_node(ctx) */
    int32_t numa_node_buffers;  /* the node the ring's pages live on (bs_host_page_node); -1 unknown, -2 not all on one node */
    int32_t threads_bound;      /* 1: the writer and the pipeline's thread ran on the CPUs of numa_node_gpu (bound by the library, or the process is confined to them) */
    int32_t _pad;
} bs_files_stats_t;
int bs_files_stats(const bs_ctx *ctx, bs_files_stats_t *out);

/* Where the context's GPU sits in the host: its NUMA node as /sys/bus/pci/devices/<bdf>/numa_node gives it, -1 when the host does not
 * say.  The threads the library starts for a context (one per context in every *_batch / split / files call, the file writers) and the
 * calling thread while it drives a context inside those calls run on that node's CPUs (BLACKSTAR_NUMA_BIND=0: left alone); the caller's
 * thread gets its own affinity back before the call returns. */
int bs_numa_node(const bs_ctx *ctx);
/* The NUMA node the page of host memory at p lives on right now (move_pages(2) as a query), -1 unknown.  bs_host_alloc memory lands on
 * bs_numa_node(ctx) on the hosts measured (profiles/r06_host_topology.txt); bench.py reports it for every delivered form. */
int bs_host_page_node(const void *p);

/* Replaces: starLookup starmap intensity saturation vel (src/StarMap.hs:93-115), batched: dirs holds n
 * un-normalised direction vectors (x,y,z interleaved, host); out_rgb gets n RGB triples, out_hits (may be
 * NULL) the number of stars within the radius.  Runs the same device function the trace kernel calls. */
int bs_star_lookup(bs_ctx *ctx, double intensity, double saturation, const double *dirs, size_t n, double *out_rgb, int32_t *out_hits);

/* Environment read ONCE, at bs_create (A/B switches for measurements; a host application sets none of them): BLACKSTAR_MODE=strict|fast
 * (initial bs_set_mode), BLACKSTAR_POST_CUS (see bs_render_rgb8_batch), BLACKSTAR_ZERO_COPY=0, BLACKSTAR_FAST_GUARD=0, BLACKSTAR_HOST_BANDS,
 * BLACKSTAR_STAGGER, BLACKSTAR_STAGGER_MIN_TILES, BLACKSTAR_STATIC_FIRST_BELOW, BLACKSTAR_LATE_POP_SLOT, BLACKSTAR_BLOCKS_PER_CU, BLACKSTAR_POST_PLAN_CUS, BLACKSTAR_BLOOM_PLAN_CUS, BLACKSTAR_NUMA_BIND=0, BLACKSTAR_FAST_MAX_STEPS. */
int bs_set_mode(bs_ctx *ctx, int mode);           /* BS_MODE_*; default BS_MODE_FAST */
int bs_get_mode(const bs_ctx *ctx);
/* The arithmetic a render of `cfg` on this context would be traced with: bs_get_mode(), except that a FAST context traces frames
 * with stepSize > 0.5, or with more than BS_FAST_MAX_EXPECTED_STEPS expected steps per ray, in STRICT (see BS_MODE_FAST above).  Returns BS_MODE_* or BS_EINVAL.  For batch frames (whose statistics
 * are not kept) this is the only way to know; perf numbers and A/B comparisons should record it. */
int bs_effective_mode(const bs_ctx *ctx, const bs_config *cfg);
/* Safety cap on colorize' iterations per ray; the reference has none (src/Raytracer.hs:80-86).  Default 100000.  1 <= max_steps <=
 * BS_MAX_STEPS_LIMIT, BS_EINVAL otherwise: a ray's steps are an int in the kernel and bs_stats_t.steps a 64-bit sum (per lane, per
 * wavefront and per frame), so the largest frame (2^30 traced rays) of capped rays still counts exactly: 2^30 x 2^30 = 2^60. */
#define BS_MAX_STEPS_LIMIT (1 << 30)
int bs_set_max_steps(bs_ctx *ctx, int max_steps);
int bs_stats(bs_ctx *ctx, bs_stats_t *out);       /* synchronises the context's last render first */
const char *bs_last_error(void);
int bs_abi_version(void);

/* Replaces: readMap + starColor' (src/StarMap.hs:45-75): parse a PPM catalogue image held in memory
 * (28-byte header, 28-byte big-endian records) into bs_star.  Returns the number of stars the buffer
 * holds; writes at most `cap` of them.  Host-only; returns BS_EINVAL if nbytes < 28. */
long bs_read_ppm(const void *bytes, size_t nbytes, bs_star *out, size_t cap);

/* Host-only: the checks every bs_render* entry point applies to its bs_config before any GPU work (see BS_EINVAL): BS_OK, or
 * BS_EINVAL with the reason in bs_last_error().  No reference counterpart: `render` is total in the types and simply never
 * returns on such inputs (src/Raytracer.hs:80-86).  Only inputs on which it would not return are refused. */
int bs_validate_config(const bs_config *cfg);

/* Replaces: toPixelRGB on PixelHSI (massiv-io Graphics.ColorSpace; call sites src/Raytracer.hs:65,
 * src/StarMap.hs:114).  Host-only; returns BS_EINVAL if the hue is outside [0,1). */
int bs_hsi_to_rgb(double hue, double sat, double intensity, double rgb[3]);

#ifdef __cplusplus
}
#endif
#endif /* BLACKSTAR_GPU_H */
