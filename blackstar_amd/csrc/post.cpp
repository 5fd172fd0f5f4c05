// post.cpp -- the steps after render in app/Main.hs:113-123, on the device, one frame at a time: bloom (ImageFilters.hs:80-86), supersample
// (:88-97), writeImg's pixel map (Raytracer.hs:23-32) and its PNG file (png_kernels.hip), and the single-frame pipelines bs_render_rgb8 /
// bs_render_png.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <string>

#include "bs_context.h"

using namespace bs;

// (the bs_* entry points get C linkage from their declarations in include/blackstar_gpu.h)

int bs::ensure_post(bs_ctx *ctx, size_t n)
{
    if (ctx->post_cap >= n) return BS_OK;
    // growing = freeing: a blur an earlier *_device call enqueued on a caller's stream may still be using the old pair (nothing here relies
    // on hipFree waiting for the device)
    if (ctx->post_busy && ctx->ev_post) HIP_TRY(hipEventSynchronize(ctx->ev_post));
    for (double *&b : ctx->d_post) {
        if (b) (void)hipFree(b);
        b = nullptr;
    }
    ctx->post_cap = 0;
    for (double *&b : ctx->d_post)
        if (hipMalloc((void **)&b, n * sizeof(double)) != hipSuccess) return fail(BS_ENOMEM, "hipMalloc bloom scratch failed");
    ctx->post_cap = n;
    return BS_OK;
}

// The blur scratch d_post[0..1] is one pair per context.  *_device calls only enqueue, so two of them on different streams
// would otherwise share it unordered: a user on another stream than the previous one first waits for that one's event.
static int acquire_post(bs_ctx *ctx, hipStream_t s)
{
    if (!ctx->ev_post) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_post, hipEventDisableTiming));
    if (ctx->post_busy && ctx->post_stream != s) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_post, 0));
    return BS_OK;
}

static int release_post(bs_ctx *ctx, hipStream_t s)
{
    HIP_TRY(hipEventRecord(ctx->ev_post, s));
    ctx->post_busy = true;
    ctx->post_stream = s;
    return BS_OK;
}

int bs_bloom_device(bs_ctx *ctx, const void *d_in, void *d_out, int width, int height, double strength, int divider, void *hip_stream)
try {
    if (!ctx || !d_in || !d_out || width <= 0 || height <= 0) return fail(BS_EINVAL, "bad argument");
    if (divider <= 0 || width / divider == 0)  // the reference crashes here: foldl1' over an empty window (ImageFilters.hs:59)
        return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
    BS_ON_DEVICE(ctx);
    ForeignWork seen_by_destroy(ctx, hip_stream);
    size_t n = (size_t)width * height * 3;
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    rc = acquire_post(ctx, static_cast<hipStream_t>(hip_stream));
    if (rc) return rc;
    if (bs::launch_bloom((const double *)d_in, (double *)d_out, ctx->d_post[0], ctx->d_post[1], width, height, strength, divider,
                         ctx->bloom_plan_cus > 0 ? ctx->bloom_plan_cus : ctx->n_cu, hip_stream))
        return fail(BS_EDEVICE, "bloom launch failed");
    return release_post(ctx, static_cast<hipStream_t>(hip_stream));
} catch (...) { return bs::abi_exception("bs_bloom_device"); }

int bs_bloom(bs_ctx *ctx, const double *in, double *out, int width, int height, double strength, int divider)
try {
    if (!ctx || !in || !out || width <= 0 || height <= 0) return fail(BS_EINVAL, "bad argument");
    BS_ON_DEVICE(ctx);
    size_t n = (size_t)width * height * 3;
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    StreamDrain drain(ctx);
    rc = copy_in(ctx, ctx->d_post[2], in, n * sizeof(double), ctx->stream);
    if (rc) return rc;
    rc = bs_bloom_device(ctx, ctx->d_post[2], ctx->d_post[2], width, height, strength, divider, ctx->stream);
    if (rc) return rc;
    rc = copy_out(ctx, out, ctx->d_post[2], n * sizeof(double), ctx->stream);
    if (rc) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_bloom"); }

int bs_supersample(bs_ctx *ctx, const double *in, double *out, int width2, int height2)
try {
    if (!ctx || !in || !out || width2 < 0 || height2 < 0) return fail(BS_EINVAL, "bad argument");
    const size_t n_in = (size_t)width2 * height2 * 3, n_out = (size_t)(width2 / 2) * (height2 / 2) * 3;
    if (n_out == 0) return BS_OK;
    BS_ON_DEVICE(ctx);
    int rc = ensure_post(ctx, n_in);
    if (rc) return rc;
    StreamDrain drain(ctx);
    rc = copy_in(ctx, ctx->d_post[2], in, n_in * sizeof(double), ctx->stream);
    if (rc) return rc;
    rc = acquire_post(ctx, ctx->stream);
    if (rc) return rc;
    if (bs::launch_supersample(ctx->d_post[2], ctx->d_post[0], width2, height2, ctx->stream)) return fail(BS_EDEVICE, "supersample launch failed");
    rc = release_post(ctx, ctx->stream);
    if (rc) return rc;
    rc = copy_out(ctx, out, ctx->d_post[0], n_out * sizeof(double), ctx->stream);
    if (rc) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_supersample"); }

int bs_srgb8_device(bs_ctx *ctx, const void *d_in, void *d_out_u8, size_t n_values, void *hip_stream)
try {
    if (!ctx || (n_values && (!d_in || !d_out_u8))) return fail(BS_EINVAL, "bad argument");
    BS_ON_DEVICE(ctx);
    ForeignWork seen_by_destroy(ctx, hip_stream);   // (the kernel reads the context's sRGB8 table)
    if (bs::launch_srgb8((const double *)d_in, (unsigned char *)d_out_u8, n_values, ctx->d_srgb_table, hip_stream)) return fail(BS_EDEVICE, "srgb8 launch failed");
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_srgb8_device"); }

int bs_srgb8(bs_ctx *ctx, const double *in, unsigned char *out, size_t n_values)
try {
    if (!ctx || (n_values && (!in || !out))) return fail(BS_EINVAL, "bad argument");
    if (n_values == 0) return BS_OK;
    BS_ON_DEVICE(ctx);
    int rc = ensure_post(ctx, n_values);
    if (rc) return rc;
    if (ctx->u8_cap < n_values) {
        if (ctx->d_u8) (void)hipFree(ctx->d_u8);
        ctx->d_u8 = nullptr;
        ctx->u8_cap = 0;
        if (hipMalloc((void **)&ctx->d_u8, n_values) != hipSuccess) return fail(BS_ENOMEM, "hipMalloc failed");
        ctx->u8_cap = n_values;
    }
    StreamDrain drain(ctx);
    rc = copy_in(ctx, ctx->d_post[2], in, n_values * sizeof(double), ctx->stream);
    if (rc) return rc;
    rc = bs_srgb8_device(ctx, ctx->d_post[2], ctx->d_u8, n_values, ctx->stream);
    if (rc) return rc;
    rc = copy_out(ctx, out, ctx->d_u8, n_values, ctx->stream);
    if (rc) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_srgb8"); }

int bs::check_bloom_args(int width, double strength, int divider)
{
    if (strength != 0 && (divider <= 0 || width / divider == 0))  // the reference crashes here: foldl1' over an empty window (ImageFilters.hs:59)
        return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
    return BS_OK;
}

// What doRender does with the rendered image (app/Main.hs:113-123): bloom when bloomStrength /= 0, then writeImg's pixel map -- d_img
// (f64, w x h x 3) -> d_u8 (RGB8), on stream s.  The final img + strength * blurred is fused with the sRGB8 map: the bloomed f64
// image is never written.  plan_cus: the CUs the blur sweeps are planned for.
int bs::enqueue_post_rgb8(bs_ctx *ctx, const double *d_img, int w, int h, double strength, int divider, unsigned char *d_u8, int plan_cus, hipStream_t s)
{
    if (strength != 0) {
        int rc = acquire_post(ctx, s);
        if (rc) return rc;
        if (bs::launch_bloom_srgb8(d_img, d_u8, ctx->d_post[0], ctx->d_post[1], w, h, strength, divider, plan_cus, ctx->d_srgb_table, s))
            return fail(BS_EDEVICE, "bloom launch failed");
        return release_post(ctx, s);
    }
    if (bs::launch_srgb8(d_img, d_u8, (size_t)w * h * 3, ctx->d_srgb_table, s)) return fail(BS_EDEVICE, "srgb8 launch failed");
    return BS_OK;
}

int bs_render_rgb8(bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, unsigned char *out_rgb8, size_t out_bytes)
try {
    if (!ctx || !cfg || !out_rgb8) return fail(BS_EINVAL, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = (size_t)cfg->width * cfg->height * 3;
    if (out_bytes < n) return fail(BS_EINVAL, "output buffer too small");
    if (int rc = check_bloom_args(cfg->width, bloom_strength, bloom_divider)) return rc;
    BS_ON_DEVICE(ctx);
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    // a page-locked out_rgb8 is written by the sRGB8 kernel itself (zero copy), otherwise staged through d_u8
    unsigned char *u8_target = ctx->d_u8;
    bool straddles = false;
    if (double *alias = device_alias_of_pinned(ctx, out_rgb8, n, &straddles)) u8_target = reinterpret_cast<unsigned char *>(alias);
    if (straddles) return fail(BS_EINVAL, kStraddleMsg);
    StreamDrain drain(ctx);
    // doRender (app/Main.hs:105-123): render -> bloom if bloomStrength /= 0 -> writeImg's sRGB + toWord8, all in HBM
    rc = enqueue_render(ctx, cfg, ctx->d_post[2], n, ctx->stream);
    if (rc) return rc;
    rc = enqueue_post_rgb8(ctx, ctx->d_post[2], cfg->width, cfg->height, bloom_strength, bloom_divider, u8_target, ctx->n_cu, ctx->stream);
    if (rc) return rc;
    if (u8_target == ctx->d_u8) {
        rc = copy_out(ctx, out_rgb8, ctx->d_u8, n, ctx->stream);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_zero_copy = u8_target != ctx->d_u8;
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_rgb8"); }

// ---- writeImg's file on the device (png_kernels.hip) --------------------------------------------------------------------------------

int bs::check_png_frame(int width, int height)
{
    if (width <= 0 || height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    if ((double)width * (double)height > 1.0e9 || bs::png_file_bound(width, height) > 0xFFFFFFFFull)
        return fail(BS_EINVAL, "frame too large for one PNG file of this encoder (chunk offsets are 32 bits)");
    return BS_OK;
}

int bs_png_bound(int width, int height, size_t *out_bytes)
try {
    if (!out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    *out_bytes = (size_t)bs::png_file_bound(width, height);
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_png_bound"); }

// PNG slot k of the context sized for a w x h frame: the encoder's scratch, the page-locked size slots, and (device_file) a device copy
// of the file for a caller whose buffer the GPU cannot write.
int bs::ensure_png(bs_ctx *ctx, int k, int w, int h, bool device_file)
{
    // (slot kPngSingle may still be in use by an encoder an earlier bs_encode_png_device enqueued on a caller's stream: wait before it can be
    // freed to grow; slots 0..2 belong to the batch pipelines, which drain their streams before they return)
    if (k == bs_ctx::kPngSingle && ctx->png_busy && ctx->ev_png &&
        (ctx->png_scratch_cap[k] < bs::png_scratch_bytes(w, h) || (device_file && ctx->png_file_cap[k] < (size_t)bs::png_file_bound(w, h))))
        HIP_TRY(hipEventSynchronize(ctx->ev_png));
    if (!grow_device(ctx->d_png_scratch[k], ctx->png_scratch_cap[k], bs::png_scratch_bytes(w, h)))
        return fail(BS_ENOMEM, "hipMalloc PNG scratch failed");
    if (device_file && !grow_device(ctx->d_png_file[k], ctx->png_file_cap[k], (size_t)bs::png_file_bound(w, h)))
        return fail(BS_ENOMEM, "hipMalloc PNG file failed");
    if (!ctx->h_png_bytes) HIP_TRY(hipHostMalloc((void **)&ctx->h_png_bytes, bs_ctx::kPngSlots * sizeof(uint64_t), hipHostMallocDefault));
    return BS_OK;
}

uint64_t *bs::png_bytes_slot(bs_ctx *ctx, int k)
{
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, ctx->h_png_bytes, 0) != hipSuccess || !d) return nullptr;
    return static_cast<uint64_t *>(d) + k;
}

// PNG slot kPngSingle serves the enqueue-only and single-frame entry points (the batch pipelines have slots 0..2 to themselves, so a batch
// call issued right behind a bs_encode_png_device on a caller's stream cannot touch its scratch): like the blur scratch, a user on
// another stream first waits for the previous one.
static int acquire_png(bs_ctx *ctx, hipStream_t s)
{
    if (!ctx->ev_png) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_png, hipEventDisableTiming));
    if (ctx->png_busy && ctx->png_stream != s) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_png, 0));
    return BS_OK;
}

static int release_png(bs_ctx *ctx, hipStream_t s)
{
    HIP_TRY(hipEventRecord(ctx->ev_png, s));
    ctx->png_busy = true;
    ctx->png_stream = s;
    return BS_OK;
}

int bs_encode_png_device(bs_ctx *ctx, const void *d_rgb8, int width, int height, void *d_png, size_t cap, void *d_file_bytes, void *hip_stream)
try {
    if (!ctx || !d_rgb8 || !d_png || !d_file_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    if (cap < bs::png_file_bound(width, height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    BS_ON_DEVICE(ctx);
    ForeignWork seen_by_destroy(ctx, hip_stream);
    int rc = ensure_png(ctx, bs_ctx::kPngSingle, width, height, false);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    rc = acquire_png(ctx, s);
    if (rc) return rc;
    if (bs::launch_png_encode(static_cast<const unsigned char *>(d_rgb8), width, height, ctx->d_png_scratch[bs_ctx::kPngSingle], static_cast<unsigned char *>(d_png),
                              static_cast<uint64_t *>(d_file_bytes), s))
        return fail(BS_EDEVICE, "PNG encoder launch failed");
    return release_png(ctx, s);
} catch (...) { return bs::abi_exception("bs_encode_png_device"); }

// d_u8 (w x h RGB8 in HBM) -> the PNG file in the caller's out_png, on ctx->stream, blocking.  A page-locked out_png is written by the
// encoder's last kernel itself; otherwise the file is assembled in HBM and exactly its bytes are copied.
static int png_to_host(bs_ctx *ctx, const unsigned char *d_u8, int w, int h, unsigned char *out_png, size_t *out_bytes)
{
    bool straddles = false;
    double *alias = device_alias_of_pinned(ctx, out_png, (size_t)bs::png_file_bound(w, h), &straddles);
    if (straddles) return fail(BS_EINVAL, kStraddleMsg);
    int rc = ensure_png(ctx, bs_ctx::kPngSingle, w, h, alias == nullptr);
    if (rc) return rc;
    uint64_t *d_bytes = png_bytes_slot(ctx, bs_ctx::kPngSingle);
    if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
    unsigned char *target = alias ? reinterpret_cast<unsigned char *>(alias) : ctx->d_png_file[bs_ctx::kPngSingle];
    rc = acquire_png(ctx, ctx->stream);
    if (rc) return rc;
    if (bs::launch_png_encode(d_u8, w, h, ctx->d_png_scratch[bs_ctx::kPngSingle], target, d_bytes, ctx->stream)) return fail(BS_EDEVICE, "PNG encoder launch failed");
    rc = release_png(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t bytes = (size_t)ctx->h_png_bytes[bs_ctx::kPngSingle];
    if (!alias) {
        rc = copy_out(ctx, out_png, ctx->d_png_file[bs_ctx::kPngSingle], bytes, ctx->stream);
        if (rc) return rc;
    }
    ctx->last_zero_copy = alias != nullptr;
    *out_bytes = bytes;
    return BS_OK;
}

int bs_encode_png(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned char *out_png, size_t cap, size_t *out_bytes)
try {
    if (!ctx || !rgb8 || !out_png || !out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    if (cap < bs::png_file_bound(width, height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    BS_ON_DEVICE(ctx);
    const size_t n = (size_t)width * height * 3;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    StreamDrain drain(ctx);
    if (int rc = copy_in(ctx, ctx->d_u8, rgb8, n, ctx->stream)) return rc;
    return png_to_host(ctx, ctx->d_u8, width, height, out_png, out_bytes);
} catch (...) { return bs::abi_exception("bs_encode_png"); }
int bs_render_png(bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, unsigned char *out_png, size_t cap, size_t *out_bytes)
try {
    if (!ctx || !cfg || !out_png || !out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(cfg->width, cfg->height)) return rc;
    auto t0 = std::chrono::steady_clock::now();
    if (cap < bs::png_file_bound(cfg->width, cfg->height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    if (int rc = check_bloom_args(cfg->width, bloom_strength, bloom_divider)) return rc;
    BS_ON_DEVICE(ctx);
    const size_t n = (size_t)cfg->width * cfg->height * 3;
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    StreamDrain drain(ctx);
    // doRender (app/Main.hs:105-123) to the end: render -> bloom -> sRGB8 -> the PNG file, all on the device
    rc = enqueue_render(ctx, cfg, ctx->d_post[2], n, ctx->stream);
    if (rc) return rc;
    rc = enqueue_post_rgb8(ctx, ctx->d_post[2], cfg->width, cfg->height, bloom_strength, bloom_divider, ctx->d_u8, ctx->n_cu, ctx->stream);
    if (rc) return rc;
    rc = png_to_host(ctx, ctx->d_u8, cfg->width, cfg->height, out_png, out_bytes);
    if (rc) return rc;
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_png"); }

