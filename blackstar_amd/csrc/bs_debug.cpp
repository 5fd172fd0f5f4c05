// bs_debug.cpp -- the TEST HOOKS of include/blackstar_gpu_debug.h.  Built into blackstar_amd/libblackstar_gpu_debug.so together with
// debug_kernels.hip; that library links against the product library (libblackstar_gpu.so) and reaches into its contexts through
// bs_context.h -- both are built from one tree by one make run, and bs_debug_abi_check() refuses a product library of another layout.
// Nothing here is part of the product: `nm -D libblackstar_gpu.so | grep bs_debug` is empty, and tests, scripts and bench.py's
// issue-rate probe are the only things that load this library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/blackstar_gpu_debug.h"
#include "bs_context.h"

using namespace bs;

// Every entry point is a function-try-block ending in bs::abi_exception, like the product's (std::vector, std::string and bs::fail can
// throw; an exception must not travel through ctypes into std::terminate).  The one-line getters cannot throw.

int bs_debug_abi_check(void)
try {
    if (bs_abi_version() != BS_ABI_VERSION) return fail(BS_EINTERNAL, "libblackstar_gpu_debug.so was built against another BS_ABI_VERSION than the libblackstar_gpu.so that is loaded");
    if (bs::ctx_layout_bytes() != sizeof(bs_ctx)) return fail(BS_EINTERNAL, "libblackstar_gpu_debug.so and libblackstar_gpu.so are builds of different trees (context layout differs): rebuild both");
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_abi_check"); }

int bs_debug_srgb8_table(double table[257])
try {
    if (!table) return fail(BS_EINVAL, "null argument");
    bs::srgb8_thresholds(table);
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_srgb8_table"); }
int bs_debug_png_phases(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned long long *clocks, size_t n_clocks)
try {
    if (!ctx || !rgb8 || !clocks) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    const size_t nb = bs::png_block_count(width, height);
    if (n_clocks < nb * bs::kPngPhases) return fail(BS_EINVAL, "clocks: blocks * 23 entries are required (blocks = ceil(height * (3 width + 1) / 8192))");
    BS_ON_DEVICE(ctx);
    const size_t n = (size_t)width * height * 3;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    int rc = ensure_png(ctx, bs_ctx::kPngSingle, width, height, true);
    if (rc) return rc;
    rc = ensure_scratch(ctx, nb * bs::kPngPhases * sizeof(unsigned long long));
    if (rc) return rc;
    uint64_t *d_bytes = png_bytes_slot(ctx, bs_ctx::kPngSingle);
    if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
    StreamDrain drain(ctx);
    if ((rc = copy_in(ctx, ctx->d_u8, rgb8, n, ctx->stream))) return rc;
    HIP_TRY(hipDeviceSynchronize());  // (an enqueue-only user of the slot on a caller's stream: a probe may simply wait)
    if (bs::launch_png_encode(ctx->d_u8, width, height, ctx->d_png_scratch[bs_ctx::kPngSingle], ctx->d_png_file[bs_ctx::kPngSingle], d_bytes, ctx->stream,
                              static_cast<unsigned long long *>(ctx->d_scratch)))
        return fail(BS_EDEVICE, "PNG encoder launch failed");
    if ((rc = copy_out(ctx, clocks, ctx->d_scratch, nb * bs::kPngPhases * sizeof(unsigned long long), ctx->stream))) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_png_phases"); }
int bs_debug_set_disk_slots(bs_ctx *ctx, int slots)
try {
    if (!ctx || slots < 0 || slots > 4) return fail(BS_EINVAL, "bad slots");
    ctx->disk_slots = slots;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_set_disk_slots"); }
int bs_debug_last_post_cus(const bs_ctx *ctx) { return ctx ? ctx->last_post_cus : BS_EINVAL; }

int bs_debug_last_trial(const bs_ctx *ctx) { return ctx ? ctx->last_trial : BS_EINVAL; }

int bs_debug_partition_choice(const bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, int png, double ms[3])
try {
    if (!ctx || !cfg) return fail(BS_EINVAL, "null argument");
    const bs_ctx::PartitionKey key{cfg->width, cfg->height, cfg->supersampling ? 1 : 0, bloom_strength != 0 ? bloom_divider : 0, png ? 1 : 0,
                                   effective_mode(ctx, cfg)};
    for (const auto &c : ctx->partition_cache) {
        if (!(c.key == key)) continue;
        if (ms) std::memcpy(ms, c.ms, sizeof c.ms);
        return c.post_cus;
    }
    return -1;
} catch (...) { return bs::abi_exception("bs_debug_partition_choice"); }

int bs_debug_pick_partition(const double *ms, const int *cus, int n)
try {
    if (!ms || !cus || n <= 0) return fail(BS_EINVAL, "bad argument");
    return bs::pick_partition(ms, cus, n);
} catch (...) { return bs::abi_exception("bs_debug_pick_partition"); }

int bs_debug_forget_partitions(bs_ctx *ctx)
try {
    if (!ctx) return fail(BS_EINVAL, "null argument");
    ctx->partition_cache.clear();
    ctx->trial = {};      // a half-finished trial's segment times must not carry into the next measurement
    ctx->last_trial = 0;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_forget_partitions"); }

long bs_debug_star_grid(const bs_star *stars, size_t n_stars, uint32_t *cell_start, int32_t *entry_star, size_t cap)
try {
    if ((n_stars && !stars) || !cell_start || (cap && !entry_star)) return BS_EINVAL;
    std::vector<bs::StarNode> nodes;
    std::vector<bs::StarColor> colors;
    std::vector<uint32_t> cs;
    bs::build_star_index(stars, n_stars, nodes, colors, cs);
    std::copy(cs.begin(), cs.end(), cell_start);
    for (size_t k = 0; k < nodes.size() && k < cap; k++) entry_star[k] = nodes[k].id;
    return (long)nodes.size();
} catch (...) { return (long)bs::abi_exception("bs_debug_star_grid"); }

int bs_trace_rays(bs_ctx *ctx, const bs_config *cfg, const int32_t *yx, size_t n_rays, bs_ray_record *out)
try {
    if (!ctx || !cfg || (n_rays && (!yx || !out))) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    int rc = fill_params(ctx, cfg, p);
    if (rc) return rc;
    if (n_rays == 0) return BS_OK;
    BS_ON_DEVICE(ctx);
    const size_t yx_bytes = (n_rays * 2 * sizeof(int32_t) + 255) & ~size_t(255);
    rc = ensure_scratch(ctx, yx_bytes + n_rays * sizeof(bs_ray_record));
    if (rc) return rc;
    int32_t *d_yx = static_cast<int32_t *>(ctx->d_scratch);
    bs_ray_record *d_out = reinterpret_cast<bs_ray_record *>(static_cast<char *>(ctx->d_scratch) + yx_bytes);
    StreamDrain drain(ctx);
    if ((rc = copy_in(ctx, d_yx, yx, n_rays * 2 * sizeof(int32_t), ctx->stream))) return rc;
    if (bs::launch_trace_records(p, effective_mode(ctx, cfg), d_yx, n_rays, d_out, ctx->stream)) return fail(BS_EDEVICE, "kernel launch failed");
    if ((rc = copy_out(ctx, out, d_out, n_rays * sizeof(bs_ray_record), ctx->stream))) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_trace_rays"); }
int bs_debug_ubench(bs_ctx *ctx, int kind, int blocks, int iters, double *out_ms, double *out_ginstr)
try {
    if (!ctx || !out_ms || blocks <= 0 || iters <= 0) return fail(BS_EINVAL, "bad argument");
    BS_ON_DEVICE(ctx);
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 64));
    hipError_t e = hipSuccess;
    if (bs::launch_ubench(kind, blocks, 16, d, ctx->stream)) e = hipErrorLaunchFailure;  // warm-up
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_u0, ctx->stream);
    if (e == hipSuccess && bs::launch_ubench(kind, blocks, iters, d, ctx->stream)) e = hipErrorLaunchFailure;
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_u1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev_u1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev_u0, ctx->ev_u1);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(BS_EDEVICE, std::string("bs_debug_ubench: ") + hipGetErrorString(e));
    *out_ms = ms;
    if (out_ginstr) *out_ginstr = (double)blocks * 256.0 * iters * 32.0 / 1e9;  // lane-instructions, in 1e9
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_debug_ubench"); }

int bs_debug_sqrt_div(bs_ctx *ctx, const double *a, const double *b, size_t n, double *out_sqrt, double *out_div, int bare)
try {
    if (!ctx || (n && (!a || !b || !out_sqrt || !out_div))) return fail(BS_EINVAL, "null argument");
    if (n == 0) return BS_OK;
    BS_ON_DEVICE(ctx);
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 4 * n * sizeof(double)));
    int rc = copy_in(ctx, d, a, n * sizeof(double), ctx->stream);
    if (!rc) rc = copy_in(ctx, d + n, b, n * sizeof(double), ctx->stream);
    if (!rc && bs::launch_sqrt_div(d, d + n, n, d + 2 * n, d + 3 * n, bare, ctx->stream)) rc = fail(BS_EDEVICE, "bs_debug_sqrt_div: launch failed");
    if (!rc) rc = copy_out(ctx, out_sqrt, d + 2 * n, n * sizeof(double), ctx->stream);
    if (!rc) rc = copy_out(ctx, out_div, d + 3 * n, n * sizeof(double), ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
} catch (...) { return bs::abi_exception("bs_debug_sqrt_div"); }
