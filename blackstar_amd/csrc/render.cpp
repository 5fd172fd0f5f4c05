// render.cpp -- Raytracer.render behind the C ABI: parameter derivation per launch, the launch slots (tile queue + statistics), the
// blocking and enqueue-only render entry points, the batched starLookup and bs_stats.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "bs_context.h"

namespace bs {

// row0/row1: the band of OUTPUT rows to render ([0, height) = the frame).
int fill_params(bs_ctx *ctx, const bs_config *cfg, TraceParams &p, int row0, int row1)
{
    std::string err;
    std::memset(&p, 0, sizeof p);
    if (!bs::derive_params(*cfg, p, err)) return fail(BS_EINVAL, err);
    if (row1 < 0) row1 = cfg->height;
    if (row0 < 0 || row1 > cfg->height || row0 >= row1) return fail(BS_EINVAL, "row band must satisfy 0 <= row0 < row1 <= height");
    p.band_t0 = p.ss ? 2 * row0 : row0;
    p.band_t1 = p.ss ? 2 * row1 : row1;
    p.max_steps = ctx->max_steps;
    if (!ctx->fast_guard) p.guard_steps = INT32_MAX;
    p.disk_slots = ctx->disk_slots;
    {
        const long tiles = (long)((p.wt + 7) / 8) * ((p.band_t1 - p.band_t0 + 7) / 8);
        const int cus = ctx->launch_cus > 0 ? ctx->launch_cus : ctx->n_cu;  // (a CU-masked stream offers fewer)
        const long waves = (long)cus * 4 * ctx->blocks_per_cu;  // resident wavefronts: blocks_per_cu workgroups of 4 per CU
        p.blocks_per_slot = cus;
        p.grid_blocks = (int32_t)std::max<long>(1, std::min<long>((tiles + 3) / 4, waves / 4));
        p.stagger_cycles = tiles >= (long)ctx->stagger_min_tiles * waves ? ctx->stagger_cycles : 0;  // only worth it when a wave runs several tiles
        // wavefront g starts on tile g, no initial pop (trace_kernel.hip; one binary, the knob alone, profiles/r06_static_first_tile_ab.txt: the C3
        // frame 0.9 % faster, default.yaml at 1080p 2.3 %, 640 x 360 15 %, lensing-disk at 4K level)
        p.queue_base = tiles < (long)ctx->static_first_below * waves ? 4 * p.grid_blocks : 0;
        p.late_pop_slot = ctx->late_pop_slot;
    }
    p.n_entries = (int32_t)ctx->n_entries;
    p.nodes = ctx->d_nodes;
    p.colors = ctx->d_colors;
    p.cell_start = ctx->d_cell_start;
    p.counters = ctx->d_counters;  // enqueue_render substitutes the launch slot's block
    return BS_OK;
}
int resolve_stats(bs_ctx *ctx)
{
    if (!ctx->pending) return BS_OK;
    BS_ON_DEVICE(ctx);
    bs_ctx::LaunchSlot &sl = ctx->slots[ctx->stats_slot];
    HIP_TRY(hipEventSynchronize(sl.ev_done));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
    bs_stats_t &st = ctx->stats;
    st.rays = sl.rays;
    st.steps = sl.h_counters[0];
    st.capped = sl.h_counters[1];
    st.horizon = sl.h_counters[2];
    st.escaped = sl.h_counters[3];
    st.disk_hits = sl.h_counters[4];
    st.star_hits = sl.h_counters[5];
    st.wave_iters = sl.h_counters[6];
    st.kernel_ms = ms;
    st.wall_ms = ctx->last_wall_ms;
    st.effective_mode = sl.mode;
    ctx->pending = false;
    return BS_OK;
}

// first/last: a frame (or band) delivered as several consecutive launches accumulates ONE set of statistics: the first
// launch takes the next LaunchSlot, clears its counters and records the start event (later ones reset just the tile queue
// head); the end event, the counter read-back and ev_done belong to the last.
// quiet: a batch frame (bs_render_batch) -- same slot discipline, but no timing events, no read-back, bs_stats untouched.
int enqueue_render(bs_ctx *ctx, const bs_config *cfg, double *d_out, size_t out_doubles, hipStream_t s, int row0, int row1, bool first, bool last, bool quiet)
{
    if (!ctx || !cfg || !d_out) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    int rc = fill_params(ctx, cfg, p, row0, row1);
    if (rc) return rc;
    if (row1 < 0) row1 = cfg->height;
    if (out_doubles < (size_t)cfg->width * (size_t)(row1 - row0) * 3) return fail(BS_EINVAL, "output buffer too small");
    p.out = d_out;
    BS_ON_DEVICE(ctx);
    if (first) {
        if (ctx->pending && ctx->stats_slot == ctx->next_slot) {  // bs_stats still owes the numbers of this slot's previous owner
            rc = resolve_stats(ctx);
            if (rc) return rc;
        }
        bs_ctx::LaunchSlot &sl = ctx->slots[ctx->next_slot];
        if (sl.used) HIP_TRY(hipEventSynchronize(sl.ev_done));  // its owner of kSlots renders ago (normally long finished)
        ctx->cur_slot = ctx->next_slot;
        ctx->next_slot = (ctx->next_slot + 1) % bs_ctx::kSlots;
        sl.rays = 0;
        HIP_TRY(hipMemsetAsync(sl.d_counters, 0, bs::kCounters * sizeof(unsigned long long), s));
        if (!quiet) HIP_TRY(hipEventRecord(sl.ev0, s));
    }
    if (ctx->cur_slot < 0) return fail(BS_EINTERNAL, "continuation launch without a first one");
    bs_ctx::LaunchSlot &sl = ctx->slots[ctx->cur_slot];
    if (!first) HIP_TRY(hipMemsetAsync(sl.d_counters + (bs::kCounters - 1), 0, sizeof(unsigned long long), s));  // tile queue head
    p.counters = sl.d_counters;
    sl.mode = effective_mode(ctx, cfg);
    if (bs::launch_trace(p, sl.mode, s)) return fail(BS_EDEVICE, "kernel launch failed");
    sl.rays += (uint64_t)p.wt * (uint64_t)(p.band_t1 - p.band_t0);
    if (last) {
        if (!quiet) {
            HIP_TRY(hipEventRecord(sl.ev1, s));
            HIP_TRY(hipMemcpyAsync(sl.h_counters, sl.d_counters, bs::kCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipEventRecord(sl.ev_done, s));
        sl.used = true;
        if (!quiet) {
            ctx->stats_slot = ctx->cur_slot;
            ctx->pending = true;
        }
    }
    return BS_OK;
}
}  // namespace bs

using namespace bs;

extern "C" {

int bs_render_device(bs_ctx *ctx, const bs_config *cfg, void *d_out_rgb, size_t out_doubles, void *hip_stream)
try {
    ForeignWork seen_by_destroy(ctx, hip_stream);
    return enqueue_render(ctx, cfg, static_cast<double *>(d_out_rgb), out_doubles, static_cast<hipStream_t>(hip_stream));
} catch (...) { return bs::abi_exception("bs_render_device"); }
int bs_render_rows_device(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, void *d_out_rgb, size_t out_doubles, void *hip_stream)
try {
    if (row1 < 0) return fail(BS_EINVAL, "row band must satisfy 0 <= row0 < row1 <= height");
    ForeignWork seen_by_destroy(ctx, hip_stream);
    return enqueue_render(ctx, cfg, static_cast<double *>(d_out_rgb), out_doubles, static_cast<hipStream_t>(hip_stream), row0, row1);
} catch (...) { return bs::abi_exception("bs_render_rows_device"); }

int bs_render(bs_ctx *ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles)
try {
    if (!cfg) return fail(BS_EINVAL, "null argument");
    return bs_render_rows(ctx, cfg, 0, cfg->height, out_rgb, out_doubles);
} catch (...) { return bs::abi_exception("bs_render"); }

int bs_render_rows(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, double *out_rgb, size_t out_doubles)
try {
    if (!ctx || !cfg || !out_rgb) return fail(BS_EINVAL, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    if (row0 < 0 || row1 > cfg->height || row0 >= row1) return fail(BS_EINVAL, "row band must satisfy 0 <= row0 < row1 <= height");
    auto t0 = std::chrono::steady_clock::now();
    size_t need = (size_t)cfg->width * (size_t)(row1 - row0) * 3;
    if (out_doubles < need) return fail(BS_EINVAL, "output buffer too small");
    BS_ON_DEVICE(ctx);
    bool straddles = false;
    double *alias = device_alias_of_pinned(ctx, out_rgb, need * sizeof(double), &straddles);
    if (straddles) return fail(BS_EINVAL, kStraddleMsg);
    if (alias) {  // page-locked buffer: the kernel writes it
        StreamDrain drain(ctx);
        int rc = enqueue_render(ctx, cfg, alias, need, ctx->stream, row0, row1);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->last_zero_copy = 1;
        return BS_OK;
    }
    ctx->last_zero_copy = 0;
    if (ctx->img_cap < need) {
        if (ctx->d_img) (void)hipFree(ctx->d_img);
        ctx->d_img = nullptr;
        ctx->img_cap = 0;
        if (hipMalloc((void **)&ctx->d_img, need * sizeof(double)) != hipSuccess) return fail(BS_ENOMEM, "hipMalloc image failed");
        ctx->img_cap = need;
    }
    // Host delivery of a big image into PAGEABLE memory: the frame goes out as a few consecutive launches (sub-bands of rows) and the copy
    // stream moves sub-band k to the caller while sub-band k+1 is being traced -- all but the last copy are hidden behind
    // the kernels (49.8 MB of f64 take about 1 ms to reach host memory that has been touched before, pinned or not).
    StreamDrain drain(ctx);  // no DMA into out_rgb may outlive this call, whichever way it returns
    const int rows = row1 - row0;
    int nb = need * sizeof(double) >= (size_t(8) << 20) ? ctx->host_bands : 1;
    nb = std::max(1, std::min(nb, std::min(rows / 4, (int)bs_ctx::kMaxHostBands)));
    if (nb > 1) {
        if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < nb; b++)
            if (!ctx->ev_band[b]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_band[b], hipEventDisableTiming));
    }
    const size_t row_doubles = (size_t)cfg->width * 3;
    auto cut = [&](int b) { return row0 + (int)((long)rows * b / nb); };
    for (int b = 0; b < nb; b++) {
        const int a = cut(b), e = cut(b + 1);
        int rc = enqueue_render(ctx, cfg, ctx->d_img + (size_t)(a - row0) * row_doubles, (size_t)(e - a) * row_doubles, ctx->stream, a, e, b == 0, b == nb - 1);
        if (rc) return rc;
        if (nb > 1) HIP_TRY(hipEventRecord(ctx->ev_band[b], ctx->stream));
    }
    if (nb == 1) {
        if (int rc = copy_out(ctx, out_rgb, ctx->d_img, need * sizeof(double), ctx->stream)) return rc;
    } else {
        for (int b = 0; b < nb; b++) {   // (every band's kernel is enqueued already: band b reaches the caller while band b + 1 is traced)
            const int a = cut(b), e = cut(b + 1);
            HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_band[b], 0));
            if (int rc = copy_out(ctx, out_rgb + (size_t)(a - row0) * row_doubles, ctx->d_img + (size_t)(a - row0) * row_doubles,
                                  (size_t)(e - a) * row_doubles * sizeof(double), ctx->copy_stream))
                return rc;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_rows"); }
int bs_stats(bs_ctx *ctx, bs_stats_t *out)
try {
    if (!ctx || !out) return fail(BS_EINVAL, "null argument");
    int rc = resolve_stats(ctx);
    if (rc) return rc;
    ctx->stats.wall_ms = ctx->last_wall_ms;
    ctx->stats.zero_copy = ctx->last_zero_copy;
    *out = ctx->stats;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_stats"); }

int bs_star_lookup(bs_ctx *ctx, double intensity, double saturation, const double *dirs, size_t n, double *out_rgb, int32_t *out_hits)
try {
    if (!ctx || (n && (!dirs || !out_rgb))) return fail(BS_EINVAL, "null argument");
    if (n == 0) return BS_OK;
    bs::TraceParams p;
    std::memset(&p, 0, sizeof p);
    p.star_intensity = intensity;
    p.star_saturation = saturation;
    p.star_a = std::log(2.0) / 50;
    p.n_entries = (int32_t)ctx->n_entries;
    p.nodes = ctx->d_nodes;
    p.colors = ctx->d_colors;
    p.cell_start = ctx->d_cell_start;
    BS_ON_DEVICE(ctx);
    // persistent scratch (grown on demand, kept for the life of the context): [dirs 3n | rgb 3n] doubles, then n hit counts
    int rc = ensure_scratch(ctx, 6 * n * sizeof(double) + n * sizeof(int32_t));
    if (rc) return rc;
    double *d = static_cast<double *>(ctx->d_scratch);
    int32_t *dh = reinterpret_cast<int32_t *>(d + 6 * n);
    StreamDrain drain(ctx);
    rc = copy_in(ctx, d, dirs, 3 * n * sizeof(double), ctx->stream);
    if (rc) return rc;
    if (bs::launch_star_lookup(p, d, n, d + 3 * n, dh, ctx->stream)) return fail(BS_EDEVICE, "kernel launch failed");
    rc = copy_out(ctx, out_rgb, d + 3 * n, 3 * n * sizeof(double), ctx->stream);
    if (rc) return rc;
    if (out_hits && (rc = copy_out(ctx, out_hits, dh, n * sizeof(int32_t), ctx->stream))) return rc;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_star_lookup"); }

}  // extern "C"
