// bs_api.cpp -- the C ABI of include/blackstar_gpu.h: context lifetime, parameter derivation, launches,
// result / statistics read-back.  No CPU rendering path exists here: without a HIP device every render
// entry point fails with BS_EDEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "bs_internal.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(BS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct bs_ctx {
    int device = -1;
    int mode = BS_MODE_FAST;
    int max_steps = 100000;
    int disk_slots = 4;
    bool zero_copy = true;       // page-locked caller buffers are written by the kernel itself (env BLACKSTAR_ZERO_COPY=0: always stage + copy)
    bool fast_guard = true;      // FAST mode re-traces photon-sphere-grazing rays in STRICT (env BLACKSTAR_FAST_GUARD=0 turns it off for A/B)
    int n_cu = 256;
    int blocks_per_cu = 4;       // resident workgroups per CU (VGPR/LDS-limited); env BLACKSTAR_BLOCKS_PER_CU for A/B builds
    int stagger_cycles = 16000;  // first-tile phase offset per SIMD slot (env BLACKSTAR_STAGGER overrides; 0 = off)
    int stagger_min_tiles = 6;   // ... applied to launches of at least this many tiles per resident wavefront (env BLACKSTAR_STAGGER_MIN_TILES; C2 = 7.9 tiles per wavefront gains 2 %, frames of 3-5 do not)
    size_t n_stars = 0;
    bs::StarNode *d_nodes = nullptr;
    bs::StarColor *d_colors = nullptr;
    uint32_t *d_cell_start = nullptr;
    size_t n_entries = 0;  // stars + border duplicates in the direction grid
    // Every render (one launch, or the consecutive launches of one host-delivered frame) owns a LaunchSlot: its tile queue
    // head + statistics block in HBM, the pinned landing area of that block and its events.  Launches of one context on
    // DIFFERENT streams therefore never share a queue head (two persistent kernels popping one counter would each skip
    // the tiles the other took); a slot is reused kSlots renders later, after waiting for its previous owner.
    struct LaunchSlot {
        unsigned long long *d_counters = nullptr;  // device, bs::kCounters
        unsigned long long *h_counters = nullptr;  // pinned, bs::kCounters
        hipEvent_t ev0 = nullptr, ev1 = nullptr;   // kernel start / kernel end (timing)
        hipEvent_t ev_done = nullptr;              // everything of the render (incl. the counter read-back) has been enqueued before it
        bool used = false;
        uint64_t rays = 0;
        int mode = BS_MODE_FAST;                   // the arithmetic this render was traced with (effective_mode)
    };
    static constexpr int kSlots = 8;
    LaunchSlot slots[kSlots];
    unsigned long long *d_counters = nullptr;  // kSlots * bs::kCounters, carved into the slots
    unsigned long long *h_counters = nullptr;  // pinned, same shape
    int next_slot = 0;
    int cur_slot = -1;    // slot of the render being enqueued (first .. last launch)
    int stats_slot = -1;  // slot whose statistics bs_stats reports
    void *d_scratch = nullptr;  // persistent device scratch of the batched hooks (bs_star_lookup, bs_trace_rays, ...)
    size_t scratch_cap = 0;
    double *d_img = nullptr;                   // scratch image for bs_render (host-output variant)
    size_t img_cap = 0;
    double *d_img2 = nullptr;                  // second image + copy stream: bs_render_batch overlaps frame i's D2H with frame i+1's kernel
    size_t img2_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_frame[2] = {nullptr, nullptr};
    hipStream_t stream2 = nullptr;                  // bs_render_batch: odd frames run on a second compute stream, so that a frame's
                                                    // first wavefronts fill the slots the previous frame's last tiles leave idle
    double *d_post[3] = {nullptr, nullptr, nullptr};  // bloom ping-pong buffers + host-variant staging
    size_t post_cap = 0;
    hipEvent_t ev_post = nullptr;     // recorded behind the last user of d_post[0..1]; a user on another stream waits for it first
    hipStream_t post_stream = nullptr;
    bool post_busy = false;
    unsigned char *d_u8 = nullptr;
    size_t u8_cap = 0;
    unsigned char *d_u8b = nullptr;  // bs_render_rgb8_batch: staging of the frame on the second stream (pageable outputs only)
    size_t u8b_cap = 0;
    double *d_srgb_table = nullptr;  // 257 thresholds of the sRGB8 pixel map (bs::srgb8_thresholds)
    hipStream_t stream = nullptr;
    hipEvent_t ev_u0 = nullptr, ev_u1 = nullptr;  // bs_debug_ubench timing
    static constexpr int kMaxHostBands = 8;
    int host_bands = 2;  // bs_render[_rows]: launches per frame (2 measured best: 5.22 ms vs 5.62 with 1 and 5.35 with 4 for a 1080p 4xSS frame), so that band k's device-to-host copy overlaps band k+1's kernel (env BLACKSTAR_HOST_BANDS)
    hipEvent_t ev_band[kMaxHostBands] = {};
    bool pending = false;  // a render has been enqueued whose stats were not read back yet
    double last_wall_ms = 0;
    int last_zero_copy = 0;  // the last blocking render wrote the caller's page-locked buffer itself (no device image, no copy)
    // bs_render_rgb8_batch with a PARTITIONED chip: the trace kernels run on streams whose CU mask leaves M CUs out (M / 8 in every
    // XCD), and bloom + sRGB8 run on a stream that owns exactly those -- see render_rgb8_frames_partitioned / choose_post_cus
    int post_cus_req = -1;       // env BLACKSTAR_POST_CUS: -1 = choose per batch (default), 0 = never partition, 8..32 = always that many
    int post_plan_cus = 0;       // CUs the blur sweeps are PLANNED for on the post stream (env BLACKSTAR_POST_PLAN_CUS; 0 = the partition's)
    int launch_cus = 0;          // CUs the next trace launches may use (0 = n_cu): sizes the persistent grid
    int last_post_cus = -1;      // CUs the post stage owned in this context's share of the last bs_render_rgb8_batch (0: shared chip; -1: none yet)
    int bloom_plan_cus = 0;      // probe only (env BLACKSTAR_BLOOM_PLAN_CUS): CU count bs_bloom_device plans its sweeps for (0 = n_cu)
    struct Partition {
        hipStream_t trace[2] = {nullptr, nullptr};  // CU mask: every CU but the post stage's
        hipStream_t post = nullptr;                 // CU mask: the post stage's CUs
    };
    static constexpr int kPartitions = 7;           // post stage on 8, 12, ... 32 CUs (at least one in every XCD: an XCD without a mask bit gets all its CUs)
    Partition parts[kPartitions];
    hipEvent_t ev_traced[3] = {nullptr, nullptr, nullptr}, ev_posted[3] = {nullptr, nullptr, nullptr};
    double *d_img3 = nullptr;
    size_t img3_cap = 0;
    unsigned char *d_u8c = nullptr;
    size_t u8c_cap = 0;
    // writeImg's PNG encoder on the device (png_kernels.hip): per frame in flight its scratch (filter types, chunk sizes / offsets, staging
    // slots), a device copy of the file for callers with pageable buffers, and the file's size in page-locked memory
    static constexpr int kPngSlots = 3;
    unsigned char *d_png_scratch[kPngSlots] = {nullptr, nullptr, nullptr};
    size_t png_scratch_cap[kPngSlots] = {0, 0, 0};
    unsigned char *d_png_file[kPngSlots] = {nullptr, nullptr, nullptr};
    size_t png_file_cap[kPngSlots] = {0, 0, 0};
    uint64_t *h_png_bytes = nullptr;   // page-locked, kPngSlots entries (the kernels write them through the device alias)
    hipEvent_t ev_png = nullptr;       // behind the last user of PNG slot 0 (the enqueue-only entry point): see acquire_png
    hipStream_t png_stream = nullptr;
    bool png_busy = false;
    // page-locked file buffers bs_render_png_files keeps between calls (page-locking 6 MB costs 1-2 ms): (pointer, capacity)
    std::vector<std::pair<unsigned char *, size_t>> file_pool;
    struct VerifiedRange { const void *host = nullptr; size_t bytes = 0; };
    VerifiedRange verified[8];  // host buffers device_alias_of_pinned has walked page by page (registered memory without a queryable range)
    int verified_next = 0;
    bs_stats_t stats{};
};

namespace {

// row0/row1: the band of OUTPUT rows to render ([0, height) = the frame).
int fill_params(bs_ctx *ctx, const bs_config *cfg, bs::TraceParams &p, int row0 = 0, int row1 = -1)
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
    }
    p.n_entries = (int32_t)ctx->n_entries;
    p.nodes = ctx->d_nodes;
    p.colors = ctx->d_colors;
    p.cell_start = ctx->d_cell_start;
    p.counters = ctx->d_counters;  // enqueue_render substitutes the launch slot's block
    return BS_OK;
}

// The arithmetic a frame is traced with.  FAST's error model (rounding differences of ~1 ulp per right-hand side, amplified by
// the discrete map) needs the RK4 step to resolve the field: with stepSize above 0.5 Schwarzschild radii a single step past the
// hole amplifies a perturbation by >10x and FAST and STRICT trajectories part ways (scripts/fuzz_worst.py: terminal directions
// 4e-3 apart at stepSize 1.0, all of the fuzz's largest colour deviations), so such frames are traced with STRICT arithmetic even
// in FAST mode -- the reference's default is 0.3 and every scene file it ships uses that.  (BLACKSTAR_FAST_GUARD=0: off, for A/B.)
int effective_mode(const bs_ctx *ctx, const bs_config *cfg)
{
    if (ctx->mode == BS_MODE_FAST && ctx->fast_guard && !(cfg->step_size <= 0.5)) return BS_MODE_STRICT;
    return ctx->mode;
}

// On every exit path of a blocking entry point nothing of the call may still be in flight: the caller's buffers are DMA
// targets, and after an error return the caller is free to release them.  (On the success path the streams have been
// synchronised already and this costs a few microseconds.)
struct StreamDrain {
    bs_ctx *ctx;
    explicit StreamDrain(bs_ctx *c) : ctx(c) {}
    StreamDrain(const StreamDrain &) = delete;
    StreamDrain &operator=(const StreamDrain &) = delete;
    ~StreamDrain()
    {
        if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return;
        for (hipStream_t s : {ctx->stream, ctx->stream2, ctx->copy_stream})
            if (s) (void)hipStreamSynchronize(s);
        for (const bs_ctx::Partition &pt : ctx->parts)
            for (hipStream_t s : {pt.trace[0], pt.trace[1], pt.post})
                if (s) (void)hipStreamSynchronize(s);
    }
};

// Zero copy: if a caller's HOST buffer is page-locked (bs_host_alloc, hipHostMalloc, hipHostRegister) the device can write it
// directly over PCIe, so the kernel's own image stores deliver the frame -- no device image, no copy, and the transfer is
// spread over the whole kernel instead of trailing it.  Measured (scripts/zero_copy_probe.py, profiles/r02_zero_copy.txt):
// bs_render of the C3 frame 5.45 -> 4.57 ms (kernel 4.38), C2 2.14 -> 1.49 ms (49.8 MB in 1.47 ms = 34 GB/s while tracing), C4
// 21.6 -> 18.7 ms; the kernel time itself does not change (11 GB/s average is far below what PCIe takes in 96-B segments).
// Returns the device alias of `host`, or nullptr for pageable memory (which takes the staged path).  BLACKSTAR_ZERO_COPY=0: off.
// *straddles (optional): set when `host` STARTS in page-locked memory but [host, host + bytes) is not contained in it -- a buffer
// no path can deliver into: the kernel's stores would fault, and the runtime's own hipMemcpyAsync refuses it ("invalid argument":
// it finds the registered range the pointer starts in and the size does not fit).  Callers turn that into BS_EINVAL.
double *device_alias_of_pinned(bs_ctx *ctx, const void *host, size_t bytes, bool *straddles = nullptr)
{
    if (straddles) *straddles = false;
    if (!host || bytes == 0) return nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess || a.type != hipMemoryTypeHost || !a.devicePointer) {
        (void)hipGetLastError();  // pageable memory is reported as an error: not one of ours
        return nullptr;
    }
    // [host, host + bytes) must lie inside page-locked memory from end to end.  Probing the two ends is not enough: a buffer that
    // starts in one hipHostRegister range and ends in another, with pageable memory in between, passes that test, and the kernel's
    // stores through base alias + offset then fault on the GPU -- which ends the process instead of returning an error.
    const char *hp = static_cast<const char *>(host);
    bool covered = false, known = false;
    void *base = nullptr;
    size_t size = 0;
    // (1) the range the pointer belongs to, as the driver-style attributes report it: exact for hipHostMalloc (bs_host_alloc,
    //     torch's pinned allocator) AND for hipHostRegister'ed memory -- for which hipMemGetAddressRange on ROCm 7.2 returns the
    //     size but a NULL base (scripts/pinned_probe.py -> profiles/r03_pinned_probe.txt)
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, const_cast<void *>(host)) == hipSuccess && base &&
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, const_cast<void *>(host)) == hipSuccess && size) {
        const char *hb = static_cast<const char *>(base);
        known = true;
        covered = hp >= hb && bytes <= size && static_cast<size_t>(hp - hb) <= size - bytes;
    } else {
        (void)hipGetLastError();
        base = nullptr;
        size = 0;
        if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, a.devicePointer) == hipSuccess && base) {
            const char *db = static_cast<const char *>(base), *dp = static_cast<const char *>(a.devicePointer);
            known = true;
            covered = dp >= db && bytes <= size && static_cast<size_t>(dp - db) <= size - bytes;
        } else {
            (void)hipGetLastError();
        }
    }
    if (!known) {
        // (2) no range to be had: walk the buffer page by page (every page page-locked, the device alias contiguous) -- once per
        //     buffer: the last few verified (pointer, size) pairs are remembered, callers reuse their frame buffers
        for (const auto &v : ctx->verified)
            if (v.host == host && v.bytes == bytes) covered = true;
        if (!covered) {
            const char *dp = static_cast<const char *>(a.devicePointer);
            const uintptr_t page = 4096;
            covered = true;
            for (uintptr_t q = (reinterpret_cast<uintptr_t>(hp) & ~(page - 1)) + page; covered && q < reinterpret_cast<uintptr_t>(hp) + bytes; q += page) {
                hipPointerAttribute_t b;
                if (hipPointerGetAttributes(&b, reinterpret_cast<const void *>(q)) != hipSuccess || b.type != hipMemoryTypeHost ||
                    static_cast<const char *>(b.devicePointer) - reinterpret_cast<const char *>(q) != dp - hp) {
                    (void)hipGetLastError();
                    covered = false;
                }
            }
            if (covered) {
                ctx->verified[ctx->verified_next] = {host, bytes};
                ctx->verified_next = (ctx->verified_next + 1) % (int)(sizeof ctx->verified / sizeof ctx->verified[0]);
            }
        }
    }
    if (!covered) {
        if (straddles) *straddles = true;
        return nullptr;
    }
    if (!ctx->zero_copy) return nullptr;  // BLACKSTAR_ZERO_COPY=0: stage + copy (the runtime copies into page-locked memory directly)
    // page-locked for ANOTHER device only (hipHostMalloc / hipHostRegister there without the Portable flag): not ours to write
    if (a.device != ctx->device && !(a.allocationFlags & hipHostMallocPortable)) return nullptr;
    return static_cast<double *>(a.devicePointer);
}

const char *kStraddleMsg = "output buffer starts in page-locked memory but is not contained in it (it runs past the end of its hipHostMalloc / "
                           "hipHostRegister range, e.g. into pageable memory between two registered ranges): neither the kernel nor the runtime's copy can deliver into it";

template <class T>
bool grow_device(T *&buf, size_t &cap, size_t elems)
{
    if (cap >= elems) return true;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    cap = 0;
    if (hipMalloc((void **)&buf, elems * sizeof(T)) != hipSuccess) return false;
    cap = elems;
    return true;
}

// BLACKSTAR_POST_CUS as a number: 0 = never partition, otherwise a multiple of 4 in [8, 32] (rounded down, at least 8)
int post_cus_setting(int v)
{
    if (v <= 0) return 0;
    return std::max(8, std::min(32, v) / 4 * 4);
}

int ensure_scratch(bs_ctx *ctx, size_t bytes)
{
    if (ctx->scratch_cap >= bytes) return BS_OK;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_cap = 0;
    const size_t cap = std::max<size_t>(bytes, size_t(1) << 20);
    if (hipMalloc(&ctx->d_scratch, cap) != hipSuccess) return fail(BS_ENOMEM, "hipMalloc scratch failed");
    ctx->scratch_cap = cap;
    return BS_OK;
}

int resolve_stats(bs_ctx *ctx)
{
    if (!ctx->pending) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
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
int enqueue_render(bs_ctx *ctx, const bs_config *cfg, double *d_out, size_t out_doubles, hipStream_t s, int row0 = 0, int row1 = -1,
                   bool first = true, bool last = true, bool quiet = false)
{
    if (!ctx || !cfg || !d_out) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    int rc = fill_params(ctx, cfg, p, row0, row1);
    if (rc) return rc;
    if (row1 < 0) row1 = cfg->height;
    if (out_doubles < (size_t)cfg->width * (size_t)(row1 - row0) * 3) return fail(BS_EINVAL, "output buffer too small");
    p.out = d_out;
    HIP_TRY(hipSetDevice(ctx->device));
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

}  // namespace

extern "C" {

int bs_abi_version(void) { return BS_ABI_VERSION; }

const char *bs_last_error(void) { return g_err.c_str(); }

bs_ctx *bs_create(int device, const bs_star *stars, size_t n_stars)
{
    if (device < 0) { fail(BS_EDEVICE, "this library has no CPU backend: device must be a HIP device ordinal >= 0"); return nullptr; }
    if (n_stars && !stars) { fail(BS_EINVAL, "stars is null"); return nullptr; }
    if (n_stars >= (size_t(1) << 30)) { fail(BS_EINVAL, "too many stars"); return nullptr; }
    for (size_t i = 0; i < n_stars; i++) {
        double h = stars[i].hue * 2 * 3.141592653589793;
        if (!(h >= 0 && h < 2 * 3.141592653589793)) { fail(BS_EINVAL, "HSI pixel is not properly scaled (star hue outside [0,1))"); return nullptr; }
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device >= count) {
        fail(BS_EDEVICE, e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no such HIP device");
        return nullptr;
    }
    bs_ctx *ctx = new (std::nothrow) bs_ctx();
    if (!ctx) { fail(BS_ENOMEM, "out of host memory"); return nullptr; }
    ctx->device = device;
    ctx->n_stars = n_stars;
    if (const char *m = std::getenv("BLACKSTAR_STAGGER")) ctx->stagger_cycles = std::atoi(m);
    if (const char *m = std::getenv("BLACKSTAR_STAGGER_MIN_TILES")) ctx->stagger_min_tiles = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_FAST_GUARD")) ctx->fast_guard = std::atoi(m) != 0;
    if (const char *m = std::getenv("BLACKSTAR_ZERO_COPY")) ctx->zero_copy = std::atoi(m) != 0;
    if (const char *m = std::getenv("BLACKSTAR_HOST_BANDS")) ctx->host_bands = std::max(1, std::min((int)bs_ctx::kMaxHostBands, std::atoi(m)));
    if (const char *m = std::getenv("BLACKSTAR_BLOCKS_PER_CU")) ctx->blocks_per_cu = std::max(1, std::min(8, std::atoi(m)));
    if (const char *m = std::getenv("BLACKSTAR_POST_CUS")) ctx->post_cus_req = std::strcmp(m, "auto") ? post_cus_setting(std::atoi(m)) : -1;
    if (const char *m = std::getenv("BLACKSTAR_POST_PLAN_CUS")) ctx->post_plan_cus = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_BLOOM_PLAN_CUS")) ctx->bloom_plan_cus = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_MODE")) {
        if (!std::strcmp(m, "fast")) ctx->mode = BS_MODE_FAST;
        else if (!std::strcmp(m, "strict")) ctx->mode = BS_MODE_STRICT;
    }
    std::vector<bs::StarNode> nodes;
    std::vector<bs::StarColor> colors;
    std::vector<uint32_t> cell_start;
    bs::build_star_index(stars, n_stars, nodes, colors, cell_start);
    ctx->n_entries = nodes.size();
    auto ok = [&](hipError_t r, const char *what) {
        if (r == hipSuccess) return true;
        fail(BS_EDEVICE, std::string(what) + ": " + hipGetErrorString(r));
        return false;
    };
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    bool good = ok(hipSetDevice(device), "hipSetDevice") &&
                ok(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate") &&
                ok(hipEventCreate(&ctx->ev_u0), "hipEventCreate") && ok(hipEventCreate(&ctx->ev_u1), "hipEventCreate") &&
                ok(hipMalloc((void **)&ctx->d_nodes, std::max<size_t>(1, nodes.size()) * sizeof(bs::StarNode)), "hipMalloc nodes") &&
                ok(hipMalloc((void **)&ctx->d_colors, std::max<size_t>(1, colors.size()) * sizeof(bs::StarColor)), "hipMalloc colors") &&
                ok(hipMalloc((void **)&ctx->d_cell_start, cell_start.size() * sizeof(uint32_t)), "hipMalloc cell_start") &&
                ok(hipMemcpy(ctx->d_cell_start, cell_start.data(), cell_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "upload cell_start") &&
                ok(hipMalloc((void **)&ctx->d_counters, bs_ctx::kSlots * bs::kCounters * sizeof(unsigned long long)), "hipMalloc counters") &&
                ok(hipHostMalloc((void **)&ctx->h_counters, bs_ctx::kSlots * bs::kCounters * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc") &&
                ok(hipMemcpy(ctx->d_nodes, nodes.data(), nodes.size() * sizeof(bs::StarNode), hipMemcpyHostToDevice), "upload nodes") &&
                ok(hipMemcpy(ctx->d_colors, colors.data(), colors.size() * sizeof(bs::StarColor), hipMemcpyHostToDevice), "upload colors");
    if (good) {
        static const std::vector<double> table = [] { std::vector<double> t(257); bs::srgb8_thresholds(t.data()); return t; }();
        good = ok(hipMalloc((void **)&ctx->d_srgb_table, 257 * sizeof(double)), "hipMalloc srgb table") &&
               ok(hipMemcpy(ctx->d_srgb_table, table.data(), 257 * sizeof(double), hipMemcpyHostToDevice), "upload srgb table");
    }
    for (int k = 0; good && k < bs_ctx::kSlots; k++) {
        bs_ctx::LaunchSlot &sl = ctx->slots[k];
        sl.d_counters = ctx->d_counters + (size_t)k * bs::kCounters;
        sl.h_counters = ctx->h_counters + (size_t)k * bs::kCounters;
        good = ok(hipEventCreate(&sl.ev0), "hipEventCreate") && ok(hipEventCreate(&sl.ev1), "hipEventCreate") &&
               ok(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming), "hipEventCreate");
    }
    if (!good) {
        std::string keep = g_err;
        bs_destroy(ctx);
        g_err = keep;
        return nullptr;
    }
    return ctx;
}

int bs_debug_srgb8_table(double table[257])
{
    if (!table) return fail(BS_EINVAL, "null argument");
    bs::srgb8_thresholds(table);
    return BS_OK;
}

int bs_device_count(void)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return fail(BS_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return count;
}

void bs_destroy(bs_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device >= 0 && hipSetDevice(ctx->device) == hipSuccess) {
        (void)hipDeviceSynchronize();
        if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
        if (ctx->d_colors) (void)hipFree(ctx->d_colors);
        if (ctx->d_cell_start) (void)hipFree(ctx->d_cell_start);
        if (ctx->d_counters) (void)hipFree(ctx->d_counters);
        if (ctx->d_img) (void)hipFree(ctx->d_img);
        if (ctx->d_img2) (void)hipFree(ctx->d_img2);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        for (bs_ctx::LaunchSlot &sl : ctx->slots)
            for (hipEvent_t e : {sl.ev0, sl.ev1, sl.ev_done})
                if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->ev_frame)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->ev_band)
            if (e) (void)hipEventDestroy(e);
        for (double *b : ctx->d_post)
            if (b) (void)hipFree(b);
        if (ctx->d_u8) (void)hipFree(ctx->d_u8);
        if (ctx->d_u8b) (void)hipFree(ctx->d_u8b);
        if (ctx->d_u8c) (void)hipFree(ctx->d_u8c);
        if (ctx->d_img3) (void)hipFree(ctx->d_img3);
        for (unsigned char *b : ctx->d_png_scratch)
            if (b) (void)hipFree(b);
        for (unsigned char *b : ctx->d_png_file)
            if (b) (void)hipFree(b);
        if (ctx->h_png_bytes) (void)hipHostFree(ctx->h_png_bytes);
        for (auto &b : ctx->file_pool)
            if (b.first) (void)hipHostFree(b.first);
        if (ctx->ev_png) (void)hipEventDestroy(ctx->ev_png);
        for (bs_ctx::Partition &pt : ctx->parts)
            for (hipStream_t st : {pt.trace[0], pt.trace[1], pt.post})
                if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t e : ctx->ev_traced)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->ev_posted)
            if (e) (void)hipEventDestroy(e);
        if (ctx->ev_post) (void)hipEventDestroy(ctx->ev_post);
        if (ctx->d_srgb_table) (void)hipFree(ctx->d_srgb_table);
        if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
        if (ctx->ev_u0) (void)hipEventDestroy(ctx->ev_u0);
        if (ctx->ev_u1) (void)hipEventDestroy(ctx->ev_u1);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

int bs_set_mode(bs_ctx *ctx, int mode)
{
    if (!ctx || (mode != BS_MODE_STRICT && mode != BS_MODE_FAST)) return fail(BS_EINVAL, "bad mode");
    ctx->mode = mode;
    return BS_OK;
}

int bs_get_mode(const bs_ctx *ctx) { return ctx ? ctx->mode : BS_EINVAL; }

int bs_validate_config(const bs_config *cfg)
{
    if (!cfg) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    std::memset(&p, 0, sizeof p);
    std::string err;
    if (!bs::derive_params(*cfg, p, err)) return fail(BS_EINVAL, err);
    return BS_OK;
}

int bs_effective_mode(const bs_ctx *ctx, const bs_config *cfg)
{
    if (!ctx || !cfg) return fail(BS_EINVAL, "null argument");
    return effective_mode(ctx, cfg);
}

int bs_set_max_steps(bs_ctx *ctx, int max_steps)
{
    if (!ctx || max_steps <= 0) return fail(BS_EINVAL, "bad max_steps");
    // the kernel counts a ray's steps in an int and the frame's in 64 bits: 2^30 rays (the largest frame) x 2^30 steps = 2^60
    if (max_steps > BS_MAX_STEPS_LIMIT) return fail(BS_EINVAL, "max_steps above BS_MAX_STEPS_LIMIT (2^30): the step counters could not hold a frame of capped rays");
    ctx->max_steps = max_steps;
    return BS_OK;
}

static int ensure_post(bs_ctx *ctx, size_t n)
{
    if (ctx->post_cap >= n) return BS_OK;
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
{
    if (!ctx || !d_in || !d_out || width <= 0 || height <= 0) return fail(BS_EINVAL, "bad argument");
    if (divider <= 0 || width / divider == 0)  // the reference crashes here: foldl1' over an empty window (ImageFilters.hs:59)
        return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t n = (size_t)width * height * 3;
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    rc = acquire_post(ctx, static_cast<hipStream_t>(hip_stream));
    if (rc) return rc;
    if (bs::launch_bloom((const double *)d_in, (double *)d_out, ctx->d_post[0], ctx->d_post[1], width, height, strength, divider,
                         ctx->bloom_plan_cus > 0 ? ctx->bloom_plan_cus : ctx->n_cu, hip_stream))
        return fail(BS_EDEVICE, "bloom launch failed");
    return release_post(ctx, static_cast<hipStream_t>(hip_stream));
}

int bs_bloom(bs_ctx *ctx, const double *in, double *out, int width, int height, double strength, int divider)
{
    if (!ctx || !in || !out || width <= 0 || height <= 0) return fail(BS_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t n = (size_t)width * height * 3;
    int rc = ensure_post(ctx, n);
    if (rc) return rc;
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(ctx->d_post[2], in, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = bs_bloom_device(ctx, ctx->d_post[2], ctx->d_post[2], width, height, strength, divider, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->d_post[2], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

int bs_supersample(bs_ctx *ctx, const double *in, double *out, int width2, int height2)
{
    if (!ctx || !in || !out || width2 < 0 || height2 < 0) return fail(BS_EINVAL, "bad argument");
    const size_t n_in = (size_t)width2 * height2 * 3, n_out = (size_t)(width2 / 2) * (height2 / 2) * 3;
    if (n_out == 0) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_post(ctx, n_in);
    if (rc) return rc;
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(ctx->d_post[2], in, n_in * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = acquire_post(ctx, ctx->stream);
    if (rc) return rc;
    if (bs::launch_supersample(ctx->d_post[2], ctx->d_post[0], width2, height2, ctx->stream)) return fail(BS_EDEVICE, "supersample launch failed");
    rc = release_post(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->d_post[0], n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

int bs_srgb8_device(bs_ctx *ctx, const void *d_in, void *d_out_u8, size_t n_values, void *hip_stream)
{
    if (!ctx || (n_values && (!d_in || !d_out_u8))) return fail(BS_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (bs::launch_srgb8((const double *)d_in, (unsigned char *)d_out_u8, n_values, ctx->d_srgb_table, hip_stream)) return fail(BS_EDEVICE, "srgb8 launch failed");
    return BS_OK;
}

int bs_srgb8(bs_ctx *ctx, const double *in, unsigned char *out, size_t n_values)
{
    if (!ctx || (n_values && (!in || !out))) return fail(BS_EINVAL, "bad argument");
    if (n_values == 0) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
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
    HIP_TRY(hipMemcpyAsync(ctx->d_post[2], in, n_values * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = bs_srgb8_device(ctx, ctx->d_post[2], ctx->d_u8, n_values, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->d_u8, n_values, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

static int check_bloom_args(int width, double strength, int divider)
{
    if (strength != 0 && (divider <= 0 || width / divider == 0))  // the reference crashes here: foldl1' over an empty window (ImageFilters.hs:59)
        return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
    return BS_OK;
}

// What doRender does with the rendered image (app/Main.hs:113-123): bloom when bloomStrength /= 0, then writeImg's pixel map -- d_img
// (f64, w x h x 3) -> d_u8 (RGB8), on stream s.  The final img + strength * blurred is fused with the sRGB8 map: the bloomed f64
// image is never written.  plan_cus: the CUs the blur sweeps are planned for.
static int enqueue_post_rgb8(bs_ctx *ctx, const double *d_img, int w, int h, double strength, int divider, unsigned char *d_u8, int plan_cus, hipStream_t s)
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
{
    if (!ctx || !cfg || !out_rgb8) return fail(BS_EINVAL, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = (size_t)cfg->width * cfg->height * 3;
    if (out_bytes < n) return fail(BS_EINVAL, "output buffer too small");
    if (int rc = check_bloom_args(cfg->width, bloom_strength, bloom_divider)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
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
    if (u8_target == ctx->d_u8) HIP_TRY(hipMemcpyAsync(out_rgb8, ctx->d_u8, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_zero_copy = u8_target != ctx->d_u8;
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BS_OK;
}

// ---- writeImg's file on the device (png_kernels.hip) --------------------------------------------------------------------------------

static int check_png_frame(int width, int height)
{
    if (width <= 0 || height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    if ((double)width * (double)height > 1.0e9 || bs::png_file_bound(width, height) > 0xFFFFFFFFull)
        return fail(BS_EINVAL, "frame too large for one PNG file of this encoder (chunk offsets are 32 bits)");
    return BS_OK;
}

int bs_png_bound(int width, int height, size_t *out_bytes)
{
    if (!out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    *out_bytes = (size_t)bs::png_file_bound(width, height);
    return BS_OK;
}

// PNG slot k of the context sized for a w x h frame: the encoder's scratch, the page-locked size slots, and (device_file) a device copy
// of the file for a caller whose buffer the GPU cannot write.
static int ensure_png(bs_ctx *ctx, int k, int w, int h, bool device_file)
{
    if (!grow_device(ctx->d_png_scratch[k], ctx->png_scratch_cap[k], bs::png_scratch_bytes(w, h)))
        return fail(BS_ENOMEM, "hipMalloc PNG scratch failed");
    if (device_file && !grow_device(ctx->d_png_file[k], ctx->png_file_cap[k], (size_t)bs::png_file_bound(w, h)))
        return fail(BS_ENOMEM, "hipMalloc PNG file failed");
    if (!ctx->h_png_bytes) HIP_TRY(hipHostMalloc((void **)&ctx->h_png_bytes, bs_ctx::kPngSlots * sizeof(uint64_t), hipHostMallocDefault));
    return BS_OK;
}

static uint64_t *png_bytes_slot(bs_ctx *ctx, int k)
{
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, ctx->h_png_bytes, 0) != hipSuccess || !d) return nullptr;
    return static_cast<uint64_t *>(d) + k;
}

// PNG slot 0 serves the enqueue-only entry point: like the blur scratch, a user on another stream first waits for the previous one.
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
{
    if (!ctx || !d_rgb8 || !d_png || !d_file_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    if (cap < bs::png_file_bound(width, height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_png(ctx, 0, width, height, false);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    rc = acquire_png(ctx, s);
    if (rc) return rc;
    if (bs::launch_png_encode(static_cast<const unsigned char *>(d_rgb8), width, height, ctx->d_png_scratch[0], static_cast<unsigned char *>(d_png),
                              static_cast<uint64_t *>(d_file_bytes), s))
        return fail(BS_EDEVICE, "PNG encoder launch failed");
    return release_png(ctx, s);
}

// d_u8 (w x h RGB8 in HBM) -> the PNG file in the caller's out_png, on ctx->stream, blocking.  A page-locked out_png is written by the
// encoder's last kernel itself; otherwise the file is assembled in HBM and exactly its bytes are copied.
static int png_to_host(bs_ctx *ctx, const unsigned char *d_u8, int w, int h, unsigned char *out_png, size_t *out_bytes)
{
    bool straddles = false;
    double *alias = device_alias_of_pinned(ctx, out_png, (size_t)bs::png_file_bound(w, h), &straddles);
    if (straddles) return fail(BS_EINVAL, kStraddleMsg);
    int rc = ensure_png(ctx, 0, w, h, alias == nullptr);
    if (rc) return rc;
    uint64_t *d_bytes = png_bytes_slot(ctx, 0);
    if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
    unsigned char *target = alias ? reinterpret_cast<unsigned char *>(alias) : ctx->d_png_file[0];
    rc = acquire_png(ctx, ctx->stream);
    if (rc) return rc;
    if (bs::launch_png_encode(d_u8, w, h, ctx->d_png_scratch[0], target, d_bytes, ctx->stream)) return fail(BS_EDEVICE, "PNG encoder launch failed");
    rc = release_png(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t bytes = (size_t)ctx->h_png_bytes[0];
    if (!alias) {
        HIP_TRY(hipMemcpyAsync(out_png, ctx->d_png_file[0], bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    ctx->last_zero_copy = alias != nullptr;
    *out_bytes = bytes;
    return BS_OK;
}

int bs_encode_png(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned char *out_png, size_t cap, size_t *out_bytes)
{
    if (!ctx || !rgb8 || !out_png || !out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    if (cap < bs::png_file_bound(width, height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)width * height * 3;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(ctx->d_u8, rgb8, n, hipMemcpyHostToDevice, ctx->stream));
    return png_to_host(ctx, ctx->d_u8, width, height, out_png, out_bytes);
}

int bs_debug_png_phases(bs_ctx *ctx, const unsigned char *rgb8, int width, int height, unsigned long long *clocks, size_t n_clocks)
{
    if (!ctx || !rgb8 || !clocks) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(width, height)) return rc;
    const size_t nb = bs::png_block_count(width, height);
    if (n_clocks < nb * bs::kPngPhases) return fail(BS_EINVAL, "clocks: blocks * 23 entries are required (blocks = ceil(height * (3 width + 1) / 8192))");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)width * height * 3;
    if (!grow_device(ctx->d_u8, ctx->u8_cap, n)) return fail(BS_ENOMEM, "hipMalloc failed");
    int rc = ensure_png(ctx, 0, width, height, true);
    if (rc) return rc;
    rc = ensure_scratch(ctx, nb * bs::kPngPhases * sizeof(unsigned long long));
    if (rc) return rc;
    uint64_t *d_bytes = png_bytes_slot(ctx, 0);
    if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(ctx->d_u8, rgb8, n, hipMemcpyHostToDevice, ctx->stream));
    rc = acquire_png(ctx, ctx->stream);
    if (rc) return rc;
    if (bs::launch_png_encode(ctx->d_u8, width, height, ctx->d_png_scratch[0], ctx->d_png_file[0], d_bytes, ctx->stream,
                              static_cast<unsigned long long *>(ctx->d_scratch)))
        return fail(BS_EDEVICE, "PNG encoder launch failed");
    rc = release_png(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(clocks, ctx->d_scratch, nb * bs::kPngPhases * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

int bs_render_png(bs_ctx *ctx, const bs_config *cfg, double bloom_strength, int bloom_divider, unsigned char *out_png, size_t cap, size_t *out_bytes)
{
    if (!ctx || !cfg || !out_png || !out_bytes) return fail(BS_EINVAL, "null argument");
    if (int rc = check_png_frame(cfg->width, cfg->height)) return rc;
    auto t0 = std::chrono::steady_clock::now();
    if (cap < bs::png_file_bound(cfg->width, cfg->height)) return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
    if (int rc = check_bloom_args(cfg->width, bloom_strength, bloom_divider)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
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
}

int bs_debug_set_disk_slots(bs_ctx *ctx, int slots)
{
    if (!ctx || slots < 0 || slots > 4) return fail(BS_EINVAL, "bad slots");
    ctx->disk_slots = slots;
    return BS_OK;
}

int bs_render_device(bs_ctx *ctx, const bs_config *cfg, void *d_out_rgb, size_t out_doubles, void *hip_stream)
{
    return enqueue_render(ctx, cfg, static_cast<double *>(d_out_rgb), out_doubles, static_cast<hipStream_t>(hip_stream));
}

void *bs_host_alloc(bs_ctx *ctx, size_t bytes)
{
    if (!ctx || bytes == 0) { fail(BS_EINVAL, "null context or zero size"); return nullptr; }
    void *p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        fail(BS_ENOMEM, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void bs_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int bs_render_rows_device(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, void *d_out_rgb, size_t out_doubles, void *hip_stream)
{
    if (row1 < 0) return fail(BS_EINVAL, "row band must satisfy 0 <= row0 < row1 <= height");
    return enqueue_render(ctx, cfg, static_cast<double *>(d_out_rgb), out_doubles, static_cast<hipStream_t>(hip_stream), row0, row1);
}

int bs_render(bs_ctx *ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles)
{
    if (!cfg) return fail(BS_EINVAL, "null argument");
    return bs_render_rows(ctx, cfg, 0, cfg->height, out_rgb, out_doubles);
}

int bs_render_rows(bs_ctx *ctx, const bs_config *cfg, int row0, int row1, double *out_rgb, size_t out_doubles)
{
    if (!ctx || !cfg || !out_rgb) return fail(BS_EINVAL, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    if (row0 < 0 || row1 > cfg->height || row0 >= row1) return fail(BS_EINVAL, "row band must satisfy 0 <= row0 < row1 <= height");
    auto t0 = std::chrono::steady_clock::now();
    size_t need = (size_t)cfg->width * (size_t)(row1 - row0) * 3;
    if (out_doubles < need) return fail(BS_EINVAL, "output buffer too small");
    HIP_TRY(hipSetDevice(ctx->device));
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
        HIP_TRY(hipMemcpyAsync(out_rgb, ctx->d_img, need * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        for (int b = 0; b < nb; b++) {
            const int a = cut(b), e = cut(b + 1);
            HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_band[b], 0));
            HIP_TRY(hipMemcpyAsync(out_rgb + (size_t)(a - row0) * row_doubles, ctx->d_img + (size_t)(a - row0) * row_doubles,
                                   (size_t)(e - a) * row_doubles * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BS_OK;
}

// Frames first, first+step, ... on one context, double-buffered: while frame k's image is copied to the host (copy
// stream), frame k+1's kernel already runs (compute stream).  Pageable host buffers: the copy itself is the runtime's
// staged D2H (about 19 GB/s), but it no longer sits between two kernels.
static int render_frames_pipelined(bs_ctx *ctx, const bs_config *cfgs, double *const *outs, int first, int n_frames, int step)
{
    if (first >= n_frames) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t need = 0;
    for (int i = first; i < n_frames; i += step) {
        if (cfgs[i].width <= 0 || cfgs[i].height <= 0 || !outs[i]) return fail(BS_EINVAL, "bad frame");
        need = std::max(need, (size_t)cfgs[i].width * cfgs[i].height * 3);
    }
    {   // every frame's buffer page-locked: the kernels write them directly, two frames in flight on two streams, no copies
        std::vector<double *> alias;
        bool all = true;
        for (int i = first; i < n_frames; i += step) {  // (every buffer is looked at: one that straddles fails the call before any launch)
            bool straddles = false;
            double *a = device_alias_of_pinned(ctx, outs[i], (size_t)cfgs[i].width * cfgs[i].height * 3 * sizeof(double), &straddles);
            if (straddles) return fail(BS_EINVAL, kStraddleMsg);
            all = all && a != nullptr;
            alias.push_back(a);
        }
        if (all) {
            if (!ctx->stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            for (hipEvent_t &e : ctx->ev_frame)
                if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            hipStream_t cs[2] = {ctx->stream, ctx->stream2};
            StreamDrain drain(ctx);
            int k = 0;
            for (int i = first; i < n_frames; i += step, k++) {
                if (k >= 2) HIP_TRY(hipEventSynchronize(ctx->ev_frame[k & 1]));  // at most two frames in flight
                int rc = enqueue_render(ctx, &cfgs[i], alias[k], (size_t)cfgs[i].width * cfgs[i].height * 3, cs[k & 1], 0, -1, true, true, /*quiet=*/true);
                if (rc) return rc;
                HIP_TRY(hipEventRecord(ctx->ev_frame[k & 1], cs[k & 1]));
            }
            HIP_TRY(hipStreamSynchronize(cs[0]));
            HIP_TRY(hipStreamSynchronize(cs[1]));
            return BS_OK;
        }
    }
    auto grow = [&](double *&buf, size_t &cap) {
        if (cap >= need) return true;
        if (buf) (void)hipFree(buf);
        buf = nullptr;
        cap = 0;
        if (hipMalloc((void **)&buf, need * sizeof(double)) != hipSuccess) return false;
        cap = need;
        return true;
    };
    if (!grow(ctx->d_img, ctx->img_cap) || !grow(ctx->d_img2, ctx->img2_cap)) return fail(BS_ENOMEM, "hipMalloc image failed");
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (hipEvent_t &e : ctx->ev_frame)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // Frame k: image buf[k&1], compute stream cs[k&1], its own launch slot -- two frames can be in flight, and the persistent
    // wavefronts of frame k+1 take over the slots frame k's wavefronts leave as its tile queue runs dry (the end-of-frame
    // tail and the copy both disappear behind the neighbouring frame).
    double *buf[2] = {ctx->d_img, ctx->d_img2};
    hipStream_t cs[2] = {ctx->stream, ctx->stream2};
    // From here on work is in flight whose DMA targets are the caller's outs[]: every return path drains the streams first.
    StreamDrain drain(ctx);
    int k = 0;
    int rc = enqueue_render(ctx, &cfgs[first], buf[0], need, cs[0], 0, -1, true, true, /*quiet=*/true);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_frame[0], cs[0]));
    for (int i = first; i < n_frames; i += step, k++) {
        const int nxt = i + step;
        if (nxt < n_frames) {  // buf[(k+1)&1] is free: its previous copy was waited for before this point
            rc = enqueue_render(ctx, &cfgs[nxt], buf[(k + 1) & 1], need, cs[(k + 1) & 1], 0, -1, true, true, /*quiet=*/true);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ctx->ev_frame[(k + 1) & 1], cs[(k + 1) & 1]));
        }
        HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_frame[k & 1], 0));
        HIP_TRY(hipMemcpyAsync(outs[i], buf[k & 1], (size_t)cfgs[i].width * cfgs[i].height * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
        HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    }
    return BS_OK;
}

// every context of a batch / split call is driven by its own host thread: the same context twice would be driven by two
static int distinct_contexts(bs_ctx *const *ctxs, int n_ctx)
{
    for (int c = 0; c < n_ctx; c++) {
        if (!ctxs[c]) return fail(BS_EINVAL, "null context");
        for (int d = 0; d < c; d++)
            if (ctxs[d] == ctxs[c]) return fail(BS_EINVAL, "the same context appears twice (one context per device, each named once)");
    }
    return BS_OK;
}

int bs_render_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, double *const *outs)
{
    if (!ctxs || n_ctx <= 0 || (n_frames > 0 && (!cfgs || !outs))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    // One host thread per context (= per device); frame i goes to context i % n_ctx.  No data-path
    // collective: frames are independent (app/Main.hs:72-77 renders them one after another).
    std::vector<int> rcs(n_ctx, BS_OK);
    std::vector<std::string> errs(n_ctx);
    std::vector<std::thread> th;
    for (int c = 0; c < n_ctx; c++) {
        th.emplace_back([&, c]() {
            rcs[c] = render_frames_pipelined(ctxs[c], cfgs, outs, c, n_frames, n_ctx);
            if (rcs[c]) errs[c] = g_err;
        });
    }
    for (auto &t : th) t.join();
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
}

// Where a batch's frames go: RGB8 pixels (png == nullptr) or finished PNG files (bs_render_png_batch).
struct PngSink {
    const size_t *caps;   // capacity of outs[i]
    size_t *sizes;        // receives the size of file i
};

// What every frame of a context's share must satisfy before anything is launched; returns the largest frame (values) in *need.
static int check_rgb8_share(const bs_config *cfgs, const double *strengths, const int *dividers, unsigned char *const *outs, const PngSink *png,
                            int first, int n_frames, int step, size_t *need)
{
    *need = 0;
    for (int i = first; i < n_frames; i += step) {
        if (cfgs[i].width <= 0 || cfgs[i].height <= 0 || !outs[i]) return fail(BS_EINVAL, "bad frame");
        const double st = strengths ? strengths[i] : 0.0;
        if (st != 0 && !dividers) return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
        if (int rc = check_bloom_args(cfgs[i].width, st, st != 0 ? dividers[i] : 1)) return rc;
        if (png) {
            if (int rc = check_png_frame(cfgs[i].width, cfgs[i].height)) return rc;
            if (png->caps[i] < bs::png_file_bound(cfgs[i].width, cfgs[i].height))
                return fail(BS_EINVAL, "output buffer too small: bs_png_bound(width, height) bytes are required");
        }
        *need = std::max(*need, (size_t)cfgs[i].width * cfgs[i].height * 3);
    }
    return BS_OK;
}

// The PNG files of a batch in flight: frame `frame` was encoded into slot k (its scratch, its size word), into the caller's page-locked
// buffer or -- staged -- into the slot's device copy of the file.  retire() is called once everything enqueued for the slot has
// finished: it reports the size and, for a staged file, copies exactly its bytes (a copy of unknown length cannot be enqueued ahead).
struct PngSlots {
    int frame[bs_ctx::kPngSlots] = {-1, -1, -1};
    bool staged[bs_ctx::kPngSlots] = {false, false, false};

    int enqueue(bs_ctx *ctx, int k, int i, const unsigned char *d_u8, const bs_config &cfg, unsigned char *out, hipStream_t s)
    {
        bool straddles = false;
        double *alias = device_alias_of_pinned(ctx, out, (size_t)bs::png_file_bound(cfg.width, cfg.height), &straddles);
        if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        int rc = ensure_png(ctx, k, cfg.width, cfg.height, alias == nullptr);
        if (rc) return rc;
        uint64_t *d_bytes = png_bytes_slot(ctx, k);
        if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
        unsigned char *target = alias ? reinterpret_cast<unsigned char *>(alias) : ctx->d_png_file[k];
        if (bs::launch_png_encode(d_u8, cfg.width, cfg.height, ctx->d_png_scratch[k], target, d_bytes, s)) return fail(BS_EDEVICE, "PNG encoder launch failed");
        frame[k] = i;
        staged[k] = alias == nullptr;
        return BS_OK;
    }

    int retire(bs_ctx *ctx, int k, unsigned char *const *outs, const PngSink &png, hipStream_t s)
    {
        if (frame[k] < 0) return BS_OK;
        const size_t bytes = (size_t)ctx->h_png_bytes[k];
        png.sizes[frame[k]] = bytes;
        if (staged[k]) {
            HIP_TRY(hipMemcpyAsync(outs[frame[k]], ctx->d_png_file[k], bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        frame[k] = -1;
        return BS_OK;
    }
};

// doRender (app/Main.hs:105-123) for frames first, first+step, ... on one context, two frames in flight: frame k runs render ->
// bloom -> sRGB8 (-> PNG) on compute stream k & 1 with its own f64 image, so frame k+1's trace kernel fills the SIMDs frame k's last
// tiles leave (the fixed ~0.25 ms of a launch) and frame k's bloom runs on the CUs the next trace kernel frees
// first.  The blur scratch is one pair per context: the bloom of frame k+1 is ordered behind frame k's by acquire/release_post.
static int render_rgb8_frames_pipelined(bs_ctx *ctx, const bs_config *cfgs, const double *strengths, const int *dividers, unsigned char *const *outs,
                                        int first, int n_frames, int step, const PngSink *png)
{
    if (first >= n_frames) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t need = 0;
    int rc = check_rgb8_share(cfgs, strengths, dividers, outs, png, first, n_frames, step, &need);
    if (rc) return rc;
    if (!grow_device(ctx->d_img, ctx->img_cap, need) || !grow_device(ctx->d_img2, ctx->img2_cap, need) || !grow_device(ctx->d_u8, ctx->u8_cap, need) ||
        !grow_device(ctx->d_u8b, ctx->u8b_cap, need))
        return fail(BS_ENOMEM, "hipMalloc image failed");
    rc = ensure_post(ctx, need);
    if (rc) return rc;
    if (!ctx->stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (hipEvent_t &e : ctx->ev_frame)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double *img[2] = {ctx->d_img, ctx->d_img2};
    unsigned char *stage[2] = {ctx->d_u8, ctx->d_u8b};
    hipStream_t cs[2] = {ctx->stream, ctx->stream2};
    PngSlots files;
    StreamDrain drain(ctx);  // the caller's outs[] are DMA targets from here on: every return path drains the streams first
    int k = 0;
    for (int i = first; i < n_frames; i += step, k++) {
        const int b = k & 1;
        if (k >= 2) {
            HIP_TRY(hipEventSynchronize(ctx->ev_frame[b]));  // frame k-2 (same image, same staging) has left the device
            if (png && (rc = files.retire(ctx, b, outs, *png, cs[b]))) return rc;
        }
        const size_t n = (size_t)cfgs[i].width * cfgs[i].height * 3;
        unsigned char *target = stage[b];
        if (!png) {
            bool straddles = false;
            if (double *alias = device_alias_of_pinned(ctx, outs[i], n, &straddles)) target = reinterpret_cast<unsigned char *>(alias);  // page-locked: written in place
            if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        }
        rc = enqueue_render(ctx, &cfgs[i], img[b], n, cs[b], 0, -1, true, true, /*quiet=*/true);
        if (rc) return rc;
        rc = enqueue_post_rgb8(ctx, img[b], cfgs[i].width, cfgs[i].height, strengths ? strengths[i] : 0.0, dividers ? dividers[i] : 1, target, ctx->n_cu, cs[b]);
        if (rc) return rc;
        if (png) {
            rc = files.enqueue(ctx, b, i, stage[b], cfgs[i], outs[i], cs[b]);
            if (rc) return rc;
        } else if (target == stage[b]) {
            HIP_TRY(hipMemcpyAsync(outs[i], stage[b], n, hipMemcpyDeviceToHost, cs[b]));
        }
        HIP_TRY(hipEventRecord(ctx->ev_frame[b], cs[b]));
    }
    HIP_TRY(hipStreamSynchronize(cs[0]));
    HIP_TRY(hipStreamSynchronize(cs[1]));
    for (int b = 0; png && b < 2; b++)
        if ((rc = files.retire(ctx, b, outs, *png, cs[b]))) return rc;
    return BS_OK;
}

// How many CUs the post stage of a batch should own (0: none -- the shared-chip pipeline above).  With the chip partitioned a frame
// costs trace x n_cu / (n_cu - M), provided the post stage confined to M CUs keeps up; on the shared chip it costs trace + post + a
// hand-over stall (a blur workgroup only ever gets a CU in the drain of a later trace kernel, and delays the one behind it) that
// grows with the post stage: 0.33 ms at 1080p, 0.46 at 1440p, 0.7-0.95 at 4K -- about 1.4 x the post stage's own time.
// Measured (scripts/post_partition_ab.py, partition_large_ab.py, partition_more_ab.py -> profiles/r03_post_partition_ab.txt,
// r03_partition_large_ab.jsonl, r03_partition_more_ab.jsonl), ms per frame partitioned / shared: C3 1080p 4.27-4.38 / 4.67 (M = 8); 720p
// 2.00 / 2.25 (16; with 8 the post stage is the bottleneck: 2.26); bloomDivider 10 (r = 192) 4.47 / 4.72 (16; 8: 7.1); 1440p 7.80 /
// 8.04 (16); 3200x1800 12.15 / 12.45 (16); 3840x2160 17.39 / 17.83 (16; 8: 18.6); lensing-disk at 4K 19.23 / 19.90 (8: the longer
// trace hides the post stage on 8 CUs) and at 1440p 8.59 / 8.98 (8); C3 in STRICT 10.59 / 10.71 (8); frames without supersampling
// lose with any M (1.33 -> 2.3-4.0: too cheap to trace per pixel).  Both sides are ESTIMATED per frame -- trace: rays x straight-path
// steps / the measured FAST rate of 4.5e11 ray-steps per second and chip (STRICT: / 2.4); post: bs::estimate_post_us -- and the smallest
// M of {8, 16, 24} is taken for which, on EVERY frame of the share, (a) the post stage ALONE on M CUs needs at most 86 % of the trace
// time on the rest (next to the trace kernels it runs ~20 % slower than alone: 720p on 8 CUs 1.80 ms alone, 2.25 in the pipeline) and
// (b) the partitioned frame time undercuts the shared one by 0.5 % (a wrong call either way costs about 1 %).  BLACKSTAR_POST_CUS=0 | 8 | 16 | 24 | 32 overrides (A/B).
// png: the post stage also makes the frame's PNG file (bs_render_png_batch; bs::estimate_png_us).  Its small workgroups slip into
// the trace kernels' drains on the shared chip (about 45 % of their time shows: 4.73 against 4.65 ms), but confined to M CUs they count in
// full: the C3 frame then needs M = 16 -- 4.41 ms per frame against 4.73 shared and 7.4-8.1 with M = 8 (scripts/png_partition_ab.py).
static int post_cus_for_frame(const bs_config &cfg, double strength, int divider, int n_cu, bool fast, int m, bool png)
{
    bs::TraceParams p;
    std::memset(&p, 0, sizeof p);
    std::string err;
    if ((strength == 0 && !png) || !bs::derive_params(cfg, p, err)) return 0;
    // (without bloom the pixel map alone: about 30 us on the chip, 60 on a slice of it)
    double post_m = strength != 0 ? bs::estimate_post_us(cfg.width, cfg.height, divider, m) : 60.0;
    const double post_all = strength != 0 ? bs::estimate_post_us(cfg.width, cfg.height, divider, n_cu) : 30.0;
    if (post_m <= 0 || post_all <= 0) return 0;
    double shared_extra = post_all + std::max(330.0, 1.4 * post_all);   // the post stage and the stall it causes
    if (png) {
        post_m += bs::estimate_png_us(cfg.width, cfg.height, m);
        shared_extra += 0.45 * bs::estimate_png_us(cfg.width, cfg.height, n_cu);
    }
    const double steps = (p.rcam + std::sqrt(p.safe)) / p.h;  // the longest straight path through the scene, in steps
    const double rate = 4.5e11 * n_cu / 256.0 / (fast ? 1.0 : 2.4);
    const double trace_all = (double)p.wt * p.ht * steps / rate * 1e6;
    const double trace_m = trace_all * n_cu / (n_cu - m);
    if (trace_all < 1500.0) return 0;  // small frames (below ~720p supersampled): launch overheads dominate both stages; not measured, not partitioned
    return post_m <= 0.86 * trace_m && trace_m < 0.995 * (trace_all + shared_extra) ? m : 0;
}

static int choose_post_cus(bs_ctx *ctx, const bs_config *cfgs, const double *strengths, const int *dividers, int first, int n_frames, int step, bool png)
{
    if (ctx->post_cus_req == 0 || ctx->n_cu < 128 || ctx->n_cu % 8 != 0) return 0;
    if (ctx->post_cus_req > 0) return ctx->post_cus_req;
    if (first + 2 * step >= n_frames) return 0;  // fewer than three frames for this context: nothing to hide the post stage behind (1 frame: 5.38 against 5.22 ms)
    for (int m : {8, 16, 24}) {
        bool ok = true;
        for (int i = first; ok && i < n_frames; i += step) {
            const double st = strengths ? strengths[i] : 0.0;
            ok = (st == 0 || dividers) &&
                 post_cus_for_frame(cfgs[i], st, st != 0 ? dividers[i] : 1, ctx->n_cu, effective_mode(ctx, &cfgs[i]) == BS_MODE_FAST, m, png) == m;
        }
        if (ok) return m;
    }
    return 0;
}

// The CU-masked streams of a partition (post stage on bits [0, post_cus), trace kernels on the rest), made once per context and M.
// (hipExtStreamCreateWithCUMask takes no flags: unlike the context's other streams these synchronise with the NULL stream -- only a
// matter of overlap, and only if another thread of the caller keeps the NULL stream busy during a bs_render_rgb8_batch call.)
// false: the runtime would not make them (no CU-mask support on this device / driver) -- the caller falls back to the shared chip.
static bool ensure_partition(bs_ctx *ctx, int post_cus)
{
    if (post_cus < 8 || post_cus > 32 || post_cus % 4 != 0 || hipSetDevice(ctx->device) != hipSuccess) return false;
    bs_ctx::Partition &pt = ctx->parts[(post_cus - 8) / 4];
    if (pt.post) return true;
    const int words = (ctx->n_cu + 31) / 32;
    std::vector<uint32_t> post(words, 0u), trace(words, 0u);
    for (int b = 0; b < ctx->n_cu; b++) (b < post_cus ? post : trace)[b / 32] |= 1u << (b % 32);
    hipStream_t sp = nullptr, st0 = nullptr, st1 = nullptr;
    const bool ok = hipExtStreamCreateWithCUMask(&sp, (uint32_t)words, post.data()) == hipSuccess &&
                    hipExtStreamCreateWithCUMask(&st0, (uint32_t)words, trace.data()) == hipSuccess &&
                    hipExtStreamCreateWithCUMask(&st1, (uint32_t)words, trace.data()) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (hipStream_t s : {sp, st0, st1})
            if (s) (void)hipStreamDestroy(s);
        return false;
    }
    pt.post = sp; pt.trace[0] = st0; pt.trace[1] = st1;
    return true;
}

// The same with the chip PARTITIONED between the two stages (ctx->post_cus > 0).  A blur workgroup needs a whole CU (152 KiB of LDS,
// 8 wavefronts of 202 VGPRs) and the trace kernels' persistent workgroups hold every CU until their tile queue runs dry, so on shared
// streams the post stage of frame k only ever runs in the drain of a later trace kernel, and delays the one behind it (4.67 against
// 4.13 ms per frame without the post stage).  Here the trace kernels run on streams whose CU mask leaves post_cus CUs out and the post
// stage on a stream that owns exactly those: frame k's bloom + sRGB8 run WHILE frames k+1, k+2 are traced, at the price of post_cus /
// n_cu of the trace rate.  Mask bit i is CU i / 8 of XCD i % 8 (scripts/cumask_probe.py, profiles/r03_cumask_probe.txt: an XCD
// without a single bit gets ALL its CUs), so bits [0, post_cus) are post_cus / 8 CUs in every XCD (for 12, 20, 28: one more in the first
// four XCDs -- the trace kernels' tile queue and the blur sweeps' plans balance themselves).  Three images in flight.
static int render_rgb8_frames_partitioned(bs_ctx *ctx, int post_cus, const bs_config *cfgs, const double *strengths, const int *dividers,
                                          unsigned char *const *outs, int first, int n_frames, int step, const PngSink *png)
{
    if (first >= n_frames) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t need = 0;
    int rc = check_rgb8_share(cfgs, strengths, dividers, outs, png, first, n_frames, step, &need);
    if (rc) return rc;
    if (!grow_device(ctx->d_img, ctx->img_cap, need) || !grow_device(ctx->d_img2, ctx->img2_cap, need) || !grow_device(ctx->d_img3, ctx->img3_cap, need) ||
        !grow_device(ctx->d_u8, ctx->u8_cap, need) || !grow_device(ctx->d_u8b, ctx->u8b_cap, need) || !grow_device(ctx->d_u8c, ctx->u8c_cap, need))
        return fail(BS_ENOMEM, "hipMalloc image failed");
    rc = ensure_post(ctx, need);
    if (rc) return rc;
    bs_ctx::Partition &pt = ctx->parts[(post_cus - 8) / 4];  // streams made by ensure_partition
    for (hipEvent_t &e : ctx->ev_traced)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : ctx->ev_posted)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double *img[3] = {ctx->d_img, ctx->d_img2, ctx->d_img3};
    unsigned char *stage[3] = {ctx->d_u8, ctx->d_u8b, ctx->d_u8c};
    const int plan_cus = ctx->post_plan_cus > 0 ? ctx->post_plan_cus : post_cus;
    struct LaunchCus {  // the trace launches of this call size their persistent grids for the CUs their streams may use
        bs_ctx *c;
        LaunchCus(bs_ctx *c_, int n) : c(c_) { c->launch_cus = n; }
        ~LaunchCus() { c->launch_cus = 0; }
    } cus(ctx, ctx->n_cu - post_cus);
    PngSlots files;
    hipStream_t posted_on[3] = {pt.post, pt.post, pt.post};
    StreamDrain drain(ctx);  // the caller's outs[] are DMA targets from here on: every return path drains the streams first
    int k = 0;
    for (int i = first; i < n_frames; i += step, k++) {
        const int b = k % 3;
        hipStream_t ts = pt.trace[k & 1];
        if (k >= 3) {
            HIP_TRY(hipEventSynchronize(ctx->ev_posted[b]));  // frame k-3 (same image, same staging) has left the device
            if (png && (rc = files.retire(ctx, b, outs, *png, posted_on[b]))) return rc;
        }
        const size_t n = (size_t)cfgs[i].width * cfgs[i].height * 3;
        unsigned char *target = stage[b];
        if (!png) {
            bool straddles = false;
            if (double *alias = device_alias_of_pinned(ctx, outs[i], n, &straddles)) target = reinterpret_cast<unsigned char *>(alias);
            if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        }
        rc = enqueue_render(ctx, &cfgs[i], img[b], n, ts, 0, -1, true, true, /*quiet=*/true);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ctx->ev_traced[b], ts));
        // The LAST frame of the share has no tracing left to hide behind: its post stage takes the whole chip (ctx->stream, sweeps
        // planned for all CUs: 0.2 instead of 3.8 ms for a 1080p frame -- on a 20-frame batch that tail alone was 0.19 ms per frame).
        const bool last = i + step >= n_frames;
        hipStream_t ps = last ? ctx->stream : pt.post;
        posted_on[b] = ps;
        HIP_TRY(hipStreamWaitEvent(ps, ctx->ev_traced[b], 0));
        rc = enqueue_post_rgb8(ctx, img[b], cfgs[i].width, cfgs[i].height, strengths ? strengths[i] : 0.0, dividers ? dividers[i] : 1, target,
                               last ? ctx->n_cu : plan_cus, ps);
        if (rc) return rc;
        if (png) {
            rc = files.enqueue(ctx, b, i, stage[b], cfgs[i], outs[i], ps);
            if (rc) return rc;
        } else if (target == stage[b]) {
            HIP_TRY(hipMemcpyAsync(outs[i], stage[b], n, hipMemcpyDeviceToHost, ps));
        }
        HIP_TRY(hipEventRecord(ctx->ev_posted[b], ps));
    }
    HIP_TRY(hipStreamSynchronize(pt.trace[0]));
    HIP_TRY(hipStreamSynchronize(pt.trace[1]));
    HIP_TRY(hipStreamSynchronize(pt.post));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int b = 0; png && b < 3; b++)
        if ((rc = files.retire(ctx, b, outs, *png, posted_on[b]))) return rc;
    return BS_OK;
}

int bs_debug_last_post_cus(const bs_ctx *ctx) { return ctx ? ctx->last_post_cus : BS_EINVAL; }

int bs_debug_post_cus(const bs_config *cfg, double bloom_strength, int bloom_divider, int n_cu, int mode)
{
    if (!cfg || n_cu < 1) return fail(BS_EINVAL, "bad argument");
    if (n_cu < 128 || n_cu % 8 != 0) return 0;
    const bool png = (mode & BS_DEBUG_POST_CUS_PNG) != 0;
    mode &= ~BS_DEBUG_POST_CUS_PNG;
    const bool fast = mode == BS_MODE_FAST && cfg->step_size <= 0.5;
    for (int m : {8, 16, 24})
        if (post_cus_for_frame(*cfg, bloom_strength, bloom_divider, n_cu, fast, m, png) == m) return m;
    return 0;
}

// bs_render_rgb8_batch / bs_render_png_batch: one host thread per context, frame i on context i % n_ctx.
static int render_post_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                             unsigned char *const *outs, const PngSink *png)
{
    if (!ctxs || n_ctx <= 0 || (n_frames > 0 && (!cfgs || !outs))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    std::vector<int> rcs(n_ctx, BS_OK);
    std::vector<std::string> errs(n_ctx);
    std::vector<std::thread> th;
    for (int c = 0; c < n_ctx; c++) {
        th.emplace_back([&, c]() {
            bs_ctx *x = ctxs[c];
            int post_cus = choose_post_cus(x, cfgs, bloom_strengths, bloom_dividers, c, n_frames, n_ctx, png != nullptr);
            // Only with page-locked outputs, which the last kernel of a frame writes itself: a copy into PAGEABLE memory blocks the
            // host thread until the frame's post stage has finished -- 3.8 ms on 8 CUs instead of 0.2 ms on the whole chip -- and the
            // next trace kernel is not enqueued meanwhile (measured 9.1 against 4.8 ms per frame: scripts/post_partition_pageable_ab.py)
            for (int i = c; post_cus && i < n_frames; i += n_ctx) {
                if (!outs[i] || cfgs[i].width <= 0 || cfgs[i].height <= 0 || hipSetDevice(x->device) != hipSuccess ||
                    !device_alias_of_pinned(x, outs[i], png ? (size_t)bs::png_file_bound(cfgs[i].width, cfgs[i].height) : (size_t)cfgs[i].width * cfgs[i].height * 3))
                    post_cus = 0;
            }
            if (post_cus && !ensure_partition(x, post_cus)) post_cus = 0;
            x->last_post_cus = post_cus;
            rcs[c] = post_cus ? render_rgb8_frames_partitioned(x, post_cus, cfgs, bloom_strengths, bloom_dividers, outs, c, n_frames, n_ctx, png)
                              : render_rgb8_frames_pipelined(x, cfgs, bloom_strengths, bloom_dividers, outs, c, n_frames, n_ctx, png);
            if (rcs[c]) errs[c] = g_err;
        });
    }
    for (auto &t : th) t.join();
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
}

int bs_render_rgb8_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                         unsigned char *const *outs)
{
    return render_post_batch(ctxs, n_ctx, cfgs, n_frames, bloom_strengths, bloom_dividers, outs, nullptr);
}

int bs_render_png_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                        unsigned char *const *outs, const size_t *caps, size_t *out_bytes)
{
    if (n_frames > 0 && (!caps || !out_bytes)) return fail(BS_EINVAL, "null argument");
    const PngSink sink{caps, out_bytes};
    return render_post_batch(ctxs, n_ctx, cfgs, n_frames, bloom_strengths, bloom_dividers, outs, &sink);
}

// One file: create / truncate, write, close.  Empty string on success, else what failed.
static std::string write_whole_file(const char *path, const unsigned char *data, size_t n)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return std::string(path) + ": " + std::strerror(errno);
    const size_t wrote = n ? std::fwrite(data, 1, n, f) : 0;
    const int close_rc = std::fclose(f);
    if (wrote != n || close_rc != 0) return std::string(path) + ": " + std::strerror(errno ? errno : EIO);
    return std::string();
}

int bs_render_png_files(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                        const char *const *paths, int pipe)
{
    if (!ctxs || n_ctx <= 0 || n_frames < 0 || (n_frames > 0 && (!cfgs || !paths))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    size_t cap = 0;
    for (int i = 0; i < n_frames; i++) {
        if (!paths[i]) return fail(BS_EINVAL, "null path");
        if (int rc = check_png_frame(cfgs[i].width, cfgs[i].height)) return rc;
        cap = std::max(cap, (size_t)bs::png_file_bound(cfgs[i].width, cfgs[i].height));
    }
    if (n_frames == 0) return BS_OK;
    // `chunk` frames per bs_render_png_batch call into one of two sets of page-locked file buffers; a writer thread writes the set of the
    // call before while the GPUs fill the other.  (The second set exists only if there is a second call.)
    const int chunk = (pipe > 0 ? pipe : 16) * n_ctx;
    const int slots = std::min(chunk, n_frames);
    const int n_sets = n_frames > chunk ? 2 : 1;
    // buffer k of the call lives in the pool of context k % n_ctx (entry k / n_ctx), kept for the next call and freed with the context
    std::vector<unsigned char *> bufs((size_t)n_sets * slots, nullptr);
    for (size_t k = 0; k < bufs.size(); k++) {
        bs_ctx *owner = ctxs[k % n_ctx];
        const size_t e = k / n_ctx;
        if (owner->file_pool.size() <= e) owner->file_pool.resize(e + 1, {nullptr, 0});
        auto &slot = owner->file_pool[e];
        if (slot.second < cap) {
            if (slot.first) bs_host_free(slot.first);
            slot = {nullptr, 0};
            slot.first = static_cast<unsigned char *>(bs_host_alloc(owner, cap));
            if (!slot.first) return BS_ENOMEM;   // (bs_host_alloc has set the message)
            slot.second = cap;
        }
        bufs[k] = slot.first;
    }
    std::vector<size_t> caps(slots, cap), sizes((size_t)n_sets * slots, 0);
    std::thread writer;
    std::string write_error;   // owned by the writer thread until it is joined
    struct JoinWriter {
        std::thread &t;
        ~JoinWriter() { if (t.joinable()) t.join(); }
    } join_writer{writer};
    int rc = BS_OK;
    for (int pos = 0, it = 0; pos < n_frames && rc == BS_OK; pos += chunk, it++) {
        const int count = std::min(chunk, n_frames - pos), set = it & 1;
        // (the set being refilled was written out by the writer of the call before the last, joined below one iteration ago;
        //  the writer of the last call -- the other set -- may still be running: that is the overlap)
        unsigned char *const *outs = bufs.data() + (size_t)set * slots;
        size_t *sz = sizes.data() + (size_t)set * slots;
        const PngSink sink{caps.data(), sz};
        rc = render_post_batch(ctxs, n_ctx, cfgs + pos, count, bloom_strengths ? bloom_strengths + pos : nullptr, bloom_dividers ? bloom_dividers + pos : nullptr, outs, &sink);
        if (rc) break;
        if (writer.joinable()) {              // the call before this one: its files (the other set) must be out before a new writer starts
            writer.join();
            if (!write_error.empty()) return fail(BS_EIO, write_error);
        }
        writer = std::thread([&write_error, outs, sz, paths, pos, count]() {
            for (int j = 0; j < count && write_error.empty(); j++) write_error = write_whole_file(paths[pos + j], outs[j], sz[j]);
        });
    }
    if (writer.joinable()) writer.join();
    if (rc) return rc;   // (the failing call has set the message)
    if (!write_error.empty()) return fail(BS_EIO, write_error);
    return BS_OK;
}

int bs_render_split(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles)
{
    if (!ctxs || n_ctx <= 0 || !cfg || !out_rgb) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    if (cfg->width <= 0 || cfg->height <= 0) return fail(BS_EINVAL, "resolution must be positive");
    if (out_doubles < (size_t)cfg->width * cfg->height * 3) return fail(BS_EINVAL, "output buffer too small");
    // Context c renders the c-th of n contiguous row bands (sizes differ by at most one row; contexts beyond the number
    // of rows stay idle), one host thread per context, each copying its band straight into its place in out_rgb.
    const int n = std::min(n_ctx, cfg->height);
    const int base = cfg->height / n, extra = cfg->height % n;
    std::vector<int> rcs(n, BS_OK);
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (int c = 0; c < n; c++) {
        const int row0 = c * base + std::min(c, extra), row1 = row0 + base + (c < extra ? 1 : 0);
        th.emplace_back([&, c, row0, row1]() {
            rcs[c] = bs_render_rows(ctxs[c], cfg, row0, row1, out_rgb + (size_t)row0 * cfg->width * 3, (size_t)(row1 - row0) * cfg->width * 3);
            if (rcs[c]) errs[c] = g_err;
        });
    }
    for (auto &t : th) t.join();
    for (int c = 0; c < n; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
}

int bs_trace_rays(bs_ctx *ctx, const bs_config *cfg, const int32_t *yx, size_t n_rays, bs_ray_record *out)
{
    if (!ctx || !cfg || (n_rays && (!yx || !out))) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    int rc = fill_params(ctx, cfg, p);
    if (rc) return rc;
    if (n_rays == 0) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t yx_bytes = (n_rays * 2 * sizeof(int32_t) + 255) & ~size_t(255);
    rc = ensure_scratch(ctx, yx_bytes + n_rays * sizeof(bs_ray_record));
    if (rc) return rc;
    int32_t *d_yx = static_cast<int32_t *>(ctx->d_scratch);
    bs_ray_record *d_out = reinterpret_cast<bs_ray_record *>(static_cast<char *>(ctx->d_scratch) + yx_bytes);
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(d_yx, yx, n_rays * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (bs::launch_trace_records(p, effective_mode(ctx, cfg), d_yx, n_rays, d_out, ctx->stream)) return fail(BS_EDEVICE, "kernel launch failed");
    HIP_TRY(hipMemcpyAsync(out, d_out, n_rays * sizeof(bs_ray_record), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

int bs_stats(bs_ctx *ctx, bs_stats_t *out)
{
    if (!ctx || !out) return fail(BS_EINVAL, "null argument");
    int rc = resolve_stats(ctx);
    if (rc) return rc;
    ctx->stats.wall_ms = ctx->last_wall_ms;
    ctx->stats.zero_copy = ctx->last_zero_copy;
    *out = ctx->stats;
    return BS_OK;
}

int bs_star_lookup(bs_ctx *ctx, double intensity, double saturation, const double *dirs, size_t n, double *out_rgb, int32_t *out_hits)
{
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
    HIP_TRY(hipSetDevice(ctx->device));
    // persistent scratch (grown on demand, kept for the life of the context): [dirs 3n | rgb 3n] doubles, then n hit counts
    int rc = ensure_scratch(ctx, 6 * n * sizeof(double) + n * sizeof(int32_t));
    if (rc) return rc;
    double *d = static_cast<double *>(ctx->d_scratch);
    int32_t *dh = reinterpret_cast<int32_t *>(d + 6 * n);
    StreamDrain drain(ctx);
    HIP_TRY(hipMemcpyAsync(d, dirs, 3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (bs::launch_star_lookup(p, d, n, d + 3 * n, dh, ctx->stream)) return fail(BS_EDEVICE, "kernel launch failed");
    HIP_TRY(hipMemcpyAsync(out_rgb, d + 3 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (out_hits) HIP_TRY(hipMemcpyAsync(out_hits, dh, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}

int bs_debug_ubench(bs_ctx *ctx, int kind, int blocks, int iters, double *out_ms, double *out_ginstr)
{
    if (!ctx || !out_ms || blocks <= 0 || iters <= 0) return fail(BS_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
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
}

int bs_debug_sqrt_div(bs_ctx *ctx, const double *a, const double *b, size_t n, double *out_sqrt, double *out_div, int bare)
{
    if (!ctx || (n && (!a || !b || !out_sqrt || !out_div))) return fail(BS_EINVAL, "null argument");
    if (n == 0) return BS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 4 * n * sizeof(double)));
    hipError_t e = hipMemcpy(d, a, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && bs::launch_sqrt_div(d, d + n, n, d + 2 * n, d + 3 * n, bare, ctx->stream)) e = hipErrorLaunchFailure;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out_sqrt, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_div, d + 3 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(BS_EDEVICE, std::string("bs_debug_sqrt_div: ") + hipGetErrorString(e));
    return BS_OK;
}

}  // extern "C"
