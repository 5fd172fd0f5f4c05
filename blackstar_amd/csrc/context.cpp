// context.cpp -- the context behind the C ABI of include/blackstar_gpu.h: lifetime, settings, the thread-local error message, page-locked
// caller buffers (zero copy) and device scratch.  No CPU rendering path exists in this library: without a HIP device every render
// entry point fails with BS_EDEVICE.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include <memory>

#include "bs_context.h"

namespace bs {

namespace {
thread_local std::string g_err;
}

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

const std::string &error_message() { return g_err; }

int abi_exception(const char *where) noexcept
{
    int code = BS_EINTERNAL;
    try {
        try {
            throw;   // (the exception the caller's catch (...) is handling)
        } catch (const std::bad_alloc &) {
            code = BS_ENOMEM;
            g_err = std::string(where) + ": out of host memory";
        } catch (const std::exception &e) {
            g_err = std::string(where) + ": unexpected C++ exception: " + e.what();
        } catch (...) {
            g_err = std::string(where) + ": unexpected C++ exception";
        }
    } catch (...) {  // building the message failed too: the code alone has to do
    }
    return code;
}

size_t ctx_layout_bytes() { return sizeof(bs_ctx); }

// Waits for everything the context has enqueued anywhere: its own streams, then every event it recorded behind work on a caller's stream.
// (hipEventSynchronize on an event that was never recorded returns at once.)  The device is set by the caller.
static void quiesce(bs_ctx *ctx)
{
    { StreamDrain own(ctx); }
    for (bs_ctx::LaunchSlot &sl : ctx->slots)
        if (sl.used && sl.ev_done) (void)hipEventSynchronize(sl.ev_done);
    for (hipEvent_t e : {ctx->ev_post, ctx->ev_png, ctx->ev_frame[0], ctx->ev_frame[1], ctx->ev_stage[0], ctx->ev_stage[1]})
        if (e) (void)hipEventSynchronize(e);
    for (hipEvent_t e : ctx->ev_traced)
        if (e) (void)hipEventSynchronize(e);
    for (hipEvent_t e : ctx->ev_posted)
        if (e) (void)hipEventSynchronize(e);
    for (const bs_ctx::Foreign &f : ctx->foreign)
        if (f.used && f.ev) (void)hipEventSynchronize(f.ev);
}

ForeignWork::~ForeignWork()
{
    if (!ctx) return;
    const OnDevice on_device(ctx->device);
    if (!on_device.ok()) return;
    for (hipStream_t own : {ctx->stream, ctx->stream2, ctx->copy_stream})
        if (own && own == s) return;
    for (const bs_ctx::Partition &pt : ctx->parts)
        for (hipStream_t own : {pt.trace[0], pt.trace[1], pt.post})
            if (own && own == s) return;
    bs_ctx::Foreign *f = nullptr;
    for (bs_ctx::Foreign &g : ctx->foreign)
        if (g.used && g.s == s) f = &g;
    if (!f) {   // a stream not seen lately: the oldest entry makes room -- after what it stands for has finished (nothing may get lost)
        f = &ctx->foreign[ctx->foreign_next];
        ctx->foreign_next = (ctx->foreign_next + 1) % (int)(sizeof ctx->foreign / sizeof ctx->foreign[0]);
        if (f->used && f->ev) (void)hipEventSynchronize(f->ev);
        f->used = false;
    }
    if (!f->ev && hipEventCreateWithFlags(&f->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);   // no event to be had: wait here rather than leave work bs_destroy cannot see
        return;
    }
    if (hipEventRecord(f->ev, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        return;
    }
    f->s = s;
    f->used = true;
}

StreamDrain::~StreamDrain()
{
    if (!ctx) return;
    const OnDevice on_device(ctx->device);
    if (!on_device.ok()) return;
    for (hipStream_t s : {ctx->stream, ctx->stream2, ctx->copy_stream})
        if (s) (void)hipStreamSynchronize(s);
    for (const bs_ctx::Partition &pt : ctx->parts)
        for (hipStream_t s : {pt.trace[0], pt.trace[1], pt.post})
            if (s) (void)hipStreamSynchronize(s);
}

// Zero copy: if a caller's HOST buffer is page-locked (bs_host_alloc, hipHostMalloc, hipHostRegister) the device can write it
// directly over PCIe, so the kernel's own image stores deliver the frame -- no device image, no copy, and the transfer is
// spread over the whole kernel instead of trailing it.  Measured (scripts/zero_copy_probe.py, profiles/r02_zero_copy.txt):
// bs_render of the C3 frame 5.45 -> 4.57 ms (kernel 4.38), C2 2.14 -> 1.49 ms (49.8 MB in 1.47 ms = 34 GB/s while tracing), C4
// 21.6 -> 18.7 ms; the kernel time itself does not change (11 GB/s average is far below what PCIe takes in 96-B segments).
// Returns the device alias of `host`, or nullptr for pageable memory (which takes the staged path).  BLACKSTAR_ZERO_COPY=0: off.
// *straddles (optional): set when `host` STARTS in page-locked memory but [host, host + bytes) is not contained in it -- a buffer
// no path can deliver into: the kernel's stores would fault, and the runtime's own hipMemcpyAsync refuses it ("invalid argument":
// it finds the registered range the pointer starts in and the size does not fit).  Callers turn that into BS_EINVAL.
double *device_alias_of_pinned(bs_ctx *ctx, const void *host, size_t bytes, bool *straddles)
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
        // (2) no range to be had: walk the buffer page by page (every page page-locked, the device alias contiguous).  The walk costs a
        //     runtime query per 4 KiB page, so the last few (pointer, size) pairs that passed are remembered -- callers reuse their frame
        //     buffers -- and a remembered pair is only trusted after its LAST page has been looked at again: memory that was unregistered
        //     and re-registered shorter at the same address (the entry would otherwise vouch for pages that are pageable now) fails there.
        const char *dp = static_cast<const char *>(a.devicePointer);
        const uintptr_t page = 4096;
        auto page_ok = [&](uintptr_t q) {
            hipPointerAttribute_t b;
            if (hipPointerGetAttributes(&b, reinterpret_cast<const void *>(q)) == hipSuccess && b.type == hipMemoryTypeHost &&
                static_cast<const char *>(b.devicePointer) - reinterpret_cast<const char *>(q) == dp - hp)
                return true;
            (void)hipGetLastError();
            return false;
        };
        const uintptr_t last = (reinterpret_cast<uintptr_t>(hp) + bytes - 1) & ~(page - 1);
        for (auto &v : ctx->verified) {
            if (v.host != host || v.bytes != bytes) continue;
            if (last <= reinterpret_cast<uintptr_t>(hp) || page_ok(last)) covered = true;
            else v = {};  // stale: forget it, walk again below
        }
        if (!covered) {
            covered = true;
            for (uintptr_t q = (reinterpret_cast<uintptr_t>(hp) & ~(page - 1)) + page; covered && q < reinterpret_cast<uintptr_t>(hp) + bytes; q += page)
                covered = page_ok(q);
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

const char *const kStraddleMsg = "output buffer starts in page-locked memory but is not contained in it (it runs past the end of its hipHostMalloc / "
                                 "hipHostRegister range, e.g. into pageable memory between two registered ranges): neither the kernel nor the runtime's copy can deliver into it";

namespace {

constexpr size_t kDirectCopyBytes = size_t(1) << 20;   // up to here the runtime stages a pageable copy itself (measured: "HSA Copy Using Staging resource")

// Is [h, h + bytes) inside ONE page-locked range from end to end?  The same containment test device_alias_of_pinned makes, with the driver's
// range attributes: probing the two ends is not enough (a buffer that starts in one hipHostRegister range and ends in another, with pageable
// memory between them, passes that), and the runtime would then either refuse the copy or pin the pageable middle on the fly -- the very
// path the staging exists to avoid.  Anything that cannot be PROVED contained is staged.
bool inside_one_page_locked_range(const void *h, size_t bytes)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, h) != hipSuccess || a.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    void *base = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, const_cast<void *>(h)) != hipSuccess || !base ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, const_cast<void *>(h)) != hipSuccess || !size) {
        (void)hipGetLastError();
        return false;
    }
    const char *hb = static_cast<const char *>(base), *hp = static_cast<const char *>(h);
    return hp >= hb && bytes <= size && static_cast<size_t>(hp - hb) <= size - bytes;
}

// may the runtime have this caller buffer as it is?
bool direct_ok(const void *h, size_t bytes)
{
    if (bytes <= kDirectCopyBytes) return true;
    return inside_one_page_locked_range(h, bytes);
}

int ensure_stage(bs_ctx *ctx)
{
    for (int b = 0; b < 2; b++) {
        if (!ctx->h_stage[b] && hipHostMalloc((void **)&ctx->h_stage[b], bs_ctx::kStageBytes, hipHostMallocDefault) != hipSuccess)
            return fail(BS_ENOMEM, "hipHostMalloc staging failed");
        if (!ctx->ev_stage[b]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_stage[b], hipEventDisableTiming));
    }
    return BS_OK;
}

int wait_stage(bs_ctx *ctx, int b)
{
    if (ctx->stage_busy[b]) {
        HIP_TRY(hipEventSynchronize(ctx->ev_stage[b]));
        ctx->stage_busy[b] = false;
    }
    return BS_OK;
}

}  // namespace

int copy_in(bs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return BS_OK;
    if (direct_ok(h_src, bytes)) {
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
        return BS_OK;
    }
    if (int rc = ensure_stage(ctx)) return rc;
    size_t off = 0;
    for (int k = 0; off < bytes; k++) {
        const int b = k & 1;
        const size_t len = std::min(bs_ctx::kStageBytes, bytes - off);
        if (int rc = wait_stage(ctx, b)) return rc;   // the DMA that last read this piece has finished
        std::memcpy(ctx->h_stage[b], static_cast<const char *>(h_src) + off, len);
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_dst) + off, ctx->h_stage[b], len, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(ctx->ev_stage[b], s));
        ctx->stage_busy[b] = true;
        off += len;
    }
    return BS_OK;
}

int copy_out(bs_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) {
        HIP_TRY(hipStreamSynchronize(s));
        return BS_OK;
    }
    if (direct_ok(h_dst, bytes)) {
        HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return BS_OK;
    }
    if (int rc = ensure_stage(ctx)) return rc;
    // piece k lands in staging piece k & 1 while piece k - 1 is copied out of the other one by the host
    size_t off = 0, prev_off = 0, prev_len = 0;
    int prev_b = -1;
    for (int k = 0; off < bytes; k++) {
        const int b = k & 1;
        const size_t len = std::min(bs_ctx::kStageBytes, bytes - off);
        if (int rc = wait_stage(ctx, b)) return rc;   // (a copy_in's last DMA may still be reading it)
        HIP_TRY(hipMemcpyAsync(ctx->h_stage[b], static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(ctx->ev_stage[b], s));
        ctx->stage_busy[b] = true;
        if (prev_b >= 0) {
            if (int rc = wait_stage(ctx, prev_b)) return rc;
            std::memcpy(static_cast<char *>(h_dst) + prev_off, ctx->h_stage[prev_b], prev_len);
        }
        prev_b = b; prev_off = off; prev_len = len;
        off += len;
    }
    if (int rc = wait_stage(ctx, prev_b)) return rc;
    std::memcpy(static_cast<char *>(h_dst) + prev_off, ctx->h_stage[prev_b], prev_len);
    return BS_OK;
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
// The arithmetic a frame is traced with.  FAST's error model (rounding differences of ~1 ulp per right-hand side, amplified by
// the discrete map) needs two things of the frame, and a FAST context traces frames that lack either in STRICT -- at STRICT's cost, 2.4x
// per step -- so that every frame stays inside the 1e-4 parity bar with a measured margin (BLACKSTAR_FAST_GUARD=0: both rules off, A/B):
//  (1) a step that resolves the field: with stepSize above 0.5 Schwarzschild radii a single step past the hole amplifies a perturbation
//      by >10x and FAST and STRICT trajectories part ways (terminal directions 4e-3 apart at stepSize 1.0, all of the round-2 fuzz's
//      largest colour deviations).  The reference's default is 0.3 and every scene file it ships uses that.
//  (2) a path of bounded length (round 6): the difference between FAST's and STRICT's terminal direction grows with the number of steps,
//      about 1e-9 relative colour difference per expected step on a clustered sky (the star PSF exp(-d^2 / 2w^2) turns a direction
//      difference e into up to 6000 e): over 3 560 random long-path scenes against the ORACLE (scripts/fuzz_longpath.py,
//      profiles/r06_fuzz_oracle_longpath.json) the worst value is 1.8e-6 below N0 = 2 000 expected steps per ray, 1.2e-5 below 10 000,
//      3.6e-5 below 30 000 and 2.2e-4 -- OUTSIDE the bar, 3 values -- between 30 000 and 100 000.  N0 = (|camera| + sqrt safeDistance) /
//      stepSize, the longest straight path through the traced volume; above fast_max_steps (BS_FAST_MAX_EXPECTED_STEPS = 2 000: 57x
//      inside the bar on that sample; 26x over the 2 x 100 000-scene fuzz of the library with the rule, whose worst case is now a
//      stepSize-0.5 frame) the frame is traced in STRICT.  The scenes the reference ships have N0 = 233 .. 523.
double expected_steps(const bs_config *cfg)
{
    const double px = cfg->cam_pos[0], py = cfg->cam_pos[1], pz = cfg->cam_pos[2];
    const double r2 = (px * px + py * py) + pz * pz;
    const double twice = 2.0 * r2;
    const double safe = (2500.0 <= twice) ? twice : 2500.0;   // src/Raytracer.hs:59-60
    return (std::sqrt(r2) + std::sqrt(safe)) / cfg->step_size;
}

int effective_mode(const bs_ctx *ctx, const bs_config *cfg)
{
    if (ctx->mode == BS_MODE_FAST && ctx->fast_guard) {
        if (!(cfg->step_size <= 0.5)) return BS_MODE_STRICT;
        if (ctx->fast_max_steps > 0 && !(expected_steps(cfg) <= ctx->fast_max_steps)) return BS_MODE_STRICT;
    }
    return ctx->mode;
}
}  // namespace bs

using bs::fail;

extern "C" {

int bs_abi_version(void) { return BS_ABI_VERSION; }

const char *bs_last_error(void) { return bs::error_message().c_str(); }

bs_ctx *bs_create(int device, const bs_star *stars, size_t n_stars)
try {
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
    // Owned until the very end: anything below that throws (std::bad_alloc in build_star_index, in a vector or a message string) lands in
    // the function's catch with the half-made context -- its streams, events and device memory -- destroyed on the way out.
    std::unique_ptr<bs_ctx, void (*)(bs_ctx *)> owner(new (std::nothrow) bs_ctx(), bs_destroy);
    bs_ctx *ctx = owner.get();
    if (!ctx) { fail(BS_ENOMEM, "out of host memory"); return nullptr; }
    ctx->device = device;
    ctx->n_stars = n_stars;
    bs::probe_host_topology(ctx);
    if (const char *m = std::getenv("BLACKSTAR_STAGGER")) ctx->stagger_cycles = std::atoi(m);
    if (const char *m = std::getenv("BLACKSTAR_STAGGER_MIN_TILES")) ctx->stagger_min_tiles = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_STATIC_FIRST_BELOW")) ctx->static_first_below = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_LATE_POP_SLOT")) ctx->late_pop_slot = std::max(0, std::atoi(m));
    if (const char *m = std::getenv("BLACKSTAR_FAST_GUARD")) ctx->fast_guard = std::atoi(m) != 0;
    if (const char *m = std::getenv("BLACKSTAR_FAST_MAX_STEPS")) ctx->fast_max_steps = std::max(0.0, std::atof(m));   // (0: no long-path rule; for measurements)
    if (const char *m = std::getenv("BLACKSTAR_ZERO_COPY")) ctx->zero_copy = std::atoi(m) != 0;
    if (const char *m = std::getenv("BLACKSTAR_HOST_BANDS")) ctx->host_bands = std::max(1, std::min((int)bs_ctx::kMaxHostBands, std::atoi(m)));
    if (const char *m = std::getenv("BLACKSTAR_BLOCKS_PER_CU")) ctx->blocks_per_cu = std::max(1, std::min(8, std::atoi(m)));
    if (const char *m = std::getenv("BLACKSTAR_POST_CUS")) ctx->post_cus_req = std::strcmp(m, "auto") ? bs::post_cus_setting(std::atoi(m)) : -1;
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
    auto up = [&](void *d_dst, const void *h_src, size_t bytes, const char *what) {   // (after the stream exists: the && chain below makes it first)
        if (bs::copy_in(ctx, d_dst, h_src, bytes, ctx->stream) == BS_OK) return true;
        fail(BS_EDEVICE, std::string(what) + ": " + bs::error_message());
        return false;
    };
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    const bs::OnDevice on_device(device);   // (declared after `owner`: the caller's device comes back before a failed context is destroyed, bs_destroy sets its own)
    bool good = ok(on_device.err, "hipSetDevice") &&
                ok(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate") &&
                ok(hipEventCreate(&ctx->ev_u0), "hipEventCreate") && ok(hipEventCreate(&ctx->ev_u1), "hipEventCreate") &&
                ok(hipMalloc((void **)&ctx->d_nodes, std::max<size_t>(1, nodes.size()) * sizeof(bs::StarNode)), "hipMalloc nodes") &&
                ok(hipMalloc((void **)&ctx->d_colors, std::max<size_t>(1, colors.size()) * sizeof(bs::StarColor)), "hipMalloc colors") &&
                ok(hipMalloc((void **)&ctx->d_cell_start, cell_start.size() * sizeof(uint32_t)), "hipMalloc cell_start") &&
                up(ctx->d_cell_start, cell_start.data(), cell_start.size() * sizeof(uint32_t), "upload cell_start") &&
                ok(hipMalloc((void **)&ctx->d_counters, bs_ctx::kSlots * bs::kCounters * sizeof(unsigned long long)), "hipMalloc counters") &&
                ok(hipHostMalloc((void **)&ctx->h_counters, bs_ctx::kSlots * bs::kCounters * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc") &&
                up(ctx->d_nodes, nodes.data(), nodes.size() * sizeof(bs::StarNode), "upload nodes") &&
                up(ctx->d_colors, colors.data(), colors.size() * sizeof(bs::StarColor), "upload colors");
    if (good) {
        static const std::vector<double> table = [] { std::vector<double> t(257); bs::srgb8_thresholds(t.data()); return t; }();
        good = ok(hipMalloc((void **)&ctx->d_srgb_table, 257 * sizeof(double)), "hipMalloc srgb table") &&
               up(ctx->d_srgb_table, table.data(), 257 * sizeof(double), "upload srgb table");
    }
    for (int k = 0; good && k < bs_ctx::kSlots; k++) {
        bs_ctx::LaunchSlot &sl = ctx->slots[k];
        sl.d_counters = ctx->d_counters + (size_t)k * bs::kCounters;
        sl.h_counters = ctx->h_counters + (size_t)k * bs::kCounters;
        good = ok(hipEventCreate(&sl.ev0), "hipEventCreate") && ok(hipEventCreate(&sl.ev1), "hipEventCreate") &&
               ok(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming), "hipEventCreate");
    }
    // The uploads above are enqueued on the context's stream through its own page-locked staging pieces (bs::copy_in: the std::vectors
    // are pageable and go out of scope below); the context is only handed out when the device has really finished with them.
    // (The context's OWN stream, not the device: a host application with other streams on this device does not stall on a bs_create.)
    if (good) good = ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if (!good) {
        const std::string keep = bs::error_message();
        owner.reset();   // bs_destroy
        (void)fail(BS_EDEVICE, keep);
        return nullptr;
    }
    return owner.release();
} catch (...) { (void)bs::abi_exception("bs_create"); return nullptr; }
int bs_device_count(void)
try {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return fail(BS_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return count;
} catch (...) { return bs::abi_exception("bs_device_count"); }

void bs_destroy(bs_ctx *ctx)
try {
    if (!ctx) return;
    const bs::OnDevice on_device(ctx->device);
    if (on_device.ok()) {
        // Everything THIS context enqueued has finished before its memory goes -- and nothing else is waited for: the context's own streams,
        // and, for work the caller had enqueued on streams of their own (*_device entry points), the event the context recorded behind
        // every such call, error returns included (ForeignWork; also every launch slot's ev_done, the blur / PNG scratch's ev_post / ev_png).
        // No hipDeviceSynchronize of our own.  (The runtime's hipFree below waits for the whole device on ROCm 7.2 -- measured: a bs_destroy
        // behind another context's 19 ms kernel takes 18.9 ms -- but nothing here relies on it: quiesce() has waited for what is ours.)
        bs::quiesce(ctx);
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
        for (bs_ctx::Foreign &f : ctx->foreign)
            if (f.ev) (void)hipEventDestroy(f.ev);
        if (ctx->d_srgb_table) (void)hipFree(ctx->d_srgb_table);
        for (int b = 0; b < 2; b++) {
            if (ctx->h_stage[b]) (void)hipHostFree(ctx->h_stage[b]);
            if (ctx->ev_stage[b]) (void)hipEventDestroy(ctx->ev_stage[b]);
        }
        if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
        if (ctx->ev_u0) (void)hipEventDestroy(ctx->ev_u0);
        if (ctx->ev_u1) (void)hipEventDestroy(ctx->ev_u1);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
} catch (...) { (void)bs::abi_exception("bs_destroy"); }

int bs_set_mode(bs_ctx *ctx, int mode)
try {
    if (!ctx || (mode != BS_MODE_STRICT && mode != BS_MODE_FAST)) return fail(BS_EINVAL, "bad mode");
    ctx->mode = mode;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_set_mode"); }

int bs_get_mode(const bs_ctx *ctx) { return ctx ? ctx->mode : BS_EINVAL; }
int bs_validate_config(const bs_config *cfg)
try {
    if (!cfg) return fail(BS_EINVAL, "null argument");
    bs::TraceParams p;
    std::memset(&p, 0, sizeof p);
    std::string err;
    if (!bs::derive_params(*cfg, p, err)) return fail(BS_EINVAL, err);
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_validate_config"); }

int bs_effective_mode(const bs_ctx *ctx, const bs_config *cfg)
try {
    if (!ctx || !cfg) return fail(BS_EINVAL, "null argument");
    return bs::effective_mode(ctx, cfg);
} catch (...) { return bs::abi_exception("bs_effective_mode"); }

int bs_set_max_steps(bs_ctx *ctx, int max_steps)
try {
    if (!ctx || max_steps <= 0) return fail(BS_EINVAL, "bad max_steps");
    // the kernel counts a ray's steps in an int and the frame's in 64 bits: 2^30 rays (the largest frame) x 2^30 steps = 2^60
    if (max_steps > BS_MAX_STEPS_LIMIT) return fail(BS_EINVAL, "max_steps above BS_MAX_STEPS_LIMIT (2^30): the step counters could not hold a frame of capped rays");
    ctx->max_steps = max_steps;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_set_max_steps"); }
void *bs_host_alloc(bs_ctx *ctx, size_t bytes)
try {
    if (!ctx || bytes == 0) { fail(BS_EINVAL, "null context or zero size"); return nullptr; }
    void *p = nullptr;
    const bs::OnDevice on_device(ctx->device);
    if (!on_device.ok() || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        fail(BS_ENOMEM, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
} catch (...) { (void)bs::abi_exception("bs_host_alloc"); return nullptr; }

void bs_host_free(void *p)
try {
    if (p) (void)hipHostFree(p);
} catch (...) { (void)bs::abi_exception("bs_host_free"); }

int bs_numa_node(const bs_ctx *ctx) { return ctx ? ctx->numa_node : -1; }

int bs_host_page_node(const void *p)
try {
    return p ? bs::numa_node_of_page(p) : -1;
} catch (...) { (void)bs::abi_exception("bs_host_page_node"); return -1; }

int bs_files_stats(const bs_ctx *ctx, bs_files_stats_t *out)
try {
    if (!ctx || !out) return fail(BS_EINVAL, "null argument");
    *out = ctx->files_stats;
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_files_stats"); }

}  // extern "C"
