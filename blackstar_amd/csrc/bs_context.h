// bs_context.h -- the context behind the C ABI (struct bs_ctx) and the host-side helpers its translation units share:
//   context.cpp  lifetime, settings, error state, page-locked buffers (zero copy), scratch
//   render.cpp   Raytracer.render: parameter derivation, launch slots, bs_render* / bs_star_lookup / bs_stats
//   post.cpp     the steps after render on the device: bloom, supersample, sRGB8, the PNG file (single frame)
//   batch.cpp    many frames / many contexts: pipelines, the CU partition, bs_render_*_batch, bs_render_png_files, bs_render_split
//   bs_debug.cpp the test hooks -- built into libblackstar_gpu_debug.so, NOT into the product library
// Not part of the ABI.  The debug library is built from the same tree in the same make run and reads this struct directly;
// ctx_layout_bytes() lets it refuse a product library of another build.
#pragma once

#include <hip/hip_runtime.h>
#include <sched.h>

#include <string>
#include <utility>
#include <vector>

#include "bs_internal.h"

struct bs_ctx {
    int device = -1;
    int mode = BS_MODE_FAST;
    int max_steps = 100000;
    int disk_slots = 4;
    bool zero_copy = true;       // page-locked caller buffers are written by the kernel itself (env BLACKSTAR_ZERO_COPY=0: always stage + copy)
    bool fast_guard = true;      // FAST mode re-traces photon-sphere-grazing rays in STRICT (env BLACKSTAR_FAST_GUARD=0 turns it off for A/B)
    double fast_max_steps = BS_FAST_MAX_EXPECTED_STEPS;   // FAST frames whose expected steps per ray exceed this are traced in STRICT (env BLACKSTAR_FAST_MAX_STEPS; 0 = no limit)
    int n_cu = 256;
    int blocks_per_cu = 4;       // resident workgroups per CU (VGPR/LDS-limited); env BLACKSTAR_BLOCKS_PER_CU for A/B builds
    int stagger_cycles = 16000;  // first-tile phase offset per SIMD slot (env BLACKSTAR_STAGGER overrides; 0 = off)
    int static_first_below = 1 << 20;  // launches of fewer tiles per resident wavefront than this (= every launch) start wavefront g on tile g, without a queue pop (env BLACKSTAR_STATIC_FIRST_BELOW for A/B; 0 = never, as in rounds 1-5)
    int late_pop_slot = 0;       // wavefronts of residency slots >= this pop their next tile AFTER tracing the current one, not before: 0 = everybody (the product), 4 = nobody (rounds 1-5); env BLACKSTAR_LATE_POP_SLOT for A/B (trace_kernel.hip)
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
    // bs_render_rgb8_batch / bs_render_png_batch with a PARTITIONED chip: the trace kernels run on streams whose CU mask leaves M CUs
    // out (M / 8 in every XCD), and bloom + sRGB8 (+ the PNG encoder) run on a stream that owns exactly those -- batch.cpp
    int post_cus_req = -1;       // env BLACKSTAR_POST_CUS: -1 = measure per frame shape (default), 0 = never partition, 8..32 = always that many
    int post_plan_cus = 0;       // CUs the blur sweeps are PLANNED for on the post stream (env BLACKSTAR_POST_PLAN_CUS; 0 = the partition's)
    int launch_cus = 0;          // CUs the next trace launches may use (0 = n_cu): sizes the persistent grid
    int last_post_cus = -1;      // CUs the post stage owned in this context's share of the last batch call (0: shared chip; -1: none yet)
    int bloom_plan_cus = 0;      // probe only (env BLACKSTAR_BLOOM_PLAN_CUS): CU count bs_bloom_device plans its sweeps for (0 = n_cu)
    // The partition decision is MEASURED, once per frame shape and context (batch.cpp: "measured, not modelled"): shares of one unmeasured
    // shape run segments of 8 frames shared / with 8 / with 16 post CUs, their steady state timed, and the fastest is remembered here.
    struct PartitionKey {
        int32_t w, h, ss, divider, png, mode;  // divider 0 = no bloom
        bool operator==(const PartitionKey &o) const { return w == o.w && h == o.h && ss == o.ss && divider == o.divider && png == o.png && mode == o.mode; }
    };
    struct PartitionChoice {
        PartitionKey key;
        int post_cus;           // the measured best: 0 (shared chip), 8 or 16
        double ms[3];           // per-frame wall time of the trial segments: shared, 8, 16 (0: not run)
    };
    std::vector<PartitionChoice> partition_cache;
    struct Trial {               // a trial in progress (its segments may span several batch calls)
        bool active = false;
        PartitionKey key{};
        int stage = 0;           // segments timed so far: 0..3
        double ms[3] = {0, 0, 0};
    } trial;
    double last_batch_end_ms = 0;  // host clock when this context's last rgb8 / png batch work ended (0: never): idle contexts warm up first
    int last_trial = 0;          // the last batch call of this context: 0 no trial, 1 a trial ended in it (shape remembered), 2 a trial progressed (test hook)
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
    // slots), a device copy of the file for callers with pageable buffers, and the file's size in page-locked memory.  Slots 0..2 belong
    // to the batch pipelines, slot 3 to the enqueue-only / single-frame entry points (bs_encode_png_device, bs_encode_png, bs_render_png).
    static constexpr int kPngSlots = 4;
    static constexpr int kPngSingle = 3;
    unsigned char *d_png_scratch[kPngSlots] = {};
    size_t png_scratch_cap[kPngSlots] = {};
    unsigned char *d_png_file[kPngSlots] = {};
    size_t png_file_cap[kPngSlots] = {};
    uint64_t *h_png_bytes = nullptr;   // page-locked, kPngSlots entries (the kernels write them through the device alias)
    hipEvent_t ev_png = nullptr;       // behind the last user of PNG slot kPngSingle: see acquire_png
    hipStream_t png_stream = nullptr;
    bool png_busy = false;
    // page-locked file buffers bs_render_png_files keeps between calls (page-locking 6 MB costs 1-2 ms): (pointer, capacity) -- the ring its
    // writer thread drains
    std::vector<std::pair<unsigned char *, size_t>> file_pool;
    bs_files_stats_t files_stats{};   // this context's share of the last bs_render_png_files call (bs_files_stats)
    // Where the GPU sits in the host (host_topology.cpp, probed once in bs_create): the host threads that drive this context -- the
    // per-context thread of every batch form and the file writer -- run on the CPUs of the GPU's NUMA node.
    int numa_node = -1;               // /sys/bus/pci/devices/<bdf>/numa_node; -1: unknown
    cpu_set_t numa_cpus;              // that node's CPUs, restricted to what the process may use
    bool numa_bind = false;           // binding would change something (env BLACKSTAR_NUMA_BIND=0: never)
    bool numa_confined = false;       // the whole process is confined to that node's CPUs already: nothing to bind, every thread is on the node
    // Staging for caller memory that is NOT page-locked (context.cpp: copy_in / copy_out): two page-locked pieces, used alternately
    static constexpr size_t kStageBytes = size_t(8) << 20;
    unsigned char *h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};   // ev_stage[b] has been recorded behind a DMA that reads stage b and nobody has waited for it yet
    // Work the *_device entry points enqueue on a CALLER's stream reads and writes memory this context owns (the star grid, counters, the
    // sRGB8 table, blur / PNG scratch): behind every such call -- on every exit path -- an event is recorded on that stream (ForeignWork), and
    // bs_destroy waits for them before anything is freed.  One entry per caller stream seen lately.
    struct Foreign { hipStream_t s = nullptr; hipEvent_t ev = nullptr; bool used = false; };
    Foreign foreign[4];
    int foreign_next = 0;
    struct VerifiedRange { const void *host = nullptr; size_t bytes = 0; };
    VerifiedRange verified[8];  // host buffers device_alias_of_pinned has walked page by page (registered memory without a queryable range)
    int verified_next = 0;
    bs_stats_t stats{};
};

namespace bs {

// ---- context.cpp ----------------------------------------------------------------------------------------------------------------
int fail(int code, const std::string &msg);   // sets the calling thread's bs_last_error() message, returns code
const std::string &error_message();           // the calling thread's message (to carry a worker thread's error to the caller's)
size_t ctx_layout_bytes();                    // sizeof(bs_ctx) as the product library was compiled (checked by the debug library)

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return bs::fail(BS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The calling thread's current HIP device is the context's while this object lives and the CALLER's again afterwards: a host application
// that drives other devices from the same thread (torch, another library, its own kernels) finds its current device as it left it after
// every call into this library.  (A thread whose current device is the context's already pays one hipGetDevice.)
struct OnDevice {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipErrorInvalidDevice;
    explicit OnDevice(int device)
    {
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        err = prev == device ? hipSuccess : hipSetDevice(device);
        switched = err == hipSuccess && prev >= 0 && prev != device;
    }
    OnDevice(const OnDevice &) = delete;
    OnDevice &operator=(const OnDevice &) = delete;
    ~OnDevice()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    bool ok() const { return err == hipSuccess; }
};
#define BS_ON_DEVICE(ctx)                     \
    bs::OnDevice on_device_((ctx)->device);   \
    if (!on_device_.ok()) return bs::fail(BS_EDEVICE, std::string("hipSetDevice: ") + hipGetErrorString(on_device_.err))

// On every exit path of a blocking entry point nothing of the call may still be in flight: the caller's buffers are DMA
// targets, and after an error return the caller is free to release them.  (On the success path the streams have been
// synchronised already and this costs a few microseconds.)
struct StreamDrain {
    bs_ctx *ctx;
    explicit StreamDrain(bs_ctx *c) : ctx(c) {}
    StreamDrain(const StreamDrain &) = delete;
    StreamDrain &operator=(const StreamDrain &) = delete;
    ~StreamDrain();
};

// Declared first thing in every *_device entry point: when the call returns -- with or without an error, whatever it had enqueued by then --
// an event of the context is recorded on the caller's stream behind it (nothing for the context's own streams: StreamDrain covers those).
struct ForeignWork {
    bs_ctx *ctx;
    hipStream_t s;
    ForeignWork(bs_ctx *c, void *hip_stream) : ctx(c), s(static_cast<hipStream_t>(hip_stream)) {}
    ForeignWork(const ForeignWork &) = delete;
    ForeignWork &operator=(const ForeignWork &) = delete;
    ~ForeignWork();
};

// The device alias of a caller's page-locked HOST buffer (zero copy), or nullptr for pageable memory; *straddles: the buffer starts in
// page-locked memory but is not contained in it (BS_EINVAL for the caller, with kStraddleMsg).  See context.cpp.
double *device_alias_of_pinned(bs_ctx *ctx, const void *host, size_t bytes, bool *straddles = nullptr);
extern const char *const kStraddleMsg;

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

// Host <-> device copies of CALLER memory.  Page-locked caller memory and pieces of at most 1 MiB go to hipMemcpyAsync as they are; larger
// PAGEABLE memory is moved through the context's own page-locked staging pieces with a host memcpy, because above 1 MiB the runtime pins the
// caller's pages on the fly and DMAs them at the caller's address ("HSA Copy Using Pinned resource", 55 GB/s) -- a path on which a GPU memory
// fault at a host heap address was caught twice in ~30 runs of the GPU suite (a process that allocates and frees multi-megabyte buffers all
// the time; profiles/EXPERIMENTS.md section 5).  A fault ends the process; 10 GB/s on the path the header calls the slow one does not.
//   copy_in : enqueued on s; returns once h_src has been consumed (the last pieces may still be in flight ON s, from the staging pieces)
//   copy_out: BLOCKING -- returns when everything enqueued on s before it has finished and h_dst holds the bytes
int copy_in(bs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t s);
int copy_out(bs_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, hipStream_t s);
int ensure_scratch(bs_ctx *ctx, size_t bytes);
int post_cus_setting(int v);  // BLACKSTAR_POST_CUS as a number: 0 = never partition, otherwise a multiple of 4 in [8, 32]
int effective_mode(const bs_ctx *ctx, const bs_config *cfg);
double expected_steps(const bs_config *cfg);   // N0 = (|camera| + sqrt safeDistance) / stepSize

// ---- host_topology.cpp ------------------------------------------------------------------------------------------------------------
void probe_host_topology(bs_ctx *ctx);        // fills numa_node / numa_cpus / numa_bind (never fails: no information = no binding)
int numa_node_of_page(const void *p);         // the NUMA node the page at p lives on, -1 unknown
// The calling thread runs on the CPUs of the context's NUMA node while this object lives; its previous affinity comes back afterwards
// (a caller's own thread is only borrowed).  bound(): the affinity was really changed.
class NumaBind {
public:
    explicit NumaBind(const bs_ctx *ctx);
    ~NumaBind();
    NumaBind(const NumaBind &) = delete;
    NumaBind &operator=(const NumaBind &) = delete;
    bool bound() const { return bound_; }
private:
    cpu_set_t saved_;
    bool bound_ = false;
};

// ---- render.cpp -----------------------------------------------------------------------------------------------------------------
// row0/row1: the band of OUTPUT rows to render ([0, height) = the frame).
int fill_params(bs_ctx *ctx, const bs_config *cfg, TraceParams &p, int row0 = 0, int row1 = -1);
int resolve_stats(bs_ctx *ctx);
// first/last: a frame (or band) delivered as several consecutive launches accumulates ONE set of statistics; quiet: a batch frame
// (no timing events, no read-back, bs_stats untouched).  See render.cpp.
int enqueue_render(bs_ctx *ctx, const bs_config *cfg, double *d_out, size_t out_doubles, hipStream_t s, int row0 = 0, int row1 = -1,
                   bool first = true, bool last = true, bool quiet = false);

// ---- post.cpp -------------------------------------------------------------------------------------------------------------------
int ensure_post(bs_ctx *ctx, size_t n);
int check_bloom_args(int width, double strength, int divider);
// bloom when strength != 0, then writeImg's pixel map: d_img (f64, w x h x 3) -> d_u8 (RGB8) on stream s; plan_cus: the CUs the blur sweeps are planned for
int enqueue_post_rgb8(bs_ctx *ctx, const double *d_img, int w, int h, double strength, int divider, unsigned char *d_u8, int plan_cus, hipStream_t s);
int check_png_frame(int width, int height);
int ensure_png(bs_ctx *ctx, int k, int w, int h, bool device_file);
uint64_t *png_bytes_slot(bs_ctx *ctx, int k);

// ---- batch.cpp ------------------------------------------------------------------------------------------------------------------
// The decision rule of the partition trial (host-only, pure): ms[i] = measured per-frame time with cus[i] post-stage CUs (cus[0] == 0:
// the shared chip).  Returns the CU count of the fastest entry, the shared chip unless a partition beats it by more than kTrialMargin.
int pick_partition(const double *ms, const int *cus, int n);
constexpr double kTrialMargin = 0.015;  // (4-5 steady-state frame intervals per segment resolve the per-frame time to about 1 %)

}  // namespace bs
