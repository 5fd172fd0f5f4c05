// batch.cpp -- many frames and many contexts: the reference's batch loop (app/Main.hs:68-77) as pipelines of frames in flight per
// context, one host thread per context (on the CPUs of the GPU's NUMA node), frame i on context i % n_ctx; the post stage on its own CUs
// where that measures faster; files written by a writer thread PER CONTEXT out of that context's ring of page-locked buffers; one huge
// frame split into row bands.  No data-path collective and no cross-context synchronisation anywhere: frames and bands are independent.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "bs_context.h"

using namespace bs;

// (the bs_* entry points get C linkage from their declarations in include/blackstar_gpu.h)

// One host thread per context: body(c) for c = 0 .. n-1, each on its own thread (a single context: on the caller's).  A thread that cannot
// be started (std::system_error: EAGAIN under a process / thread limit) must not become an exception in an `extern "C"` function, which
// would end the process: its body runs on the calling thread instead, after the others have been started.
// Whichever thread runs body(c) does so on the CPUs of context c's NUMA node (bs::NumaBind; the caller's own thread gets its affinity back).
// No exception leaves a body -- on a worker thread it would end the process, on the calling thread it would unwind past joinable
// threads --: it is translated like at the ABI (abi_exception: std::bad_alloc -> BS_ENOMEM, else BS_EINTERNAL) and returned, with its
// message, after every thread has been joined.  BS_OK otherwise (the bodies report their own errors through what they capture).
template <class Body>
static int per_context(bs_ctx *const *ctxs, int n, Body inner)
{
    std::vector<int> thrown((size_t)n, BS_OK);
    std::vector<std::string> what((size_t)n);
    auto body = [&](int c) {
        try {
            bs::NumaBind on_node(ctxs[c]);
            inner(c);
        } catch (...) {
            thrown[(size_t)c] = bs::abi_exception("a per-context thread");
            try { what[(size_t)c] = bs::error_message(); } catch (...) {}
        }
    };
    if (n == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        std::vector<int> inline_later;
        th.reserve((size_t)n);
        for (int c = 0; c < n; c++) {
            try {
                th.emplace_back(body, c);
            } catch (const std::system_error &) {
                inline_later.push_back(c);
            }
        }
        for (int c : inline_later) body(c);
        for (auto &t : th) t.join();
    }
    for (int c = 0; c < n; c++)
        if (thrown[(size_t)c]) return fail(thrown[(size_t)c], what[(size_t)c]);
    return BS_OK;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Frames first, first+step, ... on one context, double-buffered: while frame k's image is copied to the host (copy
// stream), frame k+1's kernel already runs (compute stream).  Pageable host buffers: the copy itself is the runtime's
// staged D2H (about 19 GB/s), but it no longer sits between two kernels.
static int render_frames_pipelined(bs_ctx *ctx, const bs_config *cfgs, double *const *outs, int first, int n_frames, int step)
{
    if (first >= n_frames) return BS_OK;
    BS_ON_DEVICE(ctx);
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
        rc = copy_out(ctx, outs[i], buf[k & 1], (size_t)cfgs[i].width * cfgs[i].height * 3 * sizeof(double), ctx->copy_stream);
        if (rc) return rc;
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
try {
    if (!ctxs || n_ctx <= 0 || (n_frames > 0 && (!cfgs || !outs))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    // One host thread per context (= per device); frame i goes to context i % n_ctx.  No data-path
    // collective: frames are independent (app/Main.hs:72-77 renders them one after another).
    std::vector<int> rcs(n_ctx, BS_OK);
    std::vector<std::string> errs(n_ctx);
    const int thrown = per_context(ctxs, n_ctx, [&](int c) {
        rcs[c] = render_frames_pipelined(ctxs[c], cfgs, outs, c, n_frames, n_ctx);
        if (rcs[c]) errs[c] = bs::error_message();
    });
    if (thrown) return thrown;
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_batch"); }

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

// ---- bs_render_png_files: one context's file buffers and its writer ---------------------------------------------------------------
// The reference's batch loop ends every scene with writeImg's write (app/Main.hs:119-123, src/Raytracer.hs:29-32).  Here each context has
// a RING of page-locked file buffers of its own (bs_host_alloc on its device: pages on the GPU's NUMA node) and a WRITER thread of its
// own (on that node's CPUs): the render pipeline takes a free buffer for the frame it is about to enqueue (acquire), the PNG encoder
// writes the file into it over PCIe, and when the frame has left the device the buffer goes to the writer (submit), which creates /
// writes / closes the file and gives the buffer back.  The pipeline only ever waits for ITS writer, and only when every buffer of the
// ring is still to be written -- a context never waits for another context's frames or files, and there is no chunk boundary at which
// the pipeline drains (round 5 had one writer thread for all contexts and a barrier across them every 16 frames per context).
// An error anywhere -- a file that cannot be written, a device error -- raises the call's `stop` flag: every pipeline stops taking new
// frames (acquire returns nullptr -> kCancelled), drains what it has in flight, and every writer is joined before the call returns.
constexpr int kCancelled = -1000;   // internal: never crosses the ABI
constexpr int kMinRing = 4;         // frames in flight per context (<= 3) + one with the writer: the pipeline can always make progress
constexpr int kDefaultRing = 16;

struct FileRing {
    struct Job { int frame; int buf; size_t bytes; };
    bs_ctx *ctx = nullptr;
    const char *const *paths = nullptr;
    std::vector<unsigned char *> bufs;
    size_t cap = 0;
    std::atomic<bool> *stop = nullptr;   // the call's
    std::mutex m;
    std::condition_variable cv_free, cv_work;
    std::vector<int> free_bufs;
    std::vector<std::pair<int, int>> taken;   // (frame, buffer) of frames acquired and not yet submitted
    std::deque<Job> jobs;
    bool closing = false;
    bool threaded = false;
    std::thread writer;
    std::string error;                   // the writer's (read after join, or under m)
    // what bs_files_stats reports
    uint64_t files = 0, bytes = 0;
    double busy_ms = 0, wait_ms = 0, t_last = 0;
    bool writer_bound = false;

    void start()
    {
        for (int b = (int)bufs.size() - 1; b >= 0; b--) free_bufs.push_back(b);
        try {
            writer = std::thread([this] { loop(); });
            threaded = true;
        } catch (const std::system_error &) {   // no thread to be had: submit() writes the file itself, without the overlap
            threaded = false;
        }
    }

    // A free buffer for frame `frame`; blocks while the writer still owns all of them.  nullptr: the call is being stopped.
    unsigned char *acquire(int frame)
    {
        std::unique_lock<std::mutex> lk(m);
        if (free_bufs.empty()) {
            const double t0 = now_ms();
            cv_free.wait(lk, [&] { return !free_bufs.empty() || stop->load(); });
            wait_ms += now_ms() - t0;
        }
        if (stop->load() || free_bufs.empty()) return nullptr;
        const int b = free_bufs.back();
        free_bufs.pop_back();
        taken.emplace_back(frame, b);
        return bufs[(size_t)b];
    }

    // Frame `frame`'s file (bytes long) is complete in the buffer it acquired: over to the writer.
    int submit(int frame, size_t n)
    {
        int b = -1;
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t k = 0; k < taken.size(); k++)
                if (taken[k].first == frame) { b = taken[k].second; taken.erase(taken.begin() + (long)k); break; }
            if (b < 0) return fail(BS_EINTERNAL, "file ring: frame without a buffer");
            if (threaded) {
                jobs.push_back(Job{frame, b, n});
                cv_work.notify_one();
                return BS_OK;
            }
        }
        write_one(Job{frame, b, n});   // (no writer thread: here, on the pipeline's)
        std::lock_guard<std::mutex> lk(m);
        free_bufs.push_back(b);
        return error.empty() ? BS_OK : kCancelled;
    }

    void write_one(const Job &j)
    {
        if (!error.empty()) return;   // after the first failure nothing more is written
        const double t0 = now_ms();
        std::string e = write_whole_file(paths[j.frame], bufs[(size_t)j.buf], j.bytes);
        t_last = now_ms();
        busy_ms += t_last - t0;
        if (e.empty()) {
            files++;
            bytes += j.bytes;
        } else {
            std::lock_guard<std::mutex> lk(m);   // (not held by either caller at this point)
            error = std::move(e);
            stop->store(true);
        }
    }

    void loop()
    {
        bs::NumaBind on_node(ctx);
        writer_bound = on_node.bound() || ctx->numa_confined;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return !jobs.empty() || closing; });
            if (jobs.empty()) return;   // closing, and everything handed over has been dealt with
            const Job j = jobs.front();
            jobs.pop_front();
            lk.unlock();
            write_one(j);
            lk.lock();
            free_bufs.push_back(j.buf);
            cv_free.notify_all();
        }
    }

    // No more frames will be submitted: the writer finishes what it has and is joined.  Called on every exit path (the destructor does).
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            closing = true;
        }
        cv_work.notify_all();
        cv_free.notify_all();
        if (writer.joinable()) writer.join();
    }
    void wake()   // (the call's stop flag was raised by somebody else; m is taken so that a pipeline between its check and its wait is not missed)
    {
        { std::lock_guard<std::mutex> lk(m); }
        cv_free.notify_all();
    }
    ~FileRing() { close(); }
};

// Where a batch's frames go: RGB8 pixels (png == nullptr), finished PNG files in the caller's buffers (bs_render_png_batch: caps / sizes),
// or files on disk through the context's ring (bs_render_png_files: ring; the pipelines' outs[] is null then).
struct PngSink {
    const size_t *caps;   // capacity of outs[i]
    size_t *sizes;        // receives the size of file i
    FileRing *ring;
};

// What every frame of a context's share must satisfy before anything is launched; returns the largest frame (values) in *need.
static int check_rgb8_share(const bs_config *cfgs, const double *strengths, const int *dividers, unsigned char *const *outs, const PngSink *png,
                            int first, int n_frames, int step, size_t *need)
{
    *need = 0;
    for (int i = first; i < n_frames; i += step) {
        const bool to_ring = png && png->ring;
        if (cfgs[i].width <= 0 || cfgs[i].height <= 0 || (!to_ring && !outs[i])) return fail(BS_EINVAL, "bad frame");
        const double st = strengths ? strengths[i] : 0.0;
        if (st != 0 && !dividers) return fail(BS_EINVAL, "bloom radius (width `div` bloomDivider) must be >= 1");
        if (int rc = check_bloom_args(cfgs[i].width, st, st != 0 ? dividers[i] : 1)) return rc;
        if (png) {
            if (int rc = check_png_frame(cfgs[i].width, cfgs[i].height)) return rc;
            if ((to_ring ? png->ring->cap : png->caps[i]) < bs::png_file_bound(cfgs[i].width, cfgs[i].height))
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
    int frame[3] = {-1, -1, -1};   // (the context's PNG slots 0..2; slot bs_ctx::kPngSingle belongs to the single-frame entry points)
    bool staged[3] = {false, false, false};
    unsigned char *out[3] = {nullptr, nullptr, nullptr};   // where slot k's file goes: the caller's outs[i], or the ring buffer the frame acquired

    // frame i's file buffer: the caller's, or -- files on disk -- a free one of the context's ring (may wait for the writer; nullptr: stopped)
    static unsigned char *target_of(const PngSink &png, unsigned char *const *outs, int i) { return png.ring ? png.ring->acquire(i) : outs[i]; }

    int enqueue(bs_ctx *ctx, int k, int i, const unsigned char *d_u8, const bs_config &cfg, unsigned char *file_out, hipStream_t s)
    {
        bool straddles = false;
        double *alias = device_alias_of_pinned(ctx, file_out, (size_t)bs::png_file_bound(cfg.width, cfg.height), &straddles);
        if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        int rc = ensure_png(ctx, k, cfg.width, cfg.height, alias == nullptr);
        if (rc) return rc;
        uint64_t *d_bytes = png_bytes_slot(ctx, k);
        if (!d_bytes) return fail(BS_EDEVICE, "hipHostGetDevicePointer failed");
        unsigned char *target = alias ? reinterpret_cast<unsigned char *>(alias) : ctx->d_png_file[k];
        if (bs::launch_png_encode(d_u8, cfg.width, cfg.height, ctx->d_png_scratch[k], target, d_bytes, s)) return fail(BS_EDEVICE, "PNG encoder launch failed");
        frame[k] = i;
        staged[k] = alias == nullptr;
        out[k] = file_out;
        return BS_OK;
    }

    int retire(bs_ctx *ctx, int k, const PngSink &png, hipStream_t s)
    {
        if (frame[k] < 0) return BS_OK;
        const size_t bytes = (size_t)ctx->h_png_bytes[k];
        if (staged[k]) {
            if (int rc = copy_out(ctx, out[k], ctx->d_png_file[k], bytes, s)) return rc;
        }
        const int i = frame[k];
        frame[k] = -1;
        if (png.ring) return png.ring->submit(i, bytes);
        png.sizes[i] = bytes;
        return BS_OK;
    }
};

// doRender (app/Main.hs:105-123) for frames first, first+step, ... on one context, two frames in flight: frame k runs render ->
// bloom -> sRGB8 (-> PNG) on compute stream k & 1 with its own f64 image, so frame k+1's trace kernel fills the SIMDs frame k's last
// tiles leave (the fixed ~0.25 ms of a launch) and frame k's bloom runs on the CUs the next trace kernel frees
// first.  The blur scratch is one pair per context: the bloom of frame k+1 is ordered behind frame k's by acquire/release_post.
static int render_rgb8_frames_pipelined(bs_ctx *ctx, const bs_config *cfgs, const double *strengths, const int *dividers, unsigned char *const *outs,
                                        int first, int n_frames, int step, const PngSink *png, std::vector<double> *done_ms = nullptr)
{
    if (first >= n_frames) return BS_OK;
    BS_ON_DEVICE(ctx);
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
    // RGB8 frames for PAGEABLE outputs wait in their staging image until the slot is retired: { where to, how many bytes }
    struct Pending { unsigned char *out = nullptr; size_t bytes = 0; } pageable[2];
    auto deliver = [&](int b) -> int {   // (the slot's frame has left the device: bs::copy_out moves it without letting the runtime pin caller pages)
        if (!pageable[b].out) return BS_OK;
        const int r = copy_out(ctx, pageable[b].out, stage[b], pageable[b].bytes, cs[b]);
        pageable[b] = Pending{};
        return r;
    };
    StreamDrain drain(ctx);  // the caller's outs[] are DMA targets from here on: every return path drains the streams first
    int k = 0;
    for (int i = first; i < n_frames; i += step, k++) {
        const int b = k & 1;
        if (k >= 2) {
            HIP_TRY(hipEventSynchronize(ctx->ev_frame[b]));  // frame k-2 (same image, same staging) has left the device
            if (done_ms) done_ms->push_back(now_ms());       // (the partition trial: when each frame of the pipeline completed)
            if (png && (rc = files.retire(ctx, b, *png, cs[b]))) return rc;
            if ((rc = deliver(b))) return rc;
        }
        const size_t n = (size_t)cfgs[i].width * cfgs[i].height * 3;
        unsigned char *target = stage[b], *file_out = nullptr;
        if (!png) {
            bool straddles = false;
            if (double *alias = device_alias_of_pinned(ctx, outs[i], n, &straddles)) target = reinterpret_cast<unsigned char *>(alias);  // page-locked: written in place
            if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        } else if (!(file_out = PngSlots::target_of(*png, outs, i))) {
            return kCancelled;   // (only a ring says no: the call is being stopped)
        }
        rc = enqueue_render(ctx, &cfgs[i], img[b], n, cs[b], 0, -1, true, true, /*quiet=*/true);
        if (rc) return rc;
        rc = enqueue_post_rgb8(ctx, img[b], cfgs[i].width, cfgs[i].height, strengths ? strengths[i] : 0.0, dividers ? dividers[i] : 1, target, ctx->n_cu, cs[b]);
        if (rc) return rc;
        if (png) {
            rc = files.enqueue(ctx, b, i, stage[b], cfgs[i], file_out, cs[b]);
            if (rc) return rc;
        } else if (target == stage[b]) {
            pageable[b] = Pending{outs[i], n};
        }
        HIP_TRY(hipEventRecord(ctx->ev_frame[b], cs[b]));
    }
    // The last two frames, in frame order (the older slot first), each retired as soon as IT has left the device: a file goes to the
    // writer while the frame behind it is still on the GPU, so only the very last file's write is not hidden behind rendering.
    for (int b = 0; b < 2; b++) {
        const int slot = (k + b) & 1;
        if (k - 2 + b < 0) continue;   // (a share of one frame has no older slot)
        HIP_TRY(hipEventSynchronize(ctx->ev_frame[slot]));
        if (png && (rc = files.retire(ctx, slot, *png, cs[slot]))) return rc;
        if ((rc = deliver(slot))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(cs[0]));
    HIP_TRY(hipStreamSynchronize(cs[1]));
    return BS_OK;
}

// ---- the CU partition: measured, not modelled -----------------------------------------------------------------------------------
// With the chip partitioned a frame costs trace x n_cu / (n_cu - M), provided the post stage (bloom + sRGB8, + the PNG encoder) confined
// to M CUs keeps up; on the shared chip it costs trace + post + a hand-over stall (a blur workgroup only ever gets a CU in the drain of a
// later trace kernel, and delays the one behind it).  Which side wins depends on the frame's shape, the scene, the arithmetic mode and the
// chip's clocks -- rounds 2-3 measured 26 combinations (profiles/r03_post_partition_ab.txt, r03_partition_large_ab.jsonl,
// r03_partition_more_ab.jsonl, r03_png_partition_ab.jsonl: C3 at 1080p 4.27 partitioned on 8 CUs / 4.67 ms shared, 720p wants 16, frames
// without supersampling lose with any M, ...) and fitted a model with five constants to them.  Round 4 replaced the model with the
// measurement itself.  A share of frames of ONE shape that the context has not measured yet is rendered as a TRIAL: segments of
// kTrialSegment frames, each a self-contained blocking pipeline -- shared chip, 16 post-stage CUs, 8 -- preceded, if the context's last
// batch work ended more than a few milliseconds ago, by at least kTrialWarm frames and 30 ms on the shared chip (the first launches after
// an idle spell run up to 20 % slow).  The segments need not fit one call: the context remembers how far it got (bs_ctx::trial), so 32 frames in one call
// measure a shape, and so do three calls of 16 (bs_render_png_files' internal calls).  When all three are timed the fastest is remembered
// for that shape (PartitionKey) for the life of the context.  What is timed is the segment's STEADY STATE: the host clock at the moments
// the pipeline's own loop learns that frame k has left the device (it waits for frame k - depth before it enqueues frame k), first to
// last, divided by the frames between.  The wall time of a short segment would not do: its fill and drain cost about one post stage on M
// CUs -- a whole frame time -- and a first version that timed 4-frame segments end to end chose the shared chip for 16 of 26 shapes on
// which a partition is 2-15 % faster (profiles/r04_partition_trial_ab_wall_time_v1.jsonl).  Every trial frame is a frame of the batch,
// delivered like any other (byte-identical whichever way it was made); what the trial costs is the difference between the segments, a
// few per cent of 32 frames, once.  The key is the frame's SHAPE (size, supersampling, bloom divider, pixels or file, arithmetic), not its
// scene: a heavier scene of the same shape may prefer the other partition (lensing-disk at 4K: 8 CUs, default-aa at 4K: 16) and then runs
// 2-3 % above its own optimum with the remembered one -- still ahead of the shared chip in every measured case.
// Shares that are too short, or mix shapes of which one has not been measured, run on the shared chip
// (the safe side: a partition that is too small for its post stage costs 50-70 %, none costs <= 9 %).
// BLACKSTAR_POST_CUS=0 | 8 | 16 | 24 | 32 overrides (A/B).
// Order: shared, 16, 8 -- NOT ascending.  A partition that is too small for its post stage leaves the trace CUs idle half of the time, the
// chip lowers its clocks, and the segment measured NEXT starts slow: with the order shared / 8 / 16 the 16-CU segment of every shape whose
// 8-CU segment was starved read 5-13 % high and lost to the shared chip on 4 of 26 shapes (profiles/r04_partition_trial_ab_order_v2.jsonl).
// 16 CUs are rarely starved; when they are (more than kStarved x the shared time) 8 can only be worse and is not run at all.
constexpr int kTrialWarm = 8, kTrialSegment = 8, kTrialWarmMax = 64;
constexpr double kIdleMs = 5.0;   // a context whose last batch work ended longer ago than this starts its next trial segment with a warm-up ...
constexpr double kWarmMs = 30.0;  // ... of at least this long (and kTrialWarm frames): the clocks need ~35 ms of work to come back from idle
constexpr double kStarved = 1.25;
static const int kTrialCus[3] = {0, 8, 16};   // (index into PartitionChoice::ms)
static const int kTrialOrder[3] = {0, 2, 1};  // stage -> index into kTrialCus

int bs::pick_partition(const double *ms, const int *cus, int n)
{
    int best = -1, shared = -1;
    for (int i = 0; i < n; i++) {
        if (!(ms[i] > 0)) continue;   // not run
        if (cus[i] == 0) shared = i;
        if (best < 0 || ms[i] < ms[best]) best = i;
    }
    if (best < 0) return 0;
    if (cus[best] != 0 && shared >= 0 && !(ms[best] < (1.0 - bs::kTrialMargin) * ms[shared])) return 0;   // not clearly better than doing nothing
    return cus[best];
}

static bool partition_key(const bs_ctx *ctx, const bs_config &cfg, double strength, int divider, bool png, bs_ctx::PartitionKey *key)
{
    if (cfg.width <= 0 || cfg.height <= 0) return false;
    *key = {cfg.width, cfg.height, cfg.supersampling ? 1 : 0, strength != 0 ? divider : 0, png ? 1 : 0, effective_mode(ctx, &cfg)};
    return true;
}

static const bs_ctx::PartitionChoice *find_choice(const bs_ctx *ctx, const bs_ctx::PartitionKey &key)
{
    for (const auto &c : ctx->partition_cache)
        if (c.key == key) return &c;
    return nullptr;
}

// What this context's share of a batch does about the partition: *post_cus = the CUs to set aside (0: shared chip), *trial = measure now.
static void plan_share(bs_ctx *ctx, const bs_config *cfgs, const double *strengths, const int *dividers, int first, int n_frames, int step, bool png,
                       int *post_cus, bool *trial, bs_ctx::PartitionKey *trial_key)
{
    *post_cus = 0;
    *trial = false;
    if (ctx->post_cus_req == 0 || ctx->n_cu < 128 || ctx->n_cu % 8 != 0) return;
    if (first + 2 * step >= n_frames) return;  // fewer than three frames for this context: nothing to hide the post stage behind (1 frame: 5.38 against 5.22 ms)
    if (ctx->post_cus_req > 0) { *post_cus = ctx->post_cus_req; return; }
    bool one_shape = true, all_known = true, with_post = false;
    int count = 0, widest = 0;
    bs_ctx::PartitionKey k0{};
    for (int i = first; i < n_frames; i += step, count++) {
        const double st = strengths ? strengths[i] : 0.0;
        bs_ctx::PartitionKey k;
        if ((st != 0 && !dividers) || !partition_key(ctx, cfgs[i], st, st != 0 ? dividers[i] : 0, png, &k)) return;  // (the pipeline will refuse the frame)
        with_post = with_post || st != 0 || png;
        if (count == 0) k0 = k;
        one_shape = one_shape && k == k0;
        if (const bs_ctx::PartitionChoice *c = find_choice(ctx, k)) {
            if (c->post_cus == 0) widest = -1;              // a shape that measured faster on the shared chip: the whole share stays there
            else if (widest >= 0) widest = std::max(widest, c->post_cus);
        } else {
            all_known = false;
        }
    }
    if (!with_post) return;          // no bloom and no file anywhere: the post stage is one 30-us pixel map, nothing to set CUs aside for
    if (all_known) { *post_cus = std::max(widest, 0); return; }
    if (one_shape && count >= kTrialSegment) { *trial = true; *trial_key = k0; }
}

// The CU-masked streams of a partition (post stage on bits [0, post_cus), trace kernels on the rest), made once per context and M.
// (hipExtStreamCreateWithCUMask takes no flags: unlike the context's other streams these synchronise with the NULL stream -- only a
// matter of overlap, and only if another thread of the caller keeps the NULL stream busy during a bs_render_rgb8_batch call.)
// false: the runtime would not make them (no CU-mask support on this device / driver) -- the caller falls back to the shared chip.
static bool ensure_partition(bs_ctx *ctx, int post_cus)
{
    const bs::OnDevice on_device(ctx->device);
    if (post_cus < 8 || post_cus > 32 || post_cus % 4 != 0 || !on_device.ok()) return false;
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
                                          unsigned char *const *outs, int first, int n_frames, int step, const PngSink *png, std::vector<double> *done_ms = nullptr)
{
    if (first >= n_frames) return BS_OK;
    BS_ON_DEVICE(ctx);
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
            if (done_ms) done_ms->push_back(now_ms());
            if (png && (rc = files.retire(ctx, b, *png, posted_on[b]))) return rc;
        }
        const size_t n = (size_t)cfgs[i].width * cfgs[i].height * 3;
        unsigned char *target = stage[b], *file_out = nullptr;
        if (!png) {
            bool straddles = false;
            if (double *alias = device_alias_of_pinned(ctx, outs[i], n, &straddles)) target = reinterpret_cast<unsigned char *>(alias);
            if (straddles) return fail(BS_EINVAL, kStraddleMsg);
        } else if (!(file_out = PngSlots::target_of(*png, outs, i))) {
            return kCancelled;
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
            rc = files.enqueue(ctx, b, i, stage[b], cfgs[i], file_out, ps);
            if (rc) return rc;
        } else if (target == stage[b]) {   // (run_share only partitions shares whose outputs are all page-locked)
            return fail(BS_EINTERNAL, "the partitioned pipeline was given a pageable output buffer");
        }
        HIP_TRY(hipEventRecord(ctx->ev_posted[b], ps));
    }
    // The last three frames in frame order, each retired as soon as IT has left the device (see the shared-chip pipeline above).
    for (int j = 0; png && j < 3; j++) {
        if (k - 3 + j < 0) continue;
        const int b = (k + j) % 3;   // slot of frame k - 3 + j
        HIP_TRY(hipEventSynchronize(ctx->ev_posted[b]));
        if ((rc = files.retire(ctx, b, *png, posted_on[b]))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(pt.trace[0]));
    HIP_TRY(hipStreamSynchronize(pt.trace[1]));
    HIP_TRY(hipStreamSynchronize(pt.post));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BS_OK;
}
// One context's share of bs_render_rgb8_batch / bs_render_png_batch: frames c, c + step, ... -- shared chip, partitioned, or the trial.
static int run_share(bs_ctx *x, const bs_config *cfgs, int n_frames, const double *strengths, const int *dividers, unsigned char *const *outs,
                     const PngSink *png, int c, int step)
{
    int post_cus = 0;
    bool trial = false;
    bs_ctx::PartitionKey key{};
    const bs::OnDevice on_device(x->device);   // (the pipelines below check it themselves: BS_ON_DEVICE)
    plan_share(x, cfgs, strengths, dividers, c, n_frames, step, png != nullptr, &post_cus, &trial, &key);
    // Only with page-locked outputs, which the last kernel of a frame writes itself: a copy into PAGEABLE memory blocks the
    // host thread until the frame's post stage has finished -- 3.8 ms on 8 CUs instead of 0.2 ms on the whole chip -- and the
    // next trace kernel is not enqueued meanwhile (measured 9.1 against 4.8 ms per frame: scripts/post_partition_pageable_ab.py)
    const bool to_ring = png && png->ring;   // (a ring's buffers are the library's own page-locked memory: one look at the first is enough)
    if ((post_cus || trial) && to_ring && (!on_device.ok() || !device_alias_of_pinned(x, png->ring->bufs[0], png->ring->cap))) {
        post_cus = 0;
        trial = false;
    }
    for (int i = c; (post_cus || trial) && !to_ring && i < n_frames; i += step) {
        if (!outs[i] || cfgs[i].width <= 0 || cfgs[i].height <= 0 || !on_device.ok() ||
            !device_alias_of_pinned(x, outs[i], png ? (size_t)bs::png_file_bound(cfgs[i].width, cfgs[i].height) : (size_t)cfgs[i].width * cfgs[i].height * 3)) {
            post_cus = 0;
            trial = false;
        }
    }
    if (post_cus && !ensure_partition(x, post_cus)) post_cus = 0;
    x->last_trial = 0;
    auto run = [&](int m, int a, int b, std::vector<double> *done_ms = nullptr) {  // frames a .. b-1 OF THE SHARE, with m post-stage CUs
        const int first = c + a * step, bound = (int)std::min<long>(n_frames, (long)c + (long)b * step);
        return m ? render_rgb8_frames_partitioned(x, m, cfgs, strengths, dividers, outs, first, bound, step, png, done_ms)
                 : render_rgb8_frames_pipelined(x, cfgs, strengths, dividers, outs, first, bound, step, png, done_ms);
    };
    const int count = (n_frames - c + step - 1) / step;
    struct Stamp {   // when this context's batch work last ended (whatever path the call took)
        bs_ctx *x;
        ~Stamp() { x->last_batch_end_ms = now_ms(); }
    } stamp{x};
    if (!trial) {
        x->last_post_cus = post_cus;
        return run(post_cus, 0, count);
    }
    // The trial (see "the CU partition: measured, not modelled" above), possibly continued from an earlier call.
    bs_ctx::Trial &T = x->trial;
    if (!T.active || !(T.key == key)) T = bs_ctx::Trial{true, key, 0, {0, 0, 0}};
    int done = 0, choice = 0;
    bool decided = false;
    if (!(ensure_partition(x, kTrialCus[1]) && ensure_partition(x, kTrialCus[2]))) {
        decided = true;   // a chip without CU-mask support measures nothing and stays shared
    } else {
        {   // every buffer either pipeline needs exists before anything is timed (the first segment would otherwise pay the allocations)
            size_t need = 0;
            if (int rc = check_rgb8_share(cfgs, strengths, dividers, outs, png, c, n_frames, step, &need)) return rc;
            if (!grow_device(x->d_img, x->img_cap, need) || !grow_device(x->d_img2, x->img2_cap, need) || !grow_device(x->d_img3, x->img3_cap, need) ||
                !grow_device(x->d_u8, x->u8_cap, need) || !grow_device(x->d_u8b, x->u8b_cap, need) || !grow_device(x->d_u8c, x->u8c_cap, need))
                return fail(BS_ENOMEM, "hipMalloc image failed");
            if (int rc = ensure_post(x, need)) return rc;
        }
        const bool cold = x->last_batch_end_ms == 0 || now_ms() - x->last_batch_end_ms > kIdleMs;
        if (cold) {   // shared-chip frames until the clocks are back: at least kTrialWarm frames and kWarmMs, while a segment still fits behind them
            const double t0 = now_ms();
            while (count - done >= kTrialWarm + kTrialSegment && done < kTrialWarmMax && (done == 0 || now_ms() - t0 < kWarmMs)) {
                if (int rc = run(0, done, done + kTrialWarm)) return rc;
                done += kTrialWarm;
            }
        }
        if (!cold || done) {   // (a cold context with too few frames to warm up AND measure: this call goes by on the shared chip)
            while (T.stage < 3 && count - done >= kTrialSegment) {
                const int v = kTrialOrder[T.stage];
                std::vector<double> t;   // completion times of the segment's frames but the last two (shared) / three (partitioned)
                if (int rc = run(kTrialCus[v], done, done + kTrialSegment, &t)) return rc;
                T.ms[v] = t.size() >= 2 ? (t.back() - t.front()) / (double)(t.size() - 1) : 0.0;
                done += kTrialSegment;
                T.stage++;
                if (T.stage == 2 && T.ms[0] > 0 && T.ms[2] > kStarved * T.ms[0]) T.stage = 3;   // 16 CUs starved: 8 is not worth a segment
            }
        }
        if (T.stage == 3) {
            decided = true;
            choice = pick_partition(T.ms, kTrialCus, 3);
        }
    }
    if (decided) {
        x->partition_cache.push_back(bs_ctx::PartitionChoice{key, choice, {T.ms[0], T.ms[1], T.ms[2]}});
        T.active = false;
    }
    x->last_trial = decided ? 1 : 2;
    x->last_post_cus = decided ? choice : 0;
    return done < count ? run(decided ? choice : 0, done, count) : BS_OK;
}

// bs_render_rgb8_batch / bs_render_png_batch: one host thread per context, frame i on context i % n_ctx.
static int render_post_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                             unsigned char *const *outs, const PngSink *png)
{
    if (!ctxs || n_ctx <= 0 || (n_frames > 0 && (!cfgs || !outs))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    std::vector<int> rcs(n_ctx, BS_OK);
    std::vector<std::string> errs(n_ctx);
    const int thrown = per_context(ctxs, n_ctx, [&](int c) {
        rcs[c] = run_share(ctxs[c], cfgs, n_frames, bloom_strengths, bloom_dividers, outs, png, c, n_ctx);
        if (rcs[c]) errs[c] = bs::error_message();
    });
    if (thrown) return thrown;
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
}

int bs_render_rgb8_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                         unsigned char *const *outs)
try {
    return render_post_batch(ctxs, n_ctx, cfgs, n_frames, bloom_strengths, bloom_dividers, outs, nullptr);
} catch (...) { return bs::abi_exception("bs_render_rgb8_batch"); }

int bs_render_png_batch(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                        unsigned char *const *outs, const size_t *caps, size_t *out_bytes)
try {
    if (n_frames > 0 && (!caps || !out_bytes)) return fail(BS_EINVAL, "null argument");
    const PngSink sink{caps, out_bytes, nullptr};
    return render_post_batch(ctxs, n_ctx, cfgs, n_frames, bloom_strengths, bloom_dividers, outs, &sink);
} catch (...) { return bs::abi_exception("bs_render_png_batch"); }

int bs_render_png_files(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfgs, int n_frames, const double *bloom_strengths, const int *bloom_dividers,
                        const char *const *paths, int pipe)
try {
    if (!ctxs || n_ctx <= 0 || n_frames < 0 || (n_frames > 0 && (!cfgs || !paths))) return fail(BS_EINVAL, "null argument");
    if (int rc = distinct_contexts(ctxs, n_ctx)) return rc;
    size_t cap = 0;
    for (int i = 0; i < n_frames; i++) {
        if (!paths[i]) return fail(BS_EINVAL, "null path");
        if (int rc = check_png_frame(cfgs[i].width, cfgs[i].height)) return rc;
        cap = std::max(cap, (size_t)bs::png_file_bound(cfgs[i].width, cfgs[i].height));
    }
    for (int c = 0; c < n_ctx; c++) ctxs[c]->files_stats = bs_files_stats_t{};
    if (n_frames == 0) return BS_OK;
    // Context c: frames c, c + n_ctx, ... as ONE rolling pipeline (run_share, exactly what bs_render_png_batch runs) into a ring of `pipe`
    // page-locked file buffers of its own, drained by a writer thread of its own (FileRing).  No context ever waits for another.
    const int ring_size = std::max(kMinRing, std::min(pipe > 0 ? pipe : kDefaultRing, (n_frames + n_ctx - 1) / n_ctx));
    std::atomic<bool> stop{false};
    std::vector<int> rcs(n_ctx, BS_OK);
    std::vector<std::string> errs(n_ctx);
    std::vector<FileRing *> rings(n_ctx, nullptr);   // (for wake-ups across contexts; each ring lives on its context's thread's stack)
    std::mutex rings_m;
    auto stop_all = [&] {
        stop.store(true);
        std::lock_guard<std::mutex> lk(rings_m);
        for (FileRing *r : rings)
            if (r) r->wake();
    };
    // (an exception in one context's body must stop the others too: they would otherwise render their whole share into a call that fails)
    struct StopOnUnwind {
        decltype(stop_all) &f;
        bool armed = true;
        ~StopOnUnwind() { if (armed) f(); }
    };
    const int thrown = per_context(ctxs, n_ctx, [&](int c) {
        bs_ctx *x = ctxs[c];
        if (c >= n_frames) return;
        StopOnUnwind on_unwind{stop_all};
        const double t0 = now_ms();
        FileRing ring;
        ring.ctx = x;
        ring.paths = paths;
        ring.cap = cap;
        ring.stop = &stop;
        // the ring's buffers live in the context's pool (kept for the next call, freed with the context): allocated by THIS thread, which
        // runs on the GPU's NUMA node, through bs_host_alloc on the context's device
        if (x->file_pool.size() < (size_t)ring_size) x->file_pool.resize((size_t)ring_size, {nullptr, 0});
        for (int k = 0; k < ring_size && rcs[c] == BS_OK; k++) {
            auto &slot = x->file_pool[(size_t)k];
            if (slot.second < cap) {
                if (slot.first) bs_host_free(slot.first);
                slot = {nullptr, 0};
                slot.first = static_cast<unsigned char *>(bs_host_alloc(x, cap));
                if (!slot.first) { rcs[c] = BS_ENOMEM; errs[c] = bs::error_message(); break; }
                slot.second = cap;
            }
            ring.bufs.push_back(slot.first);
        }
        if (rcs[c] != BS_OK) { stop_all(); return; }
        ring.start();
        struct Registered {   // other contexts' stop_all may wake this ring only while it is alive (declared after it: destroyed before it)
            std::vector<FileRing *> &all;
            std::mutex &m;
            int c;
            Registered(std::vector<FileRing *> &a, std::mutex &mm, int cc, FileRing *r) : all(a), m(mm), c(cc)
            {
                std::lock_guard<std::mutex> lk(m);
                all[(size_t)c] = r;
            }
            ~Registered()
            {
                std::lock_guard<std::mutex> lk(m);
                all[(size_t)c] = nullptr;
            }
        };
        int rc;
        {
            Registered reg(rings, rings_m, c, &ring);
            const PngSink sink{nullptr, nullptr, &ring};
            rc = run_share(x, cfgs, n_frames, bloom_strengths, bloom_dividers, nullptr, &sink, c, n_ctx);
            if (rc && rc != kCancelled) {
                errs[c] = bs::error_message();
                stop_all();
            }
        }
        ring.close();   // the writer finishes what it was handed and is joined (also on every error path: ~FileRing does the same)
        if (!ring.error.empty()) {   // its own file failed: that is this context's error, whatever the pipeline returned because of it
            rc = BS_EIO;
            errs[c] = ring.error;
            stop_all();
        }
        rcs[c] = rc;
        bs_files_stats_t &st = x->files_stats;
        st.files = ring.files;
        st.bytes = ring.bytes;
        st.wall_ms = (ring.t_last > 0 ? ring.t_last : now_ms()) - t0;
        st.writer_busy_ms = ring.busy_ms;
        st.buffer_wait_ms = ring.wait_ms;
        st.ring = ring_size;
        st.writer_threads = ring.threaded ? 1 : 0;
        st.numa_node_gpu = x->numa_node;
        int node = bs::numa_node_of_page(ring.bufs[0]);
        for (unsigned char *b : ring.bufs)
            if (bs::numa_node_of_page(b) != node) node = -2;   // (mixed)
        st.numa_node_buffers = node;
        st.threads_bound = ring.threaded ? (ring.writer_bound ? 1 : 0) : 0;
        on_unwind.armed = false;
    });
    if (thrown) return thrown;
    // the first context (in index order) with an error of its own decides; a context that was merely stopped has none
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c] && rcs[c] != kCancelled) return fail(rcs[c], errs[c]);
    for (int c = 0; c < n_ctx; c++)
        if (rcs[c] == kCancelled) return fail(BS_EINTERNAL, "bs_render_png_files: stopped without an error on record");
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_png_files"); }

int bs_render_split(bs_ctx *const *ctxs, int n_ctx, const bs_config *cfg, double *out_rgb, size_t out_doubles)
try {
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
    const int thrown = per_context(ctxs, n, [&](int c) {
        const int row0 = c * base + std::min(c, extra), row1 = row0 + base + (c < extra ? 1 : 0);
        rcs[c] = bs_render_rows(ctxs[c], cfg, row0, row1, out_rgb + (size_t)row0 * cfg->width * 3, (size_t)(row1 - row0) * cfg->width * 3);
        if (rcs[c]) errs[c] = bs::error_message();
    });
    if (thrown) return thrown;
    for (int c = 0; c < n; c++)
        if (rcs[c]) return fail(rcs[c], errs[c]);
    return BS_OK;
} catch (...) { return bs::abi_exception("bs_render_split"); }
