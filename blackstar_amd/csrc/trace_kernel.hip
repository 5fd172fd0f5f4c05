// trace_kernel.hip -- gfx950 (CDNA4) geodesic trace: the frame kernel (persistent wavefronts, one lane per ray) and the batched
// starLookup kernel.  The per-ray device functions are in trace_device.h (see its header comment for the reference file:line map).
#include "trace_device.h"

namespace bs {
namespace {

#ifdef BS_TRACE_PROBE
// Timeline probe (scripts/trace_timeline.py; never in the product build): per wavefront of the frame kernel, wall-clock stamps
// (100 MHz, comparable across CUs) at kernel entry, first tile start, start and end of the last tile and at exit, the number of
// tiles it traced, the iterations of the last one, and where it ran (HW_ID: SIMD, CU, SE; XCC_ID).
constexpr int kProbeWords = 8, kProbeWaves = 8192;
__device__ unsigned long long g_trace_probe[kProbeWords * kProbeWaves];
#endif

// Frame kernel: PERSISTENT wavefronts.  The grid is sized to fill the chip once (P.grid_blocks workgroups of 4
// wavefronts, <= 4 per CU); every wavefront independently pulls 8x8-pixel tiles of traced rays off a device-wide
// counter until the frame is done -- no workgroup barrier at all, no per-tile dispatch, one flush of the statistics per
// wavefront.  (Star data is read straight from the L2-resident direction grid; nothing is staged in LDS up front.)
// With supersampling the four rays of an output pixel sit in four adjacent lanes (a quad), in the order
// p(2y,2x), p(2y+1,2x), p(2y,2x+1), p(2y+1,2x+1) of ImageFilters.hs:94-96, and are reduced with lane
// shuffles, so only the h x w image is ever written.
//
// Why stagger: every tile costs nearly the same (lane efficiency 0.9987; step counts vary by a few percent
// across the frame), so wavefronts that start together stay in phase and all four waves of a SIMD reach
// their latency-bound tail (star lookup, shading, image write, next tile's setup) at the same time, leaving the
// f64 pipe idle (PMC: VALU busy 91 % with stars vs 96 % without).  Delaying the first tile of the wave in
// SIMD slot k by k/4 of a tile time keeps the four phases apart for the rest of the frame.
// How a wavefront's statistics reach the launch slot's counters when it runs out of tiles (BS_EXIT_STATS, an A/B knob):
//   2 (the product)  the four wavefronts of a workgroup add theirs up in LDS behind one barrier and seven lanes issue ONE set of atomics
//   0 (rounds 1-5)   every wavefront issues its own: 4096 x 7 device-scope atomics on ONE 64-byte line -- the line the tile queue's head
//                    lives in -- in the last 0.3 ms of the launch.  Round 6 (scripts/launch_cost_probe.py, profiles/r06_exit_stats_ab.txt):
//                    the C3 frame 4.25 -> 4.13 ms with 2, and 4.07 -> 4.06 for a build with no statistics at all (1): the whole of it
//   1                none (probe only: bs_stats reads zeros)
#ifndef BS_EXIT_STATS
#define BS_EXIT_STATS 2
#endif
template <bool FAST>
__global__ __launch_bounds__(kBlock, BS_MIN_WAVES) void trace_frame_kernel(const TraceParams P)
{
    __shared__ double s_lane[kLaneLdsDoubles];
    __shared__ int s_ints[3 * kBlock];
    __shared__ unsigned long long s_stats[6 * kBlock];
    const LaneLds lds(s_lane, s_ints);

    const int lane = threadIdx.x & 63;
#if defined(BS_PAD_NOPS) && BS_PAD_NOPS > 0
    // Shifts every instruction behind it by 4 bytes per s_nop, executed once per wavefront.  Where the kernel's code lands modulo 32 bytes is
    // worth 1.6-2.3 % of the C3 frame (profiles/r06_code_alignment_ab.txt: offsets 24, 28, 0, 4 good, 8-20 bad; what rounds 2-5 booked as "extra
    // scalar state in the rare path costs 1-2.5 % although the hot path's ISA is unchanged").  The Makefile's PAD picks the offset;
    // scripts/alignment_sweep.sh measures all eight.
#define BS_STR2(x) #x
#define BS_STR(x) BS_STR2(x)
    asm volatile(".rept " BS_STR(BS_PAD_NOPS) "\n s_nop 0\n .endr");
#endif
#ifdef BS_TRACE_PROBE
    const unsigned long long probe_t0 = wall_clock64();
    unsigned long long probe_t1 = 0, probe_t2 = 0, probe_ts = 0;
    unsigned probe_tiles = 0, probe_last_iters = 0;
#endif
    int lx, ly;
    if (P.ss) {
        int q = lane >> 2, sub = lane & 3;
        lx = (q & 3) * 2 + (sub >> 1);
        ly = (q >> 2) * 2 + (sub & 1);
    } else {
        lx = lane & 7;
        ly = lane >> 3;
    }
    const int tiles_x = (P.wt + 7) >> 3;
    const int n_tiles = tiles_x * ((P.band_t1 - P.band_t0 + 7) >> 3);  // the band's tiles (a whole frame: band = [0, ht))

    // phase stagger (performance only): workgroups b, b+#CU, b+2#CU, ... are the ones observed to share a CU
    const int slot = (int)(blockIdx.x / (unsigned)P.blocks_per_slot);
    if (P.stagger_cycles > 0) {
        for (int c = 0; c < slot * P.stagger_cycles; c += 64 * 100) __builtin_amdgcn_s_sleep(100);
    }
    // A wavefront pops its next tile when it has traced the current one -- not before it, as rounds 1-5 did to hide the pop's round trip (~2 us).
    // A tile popped ahead sits untouched while its owner traces: 35 us for the wavefront its SIMD favours, but under oldest-first arbitration
    // 0.15 / 0.9 / 4 ms for the three younger ones (C3: 27, 5 and 1-2 tiles per wavefront and frame) -- and the tiles still held like that when the
    // queue runs dry are started after everybody else has finished: 12 wavefronts working for the last 80 us of a 4.2 ms frame, one of them on a
    // tile it had owned for a millisecond (scripts/trace_timeline.py -> profiles/r06_late_pop_timeline.txt).  One binary, the knob alone, two
    // boxes (profiles/r06_late_pop_ab.txt): everybody late (P.late_pop_slot 0, the product) C3 -0.4 ... -0.7 %, default.yaml at 1080p -3.4 %;
    // only slots 1-3 late: -0.2 % / -4.1 %; only 2-3: -0.2 % / -1.2 %.  (Round 2 had measured exposed pops as a LOSS of 1.4-2.3 % and round 5's
    // "late pops" chose by the INDEX of the tile being started, which a young wavefront popped a millisecond earlier: both were read against
    // builds whose loop had landed differently -- EXPERIMENTS.md 6.6.)
    const bool late_pop = slot >= P.late_pop_slot;

    // per-lane statistics summed over this wave's tiles live in LDS (six more registers held across trace_ray spill), as 64-bit
    // words: touched once per tile, never in the stepping loop, and a lane's step total is unbounded in the tiles it traces
    // (bs_set_max_steps x tiles per lane passes 2^32 on large frames of capped rays)
    unsigned long long *stat = s_stats + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 6; k++) stat[k * kBlock] = 0ull;
    unsigned long long a_iters = 0;  // wave-uniform
    // Over-fetching past the end of the queue is harmless: indices >= n_tiles just end the loop.
    // The FIRST tile needs no atomic (P.queue_base != 0, render.cpp): wavefront g of the grid takes tile g and the queue hands out the tiles
    // from queue_base = the number of wavefronts on.  Rounds 1-5 started every launch with up to 4096 pops on one address before anything
    // was traced: 25-45 us -- 2 % of the reference's default.yaml at 1080p, 15 % of a 640 x 360 frame, 0.9 % of the C3 frame
    // (profiles/r06_static_first_tile_ab.txt: one binary, BLACKSTAR_STATIC_FIRST_BELOW alone).
    const int queue_base = P.queue_base;
    int next_tile = (int)blockIdx.x * (kBlock / 64) + (int)(threadIdx.x >> 6);
    if (queue_base == 0) {
        next_tile = 0;
        if (lane == 0) next_tile = (int)atomicAdd(&P.counters[7], 1ull);
    }
#ifdef BS_TRACE_PROBE
    probe_t1 = wall_clock64();
#endif
    for (;;) {
        const int tile = __builtin_amdgcn_readfirstlane(next_tile);
        if (tile >= n_tiles) break;
        if (!late_pop && lane == 0) next_tile = queue_base + (int)atomicAdd(&P.counters[7], 1ull);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int xi = tx * 8 + lx, yb = ty * 8 + ly, yi = P.band_t0 + yb;  // yb: traced row within the band (band_t0 is even with supersampling)
        const bool inb = xi < P.wt && yi < P.band_t1;

        RayResult res;
        unsigned w_iters;
#ifdef BS_TRACE_PROBE
        probe_ts = wall_clock64();
#endif
        trace_ray<FAST>(P, lds, inb, yi, xi, res, w_iters);
#ifdef BS_TRACE_PROBE
        probe_last_iters = w_iters;
#endif

        if (P.ss) {
            const int base = lane & ~3;
            double o[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double a = __shfl(res.rgba[c], base + 0, 64);
                double b = __shfl(res.rgba[c], base + 1, 64);
                double cc = __shfl(res.rgba[c], base + 2, 64);
                double dd = __shfl(res.rgba[c], base + 3, 64);
                o[c] = 0.25 * (((a + b) + cc) + dd);
            }
            if (inb && (lane & 3) == 0) {
                double *dst = P.out + ((size_t)(yb >> 1) * P.out_w + (xi >> 1)) * 3;
                dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
            }
        } else if (inb) {
            double *dst = P.out + ((size_t)yb * P.out_w + xi) * 3;
            dst[0] = res.rgba[0]; dst[1] = res.rgba[1]; dst[2] = res.rgba[2];  // dropAlpha
        }
        stat[0 * kBlock] += (unsigned long long)(unsigned)res.steps;
        stat[1 * kBlock] += res.fate == 2 ? 1ull : 0ull;
        stat[2 * kBlock] += res.fate == 0 ? 1ull : 0ull;
        stat[3 * kBlock] += res.fate == 1 ? 1ull : 0ull;
        stat[4 * kBlock] += (unsigned long long)(unsigned)res.disk_hits;
        stat[5 * kBlock] += (unsigned long long)(unsigned)res.star_hits;
        a_iters += w_iters;
        if (late_pop && lane == 0) next_tile = queue_base + (int)atomicAdd(&P.counters[7], 1ull);
#ifdef BS_TRACE_PROBE
        probe_t2 = wall_clock64();
        probe_tiles++;
#endif
    }
    const unsigned long long s_steps = wave_sum(stat[0 * kBlock]), s_cap = wave_sum(stat[1 * kBlock]), s_hor = wave_sum(stat[2 * kBlock]),
                             s_esc = wave_sum(stat[3 * kBlock]), s_disk = wave_sum(stat[4 * kBlock]), s_star = wave_sum(stat[5 * kBlock]);
#if BS_EXIT_STATS == 1
    if (lane == 0 && s_steps == 0xFFFFFFFFFFFFFFFFull) atomicAdd(&P.counters[0], s_steps + s_cap + s_hor + s_esc + s_disk + s_star + a_iters);
#elif BS_EXIT_STATS == 2
    {
        __shared__ unsigned long long s_exit[7 * (kBlock / 64)];
        const int wv = threadIdx.x >> 6;
        if (lane == 0) {
            unsigned long long *e = s_exit + 7 * wv;
            e[0] = s_steps; e[1] = s_cap; e[2] = s_hor; e[3] = s_esc; e[4] = s_disk; e[5] = s_star; e[6] = a_iters;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            unsigned long long v = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) v += s_exit[7 * w + threadIdx.x];
            if (v) atomicAdd(&P.counters[threadIdx.x], v);
        }
    }
#else
    if (lane == 0) {
        atomicAdd(&P.counters[6], a_iters);
        atomicAdd(&P.counters[0], s_steps);
        if (s_cap) atomicAdd(&P.counters[1], s_cap);
        if (s_hor) atomicAdd(&P.counters[2], s_hor);
        if (s_esc) atomicAdd(&P.counters[3], s_esc);
        if (s_disk) atomicAdd(&P.counters[4], s_disk);
        if (s_star) atomicAdd(&P.counters[5], s_star);
    }
#endif
#ifdef BS_TRACE_PROBE
    const unsigned gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (lane == 0 && gw < (unsigned)kProbeWaves) {
        unsigned long long *q = g_trace_probe + (size_t)gw * kProbeWords;
        q[0] = probe_t0; q[1] = probe_t1; q[2] = probe_t2; q[3] = wall_clock64(); q[4] = probe_tiles;
        q[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID, all 32 bits
        q[6] = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) | ((unsigned long long)probe_last_iters << 8);  // XCC_ID[3:0]; iterations of the last tile
        q[7] = probe_ts;  // start of the last tile
    }
#endif
}

// starLookup over a batch of directions (same device function as the trace kernel's escape branch).
__global__ __launch_bounds__(kBlock) void star_lookup_kernel(const TraceParams P, const double *dirs, size_t n, double *rgb, int32_t *hits)
{
    __shared__ double s_lane[2 * kHitSlots * kBlock];
    size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    double r, g, b;
    int h = star_lookup(P, s_lane + threadIdx.x, dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2], r, g, b);
    rgb[3 * k] = r; rgb[3 * k + 1] = g; rgb[3 * k + 2] = b;
    if (hits) hits[k] = h;
}

}  // namespace

#ifdef BS_TRACE_PROBE
extern "C" int bs_debug_trace_probe(unsigned long long *out, int n_words)
{
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n_words < kProbeWords * kProbeWaves ? n_words : kProbeWords * kProbeWaves);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace_probe), bytes) == hipSuccess ? 0 : -1;
}
#endif

int launch_trace(const TraceParams &p, int mode, void *stream)
{
    dim3 grid((unsigned)p.grid_blocks);
    hipStream_t s = (hipStream_t)stream;
    if (mode == BS_MODE_FAST) hipLaunchKernelGGL(trace_frame_kernel<true>, grid, dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL(trace_frame_kernel<false>, grid, dim3(kBlock), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_star_lookup(const TraceParams &p, const double *d_dirs, size_t n, double *d_rgb, int32_t *d_hits, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(star_lookup_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, p, d_dirs, n,
                       d_rgb, d_hits);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bs
