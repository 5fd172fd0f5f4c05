// post_kernels.hip -- the two steps that follow render in the reference's doRender (app/Main.hs:113-123):
//   bloom / boxBlur   src/ImageFilters.hs:28-86     
//   sRGB + toWord8    src/Raytracer.hs:23-32        
// Both are kept on the device so a frame can leave the GPU as 6.2 MB of RGB8 instead of 49.8 MB of f64.
//
// boxBlur is a RUNNING sum in the reference -- S <- (S + pix(x+r)) - pix(x-r), out = S/(2r+1) -- so each
// row (column) is a sequential floating-point chain; reproducing its bits means walking it in order.  The
// parallelism is across chains: one lane per (row, channel) for the horizontal sweep, one per (column,
// channel) for the vertical one.  Quirks preserved: the window is [x-r+1, x+r] (2r
// samples) but the normalisation is 1/(2r+1); out-of-range pixels read as 0; each pass is H then V with V
// reading the H result; 3 passes.
// Three implementations of one sweep, all bit-exact (blur_passes picks): box_blur_sweep_rot (the default: LDS-DMA loader,
// three chain wavefronts taking turns, four store wavefronts -- see its header for the measurements that shaped it), box_blur_sweep_lds (round 1:
// register-staged LDS ring; takes the sizes whose chain runs are not 16-byte aligned) and box_blur_sweep between two
// transposes (any size).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bs_internal.h"

namespace bs {
namespace {

// Sweep along the slow axis of a row-major [n][chains] array of doubles (chains = pixels_per_row * 3): lane k
// owns the chain { a[i][k] : i = 0..n-1 }.  Adjacent lanes touch adjacent doubles -> every load and store is a
// fully coalesced 512-B wave access.  The chain itself is sequential by definition (a running sum), so the
// only latency hiding is memory-level: the next kUnroll leading / trailing samples are fetched ahead of the
// add-subtract chain that consumes them.
constexpr int kUnroll = 16;

__global__ __launch_bounds__(64) void box_blur_sweep(const double *__restrict__ in, double *__restrict__ out, int chains, int n, int r, double norm)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= chains) return;
    const double *src = in + k;
    double *dst = out + k;
    const long stride = chains;
    // startVal = foldl1' add (pix <$> take r crds)                       (ImageFilters.hs:59)
    const int m = r < n ? r : n;
    double s = src[0];
    for (int i = 1; i < m; i++) s = s + src[(long)i * stride];
    int x = 0;
    for (; x + kUnroll <= n; x += kUnroll) {  // accumulate (:61-64), kUnroll samples per trip
        double lead[kUnroll], trail[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int xl = x + u + r, xt = x + u - r;
            lead[u] = (xl < n) ? src[(long)xl * stride] : 0.0;   // ixh / ixv: out of bounds -> black
            trail[u] = (xt >= 0) ? src[(long)xt * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            s = (s + lead[u]) - trail[u];
            dst[(long)(x + u) * stride] = norm * s;
        }
    }
    for (; x < n; x++) {
        double lead = (x + r < n) ? src[(long)(x + r) * stride] : 0.0;
        double trail = (x - r >= 0) ? src[(long)(x - r) * stride] : 0.0;
        s = (s + lead) - trail;
        dst[(long)x * stride] = norm * s;
    }
}

// LDS-staged variant of the same sweep (round 1; now the path for odd dimensions, windows up to r = 192): one workgroup owns kLC adjacent chains.
// Its INPUT is chain-major -- in[(p * n + row) * 3 + c] for chain (pixel p, channel c) -- and its OUTPUT row-major --
// out[(row * P + p) * 3 + c]: the horizontal sweep reads the image as it is (p = y, row = x) and writes it transposed,
// the vertical sweep reads that (p = x, row = y) and writes the image layout back, so no transpose kernels are needed:
// the loaders fetch a chain-pixel's T x 3 contiguous doubles at a time (1.5 KB coalesced) and scatter them into the
// ring, the consumer's 33 lanes write 33 contiguous doubles per row.
// The chain is inherently sequential, so ONE wavefront (lanes 0..kLC-1) walks it -- but it never waits on
// HBM: the other three wavefronts stream the rows it will need next into an LDS ring (each element is fetched
// from global memory once, as the leading sample, and re-read from the ring 2r rows later as the trailing one).
// Per tile of kLT rows: consumer processes rows [kT, kT+T) from the ring while the loaders fill the rows tile
// k+1 will lead with; one workgroup barrier per tile.  Ring depth kLR rows must cover 2r + 2T.
constexpr int kLP = 11;    // pixels (chain triples) per workgroup
constexpr int kLC = 3 * kLP;  // 33 chains per workgroup: whole RGB pixels; odd row stride in the ring
constexpr int kLT = 64;    // rows per tile
constexpr int kLR = 512;   // ring rows (power of two): 512 * 33 * 8 B = 132 KiB of LDS -> supports r <= (512 - 128) / 2 = 192

__global__ __launch_bounds__(256) void box_blur_sweep_lds(const double *__restrict__ in, double *__restrict__ out, int chains, int n, int r, double norm)
{
    // [kLR + 8][kLC]: rows 0..7 are mirrored at kLR..kLR+7, so a run of 8 consecutive rows never wraps and the
    // consumer can address it as one base + immediate offsets
    __shared__ double ring[(kLR + 8) * kLC];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * kLC;   // first chain of this workgroup
    const int p0 = blockIdx.x * kLP;   // first chain-pixel
    const int P = chains / 3;          // chain-pixels in the image
    const long stride = chains;        // output row stride
    const bool consumer = tid < 64;
    // loader thread j = tid - 64 (0..191) handles, for each of the kLP chain-pixels, element j of that pixel's T x 3
    // contiguous doubles: row = row_lo + j / 3, channel = j % 3
    constexpr int kPref = kLP;
    double pref[kPref];
    const int j = tid - 64, jr = j / 3, jc = j - 3 * jr;
    auto fetch = [&](int row_lo) {  // issue the global loads of tile rows [row_lo, row_lo + T) into registers
#pragma unroll
        for (int m = 0; m < kPref; m++) {
            const int row = row_lo + jr;
            pref[m] = (row < n && p0 + m < P) ? in[((long)(p0 + m) * n + row) * 3 + jc] : 0.0;
        }
    };
    auto commit = [&](int row_lo) {  // registers -> ring
        const int ri = (row_lo + jr) & (kLR - 1);
#pragma unroll
        for (int m = 0; m < kPref; m++) {
            ring[ri * kLC + 3 * m + jc] = pref[m];
            if (ri < 8) ring[(ri + kLR) * kLC + 3 * m + jc] = pref[m];
        }
    };
    // prologue: everything the first tile touches (rows [0, T + r)) straight into the ring, all 256 threads;
    // the loaders also start fetching what tile 1 will lead with.
    {   // eight loads in flight per thread, THEN their ring writes: one load -> wait -> write per trip exposed the full HBM
        // latency 18 times per sweep (the 25 us fixed cost of a sweep)
        constexpr int kU = 8;
        const int per_px = (kLT + r) * 3, total = per_px * kLP;
        for (int e0 = tid; e0 < total; e0 += 256 * kU) {
            double val[kU];
            int at[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int e = e0 + 256 * u;
                const int m = e / per_px, q = e - m * per_px, row = q / 3, c = q - 3 * row;
                val[u] = (e < total && row < n && p0 + m < P) ? in[((long)(p0 + m) * n + row) * 3 + c] : 0.0;
                at[u] = e < total ? (row & (kLR - 1)) * kLC + 3 * m + c : -1;
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {
                if (at[u] >= 0) {
                    ring[at[u]] = val[u];
                    if (at[u] < 8 * kLC) ring[at[u] + kLR * kLC] = val[u];  // rows 0..7 mirrored past the end
                }
            }
        }
    }
    if (!consumer) fetch(kLT + r);
    __syncthreads();
    const int lane = tid;  // consumer lanes 0..kLC-1 own one chain each
    const bool owner = consumer && lane < kLC && c0 + lane < chains;
    double s = 0.0;
    if (owner) {  // startVal = foldl1' add (pix <$> take r crds)   (ImageFilters.hs:59)
        const int m = r < n ? r : n;
        s = ring[lane];
        int i = 1;
        for (; i + 8 <= m; i += 8) {  // the eight ring reads first, then the eight (ordered) adds
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t8[u] = ring[((i + u) & (kLR - 1)) * kLC + lane];
#pragma unroll
            for (int u = 0; u < 8; u++) s = s + t8[u];
        }
        for (; i < m; i++) s = s + ring[(i & (kLR - 1)) * kLC + lane];
    }
    const int tiles = (n + kLT - 1) / kLT;
    for (int k = 0; k < tiles; k++) {
        if (consumer) {
            if (owner) {
                double *dst = out + (long)(k * kLT) * stride + c0 + lane;
                const int x_lo = k * kLT, x_hi = (k + 1) * kLT < n ? (k + 1) * kLT : n;
                const double *rl = ring + lane;
                if (x_lo - r >= 0 && x_hi - 1 + r < n && x_hi - x_lo == kLT) {
                    // interior tile: every leading and trailing sample exists -> no bounds selects; the ring reads of
                    // the NEXT 8 rows are issued before the add/subtract chain of the current 8 runs (accumulate, :61-64)
                    double la[8], ta[8], lb[8], tb[8];
                    const double *pl = rl + ((x_lo + r) & (kLR - 1)) * kLC, *pt = rl + ((x_lo - r) & (kLR - 1)) * kLC;
#pragma unroll
                    for (int u = 0; u < 8; u++) { la[u] = pl[u * kLC]; ta[u] = pt[u * kLC]; }
#pragma unroll 1
                    for (int x = x_lo; x < x_hi; x += 16) {
                        pl = rl + ((x + 8 + r) & (kLR - 1)) * kLC; pt = rl + ((x + 8 - r) & (kLR - 1)) * kLC;
#pragma unroll
                        for (int u = 0; u < 8; u++) { lb[u] = pl[u * kLC]; tb[u] = pt[u * kLC]; }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            s = (s + la[u]) - ta[u];
                            *dst = norm * s;
                            dst += stride;
                        }
                        if (x + 16 < x_hi) {
                            pl = rl + ((x + 16 + r) & (kLR - 1)) * kLC; pt = rl + ((x + 16 - r) & (kLR - 1)) * kLC;
#pragma unroll
                            for (int u = 0; u < 8; u++) { la[u] = pl[u * kLC]; ta[u] = pt[u * kLC]; }
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            s = (s + lb[u]) - tb[u];
                            *dst = norm * s;
                            dst += stride;
                        }
                    }
                } else {  // edge tiles: out-of-range samples read as black (ixh / ixv)
                    for (int x = x_lo; x < x_hi; x++) {
                        const int xl = x + r, xt = x - r;
                        const double lead = (xl < n) ? rl[(xl & (kLR - 1)) * kLC] : 0.0;
                        const double trail = (xt >= 0) ? rl[(xt & (kLR - 1)) * kLC] : 0.0;
                        s = (s + lead) - trail;
                        *dst = norm * s;
                        dst += stride;
                    }
                }
            }
        } else {
            // software pipeline: what was fetched during the previous tile lands in the ring now (tile k+1 leads with it),
            // and the loads for tile k+2 are issued -- their latency hides behind the consumer's next tile.
            commit((k + 1) * kLT + r);
            fetch((k + 2) * kLT + r);
        }
        // Tile barrier WITHOUT a memory fence: only the ring (LDS) is handed over between wavefronts, so wait for this
        // wave's LDS traffic and rendezvous.  __syncthreads() would also drain vmcnt -- i.e. wait for the loads just
        // issued for tile k+2 and for the consumer's stores -- which is exactly the latency this pipeline hides.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---- box blur sweep, LDS-DMA version (the default path) --------------------------------------------------------------
// Same layouts as box_blur_sweep_lds: INPUT chain-major in[(p * n + row) * 3 + c], OUTPUT row-major out[(row * P + p) * 3 + c],
// so an H sweep followed by a V sweep needs no transpose.  What changed is who does what, driven by measurements of the old
// kernel (profiles/r01_rgb8_kernel_stats.csv: 1.78 TB/s) and the CDNA4 price list: a CU ingests ~25 GB/s from HBM whatever it
// does, so ALL 256 CUs have to stream, each a 1/256 slice of the chains.
//   * One workgroup per CU: px = ceil(P / #CU) adjacent chain-pixels (3 px chains, <= 63), i.e. 216 / 240 workgroups for the
//     H / V sweep of a 1080p frame instead of 99 / 175.  152 KiB of static LDS keeps it at one workgroup per CU.
//   * The LOADER wavefront moves 1 KiB (64 lanes x 16 B) of one chain-pixel's contiguous run per `global_load_lds_dwordx4`
//     straight into that pixel's LDS ring -- no VGPR staging, no ds_write pass, and the prefetch depth (Dp phases = Dp x 3 KiB
//     per pixel, 48-72 KiB per workgroup) is limited by the ring, not by registers.  Counted `s_waitcnt vmcnt(N)` leaves the
//     younger batches in flight across the block barriers (raw s_barrier: a fence would drain them).
//   * CHAIN wavefronts: one lane per chain walks the reference's running sum S <- (S + pix(x+r)) - pix(x-r) in order
//     (bit-exact), operands from one ring offset + immediates (slot 0 of each ring is mirrored behind the last slot so a
//     block never wraps).
//   * blockIdx -> chain group is XCD-aware: workgroup b runs on XCD b % 8, and XCD x gets a CONTIGUOUS range of groups, so the
//     partial 128-B lines two neighbouring groups write (a group's row is 3 px doubles = 120-192 B) meet in one L2.
// A phase = 128 rows = exactly 3 chunks per pixel.  Batch p of chunks is what phase p needs beyond phase p-1:
// [3p + Lr, 3p + 3 + Lr) with Lr = ceil(24 r / 1024) (batch 0: [0, 3 + Lr)).  During phase k the loader issues batch k + Dp into
// the slots of chunks that died with phase k-1 (ring of S = 3 (1 + Dp) + 2 Lr slots), then waits for batch k+1.
constexpr int kDmaChunk = 1024;      // bytes per global_load_lds_dwordx4 wave-instruction
constexpr int kDmaPhaseRows = 128;   // 128 rows x 24 B = 3 chunks
constexpr int kDmaLds = 152 * 1024;  // static LDS of the sweep kernel (of 160 KiB per CU): one workgroup per CU, and the plan's budget
constexpr int kDmaMaxPx = 21;        // 63 chains = one consumer wavefront

struct SweepPlan {
    int px;       // chain-pixels per workgroup
    int S;        // ring slots (1 KiB each) per pixel, + 1 mirror slot
    int Dp;       // phases of prefetch in flight
    int Lr;       // chunks the +-r window reaches beyond a phase
    int stride;   // bytes between the rings of two pixels: (S + 1) * 1024 + 32 (the 32 skews the banks of neighbouring pixels)
    int groups;   // ceil(P / px)
    int per_xcd;  // ceil(groups / 8); grid = 8 * per_xcd
    int lds_bytes;  // LDS the plan needs: rings + hand-off tiles (<= kDmaLds, the kernel's static allocation)
#ifdef BS_SWEEP_PROBE  // scripts/sweep_probe.hip only: switch parts of the kernel off, report shader / wall clocks of workgroup 0
    int dbg;                     // see box_blur_sweep_rot
    unsigned long long *clocks;  // per wavefront of group 0: [2w] shader clocks (chain: in compute intervals; others: up to the barriers), [2w+1] in the loop
#endif
};

// One LDS-DMA wave-instruction: lane i's 16 bytes at g land at LDS byte address lds_addr + 16 i (lds_addr wave-uniform, via M0).
// Issued as inline assembly ON PURPOSE: with the builtin (__builtin_amdgcn_global_load_lds) anywhere in a kernel, hipcc's waitcnt
// insertion treats lgkmcnt as out of order for the whole loop and puts `s_waitcnt lgkmcnt(0)` in front of every use of an LDS
// read -- the chain wavefront then stalls a full LDS round trip per batch instead of running its reads 8 rows ahead (seen in the
// ISA; a 20-line reproducer is in scripts/probe/README).  The loader counts its own vmcnt (wait_vmcnt_le), nobody else needs to.
__device__ __forceinline__ void dma_1k(const unsigned char *g, const unsigned char *l)
{
    const unsigned lds_addr = (unsigned)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char *)l);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory");
}

// The same with the global address as a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: a loader that walks
// chunks and pixels then needs only scalar arithmetic per wave-instruction (about 7 SALU + the DMA itself).
__device__ __forceinline__ void dma_1k_uniform(const unsigned char *base_uniform, unsigned lane_off, const unsigned char *l)
{
    const unsigned lds_addr = (unsigned)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char *)l);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(base_uniform), "v"(lane_off), "s"(lds_addr) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
#define BS_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : n) {
        BS_W(0) BS_W(1) BS_W(2) BS_W(3) BS_W(4) BS_W(5) BS_W(6) BS_W(7) BS_W(8) BS_W(9) BS_W(10) BS_W(11) BS_W(12) BS_W(13) BS_W(14) BS_W(15)
        BS_W(16) BS_W(17) BS_W(18) BS_W(19) BS_W(20) BS_W(21) BS_W(22) BS_W(23) BS_W(24) BS_W(25) BS_W(26) BS_W(27) BS_W(28) BS_W(29) BS_W(30) BS_W(31)
        BS_W(32) BS_W(33) BS_W(34) BS_W(35) BS_W(36) BS_W(37) BS_W(38) BS_W(39) BS_W(40) BS_W(41) BS_W(42) BS_W(43) BS_W(44) BS_W(45) BS_W(46) BS_W(47)
        BS_W(48) BS_W(49) BS_W(50) BS_W(51) BS_W(52) BS_W(53) BS_W(54) BS_W(55) BS_W(56) BS_W(57) BS_W(58) BS_W(59) BS_W(60) BS_W(61) BS_W(62)
    default: break;  // >= 63: the counter saturates at 63, so by the time the issuing loop has finished the awaited batch has landed
    }
#undef BS_W
}

// ---- what a LONE wavefront costs (the measurements that shaped the kernel below) ---------------------------------------
// scripts/probe/sweep_probe.hip -> profiles/r02_sweep_probe*.txt, on intermediate versions of this kernel: the loader alone
// streams a sweep in 20-24 us, but a lone wavefront on a SIMD is slow at everything: a dependent v_add_f64 issues every ~10.5
// clocks (two per row: 21), an LDS instruction holds its issue slot for ~10-20 clocks (operand reads 17 per row, result
// writes 15), a global store for ~20 (64 lanes x 16 B of address + data), scalar loop control is not free either -- and none
// of it can be hidden behind another wavefront, because a chain IS one lane of one wavefront and its time is rows x clocks per
// row whatever the lane count.  Versions and their chain-wavefront clocks per row: consumer doing everything (round-1
// structure, and a first LDS-DMA version) 92-98; chain wavefront + store wavefronts 54-58; forcing the LDS instructions into the
// add latency bubbles with sched_group_barrier 58 (worse than the compiler's clustering, 48: isolated LDS instructions of a lone
// wavefront do not pipeline); three rotating chain wavefronts (below) 26 inside the compute intervals.
constexpr int kBlkRows = 32;      // rows per block: the unit the chain wavefronts take turns on; 4 blocks per loader phase
constexpr int kStoreWaves = 4;    // STORE wavefronts, kBlkRows / kStoreWaves rows of a block each
constexpr int kTileColBytes = (kBlkRows + 1) * 8;  // hand-off tile: [chain][row], 33 doubles per chain (odd: conflict-free both ways)
static_assert(kDmaPhaseRows == 4 * kBlkRows && kDmaPhaseRows * 24 == 3 * kDmaChunk, "a loader phase is 4 blocks = 3 chunks per pixel");

// ---- rotating version: THREE chain wavefronts take turns (the default path) -------------------------------------------
// With ONE chain wavefront (and store wavefronts for the rest) the chain wavefront still spends 54-58 clocks per row, of which only 21 are the two dependent adds;
// the rest is its own LDS traffic (operand reads 17, result writes 15), which a lone in-order wavefront cannot overlap with
// the adds.  But the chain only hands ONE number per lane from row to row.  So three wavefronts take turns, block by block (32
// rows): in interval j the wavefront j mod 3 does nothing but the 64 dependent adds of block j -- operands already in its
// registers, each result kept in the register of the leading sample it consumed, S taken from and returned to a 512-byte LDS
// slot -- while the other two do their LDS work off the critical path: the one that finished block j-1 writes its 32 results
// to the hand-off tile and fetches the first operands of block j+2, the one that finished block j-2 fetches the rest of the
// operands of block j+1.  Critical path per row: the adds plus 1/32 of a barrier and of an LDS round trip for S (measured:
// 26 clocks per row inside the compute intervals).
//   waves 0-2 CHAIN (rotating)   wave 3 LOADER (LDS-DMA)   waves 4-7 STORE (tile of block j-2 -> HBM in interval j)
// A lone wavefront needs ~10 clocks per instruction of ANY kind, so an interval is as long as its busiest wavefront's
// instruction count: the loader and the store wavefronts are written for few instructions (running chunk state and scalar
// address arithmetic in the loader, per-lane constants hoisted out of the store loop).  Measured and rejected: no store
// wavefronts, the chain wavefronts writing their rows to HBM in their off intervals (one wavefront per SIMD) -- 16 global stores
// per interval cost more than the tile writes (287 vs 266 us for the bloom of a 1080p frame).
constexpr int kChainWaves = 3;
constexpr int kRotThreads = 64 * (kChainWaves + 1 + kStoreWaves);

__global__ __launch_bounds__(kRotThreads) void box_blur_sweep_rot(const double *__restrict__ in, double *__restrict__ out, int P, int n, int r, double norm,
                                                                   const SweepPlan pl)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDmaLds];
    // Every kernel argument is consumed HERE, by all wavefronts: a scalar load whose result only one role uses would stay
    // "pending" on the other roles' paths, and the compiler then treats lgkmcnt as out of order there.
    asm volatile("" ::"s"(in), "s"(out), "s"(P), "s"(n), "s"(r), "s"(norm), "s"(pl.px), "s"(pl.S), "s"(pl.Dp), "s"(pl.Lr), "s"(pl.stride),
                 "s"(pl.groups), "s"(pl.per_xcd));
    const int g = (int)(blockIdx.x & 7u) * pl.per_xcd + (int)(blockIdx.x >> 3);  // XCD-contiguous ranges of chain groups
    if (g >= pl.groups) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63u);
    const int p0 = g * pl.px;
    const int npx = (P - p0) < pl.px ? (P - p0) : pl.px;
    const int S = pl.S, Dp = pl.Dp, Lr = pl.Lr, stride = pl.stride;
    const long run_bytes = (long)n * 24;
    const int nchunks = (int)((run_bytes + kDmaChunk - 1) / kDmaChunk);
    const int blocks = (n + kBlkRows - 1) / kBlkRows;
    const int tile_bytes = 3 * pl.px * kTileColBytes;        // one hand-off tile: [3 px chains][kBlkRows (+1 pad) rows]
    unsigned char *tiles = lds + pl.px * stride;             // two of them behind the rings ...
    double *s_hand = reinterpret_cast<double *>(tiles + 2 * tile_bytes);  // ... then the running sums in transit, one per chain

    // ---- loader (wave 3): chunks are issued in increasing order, one call = the next chunk of every pixel of the group ----
    const unsigned char *gin_u = reinterpret_cast<const unsigned char *>(in) + (long)p0 * run_bytes;  // wave-uniform: pixel p0, chunk 0
    const long group_bytes = (long)(P - p0) * run_bytes;  // from gin_u to the end of the array
    const unsigned lane16 = (unsigned)lane * 16u;
    int ld_chunk = 0, ld_slot = 0;  // next chunk to issue and its ring slot
    auto issue_next = [&]() -> int {
        if (ld_chunk >= nchunks) return 0;
        const unsigned char *gsrc = gin_u + (long)ld_chunk * kDmaChunk;
        unsigned char *l = lds + ld_slot * kDmaChunk;
        // only the LAST chunk of a run can reach past it (into the next pixels' rows, which nobody looks at) -- and past the end of
        // the ARRAY there is nothing to read: `safe` pixels' chunks lie wholly inside it, the others are issued lane by lane
        int safe = npx;
        const long chunk_end = (long)(ld_chunk + 1) * kDmaChunk;
        if (ld_chunk + 1 >= nchunks) {
            const long fit = chunk_end <= group_bytes ? (group_bytes - chunk_end) / run_bytes + 1 : 0;
            safe = fit < npx ? (int)fit : npx;
        }
        for (int m = 0; m < safe; m++, gsrc += run_bytes, l += stride) {
            dma_1k_uniform(gsrc, lane16, l);
            if (ld_slot == 0) dma_1k_uniform(gsrc, lane16, l + S * kDmaChunk);  // mirror of slot 0 behind the last slot
        }
        for (int m = safe; m < npx; m++, gsrc += run_bytes, l += stride) {  // lane 0 is in range, so the instruction issues
            const long left = group_bytes - ((long)ld_chunk * kDmaChunk + (long)m * run_bytes);
            if ((long)lane16 + 16 <= left) {
                dma_1k(gsrc + lane16, l);
                if (ld_slot == 0) dma_1k(gsrc + lane16, l + S * kDmaChunk);
            }
        }
        const int cnt = ld_slot == 0 ? 2 * npx : npx;
        ld_chunk++;
        ld_slot = ld_slot + 1 == S ? 0 : ld_slot + 1;
        return cnt;
    };
    int v1 = 0, v2 = 0;  // wave-instruction counts of the (up to two) batches in flight BEHIND the one that is awaited next

    // ---- chain (waves 0..2) ----
    const int ncol = 3 * npx;
    const bool active = lane < ncol;
    const int lm = active ? lane / 3 : 0, lc = active ? lane - 3 * (lane / 3) : 0;
    const unsigned char *lbase = lds + lm * stride + 8 * lc;
    const unsigned ring = (unsigned)S * kDmaChunk;
    // ring offsets of rows x + r and x - r for x = the first row of this wavefront's NEXT block to fetch (block `wave` at first)
    unsigned offL = (unsigned)((24l * r + 24l * kBlkRows * wave) % ring);
    unsigned offT = (unsigned)(((ring - (24l * r) % ring) + 24l * kBlkRows * wave) % ring);
    const unsigned adv = (unsigned)((24l * kBlkRows * kChainWaves) % ring);
    // operands of the block this wavefront computes next; after the compute L holds the block's RESULTS (each result takes the
    // register of the leading sample it consumed) until they are written to the tile, one interval later
    double L[kBlkRows], T[kBlkRows];
    double s = 0.0;
    auto fetch = [&](int u0, int u1) {  // operands of rows [u0, u1) of the block at (offL, offT)
        const unsigned char *pL = lbase + offL, *pT = lbase + offT;
#pragma unroll
        for (int u = 0; u < kBlkRows; u++) {
            if (u >= u0 && u < u1) {
                L[u] = *reinterpret_cast<const double *>(pL + 24 * u);
                T[u] = *reinterpret_cast<const double *>(pT + 24 * u);
            }
        }
    };
    auto advance = [&]() {
        offL += adv; offL = offL >= ring ? offL - ring : offL;
        offT += adv; offT = offT >= ring ? offT - ring : offT;
    };

    // ---- store (waves 4..7): 8 rows of every block each.  A lane takes TWO adjacent chains of one row (one ds_read2_b64, one
    // 16-byte store), so a store instruction covers 64 / ceil(3 px / 2) rows: all 8 at px = 5, five at px = 8.  (One store
    // wavefront for all 32 rows, the other three leaving their SIMDs to the chain wavefronts, was measured: 291 vs 226 us.) ----
    constexpr int kRowsPerStore = kBlkRows / kStoreWaves;
    const int cpairs = (ncol + 1) / 2;     // column pairs per row (the last one is a single column when 3 px is odd)
    const int G = 64 / cpairs;             // rows one store instruction covers
    const int rsub = lane / cpairs, cp = lane - rsub * cpairs;
    const bool has2 = 2 * cp + 1 < ncol;   // this lane's second column exists
    const size_t ostride = (size_t)P * 3;  // doubles between two output rows
    const int mw = wave - (kChainWaves + 1);
    const int row0 = kRowsPerStore * (mw > 0 ? mw : 0) + rsub;                                      // this lane's first row within a block
    const int iters = (kRowsPerStore + G - 1) / G;                                    // store instructions per block (wave-uniform)
    const int my_iters = rsub < G ? (kRowsPerStore - rsub + G - 1) / G : 0;           // ... in which this lane has a row
    const unsigned tile_lane = (unsigned)(2 * cp * kTileColBytes + 8 * row0);         // this lane's first byte within a tile
    const size_t step = (size_t)G * ostride;                                          // doubles between the rows of two store instructions
    double *dst_blk = out + ((size_t)row0 * P + p0) * 3 + 2 * cp;                     // this lane's first output of block 0
    struct __attribute__((packed, aligned(8))) Pair { double a, b; };                 // 16-byte store at 8-byte alignment

    // batch p = what phase p needs beyond phase p-1: chunks [3p + Lr, 3p + 3 + Lr), batch 0 from chunk 0.  Batch 0 first, alone:
    // the start value and the chain wavefronts' first operands need nothing else, and batches 1 .. Dp-1 are issued while they run.
    if (wave == kChainWaves) {
        while (ld_chunk < 3 + Lr && ld_chunk < nchunks) issue_next();
        wait_vmcnt_le(0);  // batch 0 has landed
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    if (wave == kChainWaves) {
        int c[3] = {0, 0, 0};
        for (int p = 1; p < Dp; p++) {
            int cnt = 0;
            const int hi = 3 * p + 3 + Lr;
            while (ld_chunk < hi && ld_chunk < nchunks) cnt += issue_next();
            c[p - 1] = cnt;
        }
        v1 = c[1]; v2 = c[2];  // (c[0] is batch 1, the one awaited first: its own count is never needed)
    } else if (wave < kChainWaves && active) {
        if (wave == 0) {  // startVal = foldl1' add (pix <$> take r crds)   (ImageFilters.hs:59)
            const int mr = r < n ? r : n;
            s = *reinterpret_cast<const double *>(lbase);
            int i = 1;
            for (; i + 8 <= mr; i += 8) {
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = *reinterpret_cast<const double *>(lbase + 24 * (i + u));
#pragma unroll
                for (int u = 0; u < 8; u++) s = s + t8[u];
            }
            for (; i < mr; i++) s = s + *reinterpret_cast<const double *>(lbase + 24 * i);
            s_hand[lane] = s;
        }
        if (wave < blocks) { fetch(0, kBlkRows); advance(); }  // blocks 0, 1, 2 lie in loader phase 0: landed
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // Interval `it` (it = 0 .. blocks + 1): chain wave it % 3 computes block it; STORE writes block it - 2; a raw barrier WITHOUT a
    // memory fence closes the interval (only LDS -- rings, tiles, s_hand -- is handed over between the wavefronts; a fence would
    // wait for the LDS-DMA batches in flight and for the stores).  Every wavefront executes exactly blocks + 2 of them.  The
    // roles are separate straight-line loops, so that the 128 operand registers of a chain wavefront are updated in ONE
    // sequence (a role switch inside a common loop made the register allocator spill them).
    const int n_int = blocks + 2;
#ifdef BS_SWEEP_PROBE  // dbg bits: 16 = per-interval timers (they cost a few hundred clocks per interval); 1 = STORE wavefronts idle,
                       // 2 = loader idle, 4 = chain wavefronts skip their off-interval LDS work (tile writes, operand fetches),
                       // 128 = STORE wavefronts skip their global stores, 256 = ... their tile reads
    const int dbg = pl.dbg;
    unsigned long long probe_work = 0, probe_start = __builtin_readcyclecounter();
#define BS_INTERVAL_END()                                                      \
    do {                                                                       \
        if (dbg & 16) {                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
            probe_work += __builtin_readcyclecounter() - probe_t;              \
        }                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        \
        if (dbg & 16) probe_t = __builtin_readcyclecounter();                  \
    } while (0)
    unsigned long long probe_t = __builtin_readcyclecounter(), probe_compute = 0;
#define BS_DBG(bit) (dbg & (bit))
#else
#define BS_DBG(bit) false
#define BS_INTERVAL_END() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
    if (wave < kChainWaves) {
        int it = 0;
        for (; it < wave && it < n_int; it++) BS_INTERVAL_END();  // lead-in: this wavefront's first block is block `wave`
        for (int b = wave; it < n_int; b += kChainWaves) {
            // interval b: COMPUTE block b -- nothing but the chain
            if (active && b < blocks) {
                const int x0 = b * kBlkRows;
                s = s_hand[lane];
                if (x0 >= r && x0 + (kBlkRows - 1) + r < n) {
#pragma unroll
                    for (int u = 0; u < kBlkRows; u++) {
                        s = (s + L[u]) - T[u];  // accumulate (ImageFilters.hs:61-64)
                        L[u] = s;
                    }
                } else {  // image edges: out-of-range samples are black (ixh / ixv); rows past the end are computed but never stored
#pragma unroll
                    for (int u = 0; u < kBlkRows; u++) {
                        const int x = x0 + u;
                        const double lead = (x + r < n) ? L[u] : 0.0, trail = (x - r >= 0) ? T[u] : 0.0;
                        s = (s + lead) - trail;
                        L[u] = s;
                    }
                }
                s_hand[lane] = s;
            }
#ifdef BS_SWEEP_PROBE
            if (dbg & 16) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                probe_compute += __builtin_readcyclecounter() - probe_t;
            }
#endif
            BS_INTERVAL_END();
            if (++it >= n_int) break;
            // interval b + 1: results of block b -> tile; the first operands of block b + 3
            if (active && !BS_DBG(4)) {
                if (b < blocks) {
                    unsigned char *tile = tiles + (b & 1) * tile_bytes + lane * kTileColBytes;
#pragma unroll
                    for (int u = 0; u < kBlkRows; u++) *reinterpret_cast<double *>(tile + 8 * u) = L[u];
                }
                if (b + kChainWaves < blocks) fetch(0, 8);
            }
            BS_INTERVAL_END();
            if (++it >= n_int) break;
            // interval b + 2: the rest of the operands of block b + 3
            if (active && b + kChainWaves < blocks && !BS_DBG(4)) { fetch(8, kBlkRows); advance(); }
            BS_INTERVAL_END();
            ++it;
        }
    } else if (wave == kChainWaves) {
        // loader phase k = 4 intervals = 128 rows = 3 chunks per pixel.  The chain wavefronts fetch the operands of block b in
        // intervals b - 2 and b - 1, so batch k+1 (what phase k+1 needs beyond phase k) has to have landed when interval
        // 4k + 2 opens: interval 1 of the phase waits for it.  Batch k + Dp goes into the slots of chunks that died with phase k-1.
        for (int it = 0; it < n_int; it++) {
            const int q = it & 3;
            int knew = 0;
            if (BS_DBG(2)) {
            } else if (Dp >= 2) {  // one chunk per interval (a burst would hold up the barrier): the batch is not needed for another phase
                if (q < 3) knew = issue_next();
            } else if (q == 0) {  // Dp = 1: batch k+1 itself, needed when interval 4k + 2 opens
                knew = issue_next();
                knew += issue_next();
            } else if (q == 1) {
                knew = issue_next();
            }
            if (Dp == 2) v1 += knew; else if (Dp >= 3) v2 += knew;
            if (q == 1) {
                // everything older than the batches behind batch k+1 has landed.  With Dp >= 2 the newest batch is still being
                // issued (its third chunk follows in interval 2): its count so far is exactly what is outstanding behind batch k+1.
                wait_vmcnt_le(v1 + v2);
            }
            if (q == 3 && Dp >= 2) { v1 = v2; v2 = 0; }  // the phase's batch is complete: shift the window
            BS_INTERVAL_END();
        }
    } else {
        // STORE, software-pipelined across the barrier: in interval `it` the tile of block it - 2 is READ into one register set
        // while the values read in the previous interval (block it - 3) are multiplied and written to HBM from the other -- the
        // wavefront reaches the barrier after max(LDS round trip, store issue), not their sum, and its stores never pace an
        // interval.  (The tile may be overwritten from interval it + 1 on; by then it is in registers.)
        constexpr int kMaxIt = 4;  // store instructions per block and wavefront: ceil(8 / G) with G >= 2
        struct Held { double a[kMaxIt], b[kMaxIt]; double *dst; int lim; };
        Held hx, hy;
        hx.lim = hy.lim = 0; hx.dst = hy.dst = dst_blk;
#pragma unroll
        for (int q = 0; q < kMaxIt; q++) hx.a[q] = hx.b[q] = hy.a[q] = hy.b[q] = 0.0;
        auto read_block = [&](Held &h, int it) {  // tile of block it - 2 -> registers (lanes without a row read harmless bytes)
            h.lim = 0;
            if (it >= 2 && it - 2 < blocks && !BS_DBG(1)) {
                const int jm = it - 2;
                const unsigned char *tp = tiles + (jm & 1) * tile_bytes + tile_lane;
                int lim = my_iters;  // store instructions in which this lane has a row that exists
                if ((jm + 1) * kBlkRows > n) {  // (wave-uniform) the image ends inside this block
                    const int rows_left = n - jm * kBlkRows - row0;
                    const int cap = rows_left <= 0 ? 0 : (rows_left + G - 1) / G;
                    lim = lim < cap ? lim : cap;
                }
                h.lim = lim;
                h.dst = dst_blk;
                dst_blk += (size_t)kBlkRows * ostride;
#pragma unroll
                for (int q = 0; q < kMaxIt; q++) {
                    if (q < iters && !BS_DBG(256)) {
                        h.a[q] = *reinterpret_cast<const double *>(tp + (8 * G) * q);
                        h.b[q] = *reinterpret_cast<const double *>(tp + (8 * G) * q + kTileColBytes);
                    }
                }
            }
        };
        auto write_block = [&](const Held &h) {  // mul normFactor newRGB (ImageFilters.hs:62-63) -> HBM
#pragma unroll
            for (int q = 0; q < kMaxIt; q++) {
                if (q < h.lim && !BS_DBG(128)) {
                    double *d = h.dst + q * step;
                    if (has2) {
                        Pair pr;
                        pr.a = norm * h.a[q];
                        pr.b = norm * h.b[q];
                        *reinterpret_cast<Pair *>(d) = pr;
                    } else {
                        *d = norm * h.a[q];
                    }
                }
            }
        };
        // The barrier's own lgkmcnt(0) has retired the reads, but the compiler cannot see that and would wait for them -- AND for
        // the next block's reads issued in front of the stores -- before the first multiply: passing the registers through an
        // empty asm makes the (by then free) wait happen here, so the next interval's stores go out while its reads are in flight.
        auto landed = [&](Held &h) {
#pragma unroll
            for (int q = 0; q < kMaxIt; q++) asm volatile("" : "+v"(h.a[q]), "+v"(h.b[q]));
        };
        int it = 0;
        for (; it + 1 < n_int; it += 2) {
            read_block(hx, it);
            write_block(hy);
            BS_INTERVAL_END();
            landed(hx);
            read_block(hy, it + 1);
            write_block(hx);
            BS_INTERVAL_END();
            landed(hy);
        }
        if (it < n_int) {  // odd number of intervals
            read_block(hx, it);
            write_block(hy);
            BS_INTERVAL_END();
            landed(hx);
            write_block(hx);
        } else {
            write_block(hy);
        }
    }
#undef BS_INTERVAL_END
#undef BS_DBG
#ifdef BS_SWEEP_PROBE
    if (g == 0 && lane == 0 && pl.clocks) {  // per wavefront of group 0: clocks up to the barriers (chain: in COMPUTE intervals only) / in the loop
        pl.clocks[2 * wave] = wave < kChainWaves ? probe_compute : probe_work;
        pl.clocks[2 * wave + 1] = __builtin_readcyclecounter() - probe_start;
    }
#endif
}

// [rows][cols] pixels of 3 doubles -> [cols][rows]; 32x32-pixel tiles through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void transpose_rgb(const double *__restrict__ in, double *__restrict__ out, int rows, int cols)
{
    __shared__ double tile[32][32 * 3 + 1];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * 96; i += 256) {
        int rr = i / 96, cc = i % 96;  // cc indexes doubles within the tile row (pixel cc/3, channel cc%3)
        int r = r0 + rr, c = c0 + cc / 3;
        if (r < rows && c < cols) tile[rr][cc] = in[((size_t)r * cols + c0) * 3 + cc];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 96; i += 256) {
        int cc = i / 96, rr3 = i % 96;  // output row = input column c0+cc; within it pixel rr3/3 (= input row), channel rr3%3
        int rr = rr3 / 3, ch = rr3 % 3;
        int r = r0 + rr, c = c0 + cc;
        if (r < rows && c < cols) out[((size_t)c * rows + r0) * 3 + rr3] = tile[rr][cc * 3 + ch];
    }
}

// bloom: img + strength * blurred   (ImageFilters.hs:84-86)
__global__ void bloom_combine(const double *img, const double *blurred, double *out, size_t n, double strength)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = img[i] + strength * blurred[i];
}

// supersample (ImageFilters.hs:88-97) as a standalone op (render fuses it into the trace kernel's epilogue):
// out(y,x) = 0.25 * (((p(2y,2x) + p(2y+1,2x)) + p(2y,2x+1)) + p(2y+1,2x+1)); output (h2 div 2) x (w2 div 2).
__global__ void supersample_kernel(const double *in, double *out, int w2, int h, int w)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)h * w * 3) return;
    int c = (int)(i % 3);
    size_t px = i / 3;
    int x = (int)(px % w), y = (int)(px / w);
    const double *p = in + ((size_t)(2 * y) * w2 + 2 * x) * 3 + c;
    double a = p[0], b = p[(size_t)w2 * 3], cc = p[3], d = p[(size_t)w2 * 3 + 3];
    out[i] = 0.25 * (((a + b) + cc) + d);
}

// writeImg's pixel map: toWord8 . fmap sRGB   (Raytracer.hs:23-32); toWord8 = round-half-even (255 * clamp01 x).
// The map x -> byte is monotone, so it is fully described by 255 thresholds T[k] = the smallest double that maps to a byte
// >= k.  The HOST finds them once per context by bisection over the bit patterns of x with the reference's own formula
// evaluated with the host's libm `pow` (host_math.cpp: srgb8_thresholds) -- so the device reproduces the CPU's bytes
// EXACTLY (no device pow, no 1-LSB flips next to a .5 boundary) with a few float operations per value: an f32 estimate
// of the byte (v_log_f32 / v_exp_f32, good to ~1e-6 of 255) corrected against the neighbouring thresholds.
// T has 257 entries: T[0] = -inf, T[1..255], T[256] = +inf.  NaN -> 0 (what the x86 cast of rint(NaN) gives the reference
// restatement; every comparison with NaN is false).
__device__ __forceinline__ unsigned srgb8_byte(double x, const double *T)
{
    const float xf = (float)x;
    float yf = xf < 0.0031308f ? 12.92f * xf : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (1.0f / 2.4f)) - 0.055f;
    yf = yf > 0.0f ? yf : 0.0f;   // also catches NaN
    yf = yf < 1.0f ? yf : 1.0f;
    int k = (int)(255.0f * yf + 0.5f);
    // the estimate is off by at most one; the loops are bounded by the table and run zero or one times
#pragma unroll 1
    while (k < 255 && x >= T[k + 1]) k++;
#pragma unroll 1
    while (k > 0 && x < T[k]) k--;
    return (unsigned)k;
}

__global__ __launch_bounds__(256) void srgb8_kernel(const double *__restrict__ in, unsigned char *__restrict__ out, size_t n, const double *__restrict__ table)
{
    __shared__ double T[257];
    for (int i = threadIdx.x; i < 257; i += 256) T[i] = table[i];
    __syncthreads();
    // four values per lane -> one 32-bit store
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 4 <= n && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        unsigned v = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) v |= srgb8_byte(in[i4 + u], T) << (8 * u);
        *reinterpret_cast<unsigned *>(out + i4) = v;
    } else {
        for (size_t i = i4; i < n && i < i4 + 4; i++) out[i] = (unsigned char)srgb8_byte(in[i], T);
    }
}

// bloom's last step fused with writeImg's pixel map: toWord8 (sRGB (img + strength * blurred))   (ImageFilters.hs:84-86 then
// Raytracer.hs:31-32) -- the combined f64 image is never written (49.8 MB write + read and one launch less per frame).
__global__ __launch_bounds__(256) void bloom_combine_srgb8(const double *__restrict__ img, const double *__restrict__ blurred, unsigned char *__restrict__ out,
                                                            size_t n, double strength, const double *__restrict__ table)
{
    __shared__ double T[257];
    for (int i = threadIdx.x; i < 257; i += 256) T[i] = table[i];
    __syncthreads();
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 4 <= n && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        unsigned v = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) v |= srgb8_byte(img[i4 + u] + strength * blurred[i4 + u], T) << (8 * u);
        *reinterpret_cast<unsigned *>(out + i4) = v;
    } else {
        for (size_t i = i4; i < n && i < i4 + 4; i++) out[i] = (unsigned char)srgb8_byte(img[i] + strength * blurred[i], T);
    }
}

}  // namespace

// Geometry of the LDS-DMA sweep for P chain-pixels of n rows, window r, on a chip of n_cu CUs; false = not applicable.
// single_round: only accept a plan whose workgroups all run at once (one per CU) -- with the LDS this kernel takes, a second
// round of workgroups costs a whole extra sweep time.
static bool plan_dma_sweep(const void *in, int P, int n, int r, int n_cu, SweepPlan &pl, bool single_round, int lds_budget = kDmaLds)
{
    if ((n & 1) || r < 1 || P < 1) return false;  // every chain-pixel's run must start 16-B aligned: n * 24 B a multiple of 16
    if (reinterpret_cast<uintptr_t>(in) & 15) return false;
    const long lr = (24l * r + kDmaChunk - 1) / kDmaChunk;
    if (lr > 64) return false;
    auto stride_of = [&](int dp) { return (3 * (1 + dp) + 2 * (int)lr + 1) * kDmaChunk + 32; };
    const int tile_bytes = 2 * 3 * kTileColBytes;  // per pixel: two hand-off tiles of 3 chains x (kBlkRows + 1) doubles
    const int px0 = std::max(1, std::min(kDmaMaxPx, (P + n_cu - 1) / n_cu));
    int px = 0, dp = 0;
    for (int d : {3, 2, 1}) {  // the px that covers the chip with one round of workgroups, as deep a prefetch as the LDS allows
        if ((long)px0 * (stride_of(d) + tile_bytes) <= lds_budget) { px = px0; dp = d; break; }
    }
    if (!px && !single_round) {  // fewer pixels per workgroup: more workgroups than CUs, i.e. several ROUNDS of workgroups per CU -- then what
        // counts is the CU-time of a sweep, not a workgroup's latency: the most chain-pixels the LDS takes (prefetch depth 1) first.  Measured
        // (profiles/r03_post_partition_ab.txt, "fat"): r = 192 on 8 CUs 5.0 against 5.9 ms, 1080p on 16 CUs 1.91 against 2.01, 4K on all 256 CUs
        // 0.80-0.87 either way.
        for (int d : {1, 2}) {
            const int fit = (int)(lds_budget / (stride_of(d) + tile_bytes));
            if (fit >= 1) { px = std::min(px0, fit); dp = d; break; }
        }
    }
    if (!px) return false;
    pl.px = px;
    pl.Dp = dp;
    pl.Lr = (int)lr;
    pl.S = 3 * (1 + dp) + 2 * (int)lr;
    pl.stride = stride_of(dp);
    pl.groups = (P + px - 1) / px;
    pl.per_xcd = (pl.groups + 7) / 8;
    pl.lds_bytes = px * (pl.stride + tile_bytes);
    return !single_round || pl.groups <= n_cu;
}

// Sweep paths: "dma" = the LDS-DMA kernel with rotating chain wavefronts (box_blur_sweep_rot), "lds" = the round-1 register-staged LDS ring (windows up to r = 192),
// "direct" = transpose + register-prefetch sweep (any size).  Default: dma wherever its plan exists (even dimensions, 16-B aligned
// input; measured faster than lds at 720p / 1080p / 2160p and windows up to r = 192: scripts/bloom_ab.py), else lds when its
// ring covers the window (odd sizes), else direct.  env BLACKSTAR_BLOOM_PATH=dma|lds|direct forces one where it applies (A/B and
// tests), read per call.
static int bloom_path()
{
    const char *e = std::getenv("BLACKSTAR_BLOOM_PATH");
    if (!e) return 0;
    if (!std::strcmp(e, "dma")) return 3;
    if (!std::strcmp(e, "lds")) return 1;
    if (!std::strcmp(e, "direct")) return 2;
    return 0;
}

// The three boxBlur passes (ImageFilters.hs:66-77): src -> ... -> d_b.  d_a, d_b: scratch of w*h*3 doubles, never aliasing src.
static void blur_passes(const double *src, double *d_a, double *d_b, int w, int h, int r, int n_cu, hipStream_t s)
{
    const double norm = 1 / (2 * (double)r + 1);
    const dim3 tgrid_hw((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32));  // transposing an h x w image
    const dim3 tgrid_wh((unsigned)((h + 31) / 32), (unsigned)((w + 31) / 32));  // transposing a  w x h image
    const int path = bloom_path();
    const bool lds_fits = 2 * r + 2 * kLT <= kLR;

    for (int pass = 0; pass < 3; pass++) {
        SweepPlan ph, pv;
        bool dma = false;
        if (path == 0 || path == 3)
            dma = plan_dma_sweep(src, h, w, r, n_cu, ph, false, kDmaLds - 1024) && plan_dma_sweep(d_a, w, h, r, n_cu, pv, false, kDmaLds - 1024);
        // H: image layout (h x w) -> transposed layout (w x h); V: transposed -> image layout.  No transpose kernels.
        if (dma) {
            hipLaunchKernelGGL(box_blur_sweep_rot, dim3((unsigned)(8 * ph.per_xcd)), dim3(kRotThreads), 0, s, src, d_a, h, w, r, norm, ph);
            hipLaunchKernelGGL(box_blur_sweep_rot, dim3((unsigned)(8 * pv.per_xcd)), dim3(kRotThreads), 0, s, (const double *)d_a, d_b, w, h, r, norm, pv);
        } else if (path != 2 && lds_fits) {
            hipLaunchKernelGGL(box_blur_sweep_lds, dim3((unsigned)((h + kLP - 1) / kLP)), dim3(256), 0, s, src, d_a, h * 3, w, r, norm);
            hipLaunchKernelGGL(box_blur_sweep_lds, dim3((unsigned)((w + kLP - 1) / kLP)), dim3(256), 0, s, (const double *)d_a, d_b, w * 3, h, r, norm);
        } else {
            // any size (odd dimensions, windows wider than the LDS): transpose, sweep along the slow axis with register prefetch, transpose back
            hipLaunchKernelGGL(transpose_rgb, tgrid_hw, dim3(256), 0, s, src, d_a, h, w);                  // src (h x w) -> A (w x h)
            hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((h * 3 + 63) / 64)), dim3(64), 0, s, (const double *)d_a, d_b, h * 3, w, r, norm);
            hipLaunchKernelGGL(transpose_rgb, tgrid_wh, dim3(256), 0, s, (const double *)d_b, d_a, w, h);  // B (w x h) -> A (h x w)
            hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((w * 3 + 63) / 64)), dim3(64), 0, s, (const double *)d_a, d_b, w * 3, h, r, norm);
        }
        src = d_b;  // pass p+1 reads B while writing A, then A -> B: A and B never alias
    }
}

int launch_bloom(const double *d_in, double *d_out, double *d_a, double *d_b, int w, int h, double strength, int divider, int n_cu, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int r = w / divider;  // boxBlur (w `div` divider) 3 img   (ImageFilters.hs:83)
    const size_t n = (size_t)w * h * 3;
    blur_passes(d_in, d_a, d_b, w, h, r, n_cu, s);
    hipLaunchKernelGGL(bloom_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (const double *)d_b, d_out, n, strength);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_bloom_srgb8(const double *d_in, unsigned char *d_out_u8, double *d_a, double *d_b, int w, int h, double strength, int divider, int n_cu,
                       const double *d_table, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int r = w / divider;
    const size_t n = (size_t)w * h * 3;
    blur_passes(d_in, d_a, d_b, w, h, r, n_cu, s);
    hipLaunchKernelGGL(bloom_combine_srgb8, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, d_in, (const double *)d_b, d_out_u8, n, strength, d_table);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_supersample(const double *d_in, double *d_out, int w2, int h2, void *stream)
{
    const int w = w2 / 2, h = h2 / 2;
    const size_t n = (size_t)w * h * 3;
    if (n == 0) return 0;
    hipLaunchKernelGGL(supersample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, w2, h, w);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_srgb8(const double *d_in, unsigned char *d_out, size_t n, const double *d_table, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(srgb8_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n, d_table);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bs
