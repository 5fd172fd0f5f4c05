// post_kernels.hip -- the two steps that follow render in the reference's doRender (app/Main.hs:113-123):
//   bloom / boxBlur   src/ImageFilters.hs:28-86     
//   sRGB + toWord8    src/Raytracer.hs:23-32        
// Both are HBM-bound byte/float streaming, kept on the device so a frame can leave the GPU as 6.2 MB of RGB8
// instead of 49.8 MB of f64.
//
// boxBlur is a RUNNING sum in the reference -- S <- (S + pix(x+r)) - pix(x-r), out = S/(2r+1) -- so each
// row (column) is a sequential floating-point chain; reproducing its bits means walking it in order.  The
// parallelism is across chains: one lane per (row, channel) for the horizontal sweep, one per (column,
// channel) for the vertical one.  Both run as a sweep along the slow axis of a row-major array (the
// horizontal one on a transposed copy) so that adjacent lanes touch adjacent doubles.  Quirks preserved: the window is [x-r+1, x+r] (2r
// samples) but the normalisation is 1/(2r+1); out-of-range pixels read as 0; each pass is H then V with V
// reading the H result; 3 passes.
#include <hip/hip_runtime.h>

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

// LDS-staged variant of the same sweep (used whenever the window fits): one workgroup owns kLC adjacent chains.
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
static_assert(true, "");
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

// writeImg's pixel map: toWord8 . fmap sRGB   (Raytracer.hs:23-32); toWord8 = round-half-even (255 * clamp01 x)
__global__ void srgb8_kernel(const double *in, unsigned char *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = in[i];
    double y = (x < 0.0031308) ? 12.92 * x : (1 + 0.055) * pow(x, 1.0 / 2.4) - 0.055;
    y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);  // NaN falls through both compares; rint(NaN)->0 below
    out[i] = (unsigned char)(int)rint(255.0 * y);
}

}  // namespace

int launch_bloom(const double *d_in, double *d_out, double *d_a, double *d_b, int w, int h, double strength, int divider, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int r = w / divider;  // boxBlur (w `div` divider) 3 img   (ImageFilters.hs:83)
    const double norm = 1 / (2 * (double)r + 1);
    const size_t n = (size_t)w * h * 3;
    const dim3 tgrid_hw((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32));  // transposing an h x w image
    const dim3 tgrid_wh((unsigned)((h + 31) / 32), (unsigned)((w + 31) / 32));  // transposing a  w x h image
    const double *src = d_in;
    const bool staged = 2 * r + 2 * kLT <= kLR;  // the LDS ring covers the window
    for (int pass = 0; pass < 3; pass++) {
        if (staged) {
            // H: image layout (h x w) -> transposed layout (w x h); V: transposed -> image layout.  No transpose kernels.
            hipLaunchKernelGGL(box_blur_sweep_lds, dim3((unsigned)((h + kLP - 1) / kLP)), dim3(256), 0, s, src, d_a, h * 3, w, r, norm);
            hipLaunchKernelGGL(box_blur_sweep_lds, dim3((unsigned)((w + kLP - 1) / kLP)), dim3(256), 0, s, (const double *)d_a, d_b, w * 3, h, r, norm);
        } else {
            // window wider than the ring: transpose, sweep along the slow axis with register prefetch, transpose back
            hipLaunchKernelGGL(transpose_rgb, tgrid_hw, dim3(256), 0, s, src, d_a, h, w);                  // src (h x w) -> A (w x h)
            hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((h * 3 + 63) / 64)), dim3(64), 0, s, (const double *)d_a, d_b, h * 3, w, r, norm);
            hipLaunchKernelGGL(transpose_rgb, tgrid_wh, dim3(256), 0, s, (const double *)d_b, d_a, w, h);  // B (w x h) -> A (h x w)
            hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((w * 3 + 63) / 64)), dim3(64), 0, s, (const double *)d_a, d_b, w * 3, h, r, norm);
        }
        src = d_b;
    }
    // NOTE: pass p+1 reads B while writing A -- A and B never alias, so this is safe.
    hipLaunchKernelGGL(bloom_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (const double *)d_b, d_out, n, strength);
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

int launch_srgb8(const double *d_in, unsigned char *d_out, size_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(srgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bs
