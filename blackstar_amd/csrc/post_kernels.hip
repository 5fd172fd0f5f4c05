// post_kernels.hip -- the two steps that follow render in the reference's doRender (app/Main.hs:113-123):
//   bloom / boxBlur   src/ImageFilters.hs:28-86     
//   sRGB + toWord8    src/Raytracer.hs:23-32        
// Both are HBM-bound byte/float streaming, kept on the device so a frame can leave the GPU as 6.2 MB of RGB8
// instead of 49.8 MB of f64.
//
// boxBlur is a RUNNING sum in the reference -- S <- (S + pix(x+r)) - pix(x-r), out = S/(2r+1) -- so each
// row (column) is a sequential floating-point chain; reproducing its bits means walking it in order.  The
// parallelism is across chains: one lane per (row, channel) for the horizontal sweep, one per (column,
// channel) for the vertical one.  Quirks preserved: the window is [x-r+1, x+r] (2r
// samples) but the normalisation is 1/(2r+1); out-of-range pixels read as 0; each pass is H then V with V
// reading the H result; 3 passes.
#include <hip/hip_runtime.h>

#include "bs_internal.h"

namespace bs {
namespace {

// One chain = `n` samples `stride` doubles apart starting at in + base (same addressing for out).
__global__ __launch_bounds__(256) void box_blur_sweep(const double *in, double *out, int n_chains, int n, long chain_stride, long stride, int chan,
                                                      int r, double norm)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_chains * chan) return;
    long base = (long)(k / chan) * chain_stride + (k % chan);
    const double *src = in + base;
    double *dst = out + base;
    // startVal = foldl1' add (pix <$> take r crds)                       (ImageFilters.hs:59)
    int m = r < n ? r : n;
    double s = src[0];
    for (int i = 1; i < m; i++) s = s + src[(long)i * stride];
    for (int x = 0; x < n; x++) {  // accumulate (:61-64)
        double lead = (x + r < n) ? src[(long)(x + r) * stride] : 0.0;   // ixh / ixv: out of bounds -> black
        double trail = (x - r >= 0) ? src[(long)(x - r) * stride] : 0.0;
        s = (s + lead) - trail;
        dst[(long)x * stride] = norm * s;
    }
}

// bloom: img + strength * blurred   (ImageFilters.hs:84-86)
__global__ void bloom_combine(const double *img, const double *blurred, double *out, size_t n, double strength)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = img[i] + strength * blurred[i];
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
    const double *src = d_in;
    for (int pass = 0; pass < 3; pass++) {
        // horizontal: one chain per row, samples 3 doubles apart; reads src, writes A
        hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((h * 3 + 255) / 256)), dim3(256), 0, s, src, d_a, h, w, (long)w * 3, 3L, 3, r, norm);
        // vertical: one chain per column, samples one row apart; reads A, writes B
        hipLaunchKernelGGL(box_blur_sweep, dim3((unsigned)((w * 3 + 255) / 256)), dim3(256), 0, s, (const double *)d_a, d_b, w, h, 3L, (long)w * 3, 3,
                           r, norm);
        src = d_b;
    }
    hipLaunchKernelGGL(bloom_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (const double *)d_b, d_out, n, strength);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_srgb8(const double *d_in, unsigned char *d_out, size_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(srgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bs
