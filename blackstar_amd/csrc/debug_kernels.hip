// debug_kernels.hip -- kernels of the TEST HOOKS (libblackstar_gpu_debug.so; never in the product library): per-ray terminal records
// through the same trace_ray device function the frame kernel inlines, the correctly-rounded sqrt / divide check, the FP64 VALU
// issue-rate probe.  Compiled with the product's flags from the product's device header.
#include "../../include/blackstar_gpu_debug.h"
#include "trace_device.h"

namespace bs {
namespace {

// Test hook: trace an explicit list of traced-resolution pixels, one lane per listed ray.
template <bool FAST>
__global__ __launch_bounds__(kBlock) void trace_records_kernel(const TraceParams P, const int32_t *yx, size_t n_rays, bs_ray_record *out)
{
    __shared__ double s_lane[kLaneLdsDoubles];
    __shared__ int s_ints[3 * kBlock];
    const LaneLds lds(s_lane, s_ints);
    size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = k < n_rays;
    RayResult res;
    unsigned w_iters;
    trace_ray<FAST>(P, lds, live, live ? yx[2 * k] : 0, live ? yx[2 * k + 1] : 0, res, w_iters);
    if (!live) return;
    bs_ray_record r;
    for (int i = 0; i < 3; i++) { r.vel[i] = res.vel[i]; r.pos[i] = res.pos[i]; }
    for (int i = 0; i < 4; i++) r.rgba[i] = res.rgba[i];
    r.steps = res.steps; r.fate = res.fate; r.disk_hits = res.disk_hits; r.star_hits = res.star_hits;
    out[k] = r;
}

// Test hook: device sqrt / divide, to check that the f64 lowerings are correctly rounded.
__global__ void sqrt_div_kernel(const double *a, const double *b, size_t n, double *s, double *d, int bare)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (bare == 2) {  // raw hardware seeds (precision probe for FAST mode's Newton step)
        s[k] = __builtin_amdgcn_rsq(a[k]);
        d[k] = __builtin_amdgcn_rcp(b[k]);
    } else if (bare) {  // the scaling-free sequences the STRICT RK4 RHS uses
        s[k] = sqrt_rn(a[k]);
        d[k] = div_rn(a[k], b[k]);
    } else {     // hipcc's own lowering of sqrt and '/'
        s[k] = __builtin_sqrt(a[k]);
        d[k] = a[k] / b[k];
    }
}

// Roofline probe: NCH independent dependency chains per lane of one FP64 VALU instruction kind (32 instructions
// per lane per trip).  kind 0: v_fma_f64   1: v_mul_f64   2: v_add_f64   3: v_rsq_f64   4: v_rcp_f64, 8 chains
// (issue rate); kind 5/6/7: v_fma_f64 with 1/2/4 chains, kind 8: v_rsq_f64 with 1 chain (dependent latency when
// launched at one wave per SIMD).
template <int KIND, int NCH>
__global__ __launch_bounds__(256) void ubench_kernel(double *out, int iters, double a, double b)
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32 / NCH; u++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if constexpr (KIND == 0) x[i] = __builtin_fma(x[i], a, b);
                if constexpr (KIND == 1) x[i] = x[i] * a;
                if constexpr (KIND == 2) x[i] = x[i] + b;
                if constexpr (KIND == 3) x[i] = __builtin_amdgcn_rsq(x[i]);
                if constexpr (KIND == 4) x[i] = __builtin_amdgcn_rcp(x[i]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    if (s == 12345.678) out[0] = s;  // keep the chains live without a store on the common path
}

// More of the same for the question "is there a cheaper seed than v_rsq_f64 (16 cycles)?": kind 9 v_rsq_f32, kind 10 the round trip
// v_cvt_f32_f64 + v_cvt_f64_f32, kind 11 the whole candidate seed v_cvt_f32_f64 -> v_rsq_f32 -> v_cvt_f64_f32, kind 12 v_rsq_f64 with three
// independent v_fma_f64 behind each (does anything overlap with the transcendental?).  8 chains per lane, 32 / 32 / 96 / 128 instructions per trip.
// Kind 13: v_mfma_f64_4x4x4_4b_f64 alone (8 independent accumulators, 32 per trip); kind 14: each of them followed by three independent
// v_fma_f64 (128 per trip) -- does the DP matrix pipe run BESIDE the DP VALU on this chip, i.e. could it serve as a second adder for the
// uniform-scalar FMAs of a step?  (A measurement for the record: north_star rules MFMA out for this path, and a matrix instruction mixes
// the lanes of a row, which the free-running finished lanes of the stepping loop would poison with their inf / NaN.)
template <int KIND>
__global__ __launch_bounds__(256) void ubench_asm_kernel(double *out, int iters, double a, double b)
{
    double x[8], y[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = 1.0 + 1e-3 * (threadIdx.x + i); f[i] = (float)x[i]; y[i] = x[i] + 0.5; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (KIND == 9) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
                if constexpr (KIND == 10 && true) { if (u < 2) asm volatile("v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %0, %1" : "+v"(x[i]), "+v"(f[i])); }
                if constexpr (KIND == 11) asm volatile("v_cvt_f32_f64 %1, %0\n\tv_rsq_f32 %1, %1\n\tv_cvt_f64_f32 %0, %1" : "+v"(x[i]), "+v"(f[i]));
                if constexpr (KIND == 13) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                if constexpr (KIND == 14) {
                    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                    asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a), "v"(b));
                }
                if constexpr (KIND == 12) {
                    asm volatile("v_rsq_f64 %0, %0" : "+v"(x[i]));
                    asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a), "v"(b));
                }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // kinds 13 / 14: the matrix pipe's last results before a VALU reads them
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + (double)f[i] + y[i];
    if (s == 12345.678) out[0] = s;
}

}  // namespace

int launch_ubench(int kind, int blocks, int iters, double *d_out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)blocks), b(256);
    switch (kind) {
    case 0: hipLaunchKernelGGL((ubench_kernel<0, 8>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 1: hipLaunchKernelGGL((ubench_kernel<1, 8>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 2: hipLaunchKernelGGL((ubench_kernel<2, 8>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 3: hipLaunchKernelGGL((ubench_kernel<3, 8>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 4: hipLaunchKernelGGL((ubench_kernel<4, 8>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 5: hipLaunchKernelGGL((ubench_kernel<0, 1>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 6: hipLaunchKernelGGL((ubench_kernel<0, 2>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 7: hipLaunchKernelGGL((ubench_kernel<0, 4>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 8: hipLaunchKernelGGL((ubench_kernel<3, 1>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 9: hipLaunchKernelGGL((ubench_asm_kernel<9>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 10: hipLaunchKernelGGL((ubench_asm_kernel<10>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 11: hipLaunchKernelGGL((ubench_asm_kernel<11>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 12: hipLaunchKernelGGL((ubench_asm_kernel<12>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 13: hipLaunchKernelGGL((ubench_asm_kernel<13>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    case 14: hipLaunchKernelGGL((ubench_asm_kernel<14>), g, b, 0, s, d_out, iters, 0.999999, 1e-6); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_trace_records(const TraceParams &p, int mode, const int32_t *d_yx, size_t n_rays, bs_ray_record *d_out, void *stream)
{
    if (n_rays == 0) return 0;
    dim3 grid((unsigned)((n_rays + kBlock - 1) / kBlock));
    hipStream_t s = (hipStream_t)stream;
    if (mode == BS_MODE_FAST) hipLaunchKernelGGL(trace_records_kernel<true>, grid, dim3(kBlock), 0, s, p, d_yx, n_rays, d_out);
    else hipLaunchKernelGGL(trace_records_kernel<false>, grid, dim3(kBlock), 0, s, p, d_yx, n_rays, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sqrt_div(const double *d_a, const double *d_b, size_t n, double *d_sqrt, double *d_div, int bare, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(sqrt_div_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, n, d_sqrt, d_div, bare);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bs
