// sweep_probe.hip -- where does a box-blur sweep's time go?  Builds the product's own box_blur_sweep_rot with BS_SWEEP_PROBE
// (per-wavefront clocks of workgroup 0: chain wavefronts in their compute intervals, the others up to the barriers) and times the
// H and V sweep of a frame.  NOTE: the timers (s_memtime, two per interval) inflate every interval by a few hundred clocks --
// compare variants with scripts/bloom_ab.py on the product build, use this for attribution only.
// Build (cross-compiles):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/probe/sweep_probe.hip -o scripts/probe/sweep_probe
#define BS_SWEEP_PROBE 1
#include "../../blackstar_amd/csrc/post_kernels.hip"

#include <cstdio>
#include <vector>

using namespace bs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int w = argc > 1 ? std::atoi(argv[1]) : 1920, h = argc > 2 ? std::atoi(argv[2]) : 1080, r = argc > 3 ? std::atoi(argv[3]) : 76;
    const size_t n = (size_t)w * h * 3;
    double *a = nullptr, *b = nullptr;
    unsigned long long *clk = nullptr;
    CK(hipMalloc((void **)&a, n * 8 + 4096));
    CK(hipMalloc((void **)&b, n * 8 + 4096));
    CK(hipMalloc((void **)&clk, 256));
    std::vector<double> host(n);
    for (size_t i = 0; i < n; i++) host[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(a, host.data(), n * 8, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double norm = 1.0 / (2 * r + 1);
    struct Sweep { const char *name; int P, n; } sweeps[2] = {{"H", h, w}, {"V", w, h}};
    for (const Sweep &sw : sweeps) {
        {   // the rotating three-chain-wavefront kernel (the product path)
            SweepPlan p8;
            if (plan_dma_sweep(a, sw.P, sw.n, r, n_cu, p8, false, kDmaLds - 1024)) {
                for (int dbg : {0, 1, 128, 256, 2, 4, 5, 7, 16}) {
                    p8.dbg = dbg; p8.clocks = clk;
                    float best = 1e9f;
                    for (int it = 0; it < 6; it++) {
                        CK(hipEventRecord(e0, nullptr));
                        hipLaunchKernelGGL(box_blur_sweep_rot, dim3((unsigned)(8 * p8.per_xcd)), dim3(kRotThreads), 0, nullptr, (const double *)a, b, sw.P, sw.n, r, norm, p8);
                        CK(hipEventRecord(e1, nullptr));
                        CK(hipEventSynchronize(e1));
                        float ms = 0;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    unsigned long long c16[16];
                    CK(hipMemcpy(c16, clk, 128, hipMemcpyDeviceToHost));
                    std::printf("%s sweep rot dbg=%2d: px=%d Dp=%d S=%d groups=%d: %.1f us, loop %.1f clk/row", sw.name, dbg, p8.px, p8.Dp, p8.S, p8.groups, best * 1e3,
                                (double)c16[1] / sw.n);
                    if (dbg & 16) {
                        std::printf("; with interval timers, work clocks per row:");
                        for (int wv = 0; wv < 8; wv++) std::printf(" w%d %.1f", wv, (double)c16[2 * wv] / sw.n);
                    }
                    std::printf("\n");
                }
            }
        }
    }
    return 0;
}
