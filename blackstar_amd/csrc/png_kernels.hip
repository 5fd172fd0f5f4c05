// png_kernels.hip -- writeImg's file format on the device (src/Raytracer.hs:23-32: `writeImage path . A.map (toWord8 . fmap sRGB)`; app/Main.hs:119-123;
// SURVEY.md 8f-2): the RGB8 frame the sRGB8 kernel leaves in HBM becomes a complete PNG file without visiting the host, so a frame
// crosses PCIe as ~1 MB of file instead of 6.2 MB of pixels and no host core spends 0.1-0.25 s per frame in zlib (the reference's
// writer, and this package's before: 25-60x the time the GPU needs to render the frame).
//
// Four kernels per frame, all enqueue-only:
//   png_choose_filter   one workgroup per row: the scanline filter with the smallest sum of |residuals|
//   png_encode_blocks   one workgroup (256 threads) per 8 KiB of the filtered stream: the phase program of png_block.h (filter, tokenise,
//                       Huffman code, bit packing, CRC) -> the block's IDAT chunk in its staging slot, its size, its Adler-32 partial sums
//   png_finish          one workgroup: chunk offsets (prefix sum), Adler-32, signature / IHDR / zlib header / final block / IEND, file size
//   png_gather          staging slots -> their places in the file (device memory, or the caller's page-locked buffer: zero copy)
// The algorithm, the format decisions and what pins the byte stream on the CPU: png_block.h.  Integer / byte work, bound by LDS
// latency, not by HBM (6.2 MB in, ~2 MB out per 1080p frame).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "bs_internal.h"
#include "png_block.h"

namespace bs {
namespace {

using namespace png;

__global__ __launch_bounds__(kLanes) void png_choose_filter(const uint8_t *__restrict__ rgb, int w, uint8_t *__restrict__ filt)
{
    __shared__ uint32_t total[5];
    const int row = blockIdx.x, lane = threadIdx.x;
    if (lane < 5) total[lane] = 0;
    __syncthreads();
    uint32_t cost[5] = {0, 0, 0, 0, 0};
    const int n = 3 * w;
    for (int x0 = lane; x0 < n; x0 += 4 * kLanes) {   // four bytes' neighbourhoods are loaded before any is costed
        Neighbourhood nb[4];
        for (int u = 0; u < 4; u++) {
            const int x = x0 + u * kLanes;
            nb[u] = x < n ? neighbourhood(rgb, w, row, x) : Neighbourhood{0, 0, 0, 0};
        }
        for (int u = 0; u < 4; u++)
            if (x0 + u * kLanes < n) neighbourhood_cost(nb[u], cost);
    }
    for (int f = 0; f < 5; f++) atomicAdd(&total[f], cost[f]);
    __syncthreads();
    if (lane == 0) {
        const uint32_t t[5] = {total[0], total[1], total[2], total[3], total[4]};
        filt[row] = (uint8_t)best_filter(t);
    }
}

#define BS_COUNT(f) +1
#define BS_COUNT_ALPHABET(f, which) +1
static_assert(kPngPhases == 1 BS_PNG_BLOCK_PROGRAM(BS_COUNT, BS_COUNT_ALPHABET), "bs_internal.h: kPngPhases = phases of the block program + 1");
#undef BS_COUNT
#undef BS_COUNT_ALPHABET

// kProfile: lane 0 of every block also stores the shader clock after each phase (bs_debug_png_phases: where a block's time goes)
template <bool kProfile>
__global__ __launch_bounds__(kLanes) void png_encode_blocks(Args A, unsigned long long *__restrict__ clocks)
{
    __shared__ Block S;
    const uint32_t lane = threadIdx.x, blk = blockIdx.x;
    uint32_t phase = 0;
    auto tick = [&]() {
        if (kProfile && lane == 0) clocks[(size_t)blk * kPngPhases + phase] = clock64();
        phase++;
    };
    tick();
#define BS_RUN(f) f(lane, S, A, blk); __syncthreads(); tick();
#define BS_RUN_ALPHABET(f, which) f(lane, S, which); __syncthreads(); tick();
    BS_PNG_BLOCK_PROGRAM(BS_RUN, BS_RUN_ALPHABET)
#undef BS_RUN
#undef BS_RUN_ALPHABET
}

__global__ __launch_bounds__(kLanes) void png_finish(FinishArgs A)
{
    __shared__ Finish F;
    const uint32_t lane = threadIdx.x;
    fin_sum(lane, F, A);
    __syncthreads();
    fin_place(lane, F, A);
    __syncthreads();
    fin_copy(lane, F, A);
}

__global__ __launch_bounds__(256) void png_gather(const uint8_t *__restrict__ staging, const uint32_t *__restrict__ sizes,
                                                  const uint32_t *__restrict__ offsets, uint8_t *__restrict__ out)
{
    const uint32_t blk = blockIdx.x;
    const uint8_t *src = staging + (size_t)blk * kSlot;
    uint8_t *dst = out + offsets[blk];
    const uint32_t n = sizes[blk];
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

uint64_t png_file_bound(int w, int h) { return png::file_bound(w, h); }

size_t png_block_count(int w, int h)
{
    const uint64_t total = (uint64_t)h * ((uint64_t)3 * w + 1);
    return (size_t)((total + kBlock - 1) / kBlock);
}

size_t png_scratch_bytes(int w, int h)
{
    const uint64_t total = (uint64_t)h * ((uint64_t)3 * w + 1);
    const size_t nb = (size_t)((total + kBlock - 1) / kBlock);
    return align256((size_t)h) + 3 * align256(nb * sizeof(uint32_t)) + align256(nb * sizeof(uint32_t)) + nb * (size_t)kSlot;
}

// d_rgb8: h x w x 3 bytes (device).  d_scratch: png_scratch_bytes(w, h) (device).  d_out: png_file_bound(w, h) bytes, device memory or the
// device alias of page-locked host memory.  d_file_bytes: one uint64 (same choice).  Enqueues on `stream`; returns non-zero if a launch failed.
// d_clocks (optional): n_blocks * kPngPhases shader-clock stamps (bs_debug_png_phases).
int launch_png_encode(const unsigned char *d_rgb8, int w, int h, void *d_scratch, unsigned char *d_out, uint64_t *d_file_bytes, void *stream,
                      unsigned long long *d_clocks)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    Args A{};
    A.rgb = d_rgb8; A.w = w; A.h = h;
    A.stride = 3u * (uint32_t)w + 1u;
    A.total = (uint64_t)h * A.stride;
    A.n_blocks = (uint32_t)((A.total + kBlock - 1) / kBlock);
    const size_t nb = A.n_blocks;
    uint8_t *p = static_cast<uint8_t *>(d_scratch);
    uint8_t *filt = p;                                          p += align256((size_t)h);
    uint32_t *sizes = reinterpret_cast<uint32_t *>(p);          p += align256(nb * sizeof(uint32_t));
    uint32_t *offsets = reinterpret_cast<uint32_t *>(p);        p += align256(nb * sizeof(uint32_t));
    uint32_t *adler = reinterpret_cast<uint32_t *>(p);          p += 2 * align256(nb * sizeof(uint32_t));
    uint8_t *staging = p;
    A.filt = filt; A.staging = staging; A.sizes = sizes; A.adler = adler;
    hipLaunchKernelGGL(png_choose_filter, dim3(h), dim3(kLanes), 0, s, d_rgb8, w, filt);
    if (d_clocks)
        hipLaunchKernelGGL(png_encode_blocks<true>, dim3(A.n_blocks), dim3(kLanes), 0, s, A, d_clocks);
    else
        hipLaunchKernelGGL(png_encode_blocks<false>, dim3(A.n_blocks), dim3(kLanes), 0, s, A, (unsigned long long *)nullptr);
    FinishArgs FA{sizes, adler, offsets, A.n_blocks, A.total, w, h, d_out, d_file_bytes};
    hipLaunchKernelGGL(png_finish, dim3(1), dim3(kLanes), 0, s, FA);
    hipLaunchKernelGGL(png_gather, dim3(A.n_blocks), dim3(256), 0, s, staging, sizes, offsets, d_out);
    return hipGetLastError() != hipSuccess;
}

}  // namespace bs
