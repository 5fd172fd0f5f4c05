// TEST INFRASTRUCTURE: the PNG encoder's phase program (blackstar_amd/csrc/png_block.h -- the very functions png_kernels.hip runs on
// the GPU) executed lane by lane on the host, so that its byte stream can be checked against zlib / Pillow without a device.  The lane
// order inside a phase is the caller's choice (forwards, backwards, shuffled): on the GPU the 64 lanes of a phase run concurrently, so the
// result must not depend on it.  Not part of the product library (which has no CPU path); built by tests/test_png.py with g++.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <numeric>
#include <random>
#include <vector>

#include "../../blackstar_amd/csrc/png_block.h"

using namespace bs::png;

namespace {

std::vector<uint32_t> lane_order(int order, uint32_t salt)
{
    std::vector<uint32_t> l(kLanes);
    std::iota(l.begin(), l.end(), 0u);
    if (order == 1) std::reverse(l.begin(), l.end());
    if (order >= 2) {
        std::mt19937 g(order * 7919u + salt);
        std::shuffle(l.begin(), l.end(), g);
    }
    return l;
}

}  // namespace

extern "C" {

uint64_t png_emul_bound(int w, int h) { return file_bound(w, h); }

// rgb: h * w * 3 bytes.  out: at least png_emul_bound(w, h) bytes.  filters (optional): receives the h filter types chosen.
// stats (optional): [0] blocks, [1] blocks emitted as stored.
// tables (optional): per block 2 * 286 words -- the literal/length frequencies, then the code lengths chosen for them
static uint32_t *g_tables = nullptr;
void png_emul_set_tables(uint32_t *tables) { g_tables = tables; }
// forced (optional): h filter types to use instead of the encoder's own choice (experiments with other selection rules)
static const uint8_t *g_forced_filters = nullptr;
void png_emul_force_filters(const uint8_t *forced) { g_forced_filters = forced; }

int png_emul_encode(const uint8_t *rgb, int w, int h, uint8_t *out, uint64_t cap, uint64_t *out_bytes, int order, uint8_t *filters, uint32_t *stats)
{
    if (!rgb || !out || !out_bytes || w <= 0 || h <= 0 || cap < file_bound(w, h)) return -1;
    std::vector<uint8_t> filt(h);
    for (int row = 0; row < h; row++) {   // png_choose_filter: one wavefront per row, lanes stride over the bytes, costs summed
        uint32_t cost[5] = {0, 0, 0, 0, 0};
        for (int x = 0; x < 3 * w; x++) filter_cost(rgb, w, row, x, cost);
        filt[row] = g_forced_filters ? g_forced_filters[row] : (uint8_t)best_filter(cost);
    }
    if (filters) std::memcpy(filters, filt.data(), h);
    Args A{};
    A.rgb = rgb; A.filt = filt.data(); A.w = w; A.h = h;
    A.stride = 3u * (uint32_t)w + 1u;
    A.total = (uint64_t)h * A.stride;
    A.n_blocks = (uint32_t)((A.total + kBlock - 1) / kBlock);
    std::vector<uint8_t> staging((size_t)A.n_blocks * kSlot, 0xAA);
    std::vector<uint32_t> sizes(A.n_blocks), adler(2 * (size_t)A.n_blocks), offsets(A.n_blocks);
    A.staging = staging.data(); A.sizes = sizes.data(); A.adler = adler.data();
    auto S = std::make_unique<Block>();
    uint32_t n_stored = 0;
    for (uint32_t blk = 0; blk < A.n_blocks; blk++) {
        std::memset(S.get(), 0xCD, sizeof(Block));   // LDS is not zeroed on the GPU either
        uint32_t phase = 0;
#define RUN(f) { for (uint32_t lane : lane_order(order, blk * 64 + phase)) { f(lane, *S, A, blk); } phase++; }
#define RUN_ALPHABET(f, which) { if (g_tables && phase == 4) { for (int i = 0; i < kLL; i++) g_tables[(size_t)blk * 2 * kLL + i] = S->freq[i]; } \
                                 for (uint32_t lane : lane_order(order, blk * 64 + phase)) { f(lane, *S, which); } \
                                 phase++; }
        BS_PNG_BLOCK_PROGRAM(RUN, RUN_ALPHABET)
#undef RUN
#undef RUN_ALPHABET
        n_stored += !S->use_dyn;
        if (g_tables) for (int i = 0; i < kLL; i++) g_tables[(size_t)blk * 2 * kLL + kLL + i] = S->len[i];
    }
    uint64_t file_bytes = 0;
    FinishArgs FA{sizes.data(), adler.data(), offsets.data(), A.n_blocks, A.total, w, h, out, &file_bytes};
    auto F = std::make_unique<Finish>();
    for (uint32_t lane : lane_order(order, 1)) fin_sum(lane, *F, FA);
    for (uint32_t lane : lane_order(order, 2)) fin_place(lane, *F, FA);
    for (uint32_t lane : lane_order(order, 3)) fin_copy(lane, *F, FA);
    for (uint32_t blk = 0; blk < A.n_blocks; blk++)   // png_gather
        std::memcpy(out + offsets[blk], staging.data() + (size_t)blk * kSlot, sizes[blk]);
    *out_bytes = file_bytes;
    if (stats) { stats[0] = A.n_blocks; stats[1] = n_stored; }
    return 0;
}

uint32_t png_emul_crc(const uint8_t *p, uint32_t n) { return crc_bytes(p, n); }
uint32_t png_emul_crc_combine(uint32_t crc_a, uint32_t crc_b, uint32_t len_b)
{
    uint32_t pow2[16];
    for (uint32_t k = 0; k < 16; k++) pow2[k] = crc_x8_pow2(k);
    return crc_shift(pow2, crc_a, len_b) ^ crc_b;
}

}  // extern "C"
