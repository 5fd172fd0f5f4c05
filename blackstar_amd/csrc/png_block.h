// png_block.h -- the PNG encoder of writeImg (src/Raytracer.hs:23-32: writeImage of the sRGB8 frame) as a PROGRAM OF
// PHASES for one 256-thread workgroup (four wavefronts) per 8 KiB of filtered scanline bytes.  The phases are plain functions of
// (lane, block state, args); png_kernels.hip runs them with a workgroup barrier between consecutive phases, tests/cpp/png_emul.cpp runs
// the same functions lane by lane on the host (forwards, backwards and shuffled: a phase must not depend on what another lane wrote IN
// THE SAME PHASE), so the byte stream the GPU produces is pinned on the CPU by zlib / Pillow decoders before it ever runs on a device.
//
// The reference hands the frame to massiv-io's writeImage (JuicyPixels' PNG encoder over zlib): what is specified is the DECODED image,
// not the file's bytes -- those depend on the zlib version.  This encoder therefore only has to be a valid PNG whose pixels are the
// RGB8 frame, and is built for the GPU instead of for ratio:
//   * scanline filter per row by the minimum sum of the residuals' BIT LENGTHS (filter_cost: about sum log2(1 + |r|); libpng's rule
//     sums |r|), computed by its own small kernel;
//   * the filtered stream is cut into 8 KiB blocks; a block is ONE deflate block with its OWN dynamic Huffman code, closed by an empty
//     stored block (zlib's Z_SYNC_FLUSH marker) so that it ends on a byte boundary, and travels in its own IDAT chunk -- blocks are
//     independent: no bit-level concatenation, no cross-block CRC;
//   * LZ77 is reduced to distance-1 matches (runs of one byte value): after the Sub / Up / Paeth filters a rendered frame is mostly
//     runs of zeros, and a run needs no hash chains -- every lane tokenises its own 32 bytes from a 32-bit equality mask; a run that
//     reaches the end of its lane continues through the next lanes of its group of eight (matches of up to 256 bytes);
//   * code lengths: Shannon lengths ceil(log2(N / f)), clamped to the limit, repaired / filled to an exactly complete code in COUNT
//     space (16 counters, one lane) and handed back to the symbols in frequency order -- everything per symbol (lengths, ranks,
//     canonical codes) runs on all lanes;
//   * the block header is decided per POSITION: a non-zero code length is its own symbol, a run of zero lengths is found on a bit mask
//     and coded by the run symbols 17 / 18 at its pieces' first positions -- every position is costed and emitted by its own lane;
//   * prefix sums (bit offsets of the lanes, of the header's symbols) are each lane summing what lies before it -- sums of groups of 16
//     (accumulated with LDS atomics where the values are produced), then its 15 neighbours: about 30 LDS reads, no log-step scans, no
//     extra barriers; ranks among equal code lengths are counted four bytes per read;
//   * a block whose dynamic encoding is not smaller than the bytes themselves is emitted as a stored block.
// Adler-32 of the filtered stream: per-block sums with block-global weights (no prefix needed), combined by png_finish.  CRC-32 of a
// chunk: 256 partial CRC registers combined in two levels with the x^n mod P operator (the construction zlib's crc32_combine uses).
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define BS_HD __host__ __device__ inline
#else
#define BS_HD inline
#endif

namespace bs {
namespace png {

constexpr int kLanes = 256;               // threads of a block's workgroup
constexpr int kSeg = 32;                  // bytes of the filtered stream one lane tokenises
constexpr int kBlock = kLanes * kSeg;     // bytes per deflate block / IDAT chunk
constexpr int kLL = 286, kLLPad = 320;    // literal/length alphabet (padded)
constexpr int kCL = 19, kCLPad = 32;      // code-length alphabet
constexpr int kDataWords = (kSeg / 4) * (kLanes + 1);  // word (kw, lane) at kw * 257 + lane: conflict-free both ways
constexpr int kStoredMax = 5 + kBlock;    // data bytes of a stored block
constexpr int kOutWords = (kStoredMax + 3) / 4 + 2;
constexpr int kSlot = 8224;               // staging bytes per block: 4 length + 4 type + <= 8197 data + 4 crc, rounded up to 32
constexpr int kHdrMax = 288;              // positions of a block header: <= 286 literal/length lengths + 1 distance length
constexpr uint32_t kHdrNone = 0xFF;
constexpr int kGroup = 16;                // prefix sums go in two levels: sums of groups of 16, then the 15 neighbours
constexpr uint32_t kAdlerMod = 65521;
constexpr uint32_t kHeadBytes = 8 + 25 + 14;   // signature, IHDR chunk, IDAT chunk holding the 2-byte zlib header
constexpr uint32_t kTailBytes = 21 + 12;       // IDAT chunk holding the final empty stored block + Adler-32, IEND chunk

struct Args {
    const uint8_t *rgb;      // h rows of 3 w bytes
    const uint8_t *filt;     // h filter types (png_choose_filter)
    int32_t w, h;
    uint32_t stride;         // 3 w + 1
    uint64_t total;          // h * stride: bytes of the filtered stream
    uint32_t n_blocks;
    uint8_t *staging;        // n_blocks * kSlot
    uint32_t *sizes;         // n_blocks: chunk bytes (12 + data)
    uint32_t *adler;         // n_blocks * 2: partial sums (sum d, sum (n - i) d_i) mod 65521
};

struct Block {
    uint32_t data[kDataWords];
    uint16_t tok[kSeg * kLanes];   // token k of lane j at k * 256 + j: literal = byte value; match = 0x8000 | length (distance 1)
    uint32_t ntok[kLanes];
    uint32_t eq[kLanes];           // bit k: byte k of the lane equals the byte before it (ph_runs)
    uint32_t lane_bits[kLanes], lane_off[kLanes];
    union {                        // the frequencies are dead once the Shannon lengths exist (ph_len_shannon); the words the token
        uint32_t freq[kLLPad];     // walks read are written two barriers later (ph_len_codes)
        uint32_t codelen[kLLPad];  // code | length << 16
    };
    uint8_t len0[kLLPad], len[kLLPad];
    uint16_t rank0[kLLPad];
    uint16_t code[kLLPad];         // bit-reversed canonical codes
    uint32_t cl_freq[kCLPad];
    uint8_t cl_len0[kCLPad], cl_len[kCLPad];
    uint16_t cl_rank0[kCLPad], cl_code[kCLPad];
    uint32_t cnt0[2][16], cum0[2][17], cumf[2][17], next_code[2][17], nused[2];
    uint8_t hdr_cost[kHdrMax];     // bits header position i contributes (0: it lies inside a run of zeros another position codes)
    uint8_t hdr_tok[kHdrMax], hdr_ext[kHdrMax];   // its code-length symbol (0..15, 17, 18; kHdrNone: none) and the run symbols' extra bits
    uint32_t hdr_zero[kHdrMax / 32];               // bit i: the length at header position i is 0
    uint32_t nhdr;                                 // code-length symbols in the header
    uint32_t lane_gsum[kLanes / kGroup], hdr_gsum[kHdrMax / kGroup];   // the same summed over groups of 16 lanes / positions
    uint32_t hlit_n, hclen_n;
    uint32_t out[kOutWords];
    uint32_t n_bytes;      // bytes of the filtered stream in this block
    int32_t prev0;         // the byte before the block (-1: the block starts the stream)
    uint32_t ntot;         // tokens in the block
    uint32_t has_match;
    uint32_t adler_a, adler_b;     // sum d, sum (n_bytes - position) d over the block, each lane's share reduced mod 65521 first
    uint32_t hdr_bits, tok_bits, eob_off, total_bits, use_dyn, data_len;
    uint32_t crc;                                // register form: the chunk's CRC-32 is its complement
    uint32_t crc_group[16], crc_p[16], crc_q[16];   // the groups' registers; crc_seg_pow, crc_group_pow
};

// ---- small helpers -----------------------------------------------------------------------------------------------------------------

BS_HD void lds_add(uint32_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
BS_HD void lds_or(uint32_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
BS_HD void lds_xor(uint32_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicXor(p, v);
#else
    *p ^= v;
#endif
}
BS_HD void lds_max(uint32_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}

BS_HD int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// The four bytes a PNG filter looks at (PNG specification, section 9; bpp = 3): the byte itself, its left (a), upper (b) and upper-left
// (c) neighbours, 0 outside the image.  All four loads are issued whatever the filter: none depends on another.
struct Neighbourhood { int raw, a, b, c; };
BS_HD Neighbourhood neighbourhood(const uint8_t *rgb, int32_t w, int32_t row, int32_t x)
{
    const ptrdiff_t rb = (ptrdiff_t)3 * w;
    const uint8_t *cur = rgb + (ptrdiff_t)row * rb + x;
    Neighbourhood n;
    n.raw = cur[0];
    n.a = x >= 3 ? cur[-3] : 0;
    n.b = row > 0 ? cur[-rb] : 0;
    n.c = (row > 0 && x >= 3) ? cur[-rb - 3] : 0;
    return n;
}
BS_HD uint8_t filtered(const Neighbourhood &n, int f)
{
    const int pred = f == 0 ? 0 : f == 1 ? n.a : f == 2 ? n.b : f == 3 ? ((n.a + n.b) >> 1) : paeth(n.a, n.b, n.c);
    return (uint8_t)(n.raw - pred);
}

// Byte (row, col) of the filtered stream: each row is its filter type followed by its 3 w filtered bytes.
BS_HD uint8_t stream_byte(const Args &A, uint32_t row, uint32_t col)
{
    const int f = A.filt[row];
    return col == 0 ? (uint8_t)f : filtered(neighbourhood(A.rgb, A.w, (int32_t)row, (int32_t)col - 1), f);
}

// What byte x of a row costs under each of the five filters: the BIT LENGTH of |signed residual| (0 for 0, 1 for +-1, 2 for +-2..3, ...),
// i.e. about log2(1 + |r|) -- closer to what the residual will cost after Huffman coding than libpng's sum of |r|, which lets a few large
// residuals outvote many small ones: 2.6 % smaller files on a bloomed frame (779 -> 758 KB at 960x540; a per-row entropy rule, which
// needs five histograms per row, gives 757), 0.4 % on frames that are mostly runs, never larger on the frames tried.
BS_HD uint32_t bit_length(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 32u - (uint32_t)__clz((int)v);   // __clz(0) = 32
#else
    return v ? 32u - (uint32_t)__builtin_clz(v) : 0u;
#endif
}
BS_HD void neighbourhood_cost(const Neighbourhood &n, uint32_t cost[5])
{
    for (int f = 0; f < 5; f++) {
        const int v = (int8_t)filtered(n, f);
        cost[f] += bit_length((uint32_t)(v < 0 ? -v : v));
    }
}
BS_HD void filter_cost(const uint8_t *rgb, int32_t w, int32_t row, int32_t x, uint32_t cost[5])
{
    neighbourhood_cost(neighbourhood(rgb, w, row, x), cost);
}

BS_HD uint32_t best_filter(const uint32_t cost[5])
{
    uint32_t best = 0;
    for (uint32_t f = 1; f < 5; f++)
        if (cost[f] < cost[best]) best = f;
    return best;
}

// Deflate length symbol of a match length 3..258 (RFC 1951, 3.2.5): symbol, number of extra bits, their value.
BS_HD void length_code(uint32_t L, uint32_t &sym, uint32_t &ebits, uint32_t &eval)
{
    if (L == 258) { sym = 285; ebits = 0; eval = 0; return; }
    const uint32_t l = L - 3;
    if (l < 8) { sym = 257 + l; ebits = 0; eval = 0; return; }
    uint32_t e = 1;
    while ((l >> (e + 3)) != 0) e++;   // e = floor(log2 l) - 2
    sym = 257 + 4 * e + 4 + ((l >> e) & 3);
    ebits = e;
    eval = l & ((1u << e) - 1);
}

BS_HD uint32_t bit_reverse(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// ---- CRC-32 (reflected polynomial 0xEDB88320) ----------------------------------------------------------------------------------------
constexpr uint32_t kCrcPoly = 0xEDB88320u;

BS_HD uint32_t crc_update_byte(uint32_t c, uint8_t b)  // the register form (no pre/post inversion)
{
    c ^= b;
    for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
    return c;
}

// a * b mod P over GF(2), reflected bit order (bit 31 = x^0)
BS_HD uint32_t crc_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// x^(8 2^k) mod P, k < 16 (0x00800000 = x^8, each the square of the one before): a switch of immediates -- the block kernel copies them
// into LDS once (a constant array indexed at run time would live in memory behind a 500-cycle load)
BS_HD uint32_t crc_x8_pow2(uint32_t k)
{
    switch (k) {
    case 0: return 0x00800000u; case 1: return 0x00008000u; case 2: return 0xEDB88320u; case 3: return 0xB1E6B092u;
    case 4: return 0xA06A2517u; case 5: return 0xED627DAEu; case 6: return 0x88D14467u; case 7: return 0xD7BBFE6Au;
    case 8: return 0xEC447F11u; case 9: return 0x8E7EA170u; case 10: return 0x6427800Eu; case 11: return 0x4D47BAE0u;
    case 12: return 0x09FE548Fu; case 13: return 0x83852D0Fu; case 14: return 0x30362F1Au; default: return 0x7B5A9CC3u;
    }
}

// x^(8 n) mod P for n < 65536: the product of x^(8 2^k) over the set bits k of n; pow2[k] = crc_x8_pow2(k)
BS_HD uint32_t crc_x8n(const uint32_t *pow2, uint32_t n)
{
    uint32_t p = 1u << 31;   // x^0
    for (uint32_t k = 0; n; k++, n >>= 1)
        if (n & 1u) p = crc_multmodp(pow2[k], p);
    return p;
}

// the CRC-32 of A || B from the CRC-32s of A and B and the length of B (< 65536)
BS_HD uint32_t crc_shift(const uint32_t *pow2, uint32_t crc_a, uint32_t len_b) { return crc_multmodp(crc_x8n(pow2, len_b), crc_a); }

// The block kernel's two-level combination: the chunk's bytes are cut into 256 pieces of kCrcSeg bytes aligned to the chunk's END (the
// ragged piece is the first one), so piece j is followed by (255 - j) kCrcSeg bytes whatever the chunk's length: sixteen pieces
// form a group (shift inside the group: X^(15 - j % 16), X = x^(8 kCrcSeg)), sixteen groups the chunk (shift of group g: Y^(15 - g), Y = X^16).
// One multiplication per lane, constants only.
constexpr uint32_t kCrcSeg = 33;   // 256 * 33 >= 4 + kStoredMax
BS_HD uint32_t crc_seg_pow(uint32_t i)     // X^i, i < 16
{
    switch (i) {
    case 0: return 0x80000000u; case 1: return 0x3183EC92u; case 2: return 0x5B0DF038u; case 3: return 0x333100D6u;
    case 4: return 0xF44779B9u; case 5: return 0xBDCC5801u; case 6: return 0xB57004DDu; case 7: return 0x0B0125EEu;
    case 8: return 0xD3DCF3D3u; case 9: return 0xF45CE70Bu; case 10: return 0xCE48B184u; case 11: return 0x7F8CB060u;
    case 12: return 0x22385622u; case 13: return 0x23A58B5Cu; case 14: return 0xF657F322u; default: return 0x8CBD6CA3u;
    }
}
BS_HD uint32_t crc_group_pow(uint32_t i)   // Y^i, i < 16
{
    switch (i) {
    case 0: return 0x80000000u; case 1: return 0xDBA769B6u; case 2: return 0x17AEC39Eu; case 3: return 0x28DA6DB3u;
    case 4: return 0x447D3DCEu; case 5: return 0x76FC39ACu; case 6: return 0x221BAA2Bu; case 7: return 0xB3BFECFAu;
    case 8: return 0x0D63715Du; case 9: return 0x3C20FE04u; case 10: return 0x6105EA4Au; case 11: return 0xF6CE12A3u;
    case 12: return 0x94C61C3Cu; case 13: return 0x77DCCE78u; case 14: return 0x00025BFBu; default: return 0x396A93D0u;
    }
}

BS_HD uint32_t crc_bytes(const uint8_t *p, uint32_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) c = crc_update_byte(c, p[i]);
    return c ^ 0xFFFFFFFFu;
}

BS_HD void put_be32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

// ---- the block's bytes in LDS ----------------------------------------------------------------------------------------------------------
BS_HD uint32_t data_word(uint32_t lane, uint32_t kw) { return kw * (kLanes + 1) + lane; }
BS_HD uint32_t data_index(uint32_t lane, uint32_t k) { return data_word(lane, k >> 2) * 4 + (k & 3); }
BS_HD uint8_t data_get(const Block &S, uint32_t lane, uint32_t k) { return reinterpret_cast<const uint8_t *>(S.data)[data_index(lane, k)]; }
BS_HD uint32_t lane_bytes(const Block &S, uint32_t lane)
{
    const uint32_t first = lane * kSeg;
    return S.n_bytes <= first ? 0u : (S.n_bytes - first < (uint32_t)kSeg ? S.n_bytes - first : (uint32_t)kSeg);
}

// How many of the bytes v[0 .. s) equal l (v: 4-byte aligned, readable up to the next multiple of 4; l < 128): four bytes per read.
BS_HD uint32_t count_equal_before(const uint8_t *v, uint32_t s, uint32_t l)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(v);
    const uint32_t pat = l * 0x01010101u;
    uint32_t r = 0;
    for (uint32_t i = 0; i * 4 < s; i++) {
        uint32_t x = w[i] ^ pat;                                       // a zero byte where v == l
        x = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in exactly those bytes
        if (s - i * 4 < 4) x &= (1u << (8 * (s - i * 4))) - 1u;        // bytes at or beyond s do not count
#if defined(__HIP_DEVICE_COMPILE__)
        r += (uint32_t)__popc(x);
#else
        r += (uint32_t)__builtin_popcount(x);
#endif
    }
    return r;
}

// ---- phases ----------------------------------------------------------------------------------------------------------------------------
// (every phase: all 256 lanes, then a barrier)

BS_HD void ph_init(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    for (uint32_t i = lane; i < (uint32_t)kLLPad; i += kLanes) {
        S.freq[i] = i == 256 ? 1u : 0u;   // end-of-block is always coded once
        S.len0[i] = 0; S.len[i] = 0; S.rank0[i] = 0; S.code[i] = 0;
    }
    for (uint32_t i = lane; i < (uint32_t)kOutWords; i += kLanes) S.out[i] = 0;
    if (lane < (uint32_t)(kLanes / kGroup)) { S.lane_gsum[lane] = 0; S.crc_group[lane] = 0; S.crc_p[lane] = crc_seg_pow(lane); S.crc_q[lane] = crc_group_pow(lane); }
    if (lane < (uint32_t)(kHdrMax / kGroup)) S.hdr_gsum[lane] = 0;
    if (lane < (uint32_t)(kHdrMax / 32)) S.hdr_zero[lane] = 0;
    if (lane < (uint32_t)kCLPad) {
        S.cl_freq[lane] = 0; S.cl_len0[lane] = 0; S.cl_len[lane] = 0; S.cl_rank0[lane] = 0; S.cl_code[lane] = 0;
        S.cnt0[lane >> 4][lane & 15] = 0;
    }
    if (lane == 0) {
        const uint64_t first = (uint64_t)blk * kBlock;
        const uint64_t left = A.total - first;
        S.n_bytes = left < (uint64_t)kBlock ? (uint32_t)left : (uint32_t)kBlock;
        S.ntot = 0; S.has_match = 0; S.nused[0] = 0; S.nused[1] = 0; S.crc = 0;
        S.adler_a = 0; S.adler_b = 0; S.hlit_n = 257; S.hdr_bits = 0; S.tok_bits = 0; S.nhdr = 0;
        S.prev0 = -1;
        if (first > 0) {
            const uint64_t p = first - 1;
            S.prev0 = stream_byte(A, (uint32_t)(p / A.stride), (uint32_t)(p % A.stride));
        }
    }
}

BS_HD uint32_t load_u32(const uint8_t *p)   // four bytes at any alignment (one global_load_dword on the device)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// filtered bytes of the block into LDS, four at a time: lane l takes positions 4 l .. 4 l + 3, then + 1024, ... -- one word of a
// segment.  Where the four lie inside one row, clear of its first pixel, each of their neighbourhoods is ONE unaligned 4-byte load
// (raw, left, up, up-left: 4 loads for 4 bytes instead of 16); the two groups per row that touch the filter byte or the left edge go
// byte by byte.  A lane's eight groups are loaded four at a time before any of the four is filtered (all eight at once cost 56 registers
// and measured slower: 30 k against 19 k clocks for the phase).
constexpr uint32_t kLoadGroups = kBlock / (4 * kLanes);   // 8
constexpr uint32_t kLoadBatch = 4;
BS_HD void ph_load(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    const uint64_t first = (uint64_t)blk * kBlock + 4 * lane;
    uint32_t row = (uint32_t)(first / A.stride), col = (uint32_t)(first % A.stride);
    const ptrdiff_t rb = (ptrdiff_t)3 * A.w;
    for (uint32_t g0 = 0; g0 < kLoadGroups; g0 += kLoadBatch) {
        uint32_t wraw[kLoadBatch], wa[kLoadBatch], wb[kLoadBatch], wc[kLoadBatch], rows[kLoadBatch], cols[kLoadBatch];
        int f[kLoadBatch];
        for (uint32_t u = 0; u < kLoadBatch; u++) {
            const uint32_t q = ((g0 + u) * kLanes + lane) * 4;   // position in the block
            rows[u] = row; cols[u] = col;
            wraw[u] = wa[u] = wb[u] = wc[u] = 0;
            f[u] = 0;
            if (q + 4 <= S.n_bytes && col >= 4 && col + 4 <= A.stride) {   // fast: one row, x >= 3
                const uint8_t *cur = A.rgb + (ptrdiff_t)row * rb + (col - 1);
                f[u] = A.filt[row];
                wraw[u] = load_u32(cur);
                wa[u] = load_u32(cur - 3);
                if (row > 0) {
                    wb[u] = load_u32(cur - rb);
                    wc[u] = load_u32(cur - rb - 3);
                }
            }
            col += 4 * kLanes;
            while (col >= A.stride) { col -= A.stride; row++; }
        }
        for (uint32_t u = 0; u < kLoadBatch; u++) {
            const uint32_t q = ((g0 + u) * kLanes + lane) * 4;
            if (q >= S.n_bytes) continue;
            uint32_t out = 0;
            if (q + 4 <= S.n_bytes && cols[u] >= 4 && cols[u] + 4 <= A.stride) {
                for (uint32_t k = 0; k < 4; k++) {
                    const Neighbourhood nb{(int)((wraw[u] >> (8 * k)) & 0xFFu), (int)((wa[u] >> (8 * k)) & 0xFFu), (int)((wb[u] >> (8 * k)) & 0xFFu),
                                           (int)((wc[u] >> (8 * k)) & 0xFFu)};
                    out |= (uint32_t)filtered(nb, f[u]) << (8 * k);
                }
            } else {
                uint32_t r = rows[u], c = cols[u];
                for (uint32_t k = 0; k < 4 && q + k < S.n_bytes; k++) {
                    out |= (uint32_t)stream_byte(A, r, c) << (8 * k);
                    if (++c == A.stride) { c = 0; r++; }
                }
            }
            S.data[data_word(q / kSeg, (q % kSeg) / 4)] = out;
        }
    }
}

BS_HD uint32_t count_trailing_zeros(uint32_t v)   // v != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffs((int)v) - 1u;
#else
    return (uint32_t)__builtin_ctz(v);
#endif
}
BS_HD uint32_t population(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(v);
#else
    return (uint32_t)__builtin_popcount(v);
#endif
}

// Bit k of a lane's equality mask: byte k of its 32 equals the byte before it (the previous lane's last byte for k = 0).  Eight words in
// registers, one SWAR zero-byte test per word; the Adler partial sums come from the same registers.
BS_HD void ph_runs(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t n = lane_bytes(S, lane);
    if (n == 0) { S.eq[lane] = 0; return; }
    const int prev = lane == 0 ? S.prev0 : (int)data_get(S, lane - 1, kSeg - 1);
    uint32_t E = 0, a = 0, b = 0, before_word = (uint32_t)(prev & 0xFF) << 24;
    const uint32_t weight0 = S.n_bytes - lane * kSeg;   // Adler weight of the lane's first byte: bytes from it to the end of the block
    for (uint32_t kw = 0; kw < (uint32_t)kSeg / 4; kw++) {
        const uint32_t cur = kw * 4 < n ? S.data[data_word(lane, kw)] : 0u;
        const uint32_t before = (cur << 8) | (before_word >> 24);
        before_word = cur;
        uint32_t x = cur ^ before;                                     // a zero byte where a byte equals the one before
        x = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in exactly those bytes
        E |= (((x >> 7) & 1u) | ((x >> 14) & 2u) | ((x >> 21) & 4u) | ((x >> 28) & 8u)) << (4 * kw);
        for (uint32_t u = 0; u < 4; u++) {                             // (bytes at or beyond n are 0: they add nothing)
            const uint32_t v = (cur >> (8 * u)) & 0xFFu;
            a += v;
            b += v * (weight0 - (4 * kw + u));
        }
    }
    E &= n == 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
    if (prev < 0) E &= ~1u;                                            // the stream's first byte has nothing before it
    S.eq[lane] = E;
    lds_add(&S.adler_a, a % kAdlerMod);
    lds_add(&S.adler_b, b % kAdlerMod);
}

BS_HD uint32_t low_ones(uint32_t E) { return ~E ? count_trailing_zeros(~E) : 32u; }
BS_HD uint32_t high_ones(uint32_t E)
{
    uint32_t t = 0;
    while (t < 32 && ((E >> (31 - t)) & 1u)) t++;
    return t;
}

// The lane's equality mask -> literals and distance-1 matches (runs of three or more 1-bits become one match each, every other byte a
// literal: found with a handful of shifts, visited one set bit at a time), and the symbols' frequencies.
// Runs cross lanes inside a group of kRunGroup = 8 lanes: a match that reaches the end of its lane takes over the leading 1-bits of the
// lanes that follow (a lane that is ONE run hands on to the next), and those lanes drop the bytes they gave away -- matches of up to
// 256 bytes instead of 32.  A group's first lane never gives bytes away, so nobody looks further than seven lanes.
constexpr uint32_t kRunGroup = 8;
BS_HD void ph_tokenize(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t n = lane_bytes(S, lane);
    if (n == 0) { S.ntok[lane] = 0; return; }
    const uint32_t E = S.eq[lane];
    const uint32_t valid = n == 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
    const uint32_t third = E & (E << 1) & (E << 2);                    // third or later position of a run of 1-bits
    const uint32_t in_run = (third | (third >> 1) | (third >> 2)) & E; // every position of a run of three or more
    const uint32_t starts = in_run & ~(in_run << 1);
    uint32_t todo = (valid & ~in_run) | starts;
    // bytes handed to the match that ends the lane before (it is a match if that lane ends in three or more 1-bits)
    const uint32_t lead = low_ones(E);
    const bool gives = lane % kRunGroup != 0 && lead > 0 && high_ones(S.eq[lane - 1]) >= 3;
    if (gives) todo &= lead == 32 ? 0u : ~((1u << lead) - 1u);
    // bytes taken over by this lane's last match, if it reaches the lane's end
    uint32_t takes = 0;
    if (n == 32 && (in_run >> 31))
        for (uint32_t i = lane + 1; i % kRunGroup != 0 && lane_bytes(S, i) > 0; i++) {
            const uint32_t l = low_ones(S.eq[i]);   // (a partial lane's mask has no bits beyond its bytes)
            takes += l;
            if (l < 32) break;
        }
    const uint32_t nt = population(todo);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(S.data);
    for (uint32_t i = 0; i < nt; i++) {
        const uint32_t k = count_trailing_zeros(todo);
        todo &= todo - 1;
        uint32_t tok, sym;
        if ((in_run >> k) & 1u) {                                      // (a set bit of todo inside a run is the run's first position --
            const uint32_t above = ~(in_run >> k);                     //  or what is left of a leading run: then it was given away whole)
            uint32_t run = above ? count_trailing_zeros(above) : 32u;  // the lowest 0 above k ends the run; none: it reaches bit 31
            if (k + run == 32) run += takes;
            uint32_t eb, ev;
            length_code(run, sym, eb, ev);
            tok = 0x8000u | run;
        } else {
            sym = tok = bytes[data_index(lane, k)];
        }
        lds_add(&S.freq[sym], 1);
        S.tok[i * kLanes + lane] = (uint16_t)tok;
    }
    S.ntok[lane] = nt;
    if (nt) lds_add(&S.ntot, nt);
    if (starts) lds_or(&S.has_match, 1u);
}

// Code lengths of one alphabet (which = 0: literal/length, limit 15; 1: code lengths, limit 7), in five phases.
struct Alphabet {
    uint32_t *freq; uint8_t *len0; uint8_t *len; uint16_t *rank0; uint16_t *code;
    uint32_t n, limit, total;
};
BS_HD Alphabet alphabet(Block &S, int which)
{
    if (which == 0) return Alphabet{S.freq, S.len0, S.len, S.rank0, S.code, (uint32_t)kLL, 15u, S.ntot + 1u};
    return Alphabet{S.cl_freq, S.cl_len0, S.cl_len, S.cl_rank0, S.cl_code, (uint32_t)kCL, 7u, S.nhdr};
}

// (1) Shannon length of every used symbol: the smallest l with f 2^l >= N, clamped to [1, limit]
BS_HD void ph_len_shannon(uint32_t lane, Block &S, int which)
{
    const Alphabet al = alphabet(S, which);
    for (uint32_t s = lane; s < al.n; s += kLanes) {
        const uint32_t f = al.freq[s];
        uint32_t l = 0;
        if (f) {
            l = 1;
            while (l < al.limit && ((uint64_t)f << l) < al.total) l++;
            lds_add(&S.cnt0[which][l], 1);
            lds_add(&S.nused[which], 1);
            if (which == 0) lds_max(&S.hlit_n, s + 1);   // HLIT: the header lists the lengths up to the last used symbol (at least 257)
        }
        al.len0[s] = (uint8_t)l;
    }
}

// (2) rank of a symbol among the symbols of its Shannon length (by index)
BS_HD void ph_len_rank(uint32_t lane, Block &S, int which)
{
    const Alphabet al = alphabet(S, which);
    for (uint32_t s = lane; s < al.n; s += kLanes) {
        const uint32_t l = al.len0[s];
        al.rank0[s] = (uint16_t)(l ? count_equal_before(al.len0, s, l) : 0u);
    }
}

// (3) one lane, on the 16 counters: make the code exactly complete (Kraft sum = 1).  Too long (only after clamping): symbols move from
// the longest length below the limit one step down; too short: the affordable move with the largest gain (the shortest length whose step
// 2^-L fits into what is left) until nothing is left -- the longest length in use always fits, so this ends.
BS_HD void ph_len_counts(uint32_t lane, Block &S, int which)
{
    if (lane != 0) return;
    const uint32_t limit = which == 0 ? 15u : 7u;
    uint32_t c[17];
    for (uint32_t l = 0; l <= 16; l++) c[l] = l >= 1 && l <= limit ? S.cnt0[which][l] : 0u;
    uint32_t acc = 0;
    for (uint32_t l = 0; l <= 16; l++) {   // cum0[l]: used symbols with a Shannon length below l
        S.cum0[which][l] = acc;
        acc += c[l];
    }
    if (S.nused[which] == 1) {
        // a single symbol: one bit, and a second (unused) code of one bit so that the set is complete (zlib's inflate rejects an
        // incomplete code-length code; the literal/length alphabet always has two symbols)
        for (uint32_t l = 0; l <= 16; l++) c[l] = 0;
        c[1] = 2;
    } else {
        const uint32_t T = 1u << limit;
        uint32_t K = 0;
        for (uint32_t l = 1; l <= limit; l++) K += c[l] << (limit - l);
        while (K > T) {
            uint32_t l = limit - 1;
            while (l >= 1 && c[l] == 0) l--;
            c[l]--; c[l + 1]++;
            K -= 1u << (limit - l - 1);
        }
        while (K < T) {
            const uint32_t left = T - K;
            uint32_t l = 2;
            while (l <= limit && (c[l] == 0 || (1u << (limit - l)) > left)) l++;
            if (l > limit) break;   // (cannot happen: what is left is a multiple of the longest code's step)
            uint32_t m = left >> (limit - l);
            if (m > c[l]) m = c[l];
            c[l] -= m; c[l - 1] += m;
            K += m << (limit - l);
        }
    }
    acc = 0;
    uint32_t code = 0;
    for (uint32_t l = 0; l <= 16; l++) {   // cumf[l]: symbols with a final length <= l; next_code: RFC 1951, 3.2.2
        acc += c[l];
        S.cumf[which][l] = acc;
        if (l >= 1) {
            code = (code + (l >= 2 ? c[l - 1] : 0u)) << 1;
            S.next_code[which][l] = code;
        }
    }
}

// (4) final lengths handed out in the order (Shannon length, index): more frequent symbols never get longer codes
BS_HD void ph_len_assign(uint32_t lane, Block &S, int which)
{
    const Alphabet al = alphabet(S, which);
    for (uint32_t s = lane; s < al.n; s += kLanes) {
        const uint32_t l0 = al.len0[s];
        uint32_t l = 0;
        if (l0) {
            const uint32_t q = S.cum0[which][l0] + al.rank0[s];
            l = 1;
            while (S.cumf[which][l] <= q) l++;
        }
        if (which == 1 && S.nused[1] == 1 && s < 2 && !l0) {   // the unused second code of a one-symbol alphabet (see ph_len_counts)
            uint32_t only = 0;
            while (S.cl_freq[only] == 0) only++;
            if (s == (only == 0 ? 1u : 0u)) l = 1;
        }
        al.len[s] = (uint8_t)l;
    }
}

// (5) canonical codes (RFC 1951, 3.2.2), stored bit-reversed: deflate sends Huffman codes most significant bit first
BS_HD void ph_len_codes(uint32_t lane, Block &S, int which)
{
    const Alphabet al = alphabet(S, which);
    for (uint32_t s = lane; s < al.n; s += kLanes) {
        const uint32_t l = al.len[s];
        if (!l) continue;
        al.code[s] = (uint16_t)bit_reverse(S.next_code[which][l] + count_equal_before(al.len, s, l), l);
        if (which == 0) S.codelen[s] = al.code[s] | (l << 16);
    }
}

// The block header lists HLIT + 257 literal/length code lengths and the one distance code length; position i of that list:
BS_HD uint32_t header_value(const Block &S, uint32_t i) { return i < S.hlit_n ? S.len[i] : (S.has_match ? 1u : 0u); }
BS_HD uint32_t header_positions(const Block &S) { return S.hlit_n + 1; }

// which header positions hold a zero length (runs of them are coded by the symbols 17 and 18)
BS_HD void ph_header_zeros(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t n = header_positions(S);
    for (uint32_t i = lane; i < n; i += kLanes)
        if (header_value(S, i) == 0) lds_or(&S.hdr_zero[i >> 5], 1u << (i & 31));
}

BS_HD uint32_t count_leading_zeros(uint32_t v)   // v != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)v);
#else
    return (uint32_t)__builtin_clz(v);
#endif
}

// The header as code-length symbols (RFC 1951, 3.2.7), one decision per position: a non-zero length is its own symbol; a run [a, b) of
// zero lengths is cut into pieces of 138 from its start -- a piece of 11..138 is ONE symbol 18 (7 extra bits) carried by the piece's
// first position, 3..10 one symbol 17 (3 extra bits), 1..2 literal zeros; the other positions of a piece carry nothing.  The run
// around a position is found on the bit mask of zero lengths, a word at a time.  (Symbol 16, "repeat the previous length", is not used.)
BS_HD void ph_header_tokens(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t n = header_positions(S);
    for (uint32_t i = lane; i < n; i += kLanes) {
        const uint32_t v = header_value(S, i);
        uint32_t tok = v, ext = 0;
        if (v == 0) {
            uint32_t a = i;                                   // the run's first position: below it a non-zero length, or nothing
            while (a > 0) {
                const uint32_t off = (a - 1) & 31;
                const uint32_t m = ~S.hdr_zero[(a - 1) >> 5] << (31 - off);   // bit 31: position a - 1 is NOT zero, bit 30: a - 2, ...
                if (m) { a -= count_leading_zeros(m); break; }
                a -= off + 1;
            }
            uint32_t b = i + 1;                               // one past the run's last position (mask bits at or beyond n are 0)
            while (b < n) {
                const uint32_t off = b & 31;
                const uint32_t m = ~S.hdr_zero[b >> 5] >> off;                 // bit 0: position b is NOT zero
                if (m) { b += count_trailing_zeros(m); break; }
                b += 32 - off;
            }
            if (b > n) b = n;
            const uint32_t p0 = a + (i - a) / 138 * 138;
            const uint32_t plen = b - p0 < 138 ? b - p0 : 138;
            if (plen >= 3) {
                if (i != p0) tok = kHdrNone;
                else if (plen >= 11) { tok = 18; ext = plen - 11; }
                else { tok = 17; ext = plen - 3; }
            }
        }
        S.hdr_tok[i] = (uint8_t)tok;
        S.hdr_ext[i] = (uint8_t)ext;
        if (tok != kHdrNone) {
            lds_add(&S.cl_freq[tok], 1);
            lds_add(&S.nhdr, 1);
        }
    }
}

BS_HD uint32_t cl_order(uint32_t i)
{
    const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[i];
}

BS_HD uint32_t token_bits(const Block &S, uint32_t t)
{
    if (!(t & 0x8000u)) return S.codelen[t] >> 16;
    uint32_t sym, eb, ev;
    length_code(t & 0x7FFFu, sym, eb, ev);
    return (S.codelen[sym] >> 16) + eb + 1;   // + the distance code: one bit
}

// bits of every lane's tokens and of every header position
BS_HD void ph_bitcount(uint32_t lane, Block &S, const Args &, uint32_t)
{
    uint32_t bits = 0;
    for (uint32_t i = 0; i < S.ntok[lane]; i++) bits += token_bits(S, S.tok[i * kLanes + lane]);
    S.lane_bits[lane] = bits;
    if (bits) {
        lds_add(&S.tok_bits, bits);
        lds_add(&S.lane_gsum[lane / kGroup], bits);
    }
    const uint32_t n = header_positions(S);
    for (uint32_t i = lane; i < n; i += kLanes) {
        const uint32_t t = S.hdr_tok[i];
        const uint32_t c = t == kHdrNone ? 0u : S.cl_len[t] + (t == 17 ? 3u : t == 18 ? 7u : 0u);
        S.hdr_cost[i] = (uint8_t)c;
        if (c) {
            lds_add(&S.hdr_bits, c);
            lds_add(&S.hdr_gsum[i / kGroup], c);
        }
    }
    if (lane == 0) {
        uint32_t hc = kCL;
        while (hc > 4 && S.cl_len[cl_order(hc - 1)] == 0) hc--;
        S.hclen_n = hc;
    }
}

// where every lane's bits go (the groups before its group, then the lanes before it in its group), and whether the block is worth coding
BS_HD void ph_plan(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t head = 3 + 5 + 5 + 4 + 3 * S.hclen_n + S.hdr_bits;
    uint32_t off = head;
    for (uint32_t g = 0; g < lane / kGroup; g++) off += S.lane_gsum[g];
    for (uint32_t j = lane / kGroup * kGroup; j < lane; j++) off += S.lane_bits[j];
    S.lane_off[lane] = off;
    if (lane != 0) return;
    const uint32_t eob = head + S.tok_bits;
    S.eob_off = eob;
    S.total_bits = eob + S.len[256];
    const uint32_t dyn = (S.total_bits + 3 + 7) / 8 + 4;   // + the empty stored block: 3 bits, padding to the byte, LEN, NLEN
    const uint32_t stored = 5 + S.n_bytes;
    S.use_dyn = dyn < stored;
    S.data_len = S.use_dyn ? dyn : stored;
}

struct BitWriter {
    uint32_t *out;
    uint64_t acc;
    uint32_t nacc, word;
    BS_HD BitWriter(uint32_t *o, uint32_t bitpos) : out(o), acc(0), nacc(bitpos & 31u), word(bitpos >> 5) {}
    BS_HD void put(uint32_t v, uint32_t n)   // n <= 16
    {
        acc |= (uint64_t)v << nacc;
        nacc += n;
        if (nacc >= 32) {
            lds_or(&out[word], (uint32_t)acc);
            acc >>= 32;
            nacc -= 32;
            word++;
        }
    }
    BS_HD void flush()
    {
        if (nacc) lds_or(&out[word], (uint32_t)acc);
        acc = 0;
    }
};

BS_HD void ph_emit(uint32_t lane, Block &S, const Args &, uint32_t)
{
    uint8_t *ob = reinterpret_cast<uint8_t *>(S.out);
    if (!S.use_dyn) {   // stored block: BFINAL = 0, BTYPE = 00, LEN, NLEN, the bytes
        if (lane == 0) {
            ob[0] = 0;
            ob[1] = (uint8_t)S.n_bytes; ob[2] = (uint8_t)(S.n_bytes >> 8);
            ob[3] = (uint8_t)~S.n_bytes; ob[4] = (uint8_t)(~S.n_bytes >> 8);
        }
        const uint32_t n = lane_bytes(S, lane);
        for (uint32_t k = 0; k < n; k++) ob[5 + lane * kSeg + k] = data_get(S, lane, k);
        return;
    }
    if (lane == 0) {
        BitWriter bw(S.out, 0);
        bw.put(0u | (2u << 1), 3);   // BFINAL = 0, BTYPE = 10 (dynamic)
        bw.put(S.hlit_n - 257, 5);
        bw.put(0, 5);                // HDIST: one distance code
        bw.put(S.hclen_n - 4, 4);
        for (uint32_t i = 0; i < S.hclen_n; i++) bw.put(S.cl_len[cl_order(i)], 3);
        bw.flush();
        BitWriter be(S.out, S.eob_off);
        be.put(S.code[256], S.len[256]);
        be.put(0, 3);                // the empty stored block: BFINAL = 0, BTYPE = 00
        be.flush();
        BitWriter bn(S.out, ((S.total_bits + 3 + 7) / 8 + 2) * 8);   // behind the padding: LEN = 0 (already there), NLEN = 0xFFFF
        bn.put(0xFFFFu, 16);
        bn.flush();
    }
    // header positions lane, lane + 256: each sums the cost of the positions before it (whole groups, then its group's)
    const uint32_t n = header_positions(S);
    for (uint32_t i = lane; i < n; i += kLanes) {
        uint32_t off = 3 + 5 + 5 + 4 + 3 * S.hclen_n;
        for (uint32_t g = 0; g < i / kGroup; g++) off += S.hdr_gsum[g];
        for (uint32_t j = i / kGroup * kGroup; j < i; j++) off += S.hdr_cost[j];
        const uint32_t t = S.hdr_tok[i];
        if (t == kHdrNone) continue;
        BitWriter bh(S.out, off);
        bh.put(S.cl_code[t], S.cl_len[t]);
        if (t == 17) bh.put(S.hdr_ext[i], 3);
        if (t == 18) bh.put(S.hdr_ext[i], 7);
        bh.flush();
    }
    BitWriter bw(S.out, S.lane_off[lane]);
    for (uint32_t i = 0; i < S.ntok[lane]; i++) {
        const uint32_t t = S.tok[i * kLanes + lane];
        if (!(t & 0x8000u)) {
            const uint32_t cl = S.codelen[t];
            bw.put(cl & 0xFFFFu, cl >> 16);
        } else {
            uint32_t sym, eb, ev;
            length_code(t & 0x7FFFu, sym, eb, ev);
            const uint32_t cl = S.codelen[sym];
            bw.put(cl & 0xFFFFu, cl >> 16);
            bw.put(ev, eb + 1);      // the extra bits, then distance 1: the one distance code, one bit (0)
        }
    }
    bw.flush();
}

// CRC-32 of the chunk ("IDAT" + data), level one: every lane runs the register over its piece (the piece holding the chunk's first byte
// starts from 0xFFFFFFFF, the others from 0), shifts it to the end of its group, XOR into the group's register
BS_HD void ph_crc(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint8_t *ob = reinterpret_cast<const uint8_t *>(S.out);
    const int32_t m = 4 + (int32_t)S.data_len;
    const int32_t hi = m - (int32_t)((kLanes - 1 - lane) * kCrcSeg);
    if (hi <= 0) return;
    const int32_t lo = hi > (int32_t)kCrcSeg ? hi - (int32_t)kCrcSeg : 0;
    const uint8_t type[4] = {'I', 'D', 'A', 'T'};
    uint32_t c = lo == 0 ? 0xFFFFFFFFu : 0u;
    for (int32_t i = lo; i < hi; i++) c = crc_update_byte(c, i < 4 ? type[i] : ob[i - 4]);
    lds_xor(&S.crc_group[lane / 16], crc_multmodp(S.crc_p[15 - lane % 16], c));
}

// level two: the sixteen groups' registers, each shifted to the end of the chunk
BS_HD void ph_crc_groups(uint32_t lane, Block &S, const Args &, uint32_t)
{
    if (lane < 16 && S.crc_group[lane]) lds_xor(&S.crc, crc_multmodp(S.crc_q[15 - lane], S.crc_group[lane]));
}

BS_HD void ph_write(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    uint8_t *slot = A.staging + (size_t)blk * kSlot;
    uint32_t *slot_words = reinterpret_cast<uint32_t *>(slot + 8);   // kSlot and 8 are multiples of 4
    const uint32_t full = S.data_len / 4;
    for (uint32_t i = lane; i < full; i += kLanes) slot_words[i] = S.out[i];
    if (lane == 0) {
        const uint8_t *ob = reinterpret_cast<const uint8_t *>(S.out);
        for (uint32_t i = full * 4; i < S.data_len; i++) slot[8 + i] = ob[i];
        put_be32(slot, S.data_len);
        slot[4] = 'I'; slot[5] = 'D'; slot[6] = 'A'; slot[7] = 'T';
        put_be32(slot + 8 + S.data_len, ~S.crc);
        A.sizes[blk] = 12 + S.data_len;
        A.adler[2 * blk] = S.adler_a % kAdlerMod;
        A.adler[2 * blk + 1] = S.adler_b % kAdlerMod;
    }
}

// The phases of a block, in order: RUN(f) runs f on every lane, then a barrier.
#define BS_PNG_BLOCK_PROGRAM(RUN, RUN_ALPHABET)                                                       \
    RUN(ph_init) RUN(ph_load) RUN(ph_runs) RUN(ph_tokenize)                                                        \
    RUN_ALPHABET(ph_len_shannon, 0) RUN_ALPHABET(ph_len_rank, 0) RUN_ALPHABET(ph_len_counts, 0)       \
    RUN_ALPHABET(ph_len_assign, 0) RUN_ALPHABET(ph_len_codes, 0)                                      \
    RUN(ph_header_zeros) RUN(ph_header_tokens)                                                        \
    RUN_ALPHABET(ph_len_shannon, 1) RUN_ALPHABET(ph_len_rank, 1) RUN_ALPHABET(ph_len_counts, 1)       \
    RUN_ALPHABET(ph_len_assign, 1) RUN_ALPHABET(ph_len_codes, 1)                                      \
    RUN(ph_bitcount) RUN(ph_plan) RUN(ph_emit) RUN(ph_crc) RUN(ph_crc_groups) RUN(ph_write)

// ---- the frame: offsets of the chunks, Adler-32, the fixed chunks around them (png_finish: ONE workgroup) ---------------------------------
struct Finish {
    uint32_t lane_sum[kLanes];
    uint32_t lane_a[kLanes], lane_b[kLanes], lane_n[kLanes];   // all mod 65521
    uint8_t head[kHeadBytes + 1], tail[kTailBytes + 3];        // the fixed chunks, built by one lane, copied out by eighty
    uint32_t end;                                              // where the tail goes
};

struct FinishArgs {
    const uint32_t *sizes;   // n_blocks
    const uint32_t *adler;   // n_blocks * 2
    uint32_t *offsets;       // n_blocks: where each block's chunk goes in the file
    uint32_t n_blocks;
    uint64_t total;          // bytes of the filtered stream
    int32_t w, h;
    uint8_t *out;            // the file
    uint64_t *file_bytes;    // its size
};

BS_HD void fin_range(uint32_t lane, uint32_t n, uint32_t &b0, uint32_t &b1)
{
    const uint32_t per = (n + kLanes - 1) / kLanes;
    b0 = lane * per < n ? lane * per : n;
    b1 = b0 + per < n ? b0 + per : n;
}

// Adler-32 sums of X || Y from those of X and Y: (aX + aY, bX + nY aX + bY), everything mod 65521 (65520^2 < 2^32: 32-bit arithmetic)
BS_HD void adler_append(uint32_t &a, uint32_t &b, uint32_t &n, uint32_t ay, uint32_t by, uint32_t ny)
{
    b = (b + (ny * a) % kAdlerMod + by) % kAdlerMod;
    a = (a + ay) % kAdlerMod;
    n = (n + ny) % kAdlerMod;
}

BS_HD void fin_sum(uint32_t lane, Finish &F, const FinishArgs &A)
{
    uint32_t b0, b1;
    fin_range(lane, A.n_blocks, b0, b1);
    uint32_t s = 0, a = 0, b = 0, n = 0;
    for (uint32_t k = b0; k < b1; k++) {
        s += A.sizes[k];
        const uint64_t first = (uint64_t)k * kBlock;
        const uint32_t nk = A.total - first < (uint64_t)kBlock ? (uint32_t)(A.total - first) : (uint32_t)kBlock;
        adler_append(a, b, n, A.adler[2 * k], A.adler[2 * k + 1], nk);
    }
    F.lane_sum[lane] = s;
    F.lane_a[lane] = a; F.lane_b[lane] = b; F.lane_n[lane] = n;
}

BS_HD void write_chunk(uint8_t *p, const char type[4], const uint8_t *data, uint32_t n)
{
    put_be32(p, n);
    for (int i = 0; i < 4; i++) p[4 + i] = (uint8_t)type[i];
    for (uint32_t i = 0; i < n; i++) p[8 + i] = data[i];
    put_be32(p + 8 + n, crc_bytes(p + 4, 4 + n));
}

BS_HD void fin_place(uint32_t lane, Finish &F, const FinishArgs &A)
{
    uint32_t b0, b1;
    fin_range(lane, A.n_blocks, b0, b1);
    uint32_t at = kHeadBytes;
    for (uint32_t j = 0; j < lane; j++) at += F.lane_sum[j];
    for (uint32_t k = b0; k < b1; k++) {
        A.offsets[k] = at;
        at += A.sizes[k];
    }
    if (lane != 0) return;
    uint32_t end = kHeadBytes, a = 0, b = 0, n = 0;
    for (uint32_t j = 0; j < (uint32_t)kLanes; j++) {
        end += F.lane_sum[j];
        adler_append(a, b, n, F.lane_a[j], F.lane_b[j], F.lane_n[j]);
    }
    a = (a + 1) % kAdlerMod;                            // Adler-32 starts at A = 1, which every byte adds to B once
    b = (b + n) % kAdlerMod;
    uint8_t *o = F.head;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    for (int i = 0; i < 8; i++) o[i] = sig[i];
    uint8_t ihdr[13];
    put_be32(ihdr, (uint32_t)A.w);
    put_be32(ihdr + 4, (uint32_t)A.h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8 bits, RGB, deflate, adaptive filtering, no interlace
    write_chunk(o + 8, "IHDR", ihdr, 13);
    const uint8_t zhdr[2] = {0x78, 0x01};               // zlib: deflate, 32 KiB window; no preset dictionary, fastest-compression hint
    write_chunk(o + 33, "IDAT", zhdr, 2);
    uint8_t tail[9] = {0x01, 0x00, 0x00, 0xFF, 0xFF, 0, 0, 0, 0};   // the final (empty, stored) block, then Adler-32
    put_be32(tail + 5, (b << 16) | a);
    write_chunk(F.tail, "IDAT", tail, 9);
    write_chunk(F.tail + 21, "IEND", tail, 0);
    F.end = end;
}

// the fixed chunks into the file (one byte per lane: the file may live in the caller's page-locked memory, across PCIe)
BS_HD void fin_copy(uint32_t lane, Finish &F, const FinishArgs &A)
{
    if (lane < kHeadBytes) A.out[lane] = F.head[lane];
    if (lane >= 64 && lane < 64 + kTailBytes) A.out[F.end + lane - 64] = F.tail[lane - 64];
    if (lane == 0) *A.file_bytes = (uint64_t)F.end + kTailBytes;
}

// Bytes a w x h RGB8 frame can take at most as a file of this encoder (every block stored).
BS_HD uint64_t file_bound(int32_t w, int32_t h)
{
    const uint64_t total = (uint64_t)h * ((uint64_t)3 * w + 1);
    const uint64_t nb = (total + kBlock - 1) / kBlock;
    return kHeadBytes + kTailBytes + total + nb * (5 + 12);
}

}  // namespace png
}  // namespace bs
