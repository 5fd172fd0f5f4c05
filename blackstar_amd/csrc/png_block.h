// png_block.h -- the PNG encoder of writeImg (src/Raytracer.hs:23-32: writeImage of the sRGB8 frame) as a PROGRAM OF
// PHASES for one 64-lane wavefront per 8 KiB of filtered scanline bytes.  The phases are plain functions of (lane, block state, args);
// png_kernels.hip runs them with a workgroup barrier between consecutive phases, tests/cpp/png_emul.cpp runs the same functions lane by
// lane on the host (forwards, backwards and shuffled: a phase must not depend on what another lane wrote IN THE SAME PHASE), so the
// byte stream the GPU produces is pinned on the CPU by zlib / Pillow decoders before it ever runs on a device.
//
// The reference hands the frame to massiv-io's writeImage (JuicyPixels' PNG encoder over zlib): what is specified is the DECODED image,
// not the file's bytes -- those depend on the zlib version.  This encoder therefore only has to be a valid PNG whose pixels are the
// RGB8 frame, and is built for the GPU instead of for ratio:
//   * scanline filter per row by the minimum-sum-of-absolute-differences rule (png_filter_cost), computed by its own small kernel;
//   * the filtered stream is cut into 8 KiB blocks; a block is ONE deflate block with its OWN dynamic Huffman code, closed by an empty
//     stored block (zlib's Z_SYNC_FLUSH marker) so that it ends on a byte boundary, and travels in its own IDAT chunk -- blocks are
//     independent: no bit-level concatenation, no cross-block CRC;
//   * LZ77 is reduced to distance-1 matches (runs of one byte value): after the Sub / Up / Paeth filters a rendered frame is mostly
//     runs of zeros, and a run needs no hash chains -- every lane tokenises its own 128 bytes serially;
//   * code lengths: Shannon lengths ceil(log2(N / f)), clamped to the limit, repaired / filled to an exactly complete code in COUNT
//     space (16 counters, one lane) and handed back to the symbols in frequency order -- the symbol-parallel parts (lengths, ranks,
//     canonical codes) run on all lanes;
//   * a block whose dynamic encoding is not smaller than the bytes themselves is emitted as a stored block.
// Adler-32 of the filtered stream: per-block partial sums, combined by png_finish.  CRC-32 of a chunk: 64 partial CRCs combined with
// the x^n mod P operator (the construction zlib's crc32_combine uses).
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

constexpr int kLanes = 64;
constexpr int kSeg = 128;                 // bytes of the filtered stream one lane tokenises
constexpr int kBlock = kLanes * kSeg;     // bytes per deflate block / IDAT chunk
constexpr int kLL = 286, kLLPad = 320;    // literal/length alphabet (padded to a multiple of kLanes)
constexpr int kCL = 19;                   // code-length alphabet
constexpr int kDataWords = (kSeg / 4) * (kLanes + 1);  // word (kw, lane) at kw * 65 + lane: conflict-free both ways
constexpr int kStoredMax = 5 + kBlock;    // data bytes of a stored block
constexpr int kOutWords = (kStoredMax + 3) / 4 + 2;
constexpr int kSlot = 8224;               // staging bytes per block: 4 length + 4 type + <= 8197 data + 4 crc, rounded up to 32
constexpr int kHdrMax = kLL + 2;          // code-length tokens of a block header (no run symbol 16: one token per length at most)
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
    uint16_t tok[kSeg * kLanes];   // token k of lane j at k * 64 + j: literal = byte value; match = 0x8000 | length (distance 1)
    uint32_t ntok[kLanes];
    uint32_t lane_a[kLanes], lane_b[kLanes];   // Adler partial sums of the lane's bytes
    uint32_t lane_bits[kLanes], lane_off[kLanes];
    uint32_t freq[kLLPad];
    uint8_t len0[kLLPad], len[kLLPad];
    uint16_t rank0[kLLPad];
    uint16_t code[kLLPad];         // bit-reversed canonical codes
    uint32_t cl_freq[kLanes];
    uint8_t cl_len0[kLanes], cl_len[kLanes];
    uint16_t cl_rank0[kLanes], cl_code[kLanes];
    uint32_t cnt0[2][16], cum0[2][17], cumf[2][17], next_code[2][17], nused[2];
    uint8_t hdr_sym[kHdrMax], hdr_ext[kHdrMax];
    uint32_t nhdr, hlit_n, hclen_n;
    uint32_t out[kOutWords];
    uint32_t n_bytes;      // bytes of the filtered stream in this block
    int32_t prev0;         // the byte before the block (-1: the block starts the stream)
    uint32_t ntot;         // tokens in the block
    uint32_t has_match;
    uint32_t hdr_bits, eob_off, total_bits, use_dyn, data_len;
    uint32_t crc;
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

BS_HD int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// The PNG filter `f` applied at byte x of row `row` (PNG specification, section 9: bpp = 3).
BS_HD uint8_t filter_byte(const uint8_t *rgb, int32_t w, int32_t row, int32_t x, int f)
{
    const size_t rb = (size_t)3 * w;
    const uint8_t *cur = rgb + (size_t)row * rb;
    const int raw = cur[x];
    if (f == 0) return (uint8_t)raw;
    const int a = x >= 3 ? cur[x - 3] : 0;
    if (f == 1) return (uint8_t)(raw - a);
    const int b = row > 0 ? cur[(ptrdiff_t)x - (ptrdiff_t)rb] : 0;
    if (f == 2) return (uint8_t)(raw - b);
    if (f == 3) return (uint8_t)(raw - ((a + b) >> 1));
    const int c = (row > 0 && x >= 3) ? cur[(ptrdiff_t)x - 3 - (ptrdiff_t)rb] : 0;
    return (uint8_t)(raw - paeth(a, b, c));
}

// Byte p of the filtered stream: each row is its filter type followed by its 3 w filtered bytes.
BS_HD uint8_t stream_byte(const Args &A, uint32_t row, uint32_t col)
{
    const int f = A.filt[row];
    return col == 0 ? (uint8_t)f : filter_byte(A.rgb, A.w, (int32_t)row, (int32_t)col - 1, f);
}

// What byte x of a row costs under each of the five filters: |signed residual| (libpng's heuristic).
BS_HD void filter_cost(const uint8_t *rgb, int32_t w, int32_t row, int32_t x, uint32_t cost[5])
{
    for (int f = 0; f < 5; f++) {
        const int v = (int8_t)filter_byte(rgb, w, row, x, f);
        cost[f] += (uint32_t)(v < 0 ? -v : v);
    }
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

// x^(8 n) mod P
BS_HD uint32_t crc_x8n(uint32_t n)
{
    uint32_t p = 1u << 31;          // x^0
    uint32_t base = 1u << (31 - 8);  // x^8
    while (n) {
        if (n & 1u) p = crc_multmodp(base, p);
        n >>= 1;
        if (n) base = crc_multmodp(base, base);
    }
    return p;
}

// the CRC-32 of A || B from the CRC-32s of A and B and the length of B
BS_HD uint32_t crc_shift(uint32_t crc_a, uint32_t len_b) { return crc_multmodp(crc_x8n(len_b), crc_a); }

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
BS_HD uint32_t data_index(uint32_t lane, uint32_t k) { return ((k >> 2) * (kLanes + 1) + lane) * 4 + (k & 3); }
BS_HD uint8_t data_get(const Block &S, uint32_t lane, uint32_t k) { return reinterpret_cast<const uint8_t *>(S.data)[data_index(lane, k)]; }
BS_HD uint32_t lane_bytes(const Block &S, uint32_t lane)
{
    const uint32_t first = lane * kSeg;
    return S.n_bytes <= first ? 0u : (S.n_bytes - first < (uint32_t)kSeg ? S.n_bytes - first : (uint32_t)kSeg);
}

// ---- phases ----------------------------------------------------------------------------------------------------------------------------
// (every phase: all 64 lanes, then a barrier)

BS_HD void ph_init(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    for (uint32_t i = lane; i < (uint32_t)kLLPad; i += kLanes) {
        S.freq[i] = i == 256 ? 1u : 0u;   // end-of-block is always coded once
        S.len0[i] = 0; S.len[i] = 0; S.rank0[i] = 0; S.code[i] = 0;
    }
    for (uint32_t i = lane; i < (uint32_t)kOutWords; i += kLanes) S.out[i] = 0;
    S.cl_freq[lane] = 0; S.cl_len0[lane] = 0; S.cl_len[lane] = 0; S.cl_rank0[lane] = 0; S.cl_code[lane] = 0;
    if (lane < 32) S.cnt0[lane >> 4][lane & 15] = 0;
    if (lane == 0) {
        const uint64_t first = (uint64_t)blk * kBlock;
        const uint64_t left = A.total - first;
        S.n_bytes = left < (uint64_t)kBlock ? (uint32_t)left : (uint32_t)kBlock;
        S.ntot = 0; S.has_match = 0; S.nused[0] = 0; S.nused[1] = 0; S.nhdr = 0; S.crc = 0;
        S.prev0 = -1;
        if (first > 0) {
            const uint64_t p = first - 1;
            S.prev0 = stream_byte(A, (uint32_t)(p / A.stride), (uint32_t)(p % A.stride));
        }
    }
}

// filtered bytes of the block into LDS: lane l takes positions l, l + 64, ... (adjacent lanes read adjacent bytes of the frame)
BS_HD void ph_load(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    const uint64_t first = (uint64_t)blk * kBlock + lane;
    uint32_t row = (uint32_t)(first / A.stride), col = (uint32_t)(first % A.stride);
    uint8_t *bytes = reinterpret_cast<uint8_t *>(S.data);
    for (uint32_t it = 0; it < (uint32_t)kBlock / kLanes; it++) {
        const uint32_t q = it * kLanes + lane;   // position in the block
        if (q < S.n_bytes) bytes[data_index(q / kSeg, q % kSeg)] = stream_byte(A, row, col);
        col += kLanes;
        while (col >= A.stride) { col -= A.stride; row++; }
    }
}

// the lane's 128 bytes -> literals and distance-1 matches; symbol frequencies; Adler partial sums
BS_HD void ph_tokenize(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint32_t n = lane_bytes(S, lane);
    int prev = lane == 0 ? S.prev0 : (n ? (int)data_get(S, lane - 1, kSeg - 1) : -1);
    uint32_t nt = 0, a = 0, b = 0, k = 0;
    bool any = false;
    while (k < n) {
        const int v = data_get(S, lane, k);
        uint32_t run = 0;
        if (v == prev) {
            run = 1;
            while (k + run < n && data_get(S, lane, k + run) == v) run++;
        }
        if (run >= 3) {
            uint32_t sym, eb, ev;
            length_code(run, sym, eb, ev);
            lds_add(&S.freq[sym], 1);
            S.tok[nt * kLanes + lane] = (uint16_t)(0x8000u | run);
            nt++;
            // Adler: `run` bytes of value v at positions k .. k + run - 1 (weights n - k, n - k - 1, ...)
            a += run * (uint32_t)v;
            b += (uint32_t)v * (run * (n - k) - run * (run - 1) / 2);
            k += run;
            any = true;
        } else {
            lds_add(&S.freq[v], 1);
            S.tok[nt * kLanes + lane] = (uint16_t)v;
            nt++;
            a += (uint32_t)v;
            b += (uint32_t)v * (n - k);
            prev = v;
            k++;
        }
    }
    S.ntok[lane] = nt;
    S.lane_a[lane] = a % kAdlerMod;
    S.lane_b[lane] = b % kAdlerMod;
    if (nt) lds_add(&S.ntot, nt);
    if (any) lds_or(&S.has_match, 1u);
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
        uint32_t r = 0;
        if (l)
            for (uint32_t t = 0; t < s; t++) r += al.len0[t] == l;
        al.rank0[s] = (uint16_t)r;
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
        uint32_t r = 0;
        for (uint32_t t = 0; t < s; t++) r += al.len[t] == l;
        al.code[s] = (uint16_t)bit_reverse(S.next_code[which][l] + r, l);
    }
}

// The block header's code lengths (HLIT + 257 literal/length lengths, then the one distance length) as code-length symbols: runs of
// zeros as 17 / 18, everything else literally.
BS_HD void ph_header_rle(uint32_t lane, Block &S, const Args &, uint32_t)
{
    if (lane != 0) return;
    uint32_t hl = kLL;
    while (hl > 257 && S.len[hl - 1] == 0) hl--;
    S.hlit_n = hl;
    const uint32_t n = hl + 1;   // + the distance alphabet's single length
    uint32_t nh = 0, i = 0;
    while (i < n) {
        const uint32_t v = i < hl ? S.len[i] : (S.has_match ? 1u : 0u);
        if (v == 0) {
            uint32_t run = 1;
            while (i + run < n && run < 138 && (i + run < hl ? S.len[i + run] : (S.has_match ? 1u : 0u)) == 0) run++;
            if (run >= 11) { S.hdr_sym[nh] = 18; S.hdr_ext[nh] = (uint8_t)(run - 11); }
            else if (run >= 3) { S.hdr_sym[nh] = 17; S.hdr_ext[nh] = (uint8_t)(run - 3); }
            else { run = 1; S.hdr_sym[nh] = 0; S.hdr_ext[nh] = 0; }
            S.cl_freq[S.hdr_sym[nh]]++;
            nh++;
            i += run;
        } else {
            S.hdr_sym[nh] = (uint8_t)v; S.hdr_ext[nh] = 0;
            S.cl_freq[v]++;
            nh++;
            i++;
        }
    }
    S.nhdr = nh;
}

BS_HD uint32_t cl_order(uint32_t i)
{
    const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[i];
}

BS_HD uint32_t token_bits(const Block &S, uint32_t t)
{
    if (!(t & 0x8000u)) return S.len[t];
    uint32_t sym, eb, ev;
    length_code(t & 0x7FFFu, sym, eb, ev);
    return S.len[sym] + eb + 1;   // + the distance code: one bit
}

BS_HD void ph_bitcount(uint32_t lane, Block &S, const Args &, uint32_t)
{
    uint32_t bits = 0;
    for (uint32_t i = 0; i < S.ntok[lane]; i++) bits += token_bits(S, S.tok[i * kLanes + lane]);
    S.lane_bits[lane] = bits;
}

BS_HD void ph_plan(uint32_t lane, Block &S, const Args &, uint32_t)
{
    if (lane != 0) return;
    uint32_t hc = kCL;
    while (hc > 4 && S.cl_len[cl_order(hc - 1)] == 0) hc--;
    S.hclen_n = hc;
    uint32_t bits = 3 + 5 + 5 + 4 + 3 * hc;
    for (uint32_t i = 0; i < S.nhdr; i++) {
        const uint32_t s = S.hdr_sym[i];
        bits += S.cl_len[s] + (s == 17 ? 3u : s == 18 ? 7u : 0u);
    }
    S.hdr_bits = bits;
    for (uint32_t j = 0; j < (uint32_t)kLanes; j++) {
        S.lane_off[j] = bits;
        bits += S.lane_bits[j];
    }
    S.eob_off = bits;
    bits += S.len[256];
    S.total_bits = bits;
    const uint32_t dyn = (bits + 3 + 7) / 8 + 4;   // + the empty stored block: 3 bits, padding to the byte, LEN, NLEN
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
        for (uint32_t i = 0; i < S.nhdr; i++) {
            const uint32_t s = S.hdr_sym[i];
            bw.put(S.cl_code[s], S.cl_len[s]);
            if (s == 17) bw.put(S.hdr_ext[i], 3);
            if (s == 18) bw.put(S.hdr_ext[i], 7);
        }
        bw.flush();
        BitWriter be(S.out, S.eob_off);
        be.put(S.code[256], S.len[256]);
        be.put(0, 3);                // the empty stored block: BFINAL = 0, BTYPE = 00
        be.flush();
        BitWriter bn(S.out, ((S.total_bits + 3 + 7) / 8 + 2) * 8);   // behind the padding: LEN = 0 (already there), NLEN = 0xFFFF
        bn.put(0xFFFFu, 16);
        bn.flush();
    }
    BitWriter bw(S.out, S.lane_off[lane]);
    for (uint32_t i = 0; i < S.ntok[lane]; i++) {
        const uint32_t t = S.tok[i * kLanes + lane];
        if (!(t & 0x8000u)) {
            bw.put(S.code[t], S.len[t]);
        } else {
            uint32_t sym, eb, ev;
            length_code(t & 0x7FFFu, sym, eb, ev);
            bw.put(S.code[sym], S.len[sym]);
            if (eb) bw.put(ev, eb);
            bw.put(0, 1);            // distance 1: the one distance code, one bit
        }
    }
    bw.flush();
}

// CRC-32 of the chunk ("IDAT" + data): every lane takes 1/64 of the bytes, shifts its CRC over what follows, XOR of all = the chunk's
BS_HD void ph_crc(uint32_t lane, Block &S, const Args &, uint32_t)
{
    const uint8_t *ob = reinterpret_cast<const uint8_t *>(S.out);
    const uint32_t m = 4 + S.data_len;
    const uint32_t per = (m + kLanes - 1) / kLanes;
    const uint32_t b0 = lane * per < m ? lane * per : m, b1 = b0 + per < m ? b0 + per : m;
    if (b0 == b1) return;
    const uint8_t type[4] = {'I', 'D', 'A', 'T'};
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = b0; i < b1; i++) c = crc_update_byte(c, i < 4 ? type[i] : ob[i - 4]);
    c ^= 0xFFFFFFFFu;
    lds_xor(&S.crc, crc_shift(c, m - b1));
}

BS_HD void ph_write(uint32_t lane, Block &S, const Args &A, uint32_t blk)
{
    const uint8_t *ob = reinterpret_cast<const uint8_t *>(S.out);
    uint8_t *slot = A.staging + (size_t)blk * kSlot;
    for (uint32_t i = lane; i < S.data_len; i += kLanes) slot[8 + i] = ob[i];
    if (lane == 0) {
        put_be32(slot, S.data_len);
        slot[4] = 'I'; slot[5] = 'D'; slot[6] = 'A'; slot[7] = 'T';
        put_be32(slot + 8 + S.data_len, S.crc);
        A.sizes[blk] = 12 + S.data_len;
        uint64_t a = 0, b = 0;   // Adler partial sums of the block from the lanes': X || Y -> (aX + aY, bX + nY aX + bY)
        for (uint32_t j = 0; j < (uint32_t)kLanes; j++) {
            const uint32_t n = lane_bytes(S, j);
            b = (b + (uint64_t)n * a + S.lane_b[j]) % kAdlerMod;
            a = (a + S.lane_a[j]) % kAdlerMod;
        }
        A.adler[2 * blk] = (uint32_t)a;
        A.adler[2 * blk + 1] = (uint32_t)b;
    }
}

// The phases of a block, in order: RUN(f) runs f on every lane, then a barrier.
#define BS_PNG_BLOCK_PROGRAM(RUN, RUN_ALPHABET)                                                       \
    RUN(ph_init) RUN(ph_load) RUN(ph_tokenize)                                                        \
    RUN_ALPHABET(ph_len_shannon, 0) RUN_ALPHABET(ph_len_rank, 0) RUN_ALPHABET(ph_len_counts, 0)       \
    RUN_ALPHABET(ph_len_assign, 0) RUN_ALPHABET(ph_len_codes, 0)                                      \
    RUN(ph_header_rle)                                                                                \
    RUN_ALPHABET(ph_len_shannon, 1) RUN_ALPHABET(ph_len_rank, 1) RUN_ALPHABET(ph_len_counts, 1)       \
    RUN_ALPHABET(ph_len_assign, 1) RUN_ALPHABET(ph_len_codes, 1)                                      \
    RUN(ph_bitcount) RUN(ph_plan) RUN(ph_emit) RUN(ph_crc) RUN(ph_write)

// ---- the frame: offsets of the chunks, Adler-32, the fixed chunks around them (png_finish: ONE wavefront) ---------------------------------
struct Finish {
    uint32_t lane_sum[kLanes];
    uint64_t lane_a[kLanes], lane_b[kLanes], lane_n[kLanes];
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

BS_HD void fin_sum(uint32_t lane, Finish &F, const FinishArgs &A)
{
    uint32_t b0, b1;
    fin_range(lane, A.n_blocks, b0, b1);
    uint32_t s = 0;
    uint64_t a = 0, b = 0, n = 0;
    for (uint32_t k = b0; k < b1; k++) {
        s += A.sizes[k];
        const uint64_t first = (uint64_t)k * kBlock;
        const uint64_t nk = A.total - first < (uint64_t)kBlock ? A.total - first : (uint64_t)kBlock;
        b = (b + (nk % kAdlerMod) * a + A.adler[2 * k + 1]) % kAdlerMod;
        a = (a + A.adler[2 * k]) % kAdlerMod;
        n += nk;
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
    uint32_t end = kHeadBytes;
    uint64_t a = 0, b = 0;
    for (uint32_t j = 0; j < (uint32_t)kLanes; j++) {
        end += F.lane_sum[j];
        b = (b + (F.lane_n[j] % kAdlerMod) * a + F.lane_b[j]) % kAdlerMod;
        a = (a + F.lane_a[j]) % kAdlerMod;
    }
    a = (a + 1) % kAdlerMod;                            // Adler-32 starts at A = 1, which every byte adds to B once
    b = (b + A.total % kAdlerMod) % kAdlerMod;
    uint8_t *o = A.out;
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
    put_be32(tail + 5, (uint32_t)((b << 16) | a));
    write_chunk(o + end, "IDAT", tail, 9);
    write_chunk(o + end + 21, "IEND", tail, 0);
    *A.file_bytes = (uint64_t)end + kTailBytes;
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
