// Internal types shared by the host side (context.cpp, render.cpp, post.cpp, batch.cpp, star_index.cpp) and the gfx950 kernels
// (trace_device.h / trace_kernel.hip, post_kernels.hip, png_kernels.hip).  Not part of the ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/blackstar_gpu.h"

struct bs_ray_record;  // include/blackstar_gpu_debug.h (test hook)

namespace bs {

// One entry of the star grid: 32 B, loaded as two dwordx4.
struct alignas(32) StarNode {
    double x, y, z;
    int32_t mag;  // magnitude * 100 (StarMap.hs:57)
    int32_t id;   // index into the caller's star array (for tests / canonical ordering)
};

// What a HIT needs besides d^2 and mag, indexed like StarNode: the hue enters toPixelRGB (PixelHSI h s i) only through
// cos a / cos b of its sector (massiv-io), which do not depend on the ray -- the host evaluates the two
// cosines once per star (libm, like the reference) and the kernel keeps the reference's i + is*ca/cb arithmetic.
struct alignas(32) StarColor {
    double ca, cb;   // cos a, cos b of the hue's sector
    double sat;      // starColor' saturation (StarMap.hs:61-72), before the scene's starSaturation factor
    int32_t sector;  // 0, 1, 2: which of (r,g,b) is `first`
    int32_t pad_;
};

// ---- star index: a cube-map grid of directions -------------------------------------------------------------------
// The reference queries kdt's `inRadius tree (3*w) nvel` (StarMap.hs:104): all stars p with |p - nvel|^2 <= r^2,
// r = 0.0015.  Only the SET matters, so the index is free to be anything that returns that set.  A star within chord
// distance r of a unit vector q lies within the angle asin(r) of q's direction, whatever the star's own length, so
// the index bins DIRECTIONS: the cube face of the largest |component| (face = 2*axis + (component < 0)) and the
// gnomonic coordinates u = a/|m|, v = b/|m| of the other two components (cyclic order), cut into kGridG x kGridG cells
// per face, cell rows contiguous in u.  On a face (|u|,|v| <= 1 + kGridDelta) the coordinate u moves by at most
// sqrt(1+u^2)*sqrt(1+u^2+v^2) <= 2.47 per radian, so everything within asin(r) of q projects into the box
// [u-D, u+D] x [v-D, v+D], D = kGridDelta = 0.0039 (2.6 r: 4 % slack for curvature and the kernel's approximate
// reciprocal).  Stars whose box neighbourhood reaches over a face edge are ALSO listed in that neighbouring face (in
// its clamped border cells), so a query only ever reads its own face: at most 2 x 2 cells (2 D <= cell width), two
// contiguous runs of entries.  6 * 256^2 cell offsets = 1.5 MB stay in every XCD's L2; a lookup is ~40 VALU of
// addressing plus ~12 per candidate star (about 5), against ~25 dependent node visits in a k-d descent.
constexpr double kStarW = 0.0005;                 // StarMap.hs:103
constexpr double kStarRadius = 3 * kStarW;        // StarMap.hs:104
constexpr int kGridG = 256;                       // cells per face edge: width 2/256 = 0.0078125 >= 2 * kGridDelta
constexpr double kGridDelta = 0.0039;
constexpr int kGridCells = 6 * kGridG * kGridG;
constexpr double kOriginReach = 0.0016;           // > kStarRadius + 1e-6: stars a non-normalised (|v|^2 <= 1e-12) query can reach
constexpr int kCounters = 8;                        // steps, capped, horizon, escaped, disk_hits, star_hits, wave_iters, tile queue head

// Everything the trace kernel needs, passed by value as the kernel argument (lands in SGPRs / kernarg).
struct TraceParams {
    double cam[3];
    double xa[3], ya[3], za[3];  // look-at basis, computed once on the host with the reference's op order
    double fov, W, H;            // traced resolution as doubles (cfg' of Raytracer.hs:63)
    double inv_W, inv_H;         // 1.0 / W, 1.0 / H correctly rounded (host): generate_ray's divisions by W and H as three instructions each (trace_device.h div_by);
                                 // both 0 when the camera holds a tiny non-zero number (|x| < 2^-300): the device then divides the compiler's way
    double h, hh, h6;            // stepSize, h/2, h/6
    double hh2, hhh, h2_6;       // FAST mode regrouping: h^2/4, h^2/2, h^2/6
    double e1[3], rcam;          // FAST mode orbital-plane frame: cam/|cam| and |cam|
    double safe, in2, out2;      // safeDistance, diskInner^2, diskOuter^2 (Raytracer.hs:59-62)
    double rI, rO;               // sqrt in2, sqrt out2 (Raytracer.hs:107-108)
    double disk_rgb[3];          // toPixelRGB diskColor (:65)
    double disk_opacity;
    double star_intensity, star_saturation;
    double star_a;               // log 2 / 50 (StarMap.hs:108)
    int32_t wt, ht;              // traced resolution
    int32_t band_t0, band_t1;    // traced rows [band_t0, band_t1) this launch renders (0, ht for a whole frame); `out` is the band's first row
    int32_t ss;                  // supersampling
    int32_t out_w, out_h;
    int32_t max_steps;
    int32_t guard_steps;         // FAST mode: rays that take more steps than this are re-traced with STRICT arithmetic (derive_params)
    int32_t n_entries;           // entries in nodes/colors (stars + border duplicates)
    int32_t grid_blocks;         // persistent workgroups launched (<= 4 per CU)
    int32_t stagger_cycles;      // first-tile phase offset per SIMD slot, in shader cycles (0 = off)
    int32_t blocks_per_slot;     // workgroups per residency slot (= CUs): workgroup b sits in slot b / blocks_per_slot
    int32_t disk_slots;          // LDS crossing-queue depth in use (<= 4; tests shrink it to force the overflow path)
    int32_t queue_base;          // 0: every tile comes off the queue.  4 * grid_blocks: wavefront g of the grid starts on tile g without an atomic and the queue hands out the tiles from there on (short launches: render.cpp fill_params)
    int32_t late_pop_slot;       // wavefronts of residency slots >= this pop their next tile AFTER tracing the current one (0: all of them -- nobody holds an untouched tile): trace_kernel.hip
    const StarNode *nodes;       // device, n_entries, sorted by cell
    const StarColor *colors;     // device, n_entries
    const uint32_t *cell_start;  // device, kGridCells + 2: entries of cell c are [cell_start[c], cell_start[c+1]); cell kGridCells = origin list
    double *out;                 // device, out_h * out_w * 3
    unsigned long long *counters;  // device, kCounters
};

// context.cpp: the catch-all of every `extern "C"` entry point (each is a function-try-block): no C++ exception crosses the ABI.  Must be
// called from inside a catch handler; sets the thread's bs_last_error() and returns BS_ENOMEM for std::bad_alloc, else BS_EINTERNAL.
int abi_exception(const char *where) noexcept;

// Host-side strict math (host_math.cpp, compiled -ffp-contract=off).
void host_hsi_to_rgb(double hue, double s, double i, double rgb[3], bool *ok);
// writeImg's pixel map x -> toWord8 (sRGB x) (Raytracer.hs:23-32) as its 255 thresholds: T[k] (k = 1..255) = the smallest double
// the map sends to a byte >= k, found by bisection with the host's libm `pow`; T[0] = -inf, T[256] = +inf.
void srgb8_thresholds(double T[257]);
// Fills every derived field of TraceParams except the device pointers.  Returns false + message on bad input.
bool derive_params(const bs_config &cfg, TraceParams &p, std::string &err);

// star_index.cpp: bin the caller's stars into the cube-map grid (entries sorted by cell, border duplicates included).
void build_star_index(const bs_star *stars, size_t n, std::vector<StarNode> &nodes, std::vector<StarColor> &colors, std::vector<uint32_t> &cell_start);
// The cell coordinate of a gnomonic coordinate t (shared by the builder and, with the same operations, the kernel).
inline int grid_cell(double t)
{
    double c = (t + 1.0) * (0.5 * kGridG);
    if (!(c > 0.0)) return 0;
    if (c >= (double)kGridG) return kGridG - 1;
    return (int)c;
}

// trace_kernel.hip launchers (enqueue on `stream`, no sync).
int launch_trace(const TraceParams &p, int mode, void *stream);
int launch_star_lookup(const TraceParams &p, const double *d_dirs, size_t n, double *d_rgb, int32_t *d_hits, void *stream);
// debug_kernels.hip (libblackstar_gpu_debug.so only)
int launch_trace_records(const TraceParams &p, int mode, const int32_t *d_yx, size_t n_rays, bs_ray_record *d_out, void *stream);
int launch_ubench(int kind, int blocks, int iters, double *d_out, void *stream);
int launch_sqrt_div(const double *d_a, const double *d_b, size_t n, double *d_sqrt, double *d_div, int bare, void *stream);
// post_kernels.hip
int launch_bloom(const double *d_in, double *d_out, double *d_a, double *d_b, int w, int h, double strength, int divider, int n_cu, void *stream);
// bloom + writeImg's pixel map in one go: only RGB8 is written (d_table: the 257 sRGB8 thresholds, srgb8_thresholds)
int launch_bloom_srgb8(const double *d_in, unsigned char *d_out_u8, double *d_a, double *d_b, int w, int h, double strength, int divider, int n_cu,
                       const double *d_table, void *stream);
int launch_supersample(const double *d_in, double *d_out, int w2, int h2, void *stream);
int launch_srgb8(const double *d_in, unsigned char *d_out, size_t n, const double *d_table, void *stream);
// png_kernels.hip: writeImg's file format on the device (algorithm: png_block.h)
uint64_t png_file_bound(int w, int h);   // bytes a w x h RGB8 frame can take at most as a file of this encoder
size_t png_scratch_bytes(int w, int h);  // device scratch one encode needs
size_t png_block_count(int w, int h);     // 8 KiB blocks of the filtered stream = workgroups of the encoding kernel
constexpr int kPngPhases = 23;            // clock stamps per block of the profiling variant: before the first phase, after each of the 22
int launch_png_encode(const unsigned char *d_rgb8, int w, int h, void *d_scratch, unsigned char *d_out, uint64_t *d_file_bytes, void *stream,
                      unsigned long long *d_clocks = nullptr);

}  // namespace bs
