// Internal types shared by the host side (bs_api.cpp, star_index.cpp) and the gfx950 kernels
// (trace_kernel.hip).  Not part of the ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/blackstar_gpu.h"

namespace bs {

// One node of the flat k-d array: 32 B, loaded as two dwordx4 (or one ds_read_b128 pair from LDS).
// Layout: 1-based Eytzinger order of a left-balanced k-d tree (children of i are 2i, 2i+1; the split
// axis of a node at depth d is d % 3, mirroring kdt's `cycle (pointAsList q)`), so the top L levels are
// the first 2^L - 1 entries and can be staged in LDS as one contiguous block.
struct alignas(32) StarNode {
    double x, y, z;
    int32_t mag;  // magnitude * 100 (StarMap.hs:57)
    int32_t id;   // index into the caller's star array (for tests / canonical ordering)
};

struct alignas(16) StarColor {
    double hue, sat;  // starColor' (StarMap.hs:61-72), indexed like StarNode
};

// The traversal only ever needs a node's SPLIT coordinate (8 B); they live in their own Eytzinger-ordered f64
// array (3.8 MB for 470 k stars -> resident in each XCD's 4 MB L2, where the 15 MB node array is not), and the
// top kLdsLevels levels of it are staged in LDS.  The full 32-B node is read only when the query ball reaches
// the node's splitting plane.
#ifndef BS_LDS_LEVELS
#define BS_LDS_LEVELS 10
#endif
constexpr int kLdsLevels = BS_LDS_LEVELS;                      // top levels of the split array staged in LDS
constexpr int kLdsNodes = (1 << kLdsLevels) - 1;    // 1023 splits * 8 B = 8184 B per workgroup (4 workgroups/CU with the per-lane scratch)
constexpr int kCounters = 8;                        // steps, capped, horizon, escaped, disk_hits, star_hits, wave_iters, tile queue head

// Everything the trace kernel needs, passed by value as the kernel argument (lands in SGPRs / kernarg).
struct TraceParams {
    double cam[3];
    double xa[3], ya[3], za[3];  // look-at basis, computed once on the host with the reference's op order
    double fov, W, H;            // traced resolution as doubles (cfg' of Raytracer.hs:63)
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
    int32_t ss;                  // supersampling
    int32_t out_w, out_h;
    int32_t max_steps;
    int32_t n_stars;
    int32_t lds_nodes;           // min(n_stars, kLdsNodes)
    int32_t grid_blocks;         // persistent workgroups launched (<= 4 per CU)
    int32_t stagger_cycles;      // first-tile phase offset per SIMD slot, in shader cycles (0 = off)
    int32_t blocks_per_slot;     // workgroups per residency slot (= CUs): workgroup b sits in slot b / blocks_per_slot
    int32_t disk_slots;          // LDS crossing-queue depth in use (<= 4; tests shrink it to force the overflow path)
    const StarNode *nodes;       // device, n_stars + 1 entries (entry 0 unused)
    const double *splits;        // device, n_stars + 1 entries: the coordinate of node i along axis depth(i) % 3
    const StarColor *colors;     // device, n_stars + 1 entries
    double *out;                 // device, out_h * out_w * 3
    unsigned long long *counters;  // device, kCounters
};

// Host-side strict math (host_math.cpp, compiled -ffp-contract=off).
void host_hsi_to_rgb(double hue, double s, double i, double rgb[3], bool *ok);
// Fills every derived field of TraceParams except the device pointers.  Returns false + message on bad input.
bool derive_params(const bs_config &cfg, TraceParams &p, std::string &err);

// star_index.cpp: build the 1-based Eytzinger left-balanced k-d array from the caller's stars.
void build_star_index(const bs_star *stars, size_t n, std::vector<StarNode> &nodes, std::vector<StarColor> &colors, std::vector<double> &splits);

// trace_kernel.hip launchers (enqueue on `stream`, no sync).
int launch_trace(const TraceParams &p, int mode, void *stream);
int launch_trace_records(const TraceParams &p, int mode, const int32_t *d_yx, size_t n_rays, bs_ray_record *d_out, void *stream);
int launch_star_lookup(const TraceParams &p, const double *d_dirs, size_t n, double *d_rgb, int32_t *d_hits, void *stream);
// post_kernels.hip
int launch_bloom(const double *d_in, double *d_out, double *d_a, double *d_b, int w, int h, double strength, int divider, void *stream);
int launch_supersample(const double *d_in, double *d_out, int w2, int h2, void *stream);
int launch_srgb8(const double *d_in, unsigned char *d_out, size_t n, void *stream);
int launch_ubench(int kind, int blocks, int iters, double *d_out, void *stream);
int launch_sqrt_div(const double *d_a, const double *d_b, size_t n, double *d_sqrt, double *d_div, int bare, void *stream);

}  // namespace bs
