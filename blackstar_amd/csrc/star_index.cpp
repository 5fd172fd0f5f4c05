// star_index.cpp -- bin the star set into the cube-map direction grid the kernel queries (bs_internal.h).
//
// The reference keeps stars in kdt's `KdMap` (src/StarMap.hs:26,91: build toList) and queries it with
// `inRadius tree (3*w) nvel` (src/StarMap.hs:104).  Only the SET of stars within the radius affects the result, so
// the GPU index is free to choose its own structure: here a counting sort of the stars by (face, v cell, u cell) of
// their direction, plus a copy of every star into each neighbouring face whose border cells its neighbourhood can
// reach.  Entries of one cell are ordered by the caller's star index, so the result does not depend on input order
// beyond that.
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "bs_internal.h"

namespace bs {

namespace {

// cos a, cos b and the sector of toPixelRGB (PixelHSI h' s i) for hue h' in [0,1)  (massiv-io):
// the same expressions as host_hsi_to_rgb (host_math.cpp), evaluated once per star.
StarColor star_color(double hue, double sat)
{
    const double pi = 3.141592653589793;
    const double h = hue * 2 * pi;
    const int k = (h < 2 * pi / 3) ? 0 : ((h < 4 * pi / 3) ? 1 : 2);
    const double a = k == 0 ? h : (k == 1 ? h - 2 * pi / 3 : h - 4 * pi / 3);
    const double b = k == 0 ? pi / 3 - h : (k == 1 ? h + pi : 2 * pi - pi / 3 - h);
    return StarColor{std::cos(a), std::cos(b), sat, k, 0};
}

struct Placement {
    uint32_t cell, star;
};

}  // namespace

void build_star_index(const bs_star *stars, size_t n, std::vector<StarNode> &nodes, std::vector<StarColor> &colors, std::vector<uint32_t> &cell_start)
{
    std::vector<Placement> placed;
    placed.reserve(n + n / 16);
    const double lim = 1.0 + kGridDelta;
    for (size_t i = 0; i < n; i++) {
        const double d[3] = {stars[i].x, stars[i].y, stars[i].z};
        if (!(std::isfinite(d[0]) && std::isfinite(d[1]) && std::isfinite(d[2]))) continue;  // can never be within the radius
        // linear's normalize leaves a vector with |v|^2 <= 1e-12 as it is, so such a (degenerate) query is NOT a unit
        // vector: it can only reach stars within the radius of the origin.  Those go to one extra list after the cells.
        if ((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] <= kOriginReach * kOriginReach) placed.push_back(Placement{(uint32_t)kGridCells, (uint32_t)i});
        for (int f = 0; f < 6; f++) {
            const int a = f >> 1;
            const double m = (f & 1) ? -d[a] : d[a];
            if (!(m > 0.0)) continue;  // the zero vector has no direction (a unit query is 1 > radius away from it)
            const double u = d[(a + 1) % 3] / m, v = d[(a + 2) % 3] / m;
            if (!(std::fabs(u) <= lim && std::fabs(v) <= lim)) continue;
            placed.push_back(Placement{(uint32_t)((f * kGridG + grid_cell(v)) * kGridG + grid_cell(u)), (uint32_t)i});
        }
    }
    // counting sort by cell; `placed` is in star order, so entries of a cell come out in star order
    cell_start.assign((size_t)kGridCells + 2, 0u);  // cells, the origin list, end
    for (const Placement &p : placed) cell_start[p.cell + 1]++;
    for (size_t c = 0; c <= (size_t)kGridCells; c++) cell_start[c + 1] += cell_start[c];
    nodes.assign(placed.size(), StarNode{0, 0, 0, 0, -1});
    colors.assign(placed.size(), StarColor{0, 0, 0, 0, 0});
    std::vector<uint32_t> fill(cell_start.begin(), cell_start.end() - 1);
    for (const Placement &p : placed) {
        const uint32_t k = fill[p.cell]++;
        const bs_star &s = stars[p.star];
        nodes[k] = StarNode{s.x, s.y, s.z, s.mag, (int32_t)p.star};
        colors[k] = star_color(s.hue, s.sat);
    }
}

}  // namespace bs
