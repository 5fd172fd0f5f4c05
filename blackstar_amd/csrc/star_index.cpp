// star_index.cpp -- flatten the star set into a pointer-free k-d array for the GPU.
//
// The reference keeps stars in kdt's `KdMap` (src/StarMap.hs:26,91: build toList), a balanced static k-d
// tree of boxed nodes whose split axis cycles x,y,z with depth, and queries it with `inRadius`
// (src/StarMap.hs:104).  Only the SET of stars within the radius affects the result, so the GPU index is
// free to choose its own node order: a left-balanced (complete) tree stored in 1-based Eytzinger order.
//   * children of node i are 2i and 2i+1, no pointers, no per-node axis (axis = depth % 3);
//   * nodes 1..2^L-1 are exactly the top L levels -> one contiguous block the kernel stages in LDS;
//   * a complete tree makes "index <= n" the only existence test.
#include <algorithm>
#include <cstdint>
#include <numeric>

#include "bs_internal.h"

namespace bs {

namespace {

// Size of the left subtree of a complete binary tree with n nodes.
size_t left_subtree_size(size_t n)
{
    if (n <= 1) return 0;
    size_t h = 0;  // floor(log2(n))
    while ((size_t(2) << h) <= n) h++;
    size_t full = (size_t(1) << h) - 1;       // nodes above the last level
    size_t last = n - full;                   // nodes on the last level
    size_t half = size_t(1) << (h - 1);       // capacity of the left half of the last level
    return (half - 1) + std::min(last, half);
}

struct Builder {
    const bs_star *stars;
    std::vector<uint32_t> order;
    std::vector<StarNode> *nodes;
    std::vector<StarColor> *colors;

    double coord(uint32_t id, int axis) const { return axis == 0 ? stars[id].x : (axis == 1 ? stars[id].y : stars[id].z); }

    void build(size_t node, size_t lo, size_t hi, int axis)
    {
        // iterative on the right child to bound recursion depth to log2(n)
        while (lo < hi) {
            size_t n = hi - lo;
            size_t mid = lo + left_subtree_size(n);
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t a, uint32_t b) {
                double ca = coord(a, axis), cb = coord(b, axis);
                return ca < cb || (ca == cb && a < b);  // deterministic under ties
            });
            uint32_t id = order[mid];
            StarNode &nd = (*nodes)[node];
            nd.x = stars[id].x; nd.y = stars[id].y; nd.z = stars[id].z;
            nd.mag = stars[id].mag;
            nd.id = (int32_t)id;
            (*colors)[node] = StarColor{stars[id].hue, stars[id].sat};
            int next = axis == 2 ? 0 : axis + 1;
            build(2 * node, lo, mid, next);
            node = 2 * node + 1;
            lo = mid + 1;
            axis = next;
        }
    }
};

}  // namespace

void build_star_index(const bs_star *stars, size_t n, std::vector<StarNode> &nodes, std::vector<StarColor> &colors, std::vector<double> &splits)
{
    nodes.assign(n + 1, StarNode{0, 0, 0, 0, -1});
    colors.assign(n + 1, StarColor{0, 0});
    splits.assign(n + 1, 0.0);
    if (n == 0) return;
    Builder b{stars, {}, &nodes, &colors};
    b.order.resize(n);
    std::iota(b.order.begin(), b.order.end(), 0u);
    b.build(1, 0, n, 0);
    // split coordinate of node i = its point's coordinate along axis depth(i) % 3, depth(i) = floor(log2 i)
    for (size_t i = 1; i <= n; i++) {
        int depth = 0;
        for (size_t t = i; t > 1; t >>= 1) depth++;
        const StarNode &nd = nodes[i];
        splits[i] = depth % 3 == 0 ? nd.x : (depth % 3 == 1 ? nd.y : nd.z);
    }
}

}  // namespace bs
