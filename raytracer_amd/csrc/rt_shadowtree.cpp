// Host-side build of a shadow tree (rtamd_tree::shadow_root): a kd-tree over triangle boxes for
// shadow queries that any hit decides.  Such a query only asks whether some triangle on the ray
// is hit, so the tree it walks need not be the reference's (kdtree.py:27-131 builds that one, and
// rt_kdbuild.hip reproduces it); this one is built for the walk: surface-area heuristic over the
// faces of the boxes, small leaves, triangles whose box straddles a plane on both sides.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace {

struct Builder {
    const double* lo;           // [n][3]
    const double* hi;
    int32_t max_depth, leaf_rows, max_leaf_rows;
    double step_cost;
    std::vector<int32_t> axis;  // per node: 0..2, or -1 for a leaf
    std::vector<double> split;
    std::vector<int32_t> link;  // interior: index of the right child; leaf: its number
    std::vector<int64_t> leaf_offset;
    std::vector<int32_t> leaf_rows_out;
    int32_t depth_reached = 0;
    bool overflow = false;
    std::vector<double> sorted_lo, sorted_hi;

    void build(const std::vector<int32_t>& idx, const double cell_lo[3], const double cell_hi[3], int32_t depth) {
        if (overflow) return;
        const size_t n = idx.size();
        const size_t me = axis.size();
        depth_reached = std::max(depth_reached, depth);
        double best_cost = std::numeric_limits<double>::infinity(), best_plane = 0;
        int best_axis = -1;
        if ((int64_t)n > leaf_rows && depth < max_depth) {
            const double d[3] = {cell_hi[0] - cell_lo[0], cell_hi[1] - cell_lo[1], cell_hi[2] - cell_lo[2]};
            const double whole = std::max(d[0] * d[1] + d[1] * d[2] + d[2] * d[0], 1e-300);
            sorted_lo.resize(n);
            sorted_hi.resize(n);
            for (int ax = 0; ax < 3; ++ax) {
                for (size_t i = 0; i < n; ++i) {
                    sorted_lo[i] = lo[(size_t)idx[i] * 3 + ax];
                    sorted_hi[i] = hi[(size_t)idx[i] * 3 + ax];
                }
                std::sort(sorted_lo.begin(), sorted_lo.end());
                std::sort(sorted_hi.begin(), sorted_hi.end());
                const double d1 = d[(ax + 1) % 3], d2 = d[(ax + 2) % 3];
                // candidate planes: every box face inside the cell, swept in ascending order with
                // the two counts kept along; the lowest of equally cheap planes wins, a lower
                // axis wins ties between axes
                size_t pl = 0, ph = 0, below = 0, not_above = 0;      // next low / high face; lo < c; hi <= c
                while (pl < n || ph < n) {
                    const double c = (ph >= n || (pl < n && sorted_lo[pl] <= sorted_hi[ph])) ? sorted_lo[pl] : sorted_hi[ph];
                    while (pl < n && sorted_lo[pl] == c) ++pl;
                    while (ph < n && sorted_hi[ph] == c) ++ph;
                    while (below < n && sorted_lo[below] < c) ++below;
                    while (not_above < n && sorted_hi[not_above] <= c) ++not_above;
                    if (!(c > cell_lo[ax]) || !(c < cell_hi[ax])) continue;
                    const size_t n_left = below, n_right = n - not_above;
                    if (n_left == n && n_right == n) continue;
                    // surface of a box that is x long on this axis: x * (d1 + d2) + d1 * d2
                    const double cost = step_cost + (((c - cell_lo[ax]) * (d1 + d2) + d1 * d2) * (double)n_left
                                                     + ((cell_hi[ax] - c) * (d1 + d2) + d1 * d2) * (double)n_right) / whole;
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_plane = c; }
                }
            }
        }
        if (best_axis < 0 || (best_cost >= (double)n && (int64_t)n <= max_leaf_rows)) {
            if ((int64_t)n > max_leaf_rows) { overflow = true; return; }
            axis.push_back(-1);
            split.push_back(-1.0);
            link.push_back((int32_t)(leaf_offset.size() - 1));
            leaf_rows_out.insert(leaf_rows_out.end(), idx.begin(), idx.end());
            leaf_offset.push_back((int64_t)leaf_rows_out.size());
            return;
        }
        axis.push_back(best_axis);
        split.push_back(best_plane);
        link.push_back(0);
        std::vector<int32_t> left, right;
        for (int32_t i : idx) {
            if (lo[(size_t)i * 3 + best_axis] < best_plane) left.push_back(i);
            if (hi[(size_t)i * 3 + best_axis] > best_plane) right.push_back(i);
        }
        double left_hi[3] = {cell_hi[0], cell_hi[1], cell_hi[2]}, right_lo[3] = {cell_lo[0], cell_lo[1], cell_lo[2]};
        left_hi[best_axis] = best_plane;
        right_lo[best_axis] = best_plane;
        build(left, cell_lo, left_hi, depth + 1);
        std::vector<int32_t>().swap(left);
        link[me] = (int32_t)axis.size();
        build(right, right_lo, cell_hi, depth + 1);
    }
};

}  // namespace

extern "C" int rtamd_shadow_tree_build(const double* box_lo, const double* box_hi, int64_t n_boxes, int32_t max_depth,
                                       int32_t leaf_rows, int32_t max_leaf_rows, double step_cost,
                                       int32_t* node_axis, double* node_split, int32_t* node_link, int64_t max_nodes,
                                       int64_t* leaf_offset, int32_t* leaf_row, int64_t max_rows,
                                       int64_t* n_nodes, int64_t* n_leaves, int64_t* n_rows, int32_t* depth) {
    if (n_nodes) *n_nodes = 0;
    if (n_rows) *n_rows = 0;
    if (!box_lo || !box_hi || n_boxes <= 0 || n_boxes > 0x7fffffffLL || !n_nodes || !n_leaves || !n_rows || !depth ||
        max_depth < 0 || leaf_rows < 1 || max_leaf_rows < leaf_rows)
        return RTAMD_EINVAL;
    try {
    Builder b;
    b.lo = box_lo;
    b.hi = box_hi;
    b.max_depth = max_depth;
    b.leaf_rows = leaf_rows;
    b.max_leaf_rows = max_leaf_rows;
    b.step_cost = step_cost;
    b.leaf_offset.push_back(0);
    std::vector<int32_t> all((size_t)n_boxes);
    double cell_lo[3], cell_hi[3];
    for (int ax = 0; ax < 3; ++ax) { cell_lo[ax] = box_lo[ax]; cell_hi[ax] = box_hi[ax]; }
    for (int64_t i = 0; i < n_boxes; ++i) {
        all[(size_t)i] = (int32_t)i;
        for (int ax = 0; ax < 3; ++ax) {
            cell_lo[ax] = std::min(cell_lo[ax], box_lo[i * 3 + ax]);
            cell_hi[ax] = std::max(cell_hi[ax], box_hi[i * 3 + ax]);
        }
    }
    b.build(all, cell_lo, cell_hi, 0);
    if (b.overflow) return RTAMD_EUNSUPPORTED;          // some leaf keeps more than max_leaf_rows rows
    *n_nodes = (int64_t)b.axis.size();
    *n_leaves = (int64_t)b.leaf_offset.size() - 1;
    *n_rows = (int64_t)b.leaf_rows_out.size();
    *depth = b.depth_reached;
    if (!node_axis || !node_split || !node_link || !leaf_offset || !leaf_row) return RTAMD_OK;       // sizes only
    if (*n_nodes > max_nodes || *n_rows > max_rows) return RTAMD_EINVAL;
    std::copy(b.axis.begin(), b.axis.end(), node_axis);
    std::copy(b.split.begin(), b.split.end(), node_split);
    std::copy(b.link.begin(), b.link.end(), node_link);
    std::copy(b.leaf_offset.begin(), b.leaf_offset.end(), leaf_offset);
    std::copy(b.leaf_rows_out.begin(), b.leaf_rows_out.end(), leaf_row);
    return RTAMD_OK;
    } catch (...) {             // the builder's lists are std::vectors: out of host memory
        return RTAMD_EINVAL;
    }
}
