// Host-side build of a shadow tree (rtamd_tree::shadow_root): a kd-tree over triangles for shadow
// queries that any hit decides.  Such a query only asks whether some triangle on the ray is hit, so
// the tree it walks need not be the reference's (kdtree.py:27-131 builds that one, and
// rt_kdbuild.hip reproduces it); this one is built for the walk: surface-area heuristic over the
// faces of the triangles' boxes, small leaves, a triangle on both sides of every plane it reaches
// across.
//
// With the triangles' corners at hand the build splits "perfectly": the box a triangle brings to a
// node is not the box of the whole triangle but of the part of it that lies inside the node's cell
// (the triangle clipped against the cell widened by the caller's slack, that part's box widened by
// the slack again).  A triangle whose box reaches across a plane while the triangle itself stays on
// one side is then listed on that side only, candidate planes sit where the geometry inside the cell
// ends, and leaf boxes close around what the leaf's cell really holds.  What the walk needs still
// holds: a point of a triangle within the slack of a leaf's cell is within that leaf's box, and the
// triangle is on the leaf's list.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace {

struct Ref {
    int32_t tri;
    double lo[3], hi[3];
};

struct Builder {
    const double* verts;        // [n][9] corners, or null: boxes are split as boxes
    double slack;
    int32_t max_depth, leaf_rows, max_leaf_rows;
    double step_cost;
    std::vector<int32_t> axis;  // per node: 0..2, or -1 for a leaf
    std::vector<double> split;
    std::vector<int32_t> link;  // interior: index of the right child; leaf: its number
    std::vector<int64_t> leaf_offset;
    std::vector<int32_t> leaf_rows_out;
    std::vector<double> leaf_box;       // [n_leaves][6]: lo.xyz hi.xyz around what the leaf lists (zeros for an empty leaf)
    int32_t depth_reached = 0;
    bool overflow = false;
    std::vector<double> sorted_lo, sorted_hi;

    // The box of triangle `tri` inside the cell [clo, chi] widened by the slack: Sutherland-Hodgman
    // clipping of the triangle against the six planes, the box of what is left widened by the slack
    // and kept inside `keep` (the box the triangle came with).  False: nothing of the triangle is
    // within the slack of the cell.
    bool clipped_box(int32_t tri, const double clo[3], const double chi[3], const Ref& keep, Ref& out) const {
        double poly[16][3], next[16][3];
        int n = 3;
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) poly[k][c] = verts[(size_t)tri * 9 + 3 * k + c];
        for (int ax = 0; ax < 3 && n > 0; ++ax) {
            for (int side = 0; side < 2 && n > 0; ++side) {
                const double plane = side == 0 ? clo[ax] - slack : chi[ax] + slack;
                int m = 0;
                for (int i = 0; i < n; ++i) {
                    const double* a = poly[i];
                    const double* b = poly[(i + 1) % n];
                    const bool a_in = side == 0 ? a[ax] >= plane : a[ax] <= plane;
                    const bool b_in = side == 0 ? b[ax] >= plane : b[ax] <= plane;
                    if (a_in) { for (int c = 0; c < 3; ++c) next[m][c] = a[c]; ++m; }
                    if (a_in != b_in) {
                        const double t = (plane - a[ax]) / (b[ax] - a[ax]);
                        for (int c = 0; c < 3; ++c) next[m][c] = c == ax ? plane : a[c] + t * (b[c] - a[c]);
                        ++m;
                    }
                }
                n = m;
                for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) poly[i][c] = next[i][c];
            }
        }
        if (n == 0) return false;
        out.tri = tri;
        for (int c = 0; c < 3; ++c) {
            double lo = poly[0][c], hi = poly[0][c];
            for (int i = 1; i < n; ++i) { lo = std::min(lo, poly[i][c]); hi = std::max(hi, poly[i][c]); }
            // widened as the caller widened the whole triangle's box (slack, and a few ulps for the clipping's own rounding)
            const double pad = slack + 1e-12 * (1.0 + std::fabs(lo) + std::fabs(hi));
            out.lo[c] = std::max(keep.lo[c], lo - pad);
            out.hi[c] = std::min(keep.hi[c], hi + pad);
        }
        return true;
    }

    void build(const std::vector<Ref>& refs, const double cell_lo[3], const double cell_hi[3], int32_t depth) {
        if (overflow) return;
        const size_t n = refs.size();
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
                    sorted_lo[i] = refs[i].lo[ax];
                    sorted_hi[i] = refs[i].hi[ax];
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
            double box[6] = {0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < n; ++i) {
                leaf_rows_out.push_back(refs[i].tri);
                for (int c = 0; c < 3; ++c) {
                    box[c] = i == 0 ? refs[i].lo[c] : std::min(box[c], refs[i].lo[c]);
                    box[3 + c] = i == 0 ? refs[i].hi[c] : std::max(box[3 + c], refs[i].hi[c]);
                }
            }
            leaf_box.insert(leaf_box.end(), box, box + 6);
            leaf_offset.push_back((int64_t)leaf_rows_out.size());
            return;
        }
        axis.push_back(best_axis);
        split.push_back(best_plane);
        link.push_back(0);
        double left_hi[3] = {cell_hi[0], cell_hi[1], cell_hi[2]}, right_lo[3] = {cell_lo[0], cell_lo[1], cell_lo[2]};
        left_hi[best_axis] = best_plane;
        right_lo[best_axis] = best_plane;
        std::vector<Ref> left, right;
        for (const Ref& r : refs) {
            const bool to_left = r.lo[best_axis] < best_plane, to_right = r.hi[best_axis] > best_plane;
            if (to_left && to_right && verts) {
                // the triangle's box reaches across the plane: each side gets the box of the part on its side, or nothing
                Ref part;
                if (clipped_box(r.tri, cell_lo, left_hi, r, part)) left.push_back(part);
                if (clipped_box(r.tri, right_lo, cell_hi, r, part)) right.push_back(part);
                continue;
            }
            if (to_left) left.push_back(r);
            if (to_right) right.push_back(r);
        }
        build(left, cell_lo, left_hi, depth + 1);
        std::vector<Ref>().swap(left);
        link[me] = (int32_t)axis.size();
        build(right, right_lo, cell_hi, depth + 1);
    }
};

}  // namespace

extern "C" int rtamd_shadow_tree_build(const double* box_lo, const double* box_hi, int64_t n_boxes, int32_t max_depth,
                                       int32_t leaf_rows, int32_t max_leaf_rows, double step_cost,
                                       const double* tri_corners, double slack,
                                       int32_t* node_axis, double* node_split, int32_t* node_link, int64_t max_nodes,
                                       int64_t* leaf_offset, int32_t* leaf_row, double* leaf_box, int64_t max_rows,
                                       int64_t* n_nodes, int64_t* n_leaves, int64_t* n_rows, int32_t* depth) {
    if (n_nodes) *n_nodes = 0;
    if (n_rows) *n_rows = 0;
    if (!box_lo || !box_hi || n_boxes <= 0 || n_boxes > 0x7fffffffLL || !n_nodes || !n_leaves || !n_rows || !depth ||
        max_depth < 0 || leaf_rows < 1 || max_leaf_rows < leaf_rows || !(slack >= 0.0))
        return RTAMD_EINVAL;
    try {
    Builder b;
    b.verts = tri_corners;
    b.slack = slack;
    b.max_depth = max_depth;
    b.leaf_rows = leaf_rows;
    b.max_leaf_rows = max_leaf_rows;
    b.step_cost = step_cost;
    b.leaf_offset.push_back(0);
    std::vector<Ref> all((size_t)n_boxes);
    double cell_lo[3], cell_hi[3];
    for (int ax = 0; ax < 3; ++ax) { cell_lo[ax] = box_lo[ax]; cell_hi[ax] = box_hi[ax]; }
    for (int64_t i = 0; i < n_boxes; ++i) {
        all[(size_t)i].tri = (int32_t)i;
        for (int ax = 0; ax < 3; ++ax) {
            // (a coordinate that is not a number would break the order the sweeps below sort by)
            if (!std::isfinite(box_lo[i * 3 + ax]) || !std::isfinite(box_hi[i * 3 + ax])) return RTAMD_EINVAL;
            all[(size_t)i].lo[ax] = box_lo[i * 3 + ax];
            all[(size_t)i].hi[ax] = box_hi[i * 3 + ax];
            cell_lo[ax] = std::min(cell_lo[ax], box_lo[i * 3 + ax]);
            cell_hi[ax] = std::max(cell_hi[ax], box_hi[i * 3 + ax]);
        }
    }
    if (tri_corners)
        for (int64_t i = 0; i < n_boxes * 9; ++i)
            if (!std::isfinite(tri_corners[i])) return RTAMD_EINVAL;
    b.build(all, cell_lo, cell_hi, 0);
    std::vector<Ref>().swap(all);
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
    if (leaf_box) std::copy(b.leaf_box.begin(), b.leaf_box.end(), leaf_box);
    return RTAMD_OK;
    } catch (...) {             // the builder's lists are std::vectors: out of host memory
        return RTAMD_EINVAL;
    }
}
