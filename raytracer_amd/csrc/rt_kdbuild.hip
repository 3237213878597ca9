// Device-side build of the reference's kd-tree (/root/reference/kdtree.py:8-24 BoundingBoxes,
// :27-52 SAH, :92-131 KDTree.sub_divide), level by level, emitting exactly the tree the
// reference's recursion builds: same planes (bit for bit), same leaf lists in the same order.
//
// The reference recurses depth first over Python lists; nothing in a node depends on its
// siblings, so here all nodes of one depth are processed together:
//   * a node's list is a segment of an index array; `sorted` / `list.sort` are stable, and so
//     is the segmented merge sort below (it ranks every element inside the neighbouring run,
//     `<` for elements of the left run and `<=` for those of the right run);
//   * SAH sorts the node's list IN PLACE by upper bound (kdtree.py:28), so the list axis 1 sees
//     is the one axis 0 left behind, and the list a degenerate node keeps as its leaf is the one
//     the third sort left behind: the segment is threaded through the three axes the same way;
//   * both `while` scans of kdtree.py:38-42 count entries <= plane in a sorted list with a plane
//     that never decreases: a binary search for the upper bound;
//   * the cost (kdtree.py:43) is two products and a sum in double precision, the first strict
//     minimum wins, axes compete with a strict `<` after scaling by the cell's cross-section
//     (kdtree.py:104-108), and children re-sort their lists by box centre along x (kdtree.py:10).
// Nodes are numbered breadth first while the tree grows and renumbered to preorder (a node's
// left subtree follows it directly) at the end, which is the layout the traversal kernels read.
//
// Every buffer beyond inputs and outputs lives in one workspace that the caller passes in
// (rtamd_kd_build) or in a static device array (rtamd_kd_build_static).  The host side reads a
// few counters back per level, so the build is synchronous.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/raytracer_amd.h"

namespace {

constexpr double KD_EPS = 1.0e-7;                           // kdtree.py:5
constexpr double KD_INF = 9223372036854775807.0;            // sys.maxint as the comparisons see it
constexpr int LEAF_LIMIT = 20;                              // kdtree.py:94
constexpr int TPB = 256;

enum { H_NEXT_NODES = 0, H_NEXT_ITEMS = 1, H_LEAF_ITEMS = 2, H_MAX_CANDIDATE = 3, H_WORDS = 8 };

// One level of the tree under construction: node m owns items [begin[m], begin[m]+count[m]).
struct Level {
    int32_t* begin;
    int32_t* count;
    int32_t* forced;        // kdtree.py:124, :128: a child that received every box becomes a leaf
    int32_t* id;            // breadth-first node number
    double* lo;             // [3][capacity] cell bounds (min_d / max_d of kdtree.py:92)
    double* hi;
};

struct Build {
    const double* box_lo;   // [n][3]
    const double* box_hi;
    int64_t n_boxes, cap_items, cap_nodes, cap_refs;
    Level cur, next;
    int32_t *items, *items_next, *seg, *seg_next, *tmp_items;
    int32_t* items_u[3];    // the list after the in-place sort by upper bound along each axis
    int32_t* items_l[3];    // ... and sorted by lower bound (x0_sorted_array of kdtree.py:34)
    double *keys, *tmp_keys, *keys_next;
    double* up_keys[3];
    double* lo_keys[3];
    double* best_cost[3];   // per node and axis: the winning candidate of kdtree.py:35-48
    double* best_plane[3];
    int32_t* best_nl[3];
    int32_t* best_nu[3];
    // per node decisions
    int32_t *d_axis, *d_k, *d_j, *d_leaf_items, *d_child_items, *d_children;
    double* d_center;
    int32_t *o_leaf, *o_child_items, *o_children;      // exclusive scans of the three d_ arrays
    // breadth-first tree
    int32_t *b_axis, *b_left, *b_right, *b_first, *b_count, *b_size, *b_pre, *b_cnt_pre, *b_first_pre;
    double* b_split;
    int32_t* leaf_pool;
    int32_t* header;
};

__device__ __forceinline__ double canonical(double x) { return x + 0.0; }    // -0.0 sorts as +0.0, as Python compares them

// ------------------------------------------------------------------ stable segmented merge sort
// One pass: inside every segment, runs of `width` sorted elements are merged pairwise.
__global__ void merge_pass(const double* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                           double* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                           const int32_t* __restrict__ seg, const int32_t* __restrict__ begin,
                           const int32_t* __restrict__ count, int32_t total, int32_t width) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int32_t s = seg[i], b = begin[s], n = count[s];
    const int32_t r = i - b;
    const double key = keys_in[i];
    const int32_t start = (r / (2 * width)) * (2 * width);
    const int32_t mid = min(start + width, n), end = min(start + 2 * width, n);
    int32_t pos;
    if (r < mid) {
        // rank among the right run: elements strictly smaller go first
        int32_t lo = mid, hi = end;
        while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (keys_in[b + m] < key) lo = m + 1; else hi = m; }
        pos = r + (lo - mid);
    } else {
        // rank among the left run: elements smaller or equal go first
        int32_t lo = start, hi = mid;
        while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (keys_in[b + m] <= key) lo = m + 1; else hi = m; }
        pos = (r - mid) + lo;
    }
    keys_out[b + pos] = key;
    vals_out[b + pos] = vals_in[i];
}

__global__ void copy_pairs(const double* keys_in, const int32_t* vals_in, double* keys_out, int32_t* vals_out, int32_t total) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { keys_out[i] = keys_in[i]; vals_out[i] = vals_in[i]; }
}

// keys[i] = coordinate `axis` of box items[i] (which: 0 lower bound, 1 upper bound, 2 centre along x)
__global__ void gather_keys(const double* __restrict__ box_lo, const double* __restrict__ box_hi,
                            const int32_t* __restrict__ items, double* __restrict__ keys, int32_t total, int axis, int which) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t bx = items[i];
    double k;
    if (which == 0) k = box_lo[bx * 3 + axis];
    else if (which == 1) k = box_hi[bx * 3 + axis];
    else k = (box_lo[bx * 3] + box_hi[bx * 3]) / 2;          // BoundingBox.get_mid_x
    keys[i] = canonical(k);
}

// ------------------------------------------------------------------ SAH (kdtree.py:27-52), one workgroup per node
__global__ void sah_eval(const double* __restrict__ up_keys, const double* __restrict__ lo_keys, Level lv, int axis,
                         double* __restrict__ best_cost, double* __restrict__ best_plane,
                         int32_t* __restrict__ best_nl, int32_t* __restrict__ best_nu) {
    const int32_t m = blockIdx.x;
    const int32_t n = lv.count[m], b = lv.begin[m];
    __shared__ double s_cost[TPB];
    __shared__ int32_t s_idx[TPB];
    double my_cost = 0;
    int32_t my_idx = -1;
    if (!lv.forced[m] && n > LEAF_LIMIT) {
        const double min_d = lv.lo[axis * (int64_t)gridDim.x + m], max_d = lv.hi[axis * (int64_t)gridDim.x + m];
        for (int32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const double plane = up_keys[b + i] + KD_EPS;
            int32_t lo = 0, hi = n;
            while (lo < hi) { const int32_t c = (lo + hi) >> 1; if (up_keys[b + c] <= plane) lo = c + 1; else hi = c; }
            const int32_t nu = lo;
            lo = 0; hi = n;
            while (lo < hi) { const int32_t c = (lo + hi) >> 1; if (lo_keys[b + c] <= plane) lo = c + 1; else hi = c; }
            const int32_t nl = lo;
            const double cost = (double)(nl + 1) * (plane - min_d) + (double)(n - nu - 1) * (max_d - plane);
            if (cost < KD_INF && (my_idx < 0 || cost < my_cost)) { my_cost = cost; my_idx = i; }   // ascending i: first minimum kept
        }
    }
    s_cost[threadIdx.x] = my_cost;
    s_idx[threadIdx.x] = my_idx;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double oc = s_cost[threadIdx.x + off];
            const int32_t oi = s_idx[threadIdx.x + off];
            const int32_t mi = s_idx[threadIdx.x];
            if (oi >= 0 && (mi < 0 || oc < s_cost[threadIdx.x] || (oc == s_cost[threadIdx.x] && oi < mi))) {
                s_cost[threadIdx.x] = oc;
                s_idx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int32_t i = s_idx[0];
        const int32_t at = m;
        if (i < 0) {
            best_cost[at] = KD_INF; best_plane[at] = 0; best_nl[at] = -1; best_nu[at] = -1;
        } else {
            const double plane = up_keys[b + i] + KD_EPS;
            int32_t lo = 0, hi = n;
            while (lo < hi) { const int32_t c = (lo + hi) >> 1; if (up_keys[b + c] <= plane) lo = c + 1; else hi = c; }
            best_nu[at] = lo;
            lo = 0; hi = n;
            while (lo < hi) { const int32_t c = (lo + hi) >> 1; if (lo_keys[b + c] <= plane) lo = c + 1; else hi = c; }
            best_nl[at] = lo;
            best_cost[at] = s_cost[0];
            best_plane[at] = plane;
        }
    }
}

// KDTree.sub_divide's choice (kdtree.py:94-118) for every node of the level.
__global__ void decide(Build B, int32_t n_nodes) {
    const int32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_nodes) return;
    const Level& lv = B.cur;
    const int32_t n = lv.count[m];
    int32_t axis = -1, k = 0, j = 0;
    double center = 0;
    if (!lv.forced[m] && n > LEAF_LIMIT) {
        double best = KD_INF;
        for (int a = 0; a < 3; ++a) {
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
            const double dim_a = lv.hi[a1 * (int64_t)n_nodes + m] - lv.lo[a1 * (int64_t)n_nodes + m];
            const double dim_b = lv.hi[a2 * (int64_t)n_nodes + m] - lv.lo[a2 * (int64_t)n_nodes + m];
            const double cost = B.best_cost[a][m] * (dim_a * dim_b);
            if (cost < best) {
                best = cost;
                axis = a;
                const double min_d = lv.lo[a * (int64_t)n_nodes + m], max_d = lv.hi[a * (int64_t)n_nodes + m];
                if (B.best_nl[a][m] < 0) {
                    // no candidate beat the initial cost: both slices take everything (kdtree.py:52 with None)
                    k = n; j = 0; center = min_d;
                } else {
                    k = B.best_nl[a][m];
                    j = B.best_nu[a][m];
                    const double plane = B.best_plane[a][m];
                    const double c = (min_d > plane) ? min_d : plane;          // max(split_center, min_d)
                    center = (c < max_d) ? c : max_d;                          // min(max_d, ...)
                }
            }
        }
        // kdtree.py:114: no axis won, or one side is empty: a leaf that keeps the list as the third sort left it
        if (axis < 0 || k == 0 || n - j == 0) axis = -2;
    }
    const bool interior = axis >= 0;
    B.d_axis[m] = axis;     // -1: leaf with the list as it came; -2: leaf with the list as the third sort left it
    B.d_k[m] = k;
    B.d_j[m] = j;
    B.d_center[m] = center;
    B.d_leaf_items[m] = interior ? 0 : n;
    B.d_child_items[m] = interior ? k + (n - j) : 0;
    B.d_children[m] = interior ? 2 : 0;
}

// Exclusive scans of three int arrays by one workgroup; totals go to the header.
__global__ void scan3(const int32_t* a0, const int32_t* a1, const int32_t* a2, int32_t* o0, int32_t* o1, int32_t* o2,
                      int32_t n, int32_t* header, int h0, int h1, int h2) {
    __shared__ int32_t part[3][TPB];
    const int32_t* in[3] = {a0, a1, a2};
    int32_t* out[3] = {o0, o1, o2};
    const int hs[3] = {h0, h1, h2};
    const int32_t per = (n + TPB - 1) / TPB;
    const int32_t lo = min(n, (int32_t)threadIdx.x * per), hi = min(n, lo + per);
    for (int q = 0; q < 3; ++q) {
        if (!in[q]) continue;
        int32_t sum = 0;
        for (int32_t i = lo; i < hi; ++i) sum += in[q][i];
        part[q][threadIdx.x] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) {
            if (!in[q]) continue;
            int32_t run = 0;
            for (int t = 0; t < TPB; ++t) { const int32_t v = part[q][t]; part[q][t] = run; run += v; }
            if (hs[q] >= 0) header[hs[q]] = run;
        }
    }
    __syncthreads();
    for (int q = 0; q < 3; ++q) {
        if (!in[q]) continue;
        int32_t run = part[q][threadIdx.x];
        for (int32_t i = lo; i < hi; ++i) { const int32_t v = in[q][i]; out[q][i] = run; run += v; }
    }
}

// Writes the level's node records, parks leaf lists in the pool and hands the children their
// lists (kdtree.py:119-131).  One workgroup per node.
__global__ void emit_level(Build B, int32_t n_nodes, int32_t n_next_nodes, int32_t next_id_base, int32_t leaf_base) {
    const int32_t m = blockIdx.x;
    const Level& lv = B.cur;
    const int32_t n = lv.count[m], b = lv.begin[m], id = lv.id[m];
    const int32_t axis = B.d_axis[m];
    if (axis < 0) {
        const int32_t* src = axis == -1 ? B.items : B.items_u[2];
        const int32_t at = leaf_base + B.o_leaf[m];
        for (int32_t i = threadIdx.x; i < n; i += blockDim.x) B.leaf_pool[at + i] = src[b + i];
        if (threadIdx.x == 0) {
            B.b_axis[id] = -1; B.b_split[id] = 0; B.b_left[id] = -1; B.b_right[id] = -1;
            B.b_first[id] = at; B.b_count[id] = n;
        }
        return;
    }
    const int32_t k = B.d_k[m], j = B.d_j[m];
    const int32_t c0 = B.o_children[m], c1 = c0 + 1;                 // positions of the children in the next level
    const int32_t off0 = B.o_child_items[m], off1 = off0 + k;
    const int32_t* left = B.items_l[axis] + b;                       // x0_sorted_array[:best_left]
    const int32_t* right = B.items_u[axis] + b + j;                  // array[best_right:]
    for (int32_t i = threadIdx.x; i < k; i += blockDim.x) { B.items_next[off0 + i] = left[i]; B.seg_next[off0 + i] = c0; }
    for (int32_t i = threadIdx.x; i < n - j; i += blockDim.x) { B.items_next[off1 + i] = right[i]; B.seg_next[off1 + i] = c1; }
    if (threadIdx.x == 0) {
        const double center = B.d_center[m];
        B.b_axis[id] = axis; B.b_split[id] = center;
        B.b_left[id] = next_id_base + c0; B.b_right[id] = next_id_base + c1;
        B.b_first[id] = 0; B.b_count[id] = 0;
        Level& nx = B.next;
        nx.begin[c0] = off0; nx.count[c0] = k; nx.forced[c0] = k == n; nx.id[c0] = next_id_base + c0;
        nx.begin[c1] = off1; nx.count[c1] = n - j; nx.forced[c1] = (n - j) == n; nx.id[c1] = next_id_base + c1;
        for (int a = 0; a < 3; ++a) {
            const double lo = lv.lo[a * (int64_t)n_nodes + m], hi = lv.hi[a * (int64_t)n_nodes + m];
            nx.lo[a * (int64_t)n_next_nodes + c0] = lo;
            nx.hi[a * (int64_t)n_next_nodes + c0] = a == axis ? center : hi;      // max_d_left
            nx.lo[a * (int64_t)n_next_nodes + c1] = a == axis ? center : lo;      // min_d_right
            nx.hi[a * (int64_t)n_next_nodes + c1] = hi;
        }
        // every child list is sorted by centre along x, also the ones that become leaves at once
        atomicMax(&B.header[H_MAX_CANDIDATE], max(k, n - j));
    }
}

// ------------------------------------------------------------------ breadth first -> preorder
__global__ void subtree_sizes(Build B, int32_t first, int32_t last) {
    const int32_t id = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= last) return;
    B.b_size[id] = B.b_axis[id] < 0 ? 1 : 1 + B.b_size[B.b_left[id]] + B.b_size[B.b_right[id]];
}
__global__ void preorder_numbers(Build B, int32_t first, int32_t last) {
    const int32_t id = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= last) return;
    if (id == 0) B.b_pre[0] = 0;
    const int32_t pre = B.b_pre[id];
    B.b_cnt_pre[pre] = B.b_count[id];
    if (B.b_axis[id] >= 0) {
        B.b_pre[B.b_left[id]] = pre + 1;
        B.b_pre[B.b_right[id]] = pre + 1 + B.b_size[B.b_left[id]];
    }
}
__global__ void write_tree(Build B, int32_t n_nodes, rtamd_kd_tree_out out) {
    const int32_t id = blockIdx.x;
    const int32_t pre = B.b_pre[id];
    const int32_t axis = B.b_axis[id];
    if (threadIdx.x == 0) {
        out.axis[pre] = axis < 0 ? -1 : axis;
        out.split[pre] = B.b_split[id];
        out.left[pre] = axis < 0 ? -1 : B.b_pre[B.b_left[id]];
        out.right[pre] = axis < 0 ? -1 : B.b_pre[B.b_right[id]];
        out.first[pre] = axis < 0 ? B.b_first_pre[pre] : 0;
        out.count[pre] = axis < 0 ? B.b_count[id] : 0;
    }
    if (axis < 0) {
        const int32_t n = B.b_count[id], from = B.b_first[id], to = B.b_first_pre[pre];
        for (int32_t i = threadIdx.x; i < n; i += blockDim.x) out.leaf_refs[to + i] = B.leaf_pool[from + i];
    }
}
__global__ void iota_segment(int32_t* items, int32_t* seg, int32_t n) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { items[i] = i; seg[i] = 0; }
}
__global__ void root_level(Build B, int32_t n) {
    // BoundingBoxes.__init__ (kdtree.py:14-21): the extent of all boxes; one workgroup
    __shared__ double s_lo[3][TPB], s_hi[3][TPB];
    for (int a = 0; a < 3; ++a) {
        double lo = 0, hi = 0;
        bool any = false;
        for (int32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const double l = B.box_lo[(int64_t)i * 3 + a], h = B.box_hi[(int64_t)i * 3 + a];
            if (!any || l < lo) lo = l;
            if (!any || h > hi) hi = h;
            any = true;
        }
        s_lo[a][threadIdx.x] = any ? lo : 1e308 * 10;
        s_hi[a][threadIdx.x] = any ? hi : -1e308 * 10;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            double lo = s_lo[a][0], hi = s_hi[a][0];
            for (int t = 1; t < TPB; ++t) { if (s_lo[a][t] < lo) lo = s_lo[a][t]; if (s_hi[a][t] > hi) hi = s_hi[a][t]; }
            B.cur.lo[a] = lo;           // one node: stride 1
            B.cur.hi[a] = hi;
        }
        B.cur.begin[0] = 0; B.cur.count[0] = n; B.cur.forced[0] = 0; B.cur.id[0] = 0;
    }
}

// ------------------------------------------------------------------ host side
thread_local char g_kd_error[256] = "";
int kd_fail(int code, const char* msg) { snprintf(g_kd_error, sizeof(g_kd_error), "%s", msg); return code; }
int kd_hip(hipError_t e, const char* what) { snprintf(g_kd_error, sizeof(g_kd_error), "%s: %s", what, hipGetErrorString(e)); return RTAMD_EHIP; }

struct Carver {
    char* base;
    size_t used, size;
    template <class T> T* take(int64_t n) {
        used = (used + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base ? base + used : nullptr);
        used += sizeof(T) * (size_t)n;
        return p;
    }
};

void carve(Build& B, Carver& c) {
    const int64_t L = B.cap_items, N = B.cap_nodes, R = B.cap_refs;
    for (Level* lv : {&B.cur, &B.next}) {
        lv->begin = c.take<int32_t>(L); lv->count = c.take<int32_t>(L); lv->forced = c.take<int32_t>(L);
        lv->id = c.take<int32_t>(L); lv->lo = c.take<double>(3 * L); lv->hi = c.take<double>(3 * L);
    }
    B.items = c.take<int32_t>(L); B.items_next = c.take<int32_t>(L); B.seg = c.take<int32_t>(L);
    B.seg_next = c.take<int32_t>(L); B.tmp_items = c.take<int32_t>(L);
    B.keys = c.take<double>(L); B.tmp_keys = c.take<double>(L); B.keys_next = c.take<double>(L);
    for (int a = 0; a < 3; ++a) {
        B.items_u[a] = c.take<int32_t>(L); B.items_l[a] = c.take<int32_t>(L);
        B.up_keys[a] = c.take<double>(L); B.lo_keys[a] = c.take<double>(L);
        B.best_cost[a] = c.take<double>(L); B.best_plane[a] = c.take<double>(L);
        B.best_nl[a] = c.take<int32_t>(L); B.best_nu[a] = c.take<int32_t>(L);
    }
    B.d_axis = c.take<int32_t>(L); B.d_k = c.take<int32_t>(L); B.d_j = c.take<int32_t>(L);
    B.d_leaf_items = c.take<int32_t>(L); B.d_child_items = c.take<int32_t>(L); B.d_children = c.take<int32_t>(L);
    B.d_center = c.take<double>(L);
    B.o_leaf = c.take<int32_t>(L); B.o_child_items = c.take<int32_t>(L); B.o_children = c.take<int32_t>(L);
    B.b_axis = c.take<int32_t>(N); B.b_left = c.take<int32_t>(N); B.b_right = c.take<int32_t>(N);
    B.b_first = c.take<int32_t>(N); B.b_count = c.take<int32_t>(N); B.b_size = c.take<int32_t>(N);
    B.b_pre = c.take<int32_t>(N); B.b_cnt_pre = c.take<int32_t>(N); B.b_first_pre = c.take<int32_t>(N);
    B.b_split = c.take<double>(N);
    B.leaf_pool = c.take<int32_t>(R);
    B.header = c.take<int32_t>(H_WORDS);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + TPB - 1) / TPB); }

// Stable sort of (keys, vals) inside the segments of `lv`; the result lands in (dst_keys, dst_vals).
void segmented_sort(hipStream_t st, const double* src_keys, const int32_t* src_vals, double* tmp_keys, int32_t* tmp_vals,
                    double* dst_keys, int32_t* dst_vals, const int32_t* seg, const Level& lv, int32_t total, int32_t longest) {
    int passes = 0;
    while ((1 << passes) < longest) ++passes;
    if (passes == 0) {
        hipLaunchKernelGGL(copy_pairs, dim3(blocks_for(total)), dim3(TPB), 0, st, src_keys, src_vals, dst_keys, dst_vals, total);
        return;
    }
    const double* in_k = src_keys;
    const int32_t* in_v = src_vals;
    for (int p = 0; p < passes; ++p) {
        const bool to_dst = ((passes - 1 - p) & 1) == 0;
        double* out_k = to_dst ? dst_keys : tmp_keys;
        int32_t* out_v = to_dst ? dst_vals : tmp_vals;
        hipLaunchKernelGGL(merge_pass, dim3(blocks_for(total)), dim3(TPB), 0, st, in_k, in_v, out_k, out_v, seg, lv.begin,
                           lv.count, total, 1 << p);
        in_k = out_k;
        in_v = out_v;
    }
}

int build_tree(Build& B, const rtamd_kd_tree_out* out, int64_t* n_nodes_out, int64_t* n_refs_out, int32_t* depth_out,
               hipStream_t st) {
    const int32_t n = (int32_t)B.n_boxes;
    int32_t header[H_WORDS];
    hipError_t e;
    // level 0: all boxes, ordered by centre along x
    hipLaunchKernelGGL(iota_segment, dim3(blocks_for(n)), dim3(TPB), 0, st, B.items_next, B.seg, n);
    hipLaunchKernelGGL(root_level, dim3(1), dim3(TPB), 0, st, B, n);
    hipLaunchKernelGGL(gather_keys, dim3(blocks_for(n)), dim3(TPB), 0, st, B.box_lo, B.box_hi, B.items_next, B.keys_next, n, 0, 2);
    segmented_sort(st, B.keys_next, B.items_next, B.tmp_keys, B.tmp_items, B.keys, B.items, B.seg, B.cur, n, n);

    int32_t level_nodes = 1, level_items = n, longest = n, nodes_so_far = 1, leaf_items = 0;
    int32_t level_first[256];
    int n_levels = 0;
    while (level_nodes > 0) {
        if (n_levels >= 255) return kd_fail(RTAMD_EUNSUPPORTED, "kd-tree deeper than 255 levels");
        level_first[n_levels++] = nodes_so_far - level_nodes;
        const dim3 item_grid(blocks_for(level_items)), node_grid(level_nodes);
        for (int a = 0; a < 3; ++a) {
            const int32_t* in_items = a == 0 ? B.items : B.items_u[a - 1];
            hipLaunchKernelGGL(gather_keys, item_grid, dim3(TPB), 0, st, B.box_lo, B.box_hi, in_items, B.keys_next, level_items, a, 1);
            segmented_sort(st, B.keys_next, in_items, B.tmp_keys, B.tmp_items, B.up_keys[a], B.items_u[a], B.seg, B.cur,
                           level_items, longest);
            hipLaunchKernelGGL(gather_keys, item_grid, dim3(TPB), 0, st, B.box_lo, B.box_hi, B.items_u[a], B.keys_next, level_items, a, 0);
            segmented_sort(st, B.keys_next, B.items_u[a], B.tmp_keys, B.tmp_items, B.lo_keys[a], B.items_l[a], B.seg, B.cur,
                           level_items, longest);
            hipLaunchKernelGGL(sah_eval, node_grid, dim3(TPB), 0, st, B.up_keys[a], B.lo_keys[a], B.cur, a, B.best_cost[a],
                               B.best_plane[a], B.best_nl[a], B.best_nu[a]);
        }
        e = hipMemsetAsync(B.header, 0, sizeof(int32_t) * H_WORDS, st);
        if (e != hipSuccess) return kd_hip(e, "clearing the level header");
        hipLaunchKernelGGL(decide, dim3(blocks_for(level_nodes)), dim3(TPB), 0, st, B, level_nodes);
        hipLaunchKernelGGL(scan3, dim3(1), dim3(TPB), 0, st, B.d_leaf_items, B.d_child_items, B.d_children, B.o_leaf,
                           B.o_child_items, B.o_children, level_nodes, B.header, (int)H_LEAF_ITEMS, (int)H_NEXT_ITEMS, (int)H_NEXT_NODES);
        e = hipMemcpyAsync(header, B.header, sizeof(header), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return kd_hip(e, "reading the level totals");
        const int32_t next_nodes = header[H_NEXT_NODES], next_items = header[H_NEXT_ITEMS];
        if (next_items > B.cap_items || next_nodes > B.cap_items)
            return kd_fail(RTAMD_EUNSUPPORTED, "kd build: a level holds more references than the workspace was sized for");
        if ((int64_t)nodes_so_far + next_nodes > B.cap_nodes || (int64_t)nodes_so_far + next_nodes > out->max_nodes)
            return kd_fail(RTAMD_EUNSUPPORTED, "kd build: more nodes than the output arrays hold");
        if ((int64_t)leaf_items + header[H_LEAF_ITEMS] > B.cap_refs || (int64_t)leaf_items + header[H_LEAF_ITEMS] > out->max_refs)
            return kd_fail(RTAMD_EUNSUPPORTED, "kd build: more leaf references than the output arrays hold");
        hipLaunchKernelGGL(emit_level, node_grid, dim3(TPB), 0, st, B, level_nodes, next_nodes, nodes_so_far, leaf_items);
        leaf_items += header[H_LEAF_ITEMS];
        if (next_nodes > 0) {
            e = hipMemcpyAsync(header, B.header, sizeof(header), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return kd_hip(e, "reading the longest child list");
            longest = header[H_MAX_CANDIDATE];
            // children re-sort their lists by box centre along x (kdtree.py:10 through :125-131)
            hipLaunchKernelGGL(gather_keys, dim3(blocks_for(next_items)), dim3(TPB), 0, st, B.box_lo, B.box_hi, B.items_next,
                               B.keys_next, next_items, 0, 2);
            segmented_sort(st, B.keys_next, B.items_next, B.tmp_keys, B.tmp_items, B.keys, B.items, B.seg_next, B.next,
                           next_items, longest);
            Level t = B.cur; B.cur = B.next; B.next = t;
            int32_t* s = B.seg; B.seg = B.seg_next; B.seg_next = s;
        }
        nodes_so_far += next_nodes;
        level_nodes = next_nodes;
        level_items = next_items;
    }
    // breadth first -> preorder
    const int32_t total_nodes = nodes_so_far;
    level_first[n_levels] = total_nodes;
    for (int l = n_levels - 1; l >= 0; --l) {
        const int32_t cnt = level_first[l + 1] - level_first[l];
        hipLaunchKernelGGL(subtree_sizes, dim3(blocks_for(cnt)), dim3(TPB), 0, st, B, level_first[l], level_first[l + 1]);
    }
    for (int l = 0; l < n_levels; ++l) {
        const int32_t cnt = level_first[l + 1] - level_first[l];
        hipLaunchKernelGGL(preorder_numbers, dim3(blocks_for(cnt)), dim3(TPB), 0, st, B, level_first[l], level_first[l + 1]);
    }
    hipLaunchKernelGGL(scan3, dim3(1), dim3(TPB), 0, st, B.b_cnt_pre, (const int32_t*)nullptr, (const int32_t*)nullptr,
                       B.b_first_pre, (int32_t*)nullptr, (int32_t*)nullptr, total_nodes, B.header, -1, -1, -1);
    hipLaunchKernelGGL(write_tree, dim3(total_nodes), dim3(64), 0, st, B, total_nodes, *out);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return kd_hip(e, "kd build");
    e = hipGetLastError();
    if (e != hipSuccess) return kd_hip(e, "kd build kernels");
    *n_nodes_out = total_nodes;
    *n_refs_out = leaf_items;
    *depth_out = n_levels - 1;
    return RTAMD_OK;
}

int check_args(const double* lo, const double* hi, int64_t n, const rtamd_kd_tree_out* out, int64_t* a, int64_t* b, int32_t* c) {
    if (!lo || !hi || !out || !a || !b || !c) return kd_fail(RTAMD_EINVAL, "kd build: null argument");
    if (n < 1 || n > 0x3fffffff) return kd_fail(RTAMD_EINVAL, "kd build: box count out of range");
    if (!out->axis || !out->split || !out->left || !out->right || !out->first || !out->count || !out->leaf_refs)
        return kd_fail(RTAMD_EINVAL, "kd build: null output array");
    if (out->max_nodes < 1 || out->max_refs < n) return kd_fail(RTAMD_EINVAL, "kd build: output arrays too small");
    return RTAMD_OK;
}

// ------------------------------------------------------------------ the workspace that needs no allocation
// rtamd_kd_build_static keeps the build's buffers in this array of the code object instead of a
// workspace from the caller.  Every GPU the library runs on holds its own copy at its own
// address; the array's name is only a host-side handle, so the address is asked for once per
// device (hipGetSymbolAddress with that device current) and used from then on.
constexpr int64_t STATIC_MAX_BOXES = 8192;
constexpr int64_t STATIC_LEVEL_CAPACITY = 3 * STATIC_MAX_BOXES;     // references one tree level may hold
constexpr int64_t STATIC_MAX_NODES = 32768;
constexpr int64_t STATIC_MAX_REFS = 32768;
constexpr size_t STATIC_BYTES = 11 * 1024 * 1024;
static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];
bool g_scratch_known[MAX_DEVICES];
std::mutex g_scratch_lock;

int static_scratch(void** ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return kd_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= MAX_DEVICES) return kd_fail(RTAMD_EUNSUPPORTED, "kd build: device index beyond the table of static workspaces");
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (!g_scratch_known[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return kd_hip(e, "address of the static kd workspace");
        g_scratch_ptr[dev] = p;
        g_scratch_known[dev] = true;
    }
    *ptr = g_scratch_ptr[dev];
    return RTAMD_OK;
}

}  // namespace

extern "C" {

const char* rtamd_kd_last_error(void) { return g_kd_error; }

int rtamd_kd_build_static(const double* box_lo, const double* box_hi, int64_t n_boxes, const rtamd_kd_tree_out* out,
                          int64_t* n_nodes, int64_t* n_refs, int32_t* depth, void* stream) {
    int rc = check_args(box_lo, box_hi, n_boxes, out, n_nodes, n_refs, depth);
    if (rc) return rc;
    if (n_boxes > STATIC_MAX_BOXES)
        return kd_fail(RTAMD_EUNSUPPORTED, "rtamd_kd_build_static handles at most 8192 boxes; larger lists need rtamd_kd_build with a workspace");
    void* ws = nullptr;
    rc = static_scratch(&ws);
    if (rc) return rc;
    Build B;
    memset(&B, 0, sizeof(B));
    B.box_lo = box_lo; B.box_hi = box_hi; B.n_boxes = n_boxes;
    B.cap_items = STATIC_LEVEL_CAPACITY; B.cap_nodes = STATIC_MAX_NODES; B.cap_refs = STATIC_MAX_REFS;
    Carver c = {static_cast<char*>(ws), 0, STATIC_BYTES};
    carve(B, c);
    if (c.used > STATIC_BYTES) return kd_fail(RTAMD_EUNSUPPORTED, "kd build: static workspace smaller than its layout");
    return build_tree(B, out, n_nodes, n_refs, depth, static_cast<hipStream_t>(stream));
}

int rtamd_kd_build_workspace_bytes(int64_t level_capacity, int64_t max_nodes, int64_t max_refs, size_t* bytes) {
    if (!bytes || level_capacity < 1 || max_nodes < 1 || max_refs < 1) return kd_fail(RTAMD_EINVAL, "kd workspace: bad argument");
    Build B;
    memset(&B, 0, sizeof(B));
    B.cap_items = level_capacity; B.cap_nodes = max_nodes; B.cap_refs = max_refs;
    Carver c = {nullptr, 0, 0};
    carve(B, c);
    *bytes = (c.used + 255) & ~size_t(255);
    return RTAMD_OK;
}

int rtamd_kd_build(const double* box_lo, const double* box_hi, int64_t n_boxes, const rtamd_kd_tree_out* out,
                   void* workspace, size_t workspace_bytes, int64_t level_capacity, int64_t* n_nodes, int64_t* n_refs,
                   int32_t* depth, void* stream) {
    int rc = check_args(box_lo, box_hi, n_boxes, out, n_nodes, n_refs, depth);
    if (rc) return rc;
    if (!workspace) return kd_fail(RTAMD_EINVAL, "kd build: null workspace");
    if (level_capacity < n_boxes) return kd_fail(RTAMD_EINVAL, "kd build: level capacity below the box count");
    size_t need = 0;
    rtamd_kd_build_workspace_bytes(level_capacity, out->max_nodes, out->max_refs, &need);
    if (workspace_bytes < need) return kd_fail(RTAMD_EINVAL, "kd build: workspace smaller than rtamd_kd_build_workspace_bytes says");
    Build B;
    memset(&B, 0, sizeof(B));
    B.box_lo = box_lo; B.box_hi = box_hi; B.n_boxes = n_boxes;
    B.cap_items = level_capacity; B.cap_nodes = out->max_nodes; B.cap_refs = out->max_refs;
    Carver c = {static_cast<char*>(workspace), 0, workspace_bytes};
    carve(B, c);
    return build_tree(B, out, n_nodes, n_refs, depth, static_cast<hipStream_t>(stream));
}

}  // extern "C"
