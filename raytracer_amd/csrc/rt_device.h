// Per-sample ray tracing path for gfx950, IEEE double throughout.
//
// Every function states the reference code it reproduces (paths under
// /root/reference).  Comparisons are written in the reference's own form and order
// (`not t > eps`, strict `<`, Python's min/max argument rules) because NaN and
// infinities from zero direction components flow through them.
//
// Compile-time switches of this file and rt_kernels.hip (all off / at their defaults in the product library):
//
//   builds that check or measure (tools/, tests/fuzz_sweep.py; never shipped)
//     RT_PHASE_TIMERS       shader-clock timers and walk statistics per part of a kd query (tools/phase_profile.py)
//     RT_REF_ORDER          a path's radiance summed in the reference's order (PathSum below)
//     RT_TRI_LU             triangle systems solved by LU with pivoting, as numpy.linalg.solve does (tri_solve_v)
//     RT_POW_ALT            another evaluation of pow (rt_pow)
//     RT_ONLY_HOT / RT_ONLY_TILES   libraries that hold only the benchmark's kernel / only the whole-frame kernels, for
//                           quick experiment builds (tools/build_variant.sh); RT_NO_REFILL: no refilled waves
//   tunables, each with the measurements that set its default next to its definition
//     RT_STACK_MAX (40), RT_LDS_STACK (10), RT_TINY_DEPTH (8), RT_MAX_SKIPS (1), RT_MAX_LOADS (3),
//     RT_ST_MAX_SKIPS (1), RT_ST_MAX_LOADS (3), RT_STEAL_PASSES (1), RT_STEAL_MIN_PAIRS (1) here;
//     RT_BLOCK (64), RT_MIN_WAVES (4), RT_MIN_WAVES_ANALYTIC (4), RT_REFILL_MAX (16), RT_GROUP_SHIFT_MAX (4) in rt_kernels.hip
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/raytracer_amd.h"

namespace rt {

constexpr double EPS = 1.0e-7;                    // geometry.py:8, tracer.py:4, kdtree.py:5
constexpr double T_INF = 9223372036854775807.0;   // sys.maxint, the initial tmin (main.py:37)
constexpr double PI = 3.141592653589793;
#ifndef RT_STACK_MAX
#define RT_STACK_MAX 40
#endif
constexpr int STACK_MAX = RT_STACK_MAX;           // >= tree depth + 2, checked on the host
constexpr int NO_SHADOW_BIT = 1 << 30;

// ---------------------------------------------------------------- small vector type
struct V3 { double x, y, z; };

__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ double axis_of(V3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }
__device__ __forceinline__ float axis_of_f(float x, float y, float z, int ax) { return ax == 0 ? x : (ax == 1 ? y : z); }

// Python's min/max keep the first argument unless a later one compares strictly
// beyond it; with NaN operands that is not fmin/fmax.
__device__ __forceinline__ double pymin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }

struct Ray { V3 o, d; };

// ---------------------------------------------------------------- random draws
// Same function as raytracer_amd/rng.py: draw = mix(mix(mix(seed ^ phase*C) ^ a) ^ (b<<32 | counter)).
constexpr uint64_t PHASE_MUL = 0xD6E8FEB86659FD93ull;
constexpr uint64_t PHASE_PIXEL = 2, PHASE_TRACE = 4;

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t phase_key(uint64_t seed, uint64_t phase) {
    return mix64(seed ^ (phase * PHASE_MUL));
}
__device__ __forceinline__ uint32_t index_from(uint64_t h, uint32_t n) {
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

struct Draw {
    uint64_t key;       // mix(phase_key ^ pixel index)
    uint32_t sample;
    uint32_t counter;
    __device__ __forceinline__ uint32_t index(uint32_t n) {
        uint64_t h = mix64(key ^ (((uint64_t)sample << 32) | (uint64_t)counter));
        counter++;
        return index_from(h, n);
    }
};

// sampler.py:84-85 (one uniform choice over the flat hemisphere list); a RegularSampler
// returns its single vector without drawing (sampler.py:13-14).
__device__ __forceinline__ V3 draw_hemisphere(const rtamd_scene_desc& S, int sid, Draw& dr) {
    const rtamd_sampler sm = S.samplers[sid];
    uint32_t i = sm.regular ? 0u : dr.index((uint32_t)(sm.num_patterns * sm.samples_per_pattern));
    return ld3(S.sampler_data + sm.hemi_offset + 3 * (int64_t)i);
}

// geometry.py:379: random.choice(sampler.sample()) - pattern choice, then element choice.
__device__ __forceinline__ void draw_square(const rtamd_scene_desc& S, int sid, Draw& dr, double& sx, double& sy) {
    const rtamd_sampler sm = S.samplers[sid];
    uint32_t pat = sm.regular ? 0u : dr.index((uint32_t)sm.num_patterns);
    uint32_t el = dr.index((uint32_t)sm.samples_per_pattern);
    const double* p = S.sampler_data + sm.square_offset + 2 * ((int64_t)pat * sm.samples_per_pattern + el);
    sx = p[0];
    sy = p[1];
}

// ---------------------------------------------------------------- work counters
#ifdef RT_PHASE_TIMERS
__device__ unsigned long long g_phase_cycles[64];
__device__ unsigned long long g_phase_hist[64];   // shadow-tree queries by their number of rays (8 per bucket): queries, rounds, cycles, lane-rounds
// coarse marks along a wave's life (profiling builds): cycles since the previous mark go to `slot`
struct WaveMarks {
    long long mark_t = 0;
    __device__ __forceinline__ void mark(int slot) {
        const long long now = clock64();
        if (__lane_id() == 0) atomicAdd(&g_phase_cycles[slot], (unsigned long long)(now - mark_t));
        mark_t = clock64();
    }
};
#else
struct WaveMarks { __device__ __forceinline__ void mark(int) {} };
#endif
// COUNT = 0: the timed kernels count rays only.  COUNT = 1: the work of the REFERENCE algorithm
// (kd steps, leaves, triangle and analytic tests as kdtree.py / geometry.py would spend them);
// such a launch also takes no shortcut that changes those numbers.  COUNT = 2: the work this
// kernel EXECUTES on its fast path (the same code as COUNT = 0 plus the tallies): node records,
// culling boxes and triangle rows loaded, ray-triangle tests run, bytes those loads ask for.
template <int COUNT> struct Work : WaveMarks {
    uint32_t closest = 0, shadow = 0;
    __device__ __forceinline__ void node() {}
    __device__ __forceinline__ void leaf() {}
    __device__ __forceinline__ void tri(uint32_t) {}
    __device__ __forceinline__ void analytic() {}
    __device__ __forceinline__ void x_record() {}
    __device__ __forceinline__ void x_box(uint32_t) {}
    __device__ __forceinline__ void x_rows(uint32_t, uint32_t) {}
    __device__ __forceinline__ void x_tests(uint32_t) {}
    __device__ __forceinline__ void x_skip() {}
    __device__ __forceinline__ void x_bytes(uint32_t) {}
    __device__ __forceinline__ uint32_t slot(int) const { return 0; }
};
template <> struct Work<1> : WaveMarks {
    uint32_t closest = 0, shadow = 0, n_node = 0, n_leaf = 0, n_tri = 0, n_analytic = 0;
    __device__ __forceinline__ void node() { n_node++; }
    __device__ __forceinline__ void leaf() { n_leaf++; }
    __device__ __forceinline__ void tri(uint32_t k) { n_tri += k; }
    __device__ __forceinline__ void analytic() { n_analytic++; }
    __device__ __forceinline__ void x_record() {}
    __device__ __forceinline__ void x_box(uint32_t) {}
    __device__ __forceinline__ void x_rows(uint32_t, uint32_t) {}
    __device__ __forceinline__ void x_tests(uint32_t) {}
    __device__ __forceinline__ void x_skip() {}
    __device__ __forceinline__ void x_bytes(uint32_t) {}
    __device__ __forceinline__ uint32_t slot(int i) const {
        return i == 3 ? n_node : i == 4 ? n_leaf : i == 5 ? n_tri : i == 6 ? n_analytic : 0;
    }
};
template <> struct Work<2> : WaveMarks {
    uint32_t closest = 0, shadow = 0, n_node = 0, n_leaf = 0, n_test = 0, n_analytic = 0;
    uint32_t n_skip = 0, n_record = 0, n_box = 0, n_row = 0, n_byte = 0;
    __device__ __forceinline__ void node() { n_node++; }                     // interior steps taken
    __device__ __forceinline__ void leaf() { n_leaf++; }                     // leaves reached
    __device__ __forceinline__ void tri(uint32_t) {}
    __device__ __forceinline__ void analytic() { n_analytic++; n_byte += 48; }
    __device__ __forceinline__ void x_record() { n_record++; n_byte += 48; } // node + both children, 3 x 16 B
    __device__ __forceinline__ void x_box(uint32_t bytes) { n_box++; n_byte += bytes; }
    __device__ __forceinline__ void x_rows(uint32_t k, uint32_t bytes_each) { n_row += k; n_byte += k * bytes_each; }
    __device__ __forceinline__ void x_tests(uint32_t k) { n_test += k; }
    __device__ __forceinline__ void x_skip() { n_skip++; }
    __device__ __forceinline__ void x_bytes(uint32_t b) { n_byte += b; }
    __device__ __forceinline__ uint32_t slot(int i) const {
        switch (i) {
        case 3: return n_node; case 4: return n_leaf; case 5: return n_test; case 6: return n_analytic;
        case 7: return n_skip; case 8: return n_record; case 9: return n_box; case 10: return n_row;
        case 11: return n_byte; default: return 0;
        }
    }
};

// ---------------------------------------------------------------- wave helpers
// A wave traces 64 samples in lock step.  Per-lane work (kd descent, analytic
// primitives, shading arithmetic) runs under the usual exec mask; the triangle lists of
// big kd leaves are tested by the whole wave for one ray at a time, so the functions
// below are entered by all 64 lanes together and take a per-lane `active` flag instead
// of being called from divergent code.
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double bcast_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int first_lane(unsigned long long mask) {
    return __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mask));
}
// ---------------------------------------------------------------- triangles
// geometry.py:447-455 solves [v1-v0, v2-v0, -d] (alpha, beta, t)^T = o - v0 with LAPACK.
// Here the same system is solved by Cramer's rule in the Moeller-Trumbore arrangement;
// a singular system (the reference raises) is a miss because every comparison with
// the resulting inf/NaN is false.
#ifdef RT_TRI_LU
// Checking builds only: the system solved the way numpy.linalg.solve does at geometry.py:440
// (LAPACK dgesv: LU with partial pivoting, the pivot's reciprocal, a rank-1 update, the two
// triangular solves), operation for operation like the CPU checker.  Cramer's rule (below) agrees
// with it to rounding, which decides a test differently only where a ray grazes an edge to within
// that rounding; this build tells such knife-edge differences from real ones.
__device__ __forceinline__ bool tri_solve_v(V3 v0, V3 e1, V3 e2, const Ray& r,
                                            double& alpha, double& beta, double& t) {
    const V3 f = v0 - r.o;
    double a0[4] = {e1.x, e2.x, -r.d.x, -f.x}, a1[4] = {e1.y, e2.y, -r.d.y, -f.y}, a2[4] = {e1.z, e2.z, -r.d.z, -f.z};
    auto swap_rows = [](double* x, double* y) {
        for (int k = 0; k < 4; ++k) { const double tmp = x[k]; x[k] = y[k]; y[k] = tmp; }
    };
    {
        const double m0 = fabs(a0[0]), m1 = fabs(a1[0]), m2 = fabs(a2[0]);
        if (m1 > m0 && !(m2 > m1)) swap_rows(a0, a1);
        else if (m2 > m0 && m2 > m1) swap_rows(a0, a2);
    }
    if (a0[0] == 0.0) return false;
    {
        const double rcp = 1.0 / a0[0];
        a1[0] *= rcp; a2[0] *= rcp;
        a1[1] -= a1[0] * a0[1]; a1[2] -= a1[0] * a0[2];
        a2[1] -= a2[0] * a0[1]; a2[2] -= a2[0] * a0[2];
    }
    if (fabs(a2[1]) > fabs(a1[1])) swap_rows(a1, a2);
    if (a1[1] == 0.0) return false;
    {
        const double rcp = 1.0 / a1[1];
        a2[1] *= rcp;
        a2[2] -= a2[1] * a1[2];
    }
    if (a2[2] == 0.0) return false;
    double b0 = a0[3], b1 = a1[3], b2 = a2[3];
    b1 -= b0 * a1[0];
    b2 -= b0 * a2[0];
    b2 -= b1 * a2[1];
    b2 = b2 / a2[2];
    b0 -= b2 * a0[2];
    b1 -= b2 * a1[2];
    b1 = b1 / a1[1];
    b0 -= b1 * a0[1];
    b0 = b0 / a0[0];
    alpha = b0; beta = b1; t = b2;
    return alpha > EPS && beta > EPS && t > EPS && alpha + beta < 1.0;
}
#else
__device__ __forceinline__ bool tri_solve_v(V3 v0, V3 e1, V3 e2, const Ray& r,
                                            double& alpha, double& beta, double& t) {
    const V3 pvec = cross(r.d, e2);
    const double det = dot(e1, pvec);
    // reciprocal by two Newton steps on the hardware estimate (within an ulp or so of
    // 1/det; a zero or denormal det gives inf/NaN and with it a miss, as the division does)
    double inv = __builtin_amdgcn_rcp(det);
    inv = fma(inv, fma(-det, inv, 1.0), inv);
    inv = fma(inv, fma(-det, inv, 1.0), inv);
    const V3 tvec = r.o - v0;
    alpha = dot(tvec, pvec) * inv;
    const V3 qvec = cross(tvec, e1);
    beta = dot(r.d, qvec) * inv;
    t = dot(e2, qvec) * inv;
    return alpha > EPS && beta > EPS && t > EPS && alpha + beta < 1.0;
}
#endif
// bare triangle given as 9 contiguous doubles v0, e1, e2
__device__ __forceinline__ bool tri_solve(const double* g, const Ray& r, double& alpha, double& beta, double& t) {
    return tri_solve_v(ld3(g), ld3(g + 3), ld3(g + 6), r, alpha, beta, t);
}
// The scene's struct-of-arrays triangle table seen through a buffer descriptor: component c
// of row `ref` is tri_geom[c*n_refs + ref], so 64 consecutive rows are 9 coalesced loads.
// With buffer loads the per-lane part of the address is one 32-bit byte offset (ref*8) and
// the component's plane offset (c*n_refs*8) rides in a scalar register, which removes the
// 64-bit address arithmetic of nine separate global loads; rows past the table read as 0.
struct TriTable {
    __amdgpu_buffer_rsrc_t rsrc;
    int32_t plane;              // bytes between two component planes = n_refs * 8
};
__device__ __forceinline__ TriTable tri_table(const rtamd_scene_desc& S) {
    TriTable t;
    t.plane = (int32_t)(S.n_refs * 8);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(S.tri_geom), 0, (int32_t)(S.n_refs * 72), 0x00020000);
    return t;
}
__device__ __forceinline__ double tri_component(const TriTable& tt, int32_t byte_offset, int c) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(tt.rsrc, byte_offset, c * tt.plane, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ bool tri_solve_ref(const TriTable& tt, int32_t ref, const Ray& r,
                                              double& alpha, double& beta, double& t) {
    const int32_t off = ref * 8;
    const V3 v0 = v3(tri_component(tt, off, 0), tri_component(tt, off, 1), tri_component(tt, off, 2));
    const V3 e1 = v3(tri_component(tt, off, 3), tri_component(tt, off, 4), tri_component(tt, off, 5));
    const V3 e2 = v3(tri_component(tt, off, 6), tri_component(tt, off, 7), tri_component(tt, off, 8));
    return tri_solve_v(v0, e1, e2, r, alpha, beta, t);
}

// Row-major copy of the triangle table for scans by a single lane: one row = 80 contiguous
// bytes (v0, e1, e2 and {tri_prim, tri_order}), five 16-byte loads.
struct TriRow { V3 v0, e1, e2; int32_t prim, order; };
__device__ __forceinline__ TriRow load_row(const rtamd_scene_desc& S, int32_t ref) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = reinterpret_cast<const d2*>(S.tri_rows + (int64_t)ref * 10);
    const d2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriRow r;
    r.v0 = v3(a.x, a.y, b.x);
    r.e1 = v3(b.y, c.x, c.y);
    r.e2 = v3(d.x, d.y, e.x);
    r.prim = __double2loint(e.y);
    r.order = __double2hiint(e.y);
    return r;
}

// ---------------------------------------------------------------- phase timers (profiling builds)
// With RT_PHASE_TIMERS every wave adds the shader-clock cycles it spends in each part of a
// kd query to a global table (closest-hit queries in slots 0-7, shadow queries in 8-15,
// whole waves in slot 16); tools/phase_profile.py builds such a library and prints the split.
#ifdef RT_PHASE_TIMERS
struct PhaseClock {
    long long last;
    int base;
    __device__ __forceinline__ void start(int b) { base = b; last = clock64(); }
    __device__ __forceinline__ void lap(int slot) {
        const long long now = clock64();
        if (lane_id() == 0) atomicAdd(&g_phase_cycles[base + slot], (unsigned long long)(now - last));
        last = clock64();
    }
    __device__ __forceinline__ void count(int slot) {
        if (lane_id() == 0) atomicAdd(&g_phase_cycles[base + slot], 1ull);
    }
};
#else
struct PhaseClock {
    __device__ __forceinline__ void start(int) {}
    __device__ __forceinline__ void lap(int) {}
    __device__ __forceinline__ void count(int) {}
};
#endif
enum { PH_DESCENT = 0, PH_PACK16 = 1, PH_PACK32 = 2, PH_BIG_PLAIN = 3, PH_BIG_BOXES = 4, PH_POP = 5, PH_QUERIES = 6, PH_HANDOVER = 7 };

// ---------------------------------------------------------------- kd-tree
// Traversal stack of one wave.  An entry is (far child, t at the split plane); the far
// bound that kdtree.py:152 pushes along with them is always the t of the entry below (or
// the root box's exit distance for the bottom entry), so it is recovered on pop instead
// of being stored.  The first LDS_STACK entries of every lane live in LDS, lane-
// interleaved (conflict-free); deeper entries go to per-lane private memory.
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 10
#endif
constexpr int LDS_STACK = RT_LDS_STACK;
struct KdStack {
    int32_t* node;      // LDS, [LDS_STACK][64]
    double* t;          // LDS, [LDS_STACK][64]
    int32_t* slot;      // LDS, [64]: lane ids by rank while walks are handed over (see kd_query_wave)
    bool tiny_ok;       // trees of a few levels may be scanned by single lanes throughout (a constant
                        // false in the kernel that is only launched for deep trees drops that code)
    bool fused_always;  // whole-frame area-light kernel: the host has shown that every wave takes the path
                        // that traces both shadow rays of a hit in one shadow-tree query (direct_light_wave);
                        // a constant true leaves the general path out of the kernel
    bool prefilter_ok;  // shadow rays that end inside a tree's box may ask the shadow tree first (a
                        // constant false keeps that second copy of the walk out of a kernel)
};
// Shadow queries keep their rounds short so that lanes do not wait for the one with the
// longest walk and idle lanes can be handed part of it.  1080p dragon frame: skips 1: 15.95,
// 2: 16.46, 4: 17.20, 8: 17.79, unbounded 18.03 ms; node loads unbounded 15.97, 4: 15.76,
// 2: 15.77, 1: 16.43 ms.
#ifndef RT_MAX_SKIPS
#define RT_MAX_SKIPS 1            // leaves a lane may skip by their boxes before its round ends
#endif
#ifndef RT_MAX_LOADS
#define RT_MAX_LOADS 3            // node records a lane may load on the way down before its round ends
#endif
#ifndef RT_STEAL_PASSES
#define RT_STEAL_PASSES 1         // hand-over passes per round (a giver gives one stack entry per pass)
#endif
#ifndef RT_STEAL_MIN_PAIRS
#define RT_STEAL_MIN_PAIRS 1      // hand walks over only when at least this many lanes can take one (1: 17.72, 4: 17.81, 12: 18.36, never: 18.19 ms)
#endif
// the same two limits for the single-precision walk of shadow trees
#ifndef RT_TINY_DEPTH
#define RT_TINY_DEPTH 8           // trees of fewer levels than this are scanned by single lanes throughout (Cornell box, area light, 16 spp: 0.253 -> 0.207 ms; path traced: 4.61 -> 4.44 ms)
#endif
#ifndef RT_ST_MAX_SKIPS
#define RT_ST_MAX_SKIPS 1
#endif
#ifndef RT_ST_MAX_LOADS
#define RT_ST_MAX_LOADS 3
#endif

// BoundingBox.hit (geometry.py:49-70) on the tree's root box.
__device__ __forceinline__ bool root_box(const rtamd_tree& T, const Ray& r, double& tmin, double& tmax) {
    const double dx = 1.0 / r.d.x, dy = 1.0 / r.d.y, dz = 1.0 / r.d.z;
    const double t1 = (T.bbox_lo[0] - r.o.x) * dx, t2 = (T.bbox_hi[0] - r.o.x) * dx;
    const double t3 = (T.bbox_lo[1] - r.o.y) * dy, t4 = (T.bbox_hi[1] - r.o.y) * dy;
    const double t5 = (T.bbox_lo[2] - r.o.z) * dz, t6 = (T.bbox_hi[2] - r.o.z) * dz;
    tmin = pymax(pymax(pymin(t1, t2), pymin(t3, t4)), pymin(t5, t6));
    tmax = pymin(pymin(pymax(t1, t2), pymax(t3, t4)), pymax(t5, t6));
    if (tmax < 0) return false;
    if (tmin > tmax) return false;
    return true;
}

struct TriHit { double t, alpha, beta; int32_t ref; };

// Leaves of at most W rows (W = 8, 16 or 32), 64/W rays per step: the wave is split into
// groups of W lanes, each group takes the ray of one pending lane (ds_bpermute), tests
// that lane's leaf with one row per group lane and reduces inside the group; the pending
// lane then collects its group's result.  Rows of such leaves are stored in scan order,
// so the scan position of a row is its lane's position in the group.
template <int COUNT, bool SHADOW, int W>
__device__ __forceinline__ void packed_leaves(const rtamd_scene_desc& S, bool flagged, unsigned long long pending,
                                              const Ray& r, int32_t first, int32_t count, Work<COUNT>& w,
                                              bool& found, TriHit& best) {
    constexpr int R = 64 / W;
    const int lane = lane_id();
    const int grp = lane / W, sub = lane % W;
    const TriTable geom = tri_table(S);
    while (pending) {
        int owner = -1, my_group = -1;
#pragma unroll
        for (int g = 0; g < R; ++g) {
            if (pending) {
                const int o = first_lane(pending);
                pending &= pending - 1;
                if (g == grp) owner = o;
                if (o == lane) my_group = g;
            }
        }
        const int src = owner < 0 ? lane : owner;
        Ray q;
        q.o = v3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
        q.d = v3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
        const int32_t qfirst = __shfl(first, src);
        const int32_t src_count = __shfl(count, src);     // unconditional: every lane must take part
        const int32_t qcount = owner < 0 ? 0 : src_count;
        const bool in = sub < qcount;
        const int32_t ref = qfirst + (in ? sub : 0);
        bool usable = in;
        if (SHADOW && flagged) usable = in && !(S.tri_prim[ref] & NO_SHADOW_BIT);         // geometry.py:433-434
        double a = 0, b = 0, t = 0;
        const bool hit = usable && tri_solve_ref(geom, ref, q, a, b, t);
        w.x_rows(usable ? 1u : 0u, 72);
        w.x_tests(usable ? 1u : 0u);
        const unsigned long long hm = __ballot(hit);
        const unsigned long long group_mask = (1ull << W) - 1ull;
        bool g_found = false;
        double g_t = 0, g_a = 0, g_b = 0;
        int g_sub = 0;
        if (hm) {
            if (SHADOW) {
                const uint32_t gb = (uint32_t)((hm >> (grp * W)) & group_mask);
                g_found = gb != 0;
                g_sub = g_found ? (int)__builtin_ctz(gb) : 0;
                g_t = __shfl(t, grp * W + g_sub);
            } else {
                double bt = hit ? t : T_INF * 4.0;
                int bs = sub;
                for (int off = W / 2; off > 0; off >>= 1) {
                    const double ot = __shfl_xor(bt, off);
                    const int os = __shfl_xor(bs, off);
                    const bool take = ot < bt || (ot == bt && os < bs);
                    bt = take ? ot : bt;
                    bs = take ? os : bs;
                }
                g_found = bt < T_INF * 4.0;
                g_sub = bs;
                g_t = bt;
                g_a = __shfl(a, grp * W + bs);
                g_b = __shfl(b, grp * W + bs);
            }
        }
        uint32_t g_tests = 0;
        if (COUNT == 1 && SHADOW) {
            const uint32_t ub = (uint32_t)((__ballot(usable) >> (grp * W)) & group_mask);
            g_tests = g_found ? (uint32_t)__popc(ub & ((2u << g_sub) - 1u)) : (uint32_t)__popc(ub);
        }
        // hand the group results to the lanes that own the rays
        const int from = (my_group < 0 ? 0 : my_group) * W;
        if (COUNT == 1) {
            const uint32_t n = SHADOW ? (uint32_t)__shfl((int)g_tests, from) : (uint32_t)count;
            if (my_group >= 0) w.tri(n);
        }
        if (hm) {
            const int o_found = __shfl((int)g_found, from);
            const double o_t = __shfl(g_t, from);
            const int o_sub = __shfl(g_sub, from);
            double o_a = 0, o_b = 0;
            if (!SHADOW) { o_a = __shfl(g_a, from); o_b = __shfl(g_b, from); }
            if (my_group >= 0 && o_found) {
                best.t = o_t; best.alpha = o_a; best.beta = o_b; best.ref = first + o_sub; found = true;
            }
        }
    }
}

// Slab test of a box given as lo.xyz, hi.xyz (six contiguous doubles) against the ray's
// line; conservative (fmin/fmax drop the NaNs of 0*inf on a slab boundary).  tn = entry
// distance.
__device__ __forceinline__ bool box_touch(const double* bx, const Ray& q, V3 inv, double& tn) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = reinterpret_cast<const d2*>(bx);
    const d2 a = p[0], b = p[1], c = p[2];          // lo.x lo.y | lo.z hi.x | hi.y hi.z
    const double x0 = (a.x - q.o.x) * inv.x, x1 = (b.y - q.o.x) * inv.x;
    const double y0 = (a.y - q.o.y) * inv.y, y1 = (c.x - q.o.y) * inv.y;
    const double z0 = (b.x - q.o.z) * inv.z, z1 = (c.y - q.o.z) * inv.z;
    tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    return !(tn > tf) && !(tf < 0.0);
}

// The same tests on a box given as six floats lo.xyz, hi.xyz inside two 16-byte quads
// q0 = f[0..3], q1 = f[4..7] starting at float `at` (0 or 2) of q0: boxes that only cull are
// kept in single precision, rounded outward, which halves what a lane loads for them.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool box_touch6(double lx, double ly, double lz, double hx, double hy, double hz,
                                           const Ray& q, V3 inv, double& tn, double& tf) {
    const double x0 = (lx - q.o.x) * inv.x, x1 = (hx - q.o.x) * inv.x;
    const double y0 = (ly - q.o.y) * inv.y, y1 = (hy - q.o.y) * inv.y;
    const double z0 = (lz - q.o.z) * inv.z, z1 = (hz - q.o.z) * inv.z;
    tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    return !(tn > tf) && !(tf < 0.0);
}
// box in floats 0..5 of the 32 bytes at p
__device__ __forceinline__ bool box_touch_f(const float* p, const Ray& q, V3 inv, double& tn, double& tf) {
    const f4* v = reinterpret_cast<const f4*>(p);
    const f4 a = v[0], b = v[1];
    return box_touch6(a.x, a.y, a.z, a.w, b.x, b.y, q, inv, tn, tf);
}
// box in floats 6..11 of the 48 bytes at p (the group box of a chunk_tree row)
__device__ __forceinline__ bool box_touch_f_second(const float* p, const Ray& q, V3 inv, double& tn, double& tf) {
    const f4* v = reinterpret_cast<const f4*>(p);
    const f4 b = v[1], c = v[2];
    return box_touch6(b.z, b.w, c.x, c.y, c.z, c.w, q, inv, tn, tf);
}

// KDTree.hit's scan of one boxed leaf (kdtree.py:169-173) by the lane that owns the ray:
// chunk boxes, then the boxes of a touched chunk's 8-row cells, then the rows of a touched
// cell.  A box the ray enters beyond the best t so far cannot hold a closer or equal hit.
// Meant for waves whose rays are close together (primary rays of a tile): their lanes read
// the same boxes and rows, so the loads coalesce.
template <int COUNT>
__device__ __forceinline__ void boxed_leaf_closest_lane(const rtamd_scene_desc& S, const Ray& r, int32_t first,
                                                        int32_t count, int32_t cbase, Work<COUNT>& w, bool& found,
                                                        TriHit& best) {
    const V3 inv = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    const int32_t n_chunks = (count + 63) >> 6;
    int32_t best_ord = 0;
    for (int32_t j = 0; j < n_chunks; ++j) {
        const double* rec = S.leaf_boxes + (int64_t)(cbase + j) * 54;
        double tn;
        w.x_box(48);
        if (!box_touch(rec, r, inv, tn) || (found && tn > best.t)) continue;
        const int32_t rows_here = min(64, count - (j << 6));
        for (int32_t k = 0; k * 8 < rows_here; ++k) {
            w.x_box(48);
            if (!box_touch(rec + 6 * (1 + k), r, inv, tn) || (found && tn > best.t)) continue;
            const int32_t r0 = first + (j << 6) + k * 8;
            const int32_t nr = min(8, rows_here - k * 8);
            w.x_rows((uint32_t)nr, 80);
            w.x_tests((uint32_t)nr);
            for (int32_t i = 0; i < nr; ++i) {
                const TriRow row = load_row(S, r0 + i);
                double a, b, t;
                if (tri_solve_v(row.v0, row.e1, row.e2, r, a, b, t) &&
                    (!found || t < best.t || (t == best.t && row.order < best_ord))) {
                    best.t = t; best.alpha = a; best.beta = b; best.ref = r0 + i; best_ord = row.order; found = true;
                }
            }
        }
    }
}

// The single-precision walk of a shadow tree (below) is only as exact as the point it starts from is in
// single precision: coordinates much larger than the tree's own would move the walk by more than
// the slack the tree was built with.  A ray whose origin lies farther than 16 times the tree's
// largest coordinate from the coordinate origin therefore starts its walk from a point moved along
// the ray to within about that largest coordinate of the tree's box (walk_shift: the distance, as
// a float; 0 for other rays); the triangle tests keep the ray as it is.
__device__ __forceinline__ double tree_reach(const rtamd_tree& T) {
    return fmax(fmax(fmax(fabs(T.bbox_lo[0]), fabs(T.bbox_hi[0])), fmax(fabs(T.bbox_lo[1]), fabs(T.bbox_hi[1]))),
                fmax(fabs(T.bbox_lo[2]), fabs(T.bbox_hi[2])));
}
__device__ __forceinline__ float walk_shift(const rtamd_tree& T, const Ray& q, double near) {
    const double reach = tree_reach(T);
    if (fmax(fmax(fabs(q.o.x), fabs(q.o.y)), fabs(q.o.z)) <= 16.0 * reach) return 0.0f;
    const double dmax = fmax(fmax(fabs(q.d.x), fabs(q.d.y)), fabs(q.d.z));
    return (float)fmax(0.0, near - reach / dmax);
}

// Shadow queries that any hit decides, on a tree's shadow tree (rtamd_tree::shadow_root).
//
// Such a query asks only whether some triangle is hit anywhere on the ray inside the tree's box:
// neither the order of the leaves nor the leaf that lists the triangle matters, only that the
// exact test runs on every triangle the ray can hit.  The walk is therefore free to be
// approximate as long as it errs towards visiting: it runs in single precision (origin,
// reciprocal direction, plane distances, interval bounds and leaf boxes), which halves the
// registers and the instructions of a step.  What makes that safe is in the tree, not here: the
// host lists a triangle in every leaf whose cell its box reaches into after widening the box
// by far more (4e-5 of the tree's largest coordinate) than the walk's rounding can move a point
// (about 4e-6 of it for a starting point within 16 times that coordinate, see walk_shift), and
// widens the leaf boxes by the same amount.  The triangle tests are
// the exact double-precision ones.
//
// Rounds, short rounds, hand-over of the bottom stack entry to idle lanes and packed leaf steps
// work as in kd_query_wave below.  A stack entry is (node, t at the plane as a float) in eight
// bytes, the first LDS_STACK of every lane in the LDS array the double-precision walk keeps its
// t values in.
//
// A lane may bring a second ray (TWO): it starts on it as soon as its part of the first ray's
// walk is done, which keeps more lanes walking per round than two queries in a row would.
// Rays are told apart by `owner` = lane + 64 * (second ray).
template <int COUNT, bool TWO = false>
__device__ __forceinline__ void shadow_tree_wave(const rtamd_scene_desc& S, const rtamd_tree& T, bool walking, const Ray& ray_in,
                                 double near_in, double exit_in, bool unlimited, const KdStack& stk,
                                 Work<COUNT>& w, bool& found, TriHit& best,
                                 bool pending2 = false, const Ray* ray2 = nullptr, double near2 = 0, double exit2 = 0,
                                 bool* found2 = nullptr) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int lane = lane_id();
    Ray r = ray_in;                         // own ray, or the one whose walk this lane took over
    // the walk's own parameter starts `shift` along the ray (walk_shift): its origin, and every distance
    // it keeps, are taken from there
    float shift = walking ? walk_shift(T, r, near_in) : 0.0f;
    float ox, oy, oz;
    auto walk_origin = [&]() {
        ox = (float)r.o.x; oy = (float)r.o.y; oz = (float)r.o.z;
        if (shift != 0.0f) {
            ox = (float)(r.o.x + (double)shift * r.d.x); oy = (float)(r.o.y + (double)shift * r.d.y); oz = (float)(r.o.z + (double)shift * r.d.z);
        }
    };
    walk_origin();
    auto reciprocal = [](double d) { return fminf(fmaxf(__builtin_amdgcn_rcpf((float)d), -3.0e37f), 3.0e37f); };
    float ix = reciprocal(r.d.x), iy = reciprocal(r.d.y), iz = reciprocal(r.d.z);
    float t_near = (float)(near_in - (double)shift), t_far = (float)(exit_in - (double)shift), t_exit = t_far;
    // the same for the lane's second ray
    const float shift2 = (TWO && pending2) ? walk_shift(T, *ray2, near2) : 0.0f;
    const float near2w = TWO ? (float)(near2 - (double)shift2) : 0.0f, exit2w = TWO ? (float)(exit2 - (double)shift2) : 0.0f;
    u32x2* lds = reinterpret_cast<u32x2*>(stk.t);                   // [LDS_STACK][64] entries
    u32x2 deep[STACK_MAX - LDS_STACK];
    int sp = 0, sb = 0, owner = lane;
    int32_t node = T.shadow_root;
    bool out_found = false, out_found2 = false;
    double out_t = 0;
    found = false;
    PhaseClock clk;
    clk.start(unlimited ? 36 : 8);
    clk.count(PH_QUERIES);
#ifdef RT_PHASE_TIMERS
    // walk statistics: fresh = 1 while a walk that started at the root has not reached its first leaf, 2 the same
    // for a walk that started from a stack entry handed over; levels = interior steps of such a walk so far
    int fresh = walking ? 1 : 0, levels = 0;
    auto note_start = [&](int kind) {
        fresh = kind; levels = 0;
        if (kind == 1) { atomicAdd(&g_phase_cycles[50], 1ull); if (t_near < 0.0f) atomicAdd(&g_phase_cycles[51], 1ull); }
        else atomicAdd(&g_phase_cycles[58], 1ull);
    };
    if (walking) note_start(1);
    const int q_rays = (int)__popcll(__ballot(walking)) + (int)__popcll(__ballot(TWO && pending2));
    const long long q_start = clock64();
    unsigned long long q_rounds = 0, q_lane_rounds = 0;
#define RT_NOTE_START(kind) note_start(kind)
#else
#define RT_NOTE_START(kind)
#endif

    auto entry_at = [&](int i) -> u32x2 { return i < LDS_STACK ? lds[i * 64 + lane] : deep[i - LDS_STACK]; };
    auto pop = [&]() {
        sp--;
        const u32x2 e = entry_at(sp);
        node = (int32_t)e.x;
        t_near = __uint_as_float(e.y);
        t_far = sp == sb ? t_exit : __uint_as_float(entry_at(sp - 1).y);
    };

    while (__ballot(walking || (TWO && pending2))) {
        if (TWO && pending2 && !walking) {
            // this lane's part of its first ray's walk is done: on to its second ray
            pending2 = false;
            r = *ray2;
            shift = shift2;
            walk_origin();
            ix = reciprocal(r.d.x); iy = reciprocal(r.d.y); iz = reciprocal(r.d.z);
            node = T.shadow_root;
            t_near = near2w; t_far = t_exit = exit2w;
            sp = 0; sb = 0; owner = lane + 64;
            found = false; walking = true;
            RT_NOTE_START(1);
        }
        if (TWO) {
            // second rays that wait behind a first ray still being walked go to lanes with nothing to do
            const unsigned long long idle_m = ~__ballot(walking);
            const unsigned long long donor_m = __ballot(walking && pending2);
            const int n_pairs = min((int)__popcll(idle_m), (int)__popcll(donor_m));
            if (n_pairs > 0) {
                const int rank_idle = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                const int rank_donor = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(donor_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donor_m, 0u));
                const bool gives = walking && pending2 && rank_donor < n_pairs;
                const bool takes = !walking && rank_idle < n_pairs;
                if (gives) stk.slot[rank_donor] = lane;
                __syncthreads();                    // one wave per workgroup: orders the LDS write and read
                int src = lane;
                if (takes) src = stk.slot[rank_idle];
                __syncthreads();
                const V3 no = v3(__shfl(ray2->o.x, src), __shfl(ray2->o.y, src), __shfl(ray2->o.z, src));
                const V3 nd = v3(__shfl(ray2->d.x, src), __shfl(ray2->d.y, src), __shfl(ray2->d.z, src));
                const float n_near = __shfl(near2w, src), n_exit = __shfl(exit2w, src), n_shift = __shfl(shift2, src);
                if (gives) pending2 = false;
                if (takes) {
                    r.o = no; r.d = nd; owner = src + 64;
                    shift = n_shift;
                    walk_origin();
                    ix = reciprocal(nd.x); iy = reciprocal(nd.y); iz = reciprocal(nd.z);
                    node = T.shadow_root; t_near = n_near; t_far = n_exit; t_exit = n_exit;
                    sp = 0; sb = 0;
                    found = false; walking = true;
                    RT_NOTE_START(1);
                }
            }
        }
        {
            // hand the bottom stack entry - the farthest, largest pending subtree - of lanes still
            // walking to lanes that have their answer
            const unsigned long long idle_m = ~__ballot(walking);
            const bool can_give = walking && sp > sb;
            const unsigned long long donor_m = __ballot(can_give);
            const int n_pairs = min((int)__popcll(idle_m), (int)__popcll(donor_m));
            if (n_pairs >= RT_STEAL_MIN_PAIRS) {
                const int rank_idle = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                const int rank_donor = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(donor_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donor_m, 0u));
                const bool gives = can_give && rank_donor < n_pairs;
                const bool takes = !walking && rank_idle < n_pairs;
                if (gives) stk.slot[rank_donor] = lane;
                __syncthreads();                    // one wave per workgroup: orders the LDS write and read
                int src = lane;
                if (takes) src = stk.slot[rank_idle];
                __syncthreads();
                u32x2 g = {0u, 0u};
                float g_exit = 0;
                if (gives) {
                    g = entry_at(sb);
                    g_exit = t_exit;
                    t_exit = __uint_as_float(g.y);  // what remains with the giver ends where the given part begins
                    sb++;
                }
                const V3 no = v3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
                const V3 nd = v3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
                const int n_owner = __shfl(owner, src);
                const int32_t n_node = __shfl((int)g.x, src);
                const float n_t = __uint_as_float((unsigned)__shfl((int)g.y, src)), n_exit = __shfl(g_exit, src);
                const float n_shift = __shfl(shift, src);
                if (takes) {
                    r.o = no; r.d = nd; owner = n_owner;
                    shift = n_shift;
                    walk_origin();
                    ix = reciprocal(nd.x); iy = reciprocal(nd.y); iz = reciprocal(nd.z);
                    node = n_node; t_near = n_t; t_far = n_exit; t_exit = n_exit;
                    sp = 0; sb = 0;
                    found = false; walking = true;
                    RT_NOTE_START(2);
                }
            }
        }
        clk.lap(PH_HANDOVER);
#ifdef RT_PHASE_TIMERS
        {
            const unsigned long long n_walking = __popcll(__ballot(walking));
            const unsigned long long n_fresh1 = __popcll(__ballot(walking && fresh == 1)), n_fresh2 = __popcll(__ballot(walking && fresh == 2));
            q_rounds++; q_lane_rounds += n_walking;
            if (lane == 0) {
                atomicAdd(&g_phase_cycles[28], 1ull); atomicAdd(&g_phase_cycles[29], n_walking);
                atomicAdd(&g_phase_cycles[48], n_fresh1); atomicAdd(&g_phase_cycles[49], n_fresh2);
            }
        }
#endif
        int32_t first = 0, count = 0;
        bool parked = false;
        int skips = 0, loads = 0;
        if (walking) {
            auto step = [&](const rtamd_kdnode& nd) {
                w.node();
#ifdef RT_PHASE_TIMERS
                levels++;
#endif
                const int ax = nd.b;
                const float split = (float)nd.split;
                const float o = axis_of_f(ox, oy, oz, ax);
                const float inv = axis_of_f(ix, iy, iz, ax);
                const float t = (split - o) * inv;
                int32_t near_n, far_n;
                if (split >= o) { near_n = node + 1; far_n = nd.a; }
                else            { near_n = nd.a;     far_n = node + 1; }
                // (the sign bit, not t < 0: an origin that lies on the plane in single precision gives t = -0 for a
                // ray that leaves it towards the near side and +0 for one that leaves it towards the far side)
                if (t >= t_far || __float_as_int(t) < 0) {
                    node = near_n;
                } else if (t <= t_near) {
                    node = far_n;
                } else {
                    const u32x2 e = {(unsigned)far_n, __float_as_uint(t)};
                    if (sp < LDS_STACK) lds[sp * 64 + lane] = e;
                    else deep[sp - LDS_STACK] = e;
                    sp++;
                    node = near_n;
                    t_far = t;
                }
            };
            for (;;) {
                const rtamd_kdnode* rec = S.kd_triples + (int64_t)node * 4;
                rtamd_kdnode nd = rec[0], left = rec[1], right = rec[2];
                w.x_record();
                while ((nd.b & 3) != 3) {
                    const int32_t left_index = node + 1;
                    step(nd);
                    nd = node == left_index ? left : right;
                    if ((nd.b & 3) == 3) break;
                    step(nd);
                    if (++loads >= RT_ST_MAX_LOADS) { parked = true; break; }
                    rec = S.kd_triples + (int64_t)node * 4;
                    nd = rec[0]; left = rec[1]; right = rec[2];
                    w.x_record();
                }
                if (parked) break;
                w.leaf();
#ifdef RT_PHASE_TIMERS
                if (fresh) {
                    atomicAdd(&g_phase_cycles[fresh == 1 ? 55 : 59], (unsigned long long)levels);
                    if (fresh == 1) atomicAdd(&g_phase_cycles[56], (unsigned long long)sp);       // entries pushed on the way
                    fresh = 0;
                }
#endif
                first = nd.a;
                count = nd.b >> 2;
                bool touched = false;
                if (count != 0) {
                    w.x_box(32);
                    const f4* box = reinterpret_cast<const f4*>(S.leaf_aabb + (int64_t)node * 8);
                    const f4 a = box[0], b = box[1];
                    const float x0 = (a.x - ox) * ix, x1 = (a.w - ox) * ix;
                    const float y0 = (a.y - oy) * iy, y1 = (b.x - oy) * iy;
                    const float z0 = (a.z - oz) * iz, z1 = (b.y - oz) * iz;
                    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    touched = !(tn > tf) && !(tf < 0.0f);
                }
                if (touched) break;
                w.x_skip();
                if (sp == sb) { walking = false; first = 0; count = 0; break; }
                pop();
                if (++skips >= RT_ST_MAX_SKIPS) { parked = true; break; }
            }
        }
        clk.lap(PH_DESCENT);
        const bool at_leaf = walking && !parked;
        const unsigned long long p16 = __ballot(at_leaf && count <= 16);
        const unsigned long long p32 = __ballot(at_leaf && count > 16 && count <= 32);
#ifdef RT_PHASE_TIMERS
        {
            unsigned long long rest = p16;
            const unsigned long long n_rays = __popcll(rest);
            unsigned long long n_leaves = 0, biggest = 0;
            while (rest) {
                const int32_t lf = bcast_i(first, first_lane(rest));
                const unsigned long long same = __ballot(at_leaf && count <= 16 && first == lf);
                rest &= ~same;
                n_leaves++;
                biggest = max(biggest, (unsigned long long)__popcll(same));
            }
            if (lane == 0 && n_rays) {
                atomicAdd(&g_phase_cycles[52], n_rays); atomicAdd(&g_phase_cycles[53], n_leaves); atomicAdd(&g_phase_cycles[54], 1ull);
                atomicAdd(&g_phase_cycles[57], biggest);
            }
        }
#endif
        if (p16) packed_leaves<COUNT, true, 16>(S, false, p16, r, first, count, w, found, best);
        clk.lap(PH_PACK16);
        if (p32) packed_leaves<COUNT, true, 32>(S, false, p32, r, first, count, w, found, best);
        clk.lap(PH_PACK32);
        // a leaf the build could not bring below 33 rows: one ray at a time, 64 rows per step
        unsigned long long big = __ballot(at_leaf && count > 32);
        if (big) {
            const TriTable geom = tri_table(S);
            while (big) {
                const int src = first_lane(big);
                big &= big - 1;
                Ray q;
                q.o = v3(bcast_d(r.o.x, src), bcast_d(r.o.y, src), bcast_d(r.o.z, src));
                q.d = v3(bcast_d(r.d.x, src), bcast_d(r.d.y, src), bcast_d(r.d.z, src));
                const int32_t qfirst = bcast_i(first, src), qcount = bcast_i(count, src);
                for (int32_t c = 0; c < qcount; c += 64) {
                    const bool in = c + lane < qcount;
                    double a, b, t = 0;
                    w.x_rows(in ? 1u : 0u, 72);
                    w.x_tests(in ? 1u : 0u);
                    const bool hit = in && tri_solve_ref(geom, qfirst + c + lane, q, a, b, t);
                    const unsigned long long hm = __ballot(hit);
                    if (hm) {
                        const double ht = bcast_d(t, first_lane(hm));
                        if (lane == src) { best.t = ht; found = true; }
                        break;
                    }
                }
            }
            clk.lap(PH_BIG_PLAIN);
        }
        // a hit answers the ray's owner and ends every lane that walks a part of that ray
        unsigned long long hits = __ballot(walking && found);
        while (hits) {
            const int h = first_lane(hits);
            const int o = bcast_i(owner, h);
            const double ht = bcast_d(best.t, h);
            if (walking && owner == o) walking = false;
            if (lane == (o & 63)) {
                if (TWO && o >= 64) out_found2 = true;
                else { out_found = true; out_t = ht; }
            }
            hits &= __ballot(walking && found);
        }
        if (walking && !parked) {
            if (sp == sb) walking = false;
            else pop();
        }
        clk.lap(PH_POP);
    }
#ifdef RT_PHASE_TIMERS
    if (lane == 0) {
        const int b = min(15, q_rays >> 3);
        atomicAdd(&g_phase_hist[b], 1ull); atomicAdd(&g_phase_hist[16 + b], q_rounds);
        atomicAdd(&g_phase_hist[32 + b], (unsigned long long)(clock64() - q_start)); atomicAdd(&g_phase_hist[48 + b], q_lane_rounds);
    }
#endif
    found = out_found;
    best.t = out_t;
    if (TWO) *found2 = out_found2;
}

// One kd-tree query per active lane, all 64 lanes inside.
//
// The query runs in rounds.  In a round each lane walks the tree front to back exactly as
// KDTree.iterate_nodes does (kdtree.py:133-155), with a stack of its own, until it stands at
// a leaf it has to test (a leaf whose triangles' box the ray misses is passed by); then the leaves are tested:
//   * rays of a tile that have not bounced yet (`coherent`) reach the same leaves, so each
//     lane scans its own leaf from the row-major table, big leaves through their boxes;
//   * otherwise the wave tests the leaves together with coalesced loads of the
//     struct-of-arrays table: leaves of up to 16 / 32 rows four / two rays per step,
//     bigger leaves 64 rows per step, only the 64-row chunks a ray's line can touch, and
//     shadow rays that stand in the same leaf share the load of a chunk.
// Shadow queries of non-counting launches keep their rounds short and hand parts of long
// walks to lanes that have their answer already (see below).
// SHADOW = false is KDTree.hit (kdtree.py:166-174): every item of the leaf is tested, the
// strict minimum t wins, ties go to the earlier item, and the walk ends at the first leaf
// with a hit.  SHADOW = true is KDTree.shadow_hit (kdtree.py:157-164): the first item in
// leaf order that reports a hit ends the query and its t is returned.
template <int COUNT, bool SHADOW>
__device__ __forceinline__ void kd_query_wave(const rtamd_scene_desc& S, const rtamd_tree& T, bool active, const Ray& ray_in,
                              double limit_in, bool coherent, const KdStack& stk, Work<COUNT>& w, bool& found,
                              TriHit& best) {
    found = false;
    // a lane may come to walk part of another lane's ray (see "handing over" below)
    Ray r = ray_in;
    double limit = limit_in;
    // rows carry a does-not-cast-shadows bit; the tree says whether any row has it, so
    // that the usual mesh (all rows cast) never loads the flags, and a mesh that casts no
    // shadow at all is not walked (its walk is still counted when COUNT asks for the
    // reference's work)
    const bool flagged = SHADOW && T.shadow_casters != RTAMD_CASTERS_ALL;
    // a tree of a few levels (the Cornell box: 9 nodes, 68 rows) sits in the L1s whole: there every
    // lane scans the leaves of its own ray, scattered rays and shadow rays too - 64 scans side by
    // side cost less than the wave's packed steps, which serve four rays at a time
    const bool tiny = stk.tiny_ok && T.depth < RT_TINY_DEPTH && S.tri_rows != nullptr;
    const bool lane_scans = !SHADOW && (coherent || tiny) && S.leaf_boxes != nullptr;
    if (SHADOW && COUNT != 1 && T.shadow_casters == RTAMD_CASTERS_NONE) return;
    double t_near = 0, t_far = 0, t_exit = 0;
    bool walking = active && root_box(T, r, t_near, t_exit);
    t_far = t_exit;
    int32_t deep_node[STACK_MAX - LDS_STACK];       // only touched by trees deeper than LDS_STACK
    double deep_t[STACK_MAX - LDS_STACK];
    int sp = 0;
    int32_t node = T.node_offset;
    const int lane = lane_id();
    const TriTable geom = tri_table(S);
    // leaf boxes pay in trees with many leaves along a ray; a tree of a few levels (the Cornell
    // box: 9 nodes) is walked faster without them
    const bool use_leaf_aabb = S.leaf_aabb != nullptr && T.depth >= 8;
    V3 inv_d = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    constexpr bool STEAL = SHADOW && COUNT != 1;
    // Handing over.  A shadow ray whose every possible hit lies before its limit (no limit,
    // or the ray leaves the root box before it) is occluded exactly when any triangle on it
    // is hit, whichever part of the walk finds it.  Lanes whose own query is answered take
    // over the bottom stack entry - the farthest, largest pending subtree - of a lane still
    // walking such a ray and walk it as a query of their own; a hit by any part answers
    // the ray's owner and ends all its parts.  `sb` is the bottom of the lane's stack,
    // `owner` the lane whose ray this lane walks, out_* the answer to the lane's own ray.
    int sb = 0;
    int owner = lane;
    bool out_found = false;
    double out_t = 0;
    bool steal_ok = STEAL && walking && (t_exit * (1.0 + 1e-9) + 1e-9 < limit);
    // Whether such a ray is occluded does not depend on which leaf lists the triangle it hits:
    // it walks the tree's shadow tree (the same triangles in a kd-tree built for this walk);
    // only rays with a limit inside the tree's box go on below, on the reference's tree
    if (STEAL && S.leaf_aabb != nullptr && T.shadow_root > 0) {
        const bool fast = steal_ok;
        // A ray that ends inside the tree's box (a light among the triangles: the reference's j.py)
        // is answered by the first item in scan order that it hits, wherever on the ray that hit
        // lies - the reference's walk, below.  But when no triangle at all is hit before the limit,
        // that walk finds either nothing or a hit beyond the limit, and the ray is not occluded
        // either way: the shadow tree, walked up to the limit, proves that faster, and only rays
        // with some hit in there go on to the exact walk.  Both kinds walk the shadow tree in one query.
        const bool limited = stk.prefilter_ok && walking && !fast;
        if (__ballot(fast || limited)) {
            bool f2 = false;
            TriHit b2;
            b2.t = 0;
            const double far = fast ? t_exit : fmin(t_exit, limit * (1.0 + 1e-6) + 1e-6);
            shadow_tree_wave<COUNT>(S, T, fast || limited, r, t_near, far, limit_in > T_INF, stk, w, f2, b2);
            if (fast) { out_found = f2; out_t = b2.t; walking = false; }
            if (limited && !f2) walking = false;
        }
        if (!__ballot(walking)) { found = out_found; best.t = out_t; return; }
        steal_ok = false;
    }
    PhaseClock clk;
    clk.start(SHADOW ? (limit_in > T_INF ? 36 : 8) : 0);     // shadow queries without a limit (ambient occlusion) apart
    clk.count(PH_QUERIES);

    while (__ballot(walking)) {
        for (int pass = 0; STEAL && pass < RT_STEAL_PASSES; ++pass) {
            const unsigned long long idle_m = ~__ballot(walking);
            const bool can_give = walking && steal_ok && sp > sb;
            const unsigned long long donor_m = __ballot(can_give);
            const int n_pairs = min((int)__popcll(idle_m), (int)__popcll(donor_m));
            if (n_pairs < RT_STEAL_MIN_PAIRS) break;
            {
                const int rank_idle = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                const int rank_donor = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(donor_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donor_m, 0u));
                const bool gives = can_give && rank_donor < n_pairs;
                const bool takes = !walking && rank_idle < n_pairs;
                if (gives) stk.slot[rank_donor] = lane;
                __syncthreads();                    // one wave per workgroup: orders the LDS write and read
                int src = lane;
                if (takes) src = stk.slot[rank_idle];
                __syncthreads();
                int32_t g_node = 0;
                double g_t = 0, g_exit = 0;
                if (gives) {
                    if (sb < LDS_STACK) { g_node = stk.node[sb * 64 + lane]; g_t = stk.t[sb * 64 + lane]; }
                    else { g_node = deep_node[sb - LDS_STACK]; g_t = deep_t[sb - LDS_STACK]; }
                    g_exit = t_exit;
                    t_exit = g_t;                   // what remains with the giver now ends where the given part begins
                    sb++;
                }
                const V3 no = v3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
                const V3 nd = v3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
                const double n_limit = __shfl(limit, src);
                const int n_owner = __shfl(owner, src);
                const int32_t n_node = __shfl(g_node, src);
                const double n_t = __shfl(g_t, src), n_exit = __shfl(g_exit, src);
                if (takes) {
                    r.o = no; r.d = nd; limit = n_limit; owner = n_owner;
                    inv_d = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
                    node = n_node; t_near = n_t; t_far = n_exit; t_exit = n_exit;
                    sp = 0; sb = 0;
                    found = false; walking = true; steal_ok = true;
                }
            }
        }
        clk.lap(PH_HANDOVER);
#ifdef RT_PHASE_TIMERS
        if (SHADOW) {
            // descent rounds and the lanes walking in them
            const unsigned long long n_walking = __popcll(__ballot(walking));
            if (lane == 0) { atomicAdd(&g_phase_cycles[28], 1ull); atomicAdd(&g_phase_cycles[29], n_walking); }
        }
#endif
        int32_t first = 0, count = 0, cbase = -1;
        bool parked = false;
        int skips = 0, loads = 0;
        unsigned long long my_chunks = 0;
        bool my_any_hit_decides = true;
        bool boxes_done = false;
        // Shadow queries on boxed leaves: the lane tests its ray against the chunk boxes of its
        // leaf, leaving a mask of touched chunks.  48-byte rows of floats: the chunk's box, then
        // the box around a group of four chunks (in the row of the group's first chunk); a group
        // the ray misses is skipped whole.
        auto chunk_boxes_of_leaf = [&]() {
            const int32_t nch = (count + 63) >> 6;
            double tn, tf;
            const float* rec = S.chunk_tree + (int64_t)cbase * 12;
            my_chunks = 0;
            my_any_hit_decides = true;
            for (int32_t g = 0; g < nch; g += 4) {
                if (nch > 4) w.x_box(32);
                if (nch > 4 && !box_touch_f_second(rec + g * 12, r, inv_d, tn, tf)) continue;
                const int32_t g_end = min(g + 4, nch);
                for (int32_t j = g; j < g_end; ++j) {
                    w.x_box(32);
                    if (box_touch_f(rec + j * 12, r, inv_d, tn, tf)) {
                        my_chunks |= 1ull << j;
                        if (!(tf < limit)) my_any_hit_decides = false;
                    }
                }
            }
            boxes_done = true;
        };
        if (walking) {
            // One interior step of KDTree.iterate_nodes (kdtree.py:140-153) at `node`, whose
            // record is nd: picks the next node and pushes the far child where the ray
            // crosses the split plane inside (t_near, t_far).
            auto step = [&](const rtamd_kdnode& nd) {
                w.node();
                const int ax = nd.b;
                const double o = axis_of(r.o, ax), d = axis_of(r.d, ax);
                const double t = (nd.split - o) / d;
                int32_t near_n, far_n;
                if (nd.split >= o) { near_n = node + 1; far_n = nd.a; }
                else               { near_n = nd.a;     far_n = node + 1; }
                if (t >= t_far || t < 0) {
                    node = near_n;
                } else if (t <= t_near) {
                    node = far_n;
                } else {
                    if (sp < LDS_STACK) { stk.node[sp * 64 + lane] = far_n; stk.t[sp * 64 + lane] = t; }
                    else { deep_node[sp - LDS_STACK] = far_n; deep_t[sp - LDS_STACK] = t; }
                    sp++;
                    node = near_n;
                    t_far = t;
                }
            };
            // pops the far child pushed last; the far bound pushed with it = t of the entry
            // now on top, or the root exit
            auto pop = [&]() {
                sp--;
                if (sp < LDS_STACK) { node = stk.node[sp * 64 + lane]; t_near = stk.t[sp * 64 + lane]; }
                else { node = deep_node[sp - LDS_STACK]; t_near = deep_t[sp - LDS_STACK]; }
                if (sp == sb) t_far = t_exit;
                else if (sp - 1 < LDS_STACK) t_far = stk.t[(sp - 1) * 64 + lane];
                else t_far = deep_t[sp - 1 - LDS_STACK];
            };
            for (;;) {
                // a 64-byte record holds a node and both its children, so every load serves two levels
                const rtamd_kdnode* rec = S.kd_triples + (int64_t)node * 4;
                rtamd_kdnode nd = rec[0], left = rec[1], right = rec[2];
                w.x_record();
                while ((nd.b & 3) != 3) {
                    const int32_t left_index = node + 1;
                    step(nd);
                    nd = node == left_index ? left : right;
                    if ((nd.b & 3) == 3) break;
                    step(nd);
                    // a long descent ends the lane's round at an interior node (same reason as RT_MAX_SKIPS)
                    if (STEAL && ++loads >= RT_MAX_LOADS) { parked = true; break; }
                    rec = S.kd_triples + (int64_t)node * 4;
                    nd = rec[0]; left = rec[1]; right = rec[2];
                    w.x_record();
                }
                if (parked) break;
                w.leaf();
                first = nd.a;
                count = nd.b >> 2;
                cbase = S.chunk_boxes ? (int32_t)nd.split : -1;
                // a leaf whose triangles' box the ray misses holds no hit: the lane walks on at
                // once (not when counting: the reference tests the rows)
                if (COUNT != 1 && use_leaf_aabb) {
                    double tn;
                    double tf;
                    if (count != 0) w.x_box(32);
                    const f4* box = reinterpret_cast<const f4*>(S.leaf_aabb + (int64_t)node * 8);
                    const f4 box_a = box[0], box_b = box[1];
                    if (count == 0 || !box_touch6(box_a.x, box_a.y, box_a.z, box_a.w, box_b.x, box_b.y, r, inv_d, tn, tf)) {
                        w.x_skip();
                        if (sp == sb) { walking = false; first = 0; count = 0; cbase = -1; break; }
                        pop();
                        // a long run of skipped leaves ends the lane's round without a leaf, so
                        // that the wave's other lanes do not wait for it and part of its walk
                        // can be handed over
                        if (STEAL && ++skips >= RT_MAX_SKIPS) { parked = true; break; }
                        continue;
                    }
                    // a query that any hit decides walks on into the leaf's own subtree, if it has
                    // one (the root's index rides in the box row): small leaves with tight boxes
                    // instead of hundreds of rows
                    if (STEAL && steal_ok) {
                        const int32_t sub = __float_as_int(box_b.z);
                        if (sub > 0) {
                            node = sub;
                            continue;
                        }
                    }
                }
                break;
            }
        }

        clk.lap(PH_DESCENT);
        const bool at_leaf = walking && !parked;     // has a leaf to test in this round
        // rays of one tile that have not bounced yet run close together: their lanes reach
        // the same leaves and read the same rows, so each lane scans its own leaf; rays
        // that have scattered are tested by the wave together (packed and chunk steps below)
        const bool small = at_leaf && (lane_scans ? cbase < 0 : (count == 0 || (tiny && cbase < 0)));
        if (small) {
            if (SHADOW) {
                for (int32_t i = 0; i < count; ++i) {
                    const TriRow row = load_row(S, first + i);
                    w.x_rows(1, 80);
                    if (row.prim & NO_SHADOW_BIT) continue;                      // geometry.py:433-434
                    w.tri(1);
                    w.x_tests(1);
                    double a, b, t;
                    if (tri_solve_v(row.v0, row.e1, row.e2, r, a, b, t)) { best.t = t; found = true; break; }
                }
            } else {
                w.tri((uint32_t)count);
                w.x_rows((uint32_t)count, 80);
                w.x_tests((uint32_t)count);
                for (int32_t i = 0; i < count; ++i) {
                    const TriRow row = load_row(S, first + i);
                    double a, b, t;
                    if (tri_solve_v(row.v0, row.e1, row.e2, r, a, b, t) && (!found || best.t > t)) {
                        best.t = t; best.alpha = a; best.beta = b; best.ref = first + i; found = true;
                    }
                }
            }
        }

        // leaves of up to 16 / 32 rows: four / two rays per step (an 8-row class with eight
        // rays per step measured no faster: 2153 vs 2185 Mrays/s)
        const bool pack16 = at_leaf && !small && count <= 16;
        const bool pack32 = at_leaf && !small && !pack16 && count <= 32;
        const unsigned long long p16 = __ballot(pack16), p32 = __ballot(pack32);
#ifdef RT_PHASE_TIMERS
        if (SHADOW) {
            // sharing statistics of packed leaves: rays, distinct leaves
            unsigned long long rest = p16 | p32;
            const unsigned long long n_rays = __popcll(rest);
            unsigned long long n_leaves = 0;
            while (rest) {
                const int32_t lf = bcast_i(first, first_lane(rest));
                rest &= ~__ballot((pack16 || pack32) && first == lf);
                n_leaves++;
            }
            if (lane == 0 && n_rays) { atomicAdd(&g_phase_cycles[25], n_rays); atomicAdd(&g_phase_cycles[26], n_leaves); atomicAdd(&g_phase_cycles[27], 1ull); }
        }
#endif
        if (p16) packed_leaves<COUNT, SHADOW, 16>(S, flagged, p16, r, first, count, w, found, best);
        clk.lap(PH_PACK16);
        if (p32) packed_leaves<COUNT, SHADOW, 32>(S, flagged, p32, r, first, count, w, found, best);
        clk.lap(PH_PACK32);

        unsigned long long big = __ballot(at_leaf && !small && count > 32);
        if (lane_scans) {
            // boxed leaves as well: each lane scans its own leaf through the cell boxes
            const bool mine = at_leaf && !small && count > 32 && cbase >= 0;
            const unsigned long long pm = __ballot(mine);
            if (pm) {
                if (mine) {
                    boxed_leaf_closest_lane<COUNT>(S, r, first, count, cbase, w, found, best);
                    w.tri((uint32_t)count);
                }
                big &= ~pm;
                clk.lap(PH_BIG_BOXES);
            }
        }
        // Shadow queries on boxed leaves, first part: every lane tests its own ray against
        // the chunk boxes of its own leaf (lanes of a wave mostly share a few big leaves, so
        // these loads coalesce), leaving a mask of touched chunks.  The rays are then taken
        // one at a time below, with no box step of their own.
        const bool prepass = SHADOW && S.chunk_tree != nullptr;
        if (prepass && at_leaf && !small && cbase >= 0 && count <= 4096 && !boxes_done) chunk_boxes_of_leaf();
        // Shadow queries, second part: lanes that stand in the same leaf are served together.
        // For every chunk that any of them touches the wave loads its 64 rows once and tests
        // them against each of those rays in turn (the ray rides in scalar registers).  A
        // leaf of 33-64 rows without boxes counts as one chunk.  A ray whose boxes all end
        // before its limit is decided by any hit (see below); the others keep the hit with
        // the earliest scan position over all their chunks, as KDTree.shadow_hit reports it.
        if (SHADOW && COUNT != 1 && prepass) {
            const bool eligible = at_leaf && !small && count > 32 && (cbase < 0 ? count <= 64 : count <= 4096);
            if (eligible && cbase < 0) { my_chunks = 1; my_any_hit_decides = false; }
            const bool has_work = eligible && my_chunks != 0;
            unsigned long long rest = __ballot(has_work);
            big &= ~__ballot(eligible);
            int32_t my_ord = 0x7fffffff;
            while (rest) {
                const int s0 = first_lane(rest);
                const int32_t lf = bcast_i(first, s0), lcount = bcast_i(count, s0);
                bool open = has_work && first == lf;        // this lane's query is in this leaf and undecided
                const unsigned long long same = __ballot(open);
                rest &= ~same;
                unsigned long long uni = 0;
                for (unsigned long long m = same; m; m &= m - 1) {
                    const int s1 = first_lane(m);
                    const uint32_t lo = (uint32_t)bcast_i((int)(uint32_t)my_chunks, s1);
                    const uint32_t hi = (uint32_t)bcast_i((int)(uint32_t)(my_chunks >> 32), s1);
                    uni |= ((unsigned long long)hi << 32) | lo;
                }
                while (uni) {
                    const int32_t cj = (int32_t)__builtin_ctzll(uni);
                    uni &= uni - 1;
                    unsigned long long takers = __ballot(open && ((my_chunks >> cj) & 1ull));
                    if (!takers) continue;
                    const int32_t i = (cj << 6) + lane;
                    const bool in = i < lcount;
                    const int32_t ref = lf + (in ? i : 0);
                    bool usable = in;
                    if (flagged) usable = in && !(S.tri_prim[ref] & NO_SHADOW_BIT);               // geometry.py:433-434
                    const int32_t off = ref * 8;
                    const V3 v0 = v3(tri_component(geom, off, 0), tri_component(geom, off, 1), tri_component(geom, off, 2));
                    const V3 e1 = v3(tri_component(geom, off, 3), tri_component(geom, off, 4), tri_component(geom, off, 5));
                    const V3 e2 = v3(tri_component(geom, off, 6), tri_component(geom, off, 7), tri_component(geom, off, 8));
                    w.x_rows(in ? 1u : 0u, 72);
                    while (takers) {
                        const int rl = first_lane(takers);
                        takers &= takers - 1;
                        Ray q;
                        q.o = v3(bcast_d(r.o.x, rl), bcast_d(r.o.y, rl), bcast_d(r.o.z, rl));
                        q.d = v3(bcast_d(r.d.x, rl), bcast_d(r.d.y, rl), bcast_d(r.d.z, rl));
                        double a, b, t;
                        const bool hit = usable && tri_solve_v(v0, e1, e2, q, a, b, t);
                        w.x_tests(usable ? 1u : 0u);
                        const unsigned long long hm = __ballot(hit);
                        if (!hm) continue;
                        if (bcast_i((int)my_any_hit_decides, rl) != 0) {
                            const double ht = bcast_d(t, first_lane(hm));
                            if (lane == rl) { best.t = ht; found = true; open = false; }
                        } else {
                            const int32_t ord = hit ? S.tri_order[ref] : 0x7fffffff;
                            int wl = first_lane(hm);
                            for (;;) {
                                const int32_t ow = bcast_i(ord, wl);
                                const unsigned long long closer = __ballot(hit && ord < ow);
                                if (!closer) break;
                                wl = first_lane(closer);
                            }
                            const int32_t ow = bcast_i(ord, wl);
                            const double ht = bcast_d(t, wl);
                            if (lane == rl && ow < my_ord) { my_ord = ow; best.t = ht; found = true; }
                        }
                    }
                }
            }
            clk.lap(PH_BIG_BOXES);
        }
#ifdef RT_PHASE_TIMERS
        if (SHADOW && prepass) {
            // sharing statistics of boxed leaves: rays, distinct leaves, (ray, chunk) pairs, distinct (leaf, chunk) pairs
            unsigned long long rest = __ballot(at_leaf && !small && count > 32 && cbase >= 0);
            unsigned long long n_rays = __popcll(rest), n_leaves = 0, n_pairs = 0, n_distinct = 0;
            while (rest) {
                const int s0 = first_lane(rest);
                const int32_t lf = bcast_i(first, s0);
                unsigned long long same = __ballot(at_leaf && !small && count > 32 && cbase >= 0 && first == lf);
                rest &= ~same;
                n_leaves++;
                unsigned long long uni = 0;
                while (same) {
                    const int s1 = first_lane(same);
                    same &= same - 1;
                    const uint32_t lo = (uint32_t)bcast_i((int)(uint32_t)my_chunks, s1);
                    const uint32_t hi = (uint32_t)bcast_i((int)(uint32_t)(my_chunks >> 32), s1);
                    const unsigned long long mc = ((unsigned long long)hi << 32) | lo;
                    n_pairs += __popcll(mc);
                    uni |= mc;
                }
                n_distinct += __popcll(uni);
            }
            if (lane == 0 && n_rays) {
                atomicAdd(&g_phase_cycles[20], n_rays);
                atomicAdd(&g_phase_cycles[21], n_leaves);
                atomicAdd(&g_phase_cycles[22], n_pairs);
                atomicAdd(&g_phase_cycles[23], n_distinct);
                atomicAdd(&g_phase_cycles[24], 1ull);
            }
        }
#endif
        while (big) {
            const int src = first_lane(big);
            big &= big - 1;
            Ray q;
            q.o = v3(bcast_d(r.o.x, src), bcast_d(r.o.y, src), bcast_d(r.o.z, src));
            q.d = v3(bcast_d(r.d.x, src), bcast_d(r.d.y, src), bcast_d(r.d.z, src));
            const int32_t qfirst = bcast_i(first, src), qcount = bcast_i(count, src);
            const int32_t qbox = bcast_i(cbase, src);
            const int32_t n_chunks = (qcount + 63) >> 6;
            // Per-lane best over the rows this lane tests: for a closest-hit query the
            // minimum t (earlier scan position on ties), for a shadow query the hit with
            // the earliest scan position.  `lord` is the row's position in the reference's
            // scan order, which is what ties and "first item that hits" are decided by.
            bool lfound = false;
            double lt = 0, la = 0, lb = 0;
            int32_t lref = 0, lord = 0;
            uint32_t tests = 0;
            bool stop = false;

            auto test_chunk = [&](int32_t c) {
                const int32_t i = c + lane;
                const bool in = i < qcount;
                const int32_t ref = qfirst + i;
                bool usable = in;
                if (SHADOW && flagged) usable = in && !(S.tri_prim[in ? ref : qfirst] & NO_SHADOW_BIT);   // geometry.py:433-434
                double a, b, t;
                w.x_rows(usable ? 1u : 0u, 72);
                w.x_tests(usable ? 1u : 0u);
                if (usable && tri_solve_ref(geom, ref, q, a, b, t)) {
                    const int32_t ord = S.tri_order ? S.tri_order[ref] : ref;
                    const bool better = SHADOW ? (!lfound || ord < lord)
                                               : (!lfound || t < lt || (t == lt && ord < lord));
                    if (better) { lt = t; la = a; lb = b; lref = ref; lord = ord; lfound = true; }
                }
                if (COUNT == 1 && SHADOW && qbox < 0) {
                    // rows are in scan order here: the reference stops at the first hit
                    const unsigned long long hm = __ballot(lfound), um = __ballot(usable);
                    if (hm) { tests += (uint32_t)__popcll(um & ((2ull << first_lane(hm)) - 1ull)); }
                    else tests += (uint32_t)__popcll(um);
                }
            };

            if (qbox >= 0 && prepass && qcount <= 4096) {
                // chunks named by the lane's own box tests above; when every touched box ends
                // before `limit` any hit decides the query (see below) and the first chunk
                // with a hit ends it
                const uint32_t lo = (uint32_t)bcast_i((int)(uint32_t)my_chunks, src);
                const uint32_t hi = (uint32_t)bcast_i((int)(uint32_t)(my_chunks >> 32), src);
                unsigned long long todo = ((unsigned long long)hi << 32) | lo;
                const bool any_hit_decides = COUNT != 1 && bcast_i((int)my_any_hit_decides, src) != 0;
                while (todo) {
                    const int32_t c = (int32_t)__builtin_ctzll(todo) << 6;
                    todo &= todo - 1;
                    test_chunk(c);
                    if (any_hit_decides && __ballot(lfound)) break;
                }
            } else if (qbox >= 0) {
                // leaf with chunk boxes: lane j tests the box of chunk j against the ray's line,
                // then the wave visits the touched chunks.  Shadow queries: when every touched
                // box ends before `limit` (always, without a limit) any hit decides the query
                // the same way as the first one in scan order would (light.py:34-35,
                // :127-128), so the first chunk with a hit ends it; otherwise all touched
                // chunks are scanned for the earliest item.
                const double ix = 1.0 / q.d.x, iy = 1.0 / q.d.y, iz = 1.0 / q.d.z;
                const double* __restrict__ cb = S.chunk_boxes;
                const int64_t nc = S.n_chunks;
                const double qlimit = SHADOW ? bcast_d(limit, src) : 0.0;
                for (int32_t c0 = 0; c0 < n_chunks && !stop; c0 += 64) {
                    const int32_t j = c0 + lane;
                    bool touch = false;
                    double tf = 0;
                    if (j < n_chunks) {
                        w.x_box(48);
                        const double* bx = cb + qbox + j;
                        const double x0 = (bx[0] - q.o.x) * ix, x1 = (bx[3 * nc] - q.o.x) * ix;
                        const double y0 = (bx[nc] - q.o.y) * iy, y1 = (bx[4 * nc] - q.o.y) * iy;
                        const double z0 = (bx[2 * nc] - q.o.z) * iz, z1 = (bx[5 * nc] - q.o.z) * iz;
                        // fmin/fmax drop NaNs (0*inf on a slab boundary), which keeps the test conservative
                        const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
                        tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
                        touch = !(tn > tf) && !(tf < 0.0);
                    }
                    unsigned long long todo = __ballot(touch);
                    const bool any_hit_decides = SHADOW && COUNT != 1 && n_chunks <= 64 && !__ballot(touch && !(tf < qlimit));
                    while (todo) {
                        const int32_t c = (c0 + first_lane(todo)) << 6;
                        todo &= todo - 1;
                        test_chunk(c);
                        if (any_hit_decides && __ballot(lfound)) { stop = true; break; }
                    }
                }
            } else {
                for (int32_t c = 0; c < qcount && !stop; c += 64) {
                    test_chunk(c);
                    // rows in scan order: a shadow query is decided by the first chunk with a hit
                    if (SHADOW && __ballot(lfound)) stop = true;
                }
            }

            unsigned long long cand = __ballot(lfound);
            int winner = 0;
            if (cand) {
                // minimum over the lanes: (t, scan position) for closest hits, scan position for shadows
                winner = first_lane(cand);
                for (;;) {
                    const double tw = bcast_d(lt, winner);
                    const int32_t ow = bcast_i(lord, winner);
                    cand = SHADOW ? __ballot(lfound && lord < ow)
                                  : __ballot(lfound && (lt < tw || (lt == tw && lord < ow)));
                    if (!cand) break;
                    winner = first_lane(cand);
                }
                const double wt = bcast_d(lt, winner), wa = bcast_d(la, winner), wb = bcast_d(lb, winner);
                const int32_t wr = bcast_i(lref, winner);
                if (lane == src) { best.t = wt; best.alpha = wa; best.beta = wb; best.ref = wr; found = true; }
                if (COUNT == 1 && SHADOW && qbox >= 0) {
                    // what the reference's scan would have tested: shadow casters up to the hit
                    const int32_t ow = bcast_i(lord, winner);
                    for (int32_t c = 0; c < qcount; c += 64) {
                        const int32_t i = c + lane;
                        const bool counted = i < qcount && !(S.tri_prim[qfirst + (i < qcount ? i : 0)] & NO_SHADOW_BIT) &&
                                             S.tri_order[qfirst + (i < qcount ? i : 0)] <= ow;
                        tests += (uint32_t)__popcll(__ballot(counted));
                    }
                }
            } else if (COUNT == 1 && SHADOW && qbox >= 0) {
                for (int32_t c = 0; c < qcount; c += 64) {
                    const int32_t i = c + lane;
                    const bool counted = i < qcount && !(S.tri_prim[qfirst + (i < qcount ? i : 0)] & NO_SHADOW_BIT);
                    tests += (uint32_t)__popcll(__ballot(counted));
                }
            }
            if (lane == src) w.tri(SHADOW ? tests : (uint32_t)qcount);
            clk.lap(qbox >= 0 ? PH_BIG_BOXES : PH_BIG_PLAIN);
        }

        if (STEAL) {
            // a hit answers the ray's owner and ends every lane that walks a part of that ray
            unsigned long long hits = __ballot(walking && found);
            while (hits) {
                const int h = first_lane(hits);
                const int o = bcast_i(owner, h);
                const double ht = bcast_d(best.t, h);
                if (walking && owner == o) walking = false;
                if (lane == o) { out_found = true; out_t = ht; }
                hits &= __ballot(walking && found);
            }
        }
        if (walking && !parked) {
            if ((!STEAL && found) || sp == sb) {
                walking = false;
            } else {
                sp--;
                if (sp < LDS_STACK) { node = stk.node[sp * 64 + lane]; t_near = stk.t[sp * 64 + lane]; }
                else { node = deep_node[sp - LDS_STACK]; t_near = deep_t[sp - LDS_STACK]; }
                // the far bound pushed with this entry = t of the entry now on top, or the root exit
                if (sp == sb) t_far = t_exit;
                else if (sp - 1 < LDS_STACK) t_far = stk.t[(sp - 1) * 64 + lane];
                else t_far = deep_t[sp - 1 - LDS_STACK];
            }
        }
        clk.lap(PH_POP);
    }
    if (STEAL) { found = out_found; best.t = out_t; }
}

// ---------------------------------------------------------------- analytic primitives
// Sphere.hit / shadow_hit share this root selection (geometry.py:253-296).
__device__ __forceinline__ bool sphere_t(const double* p, const Ray& r, V3& temp, double& t) {
    temp = r.o - ld3(p);
    const double a = dot(r.d, r.d);
    const double b = 2.0 * dot(temp, r.d);
    const double c = dot(temp, temp) - p[3] * p[3];
    const double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    const double e = sqrt(disc);
    const double denom = 2.0 * a;
    t = (-b - e) / denom;
    if (!(t > EPS)) t = (-b + e) / denom;
    return t > EPS;
}

// Plane.hit (geometry.py:317-326); also the first half of CheckerPlane / Rectangle.
__device__ __forceinline__ bool plane_t(V3 point, V3 normal, const Ray& r, double& t) {
    const double dn = dot(r.d, normal);
    if (dn == 0) return false;
    t = dot(point - r.o, normal) / dn;
    return t > EPS;
}

// Rectangle.hit / shadow_hit (geometry.py:388-416): p0 is the centre here.
__device__ __forceinline__ bool rectangle_t(const double* p, const Ray& r, double& t) {
    if (!plane_t(ld3(p), ld3(p + 3), r, t)) return false;
    const V3 diff = (r.o + t * r.d) - ld3(p);
    const double a = fabs(dot(diff, ld3(p + 6)));
    const double b = fabs(dot(diff, ld3(p + 9)));
    return a * 2 <= p[12] && b * 2 <= p[13];
}

// AxisAlignedBox.hit (geometry.py:136-217).
__device__ __forceinline__ bool box_hit(const double* p, const Ray& r, double& tmin, V3& normal) {
    const double dx = 1.0 / r.d.x, dy = 1.0 / r.d.y, dz = 1.0 / r.d.z;
    double tx_min, tx_max, ty_min, ty_max, tz_min, tz_max;
    if (dx >= 0) { tx_min = (p[0] - r.o.x) * dx; tx_max = (p[3] - r.o.x) * dx; }
    else         { tx_min = (p[3] - r.o.x) * dx; tx_max = (p[0] - r.o.x) * dx; }
    if (dy >= 0) { ty_min = (p[1] - r.o.y) * dy; ty_max = (p[4] - r.o.y) * dy; }
    else         { ty_min = (p[4] - r.o.y) * dy; ty_max = (p[1] - r.o.y) * dy; }
    if (dz >= 0) { tz_min = (p[2] - r.o.z) * dz; tz_max = (p[5] - r.o.z) * dz; }
    else         { tz_min = (p[5] - r.o.z) * dz; tz_max = (p[2] - r.o.z) * dz; }
    double t0, t1;
    int face_in, face_out;
    if (tx_min > ty_min) { t0 = tx_min; face_in = (dx >= 0) ? 0 : 3; }
    else                 { t0 = ty_min; face_in = (dy >= 0) ? 1 : 4; }
    if (tz_min > t0)     { t0 = tz_min; face_in = (dz >= 0) ? 2 : 5; }
    if (tx_max < ty_max) { t1 = tx_max; face_out = (dx >= 0) ? 3 : 0; }
    else                 { t1 = ty_max; face_out = (dy >= 0) ? 4 : 1; }
    if (tz_max < t1)     { t1 = tz_max; face_out = (dz >= 0) ? 5 : 2; }
    if (t0 < t1 && t1 > EPS) {
        int face;
        if (t0 > EPS) { tmin = t0; face = face_in; }
        else          { tmin = t1; face = face_out; }
        const double s = face < 3 ? -1.0 : 1.0;          // FACE_ARRAY, geometry.py:116-121
        const int ax = face % 3;
        normal = v3(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
        return true;
    }
    return false;
}

// AxisAlignedBox.shadow_hit (geometry.py:219-242): a different slab form that reports
// the entry distance even when it is negative.
__device__ __forceinline__ bool box_shadow(const double* p, const Ray& r, double& t) {
    const double dx = 1.0 / r.d.x, dy = 1.0 / r.d.y, dz = 1.0 / r.d.z;
    const double t1 = (p[0] - r.o.x) * dx, t2 = (p[3] - r.o.x) * dx;
    const double t3 = (p[1] - r.o.y) * dy, t4 = (p[4] - r.o.y) * dy;
    const double t5 = (p[2] - r.o.z) * dz, t6 = (p[5] - r.o.z) * dz;
    const double tmin = pymax(pymax(pymin(t1, t2), pymin(t3, t4)), pymin(t5, t6));
    const double tmax = pymin(pymin(pymax(t1, t2), pymax(t3, t4)), pymax(t5, t6));
    if (tmax < EPS) return false;
    if (tmin > tmax) return false;
    t = tmin;
    return true;
}

// Instance.inv_ray (geometry.py:97-101): the direction is not renormalised.
__device__ __forceinline__ Ray inverse_ray(const double* m, const double* n, const Ray& r) {
    Ray q;
    q.o = v3(((m[0] * r.o.x + m[1] * r.o.y) + m[2] * r.o.z) + m[3],
             ((m[4] * r.o.x + m[5] * r.o.y) + m[6] * r.o.z) + m[7],
             ((m[8] * r.o.x + m[9] * r.o.y) + m[10] * r.o.z) + m[11]);
    q.d = v3((n[0] * r.d.x + n[1] * r.d.y) + n[2] * r.d.z,
             (n[3] * r.d.x + n[4] * r.d.y) + n[5] * r.d.z,
             (n[6] * r.d.x + n[7] * r.d.y) + n[8] * r.d.z);
    return q;
}
// Instance.hit (geometry.py:107): normal through the transposed inverse, not renormalised.
__device__ __forceinline__ V3 transposed_normal(const double* n, V3 nrm) {
    return v3((n[0] * nrm.x + n[3] * nrm.y) + n[6] * nrm.z,
              (n[1] * nrm.x + n[4] * nrm.y) + n[7] * nrm.z,
              (n[2] * nrm.x + n[5] * nrm.y) + n[8] * nrm.z);
}
// A ray into the space of the object an Instance wraps: through the outermost Instance first,
// then through the nested ones from the outside in (each Instance.hit calls the next).
__device__ __forceinline__ Ray to_object_space(const rtamd_scene_desc& S, const rtamd_object& ob, const Ray& r) {
    Ray q = inverse_ray(ob.inv_point, ob.inv_dir, r);
    for (int j = 0; j < ob.n_inner; ++j) {
        const rtamd_transform& t = S.transforms[ob.inner_offset + j];
        q = inverse_ray(t.inv_point, t.inv_dir, q);
    }
    return q;
}
// The hit's normal on its way back out: innermost Instance first.
__device__ __forceinline__ V3 normal_to_world(const rtamd_scene_desc& S, const rtamd_object& ob, V3 nrm) {
    for (int j = ob.n_inner - 1; j >= 0; --j) nrm = transposed_normal(S.transforms[ob.inner_offset + j].inv_dir, nrm);
    return transposed_normal(ob.inv_dir, nrm);
}

// ---------------------------------------------------------------- world queries
struct Hit {
    double t;
    V3 n;        // shading normal as the primitive reports it
    V3 p;        // hit_point = origin + tmin*direction (main.py:44)
    int32_t material, object, prim;
};

// World.hit_objects (main.py:35-46) for every active lane: objects in list order, strict `<`.
template <int COUNT, bool TREES>
__device__ __forceinline__ bool closest_hit_wave(const rtamd_scene_desc& S, const KdStack& stk, bool active, bool coherent, const Ray& ray,
                                 Work<COUNT>& w, Hit& h) {
    if (active) w.closest++;
    double tmin = T_INF;
    bool any = false;
    for (int i = 0; i < S.n_objects; ++i) {
        const rtamd_object& ob = S.objects[i];
        const Ray r = ob.has_transform ? to_object_space(S, ob, ray) : ray;
        const double* p = ob.p;
        double t = 0;
        V3 n = v3(0, 0, 0);
        int32_t mat = ob.material, prim = -1;
        bool hit = false;
        if (TREES && ob.kind == RTAMD_KDTREE) {
            if (__ballot(active)) {
                const rtamd_tree& T = S.trees[ob.tree];
                TriHit th;
                bool f;
                w.mark(44);     // closest hit: object and tree records, ray into object space
                kd_query_wave<COUNT, false>(S, T, active, r, 0.0, coherent, stk, w, f, th);
                w.mark(45);     // closest hit: the kd query, root box included
                if (active && f) {
                    hit = true;
                    t = th.t;
                    prim = S.tri_prim[th.ref] & (NO_SHADOW_BIT - 1);
                    const int64_t gp = (int64_t)T.prim_offset + prim;
                    const double* nn = S.prim_normals + S.prim_normal_offset[gp];
                    n = T.smooth ? ((1 - th.alpha - th.beta) * ld3(nn) + th.alpha * ld3(nn + 3)) + th.beta * ld3(nn + 6)
                                 : ld3(nn);
                    mat = S.prim_material[gp];
                }
                w.mark(46);     // closest hit: normal and material of the row that was hit
            }
        } else if (active) {
            w.analytic();
            switch (ob.kind) {
            case RTAMD_SPHERE: {
                V3 temp;
                hit = sphere_t(p, r, temp, t);
                if (hit) n = ((temp + t * r.d) / p[3]) * p[4];
            } break;
            case RTAMD_PLANE:
                hit = plane_t(ld3(p), ld3(p + 3), r, t);
                n = ld3(p + 3);
                break;
            case RTAMD_CHECKER_PLANE:
                hit = plane_t(ld3(p), ld3(p + 3), r, t);
                n = ld3(p + 3);
                if (hit) {                                        // geometry.py:353-360
                    const V3 rel = (r.o + t * r.d) - ld3(p);
                    const double x = round(dot(rel, ld3(p + 6)) / p[12]);
                    const double y = round(dot(rel, ld3(p + 9)) / p[12]);
                    if (fmod(x + y, 2.0) != 0) mat = ob.alt_material;
                }
                break;
            case RTAMD_RECTANGLE:
                hit = rectangle_t(p, r, t);
                n = ld3(p + 3);
                break;
            case RTAMD_BOX:
                hit = box_hit(p, r, t, n);
                break;
            case RTAMD_TRIANGLE: {
                const double g[9] = {p[0], p[1], p[2], p[3] - p[0], p[4] - p[1], p[5] - p[2],
                                     p[6] - p[0], p[7] - p[1], p[8] - p[2]};
                double a, b;
                hit = tri_solve(g, r, a, b, t);
                if (hit) n = ob.smooth ? ((1 - a - b) * ld3(p + 12) + a * ld3(p + 15)) + b * ld3(p + 18) : ld3(p + 9);
            } break;
            default: break;
            }
        }
        if (active && hit && t < tmin) {
            tmin = t;
            any = true;
            h.t = t;
            h.n = ob.has_transform ? normal_to_world(S, ob, n) : n;
            h.material = mat;
            h.object = i;
            h.prim = prim;
        }
    }
    if (any) h.p = ray.o + tmin * ray.d;
    return any;
}

// light.in_shadow for every lane with `need`: any object that reports a hit [closer than
// `limit`] (light.py:31-37 / :124-130 with a distance, :60-65 / :78-83 / :170-175 without).
// Shadow test of one analytic object (the shadow_hit methods of geometry.py).
__device__ __forceinline__ bool analytic_shadow(const rtamd_object& ob, const Ray& r, double& t) {
    const double* p = ob.p;
    switch (ob.kind) {
    case RTAMD_SPHERE: { V3 temp; return sphere_t(p, r, temp, t); }
    case RTAMD_PLANE:
    case RTAMD_CHECKER_PLANE: return plane_t(ld3(p), ld3(p + 3), r, t);
    case RTAMD_RECTANGLE: return rectangle_t(p, r, t);
    case RTAMD_BOX: return box_shadow(p, r, t);
    case RTAMD_TRIANGLE: {
        const double g[9] = {p[0], p[1], p[2], p[3] - p[0], p[4] - p[1], p[5] - p[2],
                             p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        double a, b;
        return tri_solve(g, r, a, b, t);
    }
    default: return false;
    }
}

template <int COUNT, bool TREES, bool ONE_TREE = false>
__device__ __forceinline__ bool occluded_wave(const rtamd_scene_desc& S, const KdStack& stk, bool need, const Ray& ray,
                              bool use_limit, double limit, Work<COUNT>& w) {
    if (need) w.shadow++;
    bool occ = false;
    if (ONE_TREE) {
        // Exactly one object of the scene is a kd-tree (and the launch does not count the reference's
        // work, whose object order matters): the analytic objects are asked first and the
        // tree last - "any object" does not depend on the order - so that nothing of the
        // caller's ray has to outlive the tree's query.
        int tree_object = 0;
        for (int i = 0; i < S.n_objects; ++i) {
            const rtamd_object& ob = S.objects[i];
            if (ob.kind == RTAMD_KDTREE) { tree_object = i; continue; }
            if (!(need && !occ && ob.cast_shadow)) continue;
            const Ray r = ob.has_transform ? to_object_space(S, ob, ray) : ray;
            double t = 0;
            w.analytic();
            if (analytic_shadow(ob, r, t) && (!use_limit || t < limit)) occ = true;
        }
        const rtamd_object& ob = S.objects[tree_object];
        const bool q = need && !occ;
        if (__ballot(q)) {
            const Ray r = ob.has_transform ? to_object_space(S, ob, ray) : ray;
            TriHit th;
            bool f;
            kd_query_wave<COUNT, true>(S, S.trees[ob.tree], q, r, use_limit ? limit : T_INF * 4.0, false, stk, w, f, th);
            if (q && f && (!use_limit || th.t < limit)) occ = true;
        }
        return occ;
    }
    if (TREES && COUNT != 1) {
        // "any object" does not depend on the order: the cheap analytic objects first, so that
        // a ray they occlude never walks a tree (a counting launch keeps the list order, which
        // decides how much work the reference does)
        for (int i = 0; i < S.n_objects; ++i) {
            const rtamd_object& ob = S.objects[i];
            if (ob.kind == RTAMD_KDTREE || !(need && !occ && ob.cast_shadow)) continue;
            const Ray r = ob.has_transform ? to_object_space(S, ob, ray) : ray;
            double t = 0;
            w.analytic();
            if (analytic_shadow(ob, r, t) && (!use_limit || t < limit)) occ = true;
        }
    }
    for (int i = 0; i < S.n_objects; ++i) {
        const rtamd_object& ob = S.objects[i];
        if (TREES && COUNT != 1 && ob.kind != RTAMD_KDTREE) continue;
        const bool q = need && !occ;
        const Ray r = ob.has_transform ? to_object_space(S, ob, ray) : ray;
        double t = 0;
        bool hit = false;
        if (TREES && ob.kind == RTAMD_KDTREE) {
            if (__ballot(q)) {
                TriHit th;
                bool f;
                kd_query_wave<COUNT, true>(S, S.trees[ob.tree], q, r, use_limit ? limit : T_INF * 4.0, false, stk, w, f, th);
                hit = q && f;
                t = th.t;
            }
        } else if (q && ob.cast_shadow) {
            w.analytic();
            hit = analytic_shadow(ob, r, t);
        }
        if (q && hit && (!use_limit || t < limit)) occ = true;
    }
    return occ;
}

// ---------------------------------------------------------------- shading
// Local frame used by AmbientOccluder.L and EnvironmentLight.get_direction
// (light.py:86-89, :153-156).
__device__ __forceinline__ V3 hemisphere_about(V3 w, V3 sp) {
    V3 v = cross(w, v3(0.0072, 1.0, 0.0034));
    v = v / norm(v);
    const V3 u = cross(v, w);
    return (u * sp.x + v * sp.y) + w * sp.z;
}

// pow() inlined into a kernel with a bounce loop has its polynomial constants hoisted in front of
// the loop and carried - spilled and reloaded one by one - through it.  The kernels for scenes
// without a kd-tree therefore call it as a function (mirror hall 4K frame: 125 -> 106 ms,
// foursphere 0.18 -> 0.14 ms); in the kd kernels the call's own register conventions cost more
// than they save (dragon frame 11.46 -> 11.73 ms), so those inline it.
#ifdef RT_POW_ALT
// checking builds only: another evaluation of the same power (a few ulps off), to tell differences
// against the CPU checker that hang on the last bit of pow from any others
__device__ __attribute__((noinline)) double pow_as_call(double a, double b) { return a > 0.0 ? exp2(b * log2(a)) : pow(a, b); }
template <bool CALL> __device__ __forceinline__ double rt_pow(double a, double b) { return pow_as_call(a, b); }
#else
__device__ __attribute__((noinline)) double pow_as_call(double a, double b) { return pow(a, b); }
template <bool CALL> __device__ __forceinline__ double rt_pow(double a, double b) {
    return CALL ? pow_as_call(a, b) : pow(a, b);
}
#endif

// GlossySpecular.f (material.py:88-97) with ks*cs folded into `spec`.
template <bool POW_CALL>
__device__ __forceinline__ V3 phong_lobe(V3 spec, double exp_, V3 n, V3 wo, V3 wi) {
    const double ndotwi = dot(n, wi);
    const V3 r = (-wi) + (2.0 * n) * ndotwi;
    const double rdotwo = dot(r, wo);
    if (rdotwo > 0.0) return spec * rt_pow<POW_CALL>(rdotwo, exp_);
    return v3(0, 0, 0);
}

// ShadeRecord.local_hit_point of a hit: o + t*d in the object's own space (geometry.py:210,
// :323, :456; through an Instance the child's record is kept, :103-108), relative to the
// centre for a sphere (:271).
__device__ __forceinline__ V3 local_hit_point(const rtamd_scene_desc& S, const rtamd_object& ob, const Ray& world_ray, double t) {
    const Ray r = ob.has_transform ? to_object_space(S, ob, world_ray) : world_ray;
    V3 p = r.o + t * r.d;
    if (ob.kind == RTAMD_SPHERE) p = p - ld3(ob.p);
    return p;
}

// ImageTexture.get_color through SphericalMap.get_texel_coordinates (texture.py:33-68).
// Python raises where acos leaves its domain or the texel lies outside the image; here the
// colour is NaN.
__device__ __forceinline__ V3 texture_color(const rtamd_scene_desc& S, int tex, V3 local) {
    const rtamd_texture& T = S.textures[tex];
    const double theta = acos(local.y);
    double phi = atan2(local.x, local.z);
    if (phi < 0.) phi += PI * 2;
    const double u = phi / PI / 2;
    const double v = 1 - theta / PI;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (!(theta == theta)) return v3(nan, nan, nan);
    const int column = (int)((T.hres - 1) * u);
    const int row = (int)((T.vres - 1) * v);
    if (column < 0 || column >= T.width || row < 0 || row >= T.height) return v3(nan, nan, nan);
    const uint8_t* px = S.texels + T.offset + 3 * ((int64_t)row * T.width + column);
    return v3(px[0] / 256., px[1] / 256., px[2] / 256.);
}

// The direct-lighting sum of Matte/Phong .shade and .area_light_shade (material.py:251-281,
// :346-377) for every lane with `lit`; `area` selects the G/pdf weighting.  Lights are
// visited in list order by the whole wave so that the shadow queries stay wave-wide.
// EXT: `tex` is the texture colour of an SV_Matte hit (material.py:311-326), which scales
// ka and kd and adds the viewer-side test of :319.
template <int COUNT, bool EXT, bool TREES, bool ONE_TREE = false, bool FUSED_OK = ONE_TREE, bool POW_CALL = !TREES>
__device__ __forceinline__ V3 direct_light_wave(const rtamd_scene_desc& S, const KdStack& stk, bool lit, const rtamd_material& m,
                                const Hit& h, V3 wo, bool area, V3 tex, Draw& dr, Work<COUNT>& w) {
    if (ONE_TREE && FUSED_OK && COUNT != 1 && !EXT && (stk.fused_always || (S.ambient.kind == RTAMD_AMBIENT_OCCLUDER && S.lights[0].cast_shadow &&
                                                               S.leaf_aabb != nullptr))) {
        // One kd-tree, an ambient occluder and one shadow-casting light: the occlusion ray and
        // the light's shadow ray of a hit are both known before either is traced (no draw
        // happens inside a query), and which is traced first changes neither answer.  When the
        // tree has a shadow tree and any hit decides both, the wave traces them in one query.
        int tree_object = 0;
        for (int i = 0; i < S.n_objects; ++i) if (S.objects[i].kind == RTAMD_KDTREE) tree_object = i;
        const rtamd_object& tob = S.objects[tree_object];
        const rtamd_tree& T = S.trees[tob.tree];
        if (stk.fused_always || T.shadow_root > 0) {
            const rtamd_light& lt = S.lights[0];
            const Draw drawn = dr;          // if the wave cannot take this path the draws are made again below
            // ambient_color.L(sr), AmbientOccluder (light.py:85-100)
            Ray r1;
            r1.o = h.p;
            r1.d = v3(0, 0, 1);
            if (lit) r1.d = hemisphere_about(h.n, draw_hemisphere(S, S.ambient.sampler, dr));
            w.mark(60);     // material record, occlusion ray
            // the light's direction, term and shadow ray, as in the loop below
            V3 wi = v3(0, 0, 1), sample_point = v3(0, 0, 0);
            if (lit) {
                switch (lt.kind) {
                case RTAMD_POINT: { const V3 v = ld3(lt.position) - h.p; wi = v / norm(v); } break;
                case RTAMD_DIRECTIONAL: wi = ld3(lt.position); break;
                case RTAMD_AREA: {
                    double sx, sy;
                    draw_square(S, lt.sampler, dr, sx, sy);
                    sample_point = (ld3(lt.position) + sx * ld3(lt.edge_u)) + sy * ld3(lt.edge_v);
                    const V3 v = sample_point - h.p;
                    wi = v / norm(v);
                } break;
                default: wi = hemisphere_about(h.n, draw_hemisphere(S, lt.sampler, dr)); break;
                }
            }
            w.mark(61);     // light sample and direction
            const double ndotwi = dot(h.n, wi);
            const bool facing = lit && ndotwi > 0;
            const bool phong = m.kind != RTAMD_MATTE;
            const bool gate = (phong || area) ? (m.receives_shadow != 0) : true;
            const bool wanted = facing && gate;
            V3 term = v3(0, 0, 0);
            if (wanted) {
                V3 f = ld3(m.diffuse);
                if (phong) f = f + phong_lobe<POW_CALL>(ld3(m.specular), m.exp, h.n, wo, wi);
                V3 Ll = ld3(lt.radiance);
                if (lt.kind == RTAMD_POINT && lt.attenuation != 0)
                    Ll = Ll * rt_pow<POW_CALL>(norm(h.p - ld3(lt.position)) / lt.radius, -lt.attenuation);
                if (lt.kind == RTAMD_AREA && !(dot(ld3(lt.normal), wi) < 0.)) Ll = v3(0, 0, 0);
                term = (f * Ll) * ndotwi;
                if (area) {
                    double G = 1.0, pdf = 1.0;
                    if (lt.kind == RTAMD_AREA) {
                        const double ndotd = -dot(ld3(lt.normal), wi);
                        const V3 dv = sample_point - h.p;
                        const double d2 = (((0 + dv.x * dv.x) + dv.y * dv.y) + dv.z * dv.z) / (lt.radius * lt.radius);
                        G = ndotd / pymax(1., d2);
                        pdf = lt.pdf;
                    }
                    term = (term * G) / pdf;
                }
            }
            w.mark(62);     // the light's term
            Ray r2;
            r2.o = h.p;
            r2.d = wi;
            double limit = 0;
            if (lt.kind == RTAMD_POINT) limit = norm(r2.o - ld3(lt.position));
            else if (lt.kind == RTAMD_AREA) limit = dot(sample_point - r2.o, r2.d);
            const bool bounded = lt.kind == RTAMD_POINT || lt.kind == RTAMD_AREA;
            // any hit decides a ray of the light whose limit lies beyond the tree's box (always, for
            // the occlusion ray); a wave with a light ray that ends inside the box takes the
            // general path below
            const Ray q1 = tob.has_transform ? to_object_space(S, tob, r1) : r1;
            const Ray q2 = tob.has_transform ? to_object_space(S, tob, r2) : r2;
            double near1 = 0, exit1 = 0, near2 = 0, exit2 = 0;
            const bool casts = T.shadow_casters != RTAMD_CASTERS_NONE;
            const bool in1 = casts && lit && root_box(T, q1, near1, exit1);
            // (a light ray that reaches the light before it reaches the box - a hit point beyond the emitter,
            // whose ray would enter the box on its extension - has nothing of this tree in front of the light)
            const bool in2 = casts && facing && root_box(T, q2, near2, exit2) && (!bounded || near2 < limit);
            const bool free2 = !bounded || (exit2 * (1.0 + 1e-9) + 1e-9 < limit);
            // (fused_always: the emitter lies outside the tree's box by a margin along some axis, so a light
            // ray that is inside the box before its limit has left it again by then, whatever rounding does
            // to the two distances)
            w.mark(63);     // both rays in the tree's space, their way through its box
            if (stk.fused_always || !__ballot(in2 && !free2)) {
                w.mark(34);
                // light.in_shadow for both rays: the analytic objects first, the tree last
                if (lit) w.shadow++;
                if (facing) w.shadow++;
                bool occ1 = false, occ2 = false;
                for (int i = 0; i < S.n_objects; ++i) {
                    const rtamd_object& ob = S.objects[i];
                    if (ob.kind == RTAMD_KDTREE || !ob.cast_shadow) continue;
                    double t = 0;
                    if (lit && !occ1) {
                        const Ray r = ob.has_transform ? to_object_space(S, ob, r1) : r1;
                        w.analytic();
                        if (analytic_shadow(ob, r, t)) occ1 = true;
                    }
                    if (facing && !occ2) {
                        const Ray r = ob.has_transform ? to_object_space(S, ob, r2) : r2;
                        w.analytic();
                        if (analytic_shadow(ob, r, t) && (!bounded || t < limit)) occ2 = true;
                    }
                }
                const bool walk1 = in1 && !occ1, walk2 = in2 && !occ2;
                bool f1 = false, f2 = false;
                TriHit th;
                th.t = 0;
                if (__ballot(walk1 || walk2))
                    shadow_tree_wave<COUNT, true>(S, T, walk1, q1, near1, exit1, true, stk, w, f1, th, walk2, &q2, near2, exit2, &f2);
                occ1 = occ1 || (walk1 && f1);
                occ2 = occ2 || (walk2 && f2);
                w.mark(18);
                const V3 amb = occ1 ? ld3(S.ambient.occluded) : ld3(S.ambient.radiance);
                V3 L = ld3(m.ambient) * amb;
                if (wanted && !occ2) L = L + term;
                return L;
            }
            dr = drawn;
        }
    }
    // ambient_color.L(sr): AmbientLight (light.py:14-15) or AmbientOccluder (light.py:85-100)
    V3 amb = ld3(S.ambient.radiance);
    if (S.ambient.kind == RTAMD_AMBIENT_OCCLUDER) {
        Ray sr;
        sr.o = h.p;
        sr.d = v3(0, 0, 1);
        if (lit) sr.d = hemisphere_about(h.n, draw_hemisphere(S, S.ambient.sampler, dr));
        w.mark(33);     // from the closest hit to the ambient-occlusion ray
        if (occluded_wave<COUNT, TREES, ONE_TREE>(S, stk, lit, sr, false, 0.0, w)) amb = ld3(S.ambient.occluded);
        w.mark(18);     // ambient occlusion query, kd query included
    }
    const bool textured = EXT && m.kind == RTAMD_SV_MATTE;
    V3 L = (textured ? ld3(m.ambient) * tex : ld3(m.ambient)) * amb;
    const bool phong = m.kind != RTAMD_MATTE && !textured;
    // Matte.shade alone ignores receives_shadow (material.py:263 vs :279, :358, :374)
    const bool gate = (phong || area) ? (m.receives_shadow != 0) : true;
    // (the ONE_TREE instantiation is launched for scenes with a single light: the loop is then
    // one straight pass and nothing is kept for a next light through the shadow query)
    for (int li = 0; li < (ONE_TREE ? 1 : S.n_lights); ++li) {
        const rtamd_light& lt = S.lights[li];
        V3 wi = v3(0, 0, 1), sample_point = v3(0, 0, 0);
        if (lit) {
            switch (lt.kind) {
            case RTAMD_POINT: {                                  // light.py:27-29
                const V3 v = ld3(lt.position) - h.p;
                wi = v / norm(v);
            } break;
            case RTAMD_DIRECTIONAL:                              // light.py:54-55
                wi = ld3(lt.position);
                break;
            case RTAMD_AREA: {                                   // light.py:116-122, geometry.py:378-380
                double sx, sy;
                draw_square(S, lt.sampler, dr, sx, sy);
                sample_point = (ld3(lt.position) + sx * ld3(lt.edge_u)) + sy * ld3(lt.edge_v);
                const V3 v = sample_point - h.p;
                wi = v / norm(v);
            } break;
            default:                                             // ENVIRONMENT, light.py:152-165
                wi = hemisphere_about(h.n, draw_hemisphere(S, lt.sampler, dr));
                break;
            }
        }
        const double ndotwi = dot(h.n, wi);
        const bool facing = lit && ndotwi > 0 && !(textured && !(dot(h.n, wo) > 0.));
        // With kd-trees in the scene the light's contribution is worked out before its shadow
        // query, so that only the finished term - not the vectors it is made of - waits in
        // registers through the query (1.4 % on the dragon frame); scenes of analytic objects
        // alone work it out afterwards, for the lanes that turn out lit only (7 % on the
        // mirror hall, whose Phong lobes are costly).
        const bool wanted = facing && gate;
        V3 term = v3(0, 0, 0);
        auto light_term = [&]() {
            V3 f = ld3(m.diffuse);
            if (textured) f = (f * tex) / PI;                                                    // material.py:27-28
            if (phong) f = f + phong_lobe<POW_CALL>(ld3(m.specular), m.exp, h.n, wo, wi);
            V3 Ll = ld3(lt.radiance);
            if (lt.kind == RTAMD_POINT && lt.attenuation != 0)                                    // light.py:39-44
                Ll = Ll * rt_pow<POW_CALL>(norm(h.p - ld3(lt.position)) / lt.radius, -lt.attenuation);
            if (lt.kind == RTAMD_AREA && !(dot(ld3(lt.normal), wi) < 0.)) Ll = v3(0, 0, 0);        // light.py:110-114
            term = (f * Ll) * ndotwi;
            if (area) {
                double G = 1.0, pdf = 1.0;                                                        // light.py:149-150, :177-178
                if (lt.kind == RTAMD_AREA) {                                                      // light.py:132-140
                    const double ndotd = -dot(ld3(lt.normal), wi);
                    const V3 dv = sample_point - h.p;
                    const double d2 = (((0 + dv.x * dv.x) + dv.y * dv.y) + dv.z * dv.z) / (lt.radius * lt.radius);
                    G = ndotd / pymax(1., d2);
                    pdf = lt.pdf;
                }
                term = (term * G) / pdf;
            }
        };
        if (TREES && wanted) light_term();
        bool in_shadow = false;
        if (lt.cast_shadow) {
            Ray sr;
            sr.o = h.p;
            sr.d = wi;
            double limit = 0;
            if (lt.kind == RTAMD_POINT) limit = norm(sr.o - ld3(lt.position));                    // light.py:32
            else if (lt.kind == RTAMD_AREA) limit = dot(sample_point - sr.o, sr.d);               // light.py:125
            const bool bounded = lt.kind == RTAMD_POINT || lt.kind == RTAMD_AREA;
            w.mark(34);     // light sample, its term and shadow ray set-up
            if (__ballot(facing)) in_shadow = occluded_wave<COUNT, TREES, ONE_TREE>(S, stk, facing, sr, bounded, limit, w);
            w.mark(19);     // light shadow query, kd query included
        }
        if (wanted && !in_shadow) {
            if (!TREES) light_term();
            L = L + term;
        }
    }
    return L;
}

struct Primary { int32_t object, prim; double t; };

// Side of the surface a ray arrives on, relative index and the discriminant of Snell's law:
// PerfectTransmitter (material.py:135-159) and FresnelReflector/Transmitter (:170-223).
__device__ __forceinline__ bool refraction(const rtamd_material& m, V3 n, V3 d, V3& nf, double& eta,
                                           double& cos_i, double& temp) {
    double c = -dot(n, d);
    nf = n;
    eta = m.eta_front;
    if (c < 0.) { c = -c; nf = -n; eta = m.eta_back; }
    cos_i = c;
    temp = 1. - (1. - c * c) / (eta * eta);
    return temp < 0.;
}

__device__ __forceinline__ V3 powv(V3 b, double e) { return v3(rt_pow<false>(b.x, e), rt_pow<false>(b.y, e), rt_pow<false>(b.z, e)); }

// tracer.trace_ray (tracer.py:26-35, :43-58, :65-71, :79-88) for the 64 samples of a wave.
// The single-child recursion of Reflective / GlossyReflective / Matte.path_shade
// (material.py:385-411, :283-289) is unrolled into a loop: radiance = sum over bounces of
// (product of the child weights so far) * local term.  Lanes whose path has ended stay in
// the loop as helpers for the wave-wide kd leaf tests until every lane is done.
//
// EXT: the scene holds textured (SV_Matte) or two-child materials.  Transparent and
// Dielectric spawn two children per hit
// (material.py:428-436, :473-474) make the recursion a tree.  Each lane walks its tree
// depth first in the reference's order - reflected subtree, then the refracted one - with
// the waiting siblings in a per-lane list (one per level, RTAMD_MAX_BRANCH_LEVELS), so the
// in-trace draw counter advances exactly as in the recursive formulation.
// NO_BOUNCE: every material of the scene is Matte, Phong or Emissive under the plain tracer, so no
// hit spawns a child ray and the body is left after one pass, as under AREA_ONLY.
// AREA_ONLY: the scene runs under the AreaLightTracer (tracer.py:65-71), which never bounces:
// the compiler then drops the bounce loop and the ray and weight it would carry through the
// shadow queries.  The instantiation is launched for scenes with exactly one kd-tree object,
// which the shadow tests then ask last (occluded_wave<..., ONE_TREE>), and exactly one light.
// FEED: where a lane whose path has ended hands its sample in and takes another one, so that the
// wave's lanes stay filled while paths of different lengths end at different bounces (the kernels
// that own a list of samples per wave pass one; NoFeed: a lane traces one sample).
// How a path's radiance is summed.  The kernels carry the product of the bounce weights forward: L += W * local at
// every hit, W *= (fr * n.wi) / pdf at every bounce (a waiting sibling of a two-child material keeps its own W).  The
// reference's recursion sums from the last hit back to the first, L = local + ((fr * L_child) * n.wi) [/ pdf]
// (material.py:283-289, :385-394, :402-411, :420-437, :448-483): the same factors in another order, which agree to
// rounding except where a glossy lobe is denormal.  Checking builds (-DRT_REF_ORDER; not for refilled lanes) keep a
// frame per level of the recursion - what the level has added itself, and how its value enters its parent's sum - and
// fold them in the reference's order, to tell such samples from real differences.
#ifdef RT_REF_ORDER
#define RT_REF_ORDER_ON true
#else
#define RT_REF_ORDER_ON false
#endif
// how a child's value V enters its parent's sum (a: colour, s1, s2: scalars, t: the child's own hit distance)
enum { FOLD_MIRROR = 0,         // + (a * V) * s1                   Reflective.shade, material.py:393
       FOLD_LOBE,               // + ((a * V) * s1) / s2            GlossyReflective.shade :410, Matte.path_shade :289
       FOLD_PLAIN,              // + V                              Transparent.shade, total internal reflection :430
       FOLD_GLASS_MIRROR,       // + (V * s1) * a                   Transparent.shade :434
       FOLD_SCALED,             // + (s1 * V) * s2                  Transparent.shade :435
       FOLD_FILTERED,           // + V * a ** t                     Dielectric.shade :460-465
       FOLD_FRESNEL };          // + ((s1 * V) * a ** t) * s2       Dielectric.shade :477-482
struct Fold { int kind; V3 a; double s1, s2; };
__device__ __forceinline__ Fold fold_of(int kind, V3 a, double s1, double s2) { Fold f; f.kind = kind; f.a = a; f.s1 = s1; f.s2 = s2; return f; }
template <bool REFERENCE_ORDER> struct PathSum {
    __device__ __forceinline__ void add(V3& L, V3 W, V3 x) { L = L + W * x; }
    __device__ __forceinline__ void child(const Fold&) {}
    __device__ __forceinline__ void resume(int, const Fold&) {}
    __device__ __forceinline__ void hit(double) {}
    __device__ __forceinline__ void finish(V3&) {}
};
template <> struct PathSum<true> {
    static constexpr int MAX_LEVELS = 40;
    V3 here[MAX_LEVELS];            // what each open level has added so far, folded children included
    Fold how[MAX_LEVELS];           // how the level's value will enter its parent's
    double t[MAX_LEVELS];           // the level's own hit distance (0 for a miss)
    int top = 0;
    __device__ PathSum() { here[0] = v3(0, 0, 0); t[0] = 0; }
    __device__ __forceinline__ void add(V3&, V3, V3 x) { here[top] = here[top] + x; }
    __device__ __forceinline__ void hit(double tt) { t[top] = tt; }
    __device__ __forceinline__ void fold() {
        const V3 v = here[top];
        const Fold& f = how[top];
        const V3 filter = (f.kind == FOLD_FILTERED || f.kind == FOLD_FRESNEL)
                              ? v3(pow(f.a.x, t[top]), pow(f.a.y, t[top]), pow(f.a.z, t[top])) : v3(1, 1, 1);
        V3 up;
        switch (f.kind) {
        case FOLD_MIRROR: up = (f.a * v) * f.s1; break;
        case FOLD_LOBE: up = ((f.a * v) * f.s1) / f.s2; break;
        case FOLD_PLAIN: up = v; break;
        case FOLD_GLASS_MIRROR: up = (v * f.s1) * f.a; break;
        case FOLD_SCALED: up = (f.s1 * v) * f.s2; break;
        case FOLD_FILTERED: up = v * filter; break;
        default: up = ((f.s1 * v) * filter) * f.s2; break;
        }
        top--;
        here[top] = here[top] + up;
    }
    // the level the lane stands on spawns a child and the lane follows it
    __device__ __forceinline__ void child(const Fold& f) {
        if (top + 1 < MAX_LEVELS) { top++; here[top] = v3(0, 0, 0); t[top] = 0; how[top] = f; }
    }
    // the lane takes up a sibling that waited at `level` (0 = the primary ray): everything deeper is complete
    __device__ __forceinline__ void resume(int level, const Fold& f) {
        while (top >= level && top > 0) fold();
        child(f);
    }
    __device__ __forceinline__ void finish(V3& L) {
        while (top > 0) fold();
        L = here[0];
    }
};

// A child ray of a Transparent / Dielectric hit that waits while its sibling's subtree is
// traced: ray, product of the weights down to it, and Dielectric's colour filter, which is
// raised to the child's own hit distance once that is known (material.py:463-482).
struct PendingRay {
    V3 o, d, W, filter;
    int32_t depth, has_filter;
#ifdef RT_REF_ORDER
    Fold how;           // checking builds: how the child's value enters its parent's sum (PathSum)
#endif
};

struct NoFeed {
    static constexpr bool on = false;
    __device__ __forceinline__ void retire(V3) {}
    __device__ __forceinline__ bool next(bool, Ray&, Draw&) { return false; }
    __device__ __forceinline__ bool more() const { return false; }
};

// PATH_ONLY: the scene runs under the PathTracer (tracer.py:74-88): a hit draws one direction and goes on, no
// direct lighting - the kernel then holds the closest-hit query and nothing else.
// ONE_TREE: the scene has exactly one kd-tree object and one light (AREA_ONLY implies it): shadow tests ask the
// tree last and the loop over the lights is one straight pass.
template <int COUNT, bool EXT, bool TREES, bool AREA_ONLY = false, class FEED = NoFeed, bool NO_BOUNCE = false,
          bool PATH_ONLY = false, bool ONE_TREE = AREA_ONLY>
__device__ __forceinline__ V3 trace_wave(const rtamd_scene_desc& S, const KdStack& stk, bool alive, Ray ray, int depth, Draw& dr,
                         Work<COUNT>& w, Primary& first, FEED& feed) {
    static_assert(!FEED::on || (!EXT && !AREA_ONLY), "sample feeds serve the plain bounce loop");
    static_assert(!PATH_ONLY || (!EXT && !AREA_ONLY && !NO_BOUNCE), "the path tracer's build is a plain bounce loop");
    // pow() as a call wherever the body is a loop or holds the general shadow paths (see rt_pow); the area-light
    // kernels measured faster with it inline
    constexpr bool POW_CALL = !TREES || !AREA_ONLY;
    V3 L = v3(0, 0, 0), W = v3(1, 1, 1);
    PathSum<RT_REF_ORDER_ON && !FEED::on> sum;
#define RT_ADD(X) sum.add(L, W, (X))
    const int depth0 = depth;
    bool busy = alive;                      // FEED: the lane holds a sample that is not handed in yet
    const int tracer_kind = AREA_ONLY ? (int)RTAMD_TRACER_AREA : PATH_ONLY ? (int)RTAMD_TRACER_PATH : S.tracer_kind;
    const bool shade_kind = tracer_kind == RTAMD_TRACER_SHADE || tracer_kind == RTAMD_TRACER_WHITTED;
    const V3 background = shade_kind ? ld3(S.background) : v3(0, 0, 0);
    const bool area = tracer_kind == RTAMD_TRACER_AREA;
    first.object = -1; first.prim = -1; first.t = 0;
    bool primary = true;
    PendingRay pending[EXT ? RTAMD_MAX_BRANCH_LEVELS : 1];
    int n_pending = 0;
    int lane_depth = depth;                 // EXT: lanes sit at different levels of their trees
    V3 filter = v3(1, 1, 1);
    bool has_filter = false;
    for (;; ++depth, ++lane_depth) {
        if (EXT) {
            if (!alive && n_pending > 0) {
                const PendingRay& q = pending[--n_pending];
                ray.o = q.o; ray.d = q.d; W = q.W; filter = q.filter;
                lane_depth = q.depth; has_filter = q.has_filter != 0;
                alive = true;
#ifdef RT_REF_ORDER
                sum.resume(q.depth - depth0, q.how);
#endif
            }
            if (alive && !area && lane_depth > S.depth_limit) { RT_ADD(background); alive = false; }
            if (!__ballot(alive || n_pending > 0)) break;
        } else if (FEED::on) {
            if (alive && !area && lane_depth > S.depth_limit) { RT_ADD(background); alive = false; }
            if (!alive && busy) { feed.retire(L); busy = false; }
            if (feed.next(!alive, ray, dr)) {
                alive = true; busy = true;
                L = v3(0, 0, 0); W = v3(1, 1, 1);
                lane_depth = depth0;
            }
            // (a sample of a pixel outside the image is nothing to trace: such a lane asks again in the
            // next pass, and a wave left without a live lane that way goes round once more)
            if (!__ballot(alive) && !feed.more()) break;
        } else {
            if (alive && !area && depth > S.depth_limit) { RT_ADD(background); alive = false; }
            if (!__ballot(alive)) break;
        }
        Hit h;
        h.p = v3(0, 0, 0); h.n = v3(0, 0, 1); h.material = 0;
        w.mark(32);     // ray generation / previous bounce
        // (with a feed, primary rays taken in later join lanes whose rays have bounced: only the first pass is coherent)
        const bool hit = closest_hit_wave<COUNT, TREES>(S, stk, alive, primary, ray, w, h);
        w.mark(31);     // closest hit, kd query included
        if (alive && !hit) { RT_ADD(background); alive = false; }
        if (alive && primary) { first.object = h.object; first.prim = h.prim; first.t = h.t; }
        primary = false;
        if (EXT && alive) sum.hit(h.t);
        if (EXT && alive && has_filter) { W = W * powv(filter, h.t); has_filter = false; }
        const rtamd_material& m = S.materials[alive ? h.material : 0];
        const V3 wo = -ray.d;

        if (alive && m.kind == RTAMD_EMISSIVE) {
            V3 e;
            if (shade_kind) e = v3(1., 1., 1.);                                                    // material.py:495-496
            else if (area) e = (-dot(h.n, ray.d) > 0) ? ld3(m.emit) : v3(0, 0, 0);                  // material.py:498-502
            else e = (dot(h.n, ray.d) < 0) ? ld3(m.emit) : v3(0, 0, 0);                             // material.py:504-508
            RT_ADD(e);
            alive = false;
        }

        if (tracer_kind == RTAMD_TRACER_PATH) {
            if (alive) {
                // Lambertian.sample_f (material.py:41-53) then Matte.path_shade (:283-289)
                V3 v = cross(v3(0.0034, 1, 0.0071), h.n);
                v = v / norm(v);
                const V3 u = cross(v, h.n);
                const V3 sp = draw_hemisphere(S, m.sampler, dr);
                V3 wi = (sp.x * u + sp.y * v) + sp.z * h.n;
                wi = wi / norm(wi);
                const double pdf = dot(h.n, wi) / PI;
                const double ndotwi = dot(h.n, wi);
                W = W * ((ld3(m.diffuse) * ndotwi) / pdf);
                sum.child(fold_of(FOLD_LOBE, ld3(m.diffuse), ndotwi, pdf));
                ray.o = h.p;
                ray.d = wi;
            }
            continue;
        }

        V3 tex = v3(1, 1, 1);
        if (EXT && alive && m.kind == RTAMD_SV_MATTE)
            tex = texture_color(S, m.texture, local_hit_point(S, S.objects[h.object], ray, h.t));
        const V3 Ld = direct_light_wave<COUNT, EXT, TREES, ONE_TREE, AREA_ONLY, POW_CALL>(S, stk, alive, m, h, wo, area, tex, dr, w);
        if (alive) RT_ADD(Ld);
        // the AreaLightTracer never bounces: leaving here makes the body straight-line code, so
        // nothing is hoisted in front of it and carried (or spilled) through the kd queries
        // (NO_BOUNCE: no material of the scene spawns a child ray - the same holds there)
        if (AREA_ONLY || NO_BOUNCE) break;
        if (area) { alive = false; continue; }
        if (!alive) continue;

        if (m.kind == RTAMD_REFLECTIVE) {
            // PerfectSpecular.sample_f (material.py:69-74), Reflective.shade (:385-394)
            const double ndotwo = dot(h.n, wo);
            const V3 wi = (-wo) + (2.0 * h.n) * ndotwo;
            const double pdf = dot(h.n, wi);
            const V3 fr = ld3(m.reflect) / pdf;
            W = W * (fr * dot(h.n, wi));
            sum.child(fold_of(FOLD_MIRROR, fr, dot(h.n, wi), 1.0));
            ray.o = h.p;
            ray.d = wi;
        } else if (m.kind == RTAMD_GLOSSY_REFLECTIVE) {
            // GlossySpecular.sample_f (material.py:102-121), GlossyReflective.shade (:402-411)
            const double ndotwo = dot(h.n, wo);
            const V3 wr = (-wo) + (2.0 * h.n) * ndotwo;
            V3 u = cross(v3(0.00424, 1., 0.00767), wr);
            u = u / norm(u);
            const V3 v = cross(u, wr);
            const V3 sp = draw_hemisphere(S, m.sampler, dr);
            V3 wi = (sp.x * u + sp.y * v) + sp.z * wr;
            if (dot(h.n, wi) < 0.0) wi = ((-sp.x) * u - sp.y * v) + sp.z * wr;
            const double lobe = rt_pow<POW_CALL>(dot(wr, wi), m.reflect_exp);
            double pdf = lobe * dot(h.n, wi);
            V3 fr;
            if (pdf == 0) { fr = v3(0, 0, 0); pdf = 1; }
            else fr = ld3(m.reflect) * lobe;
            W = W * ((fr * dot(h.n, wi)) / pdf);
            sum.child(fold_of(FOLD_LOBE, fr, dot(h.n, wi), pdf));
            ray.o = h.p;
            ray.d = wi;
        } else if (EXT && m.kind == RTAMD_TRANSPARENT) {
            // PerfectSpecular.sample_f (material.py:69-74), PerfectTransmitter (:135-159),
            // Transparent.shade (:420-437)
            const double ndotwo = dot(h.n, wo);
            const V3 wi = (-wo) + (2.0 * h.n) * ndotwo;
            const double pdf = dot(h.n, wi);
            const V3 fr = ld3(m.reflect) / pdf;
            V3 nf;
            double eta, cos_i, temp;
            if (!refraction(m, h.n, ray.d, nf, eta, cos_i, temp)) {
                const double cos_t = sqrt(temp);
                const V3 wt = ((-wo) / eta) - (cos_t - cos_i / eta) * nf;
                const double ft = m.kt / (eta * eta) / fabs(dot(h.n, wt));
                PendingRay& q = pending[n_pending++];
                q.o = h.p; q.d = wt; q.W = W * (ft * fabs(dot(h.n, wt)));
                q.filter = v3(1, 1, 1); q.has_filter = 0; q.depth = lane_depth + 1;
#ifdef RT_REF_ORDER
                q.how = fold_of(FOLD_SCALED, v3(0, 0, 0), ft, fabs(dot(h.n, wt)));
#endif
                W = W * (dot(h.n, wi) * fr);
                sum.child(fold_of(FOLD_GLASS_MIRROR, fr, dot(h.n, wi), 0.0));
            } else {
                sum.child(fold_of(FOLD_PLAIN, v3(0, 0, 0), 0.0, 0.0));
            }
            ray.o = h.p;
            ray.d = wi;
        } else if (EXT && m.kind == RTAMD_DIELECTRIC) {
            // FresnelReflector.fresnel / sample_f (material.py:206-229), FresnelTransmitter
            // (:170-197), Dielectric.shade (:448-483)
            V3 nf;
            double eta, cos_i, temp;
            const bool tir = refraction(m, h.n, ray.d, nf, eta, cos_i, temp);
            const double ndotwo = dot(h.n, wo);
            const V3 wi = (-wo) + (2.0 * h.n) * ndotwo;
            const double ndotwi = dot(h.n, wi);
            const bool inside = ndotwi < 0.;            // the reflected ray stays inside
            if (!tir) {
                const double cos_t = sqrt(temp);
                const double r_par = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
                const double r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
                const double kr = .5 * (r_par * r_par + r_perp * r_perp);
                const V3 wt = ((-wo) / eta) - (cos_t - cos_i / eta) * nf;
                const double ndotwt = dot(h.n, wt);
                const double ft = (1 - kr) / (eta * eta) / fabs(ndotwt);
                PendingRay& q = pending[n_pending++];
                q.o = h.p; q.d = wt; q.W = W * (ft * fabs(ndotwt));
                q.filter = inside ? ld3(m.filter_out) : ld3(m.filter_in);
                q.has_filter = 1; q.depth = lane_depth + 1;
#ifdef RT_REF_ORDER
                q.how = fold_of(FOLD_FRESNEL, q.filter, ft, fabs(ndotwt));
#endif
                W = W * ((kr / fabs(ndotwi)) * fabs(ndotwi));
                sum.child(fold_of(FOLD_FRESNEL, inside ? ld3(m.filter_in) : ld3(m.filter_out), kr / fabs(ndotwi), fabs(ndotwi)));
            } else {
                sum.child(fold_of(FOLD_FILTERED, inside ? ld3(m.filter_in) : ld3(m.filter_out), 0.0, 0.0));
            }
            filter = inside ? ld3(m.filter_in) : ld3(m.filter_out);
            has_filter = true;
            ray.o = h.p;
            ray.d = wi;
        } else {
            alive = false;
        }
    }
    sum.finish(L);
#undef RT_ADD
    return L;
}

// int(c*255) as camera.py:66 does it: truncation toward zero.  Python raises on NaN or
// infinity; those become 0 / the int32 limits here.
__device__ __forceinline__ int32_t quantise(double c) {
    const double s = c * 255;
    if (!(s == s)) return 0;
    if (s >= 2147483647.0) return 2147483647;
    if (s <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)s;
}

}  // namespace rt
