// Kernels and the C ABI of include/raytracer_amd.h.
//
// One lane traces one sample; a wave of the tile kernel holds the 64 pixels of one
// 8x8-pixel tile at one sample index, so primary rays of a wave stay close together.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "rt_device.h"

using namespace rt;

namespace {

enum { MODE_PIXELS = 0, MODE_RAYS = 1, MODE_TILES = 2 };
// One wave per workgroup: waves are independent (no LDS exchange, no barrier in the trace),
// and single-wave groups let the dispatcher refill a SIMD as soon as a long wave retires.
// RT_MIN_WAVES = 4 holds the kernel to 128 VGPRs (4 waves per SIMD); measured on the 1080p
// dragon frame: 2 waves/SIMD 575, 3 -> 636, 4 -> 757, 5 -> 656 (spills) Mrays/s.
#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_REFILL_MAX
// 64-sample groups of the work list per wave of refill_kernel: the largest divisor of the samples
// per pixel up to this.  glossyreflective 400x400x100 (the divisor in brackets): off 3.19 ms,
// 4 [4] 2.30, 8 [5] 1.93, 16 [10] 1.15, 32 [25] 1.44 ms; mirror hall 4K x 64: off 104.5 ms,
// 4 83.3, 8 82.4, 16 81.8, 32 82.1 ms.  Longer lists shorten the share of a wave's life in which
// its last paths run on with few lanes, but leave fewer waves to fill the machine with.
#define RT_REFILL_MAX 16
#endif
#ifndef RT_MIN_WAVES_ANALYTIC
// waves per SIMD the kernels for scenes without a kd-tree are held to (they need no traversal state)
#define RT_MIN_WAVES_ANALYTIC 4
#endif
#ifndef RT_GROUP_SHIFT_MAX
// Tile mode: a wave takes up to 2^RT_GROUP_SHIFT_MAX samples of each of its pixels (see the
// kernel).  1080p dragon frame, 16 samples per pixel: 8x8 pixels x 1 sample 8.61 ms,
// 4x4 x 4 -> 7.90 ms, 2x2 x 16 -> 7.69 ms (before the shadow rays had a tree of their own the
// middle one was best: 16.41 / 15.70 / 16.57 ms).
#define RT_GROUP_SHIFT_MAX 4
#endif
constexpr int BLOCK = RT_BLOCK;
static_assert(BLOCK == 64, "the LDS traversal stack is sized for one wave per workgroup");

struct LaunchArgs {
    rtamd_camera cam;
    uint64_t trace_key, pixel_key;
    int64_t n;
    const int32_t* pixel_index;
    const int32_t* sample_index;
    const double* a3;           // plane points / origins
    const double* b3;           // lens points / directions
    const int32_t* tile_ids;
    int32_t depth0;
    int32_t sample_begin, n_samples, tiles_x;
    int32_t group_shift;        // tile mode: log2 of the samples of one pixel that share a wave (0, 2, 4 or 6)
    int32_t refill;             // refill_kernel: 64-sample groups of the work list one wave owns
    rtamd_sample_out out;
    int64_t* sum_rgb8;
    double* sum_sq;
    unsigned long long* counters;
};

// Tile mode: sample k = (tile position, wave inside the tile, lane).  A wave holds a square
// block of the tile's pixels at 2^group_shift consecutive sample indices each: 8x8 pixels x 1
// sample, 4x4 x 4, 2x2 x 16 or 1 pixel x 64.  Smaller blocks keep the hit points of a wave -
// the origins of its shadow rays - closer together.
struct TileSlot {
    int32_t col, row, soff;
    int64_t acc_row;        // row of the pixel in the tile-compact accumulators
    int64_t ko;             // row of the sample in the per-sample outputs: (tile, sample, pixel in tile) order
};
__device__ __forceinline__ TileSlot tile_slot(const LaunchArgs& A, int64_t k) {
    const int32_t lane = (int32_t)(k & 63);
    const int64_t q = k >> 6;
    const int32_t waves_per_tile = A.n_samples;             // 64 * n_samples / 64
    const int32_t wv = (int32_t)(q % waves_per_tile);
    const int64_t tpos = q / waves_per_tile;
    const int32_t g = A.group_shift;
    const int32_t side_shift = 3 - (g >> 1);                // block side = 8, 4, 2, 1 pixels
    const int32_t block = wv & ((1 << g) - 1);              // blocks per tile = 64 / side^2 = 2^g
    const int32_t sgroup = wv >> g;
    const int32_t in_block = lane & ((64 >> g) - 1);
    const int32_t bx = block & ((8 >> side_shift) - 1), by = block >> (3 - side_shift);
    const int32_t px = (bx << side_shift) + (in_block & ((1 << side_shift) - 1));
    const int32_t py = (by << side_shift) + (in_block >> side_shift);
    const int32_t pix = py * 8 + px;
    const int32_t tile = A.tile_ids[tpos];
    TileSlot ts;
    ts.soff = (sgroup << g) + (lane >> (6 - g));
    ts.col = (tile % A.tiles_x) * 8 + (pix & 7);
    ts.row = (tile / A.tiles_x) * 8 + (pix >> 3);
    ts.acc_row = tpos * 64 + pix;
    ts.ko = (tpos * A.n_samples + ts.soff) * 64 + pix;
    return ts;
}

// PinholeCamera.ray_direction (camera.py:31-33) / ThinLensCamera.ray_direction (:222-225)
__device__ __forceinline__ Ray camera_ray(const rtamd_camera& c, V3 plane_point, V3 lens_point) {
    Ray r;
    const V3 u = ld3(c.u), v = ld3(c.v), w = ld3(c.w);
    if (c.thin_lens) {
        const V3 f = (plane_point * c.focal_plane_distance) / c.viewing_distance;
        const V3 d = ((f.x - lens_point.x) * u + (f.y - lens_point.y) * v) - c.focal_plane_distance * w;
        r.o = lens_point;
        r.d = d / norm(d);
    } else {
        const V3 d = (plane_point.x * u + plane_point.y * v) - c.viewing_distance * w;
        r.o = ld3(c.eye);
        r.d = d / norm(d);
    }
    return r;
}

// Tile mode: the camera ray of sample k (camera.py:78 (+ :246): one pattern choice per pixel, plus an
// independent disk pattern for a thin lens; camera.py:90-93: the view-plane point).  Returns false
// for samples of pixels outside the image.
__device__ __forceinline__ bool tile_ray(const rtamd_scene_desc& S, const LaunchArgs& A, int64_t k, Ray& ray,
                                         int32_t& pixel, int32_t& sample, int64_t& acc_row, int64_t& ko) {
    const TileSlot ts = tile_slot(A, k);
    const int32_t col = ts.col, row = ts.row, soff = ts.soff;
    if (col >= A.cam.width || row >= A.cam.height) return false;
    pixel = row * A.cam.width + col;
    sample = A.sample_begin + soff;
    acc_row = ts.acc_row;
    ko = ts.ko;
    const rtamd_sampler sm = S.samplers[A.cam.sampler];
    Draw pd;
    pd.key = mix64(A.pixel_key ^ (uint64_t)(uint32_t)pixel);
    pd.sample = 0;
    pd.counter = 0;
    const uint32_t pat = sm.regular ? 0u : pd.index((uint32_t)sm.num_patterns);
    const double* sq = S.sampler_data + sm.square_offset + 2 * ((int64_t)pat * sm.samples_per_pattern + sample);
    const V3 pp = v3(A.cam.pixel_size * (col - A.cam.width / 2 + sq[0]),
                     A.cam.pixel_size * (row - A.cam.height / 2 + sq[1]), 0.0);
    V3 lens = v3(0, 0, 0);
    if (A.cam.thin_lens) {
        const uint32_t dpat = sm.regular ? 0u : pd.index((uint32_t)sm.num_patterns);
        const double* dk = S.sampler_data + sm.disk_offset + 2 * ((int64_t)dpat * sm.samples_per_pattern + sample);
        lens = ld3(A.cam.eye) + v3(dk[0], dk[1], 0.0) * A.cam.lens_radius;   // camera.py:262
    }
    ray = camera_ray(A.cam, pp, lens);
    return true;
}

// The builds of trace_kernel.  Every build renders the same frames; which one a launch takes depends on what the
// library has read from the scene's tables (launch() below).
//   K_ANALYTIC     scenes without a kd-tree (analytic primitives only): no traversal code, no LDS stack, and the
//                  registers go to the shading loop
//   K_TREES        any scene with kd-trees: the general bounce loop
//   K_EXT          scenes with two-child materials (Transparent, Dielectric) or textured ones (SV_Matte): a per-lane
//                  list of waiting child rays in private memory and a texture colour held across the shadow queries
//   K_AREA_TINY / K_AREA / K_AREA_FUSED
//                  whole frames (accumulators only) of a scene with one kd-tree object and one light under the
//                  AreaLightTracer, which never bounces: straight-line code.  _TINY serves trees of a few levels
//                  (single lanes scan the leaves of every kind of ray); the others leave that code out; _FUSED also
//                  leaves out the general shadow path where the host has shown that no wave takes it
//   K_STRAIGHT / K_STRAIGHT_ONE
//                  whole frames of kd scenes under the plain tracer whose materials spawn no child rays (Matte, Phong,
//                  Emissive): straight-line code as well; _ONE for one kd-tree object of more than a few levels and one
//                  light (shadow tests ask the tree last, the light loop is one pass, no scans by single lanes)
//   K_PATH         whole frames under the PathTracer: the closest-hit query and the lobe sample, no direct lighting
enum { K_ANALYTIC = 0, K_TREES, K_EXT, K_AREA_TINY, K_AREA, K_AREA_FUSED, K_STRAIGHT, K_STRAIGHT_ONE, K_PATH };
template <int KIND> struct Kind {
    static constexpr bool ext = KIND == K_EXT;
    static constexpr bool trees = KIND != K_ANALYTIC;
    static constexpr bool area_only = KIND == K_AREA_TINY || KIND == K_AREA || KIND == K_AREA_FUSED;
    static constexpr bool tiny_ok = KIND != K_AREA && KIND != K_AREA_FUSED && KIND != K_STRAIGHT_ONE;
    static constexpr bool no_bounce = KIND == K_STRAIGHT || KIND == K_STRAIGHT_ONE;
    static constexpr bool fused = KIND == K_AREA_FUSED;
    static constexpr bool path = KIND == K_PATH;
    static constexpr bool one_tree = area_only || KIND == K_STRAIGHT_ONE;
    static constexpr bool sums_only = area_only || no_bounce || path;       // no per-sample outputs
};
template <int MODE, int COUNT, int KIND>
__global__ __launch_bounds__(BLOCK, Kind<KIND>::trees ? RT_MIN_WAVES : RT_MIN_WAVES_ANALYTIC) void trace_kernel(const rtamd_scene_desc S, const LaunchArgs A) {
    typedef Kind<KIND> K;
    constexpr bool TREES = K::trees;
    // Which 64 samples this wave takes: workgroup i takes samples 64 i .. 64 i + 63 of the work list.  (Giving each of
    // the 8 XCDs, which workgroups are dealt to round-robin, a few contiguous bands of the frame instead measured no
    // faster in either round - 8 / 16 / 32 / 64 segments - the scene is served from every L2 at more than 80 % hit
    // rate either way; that remap is no longer in the code.)
#ifdef RT_PHASE_TIMERS
    const long long wave_start = clock64();
#endif
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = k < A.n;
    Work<COUNT> work;
#ifdef RT_PHASE_TIMERS
    work.mark_t = wave_start;
#endif
    int64_t acc_row = -1;
    int64_t ko = k;             // row of this sample in the per-sample outputs

    Ray ray;
    int32_t pixel = 0, sample = 0;
    if (live) {
        if (MODE == MODE_RAYS) {
            pixel = A.pixel_index ? A.pixel_index[k] : 0;
            sample = A.sample_index ? A.sample_index[k] : 0;
            ray.o = ld3(A.a3 + 3 * k);
            ray.d = ld3(A.b3 + 3 * k);
        } else if (MODE == MODE_PIXELS) {
            pixel = A.pixel_index[k];
            sample = A.sample_index[k];
            const V3 lens = A.cam.thin_lens ? ld3(A.b3 + 3 * k) : v3(0, 0, 0);
            ray = camera_ray(A.cam, ld3(A.a3 + 3 * k), lens);
        } else {
            live = tile_ray(S, A, k, ray, pixel, sample, acc_row, ko);
        }

    }
    // every lane of the wave enters the trace; lanes without a sample only help
    {
        if (!live) { ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); }
        Draw dr;
        dr.key = mix64(A.trace_key ^ (uint64_t)(uint32_t)pixel);
        dr.sample = (uint32_t)sample;
        dr.counter = 0;
        Primary first;
        // the wave's kd traversal stack (one wave per workgroup)
        __shared__ double stack_t[TREES ? LDS_STACK * 64 : 1];
        __shared__ int32_t stack_node[TREES ? LDS_STACK * 64 : 1];
        __shared__ int32_t stack_slot[TREES ? 64 : 1];
        KdStack stk;
        stk.node = stack_node;
        stk.t = stack_t;
        stk.slot = stack_slot;
        stk.tiny_ok = K::tiny_ok;
        // (the benchmark's kernel keeps to the code its frame runs: its lights lie outside the tree's box)
        stk.prefilter_ok = !(K::area_only && !K::tiny_ok);
        stk.fused_always = K::fused;
        NoFeed no_feed;
        const V3 c = trace_wave<COUNT, K::ext, TREES, K::area_only, NoFeed, K::no_bounce, K::path, K::one_tree>(
            S, stk, live, ray, A.depth0, dr, work, first, no_feed);
        if (live) {
            const int32_t r8 = quantise(c.x), g8 = quantise(c.y), b8 = quantise(c.z);
            if (!K::sums_only) {
            if (A.out.rgb) { A.out.rgb[3 * ko] = c.x; A.out.rgb[3 * ko + 1] = c.y; A.out.rgb[3 * ko + 2] = c.z; }
            if (A.out.rgb8) { A.out.rgb8[3 * ko] = r8; A.out.rgb8[3 * ko + 1] = g8; A.out.rgb8[3 * ko + 2] = b8; }
            if (A.out.hit_object) A.out.hit_object[ko] = first.object;
            if (A.out.hit_prim) A.out.hit_prim[ko] = first.prim;
            if (A.out.hit_t) A.out.hit_t[ko] = first.t;
            if (A.out.n_rays) { A.out.n_rays[2 * ko] = (int32_t)work.closest; A.out.n_rays[2 * ko + 1] = (int32_t)work.shadow; }
            }
        }
        if (MODE == MODE_TILES) {
            // The 2^group_shift samples of a pixel that share this wave sit in lanes a multiple
            // of 64 >> group_shift apart (tile_slot): their contributions are summed across
            // those lanes first, and the lane of the pixel's first sample adds the total, so
            // the accumulators see one atomic per pixel, channel and wave.  All lanes of a
            // pixel are live or dead together (liveness depends on the pixel alone).
            const int32_t stride = 64 >> A.group_shift;
            const bool adds = live && (int32_t)threadIdx.x < stride;
            if (A.sum_rgb8) {
                long long r = live ? (long long)quantise(c.x) : 0, g = live ? (long long)quantise(c.y) : 0,
                          b = live ? (long long)quantise(c.z) : 0;
                for (int32_t off = stride; off < 64; off <<= 1) {
                    r += __shfl_xor(r, off); g += __shfl_xor(g, off); b += __shfl_xor(b, off);
                }
                if (adds) {
                    unsigned long long* s = reinterpret_cast<unsigned long long*>(A.sum_rgb8) + 3 * acc_row;
                    atomicAdd(s, (unsigned long long)r);
                    atomicAdd(s + 1, (unsigned long long)g);
                    atomicAdd(s + 2, (unsigned long long)b);
                }
            }
            if (A.sum_sq) {
                double r = live ? c.x * c.x : 0.0, g = live ? c.y * c.y : 0.0, b = live ? c.z * c.z : 0.0;
                for (int32_t off = stride; off < 64; off <<= 1) {
                    r += __shfl_xor(r, off); g += __shfl_xor(g, off); b += __shfl_xor(b, off);
                }
                if (adds) {
                    double* s = A.sum_sq + 3 * acc_row;
                    atomicAdd(s, r);
                    atomicAdd(s + 1, g);
                    atomicAdd(s + 2, b);
                }
            }
        }
    }

#ifdef RT_PHASE_TIMERS
    work.mark(35);      // shading after the last query, outputs, accumulators
    if (threadIdx.x == 0) {
        atomicAdd(&g_phase_cycles[16], (unsigned long long)(clock64() - wave_start));
        atomicAdd(&g_phase_cycles[17], 1ull);
    }
#endif
    if (A.counters) {
        // one add per counter per block
        __shared__ unsigned int blk[RTAMD_N_COUNTERS];
        if (threadIdx.x < RTAMD_N_COUNTERS) blk[threadIdx.x] = 0;
        __syncthreads();
        constexpr int N_USED = COUNT == 2 ? 12 : (COUNT == 1 ? 7 : 3);
#pragma unroll
        for (int i = 0; i < N_USED; ++i) {
            unsigned int v = i == 0 ? work.closest : i == 1 ? work.shadow : i == 2 ? (live ? 1u : 0u) : work.slot(i);
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&blk[i], v);
        }
        __syncthreads();
        if (threadIdx.x < N_USED && blk[threadIdx.x]) atomicAdd(&A.counters[threadIdx.x], (unsigned long long)blk[threadIdx.x]);
    }
}

// The samples one wave of refill_kernel owns, handed to its lanes as their paths end.  Paths of
// a scene of mirrors and glossy surfaces end after different numbers of bounces; a lane that is
// done hands its sample in (per-pixel sums in LDS) and takes the next one of the wave's list, so
// later bounces run with full vectors instead of the few lanes whose paths go on.  Which lane
// traces a sample does not enter its colour: every draw is keyed by (pixel, sample, position in
// the sample's own draw sequence).
struct TileFeed {
    static constexpr bool on = true;
    const rtamd_scene_desc& S;
    const LaunchArgs& A;
    int64_t next_k, end_k;                  // wave-uniform: the part of the list not handed out yet
    unsigned long long* bins;               // LDS [64][3]: int(c*255) sums per pixel of the tile
    double* sq;                             // LDS [64][3]: c*c sums
    int32_t bin;                            // the lane's current pixel in its tile
    uint32_t taken = 0;                     // samples this lane has traced or is tracing

    __device__ __forceinline__ void retire(V3 c) {
        if (A.sum_rgb8) {
            atomicAdd(&bins[3 * bin], (unsigned long long)(long long)quantise(c.x));
            atomicAdd(&bins[3 * bin + 1], (unsigned long long)(long long)quantise(c.y));
            atomicAdd(&bins[3 * bin + 2], (unsigned long long)(long long)quantise(c.z));
        }
        if (A.sum_sq) {
            atomicAdd(&sq[3 * bin], c.x * c.x);
            atomicAdd(&sq[3 * bin + 1], c.y * c.y);
            atomicAdd(&sq[3 * bin + 2], c.z * c.z);
        }
    }
    __device__ __forceinline__ bool more() const { return next_k < end_k; }
    // all lanes call; lanes with `want` are given consecutive samples by their rank among them
    __device__ __forceinline__ bool next(bool want, Ray& ray, Draw& dr) {
        const unsigned long long m = __ballot(want);
        if (!m || next_k >= end_k) return false;
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const int64_t k = next_k + rank;
        next_k += __popcll(m);
        if (!want || k >= end_k) return false;
        return start(k, ray, dr);
    }
    __device__ __forceinline__ bool start(int64_t k, Ray& ray, Draw& dr) {
        int32_t pixel = 0, sample = 0;
        int64_t acc_row = 0, ko = 0;
        if (!tile_ray(S, A, k, ray, pixel, sample, acc_row, ko)) return false;     // a pixel outside the image
        bin = (int32_t)(acc_row & 63);
        dr.key = mix64(A.trace_key ^ (uint64_t)(uint32_t)pixel);
        dr.sample = (uint32_t)sample;
        dr.counter = 0;
        taken++;
        return true;
    }
};

// Whole frames of scenes without a kd-tree, accumulators only: a wave owns A.refill consecutive
// 64-sample groups of the tile work list (all of one tile: A.refill divides the samples per
// pixel), starts on the first and refills lanes from the others (TileFeed).  Sums per pixel are
// kept in LDS and added to the accumulators once per pixel, channel and wave.
// (Scenes with a kd-tree do not refill, not even trees of a few levels under the path tracer, where single lanes
// scan the leaves of every ray and a refilled primary ray loses nothing: the path-traced Cornell box took 4.42 ms
// against 3.40 ms with one sample per lane.  Of its lanes 0.73 hold a live path per pass; what a pass costs there is
// set by how far the wave's walks diverge, and rays of different pixels and depths in one wave diverge more.)
template <int COUNT>
__global__ __launch_bounds__(BLOCK, RT_MIN_WAVES_ANALYTIC) void refill_kernel(const rtamd_scene_desc S, const LaunchArgs A) {
    __shared__ unsigned long long bins[192];
    __shared__ double sq[192];
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < 192; i += BLOCK) { bins[i] = 0; sq[i] = 0.0; }
    __syncthreads();
    const int64_t k0 = (int64_t)blockIdx.x * BLOCK * A.refill;
    TileFeed feed{S, A, k0 + BLOCK, min(k0 + (int64_t)BLOCK * A.refill, A.n), bins, sq, 0};
    Ray ray;
    ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1);
    Draw dr;
    dr.key = 0; dr.sample = 0; dr.counter = 0;
    const bool live = k0 + lane < A.n && feed.start(k0 + lane, ray, dr);
    Work<COUNT> work;
    Primary first;
    KdStack stk;
    stk.node = nullptr; stk.t = nullptr; stk.slot = nullptr;     // no kd-tree, no traversal stack
    stk.tiny_ok = false;
    stk.fused_always = false;
    stk.prefilter_ok = false;
    trace_wave<COUNT, false, false, false, TileFeed>(S, stk, live, ray, A.depth0, dr, work, first, feed);
    __syncthreads();
    // lane l adds the sums of pixel l of the wave's tile (zero where the wave traced nothing)
    const int64_t acc_row = k0 / ((int64_t)64 * A.n_samples) * 64 + lane;
    if (A.sum_rgb8) {
        unsigned long long* s = reinterpret_cast<unsigned long long*>(A.sum_rgb8) + 3 * acc_row;
        for (int c = 0; c < 3; ++c)
            if (bins[3 * lane + c]) atomicAdd(s + c, bins[3 * lane + c]);
    }
    if (A.sum_sq) {
        for (int c = 0; c < 3; ++c)
            if (sq[3 * lane + c] != 0.0) atomicAdd(A.sum_sq + 3 * acc_row + c, sq[3 * lane + c]);
    }
    if (A.counters) {
        unsigned int v[3] = {work.closest, work.shadow, feed.taken};
        for (int i = 0; i < 3; ++i) {
            for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
            if (lane == 0 && v[i]) atomicAdd(&A.counters[i], (unsigned long long)v[i]);
        }
    }
}

// A frame's integer sums on their way to the parent (the gather of the multi-GPU job): every 64-bit
// sum saturated to the int32 it travels as, and the sum cleared for the next frame if asked.  One pass
// instead of a clamp, a narrowing copy and a memset.
__global__ __launch_bounds__(256) void pack_sums_kernel(int64_t* __restrict__ sums, int32_t* __restrict__ wire, int64_t n, int clear) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t v = sums[i];
        wire[i] = v > 2147483647LL ? 2147483647 : (v < -2147483648LL ? (int32_t)0x80000000 : (int32_t)v);
        if (clear) sums[i] = 0;
    }
}

thread_local char g_error[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_error, sizeof(g_error), fmt, detail);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(e));
    return RTAMD_EHIP;
}

template <int MODE>
int launch(const rtamd_scene_desc& S, bool extended, int tree_objects, int max_tree_depth, bool bounces, bool fused_always, LaunchArgs& A, uint32_t flags, void* stream) {
    const bool trees = S.n_trees > 0;
    if (A.n <= 0) return RTAMD_OK;
    const int64_t blocks = (A.n + BLOCK - 1) / BLOCK;
    if (blocks > 0x7fffffffLL) return fail(RTAMD_EINVAL, "too many samples for one launch%s");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool count = (flags & RTAMD_COUNT_WORK) != 0;
    const bool executed = (flags & RTAMD_COUNT_EXECUTED) != 0;
    if (count && executed) return fail(RTAMD_EINVAL, "RTAMD_COUNT_WORK and RTAMD_COUNT_EXECUTED exclude each other%s");
    if (executed && !A.counters) return fail(RTAMD_EINVAL, "RTAMD_COUNT_EXECUTED needs a counter block%s");
    const dim3 grid((unsigned)blocks), block(BLOCK);
    // whole frames, accumulators only, have builds of their own where the scene allows (see Kind)
    const bool sums_only = MODE == MODE_TILES && !A.out.rgb && !A.out.rgb8 && !A.out.hit_object && !A.out.hit_prim &&
                           !A.out.hit_t && !A.out.n_rays;
    const bool tiny_trees = max_tree_depth < RT_TINY_DEPTH;
    const bool one_tree_one_light = tree_objects == 1 && S.n_lights == 1;
    bool area_only = false, no_bounce = false, path = false;
    area_only = sums_only && !extended && trees && S.tracer_kind == RTAMD_TRACER_AREA && one_tree_one_light;
    no_bounce = sums_only && !extended && trees && !area_only && !bounces && S.tracer_kind == RTAMD_TRACER_SHADE && !count && !executed;
    path = sums_only && !extended && trees && S.tracer_kind == RTAMD_TRACER_PATH && !count && !executed;
#if !defined(RT_NO_REFILL) && !defined(RT_ONLY_HOT)
    // waves that refill their lanes: whole frames of scenes without a kd-tree
    if (sums_only && !extended && !count && !executed && !trees) {
        A.refill = 0;
        for (int r = RT_REFILL_MAX; r > 1; --r)
            if (A.n_samples % r == 0) { A.refill = r; break; }
        if (A.refill > 1) {
            const int64_t waves = (A.n + (int64_t)BLOCK * A.refill - 1) / ((int64_t)BLOCK * A.refill);
            hipLaunchKernelGGL((refill_kernel<0>), dim3((unsigned)waves), block, 0, st, S, A);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "kernel launch");
            return RTAMD_OK;
        }
    }
#endif
#define RT_LAUNCH(COUNT_, KIND_) hipLaunchKernelGGL((trace_kernel<MODE, COUNT_, KIND_>), grid, block, 0, st, S, A)
#define RT_LAUNCH_TILES(COUNT_, KIND_) hipLaunchKernelGGL((trace_kernel<MODE_TILES, COUNT_, KIND_>), grid, block, 0, st, S, A)
#ifdef RT_ONLY_HOT
    // experiment builds (tools/variant_bench.sh): only the whole-frame kernel of the benchmark scene
    if (area_only && !count && !executed && fused_always) RT_LAUNCH_TILES(0, K_AREA_FUSED);
    else if (area_only && !count && !executed) RT_LAUNCH_TILES(0, K_AREA);
    else return fail(RTAMD_EUNSUPPORTED, "this build holds the whole-frame area-light kernel only%s");
#elif defined(RT_ONLY_TILES)
    // experiment builds: the uncounted whole-frame kernels only
    if (MODE != MODE_TILES || count || executed) return fail(RTAMD_EUNSUPPORTED, "this build holds the uncounted tile kernels only%s");
    if (extended) RT_LAUNCH_TILES(0, K_EXT);
    else if (area_only && !tiny_trees && fused_always) RT_LAUNCH_TILES(0, K_AREA_FUSED);
    else if (area_only && !tiny_trees) RT_LAUNCH_TILES(0, K_AREA);
    else if (area_only) RT_LAUNCH_TILES(0, K_AREA_TINY);
    else if (no_bounce && one_tree_one_light && !tiny_trees) RT_LAUNCH_TILES(0, K_STRAIGHT_ONE);
    else if (no_bounce) RT_LAUNCH_TILES(0, K_STRAIGHT);
    else if (path) RT_LAUNCH_TILES(0, K_PATH);
    else if (trees) RT_LAUNCH_TILES(0, K_TREES);
    else RT_LAUNCH_TILES(0, K_ANALYTIC);
#else
    if (extended) {
        if (count) RT_LAUNCH(1, K_EXT);
        else if (executed) RT_LAUNCH(2, K_EXT);
        else RT_LAUNCH(0, K_EXT);
    } else if (trees) {
        if (count) RT_LAUNCH(1, K_TREES);
        else if (area_only && executed) RT_LAUNCH_TILES(2, K_AREA_TINY);
        else if (area_only && !tiny_trees && fused_always) RT_LAUNCH_TILES(0, K_AREA_FUSED);
        else if (area_only && !tiny_trees) RT_LAUNCH_TILES(0, K_AREA);
        else if (area_only) RT_LAUNCH_TILES(0, K_AREA_TINY);
        else if (executed) RT_LAUNCH(2, K_TREES);
        else if (no_bounce && one_tree_one_light && !tiny_trees) RT_LAUNCH_TILES(0, K_STRAIGHT_ONE);
        else if (no_bounce) RT_LAUNCH_TILES(0, K_STRAIGHT);
        else if (path) RT_LAUNCH_TILES(0, K_PATH);
        else RT_LAUNCH(0, K_TREES);
    } else {
        if (count) RT_LAUNCH(1, K_ANALYTIC);
        else if (executed) RT_LAUNCH(2, K_ANALYTIC);
        else RT_LAUNCH(0, K_ANALYTIC);
    }
#endif
#undef RT_LAUNCH
#undef RT_LAUNCH_TILES
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return RTAMD_OK;
}

}  // namespace

struct rtamd_scene {
    rtamd_scene_desc desc;
    int device;
    rtamd_sampler* samplers_host;   // host copies of the small tables, for argument checks
    int max_tree_depth;
    bool extended;                  // two-child or textured materials: the EXT kernels run
    bool two_child;                 // Transparent / Dielectric traced recursively
    bool bounces;                   // some hit can spawn a child ray (mirror, glossy, two-child materials, path tracing)
    bool fused_always;              // see KdStack::fused_always: shown here, on the host copies of the tables
    int tree_objects;               // objects of kind RTAMD_KDTREE (one tree may stand behind several)
};

extern "C" {

int rtamd_abi_version(void) { return RTAMD_ABI_VERSION; }

#ifdef RT_PHASE_TIMERS
// profiling builds only: read (and clear) the per-phase cycle table
int rtamd_debug_phase_cycles(unsigned long long* out32, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(rt::g_phase_cycles), 64 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "reading the phase table");
    e = hipMemcpyFromSymbol(out32 + 64, HIP_SYMBOL(rt::g_phase_hist), 64 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "reading the query histogram");
    if (reset) {
        unsigned long long zero[64] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(rt::g_phase_hist), zero, sizeof(zero));
        if (e != hipSuccess) return hip_fail(e, "clearing the query histogram");
        e = hipMemcpyToSymbol(HIP_SYMBOL(rt::g_phase_cycles), zero, sizeof(zero));
        if (e != hipSuccess) return hip_fail(e, "clearing the phase table");
    }
    return RTAMD_OK;
}
#endif

const char* rtamd_last_error(void) { return g_error; }

int rtamd_device_check(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RTAMD_ENODEVICE, "no HIP device is available%s");
    return RTAMD_OK;
}

int rtamd_scene_destroy(rtamd_scene* scene);

static int scene_create(const rtamd_scene_desc* d, rtamd_scene** out);

int rtamd_scene_create(const rtamd_scene_desc* d, rtamd_scene** out) {
    try {
        return scene_create(d, out);
    } catch (...) {             // host copies of the small tables are std::vectors
        return fail(RTAMD_EINVAL, "out of host memory%s");
    }
}

static int scene_create(const rtamd_scene_desc* d, rtamd_scene** out) {
    if (!d || !out) return fail(RTAMD_EINVAL, "null argument%s");
    if (d->abi_version != RTAMD_ABI_VERSION) return fail(RTAMD_EINVAL, "descriptor ABI version mismatch%s");
    if (d->tracer_kind < RTAMD_TRACER_SHADE || d->tracer_kind > RTAMD_TRACER_WHITTED)
        return fail(RTAMD_EUNSUPPORTED, "unknown tracer kind%s");
    if (d->n_objects < 0 || d->n_materials < 0 || d->n_lights < 0 || d->n_samplers < 1 || d->n_trees < 0)
        return fail(RTAMD_EINVAL, "negative table size, or no view-plane sampler%s");
    if ((d->n_objects && !d->objects) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights) ||
        !d->samplers || !d->sampler_data)
        return fail(RTAMD_EINVAL, "a non-empty table has a null pointer%s");
    if (d->n_trees && (!d->trees || !d->kd_nodes || !d->kd_triples || !d->tri_geom || !d->tri_prim || !d->prim_normals ||
                       !d->prim_normal_offset || !d->prim_material))
        return fail(RTAMD_EINVAL, "kd-tree tables missing%s");
    if (d->leaf_boxes && (!d->tri_rows || !d->chunk_boxes || !d->tri_order))
        return fail(RTAMD_EINVAL, "leaf_boxes needs tri_rows, chunk_boxes and tri_order beside it%s");
    if (d->chunk_tree && (!d->chunk_boxes || !d->tri_order))
        return fail(RTAMD_EINVAL, "chunk_tree needs chunk_boxes and tri_order beside it%s");
    if (d->n_refs < 0 || d->n_refs * 72 > 0x7fffffffLL)
        return fail(RTAMD_EUNSUPPORTED, "more than 2^31 bytes of leaf triangle rows%s");
    if (d->ambient.kind != RTAMD_AMBIENT_CONSTANT && d->ambient.kind != RTAMD_AMBIENT_OCCLUDER)
        return fail(RTAMD_EUNSUPPORTED, "unknown ambient kind%s");
    if (d->ambient.kind == RTAMD_AMBIENT_OCCLUDER && (d->ambient.sampler < 0 || d->ambient.sampler >= d->n_samplers))
        return fail(RTAMD_EINVAL, "ambient occluder sampler out of range%s");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    rtamd_scene* s = new (std::nothrow) rtamd_scene;
    if (!s) return fail(RTAMD_EINVAL, "out of host memory%s");
    s->desc = *d;
    s->device = dev;
    s->max_tree_depth = 0;
    s->extended = false;
    s->two_child = false;
    s->bounces = d->tracer_kind == RTAMD_TRACER_PATH;
    s->tree_objects = 0;
    s->samplers_host = new (std::nothrow) rtamd_sampler[d->n_samplers];
    if (!s->samplers_host) { delete s; return fail(RTAMD_EINVAL, "out of host memory%s"); }
    e = hipMemcpy(s->samplers_host, d->samplers, sizeof(rtamd_sampler) * d->n_samplers, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the sampler table"); }
    for (int i = 0; i < d->n_samplers; ++i) {
        const rtamd_sampler& sm = s->samplers_host[i];
        const int64_t n = (int64_t)sm.num_patterns * sm.samples_per_pattern;
        if (sm.num_patterns < 1 || sm.samples_per_pattern < 1 || sm.square_offset < 0 || sm.disk_offset < 0 ||
            sm.hemi_offset < 0 || sm.square_offset + 2 * n > d->n_sampler_doubles ||
            sm.disk_offset + 2 * n > d->n_sampler_doubles || sm.hemi_offset + 3 * n > d->n_sampler_doubles) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "a sampler table lies outside sampler_data%s");
        }
    }
    // sampler references of materials and lights (small tables: checked on a host copy, one copy per table)
    std::vector<rtamd_material> materials((size_t)d->n_materials);
    if (d->n_materials) {
        e = hipMemcpy(materials.data(), d->materials, sizeof(rtamd_material) * d->n_materials, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the material table"); }
    }
    for (int i = 0; i < d->n_materials; ++i) {
        const rtamd_material& m = materials[i];
        const bool draws = m.kind == RTAMD_GLOSSY_REFLECTIVE || (d->tracer_kind == RTAMD_TRACER_PATH && m.kind == RTAMD_MATTE);
        if (m.kind < RTAMD_MATTE || m.kind > RTAMD_SV_MATTE) { rtamd_scene_destroy(s); return fail(RTAMD_EUNSUPPORTED, "unknown material kind%s"); }
        if (m.kind == RTAMD_SV_MATTE) {
            // SV_Matte has shade() only (material.py:292-326)
            if (d->tracer_kind == RTAMD_TRACER_AREA || d->tracer_kind == RTAMD_TRACER_PATH) {
                rtamd_scene_destroy(s);
                return fail(RTAMD_EUNSUPPORTED, "an SV_Matte material needs the shade or Whitted tracer%s");
            }
            if (m.texture < 0 || m.texture >= d->n_textures) {
                rtamd_scene_destroy(s);
                return fail(RTAMD_EINVAL, "a textured material refers to a texture outside the table%s");
            }
            s->extended = true;
        }
        const bool two_children = m.kind == RTAMD_TRANSPARENT || m.kind == RTAMD_DIELECTRIC;
        if (two_children || m.kind == RTAMD_REFLECTIVE || m.kind == RTAMD_GLOSSY_REFLECTIVE) s->bounces = true;
        if (two_children && d->tracer_kind != RTAMD_TRACER_AREA) {
            // Dielectric.shade hands trace_ray a third argument (material.py:459); only
            // WhittedTracer accepts it (tracer.py:43)
            if (m.kind == RTAMD_DIELECTRIC && d->tracer_kind != RTAMD_TRACER_WHITTED) {
                rtamd_scene_destroy(s);
                return fail(RTAMD_EUNSUPPORTED, "a Dielectric material needs the Whitted tracer%s");
            }
            if (d->depth_limit + 1 > RTAMD_MAX_BRANCH_LEVELS) {
                rtamd_scene_destroy(s);
                return fail(RTAMD_EUNSUPPORTED, "depth limit too large for two-child materials%s");
            }
            s->extended = true;
            s->two_child = true;
        }
        if (d->tracer_kind == RTAMD_TRACER_PATH && m.kind != RTAMD_MATTE && m.kind != RTAMD_EMISSIVE) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EUNSUPPORTED, "the path tracer needs Matte or Emissive materials%s");
        }
        if (draws && (m.sampler < 0 || m.sampler >= d->n_samplers)) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "a material that samples a lobe has no sampler%s");
        }
    }
    if (d->n_transforms < 0 || (d->n_transforms && !d->transforms)) {
        rtamd_scene_destroy(s);
        return fail(RTAMD_EINVAL, "transform table missing%s");
    }
    if (d->n_textures < 0 || (d->n_textures && (!d->textures || !d->texels))) {
        rtamd_scene_destroy(s);
        return fail(RTAMD_EINVAL, "texture tables missing%s");
    }
    std::vector<rtamd_texture> textures((size_t)d->n_textures);
    if (d->n_textures) {
        e = hipMemcpy(textures.data(), d->textures, sizeof(rtamd_texture) * d->n_textures, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the texture table"); }
    }
    for (int i = 0; i < d->n_textures; ++i) {
        const rtamd_texture& t = textures[i];
        if (t.width < 1 || t.height < 1 || t.hres < 1 || t.vres < 1 || t.hres > t.width || t.vres > t.height ||
            t.offset < 0 || t.offset + 3 * (int64_t)t.width * t.height > d->n_texel_bytes) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "a texture lies outside texels or has bad extents%s");
        }
        if (t.mapping != RTAMD_MAP_SPHERICAL) { rtamd_scene_destroy(s); return fail(RTAMD_EUNSUPPORTED, "unknown texture mapping%s"); }
    }
    std::vector<rtamd_light> lights((size_t)d->n_lights);
    if (d->n_lights) {
        e = hipMemcpy(lights.data(), d->lights, sizeof(rtamd_light) * d->n_lights, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the light table"); }
    }
    for (int i = 0; i < d->n_lights; ++i) {
        const rtamd_light& l = lights[i];
        if (l.kind < RTAMD_POINT || l.kind > RTAMD_ENVIRONMENT) { rtamd_scene_destroy(s); return fail(RTAMD_EUNSUPPORTED, "unknown light kind%s"); }
        if ((l.kind == RTAMD_AREA || l.kind == RTAMD_ENVIRONMENT) && (l.sampler < 0 || l.sampler >= d->n_samplers)) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "an area or environment light has no sampler%s");
        }
    }
    std::vector<rtamd_object> objects((size_t)d->n_objects);
    if (d->n_objects) {
        e = hipMemcpy(objects.data(), d->objects, sizeof(rtamd_object) * d->n_objects, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the object table"); }
    }
    for (int i = 0; i < d->n_objects; ++i) {
        const rtamd_object& o = objects[i];
        const bool tree = o.kind == RTAMD_KDTREE;
        if (tree) s->tree_objects++;
        if (o.n_inner < 0 || (o.n_inner > 0 && (!o.has_transform || o.inner_offset < 0 ||
                                                 (int64_t)o.inner_offset + o.n_inner > d->n_transforms))) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "an object's nested transforms lie outside the transform table%s");
        }
        if (o.kind < RTAMD_SPHERE || o.kind > RTAMD_KDTREE || (tree && (o.tree < 0 || o.tree >= d->n_trees)) ||
            (!tree && (o.material < 0 || o.material >= d->n_materials)) ||
            (o.kind == RTAMD_CHECKER_PLANE && (o.alt_material < 0 || o.alt_material >= d->n_materials))) {
            rtamd_scene_destroy(s);
            return fail(RTAMD_EINVAL, "an object refers to a tree or material outside its table%s");
        }
    }
    std::vector<rtamd_tree> trees((size_t)d->n_trees);
    if (d->n_trees) {
        e = hipMemcpy(trees.data(), d->trees, sizeof(rtamd_tree) * d->n_trees, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rtamd_scene_destroy(s); return hip_fail(e, "reading the tree table"); }
    }
    for (int i = 0; i < d->n_trees; ++i) {
        const rtamd_tree& t = trees[i];
        if (t.depth + 2 > STACK_MAX || t.walk_depth + 2 > STACK_MAX) { rtamd_scene_destroy(s); return fail(RTAMD_EUNSUPPORTED, "kd-tree deeper than the traversal stack%s"); }
        if (t.node_offset < 0 || t.node_offset >= d->n_kd_nodes) { rtamd_scene_destroy(s); return fail(RTAMD_EINVAL, "tree root outside kd_nodes%s"); }
        if (t.shadow_root < 0 || t.shadow_root >= d->n_kd_nodes) { rtamd_scene_destroy(s); return fail(RTAMD_EINVAL, "shadow tree root outside kd_nodes%s"); }
        if (t.depth > s->max_tree_depth) s->max_tree_depth = t.depth;
    }
    // Whether every wave of a whole-frame launch can trace both shadow rays of a hit in one shadow-tree
    // query (direct_light_wave): one untransformed tree with a shadow tree, an ambient occluder, and one
    // shadow-casting area light whose emitter - corner + [0,1] edge_u + [0,1] edge_v - is separated from the
    // tree's box along some axis by a hundredth of the scene's size: then every light ray leaves the box
    // before its limit, and a ray's every possible hit lies before the light.
    s->fused_always = false;
    if (d->tracer_kind == RTAMD_TRACER_AREA && !s->extended && s->tree_objects == 1 && d->n_lights == 1 &&
        d->ambient.kind == RTAMD_AMBIENT_OCCLUDER && d->leaf_aabb && lights[0].kind == RTAMD_AREA && lights[0].cast_shadow) {
        const rtamd_object* tob = nullptr;
        for (int i = 0; i < d->n_objects; ++i) if (objects[i].kind == RTAMD_KDTREE) tob = &objects[i];
        if (tob && !tob->has_transform && trees[tob->tree].shadow_root > 0) {
            const rtamd_tree& t = trees[tob->tree];
            const rtamd_light& l = lights[0];
            double size = 0;
            for (int k = 0; k < 3; ++k) size = fmax(size, fmax(fmax(fabs(t.bbox_lo[k]), fabs(t.bbox_hi[k])), fabs(l.position[k]) + fabs(l.edge_u[k]) + fabs(l.edge_v[k])));
            for (int k = 0; k < 3; ++k) {
                const double lo = l.position[k] + fmin(0.0, l.edge_u[k]) + fmin(0.0, l.edge_v[k]);
                const double hi = l.position[k] + fmax(0.0, l.edge_u[k]) + fmax(0.0, l.edge_v[k]);
                if (lo > t.bbox_hi[k] + 0.01 * size || hi < t.bbox_lo[k] - 0.01 * size) s->fused_always = true;
            }
        }
    }
    *out = s;
    return RTAMD_OK;
}

int rtamd_scene_destroy(rtamd_scene* scene) {
    if (scene) delete[] scene->samplers_host;
    delete scene;
    return RTAMD_OK;
}

// A handle's tables live on the device that was current when it was created; a launch from
// another device would hand that device's kernel foreign pointers.
static int check_device(const rtamd_scene* scene) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev != scene->device) return fail(RTAMD_EINVAL, "the scene belongs to another device than the current one%s");
    return RTAMD_OK;
}

static int check_camera(const rtamd_scene* scene, const rtamd_camera* c) {
    if (!scene || !c) return fail(RTAMD_EINVAL, "null scene or camera%s");
    if (int rc = check_device(scene)) return rc;
    if (c->width <= 0 || c->height <= 0) return fail(RTAMD_EINVAL, "empty view plane%s");
    if (c->sampler < 0 || c->sampler >= scene->desc.n_samplers) return fail(RTAMD_EINVAL, "camera sampler out of range%s");
    return RTAMD_OK;
}

int rtamd_render_pixels(rtamd_scene* scene, const rtamd_camera* camera, uint64_t seed, int64_t n,
                        const int32_t* pixel_index, const int32_t* sample_index, const double* plane_points,
                        const double* lens_points, const rtamd_sample_out* out, uint64_t* counters,
                        uint32_t flags, void* stream) {
    int rc = check_camera(scene, camera);
    if (rc) return rc;
    if (n < 0) return fail(RTAMD_EINVAL, "negative sample count%s");
    if (n == 0) return RTAMD_OK;
    if (!pixel_index || !sample_index || !plane_points || !out) return fail(RTAMD_EINVAL, "null sample buffer%s");
    if (camera->thin_lens && !lens_points) return fail(RTAMD_EINVAL, "a thin-lens camera needs lens points%s");
    LaunchArgs A;
    memset(&A, 0, sizeof(A));
    A.cam = *camera;
    A.trace_key = phase_key(seed, PHASE_TRACE);
    A.pixel_key = phase_key(seed, PHASE_PIXEL);
    A.n = n;
    A.pixel_index = pixel_index;
    A.sample_index = sample_index;
    A.a3 = plane_points;
    A.b3 = lens_points;
    A.out = *out;
    A.counters = reinterpret_cast<unsigned long long*>(counters);
    return launch<MODE_PIXELS>(scene->desc, scene->extended, scene->tree_objects, scene->max_tree_depth, scene->bounces, scene->fused_always, A, flags, stream);
}

int rtamd_trace_rays(rtamd_scene* scene, uint64_t seed, int64_t n, const int32_t* pixel_index,
                     const int32_t* sample_index, const double* origins, const double* directions,
                     int32_t depth, const rtamd_sample_out* out, uint64_t* counters, uint32_t flags,
                     void* stream) {
    if (!scene) return fail(RTAMD_EINVAL, "null scene%s");
    if (int rc = check_device(scene)) return rc;
    if (n < 0) return fail(RTAMD_EINVAL, "negative ray count%s");
    if (n == 0) return RTAMD_OK;
    if (!origins || !directions || !out) return fail(RTAMD_EINVAL, "null ray buffer%s");
    if (scene->two_child && depth < 0) return fail(RTAMD_EINVAL, "negative start depth in a scene with two-child materials%s");
    LaunchArgs A;
    memset(&A, 0, sizeof(A));
    A.trace_key = phase_key(seed, PHASE_TRACE);
    A.pixel_key = phase_key(seed, PHASE_PIXEL);
    A.n = n;
    A.pixel_index = pixel_index;
    A.sample_index = sample_index;
    A.a3 = origins;
    A.b3 = directions;
    A.depth0 = depth;
    A.out = *out;
    A.counters = reinterpret_cast<unsigned long long*>(counters);
    return launch<MODE_RAYS>(scene->desc, scene->extended, scene->tree_objects, scene->max_tree_depth, scene->bounces, scene->fused_always, A, flags, stream);
}

int rtamd_render_tiles(rtamd_scene* scene, const rtamd_camera* camera, uint64_t seed, int64_t n_tiles,
                       const int32_t* tile_ids, int32_t sample_begin, int32_t sample_end,
                       const rtamd_sample_out* per_sample, int64_t* sum_rgb8, double* sum_sq,
                       uint64_t* counters, uint32_t flags, void* stream) {
    int rc = check_camera(scene, camera);
    if (rc) return rc;
    if (n_tiles < 0 || sample_begin < 0 || sample_end < sample_begin) return fail(RTAMD_EINVAL, "bad tile or sample range%s");
    if (n_tiles == 0 || sample_end == sample_begin) return RTAMD_OK;
    if (!tile_ids) return fail(RTAMD_EINVAL, "null tile list%s");
    if (sample_end > scene->samplers_host[camera->sampler].samples_per_pattern)
        return fail(RTAMD_EINVAL, "sample range exceeds the view-plane sampler's pattern size%s");
    LaunchArgs A;
    memset(&A, 0, sizeof(A));
    A.cam = *camera;
    A.trace_key = phase_key(seed, PHASE_TRACE);
    A.pixel_key = phase_key(seed, PHASE_PIXEL);
    A.n_samples = sample_end - sample_begin;
    A.sample_begin = sample_begin;
    A.n = n_tiles * 64 * (int64_t)A.n_samples;
    A.tile_ids = tile_ids;
    A.tiles_x = (camera->width + 7) / 8;
    // several samples of a pixel per wave pay where shadow rays walk a kd-tree from nearby
    // origins, and everywhere through the accumulators: the samples of a pixel are summed inside
    // the wave and added once (glossyreflective 400x400x100: 3.54 -> 3.35 ms, mirrorhall 4K: 129 -> 125 ms)
    A.group_shift = 0;
    for (int g = RT_GROUP_SHIFT_MAX; g > 0; g -= 2)
        if (A.n_samples % (1 << g) == 0) { A.group_shift = g; break; }
    if (per_sample) A.out = *per_sample;
    A.sum_rgb8 = sum_rgb8;
    A.sum_sq = sum_sq;
    A.counters = reinterpret_cast<unsigned long long*>(counters);
    return launch<MODE_TILES>(scene->desc, scene->extended, scene->tree_objects, scene->max_tree_depth, scene->bounces, scene->fused_always, A, flags, stream);
}

int rtamd_pack_sums(int64_t* sum_rgb8, int32_t* wire, int64_t n_values, int32_t clear, void* stream) {
    if (n_values < 0) return fail(RTAMD_EINVAL, "negative count%s");
    if (n_values == 0) return RTAMD_OK;
    if (!sum_rgb8 || !wire) return fail(RTAMD_EINVAL, "null buffer%s");
    const int64_t blocks = (n_values + 255) / 256;
    hipLaunchKernelGGL(pack_sums_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sum_rgb8, wire, n_values, (int)clear);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return RTAMD_OK;
}

}  // extern "C"
