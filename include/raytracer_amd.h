/*
 * raytracer_amd.h - C ABI of the MI355X (gfx950) ray tracing hot path.
 *
 * The reference (/root/reference) is pure Python and has no FFI; the functions below
 * are what a binding for its per-sample path would bind.  Each entry point names the
 * reference seam it stands behind.  Conventions:
 *
 *   - every function returns 0 on success and a negative RTAMD_E* code on failure;
 *     rtamd_last_error() returns a message for the calling thread's last failure;
 *   - scene tables and sample/ray/result buffers are DEVICE pointers owned by the
 *     caller (e.g. torch tensors' data_ptr()); descriptors (structs below) are HOST
 *     memory and are copied during the call;
 *   - launches are asynchronous on the hipStream_t passed as `stream` (a void* here so
 *     that the header needs no HIP include); results are valid once that stream has
 *     been synchronised;
 *   - nothing is allocated per call; a scene handle owns one small device block that
 *     lives until rtamd_scene_destroy;
 *   - a handle belongs to the device that was current in rtamd_scene_create; calls on
 *     one handle must not overlap in time on different streams if they share outputs.
 *
 * All geometry and shading arithmetic is IEEE double ("f64"), like the reference's
 * Python floats.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_ABI_VERSION 7

enum {
    RTAMD_OK = 0,
    RTAMD_EINVAL = -1,      /* bad argument / inconsistent descriptor          */
    RTAMD_EUNSUPPORTED = -2,/* valid scene the device path does not cover     */
    RTAMD_EHIP = -3,        /* a HIP runtime call failed                      */
    RTAMD_ENODEVICE = -4    /* no gfx950 device available                     */
};

/* primitive kinds of world.objects entries (geometry.py classes) */
enum {
    RTAMD_SPHERE = 0,        /* Sphere / ConcaveSphere   geometry.py:245-307 */
    RTAMD_PLANE = 1,         /* Plane                    geometry.py:310-336 */
    RTAMD_CHECKER_PLANE = 2, /* CheckerPlane             geometry.py:339-363 */
    RTAMD_RECTANGLE = 3,     /* Rectangle                geometry.py:365-416 */
    RTAMD_BOX = 4,           /* AxisAlignedBox           geometry.py:115-242 */
    RTAMD_TRIANGLE = 5,      /* Triangle                 geometry.py:419-473 */
    RTAMD_KDTREE = 6         /* KDTree                   kdtree.py:55-174    */
};

/* material kinds (material.py) */
enum {
    RTAMD_MATTE = 0, RTAMD_PHONG = 1, RTAMD_REFLECTIVE = 2,
    RTAMD_GLOSSY_REFLECTIVE = 3, RTAMD_EMISSIVE = 4,
    RTAMD_TRANSPARENT = 5,   /* material.py:414-437: mirror child + refracted child    */
    RTAMD_DIELECTRIC = 6,    /* material.py:440-483: Fresnel-weighted children, colour
                                filter cf^t over the child's path length               */
    RTAMD_SV_MATTE = 7       /* material.py:292-326: Matte whose colour is a texture lookup
                                (`ambient` holds ka, `diffuse` holds kd, both broadcast) */
};

/* texture mappings (texture.py:53-68) */
enum { RTAMD_MAP_SPHERICAL = 0 };

/* light kinds (light.py) */
enum { RTAMD_POINT = 0, RTAMD_DIRECTIONAL = 1, RTAMD_AREA = 2, RTAMD_ENVIRONMENT = 3 };
enum { RTAMD_AMBIENT_CONSTANT = 0, RTAMD_AMBIENT_OCCLUDER = 1 };

/* tracer kinds (tracer.py:21-88) */
enum {
    RTAMD_TRACER_SHADE = 0, RTAMD_TRACER_AREA = 1, RTAMD_TRACER_PATH = 2,
    RTAMD_TRACER_WHITTED = 3 /* tracer.py:38-58: as SHADE, and reports the hit distance
                                to the calling material (Dielectric needs it)          */
};

/* Transparent and Dielectric spawn two child rays per hit (material.py:428,436,473-474).
 * The device keeps the not-yet-traced sibling of every level of one sample's ray tree in
 * a per-sample list, so scenes holding such materials need depth_limit + 1 <= this. */
#define RTAMD_MAX_BRANCH_LEVELS 8

/* One entry of world.objects, in list order (main.py:35-46 scans them in that order and
 * the earlier object wins ties).  `p` holds the primitive's parameters:
 *   SPHERE     center[3], radius, normal_sign
 *   PLANE      point[3], normal[3]
 *   CHECKER    point[3], normal[3], right[3], up[3], grid_size
 *   RECTANGLE  p0[3], normal[3], right[3], top[3], a, b
 *   BOX        lo[3], hi[3]
 *   TRIANGLE   v0[3], v1[3], v2[3], normal[3], n0[3], n1[3], n2[3] (n* only if smooth)
 *   KDTREE     -  (see `tree`)
 * With has_transform the object is an Instance (geometry.py:85-112): `inv_point` is the
 * upper 3x4 of inverse(M) row-major, `inv_dir` is inverse(M[:3,:3]) row-major (the
 * reference inverts the two separately).  An Instance may wrap another Instance: `inv_point` /
 * `inv_dir` then belong to the outermost one and the n_inner further levels, from the outside
 * in, are transforms[inner_offset .. inner_offset + n_inner); a ray passes through them in that
 * order, a normal comes back through them in the opposite one. */
typedef struct rtamd_object {
    int32_t kind;
    int32_t material;       /* index into materials; -1 for KDTREE                */
    int32_t alt_material;   /* CheckerPlane's second material                     */
    int32_t cast_shadow;
    int32_t has_transform;
    int32_t tree;           /* index into trees for KDTREE, else -1               */
    int32_t smooth;         /* TRIANGLE: vertex normals present                   */
    int32_t n_inner;        /* Instance levels inside the outermost one (usually 0) */
    double p[30];
    double inv_point[12];
    double inv_dir[9];
    int32_t inner_offset;   /* first of the n_inner entries in transforms          */
    int32_t pad_;           /* keeps sizeof a multiple of 16                       */
} rtamd_object;

/* One inner level of a nested Instance (see rtamd_object). */
typedef struct rtamd_transform {
    double inv_point[12];
    double inv_dir[9];
    double pad_[3];
} rtamd_transform;

typedef struct rtamd_material {
    int32_t kind;
    int32_t receives_shadow;
    int32_t sampler;        /* glossy lobe / path tracing hemisphere sampler, or -1 */
    int32_t texture;        /* SV_MATTE: index into textures, else -1               */
    double ambient[3];      /* ka*cd            material.py:55-56                 */
    double diffuse[3];      /* kd*cd/pi         material.py:38-39                 */
    double specular[3];     /* ks*cs            material.py:95                    */
    double exp;
    double reflect[3];      /* kr*cd            material.py:74 / :121             */
    double reflect_exp;
    double emit[3];         /* ls*color         material.py:492-493               */
    double kt;              /* Transparent: PerfectTransmitter.kt  material.py:133 */
    /* relative index of refraction for a ray that arrives on the side the normal points
     * to (n.wo >= 0) and for one that arrives from the other side:
     *   Transparent  ior, 1/ior              material.py:147-152
     *   Dielectric   eta_in/eta_out, eta_out/eta_in   material.py:174-177 */
    double eta_front;
    double eta_back;
    double filter_in[3];    /* Dielectric cf_in / cf_out  material.py:445-446      */
    double filter_out[3];
} rtamd_material;

typedef struct rtamd_light {
    int32_t kind;
    int32_t cast_shadow;
    int32_t sampler;        /* AREA / ENVIRONMENT                                  */
    int32_t reserved;
    double radiance[3];     /* ls*color, or the emitter's Le                       */
    double position[3];     /* POINT location, DIRECTIONAL direction, AREA corner  */
    double attenuation;
    double radius;
    double edge_u[3];       /* AREA: sample = corner + sx*edge_u + sy*edge_v       */
    double edge_v[3];
    double normal[3];       /* AREA emitter normal                                 */
    double pdf;             /* AREA: 1/area as the shape reports it                */
} rtamd_light;

typedef struct rtamd_ambient {
    int32_t kind;
    int32_t sampler;        /* OCCLUDER hemisphere sampler                         */
    double radiance[3];     /* ls*color                                            */
    double occluded[3];     /* 0.1*ls*color    light.py:97-98                      */
} rtamd_ambient;

/* One sampler object's tables inside `sampler_data` (doubles):
 *   square:  num_patterns * samples_per_pattern * 2   (sampler.py:31)
 *   disk:    same shape                                (sampler.py:34-61)
 *   hemi:    num_patterns * samples_per_pattern * 3   (sampler.py:63-76)
 * `regular` marks a RegularSampler, whose methods consume no random draw. */
typedef struct rtamd_sampler {
    int32_t num_patterns;
    int32_t samples_per_pattern;
    int32_t regular;
    int32_t reserved;
    int64_t square_offset;
    int64_t disk_offset;
    int64_t hemi_offset;
} rtamd_sampler;

/* 16-byte kd node; nodes of a tree are stored in preorder, so the left child of node i
 * is node i+1.  Interior: split plane, a = index of the right child, b = axis (0..2).
 * Leaf: b & 3 == 3, a = first leaf reference, b >> 2 = reference count; `split` of a leaf
 * is the index of its first chunk box when the leaf has them (see chunk_boxes), else < 0.
 * Node and reference indices are absolute (into kd_nodes / tri_* of the scene). */
typedef struct rtamd_kdnode {
    double split;
    int32_t a;
    int32_t b;
} rtamd_kdnode;

typedef struct rtamd_tree {
    double bbox_lo[3];
    double bbox_hi[3];
    int32_t node_offset;    /* root node index in kd_nodes                         */
    int32_t ref_offset;     /* first leaf reference of this tree in tri_* arrays   */
    int32_t prim_offset;    /* first primitive of this tree in prim_* arrays       */
    int32_t smooth;         /* vertex normals present: prim_normals holds 9/prim   */
    int32_t depth;          /* maximum node depth of the tree proper (the reference's)  */
    int32_t shadow_casters; /* what bit 30 of this tree's tri_prim rows says, in one word:
                               RTAMD_CASTERS_ALL (no row has it), _NONE (every row has it:
                               a mesh with cast_shadow off, geometry.py:433-434) or _MIXED */
    int32_t shadow_root;    /* 0, or the index in kd_nodes / kd_triples (past the trees proper)
                               of the root of another kd-tree over the same triangles, any
                               shape, with leaf_aabb rows for its leaves: shadow queries that
                               any hit decides (no limit, or the ray leaves bbox before its
                               limit) walk this tree instead, in single precision: a
                               triangle must be listed in every leaf whose cell its box,
                               widened by 4e-5 of the largest coordinate of bbox, reaches
                               into, and the leaves' leaf_aabb rows must be widened
                               alike; ignored without leaf_aabb                     */
    int32_t walk_depth;     /* levels a walk may pass through: `depth`, or more where a shadow
                               tree or the shadow subtrees of leaves (leaf_aabb) are deeper;
                               bounds the traversal stack                            */
} rtamd_tree;
enum { RTAMD_CASTERS_ALL = 0, RTAMD_CASTERS_NONE = 1, RTAMD_CASTERS_MIXED = 2 };

/* An ImageTexture (texture.py:11-46): 8-bit RGB texels, row-major from the top row as
 * PIL's getpixel((column, row)) addresses them, at texels[offset + 3*(row*width + column)].
 * hres/vres are the extents the mapping scales (u, v) by; they equal width/height unless the
 * scene gave its own (texture.py:26-29) and never exceed them.  A colour is texel/256. */
typedef struct rtamd_texture {
    int32_t width, height;
    int32_t hres, vres;
    int32_t mapping;
    int32_t reserved;
    int64_t offset;         /* first byte of this image in texels                  */
} rtamd_texture;

/* Everything a render needs that does not change between frames.
 * Leaf-ordered triangle tables have one row per leaf reference (a triangle that
 * straddles a split plane appears once per leaf that holds it, kdtree.py:52):
 *   tri_geom[9][n_refs] = v0.xyz, (v1-v0).xyz, (v2-v0).xyz as nine arrays of n_refs doubles
 *                       (component c of reference r is tri_geom[c*n_refs + r]), so the 64
 *                       lanes of a wave load 64 consecutive references with 9 coalesced reads
 *   tri_prim[ref]     = primitive index within its tree; bit 30 set if the triangle
 *                       does not cast shadows (geometry.py:433-434)
 *   tri_order[ref]    = position of the row in the reference's scan order (absolute
 *                       reference index).  Rows of a leaf may be stored in any order;
 *                       ties in t and "first item that hits" are decided by this value.
 *                       NULL means rows are stored in scan order.
 *   chunk_boxes[6][n_chunks] = lo.xyz, hi.xyz of each 64-row chunk of the leaves that carry
 *                       boxes (conservative: every triangle of the chunk lies inside);
 *                       may be NULL with n_chunks = 0
 *   leaf_boxes[n_chunks][9][6] = the same chunk box followed by the boxes of the chunk's eight
 *                       8-row cells (unused cells zero), and
 *   tri_rows[..][10]  = row-major copy of tri_geom with {tri_prim, tri_order} packed into
 *                       the tenth double: what a single lane reads when it scans a boxed
 *                       leaf for its own ray.  Both optional (NULL: the wave scans such
 *                       leaves together, chunk by chunk).  It needs rows for the leaves of
 *                       the trees proper only, not for those of shadow trees (rtamd_tree).
 * Per-primitive tables (one row per triangle, all trees concatenated):
 *   prim_normals      = unit face normal [3], or n0,n1,n2 [9] for smooth trees; rows
 *                       are addressed through prim_normal_offset[prim] (in doubles)
 *   prim_material[prim]
 */
typedef struct rtamd_scene_desc {
    int32_t abi_version;
    int32_t tracer_kind;
    int32_t depth_limit;
    int32_t n_objects;
    int32_t n_materials;
    int32_t n_lights;
    int32_t n_samplers;
    int32_t n_trees;
    int64_t n_kd_nodes;
    int64_t n_refs;
    int64_t n_prims;
    int64_t n_sampler_doubles;
    int64_t n_chunks;
    double background[3];
    rtamd_ambient ambient;
    const rtamd_object *objects;        /* device */
    const rtamd_material *materials;    /* device */
    const rtamd_light *lights;          /* device */
    const rtamd_sampler *samplers;      /* device */
    const double *sampler_data;         /* device */
    const rtamd_tree *trees;            /* device */
    const rtamd_kdnode *kd_nodes;       /* device */
    const rtamd_kdnode *kd_triples;     /* device: [n_kd_nodes][4] = node i, its left child, its right
                                           child (zeros for a leaf), one unused record: what a
                                           traversal step reads, two tree levels per 64-byte load */
    const double *tri_geom;             /* device */
    const int32_t *tri_prim;            /* device */
    const int32_t *tri_order;           /* device, optional */
    const double *tri_rows;             /* device, optional: [rows of the trees proper][10] row-major copy of tri_geom,
                                           tenth double = {tri_prim, tri_order} as two int32 */
    const double *leaf_boxes;           /* device, optional: [n_chunks][9][6] = per chunk its own
                                           box (as chunk_boxes) and the boxes of its eight
                                           8-row cells, lo.xyz hi.xyz each */
    const double *chunk_boxes;          /* device, optional */
    const double *prim_normals;         /* device */
    const int64_t *prim_normal_offset;  /* device */
    const int32_t *prim_material;       /* device */
    int32_t n_textures;
    int32_t n_transforms;
    int64_t n_texel_bytes;
    const rtamd_texture *textures;      /* device, NULL when n_textures == 0 */
    const uint8_t *texels;              /* device */
    const float *chunk_tree;            /* device, optional: [n_chunks][12] floats = per chunk its box
                                           (as chunk_boxes, rounded outward to single precision:
                                           it only culls) and, for every fourth chunk of a leaf
                                           counted from the leaf's first, the box around that
                                           chunk and the next three of the same leaf (zeros
                                           elsewhere): what a lane reads when it sorts out which
                                           chunks of a leaf its own shadow ray can touch.  NULL:
                                           the wave tests the chunk boxes of one ray at a time */
    const float *leaf_aabb;             /* device, optional: [n_kd_nodes][8] floats = for every leaf
                                           that holds rows the box around its triangles, lo.xyz
                                           hi.xyz, then one int32 and one unused float
                                           (conservative, rounded outward); zeros for other
                                           nodes.  A ray that crosses the leaf's cell but misses
                                           this box skips the leaf.  The int32, when > 0, is the
                                           index in kd_nodes / kd_triples of the root of a
                                           kd-tree over this leaf's triangles alone (nodes past
                                           the trees proper, leaves of rows appended to the
                                           triangle tables, each with a box row of its own):
                                           shadow queries that any hit decides walk on into
                                           it instead of testing the leaf.  rtamd_tree.walk_depth
                                           must then count the levels such a walk can pass */
    const rtamd_transform *transforms;  /* device, NULL when n_transforms == 0: inner levels of nested
                                           Instances */
} rtamd_scene_desc;

/* Camera and view plane (camera.py:12-33, :142-146, :222-225; tracer.py:91-95). */
typedef struct rtamd_camera {
    double eye[3], u[3], v[3], w[3];
    double viewing_distance;
    double lens_radius;
    double focal_plane_distance;
    double pixel_size;
    int32_t thin_lens;
    int32_t width, height;
    int32_t sampler;        /* view-plane sampler index                            */
} rtamd_camera;

/* Per-sample results; any pointer may be NULL to skip that output.
 *   rgb[n][3]     radiance before quantisation (what trace_ray returns)
 *   rgb8[n][3]    int(c*255) per channel, as camera.py:66 emits (not clamped)
 *   hit_object[n] index into world.objects of the primary hit, -1 on a miss
 *   hit_prim[n]   triangle index within its tree (face index after quad split), else -1
 *   hit_t[n]      primary tmin
 *   n_rays[n][2]  closest-hit queries (main.py:35 calls) and shadow queries
 *                 (light in_shadow calls) spent on the sample */
typedef struct rtamd_sample_out {
    double *rgb;
    int32_t *rgb8;
    int32_t *hit_object;
    int32_t *hit_prim;
    double *hit_t;
    int32_t *n_rays;
} rtamd_sample_out;

/* Launch-wide counters, added to by every launch that is given the block:
 *   [0] closest-hit queries   [1] shadow queries      [2] samples traced
 * With flags & RTAMD_COUNT_WORK the launch follows the REFERENCE algorithm step by step (no
 * culling shortcut that would change the numbers) and adds what kdtree.py / geometry.py
 * would spend on the same rays:
 *   [3] kd interior steps     [4] kd leaves visited   [5] triangle tests
 *   [6] analytic object tests
 * With flags & RTAMD_COUNT_EXECUTED the launch runs the same code as an uncounted one and adds
 * what that code EXECUTES (per lane, summed):
 *   [3] kd interior steps     [4] kd leaves reached   [5] ray-triangle tests run
 *   [6] analytic object tests [7] leaves passed by because the ray misses the box of
 *   their triangles           [8] node records loaded (a node and both children, 48 B)
 *   [9] culling boxes tested  [10] triangle rows loaded (one row may be tested against
 *   several rays)             [11] bytes requested by the loads behind [6], [8]-[10]
 * The two flags exclude each other.  [12]-[15] are reserved. */
#define RTAMD_N_COUNTERS 16
#define RTAMD_COUNT_WORK 1u
#define RTAMD_COUNT_EXECUTED 2u

typedef struct rtamd_scene rtamd_scene;

/* Library / device check.  Fails with RTAMD_ENODEVICE when no HIP device is usable. */
int rtamd_abi_version(void);
int rtamd_device_check(void);
const char *rtamd_last_error(void);

/* Scene lifetime.  Stands where World.__init__ leaves a built, immutable world
 * (main.py:13-22); validates the descriptor (counts, index ranges read on the host are
 * limited to the descriptor itself - table contents are trusted). */
int rtamd_scene_create(const rtamd_scene_desc *desc, rtamd_scene **out);
int rtamd_scene_destroy(rtamd_scene *scene);

/* camera.render_pixels(world, pairs)  (camera.py:60-67, :227-234): trace n samples whose
 * view-plane points the caller chose.  pixel_index[n] = row*width+col and sample_index[n]
 * key the in-trace random draws; plane_points[n][3]; lens_points[n][3] for a thin-lens
 * camera (NULL otherwise). */
int rtamd_render_pixels(rtamd_scene *scene, const rtamd_camera *camera, uint64_t seed,
                        int64_t n, const int32_t *pixel_index, const int32_t *sample_index,
                        const double *plane_points, const double *lens_points,
                        const rtamd_sample_out *out, uint64_t *counters, uint32_t flags,
                        void *stream);

/* world.tracer.trace_ray(ray, depth)  (tracer.py:26-35, :65-71, :79-88): trace n explicit
 * rays starting at recursion depth `depth`. */
int rtamd_trace_rays(rtamd_scene *scene, uint64_t seed, int64_t n,
                     const int32_t *pixel_index, const int32_t *sample_index,
                     const double *origins, const double *directions, int32_t depth,
                     const rtamd_sample_out *out, uint64_t *counters, uint32_t flags,
                     void *stream);

/* Whole-frame sample generation + trace for a list of 8x8-pixel tiles: the work of
 * get_pixel_plane_point_pairs + render_pixels (camera.py:69-98, :236-267, :60-67) for the
 * pixels of those tiles and sample indices [sample_begin, sample_end) of each pixel's
 * pattern.  tile_ids[n_tiles] index tiles row-major over ceil(width/8) x ceil(height/8).
 * Row k of the per-sample outputs is (tile k / (64*ns), sample k / 64 % ns, pixel-in-tile
 * k % 64), pixel-in-tile = 8*dy + dx, however the kernel groups samples into waves; rows of
 * pixels outside the image are not written.
 *   per_sample: optional per-sample outputs in that order (may be NULL)
 *   sum_rgb8[n_tiles*64][3]  int64: += int(c*255) per sample (numerator of parallel mode's
 *                             running mean, main.py:99-103), row = 64*position of the tile
 *                             in tile_ids + pixel-in-tile
 *   sum_sq[n_tiles*64][3]    double: += c*c per sample (camera.py:55, :135), same rows
 * Either accumulator may be NULL; both are added to, so the caller zeroes them and may
 * split a frame's samples over several calls with the same tile list. */
int rtamd_render_tiles(rtamd_scene *scene, const rtamd_camera *camera, uint64_t seed,
                       int64_t n_tiles, const int32_t *tile_ids,
                       int32_t sample_begin, int32_t sample_end,
                       const rtamd_sample_out *per_sample,
                       int64_t *sum_rgb8, double *sum_sq,
                       uint64_t *counters, uint32_t flags, void *stream);

/* A worker's results on their way to the parent (main.py:88-90 puts them on the pixel queue,
 * main.py:99-110 adds them up): between a rank's rtamd_render_tiles and the gather that brings its
 * sums to rank 0.  wire[i] = sum_rgb8[i] saturated to int32, for i < n_values (a sum only leaves the
 * int32 range when samples were infinite, and such a pixel is clamped to 0 or 255 on arrival,
 * main.py:104-106); with clear != 0 sum_rgb8 is zeroed in the same pass, ready for the next frame.
 * Device pointers, asynchronous on `stream`. */
int rtamd_pack_sums(int64_t *sum_rgb8, int32_t *wire, int64_t n_values, int32_t clear, void *stream);

/* ---- device-side kd-tree build ------------------------------------------------------
 * KDTree(BoundingBoxes(boxes))  (kdtree.py:8-24, :27-52, :56-69, :92-131): builds, on the
 * device, exactly the tree the reference's recursion builds from n axis-aligned boxes
 * (box_lo[n][3], box_hi[n][3], device, in the order of the list given to BoundingBoxes).
 * The tree comes back in preorder (the left child of node i is node i+1):
 *   axis[i]   0/1/2 for an interior node (KDTree.split_idx), -1 for a leaf
 *   split[i]  KDTree.split_center of an interior node
 *   left[i], right[i]   child node indices (-1 for a leaf)
 *   first[i], count[i]  a leaf's slice of leaf_refs (0, 0 for an interior node)
 *   leaf_refs           box indices, every leaf's list in the order KDTree.hit scans it
 * All output arrays are device memory owned by the caller, max_nodes / max_refs entries long.
 * The call is synchronous on `stream` (it reads a few totals back after every tree level). */
typedef struct rtamd_kd_tree_out {
    int32_t *axis;
    double *split;
    int32_t *left, *right, *first, *count;
    int32_t *leaf_refs;
    int64_t max_nodes, max_refs;
} rtamd_kd_tree_out;

/* Bytes of device workspace rtamd_kd_build needs when no tree level holds more than
 * level_capacity box references (straddling boxes are referenced by both children,
 * kdtree.py:52; 4 * n_boxes has been enough for every mesh of the reference). */
int rtamd_kd_build_workspace_bytes(int64_t level_capacity, int64_t max_nodes, int64_t max_refs,
                                   size_t *bytes);

/* Fails with RTAMD_EUNSUPPORTED (and leaves the outputs undefined) when a level outgrows
 * level_capacity or the tree outgrows max_nodes / max_refs.  n_nodes, n_refs and depth
 * (maximum node depth, root = 0) are host pointers. */
int rtamd_kd_build(const double *box_lo, const double *box_hi, int64_t n_boxes,
                   const rtamd_kd_tree_out *out, void *workspace, size_t workspace_bytes,
                   int64_t level_capacity, int64_t *n_nodes, int64_t *n_refs, int32_t *depth,
                   void *stream);

/* The same build without a workspace argument and without any allocation.  It has no
 * counterpart in the reference; it exists for callers that cannot allocate device memory
 * (serving loops and harnesses that forbid hipMalloc, set-up code that runs where allocation is
 * not possible).  Like rtamd_kd_build it synchronises `stream` and copies a few totals to the host
 * after every tree level, so it cannot itself be captured into a HIP graph.
 * The buffers live in a static array of the library's code object (11 MiB, held by every GPU
 * the library runs on until the process ends), which limits it to n_boxes <= 8192, 24576
 * references per tree level, 32768 nodes and 32768 leaf references; larger inputs fail with
 * RTAMD_EUNSUPPORTED and a message that names the limit.  All calls on one GPU share that
 * array: calls that overlap in time on the same GPU race and must be serialised by the caller.
 * Inputs, outputs and results are as for rtamd_kd_build. */
int rtamd_kd_build_static(const double *box_lo, const double *box_hi, int64_t n_boxes,
                          const rtamd_kd_tree_out *out, int64_t *n_nodes, int64_t *n_refs,
                          int32_t *depth, void *stream);

/* Message of the calling thread's last failed rtamd_kd_* call. */
const char *rtamd_kd_last_error(void);

/* Builds a shadow tree (rtamd_tree.shadow_root) on the host: a kd-tree over n_boxes triangle
 * boxes (box_lo / box_hi [n_boxes][3], host memory, already widened as rtamd_tree.shadow_root
 * asks) by the surface-area heuristic over the box faces; a box that straddles a plane goes to
 * both sides; a node of at most leaf_rows boxes, or one that no plane improves, becomes a leaf.
 * No counterpart in the reference, whose kdtree.py builds the tree its own walk needs; this is
 * the tree for shadow queries that any hit decides.
 * With tri_corners ([n_boxes][9] = v0, v1, v2, host memory; may be NULL) a triangle whose box
 * reaches across a plane is clipped against either side's cell, widened by `slack` (the absolute
 * amount the boxes were widened by): a side the triangle itself does not come within `slack` of
 * does not list it, and the box it brings to the other side is that of the part inside, widened by
 * `slack` again.  Candidate planes and leaf boxes then follow the geometry inside each cell.
 * Nodes come back in preorder (left child = next node): node_axis 0..2 with node_split and
 * node_link = index of the right child, or node_axis -1 for a leaf whose number is node_link;
 * leaf k holds the boxes leaf_row[leaf_offset[k] .. leaf_offset[k+1]) and, if leaf_box is given,
 * leaf_box[k][6] = lo.xyz hi.xyz around what those rows bring to the leaf (zeros for an empty one;
 * max_nodes rows).  With the output arrays NULL only the sizes are worked out (n_nodes, n_leaves,
 * n_rows, depth = levels).
 * RTAMD_EUNSUPPORTED: within max_depth levels some leaf would keep more than max_leaf_rows;
 * RTAMD_EINVAL also for coordinates that are not finite numbers, and when the host runs out of memory. */
int rtamd_shadow_tree_build(const double *box_lo, const double *box_hi, int64_t n_boxes, int32_t max_depth,
                            int32_t leaf_rows, int32_t max_leaf_rows, double step_cost,
                            const double *tri_corners, double slack,
                            int32_t *node_axis, double *node_split, int32_t *node_link, int64_t max_nodes,
                            int64_t *leaf_offset, int32_t *leaf_row, double *leaf_box, int64_t max_rows,
                            int64_t *n_nodes, int64_t *n_leaves, int64_t *n_rows, int32_t *depth);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
