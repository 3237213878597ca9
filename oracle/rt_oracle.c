/*
 * rt_oracle.c - CPU restatement (plain C, IEEE double) of the reference's per-sample path.
 *
 * TEST INFRASTRUCTURE.  This file is the checker the HIP path is compared against and the
 * "port" CPU baseline timed by bench.py; it is never linked into, imported by or called
 * from the product (raytracer_amd/).  Its parity with the reference is pinned by
 * the .npz files of tests/golden, which oracle/make_golden.py produced by running the reference
 * itself (converted to Python 3 in a temp dir) on the same seeded samples.
 *
 * It follows the reference function by function (paths under /root/reference), keeps
 * its recursion (tracer -> material.shade -> tracer) and solves the triangle system the
 * way LAPACK dgesv does (LU with partial pivoting; numpy.linalg.solve at
 * geometry.py:440,453 calls dgesv).  numpy and its LAPACK are not part of the reference
 * tree (no requirements file pins them); the restated algorithm is the textbook
 * unblocked dgetf2 + dgetrs.
 *
 * Input is the scene layout of include/raytracer_amd.h with HOST pointers.
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#ifdef RTO_DEBUG_RAYS
#include <stdio.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/raytracer_amd.h"

#define EPS 1.0e-7                        /* geometry.py:8 */
#define T_INF 9223372036854775807.0       /* sys.maxint, main.py:37 */
#define NO_SHADOW_BIT (1 << 30)
#define PHASE_MUL 0xD6E8FEB86659FD93ull
#define PHASE_PIXEL 2ull
#define PHASE_TRACE 4ull

typedef struct { double x, y, z; } vec;
typedef struct { vec o, d; } ray_t;

typedef struct {
    uint64_t closest, shadow, samples, nodes, leaves, tris, analytic;
} work_t;

typedef struct {
    const rtamd_scene_desc *S;
    uint64_t key;          /* mix(phase_key ^ pixel) */
    uint32_t sample, counter;
    work_t w;
    int first_object, first_prim;
    double first_t;
    int have_first;
} ctx_t;

/* ---------------------------------------------------------------- vectors */
static vec V(double x, double y, double z) { vec r = {x, y, z}; return r; }
static vec P3(const double *p) { return V(p[0], p[1], p[2]); }
static vec add(vec a, vec b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static vec sub(vec a, vec b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static vec neg(vec a) { return V(-a.x, -a.y, -a.z); }
static vec mul(vec a, vec b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static vec scl(vec a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static vec dvd(vec a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static double dot(vec a, vec b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static vec cross(vec a, vec b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static double norm(vec a) { return sqrt(dot(a, a)); }
static double comp(vec a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }
/* Python's min()/max(): keep the first argument unless a later one is strictly beyond it */
static double pymin(double a, double b) { return (b < a) ? b : a; }
static double pymax(double a, double b) { return (b > a) ? b : a; }

/* ---------------------------------------------------------------- draws (raytracer_amd/rng.py) */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t phase_key(uint64_t seed, uint64_t phase) { return mix64(seed ^ (phase * PHASE_MUL)); }
static uint32_t draw_index(ctx_t *c, uint32_t n) {
    uint64_t h = mix64(c->key ^ (((uint64_t)c->sample << 32) | (uint64_t)c->counter));
    c->counter++;
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}
/* sampler.sample_unit_hemisphere_surface (sampler.py:84-85; RegularSampler :13-14 draws nothing) */
static vec draw_hemisphere(ctx_t *c, int sid) {
    const rtamd_sampler *sm = &c->S->samplers[sid];
    uint32_t i = sm->regular ? 0u : draw_index(c, (uint32_t)(sm->num_patterns * sm->samples_per_pattern));
    return P3(c->S->sampler_data + sm->hemi_offset + 3 * (int64_t)i);
}
/* random.choice(sampler.sample())  (geometry.py:379) */
static void draw_square(ctx_t *c, int sid, double *sx, double *sy) {
    const rtamd_sampler *sm = &c->S->samplers[sid];
    uint32_t pat = sm->regular ? 0u : draw_index(c, (uint32_t)sm->num_patterns);
    uint32_t el = draw_index(c, (uint32_t)sm->samples_per_pattern);
    const double *p = c->S->sampler_data + sm->square_offset + 2 * ((int64_t)pat * sm->samples_per_pattern + el);
    *sx = p[0];
    *sy = p[1];
}

/* ---------------------------------------------------------------- triangle: dgesv on a 3x3 */
/* Solves M x = rhs with M = [col0 col1 col2]; returns 0 if a pivot is exactly zero
 * (dgesv info > 0 -> numpy raises LinAlgError -> treated as a miss). */
static int solve3(vec c0, vec c1, vec c2, vec rhs, double x[3]) {
    double a[3][3] = {{c0.x, c1.x, c2.x}, {c0.y, c1.y, c2.y}, {c0.z, c1.z, c2.z}};
    double b[3] = {rhs.x, rhs.y, rhs.z};
    int j, i, k;
    for (j = 0; j < 3; ++j) {
        int p = j;                                  /* idamax: first largest |a(i,j)|, i >= j */
        double big = fabs(a[j][j]);
        for (i = j + 1; i < 3; ++i)
            if (fabs(a[i][j]) > big) { big = fabs(a[i][j]); p = i; }
        if (a[p][j] == 0.0) return 0;
        if (p != j) {
            for (k = 0; k < 3; ++k) { double t = a[j][k]; a[j][k] = a[p][k]; a[p][k] = t; }
            { double t = b[j]; b[j] = b[p]; b[p] = t; }
        }
        {
            double r = 1.0 / a[j][j];               /* dgetf2 scales by the reciprocal */
            for (i = j + 1; i < 3; ++i) a[i][j] *= r;
        }
        for (i = j + 1; i < 3; ++i)                 /* rank-1 update of the trailing block */
            for (k = j + 1; k < 3; ++k) a[i][k] -= a[i][j] * a[j][k];
    }
    for (j = 0; j < 3; ++j)                         /* dgetrs: L y = P b (unit lower) */
        for (i = j + 1; i < 3; ++i) b[i] -= b[j] * a[i][j];
    for (j = 2; j >= 0; --j) {                      /* U x = y */
        b[j] = b[j] / a[j][j];
        for (i = 0; i < j; ++i) b[i] -= b[j] * a[i][j];
    }
    x[0] = b[0]; x[1] = b[1]; x[2] = b[2];
    return 1;
}

/* Triangle.hit / shadow_hit acceptance (geometry.py:435-442, :448-455); g = v0, v1-v0, v2-v0
 * with the nine components `stride` doubles apart (1 for a packed triangle, n_refs for a row
 * of the scene's struct-of-arrays table) */
static int tri_test(const double *g, int64_t stride, const ray_t *r, double *alpha, double *beta, double *t) {
    double x[3];
    vec v0 = V(g[0], g[stride], g[2 * stride]);
    vec d = V(g[3 * stride], g[4 * stride], g[5 * stride]);
    vec e = V(g[6 * stride], g[7 * stride], g[8 * stride]);
    vec f = sub(v0, r->o);
    if (!solve3(d, e, neg(r->d), neg(f), x)) return 0;
    *alpha = x[0]; *beta = x[1]; *t = x[2];
    return x[0] > EPS && x[1] > EPS && x[2] > EPS && x[0] + x[1] < 1;
}

/* ---------------------------------------------------------------- kd-tree (kdtree.py) */
typedef struct { int32_t node; double close, far; } kd_entry;

/* BoundingBox.hit, geometry.py:49-70 */
static int bbox_hit(const rtamd_tree *T, const ray_t *r, double *tmin, double *tmax) {
    double dx = 1. / r->d.x, dy = 1. / r->d.y, dz = 1. / r->d.z;
    double t1 = (T->bbox_lo[0] - r->o.x) * dx, t2 = (T->bbox_hi[0] - r->o.x) * dx;
    double t3 = (T->bbox_lo[1] - r->o.y) * dy, t4 = (T->bbox_hi[1] - r->o.y) * dy;
    double t5 = (T->bbox_lo[2] - r->o.z) * dz, t6 = (T->bbox_hi[2] - r->o.z) * dz;
    *tmin = pymax(pymax(pymin(t1, t2), pymin(t3, t4)), pymin(t5, t6));
    *tmax = pymin(pymin(pymax(t1, t2), pymax(t3, t4)), pymax(t5, t6));
    if (*tmax < 0) return 0;
    if (*tmin > *tmax) return 0;
    return 1;
}

/* Generator KDTree.iterate_nodes (kdtree.py:133-155) as an explicit iterator. */
typedef struct {
    const rtamd_scene_desc *S;
    const ray_t *r;
    kd_entry stack[128];
    int sp;
} kd_iter;

static void kd_begin(kd_iter *it, const rtamd_scene_desc *S, const rtamd_tree *T, const ray_t *r) {
    double c, f;
    it->S = S; it->r = r; it->sp = 0;
    if (bbox_hit(T, r, &c, &f)) {
        it->stack[0].node = T->node_offset; it->stack[0].close = c; it->stack[0].far = f;
        it->sp = 1;
    }
}

/* Diagnostic: leaf visits and leaf items by leaf size (bucket = bit length of the count).
 * Only meaningful for single-threaded runs; read and reset with rto_leaf_histogram. */
static uint64_t g_leaf_hist[32][2];

static void note_leaf(int32_t count) {
    int b = 0;
    int32_t c = count;
    while (c) { b++; c >>= 1; }
    g_leaf_hist[b][0] += 1;
    g_leaf_hist[b][1] += (uint64_t)count;
}

void rto_leaf_histogram(uint64_t *out, int reset) {
    memcpy(out, g_leaf_hist, sizeof(g_leaf_hist));
    if (reset) memset(g_leaf_hist, 0, sizeof(g_leaf_hist));
}

/* next leaf: returns 1 and fills first/count, or 0 when the walk is over */
static int kd_next(kd_iter *it, work_t *w, int32_t *first, int32_t *count) {
    int32_t node;
    double close_d, far_d;
    if (it->sp == 0) return 0;
    it->sp--;
    node = it->stack[it->sp].node; close_d = it->stack[it->sp].close; far_d = it->stack[it->sp].far;
    for (;;) {
        const rtamd_kdnode *nd = &it->S->kd_nodes[node];
        int ax;
        double o, d, t;
        int32_t near_n, far_n;
        if ((nd->b & 3) == 3) { *first = nd->a; *count = nd->b >> 2; w->leaves++; note_leaf(*count); return 1; }
        w->nodes++;
        ax = nd->b;
        o = comp(it->r->o, ax); d = comp(it->r->d, ax);
        t = (nd->split - o) / d;
        if (nd->split >= o) { near_n = node + 1; far_n = nd->a; }
        else { near_n = nd->a; far_n = node + 1; }
        if (t >= far_d || t < 0) node = near_n;
        else if (t <= close_d) node = far_n;
        else {
            it->stack[it->sp].node = far_n; it->stack[it->sp].close = t; it->stack[it->sp].far = far_d;
            it->sp++;
            node = near_n;
            far_d = t;
        }
    }
}

/* ---------------------------------------------------------------- objects */
typedef struct {
    double t;
    vec n, p;
    vec local;                  /* ShadeRecord.local_hit_point, in the object's own space */
    int material, object, prim;
} hit_t;

static ray_t inv_ray_one(const double *m, const double *n, const ray_t *r) {       /* geometry.py:97-101 */
    ray_t q;
    q.o = V(((m[0] * r->o.x + m[1] * r->o.y) + m[2] * r->o.z) + m[3],
            ((m[4] * r->o.x + m[5] * r->o.y) + m[6] * r->o.z) + m[7],
            ((m[8] * r->o.x + m[9] * r->o.y) + m[10] * r->o.z) + m[11]);
    q.d = V((n[0] * r->d.x + n[1] * r->d.y) + n[2] * r->d.z,
            (n[3] * r->d.x + n[4] * r->d.y) + n[5] * r->d.z,
            (n[6] * r->d.x + n[7] * r->d.y) + n[8] * r->d.z);
    return q;
}

/* Instance.inv_ray through every level of a nested Instance, outermost first (each level's hit()
 * calls the wrapped object's, geometry.py:103-105) */
static ray_t inv_ray(const rtamd_scene_desc *S, const rtamd_object *ob, const ray_t *r) {
    ray_t q = inv_ray_one(ob->inv_point, ob->inv_dir, r);
    int j;
    for (j = 0; j < ob->n_inner; ++j) {
        const rtamd_transform *t = &S->transforms[ob->inner_offset + j];
        q = inv_ray_one(t->inv_point, t->inv_dir, &q);
    }
    return q;
}

static vec normal_out(const double *m, vec n) {                                    /* geometry.py:107 */
    return V((m[0] * n.x + m[3] * n.y) + m[6] * n.z, (m[1] * n.x + m[4] * n.y) + m[7] * n.z,
             (m[2] * n.x + m[5] * n.y) + m[8] * n.z);
}

static int sphere_root(const double *p, const ray_t *r, vec *temp, double *t) {   /* geometry.py:254-269 */
    double a, b, c, disc, e, denom;
    *temp = sub(r->o, P3(p));
    a = dot(r->d, r->d);
    b = 2.0 * dot(*temp, r->d);
    c = dot(*temp, *temp) - p[3] * p[3];
    disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return 0;
    e = sqrt(disc);
    denom = 2.0 * a;
    *t = (-b - e) / denom;
    if (!(*t > EPS)) *t = (-b + e) / denom;
    return *t > EPS;
}

static int plane_root(vec point, vec normal, const ray_t *r, double *t) {          /* geometry.py:317-322 */
    double dn = dot(r->d, normal);
    if (dn == 0) return 0;
    *t = dot(sub(point, r->o), normal) / dn;
    return *t > EPS;
}

static int rectangle_root(const double *p, const ray_t *r, double *t) {            /* geometry.py:388-399 */
    vec diff;
    double a, b;
    if (!plane_root(P3(p), P3(p + 3), r, t)) return 0;
    diff = sub(add(r->o, scl(r->d, *t)), P3(p));
    a = fabs(dot(diff, P3(p + 6)));
    b = fabs(dot(diff, P3(p + 9)));
    return a * 2 <= p[12] && b * 2 <= p[13];
}

static int box_closest(const double *p, const ray_t *r, double *tmin, vec *normal) { /* geometry.py:136-217 */
    static const double FACE[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double dx = 1. / r->d.x, dy = 1. / r->d.y, dz = 1. / r->d.z;
    double tx_min, tx_max, ty_min, ty_max, tz_min, tz_max, t0, t1;
    int face_in, face_out, face;
    if (dx >= 0) { tx_min = (p[0] - r->o.x) * dx; tx_max = (p[3] - r->o.x) * dx; }
    else { tx_min = (p[3] - r->o.x) * dx; tx_max = (p[0] - r->o.x) * dx; }
    if (dy >= 0) { ty_min = (p[1] - r->o.y) * dy; ty_max = (p[4] - r->o.y) * dy; }
    else { ty_min = (p[4] - r->o.y) * dy; ty_max = (p[1] - r->o.y) * dy; }
    if (dz >= 0) { tz_min = (p[2] - r->o.z) * dz; tz_max = (p[5] - r->o.z) * dz; }
    else { tz_min = (p[5] - r->o.z) * dz; tz_max = (p[2] - r->o.z) * dz; }
    if (tx_min > ty_min) { t0 = tx_min; face_in = dx >= 0 ? 0 : 3; }
    else { t0 = ty_min; face_in = dy >= 0 ? 1 : 4; }
    if (tz_min > t0) { t0 = tz_min; face_in = dz >= 0 ? 2 : 5; }
    if (tx_max < ty_max) { t1 = tx_max; face_out = dx >= 0 ? 3 : 0; }
    else { t1 = ty_max; face_out = dy >= 0 ? 4 : 1; }
    if (tz_max < t1) { t1 = tz_max; face_out = dz >= 0 ? 5 : 2; }
    if (t0 < t1 && t1 > EPS) {
        if (t0 > EPS) { *tmin = t0; face = face_in; }
        else { *tmin = t1; face = face_out; }
        *normal = P3(FACE[face]);
        return 1;
    }
    return 0;
}

static int box_shadow(const double *p, const ray_t *r, double *t) {                /* geometry.py:219-242 */
    double dx = 1. / r->d.x, dy = 1. / r->d.y, dz = 1. / r->d.z;
    double t1 = (p[0] - r->o.x) * dx, t2 = (p[3] - r->o.x) * dx;
    double t3 = (p[1] - r->o.y) * dy, t4 = (p[4] - r->o.y) * dy;
    double t5 = (p[2] - r->o.z) * dz, t6 = (p[5] - r->o.z) * dz;
    double tmin = pymax(pymax(pymin(t1, t2), pymin(t3, t4)), pymin(t5, t6));
    double tmax = pymin(pymin(pymax(t1, t2), pymax(t3, t4)), pymax(t5, t6));
    if (tmax < EPS) return 0;
    if (tmin > tmax) return 0;
    *t = tmin;
    return 1;
}

static void bare_triangle(const double *p, double g[9]) {
    int i;
    for (i = 0; i < 3; ++i) { g[i] = p[i]; g[3 + i] = p[3 + i] - p[i]; g[6 + i] = p[6 + i] - p[i]; }
}

/* obj.hit(ray) for one entry of world.objects; Instance handled here (geometry.py:103-108) */
static int object_hit(ctx_t *c, int idx, const ray_t *world_ray, hit_t *h) {
    const rtamd_scene_desc *S = c->S;
    const rtamd_object *ob = &S->objects[idx];
    ray_t local = ob->has_transform ? inv_ray(S, ob, world_ray) : *world_ray;
    const ray_t *r = &local;
    const double *p = ob->p;
    double t = 0;
    vec n = V(0, 0, 0);
    int mat = ob->material, prim = -1, hit = 0;
    if (ob->kind != RTAMD_KDTREE) c->w.analytic++;
    switch (ob->kind) {
    case RTAMD_SPHERE: {
        vec temp;
        hit = sphere_root(p, r, &temp, &t);
        if (hit) n = scl(dvd(add(temp, scl(r->d, t)), p[3]), p[4]);     /* geometry.py:270, :306 */
    } break;
    case RTAMD_PLANE:
        hit = plane_root(P3(p), P3(p + 3), r, &t);
        n = P3(p + 3);
        break;
    case RTAMD_CHECKER_PLANE:                                             /* geometry.py:347-363 */
        hit = plane_root(P3(p), P3(p + 3), r, &t);
        n = P3(p + 3);
        if (hit) {
            vec rel = sub(add(r->o, scl(r->d, t)), P3(p));
            double x = round(dot(rel, P3(p + 6)) / p[12]);
            double y = round(dot(rel, P3(p + 9)) / p[12]);
            if (fmod(x + y, 2.0) != 0) mat = ob->alt_material;
        }
        break;
    case RTAMD_RECTANGLE:
        hit = rectangle_root(p, r, &t);
        n = P3(p + 3);
        break;
    case RTAMD_BOX:
        hit = box_closest(p, r, &t, &n);
        break;
    case RTAMD_TRIANGLE: {
        double g[9], a, b;
        bare_triangle(p, g);
        hit = tri_test(g, 1, r, &a, &b, &t);
        if (hit) {
            if (ob->smooth) n = add(add(scl(P3(p + 12), 1 - a - b), scl(P3(p + 15), a)), scl(P3(p + 18), b));
            else n = P3(p + 9);
        }
    } break;
    case RTAMD_KDTREE: {                                                  /* KDTree.hit, kdtree.py:166-174 */
        const rtamd_tree *T = &S->trees[ob->tree];
        kd_iter it;
        int32_t first, count, i, best_ref = -1;
        double best_t = 0, best_a = 0, best_b = 0;
        kd_begin(&it, S, T, r);
        while (kd_next(&it, &c->w, &first, &count)) {
            for (i = 0; i < count; ++i) {
                double a, b, tt;
                c->w.tris++;
                if (tri_test(S->tri_geom + (int64_t)(first + i), S->n_refs, r, &a, &b, &tt)) {
                    if (best_ref < 0 || best_t > tt) { best_t = tt; best_a = a; best_b = b; best_ref = first + i; }
                }
            }
            if (best_ref >= 0) break;
        }
        if (best_ref >= 0) {
            int64_t gp;
            const double *nn;
            hit = 1;
            t = best_t;
            prim = S->tri_prim[best_ref] & (NO_SHADOW_BIT - 1);
            gp = (int64_t)T->prim_offset + prim;
            nn = S->prim_normals + S->prim_normal_offset[gp];
            if (T->smooth)                                                /* geometry.py:460 */
                n = add(add(scl(P3(nn), 1 - best_a - best_b), scl(P3(nn + 3), best_a)), scl(P3(nn + 6), best_b));
            else n = P3(nn);
            mat = S->prim_material[gp];
        }
    } break;
    default: break;
    }
    if (!hit) return 0;
    if (ob->has_transform) {                                              /* geometry.py:107, innermost level first */
        int j;
        for (j = ob->n_inner - 1; j >= 0; --j) n = normal_out(S->transforms[ob->inner_offset + j].inv_dir, n);
        n = normal_out(ob->inv_dir, n);
    }
    h->t = t; h->n = n; h->material = mat; h->object = idx; h->prim = prim;
    h->local = add(r->o, scl(r->d, t));                                   /* geometry.py:210, :323, :456 ... */
    if (ob->kind == RTAMD_SPHERE) h->local = sub(h->local, P3(p));        /* geometry.py:271 */
    return 1;
}

/* obj.shadow_hit(ray) */
static int object_shadow(ctx_t *c, int idx, const ray_t *world_ray, double *t) {
    const rtamd_scene_desc *S = c->S;
    const rtamd_object *ob = &S->objects[idx];
    ray_t local = ob->has_transform ? inv_ray(S, ob, world_ray) : *world_ray;
    const ray_t *r = &local;
    const double *p = ob->p;
    if (ob->kind == RTAMD_KDTREE) {                                       /* kdtree.py:157-164 */
        kd_iter it;
        int32_t first, count, i;
        kd_begin(&it, S, &S->trees[ob->tree], r);
        while (kd_next(&it, &c->w, &first, &count)) {
            for (i = 0; i < count; ++i) {
                double a, b;
                if (S->tri_prim[first + i] & NO_SHADOW_BIT) continue;     /* geometry.py:433-434 */
                c->w.tris++;
                if (tri_test(S->tri_geom + (int64_t)(first + i), S->n_refs, r, &a, &b, t)) return 1;
            }
        }
        return 0;
    }
    if (!ob->cast_shadow) return 0;
    c->w.analytic++;
    switch (ob->kind) {
    case RTAMD_SPHERE: { vec temp; return sphere_root(p, r, &temp, t); }
    case RTAMD_PLANE:
    case RTAMD_CHECKER_PLANE: return plane_root(P3(p), P3(p + 3), r, t);
    case RTAMD_RECTANGLE: return rectangle_root(p, r, t);
    case RTAMD_BOX: return box_shadow(p, r, t);
    case RTAMD_TRIANGLE: {
        double g[9], a, b;
        bare_triangle(p, g);
        return tri_test(g, 1, r, &a, &b, t);
    }
    default: return 0;
    }
}

/* World.hit_objects, main.py:35-46 */
static int hit_objects(ctx_t *c, const ray_t *r, hit_t *best) {
    double tmin = T_INF;
    int found = 0, i;
    c->w.closest++;
#ifdef RTO_DEBUG_RAYS
    fprintf(stderr, "closest %.17g %.17g %.17g %.17g %.17g %.17g\n", r->o.x, r->o.y, r->o.z, r->d.x, r->d.y, r->d.z);
#endif
    for (i = 0; i < c->S->n_objects; ++i) {
        hit_t h;
        if (object_hit(c, i, r, &h) && h.t < tmin) {
            *best = h;
            tmin = h.t;
            best->p = add(r->o, scl(r->d, tmin));
            found = 1;
        }
    }
    return found;
}

/* light.in_shadow: with_limit -> `is_hit and t < d` (light.py:31-37, :124-130), else any hit */
static int in_shadow(ctx_t *c, const ray_t *r, int with_limit, double d) {
    int i;
    c->w.shadow++;
    for (i = 0; i < c->S->n_objects; ++i) {
        double t;
        if (object_shadow(c, i, r, &t) && (!with_limit || t < d)) return 1;
    }
    return 0;
}

/* ---------------------------------------------------------------- lights and materials */
static vec about_normal(vec w, vec s) {                                   /* light.py:86-89, :153-156, :76 */
    vec v = cross(w, V(0.0072, 1.0, 0.0034)), u;
    v = dvd(v, norm(v));
    u = cross(v, w);
    return add(add(scl(u, s.x), scl(v, s.y)), scl(w, s.z));
}

static vec ambient_L(ctx_t *c, const hit_t *h) {
    const rtamd_ambient *a = &c->S->ambient;
    ray_t sr;
    if (a->kind == RTAMD_AMBIENT_CONSTANT) return P3(a->radiance);        /* light.py:14-15 */
    sr.o = h->p;                                                          /* light.py:85-100 */
    sr.d = about_normal(h->n, draw_hemisphere(c, a->sampler));
    return in_shadow(c, &sr, 0, 0) ? P3(a->occluded) : P3(a->radiance);
}

static vec glossy_f(vec spec, double e, vec n, vec wo, vec wi) {          /* material.py:88-97 */
    double ndotwi = dot(n, wi);
    vec r = add(neg(wi), scl(scl(n, 2.0), ndotwi));
    double rdotwo = dot(r, wo);
    if (rdotwo > 0.0) return scl(spec, pow(rdotwo, e));
    return V(0, 0, 0);
}

/* ImageTexture.get_color with SphericalMap.get_texel_coordinates, texture.py:33-68.  Python
 * raises where acos leaves its domain or the texel lies outside the image; a NaN colour here. */
static vec texture_color(const rtamd_scene_desc *S, int tex, vec local) {
    const rtamd_texture *T = &S->textures[tex];
    const double pi = 3.141592653589793;
    double theta = acos(local.y), phi = atan2(local.x, local.z), u, v;
    int column, row;
    const uint8_t *px;
    if (phi < 0.) phi += pi * 2;
    u = phi / pi / 2;
    v = 1 - theta / pi;
    if (!(theta == theta)) return V(NAN, NAN, NAN);
    column = (int)((T->hres - 1) * u);
    row = (int)((T->vres - 1) * v);
    if (column < 0 || column >= T->width || row < 0 || row >= T->height) return V(NAN, NAN, NAN);
    px = S->texels + T->offset + 3 * ((int64_t)row * T->width + column);
    return V(px[0] / 256., px[1] / 256., px[2] / 256.);
}

/* Matte/Phong shade and area_light_shade: material.py:251-281, :346-377; SV_Matte.shade :311-326 */
static vec direct(ctx_t *c, const rtamd_material *m, const hit_t *h, vec wo, int area) {
    const rtamd_scene_desc *S = c->S;
    int textured = m->kind == RTAMD_SV_MATTE;
    vec ka = P3(m->ambient), kd = P3(m->diffuse), L;
    int phong = m->kind != RTAMD_MATTE && !textured;
    int gate = (phong || area) ? m->receives_shadow : 1;
    int li;
    if (textured) ka = mul(ka, texture_color(S, m->texture, h->local));   /* material.py:24-25 */
    L = mul(ka, ambient_L(c, h));
    for (li = 0; li < S->n_lights; ++li) {
        const rtamd_light *lt = &S->lights[li];
        vec wi, sp = V(0, 0, 0), f, Ll, term;
        double ndotwi;
        int shadowed = 0;
        if (lt->kind == RTAMD_POINT) {                                    /* light.py:27-29 */
            vec v = sub(P3(lt->position), h->p);
            wi = dvd(v, norm(v));
        } else if (lt->kind == RTAMD_DIRECTIONAL) {                       /* light.py:54-55 */
            wi = P3(lt->position);
        } else if (lt->kind == RTAMD_AREA) {                              /* light.py:116-122 */
            double sx, sy;
            vec v;
            draw_square(c, lt->sampler, &sx, &sy);
            sp = add(add(P3(lt->position), scl(P3(lt->edge_u), sx)), scl(P3(lt->edge_v), sy));
            v = sub(sp, h->p);
            wi = dvd(v, norm(v));
        } else {                                                          /* light.py:152-165 */
            wi = about_normal(h->n, draw_hemisphere(c, lt->sampler));
        }
        ndotwi = dot(h->n, wi);
        if (!(ndotwi > 0)) continue;
        if (textured && !(dot(h->n, wo) > 0.)) continue;                  /* material.py:317-319 */
        if (lt->cast_shadow) {
            ray_t sr;
            sr.o = h->p; sr.d = wi;
            if (lt->kind == RTAMD_POINT) shadowed = in_shadow(c, &sr, 1, norm(sub(sr.o, P3(lt->position))));
            else if (lt->kind == RTAMD_AREA) shadowed = in_shadow(c, &sr, 1, dot(sub(sp, sr.o), sr.d));
            else shadowed = in_shadow(c, &sr, 0, 0);
        }
        if (shadowed || !gate) continue;
        f = kd;
        if (textured)                                                     /* material.py:27-28: one lookup per light */
            f = dvd(mul(kd, texture_color(S, m->texture, h->local)), 3.141592653589793);
        if (phong) f = add(f, glossy_f(P3(m->specular), m->exp, h->n, wo, wi));
        Ll = P3(lt->radiance);
        if (lt->kind == RTAMD_POINT && lt->attenuation != 0)              /* light.py:39-44 */
            Ll = scl(Ll, pow(norm(sub(h->p, P3(lt->position))) / lt->radius, -lt->attenuation));
        if (lt->kind == RTAMD_AREA && !(dot(P3(lt->normal), wi) < 0.)) Ll = V(0, 0, 0);   /* light.py:110-114 */
        term = scl(mul(f, Ll), ndotwi);
        if (area) {
            double G = 1., pdf = 1.;
            if (lt->kind == RTAMD_AREA) {                                 /* light.py:132-140 */
                double ndotd = -dot(P3(lt->normal), wi);
                vec dv = sub(sp, h->p);
                double d2 = (((0 + dv.x * dv.x) + dv.y * dv.y) + dv.z * dv.z) / (lt->radius * lt->radius);
                G = ndotd / pymax(1., d2);
                pdf = lt->pdf;
            }
            term = dvd(scl(term, G), pdf);
        }
        L = add(L, term);
    }
    return L;
}

static vec trace_ray_t(ctx_t *c, ray_t ray, int depth, double *tmin_out);
static vec trace_ray(ctx_t *c, ray_t ray, int depth) { return trace_ray_t(c, ray, depth, NULL); }

/* refraction set-up shared by PerfectTransmitter (material.py:135-159) and the Fresnel pair
 * (:170-223): flips the normal and picks the relative index by the side the ray arrives on */
static int refraction(const rtamd_material *m, vec n, vec d, vec *nf, double *eta, double *cos_i, double *temp) {
    double c = -dot(n, d);
    *nf = n;
    *eta = m->eta_front;
    if (c < 0.) { c = -c; *nf = neg(n); *eta = m->eta_back; }
    *cos_i = c;
    *temp = 1. - (1. - c * c) / (*eta * *eta);
    return *temp < 0.;                       /* total internal reflection */
}

static vec powv(vec b, double e) { return V(pow(b.x, e), pow(b.y, e), pow(b.z, e)); }

/* material.shade / area_light_shade / path_shade for the hit's material */
static vec shade(ctx_t *c, const hit_t *h, const ray_t *ray, int depth) {
    const rtamd_scene_desc *S = c->S;
    const rtamd_material *m = &S->materials[h->material];
    vec wo = neg(ray->d), L;
    ray_t child;
    if (m->kind == RTAMD_EMISSIVE) {
        if (S->tracer_kind == RTAMD_TRACER_SHADE || S->tracer_kind == RTAMD_TRACER_WHITTED)
            return V(1., 1., 1.);                                                          /* material.py:495-496 */
        if (S->tracer_kind == RTAMD_TRACER_AREA)
            return (-dot(h->n, ray->d) > 0) ? P3(m->emit) : V(0, 0, 0);                    /* material.py:498-502 */
        return (dot(h->n, ray->d) < 0) ? P3(m->emit) : V(0, 0, 0);                         /* material.py:504-508 */
    }
    if (S->tracer_kind == RTAMD_TRACER_PATH) {                                             /* material.py:41-53, :283-289 */
        vec v = cross(V(0.0034, 1, 0.0071), h->n), u, sp, wi;
        double pdf, ndotwi;
        v = dvd(v, norm(v));
        u = cross(v, h->n);
        sp = draw_hemisphere(c, m->sampler);
        wi = add(add(scl(u, sp.x), scl(v, sp.y)), scl(h->n, sp.z));
        wi = dvd(wi, norm(wi));
        pdf = dot(h->n, wi) / 3.141592653589793;
        ndotwi = dot(h->n, wi);
        child.o = h->p; child.d = wi;
        return dvd(scl(mul(P3(m->diffuse), trace_ray(c, child, depth + 1)), ndotwi), pdf);
    }
    if (S->tracer_kind == RTAMD_TRACER_AREA) return direct(c, m, h, wo, 1);
    L = direct(c, m, h, wo, 0);
    if (m->kind == RTAMD_REFLECTIVE) {                                                     /* material.py:69-74, :385-394 */
        double ndotwo = dot(h->n, wo), pdf;
        vec wi = add(neg(wo), scl(scl(h->n, 2.0), ndotwo)), fr;
        pdf = dot(h->n, wi);
        fr = dvd(P3(m->reflect), pdf);
        child.o = h->p; child.d = wi;
        L = add(L, scl(mul(fr, trace_ray(c, child, depth + 1)), dot(h->n, wi)));
    } else if (m->kind == RTAMD_GLOSSY_REFLECTIVE) {                                       /* material.py:102-121, :402-411 */
        double ndotwo = dot(h->n, wo), lobe, pdf;
        vec w = add(neg(wo), scl(scl(h->n, 2.0), ndotwo));
        vec u = cross(V(0.00424, 1., 0.00767), w), v, sp, wi, fr;
        u = dvd(u, norm(u));
        v = cross(u, w);
        sp = draw_hemisphere(c, m->sampler);
        wi = add(add(scl(u, sp.x), scl(v, sp.y)), scl(w, sp.z));
        if (dot(h->n, wi) < 0.0) wi = add(sub(scl(u, -sp.x), scl(v, sp.y)), scl(w, sp.z));
        lobe = pow(dot(w, wi), m->reflect_exp);
        pdf = lobe * dot(h->n, wi);
        if (pdf == 0) { fr = V(0, 0, 0); pdf = 1; }
        else fr = scl(P3(m->reflect), lobe);
        child.o = h->p; child.d = wi;
        L = add(L, dvd(scl(mul(fr, trace_ray(c, child, depth + 1)), dot(h->n, wi)), pdf));
    } else if (m->kind == RTAMD_TRANSPARENT) {                                             /* material.py:420-437 */
        double ndotwo = dot(h->n, wo), pdf, eta, cos_i, temp;
        vec wi = add(neg(wo), scl(scl(h->n, 2.0), ndotwo)), fr, nf, Lr;
        int tir;
        pdf = dot(h->n, wi);
        fr = dvd(P3(m->reflect), pdf);
        child.o = h->p; child.d = wi;
        Lr = trace_ray(c, child, depth + 1);
        tir = refraction(m, h->n, ray->d, &nf, &eta, &cos_i, &temp);
        if (tir) {
            L = add(L, Lr);
        } else {
            double cos_t = pow(temp, 0.5), ft;
            vec wt = sub(dvd(neg(wo), eta), scl(nf, cos_t - cos_i / eta));
            ft = m->kt / (eta * eta) / fabs(dot(h->n, wt));
            L = add(L, mul(scl(Lr, dot(h->n, wi)), fr));
            child.d = wt;
            L = add(L, scl(scl(trace_ray(c, child, depth + 1), ft), fabs(dot(h->n, wt))));
        }
    } else if (m->kind == RTAMD_DIELECTRIC) {                                              /* material.py:448-483 */
        double eta, cos_i, temp, cos_t, r_par, r_perp, kr, ndotwo = dot(h->n, wo), ndotwi, tr = 0, tt = 0, fr;
        vec nf, wi, Lr;
        int tir = refraction(m, h->n, ray->d, &nf, &eta, &cos_i, &temp);
        cos_t = pow(temp, 0.5);                                                            /* NaN under TIR, unused then */
        r_par = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
        r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
        kr = .5 * (r_par * r_par + r_perp * r_perp);
        wi = add(neg(wo), scl(scl(h->n, 2.0), ndotwo));
        ndotwi = dot(h->n, wi);
        fr = kr / fabs(ndotwi);
        child.o = h->p; child.d = wi;
        if (tir) {
            Lr = trace_ray_t(c, child, depth + 1, &tr);
            L = add(L, mul(Lr, powv(ndotwi < 0. ? P3(m->filter_in) : P3(m->filter_out), tr)));
        } else {
            vec wt = sub(dvd(neg(wo), eta), scl(nf, cos_t - cos_i / eta)), Lt;
            double ft = (1 - kr) / (eta * eta) / fabs(dot(h->n, wt)), ndotwt = dot(h->n, wt);
            vec cr = ndotwi < 0. ? P3(m->filter_in) : P3(m->filter_out);
            vec ct = ndotwi < 0. ? P3(m->filter_out) : P3(m->filter_in);
            Lr = trace_ray_t(c, child, depth + 1, &tr);
            child.d = wt;
            Lt = trace_ray_t(c, child, depth + 1, &tt);
            L = add(L, scl(mul(scl(Lr, fr), powv(cr, tr)), fabs(ndotwi)));
            L = add(L, scl(mul(scl(Lt, ft), powv(ct, tt)), fabs(ndotwt)));
        }
    }
    return L;
}

/* Tracer / WhittedTracer / AreaLightTracer / PathTracer .trace_ray: tracer.py:26-35, :43-58,
 * :65-71, :79-88.  tmin_out is WhittedTracer's tminlist: the hit distance, 0 without a hit. */
static vec trace_ray_t(ctx_t *c, ray_t ray, int depth, double *tmin_out) {
    const rtamd_scene_desc *S = c->S;
    int shade_kind = S->tracer_kind == RTAMD_TRACER_SHADE || S->tracer_kind == RTAMD_TRACER_WHITTED;
    vec miss = shade_kind ? P3(S->background) : V(0, 0, 0);
    hit_t h;
    if (tmin_out) *tmin_out = 0;
    if (S->tracer_kind != RTAMD_TRACER_AREA && depth > S->depth_limit) return miss;
    if (!hit_objects(c, &ray, &h)) return miss;
    if (tmin_out) *tmin_out = h.t;
    if (!c->have_first) {
        c->have_first = 1; c->first_object = h.object; c->first_prim = h.prim; c->first_t = h.t;
    }
    return shade(c, &h, &ray, depth);
}

/* ---------------------------------------------------------------- camera (camera.py) */
static ray_t camera_ray(const rtamd_camera *cam, vec pp, vec lens) {
    vec u = P3(cam->u), v = P3(cam->v), w = P3(cam->w), d;
    ray_t r;
    if (cam->thin_lens) {                                                 /* camera.py:222-225 */
        vec f = dvd(scl(pp, cam->focal_plane_distance), cam->viewing_distance);
        d = sub(add(scl(u, f.x - lens.x), scl(v, f.y - lens.y)), scl(w, cam->focal_plane_distance));
        r.o = lens;
    } else {                                                              /* camera.py:31-33 */
        d = sub(add(scl(u, pp.x), scl(v, pp.y)), scl(w, cam->viewing_distance));
        r.o = P3(cam->eye);
    }
    r.d = dvd(d, norm(d));
    return r;
}

static int32_t quantise(double cval) {                                    /* int(c*255), camera.py:66 */
    double s = cval * 255;
    if (!(s == s)) return 0;
    if (s >= 2147483647.0) return 2147483647;
    if (s <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)s;
}

static void store(const rtamd_sample_out *out, int64_t k, const ctx_t *c, vec col) {
    if (!out) return;
    if (out->rgb) { out->rgb[3 * k] = col.x; out->rgb[3 * k + 1] = col.y; out->rgb[3 * k + 2] = col.z; }
    if (out->rgb8) { out->rgb8[3 * k] = quantise(col.x); out->rgb8[3 * k + 1] = quantise(col.y); out->rgb8[3 * k + 2] = quantise(col.z); }
    if (out->hit_object) out->hit_object[k] = c->have_first ? c->first_object : -1;
    if (out->hit_prim) out->hit_prim[k] = c->have_first ? c->first_prim : -1;
    if (out->hit_t) out->hit_t[k] = c->have_first ? c->first_t : 0;
    if (out->n_rays) { out->n_rays[2 * k] = (int32_t)c->w.closest; out->n_rays[2 * k + 1] = (int32_t)c->w.shadow; }
}

static void begin_sample(ctx_t *c, const rtamd_scene_desc *S, uint64_t trace_key, int32_t pixel, int32_t sample) {
    memset(c, 0, sizeof(*c));
    c->S = S;
    c->key = mix64(trace_key ^ (uint64_t)(uint32_t)pixel);
    c->sample = (uint32_t)sample;
    c->first_object = c->first_prim = -1;
}

static void merge(uint64_t *counters, const work_t *w) {
    if (!counters) return;
#pragma omp critical
    {
        counters[0] += w->closest; counters[1] += w->shadow; counters[2] += w->samples;
        counters[3] += w->nodes; counters[4] += w->leaves; counters[5] += w->tris; counters[6] += w->analytic;
    }
}

static void accumulate(work_t *a, const work_t *b) {
    a->closest += b->closest; a->shadow += b->shadow; a->samples += 1; a->nodes += b->nodes;
    a->leaves += b->leaves; a->tris += b->tris; a->analytic += b->analytic;
}

/* ---------------------------------------------------------------- entry points (host buffers) */
int rto_abi_version(void) { return RTAMD_ABI_VERSION; }

/* camera.render_pixels, camera.py:60-67 / :227-234.  Samples are handed to threads in
 * chunks of 1000, the job size of main.py:92. */
int rto_render_pixels(const rtamd_scene_desc *S, const rtamd_camera *cam, uint64_t seed, int64_t n,
                      const int32_t *pixel_index, const int32_t *sample_index, const double *plane_points,
                      const double *lens_points, const rtamd_sample_out *out, uint64_t *counters, int threads) {
    uint64_t trace_key = phase_key(seed, PHASE_TRACE);
    int64_t k;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        work_t w;
        memset(&w, 0, sizeof(w));
#pragma omp for schedule(dynamic, 1000)
        for (k = 0; k < n; ++k) {
            ctx_t c;
            vec lens = cam->thin_lens ? P3(lens_points + 3 * k) : V(0, 0, 0);
            vec col;
            begin_sample(&c, S, trace_key, pixel_index[k], sample_index[k]);
            col = trace_ray(&c, camera_ray(cam, P3(plane_points + 3 * k), lens), 0);
            store(out, k, &c, col);
            accumulate(&w, &c.w);
        }
        merge(counters, &w);
    }
    return 0;
}

/* world.tracer.trace_ray(ray, depth) */
int rto_trace_rays(const rtamd_scene_desc *S, uint64_t seed, int64_t n, const int32_t *pixel_index,
                   const int32_t *sample_index, const double *origins, const double *directions, int32_t depth,
                   const rtamd_sample_out *out, uint64_t *counters, int threads) {
    uint64_t trace_key = phase_key(seed, PHASE_TRACE);
    int64_t k;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        work_t w;
        memset(&w, 0, sizeof(w));
#pragma omp for schedule(dynamic, 1000)
        for (k = 0; k < n; ++k) {
            ctx_t c;
            ray_t r;
            vec col;
            begin_sample(&c, S, trace_key, pixel_index ? pixel_index[k] : 0, sample_index ? sample_index[k] : 0);
            r.o = P3(origins + 3 * k);
            r.d = P3(directions + 3 * k);
            col = trace_ray(&c, r, depth);
            store(out, k, &c, col);
            accumulate(&w, &c.w);
        }
        merge(counters, &w);
    }
    return 0;
}

/* get_pixel_plane_point_pairs + render_pixels over whole 8x8 tiles (camera.py:69-98, :236-267),
 * with the sample numbering and accumulators documented for rtamd_render_tiles. */
int rto_render_tiles(const rtamd_scene_desc *S, const rtamd_camera *cam, uint64_t seed, int64_t n_tiles,
                     const int32_t *tile_ids, int32_t sample_begin, int32_t sample_end,
                     const rtamd_sample_out *per_sample, int64_t *sum_rgb8, double *sum_sq,
                     uint64_t *counters, int threads) {
    uint64_t trace_key = phase_key(seed, PHASE_TRACE), pixel_key = phase_key(seed, PHASE_PIXEL);
    int32_t ns = sample_end - sample_begin, tiles_x = (cam->width + 7) / 8;
    int64_t n = n_tiles * 64 * (int64_t)ns, k;
    const rtamd_sampler *sm = &S->samplers[cam->sampler];
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
    work_t wsum;
    memset(&wsum, 0, sizeof(wsum));
#pragma omp for schedule(dynamic, 1000)
    for (k = 0; k < n; ++k) {
        int32_t pix = (int32_t)(k & 63), soff = (int32_t)((k >> 6) % ns);
        int64_t tpos = (k >> 6) / ns;
        int32_t tile = tile_ids[tpos];
        int32_t col = (tile % tiles_x) * 8 + (pix & 7), row = (tile / tiles_x) * 8 + (pix >> 3);
        int32_t pixel, sample;
        ctx_t pc, c;
        uint32_t pat, dpat;
        const double *sq;
        vec pp, lens = V(0, 0, 0), colr;
        if (col >= cam->width || row >= cam->height) continue;
        pixel = row * cam->width + col;
        sample = sample_begin + soff;
        memset(&pc, 0, sizeof(pc));
        pc.S = S;
        pc.key = mix64(pixel_key ^ (uint64_t)(uint32_t)pixel);
        pat = sm->regular ? 0u : draw_index(&pc, (uint32_t)sm->num_patterns);            /* camera.py:78 */
        sq = S->sampler_data + sm->square_offset + 2 * ((int64_t)pat * sm->samples_per_pattern + sample);
        pp = V(cam->pixel_size * (col - cam->width / 2 + sq[0]),                          /* camera.py:92-93 */
               cam->pixel_size * (row - cam->height / 2 + sq[1]), 0.0);
        if (cam->thin_lens) {                                                             /* camera.py:246, :262 */
            const double *dk;
            dpat = sm->regular ? 0u : draw_index(&pc, (uint32_t)sm->num_patterns);
            dk = S->sampler_data + sm->disk_offset + 2 * ((int64_t)dpat * sm->samples_per_pattern + sample);
            lens = add(P3(cam->eye), scl(V(dk[0], dk[1], 0.0), cam->lens_radius));
        }
        begin_sample(&c, S, trace_key, pixel, sample);
        colr = trace_ray(&c, camera_ray(cam, pp, lens), 0);
        store(per_sample, k, &c, colr);
        if (sum_rgb8 || sum_sq) {
            int64_t rowi = 3 * (tpos * 64 + pix);
#pragma omp critical(rto_accum)
            {
                if (sum_rgb8) { sum_rgb8[rowi] += quantise(colr.x); sum_rgb8[rowi + 1] += quantise(colr.y); sum_rgb8[rowi + 2] += quantise(colr.z); }
                if (sum_sq) { sum_sq[rowi] += colr.x * colr.x; sum_sq[rowi + 1] += colr.y * colr.y; sum_sq[rowi + 2] += colr.z * colr.z; }
            }
        }
        accumulate(&wsum, &c.w);
    }
    merge(counters, &wsum);
    }
    return 0;
}
