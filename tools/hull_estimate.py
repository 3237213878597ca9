#!/usr/bin/env python3
"""How much of a shadow ray's way through the dragon's box lies outside a hull of a few more plane pairs around the
mesh (normals with components -1 / 0 / 1), for sampled occlusion and light rays of the benchmark scene - worked out
on the host.  python tools/hull_estimate.py"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import raytracer_amd
    raytracer_amd.activate()
    from mesh_parser import read_mesh
    os.chdir(raytracer_amd.ASSET_DIR)
    with open("meshes/dragon.obj") as f:
        mesh = read_mesh(f, None)
    V = np.array(mesh.vertices, dtype=np.float64)
    F = np.array(mesh.indices, dtype=np.int64)[:, :3]
    rng = np.random.RandomState(1)
    dirs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) > (0, 0, 0)]
    N = np.array(dirs, dtype=np.float64)
    lo, hi = (V @ N.T).min(axis=0), (V @ N.T).max(axis=0)
    order = np.array([int(np.abs(d).sum()) for d in N])
    sets = {"box (6 planes)": order == 1, "+ edge diagonals (18)": order <= 2, "+ corner diagonals only (14)": order != 2,
            "all 26": order >= 1}

    def interval(o, d, sel):
        no, nd = o @ N[sel].T, d @ N[sel].T
        with np.errstate(divide="ignore", invalid="ignore"):
            t1, t2 = (lo[sel] - no) / nd, (hi[sel] - no) / nd
        tn, tf = np.fmin(t1, t2), np.fmax(t1, t2)
        par = nd == 0
        inside = (no >= lo[sel]) & (no <= hi[sel])
        tn = np.where(par, np.where(inside, -np.inf, np.inf), tn)
        tf = np.where(par, np.where(inside, np.inf, -np.inf), tf)
        return np.maximum(tn.max(axis=1), 0), tf.min(axis=1)

    def hemi(n, k):
        u, v = rng.rand(k), rng.rand(k)
        r, phi = np.sqrt(u), 2 * np.pi * v
        t = np.cross(n, np.array([0.0072, 1, 0.0034]))
        t /= np.linalg.norm(t, axis=-1, keepdims=True)
        b = np.cross(t, n)
        return (r * np.cos(phi))[:, None] * b + (r * np.sin(phi))[:, None] * t + np.sqrt(1 - u)[:, None] * n

    def report(name, o, d):
        tn6, tf6 = interval(o, d, sets["box (6 planes)"])
        in6 = tf6 > tn6
        print("%s: %.3f of the rays enter the box" % (name, in6.mean()))
        for label, sel in sets.items():
            tn, tf = interval(o, d, sel)
            inn = in6 & (tf > tn)
            print("    %-30s still walked %.3f, way inside %.2f (box %.2f)" % (
                label, inn[in6].mean(), np.where(inn, tf - tn, 0)[in6].mean(), (tf6 - tn6)[in6].mean()))

    k = 200000
    o = np.stack([rng.uniform(-12, 12, k), np.zeros(k), rng.uniform(-6, 8, k)], axis=1)
    report("occlusion rays from the ground", o, hemi(np.tile(np.array([0., 1., 0.]), (k, 1)), k))
    light = np.array([-5., 7., 9.]) + np.stack([rng.uniform(0, 6, k), rng.uniform(0, 4, k), np.zeros(k)], axis=1)
    dl = light - o
    report("light rays from the ground", o, dl / np.linalg.norm(dl, axis=1, keepdims=True))
    tri = rng.randint(0, len(F), k)
    w = rng.dirichlet((1, 1, 1), k)
    p = V[F[tri, 0]] * w[:, 0:1] + V[F[tri, 1]] * w[:, 1:2] + V[F[tri, 2]] * w[:, 2:3]
    n = np.cross(V[F[tri, 1]] - V[F[tri, 0]], V[F[tri, 2]] - V[F[tri, 0]])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    report("occlusion rays from the dragon's surface", p, hemi(n, k))
    dl = light - p
    report("light rays from the dragon's surface", p, dl / np.linalg.norm(dl, axis=1, keepdims=True))


if __name__ == "__main__":
    main()
