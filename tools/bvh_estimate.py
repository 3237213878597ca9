#!/usr/bin/env python3
"""What a wide bounding-volume hierarchy over the dragon's triangles would cost a shadow ray, next to the
shadow kd-tree the kernels walk - worked out on the host, without a GPU, by the surface-area measure: a random
line through the root box meets a convex cell with probability area(cell) / area(root), so the expected number
of visits of a line is the sum of those ratios over the nodes.

  python tools/bvh_estimate.py

Binary hierarchy by the surface-area heuristic over 16 bins of the centroids, leaves of at most 16 triangles
(each triangle once); an 8-wide node is three binary levels collapsed.  The kd figures come from the library's
own builder (rtamd_shadow_tree_build, triangles split at the planes)."""
import ctypes as C
import os
import sys

import numpy

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def area(lo, hi):
    d = numpy.maximum(hi - lo, 0.0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def main():
    import raytracer_amd
    raytracer_amd.activate()
    from mesh_parser import read_mesh
    from raytracer_amd import capi
    os.chdir(raytracer_amd.ASSET_DIR)
    with open("meshes/dragon.obj") as f:
        mesh = read_mesh(f, None)
    V = numpy.array(mesh.vertices, dtype=numpy.float64)
    F = numpy.array(mesh.indices, dtype=numpy.int64)[:, :3]
    v0, v1, v2 = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    lo = numpy.minimum(numpy.minimum(v0, v1), v2)
    hi = numpy.maximum(numpy.maximum(v0, v1), v2)
    cen = 0.5 * (lo + hi)
    root_area = float(area(lo.min(axis=0), hi.max(axis=0)))

    # ---- binary BVH, levels[d] = list of (box lo, box hi, is leaf, triangles)
    visits_by_depth, leaf_visits, leaf_tests = {}, 0.0, 0.0
    n_leaves = 0
    stack = [(numpy.arange(len(lo)), 0)]
    while stack:
        idx, depth = stack.pop()
        blo, bhi = lo[idx].min(axis=0), hi[idx].max(axis=0)
        p = float(area(blo, bhi)) / root_area
        visits_by_depth[depth] = visits_by_depth.get(depth, 0.0) + p
        if len(idx) <= 16:
            leaf_visits += p
            leaf_tests += p * len(idx)
            n_leaves += 1
            continue
        best = None
        for ax in range(3):
            c = cen[idx, ax]
            cmin, cmax = c.min(), c.max()
            if cmax <= cmin:
                continue
            bins = numpy.minimum(((c - cmin) / (cmax - cmin) * 16).astype(int), 15)
            for b in range(1, 16):
                left = bins < b
                nl = int(left.sum())
                if nl == 0 or nl == len(idx):
                    continue
                cost = float(area(lo[idx[left]].min(axis=0), hi[idx[left]].max(axis=0))) * nl + \
                    float(area(lo[idx[~left]].min(axis=0), hi[idx[~left]].max(axis=0))) * (len(idx) - nl)
                if best is None or cost < best[0]:
                    best = (cost, left)
        if best is None:
            half = len(idx) // 2
            order = numpy.argsort(cen[idx, 0])
            left = numpy.zeros(len(idx), dtype=bool)
            left[order[:half]] = True
            best = (0.0, left)
        stack.append((idx[best[1]], depth + 1))
        stack.append((idx[~best[1]], depth + 1))
    total2 = sum(visits_by_depth.values())
    # an 8-wide node visit = one visit of a binary node at depth 0, 3, 6, ...; its eight child boxes are tested there
    wide8 = sum(v for d, v in visits_by_depth.items() if d % 3 == 0)
    wide4 = sum(v for d, v in visits_by_depth.items() if d % 2 == 0)
    print("binary hierarchy: %d leaves, depth %d; per random line through the root box: %.1f node boxes met, of them %.2f leaves "
          "with %.1f triangles to test" % (n_leaves, max(visits_by_depth), total2, leaf_visits, leaf_tests))
    print("collapsed to 4-wide nodes: %.1f node visits (4 child boxes tested at each); to 8-wide nodes: %.1f node visits (8 child boxes each)"
          % (wide4, wide8))

    # ---- the shadow kd-tree the kernels walk
    lib = capi.library()
    pad = 1e-9
    klo, khi = lo - pad * (1 + numpy.abs(lo)), hi + pad * (1 + numpy.abs(hi))
    wide = 4e-5 * max(float(numpy.abs(klo).max()), float(numpy.abs(khi).max()))
    klo, khi = numpy.ascontiguousarray(klo - wide), numpy.ascontiguousarray(khi + wide)
    corners = numpy.ascontiguousarray(numpy.concatenate([v0, v1, v2], axis=1))
    n = len(lo)
    mn, mr = 8 * n, 12 * n
    axis, split, link = numpy.empty(mn, numpy.int32), numpy.empty(mn), numpy.empty(mn, numpy.int32)
    off, rows, boxes = numpy.empty(mn + 1, numpy.int64), numpy.empty(mr, numpy.int32), numpy.zeros((mn, 6))
    nn, nl, nr, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
    rc = lib.rtamd_shadow_tree_build(klo.ctypes.data, khi.ctypes.data, n, 38, 16, 64, 1.0, corners.ctypes.data, wide,
                                     axis.ctypes.data, split.ctypes.data, link.ctypes.data, mn, off.ctypes.data,
                                     rows.ctypes.data, boxes.ctypes.data, mr, C.byref(nn), C.byref(nl), C.byref(nr), C.byref(d))
    assert rc == 0
    cnt = numpy.diff(off[:nl.value + 1])
    steps = reached = boxes_hit = tests = 0.0
    stack = [(0, klo.min(axis=0), khi.max(axis=0))]
    a0 = float(area(klo.min(axis=0), khi.max(axis=0)))
    while stack:
        i, a, b = stack.pop()
        if axis[i] < 0:
            k = link[i]
            reached += float(area(a, b)) / a0
            if cnt[k] > 0:
                bl, bh = numpy.maximum(boxes[k, 0:3], a), numpy.minimum(boxes[k, 3:6], b)
                if (bh > bl).all():
                    boxes_hit += float(area(bl, bh)) / a0
                    tests += float(area(bl, bh)) / a0 * cnt[k]
            continue
        steps += float(area(a, b)) / a0
        bl, ar = b.copy(), a.copy()
        bl[axis[i]] = split[i]
        ar[axis[i]] = split[i]
        stack.append((i + 1, a, bl))
        stack.append((int(link[i]), ar, b))
    print("shadow kd-tree: %d nodes, %d rows, depth %d; per random line: %.1f plane steps, %.2f leaves reached, %.2f leaf boxes met "
          "with %.1f triangles to test" % (nn.value, nr.value, d.value, steps, reached, boxes_hit, tests))
    print("vector instructions per lane, at what the kernels spend per item (kd plane step 15, leaf box 25, child box of a wide node 23 "
          "+ 4 to push): kd walk %.0f; 4-wide walk %.0f; 8-wide walk %.0f (triangle tests not counted: %.1f kd / %.1f hierarchy)"
          % (steps * 15 + reached * 25, wide4 * 4 * 27, wide8 * 8 * 27, tests, leaf_tests))


if __name__ == "__main__":
    main()
