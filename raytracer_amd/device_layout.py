"""Device-side arrangement of the flat scene tables.

`flatten.flatten_world` yields the tables in the reference's own order: every kd leaf
lists its triangles in the order `KDTree.hit` / `shadow_hit` scan them
(/root/reference/kdtree.py:157-174).  That order is observable - ties in `t` go to the
earlier item and a shadow query reports the first item that hits - but nothing says the
items must be *stored* in it.  For leaves with more than one 64-triangle chunk this
module

  * stores the leaf's triangles in Morton order of their centroids, so that 64
    consecutive rows are spatially compact,
  * records each row's position in the reference order (`tri_order`), which the kernels
    use for every tie-break and first-in-order decision, and
  * gives every 64-row chunk an axis-aligned bounding box (`chunk_boxes`, padded), so a
    wave can skip chunks a ray's line cannot touch; `chunk_tree` holds the same boxes row by
    row (as floats, rounded outward) together with a box around every group of four chunks
    of a leaf, for a lane that sorts out the chunks its own ray can touch.

The degenerate leaves of the reference's build (hundreds to 1340 triangles, a few per
mesh, visited by most rays) are where almost all triangle tests are spent; skipping
chunks there changes no result, because a triangle the exact test accepts has its hit
point inside its own (padded) bounding box.
"""
import ctypes as C
import os

import numpy

from . import capi

CHUNK = 64
CELL = 8
GROUP = 4           # chunks per group box in `chunk_tree`
PAD = 1.0e-9
SHADOW_SUBTREES = not os.environ.get("RTAMD_NO_SHADOW_SUBTREES")
# shadow trees split triangles at the planes (rt_shadowtree.cpp) rather than their boxes
PERFECT_SPLITS = not os.environ.get("RTAMD_BOX_SPLITS")


def _padded(lo, hi):
    return lo - PAD * (1.0 + numpy.abs(lo)), hi + PAD * (1.0 + numpy.abs(hi))


def _boxes_f32(lo, hi):
    """(n, 3) double boxes as float32 that still enclose them (rounded outward)."""
    lo32, hi32 = lo.astype(numpy.float32), hi.astype(numpy.float32)
    lo32 = numpy.where(lo32.astype(numpy.float64) > lo, numpy.nextafter(lo32, numpy.float32(-numpy.inf)), lo32)
    hi32 = numpy.where(hi32.astype(numpy.float64) < hi, numpy.nextafter(hi32, numpy.float32(numpy.inf)), hi32)
    return lo32.astype(numpy.float32), hi32.astype(numpy.float32)


def _morton_order(centroids):
    lo = centroids.min(axis=0)
    span = numpy.maximum(centroids.max(axis=0) - lo, 1e-300)
    q = numpy.minimum(((centroids - lo) / span * 1024.0).astype(numpy.int64), 1023)
    code = numpy.zeros(len(centroids), dtype=numpy.int64)
    for bit in range(10):
        for axis in range(3):
            code |= ((q[:, axis] >> bit) & 1) << (3 * bit + axis)
    return numpy.argsort(code, kind="stable")


SHADOW_TREE_SLACK = 4.0e-5        # see shadow_tree_wave in csrc/rt_device.h
# trees with at least this many rows get a shadow tree of their own (RTAMD_SHADOW_TREE_MIN_ROWS: test runs that
# send the shadow rays of big meshes down the paths of small ones - the reference's tree, shadow subtrees of leaves)
SHADOW_TREE_MIN_ROWS = int(os.environ.get("RTAMD_SHADOW_TREE_MIN_ROWS", "512"))
AUX_MIN_ROWS = CHUNK  # leaves of more rows than this get a shadow subtree
AUX_LEAF = 16       # rows at which a shadow subtree stops splitting
AUX_MAX_LEAF = 64   # a shadow subtree whose build leaves more rows than this in a leaf is dropped
AUX_STEP_COST = 1.0  # cost of a traversal step in triangle tests
AUX_DEPTH = 16      # levels a shadow subtree may have (fewer below a deep leaf)


def _shadow_subtree(lo, hi, max_depth, corners=None, slack=0.0):
    """A kd-tree over the (padded) triangle boxes lo, hi [n, 3], for queries that any hit
    decides (rtamd_shadow_tree_build in csrc/rt_shadowtree.cpp: surface-area heuristic over the
    box faces, triangles whose box straddles a plane go to both sides - or, with the triangles'
    `corners` [n, 9] and the absolute `slack` the boxes were widened by, only to the sides the
    triangle itself comes within the slack of).  Returns (axis, split, link, leaf_offset,
    leaf_row, leaf_box, levels): nodes in preorder, axis -1 for a leaf whose number is link, else
    link = index of the right child; leaf k holds rows leaf_row[leaf_offset[k] :
    leaf_offset[k + 1]] (numbers into lo / hi) inside leaf_box[k] (lo.xyz, hi.xyz).  None when
    some leaf would keep more than AUX_MAX_LEAF rows within max_depth levels."""
    lib = capi.library()
    lo = numpy.ascontiguousarray(lo, dtype=numpy.float64)
    hi = numpy.ascontiguousarray(hi, dtype=numpy.float64)
    if corners is not None:
        corners = numpy.ascontiguousarray(corners, dtype=numpy.float64)
        assert corners.shape == (len(lo), 9)
    n_nodes, n_leaves, n_rows, levels = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()

    def call(axis, split, link, offsets, rows, boxes, max_nodes, max_rows):
        return lib.rtamd_shadow_tree_build(
            lo.ctypes.data, hi.ctypes.data, len(lo), int(max_depth), AUX_LEAF, AUX_MAX_LEAF, AUX_STEP_COST,
            corners.ctypes.data if corners is not None else None, float(slack),
            axis, split, link, max_nodes, offsets, rows, boxes, max_rows,
            C.byref(n_nodes), C.byref(n_leaves), C.byref(n_rows), C.byref(levels))

    max_nodes, max_rows = 2 * len(lo) + 64, 6 * len(lo) + 1024           # usually enough; the sizes come back either way
    for _ in range(2):
        axis = numpy.empty(max_nodes, dtype=numpy.int32)
        split = numpy.empty(max_nodes, dtype=numpy.float64)
        link = numpy.empty(max_nodes, dtype=numpy.int32)
        offsets = numpy.empty(max_nodes + 1, dtype=numpy.int64)
        rows = numpy.empty(max_rows, dtype=numpy.int32)
        boxes = numpy.zeros((max_nodes, 6), dtype=numpy.float64)
        status = call(axis.ctypes.data, split.ctypes.data, link.ctypes.data, offsets.ctypes.data, rows.ctypes.data,
                      boxes.ctypes.data, max_nodes, max_rows)
        if status == capi.EUNSUPPORTED:
            return None
        if status == 0:
            break
        if n_nodes.value <= max_nodes and n_rows.value <= max_rows:
            raise RuntimeError("rtamd_shadow_tree_build failed (%d)" % status)
        max_nodes, max_rows = n_nodes.value, n_rows.value
    else:
        raise RuntimeError("rtamd_shadow_tree_build failed (%d)" % status)
    axis, split, link = axis[:n_nodes.value], split[:n_nodes.value], link[:n_nodes.value]
    offsets = offsets[:n_leaves.value + 1]
    return axis, split, link, offsets, rows[:n_rows.value], boxes[:n_leaves.value], int(levels.value)


def device_arrays(flat):
    """Tables to upload (see _device_arrays), worked out once per flat scene."""
    key = (SHADOW_SUBTREES, SHADOW_TREE_MIN_ROWS, PERFECT_SPLITS)
    cached = getattr(flat, "_device_arrays", None)
    if cached is None or cached[0] != key:
        cached = flat._device_arrays = (key,) + _device_arrays(flat)
    return dict(cached[1]), cached[2]


def _device_arrays(flat):
    """Tables to upload: the canonical ones, with big leaves re-arranged, plus
    `tri_order` (int32 per row) and `chunk_boxes` (SoA [6][n_chunks]: lo xyz, hi xyz)."""
    a = dict(flat.arrays)
    n_refs = len(a["tri_prim"])
    geom = a["tri_geom"].reshape(9, n_refs).copy() if n_refs else numpy.zeros((9, 0))
    prim = a["tri_prim"].copy()
    order = numpy.arange(n_refs, dtype=numpy.int32)
    nodes = a["kd_nodes"].view([("split", "<f8"), ("a", "<i4"), ("b", "<i4")]).copy()
    boxes_lo, boxes_hi = [], []
    leaf_boxes = []                     # per chunk: [9][6] = chunk box, then its eight cell boxes
    n_chunks = 0
    leaves = numpy.nonzero((nodes["b"] & 3) == 3)[0]
    counts = nodes["b"][leaves] >> 2
    nodes["split"][leaves] = -1.0
    for node, count in zip(leaves[counts > CHUNK], counts[counts > CHUNK]):
        first = int(nodes["a"][node])
        rows = slice(first, first + int(count))
        g = geom[:, rows].copy()
        v0, v1, v2 = g[0:3].T, (g[0:3] + g[3:6]).T, (g[0:3] + g[6:9]).T
        perm = _morton_order((v0 + v1 + v2) / 3.0)
        geom[:, rows] = g[:, perm]
        prim[rows] = prim[rows][perm]
        order[rows] = first + perm.astype(numpy.int32)
        lo = numpy.minimum(numpy.minimum(v0, v1), v2)[perm]
        hi = numpy.maximum(numpy.maximum(v0, v1), v2)[perm]
        nodes["split"][node] = float(n_chunks)
        for c in range(0, int(count), CHUNK):
            blo, bhi = lo[c:c + CHUNK].min(axis=0), hi[c:c + CHUNK].max(axis=0)
            blo, bhi = _padded(blo, bhi)
            boxes_lo.append(blo)
            boxes_hi.append(bhi)
            rec = numpy.zeros((1 + CHUNK // CELL, 6))
            rec[0, 0:3], rec[0, 3:6] = blo, bhi
            for k in range(0, min(CHUNK, int(count) - c), CELL):
                klo, khi = _padded(lo[c + k:c + k + CELL].min(axis=0), hi[c + k:c + k + CELL].max(axis=0))
                rec[1 + k // CELL, 0:3], rec[1 + k // CELL, 3:6] = klo, khi
            leaf_boxes.append(rec)
            n_chunks += 1
    # per chunk: its box, and for every GROUP-th chunk of a leaf the box around the group
    chunk_tree = numpy.zeros((n_chunks, 12))
    if n_chunks:
        chunk_tree[:, 0:3], chunk_tree[:, 3:6] = numpy.array(boxes_lo), numpy.array(boxes_hi)
        for node, count in zip(leaves[counts > CHUNK], counts[counts > CHUNK]):
            c0 = int(nodes["split"][node])
            nch = (int(count) + CHUNK - 1) // CHUNK
            for s in range(0, nch, GROUP):
                members = chunk_tree[c0 + s:c0 + min(s + GROUP, nch)]
                chunk_tree[c0 + s, 6:9] = members[:, 0:3].min(axis=0)
                chunk_tree[c0 + s, 9:12] = members[:, 3:6].max(axis=0)
    if n_chunks:
        chunk_boxes = numpy.ascontiguousarray(numpy.concatenate([numpy.array(boxes_lo).T, numpy.array(boxes_hi).T], axis=0))
    else:
        chunk_boxes = numpy.zeros((6, 0))
    a["tri_geom"] = numpy.ascontiguousarray(geom).ravel()
    a["tri_prim"] = prim
    a["tri_order"] = order
    rows = numpy.zeros((n_refs, 10))
    rows[:, :9] = geom.T
    rows[:, 9] = numpy.stack([prim.astype(numpy.int32), order], axis=1).copy().view(numpy.float64)[:, 0]
    a["tri_rows"] = rows.ravel()
    a["chunk_boxes"] = chunk_boxes.ravel()
    a["leaf_boxes"] = numpy.concatenate(leaf_boxes).ravel() if leaf_boxes else numpy.zeros(0)
    # culling only: single precision, rounded outward, is enough and halves what a lane loads
    ct32 = numpy.zeros((n_chunks, 12), dtype=numpy.float32)
    if n_chunks:
        ct32[:, 0:3], ct32[:, 3:6] = _boxes_f32(chunk_tree[:, 0:3], chunk_tree[:, 3:6])
        has_group = chunk_tree[:, 6:12].any(axis=1)
        glo, ghi = _boxes_f32(chunk_tree[has_group, 6:9], chunk_tree[has_group, 9:12])
        ct32[has_group, 6:9], ct32[has_group, 9:12] = glo, ghi
    a["chunk_tree"] = ct32.ravel()
    # box around the triangles of every leaf (bigger leaves have chunk boxes as well)
    leaf_aabb = numpy.zeros((len(nodes), 6))
    boxed = counts > 0
    small = leaves[boxed]
    if len(small):
        first = nodes["a"][small].astype(numpy.int64)
        cnt = counts[boxed].astype(numpy.int64)
        row = numpy.repeat(first, cnt) + (numpy.arange(cnt.sum()) - numpy.repeat(numpy.cumsum(cnt) - cnt, cnt))
        seg = numpy.cumsum(cnt) - cnt
        v0 = geom[0:3, row]
        v1, v2 = v0 + geom[3:6, row], v0 + geom[6:9, row]
        lo = numpy.minimum.reduceat(numpy.minimum(numpy.minimum(v0, v1), v2), seg, axis=1).T
        hi = numpy.maximum.reduceat(numpy.maximum(numpy.maximum(v0, v1), v2), seg, axis=1).T
        leaf_aabb[small, 0:3], leaf_aabb[small, 3:6] = _padded(lo, hi)
    la32 = numpy.zeros((len(nodes), 8), dtype=numpy.float32)
    if len(small):
        la32[small, 0:3], la32[small, 3:6] = _boxes_f32(leaf_aabb[small, 0:3], leaf_aabb[small, 3:6])
    # Shadow subtrees.  A shadow query that any hit decides (no limit, or the ray leaves the tree's
    # box before its limit) only asks whether some triangle on the ray is hit, whichever leaf
    # lists it.  For such rays every leaf of more than CHUNK rows gets a kd-tree of its own over
    # its triangles (appended nodes, leaves of a few appended rows with tight boxes) that the
    # walk continues into; its root's index rides in the seventh float of the leaf's box row.
    trees = numpy.frombuffer(bytes(a["trees"]), dtype=numpy.uint8).copy()
    tree_recs = (capi.Tree * (len(trees) // C.sizeof(capi.Tree))).from_buffer(trees)
    tree_starts = numpy.array([t.node_offset for t in tree_recs], dtype=numpy.int64)
    big = leaves[counts > AUX_MIN_ROWS]
    if SHADOW_SUBTREES and len(tree_recs):
        # interior nodes above every node (children follow their parent in the array)
        level = numpy.zeros(len(nodes), dtype=numpy.int64)
        for i in numpy.nonzero((nodes["b"] & 3) != 3)[0]:
            level[i + 1] = level[nodes["a"][i]] = level[i] + 1
        extra_nodes, extra_rows, extra_boxes = [], [], []
        next_node, next_row = [len(nodes)], [n_refs]

        def attach(src_rows, room, slack=0.0):
            """Build a shadow subtree over the table rows src_rows; (root index, levels) or None.
            `slack`: widen every triangle's box by this fraction of the largest coordinate, for a
            walk in single precision (it lists a triangle wherever rounding could take the walk)."""
            g = geom[:, src_rows]
            v0, v1, v2 = g[0:3].T, (g[0:3] + g[3:6]).T, (g[0:3] + g[6:9]).T
            blo, bhi = _padded(numpy.minimum(numpy.minimum(v0, v1), v2), numpy.maximum(numpy.maximum(v0, v1), v2))
            reach = max(float(numpy.abs(blo).max()), float(numpy.abs(bhi).max()), 1e-30)
            wide = PAD * (1.0 + reach)              # what _padded added at the least
            if slack:
                wide = slack * reach
                blo, bhi = blo - wide, bhi + wide
            corners = numpy.concatenate([v0, v1, v2], axis=1) if PERFECT_SPLITS else None
            built = _shadow_subtree(blo, bhi, room, corners, wide)
            if built is None:
                return None
            axis, split, link, offsets, rows_of, leaf_box, levels = built
            root = next_node[0]
            more = numpy.zeros(len(axis), dtype=nodes.dtype)
            is_leaf = axis < 0
            counts_here = numpy.diff(offsets)[link[is_leaf]]
            more["split"] = numpy.where(is_leaf, -1.0, split)
            more["a"][~is_leaf] = root + link[~is_leaf]
            more["b"][~is_leaf] = axis[~is_leaf]
            more["a"][is_leaf] = next_row[0] + offsets[:-1][link[is_leaf]]
            more["b"][is_leaf] = (counts_here.astype(numpy.int32) << 2) | 3
            boxes32 = numpy.zeros((len(axis), 8), dtype=numpy.float32)
            filled = numpy.nonzero(numpy.diff(offsets) > 0)[0]                     # leaves that hold rows
            if len(filled):
                # (the builder's own boxes: around the parts of the triangles inside the leaf's cell)
                leaf_lo, leaf_hi = leaf_box[filled, 0:3], leaf_box[filled, 3:6]
                node_of_leaf = numpy.nonzero(is_leaf)[0][numpy.argsort(link[is_leaf])]
                l32, h32 = _boxes_f32(leaf_lo, leaf_hi)
                boxes32[node_of_leaf[filled], 0:3], boxes32[node_of_leaf[filled], 3:6] = l32, h32
            extra_nodes.append(more)
            extra_boxes.append(boxes32)
            extra_rows.append(src_rows[rows_of])
            next_row[0] += len(rows_of)
            next_node[0] += len(axis)
            return root, levels

        whole = set()
        ref_starts = [t.ref_offset for t in tree_recs] + [n_refs]
        for ti, t in enumerate(tree_recs):
            lo_r, hi_r = ref_starts[ti], ref_starts[ti + 1]
            if hi_r - lo_r < SHADOW_TREE_MIN_ROWS:
                continue
            # every triangle of the tree once; those that cast no shadow (bit 30) cannot occlude
            ids = prim[lo_r:hi_r]
            _, firsts = numpy.unique(ids, return_index=True)
            firsts = numpy.sort(firsts[(ids[firsts] & (1 << 30)) == 0])
            if not len(firsts):
                continue
            got = attach(lo_r + firsts, capi.MAX_TREE_DEPTH, SHADOW_TREE_SLACK)
            if got is not None:
                t.shadow_root = got[0]
                t.walk_depth = max(t.walk_depth, got[1])
                whole.add(ti)
        for node, count in zip(big, counts[counts > AUX_MIN_ROWS]):
            tree = int(numpy.searchsorted(tree_starts, node, side="right")) - 1
            room = min(AUX_DEPTH, capi.MAX_TREE_DEPTH - int(level[node]))
            if room < 1 or tree in whole:
                continue
            first = int(nodes["a"][node])
            got = attach(numpy.arange(first, first + int(count)), room)
            if got is None:
                continue
            # what a walk may now pass through
            tree_recs[tree].walk_depth = max(tree_recs[tree].walk_depth, int(level[node]) + got[1])
            la32[node, 6] = numpy.array([got[0]], dtype=numpy.int32).view(numpy.float32)[0]
        if extra_nodes:
            nodes = numpy.concatenate([nodes] + extra_nodes)
            la32 = numpy.concatenate([la32] + extra_boxes)
            src = numpy.concatenate(extra_rows) if extra_rows else numpy.zeros(0, dtype=numpy.int64)
            geom = numpy.concatenate([geom, geom[:, src]], axis=1)
            prim = numpy.concatenate([prim, prim[src]])
            order = numpy.concatenate([order, order[src]])
            a["tri_geom"] = numpy.ascontiguousarray(geom).ravel()
            a["tri_prim"] = prim
            a["tri_order"] = order
            a["trees"] = trees          # (`tri_rows` stays as it is: only scans by single lanes read it, never in these trees)
    a["leaf_aabb"] = la32.ravel()
    a["kd_nodes"] = nodes.view(numpy.uint8)
    # node, left child, right child side by side: two levels of the walk per load
    triples = numpy.zeros((len(nodes), 4), dtype=nodes.dtype)
    triples[:, 0] = nodes
    interior = numpy.nonzero((nodes["b"] & 3) != 3)[0]
    triples[interior, 1] = nodes[interior + 1]
    triples[interior, 2] = nodes[nodes["a"][interior]]
    a["kd_triples"] = triples.view(numpy.uint8).ravel()
    return a, n_chunks
