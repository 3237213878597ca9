"""The C-ABI shared library: builds for gfx950, loads without a GPU and exports every
entry point include/raytracer_amd.h declares (no compute call is made here)."""
import ctypes as C
import os
import re


def declared_functions(repo_root):
    text = open(os.path.join(repo_root, "include", "raytracer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtamd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(repo_root):
    import __graft_entry__
    __graft_entry__.build()
    from raytracer_amd import capi
    lib = C.CDLL(capi.LIB_PATH)
    names = declared_functions(repo_root)
    assert set(names) == set(capi.EXPORTS), "capi.EXPORTS and the header disagree"
    for name in names:
        assert hasattr(lib, name), "%s is declared but not exported" % name
    capi.bind(lib)
    assert lib.rtamd_abi_version() == capi.ABI_VERSION


def test_argument_checks_need_no_device(repo_root):
    from raytracer_amd import capi
    lib = capi.library()
    handle = C.c_void_p()
    assert lib.rtamd_scene_create(None, C.byref(handle)) == -1
    assert b"null" in lib.rtamd_last_error()
    bad = capi.SceneDesc()
    bad.abi_version = 999
    assert lib.rtamd_scene_create(C.byref(bad), C.byref(handle)) == -1
    assert b"ABI" in lib.rtamd_last_error()
    assert lib.rtamd_trace_rays(None, 1, 0, None, None, None, None, 0, None, None, 0, None) == -1


def test_product_does_not_import_the_checker(repo_root):
    """Nothing under raytracer_amd/ may reach into oracle/ (the checker is test infrastructure)."""
    pkg = os.path.join(repo_root, "raytracer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(base, f)


def test_shadow_tree_builder_on_the_host():
    """rtamd_shadow_tree_build is host code: sizes, capacity errors, the leaf limit, and the
    property its trees exist for - every box is listed in every leaf whose cell it reaches into."""
    import ctypes as C
    import numpy
    from raytracer_amd import capi
    lib = capi.library()
    rng = numpy.random.RandomState(3)
    lo = rng.uniform(-1, 1, (500, 3))
    hi = lo + rng.uniform(0.01, 0.3, (500, 3))
    n_nodes, n_leaves, n_rows, depth = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()

    def call(lo, hi, max_depth, leaf, max_leaf, out=None, corners=None, slack=0.0, boxes=None):
        args = [None] * 3 + [0] + [None] * 3 + [0]
        if out is not None:
            axis, split, link, offsets, rows = out
            args = [axis.ctypes.data, split.ctypes.data, link.ctypes.data, len(axis), offsets.ctypes.data, rows.ctypes.data,
                    boxes.ctypes.data if boxes is not None else None, len(rows)]
        return lib.rtamd_shadow_tree_build(lo.ctypes.data, hi.ctypes.data, len(lo), max_depth, leaf, max_leaf, 1.0,
                                           corners.ctypes.data if corners is not None else None, slack, *args,
                                           C.byref(n_nodes), C.byref(n_leaves), C.byref(n_rows), C.byref(depth))
    assert call(lo, hi, 30, 8, 64) == 0                                   # sizes only
    sizes = (n_nodes.value, n_leaves.value, n_rows.value, depth.value)
    assert sizes[0] == 2 * sizes[1] - 1 and sizes[2] >= 500 and 0 < sizes[3] <= 30
    small = (numpy.zeros(4, numpy.int32), numpy.zeros(4), numpy.zeros(4, numpy.int32), numpy.zeros(5, numpy.int64), numpy.zeros(4, numpy.int32))
    assert call(lo, hi, 30, 8, 64, small) == -1 and (n_nodes.value, n_rows.value) == (sizes[0], sizes[2])   # too small: sizes come back
    out = (numpy.zeros(sizes[0], numpy.int32), numpy.zeros(sizes[0]), numpy.zeros(sizes[0], numpy.int32),
           numpy.zeros(sizes[1] + 1, numpy.int64), numpy.zeros(sizes[2], numpy.int32))
    assert call(lo, hi, 30, 8, 64, out) == 0
    axis, split, link, offsets, rows = out
    todo = [(0, lo.min(axis=0), hi.max(axis=0), numpy.arange(500))]
    seen = numpy.zeros(500, dtype=bool)
    while todo:
        at, cell_lo, cell_hi, members = todo.pop()
        if axis[at] < 0:
            got = rows[offsets[link[at]]:offsets[link[at] + 1]]
            assert sorted(got) == sorted(members) and len(got) <= 64
            seen[members] = True
            continue
        ax, plane = int(axis[at]), float(split[at])
        assert cell_lo[ax] < plane < cell_hi[ax] and link[at] > at + 1
        left_hi, right_lo = cell_hi.copy(), cell_lo.copy()
        left_hi[ax] = right_lo[ax] = plane
        todo.append((at + 1, cell_lo, left_hi, members[lo[members, ax] < plane]))
        todo.append((int(link[at]), right_lo, cell_hi, members[hi[members, ax] > plane]))
    assert seen.all()
    # a hundred boxes no plane can tell apart, with room for 64 per leaf
    same_lo, same_hi = numpy.tile(lo[:1], (100, 1)), numpy.tile(hi[:1], (100, 1))
    assert call(same_lo, same_hi, 30, 8, 64) == capi.EUNSUPPORTED
    assert call(same_lo, same_hi, 30, 8, 128) == 0 and (n_nodes.value, n_leaves.value, n_rows.value) == (1, 1, 100)
    assert call(lo, hi, 30, 0, 64) == -1 and call(lo, hi, -1, 8, 64) == -1 and call(lo, hi, 30, 16, 8) == -1
    # coordinates that are not numbers are refused (they would break the order the builder sorts by)
    for bad in (numpy.nan, numpy.inf):
        broken = hi.copy()
        broken[123, 1] = bad
        assert call(lo, broken, 30, 8, 64) == -1
        corners = numpy.ascontiguousarray(numpy.concatenate([lo, hi, lo], axis=1))
        corners[77, 4] = bad
        assert call(lo, hi, 30, 8, 64, corners=corners, slack=1e-6) == -1


def test_shadow_tree_builder_splits_triangles_at_the_planes():
    """With the triangles' corners the builder lists a triangle only where the triangle itself comes within the
    slack of a leaf's cell - and everywhere it does: sample points of every triangle, moved by up to the slack,
    always find their triangle in the leaf whose cell they fall into, inside that leaf's box."""
    import ctypes as C
    import numpy
    from raytracer_amd import capi
    lib = capi.library()
    rng = numpy.random.RandomState(11)
    n = 400
    centre = rng.uniform(-1, 1, (n, 3))
    corners = numpy.ascontiguousarray(numpy.concatenate([centre + rng.uniform(-0.25, 0.25, (n, 3)) for _ in range(3)], axis=1))
    tri = corners.reshape(n, 3, 3)
    slack = 1e-3
    lo = numpy.ascontiguousarray(tri.min(axis=1) - slack)
    hi = numpy.ascontiguousarray(tri.max(axis=1) + slack)
    n_nodes, n_leaves, n_rows, depth = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()

    def build(with_corners):
        cap = 64 * n
        axis, split, link = numpy.zeros(cap, numpy.int32), numpy.zeros(cap), numpy.zeros(cap, numpy.int32)
        offsets, rows, boxes = numpy.zeros(cap + 1, numpy.int64), numpy.zeros(cap, numpy.int32), numpy.zeros((cap, 6))
        rc = lib.rtamd_shadow_tree_build(lo.ctypes.data, hi.ctypes.data, n, 30, 4, 64, 1.0,
                                         corners.ctypes.data if with_corners else None, slack,
                                         axis.ctypes.data, split.ctypes.data, link.ctypes.data, cap, offsets.ctypes.data,
                                         rows.ctypes.data, boxes.ctypes.data, cap,
                                         C.byref(n_nodes), C.byref(n_leaves), C.byref(n_rows), C.byref(depth))
        assert rc == 0
        return axis, split, link, offsets, rows, boxes, n_rows.value
    plain = build(False)
    axis, split, link, offsets, rows, boxes, total = build(True)
    assert total < plain[6]                              # thin slanted triangles: their boxes reach much farther than they do

    def leaf_of(p):
        at = 0
        while axis[at] >= 0:
            at = at + 1 if p[axis[at]] < split[at] else int(link[at])
        return int(link[at])
    for t in range(n):
        w = rng.dirichlet((1, 1, 1), 12)
        points = w.dot(tri[t]) + rng.uniform(-0.9 * slack, 0.9 * slack, (12, 3))
        for p in points:
            k = leaf_of(p)
            assert t in rows[offsets[k]:offsets[k + 1]]
            assert (boxes[k, 0:3] <= p).all() and (p <= boxes[k, 3:6]).all()
