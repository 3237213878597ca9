"""Host kd-tree build that yields exactly the tree of /root/reference/kdtree.py
(`BoundingBoxes` :8-24, `SAH` :27-52, `KDTree.__init__` :56-69, `sub_divide` :92-131),
kept as flat arrays ready for the device.

The reference recurses over Python lists of box objects.  Here a node's box list is an
index array into per-box `lo`/`hi` coordinate arrays, and each step of the reference
is restated on those arrays:

* `sorted(...)` / `list.sort(...)` are stable, so is `argsort(kind="stable")`;
* `SAH` sorts the node's list *in place* by upper bound (kdtree.py:28), so the list the
  next axis sees - and the list a degenerate node keeps as its leaf - is the one left
  behind by the previous axis; the index array is threaded through the three axes the
  same way;
* both `while` scans (kdtree.py:38-42) count entries `<= plane` in a sorted list with a
  monotone plane, which is `searchsorted(side="right")`;
* the cost (kdtree.py:43) is evaluated as two float products and one sum per candidate,
  the first strict minimum wins (`argmin` returns the first), and axes compete with a
  strict `<` after scaling by the cell's cross-section (kdtree.py:104-108).

Traversal and leaf tests run on the GPU (csrc/rt_device.h); `hit`/`shadow_hit`
are not methods of this class.

The same build also exists on the device (csrc/rt_kdbuild.hip, `rtamd_kd_build`): level by
level, the same stable sorts and the same arithmetic, yielding the same arrays.  `KD_BUILD`
(environment variable RTAMD_KD_BUILD) picks the builder: "host", "device", or "auto" - the
device for lists of at least `DEVICE_BUILD_MIN_BOXES` boxes when a GPU is visible.
"""
import os

import numpy

from geometry import BoundingBox, GeometryObject, INF, epsilon

LEAF_LIMIT = 20          # kdtree.py:94
KD_BUILD = os.environ.get("RTAMD_KD_BUILD", "auto")
DEVICE_BUILD_MIN_BOXES = 2000


class BoundingBoxes(object):
    """A list of boxes plus its extent (kdtree.py:8-24).  `x_boxes` is the list ordered
    by box centre along x (stable); the y/z orderings the reference also keeps are never
    read by the build and are not materialised."""

    def __init__(self, boxes):
        self.boxes = list(boxes)
        n = len(self.boxes)
        lo = numpy.empty((n, 3))
        hi = numpy.empty((n, 3))
        for i, box in enumerate(self.boxes):
            d = box.data
            lo[i, 0], hi[i, 0] = d[0]
            lo[i, 1], hi[i, 1] = d[1]
            lo[i, 2], hi[i, 2] = d[2]
        self.lo, self.hi = lo, hi
        self.order = _by_mid_x(lo, hi, numpy.arange(n))
        self.x0, self.y0, self.z0 = lo.min(axis=0)
        self.x1, self.y1, self.z1 = hi.max(axis=0)

    @property
    def x_boxes(self):
        return [self.boxes[i] for i in self.order]

    def get_bounding_box(self):
        return BoundingBox(self.x0, self.x1, self.y0, self.y1, self.z0, self.z1, None)


def _by_mid_x(lo, hi, items):
    mid = (lo[items, 0] + hi[items, 0]) / 2
    return items[numpy.argsort(mid, kind="stable")]


def _sah(items, lo, hi, axis, min_d, max_d):
    """kdtree.py:27-52 on index arrays.  Returns (cost, left, right, plane, items) where
    `items` is the list as the in-place sort left it."""
    upper = hi[items, axis]
    perm = numpy.argsort(upper, kind="stable")
    items = items[perm]
    upper = upper[perm]
    lower = lo[items, axis]
    perm = numpy.argsort(lower, kind="stable")
    by_lower = items[perm]
    lower = lower[perm]

    n = len(items)
    plane = upper + epsilon
    n_upper_le = numpy.searchsorted(upper, plane, side="right")
    n_lower_le = numpy.searchsorted(lower, plane, side="right")
    cost = (n_lower_le + 1) * (plane - min_d) + (n - n_upper_le - 1) * (max_d - plane)
    with numpy.errstate(invalid="ignore"):
        usable = cost < INF
    if not usable.any():
        # no candidate beats the initial cost: the reference then slices with None,
        # i.e. both sides get everything; such a node ends up as a leaf
        return INF, by_lower, items, min_d, items
    best = int(numpy.argmin(numpy.where(usable, cost, numpy.inf)))
    center = min(max_d, max(float(plane[best]), min_d))
    return float(cost[best]), by_lower[:n_lower_le[best]], items[n_upper_le[best]:], center, items


class KDTree(GeometryObject):
    """The reference tree, flattened while it is built.

    Node arrays, preorder (a node's left subtree follows it directly):
      axis[i]   0/1/2 for an interior node, -1 for a leaf
      split[i]  splitting plane of an interior node
      left[i], right[i]  child node indices
      first[i], count[i] slice of `leaf_refs` held by a leaf
    `leaf_refs` are indices into the box list given to the root `BoundingBoxes`, in the
    order the reference would scan each leaf.
    """

    def __init__(self, bounding_boxes, leaf=False, min_d=None, max_d=None):
        super(KDTree, self).__init__()
        self.bounding_boxes = bounding_boxes
        self.bounding_box = bounding_boxes.get_bounding_box()
        bb = self.bounding_box
        if min_d is None:
            min_d = [bb.x0, bb.y0, bb.z0]
        if max_d is None:
            max_d = [bb.x1, bb.y1, bb.z1]
        self._lo, self._hi = bounding_boxes.lo, bounding_boxes.hi
        self.built_on = "host"
        if not leaf and self._wants_device(len(bounding_boxes.boxes)) and \
                [float(v) for v in min_d] == [bb.x0, bb.y0, bb.z0] and [float(v) for v in max_d] == [bb.x1, bb.y1, bb.z1]:
            try:
                self._build_on_device()
                return
            except (RuntimeError, MemoryError) as e:    # (torch's out-of-memory error is a RuntimeError)
                # RTAMD_KD_BUILD=auto: the device build is an accelerator for the host build below, which
                # yields the same arrays; an explicit RTAMD_KD_BUILD=device fails loudly
                if KD_BUILD == "device":
                    raise
                import warnings
                warnings.warn("device kd build not available (%s): building on the host" % e)
        self._axis, self._split = [], []
        self._left, self._right = [], []
        self._first, self._count = [], []
        self._refs = []
        self._nrefs = 0
        self.depth = 0
        self._build(bounding_boxes.order, [float(v) for v in min_d], [float(v) for v in max_d],
                    bool(leaf), 0)
        self.axis = numpy.array(self._axis, dtype=numpy.int32)
        self.split = numpy.array(self._split, dtype=numpy.float64)
        self.left = numpy.array(self._left, dtype=numpy.int32)
        self.right = numpy.array(self._right, dtype=numpy.int32)
        self.first = numpy.array(self._first, dtype=numpy.int32)
        self.count = numpy.array(self._count, dtype=numpy.int32)
        self.leaf_refs = (numpy.concatenate(self._refs).astype(numpy.int32)
                          if self._refs else numpy.zeros(0, numpy.int32))
        del self._axis, self._split, self._left, self._right, self._first, self._count, self._refs

    # -- build -----------------------------------------------------------------------
    @staticmethod
    def _wants_device(n_boxes):
        if KD_BUILD == "host" or n_boxes < 1:
            return False
        if KD_BUILD == "device":
            return True
        if n_boxes < DEVICE_BUILD_MIN_BOXES:
            return False
        try:
            import torch
            return torch.cuda.is_available()
        except ImportError:
            return False

    def _build_on_device(self, level_factor=4):
        """The same tree from `rtamd_kd_build`.  Levels may hold up to `level_factor` times
        the box count (boxes that straddle a plane go to both children); when a mesh needs
        more, the build is repeated with a larger workspace."""
        import ctypes as C
        import torch
        from raytracer_amd import capi
        lib = capi.library()
        n = len(self._lo)
        dev = torch.device("cuda", torch.cuda.current_device())
        lo = torch.from_numpy(numpy.ascontiguousarray(self._lo)).to(dev)
        hi = torch.from_numpy(numpy.ascontiguousarray(self._hi)).to(dev)
        while True:
            cap_items = max(level_factor * n, 1024)
            max_nodes, max_refs = 2 * cap_items, 2 * cap_items
            i32 = dict(dtype=torch.int32, device=dev)
            arrays = dict(axis=torch.empty(max_nodes, **i32), split=torch.empty(max_nodes, dtype=torch.float64, device=dev),
                          left=torch.empty(max_nodes, **i32), right=torch.empty(max_nodes, **i32),
                          first=torch.empty(max_nodes, **i32), count=torch.empty(max_nodes, **i32),
                          leaf_refs=torch.empty(max_refs, **i32))
            out = capi.KdTreeOut(max_nodes=max_nodes, max_refs=max_refs, **{k: v.data_ptr() for k, v in arrays.items()})
            need = C.c_size_t()
            capi.check_kd(lib.rtamd_kd_build_workspace_bytes(cap_items, max_nodes, max_refs, C.byref(need)), lib)
            workspace = torch.empty(need.value, dtype=torch.uint8, device=dev)
            n_nodes, n_refs, depth = C.c_int64(), C.c_int64(), C.c_int32()
            torch.cuda.current_stream(dev).synchronize()
            rc = lib.rtamd_kd_build(lo.data_ptr(), hi.data_ptr(), n, C.byref(out), workspace.data_ptr(), need.value,
                                    cap_items, C.byref(n_nodes), C.byref(n_refs), C.byref(depth),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc == capi.EUNSUPPORTED and level_factor < 64:
                # only a level (or the tree) that outgrew the buffers is worth a larger workspace, and only
                # while that workspace fits in the memory the device has free
                why = (lib.rtamd_kd_last_error() or b"").decode()
                outgrown = ("holds more references" in why or "more nodes" in why or "more leaf references" in why)
                bigger = C.c_size_t()
                cap2 = max(2 * level_factor * n, 1024)
                lib.rtamd_kd_build_workspace_bytes(cap2, 2 * cap2, 2 * cap2, C.byref(bigger))
                free_bytes = torch.cuda.mem_get_info(dev)[0] + need.value
                if outgrown and bigger.value + 7 * 4 * 2 * cap2 + 4 * 2 * cap2 < free_bytes:
                    level_factor *= 2
                    del workspace, arrays
                    continue
            capi.check_kd(rc, lib)
            break
        nn, nr = n_nodes.value, n_refs.value
        self.axis = arrays["axis"][:nn].cpu().numpy()
        self.split = arrays["split"][:nn].cpu().numpy()
        self.left = arrays["left"][:nn].cpu().numpy()
        self.right = arrays["right"][:nn].cpu().numpy()
        self.first = arrays["first"][:nn].cpu().numpy()
        self.count = arrays["count"][:nn].cpu().numpy()
        self.leaf_refs = arrays["leaf_refs"][:nr].cpu().numpy()
        self.depth = int(depth.value)
        self.built_on = "device"

    def _new_node(self):
        for arr, v in ((self._axis, -1), (self._split, 0.0), (self._left, -1),
                       (self._right, -1), (self._first, 0), (self._count, 0)):
            arr.append(v)
        return len(self._axis) - 1

    def _make_leaf(self, node, items):
        self._first[node] = self._nrefs
        self._count[node] = len(items)
        self._refs.append(items)
        self._nrefs += len(items)

    def _build(self, items, min_d, max_d, forced_leaf, depth):
        node = self._new_node()
        self.depth = max(self.depth, depth)
        if forced_leaf or len(items) <= LEAF_LIMIT:
            self._make_leaf(node, items)
            return node
        lo, hi = self._lo, self._hi
        best_cost, best = INF, None
        for axis in range(3):
            cost, left, right, center, items = _sah(items, lo, hi, axis, min_d[axis], max_d[axis])
            dim_a = max_d[(axis + 1) % 3] - min_d[(axis + 1) % 3]
            dim_b = max_d[(axis + 2) % 3] - min_d[(axis + 2) % 3]
            cost *= dim_a * dim_b
            if cost < best_cost:
                best_cost, best = cost, (axis, center, left, right)
        if best is None or len(best[2]) == 0 or len(best[3]) == 0:
            # kdtree.py:114-115: the leaf keeps the list as the third in-place sort left it
            self._make_leaf(node, items)
            return node
        axis, center, left, right = best
        self._axis[node] = axis
        self._split[node] = center
        max_left = list(max_d)
        max_left[axis] = center
        min_right = list(min_d)
        min_right[axis] = center
        n = len(items)
        # children re-sort their list by box centre along x (kdtree.py:10 via :125-131);
        # a child that received every box is forced to be a leaf (:124, :128)
        self._left[node] = self._build(_by_mid_x(lo, hi, left), min_d, max_left, len(left) == n, depth + 1)
        self._right[node] = self._build(_by_mid_x(lo, hi, right), min_right, max_d, len(right) == n, depth + 1)
        return node

    # -- introspection ---------------------------------------------------------------
    @property
    def boxes(self):
        return self.bounding_boxes.boxes

    def stats(self):
        leaves = self.axis < 0
        return dict(nodes=int(len(self.axis)), leaves=int(leaves.sum()),
                    refs=int(len(self.leaf_refs)), depth=int(self.depth),
                    max_leaf=int(self.count[leaves].max()) if leaves.any() else 0)

    def signature(self):
        """Hex digest over the preorder node list: ('I', axis, split) for interior nodes,
        ('L', face labels...) for leaves.  Used to pin the build against the reference."""
        import hashlib
        h = hashlib.sha256()
        boxes = self.boxes
        for i in range(len(self.axis)):
            if self.axis[i] >= 0:
                h.update(("I%d:%s;" % (self.axis[i], float(self.split[i]).hex())).encode())
            else:
                refs = self.leaf_refs[self.first[i]:self.first[i] + self.count[i]]
                h.update(("L" + ",".join(str(getattr(boxes[r].obj, "label", r)) for r in refs) + ";").encode())
        return h.hexdigest()

    def print_tree(self, depth=0):
        """The reference dumps every leaf's labels to stdout (kdtree.py:82-90) and scene
        files call this right after the build; the call is accepted and stays quiet."""
        return None
