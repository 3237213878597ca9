"""The device-side kd-tree build (rtamd_kd_build) against the host build, which is pinned to
the reference's trees by signature (tests/test_host_scene.py): every array identical."""
import os

import numpy
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("axis", "split", "left", "right", "first", "count", "leaf_refs")


def _both(boxes_or_bb):
    import raytracer_amd
    raytracer_amd.activate()
    import kdtree
    bb = boxes_or_bb if isinstance(boxes_or_bb, kdtree.BoundingBoxes) else kdtree.BoundingBoxes(boxes_or_bb)
    old = kdtree.KD_BUILD
    try:
        kdtree.KD_BUILD = "host"
        host = kdtree.KDTree(bb)
        kdtree.KD_BUILD = "device"
        dev = kdtree.KDTree(bb)
    finally:
        kdtree.KD_BUILD = old
    assert host.built_on == "host" and dev.built_on == "device"
    return host, dev


def _assert_same(host, dev, what):
    assert host.depth == dev.depth, what
    for f in FIELDS:
        a, b = getattr(host, f), getattr(dev, f)
        assert a.shape == b.shape, "%s: %s has %s entries on the host, %s on the device" % (what, f, a.shape, b.shape)
        if f == "split":
            same = (a.view(numpy.int64) == b.view(numpy.int64)) | (host.axis < 0)      # bit for bit
        else:
            same = a == b
        assert same.all(), "%s: %s differs at %s" % (what, f, numpy.nonzero(~same)[0][:5])
    assert host.signature() == dev.signature()


def _mesh_boxes(path, smooth=False):
    import raytracer_amd
    raytracer_amd.activate()
    from mesh_parser import read_mesh
    with open(os.path.join(raytracer_amd.ASSET_DIR, path)) as f:
        mesh = read_mesh(f)
    if smooth:
        mesh.compute_smooth_normal()
    return mesh.get_bounding_boxes()


@pytest.mark.parametrize("mesh", ("meshes/mesh1.obj", "meshes/teapot.obj", "meshes/dragon.obj"))
def test_device_build_equals_host_build_on_the_reference_meshes(mesh):
    import torch
    assert torch.cuda.is_available()
    host, dev = _both(_mesh_boxes(mesh))
    _assert_same(host, dev, mesh)
    print("%s: %s" % (mesh, host.stats()))


def test_device_build_of_the_cornell_box_groups():
    import raytracer_amd
    raytracer_amd.activate()
    from mesh_parser import read_mesh_complex
    from sampler import RegularSampler
    cwd = os.getcwd()
    os.chdir(raytracer_amd.ASSET_DIR)
    try:
        groups = read_mesh_complex("CornellBox/CornellBox-Original.obj", mat_sampler=RegularSampler())
    finally:
        os.chdir(cwd)
    boxes = []
    for _, mesh in groups.items():
        mesh.compute_normal()
        boxes += mesh.get_bounding_boxes()
    host, dev = _both(boxes)
    _assert_same(host, dev, "cornell box")


class _Box(object):
    def __init__(self, lo, hi, label):
        self.data = [(lo[0], hi[0]), (lo[1], hi[1]), (lo[2], hi[2])]
        self.obj = type("Obj", (), {"label": str(label)})()


@pytest.mark.parametrize("case", ("ties", "tiny", "just_over_the_limit", "identical", "signed_zero", "slabs", "random"))
def test_device_build_on_synthetic_box_lists(case):
    import raytracer_amd
    raytracer_amd.activate()
    r = numpy.random.RandomState(7)
    if case == "ties":                  # coordinates on a coarse grid: many equal keys in every sort
        lo = r.randint(0, 6, size=(900, 3)).astype(float)
        hi = lo + r.randint(0, 3, size=(900, 3))
    elif case == "tiny":
        lo = r.rand(7, 3); hi = lo + 0.1
    elif case == "just_over_the_limit":
        lo = r.rand(21, 3); hi = lo + 0.05
    elif case == "identical":           # every box the same: each split hands all boxes to a child
        lo = numpy.tile([[0.0, 1.0, 2.0]], (50, 1)); hi = lo + 1.0
    elif case == "signed_zero":
        lo = numpy.where(r.rand(300, 3) < 0.3, -0.0, r.randint(-2, 2, size=(300, 3)).astype(float))
        hi = lo + numpy.where(r.rand(300, 3) < 0.5, 0.0, 1.0)
    elif case == "slabs":               # long thin boxes that straddle most planes
        lo = r.rand(600, 3) * [10, 1, 1]; hi = lo + [5.0, 0.01, 0.01]
    else:
        lo = r.rand(5000, 3) * 10; hi = lo + r.rand(5000, 3) ** 3
    boxes = [_Box(lo[i], hi[i], i) for i in range(len(lo))]
    host, dev = _both(boxes)
    _assert_same(host, dev, case)
