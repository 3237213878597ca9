"""GPU parity tests: the HIP path, called through the C ABI, against (a) the reference's
golden vectors and (b) the CPU oracle on the same seeded inputs."""
import numpy
import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch


@pytest.mark.parametrize("scene", common.GOLDEN_SCENES)
def test_gpu_matches_reference_golden(torch_gpu, scene):
    """render_pixels on the golden samples: primary-hit ids exact, radiance within tolerance."""
    from raytracer_amd import flatten, runtime
    world = common.world_for(scene)
    ds = runtime.device_scene(world)
    pix, smp, plane, lens = common.golden_inputs(scene)
    out = ds.render_pixels(flatten.camera_record(world), world.seed, pix, smp, plane, lens)
    worst = common.compare_with_golden(scene, out)
    print("%s: worst relative radiance error vs reference %.3g" % (scene, worst))


@pytest.mark.parametrize("scene,tiles,samples", [
    ("foursphere", 96, (0, 1)),
    ("glossyreflective", 64, (3, 11)),
    ("mirrorhall_dof", 48, (0, 8)),
    ("cornellbox_area", 64, (0, 16)),
    ("cornellbox", 32, (10, 20)),
    ("features_analytic", 80, (0, 9)),
    ("features_mesh", 60, (0, 4)),
    ("features_env", 60, (2, 7)),
    ("features_glass", 60, (0, 9)),
    ("features_pane", 40, (1, 4)),
    ("features_pile", 48, (0, 4)),
    ("features_nested", 80, (0, 4)),
    ("features_texture", 70, (0, 9)),
    ("glass_teapot", 80, (0, 4)),
    ("glossyreflective@realtime", 64, (0, 1)),
    ("dragon_area", 40, (0, 4)),
    ("dragon_area", 420, (0, 16)),
    ("j", 600, (0, 1)),
] + [("fuzz_%d" % k, 96, None) for k in range(9)])                  # whole frames, every sample
def test_gpu_tiles_match_oracle(torch_gpu, scene, tiles, samples):
    """Whole tiles through rtamd_render_tiles (ray generation on the device) against the
    CPU oracle: per-sample ids and counts exact, integer accumulators identical."""
    check_tiles_against_oracle(torch_gpu, common.world_for(scene), common.flat_for(scene), tiles, samples, scene)


@pytest.mark.parametrize("first", range(9, 123, 6))
def test_random_worlds_match_oracle(torch_gpu, first):
    """Six more seeded random worlds per case (raytracer_amd/scenes/fuzz.py), whole frames, every
    sample, against the oracle (which the reference pins on the first nine of the family).  (A sweep
    over thousands more, tests/fuzz_sweep.py, is run by hand.)"""
    from raytracer_amd import flatten
    from raytracer_amd.world import World
    for k in range(first, first + 6):
        world = World(viewmode="parallel", buildfunction="fuzz_%d" % k, seed=1)
        check_tiles_against_oracle(torch_gpu, world, flatten.flatten_world(world), 96, None, "fuzz_%d" % k)


@pytest.mark.parametrize("first", range(0, 40, 4))
def test_random_teapot_worlds_match_oracle(torch_gpu, first):
    """Forty seeded worlds around the teapot's kd-tree (`fuzzmesh_<k>`, raytracer_amd/scenes/fuzz.py):
    a mesh with a shadow tree under every tracer, bare and instanced, smooth and flat, with
    triangles that cast no shadow, lights inside and outside its box.  Whole frames, every sample."""
    from raytracer_amd import flatten
    from raytracer_amd.world import World
    for k in range(first, first + 4):
        world = World(viewmode="parallel", buildfunction="fuzzmesh_%d" % k, seed=1)
        check_tiles_against_oracle(torch_gpu, world, flatten.flatten_world(world), 96, None, "fuzzmesh_%d" % k)


def test_random_path_traced_worlds_match_oracle(torch_gpu):
    """Six seeded path-traced worlds (`fuzzpath_<k>`, raytracer_amd/scenes/fuzz.py): Matte and Emissive analytic
    objects, mesh1 in a tree of a few levels, the teapot bare and instanced, depth limits 1 to 5, under the
    PathTracer.  Whole frames, every sample."""
    from raytracer_amd import flatten
    from raytracer_amd.world import World
    for k in range(6):
        world = World(viewmode="parallel", buildfunction="fuzzpath_%d" % k, seed=1)
        check_tiles_against_oracle(torch_gpu, world, flatten.flatten_world(world), 96, None, "fuzzpath_%d" % k)


def test_shadow_walk_from_an_origin_on_a_split_plane(torch_gpu):
    """`fuzzmesh_2031`, a world of the hand-run sweep: an ambient-occlusion ray starts on the ground at an x that,
    rounded to single precision, equals a split plane of the teapot's shadow tree, and leaves the plane towards
    the side the walk calls near.  The plane's distance is then -0: a walk that asks `t < 0` takes it for a
    crossing at the start of the ray, goes to the far side alone and misses the triangle the ray hits (one
    sample of the frame came out unoccluded); the walk asks for the sign bit."""
    import sys
    from raytracer_amd import flatten
    from raytracer_amd.world import World
    World(viewmode="parallel", buildfunction="fuzz_0", seed=1)       # loads the scene files
    fuzz = [m for name, m in sys.modules.items() if name.startswith("rtamd_scene_fuzz")][0]
    fuzz._make_mesh(2031)
    world = World(viewmode="parallel", buildfunction="fuzzmesh_2031", seed=1)
    check_tiles_against_oracle(torch_gpu, world, flatten.flatten_world(world), 96, None, "fuzzmesh_2031")


def test_shadow_rays_that_start_on_the_planes_of_the_shadow_tree(torch_gpu):
    """Shadow rays whose origin has, in single precision, exactly the coordinate of a split plane of the mesh's
    shadow tree, for every plane of the tree: primary rays straight down onto the ground (x or z of a plane) and
    straight at the back wall (x or y of a plane), beside the teapot, so that the occlusion rays and the light's
    rays of those hits start on a plane and outside the tree's box.  The kernels that walk the shadow tree in
    single precision must answer like the kernel that follows the reference's walk."""
    import ctypes as C
    import raytracer_amd
    raytracer_amd.activate()
    from light import AmbientOccluder
    from raytracer_amd import capi, device_layout, flatten, runtime
    from raytracer_amd.world import World
    from geometry import Plane
    from material import Matte
    world = World(viewmode="parallel", buildfunction="k", seed=3)
    world.ambient_color = AmbientOccluder(numpy.array((1., 1., 1.)), .4, world.viewplane.sampler)
    world.objects.append(Plane(origin=(0, 0, 8), normal=(0, 0, -1), material=Matte(ka=1, kd=1, cd=numpy.array((1., .84, 1.)))))
    world.invalidate()
    flat = flatten.flatten_world(world)
    dev, _ = device_layout.device_arrays(flat)
    raw = bytes(dev["trees"])
    tree = (capi.Tree * (len(raw) // C.sizeof(capi.Tree))).from_buffer_copy(raw)[0]
    assert tree.shadow_root > 0
    nodes = numpy.frombuffer(dev["kd_nodes"], dtype=numpy.dtype([("split", "<f8"), ("a", "<i4"), ("b", "<i4")]))[tree.shadow_root:]
    lo, hi = numpy.array(tree.bbox_lo), numpy.array(tree.bbox_hi)
    origins, directions = [], []
    for axis in range(3):
        planes = numpy.unique(nodes["split"][nodes["b"] == axis].astype(numpy.float32)).astype(numpy.float64)
        assert len(planes) > 100
        for k, c in enumerate(planes):
            if axis != 1:
                # onto the ground (y = 0) in front of, or behind, or beside the box
                other = (hi[2 - axis] + 0.4 + 0.13 * (k % 7)) if k % 2 else (lo[2 - axis] - 0.4 - 0.13 * (k % 5))
                at = [0., 3., 0.]
                at[axis], at[2 - axis] = c, other
                origins.append(at)
                directions.append((0., -1., 0.))
            if axis != 2:
                # at the back wall (z = 8): x or y of a plane
                at = [hi[0] + 0.3 + 0.11 * (k % 5), 0.2 + 0.09 * (k % 11), 5.]
                at[axis] = c
                origins.append(at)
                directions.append((0., 0., 1.))
    origins, directions = numpy.array(origins), numpy.array(directions)
    n = len(origins)
    ds = runtime.device_scene(world)
    checked = 0
    for sample in range(4):
        pix = (numpy.arange(n) * 7 + sample) .astype(numpy.int32) % (world.viewplane.resolution[0] * world.viewplane.resolution[1])
        smp = numpy.full(n, sample, dtype=numpy.int32)
        fast = ds.trace_rays(world.seed, origins, directions, pix, smp)
        exact = ds.trace_rays(world.seed, origins, directions, pix, smp, count_work=True)
        assert numpy.array_equal(fast["hit_object"], exact["hit_object"]) and numpy.array_equal(fast["n_rays"], exact["n_rays"])
        differ = numpy.nonzero(~numpy.isclose(fast["rgb"], exact["rgb"], rtol=1e-12, atol=0, equal_nan=True).all(axis=1))[0]
        assert len(differ) == 0, "sample %d: %d of %d rays differ, first: origin %s fast %s exact %s" % (
            sample, len(differ), n, origins[differ[0]], fast["rgb"][differ[0]], exact["rgb"][differ[0]])
        checked += int((fast["n_rays"][:, 1] > 0).sum())
    assert checked > 2 * n          # the hits cast shadow rays


def check_tiles_against_oracle(torch, world, flat, tiles, samples, scene):
    from raytracer_amd import flatten, runtime
    from oracle import oracle
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    rng = numpy.random.RandomState(5)
    tile_ids = numpy.sort(rng.choice(tx * ty, size=min(tiles, tx * ty), replace=False)).astype(numpy.int32)
    # always include the last tile (partial when the frame is not a multiple of 8)
    tile_ids[-1] = tx * ty - 1
    s0, s1 = samples or (0, len(world.viewplane.sampler.patterns[0]))
    n = len(tile_ids) * 64 * (s1 - s0)
    fb = runtime.FrameBuffers(ds, tile_ids)
    per = ds.sample_buffers(n)
    ds.read_counters(reset=True)
    ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1, sum_rgb8=fb.sum_rgb8, sum_sq=fb.sum_sq, per_sample=per,
                    count_work=True)
    torch.cuda.synchronize()
    gpu_work = ds.read_counters(reset=True)
    got = {k: v.cpu().numpy() for k, v in per[0].items()}
    # the timed (non-counting) kernel and the one that counts executed work give the frame of the counting kernel,
    # which follows the reference's walk step by step (first: a world whose radiance sits on a float knife edge
    # against the CPU checker, below, is still checked for this)
    for flags in (dict(), dict(count_executed=True)):
        fb2 = runtime.FrameBuffers(ds, tile_ids)
        ds.render_tiles(cam, world.seed, fb2.tile_ids, s0, s1, sum_rgb8=fb2.sum_rgb8, **flags)
        torch.cuda.synchronize()
        assert torch.equal(fb2.sum_rgb8, fb.sum_rgb8), "%s: the timed kernels' frame differs from the reference-walk kernel's" % scene
        ds.read_counters(reset=True)
    osc = oracle.OracleScene(flat)
    want_rgb8, want_sq, want = osc.render_tiles(cam, world.seed, tile_ids, s0, s1,
                                                per_sample_want=("rgb", "hit_object", "hit_prim", "hit_t", "n_rays"),
                                                threads=8)
    assert numpy.array_equal(got["hit_object"], want["hit_object"]), scene
    assert numpy.array_equal(got["hit_prim"], want["hit_prim"]), scene
    assert numpy.array_equal(got["n_rays"], want["n_rays"]), scene
    # the counting launch reports the work of the reference algorithm (kd steps, leaves,
    # triangle and analytic tests), whatever shortcuts the kernel takes: identical to the oracle's
    assert gpu_work == osc.read_counters(reset=True), scene
    close, worst = common.rgb_close(got["rgb"], want["rgb"])
    assert close, "%s: worst relative error %g" % (scene, worst)
    # int(c*255) of one sample may differ by one only where c*255 is within rounding of an
    # integer (ambient-only pixels: 0.2*255 = 51.0 +- 1 ulp); a pixel's integer sum may
    # therefore differ by at most its number of such samples, and is identical otherwise
    mismatches = common.assert_integer_sums_match(fb.sum_rgb8.cpu().numpy(), want_rgb8, want["rgb"], len(tile_ids), s1 - s0, scene)
    assert numpy.allclose(fb.sum_sq.cpu().numpy(), want_sq, rtol=1e-8, atol=1e-12, equal_nan=True), scene
    print("%s: %d samples, worst relative radiance error vs oracle %.3g, accumulator mismatches %d"
          % (scene, n, worst, mismatches))


def test_trace_rays_entry_point(torch_gpu):
    """rtamd_trace_rays with explicit rays reproduces render_pixels for the same rays."""
    from raytracer_amd import flatten, runtime
    scene = "features_analytic"
    world = common.world_for(scene)
    g = common.golden(scene)
    ds = runtime.device_scene(world)
    pix, smp, plane, lens = common.golden_inputs(scene)
    a = ds.render_pixels(flatten.camera_record(world), world.seed, pix, smp, plane, lens)
    # the reference's own rays (origin, direction) from the golden file
    b = ds.trace_rays(world.seed, g["origin"], g["direction"], pix, smp)
    assert numpy.array_equal(a["hit_object"], b["hit_object"])
    close, worst = common.rgb_close(a["rgb"], b["rgb"], rtol=1e-9)
    assert close, worst


def test_empty_and_error_paths(torch_gpu):
    from raytracer_amd import capi, flatten, runtime
    import ctypes as C
    world = common.world_for("foursphere")
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    out = ds.render_pixels(cam, 1, numpy.zeros(0, numpy.int32), numpy.zeros(0, numpy.int32), numpy.zeros((0, 3)))
    assert out["rgb"].shape == (0, 3)
    lib = capi.library()
    bad = capi.Camera()
    assert lib.rtamd_render_pixels(ds.handle, C.byref(bad), 1, 1, None, None, None, None, None, None, 0, None) != 0
    assert b"view plane" in lib.rtamd_last_error()
    # a tree whose shadow tree lies outside the node table is refused
    world = common.world_for("features_mesh")
    ds = runtime.device_scene(world)
    trees = (capi.Tree * (ds.tensors["trees"].numel() // C.sizeof(capi.Tree))).from_buffer_copy(ds.tensors["trees"].cpu().numpy().tobytes())
    assert trees[0].shadow_root > 0
    for break_it, message in ((lambda d, t: setattr(t[0], "shadow_root", d.n_kd_nodes + 7), b"shadow tree root"),
                              (lambda d, t: setattr(t[0], "shadow_root", -3), b"shadow tree root")):
        desc = ds.table_sizes(ds.flat.desc(lambda name, arr: ds.tensors[name].data_ptr()))
        desc.kd_triples, desc.leaf_aabb = ds.tensors["kd_triples"].data_ptr(), ds.tensors["leaf_aabb"].data_ptr()
        copy = (capi.Tree * len(trees)).from_buffer_copy(bytes(trees))
        break_it(desc, copy)
        import torch
        dev_trees = torch.frombuffer(bytearray(bytes(copy)), dtype=torch.uint8).to(ds.device)
        desc.trees = dev_trees.data_ptr()
        handle = C.c_void_p()
        assert lib.rtamd_scene_create(C.byref(desc), C.byref(handle)) != 0
        assert message in lib.rtamd_last_error()


def _tiny_world(scene, resolution=(8, 6), sample_dim=2, seed=1):
    from raytracer_amd.world import World
    world = World.__new__(World)
    world.overrides = dict(resolution=resolution, sample_dim=sample_dim)
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=seed)
    return world


def test_camera_render_matches_reference(torch_gpu):
    """camera.render (mean of squared radiance, camera.py:35-58): the integer triples handed
    to pixel_func equal the reference's on the 8x6 golden frame; render_progressive ends on
    the same frame after reporting every pixel once per sample pass."""
    g = common.golden("frame_order")["render_pixels"]
    world = _tiny_world("cornellbox_area")
    got = []
    world.camera.render(world, lambda coord, r, g_, b: got.append((coord[0], coord[1], r, g_, b)))
    got = numpy.array(got, dtype=numpy.int64)
    assert numpy.array_equal(got[:, :2], g[:, :2])                   # row-major pixel order
    # mean(c^2)*255 is truncated: a triple may differ by one from the reference's only where
    # that product sits within rounding of an integer (the CPU checker supplies the float)
    from raytracer_amd import flatten, runtime
    from oracle import oracle
    cam = flatten.camera_record(world)
    _, want_sq, _ = oracle.OracleScene(flatten.flatten_world(world)).render_tiles(
        cam, world.seed, numpy.arange(1, dtype=numpy.int32), 0, 4, threads=1)
    scaled = runtime.scatter_tiles(want_sq, numpy.arange(1), 8, 6).reshape(-1, 3) / 4.0 * 255
    near_integer = numpy.abs(scaled - numpy.rint(scaled)) < 1e-6 * numpy.maximum(1.0, numpy.abs(scaled))
    d = numpy.abs(got[:, 2:] - g[:, 2:])
    assert d.max() <= 1 and near_integer[d > 0].all()
    prog = []
    world.camera.render_progressive(world, lambda coord, r, g_, b: prog.append((coord[0], coord[1], r, g_, b)))
    prog = numpy.array(prog, dtype=numpy.int64)
    assert len(prog) == 4 * len(got)                                   # one report per pixel per pass
    assert numpy.abs(prog[-len(got):, 2:] - got[:, 2:]).max() <= 1


@pytest.mark.parametrize("scene", ("cornellbox_area", "mirrorhall_dof"))
def test_render_pixels_seam_reproduces_the_frame(torch_gpu, scene):
    """The worker loop of main.py:86-90 over the producer's chunks, averaged like
    main.py:99-106, gives the same image as the whole-frame tile kernel."""
    world = _tiny_world(scene, resolution=(24, 16), sample_dim=2, seed=5)
    sums = {}
    n_pairs = 0
    for chunk in world.camera.get_pixel_plane_point_pairs(world, 100):
        for pixel, r, g, b in world.camera.render_pixels(world, chunk):
            acc = sums.setdefault(tuple(pixel), numpy.zeros(4))
            acc += (r, g, b, 1)
            n_pairs += 1
    assert n_pairs == 24 * 16 * 4
    image = world.render(mode="parallel")
    for (col, row), acc in sums.items():
        mean = numpy.minimum(acc[:3] / acc[3], 255)
        assert numpy.array_equal(mean.astype(numpy.uint8), image[row, col])


def test_tracer_and_world_methods(torch_gpu):
    """world.tracer.trace_ray and world.hit_objects answer single rays from the host."""
    import tracer
    world = common.world_for("foursphere")
    g = common.golden("foursphere")
    k = int(numpy.nonzero(g["hit_object"] >= 0)[0][0])
    ray = tracer.Ray(g["origin"][k], g["direction"][k])
    from raytracer_amd import rng
    width = world.viewplane.resolution[0]
    rng.context.point(rng.PHASE_TRACE, int(g["pixel"][k, 1]) * width + int(g["pixel"][k, 0]), int(g["sample"][k]))
    rgb = world.tracer.trace_ray(ray)
    assert common.rgb_close(numpy.array(rgb), g["rgb"][k])[0]
    rec = world.hit_objects(ray)
    assert rec.object_index == g["hit_object"][k] and abs(rec.tmin - g["hit_t"][k]) < 1e-9 * g["hit_t"][k]
    miss = tracer.Ray((0., 50., 0.), (0., 1., 0.))
    assert world.hit_objects(miss) is None
    assert tuple(world.tracer.trace_ray(miss)) == tuple(world.background_color)


def test_partial_tiles_and_determinism(torch_gpu):
    """A frame whose size is not a multiple of 8 (edge tiles partly outside), rendered twice:
    identical integer frames, equal to the CPU oracle's, nothing written outside the frame."""
    from raytracer_amd import flatten, runtime
    from oracle import oracle
    world = _tiny_world("cornellbox_area", resolution=(44, 30), sample_dim=2, seed=9)
    a = world.render(mode="parallel")
    b = world.render(mode="parallel")
    assert a.shape == (30, 44, 3) and numpy.array_equal(a, b)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(44, 30)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    want, _, per = oracle.OracleScene(flatten.flatten_world(world)).render_tiles(
        cam, world.seed, tiles, 0, 4, per_sample_want=("rgb",), threads=8)
    ds = runtime.device_scene(world)
    fb = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, fb.tile_ids, 0, 4, sum_rgb8=fb.sum_rgb8)
    torch_gpu.cuda.synchronize()
    got = fb.sum_rgb8.cpu().numpy()
    common.assert_integer_sums_match(got, want, per["rgb"], len(tiles), 4, "44x30 frame")
    # the image World.render returns is those sums averaged and clamped (main.py:99-106)
    img = numpy.minimum(runtime.scatter_tiles(got, tiles, 44, 30) / 4.0, 255)
    assert numpy.array_equal(a, numpy.clip(img, 0, 255).astype(numpy.uint8))
    # rows of pixels outside the frame stay untouched
    outside = runtime.scatter_tiles(numpy.ones_like(got), tiles, 44, 30).sum() // 3
    assert outside == 44 * 30 and got.sum() == runtime.scatter_tiles(got, tiles, 44, 30).sum()
    # seeds matter, and only through the draws
    world2 = _tiny_world("cornellbox_area", resolution=(44, 30), sample_dim=2, seed=10)
    assert not numpy.array_equal(world2.render(mode="parallel"), a)


def test_trace_rays_start_depth(torch_gpu):
    """trace_ray(ray, depth): a ray started past the depth limit returns the background,
    one started at the limit is traced exactly once (tracer.py:26-35)."""
    world = common.world_for("mirrorhall")
    g = common.golden("mirrorhall")
    from raytracer_amd import runtime
    ds = runtime.device_scene(world)
    pix, smp, _, _ = common.golden_inputs("mirrorhall")
    limit = world.tracer.depth_limit
    over = ds.trace_rays(world.seed, g["origin"], g["direction"], pix, smp, depth=limit + 1)
    assert (over["n_rays"] == 0).all() and (over["rgb"] == numpy.array(world.background_color)).all()
    last = ds.trace_rays(world.seed, g["origin"], g["direction"], pix, smp, depth=limit)
    assert (last["n_rays"][:, 0] == 1).all()
    full = ds.trace_rays(world.seed, g["origin"], g["direction"], pix, smp, depth=0)
    assert numpy.array_equal(full["n_rays"], g["n_rays"])


def test_two_child_materials_are_checked_by_the_library(torch_gpu):
    """rtamd_scene_create refuses what the reference cannot run (Dielectric under a tracer
    without the three-argument trace_ray) and ray trees deeper than its per-sample list of
    waiting rays; rtamd_trace_rays refuses a negative start depth for such scenes."""
    import ctypes as C
    from raytracer_amd import capi, runtime
    world = common.world_for("features_glass")
    ds = runtime.device_scene(world)
    lib = capi.library()

    def create(**changes):
        desc = ds.flat.desc(lambda name, arr: ds.tensors[name].data_ptr())
        desc.tri_order = ds.tensors["tri_order"].data_ptr()
        desc.chunk_boxes = ds.tensors["chunk_boxes"].data_ptr()
        for k, v in changes.items():
            setattr(desc, k, v)
        handle = C.c_void_p()
        rc = lib.rtamd_scene_create(C.byref(desc), C.byref(handle))
        if rc == 0:
            lib.rtamd_scene_destroy(handle)
        return rc, lib.rtamd_last_error()
    assert create()[0] == 0
    rc, msg = create(tracer_kind=capi.TRACER_SHADE)
    assert rc != 0 and b"Whitted" in msg
    rc, msg = create(depth_limit=capi.MAX_BRANCH_LEVELS)
    assert rc != 0 and b"depth limit" in msg
    assert create(depth_limit=capi.MAX_BRANCH_LEVELS - 1)[0] == 0
    assert create(tracer_kind=capi.TRACER_AREA, depth_limit=50)[0] == 0     # area_light_shade does not recurse
    with pytest.raises(Exception, match="start depth"):
        ds.trace_rays(world.seed, numpy.zeros((1, 3)), numpy.array([[0., 0., -1.]]), depth=-1)
    # a start depth inside the tree works and matches the oracle
    from oracle import oracle
    g = common.golden("features_glass")
    pix, smp, _, _ = common.golden_inputs("features_glass")
    k = slice(0, 512)
    got = ds.trace_rays(world.seed, g["origin"][k], g["direction"][k], pix[k], smp[k], depth=2)
    want = oracle.OracleScene(common.flat_for("features_glass")).trace_rays(
        world.seed, g["origin"][k], g["direction"][k], pix[k], smp[k], depth=2)
    assert numpy.array_equal(got["n_rays"], want["n_rays"])
    close, worst = common.rgb_close(got["rgb"], want["rgb"])
    assert close, worst


def test_full_size_bench_frame_properties(torch_gpu):
    """The benchmark's own frame (dragon, 1920x1080, 16 samples per pixel, ambient occlusion +
    area light) at full size, through properties that need no reference run of that size:
    the timed kernel equals the counting kernel (which walks the tree as the reference does,
    without leaf boxes, short rounds or handed-over walks); the frame is the sum of its
    parts, split by tiles or by sample ranges; two runs are identical; the ray counts are
    those of the tracer (one closest-hit query per sample, at most two shadow queries); and
    randomly chosen whole tiles equal the CPU oracle's."""
    import bench
    from raytracer_amd import flatten, runtime
    from raytracer_amd.world import World
    from oracle import oracle
    torch = torch_gpu
    scene, overrides = bench.WORKLOADS["dragon_1080p_16spp_ao_arealight"]
    world = World.__new__(World)
    world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    assert (cam.width, cam.height) == (1920, 1080)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    n_samples = 16

    def render(tile_ids, s0, s1, into=None, **kw):
        fb = into if into is not None else runtime.FrameBuffers(ds, tile_ids)
        ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1, sum_rgb8=fb.sum_rgb8, **kw)
        torch.cuda.synchronize()
        return fb

    ds.read_counters(reset=True)
    whole = render(tiles, 0, n_samples)
    c = ds.read_counters(reset=True)
    assert c["samples"] == 1920 * 1080 * 16 and c["closest"] == c["samples"]
    assert c["samples"] < c["shadow"] <= 2 * c["samples"]
    again = render(tiles, 0, n_samples)
    assert torch.equal(again.sum_rgb8, whole.sum_rgb8)
    assert ds.read_counters(reset=True) == c
    counted = render(tiles, 0, n_samples, count_work=True)
    cc = ds.read_counters(reset=True)
    assert torch.equal(counted.sum_rgb8, whole.sum_rgb8)
    assert (cc["closest"], cc["shadow"]) == (c["closest"], c["shadow"]) and cc["tri_tests"] > 0
    # by sample ranges into one accumulator
    parts = render(tiles, 0, 4)
    render(tiles, 4, 12, into=parts)
    render(tiles, 12, 16, into=parts)
    assert torch.equal(parts.sum_rgb8, whole.sum_rgb8)
    # by tiles: the shares of three ranks, as the multi-GPU driver deals them
    from raytracer_amd import distributed
    img = runtime.scatter_tiles(whole.sum_rgb8.cpu().numpy(), tiles, cam.width, cam.height)
    total = numpy.zeros_like(img)
    for rank in range(3):
        share = distributed.tiles_of_rank(cam.width, cam.height, rank, 3)
        fb = render(share, 0, n_samples)
        total += runtime.scatter_tiles(fb.sum_rgb8.cpu().numpy(), share, cam.width, cam.height)
    assert numpy.array_equal(total, img)
    # random whole tiles against the oracle
    pick = numpy.sort(numpy.random.RandomState(3).choice(tx * ty, size=96, replace=False)).astype(numpy.int32)
    want, _, per = oracle.OracleScene(ds.flat).render_tiles(cam, world.seed, pick, 0, n_samples,
                                                            per_sample_want=("rgb",), threads=8)
    # tiles are listed in ascending order, so the whole frame's rows of tile t start at 64*t
    got = whole.sum_rgb8.cpu().numpy().reshape(-1, 64, 3)[pick].reshape(-1, 3)
    differing = common.assert_integer_sums_match(got, want, per["rgb"], len(pick), n_samples, "bench frame")
    print("bench frame vs oracle on %d tiles: %d knife-edge entries" % (len(pick), differing))


def test_kd_shortcuts_from_other_viewpoints(torch_gpu):
    """The dragon scene from five other viewpoints (around it, above it, inside its bounding
    box), other seeds: the timed kernel - leaf boxes, short rounds, walks handed between lanes,
    shared chunk loads - gives the counting kernel's frame exactly, and the counting kernel
    gives the oracle's primary hits, ray counts and work counters."""
    import bench
    from raytracer_amd import flatten, runtime
    from raytracer_amd.world import World
    from oracle import oracle
    torch = torch_gpu
    scene, _ = bench.WORKLOADS["dragon_1080p_16spp_ao_arealight"]
    world = World.__new__(World)
    world.overrides = dict(resolution=(240, 136), sample_dim=2)
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    osc = oracle.OracleScene(ds.flat)
    eye0 = numpy.array(world.camera.eye, dtype=float)
    look0 = numpy.array(world.camera.lookat, dtype=float)
    dist = numpy.linalg.norm(eye0 - look0)
    views = [
        (look0 + numpy.array([dist, 0.3 * dist, 0.0]), look0),
        (look0 + numpy.array([-0.7 * dist, 0.2 * dist, -0.7 * dist]), look0),
        (look0 + numpy.array([0.05 * dist, 1.2 * dist, 0.05 * dist]), look0),
        (look0 + 0.25 * (eye0 - look0), look0),
        (look0 + numpy.array([0.0, 0.02 * dist, 0.0]), look0 + numpy.array([dist, 0.1 * dist, 0.3 * dist])),
    ]
    tx, ty = runtime.tile_grid(240, 136)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    for i, (eye, lookat) in enumerate(views):
        world.camera.eye = numpy.array(eye)
        world.camera.lookat = numpy.array(lookat)
        world.camera.compute_uwv()
        cam = flatten.camera_record(world)
        seed = 100 + i
        n = len(tiles) * 64 * 4
        timed = runtime.FrameBuffers(ds, tiles)
        ds.render_tiles(cam, seed, timed.tile_ids, 0, 4, sum_rgb8=timed.sum_rgb8, sum_sq=timed.sum_sq)
        counted = runtime.FrameBuffers(ds, tiles)
        per = ds.sample_buffers(n, want=("hit_object", "hit_prim", "n_rays"))
        ds.read_counters(reset=True)
        ds.render_tiles(cam, seed, counted.tile_ids, 0, 4, sum_rgb8=counted.sum_rgb8, sum_sq=counted.sum_sq,
                        per_sample=per, count_work=True)
        torch.cuda.synchronize()
        work = ds.read_counters(reset=True)
        assert torch.equal(timed.sum_rgb8, counted.sum_rgb8), "view %d" % i
        assert torch.equal(timed.sum_sq, counted.sum_sq), "view %d" % i
        _, _, want = osc.render_tiles(cam, seed, tiles, 0, 4, per_sample_want=("hit_object", "hit_prim", "n_rays"),
                                      threads=8)
        assert work == osc.read_counters(reset=True), "view %d" % i
        for key in ("hit_object", "hit_prim", "n_rays"):
            assert numpy.array_equal(per[0][key].cpu().numpy(), want[key]), "view %d: %s" % (i, key)
        assert work["shadow"] > 0 and work["tri_tests"] > 0


@pytest.mark.parametrize("tables", ("canonical", "order_and_chunk_boxes", "no_leaf_aabb"))
def test_optional_device_tables_may_be_absent(torch_gpu, tables):
    """The derived tables of rtamd_scene_desc are optional.  A scene handle made without them
    renders the same frame as the full one: with the canonical tables alone (rows in scan
    order, every big leaf scanned in full), with only tri_order + chunk_boxes beside the
    re-arranged rows (the wave tests chunk boxes ray by ray), and with everything but
    leaf_aabb."""
    import ctypes as C
    from raytracer_amd import capi, flatten, runtime
    torch = torch_gpu
    world = _tiny_world("dragon_area", resolution=(96, 64), sample_dim=2, seed=4)
    ds = runtime.device_scene(world)
    lib = capi.library()
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(96, 64)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    full = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, full.tile_ids, 0, 4, sum_rgb8=full.sum_rgb8)

    keep = []
    if tables == "canonical":
        def address_of(name, arr):
            t = torch.from_numpy(numpy.ascontiguousarray(arr).view(numpy.uint8).reshape(-1).copy())
            t = (t if t.numel() else torch.zeros(16, dtype=torch.uint8)).to(ds.device)
            keep.append(t)
            return t.data_ptr()
        desc = ds.flat.desc(address_of)
        desc.kd_triples = ds.tensors["kd_triples"].data_ptr()   # node records do not depend on the row order...
        # ...except a leaf's chunk index, which is only read when chunk_boxes is given
    else:
        desc = ds.table_sizes(ds.flat.desc(lambda name, arr: ds.tensors[name].data_ptr()))
        desc.kd_triples = ds.tensors["kd_triples"].data_ptr()
        desc.tri_order = ds.tensors["tri_order"].data_ptr()
        desc.chunk_boxes = ds.tensors["chunk_boxes"].data_ptr()
        desc.n_chunks = ds.tensors["chunk_boxes"].numel() // 48
        if tables == "no_leaf_aabb":
            desc.tri_rows = ds.tensors["tri_rows"].data_ptr()
            desc.leaf_boxes = ds.tensors["leaf_boxes"].data_ptr()
            desc.chunk_tree = ds.tensors["chunk_tree"].data_ptr()
    handle = C.c_void_p()
    capi.check(lib.rtamd_scene_create(C.byref(desc), C.byref(handle)), lib)
    try:
        other = runtime.FrameBuffers(ds, tiles)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        capi.check(lib.rtamd_render_tiles(handle, C.byref(cam), world.seed, len(tiles), other.tile_ids.data_ptr(),
                                          0, 4, None, other.sum_rgb8.data_ptr(), None, None, 0, stream), lib)
        torch.cuda.synchronize()
        assert torch.equal(other.sum_rgb8, full.sum_rgb8)
        assert int(full.sum_rgb8.sum()) > 0
    finally:
        lib.rtamd_scene_destroy(handle)


def test_two_tree_objects_under_the_area_tracer(torch_gpu):
    """The whole-frame kernel variant for AreaLightTracer scenes asks the scene's one kd-tree
    last in its shadow tests; with two tree objects (here the Cornell box and a shifted
    instance of it) the general kernel must run instead.  Timed = counting = oracle."""
    import affinetransform as transform
    from geometry import Instance
    from raytracer_amd import flatten, runtime
    from oracle import oracle
    torch = torch_gpu
    world = _tiny_world("cornellbox_area", resolution=(64, 48), sample_dim=2, seed=6)
    trees = [o for o in world.objects if type(o).__name__ == "KDTree"]
    assert len(trees) == 1
    world.objects.append(Instance(trees[0], transform.translate(0.35, 0.0, -0.4)))
    world.invalidate()
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(64, 48)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    timed = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, timed.tile_ids, 0, 4, sum_rgb8=timed.sum_rgb8)
    counted = runtime.FrameBuffers(ds, tiles)
    ds.read_counters(reset=True)
    ds.render_tiles(cam, world.seed, counted.tile_ids, 0, 4, sum_rgb8=counted.sum_rgb8, count_work=True)
    torch.cuda.synchronize()
    work = ds.read_counters(reset=True)
    assert torch.equal(timed.sum_rgb8, counted.sum_rgb8)
    osc = oracle.OracleScene(ds.flat)
    want, _, per = osc.render_tiles(cam, world.seed, tiles, 0, 4, per_sample_want=("rgb",), threads=8)
    assert work == osc.read_counters(reset=True)
    common.assert_integer_sums_match(counted.sum_rgb8.cpu().numpy(), want, per["rgb"], len(tiles), 4, "two trees")


@pytest.mark.parametrize("scene,overrides,oracle_tiles", [
    ("glossyreflective", None, 24),                                                   # 400x400, 100 samples per pixel
    ("mirrorhall_dof", dict(resolution=(3840, 2160), sample_dim=8), 6),               # 4K, 64 samples per pixel
])
def test_full_size_analytic_frames(torch_gpu, scene, overrides, oracle_tiles):
    """The two full-size configurations without a kd-tree: the timed kernel, the kernel that counts
    the reference's work and the one that counts executed work render the same integer frame;
    split over sample ranges it is the same frame again; random tiles equal the oracle's."""
    from raytracer_amd import flatten, runtime
    from raytracer_amd.world import World
    from oracle import oracle
    torch = torch_gpu
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    ns = len(world.viewplane.sampler.patterns[0])

    def render(s0, s1, into=None, **kw):
        fb = into if into is not None else runtime.FrameBuffers(ds, tiles)
        ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1, sum_rgb8=fb.sum_rgb8, **kw)
        torch.cuda.synchronize()
        return fb

    ds.read_counters(reset=True)
    whole = render(0, ns)
    c = ds.read_counters(reset=True)
    assert c["samples"] == cam.width * cam.height * ns and c["closest"] >= c["samples"]
    counted = render(0, ns, count_work=True)
    work = ds.read_counters(reset=True)
    assert torch.equal(counted.sum_rgb8, whole.sum_rgb8)
    executed = render(0, ns, count_executed=True)
    done = ds.read_counters(reset=True, executed=True)
    assert torch.equal(executed.sum_rgb8, whole.sum_rgb8)
    assert (done["closest"], done["shadow"]) == (c["closest"], c["shadow"]) == (work["closest"], work["shadow"])
    assert done["analytic_tests"] == work["analytic_tests"] > 0 and done["bytes_loaded"] == 48 * done["analytic_tests"]
    parts = render(0, ns // 2)
    render(ns // 2, ns, into=parts)
    assert torch.equal(parts.sum_rgb8, whole.sum_rgb8)
    del counted, executed, parts
    pick = numpy.sort(numpy.random.RandomState(8).choice(tx * ty, size=oracle_tiles, replace=False)).astype(numpy.int32)
    want, _, per = oracle.OracleScene(ds.flat).render_tiles(cam, world.seed, pick, 0, ns, per_sample_want=("rgb",), threads=8)
    got = whole.sum_rgb8.cpu().numpy().reshape(-1, 64, 3)[pick].reshape(-1, 3)
    common.assert_integer_sums_match(got, want, per["rgb"], len(pick), ns, scene)


@pytest.mark.parametrize("scene,sample_dim", [("glossyreflective", 2), ("glossyreflective", 3), ("glossyreflective", 5),
                                              ("mirrorhall_dof", 4), ("foursphere", 6), ("features_analytic", 4)])
def test_refilled_waves_render_the_same_frame(torch_gpu, scene, sample_dim):
    """Whole frames of scenes without a kd-tree, accumulators only, are rendered by waves that own
    several 64-sample groups and refill lanes whose paths have ended (where the sample count has a
    divisor for it; 9 samples have none up to 16 but 3 and 9, 25 have 5); per-sample outputs force
    the one-sample-per-lane kernel.  Both must agree sample for sample: equal integer sums, equal
    ray and sample counts, squared sums within rounding, on a frame with partly empty edge tiles
    and across a split of the sample range."""
    from raytracer_amd import flatten, runtime
    torch = torch_gpu
    import sampler
    world = _tiny_world(scene, seed=3)
    world.viewplane.resolution = (44, 30)                  # whatever the scene file chose
    world.viewplane.sampler = sampler.MultiJitteredSampler(sample_dim)
    world.invalidate()
    ds = runtime.device_scene(world)
    assert len(ds.flat.arrays["trees"]) == 0
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(44, 30)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    ns = sample_dim * sample_dim
    ds.read_counters(reset=True)
    fb = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, fb.tile_ids, 0, ns, sum_rgb8=fb.sum_rgb8, sum_sq=fb.sum_sq)
    torch.cuda.synchronize()
    fast = ds.read_counters(reset=True)
    ref = runtime.FrameBuffers(ds, tiles)
    per = ds.sample_buffers(len(tiles) * 64 * ns, want=("rgb", "rgb8"))
    ds.render_tiles(cam, world.seed, ref.tile_ids, 0, ns, sum_rgb8=ref.sum_rgb8, sum_sq=ref.sum_sq, per_sample=per)
    torch.cuda.synchronize()
    slow = ds.read_counters(reset=True)
    assert fast == slow and fast["samples"] == 44 * 30 * ns
    assert torch.equal(fb.sum_rgb8, ref.sum_rgb8)
    # rows of the per-sample outputs are (tile, sample, pixel in tile)
    rgb8 = per[0]["rgb8"].reshape(len(tiles), ns, 64, 3).to(torch.int64).sum(dim=1).reshape(-1, 3)
    assert torch.equal(fb.sum_rgb8, rgb8)
    assert torch.allclose(fb.sum_sq, ref.sum_sq, rtol=1e-12, atol=1e-300)
    assert int(fb.sum_rgb8.sum()) > 0
    # the same samples in two launches (their counts have other divisors)
    cut = ns // 2 + 1
    two = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, two.tile_ids, 0, cut, sum_rgb8=two.sum_rgb8)
    ds.render_tiles(cam, world.seed, two.tile_ids, cut, ns, sum_rgb8=two.sum_rgb8)
    torch.cuda.synchronize()
    assert torch.equal(two.sum_rgb8, fb.sum_rgb8)
    # a sublist of tiles in another order lands in the rows of that list
    some = tiles[::-3].copy()
    part = runtime.FrameBuffers(ds, some)
    ds.render_tiles(cam, world.seed, part.tile_ids, 0, ns, sum_rgb8=part.sum_rgb8)
    torch.cuda.synchronize()
    want = fb.sum_rgb8.reshape(len(tiles), 64, 3)[torch.as_tensor(some.astype(numpy.int64), device=fb.sum_rgb8.device)]
    assert torch.equal(part.sum_rgb8.reshape(len(some), 64, 3), want)


@pytest.mark.parametrize("scene", ("j", "k", "features_mesh", "mirror", "cornellbox", "fuzzmesh_1"))
def test_accumulator_only_frames_of_kd_scenes(torch_gpu, scene):
    """Whole frames of kd scenes with accumulators only go to kernel builds of their own where the
    scene allows (materials that spawn no child ray under the plain tracer: `j`, `k` - one tree and
    one light - and `features_mesh`; the path tracer: `cornellbox` as shipped; `mirror` has a mirror
    floor and stays with the general kernel): they must render what the general kernel reports
    sample by sample."""
    from raytracer_amd import flatten, runtime
    torch = torch_gpu
    world = common.world_for(scene)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)[::7].copy()
    ns = len(world.viewplane.sampler.patterns[0])
    ds.read_counters(reset=True)
    fb = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, fb.tile_ids, 0, ns, sum_rgb8=fb.sum_rgb8, sum_sq=fb.sum_sq)
    torch.cuda.synchronize()
    fast = ds.read_counters(reset=True)
    per = ds.sample_buffers(len(tiles) * 64 * ns, want=("rgb", "rgb8"))
    ref = runtime.FrameBuffers(ds, tiles)
    ds.render_tiles(cam, world.seed, ref.tile_ids, 0, ns, sum_rgb8=ref.sum_rgb8, sum_sq=ref.sum_sq, per_sample=per)
    torch.cuda.synchronize()
    slow = ds.read_counters(reset=True)
    assert fast == slow and fast["closest"] >= fast["samples"] > 0
    assert torch.equal(fb.sum_rgb8, ref.sum_rgb8)
    rgb8 = per[0]["rgb8"].reshape(len(tiles), ns, 64, 3).to(torch.int64).sum(dim=1).reshape(-1, 3)
    assert torch.equal(fb.sum_rgb8, rgb8)
    assert torch.allclose(fb.sum_sq, ref.sum_sq, rtol=1e-12, atol=1e-300)
    assert int(fb.sum_rgb8.sum()) > 0


def test_executed_work_counters_on_the_dragon(torch_gpu):
    """RTAMD_COUNT_EXECUTED runs the fast path and tallies it: same frame as the timed kernel, far
    fewer triangle tests than the reference's scan, and loads that add up."""
    from raytracer_amd import flatten, runtime
    torch = torch_gpu
    world = common.world_for("dragon_area")
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    frames = {}
    for mode in ("timed", "executed", "reference"):
        fb = runtime.FrameBuffers(ds, tiles)
        ds.read_counters(reset=True)
        ds.render_tiles(cam, world.seed, fb.tile_ids, 0, 16, sum_rgb8=fb.sum_rgb8,
                        count_executed=mode == "executed", count_work=mode == "reference")
        torch.cuda.synchronize()
        frames[mode] = (fb.sum_rgb8, ds.read_counters(reset=True, executed=True))
    assert torch.equal(frames["timed"][0], frames["executed"][0]) and torch.equal(frames["timed"][0], frames["reference"][0])
    done, ref = frames["executed"][1], frames["reference"][1]
    assert (done["closest"], done["shadow"]) == (ref["closest"], ref["shadow"])
    assert 0 < done["tri_tests"] < ref["tri_tests"] / 4                 # boxes cull most of the reference's tests
    assert done["rows_loaded"] <= done["tri_tests"]                      # a loaded row may serve several rays
    assert done["leaves_skipped"] <= done["kd_leaves"] and done["node_records"] > 0 and done["boxes_tested"] > 0
    floor = 48 * done["node_records"] + 32 * done["boxes_tested"] + 72 * done["rows_loaded"] + 48 * done["analytic_tests"]
    assert floor <= done["bytes_loaded"] <= floor + 16 * done["boxes_tested"] + 8 * done["rows_loaded"]
    # both flags at once are refused
    from raytracer_amd import capi
    with pytest.raises(RuntimeError):
        fb = runtime.FrameBuffers(ds, tiles)
        ds.render_tiles(cam, world.seed, fb.tile_ids, 0, 16, sum_rgb8=fb.sum_rgb8, count_executed=True, count_work=True)


def shadow_path_world(case):
    """The worlds of test_shadow_tree_paths_of_the_whole_frame_kernel: (world, tiles to check, sample range)."""
    import numpy as np
    import raytracer_amd
    raytracer_amd.activate()
    from geometry import Rectangle
    from light import AmbientOccluder, AreaLight
    from material import Emissive, Phong
    if case == "tiny_instance":
        # the teapot at a fiftieth of its size: in its own space the shadow rays of the ground
        # start hundreds of teapots away, too far for a single-precision walk from their origin:
        # it starts from a point moved along the ray to just outside the tree's box
        import affinetransform as transform
        from geometry import Instance
        from raytracer_amd.world import World
        world = World(viewmode="parallel", buildfunction="k", seed=3)
        world.objects[0] = Instance(world.objects[0], transform.translate(0.5, 0.0, 4.5).dot(transform.scale(0.02, 0.02, 0.02)))
        tiles, samples = 625, None
    elif case == "teapot_smooth":
        from raytracer_amd.world import World
        from tracer import AreaLightTracer
        world = World(viewmode="parallel", buildfunction="k", seed=3)
        sampler = world.viewplane.sampler
        world.tracer = AreaLightTracer(world)
        world.ambient_color = AmbientOccluder(np.array((1., 1., 1.)), .4, sampler)
        panel = Rectangle(np.array((-2., 6., 7.)), 4, 3, np.array((0., 0., -1.)), np.array((1., 0., 0.)),
                          Emissive(20., np.array((1., .95, .9))), sampler)
        world.lights = [AreaLight(panel.material, panel, 5.)]
        tiles, samples = 200, None
    else:
        world = _tiny_world("dragon_area", resolution=(256, 144), sample_dim=2, seed=5)
        sampler = world.viewplane.sampler
        if case == "panel_inside_the_box":
            # the emitter hangs between the dragon's coils: light rays end inside the tree's box
            panel = Rectangle(np.array((-3., 4., 1.)), 3, 2, np.array((0., 0., -1.)), np.array((1., 0., 0.)),
                              Emissive(30., np.array((1., .96, .9))), sampler)
            world.lights = [AreaLight(panel.material, panel, 6.)]
        elif case in ("panel_beside_the_box", "panel_far_from_the_box"):
            # an emitter outside the tree's box (z up to 3.15): far enough and the library drops the
            # general path from the kernel it launches (it shows on the host that every light ray leaves
            # the box before it ends); a hair outside and it must keep it
            z = 3.2 if case == "panel_beside_the_box" else 6.0
            panel = Rectangle(np.array((-3., 4., z)), 3, 2, np.array((0., 0., -1.)), np.array((1., 0., 0.)),
                              Emissive(30., np.array((1., .96, .9))), sampler)
            world.lights = [AreaLight(panel.material, panel, 6.)]
        elif case == "panel_between_floor_and_box":
            # an emitter beside the box along x (the box ends at x = 7.05) that faces +z: the floor beyond it
            # (x > 9.6, z > 0) is lit, and the rays from there to the emitter point at the box behind it -
            # they reach the light first, so nothing of the tree may shadow them
            panel = Rectangle(np.array((7.6, .5, 0.)), 2, 2, np.array((0., 0., 1.)), np.array((1., 0., 0.)),
                              Emissive(30., np.array((1., .96, .9))), sampler)
            world.lights = [AreaLight(panel.material, panel, 6.)]
        else:
            tree = world.objects[0]
            shiny = Phong(ka=.6, kd=.7, ks=.5, exp=30, cd=np.array((1., .84, .1)))
            for box in tree.boxes:
                box.obj.material = shiny
        tiles, samples = (576 if case == "panel_between_floor_and_box" else 120), None
    return world, tiles, samples


@pytest.mark.parametrize("case", ("panel_inside_the_box", "panel_beside_the_box", "panel_far_from_the_box",
                                  "panel_between_floor_and_box", "phong_dragon", "teapot_smooth", "tiny_instance"))
def test_shadow_tree_paths_of_the_whole_frame_kernel(torch_gpu, case):
    """One kd-tree, an ambient occluder, one light, AreaLightTracer: the whole-frame kernel traces
    both shadow rays of a hit in one query on the mesh's shadow tree - unless some light ray
    ends inside the mesh's box, which sends the wave down the general path.  Both ways, an emitter
    just outside and well outside the box, a glossy material and a smooth mesh: timed =
    executed-work = reference-work kernel = oracle."""
    from raytracer_amd import flatten
    world, tiles, samples = shadow_path_world(case)
    world.invalidate()
    check_tiles_against_oracle(torch_gpu, world, flatten.flatten_world(world), tiles, samples, case)


def test_pack_sums_saturates_and_clears(torch_gpu):
    """rtamd_pack_sums: a rank's 64-bit sums become the int32 values that travel to rank 0 (saturated: a sum only
    leaves the int32 range where samples were infinite, and such a pixel is clamped on arrival anyway,
    main.py:104-106); with `clear` the sums are zeroed in the same pass.  Odd sizes, nothing to do, bad arguments."""
    import ctypes as C
    from raytracer_amd import capi, runtime
    torch = torch_gpu
    ds = runtime.device_scene(common.world_for("foursphere"))
    big = 2 ** 31
    for n in (1, 255, 256, 257, 3 * 64 * 1000 + 5):
        g = torch.Generator().manual_seed(n)
        sums = torch.randint(-1000, 70000, (n,), generator=g, dtype=torch.int64)
        sums[::7] = torch.tensor([big - 1, big, big + 12345, -big, -big - 1, 2 ** 62, -2 ** 62])[torch.arange(len(sums[::7])) % 7]
        want = sums.clamp(-big, big - 1).to(torch.int32)
        for clear in (False, True):
            dev = sums.clone().cuda()
            wire = torch.full((n,), 77, dtype=torch.int32, device="cuda")
            ds.pack_sums(dev, wire, clear=clear)
            torch.cuda.synchronize()
            assert torch.equal(wire.cpu(), want)
            assert torch.equal(dev.cpu(), torch.zeros_like(sums) if clear else sums)
    lib = capi.library()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rtamd_pack_sums(None, None, 0, 1, stream) == 0                 # nothing to do
    assert lib.rtamd_pack_sums(None, None, 5, 1, stream) == -1                # RTAMD_EINVAL
    assert b"null" in lib.rtamd_last_error()
    assert lib.rtamd_pack_sums(None, None, -1, 0, stream) == -1


@pytest.mark.parametrize("scene", ("mirrorhall_dof", "cornellbox_area"))
def test_one_launch_of_more_than_two_to_the_31_samples(torch_gpu, scene):
    """A frame of 8192x4320 pixels at 64 samples per pixel is 2.26e9 samples in one launch: sample numbers beyond
    the 32-bit range (the mirror hall: waves that refill their lanes; the Cornell box: one sample per lane, a
    kd-tree).  Tiles from the start, the middle and the end of that launch equal the same tiles rendered alone."""
    from raytracer_amd import flatten, runtime
    torch = torch_gpu
    world = _tiny_world(scene, resolution=(8192, 4320), sample_dim=8)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    n_tiles = tx * ty
    assert n_tiles * 64 * 64 > 2 ** 31
    whole = runtime.FrameBuffers(ds, numpy.arange(n_tiles, dtype=numpy.int32))
    ds.render_tiles(cam, world.seed, whole.tile_ids, 0, 64, sum_rgb8=whole.sum_rgb8)
    some = numpy.array([0, 1, tx - 1, n_tiles // 2 + 17, n_tiles - tx, n_tiles - 2, n_tiles - 1], dtype=numpy.int32)
    alone = runtime.FrameBuffers(ds, some)
    ds.render_tiles(cam, world.seed, alone.tile_ids, 0, 64, sum_rgb8=alone.sum_rgb8)
    torch.cuda.synchronize()
    rows = (torch.as_tensor(some, dtype=torch.int64)[:, None] * 64 + torch.arange(64)[None, :]).reshape(-1).to(whole.sum_rgb8.device)
    assert torch.equal(whole.sum_rgb8[rows], alone.sum_rgb8)
    assert int((whole.sum_rgb8.sum(dim=1) > 0).sum()) > n_tiles * 32        # the frame is not empty
