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
    ("dragon_area", 40, (0, 4)),
    ("dragon_area", 420, (0, 16)),
    ("j", 600, (0, 1)),
])
def test_gpu_tiles_match_oracle(torch_gpu, scene, tiles, samples):
    """Whole tiles through rtamd_render_tiles (ray generation on the device) against the
    CPU oracle: per-sample ids and counts exact, integer accumulators identical."""
    from raytracer_amd import flatten, runtime
    from oracle import oracle
    torch = torch_gpu
    world = common.world_for(scene)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    rng = numpy.random.RandomState(5)
    tile_ids = numpy.sort(rng.choice(tx * ty, size=min(tiles, tx * ty), replace=False)).astype(numpy.int32)
    # always include the last tile (partial when the frame is not a multiple of 8)
    tile_ids[-1] = tx * ty - 1
    s0, s1 = samples
    n = len(tile_ids) * 64 * (s1 - s0)
    fb = runtime.FrameBuffers(ds, tile_ids)
    per = ds.sample_buffers(n)
    ds.read_counters(reset=True)
    ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1, sum_rgb8=fb.sum_rgb8, sum_sq=fb.sum_sq, per_sample=per,
                    count_work=True)
    torch.cuda.synchronize()
    gpu_work = ds.read_counters(reset=True)
    got = {k: v.cpu().numpy() for k, v in per[0].items()}
    osc = oracle.OracleScene(common.flat_for(scene))
    want_rgb8, want_sq, want = osc.render_tiles(cam, world.seed, tile_ids, s0, s1,
                                                per_sample_want=("rgb", "hit_object", "hit_prim", "hit_t", "n_rays"),
                                                threads=8)
    assert numpy.array_equal(got["hit_object"], want["hit_object"])
    assert numpy.array_equal(got["hit_prim"], want["hit_prim"])
    assert numpy.array_equal(got["n_rays"], want["n_rays"])
    # the counting launch reports the work of the reference algorithm (kd steps, leaves,
    # triangle and analytic tests), whatever shortcuts the kernel takes: identical to the oracle's
    assert gpu_work == osc.read_counters(reset=True)
    close, worst = common.rgb_close(got["rgb"], want["rgb"])
    assert close, "worst relative error %g" % worst
    # int(c*255) of one sample may differ by one only where c*255 is within rounding of an
    # integer (ambient-only pixels: 0.2*255 = 51.0 +- 1 ulp); a pixel's integer sum may
    # therefore differ by at most its number of such samples, and is identical otherwise
    scaled = want["rgb"] * 255
    with numpy.errstate(invalid="ignore"):
        edge = numpy.abs(scaled - numpy.rint(scaled)) < 1e-6 * numpy.maximum(1.0, numpy.abs(scaled))
    ns = s1 - s0
    edge_per_pixel = edge.reshape(len(tile_ids), ns, 64, 3).sum(axis=1).reshape(-1, 3)
    d = numpy.abs(fb.sum_rgb8.cpu().numpy() - want_rgb8)
    assert (d <= edge_per_pixel).all(), "integer accumulators differ beyond rounding knife-edges"
    assert numpy.allclose(fb.sum_sq.cpu().numpy(), want_sq, rtol=1e-8, atol=1e-12)
    # the timed (non-counting) kernel variant gives the same frame
    fb2 = runtime.FrameBuffers(ds, tile_ids)
    ds.render_tiles(cam, world.seed, fb2.tile_ids, s0, s1, sum_rgb8=fb2.sum_rgb8)
    torch.cuda.synchronize()
    assert torch.equal(fb2.sum_rgb8, fb.sum_rgb8)
    print("%s: %d samples, worst relative radiance error vs oracle %.3g, accumulator mismatches %d"
          % (scene, n, worst, int((d > 0).sum())))


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
