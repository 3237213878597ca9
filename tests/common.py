"""Helpers shared by the parity tests: golden fixtures, scene construction, comparisons."""
import ast
import functools
import os

import numpy

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")

GOLDEN_SCENES = ("foursphere", "glossyreflective", "mirrorhall", "mirrorhall_dof", "cornellbox",
                 "cornellbox_area", "features_analytic", "features_mesh", "features_env", "features_glass",
                 "features_pane", "features_pile", "features_texture", "glass_teapot", "j", "dragon_area",
                 "k", "mirror", "mirroralt", "features_nested",
                 "fuzz_0", "fuzz_1", "fuzz_2", "fuzz_3", "fuzz_4", "fuzz_5", "fuzz_6", "fuzz_7", "fuzz_8",
                 "fuzzmesh_0", "fuzzmesh_1", "fuzzmesh_2", "fuzzmesh_10",
                 "glossyreflective@realtime", "mirrorhall@realtime",
                 # benchmark configurations 4 and 5 at their full sizes (sparse samples of the big frames)
                 "dragon_area@1080p", "mirrorhall_dof@4k")

# float comparison bound for radiance: the tolerance of the task statement is 1e-4
# relative per channel (fp32 device vs Python float64); this build computes in double, so
# the tests hold it to a far tighter bound and print the observed maximum.
RGB_RTOL = 1e-6
RGB_ATOL = 1e-9


@functools.lru_cache(maxsize=None)
def golden(name):
    with numpy.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@functools.lru_cache(maxsize=None)
def world_for(name):
    """The world a golden file was recorded on (same scene file, view mode, overrides, seed)."""
    from raytracer_amd.world import World
    g = golden(name)
    overrides = ast.literal_eval(str(g["overrides"]))
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode=str(g["viewmode"]), buildfunction=name.split("@")[0], seed=int(g["seed"]))
    return world


@functools.lru_cache(maxsize=None)
def flat_for(name):
    from raytracer_amd import flatten
    return flatten.flatten_world(world_for(name))


def golden_inputs(name):
    g = golden(name)
    w = world_for(name)
    width = w.viewplane.resolution[0]
    pix = (g["pixel"][:, 1].astype(numpy.int64) * width + g["pixel"][:, 0]).astype(numpy.int32)
    thin = getattr(w.camera, "IS_THIN_LENS", False)
    return pix, g["sample"].astype(numpy.int32), g["plane_point"], (g["lens_point"] if thin else None)


def rgb_close(a, b, rtol=RGB_RTOL, atol=RGB_ATOL):
    """(all close?, worst relative error) treating NaN==NaN and inf==inf as equal."""
    a = numpy.asarray(a, dtype=numpy.float64)
    b = numpy.asarray(b, dtype=numpy.float64)
    same = (a == b) | (numpy.isnan(a) & numpy.isnan(b))
    with numpy.errstate(invalid="ignore", divide="ignore"):
        err = numpy.abs(a - b)
        rel = numpy.where(same, 0.0, err / numpy.maximum(numpy.abs(b), atol / rtol))
    # (an infinite value only equals the same infinity: inf <= rtol * inf must not pass for a finite partner)
    ok = same | (numpy.isfinite(a) & numpy.isfinite(b) & (err <= atol + rtol * numpy.abs(b)))
    return bool(ok.all()), float(numpy.nanmax(rel)) if rel.size else 0.0


def compare_with_golden(name, out, ids_exact=True):
    """Assert a render_pixels-style result dict against the golden record of a scene."""
    g = golden(name)
    ok = g["ok"]
    assert ok.sum() > 0.9 * len(ok)
    obj_match = out["hit_object"][ok] == g["hit_object"][ok]
    prim_match = out["hit_prim"][ok] == g["hit_prim"][ok]
    assert obj_match.all(), "%s: %d primary object ids differ" % (name, (~obj_match).sum())
    assert prim_match.all(), "%s: %d primary primitive ids differ" % (name, (~prim_match).sum())
    hit = ok & (g["hit_object"] >= 0)
    t_rel = numpy.abs(out["hit_t"][hit] - g["hit_t"][hit]) / numpy.abs(g["hit_t"][hit])
    assert t_rel.max() < 1e-9, "%s: tmin differs by %g" % (name, t_rel.max())
    close, worst = rgb_close(out["rgb"][ok], g["rgb"][ok])
    assert close, "%s: radiance differs, worst relative error %g" % (name, worst)
    assert numpy.array_equal(out["n_rays"][ok], g["n_rays"][ok]), "%s: ray counts differ" % name
    if "rgb8" in out:
        # int(c*255) may differ by one only where c*255 sits within rounding of an integer
        # (e.g. ambient-only pixels with c = 0.2 -> 51.0 +- 1 ulp)
        d = numpy.abs(out["rgb8"][ok].astype(numpy.int64) - g["rgb8"][ok])
        scaled = g["rgb"][ok] * 255
        near_integer = numpy.abs(scaled - numpy.rint(scaled)) < 1e-6 * numpy.maximum(1.0, numpy.abs(scaled))
        assert d.max() <= 1 and near_integer[d > 0].all(), "%s: quantised colours differ" % name
    return worst


def knife_edge_budget(rgb, n_tiles, n_samples):
    """Per pixel and channel, how many of its samples have c*255 within rounding of an integer.
    `rgb` is the oracle's per-sample radiance in (tile, sample, pixel-in-tile) order.  Only
    such a sample's int(c*255) may differ by one between two float64 evaluations of the same
    path (ambient-only pixels: 0.2*255 = 51.0 +- 1 ulp), so a pixel's integer sum may differ
    by at most this number and must be identical where it is zero."""
    scaled = numpy.asarray(rgb, dtype=numpy.float64) * 255
    with numpy.errstate(invalid="ignore"):
        edge = numpy.abs(scaled - numpy.rint(scaled)) < 1e-6 * numpy.maximum(1.0, numpy.abs(scaled))
    return edge.reshape(n_tiles, n_samples, 64, 3).sum(axis=1).reshape(-1, 3)


def assert_integer_sums_match(got_sum, want_sum, want_rgb, n_tiles, n_samples, what=""):
    """Integer accumulators against the oracle's under the knife-edge rule; returns the number
    of entries that differ (all of them on knife edges)."""
    d = numpy.abs(numpy.asarray(got_sum, dtype=numpy.int64) - numpy.asarray(want_sum, dtype=numpy.int64))
    budget = knife_edge_budget(want_rgb, n_tiles, n_samples)
    bad = d > budget
    assert not bad.any(), "%s: %d accumulator entries differ beyond rounding knife-edges (worst by %d)" % (
        what, int(bad.sum()), int((d - budget).max()))
    return int((d > 0).sum())
