#!/usr/bin/env python3
"""Where along its path one sample's radiance differs between the GPU and the CPU checker:
  python tests/debug_sample.py SCENE SAMPLE_ROW [N_TILES]
SAMPLE_ROW is the row `k` tests/debug_mismatch.py prints.  Renders that one sample with the tracer's
depth limit lowered step by step and with every light alone.  Needs a GPU."""
import os
import sys

import numpy
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raytracer_amd import flatten, runtime      # noqa: E402
from raytracer_amd.world import World           # noqa: E402
from oracle import oracle                       # noqa: E402

scene, k = sys.argv[1], int(sys.argv[2])
ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 96
World(viewmode="parallel", buildfunction="fuzz_0", seed=1)
fuzz = [m for name, m in sys.modules.items() if name.startswith("rtamd_scene_fuzz")][0]
if scene.startswith("fuzzmesh_") and int(scene.split("_")[1]) >= fuzz.N_MESH_WORLDS:
    fuzz._make_mesh(int(scene.split("_")[1]))
elif scene.startswith("fuzz_") and int(scene.split("_")[1]) >= fuzz.N_WORLDS:
    fuzz._make(int(scene.split("_")[1]))
world = World(viewmode="parallel", buildfunction=scene, seed=1)
cam = flatten.camera_record(world)
tx, ty = runtime.tile_grid(cam.width, cam.height)
rng = numpy.random.RandomState(5)
tile_ids = numpy.sort(rng.choice(tx * ty, size=min(ntiles, tx * ty), replace=False)).astype(numpy.int32)
tile_ids[-1] = tx * ty - 1
ns = len(world.viewplane.sampler.patterns[0])
tpos, rest = divmod(k, 64 * ns)
sample, pix = divmod(rest, 64)
tile = int(tile_ids[tpos])
col, row = (tile % tx) * 8 + pix % 8, (tile // tx) * 8 + pix // 8
print("sample row", k, "-> pixel", (col, row), "sample", sample)
pp, lens = runtime.plane_points_for(world, world.camera, world.seed, sample)
pixel_index = numpy.array([row * cam.width + col], dtype=numpy.int32)
sample_index = numpy.array([sample], dtype=numpy.int32)
plane = pp[row, col][None, :].copy()
lens_pt = lens[row, col][None, :].copy() if lens is not None else None


def both(label):
    world.invalidate()
    ds = runtime.device_scene(world)
    flat = ds.flat
    got = ds.render_pixels(flatten.camera_record(world), world.seed, pixel_index, sample_index, plane, lens_pt)
    torch.cuda.synchronize()
    g = got
    want = oracle.OracleScene(flat).render_pixels(flatten.camera_record(world), world.seed, pixel_index, sample_index, plane, lens_pt)
    d = numpy.abs(g["rgb"][0] - want["rgb"][0])
    print("%-28s gpu %s cpu %s  |diff| %s  rays %s %s  t %r %r" % (label, g["rgb"][0], want["rgb"][0], d, g["n_rays"][0], want["n_rays"][0],
                                                                float(g["hit_t"][0]), float(want["hit_t"][0])))


limit = world.tracer.depth_limit if hasattr(world.tracer, "depth_limit") else None
print("tracer", type(world.tracer).__name__, "depth limit", limit, "lights", [type(l).__name__ for l in world.lights],
      "ambient", type(world.ambient_color).__name__)
both("as built")
if limit is not None:
    for d in range(limit - 1, -1, -1):
        world.tracer.depth_limit = d
        both("depth limit %d" % d)
    world.tracer.depth_limit = limit
lights = list(world.lights)
for i, l in enumerate(lights):
    world.lights = [l]
    both("only light %d (%s)" % (i, type(l).__name__))
world.lights = []
both("no lights (ambient only)")
from raytracer_amd import capi
import ctypes as C
objs = flatten.flatten_world(world).arrays["objects"]
recs = (capi.Object * (len(objs) // C.sizeof(capi.Object))).from_buffer(bytearray(objs))
print("objects:", [(o.kind, o.has_transform, o.n_inner, o.tree, o.smooth) for o in recs])
mats = flatten.flatten_world(world).arrays["materials"]
mrec = (capi.Material * (len(mats) // C.sizeof(capi.Material))).from_buffer(bytearray(mats))
print("material kinds:", [m.kind for m in mrec])
