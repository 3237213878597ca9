#!/usr/bin/env python3
"""Where does the timed kernel's frame of one world differ from the counting kernel's (which follows the
reference's walk step by step)?
  python tests/debug_fast_path.py SCENE
Prints the pixels whose integer sums differ (accumulator-only launch and per-sample launch of the timed
build against the counting launch) and the samples of those pixels.  Needs a GPU."""
import os
import sys

import numpy
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raytracer_amd import flatten, runtime      # noqa: E402
from raytracer_amd.world import World           # noqa: E402

scene = sys.argv[1]
World(viewmode="parallel", buildfunction="fuzz_0", seed=1)
fuzz = [m for name, m in sys.modules.items() if name.startswith("rtamd_scene_fuzz")][0]
if scene.startswith("fuzzmesh_") and int(scene.split("_")[1]) >= fuzz.N_MESH_WORLDS:
    fuzz._make_mesh(int(scene.split("_")[1]))
elif scene.startswith("fuzz_") and int(scene.split("_")[1]) >= fuzz.N_WORLDS:
    fuzz._make(int(scene.split("_")[1]))
world = World(viewmode="parallel", buildfunction=scene, seed=1)
flat = flatten.flatten_world(world)
ds = runtime.device_scene(world)
cam = flatten.camera_record(world)
tx, ty = runtime.tile_grid(cam.width, cam.height)
tile_ids = numpy.arange(tx * ty, dtype=numpy.int32)
ns = len(world.viewplane.sampler.patterns[0])
n = len(tile_ids) * 64 * ns
print(scene, "tracer", flat.tracer_kind, "depth limit", flat.depth_limit, "frame", cam.width, cam.height, "spp", ns,
      "lights", [type(l).__name__ for l in world.lights], "ambient", type(world.ambient_color).__name__,
      "objects", [type(o).__name__ for o in world.objects])


def frame(per_sample, **flags):
    fb = runtime.FrameBuffers(ds, tile_ids)
    per = ds.sample_buffers(n) if per_sample else None
    ds.render_tiles(cam, world.seed, fb.tile_ids, 0, ns, sum_rgb8=fb.sum_rgb8, per_sample=per, **flags)
    torch.cuda.synchronize()
    return fb.sum_rgb8.cpu().numpy(), ({k: v.cpu().numpy() for k, v in per[0].items()} if per_sample else None)


ref_sum, ref = frame(True, count_work=True)
acc_sum, _ = frame(False)
per_sum, per = frame(True)
exe_sum, _ = frame(False, count_executed=True)
for label, s in (("accumulators only, timed", acc_sum), ("per-sample outputs, timed", per_sum), ("accumulators only, executed-work counting", exe_sum)):
    rows = numpy.nonzero((s != ref_sum).any(axis=1))[0]
    print("%-44s pixels that differ from the counting launch: %d" % (label, len(rows)))
    for r in rows[:6]:
        tile, pix = divmod(int(r), 64)
        print("   tile %d pixel-in-tile %d: %s against %s" % (tile, pix, s[r], ref_sum[r]))
        if per is not None:
            ks = (tile * ns + numpy.arange(ns)) * 64 + pix
            for k in ks:
                a, b = per["rgb"][k], ref["rgb"][k]
                if not numpy.array_equal(a, b):
                    print("      sample %d: per-sample timed %s counting %s  obj %s prim %s t %r rays %s / %s" % (
                        (k // 64) % ns, a, b, ref["hit_object"][k], ref["hit_prim"][k], ref["hit_t"][k], per["n_rays"][k], ref["n_rays"][k]))

# the same launches again and again: does any of them give another frame than its first run?
for label, flags, first in (("accumulators only, timed", dict(), acc_sum), ("executed-work counting", dict(count_executed=True), exe_sum),
                            ("counting", dict(count_work=True), None)):
    base, differ = first, 0
    for i in range(int(os.environ.get("REPEATS", "30"))):
        s, _ = frame(False, **flags)
        if base is None:
            base = s
        elif not numpy.array_equal(s, base):
            differ += 1
            rows = numpy.nonzero((s != base).any(axis=1))[0]
            print("   %s, run %d: %d pixels differ from the first run, e.g. row %d: %s against %s" % (label, i, len(rows), rows[0], s[rows[0]], base[rows[0]]))
    print("%-44s runs that differ from the first: %d" % (label, differ))
