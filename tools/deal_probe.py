#!/usr/bin/env python3
"""Which deal of 8x8 tiles to 8 ranks gives the shortest slowest rank?  Every rank's share
is rendered on this GPU in turn.  Measurement tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
from raytracer_amd.world import World
from raytracer_amd import distributed, flatten, runtime

scene, ov = bench.WORKLOADS["dragon_1080p_16spp_ao_arealight"]
world = World.__new__(World)
world.overrides = ov
World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
ds = runtime.device_scene(world)
cam = flatten.camera_record(world)
tx, ty = runtime.tile_grid(cam.width, cam.height)
tiles = numpy.arange(tx * ty)
x, y = tiles % tx, tiles // tx
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
deals = {
    "diagonal 1x1": (x + y) % N,
    "diagonal 2x2": (x // 2 + y // 2) % N,
    "diagonal 4x4": (x // 4 + y // 4) % N,
    "diagonal 2x1": (x // 2 + y) % N,
    "diagonal 4x1": (x // 4 + y) % N,
    "diagonal 8x1": (x // 8 + y) % N,
    "rows": y % N,
    "columns": x % N,
}
job = distributed.FrameJob(world, 0, 1)
for name, owner in deals.items():
    times = []
    for r in range(N):
        ids = torch.as_tensor(tiles[owner == r].astype(numpy.int32)).cuda()
        job.render_fn(ids, job.sum_rgb8, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            job.render_fn(ids, job.sum_rgb8, None)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 3)
    print("%-14s max %.3f mean %.3f ms  (%s)" % (name, max(times), sum(times) / N, " ".join("%.2f" % t for t in times)), flush=True)
