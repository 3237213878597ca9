#!/usr/bin/env python3
"""How much of a frame's time the accumulator atomics cost: whole frames with and without
the integer accumulators (measurement tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd.world import World
from raytracer_amd import distributed, flatten, runtime

CASES = [("glossyreflective", None), ("mirrorhall_dof", dict(resolution=(1920, 1080), sample_dim=8)),
         ("dragon_area", dict(resolution=(1920, 1080), sample_dim=4)), ("cornellbox", None)]
for scene, ov in CASES:
    world = World.__new__(World)
    if ov:
        world.overrides = ov
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    job = distributed.FrameJob(world, 0, 1)
    res = {}
    for name, buf in (("with", job.sum_rgb8), ("without", None)):
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ds.render_tiles(cam, world.seed, job.tile_ids, 0, job.n_samples, sum_rgb8=buf)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t) * 1e3
    print("%-18s with accumulators %.2f ms, without %.2f ms" % (scene, res["with"], res["without"]), flush=True)
