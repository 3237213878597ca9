#!/usr/bin/env python3
"""Load balance of the tile deal: kernel time of every rank's share of the bench frame,
run one after the other on this GPU (measurement tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd.world import World
from raytracer_amd import distributed

scene, ov = bench.WORKLOADS["dragon_1080p_16spp_ao_arealight"]
world = World.__new__(World)
world.overrides = ov
World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        job = distributed.FrameJob(world, r, n)
        job.render_fn(job.tile_ids, job.sum_rgb8, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            job.render_fn(job.tile_ids, job.sum_rgb8, None)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 3)
    print("ranks %d: kernel ms per rank %s; max/mean %.3f; ideal speedup %.2f" % (
        n, " ".join("%.2f" % t for t in times), max(times) / (sum(times) / n), times and (sum(times)) / max(times)), flush=True)
