#!/usr/bin/env python3
"""Does the order in which a rank's tiles are launched matter?  Kernel time of a share of
the bench frame with its tiles ascending (as dealt) and with the costliest tiles first
(cost = shadow rays per tile from a one-sample probe launch).  Measurement tool."""
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


def timed(job, tile_ids):
    job.render_fn(tile_ids, job.sum_rgb8, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        job.render_fn(tile_ids, job.sum_rgb8, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


for n in (1, 8):
    job = distributed.FrameJob(world, 0, n)
    t = len(job.tiles)
    per = ds.sample_buffers(t * 64, want=("hit_object", "n_rays"))
    ds.render_tiles(cam, world.seed, job.tile_ids, 0, 1, per_sample=per)
    torch.cuda.synchronize()
    obj = per[0]["hit_object"].view(t, 64)
    kd = (obj == 1).sum(dim=1).float() + 0.1 * (obj == 0).sum(dim=1).float()
    order = torch.argsort(kd, descending=True, stable=True)
    a = timed(job, job.tile_ids)
    b = timed(job, job.tile_ids[order].contiguous())
    c = timed(job, job.tile_ids[torch.randperm(t, device=job.tile_ids.device)].contiguous())
    print("ranks %d share 0: ascending %.3f ms, costliest first %.3f ms, random %.3f ms" % (n, a, b, c), flush=True)
