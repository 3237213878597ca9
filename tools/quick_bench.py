#!/usr/bin/env python3
"""Time the whole-frame tile kernel of the benchmark workload for one or more builds of the
library and check that every build renders the same frame.

  python tools/quick_bench.py [--steps N] LIB [LIB ...]

Each LIB is a path to a librtamd*.so; every one runs in a child process (a process binds one
library).  Prints per build: kernel ms (HIP events, median of N), a checksum of the integer
accumulators, and whether it equals the first build's."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, REPO)
    import hashlib
    import torch
    from raytracer_amd.world import World
    from raytracer_amd import flatten, runtime
    import bench
    scene, overrides = bench.WORKLOADS[args.workload]
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    import numpy
    fb = runtime.FrameBuffers(ds, numpy.arange(tx * ty, dtype=numpy.int32))
    ns = len(world.viewplane.sampler.patterns[0])
    times = []
    for i in range(args.steps + 1):
        fb.sum_rgb8.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ds.render_tiles(cam, world.seed, fb.tile_ids, 0, ns, sum_rgb8=fb.sum_rgb8)
        e1.record()
        torch.cuda.synchronize()
        if i:
            times.append(e0.elapsed_time(e1))
    times.sort()
    digest = hashlib.sha256(fb.sum_rgb8.cpu().numpy().tobytes()).hexdigest()[:16]
    c = ds.read_counters()
    print(json.dumps({"ms": times[len(times) // 2], "min_ms": times[0], "digest": digest,
                      "rays": (c["closest"] + c["shadow"]) // (args.steps + 1)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--workload", default="dragon_1080p_16spp_ao_arealight")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    first = None
    for lib in args.libs:
        env = dict(os.environ, RTAMD_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps),
                            "--workload", args.workload], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            print("%-28s FAILED: %s" % (os.path.basename(lib), p.stderr.strip().splitlines()[-1:]))
            return 1
        r = json.loads(p.stdout.strip().splitlines()[-1])
        first = first or r["digest"]
        print("%-28s %7.3f ms (min %7.3f)  %7.1f Mrays/s  frame %s %s" % (
            os.path.basename(lib), r["ms"], r["min_ms"], r["rays"] / r["ms"] / 1e3, r["digest"],
            "same" if r["digest"] == first else "DIFFERENT"), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
