#!/usr/bin/env python3
"""Throughput and vector-lane use of the five benchmark configurations (and two of the
reference's scenes as shipped) on one GPU.

  python tools/config_rates.py [--no-pmc] [--lib PATH] [--out FILE] [label substring ...]

Every configuration runs in a child process: once plain for the time per frame, once under
`rocprofv3 --pmc` (SQ counters) for the share of SIMD cycles spent on vector instructions and
the active lanes per vector instruction.  This parent never touches the GPU."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CONFIGS = [
    ("1 foursphere 200x200 1spp", "foursphere", None),
    ("2 cornellbox_area 200x200 16spp", "cornellbox_area", None),
    ("3 glossyreflective 400x400 100spp", "glossyreflective", None),
    ("4 dragon_area 1920x1080 16spp", "dragon_area", dict(resolution=(1920, 1080), sample_dim=4)),
    ("5 mirrorhall_dof 3840x2160 64spp", "mirrorhall_dof", dict(resolution=(3840, 2160), sample_dim=8)),
    ("j dragon 400x400 1spp (as shipped)", "j", None),
    ("cornellbox path traced 200x200 100spp (as shipped)", "cornellbox", None),
    ("mirror teapot on a mirror floor 500x500 1spp (as shipped)", "mirror", None),
    # the reference's scene files at a frame size that fills the GPU: (width, height, sampler dimension)
    ("J dragon of j.py at 1920x1080 4spp (Matte, point light)", "j", None, (1920, 1080, 2)),
    ("K teapot of k.py at 1920x1080 4spp", "k", None, (1920, 1080, 2)),
    ("2 cornellbox_area at 1920x1080 16spp (a frame that fills the GPU)", "cornellbox_area", dict(resolution=(1920, 1080), sample_dim=4)),
    ("1 foursphere at 1920x1080 16spp (a frame that fills the GPU)", "foursphere", None, (1920, 1080, 4)),
]


def child(index, frames):
    import hashlib
    import numpy
    import torch
    from raytracer_amd.world import World
    from raytracer_amd import flatten, runtime
    label, scene, overrides = CONFIGS[index][:3]
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    if len(CONFIGS[index]) > 3:
        import sampler
        width, height, dim = CONFIGS[index][3]
        world.viewplane.pixel_size *= world.viewplane.resolution[0] / float(width)      # same framing
        world.viewplane.resolution = (width, height)
        world.viewplane.sampler = sampler.MultiJitteredSampler(dim)
        world.invalidate()
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    fb = runtime.FrameBuffers(ds, tiles)
    ns = len(world.viewplane.sampler.patterns[0])
    # split very large frames over sample ranges so one launch stays below 2^31 threads
    step = max(1, min(ns, (1 << 30) // (len(tiles) * 64)))

    def frame():
        fb.sum_rgb8.zero_()
        for s in range(0, ns, step):
            ds.render_tiles(cam, world.seed, fb.tile_ids, s, min(ns, s + step), sum_rgb8=fb.sum_rgb8)
    frame()
    torch.cuda.synchronize()
    ds.read_counters()
    times = []
    for _ in range(frames):
        t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    c = ds.read_counters()
    rays = (c["closest"] + c["shadow"]) // frames
    dt = sorted(times)[len(times) // 2]
    print(json.dumps({"config": label, "ms_per_frame": round(dt * 1e3, 3), "samples": c["samples"] // frames, "rays": rays,
                      "rays_per_sample": round(rays / max(c["samples"] // frames, 1), 2),
                      "Mrays_per_s": round(rays / dt / 1e6, 1), "launches_per_frame": (ns + step - 1) // step,
                      "frame_digest": hashlib.sha256(fb.sum_rgb8.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("only", nargs="*")
    args = ap.parse_args()
    if args.child >= 0:
        return child(args.child, args.frames)
    from raytracer_amd import profiling
    env = dict(os.environ)
    if args.lib:
        env["RTAMD_LIB"] = os.environ["RTAMD_LIB"] = os.path.abspath(args.lib)
    rows = []
    for i, (label, *_rest) in enumerate(CONFIGS):
        if args.only and not any(o in label for o in args.only):
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--child", str(i)]
        p = subprocess.run(cmd + ["--frames", str(args.frames)], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print("%s failed: %s" % (label, p.stderr[-500:]), file=sys.stderr)
            return 1
        row = json.loads(p.stdout.strip().splitlines()[-1])
        if not args.no_pmc:
            pmc = profiling.collect(cmd + ["--frames", "1"], passes=[profiling.PASSES[2]],
                                    log=lambda m: print(m, file=sys.stderr))
            if pmc:
                launches = row["launches_per_frame"]
                kernel_ms = pmc["GRBM_GUI_ACTIVE"] / profiling.N_XCDS      # cycles per launch (profiled run)
                row.update({
                    "valu_busy": round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (kernel_ms * profiling.N_SIMDS), 3),
                    "lane_utilisation": round(pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0), 3),
                    "wave_time_waiting": round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 3),
                    "valu_insts_per_ray": round(pmc["SQ_INSTS_VALU"] * launches / max(row["rays"], 1), 1),
                    "kernel": (re.search(r"\w+_kernel<[^>]*>", pmc["kernel_name"]) or re.search(".*", pmc["kernel_name"])).group(0),
                    "scratch": pmc.get("scratch")})
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
