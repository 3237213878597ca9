"""Throughput of the five BASELINE.json configurations on one GPU (Mrays/s, ms per frame)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from raytracer_amd.world import World
from raytracer_amd import flatten, runtime

CONFIGS = [
    ("1 foursphere 200x200 1spp", "foursphere", None),
    ("2 cornellbox_area 200x200 16spp", "cornellbox_area", None),
    ("3 glossyreflective 400x400 100spp", "glossyreflective", None),
    ("4 dragon_area 1920x1080 16spp", "dragon_area", dict(resolution=(1920, 1080), sample_dim=4)),
    ("5 mirrorhall_dof 3840x2160 64spp", "mirrorhall_dof", dict(resolution=(3840, 2160), sample_dim=8)),
    ("j dragon 400x400 1spp (as shipped)", "j", None),
    ("cornellbox path traced 200x200 100spp (as shipped)", "cornellbox", None),
]
only = sys.argv[1:] 
for label, scene, overrides in CONFIGS:
    if only and not any(o in label for o in only):
        continue
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    fb = runtime.FrameBuffers(ds, tiles)
    ns = len(world.viewplane.sampler.patterns[0])
    # split very large frames over sample ranges so one launch stays below 2^31 threads
    step = max(1, min(ns, (1 << 30) // (len(tiles) * 64)))
    def frame():
        for s in range(0, ns, step):
            ds.render_tiles(cam, world.seed, fb.tile_ids, s, min(ns, s + step), sum_rgb8=fb.sum_rgb8)
    frame(); torch.cuda.synchronize(); ds.read_counters()
    t0 = time.perf_counter(); frame(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = ds.read_counters()
    rays = c["closest"] + c["shadow"]
    print(json.dumps({"config": label, "ms_per_frame": round(dt * 1e3, 2), "samples": c["samples"], "rays": rays,
                      "rays_per_sample": round(rays / max(c["samples"], 1), 2), "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)
