#!/usr/bin/env python3
"""Render one of the test worlds of tests/test_gpu_parity.shadow_path_world whole (accumulators only:
the whole-frame kernels) with several builds of the library and print a digest of each frame.

  python tools/frame_digest.py CASE LIB [LIB ...]

Each LIB runs in a child process (a process binds one library).  Used to show that a change of a
kernel changes, or does not change, a frame."""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(case):
    sys.path.insert(0, REPO)
    import numpy
    import torch
    from raytracer_amd import flatten, runtime
    from tests.test_gpu_parity import shadow_path_world
    world, _, _ = shadow_path_world(case)
    world.invalidate()
    ds = runtime.device_scene(world)
    cam = flatten.camera_record(world)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    fb = runtime.FrameBuffers(ds, numpy.arange(tx * ty, dtype=numpy.int32))
    ns = len(world.viewplane.sampler.patterns[0])
    ds.render_tiles(cam, world.seed, fb.tile_ids, 0, ns, sum_rgb8=fb.sum_rgb8)
    torch.cuda.synchronize()
    sums = fb.sum_rgb8.cpu().numpy()
    print(json.dumps({"digest": hashlib.sha256(sums.tobytes()).hexdigest()[:16], "sum": int(sums.sum()),
                      "sums_file": None}))
    numpy.save(os.environ["FRAME_DIGEST_OUT"], sums)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    case, libs = sys.argv[1], sys.argv[2:]
    frames = []
    for lib in libs:
        out = "/tmp/frame_digest_%d.npy" % len(frames)
        env = dict(os.environ, RTAMD_LIB=os.path.abspath(lib), FRAME_DIGEST_OUT=out)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case], env=env, capture_output=True,
                           text=True, timeout=300)
        if p.returncode != 0:
            print("%-28s FAILED: %s" % (os.path.basename(lib), p.stderr.strip().splitlines()[-3:]))
            return 1
        r = json.loads(p.stdout.strip().splitlines()[-1])
        import numpy
        frames.append(numpy.load(out))
        differ = int((frames[-1] != frames[0]).any(axis=1).sum())
        print("%-28s frame %s  sum %d  pixels that differ from the first build's: %d" % (
            os.path.basename(lib), r["digest"], r["sum"], differ), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
