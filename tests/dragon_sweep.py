#!/usr/bin/env python3
"""One-off sweep over random variations of the benchmark's dragon world (the mesh, its kd-tree and its shadow
tree are those of configuration 4; camera, emitter, materials and ambient light vary):
  python tests/dragon_sweep.py FIRST LAST [SAMPLER_DIMENSION]
Every world goes through tests/test_gpu_parity.check_tiles_against_oracle on 48 random tiles of a 160x96 frame
(every sample; ids / ray counts / reference work exact, the timed and both counting kernels the same frame).
Needs a GPU; about five seconds per world, most of it the host-side tables."""
import os
import sys
import time
import traceback

import numpy

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def vary(world, k):
    """Seeded variation k of the dragon_area world, in place."""
    import raytracer_amd
    raytracer_amd.activate()
    from camera import PinholeCamera
    from geometry import Rectangle
    from light import AmbientLight, AmbientOccluder, AreaLight, PointLight
    from material import Emissive, Matte, Phong
    from tracer import AreaLightTracer, Tracer
    r = numpy.random.RandomState(77000 + k)
    sampler = world.viewplane.sampler
    # a camera anywhere around the dragon, close-ups and views from inside its box among them
    eye = (float(round(r.uniform(-9, 9), 3)), float(round(r.uniform(0.3, 8), 3)), float(round(r.uniform(-6, 9), 3)))
    lookat = (float(round(r.uniform(-3, 3), 2)), float(round(r.uniform(0.5, 3), 2)), float(round(r.uniform(-1, 1), 2)))
    world.camera = PinholeCamera(eye=eye, up=(0., 1., 0.), lookat=lookat, viewing_distance=float(r.choice([60., 100., 200.])))
    tree = world.objects[0]
    mat = Matte(ka=1, kd=1, cd=numpy.array((1., .84, .1))) if r.rand() < 0.6 else \
        Phong(ka=.6, kd=.7, ks=.5, exp=float(r.choice([5, 30, 100])), cd=numpy.array((1., .84, .1)))
    for box in tree.boxes:
        box.obj.material = mat
    kind = int(r.randint(0, 3))
    if kind < 2:
        # an emitter far from, just outside, or inside the tree's box, facing -z or +z
        where = r.rand()
        if where < 0.4:
            corner = (float(r.uniform(-8, 2)), float(r.uniform(5, 9)), float(r.uniform(5, 10)))
        elif where < 0.7:
            corner = (float(r.uniform(-6, 4)), float(r.uniform(1, 5)), float(r.uniform(3.16, 3.6)))
        else:
            corner = (float(r.uniform(-5, 3)), float(r.uniform(1, 5)), float(r.uniform(-2, 2.5)))
        normal = (0., 0., -1.) if r.rand() < 0.75 else (0., 0., 1.)
        panel = Rectangle(numpy.array(corner), float(r.choice([2, 3, 6])), float(r.choice([2, 4])), numpy.array(normal),
                          numpy.array((1., 0., 0.)), Emissive(float(r.choice([10., 30.])), numpy.array((1., .96, .9))), sampler)
        world.lights = [AreaLight(panel.material, panel, float(r.choice([3., 6.])))]
        world.tracer = AreaLightTracer(world)
    else:
        # the plain tracer with point lights among and above the coils (j.py's kind of light)
        world.lights = [PointLight(numpy.array((1., 1., 1.)), 1., numpy.array((float(r.uniform(-6, 6)), float(r.uniform(1, 9)), float(r.uniform(-2, 4)))),
                                   radius=10, attenuation=int(r.choice([0, 2])), cast_shadow=True) for _ in range(int(r.randint(1, 3)))]
        world.tracer = Tracer(world)
    if r.rand() < 0.7:
        world.ambient_color = AmbientOccluder(numpy.array((1., 1., 1.)), float(round(r.uniform(.2, .5), 2)), sampler)
    else:
        world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
    world.invalidate()


def main():
    import torch
    import __graft_entry__  # noqa: F401
    from raytracer_amd import flatten
    import test_gpu_parity as t
    first, last = int(sys.argv[1]), int(sys.argv[2])
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # 2: waves of 4x4 pixels x 4 samples; 4: 2x2 pixels x 16 samples, as in the benchmark
    world = t._tiny_world("dragon_area", resolution=(160, 96) if dim <= 2 else (96, 56), sample_dim=dim, seed=5)
    bad = 0
    t0 = time.time()
    for k in range(first, last):
        name = "dragonworld_%d" % k
        if (k - first) % 10 == 0:
            print("... %d of %d worlds, %d failed so far, %.0f s" % (k - first, last - first, bad, time.time() - t0), flush=True)
        try:
            vary(world, k)
            t.check_tiles_against_oracle(torch, world, flatten.flatten_world(world), 48 if dim <= 2 else 40, None, name)
        except Exception:
            bad += 1
            print("FAILED %s" % name, flush=True)
            traceback.print_exc()
    print("%d worlds, %d failed" % (last - first, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
