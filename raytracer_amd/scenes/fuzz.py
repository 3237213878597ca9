"""Seeded random worlds: every primitive, material, light, camera and tracer of the device
path in combinations nobody wrote down by hand.  Test scenes of this project, written
against the reference's scene API only, so the reference itself renders them when golden
vectors are generated.  `fuzz_<k>`: k % 3 picks the tracer (Tracer, AreaLightTracer,
WhittedTracer), everything else follows from numpy.random.RandomState(4000 + k)."""
import os

import numpy
import affinetransform as transform
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera, ThinLensCamera
from geometry import (AxisAlignedBox, CheckerPlane, ConcaveSphere, Instance, Plane, Rectangle,
                      Sphere, Triangle)
from kdtree import BoundingBoxes, KDTree
from light import (AmbientLight, AmbientOccluder, AreaLight, DirectionLight, EnvironmentLight,
                   PointLight)
from material import (Dielectric, Emissive, GlossyReflective, Matte, Phong, Reflective,
                      Transparent)
from mesh_parser import read_mesh
from sampler import MultiJitteredSampler, RegularSampler
from tracer import AreaLightTracer, PathTracer, Tracer, ViewPlane, WhittedTracer

N_WORLDS = 128


def _vec(r, lo, hi):
    return numpy.array([float(x) for x in r.uniform(lo, hi, 3)])


def _colour(r):
    return numpy.array([float(x) for x in numpy.round(r.uniform(0.15, 1.0, 3), 2)])


def _material(r, kind, lobe, lobe_e):
    """One material the tracer `kind` (0 Tracer, 1 AreaLightTracer, 2 WhittedTracer) can shade."""
    pick = int(r.randint(0, (4, 3, 6)[kind]))
    ka, kd = float(round(r.uniform(.2, .8), 2)), float(round(r.uniform(.3, .9), 2))
    ks, ex = float(round(r.uniform(.1, .7), 2)), float(r.choice([3, 12, 40, 150]))
    cd = _colour(r)
    if pick == 0:
        return Matte(ka=ka, kd=kd, cd=cd)
    if pick == 1:
        m = Phong(ka=ka, kd=kd, ks=ks, exp=ex, cd=cd)
        if r.rand() < 0.3:
            m.receives_shadow = False
        return m
    if pick == 2:
        return Reflective(ka=ka * .5, kd=kd * .5, ks=ks, kr=float(round(r.uniform(.4, .9), 2)), exp=ex, cd=cd)
    if pick == 3:
        return GlossyReflective(ka=ka * .4, kd=kd * .4, ks=ks * .5, kr=float(round(r.uniform(.4, .8), 2)),
                                exp=lobe_e, cd=cd, sampler=lobe)
    if pick == 4:
        return Transparent(ka=ka * .2, kd=kd * .2, ks=ks, cd=cd, exp=ex, sampler=None,
                           ior=float(r.choice([0.8, 1.3, 1.5])), kt=float(round(r.uniform(.6, .9), 2)), kr=.1)
    return Dielectric(ka=0.05, kd=0.05, ks=ks * .4, cd=cd, exp=ex, sampler=None,
                      eta_in=float(r.choice([0.75, 1.33, 1.5])), eta_out=1.0,
                      cf_in=_colour(r), cf_out=numpy.array((1., .98, .96)))


def _transform(r, where):
    m = transform.translate(float(where[0]), float(where[1]), float(where[2]))
    m = m.dot(transform.rotate(_vec(r, -1, 1) + numpy.array((0., 1.5, 0.)), float(r.uniform(0, 3))))
    return m.dot(transform.scale(*[float(x) for x in numpy.round(r.uniform(0.6, 1.6, 3), 2)]))


def _soup(r, n, material):
    """n small triangles scattered in a box of side 2: a kd-tree of a few levels."""
    boxes = []
    for i in range(n):
        a = _vec(r, -1, 1)
        tri = Triangle(a, a + _vec(r, -.5, .5), a + _vec(r, -.5, .5), material=material)
        tri.label = str(i)
        tri.cast_shadow = bool(r.rand() < 0.85)
        boxes.extend(tri.get_bounding_boxes())
    return KDTree(BoundingBoxes(boxes))


def _mesh_tree(name, material, smooth):
    with open(name) as f:
        mesh = read_mesh(f)
    if smooth:
        mesh.compute_smooth_normal()
    mesh.material = material
    return KDTree(BoundingBoxes(mesh.get_bounding_boxes()))


def build_world(world, viewmode, k):
    r = numpy.random.RandomState(4000 + k)
    kind = k % 3
    world.viewmode = viewmode
    dim = 1 if viewmode == "realtime" else int(r.choice([1, 2, 3]))
    if viewmode != "realtime" and os.environ.get("RTAMD_FUZZ_DIM"):
        dim = int(os.environ["RTAMD_FUZZ_DIM"])     # hand-run sweeps at other sample counts (4: waves of 2x2 pixels x 16 samples)
    sampler = RegularSampler() if viewmode == "realtime" else MultiJitteredSampler(sample_dim=dim)
    lobe_e = float(r.choice([5, 25, 100]))
    lobe = MultiJitteredSampler(sample_dim=max(dim, 2), e=lobe_e)
    world.viewplane = ViewPlane(resolution=(96, 64), pixel_size=3.3, sampler=sampler)
    angle = float(r.uniform(0, 2 * numpy.pi))
    eye = (float(round(13 * numpy.cos(angle), 3)), float(round(r.uniform(3, 8), 3)), float(round(13 * numpy.sin(angle), 3)))
    if r.rand() < 0.5 and viewmode != "realtime":
        world.camera = ThinLensCamera(lens_radius=float(round(r.uniform(.03, .15), 3)), focal_plane_distance=13.0,
                                      eye=eye, up=(0., 1., 0.), lookat=(0., 1., 0.), viewing_distance=250.)
    else:
        world.camera = PinholeCamera(eye=eye, up=(0., 1., 0.), lookat=(0., 1., 0.), viewing_distance=250.)
    # an array: Transparent.shade multiplies what trace_ray returns by a float (material.py:435)
    world.background_color = numpy.array([float(x) for x in numpy.round(r.uniform(0, .2, 3), 2)])
    depth = int(r.randint(1, 5))
    world.tracer = (Tracer(world, depth), AreaLightTracer(world), WhittedTracer(world, min(depth, 3)))[kind]
    if r.rand() < 0.5 and viewmode != "realtime":
        world.ambient_color = AmbientOccluder(numpy.array((1., 1., 1.)), float(round(r.uniform(.2, .6), 2)), sampler)
    else:
        world.ambient_color = AmbientLight(_colour(r), float(round(r.uniform(.1, .4), 2)))

    ground_a, ground_b = _material(r, min(kind, 1), lobe, lobe_e), Matte(ka=.6, kd=.8, cd=_colour(r))
    if r.rand() < 0.5:
        objects = [CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=ground_a, alt_material=ground_b,
                                up=(1, 0, 0), grid_size=float(r.choice([0.75, 1.25, 2.0])))]
    else:
        objects = [Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=ground_a)]
    for _ in range(int(r.randint(5, 10))):
        where = _vec(r, -4, 4) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(.4, 2.5)), 0.))
        mat = _material(r, kind, lobe, lobe_e)
        shape = int(r.randint(0, 9))
        if shape == 0:
            obj = Sphere(center=where, radius=float(round(r.uniform(.4, 1.3), 2)), material=mat)
        elif shape == 1:
            half = _vec(r, .3, 1.0)
            lo, hi = where - half, where + half
            obj = AxisAlignedBox(float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1]), float(lo[2]), float(hi[2]), material=mat)
        elif shape == 2:
            obj = Triangle(where + _vec(r, -1.5, 1.5), where + _vec(r, -1.5, 1.5), where + _vec(r, -1.5, 1.5), material=mat)
        elif shape == 3:
            obj = Rectangle(where, float(round(r.uniform(1, 3), 1)), float(round(r.uniform(1, 3), 1)),
                            numpy.array((0., 0., 1.)), numpy.array((0., 1., 0.)), mat, sampler)
        elif shape == 4:
            obj = Instance(Sphere(center=numpy.zeros(3), radius=1., material=mat), _transform(r, where))
        elif shape == 5:
            obj = Instance(Instance(AxisAlignedBox(-.6, .6, -.6, .6, -.6, .6, material=mat), _transform(r, (0, 0, 0))),
                           _transform(r, where))
        elif shape == 6:
            obj = Instance(_soup(r, int(r.randint(30, 120)), mat), _transform(r, where))
        elif shape == 7:
            obj = Instance(_mesh_tree('meshes/mesh1.obj', mat, False),
                           _transform(r, where).dot(transform.scale(7., 7., 7.)))
        else:
            obj = _soup(r, int(r.randint(30, 120)), mat)
        if r.rand() < 0.2:
            obj.cast_shadow = False
        objects.append(obj)

    lights = []
    # the reference's area_light_shade asks every light for G and pdf (material.py:267-281):
    # worlds under the AreaLightTracer hold area and environment lights only
    for _ in range(int(r.randint(1, 4)) if kind != 1 else 0):
        if r.rand() < 0.6:
            lights.append(PointLight(_colour(r), float(round(r.uniform(.8, 2.0), 2)),
                                     _vec(r, -6, 6) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(4, 9)), 0.)),
                                     radius=float(r.choice([5, 8, 12])), attenuation=int(r.choice([0, 2])),
                                     cast_shadow=bool(r.rand() < 0.7)))
        else:
            lights.append(DirectionLight(_colour(r), float(round(r.uniform(.3, 1.0), 2)),
                                         _vec(r, -1, 1) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(.3, 1.)), 0.)),
                                         bool(r.rand() < 0.7)))
    if kind == 1:
        glow = Emissive(float(round(r.uniform(3, 8), 1)), _colour(r))
        panel = Rectangle(numpy.array((float(r.uniform(-3, 1)), float(r.uniform(4, 7)), float(r.uniform(-9, -6)))),
                          float(r.choice([2, 4])), float(r.choice([2, 3])),
                          numpy.array((0., 0., 1.)), numpy.array((0., 1., 0.)), glow, sampler)
        panel.cast_shadow = False
        objects.append(panel)
        lights.append(AreaLight(glow, panel, float(r.choice([3., 5., 8.]))))
        if r.rand() < 0.5:
            sky = Emissive(float(round(r.uniform(.3, .8), 2)), _colour(r))
            dome = ConcaveSphere(numpy.zeros(3), 100000., sky)
            dome.cast_shadow = False
            objects.insert(0, dome)
            lights.insert(0, EnvironmentLight(sky, sampler))
    world.objects = objects
    world.lights = lights


def _make(k):
    class Fuzz(BuildFunctionBase):
        BUILD_FUNCTION_NAME = 'fuzz_%d' % k

        @classmethod
        def build_function(cls, world, viewmode):
            build_world(world, viewmode, k)
    Fuzz.__name__ = 'Fuzz%d' % k
    return Fuzz


WORLDS = [_make(k) for k in range(N_WORLDS)]


# ---- `fuzzmesh_<k>`: the teapot (6 320 triangles, a tree big enough for a shadow tree) in random company
N_MESH_WORLDS = 40


def build_mesh_world(world, viewmode, k):
    """The teapot's kd-tree (smooth or flat, bare or behind one or two Instances, some worlds with
    triangles that cast no shadow) on a ground plane with a few analytic objects, lights inside
    and outside the mesh's box, with and without limits: what decides which walk a shadow ray
    takes on the device."""
    r = numpy.random.RandomState(9000 + k)
    kind = k % 3
    world.viewmode = viewmode
    dim = int(r.choice([1, 2]))
    if os.environ.get("RTAMD_FUZZ_DIM"):
        dim = int(os.environ["RTAMD_FUZZ_DIM"])
    sampler = MultiJitteredSampler(sample_dim=dim)
    lobe_e = float(r.choice([5, 25, 100]))
    lobe = MultiJitteredSampler(sample_dim=2, e=lobe_e)
    world.viewplane = ViewPlane(resolution=(96, 64), pixel_size=3.3, sampler=sampler)
    angle = float(r.uniform(0, 2 * numpy.pi))
    eye = (float(round(9 * numpy.cos(angle), 3)), float(round(r.uniform(2, 6), 3)), float(round(9 * numpy.sin(angle), 3)))
    world.camera = PinholeCamera(eye=eye, up=(0., 1., 0.), lookat=(0., 1.2, 0.), viewing_distance=330.)
    world.background_color = numpy.array([float(x) for x in numpy.round(r.uniform(0, .2, 3), 2)])
    world.tracer = (Tracer(world, int(r.randint(1, 4))), AreaLightTracer(world), WhittedTracer(world, 2))[kind]
    if r.rand() < 0.6:
        world.ambient_color = AmbientOccluder(numpy.array((1., 1., 1.)), float(round(r.uniform(.2, .6), 2)), sampler)
    else:
        world.ambient_color = AmbientLight(_colour(r), float(round(r.uniform(.1, .4), 2)))

    with open('meshes/teapot.obj') as f:
        mesh = read_mesh(f)
    if r.rand() < 0.5:
        mesh.compute_smooth_normal()
    mesh.material = _material(r, kind, lobe, lobe_e)
    boxes = mesh.get_bounding_boxes()
    if r.rand() < 0.4:
        for box in boxes[::int(r.randint(3, 9))]:
            box.obj.cast_shadow = False
    pot = KDTree(BoundingBoxes(boxes))
    wrap = int(r.randint(0, 3))
    if wrap >= 1:
        pot = Instance(pot, _transform(r, (0., 0., 0.)))
    if wrap == 2:
        pot = Instance(pot, transform.translate(float(r.uniform(-1, 1)), 0.2, float(r.uniform(-1, 1))))
    objects = [Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=_material(r, min(kind, 1), lobe, lobe_e)), pot]
    for _ in range(int(r.randint(0, 3))):
        where = _vec(r, -4, 4) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(.4, 2.0)), 0.))
        objects.append(Sphere(center=where, radius=float(round(r.uniform(.4, 1.0), 2)), material=_material(r, kind, lobe, lobe_e)))
    lights = []
    if kind == 1:
        glow = Emissive(float(round(r.uniform(5, 12), 1)), _colour(r))
        # half of the panels hang low and close: their rays end inside the teapot's box
        height = float(r.uniform(1.0, 2.5)) if r.rand() < 0.5 else float(r.uniform(5, 8))
        panel = Rectangle(numpy.array((float(r.uniform(-2, 0)), height, float(r.uniform(-3, -1.5)))),
                          float(r.choice([1, 2])), float(r.choice([1, 2])),
                          numpy.array((0., 0., 1.)), numpy.array((0., 1., 0.)), glow, sampler)
        panel.cast_shadow = False
        objects.append(panel)
        lights.append(AreaLight(glow, panel, float(r.choice([3., 5.]))))
    else:
        for _ in range(int(r.randint(1, 4))):
            if r.rand() < 0.6:
                inside = r.rand() < 0.5
                where = _vec(r, -1.5, 1.5) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(1.0, 2.8) if inside else r.uniform(5, 9)), 0.))
                lights.append(PointLight(_colour(r), float(round(r.uniform(.8, 2.0), 2)), where,
                                         radius=float(r.choice([5, 8])), attenuation=int(r.choice([0, 2])),
                                         cast_shadow=True))
            else:
                lights.append(DirectionLight(_colour(r), float(round(r.uniform(.3, 1.0), 2)),
                                             _vec(r, -1, 1) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(.3, 1.)), 0.)),
                                             True))
    world.objects = objects
    world.lights = lights


def _make_mesh(k):
    class FuzzMesh(BuildFunctionBase):
        BUILD_FUNCTION_NAME = 'fuzzmesh_%d' % k

        @classmethod
        def build_function(cls, world, viewmode):
            build_mesh_world(world, viewmode, k)
    FuzzMesh.__name__ = 'FuzzMesh%d' % k
    return FuzzMesh


MESH_WORLDS = [_make_mesh(k) for k in range(N_MESH_WORLDS)]


# ---- `fuzzpath_<k>`: path-traced worlds (Matte and Emissive materials, tracer.py:74-88, material.py:283-289)
N_PATH_WORLDS = 6


def build_path_world(world, viewmode, k):
    """Random analytic objects, two kinds of mesh (mesh1's 70 triangles in a tree of a few levels, the teapot in a
    tree with a shadow tree that path tracing never asks) and emitters, all Matte or Emissive, under the
    PathTracer: every hit draws one direction about the normal and goes on, up to the depth limit."""
    r = numpy.random.RandomState(52000 + k)
    world.viewmode = viewmode
    dim = int(r.choice([1, 2, 3]))
    if os.environ.get("RTAMD_FUZZ_DIM"):
        dim = int(os.environ["RTAMD_FUZZ_DIM"])
    sampler = MultiJitteredSampler(sample_dim=dim)
    bounce = MultiJitteredSampler(sample_dim=2, e=float(r.choice([1, 1, 5])))
    world.viewplane = ViewPlane(resolution=(96, 64), pixel_size=3.3, sampler=sampler)
    angle = float(r.uniform(0, 2 * numpy.pi))
    eye = (float(round(10 * numpy.cos(angle), 3)), float(round(r.uniform(2, 6), 3)), float(round(10 * numpy.sin(angle), 3)))
    world.camera = PinholeCamera(eye=eye, up=(0., 1., 0.), lookat=(0., 1.2, 0.), viewing_distance=330.)
    world.background_color = numpy.array([float(x) for x in numpy.round(r.uniform(0, .2, 3), 2)])
    world.tracer = PathTracer(world)
    world.tracer.depth_limit = int(r.randint(1, 6))
    world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
    world.lights = []

    def matte():
        return Matte(ka=.5, kd=float(round(r.uniform(.4, .95), 2)), cd=_colour(r), sampler=bounce)

    def glow():
        return Emissive(float(round(r.uniform(2, 12), 1)), _colour(r))
    objects = [Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=matte())]
    if r.rand() < 0.5:
        objects.append(Plane(origin=(0, 0, -7), normal=(0, 0, 1), material=matte()))
    if r.rand() < 0.6:
        with open('meshes/teapot.obj') as f:
            mesh = read_mesh(f)
        if r.rand() < 0.5:
            mesh.compute_smooth_normal()
        mesh.material = matte()
        pot = KDTree(BoundingBoxes(mesh.get_bounding_boxes()))
        objects.append(pot if r.rand() < 0.5 else Instance(pot, _transform(r, (0., 0., 0.))))
    for _ in range(int(r.randint(3, 8))):
        where = _vec(r, -4, 4) * numpy.array((1., 0., 1.)) + numpy.array((0., float(r.uniform(.4, 2.5)), 0.))
        mat = glow() if r.rand() < 0.3 else matte()
        shape = int(r.randint(0, 6))
        if shape == 0:
            obj = Sphere(center=where, radius=float(round(r.uniform(.4, 1.3), 2)), material=mat)
        elif shape == 1:
            half = _vec(r, .3, 1.0)
            lo, hi = where - half, where + half
            obj = AxisAlignedBox(float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1]), float(lo[2]), float(hi[2]), material=mat)
        elif shape == 2:
            obj = Rectangle(where + numpy.array((0., 2., 0.)), float(round(r.uniform(1, 3), 1)), float(round(r.uniform(1, 3), 1)),
                            numpy.array((0., -1., 0.)), numpy.array((1., 0., 0.)), mat, sampler)
        elif shape == 3:
            obj = Instance(Sphere(center=numpy.zeros(3), radius=1., material=mat), _transform(r, where))
        elif shape == 4:
            obj = Instance(_mesh_tree('meshes/mesh1.obj', mat, bool(r.rand() < 0.5)),
                           _transform(r, where).dot(transform.scale(7., 7., 7.)))
        else:
            obj = _soup(r, int(r.randint(30, 120)), mat)
        objects.append(obj)
    # something to see by
    objects.append(Sphere(center=numpy.array((float(r.uniform(-3, 3)), float(r.uniform(5, 8)), float(r.uniform(-3, 3)))),
                          radius=float(round(r.uniform(.8, 2.0), 2)), material=glow()))
    world.objects = objects


def _make_path(k):
    class FuzzPath(BuildFunctionBase):
        BUILD_FUNCTION_NAME = 'fuzzpath_%d' % k

        @classmethod
        def build_function(cls, world, viewmode):
            build_path_world(world, viewmode, k)
    FuzzPath.__name__ = 'FuzzPath%d' % k
    return FuzzPath


PATH_WORLDS = [_make_path(k) for k in range(N_PATH_WORLDS)]
