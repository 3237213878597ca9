"""Small scenes that put every primitive, material and light of the device path in front
of the camera at least once.  They are test scenes of this project (the reference ships
nothing like them); each is written against the reference's scene API only, so the
reference itself can render them when golden vectors are generated."""
import numpy
import affinetransform as transform
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera, ThinLensCamera
from geometry import (AxisAlignedBox, CheckerPlane, ConcaveSphere, Instance, Plane, Rectangle,
                      Sphere, Triangle)
from kdtree import BoundingBoxes, KDTree
from light import (AmbientLight, AmbientOccluder, AreaLight, DirectionLight, EnvironmentLight,
                   PointLight)
from material import Emissive, GlossyReflective, Matte, Phong, Reflective
from mesh_parser import read_mesh
from sampler import MultiJitteredSampler, RegularSampler
from tracer import AreaLightTracer, Tracer, ViewPlane


def _rgb(r, g, b):
    return numpy.array((float(r), float(g), float(b)))


class AnalyticZoo(BuildFunctionBase):
    """Thin lens, ambient occlusion, directional + point lights, mirror and glossy bounces
    over spheres, planes, boxes, a rectangle, a bare triangle and two instances."""
    BUILD_FUNCTION_NAME = 'features_analytic'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 3
        sampler = MultiJitteredSampler(sample_dim=dim)
        lobe = MultiJitteredSampler(sample_dim=dim, e=25)
        world.viewplane = ViewPlane(resolution=(160, 120), pixel_size=2.0, sampler=sampler)
        world.camera = ThinLensCamera(lens_radius=0.08, focal_plane_distance=14.0,
                                      eye=(9., 5., 11.), up=(0., 1., 0.), lookat=(0., 1., 0.),
                                      viewing_distance=260.)
        world.background_color = (0.05, 0.07, 0.1)
        world.tracer = Tracer(world, 4)
        world.ambient_color = AmbientOccluder(_rgb(1, 1, 1), .4, sampler)

        chalk = Matte(ka=.7, kd=.9, cd=_rgb(.9, .9, .85))
        rose = Matte(ka=.7, kd=.9, cd=_rgb(.9, .4, .45))
        brass = Phong(ka=.5, kd=.7, ks=.6, exp=40, cd=_rgb(.9, .7, .2))
        dull = Phong(ka=.5, kd=.7, ks=.3, exp=6, cd=_rgb(.3, .6, .9))
        dull.receives_shadow = False
        mirror = Reflective(ka=.2, kd=.3, ks=.5, kr=.8, exp=120, cd=_rgb(.9, .95, 1.))
        satin = GlossyReflective(ka=.1, kd=.2, ks=.2, kr=.7, exp=25, cd=_rgb(1., .9, .8), sampler=lobe)

        squash = transform.translate(-3.5, 1.2, 1.0).dot(
            transform.rotate(numpy.array((0.3, 1., 0.2)), 0.7).dot(transform.scale(1.6, 0.8, 1.1)))
        tilt = transform.translate(3.0, 0.9, -2.5).dot(transform.rotate(numpy.array((1., 0., 1.)), 0.5))
        ghost = Sphere(center=_rgb(0, 4.5, -3), radius=0.8, material=rose)
        ghost.cast_shadow = False
        world.objects = [
            CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=chalk, alt_material=rose,
                         up=(1, 0, 0), grid_size=1.25),
            Sphere(center=_rgb(0, 1.5, 0), radius=1.5, material=mirror),
            Instance(Sphere(center=_rgb(0, 0, 0), radius=1., material=brass), squash),
            AxisAlignedBox(2.0, 3.4, 0.0, 1.8, 1.0, 2.6, material=dull),
            Instance(AxisAlignedBox(-0.7, 0.7, -0.7, 0.7, -0.7, 0.7, material=satin), tilt),
            Triangle(_rgb(-1.5, 0.2, 3.5), _rgb(1.5, 0.2, 3.5), _rgb(0., 2.4, 3.0), material=brass),
            Rectangle(_rgb(-5., 2., -4.), 4, 3, _rgb(0., 0., 1.), _rgb(1., 0., 0.), rose, sampler),
            Plane(origin=(0, 0, -9), normal=(0, 0, 1), material=satin),
            ghost,
        ]
        world.lights = [
            DirectionLight(_rgb(1, .95, .9), .9, numpy.array([0.4, 0.8, 0.45]), True),
            DirectionLight(_rgb(.3, .4, .6), .5, numpy.array([-0.7, 0.5, -0.2]), False),
            PointLight(_rgb(1, 1, 1), 1.5, numpy.array((-2., 6., 4.)), radius=6, attenuation=2, cast_shadow=True),
        ]


class MeshPair(BuildFunctionBase):
    """Two kd-trees: the smooth-shaded teapot seen through an Instance and the small flat
    mesh, over a mirror floor with a shadow-casting point light."""
    BUILD_FUNCTION_NAME = 'features_mesh'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 2
        world.viewplane = ViewPlane(resolution=(200, 160), pixel_size=1.6,
                                    sampler=MultiJitteredSampler(sample_dim=dim))
        world.camera = PinholeCamera(eye=(1.5, 3., 8.), up=(0., 1., 0.), lookat=(0., 1.2, 0.),
                                     viewing_distance=300.)
        world.background_color = (0.1, 0.1, 0.12)
        world.tracer = Tracer(world, 3)
        world.ambient_color = AmbientLight(_rgb(1, 1, 1), .25)

        with open('meshes/teapot.obj') as f:
            teapot = read_mesh(f)
        teapot.compute_smooth_normal()
        teapot.material = Phong(ka=.6, kd=.8, ks=.5, exp=30, cd=_rgb(1., .84, .1))
        pot = KDTree(BoundingBoxes(teapot.get_bounding_boxes()))
        with open('meshes/mesh1.obj') as f:
            small = read_mesh(f)
        small.material = Matte(ka=.8, kd=.9, cd=_rgb(.4, .8, .5))
        blob = KDTree(BoundingBoxes(small.get_bounding_boxes()))
        lean = transform.translate(-0.4, 0.1, 0.).dot(
            transform.rotate(numpy.array((0., 1., 0.)), 0.6).dot(transform.scale(0.8, 0.9, 0.8)))
        floor = Reflective(ka=.3, kd=.4, ks=.3, kr=.6, exp=80, cd=_rgb(.9, .9, .9))
        world.objects = [
            Instance(pot, lean),
            Instance(blob, transform.translate(2.8, 0.3, 1.5).dot(transform.scale(1.4))),
            Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=floor),
        ]
        world.lights = [PointLight(_rgb(1, 1, 1), 1.2, numpy.array((2., 8., 5.)), radius=10, attenuation=2,
                                   cast_shadow=True)]


class SkyAndPanel(BuildFunctionBase):
    """AreaLightTracer with an environment light (emissive sky sphere seen from inside)
    and a rectangular panel light whose emitter is also a visible object."""
    BUILD_FUNCTION_NAME = 'features_env'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 3
        sampler = MultiJitteredSampler(sample_dim=dim)
        world.viewplane = ViewPlane(resolution=(160, 160), pixel_size=2.0, sampler=sampler)
        d = (1. / 3) ** 0.5 * 20
        world.camera = PinholeCamera(eye=(d, d, d), up=(0., 1., 0.), lookat=(0., 0., 0.), viewing_distance=200.)
        world.background_color = (0.0, 0.0, 0.0)
        world.tracer = AreaLightTracer(world)
        world.ambient_color = AmbientLight(_rgb(1, 1, 1), .1)

        sky = Emissive(0.6, _rgb(.7, .8, 1.))
        glow = Emissive(5.0, _rgb(1., .9, .7))
        dome = ConcaveSphere(_rgb(0., 0., 0.), 100000., sky)
        dome.cast_shadow = False
        panel = Rectangle(_rgb(0., 10., -20.), 4, 4, _rgb(0., 0., 1.), _rgb(0., 1., 0.), glow, sampler)
        panel.cast_shadow = False
        world.objects = [
            dome,
            panel,
            Sphere(center=_rgb(0., 2.5, 5), radius=5., material=Phong(ka=1, kd=1, ks=1, exp=3, cd=_rgb(1., 1., 0))),
            AxisAlignedBox(-9., -5., 0., 5., -2., 2., material=Matte(ka=1, kd=1, cd=_rgb(.5, .9, .6))),
            Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=Matte(ka=1, kd=1, cd=_rgb(1., 1., 1.))),
        ]
        world.lights = [EnvironmentLight(sky, sampler), AreaLight(glow, panel, 5.)]


class Pile(BuildFunctionBase):
    """4 500 triangles that all span the same box, so the SAH finds nothing to split and the
    kd-tree is one leaf of several thousand rows (more than 64 chunks of 64): the scan paths
    for very large leaves, closest hits and shadows alike."""
    BUILD_FUNCTION_NAME = 'features_pile'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        sampler = MultiJitteredSampler(sample_dim=1 if viewmode == "realtime" else 2)
        world.viewplane = ViewPlane(resolution=(64, 48), pixel_size=2.0, sampler=sampler)
        world.camera = PinholeCamera(eye=(3., 4., 9.), up=(0., 1., 0.), lookat=(0., 1., 0.), viewing_distance=90.)
        world.background_color = (0.1, 0.1, 0.15)
        world.tracer = Tracer(world, 2)
        world.ambient_color = AmbientOccluder(_rgb(1, 1, 1), .5, sampler)
        straw = Matte(ka=.6, kd=.8, cd=_rgb(.9, .8, .4))
        r = numpy.random.RandomState(7)
        boxes = []
        for i in range(4500):
            # corner to corner of the unit-ish box, with a random third vertex: every
            # bounding box is (nearly) the whole pile
            a = _rgb(-1.5, 0.02, -1.5) + r.uniform(0, 0.05, 3)
            b = _rgb(1.5, 2.5, 1.5) - r.uniform(0, 0.05, 3)
            c = _rgb(*r.uniform(-1.5, 1.5, 3)) * _rgb(1, 0, 1) + _rgb(0, r.uniform(0.02, 2.5), 0)
            tri = Triangle(a, b, c, material=straw)
            tri.label = str(i)
            tri.cast_shadow = (i % 7) != 0
            boxes.extend(tri.get_bounding_boxes())
        world.objects = [
            KDTree(BoundingBoxes(boxes)),
            Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=Matte(ka=.6, kd=.8, cd=_rgb(.8, .8, .8))),
        ]
        world.lights = [PointLight(_rgb(1, 1, 1), 2.0, numpy.array((4., 7., 5.)), radius=8, attenuation=2, cast_shadow=True)]


class NestedInstances(BuildFunctionBase):
    """Instances of Instances (geometry.py:85-112 wraps any object): a sphere seen through two
    transforms, a box through three, and the small flat mesh's kd-tree through two, over a
    checker plane with a mirror ball; point light with shadows, one mirror bounce."""
    BUILD_FUNCTION_NAME = 'features_nested'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 2
        sampler = MultiJitteredSampler(sample_dim=dim)
        world.viewplane = ViewPlane(resolution=(160, 120), pixel_size=2.0, sampler=sampler)
        world.camera = PinholeCamera(eye=(7., 5., 10.), up=(0., 1., 0.), lookat=(0., 1., 0.), viewing_distance=240.)
        world.background_color = (0.04, 0.05, 0.08)
        world.tracer = Tracer(world, 3)
        world.ambient_color = AmbientLight(_rgb(1, 1, 1), .3)

        chalk = Matte(ka=.7, kd=.9, cd=_rgb(.9, .9, .85))
        moss = Matte(ka=.7, kd=.9, cd=_rgb(.35, .7, .4))
        brass = Phong(ka=.5, kd=.7, ks=.6, exp=40, cd=_rgb(.9, .7, .2))
        clay = Matte(ka=.6, kd=.8, cd=_rgb(.85, .5, .35))
        mirror = Reflective(ka=.2, kd=.3, ks=.5, kr=.8, exp=120, cd=_rgb(.9, .95, 1.))

        stretch = transform.scale(1.5, 0.7, 1.0)
        turn = transform.rotate(numpy.array((0.2, 1., 0.4)), 0.9)
        lift = transform.translate(-3.0, 1.4, 0.5)
        lean = transform.rotate(numpy.array((1., 0., 0.3)), 0.4)
        away = transform.translate(3.2, 1.0, -2.0)
        with open('meshes/mesh1.obj') as f:
            mesh = read_mesh(f)
        mesh.material = clay
        tree = KDTree(BoundingBoxes(mesh.get_bounding_boxes()))
        grow = transform.scale(9., 9., 9.)
        spin = transform.translate(0.5, 2.6, -3.5).dot(transform.rotate(numpy.array((0., 1., 0.)), 2.2))
        world.objects = [
            CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=chalk, alt_material=moss,
                         up=(1, 0, 0), grid_size=1.5),
            Instance(Instance(Sphere(center=_rgb(0, 0, 0), radius=1., material=brass), stretch), lift.dot(turn)),
            Instance(Instance(Instance(AxisAlignedBox(-0.6, 0.6, -0.6, 0.6, -0.6, 0.6, material=moss), stretch), lean), away),
            Instance(Instance(tree, grow), spin),
            Sphere(center=_rgb(0.5, 1.0, 2.5), radius=1.0, material=mirror),
        ]
        world.lights = [
            PointLight(_rgb(1, 1, 1), 1.6, numpy.array((-1., 7., 5.)), radius=7, attenuation=2, cast_shadow=True),
        ]
