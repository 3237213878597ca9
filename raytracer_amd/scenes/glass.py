"""Scenes for the two-child materials (Transparent, Dielectric) under the Whitted tracer.
Test scenes of this project, written against the reference's scene API only, so the
reference itself can render them when golden vectors are generated."""
import numpy
import affinetransform as transform
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import AxisAlignedBox, CheckerPlane, Instance, Plane, Sphere
from kdtree import BoundingBoxes, KDTree
from light import AmbientLight, AmbientOccluder, DirectionLight, PointLight
from material import Dielectric, GlossyReflective, Matte, Phong, Reflective, Transparent
from mesh_parser import read_mesh
from sampler import MultiJitteredSampler
from tracer import Tracer, ViewPlane, WhittedTracer


def _rgb(r, g, b):
    return numpy.array((float(r), float(g), float(b)))


class GlassShelf(BuildFunctionBase):
    """A Dielectric ball (colour filter in and out), a Dielectric ball seen through an
    Instance, a Transparent slab and ball, a glossy ball and a mirror wall: every branch
    of the two-child materials, with draws (ambient occlusion, glossy lobe) inside the
    ray tree so that the order of the children matters."""
    BUILD_FUNCTION_NAME = 'features_glass'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 3
        sampler = MultiJitteredSampler(sample_dim=dim)
        lobe = MultiJitteredSampler(sample_dim=dim, e=40)
        world.viewplane = ViewPlane(resolution=(160, 120), pixel_size=2.0, sampler=sampler)
        world.camera = PinholeCamera(eye=(2., 4.5, 12.), up=(0., 1., 0.), lookat=(0., 1.2, 0.),
                                     viewing_distance=250.)
        world.background_color = numpy.array((0.15, 0.2, 0.3))
        world.tracer = WhittedTracer(world, 4)
        world.ambient_color = AmbientOccluder(_rgb(1, 1, 1), .5, sampler)

        chalk = Matte(ka=.6, kd=.8, cd=_rgb(.9, .9, .85))
        slate = Matte(ka=.6, kd=.8, cd=_rgb(.25, .3, .4))
        tinted = Dielectric(ka=0, kd=0, ks=.2, cd=_rgb(1., 1., 1.), exp=200, sampler=None,
                            eta_in=1.5, eta_out=1.0, cf_in=_rgb(.65, .85, .7), cf_out=_rgb(1., .98, .96))
        bubble = Dielectric(ka=0.05, kd=0.05, ks=.1, cd=_rgb(.9, .95, 1.), exp=80, sampler=None,
                            eta_in=0.75, eta_out=1, cf_in=.9, cf_out=_rgb(.95, .9, .85))
        pane = Transparent(ka=.1, kd=.1, ks=.3, cd=_rgb(.8, .9, 1.), exp=60, sampler=None,
                           ior=1.3, kt=.85, kr=.12)
        drop = Transparent(ka=0, kd=0, ks=.5, cd=_rgb(1., .9, .9), exp=300, sampler=None,
                           ior=0.8, kt=.9, kr=.1)
        satin = GlossyReflective(ka=.1, kd=.2, ks=.2, kr=.7, exp=40, cd=_rgb(1., .85, .6), sampler=lobe)
        mirror = Reflective(ka=.1, kd=.2, ks=.3, kr=.75, exp=100, cd=_rgb(.9, .95, 1.))

        squash = transform.translate(3.2, 1.0, 1.5).dot(
            transform.rotate(numpy.array((0.2, 1., 0.3)), 0.5).dot(transform.scale(1.3, 0.8, 1.0)))
        slab = AxisAlignedBox(-4.2, -3.6, 0.05, 3.0, -1.5, 2.5, material=pane)
        slab.cast_shadow = False
        world.objects = [
            CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=chalk, alt_material=slate,
                         up=(1, 0, 0), grid_size=1.0),
            Sphere(center=_rgb(0., 1.5, 0.), radius=1.5, material=tinted),
            Instance(Sphere(center=_rgb(0, 0, 0), radius=1., material=bubble), squash),
            slab,
            Sphere(center=_rgb(-1.8, 0.7, 3.0), radius=0.7, material=drop),
            Sphere(center=_rgb(1.6, 0.8, -3.0), radius=0.8, material=satin),
            Plane(origin=(0, 0, -7), normal=(0, 0, 1), material=mirror),
        ]
        world.lights = [
            PointLight(_rgb(1, 1, 1), 1.4, numpy.array((-3., 7., 5.)), radius=7, attenuation=2, cast_shadow=True),
            DirectionLight(_rgb(.6, .7, 1.), .5, numpy.array([0.5, 0.7, 0.3]), True),
        ]


class GlassTeapot(BuildFunctionBase):
    """The smooth-shaded teapot mesh as a Dielectric inside a kd-tree - the arrangement of
    the reference's `transparent` scene (buildfunctions/transparent.py:53-96) at test size."""
    BUILD_FUNCTION_NAME = 'glass_teapot'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        overrides = getattr(world, "overrides", None) or {}
        dim = overrides.get("sample_dim", 1 if viewmode == "realtime" else 2)
        world.viewplane = ViewPlane(resolution=overrides.get("resolution", (160, 160)), pixel_size=2.0,
                                    sampler=MultiJitteredSampler(sample_dim=dim))
        world.camera = PinholeCamera(eye=(0, 8., 0.), up=(1., 0., 0.), lookat=(0., 1.5, 0.), viewing_distance=300.)
        world.background_color = numpy.array((0.2, 0.2, 0.2))
        world.tracer = WhittedTracer(world, overrides.get("depth_limit", 5))
        world.ambient_color = AmbientLight(_rgb(1, 1, 1), .2)

        glass = Dielectric(ka=0, kd=0, ks=0, cd=_rgb(1., 1., 1.), exp=0, sampler=None,
                           eta_in=0.8, eta_out=1, cf_in=_rgb(.9, .6, .5), cf_out=_rgb(.9, .6, .5))
        with open('meshes/teapot.obj') as f:
            mesh = read_mesh(f)
        mesh.compute_smooth_normal()
        mesh.cast_shadow = False
        mesh.material = glass
        world.objects = [
            KDTree(BoundingBoxes(mesh.get_bounding_boxes())),
            Sphere(center=_rgb(-2.5, 0.5, -2.5), radius=2., material=Phong(ka=1, kd=1, ks=1, exp=100, cd=_rgb(1., .84, .1))),
            CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=Matte(ka=1, kd=1, cd=_rgb(1., 1., 1.)),
                         alt_material=Matte(ka=1, kd=1, cd=_rgb(1., .84, 1.)), up=(1, 0, 0), grid_size=0.5),
        ]
        world.lights = [PointLight(_rgb(1, 1, 1), 1., numpy.array((1., 8., 2.)), radius=10, attenuation=2,
                                   cast_shadow=True)]


class PaneUnderTracer(BuildFunctionBase):
    """Transparent under the plain `Tracer` (its shade never passes the extra argument that
    only WhittedTracer takes), in front of a mirror."""
    BUILD_FUNCTION_NAME = 'features_pane'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 2
        world.viewplane = ViewPlane(resolution=(120, 96), pixel_size=2.0,
                                    sampler=MultiJitteredSampler(sample_dim=dim))
        world.camera = PinholeCamera(eye=(3., 3., 9.), up=(0., 1., 0.), lookat=(0., 1., 0.), viewing_distance=180.)
        world.background_color = numpy.array((0.3, 0.35, 0.4))
        world.tracer = Tracer(world, 3)
        world.ambient_color = AmbientLight(_rgb(1, 1, 1), .3)
        pane = Transparent(ka=.1, kd=.2, ks=.4, cd=_rgb(.9, 1., .9), exp=50, sampler=None, ior=1.4, kt=.8, kr=.15)
        world.objects = [
            Sphere(center=_rgb(0., 1.2, 0.), radius=1.2, material=pane),
            AxisAlignedBox(1.8, 3.0, 0.04, 2.0, -1.0, 0.2, material=pane),
            Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=Matte(ka=.7, kd=.8, cd=_rgb(.8, .7, .6))),
            Plane(origin=(0, 0, -5), normal=(0, 0, 1),
                  material=Reflective(ka=.1, kd=.2, ks=.3, kr=.7, exp=100, cd=_rgb(.9, .9, 1.))),
        ]
        world.lights = [PointLight(_rgb(1, 1, 1), 2., numpy.array((4., 6., 6.)), radius=1, attenuation=0,
                                   cast_shadow=True)]
