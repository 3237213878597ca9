"""Glossy and matte spheres on a checker floor, 400x400, 100 samples per pixel.
Same world as /root/reference/buildfunctions/glossyreflective.py:18-69 (benchmark configuration 3)."""
import numpy
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import CheckerPlane, Plane, Sphere
from light import AmbientLight, PointLight
from material import GlossyReflective, Matte, Phong
from sampler import MultiJitteredSampler, RegularSampler
from tracer import Tracer, ViewPlane

LOBE = 100


class GlossyScene(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'glossyreflective'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        if viewmode == "realtime":
            size, pixel, pixels, lobe = (64, 64), 5, RegularSampler(), RegularSampler()
        else:
            size, pixel = (400, 400), .64 * 1.25
            pixels = MultiJitteredSampler(sample_dim=10)
            lobe = MultiJitteredSampler(sample_dim=10, e=LOBE)
        world.viewplane = ViewPlane(resolution=size, pixel_size=pixel, sampler=pixels)
        world.camera = PinholeCamera(eye=(0., 2., -7.), up=(0., 1., 0.), lookat=(0., 1.5, 0.),
                                     viewing_distance=300.)
        world.background_color = (0.0, 0.0, 0.0)
        world.tracer = Tracer(world)

        gold = Phong(ka=1, kd=1, ks=1, exp=100, cd=numpy.array((1., .84, .1)))
        pink = Matte(ka=1, kd=1, cd=numpy.array([1., .84, 1.]))
        white = Matte(ka=1, kd=1, cd=numpy.array([1., 1., 1.]))
        blue = Matte(ka=1, kd=1, cd=numpy.array([.2, .3, 1.]))
        sharp = GlossyReflective(ka=0., kd=0., ks=0., kr=1.0, exp=LOBE, cd=numpy.array((1., 1., 1.)), sampler=lobe)
        blurry = GlossyReflective(ka=0., kd=0., ks=0., kr=1.0, exp=10, cd=numpy.array((1., 1., 1.)), sampler=lobe)

        world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
        world.objects = [
            Sphere(center=numpy.array((0, 1, 0)), radius=1., material=sharp),
            Sphere(center=numpy.array((-2.5, 0.5, -1.5)), radius=1., material=gold),
            Sphere(center=numpy.array((3.5, 1.5, 0.5)), radius=2., material=blue),
            Sphere(center=numpy.array((2.5, 0.5, -3.5)), radius=.7, material=blurry),
            CheckerPlane(origin=(0, 0, 0), normal=(0, 1, 0), material=white, alt_material=pink,
                         up=(1, 0, 0), grid_size=0.5),
            Plane(origin=(0, 0, 8), normal=(0, 0, -1), material=pink),
        ]
        world.lights = [PointLight(numpy.array((1., 1., 1.)), 1., numpy.array((1., 8., 2.)), radius=10,
                                   attenuation=2, cast_shadow=True)]
