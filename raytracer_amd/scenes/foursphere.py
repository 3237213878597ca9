"""Four mirror spheres around a point light, 200x200, one sample per pixel, 20 bounces.
Same world as /root/reference/buildfunctions/foursphere.py:18-60 (benchmark configuration 1)."""
import numpy
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import Sphere
from light import AmbientLight, PointLight
from material import Reflective
from sampler import MultiJitteredSampler, RegularSampler
from tracer import Tracer, ViewPlane

WHITE = numpy.array((1., 1., 1.))
H = 1 / 2 ** 0.5
# (centre, surface colour) in world.objects order
BALLS = (((1, 0, -H), (1., .1, .1)),
         ((-1, 0, -H), (1., 1., 1.)),
         ((0, 1, H), (.1, 1., .1)),
         ((0, -1, H), (.1, .1, 1.)))


class FourSphere(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'foursphere'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        preview = viewmode == "realtime"
        world.viewplane = ViewPlane(resolution=(64, 64) if preview else (200, 200),
                                    pixel_size=5 if preview else .64 * 2.5,
                                    sampler=RegularSampler() if preview else MultiJitteredSampler(sample_dim=1))
        world.camera = PinholeCamera(eye=(4., 0., 4.), up=(0., 1., 0.), lookat=(0., 0., 0.),
                                     viewing_distance=1300.)
        world.background_color = (0.0, 0.0, 0.0)
        world.tracer = Tracer(world, 20)
        world.ambient_color = AmbientLight(WHITE, .05)
        world.objects = [
            Sphere(center=numpy.array(c), radius=1,
                   material=Reflective(ka=0.6, kd=0.6, ks=0.6, kr=1.0, exp=500, cd=numpy.array(rgb)))
            for c, rgb in BALLS]
        world.lights = [PointLight(WHITE, 0.4, numpy.array((0., 0., 0.)), radius=1, attenuation=2,
                                   cast_shadow=False)]
