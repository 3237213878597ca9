"""A scene for SV_Matte with image textures.  The reference's own textured scene needs an
image file it does not ship, so this one paints its textures procedurally; it is written
against the reference's scene API only, so the reference itself can render it when golden
vectors are generated."""
import numpy
import affinetransform as transform
from PIL import Image
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import Instance, Plane, Rectangle, Sphere
from light import AmbientOccluder, DirectionLight, PointLight
from material import Matte, Reflective, SV_Matte
from sampler import MultiJitteredSampler
from texture import ImageTexture, SphericalMap
from tracer import Tracer, ViewPlane


def _rgb(r, g, b):
    return numpy.array((float(r), float(g), float(b)))


def _bands(width, height):
    """An RGB test card: hue bands along the columns, a brightness ramp down the rows and
    a few hard-edged patches, so that a wrong texel is a visibly different colour."""
    col = numpy.arange(width)[None, :]
    row = numpy.arange(height)[:, None]
    img = numpy.zeros((height, width, 3), dtype=numpy.uint8)
    img[:, :, 0] = (col * 255 // max(width - 1, 1)) ^ (row * 7 % 64)
    img[:, :, 1] = (row * 255 // max(height - 1, 1))
    img[:, :, 2] = ((col // 4 + row // 4) % 2) * 200 + 30
    return Image.fromarray(img, "RGB")


def _grey(width, height):
    col = numpy.arange(width)[None, :]
    row = numpy.arange(height)[:, None]
    return Image.fromarray(((col * 5 + row * 11) % 256).astype(numpy.uint8), "L")


class TexturedBalls(BuildFunctionBase):
    """Unit spheres with an RGB and a grey texture (one seen through an Instance, one seen
    again in a mirror), and a textured rectangle seen from behind, where SV_Matte's
    viewer-side test switches the lights off."""
    BUILD_FUNCTION_NAME = 'features_texture'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        dim = 1 if viewmode == "realtime" else 3
        sampler = MultiJitteredSampler(sample_dim=dim)
        world.viewplane = ViewPlane(resolution=(160, 120), pixel_size=2.0, sampler=sampler)
        world.camera = PinholeCamera(eye=(1., 2.5, 10.), up=(0., 1., 0.), lookat=(0., 0., 0.), viewing_distance=300.)
        world.background_color = numpy.array((0.1, 0.12, 0.2))
        world.tracer = Tracer(world, 3)
        world.ambient_color = AmbientOccluder(_rgb(1, 1, 1), .6, sampler)

        mapping = SphericalMap()
        card = ImageTexture(image=_bands(96, 48), color=_rgb(1, 1, 1), mapping=mapping)
        mist = ImageTexture(image=_grey(40, 30), color=_rgb(1, 1, 1), mapping=mapping, hres=32, vres=24)
        painted = SV_Matte(ka=.7, kd=.9, cd=card)
        cloudy = SV_Matte(ka=.5, kd=1.2, cd=mist)
        chalk = Matte(ka=.5, kd=.7, cd=_rgb(.9, .9, .85))
        mirror = Reflective(ka=.1, kd=.2, ks=.3, kr=.8, exp=100, cd=_rgb(.9, .95, 1.))

        lean = transform.translate(2.6, 0.2, 1.0).dot(
            transform.rotate(numpy.array((0.3, 1., 0.1)), 0.9).dot(transform.scale(1.2, 0.9, 1.0)))
        world.objects = [
            Sphere(center=_rgb(0., 0., 0.), radius=1., material=painted),
            Instance(Sphere(center=_rgb(0., 0., 0.), radius=1., material=painted), lean),
            Sphere(center=_rgb(-2.5, -0.2, 1.5), radius=1., material=cloudy),
            Rectangle(_rgb(-1., 0., 4.), 1.5, 1.6, _rgb(0., 0., -1.), _rgb(1., 0., 0.), painted, sampler),
            Plane(origin=(0, -1.2, 0), normal=(0, 1, 0), material=chalk),
            Plane(origin=(0, 0, -4), normal=(0, 0, 1), material=mirror),
        ]
        world.lights = [
            PointLight(_rgb(1, 1, 1), 1.6, numpy.array((3., 6., 6.)), radius=7, attenuation=2, cast_shadow=True),
            DirectionLight(_rgb(1., .9, .8), .5, numpy.array([-0.4, 0.8, 0.4]), True),
        ]
