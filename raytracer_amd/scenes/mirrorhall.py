"""A mirror sphere between three mirror planes.

'mirrorhall' is the world of /root/reference/buildfunctions/mirrorhall.py:18-54
(200x200, one sample, pinhole).  'mirrorhall_dof' is the 4K depth-of-field variant of
benchmark configuration 5: 3840x2160, 64 samples per pixel, thin-lens camera, and the sphere
placed through an affine Instance (stretched and tilted) next to a second plain one."""
import numpy
import affinetransform as transform
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera, ThinLensCamera
from geometry import Instance, Plane, Sphere
from light import AmbientLight, PointLight
from material import Reflective
from sampler import MultiJitteredSampler, RegularSampler
from tracer import Tracer, ViewPlane

R2 = 2 ** 0.5 / 2
EYE, LOOKAT = (14., 0., 14.), (0., 1.5, 0.)


def _hall(world, mirror, centre_piece):
    world.background_color = (0.0, 0.0, 0.0)
    world.tracer = Tracer(world)
    world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
    world.objects = list(centre_piece) + [
        Plane(origin=(-2, 0, 0), normal=(-R2, 0, R2), material=mirror),
        Plane(origin=(0, 0, -2), normal=(R2, 0, -R2), material=mirror),
        Plane(origin=(0, -1, 0), normal=(0, 1, 0), material=mirror),
    ]
    world.lights = [PointLight(numpy.array((1., 1., 1.)), 1., numpy.array((1., 8., 2.)), radius=10,
                               attenuation=2, cast_shadow=True)]


class MirrorHall(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'mirrorhall'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        preview = viewmode == "realtime"
        world.viewplane = ViewPlane(resolution=(64, 64) if preview else (200, 200),
                                    pixel_size=5 if preview else .64 * 5 / 2,
                                    sampler=RegularSampler() if preview else MultiJitteredSampler(sample_dim=1))
        world.camera = PinholeCamera(eye=EYE, up=(0., 1., 0.), lookat=LOOKAT, viewing_distance=300.)
        mirror = Reflective(ka=.6, kd=.6, ks=.6, kr=1.0, exp=100, cd=numpy.array((1., 1., 1.)))
        _hall(world, mirror, [Sphere(center=numpy.array((0, 0, 0)), radius=1, material=mirror)])


class MirrorHallDepthOfField(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'mirrorhall_dof'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        opts = getattr(world, "overrides", None) or {}
        width, height = opts.get("resolution", (64, 36) if viewmode == "realtime" else (3840, 2160))
        dim = opts.get("sample_dim", 1 if viewmode == "realtime" else 8)
        world.viewplane = ViewPlane(resolution=(width, height), pixel_size=320. / height,
                                    sampler=MultiJitteredSampler(sample_dim=dim))
        # focus on the sphere: |eye - origin| is about 19.8, the view plane sits at 300
        world.camera = ThinLensCamera(lens_radius=0.35, focal_plane_distance=19.8,
                                      eye=EYE, up=(0., 1., 0.), lookat=LOOKAT, viewing_distance=300.)
        mirror = Reflective(ka=.6, kd=.6, ks=.6, kr=1.0, exp=100, cd=numpy.array((1., 1., 1.)))
        tint = Reflective(ka=.6, kd=.6, ks=.6, kr=.8, exp=100, cd=numpy.array((1., .7, .4)))
        unit = Sphere(center=numpy.array((0., 0., 0.)), radius=1, material=tint)
        place = transform.translate(1.2, 0.4, 1.2).dot(
            transform.rotate(numpy.array((0., 0.6, 0.8)), 0.9).dot(transform.scale(0.6, 1.3, 0.6)))
        _hall(world, mirror, [Sphere(center=numpy.array((0, 0, 0)), radius=1, material=mirror),
                              Instance(unit, place)])
