"""Ray / shade-record value types, the tracer flavours and the view plane.

Interface of /root/reference/tracer.py: `Ray` :7, `ShadeRecord` :13, `Tracer` :21,
`WhittedTracer` :38, `AreaLightTracer` :61, `PathTracer` :74, `ViewPlane` :91.  A tracer here is a
description (kind + depth limit) that the flattener turns into kernel parameters;
`trace_ray` hands the ray to the device through the owning world.
"""
import sys

import numpy

from sampler import RegularSampler, MultiJitteredSampler  # noqa: F401  (re-exported, as the reference does)

INF = sys.maxsize
epsilon = 1.0e-7


class Ray(object):
    def __init__(self, origin, direction):
        self.origin = numpy.array(origin, dtype=numpy.float64)
        self.direction = numpy.array(direction, dtype=numpy.float64)


class ShadeRecord(object):
    def __init__(self, local_hit_point, normal, tmin, material=None):
        self.local_hit_point = local_hit_point
        self.normal = normal
        self.tmin = tmin
        self.material = material


class _DeviceTracer(object):
    """Common part: remembers the world and sends rays to its device scene."""
    KIND = None
    has_depth_limit = True

    def __init__(self, world, depth_limit=5):
        self.world = world
        self.depth_limit = depth_limit

    def trace_ray(self, ray, depth=0):
        """Radiance along one ray (tracer.py:26-35); runs on the GPU.

        Draws inside the trace are keyed by the world's current draw context
        (pixel index, sample index), as every other in-trace draw is."""
        from raytracer_amd import runtime
        rgb = runtime.trace_rays(self.world, [ray], depth=depth)
        return rgb[0]


class Tracer(_DeviceTracer):
    """`material.shade`; background colour on a miss or past the depth limit."""
    KIND = "shade"


class AreaLightTracer(_DeviceTracer):
    """`material.area_light_shade`; black on a miss; no depth test (tracer.py:61-71)."""
    KIND = "area"
    has_depth_limit = False

    def __init__(self, world):
        super(AreaLightTracer, self).__init__(world, depth_limit=0)


class PathTracer(_DeviceTracer):
    """`material.path_shade`; black on a miss or past the depth limit (tracer.py:74-88)."""
    KIND = "path"


class WhittedTracer(_DeviceTracer):
    """As `Tracer`, and also tells the calling material how far the ray went
    (tracer.py:38-58); `Dielectric.shade` filters its children's radiance by that length."""
    KIND = "whitted"


class ViewPlane(object):
    def __init__(self, resolution, pixel_size, sampler=RegularSampler):
        self.resolution = resolution
        self.pixel_size = pixel_size
        # the reference's default is the class itself (tracer.py:92); accept both
        self.sampler = sampler() if isinstance(sampler, type) else sampler

    def iter_row(self, row):
        for column in range(self.resolution[0]):
            yield (column, row)

    def __iter__(self):
        for row in range(self.resolution[1]):
            yield self.iter_row(row)
