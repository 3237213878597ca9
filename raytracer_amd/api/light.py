"""Light descriptions, same constructors as /root/reference/light.py
(`AmbientLight` :5, `PointLight` :18, `DirectionLight` :47, `AmbientOccluder` :68,
`AreaLight` :103, `EnvironmentLight` :143).

The per-hit work (direction, shadow query, radiance, geometry term) runs in the
kernels; each class only says which constants the kernels need.
"""
import numpy

POINT, DIRECTIONAL, AREA, ENVIRONMENT = range(4)
AMBIENT_CONSTANT, AMBIENT_OCCLUDER = range(2)


def _vec(v):
    return numpy.array(v, dtype=numpy.float64)


class AmbientLight(object):
    AMBIENT_KIND = AMBIENT_CONSTANT

    def __init__(self, color, ls):
        self.color = color
        self.ls = ls
        self.cast_shadow = False

    def device_record(self):
        return dict(kind=self.AMBIENT_KIND, radiance=_vec(self.ls * _vec(self.color)))


class AmbientOccluder(object):
    AMBIENT_KIND = AMBIENT_OCCLUDER

    def __init__(self, color, ls, sampler):
        self.color = color
        self.ls = ls
        self.sampler = sampler

    def device_record(self):
        # light.py:97-100: occluded -> (0.1,0.1,0.1)*ls*color, else ls*color
        return dict(kind=self.AMBIENT_KIND,
                    radiance=_vec(self.ls * _vec(self.color)),
                    occluded=_vec(numpy.array((0.1, 0.1, 0.1)) * self.ls * _vec(self.color)),
                    sampler=self.sampler)


class PointLight(object):
    KIND = POINT

    def __init__(self, color, ls, location, attenuation=0, radius=300, cast_shadow=False):
        self.color = color
        self.ls = ls
        self.location = location
        self.attenuation = attenuation
        self.radius = radius
        self.cast_shadow = cast_shadow

    def device_record(self):
        return dict(kind=self.KIND, radiance=_vec(self.ls * _vec(self.color)),
                    position=_vec(self.location), attenuation=float(self.attenuation),
                    radius=float(self.radius), cast_shadow=bool(self.cast_shadow))


class DirectionLight(object):
    KIND = DIRECTIONAL

    def __init__(self, color, ls, direction, cast_shadow=False):
        self.color = color
        self.ls = ls
        self.direction = direction
        self.cast_shadow = cast_shadow

    def device_record(self):
        # the direction is used as given, not normalised (light.py:54-55)
        return dict(kind=self.KIND, radiance=_vec(self.ls * _vec(self.color)),
                    direction=_vec(self.direction), cast_shadow=bool(self.cast_shadow))


class AreaLight(object):
    KIND = AREA

    def __init__(self, material, shape, radius, cast_shadow=True):
        self.material = material
        self.radius = radius
        self.shape = shape
        self.cast_shadow = cast_shadow

    def device_record(self):
        """The emitter must describe itself as a parallelogram
        `corner + sx*edge_u + sy*edge_v` with a constant normal and pdf.

        `Rectangle` does (geometry.py:378-386: corner p0, edges (a,0,0) and (0,b,0));
        any other shape passes if it offers `emitter_record()` with the same keys."""
        shape = self.shape.emitter_record()
        return dict(kind=self.KIND, radiance=_vec(self.material.get_Le(None)),
                    radius=float(self.radius), cast_shadow=bool(self.cast_shadow),
                    corner=_vec(shape["corner"]), edge_u=_vec(shape["edge_u"]),
                    edge_v=_vec(shape["edge_v"]), normal=_vec(shape["normal"]),
                    inv_pdf_area=float(shape["pdf"]), sampler=shape["sampler"])


class EnvironmentLight(object):
    KIND = ENVIRONMENT

    def __init__(self, material, sampler, cast_shadow=True):
        self.material = material
        self.sampler = sampler
        self.cast_shadow = cast_shadow

    def device_record(self):
        return dict(kind=self.KIND, radiance=_vec(self.material.get_Le(None)),
                    cast_shadow=bool(self.cast_shadow), sampler=self.sampler)
