"""Scene geometry as the scene files construct it; intersection runs on the device.

Class names, constructor signatures and public attributes are those of
/root/reference/geometry.py: `BoundingBox` :11, `GeometryObject` :76, `Instance` :85,
`AxisAlignedBox` :115, `Sphere` :245, `ConcaveSphere` :302, `Plane` :310,
`CheckerPlane` :339, `Rectangle` :365, `Triangle` :419, `Mesh` :476.
Each class adds `device_record()`: the constants the kernels read for that primitive.
"""
import sys

import numpy

INF = sys.maxsize
epsilon = 1.0e-7

# primitive tags shared with csrc/rt_scene.h
SPHERE, PLANE, CHECKER_PLANE, RECTANGLE, BOX, TRIANGLE, KDTREE = range(7)


def _vec(v):
    return numpy.array(v, dtype=numpy.float64)


class BoundingBox(object):
    __slots__ = ("data", "obj")

    def __init__(self, x0, x1, y0, y1, z0, z1, obj):
        self.data = [[x0, x1], [y0, y1], [z0, z1]]
        self.obj = obj

    x0 = property(lambda self: self.data[0][0])
    x1 = property(lambda self: self.data[0][1])
    y0 = property(lambda self: self.data[1][0])
    y1 = property(lambda self: self.data[1][1])
    z0 = property(lambda self: self.data[2][0])
    z1 = property(lambda self: self.data[2][1])

    def get_mid_x(self):
        return (self.x0 + self.x1) / 2

    def get_mid_y(self):
        return (self.y0 + self.y1) / 2

    def get_mid_z(self):
        return (self.z0 + self.z1) / 2

    def __str__(self):
        return "BoundingBox: (%f, %f), (%f, %f), (%f, %f)" % (
            self.x0, self.x1, self.y0, self.y1, self.z0, self.z1)


class GeometryObject(object):
    def __init__(self):
        self.cast_shadow = True
        self.inv_transform = None

    def get_material(self):
        return self.material


class Instance(GeometryObject):
    """An object seen through an affine transform (geometry.py:85-112).

    The ray is taken into object space with the inverse matrix, *without*
    renormalising the direction, so `t` means the same in both spaces; the normal
    comes back through the inverse transpose, also not renormalised."""

    def __init__(self, obj, transform):
        self.obj = obj
        self.set_transform(transform)

    def set_transform(self, transform):
        transform = numpy.asarray(transform, dtype=numpy.float64)
        self.inv_transform = numpy.linalg.inv(transform)
        self.inv_normal_transform = numpy.linalg.inv(transform[:3, :3])

    def get_material(self):
        return self.obj.get_material()


class AxisAlignedBox(GeometryObject):
    def __init__(self, x0, x1, y0, y1, z0, z1, material):
        super(AxisAlignedBox, self).__init__()
        self.x0, self.x1 = x0, x1
        self.y0, self.y1 = y0, y1
        self.z0, self.z1 = z0, z1
        self.material = material

    def device_record(self):
        return dict(kind=BOX, lo=_vec((self.x0, self.y0, self.z0)),
                    hi=_vec((self.x1, self.y1, self.z1)), material=self.material)


class Sphere(GeometryObject):
    NORMAL_SIGN = 1.0

    def __init__(self, center, radius, material, sampler=None):
        super(Sphere, self).__init__()
        self.center = numpy.array(center)
        self.radius = numpy.array(radius)
        self.material = material
        self.sampler = sampler

    def device_record(self):
        return dict(kind=SPHERE, center=_vec(self.center), radius=float(self.radius),
                    normal_sign=self.NORMAL_SIGN, material=self.material)


class ConcaveSphere(Sphere):
    """Sphere seen from inside: the hit normal is negated (geometry.py:302-307)."""
    NORMAL_SIGN = -1.0


class Plane(GeometryObject):
    def __init__(self, origin, normal, material):
        super(Plane, self).__init__()
        self.origin = numpy.array(origin)
        self.normal = numpy.array(normal)
        self.material = material

    def device_record(self):
        # the normal is used as given: neither normalised nor flipped (geometry.py:324)
        return dict(kind=PLANE, point=_vec(self.origin), normal=_vec(self.normal),
                    material=self.material)


class CheckerPlane(Plane):
    def __init__(self, origin, normal, material, alt_material, up, grid_size):
        super(CheckerPlane, self).__init__(origin, normal, material)
        self.grid_size = grid_size
        self.up = up
        self.right = numpy.cross(normal, up)
        self.alt_material = alt_material

    def device_record(self):
        rec = super(CheckerPlane, self).device_record()
        rec.update(kind=CHECKER_PLANE, up=_vec(self.up), right=_vec(self.right),
                   grid_size=float(self.grid_size), alt_material=self.alt_material)
        return rec


class Rectangle(GeometryObject):
    """Area-light shape.  Hit tests treat `p0` as the centre with half extents a/2, b/2
    along `right` and `top`; `sample()` treats `p0` as a corner and offsets along world
    x and y (geometry.py:378-380 vs :392-399).  Both behaviours are kept."""

    def __init__(self, p0, a, b, normal, right, material, sampler):
        super(Rectangle, self).__init__()
        self.p0 = p0
        self.a = a
        self.b = b
        self.normal = normal
        self.right = right
        self.top = numpy.cross(normal, right)
        self.sampler = sampler
        self.inv_area = 1. / (a * b)
        self.material = material

    def pdf(self, shader_rec):
        return self.inv_area

    def get_normal(self, sample_point):
        return self.normal

    def emitter_record(self):
        return dict(corner=self.p0, edge_u=(self.a, 0., 0.), edge_v=(0., self.b, 0.),
                    normal=self.normal, pdf=self.inv_area, sampler=self.sampler)

    def device_record(self):
        return dict(kind=RECTANGLE, point=_vec(self.p0), normal=_vec(self.normal),
                    right=_vec(self.right), top=_vec(self.top),
                    a=float(self.a), b=float(self.b), material=self.material)


def _face_normal(v0, v1, v2):
    n = numpy.cross(v1 - v0, v2 - v0)
    return n / numpy.linalg.norm(n)


class Triangle(GeometryObject):
    def __init__(self, v0, v1, v2, material, normal=None):
        super(Triangle, self).__init__()
        self.v0 = v0
        self.v1 = v1
        self.v2 = v2
        self.normal = _face_normal(v0, v1, v2) if normal is None else normal
        self.material = material
        self.n0 = self.n1 = self.n2 = None

    def get_bounding_boxes(self):
        lo = numpy.minimum(numpy.minimum(self.v0, self.v1), self.v2)
        hi = numpy.maximum(numpy.maximum(self.v0, self.v1), self.v2)
        return [BoundingBox(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], self)]

    def device_record(self):
        return dict(kind=TRIANGLE, triangle=self, material=self.material)


class Mesh(GeometryObject):
    """Indexed triangle mesh; quads are split (0,1,2),(0,2,3) (geometry.py:481-486)."""

    def __init__(self, vertices, indices, material=None):
        super(Mesh, self).__init__()
        self.vertices = vertices
        self.indices = []
        for face in indices:
            face = list(face)
            if len(face) == 4:
                self.indices.append([face[0], face[1], face[2]])
                self.indices.append([face[0], face[2], face[3]])
            else:
                self.indices.append(face)
        self.normals = []
        self.material = material
        self.vertices_normal = None
        self.compute_normal()

    def _corner_arrays(self):
        verts = numpy.asarray(self.vertices, dtype=numpy.float64).reshape(-1, 3)
        idx = numpy.asarray(self.indices, dtype=numpy.int64).reshape(-1, 3)
        return verts[idx[:, 0]], verts[idx[:, 1]], verts[idx[:, 2]], idx

    def compute_normal(self):
        """Unit face normals, appended to `self.normals` (geometry.py:498-502: a second
        call appends again, the first set stays in front and stays the one in use)."""
        if not self.indices:
            return
        p0, p1, p2, _ = self._corner_arrays()
        d, e = p1 - p0, p2 - p0
        n = numpy.empty_like(d)
        n[:, 0] = d[:, 1] * e[:, 2] - d[:, 2] * e[:, 1]
        n[:, 1] = d[:, 2] * e[:, 0] - d[:, 0] * e[:, 2]
        n[:, 2] = d[:, 0] * e[:, 1] - d[:, 1] * e[:, 0]
        length = numpy.sqrt(n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
        n /= length[:, None]
        self.normals.extend(n)

    def compute_smooth_normal(self):
        """Vertex normals = normalised, area-unweighted sum of the adjacent face normals,
        accumulated in face order (geometry.py:504-515)."""
        acc = numpy.zeros((len(self.vertices), 3))
        for face, normal in zip(self.indices, self.normals):
            for v in face:
                acc[v] += normal
        with numpy.errstate(invalid="ignore", divide="ignore"):
            length = numpy.sqrt(acc[:, 0] * acc[:, 0] + acc[:, 1] * acc[:, 1] + acc[:, 2] * acc[:, 2])
            acc /= length[:, None]
        self.vertices_normal = list(acc)

    def get_bounding_boxes(self):
        """One `Triangle` in its own `BoundingBox` per face, labelled with the face index
        after quad splitting; the triangles inherit `cast_shadow` (geometry.py:575-588)."""
        if not self.indices:
            return []
        p0, p1, p2, idx = self._corner_arrays()
        lo = numpy.minimum(numpy.minimum(p0, p1), p2)
        hi = numpy.maximum(numpy.maximum(p0, p1), p2)
        boxes = []
        for i in range(len(idx)):
            tri = Triangle(p0[i], p1[i], p2[i], self.material, normal=self.normals[i])
            if self.vertices_normal:
                tri.n0, tri.n1, tri.n2 = (self.vertices_normal[v] for v in idx[i])
            tri.label = str(i)
            tri.cast_shadow = self.cast_shadow
            boxes.append(BoundingBox(lo[i, 0], hi[i, 0], lo[i, 1], hi[i, 1], lo[i, 2], hi[i, 2], tri))
        return boxes

    def device_record(self):
        raise NotImplementedError(
            "a bare Mesh in world.objects is tested brute-force by the reference "
            "(geometry.py:517-531) and its shadow_hit is unusable; wrap it in a KDTree")
