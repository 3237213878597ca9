"""4x4 affine helpers with the names and conventions of
/root/reference/affinetransform.py (rotate :4, scale :22, translate :33, apply :40,
apply_normal :44).  Scenes compose these on the host; `Instance` inverts the product
and the kernels apply the inverse to rays (geometry.py:90-112).
"""
import numpy


def rotate(axis, theta):
    """Rotation by `theta` radians about `axis`, Euler-Rodrigues form.

    Same parameter products as affinetransform.py:11-19 so that a scene's instance
    matrices come out bit-identical."""
    axis = numpy.asarray(axis)
    theta = numpy.asarray(theta)
    unit = axis / numpy.sqrt(numpy.dot(axis, axis))
    a = numpy.cos(theta / 2)
    b, c, d = -unit * numpy.sin(theta / 2)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    m = numpy.zeros((4, 4))
    m[0, :3] = (aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac))
    m[1, :3] = (2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab))
    m[2, :3] = (2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc)
    m[3, 3] = 1
    return m


def scale(x, y=None, z=None):
    if y is None:
        y = z = x
    return numpy.diag(numpy.array([x, y, z, 1], dtype=numpy.result_type(x, y, z, 1)))


def translate(x, y, z):
    m = numpy.identity(4, dtype=numpy.result_type(x, y, z, 1))
    m[:3, 3] = (x, y, z)
    return m


def apply(mat, vec):
    """Transform a point (homogeneous w = 1)."""
    return mat.dot(numpy.concatenate((vec, [1.])))[:3]


def apply_normal(mat, vec):
    """Transform a direction or normal with a 3x3 matrix."""
    return mat.dot(vec)
