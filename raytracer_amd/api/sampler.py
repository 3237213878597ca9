"""Sample-pattern tables, built on the host and consumed by the device kernels.

Mirrors the constructor signatures and attributes of /root/reference/sampler.py
(`RegularSampler` :5-14, `MultiJitteredSampler` :17-85).  The tables are what the
reference builds; the random draws come from `raytracer_amd.rng` instead of the
unseeded global `random` stream, in the same call order as sampler.py:24-31.
"""
import numpy

from raytracer_amd import rng


class RegularSampler(object):
    """One centred sample, no randomness (sampler.py:5-14)."""
    sample_dim = 1
    pattern_size = 1

    def __init__(self):
        self.patterns = [[(0.5, 0.5)]]
        self.concentric_patterns = [[numpy.array((0., 0.))]]
        self.hemisphere_patterns = [numpy.array((0., 0., 1.))]

    def sample(self):
        return self.patterns[0]

    def sample_unit_disk(self):
        return self.concentric_patterns[0]

    def sample_unit_hemisphere_surface(self):
        return self.hemisphere_patterns[0]


def _concentric(sx, sy):
    # Shirley's square-to-disk map as sampler.py:38-59 writes it
    x = sx * 2 - 1
    y = sy * 2 - 1
    if x > -y:
        if x > y:
            r, phi = x, y / x
        else:
            r, phi = y, 2 - x / y
    else:
        if x < y:
            r, phi = -x, 4 + y / x
        else:
            r = -y
            phi = 6 - x / y if y != 0 else 0
    phi *= numpy.pi / 4.0
    return numpy.array([r * numpy.cos(phi), r * numpy.sin(phi)])


def _hemisphere(x, y, e):
    # cosine-power hemisphere map, sampler.py:66-75
    cos_phi = numpy.cos(2 * numpy.pi * x)
    sin_phi = numpy.sin(2 * numpy.pi * x)
    cos_theta = (1.0 - y) ** (1.0 / (e + 1.0))
    sin_theta = (1.0 - cos_theta * cos_theta) ** 0.5
    return numpy.array([sin_theta * cos_phi, sin_theta * sin_phi, cos_theta])


class MultiJitteredSampler(object):
    """`pattern_size` multi-jittered patterns of sample_dim**2 points each, plus
    their disk and hemisphere images (sampler.py:18-76)."""

    def __init__(self, sample_dim=2, pattern_size=83, e=1):
        self.sample_dim = sample_dim
        self.pattern_size = pattern_size
        self.e = e
        dim = float(sample_dim)
        rng.context.point(rng.PHASE_TABLE)
        self.patterns = []
        for _ in range(pattern_size):
            rows = list(range(sample_dim))
            cols = list(range(sample_dim))
            rng.shuffle(cols)
            rng.shuffle(rows)
            pts = []
            for i in range(sample_dim):
                for j in range(sample_dim):
                    px = (i + rng.uniform(0, 1)) / dim ** 2 + rows[i] / dim
                    py = (j + rng.uniform(0, 1)) / dim ** 2 + cols[j] / dim
                    pts.append((px, py))
            self.patterns.append(pts)
        self.concentric_patterns = [[_concentric(px, py) for px, py in pts]
                                    for pts in self.patterns]
        # one flat list over all patterns, as sampler.py:63-76 builds it
        self.hemisphere_patterns = [_hemisphere(px, py, e)
                                    for pts in self.patterns for px, py in pts]

    # Host-side draws, for callers that sample outside a device render.
    def sample(self):
        return rng.choice(self.patterns)

    def sample_unit_disk(self):
        return rng.choice(self.concentric_patterns)

    def sample_unit_hemisphere_surface(self):
        return rng.choice(self.hemisphere_patterns)
