"""Material descriptions: what the device shading code is parameterised by.

Constructor signatures, attribute names and setters follow
/root/reference/material.py (`Matte` :232, `Phong` :329, `Reflective` :380,
`GlossyReflective` :397, `Emissive` :486 and the BRDF holders :31-121).  The shading
arithmetic itself lives in the kernels (csrc/rt_device.h); `device_record()` gives the
per-material constants the way the reference would evaluate them once per call:
`ka*cd` (Lambertian.rho :55-56), `kd*cd/pi` (Lambertian.f :38-39), `ks*cs`
(GlossySpecular.f :95) and `kr*cd` (PerfectSpecular.sample_f :74, GlossySpecular.sample_f :121).
"""
import numpy

MATTE, PHONG, REFLECTIVE, GLOSSY_REFLECTIVE, EMISSIVE = range(5)


def _rgb(value):
    """An RGB triple from a scalar or a 3-vector (numpy broadcasting, as the reference gets for free)."""
    out = numpy.empty(3, dtype=numpy.float64)
    out[:] = value
    return out


class Material(object):
    def __init__(self):
        self.receives_shadow = True

    def get_Le(self, shader_rec=None):
        return numpy.array((0., 0., 0.))


class BDRF(object):
    def __init__(self, sampler=None):
        self.sampler = sampler


class Lambertian(BDRF):
    def __init__(self, sampler, kd, cd):
        super(Lambertian, self).__init__(sampler)
        self.kd = kd
        self.cd = cd


class PerfectSpecular(Lambertian):
    pass


class GlossySpecular(BDRF):
    def __init__(self, sampler, ks, cs, exp):
        super(GlossySpecular, self).__init__(sampler)
        self.ks = ks
        self.cs = cs
        self.exp = exp


class Matte(Material):
    KIND = MATTE

    def __init__(self, ka, kd, cd, sampler=None):
        super(Matte, self).__init__()
        self.ambient_brdf = Lambertian(None, ka, cd)
        self.diffuse_brdf = Lambertian(sampler, kd, cd)

    def set_ka(self, ka):
        self.ambient_brdf.kd = ka

    def set_kd(self, kd):
        self.diffuse_brdf.kd = kd

    def set_cd(self, c):
        self.ambient_brdf.cd = c
        self.diffuse_brdf.cd = c

    def device_record(self):
        return dict(kind=self.KIND,
                    ambient=_rgb(self.ambient_brdf.kd * self.ambient_brdf.cd),
                    diffuse=_rgb(self.diffuse_brdf.kd * self.diffuse_brdf.cd / numpy.pi),
                    # Lambertian.sample_f returns kd*cd/pi and pdf = n.wi/pi (:51-53)
                    sampler=self.diffuse_brdf.sampler,
                    receives_shadow=self.receives_shadow)


class Phong(Material):
    KIND = PHONG

    def __init__(self, ka, kd, ks, cd, exp, sampler=None):
        super(Phong, self).__init__()
        self.ambient_brdf = Lambertian(sampler, ka, cd)
        self.diffuse_brdf = Lambertian(sampler, kd, cd)
        self.specular_brdf = GlossySpecular(sampler, ks, cd, exp)

    set_ka = Matte.set_ka
    set_kd = Matte.set_kd
    set_cd = Matte.set_cd   # like the reference, leaves the specular colour alone (:342-344)

    def device_record(self):
        sp = self.specular_brdf
        return dict(kind=self.KIND,
                    ambient=_rgb(self.ambient_brdf.kd * self.ambient_brdf.cd),
                    diffuse=_rgb(self.diffuse_brdf.kd * self.diffuse_brdf.cd / numpy.pi),
                    specular=_rgb(sp.ks * sp.cs), exp=float(sp.exp),
                    receives_shadow=self.receives_shadow)


class Reflective(Phong):
    KIND = REFLECTIVE

    def __init__(self, ka, kd, ks, kr, cd, exp, sampler=None):
        super(Reflective, self).__init__(ka, kd, ks, cd, exp, sampler)
        self.reflective_bdrf = PerfectSpecular(None, kr, cd)

    def device_record(self):
        rec = super(Reflective, self).device_record()
        rec["reflect"] = _rgb(self.reflective_bdrf.kd * self.reflective_bdrf.cd)
        return rec


class GlossyReflective(Phong):
    KIND = GLOSSY_REFLECTIVE

    def __init__(self, ka, kd, ks, kr, cd, exp, sampler):
        super(GlossyReflective, self).__init__(ka, kd, ks, cd, exp, sampler)
        self.reflective_bdrf = GlossySpecular(sampler, kr, cd, exp)

    def device_record(self):
        rec = super(GlossyReflective, self).device_record()
        gl = self.reflective_bdrf
        rec["reflect"] = _rgb(gl.ks * gl.cs)
        rec["reflect_exp"] = float(gl.exp)
        rec["sampler"] = gl.sampler
        return rec


class Emissive(Material):
    KIND = EMISSIVE

    def __init__(self, ls, color):
        super(Emissive, self).__init__()
        self.ls = ls
        self.color = color

    def get_Le(self, shader_rec=None):
        return self.ls * self.color

    def device_record(self):
        return dict(kind=self.KIND, emit=_rgb(self.ls * numpy.asarray(self.color, dtype=numpy.float64)),
                    receives_shadow=self.receives_shadow)


class _HostOnly(Material):
    """Materials the device path does not cover (texture lookups, two-child recursion)."""

    def __init__(self, *args, **kwargs):
        super(_HostOnly, self).__init__()
        self.args, self.kwargs = args, kwargs

    def device_record(self):
        raise NotImplementedError("%s is outside the device hot path" % type(self).__name__)


class SV_Matte(_HostOnly):
    pass


class Transparent(_HostOnly):
    pass


class Dielectric(_HostOnly):
    pass
