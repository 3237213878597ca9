"""Material descriptions: what the device shading code is parameterised by.

Constructor signatures, attribute names and setters follow
/root/reference/material.py (`Matte` :232, `SV_Matte` :292, `Phong` :329, `Reflective` :380,
`GlossyReflective` :397, `Transparent` :414, `Dielectric` :440, `Emissive` :486 and the
BRDF / BTDF holders :31-229).  The shading
arithmetic itself lives in the kernels (csrc/rt_device.h); `device_record()` gives the
per-material constants the way the reference would evaluate them once per call:
`ka*cd` (Lambertian.rho :55-56), `kd*cd/pi` (Lambertian.f :38-39), `ks*cs`
(GlossySpecular.f :95) and `kr*cd` (PerfectSpecular.sample_f :74, GlossySpecular.sample_f :121).
"""
import numpy

MATTE, PHONG, REFLECTIVE, GLOSSY_REFLECTIVE, EMISSIVE, TRANSPARENT, DIELECTRIC, SV_MATTE = range(8)


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


class SV_Lambertian(Lambertian):
    """Lambertian whose `cd` is a texture object (material.py:17-28)."""


class GlossySpecular(BDRF):
    def __init__(self, sampler, ks, cs, exp):
        super(GlossySpecular, self).__init__(sampler)
        self.ks = ks
        self.cs = cs
        self.exp = exp


class BTDF(object):
    def __init__(self, sampler=None):
        self.sampler = sampler


class PerfectTransmitter(BTDF):
    def __init__(self, ior, kt, sampler=None):
        super(PerfectTransmitter, self).__init__(sampler)
        self.ior = ior
        self.kt = kt


class FresnelTransmitter(BTDF):
    def __init__(self, sampler, kd, cd, eta_in, eta_out):
        super(FresnelTransmitter, self).__init__(sampler)
        self.eta_in = eta_in
        self.eta_out = eta_out
        self.kd = kd
        self.cd = cd


class FresnelReflector(PerfectSpecular):
    def __init__(self, sampler, kd, cd, eta_in, eta_out):
        super(FresnelReflector, self).__init__(sampler, kd, cd)
        self.eta_in = eta_in
        self.eta_out = eta_out


def _py2_div(a, b):
    """`a / b` as Python 2 evaluates it: floor division when both are plain integers."""
    if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool) and not isinstance(b, bool):
        return a // b
    return a / b


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


class Transparent(Phong):
    """Phong plus a mirror child and a refracted child (material.py:414-437)."""
    KIND = TRANSPARENT

    def __init__(self, ka, kd, ks, cd, exp, sampler, ior, kt, kr):
        super(Transparent, self).__init__(ka, kd, ks, cd, exp, sampler)
        self.reflective_bdrf = PerfectSpecular(sampler, kr, cd)
        self.specular_btrf = PerfectTransmitter(ior, kt)

    def device_record(self):
        rec = super(Transparent, self).device_record()
        rec["reflect"] = _rgb(self.reflective_bdrf.kd * self.reflective_bdrf.cd)
        ior = self.specular_btrf.ior
        rec["eta_front"] = float(ior)
        rec["eta_back"] = float(1.0 / ior)          # material.py:140, :152
        rec["kt"] = float(self.specular_btrf.kt)
        return rec


class Dielectric(Phong):
    """Phong plus Fresnel-weighted reflected and refracted children whose radiance is
    filtered by cf^t over the child's path length (material.py:440-483)."""
    KIND = DIELECTRIC

    def __init__(self, ka, kd, ks, cd, exp, sampler, eta_in, eta_out, cf_in, cf_out):
        super(Dielectric, self).__init__(ka, kd, ks, cd, exp, sampler)
        self.fresnel_brdf = FresnelReflector(sampler, kd, cd, eta_in, eta_out)
        self.fresnel_btdf = FresnelTransmitter(sampler, kd, cd, eta_in, eta_out)
        self.cf_in = cf_in
        self.cf_out = cf_out

    def device_record(self):
        rec = super(Dielectric, self).device_record()
        fb = self.fresnel_brdf
        # the reflector and the transmitter are built from the same pair (:443-444)
        rec["eta_front"] = float(_py2_div(fb.eta_in, fb.eta_out))       # material.py:177, :190, :214
        rec["eta_back"] = float(_py2_div(fb.eta_out, fb.eta_in))        # material.py:175, :187, :212
        rec["filter_in"] = _rgb(self.cf_in)
        rec["filter_out"] = _rgb(self.cf_out)
        return rec


class SV_Matte(Material):
    """Matte with a texture for `cd` (material.py:292-326).  Its `shade` differs from
    Matte's in one more way: a light only contributes when the viewer is on the normal's
    side as well (`ndotwo > 0`, :319)."""
    KIND = SV_MATTE

    def __init__(self, ka, kd, cd, sampler=None):
        super(SV_Matte, self).__init__()
        self.ambient_brdf = SV_Lambertian(None, ka, cd)
        self.diffuse_brdf = SV_Lambertian(sampler, kd, cd)

    set_ka = Matte.set_ka
    set_kd = Matte.set_kd
    set_cd = Matte.set_cd

    def device_record(self):
        if self.ambient_brdf.cd is not self.diffuse_brdf.cd:
            raise NotImplementedError("SV_Matte with different ambient and diffuse textures")
        return dict(kind=self.KIND, ambient=_rgb(self.ambient_brdf.kd), diffuse=_rgb(self.diffuse_brdf.kd),
                    texture=self.diffuse_brdf.cd, receives_shadow=self.receives_shadow)
