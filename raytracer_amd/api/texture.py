"""Image textures and their mappings: descriptions the flattener turns into a texel
table in HBM.  Interface of /root/reference/texture.py (`ImageTexture` :11-46,
`SphericalMap` :53-68); the lookup itself runs in the kernels (csrc/rt_device.h).
"""
import numpy

MAP_SPHERICAL = 0


class Texture(object):
    def __init__(self):
        pass


class Mapping(object):
    pass


class SphericalMap(Mapping):
    """(theta, phi) of the local hit point -> (column, row); texture.py:54-68."""
    KIND = MAP_SPHERICAL


class ImageTexture(Texture):
    """`image` is a file name or a PIL image, as in the reference (texture.py:24-25); an
    (height, width, 3) uint8 array is accepted as well.  `color` is kept and, as in the
    reference, never used."""

    def __init__(self, image, color, mapping, hres=None, vres=None):
        self.image = image
        self.color = color
        self.mapping = mapping
        self.hres = hres
        self.vres = vres
        self.instantiated = False

    def instantiate_for_proc(self):
        image = self.image
        if isinstance(image, str):
            from PIL import Image
            image = Image.open(image)
        if self.hres is None:
            if isinstance(image, numpy.ndarray):
                self.vres, self.hres = image.shape[:2]
            else:
                self.hres, self.vres = image.size
        self.instantiated = True
        self.image = image

    def device_record(self):
        if not self.instantiated:
            self.instantiate_for_proc()
        image = self.image
        if isinstance(image, numpy.ndarray):
            pixels = image
        else:
            # getpixel (texture.py:45) yields an RGB triple for mode "RGB" and one number,
            # broadcast over the channels, for mode "L"
            if image.mode not in ("RGB", "L"):
                raise NotImplementedError("image mode %r: only RGB and L images are handled" % image.mode)
            pixels = numpy.asarray(image)
        if pixels.ndim == 2:
            pixels = numpy.repeat(pixels[:, :, None], 3, axis=2)
        if pixels.ndim != 3 or pixels.shape[2] != 3 or pixels.dtype != numpy.uint8:
            raise ValueError("a texture image must be 8-bit RGB or grey")
        if self.mapping is None:
            # without a mapping the reference reads sr.u / sr.v, which no primitive sets (texture.py:42-43)
            raise AttributeError("ImageTexture without a mapping: hit records carry no (u, v)")
        hres, vres = int(self.hres), int(self.vres)
        if not (0 < hres <= pixels.shape[1] and 0 < vres <= pixels.shape[0]):
            raise IndexError("hres/vres (%d, %d) exceed the image size %r" % (hres, vres, pixels.shape[:2]))
        return dict(pixels=pixels, hres=hres, vres=vres, mapping=self.mapping.KIND)
