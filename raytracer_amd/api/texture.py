"""Image textures (/root/reference/texture.py) are outside the device hot path: the only
scene using them needs an image asset the reference does not ship."""


class ImageTexture(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("textured materials are outside the device hot path")


class SphericalMap(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("textured materials are outside the device hot path")
