"""`World`: builds a scene through a `build_function(world, viewmode)` plugin and renders
it on the GPU.  Constructor arguments follow /root/reference/main.py:13 (`viewmode`,
`buildfunction`, `fast`, `breakon`, `numcores`); the pygame window, keyboard camera and
process pool of main.py:61-207 are replaced by a headless `render()` that returns (and
optionally saves) the finished image.
"""
import contextlib
import glob
import importlib.util
import os
import sys

import numpy

import raytracer_amd
from . import rng

raytracer_amd.activate()

from buildfunctionbase import BuildFunctionBase  # noqa: E402

_loaded_scene_files = {}


def load_scene_file(path):
    """Import one scene file (a module defining a BuildFunctionBase subclass)."""
    path = os.path.abspath(path)
    if path in _loaded_scene_files:
        return _loaded_scene_files[path]
    name = "rtamd_scene_" + os.path.splitext(os.path.basename(path))[0] + "_%d" % len(_loaded_scene_files)
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[name] = module
    spec.loader.exec_module(module)
    _loaded_scene_files[path] = module
    return module


def load_scene_dir(directory):
    """Import every scene file of a directory, like buildfunctions/__init__.py:1-5."""
    for path in sorted(glob.glob(os.path.join(directory, "*.py"))):
        if os.path.basename(path) in ("__init__.py", "buildfunctionbase.py"):
            continue
        load_scene_file(path)


@contextlib.contextmanager
def _working_directory(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


class World(object):
    def __init__(self, viewmode="realtime", buildfunction="a", fast=False, breakon=None, numcores=3,
                 seed=1, scene_dirs=(), asset_dir=None):
        load_scene_dir(raytracer_amd.SCENE_DIR)
        for d in scene_dirs:
            load_scene_dir(d)
        fcn = BuildFunctionBase.get_build_function(buildfunction)
        if fcn is None:
            raise ValueError("no build function named %r (scene files live in %s)"
                             % (buildfunction, raytracer_amd.SCENE_DIR))
        self.seed = seed
        rng.context.reseed(seed)
        # scene files open assets by paths relative to the program directory
        # ('meshes/dragon.obj', 'CornellBox/...', j.py:41, cornellbox.py:41)
        with _working_directory(asset_dir or raytracer_amd.ASSET_DIR):
            fcn(self, viewmode)
        self.fast = fast
        self.numcores = numcores
        self.breakon = breakon

    # -- device ----------------------------------------------------------------------
    def invalidate(self):
        """Drop the device copy after editing the world (objects, lights, materials)."""
        ds = getattr(self, "_device_scene", None)
        if ds is not None:
            ds.close()
        self._device_scene = None

    def hit_objects(self, ray):
        """Closest hit along one ray (main.py:35-46), evaluated on the device.  Returns None
        on a miss, else a record with `tmin`, `hit_point`, `object_index`, `prim_index`."""
        from . import runtime
        ds = runtime.device_scene(self)
        o = numpy.asarray(ray.origin, dtype=numpy.float64).reshape(1, 3)
        d = numpy.asarray(ray.direction, dtype=numpy.float64).reshape(1, 3)
        out = ds.trace_rays(self.seed, o, d, want=("hit_object", "hit_prim", "hit_t"))
        if out["hit_object"][0] < 0:
            return None

        class _Rec(object):
            pass
        rec = _Rec()
        rec.tmin = float(out["hit_t"][0])
        rec.object_index = int(out["hit_object"][0])
        rec.prim_index = int(out["hit_prim"][0])
        rec.hit_point = o[0] + rec.tmin * d[0]
        rec.world = self
        return rec

    # camera moves of main.py:48-59
    def rotate_camera(self, roll, yaw, pitch):
        import affinetransform as transform
        cam = self.camera
        rotation = transform.rotate(cam.w, roll).dot(
            transform.rotate(cam.u, pitch).dot(transform.rotate(cam.v, yaw)))
        cam.w = transform.apply(rotation, cam.w)
        cam.u = transform.apply(rotation, cam.u)
        cam.v = transform.apply(rotation, cam.v)

    def move_camera(self, pan):
        self.camera.eye = self.camera.eye + self.camera.w * -pan

    # -- rendering -------------------------------------------------------------------
    def render(self, out_path=None, mode=None):
        """Render the frame headlessly and return it as uint8 (height, width, 3).

        mode None picks what main.py:174-184 would run for the world's viewmode:
        "realtime"/"offline" -> `camera.render` semantics, anything else -> the parallel
        accumulator's converged image."""
        from . import runtime
        if mode is None:
            mode = "render" if self.viewmode in ("realtime", "offline") else "parallel"
        image, info = runtime.render_image(self, mode=mode)
        self.last_render_info = info
        if out_path:
            save_png(out_path, image)
        return image


def save_png(path, image):
    """Minimal PNG writer (zlib + struct), so saving needs no imaging library."""
    import struct
    import zlib
    h, w, _ = image.shape
    raw = b"".join(b"\x00" + image[r].tobytes() for r in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
