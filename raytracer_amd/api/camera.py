"""Cameras: same constructors and render entry points as /root/reference/camera.py
(`PinholeCamera` :12-139, `ThinLensCamera` :142-267); every entry point that traces
rays does so on the device.

* `render_pixels(world, pairs)`            camera.py:60-67 / :227-234  (worker seam)
* `get_pixel_plane_point_pairs(world, n)`  camera.py:69-98 / :236-267  (job producer)
* `render(world, pixel_func)`              camera.py:35-58 / :148-176  (all samples of a pixel)
* `render_progressive(world, pixel_func)`  camera.py:100-139 / :178-220 (one sample per pass)

Randomness: the pattern a pixel uses, the disk pattern of a thin-lens pixel and the
order of the sample passes come from `raytracer_amd.rng` keyed by pixel index, so a
frame is reproducible and independent of traversal order.
"""
import numpy

from tracer import *  # noqa: F401,F403  (the reference re-exports its tracer names here)
from raytracer_amd import rng


def _unit(v):
    return v / numpy.linalg.norm(v)


class Camera(object):
    def break_if_needed(self, world, row, column):
        """`--breakon x,y` drops into a debugger in the reference (camera.py:6-9); a device
        render has no per-pixel host frame to stop in, so the request is ignored."""
        return None


class PinholeCamera(Camera):
    lens_radius = 0.0
    focal_plane_distance = 0.0
    IS_THIN_LENS = False

    def __init__(self, eye, lookat, up, viewing_distance):
        self.eye = numpy.array(eye)
        self.lookat = numpy.array(lookat)
        self.up = numpy.array(up)
        self.viewing_distance = viewing_distance
        self.compute_uwv()

    def compute_uwv(self):
        # w points from the target to the eye; v = w x u (camera.py:20-29)
        self.w = _unit(self.eye - self.lookat)
        self.u = _unit(numpy.cross(self.w, self.up))
        self.v = numpy.cross(self.w, self.u)

    def ray_direction(self, point):
        return _unit(point[0] * self.u + point[1] * self.v - self.viewing_distance * self.w)

    # -- device-backed entry points ---------------------------------------------------
    def render_pixels(self, world, pixel_plane_point_pairs):
        from raytracer_amd import runtime
        return runtime.render_pixels(world, self, pixel_plane_point_pairs)

    def get_pixel_plane_point_pairs(self, world, chunk_size):
        from raytracer_amd import runtime
        return runtime.pixel_plane_point_pairs(world, self, chunk_size)

    def render(self, world, pixel_func):
        from raytracer_amd import runtime
        return runtime.render_frame(world, self, pixel_func, progressive=False)

    def render_progressive(self, world, pixel_func):
        from raytracer_amd import runtime
        return runtime.render_frame(world, self, pixel_func, progressive=True)


class ThinLensCamera(PinholeCamera):
    IS_THIN_LENS = True

    def __init__(self, lens_radius, focal_plane_distance, *args, **kwargs):
        self.lens_radius = lens_radius
        self.focal_plane_distance = focal_plane_distance
        super(ThinLensCamera, self).__init__(*args, **kwargs)

    def lens_point(self, disk_sample):
        """The lens offset is applied along world x and y, not along u and v (camera.py:169)."""
        return self.eye + numpy.array([disk_sample[0], disk_sample[1], 0.0]) * self.lens_radius

    def ray_direction(self, plane_point, lens_point):
        # camera.py:222-225; note that the world-space lens coordinates enter the u/v combination
        f = plane_point * self.focal_plane_distance / self.viewing_distance
        return _unit((f[0] - lens_point[0]) * self.u + (f[1] - lens_point[1]) * self.v
                     - self.focal_plane_distance * self.w)
