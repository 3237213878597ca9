"""The Cornell box (CornellBox-Original.obj, 36 triangles in one kd-tree).

'cornellbox' is the world of /root/reference/buildfunctions/cornellbox.py:18-54: path
traced, 100 samples per pixel.
'cornellbox_area' is benchmark configuration 2: 16 samples per pixel, the ceiling quad as an
area light with soft shadows (AreaLightTracer); the quad lies in an xz-plane, which
`Rectangle.sample` cannot express (geometry.py:378-380), so the light's shape is the
small class below with the same three methods an AreaLight calls."""
import random

import numpy
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from kdtree import BoundingBoxes, KDTree
from light import AmbientLight, AreaLight, PointLight
from mesh_parser import read_mesh_complex
from sampler import MultiJitteredSampler, RegularSampler
from tracer import AreaLightTracer, PathTracer, ViewPlane


class QuadEmitter(object):
    """Parallelogram corner + sx*edge_u + sy*edge_v with a constant normal."""

    def __init__(self, corner, edge_u, edge_v, normal, sampler):
        self.corner = numpy.array(corner, dtype=float)
        self.edge_u = numpy.array(edge_u, dtype=float)
        self.edge_v = numpy.array(edge_v, dtype=float)
        self.normal = numpy.array(normal, dtype=float)
        self.sampler = sampler
        area = numpy.linalg.norm(numpy.cross(self.edge_u, self.edge_v))
        self.inv_area = 1. / area

    def sample(self):
        sx, sy = random.choice(self.sampler.sample())
        return (self.corner + sx * self.edge_u) + sy * self.edge_v

    def get_normal(self, sample_point):
        return self.normal

    def pdf(self, shader_rec):
        return self.inv_area

    def emitter_record(self):
        return dict(corner=self.corner, edge_u=self.edge_u, edge_v=self.edge_v, normal=self.normal,
                    pdf=self.inv_area, sampler=self.sampler)


def _box_tree(mat_sampler, emitter_casts_shadow=True):
    meshes = read_mesh_complex('CornellBox/CornellBox-Original.obj', mat_sampler=mat_sampler)
    boxes = []
    for name, mesh in meshes.items():
        if name == 'light':
            mesh.cast_shadow = emitter_casts_shadow
        mesh.compute_normal()
        boxes += mesh.get_bounding_boxes()
    tree = KDTree(BoundingBoxes(boxes))
    tree.print_tree()
    return tree, meshes


class CornellPath(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'cornellbox'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        preview = viewmode == "realtime"
        sampler = RegularSampler() if preview else MultiJitteredSampler(sample_dim=10)
        mat_sampler = MultiJitteredSampler(sample_dim=10)
        world.viewplane = ViewPlane(resolution=(64, 64) if preview else (200, 200),
                                    pixel_size=5 if preview else 1.6, sampler=sampler)
        world.camera = PinholeCamera(eye=(0., 1, 6.), up=(0., 1., 0.), lookat=(0., 1, 0.), viewing_distance=800.)
        world.background_color = (0.0, 0.0, 0.0)
        world.tracer = PathTracer(world)
        world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
        world.objects = [_box_tree(mat_sampler)[0]]
        world.lights = [PointLight(numpy.array((1., 1., 1.)), 1., numpy.array((0., 1., 4.)), radius=4,
                                   attenuation=2, cast_shadow=True)]


class CornellAreaLight(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'cornellbox_area'

    @classmethod
    def build_function(cls, world, viewmode):
        world.viewmode = viewmode
        opts = getattr(world, "overrides", None) or {}
        size = opts.get("resolution", (64, 64) if viewmode == "realtime" else (200, 200))
        dim = opts.get("sample_dim", 1 if viewmode == "realtime" else 4)
        sampler = MultiJitteredSampler(sample_dim=dim)
        world.viewplane = ViewPlane(resolution=size, pixel_size=320. / size[1], sampler=sampler)
        world.camera = PinholeCamera(eye=(0., 1, 6.), up=(0., 1., 0.), lookat=(0., 1, 0.), viewing_distance=800.)
        world.background_color = (0.0, 0.0, 0.0)
        world.tracer = AreaLightTracer(world)
        world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
        tree, meshes = _box_tree(sampler, emitter_casts_shadow=False)
        world.objects = [tree]
        # the light quad of CornellBox-Original.obj:160-167, facing down
        quad = QuadEmitter(corner=(-0.24, 1.98, -0.22), edge_u=(0.47, 0., 0.), edge_v=(0., 0., 0.38),
                           normal=(0., -1., 0.), sampler=sampler)
        world.lights = [AreaLight(meshes['light'].material, quad, 1.)]
