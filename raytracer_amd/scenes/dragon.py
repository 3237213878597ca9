"""The 100 000-triangle dragon in a kd-tree.

'j' is the world of /root/reference/buildfunctions/j.py:18-53: 400x400, one sample,
matte dragon on a ground plane, one shadow-casting point light.
'dragon_area' is benchmark configuration 4: 1920x1080, 16 samples per pixel, ambient occlusion
plus a rectangular area light (soft shadows), AreaLightTracer."""
import numpy
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import Plane, Rectangle
from kdtree import BoundingBoxes, KDTree
from light import AmbientLight, AmbientOccluder, AreaLight, PointLight
from material import Emissive, Matte
from mesh_parser import read_mesh
from sampler import MultiJitteredSampler, RegularSampler
from tracer import AreaLightTracer, Tracer, ViewPlane

_tree_cache = {}


def dragon_tree(material, path='meshes/dragon.obj', face_limit=None):
    with open(path) as f:
        mesh = read_mesh(f, face_limit)
    mesh.material = material
    tree = KDTree(BoundingBoxes(mesh.get_bounding_boxes()))
    tree.print_tree()
    return tree


def _common(world, viewmode):
    world.viewmode = viewmode
    world.camera = PinholeCamera(eye=(0., 2., 7.), up=(0., 1., 0.), lookat=(0., 1.5, 0.),
                                 viewing_distance=100.)
    world.background_color = (0.0, 0.0, 0.0)
    gold = Matte(ka=1, kd=1, cd=numpy.array([1., .84, .1]))
    white = Matte(ka=1, kd=1, cd=numpy.array([1., 1., 1.]))
    opts = getattr(world, "overrides", None) or {}
    world.objects = [dragon_tree(gold, face_limit=opts.get("face_limit")),
                     Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=white)]


class DragonPointLight(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'j'

    @classmethod
    def build_function(cls, world, viewmode):
        preview = viewmode == "realtime"
        world.viewplane = ViewPlane(resolution=(64, 64) if preview else (400, 400),
                                    pixel_size=5 if preview else 0.8,
                                    sampler=RegularSampler() if preview else MultiJitteredSampler(sample_dim=1))
        world.tracer = Tracer(world)
        world.ambient_color = AmbientLight(numpy.array((1., 1., 1.)), .2)
        _common(world, viewmode)
        world.lights = [PointLight(numpy.array((1., 1., 1.)), 1., numpy.array((1., 8., 2.)), radius=10,
                                   attenuation=2, cast_shadow=True)]


class DragonAreaLight(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'dragon_area'

    @classmethod
    def build_function(cls, world, viewmode):
        opts = getattr(world, "overrides", None) or {}
        width, height = opts.get("resolution", (64, 36) if viewmode == "realtime" else (1920, 1080))
        dim = opts.get("sample_dim", 1 if viewmode == "realtime" else 4)
        sampler = MultiJitteredSampler(sample_dim=dim)
        world.viewplane = ViewPlane(resolution=(width, height), pixel_size=320. / height, sampler=sampler)
        world.tracer = AreaLightTracer(world)
        world.ambient_color = AmbientOccluder(numpy.array((1., 1., 1.)), .35, sampler)
        _common(world, viewmode)
        # a 6x4 emitter above and in front of the dragon, facing back towards it (-z);
        # Rectangle.sample offsets its corner along world x and y (geometry.py:378-380)
        panel = Rectangle(numpy.array((-5., 7., 9.)), 6, 4, numpy.array((0., 0., -1.)),
                          numpy.array((1., 0., 0.)), Emissive(30., numpy.array((1., .96, .9))), sampler)
        world.lights = [AreaLight(panel.material, panel, 6.)]
