"""The smooth-shaded teapot (3 644 vertices, 6 320 faces) in a kd-tree, three worlds:

'k'         /root/reference/buildfunctions/k.py:18-55: matte teapot on a matte ground.
'mirror'    /root/reference/buildfunctions/mirror.py:18-55: the same on a mirror ground, 500x500.
'mirroralt' /root/reference/buildfunctions/mirroralt.py:18-69: a mirror teapot among three
            spheres, seen from behind, a back wall at z = 8.

All three: one sample per pixel, Tracer, constant ambient light, one shadow-casting point
light with inverse-square falloff over a radius of 10."""
import numpy
from buildfunctionbase import BuildFunctionBase
from camera import PinholeCamera
from geometry import Plane, Sphere
from kdtree import BoundingBoxes, KDTree
from light import AmbientLight, PointLight
from material import Matte, Phong, Reflective
from mesh_parser import read_mesh
from sampler import MultiJitteredSampler, RegularSampler
from tracer import Tracer, ViewPlane

WHITE = (1., 1., 1.)
GOLD = (1., .84, .1)


def teapot_tree(material, announce=True):
    with open('meshes/teapot.obj') as f:
        mesh = read_mesh(f)
    mesh.compute_smooth_normal()
    mesh.material = material
    tree = KDTree(BoundingBoxes(mesh.get_bounding_boxes()))
    if announce:
        tree.print_tree()
    return tree


def _stage(world, viewmode, resolution, pixel_size, eye_z):
    world.viewmode = viewmode
    preview = viewmode == "realtime"
    world.viewplane = ViewPlane(resolution=(64, 64) if preview else resolution,
                                pixel_size=5 if preview else pixel_size,
                                sampler=RegularSampler() if preview else MultiJitteredSampler(sample_dim=1))
    world.camera = PinholeCamera(eye=(0., 2., eye_z), up=(0., 1., 0.), lookat=(0., 1.5, 0.), viewing_distance=300.)
    world.background_color = (0.0, 0.0, 0.0)
    world.tracer = Tracer(world)
    world.ambient_color = AmbientLight(numpy.array(WHITE), .2)
    world.lights = [PointLight(numpy.array(WHITE), 1., numpy.array((1., 8., 2.)), radius=10, attenuation=2,
                               cast_shadow=True)]


def _ground(material):
    return Plane(origin=(0, 0, 0), normal=(0, 1, 0), material=material)


class TeapotMatte(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'k'

    @classmethod
    def build_function(cls, world, viewmode):
        _stage(world, viewmode, (200, 200), 1.6, 7.)
        gold = Matte(ka=1, kd=1, cd=numpy.array(GOLD))
        white = Matte(ka=1, kd=1, cd=numpy.array(WHITE))
        world.objects = [teapot_tree(gold), _ground(white)]


class TeapotOnMirror(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'mirror'

    @classmethod
    def build_function(cls, world, viewmode):
        _stage(world, viewmode, (500, 500), .64, 7.)
        gold = Matte(ka=1, kd=1, cd=numpy.array(GOLD))
        glass = Reflective(ka=0.6, kd=0.6, ks=0.6, kr=1.0, exp=100, cd=numpy.array(WHITE))
        world.objects = [teapot_tree(gold), _ground(glass)]


class MirrorTeapot(BuildFunctionBase):
    BUILD_FUNCTION_NAME = 'mirroralt'

    @classmethod
    def build_function(cls, world, viewmode):
        _stage(world, viewmode, (200, 200), .64 * 2.5, -7.)
        shiny_gold = Phong(ka=1, kd=1, ks=1, exp=100, cd=numpy.array(GOLD))
        pink = Matte(ka=1, kd=1, cd=numpy.array([1., .84, 1.]))
        blue = Matte(ka=1, kd=1, cd=numpy.array([.2, .3, 1.]))
        chrome = Reflective(ka=.6, kd=.6, ks=.6, kr=1.0, exp=500, cd=numpy.array(WHITE))
        world.objects = [
            teapot_tree(chrome, announce=False),
            Sphere(center=numpy.array((-2.5, 0.5, -1.5)), radius=1., material=shiny_gold),
            Sphere(center=numpy.array((3.5, 1.5, 2.5)), radius=2., material=pink),
            Sphere(center=numpy.array((2.5, 0.5, -1.5)), radius=.7, material=shiny_gold),
            _ground(blue),
            Plane(origin=(0, 0, 8), normal=(0, 0, -1), material=pink),
        ]
