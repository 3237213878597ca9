"""MI355X-native hot path of the titangate/raytracer scene API.

`activate()` puts the scene-API modules (`sampler`, `tracer`, `material`, `light`,
`camera`, `geometry`, `kdtree`, `mesh_parser`, `affinetransform`, `buildfunctionbase`)
on `sys.path` under the top-level names scene files import them by
(/root/reference/buildfunctions/j.py:1-10).  `World` builds a scene with a
`build_function(world, viewmode)` plugin and renders it on the GPU.
"""
import os
import sys

__all__ = ["activate", "World", "API_DIR", "ASSET_DIR", "SCENE_DIR"]

HERE = os.path.dirname(os.path.abspath(__file__))
API_DIR = os.path.join(HERE, "api")
ASSET_DIR = os.path.join(HERE, "assets")
SCENE_DIR = os.path.join(HERE, "scenes")


def activate():
    """Make the scene-API modules importable by their top-level names (idempotent)."""
    if API_DIR not in sys.path:
        sys.path.insert(0, API_DIR)


def __getattr__(name):
    if name == "World":
        from .world import World
        return World
    raise AttributeError(name)
