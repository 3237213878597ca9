#!/usr/bin/env python3
"""Host against device kd-tree build of the dragon (100 000 boxes): seconds each, same tree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd
raytracer_amd.activate()
import torch
import kdtree
from mesh_parser import read_mesh
with open(os.path.join(raytracer_amd.ASSET_DIR, sys.argv[1] if len(sys.argv) > 1 else "meshes/dragon.obj")) as f:
    mesh = read_mesh(f)
bb = kdtree.BoundingBoxes(mesh.get_bounding_boxes())
torch.cuda.init(); torch.zeros(1, device="cuda")
for mode in ("device", "device", "host"):
    kdtree.KD_BUILD = mode
    t0 = time.perf_counter()
    tree = kdtree.KDTree(bb)
    dt = time.perf_counter() - t0
    print("%-6s %.3f s  %s  %s" % (mode, dt, tree.stats(), tree.signature()[:16]), flush=True)
