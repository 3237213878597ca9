"""Compare GPU and CPU checker per sample on whole tiles and describe every difference."""
import sys, os, numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import common
from raytracer_amd import flatten, runtime
from oracle import oracle

scene, ntiles, s0, s1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
world = common.world_for(scene)
ds = runtime.device_scene(world)
cam = flatten.camera_record(world)
tx, ty = runtime.tile_grid(cam.width, cam.height)
rng = numpy.random.RandomState(5)
tile_ids = numpy.sort(rng.choice(tx * ty, size=min(ntiles, tx * ty), replace=False)).astype(numpy.int32)
n = len(tile_ids) * 64 * (s1 - s0)
fb = runtime.FrameBuffers(ds, tile_ids)
per = ds.sample_buffers(n)
ds.read_counters()
ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1, sum_rgb8=fb.sum_rgb8, per_sample=per, count_work=True)
torch.cuda.synchronize()
gw = ds.read_counters()
got = {k: v.cpu().numpy() for k, v in per[0].items()}
osc = oracle.OracleScene(common.flat_for(scene))
_, _, want = osc.render_tiles(cam, world.seed, tile_ids, s0, s1,
                              per_sample_want=("rgb", "hit_object", "hit_prim", "hit_t", "n_rays"), threads=16)
ow = osc.read_counters()
print("counters gpu", gw)
print("counters cpu", ow)
bad = numpy.nonzero((got["hit_prim"] != want["hit_prim"]) | (got["hit_object"] != want["hit_object"]))[0]
print("samples", n, "id mismatches", len(bad))
flat = common.flat_for(scene)
for k in bad[:20]:
    print(" k", k, "gpu obj/prim/t", got["hit_object"][k], got["hit_prim"][k], repr(got["hit_t"][k]),
          "cpu", want["hit_object"][k], want["hit_prim"][k], repr(want["hit_t"][k]))
rgbbad = numpy.nonzero(numpy.abs(got["rgb"] - want["rgb"]).max(axis=1) > 1e-9)[0]
print("rgb mismatches", len(rgbbad), "of which id-mismatched", len(set(rgbbad) & set(bad)))
for k in rgbbad[:10]:
    print(" k", k, got["rgb"][k], want["rgb"][k], got["n_rays"][k], want["n_rays"][k])
