#!/usr/bin/env python3
"""One-off sweep over more worlds of the random families than the test suite renders:
  python tests/fuzz_sweep.py FIRST LAST [MESH_FIRST MESH_LAST [PATH_FIRST PATH_LAST]]
  python tests/fuzz_sweep.py --worlds fuzz_232,fuzzmesh_42,fuzzpath_7,...
Every world goes through tests/test_gpu_parity.check_tiles_against_oracle (whole frames, every sample, ids /
ray counts / reference work exact, the timed and both counting kernels the same frame).  Needs a GPU."""
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    import __graft_entry__  # noqa: F401  (puts the package on the path)
    from raytracer_amd.world import World
    from raytracer_amd import flatten
    World(viewmode="parallel", buildfunction="fuzz_0", seed=1)       # loads the scene files
    import test_gpu_parity as t
    fuzz = [m for name, m in sys.modules.items() if name.startswith("rtamd_scene_fuzz")][0]
    families = {"fuzz": (fuzz._make, fuzz.N_WORLDS), "fuzzmesh": (fuzz._make_mesh, fuzz.N_MESH_WORLDS),
                "fuzzpath": (fuzz._make_path, fuzz.N_PATH_WORLDS)}
    if sys.argv[1] == "--worlds":
        jobs = [(n.split("_")[0] + "_%d", families[n.split("_")[0]][0], int(n.split("_")[1])) for n in sys.argv[2].split(",")]
    else:
        a = [int(x) for x in sys.argv[1:]]
        jobs = [("fuzz_%d", fuzz._make, k) for k in range(a[0], a[1])]
        if len(a) > 2:
            jobs += [("fuzzmesh_%d", fuzz._make_mesh, k) for k in range(a[2], a[3])]
        if len(a) > 4:
            jobs += [("fuzzpath_%d", fuzz._make_path, k) for k in range(a[4], a[5])]
    bad = 0
    for done, (pattern, make, k) in enumerate(jobs):
        name = pattern % k
        if done % 50 == 0:
            print("... %d of %d worlds, %d failed so far" % (done, len(jobs), bad), flush=True)
        if k >= families[pattern.split("_")[0]][1]:
            keep = make(k)      # (held: build functions are found through their base class, which does not hold them)
        try:
            world = World(viewmode="parallel", buildfunction=name, seed=1)
            t.check_tiles_against_oracle(torch, world, flatten.flatten_world(world), 96, None, name)
        except Exception:
            bad += 1
            print("FAILED %s" % name, flush=True)
            traceback.print_exc()
    print("%d worlds, %d failed" % (len(jobs), bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
