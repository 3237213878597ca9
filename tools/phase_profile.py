#!/usr/bin/env python3
"""Where the waves of the tile kernel spend their cycles.

Builds raytracer_amd/csrc/librtamd_phases.so (the product sources with RT_PHASE_TIMERS:
every wave adds its shader-clock cycles per part of a kd query to a table), renders frames
of a bench workload with it and prints the split.  Measurement tool, not part of the
product; the timed library is the normal build.

  python tools/phase_profile.py --build          (in the build container)
  python tools/phase_profile.py [--workload W]   (on the GPU box)
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
LIB = os.path.join(REPO, "raytracer_amd", "csrc", "librtamd_phases.so")
NAMES = ["descent", "pack16", "pack32", "big_plain", "big_boxes", "pop", "queries", "handover"]


def build():
    import __graft_entry__ as entry
    src = os.path.join(REPO, "raytracer_amd", "csrc", "rt_kernels.hip")
    kd = os.path.join(REPO, "raytracer_amd", "csrc", "rt_kdbuild.hip")
    st = os.path.join(REPO, "raytracer_amd", "csrc", "rt_shadowtree.cpp")
    cmd = ["/opt/rocm/bin/hipcc"] + entry.HIP_FLAGS + ["-DRT_PHASE_TIMERS", "-DRT_ONLY_TILES", "-I", os.path.join(REPO, "include"), src, kd, st, "-o", LIB]
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--workload", default="dragon_1080p_16spp_ao_arealight")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--config", type=int, default=-1, help="index into tools/config_rates.py CONFIGS instead of a bench workload")
    args = ap.parse_args()
    if args.build:
        build()
        return
    os.environ["RTAMD_LIB"] = LIB
    import torch
    import bench
    from raytracer_amd import capi, distributed, flatten, runtime
    from raytracer_amd.world import World
    if args.config >= 0:
        # a configuration of tools/config_rates.py (the reference's scene files at full HD among them)
        import config_rates
        entry = config_rates.CONFIGS[args.config]
        args.workload, scene, overrides = entry[:3]
    else:
        scene, overrides = bench.WORKLOADS[args.workload]
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=1)
    if args.config >= 0 and len(entry) > 3:
        import sampler
        width, height, dim = entry[3]
        world.viewplane.pixel_size *= world.viewplane.resolution[0] / float(width)
        world.viewplane.resolution = (width, height)
        world.viewplane.sampler = sampler.MultiJitteredSampler(dim)
        world.invalidate()
    job = distributed.FrameJob(world, 0, 1)
    lib = capi.library()
    table = (C.c_ulonglong * 128)()
    job.render()
    torch.cuda.synchronize()
    lib.rtamd_debug_phase_cycles(table, 1)
    for _ in range(args.frames):
        job.render()
    torch.cuda.synchronize()
    lib.rtamd_debug_phase_cycles(table, 1)
    t = [int(x) for x in table]
    total, waves = t[16], t[17]
    out = {"workload": args.workload, "frames": args.frames, "waves": waves, "cycles_per_wave": total / max(waves, 1)}
    for base, kind in ((0, "closest"), (8, "shadow_limited"), (36, "shadow_unlimited")):
        for i in (0, 1, 2, 3, 4, 5, 7):
            out["%s_%s" % (kind, NAMES[i])] = t[base + i] / float(total)
        out["%s_queries_per_wave" % kind] = t[base + 6] / float(max(waves, 1))
    out["outside_kd_queries"] = 1.0 - sum(t[0:6] + t[7:8] + t[8:14] + t[15:16] + t[36:42] + t[43:44]) / float(total)
    if t[24]:
        out["boxed_shadow_sharing"] = {"episodes": t[24], "rays": t[20], "distinct_leaves": t[21],
                                       "ray_chunk_pairs": t[22], "distinct_leaf_chunk_pairs": t[23]}
    if t[27]:
        out["packed_shadow_sharing"] = {"episodes": t[27], "rays": t[25], "distinct_leaves": t[26]}
    out["marks_inclusive"] = {"closest_hit_incl_kd": (t[31] + t[44] + t[45] + t[46]) / float(total), "ambient_occlusion_incl_kd": t[18] / float(total),
                              "light_shadow_incl_kd": t[19] / float(total), "ray_generation": t[32] / float(total),
                              "hit_to_ambient_ray": t[33] / float(total), "light_setup": (t[34] + sum(t[60:64])) / float(total),
                              "shade_and_write": t[35] / float(total)}
    if sum(t[60:64]):
        out["light_setup_parts"] = {"material_record_and_occlusion_ray": t[60] / float(total), "light_sample_and_direction": t[61] / float(total),
                                    "light_term": t[62] / float(total), "rays_in_tree_space_and_root_boxes": t[63] / float(total),
                                    "rest": t[34] / float(total)}
    out["closest_hit_parts"] = {"before_kd_query": t[44] / float(total), "kd_query_with_root_box": t[45] / float(total),
                                "hit_row_normal_and_material": t[46] / float(total),
                                "analytic_objects_and_the_rest": t[31] / float(total)}
    if t[28]:
        out["shadow_rounds"] = {"rounds": t[28], "lanes_walking_per_round": t[29] / float(t[28])}
    if t[50]:
        # walks of the shadow tree: those that start at the root (a lane's own rays, second rays taken over) and
        # those that start from a stack entry handed over; "fresh" lane-rounds are spent before the first leaf
        out["shadow_walks"] = {"root_starts": t[50], "root_starts_with_origin_inside_the_box": t[51], "handed_over_starts": t[58],
                               "lane_rounds": t[29], "lane_rounds_before_first_leaf_root_starts": t[48],
                               "lane_rounds_before_first_leaf_handed_over": t[49],
                               "levels_to_first_leaf_root_starts": t[55] / float(max(t[50], 1)),
                               "entries_pushed_to_first_leaf_root_starts": t[56] / float(max(t[50], 1)),
                               "levels_to_first_leaf_handed_over": t[59] / float(max(t[58], 1)),
                               "packed16_episodes": t[54], "packed16_rays": t[52], "packed16_distinct_leaves": t[53],
                               "packed16_rays_in_the_fullest_leaf": t[57] / float(max(t[54], 1))}
    h = t[64:128]
    if sum(h[0:16]):
        out["shadow_queries_by_rays"] = [
            {"rays": "%d-%d" % (8 * b, 8 * b + 7) if b < 15 else "120-128", "queries": h[b], "rounds_per_query": h[16 + b] / float(h[b]),
             "cycles_per_query": h[32 + b] / float(h[b]), "lanes_walking_per_round": h[48 + b] / float(max(h[16 + b], 1)),
             "share_of_query_cycles": h[32 + b] / float(sum(h[32:48]))} for b in range(16) if h[b]]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
