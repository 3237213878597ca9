#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the dragon kd-tree scene (1920x1080, 16 samples per
pixel, ambient occlusion + area-light soft shadows), whole job over N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full frame: every rank traces its share of the 8x8-pixel tiles (all 16
samples of each pixel) and the tile-compact accumulators are gathered on rank 0 (RCCL).
The frame is fixed, so scaling is "strong".  A ray is one closest-hit query or one
shadow/occlusion query, counted in the kernel.  Rank 0 prints ONE JSON line.

Extra fields:
  roofline      where the tile kernel stands against three ceilings, each a fraction <= 1:
                  hbm   bytes the L2s exchanged with memory per launch (rocprofv3 FETCH_SIZE /
                        WRITE_SIZE, collected by this run from a profiled child process
                        before the timed part, or - if that is not possible - taken from
                        profiles/ when the record there was made with this library build)
                        / kernel time / 8 TB/s: the top-level achieved / frac / traffic;
                  l2    bytes the kernel's own loads ask for (node records, culling boxes,
                        triangle rows, analytic objects, 48 B per ray for ray + hit record;
                        counted by an RTAMD_COUNT_EXECUTED launch of the same frame)
                        / kernel time / 34.5 TB/s;
                  valu  share of SIMD cycles with a vector-ALU instruction executing, and
                        the active lanes per instruction (SQ counters of the same child);
                `binding` names the largest.  `reference_work` is what the reference's
                algorithm would have moved for the same rays (per ray 48 B + 16 B per kd
                step + 72 B per triangle test + 48 B per analytic test, counted by an
                RTAMD_COUNT_WORK launch) and the ratio of its triangle tests to the tests
                the kernel ran (the culling ratio) - not a bandwidth the kernel reaches.
  cpu_baseline  the CPU checker (oracle/, a float64 port of the same path) timed on this
                host over a bounded sample of the same workload, OpenMP threads stated.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
KERNEL_SOURCES = ("raytracer_amd/csrc/rt_kernels.hip", "raytracer_amd/csrc/rt_device.h", "include/raytracer_amd.h")

WORKLOADS = {
    # name -> (scene, overrides)
    "dragon_1080p_16spp_ao_arealight": ("dragon_area", dict(resolution=(1920, 1080), sample_dim=4)),
    "dragon_480x270_16spp_ao_arealight": ("dragon_area", dict(resolution=(480, 270), sample_dim=4)),
    "cornellbox_200_16spp_arealight": ("cornellbox_area", None),
    # the other benchmark configurations, for profiling tools (bench lines are quoted on the dragon frame)
    "foursphere_200_1spp": ("foursphere", None),
    "glossyreflective_400_100spp": ("glossyreflective", None),
    "mirrorhall_4k_64spp_dof": ("mirrorhall_dof", dict(resolution=(3840, 2160), sample_dim=8)),
    "cornellbox_200_100spp_path": ("cornellbox", None),       # the reference's scene file as shipped
}


def algorithmic_bytes(c):
    rays = c["closest"] + c["shadow"]
    return 48 * rays + 16 * c["kd_nodes"] + 72 * c["tri_tests"] + 48 * c["analytic_tests"]


def executed_bytes(c):
    """Bytes the kernel's loads asked for in an RTAMD_COUNT_EXECUTED launch, plus ray and hit record."""
    return 48 * (c["closest"] + c["shadow"]) + c["bytes_loaded"]


def kernel_build_id():
    """Hash of the kernel sources: ties a stored counter record to the library it was taken with."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profile_counters(args, log):
    """Hardware counters of the tile kernel from profiled child processes (this process has
    not touched the GPU yet), or the record under profiles/ made with the same kernel
    sources, or None.  Returns (counters, source)."""
    from raytracer_amd import profiling
    stored = os.path.join(REPO, "profiles", "pmc_%s.json" % args.workload)
    if not args.no_pmc:
        cmd = [sys.executable, os.path.abspath(__file__), "--pmc-probe", "--workload", args.workload,
               "--seed", str(args.seed), "--steps", "2", "--warmup", "1"]
        t0 = time.time()
        pmc = profiling.collect(cmd, log=log)
        if pmc is not None:
            pmc["build_id"] = kernel_build_id()
            if args.save_pmc:
                with open(stored, "w") as f:
                    json.dump(pmc, f, indent=1, sort_keys=True)
            return pmc, "rocprofv3 --pmc passes over a child process of this run (%.0f s)" % (time.time() - t0)
    if os.path.exists(stored):
        with open(stored) as f:
            pmc = json.load(f)
        if pmc.get("build_id") == kernel_build_id():
            return pmc, "profiles/pmc_%s.json (same kernel sources)" % args.workload
        log("profiles/pmc_%s.json was taken with other kernel sources: ignored" % args.workload)
    return None, None


def launch_ranks(n, argv):
    """Run this script as n ranks (one per GPU) under torch.distributed.run, as child
    processes of a parent that has not touched the GPU; returns the launcher's exit status.
    Replaces the worker fan-out of /root/reference/main.py:82-96."""
    import socket
    import subprocess
    with socket.socket() as s:                      # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dragon_1080p_16spp_ao_arealight", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes")
    ap.add_argument("--save-pmc", action="store_true", help="store the counters under profiles/")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)   # the profiled child
    args = ap.parse_args()

    def log(msg):
        print("[bench] " + msg, file=sys.stderr, flush=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process touches no GPU; it starts one fresh
        # process per GPU, lets rank 0's JSON line through and leaves with their status
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries the one JSON line and nothing else: whatever a library prints to file
    # descriptor 1 from here on (collective back ends announce their connections there) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    pmc, pmc_source = None, None
    if args.gpus == 1 and not args.pmc_probe and not os.environ.get("RTAMD_BENCH_LAUNCH_ONLY"):
        pmc, pmc_source = profile_counters(args, log)

    import numpy
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world_size, args.gpus))
    # RTAMD_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (which is also
    # what happens by itself there, see below)
    backend = os.environ.get("RTAMD_BENCH_BACKEND", "nccl")
    if os.environ.get("RTAMD_BENCH_LAUNCH_ONLY"):
        # launcher rehearsal without a GPU: rendezvous, one collective, rank 0 reports
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        seen = torch.tensor([rank + 1], dtype=torch.int64)
        if world_size > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
            dist.all_reduce(seen)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            json_out.write(json.dumps({"launch_only": True, "n_gpus": world_size, "rank_sum": int(seen[0])}) + "\n")
            json_out.flush()
        return
    # RCCL needs a GPU per rank; with fewer (a rehearsal on a smaller box) the ranks share the GPUs and gather
    # over gloo, and so they do when RCCL does not come up - the line says which (config.backend)
    n_dev = max(torch.cuda.device_count(), 1)
    backend_note = None
    if backend == "nccl" and world_size > n_dev and not os.environ.get("RTAMD_BENCH_FORCE_RCCL"):
        backend, backend_note = "gloo", "%d ranks on %d GPU(s): gloo instead of RCCL" % (world_size, n_dev)
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                        device_id=torch.device("cuda", local_rank))
                ones = torch.ones(1, dtype=torch.int64, device="cuda")
                dist.all_reduce(ones)                   # the communicator is built here at the latest
                torch.cuda.synchronize()
                if int(ones[0]) != world_size:
                    raise RuntimeError("RCCL all_reduce of ones over %d ranks gave %d" % (world_size, int(ones[0])))
            except Exception as exc:                    # every rank fails alike (driver, IPC or topology trouble)
                backend_note = "RCCL did not come up (%s): gloo instead" % (str(exc).strip().splitlines() or ["?"])[0][:160]
                log("rank %d: %s" % (rank, backend_note))
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
        if backend != "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world_size)

    from raytracer_amd.world import World
    from raytracer_amd import distributed, flatten, runtime

    scene, overrides = WORKLOADS[args.workload]
    world = World.__new__(World)
    if overrides:
        world.overrides = overrides
    t_build = time.time()
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=args.seed)
    ds = runtime.device_scene(world)
    t_build = time.time() - t_build
    cam = flatten.camera_record(world)
    job = distributed.FrameJob(world, rank, world_size)
    n_samples = job.n_samples

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # two accumulators in flight: the gather of frame i (int32 per channel on the wire) runs
    # on RCCL's stream while the kernel of frame i+1 renders into the other buffer
    bufs, recv = job.pipeline(2)
    wire_bytes = job.pipe_wire[0].numel() * job.pipe_wire[0].element_size()

    # Every frame's sums end in host memory on rank 0, as the parent of main.py:99-110 holds them: once a
    # slot's gather is done its parts are copied into pinned buffers on a side stream, under the kernel
    # of the next frame (gloo rehearsals gather on the host already).
    deliver = rank == 0 and backend == "nccl"
    host_parts, copy_stream, copied = [None, None], None, [None, None]
    if deliver:
        copy_stream = torch.cuda.Stream()
        host_parts = [[torch.empty(job.pipe_wire[0].shape, dtype=torch.int32).pin_memory() for _ in range(world_size)]
                      for _ in range(2)]

    def to_host(k, parts):
        """Rank 0: the gathered parts of slot k to pinned host memory, on the side stream."""
        if not deliver or parts is None:
            return
        done = torch.cuda.Event()
        done.record()                               # the current stream is ordered after the gather here
        copy_stream.wait_event(done)
        with torch.cuda.stream(copy_stream):
            for dst, src in zip(host_parts[k], parts):
                dst.copy_(src, non_blocking=True)
            copied[k] = torch.cuda.Event()
            copied[k].record()

    def run(steps, ev0=None, ev1=None, stall=None):
        pending = [None, None]
        for i in range(steps):
            k = i % 2
            if pending[k] is not None:
                t_wait = time.perf_counter()
                if stall is not None:
                    stall["gpu"][0][i].record()
                parts = pending[k].wait()           # the slot's previous gather has read its wire buffer
                if stall is not None:
                    stall["gpu"][1][i].record()
                    stall["host_s"] += time.perf_counter() - t_wait
                to_host(k, parts)
            # (the slot's sums were cleared by the pack pass of its previous frame)
            # events bracket the tile kernel on the stream it is launched on (torch's current stream)
            if ev0 is not None:
                ev0[i].record()
            job.render_fn(job.tile_ids, bufs[k], None)
            if ev1 is not None:
                ev1[i].record()
            if copied[k] is not None:
                torch.cuda.current_stream().wait_event(copied[k])   # the slot's wire / receive buffers are free again
                copied[k] = None
            pending[k] = distributed.gather_to_root(job.send(k, clear=True), parts=recv[k], async_op=True)
        parts = None
        for k in ((steps - 2) % 2, (steps - 1) % 2) if steps > 1 else (0,):
            if pending[k] is not None:
                parts = pending[k].wait()       # the last frame's gather is waited for last
                to_host(k, parts)
        return parts

    run(args.warmup)                            # same code path as the timed steps
    barrier()
    if args.pmc_probe:                          # the child that rocprofv3 watches: kernels only
        run(args.steps)
        barrier()
        return
    ds.read_counters(reset=True)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stall = {"host_s": 0.0, "gpu": ([torch.cuda.Event(enable_timing=True) for _ in range(args.steps)],
                                    [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)])}
    t0 = time.perf_counter()
    parts = run(args.steps, ev0, ev1, stall)
    barrier()
    elapsed = time.perf_counter() - t0
    counters = ds.read_counters(reset=True)
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps
    # how long this rank's stream stood still behind a gather before it could reuse a slot (per step)
    stall_gpu_ms = sum(a.elapsed_time(b) for a, b in list(zip(*stall["gpu"]))[2:]) / args.steps
    last_wire = job.pipe_wire[(args.steps - 1) % 2].clone()
    per_rank = torch.tensor([kernel_ms, stall_gpu_ms, stall["host_s"] * 1e3 / args.steps, elapsed * 1e3 / args.steps,
                             float(len(job.tiles))], dtype=torch.float64, device="cuda")
    rank_stats = [per_rank]
    if world_size > 1:
        rank_stats = [torch.zeros_like(per_rank) for _ in range(world_size)]
        dist.all_gather(rank_stats, per_rank)
    rank_stats = [[float(x) for x in r.cpu()] for r in rank_stats]

    # whole-job numbers: max time over ranks, sum of rays over ranks
    stats = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    rays = torch.tensor([counters["closest"] + counters["shadow"], counters["samples"]], dtype=torch.int64, device="cuda")
    if world_size > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed, kernel_ms = float(stats[0]), float(stats[1])
    total_rays, total_samples = int(rays[0]), int(rays[1])
    rays_per_frame = total_rays / float(args.steps)

    # the dominant (only) kernel on rank 0: what it executes and what the reference would
    # have done for the same rays, one counting launch each of this rank's share
    ds.read_counters(reset=True)
    probe = torch.zeros_like(job.sum_rgb8)
    ds.render_tiles(cam, world.seed, job.tile_ids, 0, n_samples, sum_rgb8=probe, count_executed=True)
    torch.cuda.synchronize()
    done = ds.read_counters(reset=True, executed=True)
    same_frame = bool(torch.equal(probe.clamp(-2147483648, 2147483647).to(torch.int32), last_wire))   # against the last timed frame
    probe.zero_()
    ds.render_tiles(cam, world.seed, job.tile_ids, 0, n_samples, sum_rgb8=probe, count_work=True)
    torch.cuda.synchronize()
    work = ds.read_counters(reset=True)
    algo_bytes = algorithmic_bytes(work)
    exec_bytes = executed_bytes(done)
    n_rays = float(max(done["closest"] + done["shadow"], 1))
    from raytracer_amd import profiling
    ceil = profiling.ceilings(pmc if world_size == 1 else None, kernel_ms, exec_bytes)
    hbm = ceil.get("hbm")

    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(world, ds, cam, n_samples, args.cpu_seconds)

    if world_size > 1:
        dist.barrier()
    if rank == 0:
        if deliver:
            parts = host_parts[(args.steps - 1) % 2]        # the last frame as it was delivered to the host
        frame = job.finish(parts)
        line = {
            "metric": "Mrays/s (whole node) on dragon.obj kd-tree scene" if scene == "dragon_area" else
                      "Mrays/s (whole node) on " + args.workload,
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene, "resolution": list(world.viewplane.resolution),
                       "samples_per_pixel": n_samples, "rays_per_frame": rays_per_frame,
                       "samples_per_frame": total_samples / float(args.steps),
                       "tiles": "8x8 px, diagonal interleave over ranks, gather to rank 0",
                       "backend": ("RCCL" if backend == "nccl" else backend) if world_size > 1 else None,
                       "backend_note": backend_note,
                       "scene_bytes": ds.flat.nbytes(), "host_scene_build_s": round(t_build, 2),
                       "frame_mean_rgb": [float(x) for x in frame.reshape(-1, 3).mean(axis=0)]},
            "roofline": {"bound": "hbm",
                         "achieved": hbm["achieved_GBs"] if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm["frac"] if hbm else None, "traffic": hbm["bytes_per_launch"] if hbm else None,
                         "traffic_source": pmc_source,
                         "kernel": (pmc or {}).get("kernel_name", "trace_kernel<tiles>"), "kernel_ms": kernel_ms,
                         "binding": ceil.get("binding"),
                         "ceilings": {k: v for k, v in ceil.items() if k != "binding"},
                         "registers": {k: pmc[k] for k in ("vgpr", "accum_vgpr", "sgpr", "scratch", "lds") if pmc and k in pmc},
                         "executed": {"bytes_per_launch": exec_bytes, "bytes_per_ray": exec_bytes / n_rays,
                                      "same_frame_as_timed_kernel": same_frame,
                                      "per_ray": {k: done[k] / n_rays for k in (
                                          "kd_nodes", "kd_leaves", "leaves_skipped", "node_records", "boxes_tested",
                                          "rows_loaded", "tri_tests", "analytic_tests")}},
                         "reference_work": {"bytes_per_launch": algo_bytes, "bytes_per_ray": algo_bytes / n_rays,
                                            "equivalent_GBs": algo_bytes / (kernel_ms * 1e-3) / 1e9,
                                            "culling_ratio": work["tri_tests"] / float(max(done["tri_tests"], 1)),
                                            "per_ray": {k: work[k] / n_rays for k in (
                                                "kd_nodes", "kd_leaves", "tri_tests", "analytic_tests")}},
                         "note": "hbm = counter-measured memory traffic of the launch / kernel time; l2 = bytes the "
                                 "kernel's loads ask for / kernel time (upper bound on L2 traffic); valu = SIMD cycles "
                                 "executing vector-ALU instructions; reference_work is the reference algorithm's "
                                 "traffic for the same rays, most of which the kernel culls - not a rate it reaches"},
            "cpu_baseline": cpu,
            # per rank: the tile kernel, how long the rank's stream and its host thread stood still behind a
            # gather, its own time per step, its share of the tiles; bytes it puts on the wire per frame
            "ranks": {"kernel_ms": [r[0] for r in rank_stats], "gather_stall_gpu_ms": [r[1] for r in rank_stats],
                      "gather_stall_host_ms": [r[2] for r in rank_stats], "ms_per_step": [r[3] for r in rank_stats],
                      "tiles": [int(r[4]) for r in rank_stats], "wire_bytes_per_rank_and_frame": wire_bytes,
                      "per_step_launches": "tile kernel, pack (saturate to int32 + clear), gather"},
            "timed_region": "K frames: tile kernel, pack, gather to rank 0, copy of the gathered sums to pinned host "
                            "memory on rank 0 (side stream)" + ("" if deliver else " [host copy skipped: gloo rehearsal]") +
                            "; excludes the un-permute and divide/clamp of main.py:99-106, done once on the host after timing",
        }
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if world_size > 1:
        dist.destroy_process_group()


def cpu_baseline(world, ds, cam, n_samples, target_seconds):
    """Time the CPU checker on randomly chosen whole tiles of the same frame."""
    import numpy
    from oracle import oracle
    from raytracer_amd import runtime
    # every CPU this process may run on; a box that hands the job a share of its CPUs says so through
    # OMP_NUM_THREADS (the GPU boxes give one GPU's job 16 of the machine's CPUs), and more threads than that
    # share would only queue up behind each other
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("RTAMD_CPU_THREADS") or os.environ.get("OMP_NUM_THREADS") or avail)
    threads = max(1, min(avail, share))
    osc = oracle.OracleScene(ds.flat)
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    order = numpy.random.RandomState(11).permutation(tx * ty).astype(numpy.int32)
    # calibrate on a few tiles, then size the sample for about target_seconds
    probe = order[:4 * threads]
    t0 = time.perf_counter()
    osc.render_tiles(cam, world.seed, probe, 0, n_samples, threads=threads)
    dt = time.perf_counter() - t0
    osc.read_counters(reset=True)
    n_tiles = int(max(len(probe), min(len(order) - len(probe), len(probe) * target_seconds / max(dt, 1e-6))))
    tiles = order[len(probe):len(probe) + n_tiles]
    t0 = time.perf_counter()
    osc.render_tiles(cam, world.seed, tiles, 0, n_samples, threads=threads)
    dt = time.perf_counter() - t0
    c = osc.read_counters(reset=True)
    rays = c["closest"] + c["shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(), "cpus_available_to_this_job": min(avail, share),
            "sample": "%d random 8x8 tiles x %d spp = %d samples, %d rays, %.1f s; samples handed to the threads in "
                      "chunks of 1000 (the job size of main.py:92)" % (len(tiles), n_samples, c["samples"], rays, dt)}


if __name__ == "__main__":
    main()
