"""Frame partition over ranks and the gather to rank 0, on CPU with the gloo backend
(world_size 2 and 3).  The per-rank renderer is injected: here it is the CPU checker, on
the GPU box the default is the HIP tile kernel; partition, padding, gather and assembly
are the code under test."""
import os
import socket

import numpy
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, scene, overrides, out_path):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from raytracer_amd.world import World
    from raytracer_amd import distributed, flatten
    from oracle import oracle
    world = World.__new__(World)
    world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=3)
    osc = oracle.OracleScene(flatten.flatten_world(world))
    cam = flatten.camera_record(world)
    n_samples = len(world.viewplane.sampler.patterns[0])

    def render_fn(tile_ids, sum_rgb8, sum_sq):
        a, b, _ = osc.render_tiles(cam, world.seed, tile_ids.numpy(), 0, n_samples, threads=2)
        if sum_rgb8 is not None:
            sum_rgb8[:len(a)] += torch.from_numpy(a)
        if sum_sq is not None:
            sum_sq[:len(b)] += torch.from_numpy(b)

    for mode in ("parallel", "render"):
        job = distributed.FrameJob(world, rank, world_size, render_fn=render_fn, mode=mode)
        parts = job.render()
        frame = job.finish(parts)
        if rank == 0:
            numpy.save(out_path + "." + mode + ".npy", frame)
        else:
            assert frame is None
    # frames in flight, as the benchmark runs them: render into alternating buffers with
    # the previous frame's gather still pending
    job = distributed.FrameJob(world, rank, world_size, render_fn=render_fn, mode="parallel")
    bufs, recv = job.pipeline(2)
    pending = [None, None]
    for i in range(3):
        k = i % 2
        if pending[k] is not None:
            pending[k].wait()
        # (a slot's sums start at zero and are cleared again by the pass that packs them for the wire)
        job.render_fn(job.tile_ids, bufs[k], None)
        pending[k] = distributed.gather_to_root(job.send(k, clear=True), parts=recv[k], async_op=True)
        assert int(bufs[k].abs().sum()) == 0
    pending[1].wait()
    frame = job.finish(pending[0].wait())
    if rank == 0:
        assert numpy.array_equal(frame, numpy.load(out_path + ".parallel.npy"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size", (2, 3))
def test_frame_over_ranks_equals_single_rank(tmp_path, world_size):
    import torch.multiprocessing as mp
    scene, overrides = "cornellbox_area", dict(resolution=(44, 30), sample_dim=2)   # partial edge tiles
    out = str(tmp_path / "frame")
    mp.spawn(_worker, args=(world_size, _free_port(), scene, overrides, out), nprocs=world_size, join=True)
    single = str(tmp_path / "single")
    _worker_single(scene, overrides, single)
    for mode in ("parallel", "render"):
        a = numpy.load(out + "." + mode + ".npy")
        b = numpy.load(single + "." + mode + ".npy")
        assert a.shape == (30, 44, 3) and numpy.array_equal(a, b)
        assert a.max() > 0


def _worker_single(scene, overrides, out_path):
    from raytracer_amd.world import World
    from raytracer_amd import distributed, flatten, runtime
    from oracle import oracle
    world = World.__new__(World)
    world.overrides = overrides
    World.__init__(world, viewmode="parallel", buildfunction=scene, seed=3)
    osc = oracle.OracleScene(flatten.flatten_world(world))
    cam = flatten.camera_record(world)
    n_samples = len(world.viewplane.sampler.patterns[0])
    tx, ty = runtime.tile_grid(cam.width, cam.height)
    tiles = numpy.arange(tx * ty, dtype=numpy.int32)
    a, b, _ = osc.render_tiles(cam, world.seed, tiles, 0, n_samples, threads=2)
    par = numpy.minimum(runtime.scatter_tiles(a, tiles, cam.width, cam.height) / float(n_samples), 255)
    ren = numpy.minimum(numpy.trunc(runtime.scatter_tiles(b, tiles, cam.width, cam.height) / float(n_samples) * 255), 255)
    numpy.save(out_path + ".parallel.npy", numpy.clip(par, 0, 255).astype(numpy.uint8))
    numpy.save(out_path + ".render.npy", numpy.clip(ren, 0, 255).astype(numpy.uint8))


def test_partition_covers_every_tile_once():
    from raytracer_amd import distributed, runtime
    for (w, h) in ((1920, 1080), (44, 30), (8, 8)):
        tx, ty = runtime.tile_grid(w, h)
        for world_size in (1, 2, 3, 4, 8):
            parts = [distributed.tiles_of_rank(w, h, r, world_size) for r in range(world_size)]
            allt = numpy.sort(numpy.concatenate(parts))
            assert numpy.array_equal(allt, numpy.arange(tx * ty))
            sizes = [len(p) for p in parts]
            assert max(sizes) == distributed.padded_tile_count(w, h, world_size)
            if tx * ty >= 8 * world_size:
                assert max(sizes) - min(sizes) <= max(2, tx * ty // (world_size * 20))   # balanced


@pytest.mark.parametrize("n", (2, 4, 8))
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` without a launcher starts N ranks itself (rehearsed without
    a GPU: rendezvous and one collective over gloo) and relays rank 0's JSON line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RTAMD_BENCH_LAUNCH_ONLY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["rank_sum"] == n * (n + 1) // 2


def test_bench_launcher_reports_a_failing_rank():
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RTAMD_BENCH_LAUNCH_ONLY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--workload", "no_such_workload"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def _nccl_worker(rank, world_size, port, out_path):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", rank))
    from raytracer_amd.world import World
    from raytracer_amd import distributed
    world = World.__new__(World)
    world.overrides = dict(resolution=(88, 60), sample_dim=2)
    World.__init__(world, viewmode="parallel", buildfunction="cornellbox_area", seed=3)
    job = distributed.FrameJob(world, rank, world_size)
    frame = job.finish(job.render())
    if rank == 0:
        numpy.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


def _nccl_single_worker(rank, port, out_path):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from raytracer_amd.world import World
    from raytracer_amd import distributed
    world = World.__new__(World)
    world.overrides = dict(resolution=(88, 60), sample_dim=2)
    World.__init__(world, viewmode="parallel", buildfunction="cornellbox_area", seed=3)
    job = distributed.FrameJob(world, 0, 1)
    bufs, _ = job.pipeline(2)
    recv = [[torch.empty_like(w)] for w in job.pipe_wire]
    pending = [None, None]
    for i in range(4):                      # frames in flight, as bench.py runs them
        k = i % 2
        if pending[k] is not None:
            pending[k].wait()
        job.render_fn(job.tile_ids, bufs[k], None)
        pending[k] = distributed.gather_to_root(job.send(k, clear=True), parts=recv[k], async_op=True, always_collective=True)
    pending[0].wait()
    parts = pending[1].wait()
    torch.cuda.synchronize()
    assert parts is recv[1] and pending[1].ran_collective
    numpy.save(out_path, job.finish(parts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_gather_with_one_rank(tmp_path):
    """The gather as bench.py issues it (device_id at initialisation, asynchronous, into a reused receive
    list, int32 wire buffers packed by the library) run through RCCL itself on the one GPU of the box:
    a group of one rank, the collective forced (a group of one normally skips it)."""
    import torch.multiprocessing as mp
    from raytracer_amd.world import World
    from raytracer_amd import runtime
    out = str(tmp_path / "one.npy")
    mp.spawn(_nccl_single_worker, args=(_free_port(), out), nprocs=1, join=True)
    world = World.__new__(World)
    world.overrides = dict(resolution=(88, 60), sample_dim=2)
    World.__init__(world, viewmode="parallel", buildfunction="cornellbox_area", seed=3)
    single, _ = runtime.render_image(world, mode="parallel")
    assert numpy.array_equal(numpy.load(out), single)


@pytest.mark.gpu
def test_two_ranks_over_rccl_equal_one_rank(tmp_path):
    """Two processes, one GPU each, RCCL gather: the frame equals the single-rank frame."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from raytracer_amd.world import World
    from raytracer_amd import runtime
    out = str(tmp_path / "two.npy")
    mp.spawn(_nccl_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    world = World.__new__(World)
    world.overrides = dict(resolution=(88, 60), sample_dim=2)
    World.__init__(world, viewmode="parallel", buildfunction="cornellbox_area", seed=3)
    single, _ = runtime.render_image(world, mode="parallel")
    assert numpy.array_equal(numpy.load(out), single)


@pytest.mark.gpu
def test_bench_with_two_ranks_gathers_over_gloo_when_rccl_is_refused():
    """`bench.py --gpus 2` on a box with one GPU: RCCL refuses two ranks on one device, the ranks say so and
    gather over gloo instead; with two GPUs the same command runs over RCCL."""
    import json
    import subprocess
    import sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RTAMD_BENCH_BACKEND")}
    env["RTAMD_BENCH_FORCE_RCCL"] = "1"         # do not skip RCCL because of the device count: let it refuse
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cornellbox_200_16spp_arealight", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and len(line["ranks"]["kernel_ms"]) == 2
    assert line["roofline"]["executed"]["same_frame_as_timed_kernel"]
    if torch.cuda.device_count() < 2:
        assert line["config"]["backend"] == "gloo" and "RCCL did not come up" in line["config"]["backend_note"]
    else:
        assert line["config"]["backend"] == "RCCL" and line["config"]["backend_note"] is None
