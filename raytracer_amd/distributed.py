"""One frame over several GPUs: one process per GPU, the scene replicated, 8x8-pixel
tiles dealt out across ranks, one gather of the tile-compact accumulators to rank 0.

The reference spreads 1000-sample chunks over forked workers through a job queue and
averages what comes back in the parent (/root/reference/main.py:82-110).  Samples are
independent and the scene is read-only, so here the split is static: tile t belongs to
rank (tile_x + tile_y) mod world_size - diagonal stripes, which spread the object in the
middle of the frame evenly.  The only exchange is the final gather (torch.distributed;
backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy

from . import runtime


def tile_owner(tile_ids, tiles_x, world_size):
    tile_ids = numpy.asarray(tile_ids, dtype=numpy.int64)
    return ((tile_ids % tiles_x) + (tile_ids // tiles_x)) % world_size


def tiles_of_rank(width, height, rank, world_size):
    """Tile ids (row-major over the 8x8 tile grid) rendered by `rank`, ascending."""
    tx, ty = runtime.tile_grid(width, height)
    all_tiles = numpy.arange(tx * ty, dtype=numpy.int64)
    return all_tiles[tile_owner(all_tiles, tx, world_size) == rank].astype(numpy.int32)


def padded_tile_count(width, height, world_size):
    """Largest tile count of any rank: every rank pads its buffer to this for the gather."""
    tx, ty = runtime.tile_grid(width, height)
    owners = tile_owner(numpy.arange(tx * ty), tx, world_size)
    return int(numpy.bincount(owners, minlength=world_size).max())


class PendingGather(object):
    """A gather that may still be running on the collective's own stream."""

    def __init__(self, parts, work=None, ran_collective=False):
        self.parts, self.work, self.ran_collective = parts, work, ran_collective

    def wait(self):
        """Order the caller's stream after the gather; returns the root's list (None elsewhere)."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.parts


def gather_to_root(local, dist=None, root=0, parts=None, async_op=False, always_collective=False):
    """Gather equal-shaped per-rank tensors on `root`; returns the list there, None elsewhere.

    `parts` lets the root reuse its receive buffers.  With async_op the collective is only
    enqueued (RCCL runs it on its own stream once the producing kernel is done, so the
    next frame's kernel overlaps it) and a PendingGather is returned.  A group of one rank has
    nothing to gather and skips the collective unless always_collective asks for it (a test of the
    collective's own path on a box with one GPU)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return PendingGather([local]) if async_op else [local]
    world = dist.get_world_size()
    rank = dist.get_rank()
    if local.is_cuda and dist.get_backend() == "gloo":
        local = local.cpu()             # gloo gathers host tensors (CPU tests, single-GPU rehearsals)
        parts = None
    if rank == root and parts is None:
        parts = [torch.empty_like(local) for _ in range(world)]
    if rank != root:
        parts = None
    work = None
    try:
        work = dist.gather(local, gather_list=parts, dst=root, async_op=async_op)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: all ranks collect, only the root keeps the result
        every = [torch.empty_like(local) for _ in range(world)]
        work = dist.all_gather(every, local, async_op=async_op)
        parts = every if rank == root else None
    return PendingGather(parts, work, True) if async_op else parts


def assemble(parts, width, height, world_size):
    """Rank-ordered tile-compact buffers (numpy, padded) -> (height, width, C) image."""
    img = None
    for rank, part in enumerate(parts):
        tiles = tiles_of_rank(width, height, rank, world_size)
        part = numpy.asarray(part)[:len(tiles) * 64]
        piece = runtime.scatter_tiles(part, tiles, width, height)
        img = piece if img is None else img + piece
    return img


class FrameJob(object):
    """A rank's share of one frame: its tile list and padded accumulators.

    `render_fn(tile_ids_tensor, sum_rgb8, sum_sq)` adds the rank's samples into the
    accumulators; the default launches the HIP tile kernel of the world's device scene."""

    def __init__(self, world, rank, world_size, device=None, render_fn=None, mode="parallel"):
        import torch
        self.world, self.rank, self.world_size, self.mode = world, rank, world_size, mode
        self.width, self.height = world.viewplane.resolution
        self.tiles = tiles_of_rank(self.width, self.height, rank, world_size)
        self.padded = padded_tile_count(self.width, self.height, world_size)
        self.n_samples = len(world.viewplane.sampler.patterns[0])
        if render_fn is None:
            from . import flatten
            ds = runtime.device_scene(world)
            cam = flatten.camera_record(world)
            device = ds.device

            def render_fn(tile_ids, sum_rgb8, sum_sq):
                ds.render_tiles(cam, world.seed, tile_ids, 0, self.n_samples, sum_rgb8=sum_rgb8, sum_sq=sum_sq)
            self.ds = ds
        else:
            self.ds = None
        self.render_fn = render_fn
        self.device = device or torch.device("cpu")
        self.tile_ids = torch.as_tensor(self.tiles).to(self.device)
        self.sum_rgb8 = torch.zeros((self.padded * 64, 3), dtype=torch.int64, device=self.device)
        self.sum_sq = torch.zeros((self.padded * 64, 3), dtype=torch.float64, device=self.device)

    def render(self):
        """Zero, render this rank's tiles (asynchronous on the GPU), gather to rank 0."""
        want_int = self.mode == "parallel"
        acc = self.sum_rgb8 if want_int else self.sum_sq
        acc.zero_()
        if len(self.tiles):
            self.render_fn(self.tile_ids, self.sum_rgb8 if want_int else None, None if want_int else self.sum_sq)
        return gather_to_root(acc)

    def pipeline(self, depth=2):
        """`depth` integer accumulators with their wire buffers and receive lists for frames
        in flight: frame i renders into buffer i % depth while the gather of frame i-1
        still runs.  Returns (accumulators, receive lists); `send(slot)` packs a slot."""
        import torch
        bufs = [self.sum_rgb8] + [torch.zeros_like(self.sum_rgb8) for _ in range(depth - 1)]
        self.pipe_bufs = bufs
        self.pipe_wire = [torch.zeros(b.shape, dtype=torch.int32, device=b.device) for b in bufs]
        is_root = self.rank == 0 and self.world_size > 1
        recv = [[torch.empty_like(w) for _ in range(self.world_size)] if is_root else None for w in self.pipe_wire]
        return bufs, recv

    def send(self, slot, clear=False):
        """The slot's accumulator as it travels: int32 per channel (saturated), half the
        bytes of the kernel's 64-bit sums.  A pixel's sum only leaves the int32 range when
        samples were infinite, and such a pixel is clamped to 0 or 255 on arrival anyway.
        `clear` zeroes the accumulator for the slot's next frame.  On the GPU this is one pass of
        the library's pack kernel (rtamd_pack_sums); accumulators of a caller's own render_fn on
        the host (the CPU tests of the job's logic) take the same steps with torch."""
        import torch
        buf, wire = self.pipe_bufs[slot], self.pipe_wire[slot]
        if self.ds is not None and buf.is_cuda:
            self.ds.pack_sums(buf, wire, clear=clear)
        else:
            wire.copy_(torch.clamp(buf, -2147483648, 2147483647))
            if clear:
                buf.zero_()
        return wire

    def finish(self, parts):
        """On rank 0: the frame as uint8 (height, width, 3), by the rules of main.py:99-106
        ("parallel") or camera.py:55-58 ("render")."""
        if parts is None:
            return None
        host = [p.cpu().numpy() for p in parts]
        total = assemble(host, self.width, self.height, self.world_size) / float(self.n_samples)
        if self.mode == "parallel":
            img = numpy.minimum(total, 255)
        else:
            img = numpy.minimum(numpy.trunc(total * 255), 255)
        return numpy.clip(img, 0, 255).astype(numpy.uint8)
