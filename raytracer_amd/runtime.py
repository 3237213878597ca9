"""Device side of the scene API: uploads a flattened world and calls the HIP kernels
through the C ABI (include/raytracer_amd.h).  torch is used for device buffers and
streams only.

The functions here are what the camera / tracer classes delegate to:
`render_pixels` (camera.py:60-67, :227-234), `pixel_plane_point_pairs`
(camera.py:69-98, :236-267), `render_frame` (camera.py:35-58, :100-139) and
`trace_rays` (tracer.py:26-35) of /root/reference.
"""
import ctypes as C

import numpy

from . import capi, device_layout, flatten, rng


def _torch():
    import torch
    return torch


class PixelKey(tuple):
    """(col, row) that also remembers which sample of the pixel's pattern it stands for.
    The reference's pairs carry no sample index because its draws are positional in a
    global stream; here draws are keyed by (pixel, sample), so the producer attaches it."""
    sample_index = 0

    def __new__(cls, col, row, sample_index):
        self = tuple.__new__(cls, (col, row))
        self.sample_index = sample_index
        return self


class DeviceScene(object):
    """A world's tables resident in HBM plus the library's scene handle."""

    def __init__(self, world, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the ray tracing path has no CPU fallback")
        self.lib = capi.library()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.flat = flatten.flatten_world(world)
        self.tensors = {}
        with torch.cuda.device(self.device):
            arrays, n_chunks = device_layout.device_arrays(self.flat)
            self.n_kd_nodes = len(arrays["kd_nodes"]) // C.sizeof(capi.KdNode)
            self.n_refs = len(arrays["tri_prim"])
            for name, arr in arrays.items():
                host = torch.from_numpy(numpy.ascontiguousarray(arr).view(numpy.uint8).reshape(-1).copy())
                if host.numel() == 0:
                    host = torch.zeros(16, dtype=torch.uint8)
                self.tensors[name] = host.to(self.device)
            desc = self.flat.desc(lambda name, arr: self.tensors[name].data_ptr())
            desc.tri_order = self.tensors["tri_order"].data_ptr()
            desc.tri_rows = self.tensors["tri_rows"].data_ptr()
            desc.kd_triples = self.tensors["kd_triples"].data_ptr()
            desc.leaf_boxes = self.tensors["leaf_boxes"].data_ptr()
            desc.chunk_boxes = self.tensors["chunk_boxes"].data_ptr()
            desc.chunk_tree = self.tensors["chunk_tree"].data_ptr()
            desc.leaf_aabb = self.tensors["leaf_aabb"].data_ptr()
            desc.n_chunks = n_chunks
            self.table_sizes(desc)
            handle = C.c_void_p()
            # the library reads the small tables back on the null stream: the uploads must have landed
            torch.cuda.current_stream(self.device).synchronize()
            capi.check(self.lib.rtamd_scene_create(C.byref(desc), C.byref(handle)), self.lib)
        self.handle = handle
        self.counters = torch.zeros(capi.N_COUNTERS, dtype=torch.int64, device=self.device)

    def table_sizes(self, desc):
        """Node and row counts of the device tables, which may hold appended nodes and rows
        (shadow subtrees of big leaves) beyond the canonical ones."""
        desc.n_kd_nodes, desc.n_refs = self.n_kd_nodes, self.n_refs
        return desc

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rtamd_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------
    def on_device(self):
        """Context in which the scene's own device is the current one (the library refuses
        a launch from any other)."""
        return _torch().cuda.device(self.device)

    def stream_ptr(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, array, dtype):
        torch = _torch()
        return torch.from_numpy(numpy.ascontiguousarray(array, dtype=dtype)).to(self.device)

    def sample_buffers(self, n, want=("rgb", "rgb8", "hit_object", "hit_prim", "hit_t", "n_rays")):
        torch = _torch()
        n = max(int(n), 1)
        shapes = dict(rgb=((n, 3), torch.float64), rgb8=((n, 3), torch.int32),
                      hit_object=((n,), torch.int32), hit_prim=((n,), torch.int32),
                      hit_t=((n,), torch.float64), n_rays=((n, 2), torch.int32))
        bufs = {k: torch.zeros(shapes[k][0], dtype=shapes[k][1], device=self.device) for k in want}
        out = capi.SampleOut()
        for k, t in bufs.items():
            setattr(out, k, t.data_ptr())
        return bufs, out

    COUNTER_NAMES = ("closest", "shadow", "samples", "kd_nodes", "kd_leaves", "tri_tests", "analytic_tests")
    EXECUTED_NAMES = COUNTER_NAMES + ("leaves_skipped", "node_records", "boxes_tested", "rows_loaded", "bytes_loaded")

    def read_counters(self, reset=True, executed=False):
        """Launch-wide counters since the last reset.  With `executed` the slots a
        count_executed launch fills are named as well (include/raytracer_amd.h)."""
        vals = self.counters.cpu().numpy().copy()
        if reset:
            self.counters.zero_()
        names = self.EXECUTED_NAMES if executed else self.COUNTER_NAMES
        return dict(zip(names, (int(v) for v in vals[:len(names)])))

    @staticmethod
    def _flags(count_work, count_executed):
        return (capi.COUNT_WORK if count_work else 0) | (capi.COUNT_EXECUTED if count_executed else 0)

    # -- the three entry points ------------------------------------------------------
    def render_pixels(self, cam, seed, pixel_index, sample_index, plane_points, lens_points=None,
                      want=("rgb", "rgb8", "hit_object", "hit_prim", "hit_t", "n_rays"), count_work=False,
                      count_executed=False):
        n = len(pixel_index)
        bufs, out = self.sample_buffers(n, want)
        if n:
            pi = self.to_device(pixel_index, numpy.int32)
            si = self.to_device(sample_index, numpy.int32)
            pp = self.to_device(plane_points, numpy.float64)
            lp = self.to_device(lens_points, numpy.float64) if lens_points is not None else None
            with self.on_device():
                capi.check(self.lib.rtamd_render_pixels(
                    self.handle, C.byref(cam), seed, n, pi.data_ptr(), si.data_ptr(), pp.data_ptr(),
                    lp.data_ptr() if lp is not None else None, C.byref(out), self.counters.data_ptr(),
                    self._flags(count_work, count_executed), self.stream_ptr()), self.lib)
        return {k: v[:n].cpu().numpy() for k, v in bufs.items()}

    def trace_rays(self, seed, origins, directions, pixel_index=None, sample_index=None, depth=0,
                   want=("rgb", "hit_object", "hit_prim", "hit_t", "n_rays"), count_work=False, count_executed=False):
        n = len(origins)
        bufs, out = self.sample_buffers(n, want)
        if n:
            o = self.to_device(origins, numpy.float64)
            d = self.to_device(directions, numpy.float64)
            pi = self.to_device(pixel_index, numpy.int32) if pixel_index is not None else None
            si = self.to_device(sample_index, numpy.int32) if sample_index is not None else None
            with self.on_device():
                capi.check(self.lib.rtamd_trace_rays(
                    self.handle, seed, n, pi.data_ptr() if pi is not None else None,
                    si.data_ptr() if si is not None else None, o.data_ptr(), d.data_ptr(), depth,
                    C.byref(out), self.counters.data_ptr(), self._flags(count_work, count_executed),
                    self.stream_ptr()), self.lib)
        return {k: v[:n].cpu().numpy() for k, v in bufs.items()}

    def render_tiles(self, cam, seed, tile_ids, sample_begin, sample_end, sum_rgb8=None, sum_sq=None,
                     per_sample=None, count_work=False, count_executed=False):
        """Asynchronous: accumulates into the given device tensors on the current stream."""
        out = per_sample[1] if per_sample is not None else None
        with self.on_device():
            capi.check(self.lib.rtamd_render_tiles(
                self.handle, C.byref(cam), seed, tile_ids.numel(), tile_ids.data_ptr(), sample_begin, sample_end,
                C.byref(out) if out is not None else None,
                sum_rgb8.data_ptr() if sum_rgb8 is not None else None,
                sum_sq.data_ptr() if sum_sq is not None else None,
                self.counters.data_ptr(), self._flags(count_work, count_executed), self.stream_ptr()), self.lib)


    def pack_sums(self, sum_rgb8, wire, clear=False):
        """Asynchronous: wire (int32) = the 64-bit sums saturated; `clear` zeroes the sums in the same pass."""
        with self.on_device():
            capi.check(self.lib.rtamd_pack_sums(sum_rgb8.data_ptr(), wire.data_ptr(), sum_rgb8.numel(), 1 if clear else 0,
                                                self.stream_ptr()), self.lib)


def device_scene(world):
    """The world's DeviceScene, created on first use.  A world is immutable once built
    (as in the reference, where workers fork a copy); call `world.invalidate()` after
    editing it."""
    ds = getattr(world, "_device_scene", None)
    if ds is None:
        ds = DeviceScene(world)
        world._device_scene = ds
    return ds


# ------------------------------------------------------------------------------------
# frame geometry shared by host producers and the tile kernel

def tile_grid(width, height):
    return (width + 7) // 8, (height + 7) // 8


def pattern_choice(world, seed, counter=0):
    """Per-pixel pattern index, row-major (height, width): the `vp.sampler.sample()` call
    per pixel of camera.py:78 (counter 0) and `sample_unit_disk()` of :246 (counter 1)."""
    vp = world.viewplane
    w, h = vp.resolution
    sampler = vp.sampler
    n_pat = len(sampler.patterns)
    if type(sampler).__name__ == "RegularSampler":
        return numpy.zeros((h, w), dtype=numpy.int64)
    pix = numpy.arange(w * h, dtype=numpy.uint64)
    return rng.index_array(rng.draw64_array(seed, rng.PHASE_PIXEL, pix, 0, counter), n_pat).reshape(h, w)


def sample_order(world, seed):
    """The shuffled order in which sample indices are visited (camera.py:83-84)."""
    n = len(world.viewplane.sampler.patterns[0])
    order = list(range(n))
    rng.context.seed = seed
    rng.context.point(rng.PHASE_FRAME, 0, 0)
    rng.shuffle(order)
    return order


def plane_points_for(world, camera, seed, sample_idx):
    """View-plane points (h, w, 3) and, for a thin lens, lens points (h, w, 3) of every
    pixel for one sample index (camera.py:90-93, :258-262)."""
    vp = world.viewplane
    w, h = vp.resolution
    ps = vp.pixel_size
    sampler = vp.sampler
    square = numpy.array(sampler.patterns, dtype=numpy.float64)           # (patterns, n, 2)
    pat = pattern_choice(world, seed, 0)
    s = square[pat, sample_idx]                                           # (h, w, 2)
    cols = numpy.arange(w)[None, :]
    rows = numpy.arange(h)[:, None]
    pp = numpy.zeros((h, w, 3))
    pp[..., 0] = ps * (cols - w // 2 + s[..., 0])
    pp[..., 1] = ps * (rows - h // 2 + s[..., 1])
    lens = None
    if getattr(camera, "IS_THIN_LENS", False):
        disk = numpy.array(sampler.concentric_patterns, dtype=numpy.float64)
        dpat = pattern_choice(world, seed, 1)
        dk = disk[dpat, sample_idx]
        lens = numpy.zeros((h, w, 3))
        lens[..., 0] = dk[..., 0]
        lens[..., 1] = dk[..., 1]
        lens = numpy.asarray(camera.eye, dtype=numpy.float64) + lens * camera.lens_radius
    return pp, lens


def pixel_plane_point_pairs(world, camera, chunk_size):
    """Generator with the chunking of camera.py:69-98: sample passes in shuffled order,
    rows then columns inside a pass, lists of `chunk_size` pairs and one final (possibly
    empty) remainder list."""
    seed = world.seed
    w, h = world.viewplane.resolution
    results = []
    for sample_idx in sample_order(world, seed):
        pp, lens = plane_points_for(world, camera, seed, sample_idx)
        for row in range(h):
            for col in range(w):
                key = PixelKey(col, row, sample_idx)
                value = pp[row, col] if lens is None else (pp[row, col], lens[row, col])
                results.append((key, value))
                if len(results) >= chunk_size:
                    yield results
                    results = []
    yield results


def render_pixels(world, camera, pairs):
    """camera.render_pixels: [(pixel, r, g, b)] with r,g,b = int(c*255)."""
    pairs = list(pairs)
    if not pairs:
        return []
    ds = device_scene(world)
    w = world.viewplane.resolution[0]
    thin = getattr(camera, "IS_THIN_LENS", False)
    n = len(pairs)
    pixel_index = numpy.empty(n, dtype=numpy.int32)
    sample_index = numpy.empty(n, dtype=numpy.int32)
    plane = numpy.empty((n, 3))
    lens = numpy.empty((n, 3)) if thin else None
    seen = {}
    for i, (pixel, value) in enumerate(pairs):
        pixel_index[i] = pixel[1] * w + pixel[0]
        si = getattr(pixel, "sample_index", None)
        if si is None:                       # caller-made pair: number repeats of a pixel
            si = seen.get(tuple(pixel), 0)
            seen[tuple(pixel)] = si + 1
        sample_index[i] = si
        if thin:
            plane[i], lens[i] = value[0], value[1]
        else:
            plane[i] = value
    cam = flatten.camera_record(world, camera)
    out = ds.render_pixels(cam, world.seed, pixel_index, sample_index, plane, lens, want=("rgb8",))
    rgb8 = out["rgb8"]
    return [(pairs[i][0], int(rgb8[i, 0]), int(rgb8[i, 1]), int(rgb8[i, 2])) for i in range(n)]


def trace_rays(world, rays, depth=0):
    """tracer.trace_ray for a list of rays; draws are keyed by the current draw context."""
    ds = device_scene(world)
    n = len(rays)
    o = numpy.array([r.origin for r in rays], dtype=numpy.float64).reshape(n, 3)
    d = numpy.array([r.direction for r in rays], dtype=numpy.float64).reshape(n, 3)
    pix = numpy.full(n, rng.context.a if rng.context.phase == rng.PHASE_TRACE else 0, dtype=numpy.int32)
    smp = numpy.full(n, rng.context.b if rng.context.phase == rng.PHASE_TRACE else 0, dtype=numpy.int32)
    out = ds.trace_rays(world.seed, o, d, pix, smp, depth=depth, want=("rgb",))
    return [tuple(c) for c in out["rgb"]]


class FrameBuffers(object):
    """Tile-compact accumulators for one rank's share of a frame."""

    def __init__(self, ds, tile_ids):
        torch = _torch()
        self.tile_ids = torch.as_tensor(numpy.asarray(tile_ids, dtype=numpy.int32)).to(ds.device)
        n = max(len(tile_ids), 1) * 64
        self.sum_rgb8 = torch.zeros((n, 3), dtype=torch.int64, device=ds.device)
        self.sum_sq = torch.zeros((n, 3), dtype=torch.float64, device=ds.device)


def scatter_tiles(values, tile_ids, width, height):
    """(n_tiles*64, 3) tile-compact rows -> (height, width, 3) image (host numpy)."""
    tx, _ = tile_grid(width, height)
    tile_ids = numpy.asarray(tile_ids, dtype=numpy.int64)
    pix = numpy.arange(64)
    cols = (tile_ids[:, None] % tx) * 8 + (pix[None, :] & 7)
    rows = (tile_ids[:, None] // tx) * 8 + (pix[None, :] >> 3)
    ok = (cols < width) & (rows < height)
    img = numpy.zeros((height, width, 3), dtype=values.dtype)
    img[rows[ok], cols[ok]] = values.reshape(len(tile_ids), 64, 3)[ok]
    return img


def render_image(world, mode="parallel", tile_ids=None, samples=None):
    """Whole-frame render on the current device.

    mode "parallel": what `--viewmode=parallel` converges to - per pixel the mean of the
    per-sample integer triples int(c*255) (camera.py:66, main.py:99-103), clamped to 255
    (main.py:104-106).  mode "render": `camera.render`'s pixel - mean of c**2 over the
    samples, times 255, truncated, clamped (camera.py:55-58, main.py:112-128).
    Returns (image uint8 (h, w, 3), info dict)."""
    ds = device_scene(world)
    torch = _torch()
    w, h = world.viewplane.resolution
    n_samples = len(world.viewplane.sampler.patterns[0])
    s0, s1 = samples if samples is not None else (0, n_samples)
    tx, ty = tile_grid(w, h)
    if tile_ids is None:
        tile_ids = numpy.arange(tx * ty, dtype=numpy.int32)
    fb = FrameBuffers(ds, tile_ids)
    cam = flatten.camera_record(world)
    ds.render_tiles(cam, world.seed, fb.tile_ids, s0, s1,
                    sum_rgb8=fb.sum_rgb8 if mode == "parallel" else None,
                    sum_sq=fb.sum_sq if mode != "parallel" else None)
    torch.cuda.synchronize(ds.device)
    count = s1 - s0
    if mode == "parallel":
        mean = scatter_tiles(fb.sum_rgb8.cpu().numpy(), tile_ids, w, h) / float(count)
        img = numpy.minimum(mean, 255)
    else:
        mean = scatter_tiles(fb.sum_sq.cpu().numpy(), tile_ids, w, h) / float(count)
        img = numpy.minimum(numpy.trunc(mean * 255), 255)
    return numpy.clip(img, 0, 255).astype(numpy.uint8), dict(samples=count, counters=ds.read_counters())


def render_frame(world, camera, pixel_func, progressive=False):
    """camera.render / camera.render_progressive: calls pixel_func((col,row), r, g, b) with
    the mean of squared radiance * 255, truncated (camera.py:55-58, :135-139).  The
    progressive form reports every pixel after every sample pass, passes in shuffled order."""
    ds = device_scene(world)
    torch = _torch()
    w, h = world.viewplane.resolution
    tx, ty = tile_grid(w, h)
    tile_ids = numpy.arange(tx * ty, dtype=numpy.int32)
    fb = FrameBuffers(ds, tile_ids)
    cam = flatten.camera_record(world, camera)
    n_samples = len(world.viewplane.sampler.patterns[0])

    def report(count):
        torch.cuda.synchronize(ds.device)
        mean = scatter_tiles(fb.sum_sq.cpu().numpy(), tile_ids, w, h) / float(count)
        # int() of NaN / infinity raises in the reference (camera.py:58); report such pixels as 0 / the int32 limits
        vals = numpy.nan_to_num(mean * 255, nan=0.0, posinf=2147483647.0, neginf=-2147483648.0)
        for row in range(h):
            for col in range(w):
                r, g, b = vals[row, col]
                pixel_func((col, row), int(r), int(g), int(b))

    if not progressive:
        ds.render_tiles(cam, world.seed, fb.tile_ids, 0, n_samples, sum_sq=fb.sum_sq)
        report(n_samples)
        return
    for done, s in enumerate(sample_order(world, world.seed), 1):
        ds.render_tiles(cam, world.seed, fb.tile_ids, s, s + 1, sum_sq=fb.sum_sq)
        report(done)
