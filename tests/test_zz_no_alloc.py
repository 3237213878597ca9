"""rtamd_kd_build_static (the kd build that allocates nothing: its buffers are a static device
array of the library) against rtamd_kd_build with a caller's workspace, through the C ABI:
identical trees, on two calls in a row, the second with a smaller input; and the size limit."""
import ctypes as C

import numpy
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("axis", "split", "left", "right", "first", "count", "leaf_refs")


def _boxes(n, seed):
    r = numpy.random.RandomState(seed)
    lo = numpy.round(r.rand(n, 3) * 10, 2)          # two decimals: plenty of equal keys
    hi = lo + numpy.round(r.rand(n, 3) ** 2, 2)
    return lo, hi


def _build(lib, torch, lo, hi, static):
    from raytracer_amd import capi
    dev = torch.device("cuda", 0)
    n = len(lo)
    d_lo, d_hi = torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev)
    max_nodes = max_refs = 32768
    i32 = dict(dtype=torch.int32, device=dev)
    arrays = dict(axis=torch.full((max_nodes,), -7, **i32), split=torch.zeros(max_nodes, dtype=torch.float64, device=dev),
                  left=torch.full((max_nodes,), -7, **i32), right=torch.full((max_nodes,), -7, **i32),
                  first=torch.full((max_nodes,), -7, **i32), count=torch.full((max_nodes,), -7, **i32),
                  leaf_refs=torch.full((max_refs,), -7, **i32))
    out = capi.KdTreeOut(max_nodes=max_nodes, max_refs=max_refs, **{k: v.data_ptr() for k, v in arrays.items()})
    n_nodes, n_refs, depth = C.c_int64(), C.c_int64(), C.c_int32()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    if static:
        rc = lib.rtamd_kd_build_static(d_lo.data_ptr(), d_hi.data_ptr(), n, C.byref(out), C.byref(n_nodes),
                                       C.byref(n_refs), C.byref(depth), stream)
    else:
        need = C.c_size_t()
        cap = 3 * 8192
        assert lib.rtamd_kd_build_workspace_bytes(cap, max_nodes, max_refs, C.byref(need)) == 0
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        rc = lib.rtamd_kd_build(d_lo.data_ptr(), d_hi.data_ptr(), n, C.byref(out), ws.data_ptr(), need.value, cap,
                                C.byref(n_nodes), C.byref(n_refs), C.byref(depth), stream)
    if rc != 0:
        return rc, lib.rtamd_kd_last_error().decode()
    nn, nr = n_nodes.value, n_refs.value
    tree = {k: v[:(nr if k == "leaf_refs" else nn)].cpu().numpy() for k, v in arrays.items()}
    tree["depth"] = depth.value
    return 0, tree


def test_static_workspace_build_equals_the_workspace_build():
    import torch
    from raytracer_amd import capi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    lib = capi.library()
    for n, seed in ((7000, 3), (450, 4)):           # two calls in a row, the second smaller
        lo, hi = _boxes(n, seed)
        rc_a, want = _build(lib, torch, lo, hi, static=False)
        rc_b, got = _build(lib, torch, lo, hi, static=True)
        assert rc_a == 0 and rc_b == 0, (want, got)
        assert got["depth"] == want["depth"] and len(got["axis"]) == len(want["axis"]) > 1
        for f in FIELDS:
            assert got[f].shape == want[f].shape, f
            assert numpy.array_equal(got[f].view(numpy.int64) if f == "split" else got[f],
                                     want[f].view(numpy.int64) if f == "split" else want[f]), "%s differs (n=%d)" % (f, n)
        # every box is referenced, and only boxes that exist
        assert set(got["leaf_refs"].tolist()) == set(range(n))
    # beyond the static array's capacity: refused with a message that names the limit
    lo, hi = _boxes(8193, 5)
    rc, msg = _build(lib, torch, lo, hi, static=True)
    assert rc == capi.EUNSUPPORTED and "8192" in msg
