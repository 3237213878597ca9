"""ctypes front end of the CPU checker (oracle/rt_oracle.c).

TEST INFRASTRUCTURE: imported by tests/, by `__graft_entry__.smoke()` and by bench.py's
`cpu_baseline` leg only.  It reads the same host tables the product uploads to the GPU
(raytracer_amd.flatten.FlatScene) and the struct layouts of raytracer_amd.capi.
"""
import ctypes as C
import os
import subprocess

import numpy

from raytracer_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_oracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-C", HERE, "-s"])


def library():
    global _lib
    if _lib is None:
        build()        # make: a no-op when the library is newer than its sources
        lib = C.CDLL(LIB_PATH)
        vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
        lib.rto_render_pixels.restype = C.c_int
        lib.rto_render_pixels.argtypes = [C.POINTER(capi.SceneDesc), C.POINTER(capi.Camera), u64, i64,
                                          vp, vp, vp, vp, C.POINTER(capi.SampleOut), vp, C.c_int]
        lib.rto_trace_rays.restype = C.c_int
        lib.rto_trace_rays.argtypes = [C.POINTER(capi.SceneDesc), u64, i64, vp, vp, vp, vp, i32,
                                       C.POINTER(capi.SampleOut), vp, C.c_int]
        lib.rto_render_tiles.restype = C.c_int
        lib.rto_render_tiles.argtypes = [C.POINTER(capi.SceneDesc), C.POINTER(capi.Camera), u64, i64, vp,
                                         i32, i32, C.POINTER(capi.SampleOut), vp, vp, vp, C.c_int]
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data if a is not None else None


class OracleScene(object):
    """A FlatScene bound to host pointers."""

    COUNTER_NAMES = ("closest", "shadow", "samples", "kd_nodes", "kd_leaves", "tri_tests", "analytic_tests")

    def __init__(self, flat):
        self.flat = flat
        self.keep = {k: numpy.ascontiguousarray(v) for k, v in flat.arrays.items()}
        self.desc = flat.desc(lambda name, arr: self.keep[name].ctypes.data if self.keep[name].size else None)
        self.counters = numpy.zeros(capi.N_COUNTERS, dtype=numpy.uint64)

    def _out(self, n, want):
        n = max(int(n), 1)
        shapes = dict(rgb=((n, 3), numpy.float64), rgb8=((n, 3), numpy.int32), hit_object=((n,), numpy.int32),
                      hit_prim=((n,), numpy.int32), hit_t=((n,), numpy.float64), n_rays=((n, 2), numpy.int32))
        bufs = {k: numpy.zeros(*shapes[k]) for k in want}
        out = capi.SampleOut()
        for k, v in bufs.items():
            setattr(out, k, v.ctypes.data)
        return bufs, out

    def read_counters(self, reset=True):
        vals = dict(zip(self.COUNTER_NAMES, (int(v) for v in self.counters[:7])))
        if reset:
            self.counters[:] = 0
        return vals

    def render_pixels(self, cam, seed, pixel_index, sample_index, plane_points, lens_points=None,
                      want=("rgb", "rgb8", "hit_object", "hit_prim", "hit_t", "n_rays"), threads=1):
        n = len(pixel_index)
        bufs, out = self._out(n, want)
        pi = numpy.ascontiguousarray(pixel_index, dtype=numpy.int32)
        si = numpy.ascontiguousarray(sample_index, dtype=numpy.int32)
        pp = numpy.ascontiguousarray(plane_points, dtype=numpy.float64)
        lp = numpy.ascontiguousarray(lens_points, dtype=numpy.float64) if lens_points is not None else None
        library().rto_render_pixels(C.byref(self.desc), C.byref(cam), seed, n, _ptr(pi), _ptr(si), _ptr(pp),
                                    _ptr(lp), C.byref(out), self.counters.ctypes.data, threads)
        return {k: v[:n] for k, v in bufs.items()}

    def trace_rays(self, seed, origins, directions, pixel_index=None, sample_index=None, depth=0,
                   want=("rgb", "hit_object", "hit_prim", "hit_t", "n_rays"), threads=1):
        n = len(origins)
        bufs, out = self._out(n, want)
        o = numpy.ascontiguousarray(origins, dtype=numpy.float64)
        d = numpy.ascontiguousarray(directions, dtype=numpy.float64)
        pi = numpy.ascontiguousarray(pixel_index, dtype=numpy.int32) if pixel_index is not None else None
        si = numpy.ascontiguousarray(sample_index, dtype=numpy.int32) if sample_index is not None else None
        library().rto_trace_rays(C.byref(self.desc), seed, n, _ptr(pi), _ptr(si), _ptr(o), _ptr(d), depth,
                                 C.byref(out), self.counters.ctypes.data, threads)
        return {k: v[:n] for k, v in bufs.items()}

    def render_tiles(self, cam, seed, tile_ids, sample_begin, sample_end, per_sample_want=(), threads=1):
        tile_ids = numpy.ascontiguousarray(tile_ids, dtype=numpy.int32)
        ns = sample_end - sample_begin
        n = len(tile_ids) * 64 * ns
        bufs, out = self._out(n, per_sample_want) if per_sample_want else ({}, None)
        sum_rgb8 = numpy.zeros((len(tile_ids) * 64, 3), dtype=numpy.int64)
        sum_sq = numpy.zeros((len(tile_ids) * 64, 3), dtype=numpy.float64)
        library().rto_render_tiles(C.byref(self.desc), C.byref(cam), seed, len(tile_ids), _ptr(tile_ids),
                                   sample_begin, sample_end, C.byref(out) if out is not None else None,
                                   sum_rgb8.ctypes.data, sum_sq.ctypes.data, self.counters.ctypes.data, threads)
        return sum_rgb8, sum_sq, bufs
