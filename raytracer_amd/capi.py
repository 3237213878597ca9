"""ctypes view of include/raytracer_amd.h: struct layouts, the shared library and its
function prototypes.  Loading fails loudly when the HIP library has not been built;
there is no fallback path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB") or os.path.join(HERE, "csrc", "librtamd.so")

ABI_VERSION = 7
MAX_BRANCH_LEVELS = 8
# levels a walk may pass through: the kernels' stack holds RT_STACK_MAX = 40 entries; a library built with a larger
# -DRT_STACK_MAX takes deeper trees (RTAMD_MAX_TREE_DEPTH = that value - 2 tells the host side)
MAX_TREE_DEPTH = int(os.environ.get("RTAMD_MAX_TREE_DEPTH", "38"))
N_COUNTERS = 16
COUNT_WORK = 1
COUNT_EXECUTED = 2

SPHERE, PLANE, CHECKER_PLANE, RECTANGLE, BOX, TRIANGLE, KDTREE = range(7)
CASTERS_ALL, CASTERS_NONE, CASTERS_MIXED = range(3)
TRACER_SHADE, TRACER_AREA, TRACER_PATH, TRACER_WHITTED = range(4)

d3 = C.c_double * 3


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("alt_material", C.c_int32),
                ("cast_shadow", C.c_int32), ("has_transform", C.c_int32), ("tree", C.c_int32),
                ("smooth", C.c_int32), ("n_inner", C.c_int32),
                ("p", C.c_double * 30), ("inv_point", C.c_double * 12),
                ("inv_dir", C.c_double * 9), ("inner_offset", C.c_int32), ("pad_", C.c_int32)]


class Transform(C.Structure):
    _fields_ = [("inv_point", C.c_double * 12), ("inv_dir", C.c_double * 9), ("pad_", C.c_double * 3)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("receives_shadow", C.c_int32), ("sampler", C.c_int32),
                ("texture", C.c_int32),
                ("ambient", d3), ("diffuse", d3), ("specular", d3), ("exp", C.c_double),
                ("reflect", d3), ("reflect_exp", C.c_double), ("emit", d3), ("kt", C.c_double),
                ("eta_front", C.c_double), ("eta_back", C.c_double), ("filter_in", d3), ("filter_out", d3)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cast_shadow", C.c_int32), ("sampler", C.c_int32),
                ("reserved", C.c_int32),
                ("radiance", d3), ("position", d3), ("attenuation", C.c_double),
                ("radius", C.c_double), ("edge_u", d3), ("edge_v", d3), ("normal", d3),
                ("pdf", C.c_double)]


class Ambient(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sampler", C.c_int32), ("radiance", d3), ("occluded", d3)]


class Sampler(C.Structure):
    _fields_ = [("num_patterns", C.c_int32), ("samples_per_pattern", C.c_int32),
                ("regular", C.c_int32), ("reserved", C.c_int32),
                ("square_offset", C.c_int64), ("disk_offset", C.c_int64), ("hemi_offset", C.c_int64)]


class KdNode(C.Structure):
    _fields_ = [("split", C.c_double), ("a", C.c_int32), ("b", C.c_int32)]


class Tree(C.Structure):
    _fields_ = [("bbox_lo", d3), ("bbox_hi", d3), ("node_offset", C.c_int32),
                ("ref_offset", C.c_int32), ("prim_offset", C.c_int32), ("smooth", C.c_int32),
                ("depth", C.c_int32), ("shadow_casters", C.c_int32), ("shadow_root", C.c_int32), ("walk_depth", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("hres", C.c_int32), ("vres", C.c_int32),
                ("mapping", C.c_int32), ("reserved", C.c_int32), ("offset", C.c_int64)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("tracer_kind", C.c_int32), ("depth_limit", C.c_int32),
                ("n_objects", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32),
                ("n_samplers", C.c_int32), ("n_trees", C.c_int32),
                ("n_kd_nodes", C.c_int64), ("n_refs", C.c_int64), ("n_prims", C.c_int64),
                ("n_sampler_doubles", C.c_int64), ("n_chunks", C.c_int64),
                ("background", d3), ("ambient", Ambient),
                ("objects", C.c_void_p), ("materials", C.c_void_p), ("lights", C.c_void_p),
                ("samplers", C.c_void_p), ("sampler_data", C.c_void_p), ("trees", C.c_void_p),
                ("kd_nodes", C.c_void_p), ("kd_triples", C.c_void_p), ("tri_geom", C.c_void_p), ("tri_prim", C.c_void_p),
                ("tri_order", C.c_void_p), ("tri_rows", C.c_void_p), ("leaf_boxes", C.c_void_p), ("chunk_boxes", C.c_void_p),
                ("prim_normals", C.c_void_p), ("prim_normal_offset", C.c_void_p),
                ("prim_material", C.c_void_p),
                ("n_textures", C.c_int32), ("n_transforms", C.c_int32), ("n_texel_bytes", C.c_int64),
                ("textures", C.c_void_p), ("texels", C.c_void_p), ("chunk_tree", C.c_void_p),
                ("leaf_aabb", C.c_void_p), ("transforms", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("eye", d3), ("u", d3), ("v", d3), ("w", d3),
                ("viewing_distance", C.c_double), ("lens_radius", C.c_double),
                ("focal_plane_distance", C.c_double), ("pixel_size", C.c_double),
                ("thin_lens", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("sampler", C.c_int32)]


class SampleOut(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("rgb8", C.c_void_p), ("hit_object", C.c_void_p),
                ("hit_prim", C.c_void_p), ("hit_t", C.c_void_p), ("n_rays", C.c_void_p)]


#: every symbol include/raytracer_amd.h declares
EXPORTS = ("rtamd_abi_version", "rtamd_device_check", "rtamd_last_error",
           "rtamd_scene_create", "rtamd_scene_destroy", "rtamd_render_pixels",
           "rtamd_trace_rays", "rtamd_render_tiles", "rtamd_pack_sums",
           "rtamd_kd_build_workspace_bytes", "rtamd_kd_build", "rtamd_kd_build_static", "rtamd_kd_last_error",
           "rtamd_shadow_tree_build")


class KdTreeOut(C.Structure):
    _fields_ = [("axis", C.c_void_p), ("split", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("first", C.c_void_p), ("count", C.c_void_p), ("leaf_refs", C.c_void_p),
                ("max_nodes", C.c_int64), ("max_refs", C.c_int64)]


def bind(lib):
    """Attach argument/return types to the entry points of a loaded library (the HIP
    library, or a checker library exporting the same ABI)."""
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    lib.rtamd_abi_version.restype = C.c_int
    lib.rtamd_abi_version.argtypes = []
    lib.rtamd_device_check.restype = C.c_int
    lib.rtamd_device_check.argtypes = []
    lib.rtamd_last_error.restype = C.c_char_p
    lib.rtamd_last_error.argtypes = []
    lib.rtamd_scene_create.restype = C.c_int
    lib.rtamd_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.rtamd_scene_destroy.restype = C.c_int
    lib.rtamd_scene_destroy.argtypes = [vp]
    lib.rtamd_render_pixels.restype = C.c_int
    lib.rtamd_render_pixels.argtypes = [vp, C.POINTER(Camera), u64, i64, vp, vp, vp, vp,
                                        C.POINTER(SampleOut), vp, u32, vp]
    lib.rtamd_trace_rays.restype = C.c_int
    lib.rtamd_trace_rays.argtypes = [vp, u64, i64, vp, vp, vp, vp, i32,
                                     C.POINTER(SampleOut), vp, u32, vp]
    lib.rtamd_render_tiles.restype = C.c_int
    lib.rtamd_render_tiles.argtypes = [vp, C.POINTER(Camera), u64, i64, vp, i32, i32,
                                       C.POINTER(SampleOut), vp, vp, vp, u32, vp]
    lib.rtamd_pack_sums.restype = C.c_int
    lib.rtamd_pack_sums.argtypes = [vp, vp, i64, i32, vp]
    lib.rtamd_kd_build_workspace_bytes.restype = C.c_int
    lib.rtamd_kd_build_workspace_bytes.argtypes = [i64, i64, i64, C.POINTER(C.c_size_t)]
    lib.rtamd_kd_build.restype = C.c_int
    lib.rtamd_kd_build.argtypes = [vp, vp, i64, C.POINTER(KdTreeOut), vp, C.c_size_t, i64,
                                   C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), vp]
    lib.rtamd_kd_build_static.restype = C.c_int
    lib.rtamd_kd_build_static.argtypes = [vp, vp, i64, C.POINTER(KdTreeOut), C.POINTER(i64), C.POINTER(i64),
                                          C.POINTER(i32), vp]
    lib.rtamd_kd_last_error.restype = C.c_char_p
    lib.rtamd_kd_last_error.argtypes = []
    lib.rtamd_shadow_tree_build.restype = C.c_int
    lib.rtamd_shadow_tree_build.argtypes = [vp, vp, i64, i32, i32, i32, C.c_double, vp, C.c_double,
                                            vp, vp, vp, i64, vp, vp, vp, i64,
                                            C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    return lib


_lib = None


def library():
    """The HIP library, loaded on first use."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
        if _lib.rtamd_abi_version() != ABI_VERSION:
            raise RuntimeError("librtamd.so ABI %d != binding ABI %d" % (_lib.rtamd_abi_version(), ABI_VERSION))
    return _lib


EUNSUPPORTED = -2


def check_kd(status, lib=None):
    if status != 0:
        lib = lib or library()
        raise RuntimeError("rtamd_kd_build failed (%d): %s" % (status, lib.rtamd_kd_last_error().decode()))


def check(status, lib=None):
    if status != 0:
        lib = lib or library()
        raise RuntimeError("rtamd error %d: %s" % (status, (lib.rtamd_last_error() or b"").decode()))
