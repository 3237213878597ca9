"""World -> flat scene tables (host numpy arrays laid out as include/raytracer_amd.h says).

The world is whatever a `build_function(world, viewmode)` left behind: `objects`,
`lights`, `ambient_color`, `tracer`, `background_color`, `viewplane`, `camera`
(/root/reference/buildfunctions/*.py, consumed by main.py:35-46 and material.py /
light.py).  Flattening walks those in list order - the order is observable, since the
first object wins ties and lights are shaded in sequence.
"""
import ctypes as C

import numpy

import raytracer_amd
from . import capi

raytracer_amd.activate()

import geometry as G          # the scene API modules (raytracer_amd.activate() puts them on sys.path)
import kdtree as K
import light as LT
import material as M
import sampler as S
import tracer as T

MAX_TREE_DEPTH = capi.MAX_TREE_DEPTH + 2      # traversal stack bound compiled into the kernels
NO_SHADOW_BIT = 1 << 30


class FlatScene(object):
    """Host copy of every table plus the scalar fields of rtamd_scene_desc."""

    TABLES = ("objects", "materials", "lights", "samplers", "sampler_data", "trees",
              "kd_nodes", "tri_geom", "tri_prim", "prim_normals", "prim_normal_offset",
              "prim_material", "textures", "texels", "transforms")

    def __init__(self):
        self.arrays = {}
        self.tracer_kind = capi.TRACER_SHADE
        self.depth_limit = 0
        self.background = numpy.zeros(3)
        self.ambient = capi.Ambient()
        self.sampler_objects = []      # id() order of sampler tables
        self.material_objects = []

    def nbytes(self):
        return sum(a.nbytes for a in self.arrays.values())

    def desc(self, address_of):
        """rtamd_scene_desc whose table pointers come from address_of(name, array)."""
        d = capi.SceneDesc()
        d.abi_version = capi.ABI_VERSION
        d.tracer_kind = self.tracer_kind
        d.depth_limit = self.depth_limit
        a = self.arrays
        d.n_objects = len(a["objects"]) // C.sizeof(capi.Object)
        d.n_materials = len(a["materials"]) // C.sizeof(capi.Material)
        d.n_lights = len(a["lights"]) // C.sizeof(capi.Light)
        d.n_samplers = len(a["samplers"]) // C.sizeof(capi.Sampler)
        d.n_trees = len(a["trees"]) // C.sizeof(capi.Tree)
        d.n_kd_nodes = len(a["kd_nodes"]) // C.sizeof(capi.KdNode)
        d.n_refs = len(a["tri_prim"])
        d.n_prims = len(a["prim_material"])
        d.n_sampler_doubles = len(a["sampler_data"])
        d.n_textures = len(a["textures"]) // C.sizeof(capi.Texture)
        d.n_texel_bytes = len(a["texels"])
        d.n_transforms = len(a["transforms"]) // C.sizeof(capi.Transform)
        d.background[:] = list(self.background)
        d.ambient = self.ambient
        for name in self.TABLES:
            setattr(d, name, address_of(name, a[name]))
        return d


def _bytes_of(ctypes_array):
    return numpy.frombuffer(bytes(ctypes_array), dtype=numpy.uint8).copy()


def _fill(dst, values):
    values = numpy.asarray(values, dtype=numpy.float64).ravel()
    for i, v in enumerate(values):
        dst[i] = v


class _Flattener(object):
    def __init__(self, world):
        self.world = world
        self.samplers, self.sampler_ids = [], {}
        self.materials, self.material_ids = [], {}
        self.textures, self.texture_ids, self.texels, self.n_texel_bytes = [], {}, [], 0
        self.trees = []
        self.transforms = []
        self.kd_nodes = []
        self.tri_geom, self.tri_prim = [], []
        self.prim_normals, self.prim_normal_offset, self.prim_material = [], [], []
        self.n_nodes = self.n_refs = self.n_prims = self.n_normal_doubles = 0

    # -- interning -------------------------------------------------------------------
    def sampler_id(self, sampler):
        if sampler is None:
            return -1
        if isinstance(sampler, type):
            raise TypeError("a sampler class was given where an instance is needed")
        key = id(sampler)
        if key not in self.sampler_ids:
            self.sampler_ids[key] = len(self.samplers)
            self.samplers.append(sampler)
        return self.sampler_ids[key]

    def texture_id(self, texture):
        if texture is None:
            return -1
        key = id(texture)
        if key not in self.texture_ids:
            rec = texture.device_record()
            pixels = numpy.ascontiguousarray(rec["pixels"], dtype=numpy.uint8)
            t = capi.Texture()
            t.height, t.width = pixels.shape[:2]
            t.hres, t.vres = rec["hres"], rec["vres"]
            t.mapping = rec["mapping"]
            t.offset = self.n_texel_bytes
            self.texels.append(pixels.reshape(-1))
            self.n_texel_bytes += pixels.size
            self.texture_ids[key] = len(self.textures)
            self.textures.append(t)
        return self.texture_ids[key]

    def material_id(self, material):
        if material is None:
            raise ValueError("an object in the scene has no material")
        key = id(material)
        if key not in self.material_ids:
            rec = material.device_record()
            m = capi.Material()
            m.kind = rec["kind"]
            m.receives_shadow = int(bool(rec.get("receives_shadow", True)))
            m.sampler = self.sampler_id(rec.get("sampler"))
            m.texture = self.texture_id(rec.get("texture"))
            for field in ("ambient", "diffuse", "specular", "reflect", "emit", "filter_in", "filter_out"):
                if field in rec:
                    _fill(getattr(m, field), rec[field])
            for field in ("exp", "reflect_exp", "kt", "eta_front", "eta_back"):
                setattr(m, field, rec.get(field, 0.0))
            self.material_ids[key] = len(self.materials)
            self.materials.append(m)
        return self.material_ids[key]

    # -- trees -----------------------------------------------------------------------
    def add_tree(self, tree):
        st = tree.stats()
        if st["depth"] + 2 > MAX_TREE_DEPTH:
            raise ValueError("kd-tree depth %d exceeds the device stack bound" % st["depth"])
        boxes = tree.boxes
        n_prims = len(boxes)
        smooth = any(getattr(b.obj, "n0", None) is not None for b in boxes)
        t = capi.Tree()
        bb = tree.bounding_box
        _fill(t.bbox_lo, (bb.x0, bb.y0, bb.z0))
        _fill(t.bbox_hi, (bb.x1, bb.y1, bb.z1))
        t.node_offset, t.ref_offset, t.prim_offset = self.n_nodes, self.n_refs, self.n_prims
        t.smooth, t.depth = int(smooth), st["depth"]
        t.walk_depth = st["depth"]

        # per-primitive tables
        v0 = numpy.array([b.obj.v0 for b in boxes], dtype=numpy.float64).reshape(-1, 3)
        v1 = numpy.array([b.obj.v1 for b in boxes], dtype=numpy.float64).reshape(-1, 3)
        v2 = numpy.array([b.obj.v2 for b in boxes], dtype=numpy.float64).reshape(-1, 3)
        width = 9 if smooth else 3
        normals = numpy.empty((n_prims, width))
        flags = numpy.zeros(n_prims, dtype=numpy.int32)
        mats = numpy.empty(n_prims, dtype=numpy.int32)
        mat_cache = {}
        for i, b in enumerate(boxes):
            tri = b.obj
            if not isinstance(tri, G.Triangle):
                raise NotImplementedError("kd-tree leaves must be triangles")
            if smooth:
                if tri.n0 is None:
                    raise NotImplementedError("a tree mixes flat and smooth triangles")
                normals[i, 0:3], normals[i, 3:6], normals[i, 6:9] = tri.n0, tri.n1, tri.n2
            else:
                normals[i] = tri.normal
            if not tri.cast_shadow:
                flags[i] = NO_SHADOW_BIT
            key = id(tri.material)
            if key not in mat_cache:
                mat_cache[key] = self.material_id(tri.material)
            mats[i] = mat_cache[key]
        t.shadow_casters = (capi.CASTERS_ALL if not flags.any() else
                            capi.CASTERS_NONE if flags.all() else capi.CASTERS_MIXED)
        self.prim_normals.append(normals.ravel())
        self.prim_normal_offset.append(self.n_normal_doubles + width * numpy.arange(n_prims, dtype=numpy.int64))
        self.prim_material.append(mats)

        # leaf-ordered geometry: v0, v1-v0, v2-v0 per leaf reference
        refs = tree.leaf_refs
        geom = numpy.empty((len(refs), 9))
        geom[:, 0:3] = v0[refs]
        geom[:, 3:6] = v1[refs] - v0[refs]
        geom[:, 6:9] = v2[refs] - v0[refs]
        self.tri_geom.append(geom)
        self.tri_prim.append((refs.astype(numpy.int32) | flags[refs]).astype(numpy.int32))

        # nodes (preorder; left child = next node)
        nodes = numpy.zeros(len(tree.axis), dtype=[("split", "<f8"), ("a", "<i4"), ("b", "<i4")])
        interior = tree.axis >= 0
        nodes["split"] = tree.split
        nodes["a"][interior] = tree.right[interior] + self.n_nodes
        nodes["b"][interior] = tree.axis[interior]
        if not numpy.all(tree.left[interior] == numpy.nonzero(interior)[0] + 1):
            raise AssertionError("kd nodes are not in preorder")
        leaf = ~interior
        nodes["a"][leaf] = tree.first[leaf] + self.n_refs
        nodes["b"][leaf] = 3 | (tree.count[leaf] << 2)
        self.kd_nodes.append(nodes)

        self.n_nodes += len(nodes)
        self.n_refs += len(refs)
        self.n_prims += n_prims
        self.n_normal_doubles += normals.size
        self.trees.append(t)
        return len(self.trees) - 1

    # -- objects ---------------------------------------------------------------------
    def add_object(self, obj):
        o = capi.Object()
        o.material = o.alt_material = o.tree = -1
        o.cast_shadow = 1
        target = obj
        if isinstance(obj, G.Instance):
            target = obj.obj
            o.has_transform = 1
            _fill(o.inv_point, numpy.asarray(obj.inv_transform)[:3, :4])
            _fill(o.inv_dir, numpy.asarray(obj.inv_normal_transform)[:3, :3])
            # an Instance of an Instance (geometry.py:85-112 wraps any object): the further
            # levels, from the outside in
            o.inner_offset = len(self.transforms)
            while isinstance(target, G.Instance):
                t = capi.Transform()
                _fill(t.inv_point, numpy.asarray(target.inv_transform)[:3, :4])
                _fill(t.inv_dir, numpy.asarray(target.inv_normal_transform)[:3, :3])
                self.transforms.append(t)
                o.n_inner += 1
                target = target.obj
        if isinstance(target, K.KDTree):
            o.kind = capi.KDTREE
            o.tree = self.add_tree(target)
            # KDTree.shadow_hit has no cast_shadow test of its own (kdtree.py:157-164): a flag set
            # on the tree object changes nothing, only its triangles' flags count
            o.cast_shadow = 1
            return o
        rec = target.device_record()
        o.kind = rec["kind"]
        o.cast_shadow = int(bool(getattr(target, "cast_shadow", True)))
        o.material = self.material_id(rec["material"])
        k = rec["kind"]
        if k == G.SPHERE:
            p = list(rec["center"]) + [rec["radius"], rec["normal_sign"]]
        elif k == G.PLANE:
            p = list(rec["point"]) + list(rec["normal"])
        elif k == G.CHECKER_PLANE:
            p = list(rec["point"]) + list(rec["normal"]) + list(rec["right"]) + list(rec["up"]) + [rec["grid_size"]]
            o.alt_material = self.material_id(rec["alt_material"])
        elif k == G.RECTANGLE:
            p = (list(rec["point"]) + list(rec["normal"]) + list(rec["right"]) + list(rec["top"])
                 + [rec["a"], rec["b"]])
        elif k == G.BOX:
            p = list(rec["lo"]) + list(rec["hi"])
        elif k == G.TRIANGLE:
            tri = rec["triangle"]
            p = list(tri.v0) + list(tri.v1) + list(tri.v2) + list(tri.normal)
            if tri.n0 is not None:
                o.smooth = 1
                p += list(tri.n0) + list(tri.n1) + list(tri.n2)
        else:
            raise NotImplementedError("object kind %r" % k)
        _fill(o.p, p)
        return o

    # -- lights ----------------------------------------------------------------------
    def add_light(self, light):
        rec = light.device_record()
        l = capi.Light()
        l.kind = rec["kind"]
        l.cast_shadow = int(rec.get("cast_shadow", False))
        l.sampler = self.sampler_id(rec.get("sampler"))
        _fill(l.radiance, rec["radiance"])
        if l.kind == LT.POINT:
            _fill(l.position, rec["position"])
            l.attenuation, l.radius = rec["attenuation"], rec["radius"]
        elif l.kind == LT.DIRECTIONAL:
            _fill(l.position, rec["direction"])
        elif l.kind == LT.AREA:
            _fill(l.position, rec["corner"])
            _fill(l.edge_u, rec["edge_u"])
            _fill(l.edge_v, rec["edge_v"])
            _fill(l.normal, rec["normal"])
            l.radius, l.pdf = rec["radius"], rec["inv_pdf_area"]
        return l

    # -- samplers --------------------------------------------------------------------
    def sampler_tables(self):
        recs, chunks, offset = [], [], 0
        for s in self.samplers:
            r = capi.Sampler()
            r.regular = int(isinstance(s, S.RegularSampler))
            r.num_patterns = len(s.patterns)
            r.samples_per_pattern = len(s.patterns[0])
            square = numpy.array(s.patterns, dtype=numpy.float64).reshape(-1)
            disk = numpy.array(s.concentric_patterns, dtype=numpy.float64).reshape(-1)
            hemi = numpy.array(s.hemisphere_patterns, dtype=numpy.float64).reshape(-1)
            n = r.num_patterns * r.samples_per_pattern
            assert square.size == 2 * n and disk.size == 2 * n and hemi.size == 3 * n
            r.square_offset, r.disk_offset, r.hemi_offset = offset, offset + 2 * n, offset + 4 * n
            offset += 7 * n
            chunks += [square, disk, hemi]
            recs.append(r)
        data = numpy.concatenate(chunks) if chunks else numpy.zeros(0)
        return recs, data


def _cat(chunks, dtype):
    return numpy.concatenate(chunks).astype(dtype) if chunks else numpy.zeros(0, dtype=dtype)


def flatten_world(world):
    """Build the FlatScene for a world.  Raises NotImplementedError for features that
    are outside the device hot path (textures, dielectrics, brute-force meshes)."""
    fl = _Flattener(world)
    scene = FlatScene()

    tracer = world.tracer
    kind = getattr(tracer, "KIND", None)
    if kind == "shade":
        scene.tracer_kind = capi.TRACER_SHADE
    elif kind == "area":
        scene.tracer_kind = capi.TRACER_AREA
    elif kind == "path":
        scene.tracer_kind = capi.TRACER_PATH
    elif kind == "whitted":
        scene.tracer_kind = capi.TRACER_WHITTED
    else:
        raise NotImplementedError("tracer %r is outside the device hot path" % type(tracer).__name__)
    scene.depth_limit = int(getattr(tracer, "depth_limit", 0))
    scene.background = numpy.array(world.background_color, dtype=numpy.float64)

    # the view-plane sampler always gets index 0 so a camera can refer to it
    fl.sampler_id(world.viewplane.sampler)

    objects = [fl.add_object(o) for o in world.objects]
    lights = [fl.add_light(l) for l in world.lights]

    amb = world.ambient_color.device_record()
    scene.ambient.kind = amb["kind"]
    scene.ambient.sampler = fl.sampler_id(amb.get("sampler"))
    _fill(scene.ambient.radiance, amb["radiance"])
    if "occluded" in amb:
        _fill(scene.ambient.occluded, amb["occluded"])

    for m in fl.materials:
        if scene.tracer_kind == capi.TRACER_PATH and m.kind not in (M.MATTE, M.EMISSIVE):
            raise NotImplementedError("PathTracer needs path_shade, which only Matte and Emissive have")
        if m.kind == M.DIELECTRIC and scene.tracer_kind in (capi.TRACER_SHADE, capi.TRACER_PATH):
            # Dielectric.shade passes a third argument that only WhittedTracer.trace_ray takes
            # (material.py:459 vs tracer.py:26, :43)
            raise TypeError("Dielectric needs a WhittedTracer")
        if (m.kind in (M.TRANSPARENT, M.DIELECTRIC) and scene.tracer_kind != capi.TRACER_AREA
                and scene.depth_limit + 1 > capi.MAX_BRANCH_LEVELS):
            raise NotImplementedError("two-child materials are traced to depth_limit <= %d on the device"
                                      % (capi.MAX_BRANCH_LEVELS - 1))
        if m.kind == M.SV_MATTE and scene.tracer_kind in (capi.TRACER_AREA, capi.TRACER_PATH):
            # SV_Matte has neither area_light_shade nor path_shade (material.py:292-326)
            raise AttributeError("SV_Matte only has shade(): it needs a Tracer or a WhittedTracer")
        samples_lobe = m.kind == M.GLOSSY_REFLECTIVE or (scene.tracer_kind == capi.TRACER_PATH and m.kind == M.MATTE)
        if samples_lobe and m.sampler < 0:
            # the reference fails here too: `None.sample_unit_hemisphere_surface` (material.py:47, :110)
            raise ValueError("a material that samples a direction was built without a sampler")

    sampler_recs, sampler_data = fl.sampler_tables()

    def pack(records, cls):
        return _bytes_of((cls * max(len(records), 1))(*records))[:len(records) * C.sizeof(cls)]

    a = scene.arrays
    a["objects"] = pack(objects, capi.Object)
    a["materials"] = pack(fl.materials, capi.Material)
    a["lights"] = pack(lights, capi.Light)
    a["samplers"] = pack(sampler_recs, capi.Sampler)
    a["sampler_data"] = sampler_data
    a["trees"] = pack(fl.trees, capi.Tree)
    a["kd_nodes"] = (numpy.concatenate(fl.kd_nodes).view(numpy.uint8) if fl.kd_nodes
                     else numpy.zeros(0, numpy.uint8))
    # struct of arrays over all leaf references of all trees: [9][n_refs]
    a["tri_geom"] = (numpy.ascontiguousarray(numpy.concatenate(fl.tri_geom, axis=0).T).ravel()
                     if fl.tri_geom else numpy.zeros(0, numpy.float64))
    a["tri_prim"] = _cat(fl.tri_prim, numpy.int32)
    a["prim_normals"] = _cat(fl.prim_normals, numpy.float64)
    a["prim_normal_offset"] = _cat(fl.prim_normal_offset, numpy.int64)
    a["prim_material"] = _cat(fl.prim_material, numpy.int32)
    a["textures"] = pack(fl.textures, capi.Texture)
    a["texels"] = _cat(fl.texels, numpy.uint8)
    a["transforms"] = pack(fl.transforms, capi.Transform)
    scene.sampler_objects = fl.samplers
    scene.n_trees = len(fl.trees)
    return scene


def camera_record(world, camera=None):
    """rtamd_camera for the world's camera and view plane."""
    cam = camera or world.camera
    vp = world.viewplane
    c = capi.Camera()
    _fill(c.eye, cam.eye)
    _fill(c.u, cam.u)
    _fill(c.v, cam.v)
    _fill(c.w, cam.w)
    c.viewing_distance = cam.viewing_distance
    c.lens_radius = getattr(cam, "lens_radius", 0.0)
    c.focal_plane_distance = getattr(cam, "focal_plane_distance", 0.0)
    c.pixel_size = vp.pixel_size
    c.thin_lens = int(getattr(cam, "IS_THIN_LENS", False))
    c.width, c.height = int(vp.resolution[0]), int(vp.resolution[1])
    c.sampler = 0
    return c
