"""ctypes binding of libbdpt_amd.so (the C ABI declared in include/bdpt_amd.h).

There is no CPU fallback: if the library is missing or cannot be loaded this module raises,
and every call that fails raises :class:`BdptError` with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BDPT_LIB") or os.path.join(_HERE, "libbdpt_amd.so")     # (BDPT_LIB: another build of the same library, for A/B timing)

BDPT_RENDER_COUNT_STATS = 1     # counters with the reference's BvhStats semantics
BDPT_RENDER_COUNT_WORK = 2      # counters of the work the production path does
BDPT_RENDER_TIME_KERNEL = 4     # HIP events around the render kernels
BDPT_RENDER_TIME_STAGES = 16    # HIP event pair around every traversal launch
BDPT_RENDER_UNIDIRECTIONAL = 32  # camera::path_trace_color instead of bidirectional_color
BDPT_RENDER_MEGAKERNEL = 8      # single persistent kernel instead of the wavefront pipeline


class BdptError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("rgb", C.c_double * 3), ("fuzz", C.c_double),
                ("ior", C.c_double)]


class PerlinTables(C.Structure):
    _fields_ = [("randvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("reserved", C.c_int32), ("rgb", C.c_double * 3), ("scale", C.c_double), ("data", C.c_void_p)]


class SceneObject(C.Structure):
    _fields_ = [("first_triangle", C.c_uint32), ("n_triangles", C.c_uint32), ("flags", C.c_uint32),
                ("phase_material", C.c_int32), ("rotate_y_degrees", C.c_double), ("offset", C.c_double * 3),
                ("density", C.c_double)]


class CameraPublic(C.Structure):
    _fields_ = [("lookfrom", C.c_double * 3), ("lookat", C.c_double * 3), ("vup", C.c_double * 3),
                ("vfov", C.c_double), ("aspect_ratio", C.c_double), ("focus_dist", C.c_double),
                ("defocus_angle", C.c_double), ("background", C.c_double * 3), ("image_width", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32), ("reserved", C.c_int32)]


class CameraState(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("pixel00_loc", C.c_double * 3), ("pixel_delta_u", C.c_double * 3),
                ("pixel_delta_v", C.c_double * 3), ("background", C.c_double * 3), ("image_width", C.c_int32),
                ("image_height", C.c_int32), ("sqrt_spp", C.c_int32), ("max_depth", C.c_int32),
                ("defocus_disk_u", C.c_double * 3), ("defocus_disk_v", C.c_double * 3), ("defocus_angle", C.c_double)]

    def as_array(self) -> np.ndarray:
        """center pixel00 du dv background disk_u disk_v defocus_angle as 22 doubles (the layout the oracle takes)."""
        return np.array([*self.center, *self.pixel00_loc, *self.pixel_delta_u, *self.pixel_delta_v, *self.background,
                         *self.defocus_disk_u, *self.defocus_disk_v, self.defocus_angle])


class RenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("shard_rows", C.c_int32), ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_traced", "aabb_tests", "aabb_hits", "bvh_node_visits", "triangle_tests",
                                          "triangle_hits", "shadow_rays", "connection_pairs", "rng_draws", "samples",
                                          "path_vertices")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_materials", C.c_uint32), ("n_lights", C.c_uint32),
                ("n_nodes", C.c_uint32), ("max_depth", C.c_uint32), ("reserved", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class SceneFileView(C.Structure):
    _fields_ = [("camera", CameraPublic), ("output_name", C.c_char_p), ("tri_vertices", C.POINTER(C.c_double)),
                ("tri_material", C.POINTER(C.c_int32)), ("materials", C.POINTER(Material)),
                ("light_triangles", C.POINTER(C.c_uint32)), ("n_triangles", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_lights", C.c_uint32), ("reserved", C.c_uint32)]


# every symbol include/bdpt_amd.h declares
EXPORTED_SYMBOLS = [
    "bdpt_last_error", "bdpt_context_create", "bdpt_context_destroy", "bdpt_camera_initialize", "bdpt_scene_create",
    "bdpt_scene_destroy", "bdpt_scene_get_info", "bdpt_scene_dfs_leaves", "bdpt_scene_node_boxes", "bdpt_trace_closest",
    "bdpt_trace_closest_device", "bdpt_visible", "bdpt_render", "bdpt_render_device", "bdpt_render_get_stats",
    "bdpt_render_get_kernel_ms", "bdpt_render_get_stage_ms",
    "bdpt_colors_to_rgb8", "bdpt_colors_to_rgb8_device", "bdpt_device_sincos", "bdpt_device_random_doubles",
    "bdpt_device_arith", "bdpt_scene_file_load", "bdpt_scene_file_destroy", "bdpt_scene_file_get", "bdpt_write_png", "bdpt_debug_yaml_to_json",
    "bdpt_reduce_framebuffer", "bdpt_context_set_workspace_limit", "bdpt_context_set_query_walker",
    "bdpt_render_device_static", "bdpt_scene_create_composite", "bdpt_debug_tie_counts",
]

WALKER_PIPELINE = 0          # traversal kernels of the render pipeline (default)
WALKER_REFERENCE_TREE = 1    # stackless walk over the reference's own tree (verification)

_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64 (same soname as ROCm's).  Two HIP runtimes in one process
    # cannot both own the GPU, so when torch is going to be used (device buffers, RCCL) it has to be
    # loaded first: this library then binds to the runtime that is already there.
    if os.environ.get("BDPT_NO_TORCH", "") == "":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise BdptError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, dp, u64 = C.c_void_p, C.POINTER(C.c_double), C.c_uint64
    lib.bdpt_last_error.restype = C.c_char_p
    lib.bdpt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.bdpt_context_destroy.argtypes = [vp]
    lib.bdpt_camera_initialize.argtypes = [C.POINTER(CameraPublic), C.POINTER(CameraState)]
    lib.bdpt_scene_create.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Material), C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]
    lib.bdpt_scene_create_composite.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Material), C.c_uint32, C.POINTER(Texture), C.c_uint32,
                                                C.POINTER(SceneObject), C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]
    lib.bdpt_scene_destroy.argtypes = [vp]
    lib.bdpt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.bdpt_scene_dfs_leaves.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.bdpt_scene_node_boxes.argtypes = [vp, vp, C.c_uint32]
    lib.bdpt_trace_closest.argtypes = [vp, vp, vp, u64, vp, C.POINTER(Stats)]
    lib.bdpt_trace_closest_device.argtypes = [vp, vp, vp, u64, vp, vp]
    lib.bdpt_visible.argtypes = [vp, vp, vp, u64, vp]
    lib.bdpt_render.argtypes = [vp, vp, C.POINTER(CameraState), C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.bdpt_render_device.argtypes = [vp, vp, C.POINTER(CameraState), C.POINTER(RenderParams), vp, vp]
    lib.bdpt_render_device_static.argtypes = [vp, vp, C.POINTER(CameraState), C.POINTER(RenderParams), vp, vp]
    lib.bdpt_render_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.bdpt_render_get_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.bdpt_render_get_stage_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.bdpt_colors_to_rgb8.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.bdpt_colors_to_rgb8_device.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.bdpt_device_sincos.argtypes = [vp, vp, u64, vp, vp]
    lib.bdpt_device_random_doubles.argtypes = [vp, u64, u64, u64, u64, vp]
    lib.bdpt_device_arith.argtypes = [vp, C.c_int, vp, vp, u64, vp]
    lib.bdpt_scene_file_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.bdpt_scene_file_destroy.argtypes = [vp]
    lib.bdpt_scene_file_get.argtypes = [vp, C.POINTER(SceneFileView)]
    lib.bdpt_write_png.argtypes = [C.c_char_p, vp, C.c_int32, C.c_int32]
    lib.bdpt_reduce_framebuffer.argtypes = [vp, vp, C.c_uint64, vp]
    lib.bdpt_debug_yaml_to_json.argtypes = [C.c_char_p, C.c_char_p, u64]
    lib.bdpt_context_set_workspace_limit.argtypes = [vp, u64]
    lib.bdpt_context_set_query_walker.argtypes = [vp, C.c_int]
    lib.bdpt_debug_tie_counts.argtypes = [vp, vp, C.c_int]
    for name in EXPORTED_SYMBOLS:
        if name != "bdpt_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def reduce_framebuffer(rccl_comm: int, d_framebuffer: int, n_doubles: int, stream: int = 0) -> None:
    """Sum of the ranks' framebuffers in place (one all-reduce); `rccl_comm` is an ncclComm_t as an integer."""
    _check(load_library().bdpt_reduce_framebuffer(C.c_void_p(rccl_comm), C.c_void_p(d_framebuffer), n_doubles, C.c_void_p(stream)))


def _check(rc: int) -> None:
    if rc != 0:
        raise BdptError("bdpt error %d: %s" % (rc, load_library().bdpt_last_error().decode()))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def camera_public_from(cam) -> CameraPublic:
    """From a scenes.Camera (public fields of /root/reference/src/camera.h:26-41)."""
    p = CameraPublic()
    p.lookfrom[:] = [float(v) for v in cam.lookfrom]
    p.lookat[:] = [float(v) for v in cam.lookat]
    p.vup[:] = [float(v) for v in cam.vup]
    p.vfov, p.aspect_ratio, p.focus_dist, p.defocus_angle = cam.vfov, cam.aspect_ratio, cam.focus_dist, cam.defocus_angle
    p.background[:] = [float(v) for v in cam.background]
    p.image_width, p.samples_per_pixel, p.max_depth = cam.image_width, cam.samples_per_pixel, cam.max_depth
    return p


def camera_initialize(cam) -> CameraState:
    st = CameraState()
    pub = cam if isinstance(cam, CameraPublic) else camera_public_from(cam)
    _check(load_library().bdpt_camera_initialize(C.byref(pub), C.byref(st)))
    return st


def load_scene_file(path: str):
    """load_scene_from_yaml through the library: returns a scenes.FlatScene (camera.file_name = output name)."""
    from . import scenes
    lib = load_library()
    h = C.c_void_p()
    _check(lib.bdpt_scene_file_load(os.fsencode(path), C.byref(h)))
    try:
        v = SceneFileView()
        _check(lib.bdpt_scene_file_get(h, C.byref(v)))
        flat = scenes.FlatScene()
        c = v.camera
        flat.camera = scenes.Camera(aspect_ratio=c.aspect_ratio, image_width=c.image_width, samples_per_pixel=c.samples_per_pixel,
                                    max_depth=c.max_depth, background=tuple(c.background), vfov=c.vfov, lookfrom=tuple(c.lookfrom),
                                    lookat=tuple(c.lookat), vup=tuple(c.vup), defocus_angle=c.defocus_angle,
                                    focus_dist=c.focus_dist, file_name=v.output_name.decode())
        for i in range(v.n_materials):
            m = v.materials[i]
            flat.materials.append(scenes.Material(int(m.type), tuple(m.rgb), float(m.fuzz), float(m.ior)))
        n = int(v.n_triangles)
        if n:
            flat.vertices = np.ctypeslib.as_array(v.tri_vertices, shape=(n, 9)).copy().tolist()
            flat.tri_material = np.ctypeslib.as_array(v.tri_material, shape=(n,)).copy().tolist()
        flat.lights = [int(v.light_triangles[i]) for i in range(v.n_lights)]
        return flat
    finally:
        lib.bdpt_scene_file_destroy(h)


def yaml_to_json(text: str):
    """Parse tree of the library's YAML reader (test hook): dicts, lists, strings, None."""
    import json
    buf = C.create_string_buffer(max(4096, 8 * len(text) + 64))
    _check(load_library().bdpt_debug_yaml_to_json(text.encode("utf-8"), buf, len(buf)))
    return json.loads(buf.value.decode("utf-8"))


def write_png(file_name: str, rgb: np.ndarray) -> None:
    """8-bit RGB PNG at ./output/<file_name> (src/image/wpng.h:38-89)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(load_library().bdpt_write_png(os.fsencode(file_name), _ptr(rgb), rgb.shape[1], rgb.shape[0]))


def colors_to_rgb8(fb: np.ndarray, spp: int) -> np.ndarray:
    fb = np.ascontiguousarray(fb, dtype=np.float64)
    H, W = fb.shape[0], fb.shape[1]
    out = np.zeros((H, W, 3), dtype=np.uint8)
    _check(load_library().bdpt_colors_to_rgb8(_ptr(fb), W, H, spp, _ptr(out)))
    return out


class Context:
    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.bdpt_context_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.bdpt_context_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_workspace_limit(self, n_bytes: int) -> None:
        """Most device memory the context may take for the per-sample state of a render (0: the default)."""
        _check(self.lib.bdpt_context_set_workspace_limit(self.h, int(n_bytes)))

    def set_query_walker(self, walker: int) -> None:
        """WALKER_PIPELINE / WALKER_REFERENCE_TREE for trace_closest (without counters) and visible."""
        _check(self.lib.bdpt_context_set_query_walker(self.h, int(walker)))

    # -- test hooks ---------------------------------------------------------------
    TIE_COUNT_NAMES = ("ties", "settled_by_face_bits", "settled_by_exact_plane", "left_to_boxes", "wrong_by_face_bits",
                       "wrong_by_exact_plane", "plain_hits", "plain_hits_unreached")

    def tie_counts(self, reset: bool = False) -> dict:
        """Counters of the cross-check build (BDPT_LIB=.../libbdpt_amd_tiecheck.so); raises BdptError on the normal one."""
        o = np.zeros(8, dtype=np.uint64)
        _check(self.lib.bdpt_debug_tie_counts(self.h, _ptr(o), 1 if reset else 0))
        return {n: int(v) for n, v in zip(self.TIE_COUNT_NAMES, o)}

    def device_sincos(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        s, c = np.zeros_like(x), np.zeros_like(x)
        _check(self.lib.bdpt_device_sincos(self.h, _ptr(x), x.size, _ptr(s), _ptr(c)))
        return s, c

    def device_random_doubles(self, seed, pixel, sample, n):
        o = np.zeros(n)
        _check(self.lib.bdpt_device_random_doubles(self.h, seed, pixel, sample, n, _ptr(o)))
        return o

    def device_arith(self, op, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        o = np.zeros_like(a)
        _check(self.lib.bdpt_device_arith(self.h, op, _ptr(a), _ptr(b), a.size, _ptr(o)))
        return o

    def scene(self, flat) -> "Scene":
        return Scene(self, flat)


class Scene:
    """bdpt_scene built from a scenes.FlatScene."""

    def __init__(self, ctx: Context, flat):
        self.ctx = ctx
        self.lib = ctx.lib
        self.flat = flat
        v = flat.vertex_array()
        tm = flat.tri_material_array()
        mats = (Material * max(1, len(flat.materials)))()
        for i, m in enumerate(flat.materials):
            mats[i].type = m.type
            mats[i].rgb[:] = [float(c) for c in m.rgb]
            mats[i].fuzz = m.fuzz
            mats[i].ior = m.ior
            mats[i].reserved = getattr(m, "texture", -1) + 1
        li = flat.light_array()
        h = C.c_void_p()
        if hasattr(flat, "objects"):
            self._create_composite(ctx, flat, v, tm, mats, li, h)
        else:
            _check(self.lib.bdpt_scene_create(ctx.h, _ptr(v), _ptr(tm), flat.n_triangles, mats, len(flat.materials),
                                              _ptr(li) if len(li) else None, len(li), C.byref(h)))
        self.h = h

    def _create_composite(self, ctx, flat, v, tm, mats, li, h):
        """bdpt_scene_create_composite from a scenes.CompositeScene whose perlin_tables / image_bytes hold what the
        reference's texture objects hold (one entry per noise / image texture, in texture order)."""
        texs = (Texture * max(1, len(flat.textures)))()
        keep = []                                        # buffers the texture records point to
        n_noise = n_image = 0
        for i, t in enumerate(flat.textures):
            texs[i].type = t.type
            texs[i].even, texs[i].odd = t.even, t.odd
            texs[i].rgb[:] = [float(c) for c in t.rgb]
            texs[i].scale = t.scale
            if t.type == 2:
                tab = np.ascontiguousarray(flat.perlin_tables[n_noise], dtype=np.float64).ravel()
                n_noise += 1
                pt = PerlinTables()
                C.memmove(pt.randvec, tab[:768].ctypes.data, 768 * 8)
                perms = np.ascontiguousarray(tab[768:1536].astype(np.int32))
                C.memmove(pt.perm_x, perms.ctypes.data, 256 * 4)
                C.memmove(pt.perm_y, perms.ctypes.data + 1024, 256 * 4)
                C.memmove(pt.perm_z, perms.ctypes.data + 2048, 256 * 4)
                keep.append(pt)
                texs[i].data = C.addressof(pt)
            elif t.type == 3:
                img = np.ascontiguousarray(flat.image_bytes[n_image], dtype=np.uint8)
                n_image += 1
                keep.append(img)
                texs[i].height, texs[i].width = img.shape[0], img.shape[1]
                texs[i].data = img.ctypes.data
        objs = (SceneObject * max(1, len(flat.objects)))()
        for k, ob in enumerate(flat.objects):
            objs[k].first_triangle, objs[k].n_triangles, objs[k].flags = ob.first_triangle, ob.n_triangles, ob.flags
            objs[k].phase_material = ob.phase_material
            objs[k].rotate_y_degrees = ob.rotate_y_degrees
            objs[k].offset[:] = [float(c) for c in ob.offset]
            objs[k].density = ob.density
        _check(self.lib.bdpt_scene_create_composite(ctx.h, _ptr(v), _ptr(tm), flat.n_triangles, mats, len(flat.materials), texs,
                                                    len(flat.textures), objs, len(flat.objects), _ptr(li) if len(li) else None,
                                                    len(li), C.byref(h)))

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.bdpt_scene_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def info(self) -> SceneInfo:
        i = SceneInfo()
        _check(self.lib.bdpt_scene_get_info(self.h, C.byref(i)))
        return i

    def dfs_leaves(self) -> np.ndarray:
        cap = 2 * self.flat.n_triangles + 4
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint32()
        _check(self.lib.bdpt_scene_dfs_leaves(self.h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def node_boxes(self) -> np.ndarray:
        n = self.info().n_nodes
        out = np.zeros((n, 6))
        _check(self.lib.bdpt_scene_node_boxes(self.h, _ptr(out), n))
        return out

    def trace_closest(self, rays, want_stats=False):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 8)
        hits = np.zeros((len(rays), 11))
        st = Stats()
        _check(self.lib.bdpt_trace_closest(self.ctx.h, self.h, _ptr(rays), len(rays), _ptr(hits),
                                           C.byref(st) if want_stats else None))
        return (hits, st) if want_stats else hits

    def visible(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(len(pairs), dtype=np.uint8)
        _check(self.lib.bdpt_visible(self.ctx.h, self.h, _ptr(pairs), len(pairs), _ptr(out)))
        return out

    @staticmethod
    def _flags(want_stats, time_kernel=False, megakernel=False, time_stages=False, unidirectional=False):
        """want_stats: False / True or "reference" (BvhStats semantics) / "work" (production path's own work)"""
        f = BDPT_RENDER_TIME_KERNEL if time_kernel else 0
        if time_stages:
            f |= BDPT_RENDER_TIME_STAGES
        if unidirectional:
            f |= BDPT_RENDER_UNIDIRECTIONAL
        if megakernel:
            f |= BDPT_RENDER_MEGAKERNEL
        if want_stats == "work":
            f |= BDPT_RENDER_COUNT_WORK
        elif want_stats:
            f |= BDPT_RENDER_COUNT_STATS
        return f

    def render(self, cam_state: CameraState, seed=0, rect=None, shard=(0, 1, 1), want_stats=False, framebuffer=None,
               megakernel=False, unidirectional=False):
        """Host-buffer render; returns the full W x H x 3 framebuffer of sums (unowned pixels keep
        the values of `framebuffer`, zeros by default) and optionally the counters."""
        W, H = cam_state.image_width, cam_state.image_height
        fb = np.zeros((H, W, 3)) if framebuffer is None else np.ascontiguousarray(framebuffer, dtype=np.float64)
        p = RenderParams()
        p.seed = seed
        if rect is not None:
            p.x0, p.y0, p.x1, p.y1 = [int(v) for v in rect]
        p.shard_index, p.shard_count, p.shard_rows = shard
        p.flags = self._flags(want_stats, megakernel=megakernel, unidirectional=unidirectional)
        st = Stats()
        _check(self.lib.bdpt_render(self.ctx.h, self.h, C.byref(cam_state), C.byref(p), _ptr(fb), C.byref(st)))
        return (fb, st) if want_stats else fb

    def render_device(self, cam_state: CameraState, d_framebuffer: int, seed=0, rect=None, shard=(0, 1, 1),
                      want_stats=False, stream: int = 0, time_kernel=False, megakernel=False, time_stages=False, unidirectional=False):
        """Framebuffer already in device memory (`d_framebuffer` = device pointer, e.g. tensor.data_ptr())."""
        p = RenderParams()
        p.seed = seed
        if rect is not None:
            p.x0, p.y0, p.x1, p.y1 = [int(v) for v in rect]
        p.shard_index, p.shard_count, p.shard_rows = shard
        p.flags = self._flags(want_stats, time_kernel, megakernel, time_stages, unidirectional)
        _check(self.lib.bdpt_render_device(self.ctx.h, self.h, C.byref(cam_state), C.byref(p), C.c_void_p(d_framebuffer),
                                           C.c_void_p(stream)))

    def render_device_static(self, cam_state: CameraState, d_framebuffer: int, seed=0, rect=None, shard=(0, 1, 1),
                             want_stats=False, stream: int = 0, time_kernel=False, unidirectional=False):
        """bdpt_render_device_static: the same render out of the library's static workspace, allocating nothing."""
        p = RenderParams()
        p.seed = seed
        if rect is not None:
            p.x0, p.y0, p.x1, p.y1 = [int(v) for v in rect]
        p.shard_index, p.shard_count, p.shard_rows = shard
        p.flags = self._flags(want_stats, time_kernel, unidirectional=unidirectional)
        _check(self.lib.bdpt_render_device_static(self.ctx.h, self.h, C.byref(cam_state), C.byref(p), C.c_void_p(d_framebuffer),
                                                  C.c_void_p(stream)))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.bdpt_render_get_kernel_ms(self.ctx.h, C.byref(ms)))
        return float(ms.value)

    def last_stage_ms(self, stage: int):
        """(sum of launch durations in ms, launches) of traversal stage 0 (closest hit) / 1 (visibility)"""
        ms, n = C.c_double(), C.c_uint64()
        _check(self.lib.bdpt_render_get_stage_ms(self.ctx.h, stage, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def last_stats(self) -> Stats:
        st = Stats()
        _check(self.lib.bdpt_render_get_stats(self.ctx.h, C.byref(st)))
        return st
