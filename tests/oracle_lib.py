"""ctypes binding of oracle/liboracle.so (test infrastructure: the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")

_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build_if_needed():
    src = os.path.join(ROOT, "oracle", "bdpt_oracle.cpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True)


def _cam22(cam):
    """The oracle's camera layout is 22 doubles; shorter arrays (no defocus fields) are padded with zeros."""
    out = np.zeros(22)
    cam = np.asarray(cam, dtype=np.float64)
    out[:len(cam)] = cam
    return out


class OracleScene:
    def __init__(self, lib, flat):
        self.lib = lib
        self.flat = flat
        self._v = flat.vertex_array()
        self._tm = flat.tri_material_array()
        self._mt = flat.material_types()
        self._mp = flat.material_params()
        self._li = flat.light_array()
        if len(self._li) == 0:
            self._li = np.zeros(1, dtype=np.uint32)
        if hasattr(flat, "objects"):
            self.h = self._create_composite(lib, flat)
        else:
            self.h = lib.oracle_scene_create(self._v, self._tm, flat.n_triangles, self._mt, self._mp, len(flat.materials),
                                             self._li, len(flat.lights))
        if not self.h:
            raise RuntimeError("oracle_scene_create failed")

    def _create_composite(self, lib, flat):
        """A composite scene (instances, textures, media).  The bytes of image textures are input (flat.image_bytes: what
        the reference's loader made of the files); the perlin tables are built by the oracle itself."""
        tex_i = np.zeros((max(len(flat.textures), 1), 6), dtype=np.int32)
        tex_d = np.zeros((max(len(flat.textures), 1), 4))
        images = []
        k = 0
        for i, t in enumerate(flat.textures):
            w = h = 0
            if t.type == 3:
                img = np.ascontiguousarray(flat.image_bytes[k], dtype=np.uint8)
                k += 1
                h, w = img.shape[0], img.shape[1]
                images.append(img.reshape(-1))
            tex_i[i] = [t.type, t.even, t.odd, w, h, 0]
            tex_d[i] = [t.rgb[0], t.rgb[1], t.rgb[2], t.scale]
        img_all = np.concatenate(images) if images else np.zeros(1, dtype=np.uint8)
        obj_i = np.zeros((max(len(flat.objects), 1), 4), dtype=np.int32)
        obj_d = np.zeros((max(len(flat.objects), 1), 5))
        for i, ob in enumerate(flat.objects):
            obj_i[i] = [ob.first_triangle, ob.n_triangles, ob.flags, ob.phase_material]
            obj_d[i] = [ob.rotate_y_degrees, ob.offset[0], ob.offset[1], ob.offset[2], ob.density]
        mat_tex = np.array([m.texture for m in flat.materials], dtype=np.int32)
        return lib.oracle_scene_create_composite(self._v, self._tm, flat.n_triangles, self._mt, self._mp, mat_tex,
                                                 len(flat.materials), tex_i, tex_d, len(flat.textures), img_all,
                                                 sum(len(a) for a in images), obj_i, obj_d, len(flat.objects), self._li,
                                                 len(flat.lights))

    def world_leaves(self):
        """Leaves of the world's tree in visiting order: triangle id, or -(object + 2)."""
        cap = self.flat.n_triangles + len(getattr(self.flat, "objects", [])) + 4
        out = np.zeros(cap, dtype=np.int32)
        n = self.lib.oracle_world_leaves(self.h, out, cap)
        return out[:n].copy()

    def world_boxes(self):
        cap = 2 * (self.flat.n_triangles + len(getattr(self.flat, "objects", []))) + 4
        out = np.zeros((cap, 6))
        n = self.lib.oracle_world_boxes(self.h, out, cap)
        return out[:n].copy()

    def perlin_tables(self):
        n = sum(1 for t in self.flat.textures if t.type == 2)
        out = np.zeros((max(n, 1), 1536))
        assert self.lib.oracle_perlin_tables(self.h, out) == n
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_scene_destroy(self.h)
            self.h = None

    def node_count(self):
        return self.lib.oracle_bvh_node_count(self.h)

    def dfs_leaves(self):
        cap = 2 * self.flat.n_triangles + 4
        out = np.zeros(cap, dtype=np.uint32)
        n = self.lib.oracle_bvh_dfs_leaves(self.h, out, cap)
        return out[:n].copy()

    def node_boxes(self):
        out = np.zeros((self.node_count(), 6))
        self.lib.oracle_bvh_boxes(self.h, out)
        return out

    def triangle_data(self):
        out = np.zeros((self.flat.n_triangles, 16))
        self.lib.oracle_triangle_data(self.h, out)
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        out = np.zeros((len(rays), 11))
        stats = np.zeros(7, dtype=np.uint64)
        self.lib.oracle_trace_closest(self.h, rays, len(rays), out, stats)
        return out, stats

    def visible(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.float64)
        out = np.zeros(len(pairs))
        self.lib.oracle_visible(self.h, pairs, len(pairs), out)
        return out.astype(np.uint8)

    def render(self, cam15, W, sqrt_spp, max_depth, seed, rect, per_sample=False):
        x0, y0, x1, y1 = [int(v) for v in rect]
        fb = np.zeros((y1 - y0, x1 - x0, 3))
        stats = np.zeros(10, dtype=np.uint64)
        n_s = (y1 - y0) * (x1 - x0) * sqrt_spp * sqrt_spp
        smp = np.zeros((n_s, 4)) if per_sample else None
        self.lib.oracle_render(self.h, _cam22(cam15), W, sqrt_spp, max_depth,
                               int(seed), x0, y0, x1, y1, fb, smp.ctypes.data if per_sample else None, stats)
        return fb, stats, smp

    def render_unidirectional(self, cam15, W, sqrt_spp, max_depth, seed, rect):
        x0, y0, x1, y1 = [int(v) for v in rect]
        fb = np.zeros((y1 - y0, x1 - x0, 3))
        stats = np.zeros(10, dtype=np.uint64)
        self.lib.oracle_render_unidirectional(self.h, _cam22(cam15), W, sqrt_spp, max_depth,
                                              int(seed), x0, y0, x1, y1, fb, stats)
        return fb, stats

    def render_serial_mt(self, cam15, W, H, sqrt_spp, max_depth):
        fb = np.zeros((H, W, 3))
        stats = np.zeros(10, dtype=np.uint64)
        self.lib.oracle_render_serial_mt(self.h, _cam22(cam15), W, H, sqrt_spp,
                                         max_depth, fb, stats)
        return fb, stats


class Oracle:
    def __init__(self):
        build_if_needed()
        lib = C.CDLL(LIB)
        lib.oracle_scene_create.restype = C.c_void_p
        lib.oracle_scene_create.argtypes = [_d, _i32, C.c_uint32, _i32, _d, C.c_uint32, _u32, C.c_uint32]
        lib.oracle_scene_create_composite.restype = C.c_void_p
        lib.oracle_scene_create_composite.argtypes = [_d, _i32, C.c_uint32, _i32, _d, _i32, C.c_uint32, _i32, _d, C.c_uint32,
                                                      _u8, C.c_uint64, _i32, _d, C.c_uint32, _u32, C.c_uint32]
        lib.oracle_world_leaves.restype = C.c_uint32
        lib.oracle_world_leaves.argtypes = [C.c_void_p, _i32, C.c_uint32]
        lib.oracle_world_boxes.restype = C.c_uint32
        lib.oracle_world_boxes.argtypes = [C.c_void_p, _d, C.c_uint32]
        lib.oracle_perlin_tables.restype = C.c_uint32
        lib.oracle_perlin_tables.argtypes = [C.c_void_p, _d]
        lib.oracle_scene_destroy.argtypes = [C.c_void_p]
        lib.oracle_bvh_node_count.restype = C.c_uint32
        lib.oracle_bvh_node_count.argtypes = [C.c_void_p]
        lib.oracle_bvh_dfs_leaves.restype = C.c_uint32
        lib.oracle_bvh_dfs_leaves.argtypes = [C.c_void_p, _u32, C.c_uint32]
        lib.oracle_bvh_boxes.argtypes = [C.c_void_p, _d]
        lib.oracle_triangle_data.argtypes = [C.c_void_p, _d]
        lib.oracle_camera_initialize.argtypes = [_d, C.c_int32, C.c_int32, _d, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.oracle_trace_closest.argtypes = [C.c_void_p, _d, C.c_uint64, _d, _u64]
        lib.oracle_visible.argtypes = [C.c_void_p, _d, C.c_uint64, _d]
        lib.oracle_render.argtypes = [C.c_void_p, _d, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, _d, C.c_void_p, _u64]
        lib.oracle_render_unidirectional.argtypes = [C.c_void_p, _d, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                                     C.c_int32, C.c_int32, C.c_int32, _d, _u64]
        lib.oracle_render_serial_mt.argtypes = [C.c_void_p, _d, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _d, _u64]
        lib.oracle_colors_to_rgb8.argtypes = [_d, C.c_int32, C.c_int32, C.c_int32, _u8]
        lib.oracle_sincos_array.argtypes = [_d, C.c_uint64, _d, _d]
        lib.oracle_pow5_array.argtypes = [_d, C.c_uint64, _d]
        lib.oracle_log_array.argtypes = [_d, C.c_uint64, _d]
        lib.oracle_rng_array.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _d]
        lib.oracle_mt_array.argtypes = [C.c_uint64, _d]
        self.lib = lib

    def scene(self, flat):
        return OracleScene(self.lib, flat)

    def camera_initialize(self, cam):
        """camera::initialize on the public fields -> (22 doubles: center pixel00 du dv background disk_u disk_v
        defocus_angle, image_height, sqrt_spp)"""
        pub = np.array([*cam.lookfrom, *cam.lookat, *cam.vup, cam.vfov, cam.aspect_ratio, cam.focus_dist,
                        cam.defocus_angle], dtype=np.float64)
        out = np.zeros(19)
        h, s = C.c_int32(0), C.c_int32(0)
        self.lib.oracle_camera_initialize(pub, cam.image_width, cam.samples_per_pixel, out, C.byref(h), C.byref(s))
        return np.concatenate([out[:12], np.array(cam.background, dtype=np.float64), out[12:19]]), h.value, s.value

    def colors_to_rgb8(self, fb, spp):
        fb = np.ascontiguousarray(fb, dtype=np.float64)
        H, W = fb.shape[0], fb.shape[1]
        out = np.zeros(H * W * 3, dtype=np.uint8)
        self.lib.oracle_colors_to_rgb8(fb, W, H, spp, out)
        return out

    def sincos(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        s, c = np.zeros_like(x), np.zeros_like(x)
        self.lib.oracle_sincos_array(x, len(x), s, c)
        return s, c

    def pow5(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        o = np.zeros_like(x)
        self.lib.oracle_pow5_array(x, len(x), o)
        return o

    def log(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        o = np.zeros_like(x)
        self.lib.oracle_log_array(x, len(x), o)
        return o

    def rng(self, seed, pixel, sample, n):
        o = np.zeros(n)
        self.lib.oracle_rng_array(seed, pixel, sample, n, o)
        return o

    def mt(self, n):
        o = np.zeros(n)
        self.lib.oracle_mt_array(n, o)
        return o
