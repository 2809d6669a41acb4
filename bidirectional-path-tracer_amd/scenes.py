"""Procedural scene construction on flat tables (host side, numpy).

Mirrors the reference's procedural helpers so that scenes built here are bit-identical to
what the reference's C++ would build from the same arguments:

  * ``add_quad_triangles``  -> /root/reference/src/objects/primatives/triangle.h:232-241
  * ``add_box_triangles``   -> /root/reference/src/objects/primatives/triangle.h:243-309
  * ``add_uv_sphere``       -> /root/reference/src/scene/scene_loader.h:204-242
  * light duplication rule  -> /root/reference/src/scene/scene_loader.h:190-202

A scene is a :class:`FlatScene`: triangle soup (N x 9 float64), per-triangle material index,
a material table, the ordered list of emissive triangle ids (the reference's ``lights``
collection) and the public camera fields of /root/reference/src/camera.h:26-41.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, ISOTROPIC = 0, 1, 2, 3, 4
TEX_SOLID, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = 0, 1, 2, 3
OBJECT_ROTATE_Y, OBJECT_TRANSLATE, OBJECT_MEDIUM = 1, 2, 4

# /root/reference/src/main.h:20
PI = 3.1415926535897932385


@dataclass
class Material:
    type: int
    rgb: Sequence[float] = (0.0, 0.0, 0.0)
    fuzz: float = 0.0
    ior: float = 1.5
    texture: int = -1      # composite scenes: index of the texture that replaces `rgb` (-1: the solid colour)


@dataclass
class Camera:
    """Public camera fields (/root/reference/src/camera.h:26-41) with the reference's defaults."""
    aspect_ratio: float = 1.0
    image_width: int = 100
    samples_per_pixel: int = 50
    max_depth: int = 10
    background: Sequence[float] = (0.0, 0.0, 0.0)
    vfov: float = 90.0
    lookfrom: Sequence[float] = (0.0, 0.0, 0.0)
    lookat: Sequence[float] = (0.0, 0.0, -1.0)
    vup: Sequence[float] = (0.0, 1.0, 0.0)
    defocus_angle: float = 0.0
    focus_dist: float = 10.0
    file_name: str = "image.png"


@dataclass
class FlatScene:
    materials: List[Material] = field(default_factory=list)
    vertices: List[List[float]] = field(default_factory=list)     # 9 doubles per triangle
    tri_material: List[int] = field(default_factory=list)
    lights: List[int] = field(default_factory=list)
    camera: Camera = field(default_factory=Camera)

    # -- building -----------------------------------------------------------------
    def add_material(self, m: Material) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_triangle(self, v0, v1, v2, mat: int) -> None:
        """world.add(tri); emissive triangles are also appended to the light list
        (/root/reference/src/scene/scene_loader.h:190-202)."""
        self.vertices.append([float(v0[0]), float(v0[1]), float(v0[2]),
                              float(v1[0]), float(v1[1]), float(v1[2]),
                              float(v2[0]), float(v2[1]), float(v2[2])])
        self.tri_material.append(mat)
        if self.materials[mat].type == DIFFUSE_LIGHT:
            self.lights.append(len(self.tri_material) - 1)

    # -- arrays for the C interfaces --------------------------------------------------
    def vertex_array(self) -> np.ndarray:
        return np.ascontiguousarray(np.array(self.vertices, dtype=np.float64).reshape(-1, 9))

    def tri_material_array(self) -> np.ndarray:
        return np.ascontiguousarray(np.array(self.tri_material, dtype=np.int32))

    def light_array(self) -> np.ndarray:
        return np.ascontiguousarray(np.array(self.lights, dtype=np.uint32))

    def material_types(self) -> np.ndarray:
        return np.ascontiguousarray(np.array([m.type for m in self.materials], dtype=np.int32))

    def material_params(self) -> np.ndarray:
        return np.ascontiguousarray(np.array(
            [[m.rgb[0], m.rgb[1], m.rgb[2], m.fuzz, m.ior] for m in self.materials], dtype=np.float64))

    @property
    def n_triangles(self) -> int:
        return len(self.tri_material)

    # -- flat file consumed by the oracle tools (layout: oracle/flat_scene.h) ---------
    def to_bytes(self) -> bytes:
        out = [b"BDPTSCN1", struct.pack("<4I", len(self.materials), self.n_triangles, len(self.lights), 0)]
        for m in self.materials:
            out.append(struct.pack("<2i5d", m.type, 0, m.rgb[0], m.rgb[1], m.rgb[2], m.fuzz, m.ior))
        rec = np.zeros(self.n_triangles, dtype=[("v", "<f8", 9), ("m", "<i4"), ("pad", "<i4")])
        rec["v"] = self.vertex_array()
        rec["m"] = self.tri_material_array()
        out.append(rec.tobytes())
        lights = list(self.lights) + ([0] if len(self.lights) % 2 else [])
        out.append(struct.pack("<%dI" % len(lights), *lights))
        c = self.camera
        out.append(struct.pack("<16d4i", *c.lookfrom, *c.lookat, *c.vup, c.vfov, c.aspect_ratio, c.focus_dist,
                               c.defocus_angle, *c.background, c.image_width, c.samples_per_pixel, c.max_depth, 0))
        return b"".join(out)


def flat_scene_from_bytes(blob: bytes) -> FlatScene:
    """Inverse of FlatScene.to_bytes (layout: oracle/flat_scene.h)."""
    if blob[:8] != b"BDPTSCN1":
        raise ValueError("not a flat scene file")
    n_mat, n_tri, n_light, _ = struct.unpack_from("<4I", blob, 8)
    off = 24
    s = FlatScene()
    for _ in range(n_mat):
        t, _pad, r, g, b, fuzz, ior = struct.unpack_from("<2i5d", blob, off)
        s.materials.append(Material(t, (r, g, b), fuzz, ior))
        off += 48
    rec = np.frombuffer(blob, dtype=[("v", "<f8", 9), ("m", "<i4"), ("pad", "<i4")], count=n_tri, offset=off)
    s.vertices = _Rows(rec["v"].copy())
    s.tri_material = rec["m"].astype(np.int64).tolist()
    off += 80 * n_tri
    s.lights = list(struct.unpack_from("<%dI" % n_light, blob, off))
    off += 4 * (n_light + (n_light % 2))
    c = struct.unpack_from("<16d4i", blob, off)
    s.camera = Camera(lookfrom=c[0:3], lookat=c[3:6], vup=c[6:9], vfov=c[9], aspect_ratio=c[10], focus_dist=c[11],
                      defocus_angle=c[12], background=c[13:16], image_width=c[16], samples_per_pixel=c[17], max_depth=c[18])
    off += 16 * 8 + 4 * 4
    if blob[off:off + 8] != b"BDPTEXT1":
        return s
    # trailing block of a composite scene (CompositeScene.to_bytes); the bytes of image textures are not in the file
    n_tex, n_obj, _, _ = struct.unpack_from("<4I", blob, off + 8)
    off += 24
    cs = CompositeScene(materials=s.materials, vertices=s.vertices, tri_material=s.tri_material, lights=s.lights, camera=s.camera)
    moff = 24
    for m in cs.materials:
        m.texture = struct.unpack_from("<2i", blob, moff)[1] - 1
        moff += 48
    for _ in range(n_tex):
        t, even, odd, _w, _h, _pad, r, g, b, scale = struct.unpack_from("<6i4d", blob, off)
        cs.textures.append(Texture(t, (r, g, b), scale, even, odd))
        off += 56
    for _ in range(n_obj):
        first, n, flags, phase, deg, ox, oy, oz, density = struct.unpack_from("<3Ii5d", blob, off)
        cs.objects.append(SceneObject(first, n, flags, deg, (ox, oy, oz), density, phase))
        off += 56
    return cs


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def add_quad_triangles(scene: FlatScene, q, u, v, mat: int) -> None:
    """triangle.h:232-241: (q, q+u, q+v) and (q+u, q+u+v, q+v)."""
    qu = _add(q, u)
    qv = _add(q, v)
    quv = _add(qu, v)           # (q + u) + v, the reference's association
    scene.add_triangle(q, qu, qv, mat)
    scene.add_triangle(qu, quv, qv, mat)


def _rotate_y_point(p, sin_theta, cos_theta):
    """triangle.h:243-249"""
    return ((cos_theta * p[0]) + (sin_theta * p[2]), p[1], (-sin_theta * p[0]) + (cos_theta * p[2]))


def add_box_triangles(scene: FlatScene, a, b, mat: int, rotate_y_degrees: float = 0.0, translate=(0.0, 0.0, 0.0)):
    """triangle.h:251-309: 12 triangles, faces in the order +Z -Z -X +X +Y -Y."""
    mn = (min(a[0], b[0]), min(a[1], b[1]), min(a[2], b[2]))
    mx = (max(a[0], b[0]), max(a[1], b[1]), max(a[2], b[2]))
    v000 = (mn[0], mn[1], mn[2]); v001 = (mn[0], mn[1], mx[2])
    v010 = (mn[0], mx[1], mn[2]); v011 = (mn[0], mx[1], mx[2])
    v100 = (mx[0], mn[1], mn[2]); v101 = (mx[0], mn[1], mx[2])
    v110 = (mx[0], mx[1], mn[2]); v111 = (mx[0], mx[1], mx[2])
    faces = [
        (v001, v101, v111), (v001, v111, v011),
        (v000, v010, v110), (v000, v110, v100),
        (v000, v001, v011), (v000, v011, v010),
        (v101, v100, v110), (v101, v110, v111),
        (v011, v111, v110), (v011, v110, v010),
        (v000, v100, v101), (v000, v101, v001),
    ]
    radians = rotate_y_degrees * PI / 180.0
    sin_theta = math.sin(radians)
    cos_theta = math.cos(radians)
    for p0, p1, p2 in faces:
        if rotate_y_degrees != 0.0:
            p0 = _rotate_y_point(p0, sin_theta, cos_theta)
            p1 = _rotate_y_point(p1, sin_theta, cos_theta)
            p2 = _rotate_y_point(p2, sin_theta, cos_theta)
        scene.add_triangle(_add(p0, translate), _add(p1, translate), _add(p2, translate), mat)


def add_uv_sphere(scene: FlatScene, center, radius: float, mat: int, lat_steps: int = 16, lon_steps: int = 32):
    """scene_loader.h:204-242 (its own literal of pi, scene_loader.h:221)."""
    pi = 3.14159265358979323846

    def sphere_point(theta, phi):
        sin_theta = math.sin(theta)
        x = radius * sin_theta * math.cos(phi)
        y = radius * math.cos(theta)
        z = radius * sin_theta * math.sin(phi)
        return (center[0] + x, center[1] + y, center[2] + z)

    for lat in range(lat_steps):
        theta0 = pi * (float(lat) / lat_steps)
        theta1 = pi * (float(lat + 1) / lat_steps)
        for lon in range(lon_steps):
            phi0 = 2.0 * pi * (float(lon) / lon_steps)
            phi1 = 2.0 * pi * (float(lon + 1) / lon_steps)
            p00 = sphere_point(theta0, phi0)
            p01 = sphere_point(theta0, phi1)
            p10 = sphere_point(theta1, phi0)
            p11 = sphere_point(theta1, phi1)
            if lat > 0:
                scene.add_triangle(p00, p10, p11, mat)
            if lat < lat_steps - 1:
                scene.add_triangle(p00, p11, p01, mat)



# ------------------------------------------------------------------------------- composite scenes
@dataclass
class Texture:
    """/root/reference/src/materials/textures/texture.h:14-87"""
    type: int
    rgb: Sequence[float] = (0.0, 0.0, 0.0)     # solid
    scale: float = 1.0                         # checker, noise
    even: int = -1                             # checker: indices of earlier textures
    odd: int = -1
    image: object = None                       # image: H x W x 3 uint8 array as FILE content (the reference's loader
                                               # turns it into its own bytes, rtw_stb_image.h:98-112)


@dataclass
class SceneObject:
    """One entry of the world's list that is a mesh of its own: translate(rotate_y(bvh_node(mesh))), optionally the
    boundary of a constant_medium (/root/reference/src/objects/hittable.h:46-157,
    /root/reference/src/materials/volumes/constant_medium.h)."""
    first_triangle: int
    n_triangles: int
    flags: int = 0
    rotate_y_degrees: float = 0.0
    offset: Sequence[float] = (0.0, 0.0, 0.0)
    density: float = 0.0
    phase_material: int = -1


@dataclass
class CompositeScene(FlatScene):
    textures: List[Texture] = field(default_factory=list)
    objects: List[SceneObject] = field(default_factory=list)
    # filled from the reference harness (tests/golden) or by the host: what the reference's objects hold
    perlin_tables: List[np.ndarray] = field(default_factory=list)   # per noise texture, 1536 doubles (randvec, perms)
    image_bytes: List[np.ndarray] = field(default_factory=list)     # per image texture, H x W x 3 uint8 (rtw_image::bdata)

    def add_texture(self, t: Texture) -> int:
        self.textures.append(t)
        return len(self.textures) - 1

    def begin_object(self) -> int:
        return self.n_triangles

    def end_object(self, first: int, **kw) -> int:
        """Triangles added since begin_object() become one object (they are NOT light triangles)."""
        self.lights = [l for l in self.lights if l < first]
        self.objects.append(SceneObject(first, self.n_triangles - first, **kw))
        return len(self.objects) - 1

    def add_instance(self, of: int, **kw) -> int:
        """Another object over the mesh of object `of` (one bvh_node shared by several wrappers)."""
        src = self.objects[of]
        self.objects.append(SceneObject(src.first_triangle, src.n_triangles, **kw))
        return len(self.objects) - 1

    def to_bytes(self) -> bytes:
        out = [b"BDPTSCN1", struct.pack("<4I", len(self.materials), self.n_triangles, len(self.lights), 0)]
        for m in self.materials:
            out.append(struct.pack("<2i5d", m.type, m.texture + 1, m.rgb[0], m.rgb[1], m.rgb[2], m.fuzz, m.ior))
        rec = np.zeros(self.n_triangles, dtype=[("v", "<f8", 9), ("m", "<i4"), ("pad", "<i4")])
        rec["v"] = self.vertex_array()
        rec["m"] = self.tri_material_array()
        out.append(rec.tobytes())
        lights = list(self.lights) + ([0] if len(self.lights) % 2 else [])
        out.append(struct.pack("<%dI" % len(lights), *lights))
        c = self.camera
        out.append(struct.pack("<16d4i", *c.lookfrom, *c.lookat, *c.vup, c.vfov, c.aspect_ratio, c.focus_dist,
                               c.defocus_angle, *c.background, c.image_width, c.samples_per_pixel, c.max_depth, 0))
        out.append(b"BDPTEXT1" + struct.pack("<4I", len(self.textures), len(self.objects), 0, 0))
        for t in self.textures:
            h, w = (t.image.shape[0], t.image.shape[1]) if t.image is not None else (0, 0)
            out.append(struct.pack("<6i4d", t.type, t.even, t.odd, w, h, 0, t.rgb[0], t.rgb[1], t.rgb[2], t.scale))
        for ob in self.objects:
            out.append(struct.pack("<3Ii5d", ob.first_triangle, ob.n_triangles, ob.flags, ob.phase_material,
                                   ob.rotate_y_degrees, *ob.offset, ob.density))
        return b"".join(out)

    def write_files(self, scene_path: str) -> None:
        """The scene file and, beside it, the image files its image textures name (oracle/flat_scene.h)."""
        from PIL import Image
        with open(scene_path, "wb") as f:
            f.write(self.to_bytes())
        for i, t in enumerate(self.textures):
            if t.type == TEX_IMAGE and t.image is not None:
                Image.fromarray(np.ascontiguousarray(t.image, dtype=np.uint8), "RGB").save("%s.tex%d.png" % (scene_path, i))


def _object_box(scene: FlatScene, a, b, mat: int) -> None:
    add_box_triangles(scene, a, b, mat)


def test_image(h: int = 12, w: int = 16) -> np.ndarray:
    """Small RGB pattern with every value class the loader's byte conversion meets (0, 255, mid tones)."""
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), dtype=np.uint8)
    img[..., 0] = (x * 255) // (w - 1)
    img[..., 1] = (y * 255) // (h - 1)
    img[..., 2] = ((x * 7 + y * 13) * 5) % 256
    return img


def _cornell_walls(s: FlatScene, light_rgb=(15.0, 15.0, 15.0)):
    red = s.add_material(Material(LAMBERTIAN, (.65, .05, .05)))
    white = s.add_material(Material(LAMBERTIAN, (.73, .73, .73)))
    green = s.add_material(Material(LAMBERTIAN, (.12, .45, .15)))
    light = s.add_material(Material(DIFFUSE_LIGHT, light_rgb))
    add_quad_triangles(s, (555, 0, 0), (0, 0, 555), (0, 555, 0), green)
    add_quad_triangles(s, (0, 0, 555), (0, 0, -555), (0, 555, 0), red)
    add_quad_triangles(s, (0, 555, 0), (555, 0, 0), (0, 0, 555), white)
    add_quad_triangles(s, (0, 0, 555), (555, 0, 0), (0, 0, -555), white)
    add_quad_triangles(s, (555, 0, 555), (-555, 0, 0), (0, 555, 0), white)
    return red, white, green, light


def _cornell_camera(image_width, samples_per_pixel, max_depth, name):
    return Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel, max_depth=max_depth,
                  background=(0, 0, 0), vfov=40.0, lookfrom=(278, 278, -800), lookat=(278, 278, 0), vup=(0, 1, 0),
                  defocus_angle=0.0, file_name=name)


def cornell_instances(image_width: int = 32, samples_per_pixel: int = 16, max_depth: int = 8) -> CompositeScene:
    """The Cornell box as a host of the reference would write it with instances: both boxes are meshes at the origin,
    rotate_y(15) / rotate_y(-18) and translate put them in place (hittable.h:46-157); a third, metal box is only
    translated and a fourth, glass one only rotated."""
    s = CompositeScene()
    red, white, green, light = _cornell_walls(s)
    add_quad_triangles(s, (213, 554, 227), (130, 0, 0), (0, 0, 105), light)
    metal = s.add_material(Material(METAL, (.8, .85, .88), fuzz=0.1))
    glass = s.add_material(Material(DIELECTRIC, ior=1.5))
    f = s.begin_object(); _object_box(s, (0, 0, 0), (165, 330, 165), white)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE, rotate_y_degrees=15.0, offset=(265, 0, 295))
    f = s.begin_object(); _object_box(s, (0, 0, 0), (165, 165, 165), white)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE, rotate_y_degrees=-18.0, offset=(130, 0, 65))
    f = s.begin_object(); _object_box(s, (0, 0, 0), (60, 60, 60), metal)
    s.end_object(f, flags=OBJECT_TRANSLATE, offset=(420, 0.5, 60))
    f = s.begin_object(); _object_box(s, (300, 340, 200), (380, 420, 280), glass)
    s.end_object(f, flags=OBJECT_ROTATE_Y, rotate_y_degrees=8.0)
    s.camera = _cornell_camera(image_width, samples_per_pixel, max_depth, "cornell_instances.png")
    return s


def sphere_lattice_instanced(n_side: int = 22, image_width: int = 4096, samples_per_pixel: int = 64, max_depth: int = 10) -> CompositeScene:
    """sphere_lattice as a host of the reference would write it with instances: six sphere meshes of 960 triangles (one
    per colour), each a bvh_node of its own, and n_side^3 translate wrappers that share them (hittable.h:46-74).
    n_side = 22: 10 648 instances standing for 10.2 M triangles."""
    s = CompositeScene()
    grey = s.add_material(Material(LAMBERTIAN, (.6, .6, .6)))
    palette = [s.add_material(Material(LAMBERTIAN, c)) for c in
               ((.75, .25, .25), (.25, .75, .25), (.25, .25, .75), (.7, .7, .2), (.2, .7, .7), (.7, .2, .7))]
    light = s.add_material(Material(DIFFUSE_LIGHT, (40.0, 40.0, 40.0)))
    extent = 3.0 * n_side
    add_quad_triangles(s, (-extent, -1.5, -extent), (3 * extent, 0, 0), (0, 0, 3 * extent), grey)
    add_quad_triangles(s, (0.25 * extent, 1.6 * extent, 0.25 * extent), (0.5 * extent, 0, 0), (0, 0, 0.5 * extent), light)
    centers = [(3.0 * a, 3.0 * b, 3.0 * c) for a in range(n_side) for b in range(n_side) for c in range(n_side)]
    first_of = {}
    for k, ctr in enumerate(centers):
        colour = k % len(palette)
        if colour not in first_of:
            f = s.begin_object()
            add_uv_sphere(s, (0.0, 0.0, 0.0), 1.0, palette[colour])
            first_of[colour] = s.end_object(f, flags=OBJECT_TRANSLATE, offset=ctr)
        else:
            s.add_instance(first_of[colour], flags=OBJECT_TRANSLATE, offset=ctr)
    s.camera = Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel, max_depth=max_depth,
                      background=(0.02, 0.02, 0.03), vfov=40.0, lookfrom=(-1.2 * extent, 1.4 * extent, -1.6 * extent),
                      lookat=(0.5 * extent, 0.3 * extent, 0.5 * extent), vup=(0, 1, 0), file_name="sphere_lattice_instanced.png")
    return s


def cornell_textures(image_width: int = 32, samples_per_pixel: int = 16, max_depth: int = 8) -> CompositeScene:
    """Cornell box whose surfaces carry every texture class (texture.h:28-87): a checker floor whose odd squares are
    themselves marble noise, a noise back wall, an image on the left wall, a checkered emitter, and a box instance
    with a checker of two solid colours."""
    s = CompositeScene()
    c_white = s.add_texture(Texture(TEX_SOLID, rgb=(.73, .73, .73)))
    c_dark = s.add_texture(Texture(TEX_SOLID, rgb=(.2, .3, .1)))
    marble = s.add_texture(Texture(TEX_NOISE, scale=0.05))
    floor_t = s.add_texture(Texture(TEX_CHECKER, scale=80.0, even=c_white, odd=marble))
    wall_t = s.add_texture(Texture(TEX_NOISE, scale=0.02))
    image_t = s.add_texture(Texture(TEX_IMAGE, image=test_image()))
    e_hi = s.add_texture(Texture(TEX_SOLID, rgb=(18.0, 17.0, 15.0)))
    e_lo = s.add_texture(Texture(TEX_SOLID, rgb=(6.0, 8.0, 12.0)))
    emit_t = s.add_texture(Texture(TEX_CHECKER, scale=30.0, even=e_hi, odd=e_lo))
    box_t = s.add_texture(Texture(TEX_CHECKER, scale=40.0, even=c_dark, odd=c_white))
    red = s.add_material(Material(LAMBERTIAN, (.65, .05, .05)))
    white = s.add_material(Material(LAMBERTIAN, (.73, .73, .73)))
    floor_m = s.add_material(Material(LAMBERTIAN, texture=floor_t))
    wall_m = s.add_material(Material(LAMBERTIAN, texture=wall_t))
    image_m = s.add_material(Material(LAMBERTIAN, texture=image_t))
    light = s.add_material(Material(DIFFUSE_LIGHT, texture=emit_t))
    box_m = s.add_material(Material(LAMBERTIAN, texture=box_t))
    add_quad_triangles(s, (555, 0, 0), (0, 0, 555), (0, 555, 0), image_m)
    add_quad_triangles(s, (0, 0, 555), (0, 0, -555), (0, 555, 0), red)
    add_quad_triangles(s, (0, 555, 0), (555, 0, 0), (0, 0, 555), white)
    add_quad_triangles(s, (0, 0, 555), (555, 0, 0), (0, 0, -555), floor_m)
    add_quad_triangles(s, (555, 0, 555), (-555, 0, 0), (0, 555, 0), wall_m)
    add_quad_triangles(s, (183, 554, 197), (190, 0, 0), (0, 0, 165), light)
    f = s.begin_object(); _object_box(s, (0, 0, 0), (165, 330, 165), box_m)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE, rotate_y_degrees=15.0, offset=(265, 0, 295))
    s.camera = _cornell_camera(image_width, samples_per_pixel, max_depth, "cornell_textures.png")
    return s


def cornell_smoke(image_width: int = 32, samples_per_pixel: int = 16, max_depth: int = 8) -> CompositeScene:
    """Cornell box with the two boxes as volumes of constant density (constant_medium.h): white smoke in the rotated
    and translated tall box, dark smoke with a marble-noise albedo in the short one, a thin haze box that is not
    transformed at all, and a solid mirror box in front of them."""
    s = CompositeScene()
    red, white, green, light = _cornell_walls(s, (7.0, 7.0, 7.0))
    add_quad_triangles(s, (113, 554, 127), (330, 0, 0), (0, 0, 305), light)
    noise_t = s.add_texture(Texture(TEX_NOISE, scale=0.03))
    smoke_w = s.add_material(Material(ISOTROPIC, (1.0, 1.0, 1.0)))
    smoke_d = s.add_material(Material(ISOTROPIC, texture=noise_t))
    haze = s.add_material(Material(ISOTROPIC, (.6, .7, .9)))
    mirror = s.add_material(Material(METAL, (.9, .9, .9), fuzz=0.0))
    f = s.begin_object(); _object_box(s, (0, 0, 0), (165, 330, 165), white)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE | OBJECT_MEDIUM, rotate_y_degrees=15.0, offset=(265, 0, 295),
                 density=0.01, phase_material=smoke_w)
    f = s.begin_object(); _object_box(s, (0, 0, 0), (165, 165, 165), white)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE | OBJECT_MEDIUM, rotate_y_degrees=-18.0, offset=(130, 0, 65),
                 density=0.02, phase_material=smoke_d)
    f = s.begin_object(); _object_box(s, (20, 20, 20), (535, 200, 535), white)
    s.end_object(f, flags=OBJECT_MEDIUM, density=0.0008, phase_material=haze)
    f = s.begin_object(); _object_box(s, (0, 0, 0), (70, 70, 70), mirror)
    s.end_object(f, flags=OBJECT_ROTATE_Y | OBJECT_TRANSLATE, rotate_y_degrees=30.0, offset=(380, 0, 90))
    s.camera = _cornell_camera(image_width, samples_per_pixel, max_depth, "cornell_smoke.png")
    return s

# ------------------------------------------------------------------------------- scenes
def cornell_box(image_width: int = 512, samples_per_pixel: int = 256, max_depth: int = 10,
                short_box: bool = False) -> FlatScene:
    """Cornell geometry, materials and camera of /root/reference/src/main.cpp:17-33,42-53, with the
    emissive ceiling quad itself as the light collection (the reference's own `lights` there carries
    a null material and crashes in BDPT mode).  24 triangles; `short_box` adds the classic second
    box (not in the reference) for 36."""
    s = FlatScene()
    red = s.add_material(Material(LAMBERTIAN, (.65, .05, .05)))
    white = s.add_material(Material(LAMBERTIAN, (.73, .73, .73)))
    green = s.add_material(Material(LAMBERTIAN, (.12, .45, .15)))
    light = s.add_material(Material(DIFFUSE_LIGHT, (15.0, 15.0, 15.0)))
    add_quad_triangles(s, (555, 0, 0), (0, 0, 555), (0, 555, 0), green)
    add_quad_triangles(s, (0, 0, 555), (0, 0, -555), (0, 555, 0), red)
    add_quad_triangles(s, (0, 555, 0), (555, 0, 0), (0, 0, 555), white)
    add_quad_triangles(s, (0, 0, 555), (555, 0, 0), (0, 0, -555), white)
    add_quad_triangles(s, (555, 0, 555), (-555, 0, 0), (0, 555, 0), white)
    add_quad_triangles(s, (213, 554, 227), (130, 0, 0), (0, 0, 105), light)
    add_box_triangles(s, (0, 0, 0), (165, 330, 165), white, 15.0, (265, 0, 295))
    if short_box:
        add_box_triangles(s, (0, 0, 0), (165, 165, 165), white, -18.0, (130, 0, 65))
    s.camera = Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel,
                      max_depth=max_depth, background=(0, 0, 0), vfov=40.0, lookfrom=(278, 278, -800),
                      lookat=(278, 278, 0), vup=(0, 1, 0), defocus_angle=0.0, file_name="cornell_box.png")
    return s


def showcase_with_defocus(image_width: int = 32, samples_per_pixel: int = 16, max_depth: int = 6) -> FlatScene:
    """materials_showcase seen through a thin lens: defocus_angle 4 degrees, focus distance 7
    (/root/reference/src/camera.h:193-196,208,230-234)."""
    s = materials_showcase(image_width, samples_per_pixel, max_depth)
    s.camera.defocus_angle = 4.0
    s.camera.focus_dist = 7.0
    return s


def materials_showcase(image_width: int = 48, samples_per_pixel: int = 16, max_depth: int = 8) -> FlatScene:
    """Small closed room that exercises every material branch of the hot path: lambertian walls,
    a fuzzy metal box, a mirror (fuzz 0) floor, a dielectric box (ior 1.5) with a nested dielectric
    (ior 1.33), an emissive quad and an emissive UV sphere (many light triangles), non-black
    background seen through an open front."""
    s = FlatScene()
    white = s.add_material(Material(LAMBERTIAN, (.73, .73, .73)))
    blue = s.add_material(Material(LAMBERTIAN, (.15, .25, .65)))
    mirror = s.add_material(Material(METAL, (.9, .9, .9), fuzz=0.0))
    brushed = s.add_material(Material(METAL, (.8, .6, .2), fuzz=0.3))
    glass = s.add_material(Material(DIELECTRIC, ior=1.5))
    water = s.add_material(Material(DIELECTRIC, ior=1.33))
    light = s.add_material(Material(DIFFUSE_LIGHT, (12.0, 11.0, 9.0)))
    glow = s.add_material(Material(DIFFUSE_LIGHT, (2.0, 3.0, 6.0)))
    add_quad_triangles(s, (0, 0, 0), (10, 0, 0), (0, 0, 10), mirror)           # floor
    add_quad_triangles(s, (0, 8, 0), (0, 0, 10), (10, 0, 0), white)            # ceiling
    add_quad_triangles(s, (0, 0, 10), (10, 0, 0), (0, 8, 0), white)            # back
    add_quad_triangles(s, (0, 0, 0), (0, 0, 10), (0, 8, 0), blue)              # left
    add_quad_triangles(s, (10, 0, 0), (0, 8, 0), (0, 0, 10), white)            # right
    add_quad_triangles(s, (3.5, 7.99, 3.5), (3, 0, 0), (0, 0, 3), light)       # ceiling light (faces down)
    add_box_triangles(s, (0, 0, 0), (2, 4, 2), brushed, 20.0, (1.5, 0, 5.5))
    add_box_triangles(s, (0, 0, 0), (2.5, 2.5, 2.5), glass, -15.0, (5.5, 0.01, 3.0))
    add_box_triangles(s, (0, 0, 0), (1.5, 1.5, 1.5), water, -15.0, (6.0, 0.5, 3.6))
    add_uv_sphere(s, (2.0, 5.5, 7.5), 0.8, glow, 6, 8)
    s.camera = Camera(aspect_ratio=1.5, image_width=image_width, samples_per_pixel=samples_per_pixel,
                      max_depth=max_depth, background=(0.05, 0.08, 0.12), vfov=55.0, lookfrom=(5, 4, -9),
                      lookat=(5, 3.5, 5), vup=(0, 1, 0), defocus_angle=0.0, file_name="showcase.png")
    return s


def no_light_room(image_width: int = 24, samples_per_pixel: int = 4, max_depth: int = 5) -> FlatScene:
    """Scene without any emissive material: `lights` falls back to the whole world
    (/root/reference/src/main.cpp:67) and every light subpath dies after its three sampling
    draws (/root/reference/src/camera.h:393-396); radiance comes from the background only."""
    s = FlatScene()
    grey = s.add_material(Material(LAMBERTIAN, (.5, .5, .5)))
    metal = s.add_material(Material(METAL, (.7, .7, .7), fuzz=0.05))
    add_quad_triangles(s, (-5, 0, -5), (10, 0, 0), (0, 0, 10), grey)
    add_box_triangles(s, (-1, 0, -1), (1, 2, 1), metal, 30.0)
    add_box_triangles(s, (2, 0, 1), (3, 1, 2), grey)
    s.camera = Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel,
                      max_depth=max_depth, background=(0.7, 0.8, 1.0), vfov=50.0, lookfrom=(0, 3, -8),
                      lookat=(0, 1, 0), vup=(0, 1, 0), file_name="nolight.png")
    return s


def sphere_lattice(n_side: int = 22, image_width: int = 4096, samples_per_pixel: int = 64, max_depth: int = 10) -> FlatScene:
    """Many-triangle scene: the loader's 16 x 32 sphere tessellation (960 triangles) replicated on an
    n_side^3 grid of lambertian spheres over a floor, under one emissive quad; flattened, as the reference has no
    instancing on this path.  n_side = 22 gives 10.2 M triangles."""
    s = FlatScene()
    grey = s.add_material(Material(LAMBERTIAN, (.6, .6, .6)))
    palette = [s.add_material(Material(LAMBERTIAN, c)) for c in
               ((.75, .25, .25), (.25, .75, .25), (.25, .25, .75), (.7, .7, .2), (.2, .7, .7), (.7, .2, .7))]
    light = s.add_material(Material(DIFFUSE_LIGHT, (40.0, 40.0, 40.0)))
    extent = 3.0 * n_side
    add_quad_triangles(s, (-extent, -1.5, -extent), (3 * extent, 0, 0), (0, 0, 3 * extent), grey)
    add_quad_triangles(s, (0.25 * extent, 1.6 * extent, 0.25 * extent), (0.5 * extent, 0, 0), (0, 0, 0.5 * extent), light)
    unit = FlatScene()
    m0 = unit.add_material(Material(LAMBERTIAN, (0, 0, 0)))
    add_uv_sphere(unit, (0.0, 0.0, 0.0), 1.0, m0)
    base = np.array(unit.vertices, dtype=np.float64).reshape(-1, 3, 3)              # 960 x 3 vertices x xyz
    idx = np.arange(n_side, dtype=np.float64) * 3.0
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    centers = np.stack([cx.ravel(), cy.ravel(), cz.ravel()], axis=1)                 # n^3 x 3
    verts = (base[None, :, :, :] + centers[:, None, None, :]).reshape(-1, 9)
    mats = np.repeat(np.array(palette, dtype=np.int64)[np.arange(len(centers)) % len(palette)], len(base))
    s.vertices = np.concatenate([np.array(s.vertices, dtype=np.float64).reshape(-1, 9), verts]).tolist() if n_side <= 4 else \
        _Rows(np.concatenate([np.array(s.vertices, dtype=np.float64).reshape(-1, 9), verts]))
    s.tri_material = list(s.tri_material) + mats.tolist()
    s.camera = Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel, max_depth=max_depth,
                      background=(0.02, 0.02, 0.03), vfov=40.0, lookfrom=(-1.2 * extent, 1.4 * extent, -1.6 * extent),
                      lookat=(0.5 * extent, 0.3 * extent, 0.5 * extent), vup=(0, 1, 0), file_name="sphere_lattice.png")
    return s


class _Rows:
    """Vertex table kept as one array (large scenes); behaves like the list of 9-float rows FlatScene expects."""
    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.float64)

    def __len__(self):
        return len(self.a)

    def __iter__(self):
        return iter(self.a)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype, copy=False)


def mixed_materials_3k(image_width: int = 40, samples_per_pixel: int = 16, max_depth: int = 8) -> FlatScene:
    """Cornell room (both boxes) with three 960-triangle spheres - lambertian, glass, fuzzy metal - and a small
    emissive sphere: 3 140 triangles, all four materials, many light triangles; too large for the LDS copy of the
    whole scene, so the traversal kernels read it from global memory."""
    s = cornell_box(image_width, samples_per_pixel, max_depth, short_box=True)
    blue = s.add_material(Material(LAMBERTIAN, (.2, .3, .7)))
    glass = s.add_material(Material(DIELECTRIC, ior=1.5))
    steel = s.add_material(Material(METAL, (.8, .8, .85), fuzz=0.15))
    glow = s.add_material(Material(DIFFUSE_LIGHT, (4.0, 3.0, 2.0)))
    add_uv_sphere(s, (140.0, 245.0, 150.0), 75.0, blue)
    add_uv_sphere(s, (400.0, 110.0, 130.0), 90.0, glass)
    add_uv_sphere(s, (300.0, 420.0, 330.0), 70.0, steel)
    add_uv_sphere(s, (100.0, 80.0, 420.0), 35.0, glow, 8, 16)
    s.camera.file_name = "mixed3k.png"
    return s


def cornell_spheres_480(image_width: int = 40, samples_per_pixel: int = 16, max_depth: int = 8) -> FlatScene:
    """Cornell room (both boxes) with two coarse lambertian spheres: 484 triangles - the upper end of what the traversal
    kernels stage completely in LDS (hierarchy + triangles within 64 KiB), where the copy of the reference's tree beside
    them takes the dynamic LDS of the visibility and connection kernels well past 64 KiB."""
    s = cornell_box(image_width, samples_per_pixel, max_depth, short_box=True)
    blue = s.add_material(Material(LAMBERTIAN, (.2, .3, .7)))
    sand = s.add_material(Material(LAMBERTIAN, (.7, .6, .3)))
    add_uv_sphere(s, (140.0, 245.0, 150.0), 75.0, blue, 8, 16)
    add_uv_sphere(s, (400.0, 300.0, 330.0), 70.0, sand, 8, 16)
    s.camera.file_name = "cornell_spheres.png"
    return s


def coplanar_stack(image_width: int = 32, samples_per_pixel: int = 16, max_depth: int = 6) -> FlatScene:
    """Axis-aligned geometry chosen to land on the ties of the reference's traversal: a floor, a second quad lying
    exactly in the floor's plane (same t for both), a third partly overlapping both, an unrotated box standing
    flush on the floor (its faces coincide with faces of the boxes of the BVH nodes above them), a thin slab
    (1e-5 thick: its triangle boxes are padded), walls meeting at exact corners, and an emissive quad."""
    s = FlatScene()
    grey = s.add_material(Material(LAMBERTIAN, (.6, .6, .6)))
    red = s.add_material(Material(LAMBERTIAN, (.7, .2, .2)))
    green = s.add_material(Material(LAMBERTIAN, (.2, .7, .2)))
    mirror = s.add_material(Material(METAL, (.9, .9, .9), fuzz=0.0))
    glass = s.add_material(Material(DIELECTRIC, ior=1.5))
    light = s.add_material(Material(DIFFUSE_LIGHT, (10.0, 10.0, 10.0)))
    add_quad_triangles(s, (0, 0, 0), (8, 0, 0), (0, 0, 8), grey)              # floor
    add_quad_triangles(s, (2, 0, 2), (4, 0, 0), (0, 0, 4), red)               # in the floor's plane
    add_quad_triangles(s, (4, 0, 1), (3, 0, 0), (0, 0, 3), green)             # overlaps both
    add_quad_triangles(s, (0, 0, 8), (8, 0, 0), (0, 6, 0), grey)              # back wall
    add_quad_triangles(s, (0, 0, 0), (0, 0, 8), (0, 6, 0), red)               # left wall
    add_quad_triangles(s, (8, 0, 0), (0, 6, 0), (0, 0, 8), green)             # right wall
    add_quad_triangles(s, (0, 6, 0), (0, 0, 8), (8, 0, 0), grey)              # ceiling
    add_box_triangles(s, (1, 0, 5), (3, 2, 7), grey)                          # flush on the floor, axis-aligned
    add_box_triangles(s, (5, 0, 4.5), (7, 3, 6.5), mirror)
    add_box_triangles(s, (3.5, 0, 2.5), (4.5, 1, 3.5), glass)
    add_box_triangles(s, (2, 2.5, 2), (6, 2.50001, 4), red)                   # thin slab
    add_quad_triangles(s, (3, 5.999, 3), (2, 0, 0), (0, 0, 2), light)
    s.camera = Camera(aspect_ratio=1.0, image_width=image_width, samples_per_pixel=samples_per_pixel, max_depth=max_depth,
                      background=(0.0, 0.0, 0.0), vfov=60.0, lookfrom=(4, 3, -6), lookat=(4, 2.5, 4), vup=(0, 1, 0),
                      file_name="coplanar.png")
    return s


# ------------------------------------------------------------------------------- stand-ins for the shipped scenes
# Both scene files of the reference (scenes/glass-of-water.yaml, scenes/coffee_machine.yaml) name Wavefront OBJ
# meshes that are not part of its checkout.  The stand-ins below are procedural meshes written out under exactly
# those file names (write_standin_assets), so that the reference's own YAML files render unchanged through the
# scene loader; the builders further down assemble the same FlatScene directly (same triangles, same order).
def _lathe(profile, segments):
    """Triangles (numpy, M x 9) of the surface of revolution of the (r, y) polyline `profile`
    around the y axis.  Profile points with r == 0 produce single triangles (no degenerate ones)."""
    ang = np.linspace(0.0, 2.0 * math.pi, segments + 1)
    ca, sa = np.cos(ang), np.sin(ang)
    ca[-1], sa[-1] = ca[0], sa[0]                 # close the seam exactly
    tris = []
    for (r0, y0), (r1, y1) in zip(profile[:-1], profile[1:]):
        for k in range(segments):
            a0 = (r0 * ca[k], y0, r0 * sa[k]); a1 = (r0 * ca[k + 1], y0, r0 * sa[k + 1])
            b0 = (r1 * ca[k], y1, r1 * sa[k]); b1 = (r1 * ca[k + 1], y1, r1 * sa[k + 1])
            if r0 > 0:
                tris.append(a0 + b0 + a1)
            if r1 > 0:
                tris.append(a1 + b0 + b1)
    return np.array(tris, dtype=np.float64)


def _orient_outward(tris):
    """Flip the winding of a closed mesh if its signed volume is negative, so that cross(e1, e2)
    points out of the enclosed medium (dielectric::scatter picks 1/ior on front faces)."""
    v0, v1, v2 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
    vol = np.einsum("ij,ij->i", v0, np.cross(v1, v2)).sum() / 6.0
    if vol < 0:
        tris = np.concatenate([v0, v2, v1], axis=1)
    return tris


def _mesh_of(build) -> np.ndarray:
    """Triangles (M x 9) that `build(scene, material)` adds to an empty scene."""
    tmp = FlatScene()
    m = tmp.add_material(Material(LAMBERTIAN, (0, 0, 0)))
    build(tmp, m)
    return np.array(tmp.vertices, dtype=np.float64).reshape(-1, 9)


def _box_mesh(a, b, rotate_y_degrees=0.0, translate=(0.0, 0.0, 0.0)):
    return _mesh_of(lambda s, m: add_box_triangles(s, a, b, m, rotate_y_degrees, translate))


def _offset(tris, off):
    return tris + np.tile(np.array(off, dtype=np.float64), 3)[None, :]


def _grid_quad(q, u, v, nu, nv):
    """The parallelogram (q, u, v) as nu x nv quads of two triangles each (add_quad_triangles per cell)."""
    def build(s, m):
        q_, u_, v_ = (np.array(x, dtype=np.float64) for x in (q, u, v))
        for a in range(nu):
            for b in range(nv):
                add_quad_triangles(s, tuple(q_ + u_ * (a / nu) + v_ * (b / nv)), tuple(u_ / nu), tuple(v_ / nv), m)
    return _mesh_of(build)


def _add_mesh(scene: FlatScene, tris, mat: int):
    for t in tris:
        scene.add_triangle(t[0:3], t[3:6], t[6:9], mat)


def glass_of_water_meshes(detail: float = 1.0):
    """[(file name relative to the YAML, material name, triangles M x 9)] in the order of the `surfaces:` list of
    /root/reference/scenes/glass-of-water.yaml:47-123: a lathe-turned glass holding water with ice blocks (four of
    them with enclosed 'AirIce' pockets), on a metal floor in front of a curved metal backdrop - the scene's
    material mix (every surface except the light is a delta material), ~55 k triangles at detail 1."""
    seg = max(16, int(256 * detail))
    steps = max(4, int(40 * detail))
    # glass: outer wall up, rim, inner wall down, thick base (closed profile from the axis back to the axis)
    outer = [(2.0 + 0.45 * (k / steps) ** 1.5, 0.02 + 6.4 * k / steps) for k in range(steps + 1)]
    inner = [(r - 0.16, y) for r, y in reversed(outer) if y >= 0.55]
    prof = [(0.0, 0.02)] + outer + [(outer[-1][0] - 0.08, outer[-1][1] + 0.04)] + inner + [(0.0, inner[-1][1])]
    glass = _orient_outward(_lathe(prof, seg))
    # water body just inside the inner wall, flat surface at y = 4.4
    wsteps = max(3, int(24 * detail))
    y_lo, y_hi = inner[-1][1] + 0.01, 4.4
    rad = lambda y: 2.0 + 0.45 * ((y - 0.02) / 6.4) ** 1.5 - 0.16 - 0.01
    wprof = [(0.0, y_lo)] + [(rad(y_lo + (y_hi - y_lo) * k / wsteps), y_lo + (y_hi - y_lo) * k / wsteps)
                             for k in range(wsteps + 1)] + [(0.0, y_hi)]
    water = _orient_outward(_lathe(wprof, seg))
    # two drops of water on the floor beside the glass (closed lathe shapes)
    drop = lambda r, h, n: _orient_outward(_lathe([(0.0, 0.0)] + [(r * math.sin(0.5 * math.pi * k / n) ** 0.8, h * (1 - math.cos(0.5 * math.pi * k / n)))
                                                                   for k in range(1, n + 1)] + [(0.0, h)], max(12, seg // 8)))
    drop_a = _offset(drop(0.35, 0.12, max(3, steps // 6)), (3.4, 0.001, 1.2))
    drop_b = _offset(drop(0.22, 0.09, max(3, steps // 6)), (-3.1, 0.001, 2.0))
    # ice blocks floating in the water, four of them with an air pocket inside
    ice = [_box_mesh((-0.6, 0.0, -0.6), (0.6, 1.2, 0.6), 25.0, (-0.75, 3.7, 0.3)),
           _box_mesh((-0.55, 0.0, -0.55), (0.55, 1.1, 0.55), -40.0, (0.8, 3.6, -0.4)),
           _box_mesh((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5), 10.0, (0.1, 3.9, 0.9)),
           _box_mesh((-0.45, 0.0, -0.45), (0.45, 0.9, 0.45), 55.0, (-0.2, 3.55, -0.95)),
           _box_mesh((-0.4, 0.0, -0.4), (0.4, 0.8, 0.4), -15.0, (0.9, 3.75, 0.85))]
    air = [_box_mesh((-0.2, 0.0, -0.2), (0.2, 0.4, 0.2), 25.0, (-0.75, 4.1, 0.3)),
           _box_mesh((-0.15, 0.0, -0.15), (0.15, 0.3, 0.15), -40.0, (0.8, 4.0, -0.4)),
           _box_mesh((-0.12, 0.0, -0.12), (0.12, 0.25, 0.12), 10.0, (0.1, 4.3, 0.9)),
           _box_mesh((-0.1, 0.0, -0.1), (0.1, 0.2, 0.1), 55.0, (-0.2, 3.9, -0.95))]
    # floor and curved backdrop (a swept quarter cylinder), both metal
    floor = _mesh_of(lambda s, m: add_quad_triangles(s, (-40.0, 0.0, 30.0), (80.0, 0.0, 0.0), (0.0, 0.0, -42.0), m))
    bseg = max(8, int(48 * detail))
    bx = max(4, int(24 * detail))
    R = 12.0

    def backdrop_build(s, m):
        for k in range(bseg):
            t0, t1 = 0.5 * math.pi * k / bseg, 0.5 * math.pi * (k + 1) / bseg
            z0, y0 = -12.0 - R * math.sin(t0), R - R * math.cos(t0)
            z1, y1 = -12.0 - R * math.sin(t1), R - R * math.cos(t1)
            for j in range(bx):
                xa, xb = -40.0 + 80.0 * j / bx, -40.0 + 80.0 * (j + 1) / bx
                add_quad_triangles(s, (xa, y0, z0), (xb - xa, 0.0, 0.0), (0.0, y1 - y0, z1 - z0), m)
        add_quad_triangles(s, (-40.0, R, -12.0 - R), (80.0, 0.0, 0.0), (0.0, 30.0, 0.0), m)
    backdrop = _mesh_of(backdrop_build)
    d = "data/glass-of-water/models/"
    return [(d + "Mesh008.obj", "WaterAir", water), (d + "Mesh005.obj", "WaterAir", drop_a), (d + "Mesh000.obj", "WaterAir", drop_b),
            (d + "Mesh004.obj", "IceAir", ice[0]), (d + "Mesh010.obj", "IceAir", ice[1]),
            (d + "Mesh009.obj", "AirIce", air[0]), (d + "Mesh012.obj", "AirIce", air[1]),
            (d + "Mesh014.obj", "IceAir", ice[2]), (d + "Mesh015.obj", "AirIce", air[2]),
            (d + "Mesh006.obj", "IceAir", ice[3]), (d + "Mesh002.obj", "AirIce", air[3]),
            (d + "Mesh001.obj", "IceAir", ice[4]),
            (d + "Mesh007.obj", "Backdrop", backdrop), (d + "Mesh011.obj", "Floor", floor), (d + "Mesh013.obj", "Glass", glass)]


def glass_of_water_standin(image_width: int = 1920, image_height: int = 1080, samples_per_pixel: int = 1024,
                           max_depth: int = 12, detail: float = 1.0) -> FlatScene:
    """What the scene loader makes of /root/reference/scenes/glass-of-water.yaml with the stand-in meshes of
    glass_of_water_meshes in place of the missing OBJ files, assembled directly: the YAML's material table
    (glass-of-water.yaml:11-45, colours given in 0-255 scaled by 1/255 as scene_loader.h:81-92 does), its surfaces in
    file order, its area light (:125-134) and its camera (:1-9) with resolution, samples and depth of the
    benchmark configuration."""
    s = FlatScene()
    grey = (147 * (1.0 / 255.0),) * 3
    # (ior = 0 for everything that is not a dielectric: what the scene loader stores there)
    mats = {"Backdrop": s.add_material(Material(METAL, grey, fuzz=0.1, ior=0.0)),
            "Floor": s.add_material(Material(METAL, grey, fuzz=0.1, ior=0.0)),
            "WaterAir": s.add_material(Material(DIELECTRIC, ior=1.33)),
            "IceAir": s.add_material(Material(DIELECTRIC, ior=1.31)),
            "Glass": s.add_material(Material(DIELECTRIC, ior=1.5)),
            "AirIce": s.add_material(Material(DIELECTRIC, ior=0.763)),
            "Light": s.add_material(Material(DIFFUSE_LIGHT, (15.9155, 27.0563, 31.831), ior=0.0))}
    for _, mat, tris in glass_of_water_meshes(detail):
        _add_mesh(s, tris, mats[mat])
    lv = [(-2.53848244, 12.39156476, 24.58024509), (2.46151756, 12.39156524, 24.58024491),
          (2.46151804, 8.57023524, 27.80475491), (-2.53848196, 8.57023476, 27.80475509)]
    s.add_triangle(lv[0], lv[1], lv[2], mats["Light"])
    s.add_triangle(lv[0], lv[2], lv[3], mats["Light"])
    s.camera = Camera(aspect_ratio=float(image_width) / float(image_height), image_width=image_width,
                      samples_per_pixel=samples_per_pixel, max_depth=max_depth, background=(0.0, 0.0, 0.0), vfov=35.0,
                      lookfrom=(-0.08935849368572235, 2.694121837615967, 25.672550201416016),
                      lookat=(0.006514795124530792, 3.1010689735412598, 0.010185359977185726), vup=(0.0, 1.0, 0.0),
                      defocus_angle=0.0, focus_dist=10.0, file_name="glass_of_water.png")
    return s


def coffee_machine_meshes(detail: float = 1.0):
    """Stand-in meshes for /root/reference/scenes/coffee_machine.yaml:29-52 (data/coffee_maker/*.obj): a drip coffee
    maker of lambertian plastic (black body, orange trim) with a metal hot plate, filter ring and riser tube and a
    glass carafe with a thick wall, on a lambertian floor that curves up into a back wall.  The scene is about
    half a unit across (the YAML's lights and camera fix the scale).  ~110 k triangles at detail 1."""
    seg = max(12, int(192 * detail))
    n = max(2, int(12 * detail))
    ring = lambda prof, off, sg=seg: _offset(_orient_outward(_lathe(prof, sg)), off)
    # body: base slab, back column, head that overhangs the carafe (tessellated boxes), all black plastic
    def tess_box(a, b, k):
        (x0, y0, z0), (x1, y1, z1) = a, b
        parts = [_grid_quad((x0, y0, z1), (x1 - x0, 0, 0), (0, y1 - y0, 0), k, k), _grid_quad((x1, y0, z0), (x0 - x1, 0, 0), (0, y1 - y0, 0), k, k),
                 _grid_quad((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0), k, k), _grid_quad((x1, y0, z1), (0, 0, z0 - z1), (0, y1 - y0, 0), k, k),
                 _grid_quad((x0, y1, z1), (x1 - x0, 0, 0), (0, 0, z0 - z1), k, k), _grid_quad((x0, y0, z0), (x1 - x0, 0, 0), (0, 0, z1 - z0), k, k)]
        return np.concatenate(parts)
    black = np.concatenate([tess_box((-0.11, 0.0, -0.14), (0.11, 0.035, 0.12), n), tess_box((-0.09, 0.035, -0.14), (0.09, 0.34, -0.07), n),
                            tess_box((-0.10, 0.27, -0.07), (0.10, 0.36, 0.10), n),
                            ring([(0.0, 0.0), (0.018, 0.0), (0.018, 0.05), (0.012, 0.06), (0.0, 0.06)], (0.0, 0.17, 0.135), max(12, seg // 4))])   # carafe handle knob
    # orange trim: lid of the head, a band round the base, the carafe's lid
    orange = np.concatenate([tess_box((-0.102, 0.36, -0.072), (0.102, 0.375, 0.102), n), tess_box((-0.112, 0.012, -0.142), (0.112, 0.022, 0.122), max(2, n // 2)),
                             ring([(0.0, 0.0), (0.052, 0.0), (0.056, 0.008), (0.03, 0.02), (0.012, 0.03), (0.0, 0.03)], (0.0, 0.2005, 0.03))])
    # metal: hot plate, filter ring under the head, riser tube
    metal = np.concatenate([ring([(0.0, 0.0), (0.075, 0.0), (0.078, 0.004), (0.075, 0.008), (0.0, 0.008)], (0.0, 0.0352, 0.03)),
                            ring([(0.0, 0.0), (0.045, 0.0), (0.05, 0.03), (0.0, 0.03)], (0.0, 0.2395, 0.03)),
                            ring([(0.0, 0.0), (0.006, 0.0), (0.006, 0.22), (0.0, 0.22)], (0.07, 0.04, -0.06), max(8, seg // 6))])
    # glass carafe: outer wall up, rim, inner wall down (closed shell, 3 mm thick), standing on the hot plate
    k = max(6, int(36 * detail))
    outer = [(0.035 + 0.04 * math.sin(math.pi * (i / k) * 0.95) ** 0.8, 0.15 * i / k) for i in range(k + 1)]
    inner = [(max(r - 0.003, 0.004), y) for r, y in reversed(outer) if y >= 0.004]
    glass = ring([(0.0, 0.0)] + outer + inner + [(0.0, inner[-1][1])], (0.0, 0.0436, 0.03))
    # floor curving up into a back wall
    f = max(4, int(48 * detail))
    parts = [_grid_quad((-0.9, 0.0, 0.9), (1.8, 0.0, 0.0), (0.0, 0.0, -1.2), f, f)]
    Rr = 0.25
    for i in range(f):
        t0, t1 = 0.5 * math.pi * i / f, 0.5 * math.pi * (i + 1) / f
        z0, y0 = -0.3 - Rr * math.sin(t0), Rr - Rr * math.cos(t0)
        z1, y1 = -0.3 - Rr * math.sin(t1), Rr - Rr * math.cos(t1)
        parts.append(_grid_quad((-0.9, y0, z0), (1.8, 0.0, 0.0), (0.0, y1 - y0, z1 - z0), f, 1))
    parts.append(_grid_quad((-0.9, Rr, -0.3 - Rr), (1.8, 0.0, 0.0), (0.0, 0.9, 0.0), f, f // 2))
    floor = np.concatenate(parts)
    d = "data/coffee_maker/"
    return [(d + "Plastic_Orange.obj", "Plastic_Orange", orange), (d + "Plastic_Black.obj", "Plastic_Black", black),
            (d + "Metal.obj", "Metal", metal), (d + "Glass.obj", "Glass", glass), (d + "Floor.obj", "Floor", floor)]


def coffee_machine_standin(image_width: int = 1920, image_height: int = 1080, samples_per_pixel: int = 512,
                           max_depth: int = 10, detail: float = 1.0) -> FlatScene:
    """What the scene loader makes of /root/reference/scenes/coffee_machine.yaml (materials :8-28, three light quads
    :54-83, camera :1-6 with the defaults of src/camera.h:26-41 for what it leaves out) with the stand-in meshes,
    assembled directly; resolution, samples and depth of the benchmark configuration (512 spp = 22 x 22 samples)."""
    s = FlatScene()
    mats = {"Plastic_Orange": s.add_material(Material(LAMBERTIAN, (255 * (1.0 / 255.0), 97 * (1.0 / 255.0), 3 * (1.0 / 255.0)), ior=0.0)),
            "Plastic_Black": s.add_material(Material(LAMBERTIAN, (0.0, 0.0, 0.0), ior=0.0)),
            "Glass": s.add_material(Material(DIELECTRIC, ior=1.5)),
            "Metal": s.add_material(Material(METAL, (170 * (1.0 / 255.0),) * 3, fuzz=0.1, ior=0.0)),
            "Floor": s.add_material(Material(LAMBERTIAN, (147 * (1.0 / 255.0),) * 3, ior=0.0)),
            "Light": s.add_material(Material(DIFFUSE_LIGHT, (245.0, 245.0, 245.0), ior=0.0))}
    for _, mat, tris in coffee_machine_meshes(detail):
        _add_mesh(s, tris, mats[mat])
    quads = [[(-0.359309, 0.449693, -0.010809), (-0.196537, 0.449693, 0.338256), (-0.196537, 0.000849009, 0.338256), (-0.359309, 0.000848979, -0.010809)],
             [(0.320673, 0.027337, 0.228975), (0.320673, 0.476182, 0.228975), (0.325221, 0.476182, -0.136419), (0.325221, 0.027337, -0.136419)],
             [(0.230128, 0.50385, 0.267372), (-0.230128, 0.50385, 0.267372), (-0.230128, 0.50385, -0.192885), (0.230128, 0.50385, -0.192885)]]
    for v in quads:
        s.add_triangle(v[0], v[1], v[2], mats["Light"])
        s.add_triangle(v[0], v[2], v[3], mats["Light"])
    s.camera = Camera(aspect_ratio=float(image_width) / float(image_height), image_width=image_width,
                      samples_per_pixel=samples_per_pixel, max_depth=max_depth, background=(0.0, 0.0, 0.0), vfov=30.0,
                      lookfrom=(-0.0029600381385535, 0.19830328226089478, 0.828150749206543),
                      lookat=(0.002896534511819482, 0.16660931706428528, 0.022717338055372238), vup=(0.0, 1.0, 0.0),
                      defocus_angle=0.0, focus_dist=10.0, file_name="image.png")
    return s


def write_obj(path: str, tris: np.ndarray) -> None:
    """Triangle soup as a Wavefront OBJ file (`v` and `f` lines, what load_obj_file reads, scene_loader.h:346-397);
    coordinates with 17 significant digits so that reading them back returns the same doubles."""
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    flat = np.asarray(tris, dtype=np.float64).reshape(-1, 3)
    lines = ["v %s %s %s" % (repr(float(x)), repr(float(y)), repr(float(z))) for x, y, z in flat]
    lines += ["f %d %d %d" % (3 * k + 1, 3 * k + 2, 3 * k + 3) for k in range(len(flat) // 3)]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


STANDIN_MESHES = {"glass-of-water.yaml": glass_of_water_meshes, "coffee_machine.yaml": coffee_machine_meshes}


def write_standin_assets(yaml_path: str, out_dir: str, detail: float = 1.0) -> str:
    """Copies the scene file `yaml_path` (one of the reference's scenes/*.yaml) into `out_dir` and writes the
    stand-in OBJ meshes under the relative names it refers to.  Returns the path of the copy, ready for
    bdpt_scene_file_load / the reference's load_scene_from_yaml."""
    import os
    import shutil
    name = os.path.basename(yaml_path)
    os.makedirs(out_dir, exist_ok=True)
    dst = os.path.join(out_dir, name)
    shutil.copyfile(yaml_path, dst)
    for rel, _, tris in STANDIN_MESHES[name](detail):
        write_obj(os.path.join(out_dir, rel), tris)
    return dst
