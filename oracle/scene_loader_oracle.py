"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's scene loader, used to check the C++ loader
of the product (bdpt_scene_file_load).  Nothing outside tests/ may import this module.

Follows src/scene/scene_loader.h:32-523 function by function.  The YAML text is parsed by PyYAML's
BaseLoader (every scalar stays a string), and the scalar conversions restate yaml-cpp's rules, so two
independent YAML front ends (PyYAML here, the product's own reader there) have to agree.

Parity is UNPINNED for this path: the reference links yaml-cpp (no version pinned in vcpkg.json), which is not
available here, and the reference holds no loader tests or fixtures.  What is checked is agreement of two
independent restatements of scene_loader.h on scene files written for the purpose, plus the structure of the
reference's own two scene files up to their (absent) OBJ meshes.
"""
import math
import os
import re

import yaml

LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = 0, 1, 2, 3

_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?[ \t]*$")
_NULLS = ("", "~", "null", "Null", "NULL")


class LoadError(Exception):
    pass


def _is_scalar(n):
    return isinstance(n, str) and n not in _NULLS


def to_double(s):
    """yaml-cpp convert<double>: the whole scalar, .inf/.nan spellings, no leading blank."""
    if s in (".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF"):
        return math.inf
    if s in ("-.inf", "-.Inf", "-.INF"):
        return -math.inf
    if s in (".nan", ".NaN", ".NAN"):
        return math.nan
    if not _FLOAT.match(s):
        return None
    return float(s.strip())


def to_int(s):
    """yaml-cpp convert<int>: stream extraction with the basefield cleared (0x.., 0.. prefixes)."""
    m = re.match(r"^([+-]?)(0[xX][0-9a-fA-F]+|0[0-7]*|[1-9]\d*)[ \t]*$", s)
    if not m:
        return None
    body = m.group(2)
    v = int(body, 16) if body[:2].lower() == "0x" else (int(body, 8) if len(body) > 1 and body[0] == "0" else int(body))
    v = -v if m.group(1) == "-" else v
    return v if -2 ** 31 <= v < 2 ** 31 else None


def node_to_string(n, default=""):
    return n if _is_scalar(n) else default


def node_to_double(n, default=0.0):
    if not _is_scalar(n):
        return default
    v = to_double(n)
    return default if v is None else v


def node_to_int(n, default=0):
    if not _is_scalar(n):
        return default
    v = to_int(n)
    return default if v is None else v


def node_to_double_list(n):
    if not isinstance(n, list):
        return []
    return [node_to_double(e, 0.0) for e in n]


def read_vec3(n, fallback):
    v = node_to_double_list(n)
    return tuple(fallback) if len(v) < 3 else (v[0], v[1], v[2])


def read_color_scaled(n, fallback):
    v = node_to_double_list(n)
    if len(v) < 3:
        return tuple(fallback)
    r, g, b = v[0], v[1], v[2]
    maxc = max(abs(r), abs(g), abs(b))
    if 1.0 < maxc <= 255.0:
        s = 1.0 / 255.0
        r, g, b = r * s, g * s, b * s
    return (r, g, b)


def _get(node, key):
    """node[key] of yaml-cpp: None stands for an undefined node."""
    return node.get(key) if isinstance(node, dict) and key in node else None


def _clamp(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


def build_material(node):
    """scene_loader.h:101-170 -> (type, rgb, fuzz, ior)"""
    if not isinstance(node, dict):
        raise LoadError("Material must be a mapping")
    black = (0.0, 0.0, 0.0)
    type_str = node_to_string(_get(node, "type"), "")
    if type_str:
        color_value = read_color_scaled(
            _get(node, "color"),
            read_color_scaled(_get(node, "albedo"),
                              read_color_scaled(_get(node, "base_color"), read_color_scaled(_get(node, "base_colour"), black))))
        if type_str in ("light", "diffuse_light"):
            return (DIFFUSE_LIGHT, read_vec3(_get(node, "emission"), black), 0.0, 0.0)
        if type_str == "lambertian":
            return (LAMBERTIAN, color_value, 0.0, 0.0)
        if type_str == "metal":
            return (METAL, color_value, _clamp(node_to_double(_get(node, "roughness"), 0.0), 0.0, 1.0), 0.0)
        if type_str in ("dielectric", "glass"):
            ior = node_to_double(_get(node, "ior"), 1.5)
            return (DIELECTRIC, black, 0.0, ior if ior > 0.0 else 1.5)
    base_color = read_color_scaled(_get(node, "base_color"), black)
    if _get(node, "base_colour") is not None:
        base_color = read_color_scaled(_get(node, "base_colour"), base_color)
    emission = read_color_scaled(_get(node, "emission"), black)
    if emission[0] * emission[0] + emission[1] * emission[1] + emission[2] * emission[2] > 0.0:
        maxc = max(abs(emission[0]), abs(emission[1]), abs(emission[2]))
        if maxc > 50.0:
            s = 50.0 / maxc
            emission = (emission[0] * s, emission[1] * s, emission[2] * s)
        return (DIFFUSE_LIGHT, emission, 0.0, 0.0)
    transmission = node_to_double(_get(node, "transmission"), 0.0)
    if transmission == 0.0:
        transmission = node_to_double(_get(node, "spec_trans"), 0.0)
    ior = node_to_double(_get(node, "ior"), 1.5)
    if transmission > 0.0:
        return (DIELECTRIC, black, 0.0, ior if ior > 0.0 else 1.5)
    metallic = node_to_double(_get(node, "metallic"), 0.0)
    roughness = _clamp(node_to_double(_get(node, "roughness"), 0.0), 0.0, 1.0)
    if metallic > 0.5:
        return (METAL, base_color, roughness, 0.0)
    return (LAMBERTIAN, base_color, 0.0, 0.0)


class Scene:
    def __init__(self):
        self.camera = dict(aspect_ratio=1.0, image_width=100, samples_per_pixel=50, max_depth=10, background=(0.0, 0.0, 0.0),
                           vfov=90.0, lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vup=(0.0, 1.0, 0.0),
                           defocus_angle=0.0, focus_dist=10.0)
        self.output_name = "image.png"
        self.vertices = []        # 9 doubles per triangle
        self.tri_material = []
        self.materials = []       # (type, rgb, fuzz, ior) in order of creation
        self.lights = []
        self._named = {}
        self._grey = None

    def _add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def _add_triangle(self, a, b, c, mat):
        self.vertices.extend([*a, *b, *c])
        self.tri_material.append(mat)
        if self.materials[mat][0] == DIFFUSE_LIGHT:
            self.lights.append(len(self.tri_material) - 1)

    def _resolve_material(self, n):
        if n is not None:
            if _is_scalar(n):
                if n in self._named:
                    return self._named[n]
            elif isinstance(n, dict):
                return self._add_material(build_material(n))
        if self._grey is None:
            self._grey = self._add_material((LAMBERTIAN, (0.8, 0.8, 0.8), 0.0, 0.0))
        return self._grey

    def load_materials(self, node):
        if not isinstance(node, dict):
            return
        for name, d in node.items():
            try:
                m = build_material(d)
            except LoadError:
                continue
            if name in self._named:
                self.materials[self._named[name]] = m
            else:
                self._named[name] = self._add_material(m)

    def load_tri_mesh(self, mesh):
        data = _get(mesh, "data")
        if not isinstance(data, dict):
            raise LoadError("Mesh missing data field")
        vn = _get(data, "vertices")
        if not isinstance(vn, list):
            raise LoadError("Missing vertices")
        v = node_to_double_list(vn)
        if len(v) % 9:
            raise LoadError("Vertices length not a multiple of 9")
        if _get(mesh, "material") is None:
            raise LoadError("Missing material field")
        mat = self._add_material(build_material(_get(mesh, "material")))
        for i in range(0, len(v), 9):
            self._add_triangle(v[i:i + 3], v[i + 3:i + 6], v[i + 6:i + 9], mat)

    def load_sphere(self, mesh):
        if _get(mesh, "material") is None:
            raise LoadError("Missing material field")
        mat = self._add_material(build_material(_get(mesh, "material")))
        data = _get(mesh, "data")
        if not isinstance(data, dict):
            raise LoadError("Missing data field")
        center = read_vec3(_get(data, "center"), (0.0, 0.0, 0.0))
        radius = node_to_double(_get(data, "radius"), 0.0)
        if radius <= 0.0:
            raise LoadError("Missing or invalid radius field")
        pi = 3.14159265358979323846

        def point(theta, phi):
            st = math.sin(theta)
            return (center[0] + radius * st * math.cos(phi), center[1] + radius * math.cos(theta), center[2] + radius * st * math.sin(phi))

        lat_steps, lon_steps = 16, 32
        for lat in range(lat_steps):
            t0 = pi * (float(lat) / lat_steps)
            t1 = pi * (float(lat + 1) / lat_steps)
            for lon in range(lon_steps):
                p0 = 2.0 * pi * (float(lon) / lon_steps)
                p1 = 2.0 * pi * (float(lon + 1) / lon_steps)
                p00, p01, p10, p11 = point(t0, p0), point(t0, p1), point(t1, p0), point(t1, p1)
                if lat > 0:
                    self._add_triangle(p00, p10, p11, mat)
                if lat < lat_steps - 1:
                    self._add_triangle(p00, p11, p01, mat)

    def load_indexed_mesh(self, mesh):
        vn, tn = _get(mesh, "vertices"), _get(mesh, "triangles")
        if not isinstance(vn, list):
            raise LoadError("Indexed mesh missing vertices")
        if not isinstance(tn, list):
            raise LoadError("Indexed mesh missing triangles")
        verts = []
        for v in vn:
            vals = node_to_double_list(v)
            if len(vals) >= 3:
                verts.append(tuple(vals[:3]))
        mat = self._resolve_material(_get(mesh, "material"))
        for tri in tn:
            idx = [node_to_int(t, 0) for t in tri] if isinstance(tri, list) else []
            if len(idx) < 3:
                continue
            for i in idx[:3]:
                if not 0 <= i < len(verts):
                    raise LoadError("vector::_M_range_check")
            self._add_triangle(verts[idx[0]], verts[idx[1]], verts[idx[2]], mat)

    def load_obj_file(self, path, mat):
        try:
            fh = open(path, "r", errors="replace")
        except OSError:
            raise LoadError("Failed to open OBJ file: " + path)
        verts = []
        with fh:
            for line in fh.read().split("\n"):
                if not line:
                    continue
                toks = line.split()
                if not toks:
                    continue
                if toks[0] == "v":
                    try:
                        verts.append((float(toks[1]), float(toks[2]), float(toks[3])))
                    except (IndexError, ValueError):
                        pass
                elif toks[0] == "f":
                    fidx = []
                    for tok in toks[1:]:
                        m = re.match(r"^[+-]?\d+", tok.split("/")[0])      # std::stoi: leading integer
                        if not m:
                            continue
                        vi = int(m.group(0))
                        fidx.append(vi - 1 if vi > 0 else len(verts) + vi)
                    for k in range(2, len(fidx)):
                        for i in (fidx[0], fidx[k - 1], fidx[k]):
                            if not 0 <= i < len(verts):
                                raise LoadError("vector::_M_range_check")
                        self._add_triangle(verts[fidx[0]], verts[fidx[k - 1]], verts[fidx[k]], mat)

    def load_object(self, node, yaml_dir):
        rel = node_to_string(_get(node, "file"))
        if not rel:
            raise LoadError("Object missing file field")
        mat = self._resolve_material(_get(node, "material"))
        self.load_obj_file(os.path.join(yaml_dir, rel), mat)

    def load_camera(self, node):
        if not isinstance(node, dict):
            raise LoadError("Camera section must be a mapping")
        res = _get(node, "resolution")
        if not isinstance(res, list) or len(res) < 2:
            raise LoadError("Camera missing resolution")
        vals = node_to_double_list(res)
        width, height = int(vals[0]), int(vals[1])
        if width <= 0 or height <= 0:
            raise LoadError("Resolution values must be positive")
        cam = self.camera
        cam["image_width"] = width
        cam["aspect_ratio"] = float(width) / float(height)
        vfov = cam["vfov"]
        if _get(node, "vfov") is not None or _get(node, "fov") is not None:
            vfov = node_to_double(_get(node, "vfov"), node_to_double(_get(node, "fov"), vfov))
        cam["vfov"] = _clamp(vfov, 1.0, 179.0)
        cam["focus_dist"] = node_to_double(_get(node, "focus_distance"), cam["focus_dist"])
        cam["defocus_angle"] = 0.0
        cam["lookfrom"] = read_vec3(_get(node, "location"), cam["lookfrom"])
        cam["lookat"] = read_vec3(_get(node, "look_at"), cam["lookat"])
        cam["vup"] = read_vec3(_get(node, "up"), cam["vup"])
        cam["background"] = read_vec3(_get(node, "background"), cam["background"])
        cam["samples_per_pixel"] = node_to_int(_get(node, "samples_per_pixel"), cam["samples_per_pixel"])
        cam["max_depth"] = node_to_int(_get(node, "max_depth"), cam["max_depth"])
        out = node_to_string(_get(node, "output"))
        if out:
            self.output_name = out


def load_scene(path, skip_missing_obj=False):
    """load_scene_from_yaml, scene_loader.h:480-523.  skip_missing_obj: keep going when an OBJ file is absent
    (for the reference's own scene files, whose meshes are not in its checkout)."""
    with open(path, "rb") as fh:
        root = yaml.load(fh.read().decode("utf-8"), Loader=yaml.BaseLoader)
    if not isinstance(root, dict):
        raise LoadError("Scene root must be a mapping")
    s = Scene()
    s.load_camera(_get(root, "camera"))
    s.load_materials(_get(root, "materials"))
    surfaces = _get(root, "surfaces")
    if surfaces is None:
        surfaces = _get(root, "scene")
    if not isinstance(surfaces, list):
        raise LoadError("Scene/surfaces field missing or not a sequence")
    yaml_dir = os.path.dirname(path)
    for mesh in surfaces:
        if not isinstance(mesh, dict):
            raise LoadError("Scene entries must be mappings")
        t = node_to_string(_get(mesh, "type"))
        if not t:
            raise LoadError("Mesh missing type field")
        if t == "TriMesh":
            s.load_tri_mesh(mesh)
        elif t == "Sphere":
            s.load_sphere(mesh)
        elif t == "mesh":
            s.load_indexed_mesh(mesh)
        elif t == "object":
            try:
                s.load_object(mesh, yaml_dir)
            except LoadError as ex:
                if not (skip_missing_obj and str(ex).startswith("Failed to open OBJ file")):
                    raise
    return s
