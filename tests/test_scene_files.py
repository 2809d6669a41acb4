"""Scene files and image output: the product's loader (bdpt_scene_file_load, C++ without yaml-cpp) against the
Python restatement of src/scene/scene_loader.h (oracle/scene_loader_oracle.py, PyYAML front end), and the PNG
writer against a decoder.  CPU only.  Parity of this path is unpinned (the reference has no loader fixtures and
yaml-cpp is not available); what is checked is that two independent restatements agree."""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
import scene_loader_oracle as slo  # noqa: E402

DATA = os.path.join(ROOT, "tests", "data")


def assert_same_scene(flat, ref):
    c, r = flat.camera, ref.camera
    for k in ("aspect_ratio", "image_width", "samples_per_pixel", "max_depth", "vfov", "defocus_angle", "focus_dist"):
        assert getattr(c, k) == r[k], k
    for k in ("background", "lookfrom", "lookat", "vup"):
        assert tuple(getattr(c, k)) == tuple(r[k]), k
    assert c.file_name == ref.output_name
    assert [(m.type, tuple(m.rgb), m.fuzz, m.ior) for m in flat.materials] == [(t, tuple(rgb), f, i) for t, rgb, f, i in ref.materials]
    assert flat.tri_material == ref.tri_material
    assert flat.lights == ref.lights
    a = np.array(flat.vertices, dtype=np.float64).reshape(-1)
    b = np.array(ref.vertices, dtype=np.float64).reshape(-1)
    assert a.shape == b.shape
    # sphere vertices come from the C library's and Python's sin/cos: the same libm here, compared exactly
    assert (a.view(np.uint64) == b.view(np.uint64)).all()


@pytest.mark.parametrize("name", ["scene_typed.yaml", "scene_legacy.yaml"])
def test_loader_matches_restatement(name):
    path = os.path.join(DATA, name)
    flat = capi.load_scene_file(path)
    ref = slo.load_scene(path)
    assert_same_scene(flat, ref)


def test_typed_scene_contents():
    flat = capi.load_scene_file(os.path.join(DATA, "scene_typed.yaml"))
    assert flat.camera.image_width == 48 and flat.camera.aspect_ratio == 48 / 32 and flat.camera.vfov == 35.0
    assert flat.camera.file_name == "typed_scene.png" and flat.camera.defocus_angle == 0.0
    names = [m.type for m in flat.materials[:6]]
    assert names == [0, 1, 2, 2, 3, 1]                       # Floor, Mirror, Water, BadIor, Lamp, Odd (legacy metal); NotAMap skipped
    assert flat.materials[0].rgb == (147 / 255 * 1.0,) * 3 or np.allclose(flat.materials[0].rgb, 147 / 255)
    assert flat.materials[1].fuzz == 1.0 and flat.materials[3].ior == 1.5
    assert flat.materials[4].rgb == (15.5, 27.0, 31.8)
    # floor 2 + lamp 2 + sphere (16*32*2 - 2*32) + pyramid twice (2+1+1+1+1+1 = 7 each) + TriMesh 2
    assert flat.n_triangles == 2 + 2 + 960 + 7 + 7 + 2
    assert flat.lights == [2, 3]
    grey = flat.materials[flat.tri_material[2 + 2 + 960 + 7]]
    assert grey.type == 0 and tuple(grey.rgb) == (0.8, 0.8, 0.8)


def test_legacy_scene_contents():
    flat = capi.load_scene_file(os.path.join(DATA, "scene_legacy.yaml"))
    c = flat.camera
    assert c.image_width == 40 and c.aspect_ratio == 2.0 and c.vfov == 179.0 and c.focus_dist == 3.5
    assert c.samples_per_pixel == 9 and c.max_depth == 10 and tuple(c.lookfrom) == (0.0, 0.0, 0.0)
    m = flat.materials
    assert m[0].type == 3 and m[0].rgb == (50.0, 25.0, 12.5)                        # emission above 255: unscaled, clamped to 50
    assert m[1].type == 2 and m[1].ior == 1.5
    assert m[2].type == 1 and m[2].fuzz == 0.0 and np.allclose(m[2].rgb, (1 / 255, 128 / 255, 1.0))
    assert m[3].type == 3 and m[3].rgb == (0.5, 0.25, 0.0)
    assert m[4].type == 0 and m[4].rgb == (0.0, 0.0, 0.0)
    assert flat.lights == [0, 3]


@pytest.mark.parametrize("text, message", [
    ("camera: {resolution: [4, 4]}\n", "Scene/surfaces field missing or not a sequence"),
    ("- a\n- b\n", "Scene root must be a mapping"),
    ("camera:\n  fov: 3\nsurfaces: []\n", "Camera missing resolution"),
    ("camera:\n  resolution: [0, 4]\nsurfaces: []\n", "Resolution values must be positive"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - type: object\n    file: nowhere.obj\n", "Failed to open OBJ file"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - type: object\n", "Object missing file field"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - type: Sphere\n    material: {type: metal}\n    data: {radius: 0}\n", "Missing or invalid radius field"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - type: TriMesh\n    material: {}\n    data: {vertices: [1, 2]}\n", "Vertices length not a multiple of 9"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - type: mesh\n    vertices: [[0,0,0]]\n    triangles: [[0, 1, 2]]\n", "range"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - radius: 1\n", "Mesh missing type field"),
    ("camera: {resolution: [4, 4]}\nsurfaces:\n  - 5\n", "Scene entries must be mappings"),
    ("camera: [1, 2\n", "yaml"),
])
def test_loader_errors(tmp_path, text, message):
    p = tmp_path / "bad.yaml"
    p.write_text(text)
    with pytest.raises(capi.BdptError) as e:
        capi.load_scene_file(str(p))
    assert message in str(e.value)
    import yaml
    with pytest.raises((slo.LoadError, yaml.YAMLError)):
        slo.load_scene(str(p))


def test_missing_file_is_an_error(tmp_path):
    with pytest.raises(capi.BdptError):
        capi.load_scene_file(str(tmp_path / "absent.yaml"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenes"), reason="the reference checkout is not on this machine")
@pytest.mark.parametrize("name", ["glass-of-water.yaml", "coffee_machine.yaml"])
def test_reference_scene_files_parse(tmp_path, name):
    """The reference's own scene files (their OBJ meshes are not in its checkout: every `file:` gets the same
    small stand-in mesh), product loader against the restatement."""
    import yaml
    src = os.path.join("/root/reference/scenes", name)
    shutil.copy(src, tmp_path / name)
    doc = yaml.load(open(src, "rb").read().decode("utf-8"), Loader=yaml.BaseLoader)
    for s in doc["surfaces"]:
        if s.get("type") == "object":
            dst = tmp_path / s["file"]
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copy(os.path.join(DATA, "models", "pyramid.obj"), dst)
    flat = capi.load_scene_file(str(tmp_path / name))
    ref = slo.load_scene(str(tmp_path / name))
    assert_same_scene(flat, ref)
    assert flat.n_triangles > 0 and len(flat.lights) in (2, 6)


def test_png_round_trip(tmp_path, monkeypatch):
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    for (h, w) in [(1, 1), (7, 13), (64, 48)]:
        rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        capi.write_png("sub_%d.png" % h, rgb)
        back = np.array(Image.open(tmp_path / "output" / ("sub_%d.png" % h)).convert("RGB"))
        assert back.shape == rgb.shape and (back == rgb).all()
    with pytest.raises(capi.BdptError):
        capi.load_library().bdpt_write_png.restype   # keep the symbol referenced
        capi._check(capi.load_library().bdpt_write_png(b"x.png", None, 4, 4))


def test_cli_reports_loader_errors_like_the_reference(tmp_path):
    """main.cpp:72-75: message on stderr, exit code 1."""
    exe = os.path.join(ROOT, "bidirectional-path-tracer_amd", "bdpt_render")
    if not os.path.exists(exe):
        pytest.skip("command-line renderer not built")
    bad = tmp_path / "bad.yaml"
    bad.write_text("camera: {resolution: [4, 4]}\n")
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1
    assert "Failed to load scene: Scene/surfaces field missing or not a sequence" in r.stderr


YAML_CASES = [
    "a: 1\nb:\n  c: [1, 2, [3, 4]]\n  d: 'it''s'\n  e: \"q\\\"x\\n\"\n",
    "list:\n  - 1\n  - two\n  - [3]\n  -\n    k: v\n  - k1: v1\n    k2: v2\n",
    "top:\n- a\n- b: 1\n  c: 2\nnext: {x: 1, y: [1, 2], z: {w: 'a, b'}}\n",
    "# only a comment first\nkey: value # trailing\nother: \"has # inside\"   \nempty:\ntilde: ~\n",
    "m: [\n  1, 2,   # c\n  3\n]\nn: {a: 1,\n    b: 2}\n",
    "- - 1\n  - 2\n- - 3\n",
    "url: http://example.com:80/x\ntime: 12:30\n",
    "---\nroot:\n    deep:\n        deeper: [ ]\n        e: {}\n",
    "﻿bom: 1\n",
    "a:   \n  - x\n\n\n  - y   \nb: 'single # not comment'\n",
]


@pytest.mark.parametrize("text", YAML_CASES)
def test_yaml_reader_agrees_with_pyyaml(text):
    import yaml

    def norm(n):
        if isinstance(n, dict):
            return {str(k): norm(v) for k, v in n.items()}
        if isinstance(n, list):
            return [norm(v) for v in n]
        return None if n in ("", "~", "null", "Null", "NULL") else n

    mine = capi.yaml_to_json(text)
    theirs = norm(yaml.load(text, Loader=yaml.BaseLoader))
    assert mine == theirs


@pytest.mark.parametrize("text", ["a: &x 1\n", "a: !!int 1\n", "a: |\n  block\n", "a: 1\n---\nb: 2\n", "\ta: 1\n", "a: [1, 2\n", "a: 1\n   b: 2\n"])
def test_yaml_reader_rejects_what_it_does_not_support(text):
    with pytest.raises(capi.BdptError):
        capi.yaml_to_json(text)


def _with_benchmark_overrides(flat, twin):
    """The benchmark configurations override resolution, samples and depth of the YAML's camera block."""
    c, t = flat.camera, twin.camera
    c.image_width, c.aspect_ratio, c.samples_per_pixel, c.max_depth = t.image_width, t.aspect_ratio, t.samples_per_pixel, t.max_depth
    return flat


@pytest.mark.parametrize("name,builder,args", [("glass-of-water.yaml", "glass_of_water_standin", (1920, 1080, 1024, 12)),
                                               ("coffee_machine.yaml", "coffee_machine_standin", (1920, 1080, 512, 10))])
def test_reference_scene_files_load_unchanged(tmp_path, name, builder, args):
    """The reference's own scenes/*.yaml (tests/data/scenes, byte for byte) through bdpt_scene_file_load, with the
    stand-in OBJ meshes written under the file names the YAML refers to: the flat scene is exactly the one the
    Python builder assembles (the one the benchmarks and the full-size tests use), and the Python restatement of the
    loader agrees."""
    scenes = importlib.import_module("bidirectional-path-tracer_amd.scenes")
    path = scenes.write_standin_assets(os.path.join(DATA, "scenes", name), str(tmp_path), detail=0.25)
    flat = capi.load_scene_file(path)
    ref = slo.load_scene(path)
    assert_same_scene(flat, ref)
    twin = getattr(scenes, builder)(*args, detail=0.25)
    # the camera block of the file: resolution / samples / depth are the file's (or the reference's defaults)
    if name == "glass-of-water.yaml":
        assert (flat.camera.image_width, flat.camera.samples_per_pixel, flat.camera.max_depth) == (1280, 400, 80)
        assert flat.camera.aspect_ratio == 1280 / 720 and flat.camera.file_name == "glass_of_water.png"
    else:
        assert (flat.camera.image_width, flat.camera.samples_per_pixel, flat.camera.max_depth) == (800, 50, 10)
        assert flat.camera.aspect_ratio == 1.0 and flat.camera.file_name == "image.png"
    flat.camera.file_name = twin.camera.file_name
    _with_benchmark_overrides(flat, twin)
    assert flat.to_bytes() == twin.to_bytes()
    assert flat.lights == twin.lights and [m.type for m in flat.materials] == [m.type for m in twin.materials]


def test_reference_scene_file_without_its_meshes_fails_like_the_reference(tmp_path):
    shutil.copyfile(os.path.join(DATA, "scenes", "coffee_machine.yaml"), tmp_path / "coffee_machine.yaml")
    with pytest.raises(capi.BdptError, match="Failed to open OBJ file"):
        capi.load_scene_file(str(tmp_path / "coffee_machine.yaml"))
