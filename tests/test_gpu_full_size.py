"""The BASELINE configurations at their full sizes, through properties that do not need the CPU oracle for every
pixel: sharded renders sum to the unsharded image bit for bit, a repeated render is identical, the two independent
GPU implementations (wavefront pipeline, single kernel) agree on a crop, and a lattice of pixels is checked
against the oracle at the full sample count."""
import importlib

import numpy as np
import pytest

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
pytestmark = pytest.mark.gpu


def from_reference_yaml(workload):
    """The benchmark workload as bench.py builds it: the reference's own scene file through bdpt_scene_file_load."""
    import bench
    return bench.build_scene(workload)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check_full_size(flat, ctx, oracle, lattice, crop, shard_counts=(2, 3)):
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    W, H = cam.image_width, cam.image_height
    whole = sc.render(cam, seed=0)
    assert np.isfinite(whole).all() and whole.max() > 0
    # shards (row bands of 4 dealt to 2 and to 3 ranks) sum to the unsharded image exactly
    for n in shard_counts:
        acc = np.zeros_like(whole)
        for r in range(n):
            acc += sc.render(cam, seed=0, shard=(r, n, 4))
        assert (bits(acc) == bits(whole)).all(), "%d shards" % n
    # the single-kernel implementation on a crop
    x0, y0, x1, y1 = crop
    mk = sc.render(cam, seed=0, rect=crop, megakernel=True)
    assert (bits(mk[y0:y1, x0:x1]) == bits(whole[y0:y1, x0:x1])).all(), "single kernel vs wavefront pipeline"
    # pixels spread over the frame against the CPU oracle at the full sample count
    osc = oracle.scene(flat)
    for (i, j) in lattice(W, H):
        ofb, _, _ = osc.render(cam.as_array(), W, cam.sqrt_spp, cam.max_depth, 0, (i, j, i + 1, j + 1))
        assert (bits(ofb[0, 0]) == bits(whole[j, i])).all(), "pixel (%d, %d)" % (i, j)
    sc.close()


def test_cornell_512_256spp_depth10(scenes, ctx, oracle):
    flat = scenes.cornell_box(512, 256, 10)
    check_full_size(flat, ctx, oracle, lambda W, H: [(int((a + 0.5) * W / 4), int((b + 0.5) * H / 4)) for a in range(4) for b in range(4)],
                    (200, 240, 232, 248))


def test_glass_of_water_1080p_1024spp_depth12(scenes, ctx, oracle):
    """Configuration 3 (the headline): scenes/glass-of-water.yaml through the loader, 1920 x 1080, 1024 spp, depth 12."""
    flat = from_reference_yaml("glass-of-water-1080p")
    assert flat.camera.image_width == 1920 and flat.camera.samples_per_pixel == 1024 and flat.camera.max_depth == 12
    assert flat.n_triangles == scenes.glass_of_water_standin().n_triangles
    check_full_size(flat, ctx, oracle, lambda W, H: [(int((a + 0.5) * W / 6), int((b + 0.5) * H / 4)) for a in range(6) for b in range(4)],
                    (900, 500, 916, 504))


def test_million_triangle_lattice(scenes, ctx, oracle):
    """Many-triangle scene (0.96 M triangles: deep hierarchies, nothing fits LDS): pixels against the oracle, and the
    two GPU implementations against each other."""
    flat = scenes.sphere_lattice(10, 96, 16, 10)
    assert flat.n_triangles == 960 * 1000 + 4
    sc = ctx.scene(flat)
    info = sc.info()
    assert info.n_triangles == flat.n_triangles
    cam = capi.camera_initialize(flat.camera)
    W, H = cam.image_width, cam.image_height
    whole = sc.render(cam, seed=1)
    crop = (40, 40, 56, 48)
    mk = sc.render(cam, seed=1, rect=crop, megakernel=True)
    assert (bits(mk[40:48, 40:56]) == bits(whole[40:48, 40:56])).all()
    osc = oracle.scene(flat)
    ofb, _, _ = osc.render(cam.as_array(), W, cam.sqrt_spp, cam.max_depth, 1, (20, 60, 76, 64))
    assert (bits(ofb) == bits(whole[60:64, 20:76])).all()
    sc.close()


def test_glass_of_water_256_16spp_depth4_whole_frame_against_the_oracle(scenes, ctx, oracle):
    """Configuration 1 (the reference's CPU-runnable case): scenes/glass-of-water.yaml at 256 x 256, 16 spp, depth 4 -
    every pixel of the GPU frame against the CPU oracle."""
    flat = from_reference_yaml("glass-of-water-256")
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    assert (cam.image_width, cam.image_height, cam.sqrt_spp, cam.max_depth) == (256, 256, 4, 4)
    fb = sc.render(cam, seed=0)
    ofb, _, _ = oracle.scene(flat).render(cam.as_array(), 256, 4, 4, 0, (0, 0, 256, 256))
    assert (bits(fb) == bits(ofb)).all() and fb.max() > 0
    sc.close()


def test_coffee_machine_1080p_512spp_depth10(scenes, ctx, oracle):
    """Configuration 4: scenes/coffee_machine.yaml through the loader, 1920 x 1080, 512 spp (22 x 22 = 484 used), depth 10:
    lambertian, dielectric and metal surfaces under three light quads - light subpaths, connections and visibility
    rays all at work."""
    flat = from_reference_yaml("coffee-machine-1080p")
    cam = capi.camera_initialize(flat.camera)
    assert (cam.image_width, cam.image_height, cam.sqrt_spp, cam.max_depth) == (1920, 1080, 22, 10)
    assert len(flat.lights) == 6
    check_full_size(flat, ctx, oracle, lambda W, H: [(int((a + 0.5) * W / 4), int((b + 0.5) * H / 3)) for a in range(4) for b in range(3)],
                    (940, 600, 956, 604), shard_counts=(2,))


def test_ten_million_triangles_4096_64spp_depth10(scenes, ctx, oracle):
    """Configuration 5: 10.2 M triangles (22^3 spheres of 960), 4096 x 4096, 64 spp, depth 10."""
    flat = scenes.sphere_lattice(22, 4096, 64, 10)
    assert flat.n_triangles == 960 * 22 ** 3 + 4
    check_full_size(flat, ctx, oracle, lambda W, H: [(int((a + 0.5) * W / 3), int((b + 0.5) * H / 2)) for a in range(3) for b in range(2)],
                    (2040, 2040, 2056, 2044), shard_counts=(2,))


def test_ten_million_triangles_as_instances_4096_64spp_depth10(scenes, ctx, oracle):
    """Configuration 5 as worded ("instanced"): 10 648 translate wrappers (src/objects/hittable.h:46-74) over six shared
    960-triangle sphere meshes - 10.2 M triangles once instanced - 4096 x 4096, 64 spp, depth 10, on the pipeline's
    two-level tables: shards, the single-kernel walk over the reference's own trees on a crop, oracle pixels."""
    flat = scenes.sphere_lattice_instanced(22, 4096, 64, 10)
    assert len(flat.objects) == 22 ** 3 and sum(o.n_triangles for o in flat.objects) == 960 * 22 ** 3
    check_full_size(flat, ctx, oracle, lambda W, H: [(int((a + 0.5) * W / 3), int((b + 0.5) * H / 2)) for a in range(3) for b in range(2)],
                    (2040, 2040, 2056, 2044), shard_counts=(2,))
