"""Parity of the HIP path (through the C ABI, libbdpt_amd.so) with the oracle and with the golden
vectors produced by the reference's own headers.  Everything here is FP64 and compared BIT FOR BIT:
hit records, visibility booleans, per-pixel framebuffer sums, work counters."""
import hashlib
import importlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden, sha256_of

pytestmark = pytest.mark.gpu

capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    bad = bits(a) != bits(b)
    assert not bad.any(), "%s: %d of %d values differ, first at %s: %r vs %r" % (
        what, bad.sum(), bad.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


CASES = GOLDEN_CASES


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request, scenes, ctx):
    flat = CASES[request.param](scenes)
    g = load_golden(request.param)
    assert hashlib.sha256(flat.to_bytes()).digest() == bytes(g["scene_sha256"])
    sc = ctx.scene(flat)
    yield request.param, flat, sc, g
    sc.close()


# ------------------------------------------------------------------ device arithmetic the parity argument rests on
def test_device_ieee_arithmetic(ctx):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-1e3, 1e3, 200000), rng.uniform(0, 1, 200000) ** 8, [0.0, 1.0, 1e-300, 1e300, 555.0]])
    b = np.concatenate([rng.uniform(-1e3, 1e3, 200000), rng.uniform(0, 1, 200000) ** 8 + 1e-12, [1.0, 3.0, 7.0, 1e-10, 0.001]])
    with np.errstate(all="ignore"):
        assert_bit_equal(ctx.device_arith(0, a, b), a / b, "division is correctly rounded")
        assert_bit_equal(ctx.device_arith(7, a, b), 1.0 / a, "reciprocal is correctly rounded")
        assert_bit_equal(ctx.device_arith(1, np.abs(a), b), np.sqrt(np.abs(a)), "sqrt is correctly rounded")
        assert_bit_equal(ctx.device_arith(3, a, b), a * b + a, "a*b+a is NOT contracted into an fma")
        assert_bit_equal(ctx.device_arith(4, a, b), np.fmin(a, b), "fmin")
        assert_bit_equal(ctx.device_arith(5, a, b), np.fmax(a, b), "fmax")
    fused = ctx.device_arith(2, a, b)
    assert (bits(fused) != bits(a * b + a)).any(), "explicit fma must still be a single rounding"


def test_device_sincos_and_pow5_match_oracle(ctx, oracle):
    rng = np.random.default_rng(2)
    r1 = np.concatenate([rng.uniform(0, 1, 1 << 20), [0.0, 0.25, 0.5, 0.75, 1 - 2.0 ** -53, 2.0 ** -60, 1e-300]])
    phi = 2 * 3.1415926535897932385 * r1
    s, c = ctx.device_sincos(phi)
    so, co = oracle.sincos(phi)
    assert_bit_equal(s, so, "sin")
    assert_bit_equal(c, co, "cos")
    x = np.concatenate([rng.uniform(0, 2, 1 << 18), [0.0, 1.0, 2.0, 1e-200]])
    assert_bit_equal(ctx.device_arith(6, x, x), oracle.pow5(x), "pow5")


def test_device_random_stream_matches_oracle(ctx, oracle):
    for seed, pixel, sample in [(0, 0, 0), (7, 12345, 3), (2 ** 63 + 5, 1920 * 1080 - 1, 1023)]:
        assert_bit_equal(ctx.device_random_doubles(seed, pixel, sample, 4096), oracle.rng(seed, pixel, sample, 4096),
                         "stream (%d,%d,%d)" % (seed, pixel, sample))


# ------------------------------------------------------------------ BVH build (host) and traversal (device)
def test_bvh_matches_reference(case):
    name, flat, sc, g = case
    info = sc.info()
    assert info.n_triangles == flat.n_triangles
    if "dfs_leaves" in g:
        assert np.array_equal(sc.dfs_leaves(), g["dfs_leaves"])
        assert_bit_equal(sc.node_boxes(), g["node_boxes"], "node boxes")
        assert info.n_nodes == len(g["node_boxes"])
    else:       # large scene: the fixture holds the SHA-256 of the tables
        assert np.array_equal(sha256_of(sc.dfs_leaves().astype(np.uint32)), g["dfs_leaves_sha256"])
        assert np.array_equal(sha256_of(sc.node_boxes()), g["node_boxes_sha256"])
        assert info.n_nodes == int(g["n_nodes"][0])


def test_closest_hits_match_reference(case):
    name, flat, sc, g = case
    if "rays" not in g:
        pytest.skip("no ray fixture")
    hits, st = sc.trace_closest(g["rays"], want_stats=True)
    assert_bit_equal(hits, g["hits"], "hit records (triangle id, t, u, v, front_face, p, normal)")
    got = [st.rays_traced, st.aabb_tests, st.aabb_hits, st.bvh_node_visits, st.triangle_tests, st.triangle_hits]
    assert got[1:] == [int(v) for v in g["hits_stats"][1:6]], "BvhStats counters of the traversal"
    # without counters the query runs through the traversal kernel of the render pipeline (four-wide hierarchy)
    assert_bit_equal(sc.trace_closest(g["rays"]), g["hits"], "hit records, pipeline traversal kernel")
    # ... or, on request, through the counter-free variant of the walk over the reference's tree
    sc.ctx.set_query_walker(capi.WALKER_REFERENCE_TREE)
    try:
        assert_bit_equal(sc.trace_closest(g["rays"]), g["hits"], "hit records, reference-tree walk without counters")
    finally:
        sc.ctx.set_query_walker(capi.WALKER_PIPELINE)


def test_visible_matches_reference(case):
    name, flat, sc, g = case
    if "pairs" not in g:
        pytest.skip("no visibility fixture")
    assert np.array_equal(sc.visible(g["pairs"]), g["visible"]), "pipeline visibility kernel"
    sc.ctx.set_query_walker(capi.WALKER_REFERENCE_TREE)
    try:
        assert np.array_equal(sc.visible(g["pairs"]), g["visible"]), "reference-tree walk"
    finally:
        sc.ctx.set_query_walker(capi.WALKER_PIPELINE)


@pytest.mark.parametrize("walker", ["pipeline", "reference_tree"])
def test_adversarial_rays_and_pairs(case, walker):
    """The places where rounding or an exact tie decides what the reference returns (rays through vertices, edge
    midpoints and corners of bvh_node boxes, axis-parallel and in-plane rays, t at 0.001, upper bounds at the hit
    distance; pairs with end points on box faces, at vertices, in one plane, closer than the offsets), through the
    traversal kernels of the render pipeline and through the walk over the reference's tree."""
    name, flat, sc, g = case
    if "adv_rays" not in g:
        pytest.skip("no adversarial fixture")
    sc.ctx.set_query_walker(capi.WALKER_PIPELINE if walker == "pipeline" else capi.WALKER_REFERENCE_TREE)
    try:
        assert_bit_equal(sc.trace_closest(g["adv_rays"]), g["adv_hits"], "adversarial hit records")
        got = sc.visible(g["adv_pairs"])
        bad = np.flatnonzero(got != g["adv_visible"])
        assert len(bad) == 0, "%d of %d adversarial pairs differ, first %d: %r" % (len(bad), len(got), bad[0], g["adv_pairs"][bad[0]])
    finally:
        sc.ctx.set_query_walker(capi.WALKER_PIPELINE)


def test_empty_inputs(case):
    name, flat, sc, g = case
    assert sc.trace_closest(np.zeros((0, 8))).shape == (0, 11)
    assert sc.visible(np.zeros((0, 6))).shape == (0,)


# ------------------------------------------------------------------ the render path
def test_render_matches_reference_bit_for_bit(case):
    name, flat, sc, g = case
    cam = capi.camera_initialize(flat.camera)
    x0, y0, x1, y1 = [int(v) for v in g["render_rect"]]
    fb, st = sc.render(cam, seed=int(g["render_seed"][0]), rect=(x0, y0, x1, y1), want_stats=True)
    assert_bit_equal(fb[y0:y1, x0:x1], g["framebuffer"], "framebuffer sums")
    ref = [int(v) for v in g["render_stats"]]
    got = [st.rays_traced, st.aabb_tests, st.aabb_hits, st.bvh_node_visits, st.triangle_tests, st.triangle_hits]
    assert got == ref[:6], "BvhStats counters of the whole render"
    assert st.rng_draws == ref[7]
    assert st.samples == (y1 - y0) * (x1 - x0) * cam.sqrt_spp ** 2
    # the production paths (no counters, shadow rays stop at the first hit) give the same image:
    # the wavefront pipeline (default) and the single persistent kernel
    seed = int(g["render_seed"][0])
    assert_bit_equal(sc.render(cam, seed=seed, rect=(x0, y0, x1, y1)), fb, "wavefront pipeline")
    assert_bit_equal(sc.render(cam, seed=seed, rect=(x0, y0, x1, y1), megakernel=True), fb, "single kernel")
    # work counters of the wavefront pipeline: never more work than the reference path (light subpaths that cannot
    # contribute are not traced, visibility queries stop at the first hit)
    fb2, sw = sc.render(cam, seed=seed, rect=(x0, y0, x1, y1), want_stats="work")
    assert_bit_equal(fb2, fb, "wavefront pipeline with work counters")
    assert sw.samples == st.samples
    assert 0 < sw.rays_traced <= st.rays_traced and sw.rng_draws <= st.rng_draws
    assert sw.path_vertices <= st.path_vertices and sw.connection_pairs <= st.connection_pairs
    assert sw.shadow_rays <= st.shadow_rays


def test_render_matches_oracle_on_other_seeds(case, oracle):
    name, flat, sc, g = case
    cam = capi.camera_initialize(flat.camera)
    osc = oracle.scene(flat)
    W, H = cam.image_width, cam.image_height
    rect = (W // 4, H // 4, W // 4 + 6, H // 4 + 5)
    for seed in (12345, 2 ** 40 + 17):
        fb = sc.render(cam, seed=seed, rect=rect)
        ofb, ostats, _ = osc.render(cam.as_array(), W, cam.sqrt_spp, cam.max_depth, seed, rect)
        assert_bit_equal(fb[rect[1]:rect[3], rect[0]:rect[2]], ofb, "seed %d" % seed)
        outside = fb.copy()
        outside[rect[1]:rect[3], rect[0]:rect[2]] = 0
        assert not outside.any(), "pixels outside the rectangle must stay untouched"


def test_shards_sum_to_the_unsharded_image(case):
    name, flat, sc, g = case
    cam = capi.camera_initialize(flat.camera)
    full = sc.render(cam, seed=5)
    for count, rows in ((2, 1), (3, 4), (8, 2)):
        acc = np.zeros_like(full)
        for idx in range(count):
            part = sc.render(cam, seed=5, shard=(idx, count, rows))
            assert not (acc != 0)[part != 0].any(), "shards must own disjoint pixels"
            acc += part
        assert_bit_equal(acc, full, "%d shards of %d rows" % (count, rows))


def test_render_is_deterministic_and_keeps_unowned_pixels(case):
    name, flat, sc, g = case
    cam = capi.camera_initialize(flat.camera)
    a = sc.render(cam, seed=9)
    b = sc.render(cam, seed=9)
    assert_bit_equal(a, b, "two runs")
    marker = np.full_like(a, 7.25)
    part = sc.render(cam, seed=9, shard=(0, 2, 1), framebuffer=marker)
    assert (part[1::2] == 7.25).all() and (bits(part[0::2]) == bits(a[0::2])).all()


def test_more_than_256_samples_per_pixel(scenes, ctx, oracle):
    """spp above the workgroup size takes the multi-chunk accumulation path (484 = 22 x 22 samples,
    what 512 spp means in the reference, src/camera.h:164)."""
    flat = scenes.cornell_box(12, 512, 6)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    assert cam.sqrt_spp == 22
    rect = (3, 4, 8, 7)
    fb, st = sc.render(cam, seed=21, rect=rect, want_stats=True)
    ofb, ostats, _ = oracle.scene(flat).render(cam.as_array(), 12, 22, 6, 21, rect)
    assert_bit_equal(fb[4:7, 3:8], ofb, "484 spp")
    assert [st.rays_traced, st.aabb_tests, st.triangle_tests, st.rng_draws, st.shadow_rays, st.connection_pairs] == \
           [int(ostats[0]), int(ostats[1]), int(ostats[4]), int(ostats[7]), int(ostats[8]), int(ostats[9])]
    sc.close()


def test_spp_one_and_depth_zero(scenes, ctx, oracle):
    flat = scenes.materials_showcase(20, 1, 0)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = sc.render(cam, seed=1)
    assert not fb.any(), "depth 0: no bounce, no light path, black image"
    flat = scenes.materials_showcase(20, 1, 3)
    sc2 = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = sc2.render(cam, seed=1)
    ofb, _, _ = oracle.scene(flat).render(cam.as_array(), 20, 1, 3, 1, (0, 0, 20, cam.image_height))
    assert_bit_equal(fb, ofb, "1 spp")
    sc.close(); sc2.close()


def _moved(flat, scale, shift):
    """The scene and its camera scaled about the origin and then translated."""
    import copy
    out = copy.deepcopy(flat)
    out.vertices = [[v[k] * scale + shift[k % 3] for k in range(9)] for v in flat.vertices]
    out.camera.lookfrom = tuple(flat.camera.lookfrom[k] * scale + shift[k] for k in range(3))
    out.camera.lookat = tuple(flat.camera.lookat[k] * scale + shift[k] for k in range(3))
    return out


@pytest.mark.parametrize("scale,shift", [(1e-3, (0.0, 0.0, 0.0)), (37.0, (0.0, 0.0, 0.0)), (1.0, (1.0e5, -2.0e5, 3.0e4)),
                                         (1e4, (-7.0e6, 1.0e6, 5.0e6))])
def test_scene_scale_and_position(scenes, ctx, oracle, scale, shift):
    """Coordinates far from the origin or of a very different magnitude widen the FP32 box tests of the traversal
    hierarchy (their error allowance grows with the largest coordinate) but must not change a bit of the image:
    production pipeline = single-kernel path = oracle."""
    flat = _moved(scenes.materials_showcase(24, 16, 6), scale, shift)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = sc.render(cam, seed=5)
    ofb, _, _ = oracle.scene(flat).render(cam.as_array(), 24, cam.sqrt_spp, 6, 5, (0, 0, 24, cam.image_height))
    assert_bit_equal(fb, ofb, "moved scene, wavefront")
    assert_bit_equal(sc.render(cam, seed=5, megakernel=True), ofb, "moved scene, single kernel")
    sc.close()


def test_one_triangle_and_one_light(scenes, ctx, oracle):
    """The smallest scenes: a single triangle (one leaf, wrapped so that record 0 exists) and two triangles."""
    for n in (1, 2):
        flat = scenes.FlatScene()
        light = flat.add_material(scenes.Material(scenes.DIFFUSE_LIGHT, (4, 4, 4)))
        white = flat.add_material(scenes.Material(scenes.LAMBERTIAN, (.7, .7, .7)))
        flat.add_triangle((-1, 2, -1), (1, 2, -1), (0, 2, 1), light)      # faces down
        if n == 2:
            flat.add_triangle((-3, 0, -3), (0, 0, 3), (3, 0, -3), white)
        flat.camera = scenes.Camera(aspect_ratio=1.0, image_width=16, samples_per_pixel=16, max_depth=4,
                                    background=(0.1, 0.2, 0.3), vfov=60, lookfrom=(0, 1, -4), lookat=(0, 1, 0), vup=(0, 1, 0))
        sc = ctx.scene(flat)
        cam = capi.camera_initialize(flat.camera)
        fb = sc.render(cam, seed=2)
        ofb, _, _ = oracle.scene(flat).render(cam.as_array(), 16, cam.sqrt_spp, 4, 2, (0, 0, 16, cam.image_height))
        assert_bit_equal(fb, ofb, "%d triangle(s)" % n)
        assert_bit_equal(sc.render(cam, seed=2, megakernel=True), ofb, "%d triangle(s), single kernel" % n)
        sc.close()


def test_scene_without_lambertian_surfaces(scenes, ctx, oracle):
    """No lambertian triangle -> no vertex can ever be connected: the pipeline leaves out light subpaths and
    connections (and their storage) for the whole scene.  Same bits as the oracle, as the single-kernel path and as
    the counting modes."""
    flat = scenes.FlatScene()
    light = flat.add_material(scenes.Material(scenes.DIFFUSE_LIGHT, (6, 5, 4)))
    mirror = flat.add_material(scenes.Material(scenes.METAL, (.8, .8, .9), fuzz=0.05))
    glass = flat.add_material(scenes.Material(scenes.DIELECTRIC, ior=1.5))
    scenes.add_quad_triangles(flat, (-2, 4, -2), (4, 0, 0), (0, 0, 4), light)
    scenes.add_quad_triangles(flat, (-6, 0, -6), (12, 0, 0), (0, 0, 12), mirror)
    scenes.add_box_triangles(flat, (-1, 0.2, -1), (1, 2.2, 1), glass, 20.0)
    scenes.add_uv_sphere(flat, (2.5, 1.0, 0.5), 1.0, glass, 8, 16)
    flat.camera = scenes.Camera(aspect_ratio=1.5, image_width=36, samples_per_pixel=16, max_depth=8,
                                background=(0.05, 0.1, 0.2), vfov=45, lookfrom=(0, 3, -9), lookat=(0, 1, 0), vup=(0, 1, 0))
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = sc.render(cam, seed=17)
    ofb, ostats, _ = oracle.scene(flat).render(cam.as_array(), 36, cam.sqrt_spp, 8, 17, (0, 0, 36, cam.image_height))
    assert fb.max() > 0
    assert_bit_equal(fb, ofb, "no lambertian surface, wavefront")
    assert_bit_equal(sc.render(cam, seed=17, megakernel=True), ofb, "no lambertian surface, single kernel")
    fb2, st = sc.render(cam, seed=17, want_stats="work")
    assert_bit_equal(fb2, ofb, "no lambertian surface, counting mode")
    # the pipeline's own work counters: no light subpath was sampled, so fewer draws than the reference makes
    assert 0 < st.rng_draws < int(ostats[7]) and st.connection_pairs == 0 and st.shadow_rays == 0
    fb3, ref_st = sc.render(cam, seed=17, want_stats=True)          # reference-equivalent counters (single-kernel path)
    assert_bit_equal(fb3, ofb, "no lambertian surface, reference counters")
    assert ref_st.rng_draws == int(ostats[7]) and ref_st.rays_traced == int(ostats[0])
    sc.close()


def test_large_scene_uses_global_memory_path(scenes, ctx, oracle):
    """More nodes + triangles than the LDS staging limit: the traversal reads the tables from
    global memory; results are still the oracle's."""
    flat = scenes.cornell_box(16, 4, 5)
    mat = flat.add_material(scenes.Material(scenes.LAMBERTIAN, (.4, .6, .3)))
    for k in range(3):
        scenes.add_uv_sphere(flat, (150 + 120 * k, 90 + 40 * k, 200 + 60 * k), 70.0, mat, 16, 32)
    assert flat.n_triangles > 2000
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb, st = sc.render(cam, seed=4, want_stats=True)
    osc = oracle.scene(flat)
    ofb, ostats, _ = osc.render(cam.as_array(), 16, 2, 5, 4, (0, 0, 16, 16))
    assert_bit_equal(fb, ofb, "global-memory scene")
    assert [st.rays_traced, st.aabb_tests, st.aabb_hits, st.bvh_node_visits, st.triangle_tests, st.triangle_hits] == \
           [int(v) for v in ostats[:6]]
    assert np.array_equal(sc.dfs_leaves(), osc.dfs_leaves())
    sc.close()


@pytest.mark.parametrize("depth", [70, 300, 1200])
def test_deep_paths(scenes, ctx, oracle, depth):
    """max_depth far beyond the usual: a sample may have depth x depth pairs (rounds of their own in the connection kernel,
    larger pieces of the shadow queue), the 2 x depth vertex records of one sample decide how many the kernel stages at a
    time and with how many waves (depth 300: one wave per workgroup), and at depth 1200 they no longer fit a compute
    unit's LDS: that render takes the single-kernel path.  A closed room, so that paths do get long."""
    flat = scenes.FlatScene()
    grey = flat.add_material(scenes.Material(scenes.LAMBERTIAN, (.7, .7, .7)))
    red = flat.add_material(scenes.Material(scenes.LAMBERTIAN, (.7, .3, .3)))
    light = flat.add_material(scenes.Material(scenes.DIFFUSE_LIGHT, (30.0, 30.0, 30.0)))
    scenes.add_quad_triangles(flat, (0, 0, 0), (4, 0, 0), (0, 0, 4), grey)        # floor, ceiling, four walls: nothing escapes
    scenes.add_quad_triangles(flat, (0, 3, 0), (0, 0, 4), (4, 0, 0), grey)
    scenes.add_quad_triangles(flat, (0, 0, 0), (0, 3, 0), (4, 0, 0), red)
    scenes.add_quad_triangles(flat, (0, 0, 4), (4, 0, 0), (0, 3, 0), grey)
    scenes.add_quad_triangles(flat, (0, 0, 0), (0, 0, 4), (0, 3, 0), grey)
    scenes.add_quad_triangles(flat, (4, 0, 0), (0, 3, 0), (0, 0, 4), red)
    scenes.add_quad_triangles(flat, (1.9, 2.99, 1.9), (0.2, 0, 0), (0, 0, 0.2), light)
    W, spp = {70: (6, 4), 300: (4, 4), 1200: (3, 1)}[depth]
    flat.camera = scenes.Camera(aspect_ratio=1.0, image_width=W, samples_per_pixel=spp, max_depth=depth, background=(0, 0, 0), vfov=70,
                                lookfrom=(2, 1.5, 0.2), lookat=(2, 1.2, 3), vup=(0, 1, 0))
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb, st = sc.render(cam, seed=29, want_stats="work") if depth <= 300 else (sc.render(cam, seed=29), None)
    ofb, ostats, _ = oracle.scene(flat).render(cam.as_array(), W, cam.sqrt_spp, depth, 29, (0, 0, W, cam.image_height))
    assert ofb.max() > 0
    assert_bit_equal(fb, ofb, "depth %d" % depth)
    assert_bit_equal(sc.render(cam, seed=29), ofb, "depth %d, without counters" % depth)
    if st is not None:
        assert st.connection_pairs > 64 * st.samples        # (samples with more pairs than a wave has lanes)
    sc.close()


def test_largest_scene_staged_in_lds(scenes, ctx, oracle):
    """484 triangles: hierarchy + triangles still fit the 64 KiB the traversal kernels stage completely, and the copy of
    the reference's tree beside them (967 nodes) takes the dynamic LDS of the visibility traversal to ~140 KiB and of
    the connection kernel's own-triangle test to ~110 KiB - both far past the 64 KiB a kernel gets without asking.
    Frame (pipeline with and without work counters, single kernel), hit records and visibility against the oracle."""
    flat = scenes.cornell_spheres_480(40, 16, 8)
    assert 450 <= flat.n_triangles <= 520
    sc = ctx.scene(flat)
    osc = oracle.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    ofb, ostats, _ = osc.render(cam.as_array(), 40, cam.sqrt_spp, 8, 23, (0, 0, 40, cam.image_height))
    assert ofb.max() > 0
    assert_bit_equal(sc.render(cam, seed=23), ofb, "484 triangles, pipeline")
    fb, st = sc.render(cam, seed=23, want_stats="work")
    assert_bit_equal(fb, ofb, "484 triangles, pipeline with work counters")
    assert st.shadow_rays > 0 and st.connection_pairs > st.shadow_rays
    assert_bit_equal(sc.render(cam, seed=23, megakernel=True), ofb, "484 triangles, single kernel")
    # surface points of random rays, and visibility between them (far ends ON surfaces: the staged tree decides the ties)
    rng = np.random.default_rng(5)
    v = flat.vertex_array().reshape(-1, 3)
    lo, hi = v.min(axis=0), v.max(axis=0)
    n = 8000
    o = rng.uniform(lo, hi, (n, 3))
    rays = np.concatenate([o, rng.uniform(lo, hi, (n, 3)) - o, np.full((n, 1), 0.001), np.full((n, 1), np.inf)], axis=1)
    hits = sc.trace_closest(rays)
    ohits, _ = osc.trace_closest(rays)
    assert_bit_equal(hits, ohits, "484 triangles, hit records")
    pts = ohits[ohits[:, 0] >= 0][:, 5:8]
    pairs = np.concatenate([pts[rng.integers(0, len(pts), 6000)], pts[rng.integers(0, len(pts), 6000)]], axis=1)
    vis = sc.visible(pairs)
    ovis = osc.visible(pairs)
    assert np.array_equal(np.asarray(vis).astype(bool), np.asarray(ovis).astype(bool))
    assert 0 < np.asarray(ovis).astype(bool).sum() < len(pairs)
    sc.close()


@pytest.mark.parametrize("big", [False, True])
def test_more_materials_than_the_connection_stage_keeps_in_lds(scenes, ctx, oracle, big):
    """The connection stage works out the BSDF values of up to 128 materials once per workgroup; a scene with more
    reads the material table per pair.  Every triangle here has a material of its own (lambertian, metal and emitters
    mixed); small = whole scene staged in LDS, big = tables in global memory."""
    flat = scenes.FlatScene()
    import random
    rnd = random.Random(5)
    def own(kind):
        if kind == 0: return flat.add_material(scenes.Material(scenes.LAMBERTIAN, (rnd.uniform(.2, .9), rnd.uniform(.2, .9), rnd.uniform(.2, .9))))
        if kind == 1: return flat.add_material(scenes.Material(scenes.METAL, (rnd.uniform(.5, .9),) * 3, fuzz=rnd.uniform(0, .3)))
        return flat.add_material(scenes.Material(scenes.DIFFUSE_LIGHT, (rnd.uniform(2, 6),) * 3))
    n = 12 if big else 9
    for i in range(n):
        for j in range(n):
            kind = 2 if (i * n + j) % 23 == 0 else (1 if (i * n + j) % 7 == 3 else 0)
            x, z = -3 + 6.0 * i / n, -3 + 6.0 * j / n
            y = 0.0 if kind != 2 else 3.0
            flat.add_triangle((x, y, z), (x + 6.0 / n, y, z), (x, y + (0.3 if kind != 2 else 0), z + 6.0 / n), own(kind))
            flat.add_triangle((x + 6.0 / n, y, z + 6.0 / n), (x, y + (0.3 if kind != 2 else 0), z + 6.0 / n), (x + 6.0 / n, y, z), own(kind))
    if big:
        for k in range(3):
            scenes.add_uv_sphere(flat, (-1.5 + 1.5 * k, 1.0, 0.0), 0.6, own(0), 16, 32)
    assert len(flat.materials) > 128
    flat.camera = scenes.Camera(aspect_ratio=1.0, image_width=20, samples_per_pixel=16, max_depth=6,
                                background=(0.02, 0.03, 0.05), vfov=60, lookfrom=(0, 2.5, -6), lookat=(0, 0.5, 0), vup=(0, 1, 0))
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = sc.render(cam, seed=13)
    ofb, _, _ = oracle.scene(flat).render(cam.as_array(), 20, cam.sqrt_spp, 6, 13, (0, 0, 20, cam.image_height))
    assert fb.max() > 0
    assert_bit_equal(fb, ofb, "more than 128 materials, wavefront")
    assert_bit_equal(sc.render(cam, seed=13, megakernel=True), ofb, "more than 128 materials, single kernel")
    sc.close()


def test_reduce_framebuffer_over_rccl(ctx):
    """bdpt_reduce_framebuffer is one ncclAllReduce(sum, double) on the caller's communicator.  One GPU here, so
    the communicator has one rank and the sum must return the buffer unchanged; wrong arguments are refused."""
    import ctypes as C
    import torch
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("no RCCL library on this machine")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        fb = torch.arange(3 * 64 * 48, dtype=torch.float64, device="cuda") * 0.25 - 7.0
        want = fb.clone()
        capi.reduce_framebuffer(comm.value, fb.data_ptr(), fb.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(fb, want)
        capi.reduce_framebuffer(comm.value, fb.data_ptr(), 0)
        with pytest.raises(capi.BdptError):
            capi.reduce_framebuffer(0, fb.data_ptr(), fb.numel())
        with pytest.raises(capi.BdptError):
            capi.reduce_framebuffer(comm.value, 0, fb.numel())
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


WF_SETTINGS = [
    {"BDPT_WF_LDS_STACK": "0"},                                   # whole traversal stack in scratch
    {"BDPT_WF_LDS_STACK": "2"},                                   # two entries in LDS, pushes straddle the border
    {"BDPT_WF_LDS_STACK": "16", "BDPT_WF_LDS_KB": "1"},
    {"BDPT_WF_BATCH_LOG2": "12", "BDPT_WF_STREAMS": "3"},         # many small batches
    {"BDPT_WF_STREAMS": "1"},
    {"BDPT_WF_CHUNK": "16", "BDPT_WF_BLOCKS_PER_CU": "1"},
    {"BDPT_WF_REFILL": "1", "BDPT_WF_LEAF": "1"},
    {"BDPT_WF_REFILL": "64", "BDPT_WF_LEAF": "64"},
    {"BDPT_WF_BUDGET_GB": "1"},
    {"BDPT_WF_NO_FIRST_LEAF": "1"},                               # shadow rays start at the root (large scenes: the default tests the far end's triangle first)
    {"BDPT_WF_NO_FIRST_LEAF": "1", "BDPT_WF_LDS_STACK": "0"},
    {"BDPT_WF_PIXEL_MAJOR": "1"}, {"BDPT_WF_PIXEL_MAJOR": "0"},   # which samples are neighbours in the queues
    {"BDPT_WF_PIXEL_MAJOR": "1", "BDPT_WF_BATCH_LOG2": "12", "BDPT_WF_STREAMS": "2"},
    {"BDPT_WF_CONN_BUDGET": "16", "BDPT_WF_CONN_CAP": "5"},      # connection kernel: few records staged at a time, many passes
    {"BDPT_WF_CONN_BUDGET": "512", "BDPT_WF_CONN_CAP": "40"},
    {"BDPT_WF_REDO_BLOCKS": "7"},
    {"BDPT_WF_SORT_CELLS": "64"}, {"BDPT_WF_SORT_CELLS": "5", "BDPT_WF_PIXEL_MAJOR": "0"}, {"BDPT_WF_SORT_CELLS": "0"},
    {"BDPT_WF_NO_LDS_SCENE": "1"}, {"BDPT_WF_NO_OWN_TEST": "1"},  # a small scene on the kernels of a large one
]


@pytest.mark.parametrize("lds_scene", [True, False])
def test_image_does_not_depend_on_launch_settings(scenes, ctx, lds_scene, monkeypatch):
    """Batch size, streams in flight, queue chunks, occupancy and where the traversal stack lives only change the
    schedule: every setting returns the bits of the default one (small scene staged in LDS / larger one in global
    memory with a stack deeper than its LDS part)."""
    flat = scenes.cornell_box(48, 16, 8)
    if not lds_scene:
        mat = flat.add_material(scenes.Material(scenes.LAMBERTIAN, (.4, .6, .3)))
        glass = flat.add_material(scenes.Material(scenes.DIELECTRIC, (1, 1, 1), ior=1.5))
        for k in range(3):
            scenes.add_uv_sphere(flat, (150 + 120 * k, 90 + 40 * k, 200 + 60 * k), 70.0, glass if k == 1 else mat, 16, 32)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    want = sc.render(cam, seed=11)
    assert want.max() > 0
    sc.close()
    for setting in WF_SETTINGS:
        with monkeypatch.context() as m:
            for k, v in setting.items():
                m.setenv(k, v)
            c2 = capi.Context(0)              # the settings are read when a context is created
            sc2 = c2.scene(flat)
            assert_bit_equal(sc2.render(cam, seed=11), want, str(setting))
            sc2.close()
            c2.close()


def test_bad_arguments_are_rejected(scenes, ctx):
    flat = scenes.cornell_box(8, 4, 3)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    with pytest.raises(capi.BdptError, match="shard_index"):
        sc.render(cam, shard=(3, 2, 1))
    cam.max_depth = -1
    with pytest.raises(capi.BdptError, match="max_depth"):
        sc.render(cam)
    bad = scenes.cornell_box(8, 4, 3)
    bad.tri_material[0] = 99
    with pytest.raises(capi.BdptError, match="material index"):
        ctx.scene(bad)
    empty = scenes.FlatScene()
    empty.add_material(scenes.Material(scenes.LAMBERTIAN, (.5, .5, .5)))
    with pytest.raises(capi.BdptError, match="no triangles"):
        ctx.scene(empty)
    sc.close()


def test_rgb8_on_device_matches_reference(scenes, ctx):
    import torch
    g = load_golden("cornell24")
    r = load_golden("cornell24_rgb8")
    fb = torch.from_numpy(g["framebuffer"].copy()).cuda()
    out = torch.zeros(fb.numel(), dtype=torch.uint8, device="cuda")
    capi._check(ctx.lib.bdpt_colors_to_rgb8_device(ctx.h, fb.data_ptr(), 48, 48, int(r["spp"][0]), out.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), r["rgb8"])


def test_device_render_timers(scenes, ctx):
    """HIP-event timers of a device-buffer render: the whole call on the caller's stream, and one event
    pair per traversal launch; the framebuffer is the one the host-buffer entry point returns."""
    import torch
    flat = scenes.cornell_box(40, 16, 5)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    want = sc.render(cam, seed=4)
    fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
    sc.render_device(cam, fb.data_ptr(), seed=4, time_kernel=True, time_stages=True)
    assert sc.last_kernel_ms() > 0.0
    ms0, n0 = sc.last_stage_ms(0)
    ms1, n1 = sc.last_stage_ms(1)
    # per batch: depth + (depth - 1) closest-hit launches, and one visibility launch per pass of the connection stage
    batches = n0 // (2 * 5 - 1)
    assert n0 > 0 and n0 % (2 * 5 - 1) == 0 and n1 >= batches and n1 % batches == 0
    assert ms0 > 0.0 and ms1 >= 0.0
    assert_bit_equal(fb.cpu().numpy(), want, "device render")
    sc.render_device(cam, fb.data_ptr(), seed=4)
    with pytest.raises(capi.BdptError):
        sc.last_stage_ms(0)


def test_command_line_renderer_end_to_end(tmp_path, oracle):
    """scene file -> bdpt_render (loader, GPU render, tone mapping, PNG) against the same scene rendered by the CPU
    oracle: the decoded image must be identical byte for byte."""
    import subprocess
    from PIL import Image
    exe = os.path.join(ROOT, "bidirectional-path-tracer_amd", "bdpt_render")
    if not os.path.exists(exe):
        pytest.skip("command-line renderer not built")
    path = os.path.join(ROOT, "tests", "data", "scene_typed.yaml")
    r = subprocess.run([exe, path, "--seed", "3"], capture_output=True, text=True, cwd=tmp_path, timeout=100)
    assert r.returncode == 0, r.stderr
    assert "Triangles: 980" in r.stdout
    img = np.array(Image.open(tmp_path / "output" / "typed_scene.png").convert("RGB"))
    flat = capi.load_scene_file(path)
    cam = capi.camera_initialize(flat.camera)
    ofb, _, _ = oracle.scene(flat).render(cam.as_array(), cam.image_width, cam.sqrt_spp, cam.max_depth, 3,
                                          (0, 0, cam.image_width, cam.image_height))
    want = capi.colors_to_rgb8(ofb, cam.sqrt_spp ** 2)
    assert img.shape == want.shape and (img == want).all()


UNI_CASES = {
    "uni_cornell24": lambda s: s.cornell_box(32, 16, 10),
    "uni_showcase": lambda s: s.materials_showcase(32, 9, 8),
    "uni_nolight": lambda s: s.no_light_room(16, 4, 5),
    "uni_cornell_d1": lambda s: s.cornell_box(16, 4, 1),
}


@pytest.mark.parametrize("name", sorted(UNI_CASES))
def test_unidirectional_estimator_matches_reference(name, scenes, ctx):
    """BDPT_RENDER_UNIDIRECTIONAL = camera::path_trace_color (src/camera.h:255-292), bit for bit against the
    reference's own function, including its BvhStats counters and the number of random draws."""
    flat = UNI_CASES[name](scenes)
    g = load_golden(name)
    assert hashlib.sha256(flat.to_bytes()).digest() == bytes(g["scene_sha256"])
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    seed = int(g["render_seed"][0])
    fb = sc.render(cam, seed=seed, unidirectional=True)
    assert_bit_equal(fb, g["framebuffer"], "unidirectional framebuffer, wavefront pipeline")
    assert_bit_equal(sc.render(cam, seed=seed, unidirectional=True, megakernel=True), g["framebuffer"], "unidirectional framebuffer, single kernel")
    fbw, sw = sc.render(cam, seed=seed, unidirectional=True, want_stats="work")
    assert_bit_equal(fbw, g["framebuffer"], "unidirectional framebuffer, pipeline with work counters")
    assert sw.samples == cam.image_width * cam.image_height * cam.sqrt_spp ** 2 and sw.shadow_rays == 0 and sw.connection_pairs == 0
    fb2, st = sc.render(cam, seed=seed, unidirectional=True, want_stats=True)
    assert_bit_equal(fb2, g["framebuffer"], "unidirectional framebuffer, counting mode")
    want = [int(v) for v in g["render_stats"]]
    assert [st.rays_traced, st.aabb_tests, st.aabb_hits, st.bvh_node_visits, st.triangle_tests, st.triangle_hits] == want[:6]
    assert st.rng_draws == want[7]
    # the pipeline draws what the reference draws (its last level included) and traces one ray fewer per sample that
    # runs out of depth (the reference counts the call that returns at once, camera.h:256-259)
    assert sw.rng_draws == st.rng_draws and 0 < sw.rays_traced <= st.rays_traced
    sc.close()
