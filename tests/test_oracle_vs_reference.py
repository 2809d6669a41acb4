"""Pins the CPU restatement (oracle/bdpt_oracle.cpp) against golden vectors produced by the
reference's OWN headers (oracle/_ref, tests/golden/make_golden.py).  Everything is compared
bit for bit: FP64 framebuffers, hit records, BVH topology and boxes, counters, RNG draws."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden, sha256_of


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    bad = bits(a) != bits(b)
    assert not bad.any(), "%s: %d of %d values differ, first at %s: %r vs %r" % (
        what, bad.sum(), bad.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


CASES = GOLDEN_CASES


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request, scenes, oracle):
    flat = CASES[request.param](scenes)
    g = load_golden(request.param)
    assert hashlib.sha256(flat.to_bytes()).digest() == bytes(g["scene_sha256"]), "scene builder drifted from the fixture"
    return request.param, flat, oracle.scene(flat), g


def test_camera_initialize(case, oracle):
    name, flat, sc, g = case
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    assert_bit_equal(cam15[:12], g["camera"][:12], "camera::initialize")
    assert H == int(g["camera"][12]) and sqrt_spp == int(g["camera"][13])
    if len(g["camera"]) >= 20:      # fixtures made since the defocus disk is exported
        assert_bit_equal(cam15[15:21], g["camera"][14:20], "defocus disk")


def test_triangle_precompute(case):
    name, flat, sc, g = case
    if "triangle_data" in g:
        assert_bit_equal(sc.triangle_data(), g["triangle_data"], "triangle ctor")
    else:       # large scene: the fixture holds the SHA-256 of the table
        assert np.array_equal(sha256_of(sc.triangle_data()), g["triangle_data_sha256"]), "triangle ctor"


def test_bvh_topology_and_boxes(case):
    name, flat, sc, g = case
    if "dfs_leaves" in g:
        assert np.array_equal(sc.dfs_leaves(), g["dfs_leaves"]), "DFS leaf order"
        assert_bit_equal(sc.node_boxes(), g["node_boxes"], "node boxes")
    else:
        assert np.array_equal(sha256_of(sc.dfs_leaves().astype(np.uint32)), g["dfs_leaves_sha256"]), "DFS leaf order"
        assert np.array_equal(sha256_of(sc.node_boxes()), g["node_boxes_sha256"]), "node boxes"


def test_closest_hits(case):
    name, flat, sc, g = case
    if "rays" not in g:
        pytest.skip("no ray fixture for this case")
    hits, stats = sc.trace_closest(g["rays"])
    assert_bit_equal(hits, g["hits"], "hit records")
    # bvh_nodes_built is reset by camera::render (src/camera.h:49) after the tree exists: the reference reports 0
    assert np.array_equal(stats[:6], g["hits_stats"][:6]), "BvhStats of the hit queries"
    assert stats[6] == (len(g["node_boxes"]) if "node_boxes" in g else int(g["n_nodes"][0]))


def test_adversarial_rays_and_pairs(case):
    """Rays through vertices, edge midpoints and corners of bvh_node boxes, axis-parallel and in-plane rays, hits at
    t = 0.001 and upper bounds at the hit distance up to an ulp; visibility pairs with end points on box faces, at
    vertices, in one plane, closer than the offsets (tests/golden/make_golden.py: adversarial_rays / _pairs)."""
    name, flat, sc, g = case
    if "adv_rays" not in g:
        pytest.skip("no adversarial fixture for this case")
    hits, _ = sc.trace_closest(g["adv_rays"])
    assert_bit_equal(hits, g["adv_hits"], "adversarial hit records")
    assert np.array_equal(sc.visible(g["adv_pairs"]), g["adv_visible"])


def test_visible(case):
    name, flat, sc, g = case
    if "pairs" not in g:
        pytest.skip("no visibility fixture for this case")
    vis = sc.visible(g["pairs"])
    assert np.array_equal(vis, g["visible"])
    # the endpoint tie makes a large share of unoccluded pairs "blocked"; make sure the fixture has both outcomes
    assert 0 < g["visible"].sum() < len(g["visible"])


def test_render_framebuffer_and_counters(case, oracle):
    name, flat, sc, g = case
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    fb, stats, _ = sc.render(cam15, flat.camera.image_width, sqrt_spp, flat.camera.max_depth,
                             int(g["render_seed"][0]), g["render_rect"])
    assert_bit_equal(fb, g["framebuffer"], "framebuffer sums")
    assert np.array_equal(stats[:6], g["render_stats"][:6]), "BvhStats"
    assert stats[7] == g["render_stats"][7], "RNG draw total"


def test_per_sample_radiance_and_draw_counts(case, oracle):
    name, flat, sc, g = case
    if "samples" not in g:
        pytest.skip("no per-sample fixture for this case")
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    _, _, smp = sc.render(cam15, flat.camera.image_width, sqrt_spp, flat.camera.max_depth,
                          int(g["render_seed"][0]), g["sample_rect"], per_sample=True)
    assert_bit_equal(smp, g["samples"], "per-sample radiance / draws")


def test_serial_mt19937_matches_unshimmed_reference(scenes, oracle):
    """The restatement driven by ONE serial default-seeded mt19937 equals the reference compiled with
    its real src/main.h - pins random_double's word->double formula and the global draw order."""
    g = load_golden("serial_mt")
    flat = scenes.cornell_box(16, 4, 5)
    assert hashlib.sha256(flat.to_bytes()).digest() == bytes(g["scene_sha256"])
    assert_bit_equal(oracle.mt(64), g["rand"], "mt19937 doubles")
    # known answers measured on the unmodified reference
    assert oracle.mt(4).tolist() == [0.1354770042967805, 0.8350085899945795, 0.96886777112423139, 0.2210340429827049]
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    fb, stats = oracle.scene(flat).render_serial_mt(cam15, 16, H, sqrt_spp, 5)
    assert_bit_equal(fb, g["framebuffer"], "serial render")
    assert np.array_equal(stats[:6], g["stats"][:6])


def test_colors_to_rgb8(oracle):
    g = load_golden("cornell24")
    r = load_golden("cornell24_rgb8")
    out = oracle.colors_to_rgb8(g["framebuffer"], int(r["spp"][0]))
    assert np.array_equal(out, r["rgb8"])
    # NaN is undefined behaviour in the reference; defined as 0 here
    fb = np.full((1, 1, 3), np.nan)
    assert oracle.colors_to_rgb8(fb, 1).tolist() == [0, 0, 0]


def test_glibc_libm_noise_floor():
    """Same reference render with glibc's own sin/cos/pow instead of the correctly rounded ones:
    a few pixels move (1-ulp direction changes flip endpoint-tie visibility tests); this is the
    noise floor the 'per-pixel L2 < 1e-3' target sits on, recorded here so a regression is visible."""
    g = load_golden("cornell24")
    a, b = g["framebuffer"] / 16.0, g["framebuffer_glibc_libm"] / 16.0
    differing = (np.abs(a - b).max(axis=2) > 0).mean()
    rel_l2 = np.linalg.norm(a - b) / np.linalg.norm(a)
    assert differing < 0.05
    assert rel_l2 < 2e-3


UNI_CASES = {
    "uni_cornell24": lambda s: s.cornell_box(32, 16, 10),
    "uni_showcase": lambda s: s.materials_showcase(32, 9, 8),
    "uni_nolight": lambda s: s.no_light_room(16, 4, 5),
    "uni_cornell_d1": lambda s: s.cornell_box(16, 4, 1),
}


@pytest.mark.parametrize("name", sorted(UNI_CASES))
def test_unidirectional_estimator(name, scenes, oracle):
    """camera::path_trace_color (src/camera.h:255-292) of the reference, driven for every sample of the frame."""
    flat = UNI_CASES[name](scenes)
    g = load_golden(name)
    assert hashlib.sha256(flat.to_bytes()).digest() == bytes(g["scene_sha256"]), "scene builder drifted from the fixture"
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    W = flat.camera.image_width
    fb, stats = oracle.scene(flat).render_unidirectional(cam15, W, sqrt_spp, flat.camera.max_depth, int(g["render_seed"][0]),
                                                         (0, 0, W, H))
    assert_bit_equal(fb, g["framebuffer"], "framebuffer sums")
    assert np.array_equal(stats[:6], g["render_stats"][:6]), "BvhStats"
    assert stats[7] == g["render_stats"][7], "RNG draw total"


def test_visibility_depends_on_the_reference_boxes(scenes, oracle):
    """camera::visible asks bvh_node::hit whether ANY triangle is accepted in [0.001, |b - a| - 0.001]; the far end point
    lies on a surface at exactly the end of that interval, and a bvh_node box whose face coincides with that surface is
    entered at the same t - when rounding makes aabb::hit fail there, the reference never tests the triangle.  So the
    answer is NOT "some triangle accepts": this test keeps that fact visible (the product's visibility walk re-checks
    the reference's boxes for exactly these hits)."""
    import ctypes as C
    flat = scenes.cornell_box(64, 16, 10)
    osc = oracle.scene(flat)
    rng = np.random.default_rng(1)
    n = 60000
    rays = np.concatenate([rng.uniform(50, 500, size=(n, 3)), rng.normal(size=(n, 3)), np.full((n, 1), 0.001), np.full((n, 1), np.inf)], axis=1)
    hits = osc.trace_closest(rays)[0]
    pts = hits[hits[:, 0] >= 0][:, 5:8]
    m = len(pts) // 2
    pairs = np.ascontiguousarray(np.concatenate([pts[:m], pts[m:2 * m]], axis=1))
    with_boxes = osc.visible(pairs)
    d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    oracle.lib.oracle_visible_all_triangles.argtypes = [C.c_void_p, d, C.c_uint64, d]
    any_triangle = np.zeros(len(pairs))
    oracle.lib.oracle_visible_all_triangles(osc.h, pairs, len(pairs), any_triangle)
    differ = with_boxes != any_triangle.astype(np.uint8)
    assert differ.any(), "expected pairs whose end-point triangle accepts but is culled by a reference box"
    # always in the same direction: the hierarchy can only hide an accepting triangle
    assert (with_boxes[differ] == 1).all() and differ.mean() < 0.02
