"""The CPU restatement of composite scenes (oracle/bdpt_oracle.cpp: instances, textures, media) against vectors the
reference's own classes produced (tests/golden/make_golden.py, golden_composite).  Bit for bit: perlin tables, the
world's tree, hit records, visibility, framebuffers of both estimators, per-sample colours, BvhStats and draw counts."""
import numpy as np
import pytest

from conftest import COMPOSITE_CASES, composite_scene


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    bad = bits(a) != bits(b)
    assert not bad.any(), "%s: %d of %d values differ, first at %s: %r vs %r" % (
        what, bad.sum(), bad.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


@pytest.fixture(scope="module", params=sorted(COMPOSITE_CASES))
def case(request, scenes, oracle):
    flat, g = composite_scene(scenes, request.param)
    return request.param, flat, oracle.scene(flat), g


def test_perlin_tables(case):
    """perlin::perlin (perlin.h:6-14,59-73) on the stream the reference harness gave the constructor."""
    name, flat, sc, g = case
    tabs = sc.perlin_tables()
    assert len(tabs) == len(flat.perlin_tables)
    for a, b in zip(tabs, flat.perlin_tables):
        assert_bit_equal(a, b, name)


def test_world_tree(case):
    name, flat, sc, g = case
    assert np.array_equal(sc.world_leaves(), g["world_leaves"]), name
    assert_bit_equal(sc.world_boxes(), g["world_node_boxes"], name)


def check_hits(name, hits, ref):
    assert np.array_equal(hits[:, 0], ref[:, 0]), "%s: hit ids differ at %s" % (name, np.flatnonzero(hits[:, 0] != ref[:, 0])[:8])
    surface = ref[:, 0] != -1
    cols = [1, 4, 5, 6, 7, 8, 9, 10]
    assert_bit_equal(hits[surface][:, cols], ref[surface][:, cols], name + ": t, front_face, p, normal")
    tri = ref[:, 0] >= 0
    assert_bit_equal(hits[tri][:, 2:4], ref[tri][:, 2:4], name + ": u, v")        # (a medium leaves u, v unset)


def test_hits_and_visibility(case):
    name, flat, sc, g = case
    hits, stats = sc.trace_closest(g["rays"])
    check_hits(name, hits, g["hits"])
    assert np.array_equal(stats[:6], g["hits_stats"][:6]), name + ": BvhStats of the hit queries"
    assert np.array_equal(sc.visible(g["pairs"]), g["visible"]), name
    if "adv_rays" in g:
        check_hits(name + " (adversarial)", sc.trace_closest(g["adv_rays"])[0], g["adv_hits"])
        assert np.array_equal(sc.visible(g["adv_pairs"]), g["adv_visible"]), name


def test_render(case, oracle):
    name, flat, sc, g = case
    cam, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    W = flat.camera.image_width
    seed = int(g["render_seed"][0])
    fb, stats, _ = sc.render(cam, W, sqrt_spp, flat.camera.max_depth, seed, (0, 0, W, H))
    assert_bit_equal(fb, g["framebuffer"], name + ": framebuffer")
    assert np.array_equal(stats[:6], g["render_stats"][:6]), name + ": BvhStats"
    assert stats[7] == g["render_stats"][7], name + ": random draws (a medium draws inside the walk)"
    fb, stats = sc.render_unidirectional(cam, W, sqrt_spp, flat.camera.max_depth, seed, (0, 0, W, H))
    assert_bit_equal(fb, g["uni_framebuffer"], name + ": unidirectional framebuffer")
    assert np.array_equal(stats[:6], g["uni_render_stats"][:6]), name + ": unidirectional BvhStats"
    assert stats[7] == g["uni_render_stats"][7], name + ": unidirectional random draws"


def test_samples(case, oracle):
    name, flat, sc, g = case
    cam, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    rect = [int(v) for v in g["sample_rect"]]
    _, _, smp = sc.render(cam, flat.camera.image_width, sqrt_spp, flat.camera.max_depth, int(g["render_seed"][0]), rect,
                          per_sample=True)
    assert_bit_equal(smp, g["samples"].reshape(smp.shape), name + ": colour and draw count of every sample")
