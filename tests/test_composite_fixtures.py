"""CPU-side checks of the composite-scene fixtures (instances, textures, media): the scene builders still produce the
bytes the reference harness was given, and the recorded texture state has the shape the C ABI takes."""
import numpy as np
import pytest

from conftest import COMPOSITE_CASES, composite_scene


@pytest.mark.parametrize("name", sorted(COMPOSITE_CASES))
def test_fixture_matches_scene_builder(scenes, name):
    flat, g = composite_scene(scenes, name)
    n_noise = sum(1 for t in flat.textures if t.type == scenes.TEX_NOISE)
    n_image = sum(1 for t in flat.textures if t.type == scenes.TEX_IMAGE)
    assert len(flat.perlin_tables) == n_noise and len(flat.image_bytes) == n_image
    for tab in flat.perlin_tables:
        rv = tab[:768].reshape(256, 3)
        assert np.allclose((rv ** 2).sum(axis=1), 1.0, atol=1e-12)            # unit_vector(vec3::random(-1, 1)), perlin.h:8
        for k in range(3):
            assert sorted(tab[768 + 256 * k: 1024 + 256 * k].astype(int)) == list(range(256))     # permutations, perlin.h:59-73
    W = flat.camera.image_width
    assert g["framebuffer"].shape == (int(g["camera"][12]), W, 3)
    assert np.isfinite(g["framebuffer"]).all()
    hit = g["hits"][:, 0]
    assert ((hit >= -1) | (hit <= -2)).all()
    objs = -(g["world_leaves"][g["world_leaves"] < -1] + 2)
    assert set(objs.tolist()) == set(range(len(flat.objects)))               # every object is a leaf of the world's tree
    if name == "cx_smoke":
        assert (hit < -1).sum() > 100                                         # rays that end inside a medium
        assert g["render_stats"][7] > 0
    if name == "cx_textures":
        img = flat.image_bytes[0]
        assert img.shape == (12, 16, 3) and img.dtype == np.uint8
        # the reference's loader decodes to linear floats (gamma 2.2) and back to bytes (rtw_stb_image.h:90-112)
        src = scenes.test_image()
        want = np.where(src == 255, 255, np.floor(256.0 * (src / 255.0) ** 2.2)).astype(np.uint8)
        assert np.abs(img.astype(int) - want.astype(int)).max() <= 1
