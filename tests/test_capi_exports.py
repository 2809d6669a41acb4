"""CPU-side checks of the product library: it loads, exports every symbol include/bdpt_amd.h declares,
and its host-only entry points (camera set-up, tone mapping, argument validation) agree with the oracle.
No kernel is launched here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

capi = importlib.import_module("bidirectional-path-tracer_amd.capi")


def test_library_is_built_in_tree():
    assert os.path.exists(capi.LIB_PATH), "libbdpt_amd.so must be built in-tree (python -c 'import __graft_entry__ as g; g.build()')"


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "bdpt_amd.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(bdpt_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_header():
    assert C.sizeof(capi.Material) == 48
    assert C.sizeof(capi.CameraPublic) == 16 * 8 + 16
    assert C.sizeof(capi.CameraState) == 15 * 8 + 16 + 7 * 8
    assert C.sizeof(capi.RenderParams) == 8 + 8 * 4
    assert C.sizeof(capi.Stats) == 88
    assert C.sizeof(capi.SceneInfo) == 32
    assert C.sizeof(capi.SceneFileView) == 144 + 5 * 8 + 4 * 4


@pytest.mark.parametrize("builder", ["cornell", "showcase", "nolight"])
def test_camera_initialize_matches_oracle(scenes, oracle, builder):
    flat = {"cornell": scenes.cornell_box(48, 16, 10), "showcase": scenes.materials_showcase(48, 16, 8),
            "nolight": scenes.no_light_room(24, 4, 5)}[builder]
    st = capi.camera_initialize(flat.camera)
    cam15, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    assert np.array_equal(st.as_array().view(np.uint64), cam15.view(np.uint64))
    assert (st.image_height, st.sqrt_spp, st.max_depth) == (H, sqrt_spp, flat.camera.max_depth)


def test_camera_initialize_matches_reference_golden(scenes):
    g = load_golden("cornell24")
    st = capi.camera_initialize(scenes.cornell_box(48, 16, 10).camera)
    assert np.array_equal(st.as_array()[:12].view(np.uint64), g["camera"][:12].view(np.uint64))


def test_camera_initialize_rejects_bad_input(scenes):
    cam = scenes.cornell_box(48, 16, 10).camera
    cam.image_width = 0
    with pytest.raises(capi.BdptError):
        capi.camera_initialize(cam)


def test_camera_initialize_defocus_disk_matches_oracle(scenes, oracle):
    flat = scenes.showcase_with_defocus(32, 16, 6)
    st = capi.camera_initialize(flat.camera)
    cam22, H, sqrt_spp = oracle.camera_initialize(flat.camera)
    assert st.defocus_angle == 4.0 and any(st.defocus_disk_u) and any(st.defocus_disk_v)
    assert np.array_equal(st.as_array().view(np.uint64), cam22.view(np.uint64))


def test_sqrt_spp_truncation(scenes):
    cam = scenes.cornell_box(8, 512, 10).camera     # 512 spp -> 22 x 22 = 484 samples (src/camera.h:164)
    assert capi.camera_initialize(cam).sqrt_spp == 22


def test_colors_to_rgb8_matches_reference():
    g = load_golden("cornell24")
    r = load_golden("cornell24_rgb8")
    out = capi.colors_to_rgb8(g["framebuffer"], int(r["spp"][0]))
    assert np.array_equal(out.reshape(-1), r["rgb8"])
    assert capi.colors_to_rgb8(np.full((1, 2, 3), np.nan), 4).tolist() == [[[0, 0, 0], [0, 0, 0]]]


def test_context_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.BdptError):
        capi.Context(0)
