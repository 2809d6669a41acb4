"""Far-end ties of visibility rays: the in-place shortcuts against the reference's own boxes, tie by tie.

A shadow ray of camera::visible (src/camera.h:425-438) ends ON a surface, so that surface's triangle accepts at
t = max_t up to rounding, and whether the reference ever tests it is decided by aabb::hit (src/acceleration/aabb.h:41-66)
of the boxes above it.  The kernels settle most of these ties without walking the boxes (face bits made by the host,
one exact-plane distance).  The cross-check build of the library (libbdpt_amd_tiecheck.so) walks the boxes for EVERY tie
as well - and for every other accepted hit away from a triangle's outline, which the kernels take as reached - and counts
disagreements; tools/tie_check.py renders seven scenes with it in a child process (the binding loads one library per
process).  Asserted: no disagreement, the scenes do produce ties of every kind, and the frames equal the normal build's."""
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK_LIB = os.path.join(ROOT, "bidirectional-path-tracer_amd", "libbdpt_amd_tiecheck.so")


def run_tool(lib):
    env = dict(os.environ)
    if lib:
        env["BDPT_LIB"] = lib
    else:
        env.pop("BDPT_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tie_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_tie_shortcuts_agree_with_the_reference_boxes():
    assert os.path.exists(CHECK_LIB), "build it: make -C bidirectional-path-tracer_amd/csrc tiecheck (__graft_entry__.build does)"
    checked = run_tool(CHECK_LIB)
    assert checked["cross_check_build"]
    normal = run_tool(None)
    assert not normal["cross_check_build"]
    total = {}
    for name, c in checked["scenes"].items():
        print(name, {k: v for k, v in c.items() if k != "frame_sha256"})
        assert c["wrong_by_face_bits"] == 0 and c["wrong_by_exact_plane"] == 0, (name, c)
        assert c["plain_hits_unreached"] == 0, (name, c)
        assert c["frame_sha256"] == normal["scenes"][name]["frame_sha256"], name + ": the cross-check build answers with the reference's walk; the frame must not change"
        for k, v in c.items():
            if isinstance(v, int):
                total[k] = total.get(k, 0) + v
    # the scenes exercise every road
    assert total["ties"] > 1_000_000 and total["settled_by_face_bits"] > 100_000 and total["settled_by_exact_plane"] > 10_000
    assert total["left_to_boxes"] > 0 and total["plain_hits"] > 1_000_000
    for name in ("coffee", "coplanar", "lattice1m"):
        assert checked["scenes"][name]["ties"] > 0, name
