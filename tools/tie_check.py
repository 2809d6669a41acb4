"""Cross-check of the shortcuts that settle far-end ties of visibility rays in place, against the reference's own box walk
(src/acceleration/aabb.h:41-66 from the triangle's bvh_node up to the root; src/camera.h:425-438).

Needs the cross-check build of the library (`make -C bidirectional-path-tracer_amd/csrc tiecheck`), which walks the
reference's boxes for every such tie and for every other accepted hit of a visibility ray, answers with the walk's result
and counts what the shortcuts would have answered differently:

    BDPT_LIB=bidirectional-path-tracer_amd/libbdpt_amd_tiecheck.so python tools/tie_check.py [--full]

Prints one JSON object: per scene the eight counters (capi.Context.TIE_COUNT_NAMES) and the SHA-256 of the rendered frame
(the frame of the normal build must be the same: tests/test_gpu_tie_check.py compares).  --full: twice the frame sizes
(minutes; what profiles/r03/tie_check_full.json holds), otherwise sizes for the test suite (seconds)."""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def cases(scenes, full):
    from conftest import composite_scene
    k = 2 if full else 1
    return [
        ("cornell", scenes.cornell_box(128 * k, 64, 10)),                        # staged in LDS: own-triangle test of k_connect
        ("coplanar", scenes.coplanar_stack(64 * k, 64, 6)),                      # stacked coincident quads: exact ties
        ("mixed3k", scenes.mixed_materials_3k(80 * k, 64, 8)),
        ("coffee", scenes.coffee_machine_standin(480 * k, 270 * k, 16, 10)),     # boxes resting on boxes: faces in one plane
        ("lattice1m", scenes.sphere_lattice(10, 192 * k, 16, 10)),               # curved geometry over a flat floor
        ("cx_instances", composite_scene(scenes, "cx_instances")[0]),            # rotated / translated boxes (two-level boxes)
        ("lattice_instanced", scenes.sphere_lattice_instanced(6, 96 * k, 16, 10)),
        ("cornell_instances", scenes.cornell_instances(96 * k, 64, 10)),         # turned and moved boxes whose tops lie in a box face of both trees
    ]


def main():
    full = "--full" in sys.argv
    capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
    scenes = importlib.import_module("bidirectional-path-tracer_amd.scenes")
    ctx = capi.Context(0)
    out = {"library": os.path.basename(capi.LIB_PATH), "scenes": {}}
    try:
        ctx.tie_counts(reset=True)
        checking = True
    except capi.BdptError:
        checking = False          # the normal build: frames only
    out["cross_check_build"] = checking
    for name, flat in cases(scenes, full):
        sc = ctx.scene(flat)
        cam = capi.camera_initialize(flat.camera)
        fb = sc.render(cam, seed=5)
        entry = {"frame_sha256": hashlib.sha256(np.ascontiguousarray(fb).tobytes()).hexdigest(), "triangles": flat.n_triangles}
        if checking:
            entry.update(ctx.tie_counts(reset=True))
        out["scenes"][name] = entry
        sc.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
