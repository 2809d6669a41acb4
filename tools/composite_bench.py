"""Times a composite scene (sphere lattice of translate instances over shared meshes, single-kernel path in the
reference's visiting order) beside the same lattice flattened to world-space triangles (wavefront pipeline).
    python tools/composite_bench.py [n_side] [image_width] [spp] [frames]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
scenes = importlib.import_module("bidirectional-path-tracer_amd.scenes")

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 10
width = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ctx = capi.Context(0)
out = {}
for name, builder in (("instanced", scenes.sphere_lattice_instanced), ("flattened", scenes.sphere_lattice)):
    t0 = time.perf_counter()
    flat = builder(n_side, width, spp, 10)
    sc = ctx.scene(flat)
    build_s = time.perf_counter() - t0
    cam = capi.camera_initialize(flat.camera)
    fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
    sc.render_device(cam, fb.data_ptr(), seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        sc.render_device(cam, fb.data_ptr(), seed=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    msamples = cam.image_width * cam.image_height * cam.sqrt_spp ** 2 * cam.max_depth / dt / 1e6
    out[name] = fb.cpu().numpy()
    print("%-10s n_side %d: %9d triangles stored, %5d objects, scene bytes %.1f MB, build %.2f s, frame %.1f ms, %.0f Msamples/s"
          % (name, n_side, flat.n_triangles, len(getattr(flat, "objects", [])), sc.info().device_bytes / 1e6, build_s, dt * 1e3, msamples), flush=True)
    sc.close()
a, b = out["instanced"] / spp, out["flattened"] / spp
print("mean radiance instanced %.6f flattened %.6f (different arithmetic: object space against world space; same estimator)"
      % (a.mean(), b.mean()))
