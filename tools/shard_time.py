#!/usr/bin/env python3
"""Time one rank's share of the headline frame for world sizes 1, 2, 4, 8 on ONE GPU (no collective):
what strong scaling costs per rank besides the all-reduce."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("bidirectional-path-tracer_amd")
capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
flat = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "glass-of-water-1080p")
ctx = capi.Context(0)
scene = ctx.scene(flat)
cam = capi.camera_initialize(flat.camera)
fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    ts = []
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.render_device(cam, fb.data_ptr(), seed=0, shard=(0, n, 4), stream=stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    if n == 1:
        t1 = t
    print("world %d: %.1f ms per rank, efficiency %.2f" % (n, t * 1e3, t1 / (n * t)), flush=True)
