#!/usr/bin/env python3
"""Benchmark of the radiance-estimation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" renders one full frame of the workload: every rank renders the row bands it owns
(one process per GPU), and for N > 1 the framebuffers are summed with one RCCL all-reduce over
xGMI (disjoint pixel support -> the sum is exact).  Total work is fixed as N grows ("strong").

Metric (BASELINE.json): Msamples/s with samples = pixels x spp_used x max_depth (nominal
path-bounces), spp_used = floor(sqrt(spp))^2.  The scene, camera and the per-(pixel,sample)
random streams are already resident / derivable on the device when the timed region starts; the
output framebuffer stays in HBM.

Rank 0 prints ONE JSON line; at N = 1 it also carries
  "roofline"     : algorithmic bytes of the render kernel (counted by the kernel itself on a
                   1/8 row sample, scaled to a full launch) / its launch time measured with HIP
                   events on the launch's own stream, against the 8 TB/s HBM peak
  "cpu_baseline" : the reference's own headers (compiled in place into oracle/_ref, kind "reference") timed on the
                   host cores over a bounded pixel lattice of the same workload, with the CPU oracle (restatement of
                   the reference path, FP64, same streams; kind "port") beside it under "port".
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "bidirectional-path-tracer_amd"

HBM_PEAK_GBS = 8000.0

# Algorithmic bytes (FP64 records of the device layout):
#   closest-hit ray : origin + direction 48, queue index 4, triangle id 4, t 8
#   visibility ray  : origin + direction + max_t + tag 64, visibility word 8
#   node record     : counted by the kernels in units of 64 bytes; triangle 80; stored path vertex 80
#   per ray outside traversal (shading): ray + hit 60 in, next ray + throughput 72 out
#   per connection pair: both vertices read when its ray is generated and when it is accumulated
B_RAY_CLOSEST, B_RAY_SHADOW, B_NODE, B_TRI, B_VERTEX, B_PIXEL, B_RAY_STATE = 64, 72, 64, 80, 80, 24, 132

WORKLOADS = {
    # name: (builder, kwargs, description)
    "glass-of-water-1080p": ("glass_of_water_standin", dict(image_width=1920, image_height=1080, samples_per_pixel=1024, max_depth=12),
                             "glass-of-water.yaml camera/materials/light on the procedural stand-in geometry, 1920x1080, 1024 spp, max depth 12"),
    "cornell-512": ("cornell_box", dict(image_width=512, samples_per_pixel=256, max_depth=10),
                    "Cornell box (24 triangles, diffuse + emissive), 512x512, 256 spp, max depth 10"),
    "glass-of-water-1080p-16spp": ("glass_of_water_standin", dict(image_width=1920, image_height=1080, samples_per_pixel=16, max_depth=12),
                                   "glass-of-water stand-in, 1920x1080, 16 spp, max depth 12 (profiling size)"),
    "sphere-lattice-10m-4096": ("sphere_lattice", dict(n_side=22, image_width=4096, samples_per_pixel=64, max_depth=10),
                                "22^3 lambertian spheres of 960 triangles (10.2 M triangles) over a floor, one emissive quad, 4096x4096, 64 spp, max depth 10"),
    "sphere-lattice-1m-1024": ("sphere_lattice", dict(n_side=10, image_width=1024, samples_per_pixel=64, max_depth=10),
                               "10^3 lambertian spheres (0.96 M triangles), 1024x1024, 64 spp, max depth 10"),
    "glass-of-water-256": ("glass_of_water_standin", dict(image_width=256, image_height=256, samples_per_pixel=16, max_depth=4),
                           "glass-of-water stand-in, 256x256, 16 spp, max depth 4"),
}


def build_scene(name):
    scenes = importlib.import_module(PKG + ".scenes")
    builder, kwargs, _ = WORKLOADS[name]
    return getattr(scenes, builder)(**kwargs)


# ------------------------------------------------------------------------------------------ CPU baseline worker
def cpu_worker(args):
    """Child process: renders its share of the pixel lattice with the oracle and reports the time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    flat = build_scene(args.workload)
    o = oracle_lib.Oracle()
    sc = o.scene(flat)
    cam15, H, sqrt_spp = o.camera_initialize(flat.camera)
    W = flat.camera.image_width
    pixels = [(int(a), int(b)) for a, b in (p.split(",") for p in args.cpu_pixels.split(";") if p)]
    deadline = time.perf_counter() + args.cpu_seconds
    done = 0
    t0 = time.perf_counter()
    for (i, j) in pixels:
        sc.render(cam15, W, sqrt_spp, flat.camera.max_depth, args.seed, (i, j, i + 1, j + 1))
        done += 1
        if time.perf_counter() > deadline:
            break
    dt = time.perf_counter() - t0
    print(json.dumps({"pixels": done, "seconds": dt, "spp_used": sqrt_spp * sqrt_spp}))


REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness_glibc")


def run_cpu_baseline(workload, seed, seconds, flat):
    """The reference's own code (its headers compiled in place into oracle/_ref/ref_harness_glibc: camera::get_ray /
    ray_color on a bvh_node, per-(pixel, sample) random streams, glibc libm) timed on the host cores; the CPU
    restatement (oracle/) beside it.  One single-threaded process per core, each over its own share of a pixel
    lattice spread over the frame, so the reference's shared atomics and shared generator cost it nothing here."""
    import tempfile
    cores = len(os.sched_getaffinity(0))
    cores = max(1, min(cores, 16))
    W = flat.camera.image_width
    H = max(1, int(W / flat.camera.aspect_ratio))
    # lattice of pixels spread evenly over the frame; workers take them round-robin and stop at the deadline
    nx, ny = 192, 108
    if W < nx * 2 or H < ny * 2:
        nx, ny = max(1, W // 4), max(1, H // 4)
    lattice = [(int((a + 0.5) * W / nx), int((b + 0.5) * H / ny)) for b in range(ny) for a in range(nx)]
    # fixed scattering of the order, so that whatever prefix the workers finish before the deadline is itself spread
    # evenly over the frame (4099 is prime and does not divide the lattice size)
    lattice = [lattice[(k * 4099) % len(lattice)] for k in range(len(lattice))] if len(lattice) % 4099 else lattice
    depth = flat.camera.max_depth

    def run(make_cmd, secs):
        # interleave so that each worker's share is itself spread over the frame
        procs = []
        for w in range(cores):
            mine = ";".join("%d,%d" % p for p in lattice[w::cores])
            procs.append(subprocess.Popen(make_cmd(mine, secs), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        results = []
        for p in procs:
            out, err = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("cpu baseline worker failed: " + err[-2000:])
            results.append(json.loads(out.strip().splitlines()[-1]))
        spp_used = results[0]["spp_used"]
        # each worker ran alone on its core: aggregate rate = sum of the per-core rates
        rate = sum(r["pixels"] * spp_used * depth / r["seconds"] for r in results) / 1e6
        return rate, sum(r["pixels"] for r in results), spp_used

    def sample_text(total_px, spp_used, secs, what):
        return ("%d pixels of a %dx%d lattice over the %dx%d frame at the full %d spp / depth %d, one single-threaded "
                "%s process per core for <= %.0f s; g++ -O2 -ffp-contract=off" % (total_px, nx, ny, W, H, spp_used, depth, what, secs))

    port_secs = seconds if not os.path.exists(REF_HARNESS) else max(5.0, seconds / 2)
    rate, total_px, spp_used = run(lambda mine, secs: [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--workload", workload,
                                                       "--seed", str(seed), "--cpu-seconds", str(secs), "--cpu-pixels", mine], port_secs)
    port = {"value": rate, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": sample_text(total_px, spp_used, port_secs, "oracle (CPU restatement)")}
    if not os.path.exists(REF_HARNESS):
        return port
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(flat.to_bytes())
        scene_path = f.name
    try:
        rate, total_px, spp_used = run(lambda mine, secs: [REF_HARNESS, scene_path, "bench", str(seed), str(secs), mine], seconds)
    finally:
        os.unlink(scene_path)
    return {"value": rate, "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": sample_text(total_px, spp_used, seconds, "reference-harness (the reference's headers, scene build excluded)"),
            "port": port}


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="glass-of-water-1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--shard-rows", type=int, default=4, help="height of the row bands dealt round-robin to the GPUs")
    ap.add_argument("--megakernel", action="store_true", help="time the single persistent kernel instead of the wavefront pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pixels", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:
        cpu_worker(args)
        return

    if os.environ.get("BDPT_BENCH_WATCHDOG"):       # dump every thread's stack if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BDPT_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be started with torch.distributed.run (one process per GPU)" % args.gpus)
        args.gpus = world

    flat = build_scene(args.workload)

    # The CPU baseline runs BEFORE this process touches the GPU (its workers are plain child processes).
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = run_cpu_baseline(args.workload, args.seed, args.cpu_seconds, flat)

    import torch
    import torch.distributed as dist
    capi = importlib.import_module(PKG + ".capi")
    distributed = importlib.import_module(PKG + ".distributed")

    # one process per GPU; BDPT_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = os.environ.get("BDPT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    ctx = capi.Context(local_rank)
    # the frames here are large: let the per-sample state of the batches in flight take most of the 288 GB
    ctx.set_workspace_limit(int(os.environ.get("BDPT_BENCH_WORKSPACE_GB", "176")) << 30)
    scene = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    W, H, spp_used, depth = cam.image_width, cam.image_height, cam.sqrt_spp ** 2, cam.max_depth
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []

    def step(timed):
        def render_shard(buf, shard):
            scene.render_device(cam, buf.data_ptr(), seed=args.seed, shard=shard, stream=stream, time_kernel=timed,
                                megakernel=args.megakernel)
            if timed:
                kernel_ms.append(scene.last_kernel_ms())   # waits for the kernel's stop event
        distributed.render_frame(render_shard, fb, rank, world, args.shard_rows, dist)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = W * H * spp_used * depth
    value = samples_per_step * args.steps / elapsed / 1e6

    out = {
        "metric": "Msamples/s (pixels x spp_used x max_depth)",
        "value": value,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.workload][2],
            "triangles": flat.n_triangles,
            "resolution": [W, H],
            "spp_used": spp_used,
            "max_depth": depth,
            "sharding": "row bands of %d dealt round-robin to %d GPU(s), one RCCL all-reduce of the framebuffer" % (args.shard_rows, world)
                        if world > 1 else "single GPU",
            "paths_per_s": W * H * spp_used * args.steps / elapsed,
        },
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # untimed counting pass over every 8th row (same kernels with their work counters on)
        fb2 = torch.zeros_like(fb)
        scene.render_device(cam, fb2.data_ptr(), seed=args.seed, shard=(0, 8, 1), want_stats="work", stream=stream,
                            megakernel=args.megakernel)
        st = scene.last_stats()
        frac = st.samples / float(W * H * spp_used)
        alg_traverse = (B_RAY_CLOSEST * st.rays_traced + B_RAY_SHADOW * st.shadow_rays + B_NODE * st.bvh_node_visits +
                        B_TRI * st.triangle_tests) / frac
        alg = alg_traverse + (B_RAY_STATE * st.rays_traced + B_VERTEX * (st.path_vertices + 4 * st.connection_pairs) +
                              B_PIXEL * (st.samples / spp_used)) / frac
        ms = sum(kernel_ms) / len(kernel_ms)
        achieved = alg / (ms * 1e-3) / 1e9
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload + ("/megakernel" if args.megakernel else ""), {})
            except Exception:
                traffic = {}
        per_sample = {"trace_rays": st.rays_traced / st.samples, "shadow_rays": st.shadow_rays / st.samples,
                      "box_tests": st.aabb_tests / st.samples, "node_records_64B": st.bvh_node_visits / st.samples,
                      "triangle_tests": st.triangle_tests / st.samples,
                      "connection_pairs": st.connection_pairs / st.samples, "rng_draws": st.rng_draws / st.samples}
        # all kernels of one frame = one bdpt_render_device call, HIP events on the caller's stream
        frame = {"kernel": "render_kernel (single persistent kernel)" if args.megakernel else
                           "one bdpt_render_device call = all kernels of one frame (4 batches in flight on internal streams)",
                 "kernel_ms": ms, "algorithmic_bytes": alg, "achieved": achieved, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get("hbm_bytes_per_launch"),
                 "rays_per_s": (st.rays_traced + st.shadow_rays) / frac / (ms * 1e-3)}
        if args.megakernel:
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get("hbm_bytes_per_launch"),
                               "kernel": frame["kernel"], "kernel_ms": ms, "algorithmic_bytes_per_launch": alg,
                               "per_sample": per_sample, "rays_per_s": frame["rays_per_s"]}
        else:
            # the dominant kernel: one more untimed frame with a HIP event pair around every traversal launch, on the
            # launch's own stream
            scene.render_device(cam, fb2.data_ptr(), seed=args.seed, stream=stream, time_stages=True)
            ms0, n0 = scene.last_stage_ms(0)
            ms1, n1 = scene.last_stage_ms(1)
            n_launch = max(1, n0 + n1)
            avg_ms = (ms0 + ms1) / n_launch
            dom = alg_traverse / n_launch / (avg_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "achieved": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom / HBM_PEAK_GBS,
                "traffic": traffic.get("k_traverse_hbm_bytes_per_launch"),
                "kernel": "k_traverse (persistent BVH traversal; closest-hit and visibility instantiations together)",
                "launches_per_frame": n_launch, "avg_launch_ms": avg_ms,
                "closest_hit": {"launches": n0, "avg_launch_ms": ms0 / max(1, n0)},
                "visibility": {"launches": n1, "avg_launch_ms": ms1 / max(1, n1)},
                "algorithmic_bytes_per_launch": alg_traverse / n_launch,
                "summed_launch_ms_over_frame_ms": (ms0 + ms1) / ms,
                "note": "launches of up to 4 batches in flight share the chip, so a launch's own duration is longer than "
                        "its share of the frame; most of the algorithmic bytes (node and triangle records of a scene "
                        "of a few MB) are served by LDS and L2, see `traffic` for what reaches HBM",
                "per_sample": per_sample,
                "frame": frame,
            }
    if cpu is not None:
        out["cpu_baseline"] = cpu

    if rank == 0:
        print(json.dumps(out))
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
