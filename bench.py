#!/usr/bin/env python3
"""Benchmark of the radiance-estimation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" renders one full frame of the workload: every rank renders the row bands it owns
(one process per GPU), and for N > 1 the framebuffers are summed with one RCCL all-reduce over
xGMI (disjoint pixel support -> the sum is exact).  Total work is fixed as N grows ("strong").
Started without a launcher, `--gpus N` starts its N rank processes itself (children, one per GPU).

Metric (BASELINE.json): Msamples/s with samples = pixels x spp_used x max_depth (nominal
path-bounces), spp_used = floor(sqrt(spp))^2.  The scene, camera and the per-(pixel,sample)
random streams are already resident / derivable on the device when the timed region starts; the
output framebuffer stays in HBM.

The two scenes of the reference are rendered FROM ITS OWN YAML FILES (tests/data/scenes, byte for
byte) through the library's scene loader, with procedural stand-in meshes written under the OBJ
file names the YAML refers to (the meshes are not part of the reference's checkout).

Rank 0 prints ONE JSON line; at N = 1 it also carries
  "roofline"     : the dominant kernel (k_traverse).  `traffic` = HBM bytes per launch from rocprofv3 counter
                   passes that THIS run makes over one frame of the same workload (child processes, before this
                   process touches the GPU); `achieved` = traffic / average launch time measured with HIP events
                   on the launch's own stream; `frac` = achieved / 8 TB/s.  The algorithmic bytes (counted by the
                   kernels on every 8th row band) and what serves them - LDS/L1, L2, HBM - are beside it.
  "cpu_baseline" : the reference's own headers (compiled in place into oracle/_ref, kind "reference") timed on the
                   host cores over a bounded pixel lattice of the same workload, with the CPU oracle (restatement of
                   the reference path, FP64, same streams; kind "port") beside it under "port".
  "extra"        : connection-heavy workloads, 2-3 frames each: the Cornell box at 512 x 512 x 256 spp, the coffee machine at
                   1920 x 1080 x 484 spp and a 10.2 M-triangle lattice at 4096 x 4096 x 64 spp (flattened and as instances),
                   each with a parity check of 16 pixels against the CPU oracle at the full sample count.
  "value_default_workspace" : the headline frames with the library's default workspace limit instead of 176 GB.
For N > 1 the line carries "per_rank": each rank's render time and the duration of its all-reduce.
"""
import argparse
import csv
import glob
import importlib
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "bidirectional-path-tracer_amd"
SCENE_FILES = os.path.join(ROOT, "tests", "data", "scenes")

HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0        # aggregate over the 8 XCDs

# Algorithmic bytes (FP64 records of the device layout):
#   closest-hit ray : origin + direction 48, queue index 4, triangle id 4, t 8
#   visibility ray  : origin + direction + max_t + tag 64, visibility word 8
#   node record     : counted by the kernels in units of 64 bytes; triangle 80; stored path vertex 80
#   per ray outside traversal (shading): ray + hit 60 in, next ray + throughput 72 out
#   per connection pair: both vertices read when its ray is generated and when it is accumulated
B_RAY_CLOSEST, B_RAY_SHADOW, B_NODE, B_TRI, B_VERTEX, B_PIXEL, B_RAY_STATE = 64, 72, 64, 80, 80, 24, 132

WORKLOADS = {
    # name: (scene file of the reference or None, builder in scenes.py, keyword arguments, description)
    "glass-of-water-1080p": ("glass-of-water.yaml", "glass_of_water_standin", dict(image_width=1920, image_height=1080, samples_per_pixel=1024, max_depth=12),
                             "scenes/glass-of-water.yaml through the scene loader (stand-in OBJ meshes), 1920x1080, 1024 spp, max depth 12"),
    "coffee-machine-1080p": ("coffee_machine.yaml", "coffee_machine_standin", dict(image_width=1920, image_height=1080, samples_per_pixel=512, max_depth=10),
                             "scenes/coffee_machine.yaml through the scene loader (stand-in OBJ meshes), 1920x1080, 512 spp (484 used), max depth 10"),
    "cornell-512": (None, "cornell_box", dict(image_width=512, samples_per_pixel=256, max_depth=10),
                    "Cornell box (24 triangles, diffuse + emissive), 512x512, 256 spp, max depth 10"),
    "glass-of-water-1080p-16spp": ("glass-of-water.yaml", "glass_of_water_standin", dict(image_width=1920, image_height=1080, samples_per_pixel=16, max_depth=12),
                                   "scenes/glass-of-water.yaml, 1920x1080, 16 spp, max depth 12 (profiling size)"),
    "sphere-lattice-10m-4096": (None, "sphere_lattice", dict(n_side=22, image_width=4096, samples_per_pixel=64, max_depth=10),
                                "22^3 lambertian spheres of 960 triangles (10.2 M triangles) over a floor, one emissive quad, 4096x4096, 64 spp, max depth 10"),
    "sphere-lattice-instanced-10m-4096": (None, "sphere_lattice_instanced", dict(n_side=22, image_width=4096, samples_per_pixel=64, max_depth=10),
                                          "the same lattice as the reference's object model writes it: 10 648 translate instances sharing six 960-triangle meshes "
                                          "(10.2 M triangles once instanced), 4096x4096, 64 spp, max depth 10 (configuration 5 as worded)"),
    "sphere-lattice-instanced-1m-1024": (None, "sphere_lattice_instanced", dict(n_side=10, image_width=1024, samples_per_pixel=64, max_depth=10),
                                         "1 000 translate instances sharing six sphere meshes (0.96 M triangles once instanced), 1024x1024, 64 spp, max depth 10"),
    "sphere-lattice-1m-1024": (None, "sphere_lattice", dict(n_side=10, image_width=1024, samples_per_pixel=64, max_depth=10),
                               "10^3 lambertian spheres (0.96 M triangles), 1024x1024, 64 spp, max depth 10"),
    "glass-of-water-256": ("glass-of-water.yaml", "glass_of_water_standin", dict(image_width=256, image_height=256, samples_per_pixel=16, max_depth=4),
                           "scenes/glass-of-water.yaml, 256x256, 16 spp, max depth 4 (configuration 1)"),
}


def build_scene(name):
    """The workload's flat scene.  Scenes of the reference come out of its YAML file through bdpt_scene_file_load
    (host code of the library; no GPU work); resolution, samples and depth of the benchmark configuration replace
    the file's."""
    scenes = importlib.import_module(PKG + ".scenes")
    yaml_name, builder, kwargs, _ = WORKLOADS[name]
    if yaml_name is None:
        return getattr(scenes, builder)(**kwargs)
    capi = importlib.import_module(PKG + ".capi")
    tmp = tempfile.mkdtemp(prefix="bdpt_scene_")
    # the loader reports "Triangles: N" on stdout like the reference's (scene_loader.h:521); this program's stdout
    # carries one JSON line and nothing else, so the report goes to stderr
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        flat = capi.load_scene_file(scenes.write_standin_assets(os.path.join(SCENE_FILES, yaml_name), tmp))
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)        # the C library's own buffer, while descriptor 1 still points at stderr
        os.dup2(saved, 1)
        os.close(saved)
        shutil.rmtree(tmp, ignore_errors=True)
    c = flat.camera
    c.image_width = kwargs["image_width"]
    c.aspect_ratio = float(kwargs["image_width"]) / float(kwargs.get("image_height", kwargs["image_width"]))
    c.samples_per_pixel, c.max_depth = kwargs["samples_per_pixel"], kwargs["max_depth"]
    return flat


# ------------------------------------------------------------------------------------------ CPU baseline worker
def cpu_worker(args):
    """Child process: renders its share of the pixel lattice with the oracle and reports the time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    scenes = importlib.import_module(PKG + ".scenes")
    with open(args.cpu_scene, "rb") as f:
        flat = scenes.flat_scene_from_bytes(f.read())
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


def host_description():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "host_logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0))}


def run_cpu_baseline(seed, seconds, flat):
    """The reference's own code (its headers compiled in place into oracle/_ref/ref_harness_glibc: camera::get_ray /
    ray_color on a bvh_node, per-(pixel, sample) random streams, glibc libm) timed on the host cores; the CPU
    restatement (oracle/) beside it.  One single-threaded process per core, each over its own share of a pixel
    lattice spread over the frame, so the reference's shared atomics and shared generator cost it nothing here."""
    host = host_description()
    # a one-GPU lease comes with 16 of the machine's CPUs (whatever nproc says): never more workers than that
    cores = max(1, min(host["affinity_cpus"], 16))
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

    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(flat.to_bytes())
        scene_path = f.name

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

    try:
        port_secs = seconds if not os.path.exists(REF_HARNESS) else max(5.0, seconds / 2)
        rate, total_px, spp_used = run(lambda mine, secs: [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-scene", scene_path,
                                                           "--seed", str(seed), "--cpu-seconds", str(secs), "--cpu-pixels", mine], port_secs)
        port = {"value": rate, "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": sample_text(total_px, spp_used, port_secs, "oracle (CPU restatement)")}
        port.update(host)
        if not os.path.exists(REF_HARNESS):
            return port
        rate, total_px, spp_used = run(lambda mine, secs: [REF_HARNESS, scene_path, "bench", str(seed), str(secs), mine], seconds)
    finally:
        os.unlink(scene_path)
    out = {"value": rate, "unit": "Msamples/s", "cores": cores, "kind": "reference",
           "sample": sample_text(total_px, spp_used, seconds, "reference-harness (the reference's headers, scene build excluded)"),
           "port": port}
    out.update(host)
    return out


# ------------------------------------------------------------------------------------------ rocprofv3 counter passes
PMC_PASSES = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"]),
              # what the waves of a kernel do with their cycles (8 SQ counters = one pass; GRBM is a block of its own)
              ("issue", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                         "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"])]
N_SIMD = 256 * 4             # compute units x SIMDs
ENGINE_CLOCK_MHZ = 2400.0     # maximum engine clock of the MI355X; under load the chip runs below it
N_XCD = 8


def issue_summary(k):
    """Where the cycles of a kernel's waves go, from the SQ counters of the `issue` pass (all in the same unit, so the
    ratios need no conversion): parked on memory / a barrier, waiting to issue, executing; the share of the 64 lanes a
    VALU instruction has active (the gfx9 VALUUtilization formula); and - against the GPU-busy cycles of the same
    dispatches, which rocprofv3 reports summed over the 8 XCDs - average resident waves and VALU-busy share per SIMD
    (the gfx9 VALUBusy formula: SQ_ACTIVE_INST_VALU x 4 / SIMDs / busy cycles)."""
    wc = k.get("SQ_WAVE_CYCLES", 0.0)
    if not wc:
        return None
    gui = k.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    out = {"mem_wait_frac": k.get("SQ_WAIT_ANY", 0.0) / wc, "issue_wait_frac": k.get("SQ_WAIT_INST_ANY", 0.0) / wc,
           "executing_frac": k.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
           "lanes_active": (k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64.0)) if k.get("SQ_ACTIVE_INST_VALU") else None,
           "valu_instructions_per_wave": (k.get("SQ_INSTS_VALU", 0.0) / k["SQ_WAVES"]) if k.get("SQ_WAVES") else None}
    if gui:
        out["waves_per_simd"] = wc * 4.0 / N_SIMD / gui
        out["valu_busy_frac"] = k.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / N_SIMD / gui
    return out


def kernel_class(name):
    if "k_traverse" in name:
        return "k_traverse"
    for k in ("k_shade", "k_gen_eye", "k_gen_light", "k_connect", "k_conn_add", "k_conn_count", "k_conn_plan", "k_conn_ordered", "k_walk_ordered",
              "k_resolve", "k_redo", "render_kernel"):
        if k in name:
            return k
    return "other"


def run_pmc_passes(workload, seed, timeout_s):
    """HBM-side and L2-side traffic of one frame of `workload`, per kernel, from rocprofv3 counter passes (one pass
    per counter group, each its own child process running `bench.py --pmc-inner`).  Returns {"error": ...} when
    rocprofv3 is not there or a pass fails; the bench line then says so instead of quoting a stored number."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {"error": "rocprofv3 not found"}
    out = {"command": "rocprofv3 --pmc <group> --output-format csv -- python3 bench.py --pmc-inner --workload %s (one frame; one child process per counter group)" % workload,
           "kernels": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for tag, counters in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="bdpt_pmc_%s_" % tag, dir="/tmp")
        cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-inner",
               "--workload", workload, "--seed", str(seed)]
        t0 = time.perf_counter()
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            _, err = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)        # the group this call started, nothing else
            p.communicate()
            shutil.rmtree(d, ignore_errors=True)
            return {"error": "counter pass '%s' exceeded %d s" % (tag, timeout_s)}
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))
        if p.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return {"error": "counter pass '%s' failed (rc %s): %s" % (tag, p.returncode, (err or "")[-300:])}
        seen = set()
        with open(files[-1]) as f:
            for r in csv.DictReader(f):
                k = out["kernels"].setdefault(kernel_class(r["Kernel_Name"]), {"dispatches": 0})
                k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                key = (r.get("Dispatch_Id"), r["Kernel_Name"])
                if tag == "fetch" and key not in seen:
                    seen.add(key)
                    k["dispatches"] += 1
        out.setdefault("seconds", {})[tag] = time.perf_counter() - t0
        shutil.rmtree(d, ignore_errors=True)
    for k in out["kernels"].values():
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request of a wide coalesced
        # read at 64 bytes, so it is doubled (an upper estimate for the scattered 8-byte accesses of this path)
        k["hbm_read_bytes"] = 2.0 * 1024.0 * k.get("FETCH_SIZE", 0.0)
        k["hbm_write_bytes"] = 1024.0 * k.get("WRITE_SIZE", 0.0)
        k["hbm_bytes"] = k["hbm_read_bytes"] + k["hbm_write_bytes"]
        # one request of the vector L1 to the L2 moves one 128-byte line (reads) / 64 bytes (writes)
        k["l2_request_bytes"] = 128.0 * k.get("TCP_TCC_READ_REQ_sum", 0.0) + 64.0 * k.get("TCP_TCC_WRITE_REQ_sum", 0.0)
        k["issue"] = issue_summary(k)
    return out


def pmc_inner(args):
    """Child of run_pmc_passes, under rocprofv3: one untimed frame of the workload."""
    import torch
    capi = importlib.import_module(PKG + ".capi")
    flat = build_scene(args.workload)
    ctx = capi.Context(0)
    ctx.set_workspace_limit(int(os.environ.get("BDPT_BENCH_WORKSPACE_GB", "176")) << 30)
    scene = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
    scene.render_device(cam, fb.data_ptr(), seed=args.seed, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    scene.close()
    ctx.close()


# ------------------------------------------------------------------------------------------ rank processes
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N rank processes, one per GPU, started before anything in
    this process touches a GPU.  Rank 0 prints the line; the exit status is the worst of the ranks'."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        while procs:
            for p in list(procs):
                rc = p.poll()
                if rc is None:
                    continue
                procs.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in procs:              # a rank died: the others would wait for it for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    sys.exit(worst if 0 <= worst < 256 else 1)


# ------------------------------------------------------------------------------------------ timing
def time_frames(scene, cam, fb, steps, warmup, seed, stream, rank=0, world=1, shard_rows=4, dist=None, megakernel=False, reduce_ms=None,
                unidirectional=False):
    """(elapsed seconds of `steps` frames, device ms of each timed frame).  Barrier + synchronize on both sides.
    reduce_ms: a list that receives the duration of every timed frame's all-reduce on this rank (world > 1)."""
    import torch
    distributed = importlib.import_module(PKG + ".distributed")
    kernel_ms = []
    reduce_events = [] if (world > 1 and reduce_ms is not None) else None

    def step(timed):
        def render_shard(buf, shard):
            scene.render_device(cam, buf.data_ptr(), seed=seed, shard=shard, stream=stream, time_kernel=timed, megakernel=megakernel,
                                unidirectional=unidirectional)
            if timed:
                kernel_ms.append(scene.last_kernel_ms())   # waits for the kernel's stop event
        distributed.render_frame(render_shard, fb, rank, world, shard_rows, dist, reduce_events if timed else None)

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if reduce_events:
        reduce_ms.extend(a.elapsed_time(b) for a, b in reduce_events)
    return elapsed, kernel_ms


def oracle_parity(flat, cam, fb, seed, pixels):
    """Pixels of the GPU framebuffer against the CPU oracle at the full sample count, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    osc = oracle_lib.Oracle().scene(flat)
    host = fb.cpu().numpy()
    W = cam.image_width
    same = 0
    worst = 0.0
    for (i, j) in pixels:
        ofb, _, _ = osc.render(cam.as_array(), W, cam.sqrt_spp, cam.max_depth, seed, (i, j, i + 1, j + 1))
        if (ofb[0, 0].view(np.uint64) == host[j, i].view(np.uint64)).all():
            same += 1
        worst = max(worst, float(np.sqrt(((ofb[0, 0] - host[j, i]) ** 2).sum())) / (cam.sqrt_spp ** 2))
    return {"against": "CPU oracle (restatement of the reference path, FP64, same random streams)", "pixels_checked": len(pixels),
            "pixels_bit_identical": same, "max_per_pixel_l2_of_mean_radiance": worst}


def extra_leg(ctx, name, seed, steps, stream, parity_pixels, unidirectional_too=False):
    import torch
    capi = importlib.import_module(PKG + ".capi")
    t0 = time.perf_counter()
    flat = build_scene(name)
    scene = ctx.scene(flat)
    build_s = time.perf_counter() - t0
    cam = capi.camera_initialize(flat.camera)
    W, H, spp_used, depth = cam.image_width, cam.image_height, cam.sqrt_spp ** 2, cam.max_depth
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    elapsed, kernel_ms = time_frames(scene, cam, fb, steps, 1, seed, stream)
    out = {"workload": WORKLOADS[name][3], "triangles": flat.n_triangles, "objects": len(getattr(flat, "objects", [])), "value": W * H * spp_used * depth * steps / elapsed / 1e6,
           "unit": "Msamples/s", "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3, "scene_build_s": build_s,
           "paths_per_s": W * H * spp_used * steps / elapsed}
    if parity_pixels:
        out["parity"] = oracle_parity(flat, cam, fb, seed, [(int((a + 0.5) * W / parity_pixels), int((a * 7 % parity_pixels + 0.5) * H / parity_pixels))
                                                            for a in range(parity_pixels)])
    if unidirectional_too:
        # the reference's other estimator (camera::path_trace_color, src/camera.h:255-292) on the same pipeline
        elapsed, _ = time_frames(scene, cam, fb, steps, 1, seed, stream, unidirectional=True)
        out["unidirectional"] = {"value": W * H * spp_used * depth * steps / elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": elapsed / steps * 1e3,
                                 "estimator": "camera::path_trace_color (BDPT_RENDER_UNIDIRECTIONAL)"}
    scene.close()
    return out


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
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic is then null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads (Cornell box, 10 M-triangle lattice)")
    ap.add_argument("--pmc-timeout", type=int, default=240)
    ap.add_argument("--pmc-json", default="", help="also write the per-kernel counter sums of the counter passes to this file")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pixels", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-scene", default="", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:
        cpu_worker(args)
        return
    if args.pmc_inner:
        pmc_inner(args)
        return

    if os.environ.get("BDPT_BENCH_WATCHDOG"):       # dump every thread's stack if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BDPT_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                           # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    flat = build_scene(args.workload)

    # Everything that runs in child processes comes BEFORE this process touches the GPU: the CPU baseline and the
    # rocprofv3 counter passes.
    cpu = None
    pmc = None
    if world == 1 and rank == 0:
        if not args.no_cpu_baseline:
            cpu = run_cpu_baseline(args.seed, args.cpu_seconds, flat)
        if not args.no_roofline and not args.no_pmc and not args.megakernel:
            try:
                pmc = run_pmc_passes(args.workload, args.seed, args.pmc_timeout)
            except Exception as e:      # never lose the bench line to the profiler
                pmc = {"error": "counter passes raised %r" % (e,)}
            if args.pmc_json:
                with open(args.pmc_json, "w") as f:
                    json.dump({"workload": args.workload, "passes": {t: c for t, c in PMC_PASSES}, **pmc}, f, indent=1)

    import torch
    import torch.distributed as dist
    capi = importlib.import_module(PKG + ".capi")

    # one process per GPU; BDPT_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = os.environ.get("BDPT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    ranks_seen, devices = 1, [local_rank]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([local_rank], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        devices = [int(t.item()) for t in gathered]

    ctx = capi.Context(local_rank)
    # the frames here are large: let the per-sample state of the batches in flight take most of the 288 GB
    ctx.set_workspace_limit(int(os.environ.get("BDPT_BENCH_WORKSPACE_GB", "176")) << 30)
    scene = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    W, H, spp_used, depth = cam.image_width, cam.image_height, cam.sqrt_spp ** 2, cam.max_depth
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    reduce_ms = []
    elapsed, kernel_ms = time_frames(scene, cam, fb, args.steps, args.warmup, args.seed, stream, rank, world, args.shard_rows, dist,
                                     args.megakernel, reduce_ms)
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what each rank spent: its own share of the frame (device time of its render call) and the all-reduce, which
        # includes the wait for the slowest rank
        mine = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms)), sum(reduce_ms) / max(1, len(reduce_ms))], dtype=torch.float64,
                            device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        render = [float(g[0].item()) for g in gathered]
        red = [float(g[1].item()) for g in gathered]
        per_rank = {"render_ms": {"min": min(render), "max": max(render), "mean": sum(render) / world, "by_rank": render},
                    "reduce_ms": {"min": min(red), "max": max(red), "by_rank": red,
                                  "note": "all-reduce of the %d-byte framebuffer as timed on each rank; the shortest one is the collective itself, "
                                          "the others include the wait for the slowest rank's render" % (W * H * 3 * 8)}}

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
            "workload": WORKLOADS[args.workload][3],
            "scene_source": ("tests/data/scenes/%s (the reference's file, unchanged) via bdpt_scene_file_load; OBJ meshes: procedural stand-ins"
                             % WORKLOADS[args.workload][0]) if WORKLOADS[args.workload][0] else "procedural (scenes.py)",
            "triangles": flat.n_triangles,
            "resolution": [W, H],
            "spp_used": spp_used,
            "max_depth": depth,
            "sharding": "row bands of %d dealt round-robin to %d GPU(s), one RCCL all-reduce of the framebuffer" % (args.shard_rows, world)
                        if world > 1 else "single GPU",
            "ranks_seen": ranks_seen,
            "rank_devices": devices,
            "collective_backend": backend if world > 1 else None,
            "paths_per_s": W * H * spp_used * args.steps / elapsed,
        },
    }
    if per_rank is not None:
        out["per_rank"] = per_rank

    if rank == 0 and world == 1 and not args.no_roofline:
        # untimed counting pass over every 8th row band (same kernels with their work counters on)
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
        per_sample = {"trace_rays": st.rays_traced / st.samples, "shadow_rays": st.shadow_rays / st.samples,
                      "box_tests": st.aabb_tests / st.samples, "node_records_64B": st.bvh_node_visits / st.samples,
                      "triangle_tests": st.triangle_tests / st.samples,
                      "connection_pairs": st.connection_pairs / st.samples, "rng_draws": st.rng_draws / st.samples}
        rays_per_s = (st.rays_traced + st.shadow_rays) / frac / (ms * 1e-3)
        kernels = (pmc or {}).get("kernels") or {}
        frame_hbm = sum(k["hbm_bytes"] for k in kernels.values()) if kernels else None
        frame_l2 = sum(k["l2_request_bytes"] for k in kernels.values()) if kernels else None
        # all kernels of one frame = one bdpt_render_device call, HIP events on the caller's stream
        frame = {"kernel": "render_kernel (single persistent kernel)" if args.megakernel else
                           "one bdpt_render_device call = all kernels of one frame (4 batches in flight on internal streams)",
                 "kernel_ms": ms, "algorithmic_bytes": alg, "algorithmic_gbs": alg / (ms * 1e-3) / 1e9,
                 "traffic": frame_hbm, "achieved": (frame_hbm / (ms * 1e-3) / 1e9) if frame_hbm else None, "unit": "GB/s",
                 "frac": (frame_hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if frame_hbm else None,
                 "l2_request_bytes": frame_l2, "l2_frac_of_peak": (frame_l2 / (ms * 1e-3) / 1e9 / L2_PEAK_GBS) if frame_l2 else None,
                 "rays_per_s": rays_per_s}
        if kernels and all("SQ_ACTIVE_INST_VALU" in k for k in kernels.values()):
            # The binding resource of the frame: cycles in which a SIMD issues a VALU instruction, summed over all kernels
            # of the frame (counter pass, kernels one after the other), against the cycles the overlapped frame takes at the
            # nominal engine clock (a lower real clock makes the share larger).
            clock_hz = ENGINE_CLOCK_MHZ * 1e6
            valu_cycles = sum(k["SQ_ACTIVE_INST_VALU"] for k in kernels.values()) * 4.0 / N_SIMD
            thread_cycles = sum(k.get("SQ_THREAD_CYCLES_VALU", 0.0) for k in kernels.values())
            frame["valu"] = {"valu_issue_cycles_per_simd": valu_cycles, "engine_clock_mhz": clock_hz / 1e6,
                             "frac_of_frame": valu_cycles / (ms * 1e-3 * clock_hz),
                             "lanes_active": thread_cycles / (sum(k["SQ_ACTIVE_INST_VALU"] for k in kernels.values()) * 64.0),
                             "by_kernel": {n: k["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / (ms * 1e-3 * clock_hz) for n, k in kernels.items()},
                             "what": "share of the frame's cycles in which a SIMD issues a vector ALU instruction (FP64 triangle tests, packed-FP32 "
                                     "box tests, shading), all kernels of the frame together: the resource this pipeline is closest to"}
        if args.megakernel:
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "kernel": frame["kernel"], "kernel_ms": ms, "algorithmic_bytes_per_launch": alg,
                               "algorithmic_gbs": alg / (ms * 1e-3) / 1e9, "per_sample": per_sample, "rays_per_s": rays_per_s}
        else:
            # the dominant kernel: one more untimed frame with a HIP event pair around every traversal launch, on the
            # launch's own stream
            scene.render_device(cam, fb2.data_ptr(), seed=args.seed, stream=stream, time_stages=True)
            ms0, n0 = scene.last_stage_ms(0)
            ms1, n1 = scene.last_stage_ms(1)
            n_launch = max(1, n0 + n1)
            avg_ms = (ms0 + ms1) / n_launch
            kt = kernels.get("k_traverse")
            traffic = (kt["hbm_bytes"] / max(1, kt["dispatches"])) if kt else None
            l2_bytes = (kt["l2_request_bytes"] / max(1, kt["dispatches"])) if kt else None
            alg_per_launch = alg_traverse / n_launch
            achieved = (traffic / (avg_ms * 1e-3) / 1e9) if traffic is not None else None
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None, "traffic": traffic,
                "kernel": "k_traverse (persistent BVH traversal; closest-hit and visibility instantiations together)",
                "what": "achieved = HBM bytes per launch measured by this run's rocprofv3 counter passes / average launch time (HIP events)",
                "launches_per_frame": n_launch, "avg_launch_ms": avg_ms,
                "counter_run_launches": kt["dispatches"] if kt else None,
                "closest_hit": {"launches": n0, "avg_launch_ms": ms0 / max(1, n0)},
                "visibility": {"launches": n1, "avg_launch_ms": ms1 / max(1, n1)},
                "algorithmic": {"bytes_per_launch": alg_per_launch, "gbs": alg_per_launch / (avg_ms * 1e-3) / 1e9,
                                "note": "records the traversal asks for (rays, node records, triangles); served as listed under `levels`"},
                "levels": None if traffic is None else {
                    "lds_and_l1_bytes_per_launch": max(0.0, alg_per_launch - l2_bytes),
                    "l2_request_bytes_per_launch": l2_bytes, "l2_gbs": l2_bytes / (avg_ms * 1e-3) / 1e9,
                    "l2_frac_of_peak": l2_bytes / (avg_ms * 1e-3) / 1e9 / L2_PEAK_GBS,
                    "hbm_bytes_per_launch": traffic},
                "limiter": "latency of dependent fetches (record -> four slab tests -> next record) with FP64 triangle tests in between: "
                           "neither HBM nor L2 bandwidth is near its peak; `issue` says what the waves do with their cycles",
                "summed_launch_ms_over_frame_ms": (ms0 + ms1) / ms,
                "note": "launches of up to 4 batches in flight share the chip, so a launch's own duration is longer than its share of "
                        "the frame; the counter passes serialise the launches",
                "issue": (kt or {}).get("issue"),
                "pmc": {k: pmc[k] for k in ("command", "seconds", "error", "kernels") if k in pmc} if pmc else None,
                "per_sample": per_sample,
                "frame": frame,
            }
    if cpu is not None:
        out["cpu_baseline"] = cpu

    if rank == 0 and world == 1 and not args.no_extras and not args.megakernel and "BDPT_BENCH_WORKSPACE_GB" not in os.environ:
        # the same frames with the library's own default workspace limit (32 GiB) instead of the 176 GB this program asks for
        ctx.set_workspace_limit(0)
        dflt_elapsed, _ = time_frames(scene, cam, fb, args.steps, 1, args.seed, stream)
        out["value_default_workspace"] = samples_per_step * args.steps / dflt_elapsed / 1e6
        out["config"]["workspace"] = {"value": "bdpt_context_set_workspace_limit(176 GB): batches of up to 2^27 samples",
                                      "value_default_workspace": "library default (32 GiB, at most a quarter of the free memory)"}
        ctx.set_workspace_limit(int(os.environ.get("BDPT_BENCH_WORKSPACE_GB", "176")) << 30)

    if rank == 0 and world == 1 and not args.no_extras and not args.megakernel and args.workload == "glass-of-water-1080p":
        scene.close()
        scene = None
        extras = {}
        for name, pixels in (("cornell-512", 16), ("coffee-machine-1080p", 16), ("sphere-lattice-10m-4096", 16), ("sphere-lattice-instanced-10m-4096", 16)):
            try:
                extras[name] = extra_leg(ctx, name, args.seed, 3 if name == "cornell-512" else 2, stream, pixels, unidirectional_too=(name == "cornell-512"))
            except Exception as e:      # the headline line must not depend on an extra
                extras[name] = {"error": repr(e)}
        out["extra"] = extras
        ok = [e["parity"] for e in extras.values() if "parity" in e]
        out["parity"] = {"pixels_checked": sum(p["pixels_checked"] for p in ok), "pixels_bit_identical": sum(p["pixels_bit_identical"] for p in ok),
                         "max_per_pixel_l2_of_mean_radiance": max([p["max_per_pixel_l2_of_mean_radiance"] for p in ok] or [None]),
                         "note": "GPU pixels of the extra workloads against the CPU oracle at the full sample count; the headline scene is "
                                 "checked the same way by tests/test_gpu_full_size.py"}

    if rank == 0:
        print(json.dumps(out))
    if scene is not None:
        scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
