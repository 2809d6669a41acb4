"""Lane use of the two kinds of traversal iteration at the bench workloads' own sizes.  Needs the measuring build
(make VARIANT=phase DEFS=-DBDPT_AB_PHASE_STATS; BDPT_LIB=.../libbdpt_amd_phase.so), whose work counters report, per sample:
aabb_tests = lane slots of interior-record iterations, aabb_hits = lanes that tested a record in them, triangle_tests = lane
slots of triangle iterations, triangle_hits = lanes that tested a triangle, bvh_node_visits = all loop iterations x 64.
    python tools/phase_stats.py [workload ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
capi = importlib.import_module("bidirectional-path-tracer_amd.capi")
ctx = capi.Context(0)
ctx.set_workspace_limit(176 << 30)
for w in (sys.argv[1:] or ["glass-of-water-1080p", "cornell-512", "coffee-machine-1080p", "sphere-lattice-1m-1024"]):
    flat = bench.build_scene(w)
    sc = ctx.scene(flat)
    cam = capi.camera_initialize(flat.camera)
    fb, st = sc.render(cam, seed=0, want_stats="work")
    d = st.as_dict()
    n = float(d["samples"])
    print("%-26s record iterations: %5.1f %% of the lanes busy (%.1f slots / sample); triangle iterations: %5.1f %% (%.1f slots / sample); all iterations %.1f" % (
        w, 100.0 * d["aabb_hits"] / d["aabb_tests"], d["aabb_tests"] / n, 100.0 * d["triangle_hits"] / d["triangle_tests"], d["triangle_tests"] / n,
        d["bvh_node_visits"] / n), flush=True)
    sc.close()
