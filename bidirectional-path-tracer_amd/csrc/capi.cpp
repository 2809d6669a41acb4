// C ABI (include/bdpt_amd.h) over the HIP kernels: contexts, scene upload, render / trace entry points.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bdpt_amd.h"
#include "bdpt_types.h"
#include "composite.h"
#include "kernels.h"
#include "scene_build.h"
#include "static_scratch.h"
#include "wavefront.h"

using namespace bdpt;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(e == hipErrorOutOfMemory ? BDPT_ERR_OUT_OF_MEMORY : BDPT_ERR_DEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                             \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return hip_fail(_e, #expr);         \
    } while (0)

constexpr int kBlock = 256;
constexpr int kStatWords = 11;
constexpr int kWfStreams = 8;                               // most batches in flight (one HIP stream each)

// Tuning knobs, overridable from the environment for experiments.  Read once, when a context is created, and
// clamped to what the kernels support.
int env_int(const char* name, int fallback, int lo, int hi) {
    const char* v = std::getenv(name);
    int x = (v && *v) ? std::atoi(v) : fallback;
    return x < lo ? lo : (x > hi ? hi : x);
}

struct Knobs {
    int budget_gb;        // 0: the context's workspace limit decides
    int streams;
    int batch_log2;
    int lds_kb;
    int lds_stack;
    int blocks_per_cu;    // 0: from the occupancy query, at most 3
    int refill;
    int leaf;
    int leaf_inner;       // 0: half the leaf threshold
    int chunk;
    int no_first_leaf;    // 1: shadow rays start at the root like all others
    int no_lds_scene;     // 1: small scenes are not staged in LDS (measuring)
    int no_own_test;      // 1: the connection kernel leaves the far end's own triangle to the traversal (measuring)
    int first_leaf_mb;    // largest triangle table (MB) whose shadow rays test their far end's triangle first
    int redo_blocks;      // workgroups of k_redo; 0: one per compute unit
    int conn_cap_tenths;  // shadow-queue entries per sample slot, in tenths of max_depth (10: depth entries per slot)
    int sort_cells;       // k_shade groups the paths of a tile by where they leave from: number of cells (0: off, at most 64; -1: by scene)
    int pixel_major;      // 1: neighbouring slots hold the strata of one pixel; 0: one stratum of neighbouring pixels
    int conn_budget;      // vertex records a wave of k_connect stages at a time; 0: 128, at least 2 x depth
};

Knobs read_knobs() {
    Knobs k;
    k.budget_gb = env_int("BDPT_WF_BUDGET_GB", 0, 0, 224);
    k.streams = env_int("BDPT_WF_STREAMS", 4, 1, kWfStreams);
    k.batch_log2 = env_int("BDPT_WF_BATCH_LOG2", 27, 6, 30);
    k.lds_kb = env_int("BDPT_WF_LDS_KB", 16, 1, 128);
    k.lds_stack = env_int("BDPT_WF_LDS_STACK", 8, 0, kQuadStackEntries);
    k.blocks_per_cu = env_int("BDPT_WF_BLOCKS_PER_CU", 0, 0, 16);
    k.refill = env_int("BDPT_WF_REFILL", 16, 1, 64);         // above 64 no wave would ever fetch rays
    k.leaf = env_int("BDPT_WF_LEAF", 32, 1, 64);
    k.leaf_inner = env_int("BDPT_WF_LEAF_INNER", 0, 0, 64);
    k.chunk = env_int("BDPT_WF_CHUNK", 512, 1, 1 << 20);
    k.no_first_leaf = env_int("BDPT_WF_NO_FIRST_LEAF", 0, 0, 1);
    k.no_lds_scene = env_int("BDPT_WF_NO_LDS_SCENE", 0, 0, 1);
    k.no_own_test = env_int("BDPT_WF_NO_OWN_TEST", 0, 0, 1);
    k.first_leaf_mb = env_int("BDPT_WF_FIRST_LEAF_MB", 1 << 20, 0, 1 << 20);
    k.redo_blocks = env_int("BDPT_WF_REDO_BLOCKS", 0, 0, 1 << 16);
    k.conn_cap_tenths = env_int("BDPT_WF_CONN_CAP", 10, 5, 80);
    k.conn_budget = env_int("BDPT_WF_CONN_BUDGET", 0, 0, 2048);
    k.pixel_major = env_int("BDPT_WF_PIXEL_MAJOR", 1, 0, 1);
    k.sort_cells = env_int("BDPT_WF_SORT_CELLS", -1, -1, 64);
    return k;
}

// Default limit of the library-managed render workspace; bdpt_context_set_workspace_limit raises or lowers it.
constexpr size_t kDefaultWorkspaceLimit = (size_t)32 << 30;

}  // namespace

struct bdpt_context {
    int device = 0;
    int cu_count = 0;
    Knobs knobs{};
    size_t workspace_limit = 0;          // bytes; 0 = default (kDefaultWorkspaceLimit, at most a quarter of the free memory)
    int query_walker = BDPT_WALKER_PIPELINE;
    // this GPU's copy of the library's static workspace (bdpt_render_device_static), looked up when the context is made
    void* static_ws = nullptr;
    hipError_t static_ws_status = hipErrorNotInitialized;
    // workspace of the ray / visibility query entry points
    void* query_ws = nullptr;
    size_t query_ws_bytes = 0;
    // reusable device buffers
    double* vertex_ws = nullptr;
    size_t vertex_ws_bytes = 0;
    int32_t* rows = nullptr;
    size_t rows_capacity = 0;
    uint32_t* unit_counter = nullptr;
    unsigned long long* stats = nullptr;
    bool stats_valid = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // bracket the render kernels on their own stream
    bool timing_valid = false;
    // wavefront workspace (one allocation, carved into the WfBuffers arrays)
    void* wf_blob = nullptr;
    size_t wf_blob_bytes = 0;
    hipStream_t wf_streams[kWfStreams] = {};
    hipEvent_t wf_fork = nullptr;
    hipEvent_t wf_join[kWfStreams] = {};
    // BDPT_RENDER_TIME_STAGES: one event pair per traversal launch of the last render
    std::vector<hipEvent_t> stage_events;
    std::vector<int> stage_kind;          // per pair: 0 closest-hit traversal, 1 visibility traversal
    size_t stage_pairs = 0;
    bool stage_valid = false;
};

struct bdpt_scene {
    bdpt_context* ctx = nullptr;
    HostScene host;
    SceneDev dev{};
    // composite scenes (bdpt_scene_create_composite) keep their own tables and run their own kernels
    bool composite = false;
    CxHostScene cx_host;
    CxSceneDev cx_dev{};
    void* d_cx_blob = nullptr;
    size_t cx_blob_bytes = 0;
    bool pipeline = true;          // false: a composite scene with media (the four-wide hierarchy cannot answer for it)
    bool ordered = false;          // such a scene: the pipeline's stages around walks in the reference's own order
    void* d_blob = nullptr;
    size_t blob_bytes = 0;
    uint32_t n_materials = 0;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int ensure_rows(bdpt_context* ctx, size_t n) {
    if (ctx->rows_capacity >= n) return BDPT_OK;
    if (ctx->rows) (void)hipFree(ctx->rows);
    ctx->rows = nullptr;
    ctx->rows_capacity = 0;
    HIP_TRY(hipMalloc(&ctx->rows, sizeof(int32_t) * n));
    ctx->rows_capacity = n;
    return BDPT_OK;
}

int ensure_vertex_ws(bdpt_context* ctx, size_t bytes) {
    if (ctx->vertex_ws_bytes >= bytes) return BDPT_OK;
    if (ctx->vertex_ws) (void)hipFree(ctx->vertex_ws);
    ctx->vertex_ws = nullptr;
    ctx->vertex_ws_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->vertex_ws, bytes));
    ctx->vertex_ws_bytes = bytes;
    return BDPT_OK;
}

int ensure_query_ws(bdpt_context* ctx, size_t bytes) {
    if (ctx->query_ws_bytes >= bytes) return BDPT_OK;
    if (ctx->query_ws) (void)hipFree(ctx->query_ws);
    ctx->query_ws = nullptr;
    ctx->query_ws_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->query_ws, bytes));
    ctx->query_ws_bytes = bytes;
    return BDPT_OK;
}

void read_stats(const unsigned long long* h, bdpt_stats* out) {
    out->rays_traced = h[0]; out->aabb_tests = h[1]; out->aabb_hits = h[2]; out->bvh_node_visits = h[3];
    out->triangle_tests = h[4]; out->triangle_hits = h[5]; out->shadow_rays = h[6]; out->connection_pairs = h[7];
    out->rng_draws = h[8]; out->samples = h[9]; out->path_vertices = h[10];
}

GpuCamera to_gpu_camera(const bdpt_camera_state& c) {
    GpuCamera g;
    for (int k = 0; k < 3; k++) {
        g.center[k] = c.center[k]; g.pixel00[k] = c.pixel00_loc[k]; g.du[k] = c.pixel_delta_u[k];
        g.dv[k] = c.pixel_delta_v[k]; g.background[k] = c.background[k];
    }
    for (int k = 0; k < 3; k++) { g.disk_u[k] = c.defocus_disk_u[k]; g.disk_v[k] = c.defocus_disk_v[k]; }
    g.defocus_angle = c.defocus_angle;
    g.recip_sqrt_spp = 1.0 / c.sqrt_spp;        // camera.h:166
    g.width = c.image_width; g.height = c.image_height; g.sqrt_spp = c.sqrt_spp; g.max_depth = c.max_depth;
    return g;
}

// ---- single persistent kernel: one thread per sample does the whole of bidirectional_color
int render_megakernel(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam, uint64_t seed, const int32_t* rows,
                      int32_t n_rows, int x0, int x1, int64_t n_owned, double* d_fb, int stats_mode, bool unidirectional,
                      hipStream_t stream) {
    HIP_TRY(hipMemsetAsync(ctx->unit_counter, 0, sizeof(uint32_t), stream));
    const int spp = cam->sqrt_spp * cam->sqrt_spp;
    const int ppg = spp >= kBlock ? 1 : kBlock / spp;
    const int64_t n_units = (n_owned + ppg - 1) / ppg;
    if (n_units > 0x7FFFFFFF) return fail(BDPT_ERR_INVALID_ARGUMENT, "too many work units in one call");
    // persistent grid: enough workgroups to fill the chip, never more than there is work
    int per_cu = 1;
    HIP_TRY(render_blocks_per_cu(scene->dev, stats_mode, unidirectional, &per_cu));
    int64_t grid = (int64_t)ctx->cu_count * per_cu;
    if (grid > n_units) grid = n_units;
    const size_t ws_stride = (size_t)grid * kBlock;
    const size_t slots = 2 * (size_t)(cam->max_depth > 0 ? cam->max_depth : 1);
    int rc = ensure_vertex_ws(ctx, align256(ws_stride * slots * 10 * sizeof(double)));
    if (rc != BDPT_OK) return rc;

    RenderLaunch rl;
    rl.seed = seed;
    rl.rows = rows;
    rl.n_rows = n_rows;
    rl.x0 = x0; rl.x1 = x1;
    rl.pixels_per_group = ppg;
    rl.n_units = (int32_t)n_units;
    rl.unit_counter = ctx->unit_counter;
    rl.fb = d_fb;
    rl.vertex_ws = ctx->vertex_ws;
    rl.stats = stats_mode ? ctx->stats : nullptr;
    rl.ws_stride = ws_stride;
    HIP_TRY(launch_render(scene->dev, to_gpu_camera(*cam), rl, (int)grid, stats_mode, unidirectional, stream));
    return BDPT_OK;
}

// ---- composite scenes: the same single-kernel scheme over the scene's own tables (composite.hip)
int render_composite(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam, uint64_t seed, const int32_t* rows,
                     int32_t n_rows, int x0, int x1, int64_t n_owned, double* d_fb, bool stats, bool unidirectional, hipStream_t stream) {
    HIP_TRY(hipMemsetAsync(ctx->unit_counter, 0, sizeof(uint32_t), stream));
    const int spp = cam->sqrt_spp * cam->sqrt_spp;
    const int ppg = spp >= kBlock ? 1 : kBlock / spp;
    const int64_t n_units = (n_owned + ppg - 1) / ppg;
    if (n_units > 0x7FFFFFFF) return fail(BDPT_ERR_INVALID_ARGUMENT, "too many work units in one call");
    int per_cu = 1;
    HIP_TRY(cx_render_blocks_per_cu(stats, unidirectional, &per_cu));
    int64_t grid = (int64_t)ctx->cu_count * per_cu;
    if (grid > n_units) grid = n_units;
    const size_t ws_stride = (size_t)grid * kBlock;
    const size_t slots = 2 * (size_t)(cam->max_depth > 0 ? cam->max_depth : 1);
    int rc = ensure_vertex_ws(ctx, align256(ws_stride * slots * kCxVtxDoubles * sizeof(double)));
    if (rc != BDPT_OK) return rc;
    CxRenderLaunch rl;
    rl.seed = seed;
    rl.rows = rows;
    rl.n_rows = n_rows;
    rl.x0 = x0; rl.x1 = x1;
    rl.pixels_per_group = ppg;
    rl.n_units = (int32_t)n_units;
    rl.unit_counter = ctx->unit_counter;
    rl.fb = d_fb;
    rl.vertex_ws = ctx->vertex_ws;
    rl.stats = stats ? ctx->stats : nullptr;
    rl.ws_stride = ws_stride;
    HIP_TRY(cx_launch_render(scene->cx_dev, to_gpu_camera(*cam), rl, (int)grid, stats, unidirectional, stream));
    return BDPT_OK;
}

// launch shape of the persistent traversal kernels and of the stage kernels for one scene
// dynamic LDS of k_connect for a scene at a depth: `waves` staging areas of `budget` vertex records, then the tables of its
// own-triangle test
size_t conn_lds_need(uint32_t waves, uint32_t budget, bool textured, size_t own_bytes) {
    return (size_t)waves * budget * (textured ? 13u : 10u) * sizeof(double) + own_bytes;      // (words of a vertex record, device_core.h)
}
constexpr size_t kConnLdsMost = (size_t)150 << 10;        // of a compute unit's 160 KiB
// true when the connection kernel can stage the 2 x depth vertex records of one sample
bool pipeline_can_connect(const bdpt_scene* scene, int depth) {
    return conn_lds_need(1, 2u * (uint32_t)(depth > 0 ? depth : 1), scene->dev.tex != nullptr, 0) <= kConnLdsMost;
}

int make_launch_cfg(const bdpt_context* ctx, const bdpt_scene* scene, WfLaunchCfg& cfg, int depth = 1) {
    const Knobs& kn = ctx->knobs;
    // (the hierarchy's triangle table: one entry per triangle, or per (object, triangle) pair of a scene with instances)
    // (... and behind it, for such a scene, the objects and the object of every entry)
    cfg.lds_bytes = (size_t)scene->dev.n_xnodes * sizeof(GpuQuadNode) + scene->host.xtris.size() * sizeof(GpuTri);
    if (scene->dev.objs) cfg.lds_bytes = (cfg.lds_bytes + scene->dev.n_objs * sizeof(CxObject) + scene->host.xtris.size() * sizeof(int32_t) + 15) & ~(size_t)15;
    cfg.lds_scene = cfg.lds_bytes <= kSceneLdsLimit && !kn.no_lds_scene;
    if (!cfg.lds_scene) cfg.lds_bytes = 0;
    cfg.instanced = scene->dev.objs != nullptr;
    cfg.textured = scene->dev.tex != nullptr;
    cfg.tex_pass = cfg.textured && scene->cx_host.textured;
    {
        const size_t walk_bytes = (size_t)scene->dev.n_nodes * sizeof(GpuNode) + (size_t)scene->dev.n_tris * sizeof(GpuTri) + (size_t)scene->dev.n_objs * sizeof(CxObject);
        cfg.walk_lds_bytes = (scene->ordered && walk_bytes <= ((size_t)24 << 10) && !kn.no_lds_scene) ? walk_bytes : 0;
    }
    cfg.ordered = scene->ordered;
    if (cfg.ordered) { cfg.lds_scene = false; cfg.lds_bytes = 0; }        // (a scene with media has no such hierarchy)
    // (a scene with instances keeps the reference's two-level tree in global memory)
    cfg.ref_tree_bytes = cfg.lds_scene && !cfg.instanced ? (((size_t)scene->dev.n_nodes * (sizeof(GpuNode) + sizeof(uint32_t)) + 15) & ~(size_t)15) : 0;
    // the visibility instantiation of a staged scene holds scene, stack and the reference's tree at once: all of it has to
    // fit a compute unit's 160 KiB (it does up to ~520 triangles, which is where kSceneLdsLimit ends too)
    if (cfg.lds_scene && cfg.lds_bytes + (size_t)(kn.lds_stack + 1) * kBlock * sizeof(double) + cfg.ref_tree_bytes + 1024 > ((size_t)160 << 10)) {
        cfg.lds_scene = false; cfg.lds_bytes = 0; cfg.ref_tree_bytes = 0;
    }
    // k_connect tests the far end's own triangle from tables in LDS (the reference's tree, triangles, grazing thresholds,
    // leaf nodes): for scenes where those are small beside the vertex records the kernel stages (up to ~70 triangles)
    // (a scene with instances: the hierarchy's own triangle table, the objects, and two words per entry)
    cfg.n_entries = (uint32_t)scene->host.xtris.size();
    const size_t own_bytes = cfg.instanced ? ((scene->host.xtris.size() * (sizeof(GpuTri) + 8) + scene->dev.n_objs * sizeof(CxObject) + 15) & ~(size_t)15)
                                           : cfg.ref_tree_bytes + (size_t)scene->dev.n_tris * (sizeof(GpuTri) + 8);
    cfg.own_test = cfg.lds_scene && own_bytes <= ((size_t)16 << 10) && !kn.no_own_test;
    {
        const uint32_t d = (uint32_t)(depth > 0 ? depth : 1);
        uint32_t budget = kn.conn_budget > 0 ? (uint32_t)kn.conn_budget : 128u;      // (measured: 96 and 160 are 2 % slower, 240 is 6 %)
        if (budget < 2u * d) budget = 2u * d;
        uint32_t waves = 4;
        while (waves > 1 && conn_lds_need(waves, budget, cfg.textured, cfg.own_test ? own_bytes : 0) > ((size_t)64 << 10)) waves /= 2;
        if (cfg.own_test && conn_lds_need(waves, budget, cfg.textured, own_bytes) > kConnLdsMost) cfg.own_test = false;
        cfg.conn_waves = waves;
        cfg.conn_rec_budget = budget;
        cfg.conn_lds_bytes = conn_lds_need(waves, budget, cfg.textured, cfg.own_test ? own_bytes : 0);
        if (cfg.conn_lds_bytes > kConnLdsMost) return fail(BDPT_ERR_UNSUPPORTED, "max_depth too large for the connection kernel's staging area");
        // a wave takes the shadow queue in pieces that hold the pairs of any one sample at least four times over, so that
        // at most a quarter of a piece is lost when the entries of a sample would straddle two pieces
        const uint32_t sample_pairs = ((d * d + 63u) / 64u) * 64u;
        cfg.conn_chunk = 4u * sample_pairs < 256u ? 256u : 4u * sample_pairs;
        cfg.conn_blocks = 1; cfg.conn_passes = 1; cfg.conn_pass_pairs = 0;      // (render_wavefront sizes these with the batch)
    }
    cfg.top_nodes = (uint32_t)kn.lds_kb * 1024u / 128u;
    cfg.lds_stack = kn.lds_stack;
    int per_cu = 1;
    HIP_TRY(wf_prepare_traverse(cfg, &per_cu));
    // Three traversal workgroups per CU (3 x 88 registers per SIMD lane) leave room for TWO waves of a shading kernel
    // (<= 120 registers) on every SIMD; with four, one fits.  More resident traversal waves only thrash L1 on large
    // scenes anyway.  Measured: 3 beats 4 by about 1 % on every workload.
    if (per_cu > 3) per_cu = 3;
    if (kn.blocks_per_cu > 0) per_cu = kn.blocks_per_cu;
    cfg.persistent_blocks = ctx->cu_count * per_cu;
    cfg.refill_threshold = kn.refill;
    cfg.leaf_threshold = kn.leaf;
    cfg.leaf_inner = kn.leaf_inner > 0 ? kn.leaf_inner : (cfg.leaf_threshold > 1 ? cfg.leaf_threshold / 2 : 1);
    cfg.chunk = (uint32_t)kn.chunk;   // measured: the cursor is one address, and its atomics cost more than a coarser split
    cfg.stream_blocks = ctx->cu_count * 8;
    // Paths of a shading tile that go on are grouped by the part of the scene they leave from when the scene has no
    // lambertian surface (specular paths from one place stay together: headline scene +2.3 %); where most bounces are diffuse
    // the next rays have nothing in common anyway and the scattered writes of the grouping cost more than it gains
    // (Cornell box -17 %, coffee machine -3 %, 10.2 M-triangle lattice -5 %).
    cfg.sort_shift = -1;
    const int sort_cells = kn.sort_cells >= 0 ? kn.sort_cells : (scene->host.has_lambertian ? 0 : 64);
    if (sort_cells > 1) {             // the smallest shift that leaves at most sort_cells cells
        const uint64_t n_pos = scene->host.stris.size();
        int shift = 0;
        while ((n_pos >> shift) > (uint64_t)sort_cells) shift++;
        cfg.sort_shift = shift;
    }
    // (one workgroup per compute unit: the list is short - a few rays in 10^5 - and every workgroup of the launch has to find a
    // place beside the resident traversal waves of the other batches before the stream can go on; eight per unit cost the
    // coffee machine 4 % of its frame)
    cfg.redo_blocks = kn.redo_blocks > 0 ? kn.redo_blocks : ctx->cu_count;
    return BDPT_OK;
}

// true when the four-wide hierarchy of the scene fits the traversal stack of the pipeline's kernels
bool pipeline_can_walk(const bdpt_scene* scene) { return (int)scene->host.sah_stack + 3 <= kQuadStackEntries; }

// ---- wavefront pipeline: batches of samples stream through per-stage kernels (wavefront.h)
// Several batches are in flight at once, each on its own internal stream with its own workspace:
// every bounce of a batch ends with the few longest rays of that bounce, and while one batch sits in
// such a tail the waves of the other batches keep the chip busy.  Nothing in the pipeline waits for
// the host: kernels whose queue turned out empty return at once.
// fixed_ws != null: the per-sample state lives in that caller-provided region of fixed_bytes bytes and nothing is
// allocated; otherwise in the context's own allocation, grown on demand up to the context's workspace limit.
int render_wavefront(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam, uint64_t seed, const int32_t* rows,
                     int x0, int x1, int64_t n_owned, double* d_fb, int stats_mode, bool time_stages, bool unidirectional,
                     hipStream_t stream, void* fixed_ws, size_t fixed_bytes) {
    const Knobs& kn = ctx->knobs;
    const int depth = cam->max_depth;
    const int64_t spp = (int64_t)cam->sqrt_spp * cam->sqrt_spp;
    // A connection needs an eye vertex that is neither delta nor on an emitter (camera.h:441, material.h:35-37), i.e.
    // a lambertian one.  A scene without a lambertian triangle has none, in any sample: the light subpaths and the
    // connection stage are left out altogether (what k_gen_light decides per sample, decided once per scene), and
    // no path vertex, visibility bit or shadow ray needs room.  The counting modes still run everything: they
    // report the draws and pairs of every sample.
    // (the unidirectional estimator forms no connections; it keeps one record per level of its recursion instead)
    const bool connections = !unidirectional && (scene->host.has_lambertian || stats_mode != 0);
    const size_t vslots = unidirectional ? (size_t)(depth > 0 ? depth : 1) : (connections ? 2 * (size_t)(depth > 0 ? depth : 1) : 0);
    // (a scene with media keeps no shadow queue: its pairs are traced one after the other, k_conn_ordered)
    const bool conn_queue = connections && !scene->ordered;
    WfLaunchCfg cfg;
    { int rc = make_launch_cfg(ctx, scene, cfg, depth); if (rc != BDPT_OK) return rc; }
    const uint64_t d1 = (uint64_t)(depth > 0 ? depth : 1);
    const size_t rec_doubles = cfg.textured ? 13 : 10;                     // words of a vertex record (device_core.h)
    if (!scene->ordered && !pipeline_can_walk(scene)) return fail(BDPT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
    if (n_owned >= ((int64_t)1 << 31)) return fail(BDPT_ERR_UNSUPPORTED, "more than 2^31 pixels in one call");
    // A shadow ray of a scene whose connection kernel does not test the far end's triangle itself first tests that
    // triangle in the traversal (WfBuffers::shadow_first, a 32-bit word per ray): +17 % on the 51 k-triangle coffee machine,
    // +9 % on a 0.96 M-triangle lattice, +15 % on 10.2 M triangles.  (BDPT_WF_FIRST_LEAF_MB limits it to triangle tables up
    // to that size.)
    const bool first_leaf = conn_queue && !cfg.own_test && scene->dev.tri_xpos != nullptr &&
                            scene->host.xtris.size() * sizeof(GpuTri) <= ((size_t)ctx->knobs.first_leaf_mb << 20) && !ctx->knobs.no_first_leaf;
    // bytes of one shadow-queue entry: ray (7 doubles), flag, then either the kept contribution or the pair's indices and
    // the triangle its walk starts at
    const size_t entry_bytes = !conn_queue ? 0 : 7 * sizeof(double) + 1 + (cfg.own_test ? 3 * sizeof(double) : 4 + (first_leaf ? 4 : 0));
    // queue entries per sample slot: max_depth of them by default (a sample has up to depth x depth pairs: the passes
    // of the connection stage take as many samples at a time as the queue holds for certain)
    const double cap_per_slot = !conn_queue ? 0.0 : (double)d1 * kn.conn_cap_tenths / 10.0;
    // bytes of state per sample slot
    // (ray, throughput and stream position twice: by entry of the live queue and of the next one)
    const size_t per_slot = sizeof(double) * (2 * 6 + 2 * 3 + 1 + 3) + sizeof(uint32_t) * (5 + 2) + (scene->ordered ? 4 : 0) + (cfg.tex_pass ? 3 * sizeof(double) : 0) + vslots * rec_doubles * sizeof(double) +
                            (conn_queue ? 8 + 1 : 0) +
                            // the queue: a third more than what a pass may be given (pieces partly lost), a quarter more for
                            // the pieces the waves leave unused at the end (conn_blocks below is limited accordingly)
                            (size_t)(cap_per_slot * (4.0 / 3.0 + 0.25) * (double)entry_bytes + 1.0);
    // Passes of the connection stage: the queue holds `cap` entries, of which the pieces the waves of k_connect leave partly
    // unused at the end (`slack`) and up to a quarter of every piece may be lost; what remains is what a pass may be given
    // (pass_pairs, at least two tiles' worth).  Every pass but the last is filled to within one tile, so this many passes
    // take whatever the samples turn out to have (most of them are empty and return at once).
    const uint64_t tile_worst = (uint64_t)kConnTile * d1 * d1;
    auto conn_plan = [&](uint64_t slots, uint32_t blocks, uint64_t& cap, uint64_t& pass_pairs, uint32_t& passes) {
        const uint64_t slack = (uint64_t)blocks * cfg.conn_waves * cfg.conn_chunk;
        uint64_t useful = (uint64_t)((double)slots * cap_per_slot);
        if (useful < 3 * tile_worst) useful = 3 * tile_worst;
        cap = useful + useful / 3 + slack + 64;          // three quarters of (cap - slack) >= useful
        pass_pairs = useful;
        const uint64_t worst = slots * d1 * d1;
        passes = (uint32_t)((worst + (useful - tile_worst) - 1) / (useful - tile_worst)) + 1u;
    };
    // (a large batch needs ~ depth x depth / cap_per_slot passes, a batch of a few hundred slots up to 1.5 x depth)
    const size_t by_queue_size = conn_queue ? (size_t)((double)(d1 * d1) / cap_per_slot) : 0;
    const size_t max_passes = conn_queue ? 2 * (by_queue_size > d1 ? by_queue_size : (size_t)d1) + 8 : 0;
    const size_t n_counters = 8 + 6 * (size_t)d1 + 2 * max_passes;      // queue sizes and cursors of every launch, sizes of the k_redo lists
    // Wavefront state of all batches in flight.  Large batches mean few, long launches (a frame is tens of thousands
    // of them), so a host that wants the last tenth of throughput raises the limit (bdpt_context_set_workspace_limit).
    size_t budget;
    if (fixed_ws) {
        budget = fixed_bytes;
    } else if (kn.budget_gb > 0) {
        budget = (size_t)kn.budget_gb << 30;
    } else {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t reach = free_b + ctx->wf_blob_bytes;
        if (ctx->workspace_limit) {
            budget = ctx->workspace_limit;
            if (budget > reach / 10 * 9) budget = reach / 10 * 9;
        } else {
            budget = kDefaultWorkspaceLimit;
            if (budget > reach / 4) budget = reach / 4;
        }
    }
    int want_streams = kn.streams;
    // room for the 256-byte alignment of the arrays of a workspace, its counters, the unused pieces of the shadow queue, the
    // least size of that queue, the tile sums and the vertex arrays' last group
    const size_t per_stream_overhead = 24 * 256 + align256(n_counters * 4) + align256((max_passes + 1) * 4) +
                                       (conn_queue ? ((size_t)cfg.conn_waves * cfg.conn_chunk + 4 * tile_worst + 64) * entry_bytes : 0) +
                                       (size_t)kVtxGroup * vslots * rec_doubles * sizeof(double);
    if (budget / want_streams <= per_stream_overhead + per_slot * (size_t)spp) want_streams = 1;
    if (budget <= per_stream_overhead + per_slot * (size_t)spp)
        return fail(BDPT_ERR_UNSUPPORTED, fixed_ws ? "the samples of one pixel exceed the static workspace (samples per pixel x depth too large for bdpt_render_device_static)"
                                                   : "samples per pixel exceed the wavefront batch capacity");
    int64_t ns_target = (int64_t)((budget / want_streams - per_stream_overhead) / per_slot);
    int64_t ns_cap = (int64_t)1 << kn.batch_log2;
    // (queue entries are numbered with 32 bits: 2^31 of them leave room for the pieces the waves take)
    if (conn_queue && (double)ns_cap * cap_per_slot * (4.0 / 3.0 + 0.25) > 2147483648.0)
        ns_cap = (int64_t)(2147483648.0 / (cap_per_slot * (4.0 / 3.0 + 0.25)));
    if (ns_target > ns_cap) ns_target = ns_cap;
    if (spp > ns_target) return fail(BDPT_ERR_UNSUPPORTED, "samples per pixel exceed the wavefront batch capacity");
    int64_t pixels_per_batch = ns_target / spp;
    // small renders: still three batches per stream, so that the tails of the batches overlap
    const int64_t even_share = (n_owned + 3 * want_streams - 1) / (3 * want_streams);
    if (pixels_per_batch > even_share) pixels_per_batch = even_share > 0 ? even_share : 1;
    // The component arrays of the state lie ns elements apart: an odd pixel count keeps that distance away from the
    // large powers of two at which all components of a slot would fall into the same memory channel.
    if (pixels_per_batch > 2 && pixels_per_batch % 2 == 0) pixels_per_batch--;
    const uint64_t ns = (uint64_t)(pixels_per_batch * spp);
    // k_connect's waves take the shadow queue in pieces and may each leave one partly unused: one workgroup per 4096 slots
    // (at most four per compute unit, and never so many that those pieces would take more than a quarter of the queue)
    uint64_t shadow_cap = 0;
    if (conn_queue) {
        uint64_t blocks = ns / 4096;
        const uint64_t by_queue = (uint64_t)((double)ns * cap_per_slot) / (4ull * cfg.conn_waves * cfg.conn_chunk);
        if (blocks > by_queue) blocks = by_queue;
        if (blocks > (uint64_t)ctx->cu_count * 4) blocks = (uint64_t)ctx->cu_count * 4;
        if (blocks < 1) blocks = 1;
        cfg.conn_blocks = (uint32_t)blocks;
        uint64_t pass_pairs = 0;
        uint32_t passes = 1;
        conn_plan(ns, cfg.conn_blocks, shadow_cap, pass_pairs, passes);
        if (passes > max_passes) return fail(BDPT_ERR_UNSUPPORTED, "internal: connection passes exceed their counters");
        if (shadow_cap >= ((uint64_t)1 << 32)) return fail(BDPT_ERR_UNSUPPORTED, "internal: shadow queue larger than its 32-bit indices");
        cfg.conn_passes = passes;
        cfg.conn_pass_pairs = pass_pairs;
    }
    const int64_t n_batches = (n_owned + pixels_per_batch - 1) / pixels_per_batch;
    const int n_streams = (int)(n_batches < want_streams ? n_batches : want_streams);

    // carve one workspace per stream
    size_t off = 0;
    auto take = [&off](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t o_ray = take(ns * 6 * 8), o_ray2 = take(ns * 6 * 8), o_thr = take(ns * 3 * 8), o_thr2 = take(ns * 3 * 8), o_hit = take(ns * 8), o_res = take(ns * 3 * 8);
    const size_t o_tri = take(ns * 4), o_ctr = take(ns * 4), o_cq1 = take(ns * 4), o_cq2 = take(ns * 4), o_cnt = take(ns * 4);
    const size_t o_q0 = take(ns * 4), o_q1 = take(ns * 4), o_cq = take(scene->ordered ? ns * 4 : 0), o_tex = take(cfg.tex_pass ? ns * 3 * 8 : 0);
    const uint64_t ns_grouped = (ns + kVtxGroup - 1) / kVtxGroup * kVtxGroup;      // whole groups of slots
    const size_t o_vtx = take(ns_grouped * vslots * rec_doubles * 8), o_sh = take(shadow_cap * 7 * 8), o_shvis = take(shadow_cap);
    const size_t o_contrib = take(cfg.own_test ? shadow_cap * 3 * 8 : 0), o_pair = take(conn_queue && !cfg.own_test ? shadow_cap * 4 : 0);
    const size_t o_first = take(first_leaf ? shadow_cap * 4 : 0);
    const size_t o_ef = take(conn_queue ? ns * 4 : 0), o_ec = take(conn_queue ? ns * 4 : 0);
    const size_t o_tiles = take(conn_queue ? (ns / kConnTile + 2) * 8 : 0), o_bounds = take((max_passes + 2) * 4);
    const size_t o_counters = take(n_counters * 4);
    // Measured: the batches in flight run 2.5 % faster when their workspaces lie at least ~30 GB apart in the
    // allocation than when they are neighbours (a total that large without the spacing does not help).  A frame big
    // enough to notice therefore spreads its workspaces over the whole budget even if it needs less.
    size_t per_stream = off;
    if (!fixed_ws && per_stream * (size_t)n_streams >= ((size_t)4 << 30)) {
        const size_t spread = (budget / (size_t)n_streams) & ~(size_t)255;
        if (spread > per_stream) per_stream = spread;
    }
    const size_t need = per_stream * (size_t)n_streams;
    char* ws_base;
    if (fixed_ws) {
        if (need > fixed_bytes) return fail(BDPT_ERR_UNSUPPORTED, "internal: static workspace layout exceeds its size");
        ws_base = static_cast<char*>(fixed_ws);
    } else {
        if (ctx->wf_blob_bytes < need) {
            if (ctx->wf_blob) (void)hipFree(ctx->wf_blob);
            ctx->wf_blob = nullptr;
            ctx->wf_blob_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->wf_blob, need));
            ctx->wf_blob_bytes = need;
        }
        ws_base = static_cast<char*>(ctx->wf_blob);
    }
    // Upper bound of |direction| over all rays of the frame (the traversal kernels scale a per-triangle threshold with
    // it): camera rays run from the lens to a point of the viewport, every other ray is a unit vector, or a unit
    // vector plus fuzz (<= 1) times a unit vector (material.h:75).
    float ray_len = 2.0f;
    {
        const double disk = std::sqrt(cam->defocus_disk_u[0] * cam->defocus_disk_u[0] + cam->defocus_disk_u[1] * cam->defocus_disk_u[1] + cam->defocus_disk_u[2] * cam->defocus_disk_u[2]) +
                            std::sqrt(cam->defocus_disk_v[0] * cam->defocus_disk_v[0] + cam->defocus_disk_v[1] * cam->defocus_disk_v[1] + cam->defocus_disk_v[2] * cam->defocus_disk_v[2]);
        for (int corner = 0; corner < 4; corner++) {
            const double i = (corner & 1) ? cam->image_width + 1.0 : -1.0, j = (corner & 2) ? cam->image_height + 1.0 : -1.0;
            double l2 = 0.0;
            for (int k = 0; k < 3; k++) {
                const double c = cam->pixel00_loc[k] + i * cam->pixel_delta_u[k] + j * cam->pixel_delta_v[k] - cam->center[k];
                l2 += c * c;
            }
            const float l = (float)((std::sqrt(l2) + (cam->defocus_angle > 0 ? disk : 0.0)) * 1.001);
            if (!(l <= ray_len)) ray_len = l;        // (a NaN bound stays: every hit is then re-checked)
        }
    }
    WfBuffers wbs[kWfStreams];
    for (int s = 0; s < n_streams; s++) {
        char* base = ws_base + per_stream * (size_t)s;
        WfBuffers& wb = wbs[s];
        wb.ray = reinterpret_cast<double*>(base + o_ray); wb.thr = reinterpret_cast<double*>(base + o_thr);
        wb.ray_next = reinterpret_cast<double*>(base + o_ray2); wb.thr_next = reinterpret_cast<double*>(base + o_thr2);
        wb.ctrq = reinterpret_cast<uint32_t*>(base + o_cq1); wb.ctrq_next = reinterpret_cast<uint32_t*>(base + o_cq2);
        wb.hit = reinterpret_cast<double*>(base + o_hit); wb.result = reinterpret_cast<double*>(base + o_res);
        wb.tmax = nullptr;
        wb.hit_tri = reinterpret_cast<int32_t*>(base + o_tri); wb.ctr = reinterpret_cast<uint32_t*>(base + o_ctr);
        wb.counts = reinterpret_cast<uint32_t*>(base + o_cnt);
        wb.queue[0] = reinterpret_cast<uint32_t*>(base + o_q0); wb.queue[1] = reinterpret_cast<uint32_t*>(base + o_q1);
        wb.conn_queue = reinterpret_cast<uint32_t*>(base + o_cq);
        wb.tex_rgb = cfg.tex_pass ? reinterpret_cast<double*>(base + o_tex) : nullptr;
        wb.vtx = reinterpret_cast<double*>(base + o_vtx);
        wb.contrib = cfg.own_test ? reinterpret_cast<double*>(base + o_contrib) : nullptr;
        wb.shadow = reinterpret_cast<double*>(base + o_sh); wb.counters = reinterpret_cast<uint32_t*>(base + o_counters);
        wb.sh_vis = reinterpret_cast<uint8_t*>(base + o_shvis); wb.sh_pair = reinterpret_cast<uint32_t*>(base + o_pair);
        wb.shadow_first = first_leaf ? reinterpret_cast<uint32_t*>(base + o_first) : nullptr;
        wb.ent_first = reinterpret_cast<uint32_t*>(base + o_ef); wb.ent_count = reinterpret_cast<uint32_t*>(base + o_ec);
        wb.tile_pairs = reinterpret_cast<unsigned long long*>(base + o_tiles); wb.pass_bounds = reinterpret_cast<uint32_t*>(base + o_bounds);
        wb.ns = ns; wb.shadow_cap = shadow_cap;
        wb.vtx_words = (uint32_t)(vslots * rec_doubles);
        wb.ray_len_ptr = nullptr;
        wb.ray_len = ray_len;
    }


    const GpuCamera gcam = to_gpu_camera(*cam);
    unsigned long long* stats = stats_mode ? ctx->stats : nullptr;

    ctx->stage_pairs = 0;
    ctx->stage_kind.clear();
    // event pair around one traversal launch on its own stream
    auto timed = [&](int kind, hipStream_t st, auto&& launch) -> int {
        if (!time_stages) { HIP_TRY(launch()); return BDPT_OK; }
        const size_t at = 2 * ctx->stage_pairs;
        while (ctx->stage_events.size() < at + 2) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            ctx->stage_events.push_back(ev);
        }
        HIP_TRY(hipEventRecord(ctx->stage_events[at], st));
        HIP_TRY(launch());
        HIP_TRY(hipEventRecord(ctx->stage_events[at + 1], st));
        ctx->stage_kind.push_back(kind);
        ctx->stage_pairs++;
        return BDPT_OK;
    };

    // the internal streams start after everything already queued on the caller's stream ...
    HIP_TRY(hipEventRecord(ctx->wf_fork, stream));
    for (int s = 0; s < n_streams; s++) HIP_TRY(hipStreamWaitEvent(ctx->wf_streams[s], ctx->wf_fork, 0));

    int64_t batch = 0;
    for (int64_t first = 0; first < n_owned; first += pixels_per_batch, batch++) {
        const int si = (int)(batch % n_streams);
        hipStream_t st = ctx->wf_streams[si];
        const WfBuffers& wb = wbs[si];
        uint32_t* C = wb.counters;
        auto eye_count = [&](int k) { return C + 8 + 2 * k; };
        auto eye_cursor = [&](int k) { return C + 8 + 2 * k + 1; };
        auto light_count = [&](int k) { return C + 8 + 2 * depth + 2 * k; };
        auto light_cursor = [&](int k) { return C + 8 + 2 * depth + 2 * k + 1; };
        auto eye_redo = [&](int k) { return C + 8 + 4 * depth + k; };
        auto light_redo = [&](int k) { return C + 8 + 5 * depth + k; };
        auto shadow_count = [&](int p) { return C + 8 + 6 * depth + 2 * p; };
        auto shadow_cursor = [&](int p) { return C + 8 + 6 * depth + 2 * p + 1; };
        uint32_t* conn_count = C;

        WfBatch b;
        b.seed = seed;
        b.rows = rows;
        b.row_w = x1 - x0; b.x0 = x0;
        b.first_pixel = first;
        b.n_pixels = (int32_t)((n_owned - first < pixels_per_batch) ? (n_owned - first) : pixels_per_batch);
        b.pixels_stride = b.n_pixels;             // every slot below n_pixels * spp is a real sample
        b.sample_stride = kn.pixel_major ? (int32_t)spp : 1;
        b.n_owned = n_owned;
        const uint32_t n_slots = (uint32_t)((int64_t)b.n_pixels * spp);

        // the path state by queue entry: launch k reads the arrays of parity k & 1 and writes those of the other
        auto step = [&](int k) {
            WfBuffers w = wb;
            if (k & 1) { std::swap(w.ray, w.ray_next); std::swap(w.thr, w.thr_next); std::swap(w.ctrq, w.ctrq_next); }
            return w;
        };
        const bool keep_draws = connections || stats_mode != 0;     // (see k_shade)
        HIP_TRY(hipMemsetAsync(C, 0, n_counters * 4, st));
        HIP_TRY(wf_gen_eye(scene->dev, gcam, wb, b, cfg, st));
        int cur = 0;
        // unidirectional estimator: camera.h:255-292, one launch pair per level of the recursion
        for (int k = 0; k < depth && unidirectional; k++) {
            const WfBuffers w = step(k);
            const uint32_t* q_in = (k == 0) ? nullptr : wb.queue[cur];
            const uint32_t* c_in = (k == 0) ? nullptr : eye_count(k - 1);
            const int nxt = (k == 0) ? 0 : (cur ^ 1);
            { int rc = timed(0, st, [&] { return wf_extend(scene->dev, w, c_in, n_slots, eye_cursor(k), cfg, stats, st, wb.queue[nxt], eye_redo(k)); }); if (rc != BDPT_OK) return rc; }
            HIP_TRY(wf_redo(scene->dev, w, cfg, st, wb.queue[nxt], eye_redo(k)));
            HIP_TRY(wf_shade_unidirectional(scene->dev, gcam, w, b, q_in, c_in, n_slots, wb.queue[nxt], eye_count(k), k, k == depth - 1, cfg, stats, st));
            cur = nxt;
        }
        // eye subpath: camera.h:303 -> trace_path(r, depth, ...)
        for (int k = 0; k < depth && !unidirectional; k++) {
            const WfBuffers w = step(k);
            const uint32_t* q_in = (k == 0) ? nullptr : wb.queue[cur];
            const uint32_t* c_in = (k == 0) ? nullptr : eye_count(k - 1);
            const int nxt = (k == 0) ? 0 : (cur ^ 1);
            { int rc = timed(0, st, [&] { return cfg.ordered ? wf_walk_ordered(scene->dev, gcam, w, b, q_in, c_in, n_slots, cfg, st)
                                                             : wf_extend(scene->dev, w, c_in, n_slots, eye_cursor(k), cfg, stats, st, wb.queue[nxt], eye_redo(k)); }); if (rc != BDPT_OK) return rc; }
            if (!cfg.ordered) HIP_TRY(wf_redo(scene->dev, w, cfg, st, wb.queue[nxt], eye_redo(k)));
            if (cfg.tex_pass) HIP_TRY(wf_tex_eval(scene->dev, gcam, w, b, q_in, c_in, n_slots, cfg, st));
            HIP_TRY(wf_shade(scene->dev, gcam, w, b, true, q_in, c_in, n_slots, wb.queue[nxt], eye_count(k), k == depth - 1, connections, keep_draws, cfg,
                             stats, st));
            cur = nxt;
        }
        // light subpath: camera.h:313 -> build_light_path -> trace_path(light_ray, depth - 1, ...)
        if (depth > 0 && connections) {
            // (k_gen_light writes the `_next` arrays: those of parity 1 are the input of launch 0)
            HIP_TRY(wf_gen_light(scene->dev, gcam, step(1), b, wb.queue[0], light_count(0), cfg, stats, st));
            cur = 0;
            for (int k = 0; k < depth - 1; k++) {
                const WfBuffers w = step(k);
                { int rc = timed(0, st, [&] { return cfg.ordered ? wf_walk_ordered(scene->dev, gcam, w, b, wb.queue[cur], light_count(k), 0, cfg, st)
                                                                 : wf_extend(scene->dev, w, light_count(k), 0, light_cursor(k), cfg, stats, st, wb.queue[cur ^ 1], light_redo(k)); }); if (rc != BDPT_OK) return rc; }
                if (!cfg.ordered) HIP_TRY(wf_redo(scene->dev, w, cfg, st, wb.queue[cur ^ 1], light_redo(k)));
                if (cfg.tex_pass) HIP_TRY(wf_tex_eval(scene->dev, gcam, w, b, wb.queue[cur], light_count(k), 0, cfg, st));
                HIP_TRY(wf_shade(scene->dev, gcam, w, b, false, wb.queue[cur], light_count(k), 0, wb.queue[cur ^ 1],
                                 light_count(k + 1), k == depth - 2, true, keep_draws, cfg, stats, st));
                cur ^= 1;
            }
        }
        // connections: camera.h:316-320
        if (connections && cfg.ordered) {
            HIP_TRY(wf_conn_prepare(gcam, wb, b, conn_count, cfg, stats, st));
            HIP_TRY(wf_conn_ordered(scene->dev, gcam, wb, b, conn_count, cfg, st));
        }
        if (conn_queue) {
            HIP_TRY(wf_conn_count(wb, n_slots, cfg, stats, st));
            HIP_TRY(wf_conn_plan(wb, n_slots, cfg, st));
            for (int p = 0; p < (int)cfg.conn_passes; p++) {
                HIP_TRY(wf_connect(scene->dev, gcam, wb, n_slots, p, shadow_count(p), cfg, stats, st));
                { int rc = timed(1, st, [&] { return wf_occlude(scene->dev, wb, shadow_count(p), 0, shadow_cursor(p), cfg, stats, st); }); if (rc != BDPT_OK) return rc; }
                HIP_TRY(wf_conn_add(scene->dev, gcam, wb, n_slots, p, cfg, st));
            }
        }
        HIP_TRY(wf_resolve(gcam, wb, b, d_fb, cfg, st));
    }
    // ... and the caller's stream continues after all of them
    for (int s = 0; s < n_streams; s++) {
        HIP_TRY(hipEventRecord(ctx->wf_join[s], ctx->wf_streams[s]));
        HIP_TRY(hipStreamWaitEvent(stream, ctx->wf_join[s], 0));
    }
    return BDPT_OK;
}

// fixed_ws != null: nothing is allocated - the row table and the per-sample state live in that region
int render_impl(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam, const bdpt_render_params* p,
                double* d_fb, hipStream_t stream, void* fixed_ws = nullptr, size_t fixed_bytes = 0) {
    if (!ctx || !scene || !cam || !p || !d_fb) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->ctx != ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "scene belongs to another context");
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(BDPT_ERR_INVALID_ARGUMENT, "empty image");
    if (cam->sqrt_spp < 1) return fail(BDPT_ERR_INVALID_ARGUMENT, "sqrt_spp must be at least 1");
    if (cam->max_depth < 0 || cam->max_depth > 4095) return fail(BDPT_ERR_INVALID_ARGUMENT, "max_depth out of range");   // vertex counts are kept in 12 bits
    if ((int64_t)cam->sqrt_spp * cam->sqrt_spp > (1 << 24)) return fail(BDPT_ERR_INVALID_ARGUMENT, "too many samples per pixel");

    const int W = cam->image_width, H = cam->image_height;
    int x0 = p->x0 < 0 ? 0 : p->x0, y0 = p->y0 < 0 ? 0 : p->y0;
    int x1 = (p->x1 <= 0 || p->x1 > W) ? W : p->x1;
    int y1 = (p->y1 <= 0 || p->y1 > H) ? H : p->y1;
    const int shard_count = p->shard_count <= 0 ? 1 : p->shard_count;
    const int shard_rows = p->shard_rows <= 0 ? 1 : p->shard_rows;
    if (p->shard_index < 0 || p->shard_index >= shard_count) return fail(BDPT_ERR_INVALID_ARGUMENT, "shard_index out of range");
    const int stats_mode = (p->flags & BDPT_RENDER_COUNT_STATS) ? 1 : ((p->flags & BDPT_RENDER_COUNT_WORK) ? 2 : 0);
    const bool want_stats = stats_mode != 0;
    const bool want_time = (p->flags & BDPT_RENDER_TIME_KERNEL) != 0;
    const bool time_stages = (p->flags & BDPT_RENDER_TIME_STAGES) != 0;
    const bool unidirectional = (p->flags & BDPT_RENDER_UNIDIRECTIONAL) != 0;
    // Reference-equivalent counters need the reference's traversal of every shadow ray: that is the single-kernel
    // path, which also takes over when the four-wide hierarchy is deeper than the pipeline's traversal stack.
    // Everything else runs the wavefront pipeline unless the caller asks otherwise.
    const bool cx_single = scene->composite && (scene->ordered ? (stats_mode != 0 || (p->flags & BDPT_RENDER_MEGAKERNEL) || unidirectional)
                                                               : (!scene->pipeline || stats_mode == 1 || (p->flags & BDPT_RENDER_MEGAKERNEL) || unidirectional || !pipeline_can_walk(scene) ||
                                                                  !pipeline_can_connect(scene, cam->max_depth)));
    const bool single_kernel = cx_single || stats_mode == 1 || (p->flags & BDPT_RENDER_MEGAKERNEL) || (!scene->ordered && !pipeline_can_walk(scene)) ||
                               (!scene->ordered && !unidirectional && !pipeline_can_connect(scene, cam->max_depth));
    if (cx_single && fixed_ws) return fail(BDPT_ERR_UNSUPPORTED, "these flags put a composite scene on the single-kernel path, which allocates");
    if (fixed_ws && (single_kernel || time_stages))
        return fail(BDPT_ERR_UNSUPPORTED, pipeline_can_walk(scene)
                        ? "bdpt_render_device_static supports the flags BDPT_RENDER_COUNT_WORK, BDPT_RENDER_TIME_KERNEL and BDPT_RENDER_UNIDIRECTIONAL only"
                        : "BVH deeper than the traversal stack (the single-kernel path that would render it allocates)");
    ctx->stats_valid = false;
    ctx->stage_valid = false;
    ctx->timing_valid = false;
    if (want_stats) HIP_TRY(hipMemsetAsync(ctx->stats, 0, sizeof(unsigned long long) * kStatWords, stream));
    if (want_time) HIP_TRY(hipEventRecord(ctx->ev_start, stream));

    // owned rows: j in [y0, y1) with (j / shard_rows) % shard_count == shard_index; the table is filled on the device
    int64_t n_rows = 0;
    if (x0 < x1 && y0 < y1)
        n_rows = wf_owned_rows_before(y1, shard_rows, shard_count, p->shard_index) - wf_owned_rows_before(y0, shard_rows, shard_count, p->shard_index);
    if (n_rows > 0) {
        int32_t* rows;
        if (fixed_ws) {
            const size_t rows_bytes = align256(sizeof(int32_t) * (size_t)n_rows);
            if (rows_bytes >= fixed_bytes) return fail(BDPT_ERR_UNSUPPORTED, "image too tall for the static workspace");
            rows = static_cast<int32_t*>(fixed_ws);
            fixed_ws = static_cast<char*>(fixed_ws) + rows_bytes;
            fixed_bytes -= rows_bytes;
        } else {
            int rc = ensure_rows(ctx, (size_t)n_rows);
            if (rc != BDPT_OK) return rc;
            rows = ctx->rows;
        }
        HIP_TRY(wf_fill_rows(rows, y0, y1, shard_rows, shard_count, p->shard_index, stream));
        const int64_t n_owned = n_rows * (x1 - x0);
        int rc;
        if (cx_single)
            rc = render_composite(ctx, scene, cam, p->seed, rows, (int32_t)n_rows, x0, x1, n_owned, d_fb, want_stats, unidirectional, stream);
        else if (single_kernel)
            rc = render_megakernel(ctx, scene, cam, p->seed, rows, (int32_t)n_rows, x0, x1, n_owned, d_fb, stats_mode, unidirectional, stream);
        else
            rc = render_wavefront(ctx, scene, cam, p->seed, rows, x0, x1, n_owned, d_fb, stats_mode, time_stages, unidirectional, stream, fixed_ws, fixed_bytes);
        if (rc != BDPT_OK) return rc;
        ctx->stage_valid = time_stages && !single_kernel;
    }
    if (want_time) { HIP_TRY(hipEventRecord(ctx->ev_stop, stream)); ctx->timing_valid = true; }
    ctx->stats_valid = want_stats;
    return BDPT_OK;
}

// ---- ray and visibility queries through the traversal kernels of the pipeline
// Queries are processed in pieces of at most kQueryPiece so that the context's query workspace stays bounded.
constexpr uint64_t kQueryPiece = (uint64_t)1 << 22;

int trace_closest_pipeline(bdpt_context* ctx, const bdpt_scene* scene, const double* d_rays, uint64_t n, double* d_hits, hipStream_t stream) {
    WfLaunchCfg cfg;
    { int rc = make_launch_cfg(ctx, scene, cfg); if (rc != BDPT_OK) return rc; }
    const uint64_t piece = n < kQueryPiece ? n : kQueryPiece;
    size_t off = 0;
    auto take = [&off](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t o_ray = take(piece * 6 * 8), o_tmax = take(piece * 8), o_hit = take(piece * 8), o_tri = take(piece * 4), o_redo = take(piece * 4),
                 o_cur = take(256);   // cursor, the ray length bound, the size of k_redo's list
    { int rc = ensure_query_ws(ctx, off); if (rc != BDPT_OK) return rc; }
    char* base = static_cast<char*>(ctx->query_ws);
    WfBuffers wb{};
    wb.ray = reinterpret_cast<double*>(base + o_ray);
    double* tmax = reinterpret_cast<double*>(base + o_tmax);
    wb.tmax = tmax;
    wb.hit = reinterpret_cast<double*>(base + o_hit);
    wb.hit_tri = reinterpret_cast<int32_t*>(base + o_tri);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(base + o_cur);
    float* ray_len = reinterpret_cast<float*>(cursor + 1);
    wb.ray_len_ptr = ray_len;
    wb.ns = piece;
    for (uint64_t first = 0; first < n; first += piece) {
        const uint32_t m = (uint32_t)(n - first < piece ? n - first : piece);
        const double* rays = d_rays + 8 * first;
        HIP_TRY(hipMemsetAsync(cursor, 0, 3 * sizeof(uint32_t), stream));
        HIP_TRY(wf_rays_in(rays, m, wb, tmax, ray_len, stream));
        HIP_TRY(wf_extend(scene->dev, wb, nullptr, m, cursor, cfg, nullptr, stream, reinterpret_cast<uint32_t*>(base + o_redo), cursor + 2));
        HIP_TRY(wf_redo(scene->dev, wb, cfg, stream, reinterpret_cast<uint32_t*>(base + o_redo), cursor + 2));
        HIP_TRY(wf_hits_out(scene->dev, rays, m, wb, d_hits + 11 * first, stream));
    }
    return BDPT_OK;
}

int visible_pipeline(bdpt_context* ctx, const bdpt_scene* scene, const double* d_pairs, uint64_t n, uint8_t* d_out, hipStream_t stream) {
    WfLaunchCfg cfg;
    { int rc = make_launch_cfg(ctx, scene, cfg); if (rc != BDPT_OK) return rc; }
    const uint64_t piece = n < kQueryPiece ? n : kQueryPiece;
    size_t off = 0;
    auto take = [&off](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t o_sh = take(piece * 7 * 8), o_vis = take(piece), o_cur = take(256);
    { int rc = ensure_query_ws(ctx, off); if (rc != BDPT_OK) return rc; }
    char* base = static_cast<char*>(ctx->query_ws);
    WfBuffers wb{};
    wb.shadow = reinterpret_cast<double*>(base + o_sh);
    wb.sh_vis = reinterpret_cast<uint8_t*>(base + o_vis);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(base + o_cur);
    wb.ns = piece; wb.shadow_cap = piece;
    for (uint64_t first = 0; first < n; first += piece) {
        const uint32_t m = (uint32_t)(n - first < piece ? n - first : piece);
        HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(uint32_t), stream));
        HIP_TRY(wf_pairs_in(d_pairs + 6 * first, m, wb, stream));
        HIP_TRY(wf_occlude(scene->dev, wb, nullptr, m, cursor, cfg, nullptr, stream));
        HIP_TRY(wf_visible_out(wb, m, d_out + first, stream));
    }
    return BDPT_OK;
}

// Uploads s->host as one device allocation (tables at 256-byte aligned offsets) and fills s->dev.
int upload_pipeline_tables(bdpt_context* ctx, bdpt_scene* s) {
    (void)ctx;
    const HostScene& h = s->host;
    // one device allocation, tables at 256-byte aligned offsets
    size_t off_nodes = 0;
    size_t off_tris = off_nodes + align256(h.nodes.size() * sizeof(GpuNode));
    size_t off_np = off_tris + align256(h.tris.size() * sizeof(GpuTri));
    size_t off_st = off_np + align256(h.node_parent.size() * sizeof(uint32_t));
    size_t off_sl = off_st + align256(h.stris.size() * sizeof(GpuTri));
    size_t off_xn = off_sl + align256(h.sleaf.size() * sizeof(uint32_t));
    size_t off_xt = off_xn + align256(h.xnodes.size() * sizeof(GpuQuadNode));
    size_t off_aux = off_xt + align256(h.xtris.size() * sizeof(GpuTri));
    size_t off_mats = off_aux + align256(h.aux.size() * sizeof(GpuTriAux));
    size_t off_ltri = off_mats + align256(h.mats.size() * sizeof(GpuMaterial));
    size_t off_lacc = off_ltri + align256(h.light_tris.size() * sizeof(uint32_t));
    size_t off_tleaf = off_lacc + align256(h.light_accum.size() * sizeof(double));
    size_t off_tgraze = off_tleaf + align256(h.tri_leaf.size() * sizeof(uint32_t));
    size_t off_shade = off_tgraze + align256(h.tri_graze.size() * sizeof(int32_t));
    size_t off_tpos = off_shade + align256(h.shade.size() * sizeof(GpuShadeRec));
    size_t off_objs = off_tpos + align256(h.tri_pos.size() * sizeof(uint32_t));
    size_t off_xobj = off_objs + align256(h.objs.size() * sizeof(CxObject));
    size_t off_pobj = off_xobj + align256(h.xobj.size() * sizeof(int32_t));
    size_t off_oleaf = off_pobj + align256(h.pos_obj.size() * sizeof(int32_t));
    size_t off_opos = off_oleaf + align256(h.obj_leaf.size() * sizeof(uint32_t));
    size_t off_xpos = off_opos + align256(h.obj_pos0.size() * sizeof(uint32_t));
    size_t total = off_xpos + align256(h.tri_xpos.size() * sizeof(uint32_t));
    hipError_t e = hipMalloc(&s->d_blob, total);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(scene)");
    s->blob_bytes = total;
    char* base = static_cast<char*>(s->d_blob);
    struct { size_t off; const void* src; size_t bytes; } parts[] = {
        {off_nodes, h.nodes.data(), h.nodes.size() * sizeof(GpuNode)},
        {off_tris, h.tris.data(), h.tris.size() * sizeof(GpuTri)},
        {off_np, h.node_parent.data(), h.node_parent.size() * sizeof(uint32_t)},
        {off_st, h.stris.data(), h.stris.size() * sizeof(GpuTri)},
        {off_sl, h.sleaf.data(), h.sleaf.size() * sizeof(uint32_t)},
        {off_xn, h.xnodes.data(), h.xnodes.size() * sizeof(GpuQuadNode)},
        {off_xt, h.xtris.data(), h.xtris.size() * sizeof(GpuTri)},
        {off_aux, h.aux.data(), h.aux.size() * sizeof(GpuTriAux)},
        {off_mats, h.mats.data(), h.mats.size() * sizeof(GpuMaterial)},
        {off_ltri, h.light_tris.data(), h.light_tris.size() * sizeof(uint32_t)},
        {off_lacc, h.light_accum.data(), h.light_accum.size() * sizeof(double)},
        {off_tleaf, h.tri_leaf.data(), h.tri_leaf.size() * sizeof(uint32_t)},
        {off_tgraze, h.tri_graze.data(), h.tri_graze.size() * sizeof(int32_t)},
        {off_shade, h.shade.data(), h.shade.size() * sizeof(GpuShadeRec)},
        {off_tpos, h.tri_pos.data(), h.tri_pos.size() * sizeof(uint32_t)},
        {off_objs, h.objs.data(), h.objs.size() * sizeof(CxObject)},
        {off_xobj, h.xobj.data(), h.xobj.size() * sizeof(int32_t)},
        {off_pobj, h.pos_obj.data(), h.pos_obj.size() * sizeof(int32_t)},
        {off_oleaf, h.obj_leaf.data(), h.obj_leaf.size() * sizeof(uint32_t)},
        {off_opos, h.obj_pos0.data(), h.obj_pos0.size() * sizeof(uint32_t)},
        {off_xpos, h.tri_xpos.data(), h.tri_xpos.size() * sizeof(uint32_t)},
    };
    for (auto& p : parts) {
        if (p.bytes == 0) continue;
        e = hipMemcpy(base + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(s->d_blob); s->d_blob = nullptr; return hip_fail(e, "hipMemcpy(scene)"); }
    }
    SceneDev& d = s->dev;
    d.nodes = reinterpret_cast<const GpuNode*>(base + off_nodes);
    d.tris = reinterpret_cast<const GpuTri*>(base + off_tris);
    d.node_parent = reinterpret_cast<const uint32_t*>(base + off_np);
    d.stris = reinterpret_cast<const GpuTri*>(base + off_st);
    d.sleaf = reinterpret_cast<const uint32_t*>(base + off_sl);
    d.tri_leaf = reinterpret_cast<const uint32_t*>(base + off_tleaf);
    d.tri_graze = reinterpret_cast<const int32_t*>(base + off_tgraze);
    d.xnodes = reinterpret_cast<const GpuQuadNode*>(base + off_xn);
    d.xtris = reinterpret_cast<const GpuTri*>(base + off_xt);
    d.n_xnodes = (uint32_t)h.xnodes.size();
    d.aux = reinterpret_cast<const GpuTriAux*>(base + off_aux);
    d.mats = reinterpret_cast<const GpuMaterial*>(base + off_mats);
    d.shade = reinterpret_cast<const GpuShadeRec*>(base + off_shade);
    d.tri_pos = reinterpret_cast<const uint32_t*>(base + off_tpos);
    d.tri_xpos = h.tri_xpos.empty() ? nullptr : reinterpret_cast<const uint32_t*>(base + off_xpos);
    d.light_tris = reinterpret_cast<const uint32_t*>(base + off_ltri);
    d.light_accum = reinterpret_cast<const double*>(base + off_lacc);
    d.light_total_area = h.light_total_area;
    d.coord_bound = h.coord_bound;
    d.bary_margin = h.bary_margin;
    d.n_nodes = (uint32_t)h.nodes.size();
    d.n_tris = (uint32_t)h.tris.size();
    d.n_mats = (uint32_t)h.mats.size();
    d.n_lights = s->composite ? h.n_lights : (uint32_t)h.light_tris.size();
    d.ref_nested = h.ref_nested ? 1u : 0u;
    d.n_objs = (uint32_t)h.objs.size();
    const bool inst = h.vertex_names_entry;      // (instances, textures, media: the kernels of scenes with instances)
    d.objs = inst ? reinterpret_cast<const CxObject*>(base + off_objs) : nullptr;
    d.xobj = inst ? reinterpret_cast<const int32_t*>(base + off_xobj) : nullptr;
    d.pos_obj = inst ? reinterpret_cast<const int32_t*>(base + off_pobj) : nullptr;
    d.obj_leaf = inst ? reinterpret_cast<const uint32_t*>(base + off_oleaf) : nullptr;
    d.obj_pos0 = inst ? reinterpret_cast<const uint32_t*>(base + off_opos) : nullptr;
    const bool tex = s->composite && (s->cx_host.textured || !s->cx_host.pipeline_ok);    // (media: the kernels with 13-word vertex records)
    d.tex = tex ? s->cx_dev.tex : nullptr;
    d.perlin = tex ? s->cx_dev.perlin : nullptr;
    d.image_bytes = tex ? s->cx_dev.image_bytes : nullptr;
    return BDPT_OK;
}

}  // namespace

extern "C" {

const char* bdpt_last_error(void) { return g_last_error.c_str(); }
// used by the host-side sources of this library (scene files), not part of the public header
void bdpt_internal_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

int bdpt_context_create(int device_index, bdpt_context** out) {
    if (!out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null output pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(BDPT_ERR_DEVICE, "no HIP device available");
    if (device_index < 0 || device_index >= count) return fail(BDPT_ERR_INVALID_ARGUMENT, "device index out of range");
    DeviceGuard guard(device_index);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_index));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BDPT_ERR_UNSUPPORTED, std::string("built for gfx950 (MI355X), found ") + prop.gcnArchName);
    bdpt_context* ctx = new (std::nothrow) bdpt_context();
    if (!ctx) return fail(BDPT_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device_index;
    ctx->cu_count = prop.multiProcessorCount;
    ctx->knobs = read_knobs();
    // `device_index` is the current device here (guard above): the address of the static workspace on THIS GPU
    ctx->static_ws_status = static_scratch_address(device_index, &ctx->static_ws);
    if (hipMalloc(&ctx->unit_counter, 256) != hipSuccess || hipMalloc(&ctx->stats, 256) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->wf_fork, hipEventDisableTiming) != hipSuccess) {
        bdpt_context_destroy(ctx);
        return fail(BDPT_ERR_OUT_OF_MEMORY, "device allocation failed");
    }
    for (int i = 0; i < kWfStreams; i++) {
        if (hipStreamCreateWithFlags(&ctx->wf_streams[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->wf_join[i], hipEventDisableTiming) != hipSuccess) {
            bdpt_context_destroy(ctx);
            return fail(BDPT_ERR_DEVICE, "stream creation failed");
        }
    }
    *out = ctx;
    return BDPT_OK;
}

int bdpt_context_destroy(bdpt_context* ctx) {
    if (!ctx) return BDPT_OK;
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->vertex_ws) (void)hipFree(ctx->vertex_ws);
    if (ctx->rows) (void)hipFree(ctx->rows);
    if (ctx->unit_counter) (void)hipFree(ctx->unit_counter);
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->wf_blob) (void)hipFree(ctx->wf_blob);
    if (ctx->query_ws) (void)hipFree(ctx->query_ws);
    for (int i = 0; i < kWfStreams; i++) {
        if (ctx->wf_streams[i]) (void)hipStreamDestroy(ctx->wf_streams[i]);
        if (ctx->wf_join[i]) (void)hipEventDestroy(ctx->wf_join[i]);
    }
    for (hipEvent_t ev : ctx->stage_events) (void)hipEventDestroy(ev);
    if (ctx->wf_fork) (void)hipEventDestroy(ctx->wf_fork);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    delete ctx;
    return BDPT_OK;
}

int bdpt_context_set_workspace_limit(bdpt_context* ctx, uint64_t bytes) {
    if (!ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "null context");
    ctx->workspace_limit = (size_t)bytes;
    return BDPT_OK;
}

int bdpt_context_set_query_walker(bdpt_context* ctx, int walker) {
    if (!ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "null context");
    if (walker != BDPT_WALKER_PIPELINE && walker != BDPT_WALKER_REFERENCE_TREE) return fail(BDPT_ERR_INVALID_ARGUMENT, "unknown walker");
    ctx->query_walker = walker;
    return BDPT_OK;
}

int bdpt_camera_initialize(const bdpt_camera_public* cam, bdpt_camera_state* out) {
    if (!cam || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    std::string err = camera_initialize(*cam, *out);
    if (!err.empty()) return fail(BDPT_ERR_INVALID_ARGUMENT, err);
    return BDPT_OK;
}

int bdpt_scene_create(bdpt_context* ctx, const double* tri_vertices, const int32_t* tri_material, uint32_t n_triangles,
                      const bdpt_material* materials, uint32_t n_materials, const uint32_t* light_triangles,
                      uint32_t n_lights, bdpt_scene** out) {
    if (!ctx || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    bdpt_scene* s = new (std::nothrow) bdpt_scene();
    if (!s) return fail(BDPT_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->ctx = ctx;
    s->n_materials = n_materials;
    std::string err;
    try {
        err = build_host_scene(tri_vertices, tri_material, n_triangles, materials, n_materials, light_triangles, n_lights, s->host);
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(BDPT_ERR_OUT_OF_MEMORY, "host allocation failed while building the scene");
    }
    if (!err.empty()) { delete s; return fail(BDPT_ERR_INVALID_ARGUMENT, err); }

    DeviceGuard guard(ctx->device);
    { int rc = upload_pipeline_tables(ctx, s); if (rc != BDPT_OK) { delete s; return rc; } }
    *out = s;
    return BDPT_OK;
}

int bdpt_scene_create_composite(bdpt_context* ctx, const double* tri_vertices, const int32_t* tri_material, uint32_t n_triangles,
                                const bdpt_material* materials, uint32_t n_materials, const bdpt_texture* textures,
                                uint32_t n_textures, const bdpt_object* objects, uint32_t n_objects,
                                const uint32_t* light_triangles, uint32_t n_lights, bdpt_scene** out) {
    if (!ctx || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (n_objects > 0 && !objects) return fail(BDPT_ERR_INVALID_ARGUMENT, "null object table");
    DeviceGuard guard(ctx->device);
    // sin / cos of the rotation angles (hittable.h:79-81) from the device's deterministic routine: the same bits on
    // every host, and the ones the kernels' own sin / cos would give
    std::vector<double> radians(n_objects ? n_objects : 1, 0.0), sn(radians.size(), 0.0), cs(radians.size(), 1.0);
    const double pi = 3.1415926535897932385;                     // src/main.h:20
    for (uint32_t k = 0; k < n_objects; k++) {
        radians[k] = objects[k].rotate_y_degrees * pi / 180.0;   // degrees_to_radians, main.h:24-26
        if ((objects[k].flags & BDPT_OBJECT_ROTATE_Y) && !(std::fabs(radians[k]) <= 1073741824.0))
            return fail(BDPT_ERR_INVALID_ARGUMENT, "rotation angle out of range");
    }
    if (n_objects > 0) {
        int rc = bdpt_device_sincos(ctx, radians.data(), n_objects, sn.data(), cs.data());
        if (rc != BDPT_OK) return rc;
    }
    bdpt_scene* s = new (std::nothrow) bdpt_scene();
    if (!s) return fail(BDPT_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->ctx = ctx;
    s->composite = true;
    s->n_materials = n_materials;
    std::string err;
    try {
        err = build_composite_scene(tri_vertices, tri_material, n_triangles, materials, n_materials, textures, n_textures, objects,
                                    n_objects, sn.data(), cs.data(), light_triangles, n_lights, s->cx_host);
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(BDPT_ERR_OUT_OF_MEMORY, "host allocation failed while building the scene");
    }
    if (!err.empty()) { delete s; return fail(BDPT_ERR_INVALID_ARGUMENT, err); }

    const CxHostScene& h = s->cx_host;
    struct Part { const void* src; size_t bytes; size_t off; };
    std::vector<CxTexture> tex = h.tex;
    if (tex.empty()) tex.push_back(CxTexture());
    std::vector<CxObject> objs = h.objs;
    if (objs.empty()) objs.push_back(CxObject());
    std::vector<CxPerlin> perlin = h.perlin;
    if (perlin.empty()) perlin.resize(1);
    Part parts[] = {
        {h.nodes.data(), h.nodes.size() * sizeof(GpuNode), 0},          {h.tris.data(), h.tris.size() * sizeof(GpuTri), 0},
        {h.aux.data(), h.aux.size() * sizeof(GpuTriAux), 0},            {h.mats.data(), h.mats.size() * sizeof(GpuMaterial), 0},
        {tex.data(), tex.size() * sizeof(CxTexture), 0},                {objs.data(), objs.size() * sizeof(CxObject), 0},
        {perlin.data(), perlin.size() * sizeof(CxPerlin), 0},           {h.image_bytes.data(), h.image_bytes.size(), 0},
        {h.tri_object.data(), h.tri_object.size() * sizeof(int32_t), 0}, {h.light_tris.data(), h.light_tris.size() * sizeof(uint32_t), 0},
        {h.light_accum.data(), h.light_accum.size() * sizeof(double), 0},
    };
    size_t total = 0;
    for (Part& p : parts) { p.off = total; total += align256(p.bytes); }
    hipError_t e = hipMalloc(&s->d_cx_blob, total);
    if (e != hipSuccess) { delete s; return hip_fail(e, "hipMalloc(scene)"); }
    s->cx_blob_bytes = total;
    char* base = static_cast<char*>(s->d_cx_blob);
    for (const Part& p : parts) {
        e = hipMemcpy(base + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(s->d_cx_blob); delete s; return hip_fail(e, "hipMemcpy(scene)"); }
    }
    CxSceneDev& d = s->cx_dev;
    d.nodes = reinterpret_cast<const GpuNode*>(base + parts[0].off);
    d.tris = reinterpret_cast<const GpuTri*>(base + parts[1].off);
    d.aux = reinterpret_cast<const GpuTriAux*>(base + parts[2].off);
    d.mats = reinterpret_cast<const GpuMaterial*>(base + parts[3].off);
    d.tex = reinterpret_cast<const CxTexture*>(base + parts[4].off);
    d.objs = reinterpret_cast<const CxObject*>(base + parts[5].off);
    d.perlin = reinterpret_cast<const CxPerlin*>(base + parts[6].off);
    d.image_bytes = reinterpret_cast<const uint8_t*>(base + parts[7].off);
    d.tri_object = reinterpret_cast<const int32_t*>(base + parts[8].off);
    d.light_tris = reinterpret_cast<const uint32_t*>(base + parts[9].off);
    d.light_accum = reinterpret_cast<const double*>(base + parts[10].off);
    d.light_total_area = h.light_total_area;
    d.n_nodes = (uint32_t)h.nodes.size();
    d.n_tris = (uint32_t)h.tris.size();
    d.n_mats = (uint32_t)h.mats.size();
    d.n_lights = h.n_lights;
    d.n_objs = (uint32_t)h.objs.size();
    d.n_tex = (uint32_t)h.tex.size();
    d.n_media = 0;
    for (const CxObject& ob : h.objs) d.n_media += (ob.flags & kCxMedium) ? 1u : 0u;
    d.pad = 0;
    // Every composite scene also gets the tables of the render pipeline.  A scene with media uses only part of them: its
    // rays are walked over the two-level table in the reference's own order (its random draws are tied to that order),
    // so the four-wide hierarchy stays unused and ray / visibility queries go to the kernels of composite.hip.
    s->pipeline = false;
    if (!std::getenv("BDPT_COMPOSITE_SINGLE_KERNEL")) {
        try {
            err = build_instanced_pipeline(s->cx_host, s->host);
        } catch (const std::bad_alloc&) {
            err = "host allocation failed while building the pipeline tables";
        }
        if (!err.empty()) { (void)hipFree(s->d_cx_blob); delete s; return fail(BDPT_ERR_INVALID_ARGUMENT, err); }
        int rc = upload_pipeline_tables(ctx, s);
        if (rc != BDPT_OK) { (void)hipFree(s->d_cx_blob); delete s; return rc; }
        s->pipeline = h.pipeline_ok;
        s->ordered = !h.pipeline_ok;
    }
    *out = s;
    return BDPT_OK;
}

int bdpt_scene_destroy(bdpt_scene* scene) {
    if (!scene) return BDPT_OK;
    DeviceGuard guard(scene->ctx->device);
    (void)hipDeviceSynchronize();
    if (scene->d_blob) (void)hipFree(scene->d_blob);
    if (scene->d_cx_blob) (void)hipFree(scene->d_cx_blob);
    delete scene;
    return BDPT_OK;
}

int bdpt_scene_get_info(const bdpt_scene* scene, bdpt_scene_info* out) {
    if (!scene || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->composite) {
        out->n_triangles = scene->cx_dev.n_tris;
        out->n_materials = scene->cx_dev.n_mats;
        out->n_lights = scene->cx_dev.n_lights;
        out->n_nodes = scene->cx_dev.n_nodes;
        out->max_depth = scene->cx_host.tree_depth;
        out->reserved = 0;
        out->device_bytes = scene->blob_bytes + scene->cx_blob_bytes;
        return BDPT_OK;
    }
    out->n_triangles = scene->dev.n_tris;
    out->n_materials = scene->dev.n_mats;
    out->n_lights = scene->dev.n_lights;
    out->n_nodes = scene->dev.n_nodes;
    out->max_depth = scene->host.tree_depth;
    out->reserved = 0;
    out->device_bytes = scene->blob_bytes;
    return BDPT_OK;
}

int bdpt_scene_dfs_leaves(const bdpt_scene* scene, uint32_t* out, uint32_t capacity, uint32_t* count) {
    if (!scene || !count) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->composite) {
        const std::vector<int32_t>& w = scene->cx_host.world_leaves;
        *count = (uint32_t)w.size();
        if (out) for (uint32_t i = 0; i < w.size() && i < capacity; i++) out[i] = (uint32_t)w[i];
        return BDPT_OK;
    }
    const std::vector<uint32_t>& v = scene->host.dfs_leaves;
    *count = (uint32_t)v.size();
    if (out) for (uint32_t i = 0; i < v.size() && i < capacity; i++) out[i] = v[i];
    return BDPT_OK;
}

int bdpt_scene_node_boxes(const bdpt_scene* scene, double* out, uint32_t capacity_nodes) {
    if (!scene || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    const std::vector<GpuNode>& n = scene->composite ? scene->cx_host.nodes : scene->host.nodes;
    if (capacity_nodes < n.size()) return fail(BDPT_ERR_INVALID_ARGUMENT, "output buffer too small");
    for (size_t i = 0; i < n.size(); i++)
        for (int k = 0; k < 3; k++) { out[6 * i + 2 * k] = n[i].lo[k]; out[6 * i + 2 * k + 1] = n[i].hi[k]; }
    return BDPT_OK;
}

int bdpt_trace_closest_device(bdpt_context* ctx, const bdpt_scene* scene, const void* d_rays, uint64_t n, void* d_hits,
                              void* stream) {
    if (!ctx || !scene || (n && (!d_rays || !d_hits))) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->ctx != ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "scene belongs to another context");
    if (n == 0) return BDPT_OK;
    DeviceGuard guard(ctx->device);
    if (scene->composite && !(scene->pipeline && ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene))) {
        HIP_TRY(cx_launch_trace_closest(scene->cx_dev, static_cast<const double*>(d_rays), n, static_cast<double*>(d_hits), nullptr,
                                        static_cast<hipStream_t>(stream)));
        return BDPT_OK;
    }
    if (ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene))
        return trace_closest_pipeline(ctx, scene, static_cast<const double*>(d_rays), n, static_cast<double*>(d_hits),
                                      static_cast<hipStream_t>(stream));
    HIP_TRY(launch_trace_closest(scene->dev, static_cast<const double*>(d_rays), n, static_cast<double*>(d_hits), nullptr,
                                 static_cast<hipStream_t>(stream)));
    return BDPT_OK;
}

int bdpt_trace_closest(bdpt_context* ctx, const bdpt_scene* scene, const double* rays, uint64_t n, double* hits,
                       bdpt_stats* stats) {
    if (!ctx || !scene || (n && (!rays || !hits))) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->ctx != ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "scene belongs to another context");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return BDPT_OK;
    DeviceGuard guard(ctx->device);
    double *d_rays = nullptr, *d_hits = nullptr;
    HIP_TRY(hipMalloc(&d_rays, n * 8 * sizeof(double)));
    hipError_t e = hipMalloc(&d_hits, n * 11 * sizeof(double));
    if (e != hipSuccess) { (void)hipFree(d_rays); return hip_fail(e, "hipMalloc(hits)"); }
    int rc = BDPT_OK;
    do {
        if ((e = hipMemcpy(d_rays, rays, n * 8 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
        if (stats && (e = hipMemset(ctx->stats, 0, sizeof(unsigned long long) * kStatWords)) != hipSuccess) break;
        // counters have the reference's meaning, so a counting call walks the reference's own tree
        if (scene->composite && (stats || !(scene->pipeline && ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene)))) {
            if ((e = cx_launch_trace_closest(scene->cx_dev, d_rays, n, d_hits, stats ? ctx->stats : nullptr, nullptr)) != hipSuccess) break;
        } else if (!stats && ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene)) {
            rc = trace_closest_pipeline(ctx, scene, d_rays, n, d_hits, nullptr);
            if (rc != BDPT_OK) break;
        } else if ((e = launch_trace_closest(scene->dev, d_rays, n, d_hits, stats ? ctx->stats : nullptr, nullptr)) != hipSuccess) break;
        if ((e = hipMemcpy(hits, d_hits, n * 11 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if (stats) {
            unsigned long long h[kStatWords];
            if ((e = hipMemcpy(h, ctx->stats, sizeof h, hipMemcpyDeviceToHost)) != hipSuccess) break;
            read_stats(h, stats);
        }
    } while (0);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (rc == BDPT_OK && e != hipSuccess) rc = hip_fail(e, "bdpt_trace_closest");
    return rc;
}

int bdpt_visible(bdpt_context* ctx, const bdpt_scene* scene, const double* pairs, uint64_t n, uint8_t* out) {
    if (!ctx || !scene || (n && (!pairs || !out))) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->ctx != ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "scene belongs to another context");
    if (n == 0) return BDPT_OK;
    DeviceGuard guard(ctx->device);
    double* d_pairs = nullptr;
    uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_pairs, n * 6 * sizeof(double)));
    hipError_t e = hipMalloc(&d_out, n);
    if (e != hipSuccess) { (void)hipFree(d_pairs); return hip_fail(e, "hipMalloc(visible)"); }
    int rc = BDPT_OK;
    do {
        if ((e = hipMemcpy(d_pairs, pairs, n * 6 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
        if (scene->composite && !(scene->pipeline && ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene))) {
            if ((e = cx_launch_visible(scene->cx_dev, d_pairs, n, d_out, nullptr)) != hipSuccess) break;
        } else if (ctx->query_walker == BDPT_WALKER_PIPELINE && pipeline_can_walk(scene)) {
            rc = visible_pipeline(ctx, scene, d_pairs, n, d_out, nullptr);
            if (rc != BDPT_OK) break;
        } else if ((e = launch_visible(scene->dev, d_pairs, n, d_out, nullptr)) != hipSuccess) break;
        if ((e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    (void)hipFree(d_pairs);
    (void)hipFree(d_out);
    if (rc != BDPT_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "bdpt_visible");
    return BDPT_OK;
}

int bdpt_render_device(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam,
                       const bdpt_render_params* params, void* d_framebuffer, void* stream) {
    if (!ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(ctx->device);
    return render_impl(ctx, scene, cam, params, static_cast<double*>(d_framebuffer), static_cast<hipStream_t>(stream));
}

int bdpt_render_device_static(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam,
                              const bdpt_render_params* params, void* d_framebuffer, void* stream) {
    if (!ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "null context");
    if (ctx->static_ws_status != hipSuccess || !ctx->static_ws)
        return fail(BDPT_ERR_DEVICE, std::string("the static workspace of this GPU could not be located: ") + hipGetErrorString(ctx->static_ws_status));
    DeviceGuard guard(ctx->device);
    // every context and stream of this GPU shares the one workspace: this call starts after the previous one has ended
    HIP_TRY(static_scratch_begin(ctx->device, static_cast<hipStream_t>(stream)));
    const int rc = render_impl(ctx, scene, cam, params, static_cast<double*>(d_framebuffer), static_cast<hipStream_t>(stream),
                               ctx->static_ws, kStaticScratchBytes);
    const hipError_t e = static_scratch_end(ctx->device, static_cast<hipStream_t>(stream));
    if (rc != BDPT_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "static_scratch_end");
    return BDPT_OK;
}

int bdpt_render_get_stats(bdpt_context* ctx, bdpt_stats* out) {
    if (!ctx || !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->stats_valid) return fail(BDPT_ERR_INVALID_ARGUMENT, "the last render did not request counters");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[kStatWords];
    HIP_TRY(hipMemcpy(h, ctx->stats, sizeof h, hipMemcpyDeviceToHost));
    read_stats(h, out);
    return BDPT_OK;
}

int bdpt_render_get_kernel_ms(bdpt_context* ctx, float* out_ms) {
    if (!ctx || !out_ms) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->timing_valid) return fail(BDPT_ERR_INVALID_ARGUMENT, "the last render did not request BDPT_RENDER_TIME_KERNEL");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
    return BDPT_OK;
}

int bdpt_render_get_stage_ms(bdpt_context* ctx, int stage, double* total_ms, uint64_t* launches) {
    if (!ctx || !total_ms || !launches) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (stage < 0 || stage > 1) return fail(BDPT_ERR_INVALID_ARGUMENT, "unknown stage");
    if (!ctx->stage_valid) return fail(BDPT_ERR_INVALID_ARGUMENT, "the last render did not request BDPT_RENDER_TIME_STAGES");
    DeviceGuard guard(ctx->device);
    double sum = 0.0;
    uint64_t n = 0;
    for (size_t k = 0; k < ctx->stage_pairs; k++) {
        if (ctx->stage_kind[k] != stage) continue;
        HIP_TRY(hipEventSynchronize(ctx->stage_events[2 * k + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->stage_events[2 * k], ctx->stage_events[2 * k + 1]));
        sum += ms;
        n++;
    }
    *total_ms = sum;
    *launches = n;
    return BDPT_OK;
}

int bdpt_render(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam, const bdpt_render_params* params,
                double* framebuffer, bdpt_stats* stats) {
    if (!ctx || !cam || !framebuffer) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(BDPT_ERR_INVALID_ARGUMENT, "empty image");
    DeviceGuard guard(ctx->device);
    const size_t bytes = (size_t)cam->image_width * cam->image_height * 3 * sizeof(double);
    double* d_fb = nullptr;
    HIP_TRY(hipMalloc(&d_fb, bytes));
    // unowned pixels keep the caller's values: start from the caller's buffer
    hipError_t e = hipMemcpy(d_fb, framebuffer, bytes, hipMemcpyHostToDevice);
    int rc = BDPT_OK;
    if (e == hipSuccess) {
        rc = render_impl(ctx, scene, cam, params, d_fb, nullptr);
        if (rc == BDPT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(framebuffer, d_fb, bytes, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_fb);
    if (rc != BDPT_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "bdpt_render");
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        if (params->flags & (BDPT_RENDER_COUNT_STATS | BDPT_RENDER_COUNT_WORK)) return bdpt_render_get_stats(ctx, stats);
    }
    return BDPT_OK;
}

int bdpt_reduce_framebuffer(void* rccl_comm, void* d_framebuffer, uint64_t n_doubles, void* stream) {
    if (!rccl_comm || !d_framebuffer) return fail(BDPT_ERR_INVALID_ARGUMENT, "bdpt_reduce_framebuffer: null communicator or framebuffer");
    // resolved at run time: the library itself does not link RCCL, so hosts that never call this do not load it
    using AllReduce = decltype(&ncclAllReduce);
    using ErrorString = decltype(&ncclGetErrorString);
    static AllReduce all_reduce = nullptr;
    static ErrorString error_string = nullptr;
    if (!all_reduce) {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(BDPT_ERR_UNSUPPORTED, std::string("bdpt_reduce_framebuffer: cannot load RCCL: ") + dlerror());
        error_string = reinterpret_cast<ErrorString>(dlsym(lib, "ncclGetErrorString"));
        all_reduce = reinterpret_cast<AllReduce>(dlsym(lib, "ncclAllReduce"));
        if (!all_reduce) return fail(BDPT_ERR_UNSUPPORTED, "bdpt_reduce_framebuffer: ncclAllReduce not found in the RCCL library");
    }
    if (n_doubles == 0) return BDPT_OK;
    const ncclResult_t r = all_reduce(d_framebuffer, d_framebuffer, (size_t)n_doubles, ncclFloat64, ncclSum,
                                      static_cast<ncclComm_t>(rccl_comm), static_cast<hipStream_t>(stream));
    if (r != ncclSuccess)
        return fail(BDPT_ERR_DEVICE, std::string("ncclAllReduce: ") + (error_string ? error_string(r) : "error"));
    return BDPT_OK;
}

int bdpt_colors_to_rgb8(const double* fb, int32_t W, int32_t H, int32_t spp, uint8_t* out) {
    if (!fb || !out || W <= 0 || H <= 0) return fail(BDPT_ERR_INVALID_ARGUMENT, "bad argument");
    const double scale = 1.0 / (spp > 1 ? spp : 1);
    const size_t n = (size_t)W * H * 3;
    for (size_t i = 0; i < n; i++) {
        const double c = fb[i] * scale;
        const double cl = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
        const double v = 256.0 * std::sqrt(cl);
        out[i] = (v != v) ? 0 : static_cast<uint8_t>(v);
    }
    return BDPT_OK;
}

int bdpt_colors_to_rgb8_device(bdpt_context* ctx, const void* d_fb, int32_t W, int32_t H, int32_t spp, void* d_out,
                               void* stream) {
    if (!ctx || !d_fb || !d_out || W <= 0 || H <= 0) return fail(BDPT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(ctx->device);
    const double scale = 1.0 / (spp > 1 ? spp : 1);
    HIP_TRY(launch_rgb8(static_cast<const double*>(d_fb), (uint64_t)W * H * 3, scale, static_cast<uint8_t*>(d_out),
                        static_cast<hipStream_t>(stream)));
    return BDPT_OK;
}

static int run_unary(bdpt_context* ctx, int kind, int op, const double* a, const double* b, uint64_t n, double* o1,
                     double* o2, uint64_t seed, uint64_t pixel, uint64_t sample) {
    if (!ctx) return fail(BDPT_ERR_INVALID_ARGUMENT, "null context");
    if (n == 0) return BDPT_OK;
    DeviceGuard guard(ctx->device);
    double *d_a = nullptr, *d_b = nullptr, *d_o1 = nullptr, *d_o2 = nullptr;
    hipError_t e = hipSuccess;
    const size_t bytes = n * sizeof(double);
    do {
        if ((e = hipMalloc(&d_a, bytes)) != hipSuccess) break;
        if ((e = hipMalloc(&d_b, bytes)) != hipSuccess) break;
        if ((e = hipMalloc(&d_o1, bytes)) != hipSuccess) break;
        if ((e = hipMalloc(&d_o2, bytes)) != hipSuccess) break;
        if (a && (e = hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice)) != hipSuccess) break;
        if (b && (e = hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice)) != hipSuccess) break;
        if (kind == 0) e = launch_sincos(d_a, n, d_o1, d_o2, nullptr);
        else if (kind == 1) e = launch_random(seed, pixel, sample, n, d_o1, nullptr);
        else e = launch_arith(op, d_a, d_b, n, d_o1, nullptr);
        if (e != hipSuccess) break;
        if (o1 && (e = hipMemcpy(o1, d_o1, bytes, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if (o2 && (e = hipMemcpy(o2, d_o2, bytes, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o1); (void)hipFree(d_o2);
    if (e != hipSuccess) return hip_fail(e, "device test hook");
    return BDPT_OK;
}

int bdpt_device_sincos(bdpt_context* ctx, const double* x, uint64_t n, double* sin_out, double* cos_out) {
    if (n && (!x || !sin_out || !cos_out)) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    return run_unary(ctx, 0, 0, x, nullptr, n, sin_out, cos_out, 0, 0, 0);
}

int bdpt_device_random_doubles(bdpt_context* ctx, uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* out) {
    if (n && !out) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    return run_unary(ctx, 1, 0, nullptr, nullptr, n, out, nullptr, seed, pixel, sample);
}

int bdpt_device_arith(bdpt_context* ctx, int op, const double* a, const double* b, uint64_t n, double* out) {
    if (n && (!a || !b || !out)) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    return run_unary(ctx, 2, op, a, b, n, out, nullptr, 0, 0, 0);
}

int bdpt_debug_tie_counts(bdpt_context* ctx, uint64_t* out8, int reset) {
    if (!ctx || !out8) return fail(BDPT_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[8];
    const hipError_t e = wf_tie_counts(c, reset != 0);
    if (e == hipErrorNotSupported) return fail(BDPT_ERR_UNSUPPORTED, "this build does not cross-check far-end ties (build the `tiecheck` variant)");
    if (e != hipSuccess) return hip_fail(e, "wf_tie_counts");
    for (int k = 0; k < 8; k++) out8[k] = c[k];
    return BDPT_OK;
}

}  // extern "C"
