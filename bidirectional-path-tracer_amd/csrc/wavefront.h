// Wavefront (streaming) formulation of the radiance-estimation path: per-sample state lives in
// HBM as structure-of-arrays, every subpath stage is its own kernel over a compacted queue of
// live samples, and ray traversal runs in persistent waves that refill finished lanes.
#ifndef BDPT_WAVEFRONT_H
#define BDPT_WAVEFRONT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bdpt_types.h"

namespace bdpt {

// Per-batch state, one entry per sample slot (capacity `ns`); slot = s * pixels_in_batch + p, so
// neighbouring lanes hold the same sample index of neighbouring pixels.
// The state of a path that is under way travels with its place in the queue, not with its sample slot: entry k of the
// live queue has its ray, throughput and stream position at index k of these arrays, the traversal writes the hit to
// index k, and the shading kernel writes the state of a path that goes on to the path's place in the NEXT queue (the
// `_next` arrays).  Every such access is a contiguous piece of memory per wave; indexed by slot - the live samples
// thin out bounce by bounce - each was a gather or scatter that kept the texture addresser busy for most of the
// shading kernels' time.
struct WfBuffers {
    double* ray;          // [6][ns]  by queue entry: origin, direction of the ray to trace / that was traced
    double* ray_next;     // [6][ns]  ... of the next queue
    double* thr;          // [3][ns]  by queue entry: path throughput
    double* thr_next;
    uint32_t* ctrq;       // [ns]     by queue entry: random numbers drawn so far by the sample
    uint32_t* ctrq_next;
    double* hit;          // [ns]     by queue entry: t of the closest hit
    const double* tmax;   // [ns]     upper end of the interval of a closest-hit ray, or null: +infinity (trace_path, camera.h:336)
    int32_t* hit_tri;     // [ns]     by queue entry: position of the hit triangle in the leaf-ordered table, or -1
    uint32_t* ctr;        // [ns]     by slot: random numbers drawn by the sample when its subpath ended
    double* result;       // [3][ns]  radiance of the sample
    uint32_t* counts;     // [ns]     n_eye | n_light << 12 | flags << 24
    double* vtx;          // [ns / kVtxGroup][2*depth][10][kVtxGroup]  stored (non-delta) path vertices: eye 0.., light depth..
    uint32_t* queue[2];   // [ns]     live-sample queues (ping-pong)
    uint32_t* conn_queue; // [ns]     scenes with media only: samples that have both eye and light vertices
    // Connection stage (camera.h:316-320).  The pairs (eye vertex a, light vertex b) of a range of samples that need a
    // visibility ray are queue entries; the entries of ONE sample are contiguous and in the reference's pair order (a
    // outer, b inner), so the sample's sum is formed by walking them.
    double* shadow;       // [7][shadow_cap]  shadow rays: origin, direction, max_t (negative: an unused entry)
    double* tex_rgb;      // [3][ns] by queue entry: value of the hit material's texture (k_tex_eval -> k_shade); null without textures
    uint8_t* sh_vis;      // [shadow_cap]  1: the entry's shadow ray came back clear (cleared when the entry is made)
    uint32_t* sh_pair;    // [shadow_cap]  a << 12 | b of the entry's pair
    uint32_t* shadow_first;   // [shadow_cap]  position in SceneDev::xtris of the triangle a shadow ray's walk tests first (the
                              // one its far end lies on), or null: every walk starts at the root
    double* contrib;      // [3][shadow_cap]  what the entry's pair adds if its ray is clear (scenes whose connection kernel
                          // tests the far end's own triangle), or null: clear pairs are evaluated again when they are added
    uint32_t* ent_first;  // [ns]     first queue entry of the sample's pairs in the current pass
    uint32_t* ent_count;  // [ns]     number of its entries
    unsigned long long* tile_pairs;   // [ns / kConnTile + 1]  candidate pairs of each tile of slots
    uint32_t* pass_bounds;    // [passes + 1]  the tiles of pass p: pass_bounds[p] .. pass_bounds[p + 1]
    uint32_t* counters;   // queue sizes and work-distribution cursors, zeroed once per batch
    const float* ray_len_ptr;  // upper bound of |direction| over the closest-hit rays: *ray_len_ptr, or ray_len when null
    float ray_len;
    uint64_t ns;
    uint64_t shadow_cap;
    uint32_t vtx_words;       // words of `vtx` per slot (records x 10, or x 13 in a scene with textures)
};

// slots per group of the vertex and contribution arrays (see slot_store in wavefront.hip)
constexpr uint32_t kVtxGroup = 64;

// wb.hit_tri value of a ray whose closest hit the traversal kernel leaves to the walk over the reference's own tree
constexpr int32_t kHitRedo = -2;
// A queue entry without a path.  The shading kernels give every path that MAY go on its place in the next queue before
// they shade it (one atomic per tile of entries); the rare path that then ends after all (a degenerate surface) leaves
// this in place of its slot, and a direction of zero, which the traversal does not walk.
constexpr uint32_t kDeadSlot = 0xFFFFFFFFu;
// scenes with media (k_walk_ordered): wb.hit_tri value kHitMedium0 - k of a ray that ends inside the medium of object k
constexpr int32_t kHitMedium0 = -3;

// entries of the closest-hit walk's stack (up to three pushes per level of the four-wide tree)
constexpr int kQuadStackEntries = 64;

// the eye subpath stored at least one vertex that can form a non-zero connection (non-delta, not on an emitter)
constexpr uint32_t kCntEyeConnectable = 1u << 24;

struct WfBatch {
    uint64_t seed;
    const int32_t* rows;      // owned rows
    int32_t row_w, x0;        // owned column range
    int64_t first_pixel;      // ordinal (in owned-pixel order) of the batch's first pixel
    int32_t n_pixels;         // pixels in this batch
    int32_t pixels_stride;    // sample_stride 1: slot = s * pixels_stride + p (neighbouring slots: one stratum of neighbouring pixels)
    int32_t sample_stride;    // otherwise = spp: slot = p * spp + s (neighbouring slots: the strata of one pixel)
    int64_t n_owned;
};

// slots per tile of the connection stage's planning (k_conn_count / k_conn_plan): one group of vertex records
constexpr uint32_t kConnTile = 64;

struct WfLaunchCfg {
    int persistent_blocks;    // grid of the persistent traversal kernels
    int stream_blocks;        // grid of the grid-stride kernels
    int redo_blocks;          // grid of k_redo (the rays of a launch that met an exact tie: few)
    bool lds_scene;
    uint32_t n_entries;       // entries of the traversal hierarchy's triangle table (a scene with instances: (object, triangle) pairs)
    bool instanced;       // the scene holds instances: traversal and shading kernels with the ray transform in front of the triangle test
    bool ordered;         // the scene holds a constant_medium: rays are traced in the reference's own order (k_walk_ordered,
                          // k_conn_ordered) and the shading kernels know the isotropic phase function (implies `textured`)
    size_t walk_lds_bytes;    // a scene with media whose tables the ordered walks copy into LDS first: their size (0: read from global memory)
    bool tex_pass;        // some material really carries a texture: k_tex_eval runs in front of every k_shade
    bool textured;        // some material carries a texture (implies `instanced`: those kernels, with vertex records of kVtxDoublesTextured words)
    // connection stage: k_connect stages the vertex records of a run of samples in LDS (conn_rec_budget records per wave)
    // and takes the shadow queue conn_chunk entries at a time; every wave may leave one such piece partly unused
    uint32_t conn_blocks;     // workgroups of k_connect
    uint32_t conn_waves;      // waves per workgroup (4; fewer when a sample's 2 x depth records need a larger staging area)
    uint32_t conn_rec_budget; // vertex records a wave stages at a time (>= 2 x depth)
    uint32_t conn_chunk;      // queue entries a wave takes at a time (>= the pairs of one sample, depth x depth)
    uint32_t conn_passes;     // passes the host launches; pass p handles the tiles pass_bounds[p] .. pass_bounds[p + 1]
    unsigned long long conn_pass_pairs;   // most candidate pairs the planner gives one pass (what the queue holds for certain)
    int sort_shift;           // k_shade orders the paths of a tile that go on by (leaf position >> sort_shift), at most 64 cells; -1: not at all
    bool own_test;        // k_connect tests the light vertex's own triangle (from tables in LDS) and keeps the pairs' contributions
    size_t lds_bytes;
    size_t ref_tree_bytes;    // LDS copy of the reference's tree (nodes + parents) for scenes staged in LDS
    size_t conn_lds_bytes;    // dynamic LDS of k_connect: the waves' staging areas, then (own_test) that copy, triangles,
                              // grazing thresholds, leaf nodes
    int refill_threshold;     // a wave fetches new rays when at least this many of its lanes are idle
    int leaf_threshold;       // lanes of a wave that must stand at a leaf before the wave turns to triangle tests
    int leaf_inner;           // ... and it stays with the triangles of those leaves while this many lanes have one left
    uint32_t chunk;           // ray indices a wave takes from the global cursor at a time
    uint32_t top_nodes;       // 128-byte records of the top tree levels each such workgroup keeps in LDS
    int lds_stack;            // traversal stack entries per lane kept in LDS (the rest of the stack is scratch)
};

// counts of the work done, accumulated by the kernels when `stats` is non-null (bdpt_stats order)
hipError_t wf_gen_eye(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const WfLaunchCfg& cfg,
                      hipStream_t st);
// closest-hit traversal of a queue; wf_redo, called next on the same stream, launches k_redo over the rays it left
// open (two calls so that a caller can time the traversal launch by itself).  redo_list: room for one entry per ray of
// the launch (the queue the following shading launch will fill is free until then); redo_count: zero when called
hipError_t wf_extend(const SceneDev& sc, const WfBuffers& wb, const uint32_t* count_ptr,
                     uint32_t count_imm, uint32_t* cursor, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st,
                     uint32_t* redo_list, uint32_t* redo_count);
hipError_t wf_redo(const SceneDev& sc, const WfBuffers& wb, const WfLaunchCfg& cfg, hipStream_t st, uint32_t* redo_list, uint32_t* redo_count);
// values of the textures at the hits of the entries of a queue (before wf_shade reads them)
hipError_t wf_tex_eval(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue_in,
                       const uint32_t* count_in_ptr, uint32_t count_in_imm, const WfLaunchCfg& cfg, hipStream_t st);
hipError_t wf_shade(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, bool eye,
                    const uint32_t* queue_in, const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out,
                    uint32_t* count_out, bool last_bounce, bool store_vertices, bool keep_draw_counts, const WfLaunchCfg& cfg,
                    unsigned long long* stats, hipStream_t st);
// one level of camera::path_trace_color (the unidirectional estimator): records into wb.vtx slot `level`
hipError_t wf_shade_unidirectional(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue_in,
                                   const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out, uint32_t* count_out, int level,
                                   bool last_level, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
hipError_t wf_gen_light(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, uint32_t* queue_out,
                        uint32_t* count_out, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
// connection stage, in this order: candidate pairs per tile of slots; the passes' tile ranges; then per pass the
// shadow rays of its pairs (wf_connect), their traversal (wf_occlude with the pass's queue size) and the sums (wf_conn_add)
hipError_t wf_conn_count(const WfBuffers& wb, uint32_t n_slots, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
hipError_t wf_conn_plan(const WfBuffers& wb, uint32_t n_slots, const WfLaunchCfg& cfg, hipStream_t st);
hipError_t wf_connect(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, uint32_t n_slots, int pass, uint32_t* shadow_count,
                      const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
hipError_t wf_conn_add(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, uint32_t n_slots, int pass, const WfLaunchCfg& cfg,
                       hipStream_t st);
// scenes with media: the samples that have vertices on both sides (k_conn_ordered's queue)
hipError_t wf_conn_prepare(const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, uint32_t* conn_count,
                           const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
// scenes with media: closest-hit walk in the reference's order for the rays of a queue, and the whole connection stage
hipError_t wf_walk_ordered(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue,
                           const uint32_t* count_ptr, uint32_t count_imm, const WfLaunchCfg& cfg, hipStream_t st);
hipError_t wf_conn_ordered(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* conn_count,
                           const WfLaunchCfg& cfg, hipStream_t st);
// number of queue entries: *shadow_count, or count_imm when shadow_count is null
hipError_t wf_occlude(const SceneDev& sc, const WfBuffers& wb, const uint32_t* shadow_count, uint32_t count_imm, uint32_t* cursor,
                      const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st);
hipError_t wf_resolve(const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, double* fb, const WfLaunchCfg& cfg,
                      hipStream_t st);

// Cross-check build only (-DBDPT_TIE_CHECK): the eight counters of the comparison between the in-place settling of
// far-end ties and the reference's own box walk (see g_tie_counts in wavefront.hip); hipErrorNotSupported otherwise.
hipError_t wf_tie_counts(unsigned long long* out8, bool reset);

// sets the LDS limits of the traversal and connection kernels for this launch shape and reports how many traversal
// workgroups fit a CU
hipError_t wf_prepare_traverse(const WfLaunchCfg& cfg, int* blocks_per_cu);

// Table of the image rows a render call owns: rows j of [y0, y1) with (j / shard_rows) % shard_count == shard_index,
// in increasing order.  wf_owned_rows_before(y, ...) is the number of owned rows below y (host arithmetic); the
// kernel writes row j to rows[wf_owned_rows_before(j) - wf_owned_rows_before(y0)].
int64_t wf_owned_rows_before(int64_t y, int shard_rows, int shard_count, int shard_index);
hipError_t wf_fill_rows(int32_t* rows, int y0, int y1, int shard_rows, int shard_count, int shard_index, hipStream_t st);

// Ray batches through the traversal kernels of the pipeline (bvh_node::hit / camera::visible for arrays of queries).
// rays: n x 8 doubles (origin, direction, tmin, tmax) -> wb.ray / wb.tmax;  hits: n x 11 doubles (hit_record).
// also raises *ray_len_out (zeroed by the caller) to the largest |direction|
hipError_t wf_rays_in(const double* rays, uint32_t n, const WfBuffers& wb, double* tmax_out, float* ray_len_out, hipStream_t st);
hipError_t wf_hits_out(const SceneDev& sc, const double* rays, uint32_t n, const WfBuffers& wb, double* hits, hipStream_t st);
// pairs: n x 6 doubles (a, b) -> shadow-ray queue entries 0..n-1 (pair k answers in wb.sh_vis[k]);  out: n bytes
hipError_t wf_pairs_in(const double* pairs, uint32_t n, const WfBuffers& wb, hipStream_t st);
hipError_t wf_visible_out(const WfBuffers& wb, uint32_t n, uint8_t* out, hipStream_t st);

}  // namespace bdpt
#endif
