// Device-resident tables of a scene and the per-launch parameter blocks (plain data shared by
// the host builder and the kernels).
#ifndef BDPT_TYPES_H
#define BDPT_TYPES_H

#include <stdint.h>

namespace bdpt {

// One bvh_node of the reference tree (src/acceleration/bvh.h:11-66) in threaded form, 64 bytes =
// one half cache line.  The reference always visits left before right, so traversal is a walk
// over explicit links and needs no stack:
//   box hit  -> test tri0 / tri1 if this node's children are triangles, then go to `next`
//   box miss -> go to `skip` (the node after this node's whole subtree)
// A node's children are either two bvh_nodes or two triangles (span 2; span 1 stores the same
// triangle twice, bvh.h:35-36).  kEndOfTree terminates the walk.
struct GpuNode {
    double lo[3];
    double hi[3];
    uint32_t skip;
    uint32_t next;
    int32_t tri0;     // -1 for an interior node
    int32_t tri1;
};
static_assert(sizeof(GpuNode) == 64, "GpuNode must stay 64 bytes");
constexpr uint32_t kEndOfTree = 0xFFFFFFFFu;

// What triangle::hit reads (src/objects/primatives/triangle.h:41-74), 80 bytes.
struct GpuTri {
    double v0[3];
    double e1[3];
    double e2[3];
    int32_t material;
    int32_t pad;
};
static_assert(sizeof(GpuTri) == 80, "GpuTri must stay 80 bytes");

// Read once per accepted hit / light sample: unit normal and area (triangle.h:26-29).
struct GpuTriAux {
    double normal[3];
    double area;
};

struct GpuMaterial {
    double rgb[3];
    double fuzz;
    double ior;
    int32_t type;
    int32_t pad;
};

struct GpuCamera {
    double center[3];
    double pixel00[3];
    double du[3];
    double dv[3];
    double background[3];
    double recip_sqrt_spp;
    int32_t width, height, sqrt_spp, max_depth;
};

struct SceneDev {
    const GpuNode* nodes;
    const GpuTri* tris;
    const GpuTriAux* aux;
    const GpuMaterial* mats;
    const uint32_t* light_tris;     // the light collection (triangle ids, insertion order)
    const double* light_accum;      // running sums of the areas exactly as triangle.h:212-219 accumulates them
    double light_total_area;
    uint32_t n_nodes, n_tris, n_mats, n_lights;
};

struct RenderLaunch {
    uint64_t seed;
    const int32_t* rows;            // owned rows (image row index), n_rows entries
    int32_t n_rows;
    int32_t x0, x1;                 // owned column range
    int32_t pixels_per_group;       // pixels one workgroup handles per work unit
    int32_t n_units;                // number of work units
    uint32_t* unit_counter;         // dynamic work distribution
    double* fb;                     // W*H*3 sums
    double* vertex_ws;              // path-vertex workspace
    unsigned long long* stats;      // 10 counters (bdpt_stats order) or nullptr
    uint64_t ws_stride;             // threads in the grid (stride between vertex slots)
};

}  // namespace bdpt
#endif
