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
    int32_t pad;      // leaf-ordered copy: original triangle id
};
static_assert(sizeof(GpuTri) == 80, "GpuTri must stay 80 bytes");

// Closest-hit queries do not depend on the shape of the tree (the answer is the nearest accepted
// triangle, ties resolved by the reference's leaf order), so the wavefront kernels use a separate hierarchy: a binned
// surface-area-heuristic tree collapsed to four children per record.  128-byte records, child boxes in
// FP32 rounded OUTWARD (a box here always contains the padded FP64 box of everything below it, so
// culling is conservative), leaves of up to four triangles.  Child reference: >= 0 index of an interior
// record; kQuadEmpty for an unused slot; otherwise a leaf, ~ref = (first << 2) | (count - 1).
// The triangle tests stay the reference's FP64 arithmetic.
struct GpuQuadNode {
    float box[4][6];      // per child: xmin xmax ymin ymax zmin zmax (min/max of an axis adjacent: one packed FP32 operand)
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(GpuQuadNode) == 128, "GpuQuadNode must stay 128 bytes");
constexpr int32_t kQuadEmpty = (int32_t)0x80000000;

// Records [0, kLdsTopQuadNodes) are the levels nearest the root in breadth-first order; the traversal kernels
// copy a prefix of them into LDS, so only the deeper visits go to L1/L2.
constexpr uint32_t kLdsTopQuadNodes = 1024;

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

// What shading a hit needs, in one record per triangle in leaf order (the position a traversal reports): the kernel
// that shades a hit then makes ONE dependent load where the separate tables (leaf position -> triangle -> normal and
// material id -> material) make three in a row.
struct GpuShadeRec {
    double normal[3];     // unit normal (GpuTriAux::normal)
    double rgb[3];        // GpuMaterial::rgb
    double param;         // fuzz of a metal, refraction index of a dielectric
    uint32_t tri;         // triangle id
    uint32_t mat;         // material id | type << 30
};
static_assert(sizeof(GpuShadeRec) == 64, "GpuShadeRec must stay 64 bytes");

struct GpuCamera {
    double center[3];
    double pixel00[3];
    double du[3];
    double dv[3];
    double background[3];
    double recip_sqrt_spp;
    int32_t width, height, sqrt_spp, max_depth;
    double disk_u[3], disk_v[3];    // defocus disk, used when defocus_angle > 0
    double defocus_angle;
};

constexpr uint32_t kCxRotate = 1u, kCxTranslate = 2u, kCxMedium = 4u;

// One object of the world's list (src/objects/hittable.h:46-157, src/materials/volumes/constant_medium.h).
struct CxObject {
    double sin_theta, cos_theta;   // rotate_y::rotate_y, hittable.h:79-81
    double offset[3];              // translate
    double neg_inv_density;        // constant_medium.h:11
    uint32_t root;                 // first record of the object's own bvh_node tree in the node table
    uint32_t flags;                // kCx*
    int32_t phase_material;
    uint32_t pad;
};

// src/materials/textures/texture.h
struct CxTexture {
    double rgb[3];
    double scale;                  // noise: scale; checker: inv_scale = 1.0 / scale (texture.h:31)
    int32_t type;                  // BDPT_TEX_*
    int32_t a, b;                  // checker: even / odd texture; image: width / height
    uint32_t data;                 // noise: index of its tables in `perlin`; image: first byte in `image_bytes`
};

// What a perlin object holds (perlin.h:53-57)
struct CxPerlin {
    double randvec[256][3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
};

struct SceneDev {
    const GpuNode* nodes;
    const GpuTri* tris;
    const uint32_t* node_parent;    // parent of every node of `nodes` (kEndOfTree for the root)
    const GpuTri* stris;            // triangles in leaf order; `pad` holds the original triangle id
    const uint32_t* sleaf;          // per entry of `stris`: the node of `nodes` that holds the triangle; bit 31 = ref_nested
    const uint32_t* tri_leaf;       // the same per triangle of `tris`
    const int32_t* tri_graze;       // per triangle of `tris`: its grazing threshold (what `xtris` carries in `material`)
    const GpuQuadNode* xnodes;      // closest-hit hierarchy (surface-area heuristic, four children, FP32 boxes)
    const GpuTri* xtris;            // its triangles in its own leaf order; `pad` = position in `stris`; `material` holds
                                    // the high 32 bits of a double instead: the triangle's grazing threshold (wavefront.hip)
    uint32_t n_xnodes;
    const GpuTriAux* aux;
    const GpuMaterial* mats;
    const GpuShadeRec* shade;       // per entry of `stris`
    const uint32_t* tri_pos;        // per triangle of `tris`: its position in `stris`
    const uint32_t* tri_xpos;       // position in `xtris` of what Vertex::tri names: a triangle of `tris`, or - in a scene with
                                    // instances - an entry of `stris`
    const uint32_t* light_tris;     // the light collection (triangle ids, insertion order)
    const double* light_accum;      // running sums of the areas exactly as triangle.h:212-219 accumulates them
    double light_total_area;
    double coord_bound;             // largest |coordinate| of any box of the closest-hit hierarchy
    double bary_margin;             // a hit closer than this (in barycentric units) to its triangle's outline is re-checked
    uint32_t n_nodes, n_tris, n_mats, n_lights;
    uint32_t ref_nested;            // 1: every box of `nodes` contains the boxes of the nodes below it
    uint32_t n_objs;                // entries of `objs`
    // Scenes with instances (translate / rotate_y objects over meshes of their own) on the pipeline.  `nodes` and
    // `node_parent` then hold the two-level table of such a scene (the world's tree, whose leaves are triangles or
    // objects, followed by one tree per mesh; every tree threaded by itself), `tris` of a mesh are in the mesh's own
    // coordinates, an entry of `stris` / `xtris` is a (object, triangle) pair, `sleaf` the node of its own tree that
    // holds the triangle, and `tri_pos` of a mesh triangle its rank inside the mesh (position = obj_pos0 + rank).
    const CxObject* objs;           // null: no instances
    const int32_t* xobj;            // per entry of `xtris`: the object whose ray transform its triangle test needs, -1: none
    const int32_t* pos_obj;         // the same per entry of `stris`
    const uint32_t* obj_leaf;       // per object: the node of the world's tree that holds it (| bit 31 = ref_nested)
    const uint32_t* obj_pos0;       // per object: position in `stris` of the first triangle of its mesh
    // Scenes with textured materials on the pipeline (they run the kernels of scenes with instances): GpuMaterial::pad
    // of `mats` is then the texture index + 1 of the material, 0 for a solid colour.
    const CxTexture* tex;           // null: every material is a solid colour
    const CxPerlin* perlin;
    const uint8_t* image_bytes;
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
