// Composite scenes: a world whose hittable_list holds, beside bare triangles, meshes of their own that are rotated,
// translated or turned into a volume of constant density, with textured materials.  Plain data shared by the host
// builder (scene_build.cpp) and the kernels (composite.hip), and the launchers of those kernels.
#ifndef BDPT_COMPOSITE_H
#define BDPT_COMPOSITE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/bdpt_amd.h"
#include "bdpt_types.h"

namespace bdpt {

// In the node table a leaf reference >= 0 is a triangle id and a reference <= -2 the object -(ref + 2);
// -1 marks an interior record, as in GpuNode.  Record 0 is the root of the world's tree; every tree of the table is
// threaded by itself and ends in kEndOfTree.
struct CxSceneDev {
    const GpuNode* nodes;
    const GpuTri* tris;
    const GpuTriAux* aux;
    const GpuMaterial* mats;       // `pad` = texture index + 1, 0 = the solid colour rgb
    const CxTexture* tex;
    const CxObject* objs;
    const CxPerlin* perlin;
    const uint8_t* image_bytes;
    const int32_t* tri_object;     // per triangle: the object it belongs to, -1 for a triangle of the world's list
    const uint32_t* light_tris;
    const double* light_accum;
    double light_total_area;
    uint32_t n_nodes, n_tris, n_mats, n_lights, n_objs, n_tex;
    uint32_t n_media;              // objects that are a constant_medium: with none, a visibility query may stop at the first hit
    uint32_t pad;
};

struct CxHostScene {
    std::vector<GpuNode> nodes;
    std::vector<GpuTri> tris;
    std::vector<GpuTriAux> aux;
    std::vector<GpuMaterial> mats;
    std::vector<CxTexture> tex;
    std::vector<CxObject> objs;
    std::vector<CxPerlin> perlin;
    std::vector<uint8_t> image_bytes;
    std::vector<int32_t> tri_object;
    std::vector<uint32_t> light_tris;
    std::vector<double> light_accum;
    double light_total_area = 0.0;
    uint32_t n_lights = 0;               // size of the light collection (the tables hold one dummy entry when it is empty)
    std::vector<uint32_t> node_parent;   // parent of every record of `nodes` inside its own tree (kEndOfTree for a root)
    std::vector<uint32_t> first_triangle, mesh_triangles;   // per object: its mesh
    bool pipeline_ok = false;            // no medium: the pipeline's four-wide hierarchy can answer for the scene
    bool textured = false;               // some material carries a texture
    std::vector<int32_t> world_leaves;   // leaf references of the world's tree in the reference's visiting order
    uint32_t n_world_nodes = 0;          // records [0, n_world_nodes) are the world's tree
    uint32_t tree_depth = 0;
};

// Host build.  sin_theta / cos_theta: per object, sin and cos of degrees_to_radians(rotate_y_degrees)
// (hittable.h:79-81), evaluated by the caller with the device's deterministic routines.
std::string build_composite_scene(const double* tri_vertices, const int32_t* tri_material, uint32_t n_triangles,
                                  const bdpt_material* materials, uint32_t n_materials, const bdpt_texture* textures,
                                  uint32_t n_textures, const bdpt_object* objects, uint32_t n_objects,
                                  const double* sin_theta, const double* cos_theta, const uint32_t* light_triangles,
                                  uint32_t n_lights, CxHostScene& out);

// Tables of the render pipeline for a composite scene with pipeline_ok (scene_build.h: HostScene).
struct HostScene;
std::string build_instanced_pipeline(const CxHostScene& cx, HostScene& out);

struct CxRenderLaunch {
    uint64_t seed;
    const int32_t* rows;
    int32_t n_rows;
    int32_t x0, x1;
    int32_t pixels_per_group;
    int32_t n_units;
    uint32_t* unit_counter;
    double* fb;
    double* vertex_ws;
    unsigned long long* stats;      // 11 counters (bdpt_stats order) or nullptr
    uint64_t ws_stride;
};
constexpr int kCxVtxDoubles = 13;   // p[3] n[3] T[3] f[3] + flags

// unidirectional: camera::path_trace_color instead of bidirectional_color
hipError_t cx_render_blocks_per_cu(bool stats, bool unidirectional, int* out);
hipError_t cx_launch_render(const CxSceneDev& sc, const GpuCamera& cam, const CxRenderLaunch& rl, int grid_blocks, bool stats,
                            bool unidirectional, hipStream_t stream);
hipError_t cx_launch_trace_closest(const CxSceneDev& sc, const double* d_rays, uint64_t n, double* d_hits,
                                   unsigned long long* d_stats, hipStream_t stream);
hipError_t cx_launch_visible(const CxSceneDev& sc, const double* d_pairs, uint64_t n, uint8_t* d_out, hipStream_t stream);

}  // namespace bdpt
#endif
