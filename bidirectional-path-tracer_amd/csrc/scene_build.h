// Host-side scene preparation: per-triangle precompute, the reference-equivalent BVH build and
// the flattening into the device tables of bdpt_types.h.
#ifndef BDPT_SCENE_BUILD_H
#define BDPT_SCENE_BUILD_H

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/bdpt_amd.h"
#include "bdpt_types.h"

namespace bdpt {

struct HostScene {
    std::vector<GpuNode> nodes;          // pre-order (root first); links already threaded
    std::vector<GpuTri> tris;
    std::vector<uint32_t> node_parent;   // parent of every bvh_node (kEndOfTree for the root)
    std::vector<uint32_t> sleaf;         // per entry of `stris`: the bvh_node that holds the triangle; bit 31 = ref_nested
    std::vector<GpuTri> stris;           // triangles in leaf order (pad = original id)
    std::vector<GpuShadeRec> shade;      // per entry of `stris`: what shading a hit on it needs
    std::vector<uint32_t> tri_pos;       // per triangle: its position in `stris`
    std::vector<uint32_t> tri_xpos;      // position in `xtris`, per triangle id (per position in `stris` when vertex_names_entry)
    bool vertex_names_entry = false;     // the scene runs the kernels of scenes with instances: a path vertex names its entry of `stris`
    std::vector<uint32_t> tri_leaf;      // per triangle: the bvh_node that holds it
    std::vector<int32_t> tri_graze;      // per triangle: grazing threshold (high word of a double, see scene_build.cpp)
    std::vector<GpuQuadNode> xnodes;     // closest-hit hierarchy (binned SAH, four children per record)
    std::vector<GpuTri> xtris;           // its triangles in leaf order (pad = position in stris)
    double coord_bound = 0.0;            // largest |coordinate| of its boxes
    double bary_margin = 1e-6;           // see SceneDev::bary_margin
    bool ref_nested = false;             // every bvh_node box contains the boxes of the nodes below it
    uint32_t sah_stack = 0;              // most entries its traversal stack can hold at once
    bool has_lambertian = false;         // some triangle has a lambertian material (the only connectable kind of vertex)
    std::vector<GpuTriAux> aux;
    std::vector<GpuMaterial> mats;
    std::vector<uint32_t> light_tris;
    std::vector<double> light_accum;
    double light_total_area = 0.0;
    std::vector<uint32_t> dfs_leaves;    // triangle ids in the reference's leaf visiting order
    uint32_t n_lights = 0;
    // scenes with instances (see SceneDev): empty otherwise
    std::vector<CxObject> objs;
    std::vector<int32_t> xobj, pos_obj;
    std::vector<uint32_t> obj_leaf, obj_pos0;
    uint32_t tree_depth = 0;
};

// Returns an empty string on success, an error message otherwise.
std::string build_host_scene(const double* tri_vertices, const int32_t* tri_material, uint32_t n_triangles,
                             const bdpt_material* materials, uint32_t n_materials, const uint32_t* light_triangles,
                             uint32_t n_lights, HostScene& out);

// camera::initialize, src/camera.h:160-197
std::string camera_initialize(const bdpt_camera_public& cam, bdpt_camera_state& out);

}  // namespace bdpt
#endif
