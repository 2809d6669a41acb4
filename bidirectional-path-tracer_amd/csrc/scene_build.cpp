// Host-side scene preparation (plain C++, no device code).  Built with -ffp-contract=off: the
// triangle precompute and the box unions must round like the reference's.
#include "scene_build.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace bdpt {
namespace {

struct Box3 { double lo[3], hi[3]; };

struct BuildTri {
    Box3 box;      // aabb(point, point) of the triangle, NOT padded (src/acceleration/aabb.h:18-25)
};

struct BuildNode {
    Box3 box;
    int32_t left, right;   // >= 0: node index, < 0: ~triangle id
};

// interval::expand via aabb::pad_to_minimums, src/acceleration/aabb.h:81-88, src/core/interval.h:36-39
inline void pad_axis(double& lo, double& hi) {
    const double delta = 0.0001;
    if (hi - lo < delta) {
        const double padding = delta / 2;
        lo = lo - padding;
        hi = hi + padding;
    }
}

// aabb(const aabb&, const aabb&), aabb.h:27-33 + interval(a, b), interval.h:12-16
inline Box3 unite(const Box3& a, const Box3& b) {
    Box3 r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = a.lo[k] <= b.lo[k] ? a.lo[k] : b.lo[k];
        r.hi[k] = a.hi[k] >= b.hi[k] ? a.hi[k] : b.hi[k];
        pad_axis(r.lo[k], r.hi[k]);
    }
    return r;
}

inline int longest_axis(const Box3& b) {   // aabb.h:68-75
    const double sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

struct Builder {
    const std::vector<BuildTri>& tris;
    std::vector<BuildNode> nodes;
    std::vector<uint32_t> order;           // the hittable_list being sorted in place (bvh.h:41)

    explicit Builder(const std::vector<BuildTri>& t) : tris(t) {}

    // bvh_node(objects, start, end), src/acceleration/bvh.h:20-48.  Nodes are emitted in pre-order.
    int32_t build(size_t start, size_t end) {
        const int32_t idx = (int32_t)nodes.size();
        nodes.push_back(BuildNode());

        const double inf = std::numeric_limits<double>::infinity();
        Box3 box;                          // aabb::empty (aabb.h:92): padding leaves the infinities as they are
        for (int k = 0; k < 3; k++) { box.lo[k] = inf; box.hi[k] = -inf; }
        for (size_t i = start; i < end; i++) box = unite(box, tris[order[i]].box);

        const int axis = longest_axis(box);
        const size_t span = end - start;
        int32_t left, right;
        if (span == 1) {
            left = right = ~(int32_t)order[start];
        } else if (span == 2) {
            left = ~(int32_t)order[start];
            right = ~(int32_t)order[start + 1];
        } else {
            // Same comparator (strict < on the triangle boxes' minimum along the axis, bvh.h:68-74) and
            // the same libstdc++ std::sort as the reference, so equal keys land in the same places.
            const std::vector<BuildTri>& t = tris;
            std::sort(order.begin() + start, order.begin() + end,
                      [&t, axis](uint32_t a, uint32_t b) { return t[a].box.lo[axis] < t[b].box.lo[axis]; });
            const size_t mid = start + span / 2;
            left = build(start, mid);
            right = build(mid, end);
        }
        nodes[idx].box = box;
        nodes[idx].left = left;
        nodes[idx].right = right;
        return idx;
    }
};

inline double vx(const double* v, int k) { return v[k]; }

}  // namespace

std::string build_host_scene(const double* tv, const int32_t* tri_material, uint32_t n_tri,
                             const bdpt_material* materials, uint32_t n_mat, const uint32_t* light_triangles,
                             uint32_t n_lights, HostScene& out) {
    if (!tv || !tri_material || !materials) return "null table pointer";
    if (n_tri == 0) return "scene has no triangles (the reference cannot build a BVH over an empty list either)";
    if (n_tri >= (1u << 29)) return "too many triangles";
    if (n_mat == 0) return "scene has no materials";
    if (n_lights > 0 && !light_triangles) return "null light list";

    out = HostScene();
    out.mats.resize(n_mat);
    for (uint32_t i = 0; i < n_mat; i++) {
        const bdpt_material& m = materials[i];
        if (m.type < BDPT_LAMBERTIAN || m.type > BDPT_DIFFUSE_LIGHT) return "unknown material type";
        GpuMaterial g;
        g.rgb[0] = m.rgb[0]; g.rgb[1] = m.rgb[1]; g.rgb[2] = m.rgb[2];
        g.fuzz = m.fuzz < 1 ? m.fuzz : 1;            // metal::metal, src/materials/material.h:71
        g.ior = m.ior;
        g.type = m.type;
        g.pad = 0;
        out.mats[i] = g;
    }

    // triangle::triangle, src/objects/primatives/triangle.h:21-39
    out.tris.resize(n_tri);
    out.aux.resize(n_tri);
    std::vector<BuildTri> btris(n_tri);
    for (uint32_t i = 0; i < n_tri; i++) {
        const double* v0 = tv + 9 * (size_t)i;
        const double* v1 = v0 + 3;
        const double* v2 = v0 + 6;
        if (tri_material[i] < 0 || (uint32_t)tri_material[i] >= n_mat) return "triangle material index out of range";
        GpuTri& t = out.tris[i];
        for (int k = 0; k < 3; k++) {
            t.v0[k] = v0[k];
            t.e1[k] = v1[k] - v0[k];
            t.e2[k] = v2[k] - v0[k];
        }
        t.material = tri_material[i];
        t.pad = 0;
        const double nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
        const double ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
        const double nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
        const double nlen = std::sqrt(nx * nx + ny * ny + nz * nz);
        const double inv = 1 / nlen;                 // unit_vector: v / len == (1/len) * v, vec3.h:93-95,109-111
        GpuTriAux& a = out.aux[i];
        a.normal[0] = inv * nx; a.normal[1] = inv * ny; a.normal[2] = inv * nz;
        a.area = 0.5 * nlen;
        for (int k = 0; k < 3; k++) {
            const double lo = std::fmin(v0[k], std::fmin(v1[k], v2[k]));
            const double hi = std::fmax(v0[k], std::fmax(v1[k], v2[k]));
            if (lo <= hi) { btris[i].box.lo[k] = lo; btris[i].box.hi[k] = hi; }
            else { btris[i].box.lo[k] = hi; btris[i].box.hi[k] = lo; }
        }
    }

    // bvh_node(world.to_hittable_list()), src/main.cpp:66
    Builder b(btris);
    b.order.resize(n_tri);
    for (uint32_t i = 0; i < n_tri; i++) b.order[i] = i;
    b.nodes.reserve(n_tri);
    b.build(0, n_tri);

    // thread the pre-order array: skip = first node after the subtree
    const uint32_t n_nodes = (uint32_t)b.nodes.size();
    std::vector<uint32_t> subtree_end(n_nodes);
    std::vector<uint32_t> depth(n_nodes, 1);
    for (int64_t i = (int64_t)n_nodes - 1; i >= 0; i--) {
        const BuildNode& n = b.nodes[i];
        subtree_end[i] = (n.left >= 0) ? subtree_end[n.right] : (uint32_t)i + 1;
    }
    out.nodes.resize(n_nodes);
    out.tree_depth = 1;
    for (uint32_t i = 0; i < n_nodes; i++) {
        const BuildNode& n = b.nodes[i];
        GpuNode& g = out.nodes[i];
        for (int k = 0; k < 3; k++) { g.lo[k] = n.box.lo[k]; g.hi[k] = n.box.hi[k]; }
        const uint32_t end = subtree_end[i];
        g.skip = (end >= n_nodes) ? kEndOfTree : end;
        if (n.left >= 0) {
            g.next = (uint32_t)n.left;               // == i + 1 in pre-order
            g.tri0 = g.tri1 = -1;
            depth[n.left] = depth[n.right] = depth[i] + 1;
            if (depth[i] + 1 > out.tree_depth) out.tree_depth = depth[i] + 1;
        } else {
            g.next = g.skip;
            g.tri0 = ~n.left;
            g.tri1 = ~n.right;
            out.dfs_leaves.push_back((uint32_t)g.tri0);
            out.dfs_leaves.push_back((uint32_t)g.tri1);
        }
    }

    // the light collection handed to camera::render: `lights`, or the whole world when it is empty (main.cpp:67)
    if (n_lights == 0) {
        out.light_tris.resize(n_tri);
        for (uint32_t i = 0; i < n_tri; i++) out.light_tris[i] = i;
    } else {
        out.light_tris.assign(light_triangles, light_triangles + n_lights);
        for (uint32_t id : out.light_tris) if (id >= n_tri) return "light triangle index out of range";
    }
    // running area sums in the order triangle_collection::sample_surface adds them (triangle.h:203-219)
    out.light_accum.resize(out.light_tris.size());
    double accum = 0.0;
    for (size_t i = 0; i < out.light_tris.size(); i++) {
        accum += out.aux[out.light_tris[i]].area;
        out.light_accum[i] = accum;
    }
    out.light_total_area = accum;
    return std::string();
}

std::string camera_initialize(const bdpt_camera_public& c, bdpt_camera_state& out) {
    if (c.image_width <= 0) return "image_width must be positive";
    if (!(c.aspect_ratio > 0)) return "aspect_ratio must be positive";
    if (c.samples_per_pixel < 1) return "samples_per_pixel must be at least 1";
    if (c.defocus_angle > 0) return "defocus blur is not supported (the reference's loader disables it)";
    const double pi = 3.1415926535897932385;          // src/main.h:20
    int image_height = int(c.image_width / c.aspect_ratio);
    image_height = (image_height < 1) ? 1 : image_height;
    const int sqrt_spp = int(std::sqrt(c.samples_per_pixel));
    const double* lf = c.lookfrom;
    const double theta = c.vfov * pi / 180.0;
    const double h = std::tan(theta / 2);
    const double viewport_height = 2 * h * c.focus_dist;
    const double viewport_width = viewport_height * (double(c.image_width) / image_height);
    auto unit3 = [](const double* v, double* o) {
        const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double inv = 1 / len;
        o[0] = inv * v[0]; o[1] = inv * v[1]; o[2] = inv * v[2];
    };
    auto cross3 = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    double d[3] = {lf[0] - c.lookat[0], lf[1] - c.lookat[1], lf[2] - c.lookat[2]};
    double w[3], u[3], v[3], t[3];
    unit3(d, w);
    cross3(c.vup, w, t);
    unit3(t, u);
    cross3(w, u, v);
    double viewport_u[3], viewport_v[3];
    for (int k = 0; k < 3; k++) { viewport_u[k] = viewport_width * u[k]; viewport_v[k] = viewport_height * (-v[k]); }
    const double inv_w = 1.0 / c.image_width, inv_h = 1.0 / image_height;      // vec / int == (1/t) * vec
    for (int k = 0; k < 3; k++) {
        out.pixel_delta_u[k] = inv_w * viewport_u[k];
        out.pixel_delta_v[k] = inv_h * viewport_v[k];
    }
    const double half = 1.0 / 2;
    for (int k = 0; k < 3; k++) {
        const double upper_left = ((lf[k] - (c.focus_dist * w[k])) - half * viewport_u[k]) - half * viewport_v[k];
        out.pixel00_loc[k] = upper_left + 0.5 * (out.pixel_delta_u[k] + out.pixel_delta_v[k]);
        out.center[k] = lf[k];
        out.background[k] = c.background[k];
    }
    out.image_width = c.image_width;
    out.image_height = image_height;
    out.sqrt_spp = sqrt_spp;
    out.max_depth = c.max_depth;
    return std::string();
}

}  // namespace bdpt
