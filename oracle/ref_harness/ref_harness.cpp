// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Never shipped, never linked by the product.
//
// Drives the UNMODIFIED reference headers under /root/reference/src (compiled where
// they lie, see oracle/Makefile) as a deterministic single-threaded oracle and dumps
// what they compute, so that golden vectors can be committed under tests/golden/.
//
// The per-pixel loop below is the worker loop of camera::render (src/camera.h:116-124)
// run on one thread; everything it calls (get_ray, ray_color -> bidirectional_color,
// bvh_node::hit, materials ...) is the reference's own code.
//
// Build variants:
//   default           : src/main.h is pre-empted by main_shim.h (re-seedable RNG streams)
//   -DHARNESS_NO_SHIM : the real src/main.h is used (process-global mt19937); only the
//                       `render_serial` / `rand` operations make sense then.

// Standard headers first, so that the `private` -> `public` trick below cannot touch them.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <csetjmp>
#include <png.h>

#include "../flat_scene.h"

#define private public
#ifdef HARNESS_NO_SHIM
#include "main.h"
#else
#include "main_shim.h"
#endif
#include "bvh.h"
#include "camera.h"
#include "hittable.h"
#include "material.h"
#include "triangle.h"
#include "texture.h"
#include "constant_medium.h"
#undef private

// textures of a composite scene, in file order (children of a checker come first)
static std::vector<std::shared_ptr<texture>> g_textures;

#ifndef HARNESS_NO_SHIM
// A noise_texture fills its perlin tables from the global generator when it is constructed (perlin.h:6-14): every
// noise texture is built on a stream of its own so that the tables do not depend on what was drawn before.
static void seed_for_noise_texture(uint32_t index) {
    harness_rng().engine = 0;
    harness_rng().stream.key = orng_stream_key(0x7065726c696eULL, index, 0);
    harness_rng().stream.counter = 0;
}
#endif

static void make_textures(const fs_scene& fs, const char* scene_path) {
    g_textures.clear();
    for (uint32_t i = 0; i < fs.n_textures; i++) {
        const fs_texture& t = fs.textures[i];
        switch (t.type) {
            case 0: g_textures.push_back(std::make_shared<solid_color>(color(t.rgb[0], t.rgb[1], t.rgb[2]))); break;
            case 1: g_textures.push_back(std::make_shared<checker_texture>(t.scale, g_textures.at(t.even), g_textures.at(t.odd))); break;
            case 2:
#ifndef HARNESS_NO_SHIM
                seed_for_noise_texture(i);
#endif
                g_textures.push_back(std::make_shared<noise_texture>(t.scale));
                break;
            case 3: {
                const std::string file = std::string(scene_path) + ".tex" + std::to_string(i) + ".png";
                g_textures.push_back(std::make_shared<image_texture>(file.c_str()));
                break;
            }
            default: fprintf(stderr, "bad texture type %d\n", t.type); exit(2);
        }
    }
}

static std::shared_ptr<material> make_material(const fs_material& m) {
    color c(m.rgb[0], m.rgb[1], m.rgb[2]);
    std::shared_ptr<texture> tex = (m.pad > 0) ? g_textures.at(m.pad - 1) : nullptr;
    switch (m.type) {
        case 0: return tex ? std::make_shared<lambertian>(tex) : std::make_shared<lambertian>(c);
        case 1: return std::make_shared<metal>(c, m.fuzz);
        case 2: return std::make_shared<dielectric>(m.ior);
        case 3: return tex ? std::make_shared<diffuse_light>(tex) : std::make_shared<diffuse_light>(c);
        case 4: return tex ? std::make_shared<isotropic>(tex) : std::make_shared<isotropic>(c);
    }
    fprintf(stderr, "bad material type %d\n", m.type);
    exit(2);
}

struct built_scene {
    triangle_collection world;
    triangle_collection lights;
    std::vector<std::shared_ptr<material>> mats;
    std::shared_ptr<bvh_node> bvh;
    std::vector<std::shared_ptr<hittable>> leaf_ptrs;   // world triangle i -> the heap copy the BVH holds
    std::vector<std::shared_ptr<hittable>> objects;     // composite scenes: object k as it stands in the world's list
    std::vector<const material*> phase;                 // composite scenes: phase function of object k (or null)
    std::vector<std::shared_ptr<material>> tri_mats;    // material object of world triangle i
    camera cam;
};

static point3 P(const double* v) { return point3(v[0], v[1], v[2]); }

// unique_materials: give every triangle its own material object so that a hit_record's
// `mat` pointer identifies the triangle that was hit (used only by the `hits` operation;
// material values do not take part in intersection).
static void build(const fs_scene& fs, built_scene& b, bool unique_materials, const char* scene_path) {
    make_textures(fs, scene_path);
    for (uint32_t i = 0; i < fs.n_materials; i++) b.mats.push_back(make_material(fs.materials[i]));
    std::vector<std::shared_ptr<material>>& tri_mats = b.tri_mats;
    for (uint32_t i = 0; i < fs.n_triangles; i++) {
        const fs_triangle& t = fs.triangles[i];
        std::shared_ptr<material> m = unique_materials ? make_material(fs.materials[t.material]) : b.mats[t.material];
        tri_mats.push_back(m);
        b.world.add(triangle(P(t.v0), P(t.v1), P(t.v2), m));
    }
    if (!fs.composite) {
        for (uint32_t i = 0; i < fs.n_lights; i++) {
            const fs_triangle& t = fs.triangles[fs.lights[i]];
            b.lights.add(triangle(P(t.v0), P(t.v1), P(t.v2), tri_mats[fs.lights[i]]));
        }
        // src/main.cpp:66 : bvh_node(world.to_hittable_list())
        hittable_list list = b.world.to_hittable_list();
        b.leaf_ptrs = list.objects;
        if (fs.n_triangles > 0) b.bvh = std::make_shared<bvh_node>(list);
    } else {
        // The world a C++ host of the reference would assemble: bare triangles and, per object, the mesh as a
        // bvh_node of its own wrapped in rotate_y / translate / constant_medium (hittable.h:46-157,
        // constant_medium.h), all in one hittable_list handed to bvh_node as in src/main.cpp:66.
        std::vector<int> owner(fs.n_triangles, -1);
        for (uint32_t k = 0; k < fs.n_objects; k++)
            for (uint32_t i = 0; i < fs.objects[k].n_triangles; i++) owner[fs.objects[k].first_triangle + i] = (int)k;
        hittable_list list;
        b.leaf_ptrs.assign(fs.n_triangles, nullptr);
        b.objects.assign(fs.n_objects, nullptr);
        b.phase.assign(fs.n_objects, nullptr);
        std::vector<uint32_t> loose;
        for (uint32_t i = 0; i < fs.n_triangles;) {
            if (owner[i] < 0) {
                auto tp = make_shared<triangle>(b.world.tris[i]);
                b.leaf_ptrs[i] = tp;
                list.add(tp);
                loose.push_back(i);
                i++;
                continue;
            }
            // the mesh that starts here, once; every object that names this range wraps the same bvh_node
            const uint32_t n_mesh = fs.objects[owner[i]].n_triangles;
            hittable_list mesh;
            for (uint32_t q = 0; q < n_mesh; q++) {
                auto tp = make_shared<triangle>(b.world.tris[i + q]);
                b.leaf_ptrs[i + q] = tp;
                mesh.add(tp);
            }
            shared_ptr<hittable> shared_mesh = make_shared<bvh_node>(mesh);
            for (uint32_t k = 0; k < fs.n_objects; k++) {
                const fs_object& ob = fs.objects[k];
                if (ob.first_triangle != i) continue;
                shared_ptr<hittable> h = shared_mesh;
                if (ob.flags & 1u) h = make_shared<rotate_y>(h, ob.rotate_y_degrees);
                if (ob.flags & 2u) h = make_shared<translate>(h, vec3(ob.offset[0], ob.offset[1], ob.offset[2]));
                if (ob.flags & 4u) {
                    const fs_material& pm = fs.materials[ob.phase_material];
                    shared_ptr<constant_medium> cm = (pm.pad > 0)
                        ? make_shared<constant_medium>(h, ob.density, g_textures.at(pm.pad - 1))
                        : make_shared<constant_medium>(h, ob.density, color(pm.rgb[0], pm.rgb[1], pm.rgb[2]));
                    b.phase[k] = cm->phase_function.get();
                    h = cm;
                }
                b.objects[k] = h;
                list.add(h);
            }
            i += n_mesh;
        }
        // the light collection: the listed triangles, or every triangle of the world's list
        if (fs.n_lights > 0) { loose.clear(); for (uint32_t i = 0; i < fs.n_lights; i++) loose.push_back(fs.lights[i]); }
        for (uint32_t id : loose) {
            const fs_triangle& t = fs.triangles[id];
            b.lights.add(triangle(P(t.v0), P(t.v1), P(t.v2), tri_mats[id]));
        }
        b.bvh = std::make_shared<bvh_node>(list);
    }

    const fs_camera& c = fs.camera;
    b.cam.aspect_ratio = c.aspect_ratio;
    b.cam.image_width = c.image_width;
    b.cam.samples_per_pixel = c.samples_per_pixel;
    b.cam.max_depth = c.max_depth;
    b.cam.background = color(c.background[0], c.background[1], c.background[2]);
    b.cam.vfov = c.vfov;
    b.cam.lookfrom = P(c.lookfrom);
    b.cam.lookat = P(c.lookat);
    b.cam.vup = P(c.vup);
    b.cam.defocus_angle = c.defocus_angle;
    b.cam.focus_dist = c.focus_dist;
}

static void write_file(const char* path, const void* data, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(data, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(3); }
    fclose(f);
}

static std::vector<double> read_doubles(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(3); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<double> v(n / 8);
    if (fread(v.data(), 8, v.size(), f) != v.size()) exit(3);
    fclose(f);
    return v;
}

static void push_stats(std::vector<double>& out) {
    auto& s = bvh_stats();
    out.push_back((double)s.rays_traced.load());
    out.push_back((double)s.aabb_tests.load());
    out.push_back((double)s.aabb_hits.load());
    out.push_back((double)s.bvh_node_visits.load());
    out.push_back((double)s.triangle_tests.load());
    out.push_back((double)s.triangle_hits.load());
    out.push_back((double)s.bvh_nodes_built.load());
}

// DFS (left then right) over the reference tree: leaf triangle ids in visiting order and
// every bvh_node's box in pre-order.
static std::map<const hittable*, double> g_object_ids;    // composite scenes: object k -> -(k + 2)
static void walk(const hittable* h, const std::map<const hittable*, uint32_t>& ids,
                 std::vector<double>& leaves, std::vector<double>& boxes) {
    const bvh_node* n = g_object_ids.count(h) ? nullptr : dynamic_cast<const bvh_node*>(h);   // an object is a leaf of the world's tree
    if (n) {
        boxes.push_back(n->bbox.x.min); boxes.push_back(n->bbox.x.max);
        boxes.push_back(n->bbox.y.min); boxes.push_back(n->bbox.y.max);
        boxes.push_back(n->bbox.z.min); boxes.push_back(n->bbox.z.max);
        walk(n->left.get(), ids, leaves, boxes);
        walk(n->right.get(), ids, leaves, boxes);
    } else {
        leaves.push_back(ids.count(h) ? (double)ids.at(h) : g_object_ids.at(h));
    }
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: ref_harness scene.bin op [args]\n");
        return 1;
    }
    const std::string op = argv[2];

#ifndef HARNESS_NO_SHIM
    if (op == "rand") {   // first N doubles of the serial default-seeded engine (src/main.h:28-32 behaviour)
        int n = atoi(argv[3]);
        harness_rng().engine = 1;
        std::vector<double> out;
        for (int i = 0; i < n; i++) out.push_back(random_double());
        write_file(argv[4], out.data(), out.size() * 8);
        return 0;
    }
#else
    if (op == "rand") {
        int n = atoi(argv[3]);
        std::vector<double> out;
        for (int i = 0; i < n; i++) out.push_back(random_double());
        write_file(argv[4], out.data(), out.size() * 8);
        return 0;
    }
#endif

    fs_scene fs;
    if (fs_load(argv[1], &fs) != 0) { fprintf(stderr, "bad scene file\n"); return 2; }

    built_scene b;
    build(fs, b, op == "hits", argv[1]);
    camera& cam = b.cam;
    cam.initialize();
    // src/main.cpp:67
    const triangle_collection& lights = b.lights.empty() ? b.world : b.lights;
    for (uint32_t k = 0; k < b.objects.size(); k++) g_object_ids[b.objects[k].get()] = -(double)(k + 2);

    if (op == "perlin") {   // argv[3] out.bin: per noise texture 768 doubles (randvec) then 768 doubles (perm_x, perm_y, perm_z)
        std::vector<double> o;
        for (uint32_t i = 0; i < fs.n_textures; i++) {
            if (fs.textures[i].type != 2) continue;
            const noise_texture* nt = static_cast<const noise_texture*>(g_textures[i].get());
            for (int k = 0; k < 256; k++) for (int a = 0; a < 3; a++) o.push_back(nt->noise.randvec[k][a]);
            for (int k = 0; k < 256; k++) o.push_back(nt->noise.perm_x[k]);
            for (int k = 0; k < 256; k++) o.push_back(nt->noise.perm_y[k]);
            for (int k = 0; k < 256; k++) o.push_back(nt->noise.perm_z[k]);
        }
        write_file(argv[3], o.data(), o.size() * 8);
        return 0;
    }
    if (op == "image_bytes") {   // argv[3] out.bin: per image texture width, height (2 doubles) then its bytes as doubles
        std::vector<double> o;
        for (uint32_t i = 0; i < fs.n_textures; i++) {
            if (fs.textures[i].type != 3) continue;
            const image_texture* it = static_cast<const image_texture*>(g_textures[i].get());
            const int w = it->image.width(), h = it->image.height();
            o.push_back(w); o.push_back(h);
            for (int k = 0; k < w * h * 3; k++) o.push_back(it->image.bdata[k]);
        }
        write_file(argv[3], o.data(), o.size() * 8);
        return 0;
    }

    if (op == "camera") {
        std::vector<double> o;
        for (int k = 0; k < 3; k++) o.push_back(cam.center[k]);
        for (int k = 0; k < 3; k++) o.push_back(cam.pixel00_loc[k]);
        for (int k = 0; k < 3; k++) o.push_back(cam.pixel_delta_u[k]);
        for (int k = 0; k < 3; k++) o.push_back(cam.pixel_delta_v[k]);
        o.push_back(cam.image_height);
        o.push_back(cam.sqrt_spp);
        for (int k = 0; k < 3; k++) o.push_back(cam.defocus_disk_u[k]);
        for (int k = 0; k < 3; k++) o.push_back(cam.defocus_disk_v[k]);
        write_file(argv[3], o.data(), o.size() * 8);
        return 0;
    }

    if (op == "dfs") {   // argv[3] leaves.bin  argv[4] boxes.bin
        std::map<const hittable*, uint32_t> ids;
        for (uint32_t i = 0; i < b.leaf_ptrs.size(); i++) ids[b.leaf_ptrs[i].get()] = i;
        std::vector<double> leaves, boxes;
        walk(b.bvh.get(), ids, leaves, boxes);
        write_file(argv[3], leaves.data(), leaves.size() * 8);
        write_file(argv[4], boxes.data(), boxes.size() * 8);
        return 0;
    }

    if (op == "triangles") {  // precomputed per-triangle data of src/objects/primatives/triangle.h:21-39
        std::vector<double> o;
        for (const auto& t : b.world.tris) {
            for (int k = 0; k < 3; k++) o.push_back(t.edge1[k]);
            for (int k = 0; k < 3; k++) o.push_back(t.edge2[k]);
            for (int k = 0; k < 3; k++) o.push_back(t.normal[k]);
            o.push_back(t.area);
            o.push_back(t.bbox.x.min); o.push_back(t.bbox.x.max);
            o.push_back(t.bbox.y.min); o.push_back(t.bbox.y.max);
            o.push_back(t.bbox.z.min); o.push_back(t.bbox.z.max);
        }
        write_file(argv[3], o.data(), o.size() * 8);
        return 0;
    }

    if (op == "hits") {   // argv[3] rays.bin (o[3] d[3] tmin tmax)*  argv[4] out.bin
        std::map<const material*, double> by_mat;
        for (uint32_t i = 0; i < b.world.tris.size(); i++) by_mat[b.world.tris[i].mat.get()] = (double)i;
        for (uint32_t k = 0; k < b.phase.size(); k++) if (b.phase[k]) by_mat[b.phase[k]] = -(double)(k + 2);
        std::vector<double> rays = read_doubles(argv[3]);
        size_t n = rays.size() / 8;
        std::vector<double> out;
        out.reserve(n * 12);
        reset_bvh_stats();
        for (size_t i = 0; i < n; i++) {
            const double* r = &rays[i * 8];
            ray rr(point3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]), 0.0);
            hit_record rec;
            rec.u = 0; rec.v = 0;                        // (a medium leaves them unset, constant_medium.h:45-50)
#ifndef HARNESS_NO_SHIM
            harness_rng().engine = 0;                    // query i owns stream (seed 0, pixel i, sample 0): media draw
            harness_rng().stream.key = orng_stream_key(0, i, 0);
            harness_rng().stream.counter = 0;
#endif
            bool h = b.bvh->hit(rr, interval(r[6], r[7]), rec);
            if (!h) {
                out.push_back(-1.0);
                for (int k = 0; k < 10; k++) out.push_back(0.0);
            } else {
                out.push_back(by_mat.at(rec.mat.get()));
                out.push_back(rec.t); out.push_back(rec.u); out.push_back(rec.v);
                out.push_back(rec.front_face ? 1.0 : 0.0);
                for (int k = 0; k < 3; k++) out.push_back(rec.p[k]);
                for (int k = 0; k < 3; k++) out.push_back(rec.normal[k]);
            }
        }
        std::vector<double> st;
        push_stats(st);
        write_file(argv[4], out.data(), out.size() * 8);
        if (argc > 5) write_file(argv[5], st.data(), st.size() * 8);
        return 0;
    }

    if (op == "visible") {   // argv[3] pairs.bin (a[3] b[3])*  argv[4] out.bin
        std::vector<double> pairs = read_doubles(argv[3]);
        size_t n = pairs.size() / 6;
        std::vector<double> out;
        for (size_t i = 0; i < n; i++) {
            const double* p = &pairs[i * 6];
#ifndef HARNESS_NO_SHIM
            harness_rng().engine = 0;
            harness_rng().stream.key = orng_stream_key(0, i, 0);
            harness_rng().stream.counter = 0;
#endif
            out.push_back(cam.visible(point3(p[0], p[1], p[2]), point3(p[3], p[4], p[5]), *b.bvh) ? 1.0 : 0.0);
        }
        write_file(argv[4], out.data(), out.size() * 8);
        return 0;
    }

    if (op == "rgb8") {   // argv[3] fb.bin  W H spp  out.bin   (src/image/wpng.h:14-35)
        std::vector<double> fbd = read_doubles(argv[3]);
        int W = atoi(argv[4]), H = atoi(argv[5]), spp = atoi(argv[6]);
        std::vector<color> fb(size_t(W) * H);
        for (size_t i = 0; i < fb.size(); i++) fb[i] = color(fbd[3 * i], fbd[3 * i + 1], fbd[3 * i + 2]);
        std::vector<uint8_t> rgb;
        colors_to_rgb8(fb, W, H, spp, rgb);
        write_file(argv[7], rgb.data(), rgb.size());
        return 0;
    }

#ifndef HARNESS_NO_SHIM
    if (op == "render" || op == "samples" || op == "render_uni") {
        // argv: seed x0 y0 x1 y1 out.bin [stats.bin]
        // render_uni: the unidirectional estimator, camera::path_trace_color (src/camera.h:255-292), which
        // ray_color only selects when `lights` is not a triangle_collection (camera.h:245-253)
        const bool unidirectional = (op == "render_uni");
        uint64_t seed = strtoull(argv[3], nullptr, 10);
        int x0 = atoi(argv[4]), y0 = atoi(argv[5]), x1 = atoi(argv[6]), y1 = atoi(argv[7]);
        const int W = cam.image_width;
        const bool per_sample = (op == "samples");
        std::vector<double> out;
        reset_bvh_stats();
        harness_rng().engine = 0;
        harness_rng().total_draws = 0;
        for (int j = y0; j < y1; j++) {
            for (int i = x0; i < x1; i++) {
                color pixel_color(0, 0, 0);
                for (int s_j = 0; s_j < cam.sqrt_spp; s_j++) {
                    for (int s_i = 0; s_i < cam.sqrt_spp; s_i++) {
                        harness_rng().stream.key =
                            orng_stream_key(seed, (uint64_t)j * W + i, (uint64_t)s_j * cam.sqrt_spp + s_i);
                        harness_rng().stream.counter = 0;
                        ray r = cam.get_ray(i, j, s_i, s_j);
                        color c = unidirectional ? cam.path_trace_color(r, cam.max_depth, *b.bvh, lights)
                                                 : cam.ray_color(r, cam.max_depth, *b.bvh, lights);
                        pixel_color += c;
                        if (per_sample) {
                            out.push_back(c[0]); out.push_back(c[1]); out.push_back(c[2]);
                            out.push_back((double)harness_rng().stream.counter);
                        }
                    }
                }
                if (!per_sample) { out.push_back(pixel_color[0]); out.push_back(pixel_color[1]); out.push_back(pixel_color[2]); }
            }
        }
        write_file(argv[8], out.data(), out.size() * 8);
        if (argc > 9) {
            std::vector<double> st;
            push_stats(st);
            st.push_back((double)harness_rng().total_draws);
            write_file(argv[9], st.data(), st.size() * 8);
        }
        return 0;
    }
#endif

#ifndef HARNESS_NO_SHIM
    if (op == "bench") {
        // argv: seed seconds "i,j;i,j;..."  -> one JSON line: pixels finished and the seconds they took (scene loading
        // and the BVH build are outside).  The reference's own per-pixel loop (camera.h:116-124) over the listed
        // pixels, single-threaded, until the list or the time runs out.
        uint64_t seed = strtoull(argv[3], nullptr, 10);
        const double seconds = atof(argv[4]);
        std::vector<std::pair<int, int>> pixels;
        for (const char* c = argv[5]; *c;) {
            char* end;
            const long i = strtol(c, &end, 10);
            if (*end != ',') break;
            const long j = strtol(end + 1, &end, 10);
            pixels.push_back({(int)i, (int)j});
            c = (*end == ';') ? end + 1 : end;
            if (*end != ';') break;
        }
        const int W = cam.image_width;
        harness_rng().engine = 0;
        double sink = 0;
        size_t done = 0;
        const auto t0 = std::chrono::steady_clock::now();
        double dt = 0;
        for (const auto& px : pixels) {
            const int i = px.first, j = px.second;
            color pixel_color(0, 0, 0);
            for (int s_j = 0; s_j < cam.sqrt_spp; s_j++)
                for (int s_i = 0; s_i < cam.sqrt_spp; s_i++) {
                    harness_rng().stream.key = orng_stream_key(seed, (uint64_t)j * W + i, (uint64_t)s_j * cam.sqrt_spp + s_i);
                    harness_rng().stream.counter = 0;
                    ray r = cam.get_ray(i, j, s_i, s_j);
                    pixel_color += cam.ray_color(r, cam.max_depth, *b.bvh, lights);
                }
            sink += pixel_color[0] + pixel_color[1] + pixel_color[2];
            done++;
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > seconds) break;
        }
        printf("{\"pixels\": %zu, \"seconds\": %.6f, \"spp_used\": %d, \"checksum\": %.17g}\n", done, dt, cam.sqrt_spp * cam.sqrt_spp, sink);
        return 0;
    }
#endif

    if (op == "render_serial") {   // the reference's own global serial RNG; argv[3] out.bin
#ifndef HARNESS_NO_SHIM
        harness_rng().engine = 1;
#endif
        const int W = cam.image_width, H = cam.image_height;
        std::vector<double> out;
        reset_bvh_stats();
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                color pixel_color(0, 0, 0);
                for (int s_j = 0; s_j < cam.sqrt_spp; s_j++)
                    for (int s_i = 0; s_i < cam.sqrt_spp; s_i++) {
                        ray r = cam.get_ray(i, j, s_i, s_j);
                        pixel_color += cam.ray_color(r, cam.max_depth, *b.bvh, lights);
                    }
                out.push_back(pixel_color[0]); out.push_back(pixel_color[1]); out.push_back(pixel_color[2]);
            }
        write_file(argv[3], out.data(), out.size() * 8);
        if (argc > 4) {
            std::vector<double> st;
            push_stats(st);
            write_file(argv[4], st.data(), st.size() * 8);
        }
        return 0;
    }

    fprintf(stderr, "unknown op %s\n", op.c_str());
    return 1;
}
