// Command-line renderer with the reference's interface (src/main.cpp:63-78): `bdpt_render [scene.yaml]`
// renders the scene file, or the built-in Cornell box when no file is given, and writes
// ./output/<file name> as an 8-bit PNG.  Everything goes through the C ABI of include/bdpt_amd.h.
//
// Options after the scene file (none exist in the reference; all optional):
//   --seed N        seed of the per-(pixel, sample) random streams (default 0)
//   --unidirectional   estimate with camera::path_trace_color (src/camera.h:255-292) instead of bidirectional_color
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/bdpt_amd.h"

namespace {

void check(int rc) {
    if (rc != BDPT_OK) throw std::runtime_error(bdpt_last_error());
}

struct FlatScene {
    bdpt_camera_public camera{};
    std::string output_name = "image.png";
    std::vector<double> vertices;
    std::vector<int32_t> tri_material;
    std::vector<bdpt_material> materials;
    std::vector<uint32_t> lights;
};

struct P3 { double x, y, z; };
P3 operator+(P3 a, P3 b) { return P3{a.x + b.x, a.y + b.y, a.z + b.z}; }

void add_triangle(FlatScene& s, P3 a, P3 b, P3 c, int32_t mat) {
    const double v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
    s.vertices.insert(s.vertices.end(), v, v + 9);
    s.tri_material.push_back(mat);
}
// add_quad_triangles, src/objects/primatives/triangle.h:232-241
void add_quad(FlatScene& s, P3 q, P3 u, P3 v, int32_t mat) {
    add_triangle(s, q, q + u, q + v, mat);
    add_triangle(s, q + u, q + u + v, q + v, mat);
}
// add_box_triangles, triangle.h:251-308
void add_box(FlatScene& s, P3 a, P3 b, int32_t mat, double rotate_y_degrees, P3 translate) {
    const P3 lo{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    const P3 hi{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    const P3 v000{lo.x, lo.y, lo.z}, v001{lo.x, lo.y, hi.z}, v010{lo.x, hi.y, lo.z}, v011{lo.x, hi.y, hi.z};
    const P3 v100{hi.x, lo.y, lo.z}, v101{hi.x, lo.y, hi.z}, v110{hi.x, hi.y, lo.z}, v111{hi.x, hi.y, hi.z};
    const std::array<std::array<P3, 3>, 12> faces = {{
        {v001, v101, v111}, {v001, v111, v011},   // +Z
        {v000, v010, v110}, {v000, v110, v100},   // -Z
        {v000, v001, v011}, {v000, v011, v010},   // -X
        {v101, v100, v110}, {v101, v110, v111},   // +X
        {v011, v111, v110}, {v011, v110, v010},   // +Y
        {v000, v100, v101}, {v000, v101, v001},   // -Y
    }};
    const double pi = 3.1415926535897932385;                      // src/main.h:20
    const double radians = rotate_y_degrees * pi / 180.0;
    const double sin_theta = std::sin(radians), cos_theta = std::cos(radians);
    auto rot = [&](P3 p) { return P3{(cos_theta * p.x) + (sin_theta * p.z), p.y, (-sin_theta * p.x) + (cos_theta * p.z)}; };
    for (const auto& f : faces) {
        P3 p0 = f[0], p1 = f[1], p2 = f[2];
        if (rotate_y_degrees != 0.0) { p0 = rot(p0); p1 = rot(p1); p2 = rot(p2); }
        add_triangle(s, p0 + translate, p1 + translate, p2 + translate, mat);
    }
}

bdpt_material material(int type, double r, double g, double b) {
    bdpt_material m{};
    m.type = type;
    m.rgb[0] = r; m.rgb[1] = g; m.rgb[2] = b;
    return m;
}

// cornell_box(), src/main.cpp:13-61.  The reference hands camera::render a separate `lights` collection
// holding one quad WITHOUT a material (main.cpp:35-37), which build_light_path dereferences
// (camera.h:394) - undefined behaviour there.  Here the light collection is the emissive quad of the
// world itself, which is what a scene file with the same geometry produces (scene_loader.h:190-202).
FlatScene cornell_box() {
    FlatScene s;
    s.materials = {material(BDPT_LAMBERTIAN, .65, .05, .05), material(BDPT_LAMBERTIAN, .73, .73, .73),
                   material(BDPT_LAMBERTIAN, .12, .45, .15), material(BDPT_DIFFUSE_LIGHT, 15, 15, 15)};
    const int32_t red = 0, white = 1, green = 2, light = 3;
    add_quad(s, P3{555, 0, 0}, P3{0, 0, 555}, P3{0, 555, 0}, green);
    add_quad(s, P3{0, 0, 555}, P3{0, 0, -555}, P3{0, 555, 0}, red);
    add_quad(s, P3{0, 555, 0}, P3{555, 0, 0}, P3{0, 0, 555}, white);
    add_quad(s, P3{0, 0, 555}, P3{555, 0, 0}, P3{0, 0, -555}, white);
    add_quad(s, P3{555, 0, 555}, P3{-555, 0, 0}, P3{0, 555, 0}, white);
    const uint32_t first_light = (uint32_t)s.tri_material.size();
    add_quad(s, P3{213, 554, 227}, P3{130, 0, 0}, P3{0, 0, 105}, light);
    s.lights = {first_light, first_light + 1};
    add_box(s, P3{0, 0, 0}, P3{165, 330, 165}, white, 15.0, P3{265, 0, 295});
    bdpt_camera_public& c = s.camera;
    c.aspect_ratio = 1.0; c.image_width = 800; c.samples_per_pixel = 5; c.max_depth = 10;
    c.vfov = 40;
    c.lookfrom[0] = 278; c.lookfrom[1] = 278; c.lookfrom[2] = -800;
    c.lookat[0] = 278; c.lookat[1] = 278; c.lookat[2] = 0;
    c.vup[1] = 1;
    c.defocus_angle = 0; c.focus_dist = 10;
    s.output_name = "cornell_box.png";
    return s;
}

FlatScene from_file(const char* path) {
    bdpt_scene_file* f = nullptr;
    check(bdpt_scene_file_load(path, &f));
    bdpt_scene_file_view v{};
    check(bdpt_scene_file_get(f, &v));
    FlatScene s;
    s.camera = v.camera;
    s.output_name = v.output_name;
    s.vertices.assign(v.tri_vertices, v.tri_vertices + (size_t)v.n_triangles * 9);
    s.tri_material.assign(v.tri_material, v.tri_material + v.n_triangles);
    s.materials.assign(v.materials, v.materials + v.n_materials);
    s.lights.assign(v.light_triangles, v.light_triangles + v.n_lights);
    bdpt_scene_file_destroy(f);
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        uint64_t seed = 0;
        int flags = 0;
        const char* scene_path = nullptr;
        for (int i = 1; i < argc; i++) {
            if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
            else if (!std::strcmp(argv[i], "--unidirectional")) flags |= BDPT_RENDER_UNIDIRECTIONAL;
            else if (!scene_path) scene_path = argv[i];
            else throw std::runtime_error(std::string("unexpected argument: ") + argv[i]);
        }
        FlatScene s = scene_path ? from_file(scene_path) : cornell_box();
        if (s.tri_material.empty()) throw std::runtime_error("scene has no triangles");
        if (s.materials.empty()) s.materials.push_back(material(BDPT_LAMBERTIAN, 0.8, 0.8, 0.8));

        bdpt_camera_state st{};
        check(bdpt_camera_initialize(&s.camera, &st));
        bdpt_context* ctx = nullptr;
        bdpt_scene* scene = nullptr;
        check(bdpt_context_create(0, &ctx));
        // an empty `lights` collection selects the whole world (main.cpp:67)
        check(bdpt_scene_create(ctx, s.vertices.data(), s.tri_material.data(), (uint32_t)s.tri_material.size(), s.materials.data(),
                                (uint32_t)s.materials.size(), s.lights.data(), (uint32_t)s.lights.size(), &scene));
        std::vector<double> fb((size_t)st.image_width * st.image_height * 3, 0.0);
        bdpt_render_params rp{};
        rp.seed = seed;
        rp.flags = flags;
        check(bdpt_render(ctx, scene, &st, &rp, fb.data(), nullptr));
        std::vector<uint8_t> rgb(fb.size());
        check(bdpt_colors_to_rgb8(fb.data(), st.image_width, st.image_height, st.sqrt_spp * st.sqrt_spp, rgb.data()));
        check(bdpt_write_png(s.output_name.c_str(), rgb.data(), st.image_width, st.image_height));
        std::cout << "Done. Saved image to output/" << s.output_name << "\n";
        bdpt_scene_destroy(scene);
        bdpt_context_destroy(ctx);
    } catch (const std::exception& ex) {
        std::cerr << "Failed to load scene: " << ex.what() << "\n";       // src/main.cpp:72-75
        return 1;
    }
    return 0;
}
