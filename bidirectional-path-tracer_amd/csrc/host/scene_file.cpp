// Scene files: the host-side work of the reference's load_scene_from_yaml (src/scene/scene_loader.h:480-523)
// and its OBJ reader, producing the flat tables bdpt_scene_create takes, and the PNG writer of
// src/image/wpng.h:38-89.  No GPU work here.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/bdpt_amd.h"
#include "mini_yaml.h"

using bdpt_host::YamlNode;

extern "C" void bdpt_internal_set_last_error(const char* msg);

struct bdpt_scene_file {
    bdpt_camera_public camera{};
    std::string output_name;
    std::vector<double> vertices;            // 9 per triangle
    std::vector<int32_t> tri_material;
    std::vector<bdpt_material> materials;
    std::vector<uint32_t> light_triangles;
};

namespace {

// the error text bdpt_last_error() returns (capi.cpp)
struct ErrorSink {
    ErrorSink& operator=(const std::string& msg) { bdpt_internal_set_last_error(msg.c_str()); return *this; }
    ErrorSink& operator=(const char* msg) { bdpt_internal_set_last_error(msg); return *this; }
} g_file_error;

struct V3 { double x, y, z; };

// ---- scalar access with the fallbacks of scene_loader.h:32-99
std::string node_to_string(const YamlNode& n, const std::string& def = {}) { return n.is_scalar() ? n.scalar : def; }
double node_to_double(const YamlNode& n, double def = 0.0) {
    double v;
    return (n.is_scalar() && bdpt_host::yaml_to_double(n.scalar, v)) ? v : def;
}
int node_to_int(const YamlNode& n, int def = 0) {
    int v;
    return (n.is_scalar() && bdpt_host::yaml_to_int(n.scalar, v)) ? v : def;
}
std::vector<double> node_to_double_list(const YamlNode& n) {
    std::vector<double> out;
    if (!n.is_sequence()) return out;
    for (const YamlNode& e : n.items) out.push_back(node_to_double(e, 0.0));
    return out;
}
V3 read_vec3(const YamlNode& n, V3 fallback) {
    const std::vector<double> v = node_to_double_list(n);
    return v.size() < 3 ? fallback : V3{v[0], v[1], v[2]};
}
// scene_loader.h:81-92: components given in 0-255 are scaled to 0-1
V3 read_color_scaled(const YamlNode& n, V3 fallback) {
    const std::vector<double> v = node_to_double_list(n);
    if (v.size() < 3) return fallback;
    double r = v[0], g = v[1], b = v[2];
    const double maxc = std::max({std::fabs(r), std::fabs(g), std::fabs(b)});
    if (maxc > 1.0 && maxc <= 255.0) {
        const double s = 1.0 / 255.0;
        r *= s; g *= s; b *= s;
    }
    return V3{r, g, b};
}

bdpt_material make_material(int type, V3 rgb, double fuzz, double ior) {
    bdpt_material m{};
    m.type = type;
    m.rgb[0] = rgb.x; m.rgb[1] = rgb.y; m.rgb[2] = rgb.z;
    m.fuzz = fuzz;
    m.ior = ior;
    return m;
}

// build_material, scene_loader.h:101-170
bdpt_material build_material(const YamlNode& node) {
    if (!node.is_map()) throw std::runtime_error("Material must be a mapping");
    const V3 black{0, 0, 0};
    const std::string type_str = node_to_string(node["type"], "");
    if (!type_str.empty()) {
        const V3 color_value = read_color_scaled(
            node["color"], read_color_scaled(node["albedo"], read_color_scaled(node["base_color"], read_color_scaled(node["base_colour"], black))));
        if (type_str == "light" || type_str == "diffuse_light")
            return make_material(BDPT_DIFFUSE_LIGHT, read_vec3(node["emission"], black), 0.0, 0.0);      // linear, unscaled
        if (type_str == "lambertian") return make_material(BDPT_LAMBERTIAN, color_value, 0.0, 0.0);
        if (type_str == "metal")
            return make_material(BDPT_METAL, color_value, std::clamp(node_to_double(node["roughness"], 0.0), 0.0, 1.0), 0.0);
        if (type_str == "dielectric" || type_str == "glass") {
            const double ior = node_to_double(node["ior"], 1.5);
            return make_material(BDPT_DIELECTRIC, black, 0.0, ior > 0.0 ? ior : 1.5);
        }
        // unknown type: the legacy mapping below
    }
    V3 base_color = read_color_scaled(node["base_color"], black);
    if (node["base_colour"].defined()) base_color = read_color_scaled(node["base_colour"], base_color);
    V3 emission = read_color_scaled(node["emission"], black);
    if (emission.x * emission.x + emission.y * emission.y + emission.z * emission.z > 0.0) {
        const double maxc = std::max({std::fabs(emission.x), std::fabs(emission.y), std::fabs(emission.z)});
        if (maxc > 50.0) {
            const double s = 50.0 / maxc;
            emission = V3{emission.x * s, emission.y * s, emission.z * s};
        }
        return make_material(BDPT_DIFFUSE_LIGHT, emission, 0.0, 0.0);
    }
    double transmission = node_to_double(node["transmission"], 0.0);
    if (transmission == 0.0) transmission = node_to_double(node["spec_trans"], 0.0);
    const double ior = node_to_double(node["ior"], 1.5);
    if (transmission > 0.0) return make_material(BDPT_DIELECTRIC, black, 0.0, ior > 0.0 ? ior : 1.5);
    const double metallic = node_to_double(node["metallic"], 0.0);
    const double roughness = std::clamp(node_to_double(node["roughness"], 0.0), 0.0, 1.0);
    if (metallic > 0.5) return make_material(BDPT_METAL, base_color, roughness, 0.0);
    return make_material(BDPT_LAMBERTIAN, base_color, 0.0, 0.0);
}

struct Loader {
    bdpt_scene_file& out;
    std::unordered_map<std::string, int32_t> named;      // load_materials, scene_loader.h:173-188
    int32_t fallback_grey = -1;

    int32_t add_material(const bdpt_material& m) {
        out.materials.push_back(m);
        return (int32_t)out.materials.size() - 1;
    }
    // add_triangle_with_lights, scene_loader.h:190-202
    void add_triangle(V3 a, V3 b, V3 c, int32_t mat) {
        const double v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
        out.vertices.insert(out.vertices.end(), v, v + 9);
        out.tri_material.push_back(mat);
        if (out.materials[mat].type == BDPT_DIFFUSE_LIGHT) out.light_triangles.push_back((uint32_t)out.tri_material.size() - 1);
    }
    // material given by name or inline, else light grey (scene_loader.h:318-329, 411-422)
    int32_t resolve_material(const YamlNode& mat_node) {
        if (mat_node.defined()) {
            if (mat_node.is_scalar()) {
                auto it = named.find(mat_node.scalar);
                if (it != named.end()) return it->second;
            } else if (mat_node.is_map()) {
                return add_material(build_material(mat_node));
            }
        }
        if (fallback_grey < 0) fallback_grey = add_material(make_material(BDPT_LAMBERTIAN, V3{0.8, 0.8, 0.8}, 0.0, 0.0));
        return fallback_grey;
    }

    void load_materials(const YamlNode& node) {
        if (!node.is_map()) return;
        for (const auto& e : node.entries) {
            try {
                const bdpt_material m = build_material(e.second);
                auto it = named.find(e.first);
                if (it != named.end()) out.materials[it->second] = m;     // a repeated name replaces the earlier entry
                else named[e.first] = add_material(m);
            } catch (...) {
                // invalid material entries are skipped
            }
        }
    }

    // load_tri_mesh, scene_loader.h:244-272
    void load_tri_mesh(const YamlNode& mesh) {
        const YamlNode& data = mesh["data"];
        if (!data.is_map()) throw std::runtime_error("Mesh missing data field");
        const YamlNode& vertices_node = data["vertices"];
        if (!vertices_node.is_sequence()) throw std::runtime_error("Missing vertices");
        const std::vector<double> v = node_to_double_list(vertices_node);
        if (v.size() % 9 != 0) throw std::runtime_error("Vertices length not a multiple of 9");
        if (!mesh["material"].defined()) throw std::runtime_error("Missing material field");
        const int32_t mat = add_material(build_material(mesh["material"]));
        for (size_t i = 0; i < v.size(); i += 9)
            add_triangle(V3{v[i], v[i + 1], v[i + 2]}, V3{v[i + 3], v[i + 4], v[i + 5]}, V3{v[i + 6], v[i + 7], v[i + 8]}, mat);
    }

    static V3 sphere_point(double theta, double phi, V3 c, double radius) {       // scene_loader.h:204-210
        const double sin_theta = std::sin(theta);
        const double x = radius * sin_theta * std::cos(phi);
        const double y = radius * std::cos(theta);
        const double z = radius * sin_theta * std::sin(phi);
        return V3{c.x + x, c.y + y, c.z + z};
    }
    // load_sphere + add_uv_sphere, scene_loader.h:212-242,274-294
    void load_sphere(const YamlNode& mesh) {
        if (!mesh["material"].defined()) throw std::runtime_error("Missing material field");
        const int32_t mat = add_material(build_material(mesh["material"]));
        const YamlNode& data = mesh["data"];
        if (!data.is_map()) throw std::runtime_error("Missing data field");
        const V3 center = read_vec3(data["center"], V3{0, 0, 0});
        const double radius = node_to_double(data["radius"], 0.0);
        if (radius <= 0.0) throw std::runtime_error("Missing or invalid radius field");
        const double pi = 3.14159265358979323846;
        const int lat_steps = 16, lon_steps = 32;
        for (int lat = 0; lat < lat_steps; ++lat) {
            const double theta0 = pi * (static_cast<double>(lat) / lat_steps);
            const double theta1 = pi * (static_cast<double>(lat + 1) / lat_steps);
            for (int lon = 0; lon < lon_steps; ++lon) {
                const double phi0 = 2.0 * pi * (static_cast<double>(lon) / lon_steps);
                const double phi1 = 2.0 * pi * (static_cast<double>(lon + 1) / lon_steps);
                const V3 p00 = sphere_point(theta0, phi0, center, radius);
                const V3 p01 = sphere_point(theta0, phi1, center, radius);
                const V3 p10 = sphere_point(theta1, phi0, center, radius);
                const V3 p11 = sphere_point(theta1, phi1, center, radius);
                if (lat > 0) add_triangle(p00, p10, p11, mat);
                if (lat < lat_steps - 1) add_triangle(p00, p11, p01, mat);
            }
        }
    }

    // load_indexed_mesh, scene_loader.h:297-343
    void load_indexed_mesh(const YamlNode& mesh) {
        const YamlNode& verts_node = mesh["vertices"];
        const YamlNode& tris_node = mesh["triangles"];
        if (!verts_node.is_sequence()) throw std::runtime_error("Indexed mesh missing vertices");
        if (!tris_node.is_sequence()) throw std::runtime_error("Indexed mesh missing triangles");
        std::vector<V3> verts;
        for (const YamlNode& v : verts_node.items) {
            const std::vector<double> vals = node_to_double_list(v);
            if (vals.size() < 3) continue;
            verts.push_back(V3{vals[0], vals[1], vals[2]});
        }
        const int32_t mat = resolve_material(mesh["material"]);
        for (const YamlNode& tri : tris_node.items) {
            std::vector<int> idx;
            if (tri.is_sequence())
                for (const YamlNode& t : tri.items) idx.push_back(node_to_int(t, 0));
            if (idx.size() < 3) continue;
            add_triangle(verts.at(static_cast<size_t>(idx[0])), verts.at(static_cast<size_t>(idx[1])),
                         verts.at(static_cast<size_t>(idx[2])), mat);       // .at(): out-of-range indices are an error
        }
    }

    // load_obj_file, scene_loader.h:346-397: lines starting with "v" and "f" only
    void load_obj_file(const std::filesystem::path& obj_path, int32_t mat) {
        std::ifstream in(obj_path);
        if (!in) throw std::runtime_error("Failed to open OBJ file: " + obj_path.string());
        std::vector<V3> verts;
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::istringstream ss(line);
            std::string tag;
            ss >> tag;
            if (tag == "v") {
                double x, y, z;
                if (ss >> x >> y >> z) verts.push_back(V3{x, y, z});
            } else if (tag == "f") {
                std::vector<int> fidx;
                std::string tok;
                while (ss >> tok) {
                    const size_t slash = tok.find('/');      // vi | vi/vt | vi/vt/vn | vi//vn
                    const std::string vi_str = (slash == std::string::npos) ? tok : tok.substr(0, slash);
                    try {
                        const int vi = std::stoi(vi_str);
                        fidx.push_back(vi > 0 ? vi - 1 : static_cast<int>(verts.size()) + vi);     // 1-based; negative = relative
                    } catch (...) {
                        // malformed index: skipped
                    }
                }
                for (size_t k = 2; k < fidx.size(); ++k)     // fan triangulation
                    add_triangle(verts.at(static_cast<size_t>(fidx[0])), verts.at(static_cast<size_t>(fidx[k - 1])),
                                 verts.at(static_cast<size_t>(fidx[k])), mat);
            }
        }
    }
    // load_object, scene_loader.h:399-425
    void load_object(const YamlNode& node, const std::filesystem::path& yaml_dir) {
        const std::string file_rel = node_to_string(node["file"]);
        if (file_rel.empty()) throw std::runtime_error("Object missing file field");
        const int32_t mat = resolve_material(node["material"]);
        load_obj_file(yaml_dir / file_rel, mat);
    }

    // load_camera_from_yaml, scene_loader.h:427-476, on the defaults of src/camera.h:26-41
    void load_camera(const YamlNode& node) {
        bdpt_camera_public& cam = out.camera;
        cam = bdpt_camera_public{};
        cam.aspect_ratio = 1.0; cam.image_width = 100; cam.samples_per_pixel = 50; cam.max_depth = 10;
        cam.vfov = 90; cam.lookat[2] = -1; cam.vup[1] = 1; cam.defocus_angle = 0; cam.focus_dist = 10;
        out.output_name = "image.png";
        if (!node.is_map()) throw std::runtime_error("Camera section must be a mapping");
        const YamlNode& resolution = node["resolution"];
        if (!resolution.is_sequence() || resolution.size() < 2) throw std::runtime_error("Camera missing resolution");
        const std::vector<double> res = node_to_double_list(resolution);
        const int width = static_cast<int>(res[0]), height = static_cast<int>(res[1]);
        if (width <= 0 || height <= 0) throw std::runtime_error("Resolution values must be positive");
        cam.image_width = width;
        cam.aspect_ratio = static_cast<double>(width) / static_cast<double>(height);
        double vfov_deg = cam.vfov;
        if (node["vfov"].defined() || node["fov"].defined()) vfov_deg = node_to_double(node["vfov"], node_to_double(node["fov"], vfov_deg));
        cam.vfov = std::clamp(vfov_deg, 1.0, 179.0);
        cam.focus_dist = node_to_double(node["focus_distance"], cam.focus_dist);
        cam.defocus_angle = 0.0;                              // defocus blur is always off
        const V3 from = read_vec3(node["location"], V3{cam.lookfrom[0], cam.lookfrom[1], cam.lookfrom[2]});
        const V3 at = read_vec3(node["look_at"], V3{cam.lookat[0], cam.lookat[1], cam.lookat[2]});
        const V3 up = read_vec3(node["up"], V3{cam.vup[0], cam.vup[1], cam.vup[2]});
        const V3 bg = read_vec3(node["background"], V3{cam.background[0], cam.background[1], cam.background[2]});
        cam.lookfrom[0] = from.x; cam.lookfrom[1] = from.y; cam.lookfrom[2] = from.z;
        cam.lookat[0] = at.x; cam.lookat[1] = at.y; cam.lookat[2] = at.z;
        cam.vup[0] = up.x; cam.vup[1] = up.y; cam.vup[2] = up.z;
        cam.background[0] = bg.x; cam.background[1] = bg.y; cam.background[2] = bg.z;
        cam.samples_per_pixel = node_to_int(node["samples_per_pixel"], cam.samples_per_pixel);
        cam.max_depth = node_to_int(node["max_depth"], cam.max_depth);
        const std::string output = node_to_string(node["output"]);
        if (!output.empty()) out.output_name = output;
    }

    // load_scene_from_yaml, scene_loader.h:480-523
    void load(const std::string& path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("bad file: " + path);                 // YAML::LoadFile throws BadFile
        std::stringstream buffer;
        buffer << in.rdbuf();
        YamlNode root;
        std::string err;
        if (!bdpt_host::yaml_parse(buffer.str(), root, err)) throw std::runtime_error("yaml: " + err);
        if (!root.is_map()) throw std::runtime_error("Scene root must be a mapping");
        load_camera(root["camera"]);
        load_materials(root["materials"]);
        const YamlNode* surfaces = &root["surfaces"];
        if (!surfaces->defined()) surfaces = &root["scene"];
        if (!surfaces->is_sequence()) throw std::runtime_error("Scene/surfaces field missing or not a sequence");
        const std::filesystem::path yaml_dir = std::filesystem::path(path).parent_path();
        for (const YamlNode& mesh : surfaces->items) {
            if (!mesh.is_map()) throw std::runtime_error("Scene entries must be mappings");
            const std::string mesh_type = node_to_string(mesh["type"]);
            if (mesh_type.empty()) throw std::runtime_error("Mesh missing type field");
            if (mesh_type == "TriMesh") load_tri_mesh(mesh);
            else if (mesh_type == "Sphere") load_sphere(mesh);
            else if (mesh_type == "mesh") load_indexed_mesh(mesh);
            else if (mesh_type == "object") load_object(mesh, yaml_dir);
            else std::cerr << "Unknown mesh type: " << mesh_type << "\n";
        }
    }
};

uint32_t be32(uint32_t v) { return ((v & 0xFFu) << 24) | ((v & 0xFF00u) << 8) | ((v >> 8) & 0xFF00u) | (v >> 24); }

void png_chunk(std::FILE* fp, const char type[4], const unsigned char* data, size_t n) {
    const uint32_t len = be32((uint32_t)n);
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    const uint32_t crc_be = be32((uint32_t)crc);
    if (std::fwrite(&len, 4, 1, fp) != 1 || std::fwrite(type, 1, 4, fp) != 4 || (n && std::fwrite(data, 1, n, fp) != n) ||
        std::fwrite(&crc_be, 4, 1, fp) != 1)
        throw std::runtime_error("write_png: write failed");
}

}  // namespace

extern "C" {

int bdpt_scene_file_load(const char* yaml_path, bdpt_scene_file** out) {
    if (!yaml_path || !out) { g_file_error = "null argument"; return BDPT_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    bdpt_scene_file* f = new bdpt_scene_file();
    try {
        Loader loader{*f};
        loader.load(yaml_path);
    } catch (const std::exception& ex) {
        g_file_error = ex.what();
        delete f;
        return BDPT_ERR_INVALID_ARGUMENT;
    }
    std::cout << "Triangles: " << f->tri_material.size() << "\n";           // scene_loader.h:521
    *out = f;
    return BDPT_OK;
}

int bdpt_scene_file_destroy(bdpt_scene_file* file) {
    delete file;
    return BDPT_OK;
}

int bdpt_scene_file_get(const bdpt_scene_file* file, bdpt_scene_file_view* out) {
    if (!file || !out) { g_file_error = "null argument"; return BDPT_ERR_INVALID_ARGUMENT; }
    out->camera = file->camera;
    out->output_name = file->output_name.c_str();
    out->tri_vertices = file->vertices.data();
    out->tri_material = file->tri_material.data();
    out->n_triangles = (uint32_t)file->tri_material.size();
    out->materials = file->materials.data();
    out->n_materials = (uint32_t)file->materials.size();
    out->light_triangles = file->light_triangles.data();
    out->n_lights = (uint32_t)file->light_triangles.size();
    return BDPT_OK;
}

// Test hook: the parse tree of a YAML text as JSON (scalars as strings, null as null).
static void yaml_json(const YamlNode& n, std::string& o) {
    auto quote = [&o](const std::string& t) {
        o.push_back('"');
        for (unsigned char c : t) {
            if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
            else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o.push_back((char)c);
        }
        o.push_back('"');
    };
    switch (n.kind) {
        case YamlNode::Scalar: quote(n.scalar); break;
        case YamlNode::Sequence:
            o.push_back('[');
            for (size_t i = 0; i < n.items.size(); i++) { if (i) o.push_back(','); yaml_json(n.items[i], o); }
            o.push_back(']');
            break;
        case YamlNode::Map:
            o.push_back('{');
            for (size_t i = 0; i < n.entries.size(); i++) {
                if (i) o.push_back(',');
                quote(n.entries[i].first);
                o.push_back(':');
                yaml_json(n.entries[i].second, o);
            }
            o.push_back('}');
            break;
        default: o += "null";
    }
}
int bdpt_debug_yaml_to_json(const char* text, char* out, uint64_t capacity) {
    if (!text || !out || capacity == 0) { g_file_error = "null argument"; return BDPT_ERR_INVALID_ARGUMENT; }
    YamlNode root;
    std::string err, json;
    if (!bdpt_host::yaml_parse(text, root, err)) { g_file_error = "yaml: " + err; return BDPT_ERR_INVALID_ARGUMENT; }
    yaml_json(root, json);
    if (json.size() + 1 > capacity) { g_file_error = "output buffer too small"; return BDPT_ERR_INVALID_ARGUMENT; }
    std::memcpy(out, json.c_str(), json.size() + 1);
    return BDPT_OK;
}

int bdpt_write_png(const char* file_name, const uint8_t* rgb, int32_t width, int32_t height) {
    if (!file_name || !rgb || width <= 0 || height <= 0) { g_file_error = "write_png: bad argument"; return BDPT_ERR_INVALID_ARGUMENT; }
    std::FILE* fp = nullptr;
    try {
        const std::filesystem::path out_dir = "output";                      // wpng.h:45-49
        std::filesystem::create_directories(out_dir);
        const std::filesystem::path out_path = out_dir / file_name;
        fp = std::fopen(out_path.string().c_str(), "wb");
        if (!fp) throw std::runtime_error("write_png: cannot open output file");
        // 8-bit RGB, no interlace; every scanline with filter type 0
        const size_t row = (size_t)width * 3;
        std::vector<unsigned char> raw((row + 1) * (size_t)height);
        for (int32_t j = 0; j < height; j++) {
            raw[(row + 1) * j] = 0;
            std::memcpy(&raw[(row + 1) * j + 1], rgb + row * j, row);
        }
        uLongf zlen = compressBound((uLong)raw.size());
        std::vector<unsigned char> z(zlen);
        if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK)
            throw std::runtime_error("write_png: deflate failed");
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        if (std::fwrite(sig, 1, 8, fp) != 8) throw std::runtime_error("write_png: write failed");
        unsigned char ihdr[13];
        const uint32_t w_be = be32((uint32_t)width), h_be = be32((uint32_t)height);
        std::memcpy(ihdr, &w_be, 4);
        std::memcpy(ihdr + 4, &h_be, 4);
        ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
        png_chunk(fp, "IHDR", ihdr, 13);
        png_chunk(fp, "IDAT", z.data(), zlen);
        png_chunk(fp, "IEND", nullptr, 0);
        std::fclose(fp);
        return BDPT_OK;
    } catch (const std::exception& ex) {
        if (fp) std::fclose(fp);
        g_file_error = ex.what();
        return BDPT_ERR_INVALID_ARGUMENT;
    }
}

}  // extern "C"
