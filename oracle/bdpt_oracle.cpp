// TEST INFRASTRUCTURE ONLY - the oracle.  Never linked into or called by the product library.
//
// CPU restatement of the reference renderer's per-pixel radiance-estimation path
// (teehee567/Bidirectional-Path-Tracer, src/camera.h:111-129,160-223,245-253,294-475 and
// everything those lines call) on flat tables: no virtual calls, no shared_ptr, explicit
// RNG draw order, every floating-point expression written in the reference's evaluation
// order (compile with -ffp-contract=off).  Each function cites the reference lines it follows.
//
// Pinning: tests/test_oracle_vs_reference.py compares this file bit for bit with golden
// vectors produced by the reference's own headers (oracle/_ref, see oracle/Makefile and
// tests/golden/make_golden.py).
//
// Deliberate differences from the reference, both documented where they occur:
//   * random numbers come from per-(pixel,sample) counter streams (oracle_rng.h); the serial
//     mt19937 of src/main.h:28-32 is kept as a second engine for pinning only;
//   * sin/cos/pow(.,5) are the correctly rounded oracle_math.h versions instead of glibc's
//     (which are 1 ulp off in ~0.14 % of calls and differ between CPU models).
#include "bdpt_oracle.h"
#include "flat_scene.h"
#include "oracle_math.h"
#include "oracle_rng.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

const double kInfinity = std::numeric_limits<double>::infinity();
const double kPi = 3.1415926535897932385;   // src/main.h:20

// ---------------------------------------------------------------- vec3 (src/core/vec3.h)
struct V3 { double x, y, z; };

inline V3 mk(double a, double b, double c) { V3 v; v.x = a; v.y = b; v.z = c; return v; }
inline V3 neg(V3 v) { return mk(-v.x, -v.y, -v.z); }                                   // vec3.h:18
inline V3 add(V3 u, V3 v) { return mk(u.x + v.x, u.y + v.y, u.z + v.z); }              // vec3.h:73-75
inline V3 sub(V3 u, V3 v) { return mk(u.x - v.x, u.y - v.y, u.z - v.z); }              // vec3.h:77-79
inline V3 mulv(V3 u, V3 v) { return mk(u.x * v.x, u.y * v.y, u.z * v.z); }             // vec3.h:81-83
inline V3 scl(double t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }                // vec3.h:85-91
inline V3 divs(V3 v, double t) { return scl(1 / t, v); }                               // vec3.h:93-95
inline double dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }            // vec3.h:97-101
inline V3 cross(V3 u, V3 v) {                                                          // vec3.h:103-107
    return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline double len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                 // vec3.h:44-46
inline double len(V3 v) { return std::sqrt(len2(v)); }                                 // vec3.h:40-42
inline V3 unit(V3 v) { return divs(v, len(v)); }                                       // vec3.h:109-111
inline double comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

inline V3 reflect(V3 v, V3 n) { return sub(v, scl(2 * dot(v, n), n)); }                // vec3.h:138-140

inline V3 refract(V3 uv, V3 n, double etai_over_etat) {                                // vec3.h:142-147
    double cos_theta = std::fmin(dot(neg(uv), n), 1.0);
    V3 r_out_perp = scl(etai_over_etat, add(uv, scl(cos_theta, n)));
    V3 r_out_parallel = scl(-std::sqrt(std::fabs(1.0 - len2(r_out_perp))), n);
    return add(r_out_perp, r_out_parallel);
}

// ---------------------------------------------------------------- RNG (src/main.h:28-42)
struct Rng {
    int engine;          // 0 counter stream, 1 serial mt19937
    orng_stream st;
    orng_mt19937* mt;
    uint64_t total;
    double next() {
        total++;
        if (engine == 1) return orng_mt_next_double(mt);
        return orng_stream_next(&st);
    }
    double next(double mn, double mx) { return mn + (mx - mn) * next(); }   // main.h:34-37
};

// vec3.h:121-128.  `vec3::random(-1,1)` builds vec3(rd, rd, rd) (vec3.h:58-60); the order in
// which the three draws land in x, y, z is unspecified in C++.  The pinned oracle compiler
// (g++) evaluates the constructor arguments right to left: the FIRST draw is z, then y, then x.
inline V3 random_unit_vector(Rng& rng) {
    for (;;) {
        double pz = rng.next(-1, 1);
        double py = rng.next(-1, 1);
        double px = rng.next(-1, 1);
        V3 p = mk(px, py, pz);
        double lensq = len2(p);
        if (1e-160 < lensq && lensq <= 1) return divs(p, std::sqrt(lensq));
    }
}

// vec3.h:149-159 (sin/cos: oracle_math.h instead of glibc)
inline V3 random_cosine_direction(Rng& rng) {
    double r1 = rng.next();
    double r2 = rng.next();
    double phi = 2 * kPi * r1;
    double s, c;
    oracle_sincos(phi, &s, &c);
    double x = c * std::sqrt(r2);
    double y = s * std::sqrt(r2);
    double z = std::sqrt(1 - r2);
    return mk(x, y, z);
}

// src/acceleration/onb.h:6-20
struct Onb { V3 a0, a1, a2; };
inline Onb make_onb(V3 n) {
    Onb b;
    b.a2 = unit(n);
    V3 a = (std::fabs(b.a2.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    b.a1 = unit(cross(b.a2, a));
    b.a0 = cross(b.a2, b.a1);
    return b;
}
inline V3 onb_transform(const Onb& b, V3 v) {
    return add(add(scl(v.x, b.a0), scl(v.y, b.a1)), scl(v.z, b.a2));
}

// ---------------------------------------------------------------- geometry tables
struct Box { double mn[3], mx[3]; };     // aabb: three intervals (src/acceleration/aabb.h:8)

struct Tri {                              // src/objects/primatives/triangle.h:21-39,124-132
    V3 v0, v1, v2, edge1, edge2, normal;
    double area;
    Box bbox;
    int32_t mat;
};

struct Mat { int32_t type; V3 rgb; double fuzz; double ior; int32_t tex; };   // tex >= 0: a texture replaces rgb (material.h:45,135,152)

// child >= 0: node index; < 0: ~code of a leaf hittable, code < #triangles: that triangle, otherwise the object
// code - #triangles of a composite scene
struct Node { Box box; int32_t left, right; };

// src/materials/textures/texture.h:14-87
struct Texture { int32_t type; V3 rgb; double scale; int32_t even, odd, perlin, width, height; size_t bytes; };
// what a perlin object holds (perlin.h:53-57)
struct Perlin { V3 randvec[256]; int perm_x[256], perm_y[256], perm_z[256]; };
// One object of a composite world's list: translate(rotate_y(bvh_node(mesh))), optionally the boundary of a
// constant_medium (src/objects/hittable.h:46-157, src/materials/volumes/constant_medium.h)
struct Object {
    uint32_t first, count, flags;        // flags: 1 rotate_y, 2 translate, 4 constant_medium
    int32_t phase;                       // the medium's isotropic material
    double sin_theta, cos_theta;
    V3 offset;
    double neg_inv_density;
    int32_t root;                        // root of the mesh's own bvh_node tree
    Box bbox;                            // bounding_box() of the outermost wrapper
};

struct Stats {                            // src/core/stats.h:8-16 (+ three extra counters)
    uint64_t rays_traced, aabb_tests, aabb_hits, bvh_node_visits, triangle_tests, triangle_hits, bvh_nodes_built;
    uint64_t shadow_rays, pairs;
};

struct Ray { V3 o, d; };

struct Hit {                              // src/objects/hittable.h:10-27 (+ triangle id; inside the medium k: -(k + 2))
    V3 p, normal;
    int32_t mat, tri;
    double t, u, v;
    bool front_face;
};

inline void pad_axis(double& mn, double& mx) {   // aabb.h:81-88 + interval.h:36-39
    const double delta = 0.0001;
    if (mx - mn < delta) {
        double padding = delta / 2;
        mn = mn - padding;
        mx = mx + padding;
    }
}

inline Box box_merge(const Box& a, const Box& b) {   // aabb.h:27-33 + interval.h:12-16
    Box r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = a.mn[k] <= b.mn[k] ? a.mn[k] : b.mn[k];
        r.mx[k] = a.mx[k] >= b.mx[k] ? a.mx[k] : b.mx[k];
        pad_axis(r.mn[k], r.mx[k]);
    }
    return r;
}

inline Box box_empty() {                  // aabb.h:92 (padding an empty interval leaves it empty)
    Box r;
    for (int k = 0; k < 3; k++) { r.mn[k] = kInfinity; r.mx[k] = -kInfinity; pad_axis(r.mn[k], r.mx[k]); }
    return r;
}

inline int longest_axis(const Box& b) {   // aabb.h:68-75
    double sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

inline Tri make_tri(V3 v0, V3 v1, V3 v2, int32_t mat) {   // triangle.h:21-39
    Tri t;
    t.v0 = v0; t.v1 = v1; t.v2 = v2; t.mat = mat;
    t.edge1 = sub(v1, v0);
    t.edge2 = sub(v2, v0);
    V3 n = cross(t.edge1, t.edge2);
    t.normal = unit(n);
    double double_area = len(n);
    t.area = 0.5 * double_area;
    double lo[3] = { std::fmin(v0.x, std::fmin(v1.x, v2.x)), std::fmin(v0.y, std::fmin(v1.y, v2.y)),
                     std::fmin(v0.z, std::fmin(v1.z, v2.z)) };
    double hi[3] = { std::fmax(v0.x, std::fmax(v1.x, v2.x)), std::fmax(v0.y, std::fmax(v1.y, v2.y)),
                     std::fmax(v0.z, std::fmax(v1.z, v2.z)) };
    for (int k = 0; k < 3; k++) {         // aabb.h:18-25 (no padding in this constructor)
        if (lo[k] <= hi[k]) { t.bbox.mn[k] = lo[k]; t.bbox.mx[k] = hi[k]; }
        else { t.bbox.mn[k] = hi[k]; t.bbox.mx[k] = lo[k]; }
    }
    return t;
}

}  // namespace

struct oracle_scene {
    std::vector<Tri> tris;
    std::vector<Mat> mats;
    std::vector<uint32_t> lights;        // the `lights` collection: world triangle ids, in order
    std::vector<Node> nodes;             // every tree in pre-order; nodes[root] is the root of the world's tree
    uint64_t nodes_built = 0;
    int32_t root = 0;
    // composite scenes only
    bool composite = false;
    std::vector<Texture> textures;
    std::vector<Perlin> perlins;
    std::vector<uint8_t> image_bytes;
    std::vector<Object> objects;
    std::vector<uint32_t> loose;         // triangles that are entries of the world's list themselves
};

namespace {

// src/acceleration/bvh.h:20-48
// `boxes[code]`: bounding_box() of the leaf hittable with that code
int build_node(oracle_scene& s, const std::vector<Box>& boxes, std::vector<uint32_t>& objects, size_t start, size_t end) {
    s.nodes_built++;
    int idx = (int)s.nodes.size();
    s.nodes.push_back(Node());

    Box bbox = box_empty();
    for (size_t i = start; i < end; i++) bbox = box_merge(bbox, boxes[objects[i]]);

    int axis = longest_axis(bbox);
    size_t span = end - start;
    int32_t left, right;
    if (span == 1) {
        left = right = ~(int32_t)objects[start];
    } else if (span == 2) {
        left = ~(int32_t)objects[start];
        right = ~(int32_t)objects[start + 1];
    } else {
        // bvh.h:68-74: strict < on the (unpadded) boxes' minimum along `axis`; same
        // libstdc++ std::sort call as the reference, so equal keys end up in the same order.
        std::sort(objects.begin() + start, objects.begin() + end,
                  [&boxes, axis](uint32_t a, uint32_t b) { return boxes[a].mn[axis] < boxes[b].mn[axis]; });
        size_t mid = start + span / 2;
        left = build_node(s, boxes, objects, start, mid);
        right = build_node(s, boxes, objects, mid, end);
    }
    s.nodes[idx].box = bbox;
    s.nodes[idx].left = left;
    s.nodes[idx].right = right;
    return idx;
}

void finish_scene(oracle_scene& s) {
    std::vector<uint32_t> objects(s.tris.size());
    std::vector<Box> boxes(s.tris.size());
    for (uint32_t i = 0; i < objects.size(); i++) { objects[i] = i; boxes[i] = s.tris[i].bbox; }   // triangle.h:191-197
    s.nodes.clear();
    s.nodes_built = 0;
    s.root = 0;
    if (!objects.empty()) build_node(s, boxes, objects, 0, objects.size());
}

// src/acceleration/aabb.h:41-66
inline bool box_hit(const Box& b, const Ray& r, double tmin, double tmax, Stats& st) {
    st.aabb_tests++;
    for (int axis = 0; axis < 3; axis++) {
        const double adinv = 1.0 / comp(r.d, axis);
        double t0 = (b.mn[axis] - comp(r.o, axis)) * adinv;
        double t1 = (b.mx[axis] - comp(r.o, axis)) * adinv;
        if (t0 < t1) {
            if (t0 > tmin) tmin = t0;
            if (t1 < tmax) tmax = t1;
        } else {
            if (t1 > tmin) tmin = t1;
            if (t0 < tmax) tmax = t0;
        }
        if (tmax <= tmin) return false;
    }
    st.aabb_hits++;
    return true;
}

// src/objects/primatives/triangle.h:41-74 + src/objects/hittable.h:20-26
inline bool tri_hit(const oracle_scene& s, int32_t id, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st) {
    st.triangle_tests++;
    const Tri& tr = s.tris[id];
    const double epsilon = 1e-8;
    V3 pvec = cross(r.d, tr.edge2);
    double det = dot(tr.edge1, pvec);
    if (std::fabs(det) < epsilon) return false;
    double inv_det = 1.0 / det;
    V3 tvec = sub(r.o, tr.v0);
    double u = dot(tvec, pvec) * inv_det;
    if (u < 0.0 || u > 1.0) return false;
    V3 qvec = cross(tvec, tr.edge1);
    double v = dot(r.d, qvec) * inv_det;
    if (v < 0.0 || u + v > 1.0) return false;
    double t = dot(tr.edge2, qvec) * inv_det;
    if (!(tmin <= t && t <= tmax)) return false;      // interval.h:22-24 (inclusive)
    rec.t = t;
    rec.p = add(r.o, scl(t, r.d));                    // ray.h:21-23
    rec.u = u;
    rec.v = v;
    rec.mat = tr.mat;
    rec.tri = id;
    rec.front_face = dot(r.d, tr.normal) < 0;
    rec.normal = rec.front_face ? tr.normal : neg(tr.normal);
    st.triangle_hits++;
    return true;
}

bool node_hit(const oracle_scene& s, int32_t idx, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st, Rng* rng);

// hit() of object k of a composite world's list.  `rng`: the stream constant_medium::hit draws from.
bool object_hit(const oracle_scene& s, int32_t k, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st, Rng* rng) {
    const Object& ob = s.objects[k];
    // boundary->hit(...) through translate::hit (hittable.h:54-66) and rotate_y::hit (:110-148)
    auto boundary_hit = [&](double lo, double hi, Hit& out) {
        Ray q = r;
        if (ob.flags & 2u) q.o = sub(q.o, ob.offset);
        if (ob.flags & 1u) {
            const double c = ob.cos_theta, sn = ob.sin_theta;
            q.o = mk((c * q.o.x) - (sn * q.o.z), q.o.y, (sn * q.o.x) + (c * q.o.z));
            q.d = mk((c * q.d.x) - (sn * q.d.z), q.d.y, (sn * q.d.x) + (c * q.d.z));
        }
        if (!node_hit(s, ob.root, q, lo, hi, out, st, rng)) return false;
        if (ob.flags & 1u) {
            const double c = ob.cos_theta, sn = ob.sin_theta;
            out.p = mk((c * out.p.x) + (sn * out.p.z), out.p.y, (-sn * out.p.x) + (c * out.p.z));
            out.normal = mk((c * out.normal.x) + (sn * out.normal.z), out.normal.y, (-sn * out.normal.x) + (c * out.normal.z));
        }
        if (ob.flags & 2u) out.p = add(out.p, ob.offset);
        return true;
    };
    if (!(ob.flags & 4u)) return boundary_hit(tmin, tmax, rec);
    // constant_medium::hit, constant_medium.h:20-53
    Hit rec1, rec2;
    if (!boundary_hit(-kInfinity, kInfinity, rec1)) return false;
    if (!boundary_hit(rec1.t + 0.0001, kInfinity, rec2)) return false;
    if (rec1.t < tmin) rec1.t = tmin;
    if (rec2.t > tmax) rec2.t = tmax;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < 0) rec1.t = 0;
    double ray_length = len(r.d);
    double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    double hit_distance = ob.neg_inv_density * oracle_log(rng->next());
    if (hit_distance > distance_inside_boundary) return false;
    rec.t = rec1.t + hit_distance / ray_length;
    rec.p = add(r.o, scl(rec.t, r.d));
    rec.normal = mk(1, 0, 0);
    rec.front_face = true;
    rec.mat = ob.phase;
    rec.tri = -(k + 2);
    rec.u = 0; rec.v = 0;                 // (the reference leaves them unset; nothing a medium can carry reads them)
    return true;
}

inline bool leaf_hit(const oracle_scene& s, int32_t code, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st, Rng* rng) {
    if ((size_t)code < s.tris.size()) return tri_hit(s, code, r, tmin, tmax, rec, st);
    return object_hit(s, code - (int32_t)s.tris.size(), r, tmin, tmax, rec, st, rng);
}

// src/acceleration/bvh.h:50-59
bool node_hit(const oracle_scene& s, int32_t idx, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st, Rng* rng) {
    st.bvh_node_visits++;
    const Node& n = s.nodes[idx];
    if (!box_hit(n.box, r, tmin, tmax, st)) return false;
    bool hit_left = n.left >= 0 ? node_hit(s, n.left, r, tmin, tmax, rec, st, rng)
                                : leaf_hit(s, ~n.left, r, tmin, tmax, rec, st, rng);
    double rmax = hit_left ? rec.t : tmax;
    bool hit_right = n.right >= 0 ? node_hit(s, n.right, r, tmin, rmax, rec, st, rng)
                                  : leaf_hit(s, ~n.right, r, tmin, rmax, rec, st, rng);
    return hit_left || hit_right;
}

inline bool world_hit(const oracle_scene& s, const Ray& r, double tmin, double tmax, Hit& rec, Stats& st, Rng* rng) {
    return node_hit(s, s.root, r, tmin, tmax, rec, st, rng);
}

// ---------------------------------------------------------------- textures (src/materials/textures)
// perlin::perlin, perlin.h:6-14,59-73: unit_vector(vec3::random(-1, 1)) fills z, y, x in that order (see
// random_unit_vector above); random_int(0, i) = int(random_double(0, i + 1)), main.h:39-42
void make_perlin(Perlin& pn, Rng& rng) {
    for (int i = 0; i < 256; i++) {
        double pz = rng.next(-1, 1);
        double py = rng.next(-1, 1);
        double px = rng.next(-1, 1);
        pn.randvec[i] = unit(mk(px, py, pz));
    }
    int* perms[3] = { pn.perm_x, pn.perm_y, pn.perm_z };
    for (int* p : perms) {
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i > 0; i--) {
            int target = int(rng.next(0, i + 1));
            int tmp = p[i];
            p[i] = p[target];
            p[target] = tmp;
        }
    }
}
// perlin::noise + perlin_interp, perlin.h:16-36,75-92
double perlin_noise(const Perlin& pn, V3 p) {
    double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
    int i = int(std::floor(p.x)), j = int(std::floor(p.y)), k = int(std::floor(p.z));
    V3 c[2][2][2];
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++)
                c[di][dj][dk] = pn.randvec[pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255]];
    double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int q = 0; q < 2; q++) {
                V3 weight_v = mk(u - a, v - b, w - q);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (q * ww + (1 - q) * (1 - ww)) *
                         dot(c[a][b][q], weight_v);
            }
    return accum;
}
double perlin_turb(const Perlin& pn, V3 p, int depth) {      // perlin.h:38-50
    double accum = 0.0;
    V3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = mk(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2);
    }
    return std::fabs(accum);
}
V3 texture_value(const oracle_scene& s, int32_t id, double u, double v, V3 p) {
    const Texture& t = s.textures[id];
    switch (t.type) {
        case 1: {                                            // checker_texture::value, texture.h:36-44 (scale = inv_scale)
            int xi = int(std::floor(t.scale * p.x)), yi = int(std::floor(t.scale * p.y)), zi = int(std::floor(t.scale * p.z));
            bool is_even = (xi + yi + zi) % 2 == 0;
            return texture_value(s, is_even ? t.even : t.odd, u, v, p);
        }
        case 2: {                                            // noise_texture::value, texture.h:80-82
            double sn, cs;
            oracle_sincos(t.scale * p.z + 10 * perlin_turb(s.perlins[t.perlin], p, 7), &sn, &cs);
            return scl(1 + sn, mk(.5, .5, .5));
        }
        case 3: {                                            // image_texture::value, texture.h:56-70; rtw_stb_image.h:63-88
            if (t.height <= 0) return mk(0, 1, 1);
            u = u < 0 ? 0 : (u > 1 ? 1 : u);
            v = 1.0 - (v < 0 ? 0 : (v > 1 ? 1 : v));
            int i = int(u * t.width), j = int(v * t.height);
            i = i < 0 ? 0 : (i < t.width ? i : t.width - 1);
            j = j < 0 ? 0 : (j < t.height ? j : t.height - 1);
            const uint8_t* px = &s.image_bytes[t.bytes + ((size_t)j * t.width + i) * 3];
            double color_scale = 1.0 / 255.0;
            return mk(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
        }
        default: return t.rgb;                               // solid_color
    }
}
inline V3 mat_colour(const oracle_scene& s, const Mat& m, const Hit& rec) {
    return m.tex >= 0 ? texture_value(s, m.tex, rec.u, rec.v, rec.p) : m.rgb;
}

// ---------------------------------------------------------------- materials (src/materials/material.h)
inline V3 mat_emitted(const oracle_scene& s, const Mat& m, const Hit& rec) {       // material.h:20-24,138-143
    if (m.type == ORACLE_DIFFUSE_LIGHT) return rec.front_face ? mat_colour(s, m, rec) : mk(0, 0, 0);
    return mk(0, 0, 0);
}
inline bool mat_is_delta(const Mat& m) { return m.type == ORACLE_METAL || m.type == ORACLE_DIELECTRIC; }  // :85,:118
inline V3 mat_evaluate_bsdf(const oracle_scene& s, const Mat& m, const Hit& rec) {   // material.h:35-37,60-63,166-168
    if (m.type == ORACLE_LAMBERTIAN) return divs(mat_colour(s, m, rec), kPi);
    if (m.type == ORACLE_ISOTROPIC) return divs(mat_colour(s, m, rec), 4 * kPi);
    return mk(0, 0, 0);
}

inline double reflectance(double cosine, double refraction_index) {   // material.h:125-130
    double r0 = (1 - refraction_index) / (1 + refraction_index);
    r0 = r0 * r0;
    return r0 + (1 - r0) * oracle_pow5(1 - cosine);
}

// ---------------------------------------------------------------- integrator (src/camera.h)
struct Camera {                 // the private state camera::initialize computes (camera.h:148-158)
    V3 center, pixel00_loc, pixel_delta_u, pixel_delta_v, background;
    V3 defocus_disk_u, defocus_disk_v;
    double defocus_angle;
};

struct PathVertex {             // camera.h:236-243
    Hit rec;
    V3 throughput, wi, emission;
    bool delta, is_light;
};

// camera.h:325-370
void trace_path(const oracle_scene& s, const Camera& cam, Ray current, int depth, std::vector<PathVertex>& path,
                V3 throughput, V3* background_contrib, Rng& rng, Stats& st) {
    for (int bounce = 0; bounce < depth; ++bounce) {
        st.rays_traced++;
        Hit rec;
        if (!world_hit(s, current, 0.001, kInfinity, rec, st, &rng)) {
            if (background_contrib) *background_contrib = add(*background_contrib, mulv(throughput, cam.background));
            break;
        }
        const Mat& m = s.mats[rec.mat];
        PathVertex vertex;
        vertex.rec = rec;
        vertex.throughput = throughput;
        vertex.wi = unit(neg(current.d));
        vertex.emission = mat_emitted(s, m, rec);
        vertex.delta = mat_is_delta(m);
        vertex.is_light = (m.type == ORACLE_DIFFUSE_LIGHT);
        path.push_back(vertex);

        if (m.type == ORACLE_DIFFUSE_LIGHT) break;           // material.h:26-28: scatter() == false

        if (m.type == ORACLE_METAL) {                        // material.h:73-83, camera.h:355-359
            V3 reflected = reflect(current.d, rec.normal);
            reflected = add(unit(reflected), scl(m.fuzz, random_unit_vector(rng)));
            throughput = mulv(throughput, m.rgb);
            current.o = rec.p;
            current.d = reflected;
            continue;
        }
        if (m.type == ORACLE_DIELECTRIC) {                   // material.h:96-116, camera.h:355-359
            double ri = rec.front_face ? (1.0 / m.ior) : m.ior;
            V3 unit_direction = unit(current.d);
            double cos_theta = std::fmin(dot(neg(unit_direction), rec.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            bool cannot_refract = ri * sin_theta > 1.0;
            V3 direction;
            if (cannot_refract || reflectance(cos_theta, ri) > rng.next())   // short-circuit: no draw on TIR
                direction = reflect(unit_direction, rec.normal);
            else
                direction = refract(unit_direction, rec.normal, ri);
            throughput = mulv(throughput, mk(1.0, 1.0, 1.0));
            current.o = rec.p;
            current.d = direction;
            continue;
        }
        if (m.type == ORACLE_ISOTROPIC) {                    // material.h:154-164, pdf.h:17-28, camera.h:361-368
            V3 attenuation = mat_colour(s, m, rec);
            V3 scattered_dir = random_unit_vector(rng);
            double pdf_value = 1 / (4 * kPi);
            double scattering_pdf = 1 / (4 * kPi);
            throughput = divs(scl(scattering_pdf, mulv(throughput, attenuation)), pdf_value);
            current.o = rec.p;
            current.d = scattered_dir;
            continue;
        }
        // lambertian: material.h:47-58, pdf.h:32-47, camera.h:361-368
        Onb uvw = make_onb(rec.normal);
        V3 scattered_dir = onb_transform(uvw, random_cosine_direction(rng));
        double cosine_theta = dot(unit(scattered_dir), uvw.a2);
        double pdf_value = std::fmax(0, cosine_theta / kPi);
        if (pdf_value <= 0) break;
        double cos_theta = dot(rec.normal, unit(scattered_dir));
        double scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi;
        throughput = divs(scl(scattering_pdf, mulv(throughput, mat_colour(s, m, rec))), pdf_value);
        current.o = rec.p;
        current.d = scattered_dir;
    }
}

// triangle.h:199-224 + 107-119
bool sample_surface(const oracle_scene& s, const std::vector<uint32_t>& lights, Rng& rng, V3& position, V3& normal,
                    int32_t& mat, double& pdf) {
    if (lights.empty()) return false;
    double total_area = 0.0;
    for (uint32_t id : lights) total_area += s.tris[id].area;
    if (total_area <= 0.0) return false;
    double pick = rng.next(0.0, total_area);
    const Tri* chosen = &s.tris[lights.back()];
    double accum = 0.0;
    for (uint32_t id : lights) {
        accum += s.tris[id].area;
        if (pick <= accum) { chosen = &s.tris[id]; break; }
    }
    double u = rng.next();
    double v = rng.next();
    if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
    position = add(add(chosen->v0, scl(u, chosen->edge1)), scl(v, chosen->edge2));
    normal = chosen->normal;
    mat = chosen->mat;
    pdf = 1.0 / total_area;
    return true;
}

// camera.h:372-418
bool build_light_path(const oracle_scene& s, const Camera& cam, const std::vector<uint32_t>& lights, int depth,
                      std::vector<PathVertex>& path, Rng& rng, Stats& st) {
    if (depth <= 0 || lights.empty()) return false;
    V3 position, normal;
    int32_t mat;
    double pdf;
    if (!sample_surface(s, lights, rng, position, normal, mat, pdf)) return false;

    Hit light_rec;                                            // camera.h:385-391
    light_rec.p = position; light_rec.normal = normal; light_rec.mat = mat; light_rec.tri = -1;
    light_rec.front_face = true; light_rec.t = 0; light_rec.u = 0; light_rec.v = 0;
    V3 emission = mat_emitted(s, s.mats[mat], light_rec);
    if (len2(emission) <= 0) return false;

    PathVertex ev;
    ev.rec.p = position;
    ev.rec.normal = normal;
    ev.rec.mat = mat;
    ev.rec.tri = -1;
    ev.rec.front_face = true;
    ev.rec.t = 0; ev.rec.u = 0; ev.rec.v = 0;
    ev.throughput = divs(mk(1.0, 1.0, 1.0), std::max(pdf, 1e-8));
    ev.wi = normal;
    ev.emission = emission;
    ev.delta = false;
    ev.is_light = true;
    path.push_back(ev);

    Onb basis = make_onb(normal);                             // camera.h:420-423
    V3 dir = onb_transform(basis, random_cosine_direction(rng));
    V3 dir_unit = unit(dir);
    double cos_theta = std::max(0.0, dot(normal, dir_unit));
    if (cos_theta <= 0) return true;

    double pdf_dir = std::max(cos_theta / kPi, 1e-8);
    V3 throughput = scl(cos_theta / pdf_dir, mulv(ev.throughput, emission));
    Ray light_ray;
    light_ray.o = add(position, scl(0.001, normal));
    light_ray.d = dir_unit;
    trace_path(s, cam, light_ray, depth - 1, path, throughput, nullptr, rng, st);
    return true;
}

// camera.h:425-438
bool visible(const oracle_scene& s, V3 a, V3 b, Stats& st, Rng* rng) {
    V3 dir = sub(b, a);
    double dist = len(dir);
    if (dist <= 0) return false;
    V3 dir_unit = divs(dir, dist);
    Ray shadow;
    shadow.o = add(a, scl(0.001, dir_unit));
    shadow.d = dir_unit;
    double max_t = dist - 0.001;
    if (max_t <= 0) return false;
    st.shadow_rays++;
    Hit tmp;
    return !world_hit(s, shadow, 0.001, max_t, tmp, st, rng);
}

// camera.h:440-475
V3 connect_vertices(const oracle_scene& s, const PathVertex& cam_v, const PathVertex& light_v, Stats& st, Rng* rng) {
    if (cam_v.delta || light_v.delta) return mk(0, 0, 0);
    st.pairs++;
    V3 dir = sub(light_v.rec.p, cam_v.rec.p);
    double dist2 = len2(dir);
    if (dist2 <= 0) return mk(0, 0, 0);
    V3 dir_unit = unit(dir);
    double cos_cam = std::fabs(dot(cam_v.rec.normal, dir_unit));
    double cos_light = std::fabs(dot(light_v.rec.normal, neg(dir_unit)));
    if (cos_cam <= 0 || cos_light <= 0) return mk(0, 0, 0);
    if (!visible(s, cam_v.rec.p, light_v.rec.p, st, rng)) return mk(0, 0, 0);
    V3 f_cam = mat_evaluate_bsdf(s, s.mats[cam_v.rec.mat], cam_v.rec);
    if (len2(f_cam) <= 0) return mk(0, 0, 0);
    V3 f_light = light_v.is_light ? light_v.emission : mat_evaluate_bsdf(s, s.mats[light_v.rec.mat], light_v.rec);
    if (len2(f_light) <= 0) return mk(0, 0, 0);
    V3 contrib = mulv(cam_v.throughput, f_cam);
    contrib = mulv(mulv(contrib, light_v.throughput), f_light);
    contrib = divs(scl(cos_cam * cos_light, contrib), dist2);
    return contrib;
}

// camera.h:294-323
V3 bidirectional_color(const oracle_scene& s, const Camera& cam, const std::vector<uint32_t>& lights, const Ray& r,
                       int depth, Rng& rng, Stats& st, std::vector<PathVertex>& camera_path,
                       std::vector<PathVertex>& light_path) {
    camera_path.clear();
    light_path.clear();
    V3 background_contrib = mk(0, 0, 0);
    trace_path(s, cam, r, depth, camera_path, mk(1, 1, 1), &background_contrib, rng, st);

    V3 result = background_contrib;
    for (const PathVertex& v : camera_path)
        if (!v.delta && len2(v.emission) > 0) result = add(result, mulv(v.throughput, v.emission));

    if (!build_light_path(s, cam, lights, depth, light_path, rng, st)) return result;

    for (const PathVertex& cv : camera_path)
        for (const PathVertex& lv : light_path) result = add(result, connect_vertices(s, cv, lv, st, &rng));
    return result;
}

// triangle::pdf_value, triangle.h:78-91: a hit test of this one triangle (not of the world)
double tri_pdf_value(const oracle_scene& s, uint32_t id, V3 origin, V3 direction, Stats& st) {
    Hit rec;
    Ray r;
    r.o = origin; r.d = direction;
    if (!tri_hit(s, (int32_t)id, r, 0.001, kInfinity, rec, st)) return 0.0;
    const Tri& tr = s.tris[id];
    double distance_squared = rec.t * rec.t * len2(direction);
    double cosine = std::fabs(dot(direction, rec.normal) / len(direction));
    if (tr.area <= 0.0 || cosine <= 0.0) return 0.0;
    return distance_squared / (cosine * tr.area);
}

// triangle_collection::pdf_value, triangle.h:170-181
double lights_pdf_value(const oracle_scene& s, const std::vector<uint32_t>& lights, V3 origin, V3 direction, Stats& st) {
    if (lights.empty()) return 0.0;
    double weight = 1.0 / lights.size();
    double sum = 0.0;
    for (uint32_t id : lights) sum += weight * tri_pdf_value(s, id, origin, direction, st);
    return sum;
}

// triangle_collection::random + triangle::random, triangle.h:183-189,93-103; random_int, main.h:39-42
V3 lights_random(const oracle_scene& s, const std::vector<uint32_t>& lights, V3 origin, Rng& rng) {
    if (lights.empty()) return mk(1, 0, 0);
    int idx = int(rng.next(0, static_cast<int>(lights.size()) - 1 + 1));
    const Tri& tr = s.tris[lights[idx]];
    double u = rng.next();
    double v = rng.next();
    if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
    V3 p = add(add(tr.v0, scl(u, tr.edge1)), scl(v, tr.edge2));
    return sub(p, origin);
}

// camera::path_trace_color, camera.h:255-292 (mixture_pdf / hittable_pdf / cosine_pdf: pdf.h:32-90)
V3 path_trace_color(const oracle_scene& s, const Camera& cam, const std::vector<uint32_t>& lights, const Ray& r, int depth,
                    Rng& rng, Stats& st) {
    st.rays_traced++;
    if (depth <= 0) return mk(0, 0, 0);
    Hit rec;
    if (!world_hit(s, r, 0.001, kInfinity, rec, st, &rng)) return cam.background;
    const Mat& m = s.mats[rec.mat];
    V3 color_from_emission = mat_emitted(s, m, rec);
    if (m.type == ORACLE_DIFFUSE_LIGHT) return color_from_emission;              // scatter() == false
    if (m.type == ORACLE_METAL) {                                                // material.h:73-83
        V3 reflected = reflect(r.d, rec.normal);
        reflected = add(unit(reflected), scl(m.fuzz, random_unit_vector(rng)));
        Ray next;
        next.o = rec.p; next.d = reflected;
        return mulv(m.rgb, path_trace_color(s, cam, lights, next, depth - 1, rng, st));
    }
    if (m.type == ORACLE_DIELECTRIC) {                                           // material.h:96-116
        double ri = rec.front_face ? (1.0 / m.ior) : m.ior;
        V3 unit_direction = unit(r.d);
        double cos_theta = std::fmin(dot(neg(unit_direction), rec.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        V3 direction;
        if (cannot_refract || reflectance(cos_theta, ri) > rng.next())
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, ri);
        Ray next;
        next.o = rec.p; next.d = direction;
        return mulv(mk(1.0, 1.0, 1.0), path_trace_color(s, cam, lights, next, depth - 1, rng, st));
    }
    // lambertian / isotropic: mixture of the light pdf and the material's own pdf (cosine_pdf / sphere_pdf)
    const bool iso = (m.type == ORACLE_ISOTROPIC);
    Onb uvw = make_onb(rec.normal);
    V3 direction;
    if (rng.next() < 0.5) direction = lights_random(s, lights, rec.p, rng);
    else direction = iso ? random_unit_vector(rng) : onb_transform(uvw, random_cosine_direction(rng));
    double own = iso ? 1 / (4 * kPi) : std::fmax(0, dot(unit(direction), uvw.a2) / kPi);
    double pdf_value = 0.5 * lights_pdf_value(s, lights, rec.p, direction, st) + 0.5 * own;
    if (pdf_value <= 0) return color_from_emission;
    double scattering_pdf;
    if (iso) scattering_pdf = 1 / (4 * kPi);
    else { double cos_theta = dot(rec.normal, unit(direction)); scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi; }
    Ray scattered;
    scattered.o = rec.p; scattered.d = direction;
    V3 sample_color = path_trace_color(s, cam, lights, scattered, depth - 1, rng, st);
    V3 color_from_scatter = divs(mulv(scl(scattering_pdf, mat_colour(s, m, rec)), sample_color), pdf_value);
    return add(color_from_emission, color_from_scatter);
}

// camera.h:199-223; defocus_disk_sample camera.h:230-234 with random_in_unit_disk, vec3.h:113-119 (the second
// argument of vec3(random_double(-1,1), random_double(-1,1), 0) is evaluated first by the pinned compiler)
Ray get_ray(const Camera& cam, int i, int j, int s_i, int s_j, double recip_sqrt_spp, Rng& rng) {
    double px = ((s_i + rng.next()) * recip_sqrt_spp) - 0.5;
    double py = ((s_j + rng.next()) * recip_sqrt_spp) - 0.5;
    V3 pixel_sample = add(add(cam.pixel00_loc, scl(i + px, cam.pixel_delta_u)), scl(j + py, cam.pixel_delta_v));
    Ray r;
    r.o = cam.center;
    if (cam.defocus_angle > 0) {
        double dx, dy;
        for (;;) {
            dy = -1 + (1 - -1) * rng.next();            // main.h:36-39
            dx = -1 + (1 - -1) * rng.next();
            if (dx * dx + dy * dy + 0.0 * 0.0 < 1) break;
        }
        r.o = add(add(cam.center, scl(dx, cam.defocus_disk_u)), scl(dy, cam.defocus_disk_v));
    }
    r.d = sub(pixel_sample, r.o);
    (void)rng.next();   // ray_time (camera.h:210): drawn, never used
    return r;
}

Camera unpack_camera(const double* c) {
    Camera cam;
    cam.center = mk(c[0], c[1], c[2]);
    cam.pixel00_loc = mk(c[3], c[4], c[5]);
    cam.pixel_delta_u = mk(c[6], c[7], c[8]);
    cam.pixel_delta_v = mk(c[9], c[10], c[11]);
    cam.background = mk(c[12], c[13], c[14]);
    cam.defocus_disk_u = mk(c[15], c[16], c[17]);
    cam.defocus_disk_v = mk(c[18], c[19], c[20]);
    cam.defocus_angle = c[21];
    return cam;
}

void export_stats(const Stats& st, const oracle_scene& s, const Rng& rng, uint64_t* out) {
    out[0] = st.rays_traced; out[1] = st.aabb_tests; out[2] = st.aabb_hits; out[3] = st.bvh_node_visits;
    out[4] = st.triangle_tests; out[5] = st.triangle_hits; out[6] = s.nodes_built;
    out[7] = rng.total; out[8] = st.shadow_rays; out[9] = st.pairs;
}

// src/main.cpp:67
const std::vector<uint32_t>& effective_lights(const oracle_scene& s, std::vector<uint32_t>& scratch) {
    if (!s.lights.empty()) return s.lights;
    if (s.composite) return s.loose;                       // (every triangle that is an entry of the world's list itself)
    scratch.resize(s.tris.size());
    for (uint32_t i = 0; i < scratch.size(); i++) scratch[i] = i;
    return scratch;
}

}  // namespace

extern "C" {

oracle_scene* oracle_scene_create(const double* tv, const int32_t* tri_material, uint32_t n_tri,
                                  const int32_t* mat_type, const double* mat_params, uint32_t n_mat,
                                  const uint32_t* lights, uint32_t n_lights) {
    if (n_tri == 0) return nullptr;   // the reference's bvh_node constructor recurses forever on an empty list
    oracle_scene* s = new oracle_scene();
    for (uint32_t i = 0; i < n_mat; i++) {
        Mat m;
        m.type = mat_type[i];
        m.rgb = mk(mat_params[5 * i], mat_params[5 * i + 1], mat_params[5 * i + 2]);
        m.fuzz = mat_params[5 * i + 3] < 1 ? mat_params[5 * i + 3] : 1;     // material.h:71
        m.ior = mat_params[5 * i + 4];
        m.tex = -1;
        s->mats.push_back(m);
    }
    for (uint32_t i = 0; i < n_tri; i++) {
        const double* v = tv + 9 * (size_t)i;
        s->tris.push_back(make_tri(mk(v[0], v[1], v[2]), mk(v[3], v[4], v[5]), mk(v[6], v[7], v[8]), tri_material[i]));
    }
    s->lights.assign(lights, lights + n_lights);
    finish_scene(*s);
    return s;
}

static void dfs_leaves(const oracle_scene* s, int32_t c, std::vector<uint32_t>& out);

/* A composite scene (instances, textures, media): the world list a C++ host of the reference would assemble, see
 * oracle/flat_scene.h.  tex_ints: 6 per texture (type even odd width height pad), tex_doubles: 4 (rgb, scale);
 * image_bytes: the bytes of all image textures in texture order (what the reference's loader made of the files);
 * obj_ints: 4 per object (first_triangle n_triangles flags phase_material), obj_doubles: 5 (degrees, offset, density).
 * The perlin tables of noise texture i are built here as the reference's constructor builds them, on the stream
 * (0x7065726c696e, i, 0) the reference harness gives that constructor. */
oracle_scene* oracle_scene_create_composite(const double* tv, const int32_t* tri_material, uint32_t n_tri, const int32_t* mat_type,
                                            const double* mat_params, const int32_t* mat_tex, uint32_t n_mat,
                                            const int32_t* tex_ints, const double* tex_doubles, uint32_t n_tex,
                                            const uint8_t* image_bytes, uint64_t n_image_bytes, const int32_t* obj_ints,
                                            const double* obj_doubles, uint32_t n_obj, const uint32_t* lights, uint32_t n_lights) {
    if (n_tri == 0) return nullptr;
    oracle_scene* s = new oracle_scene();
    s->composite = true;
    for (uint32_t i = 0; i < n_mat; i++) {
        Mat m;
        m.type = mat_type[i];
        m.rgb = mk(mat_params[5 * i], mat_params[5 * i + 1], mat_params[5 * i + 2]);
        m.fuzz = mat_params[5 * i + 3] < 1 ? mat_params[5 * i + 3] : 1;
        m.ior = mat_params[5 * i + 4];
        m.tex = mat_tex[i];
        s->mats.push_back(m);
    }
    s->image_bytes.assign(image_bytes, image_bytes + n_image_bytes);
    size_t at = 0;
    for (uint32_t i = 0; i < n_tex; i++) {
        Texture t;
        t.type = tex_ints[6 * i]; t.even = tex_ints[6 * i + 1]; t.odd = tex_ints[6 * i + 2];
        t.width = tex_ints[6 * i + 3]; t.height = tex_ints[6 * i + 4];
        t.rgb = mk(tex_doubles[4 * i], tex_doubles[4 * i + 1], tex_doubles[4 * i + 2]);
        t.scale = tex_doubles[4 * i + 3];
        t.perlin = -1; t.bytes = 0;
        if (t.type == 1) t.scale = 1.0 / t.scale;            // inv_scale, texture.h:31
        if (t.type == 2) {
            Rng rng;
            rng.engine = 0; rng.mt = nullptr; rng.total = 0;
            rng.st.key = orng_stream_key(0x7065726c696eULL, i, 0);
            rng.st.counter = 0;
            Perlin pn;
            make_perlin(pn, rng);
            t.perlin = (int32_t)s->perlins.size();
            s->perlins.push_back(pn);
        }
        if (t.type == 3) { t.bytes = at; at += (size_t)(t.width > 0 ? t.width : 0) * (size_t)(t.height > 0 ? t.height : 0) * 3; }
        s->textures.push_back(t);
    }
    for (uint32_t i = 0; i < n_tri; i++) {
        const double* v = tv + 9 * (size_t)i;
        s->tris.push_back(make_tri(mk(v[0], v[1], v[2]), mk(v[3], v[4], v[5]), mk(v[6], v[7], v[8]), tri_material[i]));
    }
    s->lights.assign(lights, lights + n_lights);

    // leaf codes: triangles 0 .. n_tri - 1, then the objects
    std::vector<Box> boxes(n_tri + n_obj);
    for (uint32_t i = 0; i < n_tri; i++) boxes[i] = s->tris[i].bbox;
    std::vector<int> owner(n_tri, -1);
    std::vector<int32_t> mesh_root(n_tri, -1);               // by first triangle: the mesh is built once, objects share it
    s->objects.resize(n_obj);
    for (uint32_t k = 0; k < n_obj; k++) {
        Object& ob = s->objects[k];
        ob.first = (uint32_t)obj_ints[4 * k]; ob.count = (uint32_t)obj_ints[4 * k + 1];
        ob.flags = (uint32_t)obj_ints[4 * k + 2]; ob.phase = obj_ints[4 * k + 3];
        const double degrees = obj_doubles[5 * k];
        ob.offset = mk(obj_doubles[5 * k + 1], obj_doubles[5 * k + 2], obj_doubles[5 * k + 3]);
        ob.neg_inv_density = (ob.flags & 4u) ? -1 / obj_doubles[5 * k + 4] : 0.0;     // constant_medium.h:11
        oracle_sincos(degrees * kPi / 180.0, &ob.sin_theta, &ob.cos_theta);          // hittable.h:79-81, main.h:24-26
        for (uint32_t i = 0; i < ob.count; i++) owner[ob.first + i] = (int)k;
        if (mesh_root[ob.first] < 0) {                       // make_shared<bvh_node>(mesh)
            std::vector<uint32_t> items(ob.count);
            for (uint32_t i = 0; i < ob.count; i++) items[i] = ob.first + i;
            mesh_root[ob.first] = build_node(*s, boxes, items, 0, items.size());
        }
        ob.root = mesh_root[ob.first];
        Box b = s->nodes[ob.root].box;
        if (ob.flags & 1u) {                                 // rotate_y::rotate_y, hittable.h:82-107
            double mn[3] = { kInfinity, kInfinity, kInfinity }, mx[3] = { -kInfinity, -kInfinity, -kInfinity };
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int q = 0; q < 2; q++) {
                        double x = i * b.mx[0] + (1 - i) * b.mn[0];
                        double y = j * b.mx[1] + (1 - j) * b.mn[1];
                        double z = q * b.mx[2] + (1 - q) * b.mn[2];
                        double newx = ob.cos_theta * x + ob.sin_theta * z;
                        double newz = -ob.sin_theta * x + ob.cos_theta * z;
                        double tester[3] = { newx, y, newz };
                        for (int c = 0; c < 3; c++) {
                            mn[c] = std::fmin(mn[c], tester[c]);
                            mx[c] = std::fmax(mx[c], tester[c]);
                        }
                    }
            for (int c = 0; c < 3; c++) {                    // aabb(point, point), aabb.h:18-25
                if (mn[c] <= mx[c]) { b.mn[c] = mn[c]; b.mx[c] = mx[c]; }
                else { b.mn[c] = mx[c]; b.mx[c] = mn[c]; }
            }
        }
        if (ob.flags & 2u) {                                 // bbox + offset, aabb.h:95-97 (interval constructor: pads)
            const double off[3] = { ob.offset.x, ob.offset.y, ob.offset.z };
            for (int c = 0; c < 3; c++) { b.mn[c] = b.mn[c] + off[c]; b.mx[c] = b.mx[c] + off[c]; pad_axis(b.mn[c], b.mx[c]); }
        }
        ob.bbox = b;
        boxes[n_tri + k] = b;
    }
    // the world's list: triangles of no object, and every object at the place of its first triangle (objects that
    // share a mesh in the order of their indices); bvh_node(list), src/main.cpp:66
    std::vector<uint32_t> items;
    for (uint32_t i = 0; i < n_tri;) {
        if (owner[i] < 0) { items.push_back(i); s->loose.push_back(i); i++; continue; }
        uint32_t span = 0;
        for (uint32_t k = 0; k < n_obj; k++)
            if (s->objects[k].first == i) { items.push_back(n_tri + k); span = s->objects[k].count; }
        i += span;
    }
    s->root = build_node(*s, boxes, items, 0, items.size());
    return s;
}

/* leaves of the world's tree in visiting order: triangle id, or -(object + 2) */
uint32_t oracle_world_leaves(const oracle_scene* s, int32_t* out, uint32_t capacity) {
    std::vector<uint32_t> v;
    dfs_leaves(s, s->root, v);
    const uint32_t n_tri = (uint32_t)s->tris.size();
    for (uint32_t i = 0; i < v.size() && i < capacity; i++) out[i] = v[i] < n_tri ? (int32_t)v[i] : -((int32_t)(v[i] - n_tri) + 2);
    return (uint32_t)v.size();
}
static void world_boxes(const oracle_scene* s, int32_t c, std::vector<double>& out) {
    if (c < 0) return;
    for (int k = 0; k < 3; k++) { out.push_back(s->nodes[c].box.mn[k]); out.push_back(s->nodes[c].box.mx[k]); }
    world_boxes(s, s->nodes[c].left, out);
    world_boxes(s, s->nodes[c].right, out);
}
/* boxes of the world tree's nodes in pre-order (6 doubles each); returns the number of nodes */
uint32_t oracle_world_boxes(const oracle_scene* s, double* out, uint32_t capacity_nodes) {
    std::vector<double> v;
    world_boxes(s, s->root, v);
    for (size_t i = 0; i < v.size() && i < (size_t)capacity_nodes * 6; i++) out[i] = v[i];
    return (uint32_t)(v.size() / 6);
}
/* perlin tables of the noise textures in texture order: 768 doubles (randvec) + 768 doubles (perm_x, perm_y, perm_z) each */
uint32_t oracle_perlin_tables(const oracle_scene* s, double* out) {
    for (size_t n = 0; n < s->perlins.size(); n++) {
        double* o = out + 1536 * n;
        const Perlin& pn = s->perlins[n];
        for (int k = 0; k < 256; k++) { o[3 * k] = pn.randvec[k].x; o[3 * k + 1] = pn.randvec[k].y; o[3 * k + 2] = pn.randvec[k].z; }
        for (int k = 0; k < 256; k++) { o[768 + k] = pn.perm_x[k]; o[1024 + k] = pn.perm_y[k]; o[1280 + k] = pn.perm_z[k]; }
    }
    return (uint32_t)s->perlins.size();
}

oracle_scene* oracle_scene_load(const char* path, double* cam_pub, int32_t* cam_ints) {
    fs_scene fs;
    if (fs_load(path, &fs) != 0) return nullptr;
    std::vector<double> tv(9 * (size_t)fs.n_triangles), mp(5 * (size_t)fs.n_materials);
    std::vector<int32_t> tm(fs.n_triangles), mt(fs.n_materials);
    for (uint32_t i = 0; i < fs.n_triangles; i++) {
        memcpy(&tv[9 * (size_t)i], fs.triangles[i].v0, 72);
        tm[i] = fs.triangles[i].material;
    }
    for (uint32_t i = 0; i < fs.n_materials; i++) {
        mt[i] = fs.materials[i].type;
        memcpy(&mp[5 * (size_t)i], fs.materials[i].rgb, 24);
        mp[5 * i + 3] = fs.materials[i].fuzz;
        mp[5 * i + 4] = fs.materials[i].ior;
    }
    oracle_scene* s = oracle_scene_create(tv.data(), tm.data(), fs.n_triangles, mt.data(), mp.data(), fs.n_materials,
                                          fs.lights, fs.n_lights);
    if (cam_pub) memcpy(cam_pub, &fs.camera, 16 * sizeof(double));
    if (cam_ints) { cam_ints[0] = fs.camera.image_width; cam_ints[1] = fs.camera.samples_per_pixel; cam_ints[2] = fs.camera.max_depth; }
    fs_free(&fs);
    return s;
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

uint32_t oracle_scene_triangle_count(const oracle_scene* s) { return (uint32_t)s->tris.size(); }
uint32_t oracle_bvh_node_count(const oracle_scene* s) { return (uint32_t)s->nodes.size(); }

static void dfs_leaves(const oracle_scene* s, int32_t c, std::vector<uint32_t>& out) {
    if (c < 0) { out.push_back((uint32_t)~c); return; }
    dfs_leaves(s, s->nodes[c].left, out);
    dfs_leaves(s, s->nodes[c].right, out);
}

uint32_t oracle_bvh_dfs_leaves(const oracle_scene* s, uint32_t* out, uint32_t capacity) {
    std::vector<uint32_t> v;
    dfs_leaves(s, s->root, v);
    for (uint32_t i = 0; i < v.size() && i < capacity; i++) out[i] = v[i];
    return (uint32_t)v.size();
}

void oracle_bvh_boxes(const oracle_scene* s, double* out) {
    // nodes are stored in pre-order already
    for (size_t i = 0; i < s->nodes.size(); i++)
        for (int k = 0; k < 3; k++) { out[6 * i + 2 * k] = s->nodes[i].box.mn[k]; out[6 * i + 2 * k + 1] = s->nodes[i].box.mx[k]; }
}

void oracle_triangle_data(const oracle_scene* s, double* out) {
    for (size_t i = 0; i < s->tris.size(); i++) {
        const Tri& t = s->tris[i];
        double* o = out + 16 * i;
        o[0] = t.edge1.x; o[1] = t.edge1.y; o[2] = t.edge1.z;
        o[3] = t.edge2.x; o[4] = t.edge2.y; o[5] = t.edge2.z;
        o[6] = t.normal.x; o[7] = t.normal.y; o[8] = t.normal.z;
        o[9] = t.area;
        for (int k = 0; k < 3; k++) { o[10 + 2 * k] = t.bbox.mn[k]; o[11 + 2 * k] = t.bbox.mx[k]; }
    }
}

// src/camera.h:160-197
void oracle_camera_initialize(const double* pub, int32_t image_width, int32_t samples_per_pixel, double* out,
                              int32_t* image_height_out, int32_t* sqrt_spp_out) {
    V3 lookfrom = mk(pub[0], pub[1], pub[2]), lookat = mk(pub[3], pub[4], pub[5]), vup = mk(pub[6], pub[7], pub[8]);
    double vfov = pub[9], aspect_ratio = pub[10], focus_dist = pub[11];
    int image_height = int(image_width / aspect_ratio);
    image_height = (image_height < 1) ? 1 : image_height;
    int sqrt_spp = int(std::sqrt(samples_per_pixel));
    V3 center = lookfrom;
    double theta = vfov * kPi / 180.0;                           // main.h:24-26
    double h = std::tan(theta / 2);
    double viewport_height = 2 * h * focus_dist;
    double viewport_width = viewport_height * (double(image_width) / image_height);
    V3 w = unit(sub(lookfrom, lookat));
    V3 u = unit(cross(vup, w));
    V3 v = cross(w, u);
    V3 viewport_u = scl(viewport_width, u);
    V3 viewport_v = scl(viewport_height, neg(v));
    V3 pixel_delta_u = divs(viewport_u, image_width);
    V3 pixel_delta_v = divs(viewport_v, image_height);
    V3 viewport_upper_left = sub(sub(sub(center, scl(focus_dist, w)), divs(viewport_u, 2)), divs(viewport_v, 2));
    V3 pixel00_loc = add(viewport_upper_left, scl(0.5, add(pixel_delta_u, pixel_delta_v)));
    out[0] = center.x; out[1] = center.y; out[2] = center.z;
    out[3] = pixel00_loc.x; out[4] = pixel00_loc.y; out[5] = pixel00_loc.z;
    out[6] = pixel_delta_u.x; out[7] = pixel_delta_u.y; out[8] = pixel_delta_u.z;
    out[9] = pixel_delta_v.x; out[10] = pixel_delta_v.y; out[11] = pixel_delta_v.z;
    // camera.h:193-196; pub[12] = defocus_angle
    double defocus_radius = focus_dist * std::tan((pub[12] / 2) * kPi / 180.0);
    V3 disk_u = scl(defocus_radius, u), disk_v = scl(defocus_radius, v);
    out[12] = disk_u.x; out[13] = disk_u.y; out[14] = disk_u.z;
    out[15] = disk_v.x; out[16] = disk_v.y; out[17] = disk_v.z;
    out[18] = pub[12];
    *image_height_out = image_height;
    *sqrt_spp_out = sqrt_spp;
}

void oracle_trace_closest(const oracle_scene* s, const double* rays, uint64_t n, double* out, uint64_t* stats) {
    Stats st;
    memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        const double* r = rays + 8 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        Hit rec;
        rec.u = 0; rec.v = 0;
        Rng rng;                                            // query i draws from stream (seed 0, pixel i, sample 0): media
        rng.engine = 0; rng.mt = nullptr; rng.total = 0;
        rng.st.key = orng_stream_key(0, i, 0);
        rng.st.counter = 0;
        double* o = out + 11 * i;
        if (!world_hit(*s, ray, r[6], r[7], rec, st, &rng)) {
            o[0] = -1.0;
            for (int k = 1; k < 11; k++) o[k] = 0.0;
        } else {
            o[0] = rec.tri; o[1] = rec.t; o[2] = rec.u; o[3] = rec.v; o[4] = rec.front_face ? 1.0 : 0.0;
            o[5] = rec.p.x; o[6] = rec.p.y; o[7] = rec.p.z;
            o[8] = rec.normal.x; o[9] = rec.normal.y; o[10] = rec.normal.z;
        }
    }
    if (stats) {
        stats[0] = st.rays_traced; stats[1] = st.aabb_tests; stats[2] = st.aabb_hits; stats[3] = st.bvh_node_visits;
        stats[4] = st.triangle_tests; stats[5] = st.triangle_hits; stats[6] = s->nodes_built;
    }
}

// The same query answered without any hierarchy: is ANY triangle accepted in the interval?  Differs from
// oracle_visible exactly where the reference's box tests keep it from testing a triangle that would accept.
void oracle_visible_all_triangles(const oracle_scene* s, const double* pairs, uint64_t n, double* out) {
    Stats st;
    memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        const double* p = pairs + 6 * i;
        V3 a = mk(p[0], p[1], p[2]), b = mk(p[3], p[4], p[5]);
        V3 dir = sub(b, a);
        double dist = len(dir);
        out[i] = 0.0;
        if (dist <= 0) continue;
        V3 dir_unit = divs(dir, dist);
        Ray shadow;
        shadow.o = add(a, scl(0.001, dir_unit));
        shadow.d = dir_unit;
        double max_t = dist - 0.001;
        if (max_t <= 0) continue;
        bool blocked = false;
        Hit tmp;
        for (size_t t = 0; t < s->tris.size() && !blocked; t++) blocked = tri_hit(*s, (int32_t)t, shadow, 0.001, max_t, tmp, st);
        out[i] = blocked ? 0.0 : 1.0;
    }
}

void oracle_visible(const oracle_scene* s, const double* pairs, uint64_t n, double* out) {
    Stats st;
    memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        const double* p = pairs + 6 * i;
        Rng rng;
        rng.engine = 0; rng.mt = nullptr; rng.total = 0;
        rng.st.key = orng_stream_key(0, i, 0);
        rng.st.counter = 0;
        out[i] = visible(*s, mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), st, &rng) ? 1.0 : 0.0;
    }
}

static void render_impl(const oracle_scene* s, const double* camera, int32_t W, int32_t sqrt_spp, int32_t max_depth,
                        uint64_t seed, int32_t x0, int32_t y0, int32_t x1, int32_t y1, double* fb_out,
                        double* per_sample_out, uint64_t* stats, bool unidirectional) {
    Camera cam = unpack_camera(camera);
    Stats st;
    memset(&st, 0, sizeof st);
    Rng rng;
    rng.engine = 0; rng.mt = nullptr; rng.total = 0;
    std::vector<uint32_t> scratch;
    const std::vector<uint32_t>& lights = effective_lights(*s, scratch);
    std::vector<PathVertex> camera_path, light_path;
    const double recip_sqrt_spp = 1.0 / sqrt_spp;                  // camera.h:166
    size_t px = 0, sm = 0;
    for (int j = y0; j < y1; j++) {
        for (int i = x0; i < x1; i++) {
            V3 pixel_color = mk(0, 0, 0);
            for (int s_j = 0; s_j < sqrt_spp; s_j++) {
                for (int s_i = 0; s_i < sqrt_spp; s_i++) {
                    rng.st.key = orng_stream_key(seed, (uint64_t)j * W + i, (uint64_t)s_j * sqrt_spp + s_i);
                    rng.st.counter = 0;
                    Ray r = get_ray(cam, i, j, s_i, s_j, recip_sqrt_spp, rng);
                    V3 c = unidirectional ? path_trace_color(*s, cam, lights, r, max_depth, rng, st)
                                          : bidirectional_color(*s, cam, lights, r, max_depth, rng, st, camera_path, light_path);
                    pixel_color = add(pixel_color, c);
                    if (per_sample_out) {
                        per_sample_out[4 * sm] = c.x; per_sample_out[4 * sm + 1] = c.y; per_sample_out[4 * sm + 2] = c.z;
                        per_sample_out[4 * sm + 3] = (double)rng.st.counter;
                    }
                    sm++;
                }
            }
            fb_out[3 * px] = pixel_color.x; fb_out[3 * px + 1] = pixel_color.y; fb_out[3 * px + 2] = pixel_color.z;
            px++;
        }
    }
    if (stats) export_stats(st, *s, rng, stats);
}

void oracle_render(const oracle_scene* s, const double* camera, int32_t W, int32_t sqrt_spp, int32_t max_depth,
                   uint64_t seed, int32_t x0, int32_t y0, int32_t x1, int32_t y1, double* fb_out,
                   double* per_sample_out, uint64_t* stats) {
    render_impl(s, camera, W, sqrt_spp, max_depth, seed, x0, y0, x1, y1, fb_out, per_sample_out, stats, false);
}

// the same loop with camera::path_trace_color as the estimator (camera.h:245-253 when `lights` is no triangle_collection)
void oracle_render_unidirectional(const oracle_scene* s, const double* camera, int32_t W, int32_t sqrt_spp, int32_t max_depth,
                                  uint64_t seed, int32_t x0, int32_t y0, int32_t x1, int32_t y1, double* fb_out,
                                  uint64_t* stats) {
    render_impl(s, camera, W, sqrt_spp, max_depth, seed, x0, y0, x1, y1, fb_out, nullptr, stats, true);
}

void oracle_render_serial_mt(const oracle_scene* s, const double* camera, int32_t W, int32_t H, int32_t sqrt_spp,
                             int32_t max_depth, double* fb_out, uint64_t* stats) {
    Camera cam = unpack_camera(camera);
    Stats st;
    memset(&st, 0, sizeof st);
    orng_mt19937 mt;
    orng_mt_seed(&mt, 5489u);
    Rng rng;
    rng.engine = 1; rng.mt = &mt; rng.total = 0;
    std::vector<uint32_t> scratch;
    const std::vector<uint32_t>& lights = effective_lights(*s, scratch);
    std::vector<PathVertex> camera_path, light_path;
    const double recip_sqrt_spp = 1.0 / sqrt_spp;
    size_t px = 0;
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            V3 pixel_color = mk(0, 0, 0);
            for (int s_j = 0; s_j < sqrt_spp; s_j++)
                for (int s_i = 0; s_i < sqrt_spp; s_i++) {
                    Ray r = get_ray(cam, i, j, s_i, s_j, recip_sqrt_spp, rng);
                    pixel_color = add(pixel_color, bidirectional_color(*s, cam, lights, r, max_depth, rng, st, camera_path, light_path));
                }
            fb_out[3 * px] = pixel_color.x; fb_out[3 * px + 1] = pixel_color.y; fb_out[3 * px + 2] = pixel_color.z;
            px++;
        }
    if (stats) export_stats(st, *s, rng, stats);
}

// src/image/wpng.h:10-35.  A NaN component is undefined behaviour in the reference
// (uint8 cast of NaN); it is defined as 0 here.
void oracle_colors_to_rgb8(const double* fb, int32_t W, int32_t H, int32_t samples_per_pixel, uint8_t* out) {
    const double scale = 1.0 / std::max(1, samples_per_pixel);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i)
            for (int k = 0; k < 3; k++) {
                double c = fb[(size_t)(j * W + i) * 3 + k] * scale;
                double cl = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
                double g = std::sqrt(cl);
                double val = 256.0 * g;
                out[(size_t)(j * W + i) * 3 + k] = (val != val) ? 0 : static_cast<uint8_t>(val);
            }
}

void oracle_sincos_array(const double* x, uint64_t n, double* s, double* c) {
    for (uint64_t i = 0; i < n; i++) oracle_sincos(x[i], &s[i], &c[i]);
}
void oracle_pow5_array(const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = oracle_pow5(x[i]);
}
void oracle_log_array(const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = oracle_log(x[i]);
}
void oracle_rng_array(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* out) {
    orng_stream st;
    st.key = orng_stream_key(seed, pixel, sample);
    st.counter = 0;
    for (uint64_t i = 0; i < n; i++) out[i] = orng_stream_next(&st);
}
void oracle_mt_array(uint64_t n, double* out) {
    orng_mt19937 mt;
    orng_mt_seed(&mt, 5489u);
    for (uint64_t i = 0; i < n; i++) out[i] = orng_mt_next_double(&mt);
}

}  // extern "C"
