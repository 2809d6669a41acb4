// Device-side building blocks of the radiance-estimation path: FP64 vector algebra in the
// reference's evaluation order, the per-sample random stream, the threaded-BVH walk, the
// materials and the subpath / connection logic.  Citations are to the reference renderer's
// source tree.  Compile with -ffp-contract=off: every + - * / sqrt below must round exactly
// once, in the order written.
#ifndef BDPT_DEVICE_CORE_H
#define BDPT_DEVICE_CORE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bdpt_types.h"
#include "det_math.h"

namespace bdpt {

#define BDPT_DEV __device__ __forceinline__

constexpr double kPi = 3.1415926535897932385;                    // src/main.h:20
constexpr double kInf = __builtin_huge_val();

// ------------------------------------------------------------------ vec3 (src/core/vec3.h)
struct V3 { double x, y, z; };

BDPT_DEV V3 mk(double a, double b, double c) { V3 v; v.x = a; v.y = b; v.z = c; return v; }
BDPT_DEV V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
BDPT_DEV V3 neg(V3 v) { return mk(-v.x, -v.y, -v.z); }
BDPT_DEV V3 add(V3 u, V3 v) { return mk(u.x + v.x, u.y + v.y, u.z + v.z); }
BDPT_DEV V3 sub(V3 u, V3 v) { return mk(u.x - v.x, u.y - v.y, u.z - v.z); }
BDPT_DEV V3 mulv(V3 u, V3 v) { return mk(u.x * v.x, u.y * v.y, u.z * v.z); }
BDPT_DEV V3 scl(double t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
BDPT_DEV V3 divs(V3 v, double t) { return scl(1 / t, v); }                              // vec3.h:93-95
BDPT_DEV double dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
BDPT_DEV V3 cross(V3 u, V3 v) {
    return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
BDPT_DEV double len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
BDPT_DEV double len(V3 v) { return __builtin_sqrt(len2(v)); }
BDPT_DEV V3 unit(V3 v) { return divs(v, len(v)); }                                      // vec3.h:109-111
BDPT_DEV V3 reflect(V3 v, V3 n) { return sub(v, scl(2 * dot(v, n), n)); }               // vec3.h:138-140
BDPT_DEV V3 refract(V3 uv, V3 n, double etai_over_etat) {                               // vec3.h:142-147
    double cos_theta = __builtin_fmin(dot(neg(uv), n), 1.0);
    V3 r_out_perp = scl(etai_over_etat, add(uv, scl(cos_theta, n)));
    V3 r_out_parallel = scl(-__builtin_sqrt(__builtin_fabs(1.0 - len2(r_out_perp))), n);
    return add(r_out_perp, r_out_parallel);
}

// ------------------------------------------------------------------ random stream (src/main.h:28-37)
// Each (pixel, sample) owns a counter-based stream; a draw turns one 64-bit word into a double
// with libstdc++'s generate_canonical formula (g1 + g2*2^32) / 2^64, >= 1 -> nextafter(1, 0).
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;

BDPT_DEV uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
BDPT_DEV uint64_t stream_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t k = mix64(seed + kGolden * (pixel + 1));
    return mix64(k ^ (0xD1B54A32D192ED03ULL * (sample + 1)));
}
struct Rng {
    uint64_t key;
    uint32_t counter;
    BDPT_DEV double next() {
        counter++;
        uint64_t x = mix64(key + (uint64_t)counter * kGolden);
        double sum = (double)(uint32_t)(x & 0xFFFFFFFFu) + (double)(uint32_t)(x >> 32) * 4294967296.0;
        double r = sum / 18446744073709551616.0;
        if (r >= 1.0) r = 0x1.fffffffffffffp-1;
        return r;
    }
    BDPT_DEV double next(double mn, double mx) { return mn + (mx - mn) * next(); }          // main.h:34-37
};

// vec3.h:121-128; vec3::random(-1,1) fills z first, then y, then x (argument evaluation order of
// the compiler that built the pinned reference).
BDPT_DEV V3 random_unit_vector(Rng& rng) {
    for (;;) {
        double pz = rng.next(-1, 1);
        double py = rng.next(-1, 1);
        double px = rng.next(-1, 1);
        V3 p = mk(px, py, pz);
        double lensq = len2(p);
        if (1e-160 < lensq && lensq <= 1) return divs(p, __builtin_sqrt(lensq));
    }
}

BDPT_DEV V3 random_cosine_direction(Rng& rng) {                                           // vec3.h:149-159
    double r1 = rng.next();
    double r2 = rng.next();
    double phi = 2 * kPi * r1;
    double s, c;
    det_sincos(phi, s, c);
    double x = c * __builtin_sqrt(r2);
    double y = s * __builtin_sqrt(r2);
    double z = __builtin_sqrt(1 - r2);
    return mk(x, y, z);
}

// src/acceleration/onb.h:6-20: returns transform(v) for the basis built around n; w receives axis[2]
BDPT_DEV V3 onb_transform(V3 n, V3 v, V3& w) {
    V3 a2 = unit(n);
    V3 a = (__builtin_fabs(a2.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    V3 a1 = unit(cross(a2, a));
    V3 a0 = cross(a2, a1);
    w = a2;
    return add(add(scl(v.x, a0), scl(v.y, a1)), scl(v.z, a2));
}

// ------------------------------------------------------------------ counters
// Counting modes of the kernels: none; the reference's BvhStats semantics (shadow rays run the
// reference's full closest-hit query so that every counter equals the CPU renderer's); the work
// the production path really does (shadow rays stop at the first hit).
constexpr int kStatsOff = 0, kStatsReference = 1, kStatsWork = 2;

struct Counters {
    uint32_t rays_traced, aabb_tests, aabb_hits, node_visits, tri_tests, tri_hits, shadow_rays, pairs, vertices;
};

// ------------------------------------------------------------------ scene access
// Global-memory view of the scene tables.
struct GlobalScene {
    const GpuNode* nodes;
    const GpuTri* tris;
    BDPT_DEV GpuNode node(uint32_t i) const { return nodes[i]; }
    BDPT_DEV GpuTri tri(int32_t i) const { return tris[i]; }
};

// LDS-resident copy of the node and triangle tables (small scenes): every box / triangle fetch
// of the walk is a ds_read_b128 burst instead of a trip to L2.
struct LdsScene {
    const GpuNode* nodes;     // points into __shared__ memory
    const GpuTri* tris;
    BDPT_DEV GpuNode node(uint32_t i) const { return nodes[i]; }
    BDPT_DEV GpuTri tri(int32_t i) const { return tris[i]; }
};

struct HitLite {
    double t, u, v;
    int32_t tri;
};

// triangle::hit, src/objects/primatives/triangle.h:41-74 (interval test inclusive, interval.h:22-24)
BDPT_DEV bool tri_test(const GpuTri& tr, V3 o, V3 d, double tmin, double tmax, double& t_out, double& u_out,
                       double& v_out) {
    V3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
    V3 pvec = cross(d, e2);
    double det = dot(e1, pvec);
    if (__builtin_fabs(det) < 1e-8) return false;
    double inv_det = 1.0 / det;
    V3 tvec = sub(o, ld3(tr.v0));
    double u = dot(tvec, pvec) * inv_det;
    if (u < 0.0 || u > 1.0) return false;
    V3 qvec = cross(tvec, e1);
    double v = dot(d, qvec) * inv_det;
    if (v < 0.0 || u + v > 1.0) return false;
    double t = dot(e2, qvec) * inv_det;
    if (!(tmin <= t && t <= tmax)) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

// The same arithmetic as straight-line code for the traversal kernels: every lane of a wave evaluates the whole test
// (in a wave some lane almost always gets to the end, so the early exits above only cost branches and delay the
// loads of the later operands) and the outcome is one predicate.  Each of t, u, v, det is formed by the operations
// of triangle::hit in their order, so a lane for which tri_test returns true gets the same bits here; the upper end
// of the interval is left to the caller:  tri_test(..., tmax) == tri_candidate(...) && t <= tmax.
BDPT_DEV bool tri_candidate(const GpuTri& tr, V3 o, V3 d, double tmin, double& t, double& u, double& v, double& det) {
    const V3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
    const V3 pvec = cross(d, e2);
    det = dot(e1, pvec);
    const double inv_det = 1.0 / det;
    const V3 tvec = sub(o, ld3(tr.v0));
    u = dot(tvec, pvec) * inv_det;
    const V3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    t = dot(e2, qvec) * inv_det;
    // (negated comparisons as in the reference: a NaN fails where the reference's test lets it pass and vice versa)
    const unsigned ok = (unsigned)!(__builtin_fabs(det) < 1e-8) & (unsigned)!(u < 0.0) & (unsigned)!(u > 1.0) & (unsigned)!(v < 0.0) &
                        (unsigned)!(u + v > 1.0) & (unsigned)(tmin <= t);
    return ok != 0u;
}

// bvh_node::hit (src/acceleration/bvh.h:50-59) + aabb::hit (src/acceleration/aabb.h:41-66) as a walk
// over the threaded tree.  The reference hands every later sibling the closest t found so far
// (bvh.h:56), so one running `closest` reproduces its intervals; visiting order is the
// reference's (left subtree, then right), hence equal-t ties resolve identically.
// ANYHIT returns at the first accepted triangle: the boolean the reference derives from its full
// closest-hit query (camera.h:437) is the same.
template <int STATS, bool ANYHIT, class Mem>
BDPT_DEV bool walk(const Mem& mem, V3 o, V3 d, double tmin, double tmax, HitLite& hit, Counters& c) {
    // aabb.h:48 computes 1.0 / dir[axis] inside every box test; the value is the same each time
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
    double closest = tmax;
    bool any = false;
    uint32_t i = 0;
    while (i != kEndOfTree) {
        const GpuNode nd = mem.node(i);
        if (STATS) { c.node_visits++; c.aabb_tests++; }
        double lo = tmin, hi = closest;
        {
            double t0 = (nd.lo[0] - o.x) * ix, t1 = (nd.hi[0] - o.x) * ix;
            double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
            lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
        }
        {
            double t0 = (nd.lo[1] - o.y) * iy, t1 = (nd.hi[1] - o.y) * iy;
            double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
            lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
        }
        {
            double t0 = (nd.lo[2] - o.z) * iz, t1 = (nd.hi[2] - o.z) * iz;
            double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
            lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
        }
        // lo only grows and hi only shrinks, so the reference's per-axis exits (aabb.h:61-62)
        // fire exactly when this final comparison does
        if (hi <= lo) { i = nd.skip; continue; }
        if (STATS) c.aabb_hits++;
        if (nd.tri0 >= 0) {
            double t, u, v;
            {
                const GpuTri tr = mem.tri(nd.tri0);
                bool h = tri_test(tr, o, d, tmin, closest, t, u, v);
                const bool twice = (nd.tri1 == nd.tri0);     // one-triangle node: the reference tests it twice
                if (STATS) { c.tri_tests += twice ? 2 : 1; if (h) c.tri_hits += twice ? 2 : 1; }
                if (h) {
                    if (ANYHIT) return true;
                    closest = t; hit.t = t; hit.u = u; hit.v = v; hit.tri = nd.tri0; any = true;
                }
            }
            if (nd.tri1 != nd.tri0) {
                const GpuTri tr = mem.tri(nd.tri1);
                bool h = tri_test(tr, o, d, tmin, closest, t, u, v);
                if (STATS) { c.tri_tests++; if (h) c.tri_hits++; }
                if (h) {
                    if (ANYHIT) return true;
                    closest = t; hit.t = t; hit.u = u; hit.v = v; hit.tri = nd.tri1; any = true;
                }
            }
        }
        i = nd.next;
    }
    return any;
}

// camera::visible, src/camera.h:425-438
template <int STATS, class Mem>
BDPT_DEV bool visible(const Mem& mem, V3 a, V3 b, Counters& c) {
    V3 dir = sub(b, a);
    double dist = len(dir);
    if (dist <= 0) return false;
    V3 dir_unit = divs(dir, dist);
    V3 origin = add(a, scl(0.001, dir_unit));
    double max_t = dist - 0.001;
    if (max_t <= 0) return false;
    if (STATS) c.shadow_rays++;
    HitLite tmp;
    return !walk<STATS, STATS != kStatsReference>(mem, origin, dir_unit, 0.001, max_t, tmp, c);
}

// ------------------------------------------------------------------ path vertices
// What the connection step needs from a path_vertex (src/camera.h:236-243): position, shading
// normal, throughput, material and three flags.  Delta vertices are never stored: connect_vertices
// returns zero for them before looking at anything else (camera.h:441).
constexpr int kVtxDoubles = 10;   // p[3] n[3] T[3] + one slot holding (material | flags)
// With textured materials the colour a vertex can contribute with is a function of its hit (texture::value(u, v, p)),
// not of its material alone: the record then also holds `f`, three more slots.
constexpr int kVtxDoublesTextured = 13;
constexpr uint32_t kVtxIsLight = 1u << 30;
constexpr uint32_t kVtxFrontFace = 1u << 29;
constexpr uint32_t kVtxMatMask = (1u << 29) - 1;

struct Vertex {
    V3 p, n, T;
    uint32_t info;
    uint32_t tri;        // the triangle the vertex lies on (upper half of the record's last slot)
    V3 f;                // textured scenes only: emission of a vertex on an emitter (zero from behind), else its BSDF value
};

// SoA in global memory: element (slot, field) of thread g lives at ws[(slot*10 + field)*stride + g],
// so a wave's accesses to one field of one slot are contiguous.
struct VertexStore {
    double* ws;
    uint64_t stride;
    uint64_t g;
    template <bool TEX = false>
    BDPT_DEV void put(int slot, const Vertex& v) const {
        double* b = ws + (uint64_t)slot * (TEX ? kVtxDoublesTextured : kVtxDoubles) * stride + g;
        b[0] = v.p.x; b[stride] = v.p.y; b[2 * stride] = v.p.z;
        b[3 * stride] = v.n.x; b[4 * stride] = v.n.y; b[5 * stride] = v.n.z;
        b[6 * stride] = v.T.x; b[7 * stride] = v.T.y; b[8 * stride] = v.T.z;
        b[9 * stride] = __longlong_as_double((long long)(((unsigned long long)v.tri << 32) | v.info));
        if (TEX) { b[10 * stride] = v.f.x; b[11 * stride] = v.f.y; b[12 * stride] = v.f.z; }
    }
    template <bool TEX = false>
    BDPT_DEV Vertex get(int slot) const {
        const double* b = ws + (uint64_t)slot * (TEX ? kVtxDoublesTextured : kVtxDoubles) * stride + g;
        Vertex v;
        v.p = mk(b[0], b[stride], b[2 * stride]);
        v.n = mk(b[3 * stride], b[4 * stride], b[5 * stride]);
        v.T = mk(b[6 * stride], b[7 * stride], b[8 * stride]);
        const unsigned long long w = (unsigned long long)__double_as_longlong(b[9 * stride]);
        v.info = (uint32_t)w;
        v.tri = (uint32_t)(w >> 32);
        v.f = TEX ? mk(b[10 * stride], b[11 * stride], b[12 * stride]) : mk(0, 0, 0);
        return v;
    }
};

// ------------------------------------------------------------------ materials (src/materials/material.h)
BDPT_DEV double reflectance(double cosine, double refraction_index) {                      // material.h:125-130
    double r0 = (1 - refraction_index) / (1 + refraction_index);
    r0 = r0 * r0;
    return r0 + (1 - r0) * det_pow5(1 - cosine);
}

// BSDF value used by connect_vertices: albedo/pi for lambertian (material.h:60-63), zero otherwise (:35-37)
BDPT_DEV V3 evaluate_bsdf(const GpuMaterial& m) {
    if (m.type == 0) return divs(ld3(m.rgb), kPi);
    return mk(0, 0, 0);
}

// One iteration of the loop of camera::trace_path (src/camera.h:342-368) after a hit was found:
// builds the path vertex, evaluates emission, samples the material.
struct Bounce {
    bool keep;            // non-delta vertex: goes into the vertex store (camera.h:441 drops delta ones)
    Vertex vx;
    bool emitted;         // EYE only: non-delta emissive vertex, camera.h:306-309
    V3 emitted_term;
    bool cont;            // path continues with (o, d, T)
    V3 o, d, T;
};

// LAMBERT = false: for scenes without a lambertian triangle (that material's code is the largest part of the function)
// (the material as the function reads it: from the scene's tables, or from the one record per hit of GpuShadeRec)
struct ShadeMat {
    V3 normal, rgb;
    double fuzz, ior;
    int32_t type, id;
};
template <bool EYE, bool LAMBERT = true>
BDPT_DEV void shade_bounce(const ShadeMat& m, int32_t tri, V3 o, V3 d, double t_hit, V3 throughput, Rng& rng, Bounce& out);

template <bool EYE, bool LAMBERT = true>
BDPT_DEV void shade_bounce(const SceneDev& sc, V3 o, V3 d, const HitLite& h, V3 throughput, Rng& rng, Bounce& out) {
    const GpuTriAux ax = sc.aux[h.tri];
    const int32_t mat_id = sc.tris[h.tri].material;
    const GpuMaterial gm = sc.mats[mat_id];
    ShadeMat m;
    m.normal = ld3(ax.normal); m.rgb = ld3(gm.rgb); m.fuzz = gm.fuzz; m.ior = gm.ior; m.type = gm.type; m.id = mat_id;
    shade_bounce<EYE, LAMBERT>(m, h.tri, o, d, h.t, throughput, rng, out);
}
// the same from the record of leaf position `pos`
template <bool EYE, bool LAMBERT = true>
BDPT_DEV void shade_bounce_at(const SceneDev& sc, uint32_t pos, V3 o, V3 d, double t_hit, V3 throughput, Rng& rng, Bounce& out) {
    const GpuShadeRec r = sc.shade[pos];
    ShadeMat m;
    m.normal = ld3(r.normal); m.rgb = ld3(r.rgb); m.fuzz = r.param; m.ior = r.param;
    m.type = (int32_t)(r.mat >> 30); m.id = (int32_t)(r.mat & kVtxMatMask);
    shade_bounce<EYE, LAMBERT>(m, (int32_t)r.tri, o, d, t_hit, throughput, rng, out);
}

// (p, normal, front_face: the hit_record as the world's hit() returned it; d: the ray's direction)
// MEDIA: the material may also be the isotropic phase function of a constant_medium (material.h:149-172)
template <bool EYE, bool LAMBERT, bool MEDIA = false>
BDPT_DEV void shade_surface(const ShadeMat& m, int32_t tri, V3 p, V3 normal, bool front_face, V3 d, V3 throughput, Rng& rng, Bounce& out);

template <bool EYE, bool LAMBERT>
BDPT_DEV void shade_bounce(const ShadeMat& m, int32_t tri, V3 o, V3 d, double t_hit, V3 throughput, Rng& rng, Bounce& out) {
    V3 p = add(o, scl(t_hit, d));                                         // ray.h:21-23
    V3 outward = m.normal;
    bool front_face = dot(d, outward) < 0;                                // hittable.h:20-26
    V3 normal = front_face ? outward : neg(outward);
    shade_surface<EYE, LAMBERT>(m, tri, p, normal, front_face, d, throughput, rng, out);
}

template <bool EYE, bool LAMBERT, bool MEDIA>
BDPT_DEV void shade_surface(const ShadeMat& m, int32_t tri, V3 p, V3 normal, bool front_face, V3 d, V3 throughput, Rng& rng, Bounce& out) {
    const int32_t mat_id = m.id;

    out.keep = (m.type == 0 || m.type == 3 || (MEDIA && m.type == 4));
    out.vx.p = p; out.vx.n = normal; out.vx.T = throughput;
    out.vx.info = (uint32_t)mat_id | (m.type == 3 ? kVtxIsLight : 0u) | (front_face ? kVtxFrontFace : 0u);
    out.vx.tri = (uint32_t)tri;
    out.emitted = false;
    out.emitted_term = mk(0, 0, 0);
    out.cont = false;
    out.o = p; out.d = d; out.T = throughput;

    if (m.type == 3) {                                                    // diffuse_light: emits, never scatters (material.h:26-28)
        if (EYE && front_face) {
            V3 e = m.rgb;
            if (len2(e) > 0) { out.emitted_term = mulv(throughput, e); out.emitted = true; }
        }
        return;
    }
    if (m.type == 1) {                                                    // metal, material.h:73-83, camera.h:355-359
        V3 reflected = reflect(d, normal);
        reflected = add(unit(reflected), scl(m.fuzz, random_unit_vector(rng)));
        out.T = mulv(throughput, m.rgb);
        out.d = reflected;
        out.cont = true;
        return;
    }
    if (m.type == 2) {                                                    // dielectric, material.h:96-116
        double ri = front_face ? (1.0 / m.ior) : m.ior;
        V3 unit_direction = unit(d);
        double cos_theta = __builtin_fmin(dot(neg(unit_direction), normal), 1.0);
        double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        V3 direction;
        if (cannot_refract || reflectance(cos_theta, ri) > rng.next())   // no draw on total internal reflection
            direction = reflect(unit_direction, normal);
        else
            direction = refract(unit_direction, normal, ri);
        out.T = mulv(throughput, mk(1.0, 1.0, 1.0));
        out.d = direction;
        out.cont = true;
        return;
    }
    if (MEDIA && m.type == 4) {                                           // isotropic, material.h:154-168 + pdf.h:17-28 + camera.h:361-368
        V3 scattered = random_unit_vector(rng);
        double pdf_value = 1 / (4 * kPi);
        double scattering_pdf = 1 / (4 * kPi);
        out.T = divs(scl(scattering_pdf, mulv(throughput, m.rgb)), pdf_value);
        out.d = scattered;
        out.cont = true;
        return;
    }
    if (!LAMBERT) return;
    // lambertian, material.h:47-58 + pdf.h:32-47 + camera.h:361-368
    V3 w;
    V3 scattered = onb_transform(normal, random_cosine_direction(rng), w);
    V3 su = unit(scattered);
    double pdf_value = __builtin_fmax(0.0, dot(su, w) / kPi);
    if (pdf_value <= 0) return;
    double cos_theta = dot(normal, su);
    double scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi;
    out.T = divs(scl(scattering_pdf, mulv(throughput, m.rgb)), pdf_value);
    out.d = scattered;
    out.cont = true;
}

// camera::trace_path, src/camera.h:325-370.  Stores the non-delta vertices into `store` starting
// at slot `slot0`; returns how many were stored.  `n_hits` counts ALL vertices (delta ones too).
// Eye subpaths (EYE) also return the background term (camera.h:337-338) and the emission term of
// camera.h:306-309; a path ends at the first emissive surface (material.h:26-28), so there is at
// most one such term.
template <int STATS, bool EYE, class Mem>
BDPT_DEV int trace_path(const Mem& mem, const SceneDev& sc, const GpuCamera& cam, V3 o, V3 d, int depth, V3 throughput,
                        Rng& rng, const VertexStore& store, int slot0, Counters& c, int& n_hits, V3& background_contrib,
                        V3& emitted_term, bool& has_emitted) {
    int stored = 0;
    n_hits = 0;
    for (int bounce = 0; bounce < depth; ++bounce) {
        if (STATS) c.rays_traced++;
        HitLite h;
        if (!walk<STATS, false>(mem, o, d, 0.001, kInf, h, c)) {
            if (EYE) background_contrib = add(background_contrib, mulv(throughput, ld3(cam.background)));
            break;
        }
        n_hits++;
        Bounce b;
        shade_bounce<EYE>(sc, o, d, h, throughput, rng, b);
        if (b.keep) {
            store.put(slot0 + stored, b.vx);
            stored++;
            if (STATS) c.vertices++;
        }
        if (EYE && b.emitted) { emitted_term = b.emitted_term; has_emitted = true; }
        if (!b.cont) break;
        o = b.o; d = b.d; throughput = b.T;
    }
    return stored;
}

// triangle_collection::sample_surface + triangle::sample, triangle.h:199-224,107-119.
// light_accum[i] holds the running area sum exactly as the reference's loop accumulates it, so
// "first i with pick <= accum[i], else the last triangle" is found by bisection.
BDPT_DEV int pick_light(const SceneDev& sc, double pick) {
    int lo = 0, hi = (int)sc.n_lights - 1;       // answer in [lo, hi]; hi is also the fallback
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (pick <= sc.light_accum[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// camera::connect_vertices, src/camera.h:440-475, in two halves around the visibility query.
// Geometry half (camera.h:444-453 + the ray set-up of camera::visible, camera.h:426-436): false when the
// pair contributes nothing before any ray is needed; otherwise the shadow ray and the cosine / distance terms.
struct PairGeom {
    V3 origin, dir_unit;
    double max_t, cos_cam, cos_light, dist2;
};
BDPT_DEV bool pair_geometry(const Vertex& cv, const Vertex& lv, PairGeom& g) {
    V3 dir = sub(lv.p, cv.p);
    g.dist2 = len2(dir);
    if (g.dist2 <= 0) return false;
    g.dir_unit = unit(dir);                                   // == dir / dist of camera::visible, bit for bit
    g.cos_cam = __builtin_fabs(dot(cv.n, g.dir_unit));
    g.cos_light = __builtin_fabs(dot(lv.n, neg(g.dir_unit)));
    if (g.cos_cam <= 0 || g.cos_light <= 0) return false;
    double dist = __builtin_sqrt(g.dist2);
    if (dist <= 0) return false;
    g.origin = add(cv.p, scl(0.001, g.dir_unit));
    g.max_t = dist - 0.001;
    if (g.max_t <= 0) return false;
    return true;
}
// BSDF / emission factors of the two end points (camera.h:458-469); false when either is zero
BDPT_DEV bool pair_factors(const SceneDev& sc, const Vertex& cv, const Vertex& lv, V3& f_cam, V3& f_light) {
    f_cam = evaluate_bsdf(sc.mats[cv.info & kVtxMatMask]);
    if (len2(f_cam) <= 0) return false;
    const GpuMaterial lm = sc.mats[lv.info & kVtxMatMask];
    if (lv.info & kVtxIsLight) f_light = (lv.info & kVtxFrontFace) ? ld3(lm.rgb) : mk(0, 0, 0);
    else f_light = evaluate_bsdf(lm);
    return len2(f_light) > 0;
}
// camera.h:471-474
BDPT_DEV V3 pair_contribution(const Vertex& cv, const Vertex& lv, const PairGeom& g, V3 f_cam, V3 f_light) {
    V3 contrib = mulv(cv.T, f_cam);
    contrib = mulv(mulv(contrib, lv.T), f_light);
    return divs(scl(g.cos_cam * g.cos_light, contrib), g.dist2);
}

template <int STATS, class Mem>
BDPT_DEV V3 connect(const Mem& mem, const SceneDev& sc, const Vertex& cv, const Vertex& lv, Counters& c) {
    if (STATS) c.pairs++;
    PairGeom g;
    if (!pair_geometry(cv, lv, g)) return mk(0, 0, 0);
    if (STATS) c.shadow_rays++;
    HitLite tmp;
    if (walk<STATS, STATS != kStatsReference>(mem, g.origin, g.dir_unit, 0.001, g.max_t, tmp, c)) return mk(0, 0, 0);
    V3 f_cam, f_light;
    if (!pair_factors(sc, cv, lv, f_cam, f_light)) return mk(0, 0, 0);
    return pair_contribution(cv, lv, g, f_cam, f_light);
}

// camera::get_ray, src/camera.h:199-223: two draws for the stratum offset, the defocus disk sample when
// defocus_angle > 0 (random_in_unit_disk, vec3.h:113-119: the y argument is drawn before the x argument - argument
// evaluation order of the compiler that built the pinned reference), one draw for ray_time (unused)
BDPT_DEV void camera_ray(const GpuCamera& cam, int i, int j, int s_i, int s_j, Rng& rng, V3& origin, V3& dir) {
    double px = ((s_i + rng.next()) * cam.recip_sqrt_spp) - 0.5;
    double py = ((s_j + rng.next()) * cam.recip_sqrt_spp) - 0.5;
    V3 pixel_sample = add(add(ld3(cam.pixel00), scl(i + px, ld3(cam.du))), scl(j + py, ld3(cam.dv)));
    origin = ld3(cam.center);
    if (cam.defocus_angle > 0) {                                               // camera.h:230-234
        double dx, dy;
        for (;;) {
            dy = rng.next(-1, 1);
            dx = rng.next(-1, 1);
            if (dx * dx + dy * dy + 0.0 * 0.0 < 1) break;
        }
        origin = add(add(ld3(cam.center), scl(dx, ld3(cam.disk_u))), scl(dy, ld3(cam.disk_v)));
    }
    dir = sub(pixel_sample, origin);
    (void)rng.next();                                                          // ray_time, camera.h:210
}

// First half of camera::build_light_path, src/camera.h:378-415: the emitter vertex and the first
// light ray.  `valid` is build_light_path's return value; `has_ray` is false when the sampled
// direction is not above the surface (camera.h:410-411) or no bounce is left (depth - 1 == 0).
struct Emitter {
    bool valid, has_ray;
    Vertex ev;
    V3 o, d, T;
};
// the colour of an emitter's material at a point of it: its solid colour
struct SolidEmitterColour {
    BDPT_DEV V3 operator()(const GpuMaterial& m, V3) const { return ld3(m.rgb); }
};
template <class Colour = SolidEmitterColour>
BDPT_DEV void sample_emitter(const SceneDev& sc, int depth, Rng& rng, Emitter& e, const Colour& colour = Colour()) {
    e.valid = false;
    e.has_ray = false;
    if (!(depth > 0 && sc.n_lights > 0 && sc.light_total_area > 0.0)) return;     // camera.h:378-379, triangle.h:200-208
    double pick = rng.next(0.0, sc.light_total_area);
    const int li = pick_light(sc, pick);
    const int32_t tri_id = (int32_t)sc.light_tris[li];
    double u = rng.next();
    double v = rng.next();
    if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
    const GpuTri tr = sc.tris[tri_id];
    const GpuTriAux ax = sc.aux[tri_id];
    V3 position = add(add(ld3(tr.v0), scl(u, ld3(tr.e1))), scl(v, ld3(tr.e2)));
    V3 normal = ld3(ax.normal);
    const GpuMaterial m = sc.mats[tr.material];
    V3 emission = (m.type == 3) ? colour(m, position) : mk(0, 0, 0);       // front_face = true, u = v = 0 (camera.h:385-391)
    if (!(len2(emission) > 0)) return;                                     // camera.h:395-396
    e.valid = true;
    double pdf = 1.0 / sc.light_total_area;
    double pm = (pdf < 1e-8) ? 1e-8 : pdf;                                 // std::max(sample.pdf, 1e-8)
    e.ev.p = position; e.ev.n = normal;
    e.ev.T = divs(mk(1.0, 1.0, 1.0), pm);
    e.ev.info = (uint32_t)tr.material | kVtxIsLight | kVtxFrontFace;
    e.ev.tri = (uint32_t)tri_id;
    e.ev.f = emission;
    V3 w;
    V3 ldir = onb_transform(normal, random_cosine_direction(rng), w);      // camera.h:420-423
    V3 dir_unit = unit(ldir);
    double dt = dot(normal, dir_unit);
    double cos_theta = (0.0 < dt) ? dt : 0.0;                              // std::max(0.0, dot)
    if (!(cos_theta > 0)) return;
    double q = cos_theta / kPi;
    double pdf_dir = (q < 1e-8) ? 1e-8 : q;                                // std::max(cos/pi, 1e-8)
    e.T = scl(cos_theta / pdf_dir, mulv(e.ev.T, emission));
    e.o = add(position, scl(0.001, normal));
    e.d = dir_unit;
    e.has_ray = true;
}

// camera::get_ray + ray_color -> bidirectional_color for one sample: src/camera.h:199-223,294-323,372-418
template <int STATS, class Mem>
BDPT_DEV V3 sample_radiance(const Mem& mem, const SceneDev& sc, const GpuCamera& cam, int i, int j, int s_i, int s_j,
                            uint64_t seed, const VertexStore& store, Counters& c, uint32_t& draws) {
    Rng rng;
    rng.key = stream_key(seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s_j * cam.sqrt_spp + s_i);
    rng.counter = 0;
    const int depth = cam.max_depth;
    V3 center, dir;
    camera_ray(cam, i, j, s_i, s_j, rng, center, dir);

    // eye subpath, camera.h:300-309
    V3 background_contrib = mk(0, 0, 0), emitted_term = mk(0, 0, 0);
    bool has_emitted = false;
    int eye_hits = 0, light_hits = 0;
    int n_eye = trace_path<STATS, true>(mem, sc, cam, center, dir, depth, mk(1, 1, 1), rng, store, 0, c, eye_hits,
                                        background_contrib, emitted_term, has_emitted);
    V3 result = background_contrib;
    if (has_emitted) result = add(result, emitted_term);

    // light subpath, camera.h:372-418
    int n_light = 0;
    Emitter em;
    sample_emitter(sc, depth, rng, em);
    if (em.valid) {
        store.put(depth, em.ev);
        if (STATS) c.vertices++;
        n_light = 1;
        light_hits = 1;
        if (em.has_ray) {
            V3 dummy_bg = mk(0, 0, 0), dummy_e = mk(0, 0, 0);
            bool dummy_flag = false;
            int lh = 0;
            n_light += trace_path<STATS, false>(mem, sc, cam, em.o, em.d, depth - 1, em.T, rng, store, depth + 1, c, lh,
                                                dummy_bg, dummy_e, dummy_flag);
            light_hits += lh;
        }
    }
    draws = rng.counter;
    if (!em.valid) return result;

    // connections, camera.h:316-320.  The reference adds a (possibly zero) colour for every pair;
    // adding +0 only matters for a -0 accumulator, which the first add below normalises.
    if (eye_hits > 0 && light_hits > 0) result = add(result, mk(0, 0, 0));
    for (int a = 0; a < n_eye; a++) {
        const Vertex cv = store.get(a);
        for (int b = 0; b < n_light; b++) {
            const Vertex lv = store.get(depth + b);
            result = add(result, connect<STATS>(mem, sc, cv, lv, c));
        }
    }
    return result;
}

// ------------------------------------------------------------------ unidirectional estimator
// triangle::pdf_value, triangle.h:78-91: a hit test of this ONE triangle, not of the world
template <int STATS>
BDPT_DEV double light_tri_pdf(const SceneDev& sc, uint32_t tri_id, V3 origin, V3 direction, Counters& c) {
    const GpuTri tr = sc.tris[tri_id];
    double t, u, v;
    if (STATS) c.tri_tests++;
    if (!tri_test(tr, origin, direction, 0.001, kInf, t, u, v)) return 0.0;
    if (STATS) c.tri_hits++;
    const GpuTriAux ax = sc.aux[tri_id];
    V3 outward = ld3(ax.normal);
    V3 normal = (dot(direction, outward) < 0) ? outward : neg(outward);          // hittable.h:20-26
    double distance_squared = t * t * len2(direction);
    double cosine = __builtin_fabs(dot(direction, normal) / len(direction));
    if (ax.area <= 0.0 || cosine <= 0.0) return 0.0;
    return distance_squared / (cosine * ax.area);
}

// camera::path_trace_color, src/camera.h:255-292, with mixture_pdf(hittable_pdf(lights), cosine_pdf) of
// src/acceleration/pdf.h:32-90 and triangle_collection::pdf_value / random (triangle.h:170-189).  The
// reference recurses and combines on the way back, `emission + attenuation * scattering_pdf * sample / pdf`
// (or `attenuation * sample` for metal and dielectric); here the walk goes forward storing one record per
// bounce (factor and pdf) in the vertex workspace, and the same expressions are folded from the last
// bounce to the first.
//
// One level of that recursion after its hit was found: either the value the level returns without going deeper
// (an emitter; a lambertian surface whose mixture pdf came out as zero), or the record of the level (p = factor,
// n.x = pdf, 0 marking `attenuation * sample`) and the ray of the next level.
struct UniBounce {
    bool cont;
    V3 value;         // when !cont
    Vertex rec;       // when cont
    V3 o, d;
};
template <int STATS>
BDPT_DEV void unidirectional_bounce(const SceneDev& sc, V3 o, V3 d, const HitLite& h, Rng& rng, Counters& c, UniBounce& out) {
    const GpuTriAux ax = sc.aux[h.tri];
    const GpuMaterial m = sc.mats[sc.tris[h.tri].material];
    V3 p = add(o, scl(h.t, d));
    V3 outward = ld3(ax.normal);
    bool front_face = dot(d, outward) < 0;
    V3 normal = front_face ? outward : neg(outward);
    out.cont = false;
    out.value = mk(0, 0, 0);
    if (m.type == 3) { out.value = front_face ? ld3(m.rgb) : mk(0, 0, 0); return; }      // emitted, no scatter
    out.rec.n = mk(0, 0, 0); out.rec.T = mk(0, 0, 0); out.rec.info = 0; out.rec.tri = 0;
    if (m.type == 1 || m.type == 2) {
        Bounce bo;
        shade_bounce<true>(sc, o, d, h, mk(1.0, 1.0, 1.0), rng, bo);      // bo.T = 1 * attenuation = attenuation
        out.rec.p = bo.T;
        out.o = bo.o; out.d = bo.d;
        out.cont = true;
        return;
    }
    // lambertian: direction from the mixture, pdf.h:73-87
    V3 w;
    (void)onb_transform(normal, mk(0, 0, 0), w);
    V3 direction;
    if (rng.next() < 0.5) {
        // triangle_collection::random: random_int(0, size - 1) = int(0 + (size - 0) * random_double()), main.h:34-42
        int idx = (int)(0.0 + ((double)(int)sc.n_lights - 0.0) * rng.next());
        if (idx > (int)sc.n_lights - 1) idx = (int)sc.n_lights - 1;       // (out of range in the reference)
        const GpuTri lt = sc.tris[sc.light_tris[idx]];
        double u = rng.next();
        double v = rng.next();
        if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
        V3 lp = add(add(ld3(lt.v0), scl(u, ld3(lt.e1))), scl(v, ld3(lt.e2)));
        direction = sub(lp, p);
    } else {
        V3 w2;
        direction = onb_transform(normal, random_cosine_direction(rng), w2);
    }
    // triangle_collection::pdf_value: every light triangle, in order
    double weight = 1.0 / (double)sc.n_lights;
    double sum = 0.0;
    for (uint32_t k = 0; k < sc.n_lights; k++) sum += weight * light_tri_pdf<STATS>(sc, sc.light_tris[k], p, direction, c);
    double cosine_theta = dot(unit(direction), w);
    double pdf_value = 0.5 * sum + 0.5 * __builtin_fmax(0.0, cosine_theta / kPi);
    if (pdf_value <= 0) return;               // returns the (zero) emission of a lambertian surface
    double cos_theta = dot(normal, unit(direction));
    double scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi;
    out.rec.p = scl(scattering_pdf, ld3(m.rgb));
    out.rec.n.x = pdf_value;
    out.o = p; out.d = direction;
    out.cont = true;
}
// the value a level returns given its record and the value of the level below
BDPT_DEV V3 unidirectional_fold(const Vertex& rec, V3 below) {
    if (rec.n.x == 0.0) return mulv(rec.p, below);
    return add(mk(0, 0, 0), divs(mulv(rec.p, below), rec.n.x));
}

template <int STATS, class Mem>
BDPT_DEV V3 sample_radiance_unidirectional(const Mem& mem, const SceneDev& sc, const GpuCamera& cam, int i, int j, int s_i,
                                           int s_j, uint64_t seed, const VertexStore& store, Counters& c, uint32_t& draws) {
    Rng rng;
    rng.key = stream_key(seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s_j * cam.sqrt_spp + s_i);
    rng.counter = 0;
    V3 o, d;
    camera_ray(cam, i, j, s_i, s_j, rng, o, d);
    int n = 0;                                  // bounces recorded (at most max_depth <= workspace slots)
    V3 result = mk(0, 0, 0);                    // value of the innermost call
    for (int remaining = cam.max_depth;; remaining--) {
        if (STATS) c.rays_traced++;
        if (remaining <= 0) break;              // camera.h:258-259
        HitLite h;
        if (!walk<STATS, false>(mem, o, d, 0.001, kInf, h, c)) { result = ld3(cam.background); break; }
        UniBounce ub;
        unidirectional_bounce<STATS>(sc, o, d, h, rng, c, ub);
        if (!ub.cont) { result = ub.value; break; }
        store.put(n++, ub.rec);
        o = ub.o; d = ub.d;
    }
    for (int k = n - 1; k >= 0; k--) result = unidirectional_fold(store.get(k), result);
    draws = rng.counter;
    return result;
}

}  // namespace bdpt
#endif
