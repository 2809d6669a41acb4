// Device-side walk over a composite scene's trees in the reference's own visiting order (composite.h): the world's
// bvh_node tree whose leaves are triangles or objects, and one tree per mesh.  Shared by the single-kernel renderer of
// composite scenes (composite.hip) and by the pipeline's kernels (wavefront.hip), which use it for exact ties and, in a
// scene with media, for every ray; the texture routines (src/materials/textures) are shared the same way.
// `Sc` is any scene view with members nodes, tris, aux, objs (and tex, perlin, image_bytes for the textures).  Citations are to the reference renderer's source tree.
#ifndef BDPT_COMPOSITE_CORE_H
#define BDPT_COMPOSITE_CORE_H

#include "../../include/bdpt_amd.h"
#include "bdpt_types.h"
#include "device_core.h"

namespace bdpt {

struct CxHit {
    double t, u, v;
    int32_t tri;      // >= 0: a triangle (obj = the object it belongs to, or -1); < 0: a point inside the medium `obj`
    int32_t obj;
};

// bvh_node::hit (src/acceleration/bvh.h:50-59) over ONE tree of triangles, starting at record `root`
// ANYHIT (visibility rays of a scene without media, not counting): returns at the first accepted triangle - the
// boolean the reference derives from its full closest-hit query (camera.h:437) is the same.
template <int STATS, bool ANYHIT = false, class Sc>
BDPT_DEV bool cx_walk_mesh(const Sc& sc, uint32_t root, V3 o, V3 d, double tmin, double tmax, CxHit& hit, Counters& c) {
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;                 // aabb.h:48
    double closest = tmax;
    bool any = false;
    uint32_t i = root;
    while (i != kEndOfTree) {
        const GpuNode nd = sc.nodes[i];
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
        if (hi <= lo) { i = nd.skip; continue; }
        if (STATS) c.aabb_hits++;
        if (nd.tri0 >= 0) {
            double t, u, v;
            {
                const bool h = tri_test(sc.tris[nd.tri0], o, d, tmin, closest, t, u, v);
                const bool twice = (nd.tri1 == nd.tri0);             // one-triangle node: tested twice, same outcome
                if (STATS) { c.tri_tests += twice ? 2 : 1; if (h) c.tri_hits += twice ? 2 : 1; }
                if (h) { if (ANYHIT) return true; closest = t; hit.t = t; hit.u = u; hit.v = v; hit.tri = nd.tri0; any = true; }
            }
            if (nd.tri1 != nd.tri0) {
                const bool h = tri_test(sc.tris[nd.tri1], o, d, tmin, closest, t, u, v);
                if (STATS) { c.tri_tests++; if (h) c.tri_hits++; }
                if (h) { if (ANYHIT) return true; closest = t; hit.t = t; hit.u = u; hit.v = v; hit.tri = nd.tri1; any = true; }
            }
        }
        i = nd.next;
    }
    return any;
}

// the ray as the object's mesh sees it: translate::hit (hittable.h:56) is the outer call, rotate_y::hit (:114-126) the inner
BDPT_DEV void cx_to_object(const CxObject& ob, V3& o, V3& d) {
    if (ob.flags & kCxTranslate) o = sub(o, ld3(ob.offset));
    if (ob.flags & kCxRotate) {
        const double c = ob.cos_theta, s = ob.sin_theta;
        o = mk((c * o.x) - (s * o.z), o.y, (s * o.x) + (c * o.z));
        d = mk((c * d.x) - (s * d.z), d.y, (s * d.x) + (c * d.z));
    }
}

// hit() of one object of the world's list
// MEDIA = false: for scenes known to hold no medium (the medium's code, its logarithm included, is left out)
template <int STATS, bool ANYHIT = false, bool MEDIA = true, class Sc>
BDPT_DEV bool cx_object_hit(const Sc& sc, int32_t k, V3 o, V3 d, double tmin, double tmax, CxHit& hit, Counters& c, Rng& rng) {
    const CxObject ob = sc.objs[k];
    V3 oo = o, od = d;
    cx_to_object(ob, oo, od);
    if (!MEDIA || !(ob.flags & kCxMedium)) {
        CxHit h;
        if (!cx_walk_mesh<STATS, ANYHIT>(sc, ob.root, oo, od, tmin, tmax, h, c)) return false;
        hit = h;
        hit.obj = k;
        return true;
    }
    // constant_medium::hit, constant_medium.h:20-53
    CxHit h1, h2;
    if (!cx_walk_mesh<STATS>(sc, ob.root, oo, od, -kInf, kInf, h1, c)) return false;
    if (!cx_walk_mesh<STATS>(sc, ob.root, oo, od, h1.t + 0.0001, kInf, h2, c)) return false;
    double t1 = h1.t, t2 = h2.t;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const double ray_length = len(d);
    const double distance_inside_boundary = (t2 - t1) * ray_length;
    const double hit_distance = ob.neg_inv_density * det_log(rng.next());
    if (hit_distance > distance_inside_boundary) return false;
    hit.t = t1 + hit_distance / ray_length;
    hit.u = 0.0; hit.v = 0.0;
    hit.tri = -1;
    hit.obj = k;
    return true;
}

// bvh_node::hit over the world's tree: leaves are triangles or objects
template <int STATS, bool ANYHIT = false, bool MEDIA = true, class Sc>
BDPT_DEV bool cx_walk(const Sc& sc, V3 o, V3 d, double tmin, double tmax, CxHit& hit, Counters& c, Rng& rng) {
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
    double closest = tmax;
    bool any = false;
    uint32_t i = 0;
    while (i != kEndOfTree) {
        const GpuNode nd = sc.nodes[i];
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
        if (hi <= lo) { i = nd.skip; continue; }
        if (STATS) c.aabb_hits++;
        if (nd.tri0 != -1) {
            // left->hit(r, ray_t, rec), then right->hit(r, (min, hit_left ? rec.t : max), rec); a node over one entry
            // holds it twice and asks it twice (bvh.h:35-36) - a medium then draws a second number
            for (int side = 0; side < 2; side++) {
                const int32_t ref = side ? nd.tri1 : nd.tri0;
                if (ref >= 0) {
                    double t, u, v;
                    const bool h = tri_test(sc.tris[ref], o, d, tmin, closest, t, u, v);
                    if (STATS) { c.tri_tests++; if (h) c.tri_hits++; }
                    if (h) { if (ANYHIT) return true; closest = t; hit.t = t; hit.u = u; hit.v = v; hit.tri = ref; hit.obj = -1; any = true; }
                } else {
                    CxHit h;
                    if (cx_object_hit<STATS, ANYHIT, MEDIA>(sc, -(ref + 2), o, d, tmin, closest, h, c, rng)) { if (ANYHIT) return true; closest = h.t; hit = h; any = true; }
                }
            }
        }
        i = nd.next;
    }
    return any;
}

// What the hit_record holds once the outermost hit() has returned
struct CxSurf {
    V3 p, n;
    bool front;
    int32_t mat;
    double u, v;
};
template <class Sc>
BDPT_DEV void cx_surface(const Sc& sc, V3 o, V3 d, const CxHit& h, CxSurf& s) {
    s.u = h.u; s.v = h.v;
    if (h.tri < 0) {                                          // constant_medium.h:45-50
        s.p = add(o, scl(h.t, d));
        s.n = mk(1, 0, 0);
        s.front = true;
        s.mat = sc.objs[h.obj].phase_material;
        return;
    }
    V3 oo = o, od = d;
    CxObject ob;
    if (h.obj >= 0) { ob = sc.objs[h.obj]; cx_to_object(ob, oo, od); }
    V3 p = add(oo, scl(h.t, od));                             // triangle.h:67, in the mesh's own coordinates
    const V3 outward = ld3(sc.aux[h.tri].normal);
    s.front = dot(od, outward) < 0;                           // hittable.h:20-26
    V3 n = s.front ? outward : neg(outward);
    if (h.obj >= 0) {
        if (ob.flags & kCxRotate) {                           // hittable.h:135-145
            const double c = ob.cos_theta, sn = ob.sin_theta;
            p = mk((c * p.x) + (sn * p.z), p.y, (-sn * p.x) + (c * p.z));
            n = mk((c * n.x) + (sn * n.z), n.y, (-sn * n.x) + (c * n.z));
        }
        if (ob.flags & kCxTranslate) p = add(p, ld3(ob.offset));      // hittable.h:63
    }
    s.p = p; s.n = n;
    s.mat = sc.tris[h.tri].material;
}

// ------------------------------------------------------------------ textures (src/materials/textures)
// perlin::noise + perlin_interp, perlin.h:16-36,75-92
BDPT_DEV double cx_perlin_noise(const CxPerlin& pn, V3 p) {
    const double fx = __builtin_floor(p.x), fy = __builtin_floor(p.y), fz = __builtin_floor(p.z);
    const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const double uu = u * u * (3 - 2 * u);
    const double vv = v * v * (3 - 2 * v);
    const double ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                const int idx = pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255];
                const V3 cv = mk(pn.randvec[idx][0], pn.randvec[idx][1], pn.randvec[idx][2]);
                const V3 weight_v = mk(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) *
                         dot(cv, weight_v);
            }
    return accum;
}
// perlin::turb, perlin.h:38-50
BDPT_DEV double cx_perlin_turb(const CxPerlin& pn, V3 p, int depth) {
    double accum = 0.0;
    V3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * cx_perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = mk(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2);
    }
    return __builtin_fabs(accum);
}

// texture::value(u, v, p), texture.h
// `Sc`: any scene view with members tex, perlin, image_bytes
template <class Sc>
__device__ __noinline__ V3 cx_texture(const Sc& sc, int32_t id, double u, double v, V3 p) {
    for (;;) {
        const CxTexture t = sc.tex[id];
        if (t.type == BDPT_TEX_CHECKER) {                                  // texture.h:36-44 (scale holds inv_scale)
            const int xi = (int)__builtin_floor(t.scale * p.x);
            const int yi = (int)__builtin_floor(t.scale * p.y);
            const int zi = (int)__builtin_floor(t.scale * p.z);
            id = ((xi + yi + zi) % 2 == 0) ? t.a : t.b;
            continue;
        }
        if (t.type == BDPT_TEX_NOISE) {                                    // texture.h:80-82
            const double arg = t.scale * p.z + 10 * cx_perlin_turb(sc.perlin[t.data], p, 7);
            double s, c;
            det_sincos(arg, s, c);
            return scl(1 + s, mk(.5, .5, .5));
        }
        if (t.type == BDPT_TEX_IMAGE) {                                    // texture.h:56-70, rtw_stb_image.h:63-73
            if (t.b <= 0) return mk(0, 1, 1);
            u = (u < 0.0) ? 0.0 : ((u > 1.0) ? 1.0 : u);
            const double vc = (v < 0.0) ? 0.0 : ((v > 1.0) ? 1.0 : v);
            v = 1.0 - vc;
            int i = (int)(u * t.a);
            int j = (int)(v * t.b);
            i = (i < 0) ? 0 : ((i < t.a) ? i : t.a - 1);
            j = (j < 0) ? 0 : ((j < t.b) ? j : t.b - 1);
            const uint8_t* px = sc.image_bytes + t.data + ((size_t)j * (size_t)t.a + (size_t)i) * 3;
            const double color_scale = 1.0 / 255.0;
            return mk(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
        }
        return ld3(t.rgb);                                                 // solid_color
    }
}
template <class Sc>
BDPT_DEV V3 cx_colour(const Sc& sc, const GpuMaterial& m, double u, double v, V3 p) {
    return (m.pad > 0) ? cx_texture(sc, m.pad - 1, u, v, p) : ld3(m.rgb);
}

}  // namespace bdpt
#endif
