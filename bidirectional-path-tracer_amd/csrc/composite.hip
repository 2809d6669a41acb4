// Single-kernel renderer and query kernels for composite scenes (composite.h): instances (translate / rotate_y),
// textures and constant-density media.  (The render pipeline of wavefront.hip renders them too; this path serves the
// counting mode, the unidirectional estimator and the queries of scenes with media.)
// One thread renders one sample from start to end and walks the reference's own trees in the reference's order:
// constant_medium::hit draws a random number every time the walk reaches a medium whose boundary the ray crosses
// (src/materials/volumes/constant_medium.h:40), so the visiting order, the running upper bound of the interval and
// even the full closest-hit traversal behind camera::visible (src/camera.h:437) are part of the result.
// Citations are to the reference renderer's source tree.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "composite.h"
#include "composite_core.h"
#include "device_core.h"

namespace bdpt {
namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ path vertices
// What connect_vertices reads of a path_vertex (camera.h:236-243): position, normal, throughput and `f`, the one
// colour the vertex can contribute with: its emission if it lies on an emitter (camera.h:463-464), the value of its
// BSDF otherwise (texture lookups are functions of the hit alone, so the value is formed when the vertex is).
constexpr uint32_t kCxIsLight = 1u;
struct CxVertex {
    V3 p, n, T, f;
    uint32_t flags;
};
struct CxVertexStore {
    double* ws;
    uint64_t stride;
    uint64_t g;
    BDPT_DEV void put(int slot, const CxVertex& v) const {
        double* b = ws + (uint64_t)slot * kCxVtxDoubles * stride + g;
        b[0] = v.p.x; b[stride] = v.p.y; b[2 * stride] = v.p.z;
        b[3 * stride] = v.n.x; b[4 * stride] = v.n.y; b[5 * stride] = v.n.z;
        b[6 * stride] = v.T.x; b[7 * stride] = v.T.y; b[8 * stride] = v.T.z;
        b[9 * stride] = v.f.x; b[10 * stride] = v.f.y; b[11 * stride] = v.f.z;
        b[12 * stride] = __longlong_as_double((long long)v.flags);
    }
    BDPT_DEV CxVertex get(int slot) const {
        const double* b = ws + (uint64_t)slot * kCxVtxDoubles * stride + g;
        CxVertex v;
        v.p = mk(b[0], b[stride], b[2 * stride]);
        v.n = mk(b[3 * stride], b[4 * stride], b[5 * stride]);
        v.T = mk(b[6 * stride], b[7 * stride], b[8 * stride]);
        v.f = mk(b[9 * stride], b[10 * stride], b[11 * stride]);
        v.flags = (uint32_t)__double_as_longlong(b[12 * stride]);
        return v;
    }
};

// One iteration of camera::trace_path (camera.h:342-368) after a hit was found
struct CxBounce {
    bool keep;            // non-delta vertex (camera.h:441 drops the others)
    CxVertex vx;
    bool emitted;         // eye subpath only: camera.h:306-309
    V3 emitted_term;
    bool cont;
    V3 o, d, T;
};
template <bool EYE>
BDPT_DEV void cx_bounce(const CxSceneDev& sc, V3 o, V3 d, const CxHit& h, V3 throughput, Rng& rng, CxBounce& out) {
    CxSurf s;
    cx_surface(sc, o, d, h, s);
    const GpuMaterial m = sc.mats[s.mat];
    out.keep = (m.type == BDPT_LAMBERTIAN || m.type == BDPT_DIFFUSE_LIGHT || m.type == BDPT_ISOTROPIC);
    out.vx.p = s.p; out.vx.n = s.n; out.vx.T = throughput; out.vx.f = mk(0, 0, 0);
    out.vx.flags = (m.type == BDPT_DIFFUSE_LIGHT) ? kCxIsLight : 0u;
    out.emitted = false;
    out.emitted_term = mk(0, 0, 0);
    out.cont = false;
    out.o = s.p; out.d = d; out.T = throughput;

    if (m.type == BDPT_DIFFUSE_LIGHT) {                                   // material.h:138-143, no scatter (:26-28)
        const V3 e = s.front ? cx_colour(sc, m, s.u, s.v, s.p) : mk(0, 0, 0);
        out.vx.f = e;
        if (EYE && len2(e) > 0) { out.emitted_term = mulv(throughput, e); out.emitted = true; }
        return;
    }
    if (m.type == BDPT_METAL) {                                           // material.h:73-83
        V3 reflected = reflect(d, s.n);
        reflected = add(unit(reflected), scl(m.fuzz, random_unit_vector(rng)));
        out.T = mulv(throughput, ld3(m.rgb));
        out.d = reflected;
        out.cont = true;
        return;
    }
    if (m.type == BDPT_DIELECTRIC) {                                      // material.h:96-116
        const double ri = s.front ? (1.0 / m.ior) : m.ior;
        const V3 unit_direction = unit(d);
        const double cos_theta = __builtin_fmin(dot(neg(unit_direction), s.n), 1.0);
        const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        const bool cannot_refract = ri * sin_theta > 1.0;
        V3 direction;
        if (cannot_refract || reflectance(cos_theta, ri) > rng.next()) direction = reflect(unit_direction, s.n);
        else direction = refract(unit_direction, s.n, ri);
        out.T = mulv(throughput, mk(1.0, 1.0, 1.0));
        out.d = direction;
        out.cont = true;
        return;
    }
    const V3 attenuation = cx_colour(sc, m, s.u, s.v, s.p);
    if (m.type == BDPT_ISOTROPIC) {                                       // material.h:154-168, pdf.h:17-28
        out.vx.f = divs(attenuation, 4 * kPi);
        const V3 scattered = random_unit_vector(rng);
        const double pdf_value = 1 / (4 * kPi);
        const double scattering_pdf = 1 / (4 * kPi);
        out.T = divs(scl(scattering_pdf, mulv(throughput, attenuation)), pdf_value);
        out.d = scattered;
        out.cont = true;
        return;
    }
    // lambertian, material.h:47-63 + pdf.h:32-47 + camera.h:361-368
    out.vx.f = divs(attenuation, kPi);
    V3 w;
    const V3 scattered = onb_transform(s.n, random_cosine_direction(rng), w);
    const V3 su = unit(scattered);
    const double pdf_value = __builtin_fmax(0.0, dot(su, w) / kPi);
    if (pdf_value <= 0) return;
    const double cos_theta = dot(s.n, su);
    const double scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi;
    out.T = divs(scl(scattering_pdf, mulv(throughput, attenuation)), pdf_value);
    out.d = scattered;
    out.cont = true;
}

// camera::trace_path, camera.h:325-370
template <int STATS, bool EYE>
BDPT_DEV int cx_trace_path(const CxSceneDev& sc, const GpuCamera& cam, V3 o, V3 d, int depth, V3 throughput, Rng& rng,
                           const CxVertexStore& store, int slot0, Counters& c, int& n_hits, V3& background_contrib,
                           V3& emitted_term, bool& has_emitted) {
    int stored = 0;
    n_hits = 0;
    for (int bounce = 0; bounce < depth; ++bounce) {
        if (STATS) c.rays_traced++;
        CxHit h;
        if (!cx_walk<STATS>(sc, o, d, 0.001, kInf, h, c, rng)) {
            if (EYE) background_contrib = add(background_contrib, mulv(throughput, ld3(cam.background)));
            break;
        }
        n_hits++;
        CxBounce b;
        cx_bounce<EYE>(sc, o, d, h, throughput, rng, b);
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

// camera::connect_vertices, camera.h:440-475.  The visibility query is the reference's: a full closest-hit traversal
// of the world (camera.h:437) - a medium on the way draws from the sample's stream.
template <int STATS>
BDPT_DEV V3 cx_connect(const CxSceneDev& sc, const CxVertex& cv, const CxVertex& lv, Counters& c, Rng& rng) {
    if (STATS) c.pairs++;
    const V3 dir = sub(lv.p, cv.p);
    const double dist2 = len2(dir);
    if (dist2 <= 0) return mk(0, 0, 0);
    const V3 dir_unit = unit(dir);
    const double cos_cam = __builtin_fabs(dot(cv.n, dir_unit));
    const double cos_light = __builtin_fabs(dot(lv.n, neg(dir_unit)));
    if (cos_cam <= 0 || cos_light <= 0) return mk(0, 0, 0);
    // camera::visible, camera.h:425-438 (dir / dist == unit_vector(dir) bit for bit)
    const double dist = __builtin_sqrt(dist2);
    if (dist <= 0) return mk(0, 0, 0);
    const V3 origin = add(cv.p, scl(0.001, dir_unit));
    const double max_t = dist - 0.001;
    if (max_t <= 0) return mk(0, 0, 0);
    if (STATS) c.shadow_rays++;
    CxHit tmp;
    const bool blocked = (STATS == kStatsOff && sc.n_media == 0) ? cx_walk<kStatsOff, true>(sc, origin, dir_unit, 0.001, max_t, tmp, c, rng)
                                                                 : cx_walk<STATS>(sc, origin, dir_unit, 0.001, max_t, tmp, c, rng);
    if (blocked) return mk(0, 0, 0);
    const V3 f_cam = (cv.flags & kCxIsLight) ? mk(0, 0, 0) : cv.f;          // evaluate_bsdf of an emitter, material.h:35-37
    if (len2(f_cam) <= 0) return mk(0, 0, 0);
    const V3 f_light = lv.f;
    if (len2(f_light) <= 0) return mk(0, 0, 0);
    V3 contrib = mulv(cv.T, f_cam);
    contrib = mulv(mulv(contrib, lv.T), f_light);
    return divs(scl(cos_cam * cos_light, contrib), dist2);
}

// camera::get_ray + bidirectional_color for one sample, camera.h:199-223,294-323,372-418
template <int STATS>
BDPT_DEV V3 cx_sample_radiance(const CxSceneDev& sc, const GpuCamera& cam, int i, int j, int s_i, int s_j, uint64_t seed,
                               const CxVertexStore& store, Counters& c, uint32_t& draws) {
    Rng rng;
    rng.key = stream_key(seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s_j * cam.sqrt_spp + s_i);
    rng.counter = 0;
    const int depth = cam.max_depth;
    V3 center, dir;
    camera_ray(cam, i, j, s_i, s_j, rng, center, dir);

    V3 background_contrib = mk(0, 0, 0), emitted_term = mk(0, 0, 0);
    bool has_emitted = false;
    int eye_hits = 0, light_hits = 0;
    const int n_eye = cx_trace_path<STATS, true>(sc, cam, center, dir, depth, mk(1, 1, 1), rng, store, 0, c, eye_hits,
                                                 background_contrib, emitted_term, has_emitted);
    V3 result = background_contrib;
    if (has_emitted) result = add(result, emitted_term);

    // build_light_path, camera.h:372-418
    int n_light = 0;
    bool valid = false;
    if (depth > 0 && sc.n_lights > 0 && sc.light_total_area > 0.0) {
        const double pick = rng.next(0.0, sc.light_total_area);                 // triangle.h:210-219
        int lo = 0, hi = (int)sc.n_lights - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pick <= sc.light_accum[mid]) hi = mid; else lo = mid + 1;
        }
        const int32_t tri_id = (int32_t)sc.light_tris[lo];
        double u = rng.next();
        double v = rng.next();
        if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
        const GpuTri tr = sc.tris[tri_id];
        const V3 position = add(add(ld3(tr.v0), scl(u, ld3(tr.e1))), scl(v, ld3(tr.e2)));
        const V3 normal = ld3(sc.aux[tri_id].normal);
        const GpuMaterial m = sc.mats[tr.material];
        const V3 emission = (m.type == BDPT_DIFFUSE_LIGHT) ? cx_colour(sc, m, 0, 0, position) : mk(0, 0, 0);   // camera.h:389-394
        if (len2(emission) > 0) {
            valid = true;
            const double pdf = 1.0 / sc.light_total_area;
            const double pm = (pdf < 1e-8) ? 1e-8 : pdf;
            CxVertex ev;
            ev.p = position; ev.n = normal; ev.T = divs(mk(1.0, 1.0, 1.0), pm); ev.f = emission; ev.flags = kCxIsLight;
            store.put(depth, ev);
            if (STATS) c.vertices++;
            n_light = 1;
            light_hits = 1;
            V3 w;
            const V3 ldir = onb_transform(normal, random_cosine_direction(rng), w);
            const V3 dir_unit = unit(ldir);
            const double dt = dot(normal, dir_unit);
            const double cos_theta = (0.0 < dt) ? dt : 0.0;
            if (cos_theta > 0 && depth - 1 > 0) {
                const double q = cos_theta / kPi;
                const double pdf_dir = (q < 1e-8) ? 1e-8 : q;
                const V3 T = scl(cos_theta / pdf_dir, mulv(ev.T, emission));
                V3 dummy_bg = mk(0, 0, 0), dummy_e = mk(0, 0, 0);
                bool dummy_flag = false;
                int lh = 0;
                n_light += cx_trace_path<STATS, false>(sc, cam, add(position, scl(0.001, normal)), dir_unit, depth - 1, T, rng,
                                                       store, depth + 1, c, lh, dummy_bg, dummy_e, dummy_flag);
                light_hits += lh;
            }
        }
    }
    if (!valid) { draws = rng.counter; return result; }

    // camera.h:316-320 (the reference adds a colour for every pair, zero or not: see sample_radiance in device_core.h)
    if (eye_hits > 0 && light_hits > 0) result = add(result, mk(0, 0, 0));
    for (int a = 0; a < n_eye; a++) {
        const CxVertex cv = store.get(a);
        for (int b = 0; b < n_light; b++) {
            const CxVertex lv = store.get(depth + b);
            result = add(result, cx_connect<STATS>(sc, cv, lv, c, rng));
        }
    }
    draws = rng.counter;
    return result;
}

// ------------------------------------------------------------------ unidirectional estimator
// triangle::pdf_value, triangle.h:78-91, of one triangle of the light collection
template <int STATS>
BDPT_DEV double cx_light_tri_pdf(const CxSceneDev& sc, uint32_t tri_id, V3 origin, V3 direction, Counters& c) {
    double t, u, v;
    if (STATS) c.tri_tests++;
    if (!tri_test(sc.tris[tri_id], origin, direction, 0.001, kInf, t, u, v)) return 0.0;
    if (STATS) c.tri_hits++;
    const GpuTriAux ax = sc.aux[tri_id];
    const V3 outward = ld3(ax.normal);
    const V3 normal = (dot(direction, outward) < 0) ? outward : neg(outward);
    const double distance_squared = t * t * len2(direction);
    const double cosine = __builtin_fabs(dot(direction, normal) / len(direction));
    if (ax.area <= 0.0 || cosine <= 0.0) return 0.0;
    return distance_squared / (cosine * ax.area);
}

// camera::path_trace_color, camera.h:255-292, as in sample_radiance_unidirectional of device_core.h: the walk goes
// forward keeping one record per level (factor in p, pdf in n.x, 0 marking `attenuation * sample`) and the
// expressions of the returning recursion are folded from the last level to the first.  Beside the lambertian
// surface, the phase function of a medium mixes the light collection with its own sphere_pdf (pdf.h:17-28,70-87).
template <int STATS>
BDPT_DEV V3 cx_sample_radiance_unidirectional(const CxSceneDev& sc, const GpuCamera& cam, int i, int j, int s_i, int s_j,
                                              uint64_t seed, const CxVertexStore& store, Counters& c, uint32_t& draws) {
    Rng rng;
    rng.key = stream_key(seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s_j * cam.sqrt_spp + s_i);
    rng.counter = 0;
    V3 o, d;
    camera_ray(cam, i, j, s_i, s_j, rng, o, d);
    int n = 0;
    V3 result = mk(0, 0, 0);
    for (int remaining = cam.max_depth;; remaining--) {
        if (STATS) c.rays_traced++;
        if (remaining <= 0) break;                                        // camera.h:258-259
        CxHit h;
        if (!cx_walk<STATS>(sc, o, d, 0.001, kInf, h, c, rng)) { result = ld3(cam.background); break; }
        CxSurf s;
        cx_surface(sc, o, d, h, s);
        const GpuMaterial m = sc.mats[s.mat];
        if (m.type == BDPT_DIFFUSE_LIGHT) { result = s.front ? cx_colour(sc, m, s.u, s.v, s.p) : mk(0, 0, 0); break; }
        CxVertex rec;
        rec.n = mk(0, 0, 0); rec.T = mk(0, 0, 0); rec.f = mk(0, 0, 0); rec.flags = 0;
        if (m.type == BDPT_METAL || m.type == BDPT_DIELECTRIC) {
            CxBounce bo;
            cx_bounce<true>(sc, o, d, h, mk(1.0, 1.0, 1.0), rng, bo);      // bo.T = 1 * attenuation
            rec.p = bo.T;
            store.put(n++, rec);
            o = bo.o; d = bo.d;
            continue;
        }
        const bool iso = (m.type == BDPT_ISOTROPIC);
        const V3 attenuation = cx_colour(sc, m, s.u, s.v, s.p);
        V3 w;
        (void)onb_transform(s.n, mk(0, 0, 0), w);                          // cosine_pdf's basis (pdf.h:34)
        V3 direction;
        if (rng.next() < 0.5) {                                            // mixture_pdf::generate, pdf.h:80-85
            int idx = (int)(0.0 + ((double)(int)sc.n_lights - 0.0) * rng.next());      // triangle.h:187, main.h:34-42
            if (idx > (int)sc.n_lights - 1) idx = (int)sc.n_lights - 1;
            const GpuTri lt = sc.tris[sc.light_tris[idx]];
            double u = rng.next();
            double v = rng.next();
            if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
            const V3 lp = add(add(ld3(lt.v0), scl(u, ld3(lt.e1))), scl(v, ld3(lt.e2)));
            direction = sub(lp, s.p);
        } else if (iso) {
            direction = random_unit_vector(rng);
        } else {
            V3 w2;
            direction = onb_transform(s.n, random_cosine_direction(rng), w2);
        }
        const double weight = 1.0 / (double)sc.n_lights;                    // triangle.h:170-181
        double sum = 0.0;
        for (uint32_t k = 0; k < sc.n_lights; k++) sum += weight * cx_light_tri_pdf<STATS>(sc, sc.light_tris[k], s.p, direction, c);
        const double own = iso ? 1 / (4 * kPi) : __builtin_fmax(0.0, dot(unit(direction), w) / kPi);
        const double pdf_value = 0.5 * sum + 0.5 * own;
        if (pdf_value <= 0) break;                                         // returns the (zero) emission of the surface
        double scattering_pdf;
        if (iso) scattering_pdf = 1 / (4 * kPi);
        else { const double cos_theta = dot(s.n, unit(direction)); scattering_pdf = cos_theta < 0 ? 0 : cos_theta / kPi; }
        rec.p = scl(scattering_pdf, attenuation);
        rec.n.x = pdf_value;
        store.put(n++, rec);
        o = s.p; d = direction;
    }
    for (int k = n - 1; k >= 0; k--) {
        const CxVertex rec = store.get(k);
        result = (rec.n.x == 0.0) ? mulv(rec.p, result) : add(mk(0, 0, 0), divs(mulv(rec.p, result), rec.n.x));
    }
    draws = rng.counter;
    return result;
}

__device__ __forceinline__ void cx_flush_counters(const Counters& c, uint32_t draws, uint32_t samples, unsigned long long* stats) {
    unsigned long long v[11] = {c.rays_traced, c.aabb_tests, c.aabb_hits, c.node_visits, c.tri_tests,
                                c.tri_hits,    c.shadow_rays, c.pairs,    draws,         samples,       c.vertices};
#pragma unroll
    for (int k = 0; k < 11; k++) {
        unsigned long long x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k], x);
    }
}

// Persistent workgroups; a work unit is `pixels_per_group` consecutive owned pixels, a thread renders one sample and
// the per-pixel sum is formed in the reference's order (s_j outer, s_i inner, camera.h:118-123) by one lane per
// colour channel reading the samples from LDS.
template <int STATS, bool UNIDIRECTIONAL>
__global__ __launch_bounds__(kBlock) void cx_render_kernel(CxSceneDev sc, GpuCamera cam, CxRenderLaunch rl) {
    __shared__ double l_samples[kBlock * 3];
    __shared__ double l_carry[kBlock * 3];
    __shared__ uint32_t l_unit;

    const int spp = cam.sqrt_spp * cam.sqrt_spp;
    const int ppg = rl.pixels_per_group;
    const int row_w = rl.x1 - rl.x0;
    const long long n_owned = (long long)rl.n_rows * row_w;
    const int chunks = (spp + kBlock - 1) / kBlock;                      // > 1 only when ppg == 1

    CxVertexStore store;
    store.ws = rl.vertex_ws;
    store.stride = rl.ws_stride;
    store.g = (uint64_t)blockIdx.x * kBlock + threadIdx.x;

    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t total_draws = 0, total_samples = 0;

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) l_unit = atomicAdd(rl.unit_counter, 1u);
        __syncthreads();
        const uint32_t unit = l_unit;
        if (unit >= (uint32_t)rl.n_units) break;

        for (int chunk = 0; chunk < chunks; chunk++) {
            int g, s;
            if (chunks == 1) { g = threadIdx.x / spp; s = threadIdx.x - g * spp; }
            else { g = 0; s = chunk * kBlock + threadIdx.x; }
            const long long q = (long long)unit * ppg + g;
            const bool active = (g < ppg) && (s < spp) && (q < n_owned);
            V3 col = mk(0, 0, 0);
            if (active) {
                const int j = rl.rows[q / row_w];
                const int i = rl.x0 + (int)(q % row_w);
                const int s_j = s / cam.sqrt_spp, s_i = s - s_j * cam.sqrt_spp;
                uint32_t draws = 0;
                col = UNIDIRECTIONAL ? cx_sample_radiance_unidirectional<STATS>(sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws)
                                     : cx_sample_radiance<STATS>(sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws);
                total_draws += draws;
                total_samples++;
            }
            l_samples[threadIdx.x * 3 + 0] = col.x;
            l_samples[threadIdx.x * 3 + 1] = col.y;
            l_samples[threadIdx.x * 3 + 2] = col.z;
            __syncthreads();
            for (int idx = threadIdx.x; idx < ppg * 3; idx += kBlock) {
                const int pg = idx / 3, ch = idx - pg * 3;
                const long long pq = (long long)unit * ppg + pg;
                if (pq < n_owned) {
                    double acc = (chunk == 0) ? 0.0 : l_carry[idx];
                    const int first = (chunks == 1) ? pg * spp : 0;
                    const int count = (chunks == 1) ? spp : min(kBlock, spp - chunk * kBlock);
                    for (int k = 0; k < count; k++) acc = acc + l_samples[(first + k) * 3 + ch];
                    if (chunk == chunks - 1) {
                        const int j = rl.rows[pq / row_w];
                        const int i = rl.x0 + (int)(pq % row_w);
                        rl.fb[((size_t)j * cam.width + i) * 3 + ch] = acc;
                    } else {
                        l_carry[idx] = acc;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (STATS) cx_flush_counters(c, total_draws, total_samples, rl.stats);
}

// world.hit(r, interval(tmin, tmax), rec) for n rays; ray i draws from stream (seed 0, pixel i, sample 0)
template <int STATS>
__global__ __launch_bounds__(kBlock) void cx_trace_closest_kernel(CxSceneDev sc, const double* rays, uint64_t n, double* hits,
                                                                   unsigned long long* stats) {
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kBlock) {
        const double* in = rays + 8 * r;
        const V3 o = mk(in[0], in[1], in[2]), d = mk(in[3], in[4], in[5]);
        Rng rng;
        rng.key = stream_key(0, r, 0);
        rng.counter = 0;
        CxHit h;
        const bool hit = cx_walk<STATS>(sc, o, d, in[6], in[7], h, c, rng);
        double* out = hits + 11 * r;
        if (!hit) {
            out[0] = -1.0;
#pragma unroll
            for (int k = 1; k < 11; k++) out[k] = 0.0;
        } else {
            CxSurf s;
            cx_surface(sc, o, d, h, s);
            out[0] = (h.tri >= 0) ? (double)h.tri : -(double)(h.obj + 2);
            out[1] = h.t; out[2] = s.u; out[3] = s.v; out[4] = s.front ? 1.0 : 0.0;
            out[5] = s.p.x; out[6] = s.p.y; out[7] = s.p.z; out[8] = s.n.x; out[9] = s.n.y; out[10] = s.n.z;
        }
    }
    if (STATS) cx_flush_counters(c, 0, 0, stats);
}

// camera::visible(a, b, world), camera.h:425-438; pair i draws from stream (seed 0, pixel i, sample 0)
__global__ __launch_bounds__(kBlock) void cx_visible_kernel(CxSceneDev sc, const double* pairs, uint64_t n, uint8_t* out) {
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kBlock) {
        const double* in = pairs + 6 * r;
        const V3 a = mk(in[0], in[1], in[2]), b = mk(in[3], in[4], in[5]);
        Rng rng;
        rng.key = stream_key(0, r, 0);
        rng.counter = 0;
        uint8_t vis = 0;
        const V3 dir = sub(b, a);
        const double dist = len(dir);
        if (dist > 0) {
            const V3 dir_unit = divs(dir, dist);
            const V3 origin = add(a, scl(0.001, dir_unit));
            const double max_t = dist - 0.001;
            if (max_t > 0) {
                CxHit tmp;
                const bool blocked = (sc.n_media == 0) ? cx_walk<kStatsOff, true>(sc, origin, dir_unit, 0.001, max_t, tmp, c, rng)
                                                       : cx_walk<kStatsOff>(sc, origin, dir_unit, 0.001, max_t, tmp, c, rng);
                vis = blocked ? 0 : 1;
            }
        }
        out[r] = vis;
    }
}

}  // namespace

hipError_t cx_render_blocks_per_cu(bool stats, bool unidirectional, int* out) {
    int n = 0;
    hipError_t e;
    if (unidirectional) e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cx_render_kernel<kStatsReference, true>, kBlock, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cx_render_kernel<kStatsOff, true>, kBlock, 0);
    else e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cx_render_kernel<kStatsReference, false>, kBlock, 0)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cx_render_kernel<kStatsOff, false>, kBlock, 0);
    *out = n < 1 ? 1 : n;
    return e;
}

hipError_t cx_launch_render(const CxSceneDev& sc, const GpuCamera& cam, const CxRenderLaunch& rl, int grid_blocks, bool stats,
                            bool unidirectional, hipStream_t stream) {
    dim3 grid(grid_blocks), block(kBlock);
    if (unidirectional) {
        if (stats) hipLaunchKernelGGL((cx_render_kernel<kStatsReference, true>), grid, block, 0, stream, sc, cam, rl);
        else hipLaunchKernelGGL((cx_render_kernel<kStatsOff, true>), grid, block, 0, stream, sc, cam, rl);
    } else {
        if (stats) hipLaunchKernelGGL((cx_render_kernel<kStatsReference, false>), grid, block, 0, stream, sc, cam, rl);
        else hipLaunchKernelGGL((cx_render_kernel<kStatsOff, false>), grid, block, 0, stream, sc, cam, rl);
    }
    return hipGetLastError();
}

hipError_t cx_launch_trace_closest(const CxSceneDev& sc, const double* d_rays, uint64_t n, double* d_hits,
                                   unsigned long long* d_stats, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(kBlock);
    if (d_stats) hipLaunchKernelGGL((cx_trace_closest_kernel<kStatsReference>), grid, block, 0, stream, sc, d_rays, n, d_hits, d_stats);
    else hipLaunchKernelGGL((cx_trace_closest_kernel<kStatsOff>), grid, block, 0, stream, sc, d_rays, n, d_hits, d_stats);
    return hipGetLastError();
}

hipError_t cx_launch_visible(const CxSceneDev& sc, const double* d_pairs, uint64_t n, uint8_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cx_visible_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, sc, d_pairs, n, d_out);
    return hipGetLastError();
}

}  // namespace bdpt
