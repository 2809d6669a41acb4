// Wavefront kernels (gfx950, wave64).  See wavefront.h for the data layout.
//
// Stage kernels are grid-stride over a queue of sample slots; the two traversal kernels are
// persistent: a wave draws ray indices in chunks from a global cursor and hands them to lanes as
// they finish, so lanes whose ray ended early do not idle until the slowest ray of the wave is done.
#include <hip/hip_runtime.h>

#include "composite_core.h"
#include "device_core.h"
#include "wavefront.h"

namespace bdpt {

namespace {

constexpr int kBlock = 256;

BDPT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
BDPT_DEV uint32_t rank_in_mask(unsigned long long mask) {
    return __builtin_popcountll(mask & ((1ull << lane_id()) - 1ull));
}

// Stage kernels work on tiles of kStageTile queue entries per 256-thread workgroup, four entries per
// thread, and append to the output queue with ONE atomic per tile: every wave contributes its counts
// through LDS.  (A single queue counter saturates at under 100 atomics per microsecond, so a queue fed
// by one atomic per wave costs ~0.7 ms per 4 M entries.)  The workgroups stay small so that they fit
// beside the resident waves of the persistent traversal kernels of the other batches in flight.
// All threads of the workgroup must call tile_append together.
constexpr int kStageBlock = 256;
constexpr int kStagePerThread = 4;
constexpr int kStageTile = kStageBlock * kStagePerThread;
constexpr int kStageWaves = kStageBlock / 64;
BDPT_DEV void tile_append(const bool (&flag)[kStagePerThread], const uint32_t (&value)[kStagePerThread], uint32_t* queue,
                          uint32_t* count, uint32_t* s_wave, uint32_t* s_base) {
    unsigned long long m[kStagePerThread];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++) { m[j] = __ballot(flag[j]); mine += (uint32_t)__builtin_popcountll(m[j]); }
    const uint32_t wave = threadIdx.x >> 6;
    if (lane_id() == 0) s_wave[wave] = mine;        // `mine` is wave-uniform
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < (uint32_t)kStageWaves; w++) { const uint32_t t = s_wave[w]; s_wave[w] = run; run += t; }
        *s_base = run ? atomicAdd(count, run) : 0u;
    }
    __syncthreads();
    uint32_t at = *s_base + s_wave[wave];
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++) {
        if (flag[j]) queue[at + rank_in_mask(m[j])] = value[j];
        at += (uint32_t)__builtin_popcountll(m[j]);
    }
    __syncthreads();
}

// The same, and every flagged entry learns where it went (`at`; unchanged for the others).
BDPT_DEV void tile_append_at(const bool (&flag)[kStagePerThread], const uint32_t (&value)[kStagePerThread], uint32_t* queue,
                             uint32_t* count, uint32_t* s_wave, uint32_t* s_base, uint32_t (&at)[kStagePerThread]) {
    unsigned long long m[kStagePerThread];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++) { m[j] = __ballot(flag[j]); mine += (uint32_t)__builtin_popcountll(m[j]); }
    const uint32_t wave = threadIdx.x >> 6;
    if (lane_id() == 0) s_wave[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < (uint32_t)kStageWaves; w++) { const uint32_t t = s_wave[w]; s_wave[w] = run; run += t; }
        *s_base = run ? atomicAdd(count, run) : 0u;
    }
    __syncthreads();
    uint32_t next = *s_base + s_wave[wave];
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++) {
        if (flag[j]) { at[j] = next + rank_in_mask(m[j]); queue[at[j]] = value[j]; }
        next += (uint32_t)__builtin_popcountll(m[j]);
    }
    __syncthreads();
}

// The same, with the flagged entries of the tile ordered by a small key (stable only between keys): entries with equal keys
// end up next to one another in the queue.  The order of a queue does not touch any result - samples do not interact -
// only which rays share a wave of the traversal kernel.
constexpr uint32_t kSortBins = 64;
BDPT_DEV void tile_append_sorted(const bool (&flag)[kStagePerThread], const uint32_t (&value)[kStagePerThread], const uint32_t (&key)[kStagePerThread],
                                 uint32_t* queue, uint32_t* count, uint32_t* s_hist, uint32_t* s_base, uint32_t (&at)[kStagePerThread]) {
    for (uint32_t i = threadIdx.x; i < kSortBins; i += kStageBlock) s_hist[i] = 0u;
    __syncthreads();
    uint32_t rank[kStagePerThread];
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++) rank[j] = flag[j] ? atomicAdd(&s_hist[key[j]], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < 64u) {              // first wave: exclusive running sums of the bins (kSortBins == 64: one bin per lane)
        const uint32_t mine = s_hist[threadIdx.x];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += v;
        }
        s_hist[threadIdx.x] = incl - mine;
        if (threadIdx.x == 63u) *s_base = incl ? atomicAdd(count, incl) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kStagePerThread; j++)
        if (flag[j]) { at[j] = *s_base + s_hist[key[j]] + rank[j]; queue[at[j]] = value[j]; }
    __syncthreads();
}

BDPT_DEV void flush_wave_counter(unsigned long long* dst, unsigned long long x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane_id() == 0 && x) atomicAdd(dst, x);
}

// pixel (i, j) and stratum (s_i, s_j) of a slot; false for slots past the end of the batch / image
BDPT_DEV bool slot_to_sample(const GpuCamera& cam, const WfBatch& b, uint32_t slot, int& i, int& j, int& s) {
    // 32-bit arithmetic throughout (the host guarantees fewer than 2^31 owned pixels): 64-bit integer
    // division is a long software sequence on the GPU
    uint32_t sv, p;
    if (b.sample_stride > 1) { p = slot / (uint32_t)b.sample_stride; sv = slot - p * (uint32_t)b.sample_stride; }
    else { sv = slot / (uint32_t)b.pixels_stride; p = slot - sv * (uint32_t)b.pixels_stride; }
    s = (int)sv;
    const uint32_t q = (uint32_t)b.first_pixel + p;
    if (p >= (uint32_t)b.n_pixels || q >= (uint32_t)b.n_owned) return false;
    const uint32_t row = q / (uint32_t)b.row_w;
    j = b.rows[row];
    i = b.x0 + (int)(q - row * (uint32_t)b.row_w);
    return true;
}

BDPT_DEV Rng slot_rng(const GpuCamera& cam, const WfBatch& b, uint32_t slot, uint32_t counter) {
    int i, j, s;
    slot_to_sample(cam, b, slot, i, j, s);
    Rng rng;
    rng.key = stream_key(b.seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s);
    rng.counter = counter;
    return rng;
}

// The vertex records of a sample slot.  Slots are kept in groups of kVtxGroup: inside a group the layout is
// [record][field][slot in group], so the lanes of a wave that read one field of neighbouring slots read neighbouring
// words, and everything a slot owns lies within one group (tens of KB).  With the fields of ALL slots in one array each
// - field rows ns words apart - a kernel that reads several records of a slot touches a different page for every
// field (measured on an earlier connection kernel: 3.6 address-translation misses per memory instruction).
BDPT_DEV VertexStore slot_store(const WfBuffers& wb, uint32_t slot) {
    VertexStore st;
    st.ws = wb.vtx + (uint64_t)(slot / kVtxGroup) * ((uint64_t)kVtxGroup * wb.vtx_words);
    st.stride = kVtxGroup;
    st.g = slot % kVtxGroup;
    return st;
}
// material and flags of record `rec` (Vertex::info)
template <bool TEX = false>
BDPT_DEV uint32_t slot_vertex_info(const WfBuffers& wb, uint32_t slot, int rec) {
    const VertexStore st = slot_store(wb, slot);
    return (uint32_t)__double_as_longlong(st.ws[((uint64_t)rec * (TEX ? kVtxDoublesTextured : kVtxDoubles) + 9) * st.stride + st.g]);
}
// ------------------------------------------------------------------ camera rays
__global__ __launch_bounds__(kBlock) void k_gen_eye(GpuCamera cam, WfBuffers wb, WfBatch b) {
    const uint32_t spp = (uint32_t)(cam.sqrt_spp * cam.sqrt_spp);
    const uint64_t n = (uint64_t)b.pixels_stride * spp;
    // (the first queue is the slots themselves: entry k = slot k)
    for (uint64_t slot = (uint64_t)blockIdx.x * kBlock + threadIdx.x; slot < n; slot += (uint64_t)gridDim.x * kBlock) {
        int i, j, s;
        wb.counts[slot] = 0;
        wb.result[slot] = 0.0; wb.result[wb.ns + slot] = 0.0; wb.result[2 * wb.ns + slot] = 0.0;
        if (!slot_to_sample(cam, b, (uint32_t)slot, i, j, s)) {
            wb.ctr[slot] = 0;
            // a slot without a sample: a direction of zero, which the traversal does not walk
            wb.ray[slot] = 0; wb.ray[wb.ns + slot] = 0; wb.ray[2 * wb.ns + slot] = 0;
            wb.ray[3 * wb.ns + slot] = 0; wb.ray[4 * wb.ns + slot] = 0; wb.ray[5 * wb.ns + slot] = 0;
            continue;
        }
        Rng rng;
        rng.key = stream_key(b.seed, (uint64_t)j * (uint64_t)cam.width + (uint64_t)i, (uint64_t)s);
        rng.counter = 0;
        const int s_j = s / cam.sqrt_spp, s_i = s - s_j * cam.sqrt_spp;
        V3 o, d;
        camera_ray(cam, i, j, s_i, s_j, rng, o, d);
        wb.ray[slot] = o.x; wb.ray[wb.ns + slot] = o.y; wb.ray[2 * wb.ns + slot] = o.z;
        wb.ray[3 * wb.ns + slot] = d.x; wb.ray[4 * wb.ns + slot] = d.y; wb.ray[5 * wb.ns + slot] = d.z;
        // (the throughput of a camera ray is 1, camera.h:303: the first shading launch knows and does not read it)
        wb.ctrq[slot] = rng.counter;
        wb.ctr[slot] = rng.counter;       // (what stands when no bounce follows: max_depth 0)
    }
}

// ------------------------------------------------------------------ persistent traversal
// Ordered (near child first) walk over the four-wide records with a small per-lane stack.
//
// Why a different hierarchy and visiting order still return the reference's answers:
//  * a triangle is accepted by the reference's own inclusive interval test (triangle.h:63) in the
//    reference's FP64 arithmetic; the boxes of this hierarchy contain everything below them with a margin,
//    so culling never loses a triangle that would be accepted;
//  * closest-hit rays: the reference keeps, among equal t, the triangle it visits last (bvh.h:55-56), i.e.
//    the one later in its left-to-right leaf order; the tie is broken by that position (tr.pad);
//  * visibility rays (camera.h:425-438): the reference reports "blocked" iff some triangle accepts AND it
//    gets to test that triangle, i.e. the box of every bvh_node above it passes aabb::hit with the
//    unchanged interval [0.001, max_t] (the bound only shrinks after the first acceptance).  That second
//    condition matters: the far end point lies ON a surface at t = max_t, and a node box whose face
//    coincides with that surface is entered at t = max_t too - rounding decides, and the reference never
//    tests the triangle when the box loses.  So a triangle accepted at an end of the interval counts only if
//    the reference's own boxes, from its node up to the root, let the ray through in the reference's arithmetic
//    (reference_reaches(), or far_end_tie() where the answer follows from the host's face bits).
//
// Exact ties.  The statement above - "a different hierarchy returns the reference's answer" - rests on the
// reference reaching every triangle that would accept, i.e. on the boxes of its own tree passing aabb::hit for that
// ray.  They do whenever the hit point lies inside them with a margin, and they may not when the ray only grazes
// them: a hit on the outline of its triangle (a ray through a vertex touches the triangle's box in one point), a ray
// almost inside the triangle's plane (u, v and t then carry a large rounding error), a hit at the lower end 0.001 of
// the interval, or two hits at the same distance (coplanar surfaces: the second one's boxes are tested against an
// interval that ends exactly where they begin, and rounding decides whether the reference ever tests it).  Such
// hits are recognised by cheap sufficient conditions (grazes(), and nearly_equal() distances of a hit and the
// running closest distance).  A visibility ray then asks reference_reaches(); a closest-hit ray finishes its walk,
// is marked kHitRedo and put on a list, and k_redo - launched after every closest-hit traversal - repeats the query
// with the reference's own traversal (device_core.h: walk) and writes the reference's answer in its place.  Between
// 7 rays in 10^6 (curved geometry) and 2 in 10^3 (boxes resting on boxes) take that road.
// Why the remaining hits are safe: the hit point p is inside the triangle by at least bary_margin in every
// barycentric coordinate, hence inside the triangle's box - and every box above it - by bary_margin x the box's
// extent on every axis where that extent is at least 1e-4; on an axis where the box is thinner the reference pads
// its node boxes by 5e-5, or p lies ON a face of an unpadded larger box - on one axis only (triangles thin on two
// axes are always re-checked), so the ray still runs inside that box on one side of p.  The slab arithmetic of
// aabb.h:41-66 is accurate to a few ulp of the coordinates, far below those margins, and the interval it is
// clipped to, [0.001, closest so far], contains t with the same margins unless one of the conditions above holds.
// The conditions are evaluated on the HIGH 32 bits of the numbers (sign, exponent, 20 mantissa bits): for
// non-negative doubles the bit patterns order like the values, so "x < c" implies hi(x) <= hi(c), and every test
// below is an integer compare that holds whenever the exact condition holds (a superset; a few more rays than
// necessary take the slow road).  Distances within 2^-19 of each other count as equal.
BDPT_DEV uint32_t hi32(double x) { return (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32); }
struct GrazeBits {
    uint32_t bary_low;      // hi(bary_margin) + 1:       u or v below the margin      <=>  hi(u) < bary_low
    uint32_t bary_high;     // hi(1 - bary_margin):       u + v above 1 - margin       <=>  hi(u + v) >= bary_high
    uint32_t t_low;         // hi(0.001 + 1e-9) + 1:      t at the lower end            <=>  hi(t) < t_low
    uint32_t len_offset;    // hi(ray length bound) - hi(1.0) + one binade: added to the triangle's threshold bits.  The
                            // sum of two high words underestimates the high word of the product by less than a
                            // factor 4/3, which the extra binade (a factor 2) covers.
};
BDPT_DEV GrazeBits graze_bits(double bary_margin, float ray_len) {
    GrazeBits g;
    g.bary_low = hi32(bary_margin) + 1u;
    g.bary_high = hi32(1.0 - bary_margin);
    g.t_low = hi32(0.001 + 1e-9) + 1u;
    const uint32_t len_hi = hi32((double)ray_len);
    // (a NaN or infinite bound: every hit is re-checked)
    g.len_offset = (len_hi >= 0x7FF00000u) ? 0x7FF00000u : (len_hi > 0x3FF00000u ? len_hi - 0x3FF00000u : 0u) + 0x00100000u;
    return g;
}
// the triangle's grazing threshold - the high word of the double 1e-4 |e1 x e2|, or of +infinity for a triangle that
// is always re-checked - travels in the `material` field of the traversal hierarchy's triangle table
BDPT_DEV bool grazes(const GpuTri& tr, double u, double v, double t, double det, const GrazeBits& g) {
    const uint32_t limit = (uint32_t)tr.material + g.len_offset;
    // (bitwise, not short-circuit: five compares and four ORs without a branch)
    return ((unsigned)((hi32(u) & 0x7FFFFFFFu) < g.bary_low) | (unsigned)((hi32(v) & 0x7FFFFFFFu) < g.bary_low) |
            (unsigned)(hi32(u + v) >= g.bary_high) | (unsigned)(hi32(t) < g.t_low) | (unsigned)((hi32(det) & 0x7FFFFFFFu) < limit)) != 0u;
}
// |hi(a) - hi(b)| <= 1 for positive a, b (+infinity included)
BDPT_DEV bool nearly_equal(double a, double b) { return (hi32(a) - hi32(b) + 1u) <= 2u; }
// The low six bits of that field name the faces of the reference's boxes above the triangle that lie in the triangle's
// own plane (scene_build.cpp).  A visibility ray that ends ON the triangle (t = max_t up to rounding) is let through by
// all those boxes unless it enters one of them through such a face: bit 2a when it travels upwards on axis a, bit
// 2a + 1 when downwards.  With a component of exactly zero the reference's arithmetic works with infinities and
// NaNs: its boxes are asked.
// Six more bits (6..11) mark the sides on which all those faces lie exactly in the plane of an exactly flat triangle.
// Such a tie is settled here: every other plane of every box above the triangle is passed with a margin, so aabb::hit
// (aabb.h:45-62) comes down, for each of them alike, to whether the distance to that plane, formed as the reference
// forms it, leaves anything of the interval.
// Returns 0: the reference gets to test the triangle; 1: it does not; 2: its boxes have to be asked.
template <bool EXACT>
BDPT_DEV int far_end_tie(const GpuTri& tr, V3 o, V3 d, double max_t) {
    const uint32_t sx = hi32(d.x), sy = hi32(d.y), sz = hi32(d.z);
    const uint32_t entered = (1u + (sx >> 31)) | ((1u + (sy >> 31)) << 2) | ((1u + (sz >> 31)) << 4);
    const uint32_t e = 0x7FF00000u;
    const bool some_zero = ((sx & e) == 0u) | ((sy & e) == 0u) | ((sz & e) == 0u);
    const uint32_t faces = (uint32_t)tr.material & entered;
    if (some_zero) return 2;
    if (faces == 0u) return 0;
    if (!EXACT) return 2;
    const uint32_t exact = ((uint32_t)tr.material >> 6) & entered;
    if (exact != faces || (faces & (faces - 1u)) != 0u) return 2;          // (one face, and that one exact)
    const int axis = (faces & 3u) ? 0 : ((faces & 12u) ? 1 : 2);
    const double plane = axis == 0 ? tr.v0[0] : (axis == 1 ? tr.v0[1] : tr.v0[2]);      // (a flat triangle: all its points share it)
    const double oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
    const double da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
    const double adinv = 1.0 / da;                                          // aabb.h:47
    const double t_face = (plane - oa) * adinv;                             // aabb.h:49-50
    return (max_t <= t_face) ? 1 : 0;                                       // aabb.h:60-61
}

#ifdef BDPT_TIE_CHECK
// Cross-check build (make tiecheck): every accepted hit of a visibility ray that is not a grazing one is ALSO put to the
// reference's own boxes (aabb.h:41-66, from the triangle's bvh_node up to the root), and what the shortcuts above would
// have answered is compared with it.  The answer used is the reference's, so images do not change.
//  [0] far-end ties seen            [1] settled by the face bits alone      [2] settled by the exact-plane distance
//  [3] left to the boxes            [4] wrong among [1]                     [5] wrong among [2]
//  [6] accepted hits away from both ends of the interval and from the outline (taken as reached without a test)
//  [7] of those, not reached by the reference
__device__ unsigned long long g_tie_counts[8];
BDPT_DEV void tie_check_record(bool tie, const GpuTri& tr, V3 o, V3 d, double max_t, bool reached) {
    if (!tie) {
        atomicAdd(&g_tie_counts[6], 1ull);
        if (!reached) atomicAdd(&g_tie_counts[7], 1ull);
        return;
    }
    atomicAdd(&g_tie_counts[0], 1ull);
    const int by_bits = far_end_tie<false>(tr, o, d, max_t), by_plane = far_end_tie<true>(tr, o, d, max_t);
    if (by_bits == 0) {
        atomicAdd(&g_tie_counts[1], 1ull);
        if (!reached) atomicAdd(&g_tie_counts[4], 1ull);
    } else if (by_plane != 2) {
        atomicAdd(&g_tie_counts[2], 1ull);
        if ((by_plane == 0) != reached) atomicAdd(&g_tie_counts[5], 1ull);
    } else {
        atomicAdd(&g_tie_counts[3], 1ull);
    }
}
#endif

constexpr int kQuadStackDepth = kQuadStackEntries;
constexpr int32_t kPop = (int32_t)0x80000000;   // neither an interior index (>= 0) nor a leaf reference

template <class NodeT>
struct TreeMem {
    const NodeT* wn;          // global or LDS copy
    const GpuTri* st;
    const int32_t* entry_obj; // scenes with instances: SceneDev::xobj and SceneDev::objs (or their copies)
    const CxObject* objs;
    BDPT_DEV NodeT wnode(int32_t i) const { return wn[i]; }
    BDPT_DEV GpuTri stri(uint32_t i) const { return st[i]; }
    BDPT_DEV int32_t xobj(uint32_t i) const { return entry_obj[i]; }
    BDPT_DEV CxObject object(int32_t k) const { return objs[k]; }
};

// Large scenes: the records nearest the root come from LDS, the deeper ones from global memory.
template <class NodeT>
struct TopLdsMem {
    const NodeT* lds;         // first n_lds records
    const NodeT* wn;
    const GpuTri* st;
    int32_t n_lds;
    const int32_t* entry_obj;
    const CxObject* objs;
    BDPT_DEV NodeT wnode(int32_t i) const {
        if (i < n_lds) return lds[i];
        return wn[i];
    }
    BDPT_DEV GpuTri stri(uint32_t i) const { return st[i]; }
    BDPT_DEV int32_t xobj(uint32_t i) const { return entry_obj[i]; }
    BDPT_DEV CxObject object(int32_t k) const { return objs[k]; }
};

// Box tests of the traversal hierarchy only have to be conservative, so they run in packed FP32 with
// the rounding error carried explicitly.  Per axis the ray keeps inv ~ 1/d (FP32 reciprocal instruction, 1 ulp) and
// the pair (c_min, c_max) = fl(-o * inv) -/+ slack; the distances to the two planes of a box are then ONE packed
// fma, (box_min, box_max) * inv + (c_min, c_max) for 1/d > 0 (pair swapped otherwise), giving a lower bound of
// the distance to the near plane and an upper bound of the distance to the far plane.  Against the exact
// (b - o)/d the computed value is off by at most
//   |b - o|/|d| * 1.5 * 2^-23  (d to FP32 and the reciprocal)  +  |o|/|d| * 2^-23  (o to FP32 and the product)
//   + |t| * 2^-24  (the fma)   <=   (|b| + |o|) / |d| * 0.75 * 2^-21
// and slack = (bound + |o|) * 2^-20 * |inv| (bound = largest |coordinate| of any box) is 2.6 times that.
// A direction component of zero would give infinities and NaNs here; a component below 2^-60 of the largest one in
// magnitude is taken as +-2^-60 of it: over the whole length a ray can have inside the scene it then moves along that axis
// by less than 2^-60 of the scene's extent, far inside the slack, like the true ray (the planes of a slab that holds o lie
// at -huge and +huge, those of any other slab both at a distance no interval reaches; a box within the slack of o still
// passes).  A ray whose LARGEST component is below 1e-18 (or not a number) is not culled at all - inv = 0 and
// (c_min, c_max) = (-inf, +inf) put no bound on the interval - and so is any axis on which 1/d, o/d or the slack leave
// the FP32 range all the same.
// (The reference's own test, aabb.h:41-66, misses the box when the origin is outside that slab and passes
// otherwise; both treatments are supersets of it.)
typedef float f2 __attribute__((ext_vector_type(2)));
struct RayF32 {
    f2 inv[3];      // (1/d, 1/d)
    f2 c[3];        // additive terms of the (min, max) planes
};
BDPT_DEV RayF32 ray_f32(V3 o, V3 d, float bound) {
    RayF32 r;
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    const float largest = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf((float)d.x), __builtin_fabsf((float)d.y)), __builtin_fabsf((float)d.z));
    const float least = largest * 0x1p-60f;
    const bool sane = largest >= 1e-18f && (largest - largest == 0.0f);       // (false for a NaN or infinite component as well)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // (leaving the axis out instead sends a ray that runs inside an axis-parallel plane - two points of one floor
        // with exactly equal heights - through every box above and below it: 20 x the time for such query segments)
        const float d32 = (float)dd[a];
        const float inv = __builtin_amdgcn_rcpf(__builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(d32), least), d32));
        const float of = (float)oo[a];
        const float c = -(of * inv);
        const float slack = (bound + __builtin_fabsf(of)) * 0x1p-20f * __builtin_fabsf(inv) * 1.000001f;
        // (x - x == 0) is false exactly for infinities and NaNs
        const bool usable = sane && (inv - inv == 0.0f) && (c - c == 0.0f) && (slack - slack == 0.0f);
        r.inv[a] = usable ? (f2){inv, inv} : (f2){0.0f, 0.0f};
        // 1/d > 0: the min plane is the near one (gets -slack); 1/d < 0: the max plane is
        r.c[a] = !usable ? (f2){-__builtin_inff(), __builtin_inff()}
                         : ((inv >= 0.0f) ? (f2){c - slack, c + slack} : (f2){c + slack, c - slack});
    }
    return r;
}
// near / far distance bounds of one child box; the ray enters it iff !(far < near)
BDPT_DEV void slab_f32(const float* box, const RayF32& r, float closest_up, float& near, float& far) {
    const f2 tx = __builtin_elementwise_fma((f2){box[0], box[1]}, r.inv[0], r.c[0]);
    const f2 ty = __builtin_elementwise_fma((f2){box[2], box[3]}, r.inv[1], r.c[1]);
    const f2 tz = __builtin_elementwise_fma((f2){box[4], box[5]}, r.inv[2], r.c[2]);
    near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)),
                           __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), 0x1.0624dcp-10f));      // largest FP32 number below 0.001
    far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)),
                          __builtin_fminf(__builtin_fmaxf(tz.x, tz.y), closest_up));
}
// smallest FP32 number >= x (x > 0 or +infinity)
BDPT_DEV float float_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);
    return f;
}

template <class T>
BDPT_DEV void slab(const T* blo, const T* bhi, V3 o, double ix, double iy, double iz, double closest, double& lo,
                   double& hi) {
    lo = 0.001; hi = closest;
    {
        double t0 = ((double)blo[0] - o.x) * ix, t1 = ((double)bhi[0] - o.x) * ix;
        double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
        lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
    }
    {
        double t0 = ((double)blo[1] - o.y) * iy, t1 = ((double)bhi[1] - o.y) * iy;
        double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
        lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
    }
    {
        double t0 = ((double)blo[2] - o.z) * iz, t1 = ((double)bhi[2] - o.z) * iz;
        double nr = (t0 < t1) ? t0 : t1, fr = (t0 < t1) ? t1 : t0;
        lo = (nr > lo) ? nr : lo; hi = (fr < hi) ? fr : hi;
    }
}

// A child the ray enters, as one 64-bit word that orders like its entry distance: high half = the bits
// of the (positive FP32) lower bound of the distance, low half = the child reference.  Read as an FP64
// number the word is positive and finite, and larger distance <=> larger number, so a compare-exchange
// is one min and one max instruction.  Children the ray misses get +infinity and sort last.
BDPT_DEV double quad_entry(bool in, float near, int32_t child) {
    const uint32_t bits = in ? __float_as_uint(near) : 0x7F800000u;
    return __longlong_as_double((long long)(((unsigned long long)bits << 32) | (uint32_t)child));
}
BDPT_DEV int32_t quad_entry_child(double e) { return (int32_t)(uint32_t)(unsigned long long)__double_as_longlong(e); }
BDPT_DEV float quad_entry_near(double e) { return __uint_as_float((uint32_t)((unsigned long long)__double_as_longlong(e) >> 32)); }
BDPT_DEV void sort2(double& a, double& b) {
    // the instructions themselves: fmin()/fmax() would first quiet possible signalling NaNs of both operands
    // (one more instruction each), and these words are never NaNs
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo; b = hi;
}

// aabb::hit (aabb.h:41-66) of every bvh_node from `node` up to the root, with the interval [0.001, max_t].
// `nodes` / `parents` are the reference tree's tables in global memory or their copy in LDS (small scenes).
//
// One box decides for almost every ray.  A box of the tree contains the boxes below it coordinate by coordinate
// (`nested`, verified when the scene is built).  Per axis aabb::hit computes t = (bound - origin) * inverse: a
// subtraction of a fixed number and a multiplication by a fixed number, each correctly rounded and therefore
// monotone in `bound`; the near and far values are the min and max of the two, the interval is the running max of the
// nears and min of the fars - monotone again.  So the interval of a box contains the interval of every box below it
// exactly as the reference computes them, and a ray that passes the lowest box passes all boxes above it.  The
// argument needs numbers that compare (no NaN): an axis-parallel ray has an infinite inverse and 0 * inf = NaN when the
// origin lies in a face's plane (aabb.h:52-55 then behaves differently), so such rays walk all the way up.
constexpr uint32_t kNestedBit = 0x80000000u;      // in an entry of SceneDev::sleaf: the boxes of the tree are nested
// (a direction component that is a normal number - exponent field not zero - is at least 2^-1022 in magnitude, and its
// inverse at most 2^1022)
BDPT_DEV bool inverse_is_finite(V3 d) {
    const uint32_t e = 0x7FF00000u;
    return ((uint32_t)((hi32(d.x) & e) != 0u) & (uint32_t)((hi32(d.y) & e) != 0u) & (uint32_t)((hi32(d.z) & e) != 0u)) != 0u;
}
BDPT_DEV bool reference_reaches_inline(const GpuNode* nodes, const uint32_t* parents, bool nested, uint32_t node, V3 o, V3 d, double max_t) {
    const bool first_decides = nested && inverse_is_finite(d);
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;       // aabb.h:48
    for (uint32_t n = node; n != kEndOfTree; n = first_decides ? kEndOfTree : parents[n]) {
        const GpuNode nd = nodes[n];
        double lo, hi;
        slab(nd.lo, nd.hi, o, ix, iy, iz, max_t, lo, hi);
        if (hi <= lo) return false;
    }
    return true;
}
// out of line for the traversal kernels, whose register budget decides how many waves of other kernels fit beside them
// `node_and_nested`: an entry of SceneDev::sleaf
__device__ __attribute__((noinline)) bool reference_reaches(const GpuNode* nodes, const uint32_t* parents, uint32_t node_and_nested, V3 o, V3 d, double max_t) {
    return reference_reaches_inline(nodes, parents, (node_and_nested & kNestedBit) != 0u, node_and_nested & ~kNestedBit, o, d, max_t);
}
// (the same over the scene's tables in global memory: the variant the traversal kernel of a large scene calls, with
// one pointer to pass instead of two)
// (reads two fields of the scene, which the compiler passes as two pointers; with a third it would pass a copy of the
// whole structure on the stack - hence the flag inside the node number)
__device__ __attribute__((noinline)) bool reference_reaches(const SceneDev& sc, uint32_t node_and_nested, V3 o, V3 d, double max_t) {
    const bool first_decides = (node_and_nested & kNestedBit) != 0u && inverse_is_finite(d);
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
    for (uint32_t n = node_and_nested & ~kNestedBit; n != kEndOfTree; n = first_decides ? kEndOfTree : sc.node_parent[n]) {
        const GpuNode nd = sc.nodes[n];
        double lo, hi;
        slab(nd.lo, nd.hi, o, ix, iy, iz, max_t, lo, hi);
        if (hi <= lo) return false;
    }
    return true;
}

// The reference's tree for that check: the global tables, or - when the whole scene is staged in LDS - a copy
// behind the staged traversal hierarchy.
struct RefTree {
    const GpuNode* nodes;
    const uint32_t* parents;
    BDPT_DEV bool reaches(const SceneDev&, uint32_t node_and_nested, V3 o, V3 d, double max_t) const { return reference_reaches(nodes, parents, node_and_nested, o, d, max_t); }
};
// (no state of its own: a traversal kernel that reads the global tables keeps no extra registers for them)
struct GlobalRefTree {
    BDPT_DEV bool reaches(const SceneDev& sc, uint32_t node_and_nested, V3 o, V3 d, double max_t) const { return reference_reaches(sc, node_and_nested, o, d, max_t); }
};
// Scenes with instances: the chain of boxes above an entry of `stris` runs through the mesh's own tree with the ray as
// the mesh sees it (translate::hit / rotate_y::hit hand the interval on unchanged, hittable.h:54-66,110-148) and then
// through the world's tree, from the node that holds the object, with the ray itself.
BDPT_DEV bool instanced_reaches(const SceneDev& sc, uint32_t pos, V3 o, V3 d, double max_t) {
    const uint32_t leaf = sc.sleaf[pos];
    const int32_t ob = sc.pos_obj[pos];
    if (ob < 0) return reference_reaches(sc, leaf, o, d, max_t);
    V3 oo = o, od = d;
    cx_to_object(sc.objs[ob], oo, od);
    if (!reference_reaches(sc, leaf, oo, od, max_t)) return false;
    return reference_reaches(sc, sc.obj_leaf[ob], o, d, max_t);
}
struct InstancedRefTree {
    BDPT_DEV bool reaches(const SceneDev& sc, uint32_t pos, V3 o, V3 d, double max_t) const { return instanced_reaches(sc, pos, o, d, max_t); }
};
BDPT_DEV size_t ref_tree_lds_bytes(uint32_t n_nodes) { return (size_t)n_nodes * (sizeof(GpuNode) + sizeof(uint32_t)); }
// all threads of the workgroup; `dst` 16-byte aligned
BDPT_DEV RefTree stage_ref_tree(const SceneDev& sc, unsigned char* dst) {
    GpuNode* l_nodes = reinterpret_cast<GpuNode*>(dst);
    uint32_t* l_parents = reinterpret_cast<uint32_t*>(l_nodes + sc.n_nodes);
    const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
    uint4* d16 = reinterpret_cast<uint4*>(l_nodes);
    for (uint32_t k = threadIdx.x; k < sc.n_nodes * (uint32_t)(sizeof(GpuNode) / 16); k += blockDim.x) d16[k] = src[k];
    for (uint32_t k = threadIdx.x; k < sc.n_nodes; k += blockDim.x) l_parents[k] = sc.node_parent[k];
    __syncthreads();
    return RefTree{l_nodes, l_parents};
}

// One loop iteration is EITHER one interior record for every lane that stands at one, OR one triangle
// for every lane that stands at a leaf: a wave executes both sides of a divergent branch one after the
// other, so mixing the two in one iteration would run the (long) triangle test for a handful of lanes
// every time.  Lanes that reach a leaf wait until `leaf_threshold` lanes of the wave have one (or no lane
// has an interior record left).  SHADOW: rays come from the shadow queue, the walk stops at the first
// accepted triangle and a clear ray sets its queue entry's flag.
// INSTANCED: an entry of the triangle table may belong to an object; its test then takes the ray as the object's mesh
// sees it (the reference's translate::hit / rotate_y::hit in front of triangle::hit), everything else is unchanged -
// the ray parameter t is the same number in both coordinate systems.
template <bool SHADOW, bool STATS, bool INSTANCED, class Mem, class Ref>
BDPT_DEV void traverse_queue(const Mem& mem, const Ref& ref, const SceneDev& sc, const WfBuffers& wb, uint32_t count,
                             uint32_t* cursor, unsigned long long* stats, const int kRefillThreshold, const uint32_t kChunk,
                             const int kLeafThreshold, const int kLeafInner, const double coord_bound, double* lds_stack, const int kLdsStack,
                             const uint32_t first_base, uint32_t* redo_list, uint32_t* redo_count) {
    const float bound_f32 = float_up(coord_bound);
    // upper bound of |d| over the rays of this launch (visibility rays have unit directions)
    const GrazeBits gbits = graze_bits(sc.bary_margin, SHADOW ? 1.0001f : (wb.ray_len_ptr ? *wb.ray_len_ptr : wb.ray_len));
    bool redo = false;                  // closest-hit rays: the walk met an exact tie, see above
    bool has = false;
    uint32_t slot = 0;
    int32_t cur = kPop;                 // interior record to visit next (when leaf_left == 0)
    uint32_t leaf_next = 0, leaf_left = 0;   // triangles of the current leaf still to test
    int sp = 0;
    // One 64-bit word per entry, see quad_entry().  The first kLdsStack entries (the ones pushed and popped most)
    // live in LDS, column threadIdx.x of a [kLdsStack][kBlock] array: a pop is then an LDS read instead of a
    // scratch read that competes with the node records for the vector L1.  Deeper entries go to scratch.
    double stk_quad[kQuadStackDepth];
    double* const lstk = lds_stack + threadIdx.x;
    auto push = [&](double e) {
        if (sp < kLdsStack) lstk[sp * kBlock] = e;
        else stk_quad[sp - kLdsStack] = e;
        sp++;
    };
    // visibility rays: the queue entry of the ray (a clear ray sets the entry's flag)
    uint32_t entry = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    double closest = 0;
    RayF32 rf = {};
    HitLite h;
    h.t = 0; h.u = 0; h.v = 0; h.tri = -1;
    // first_base: the wave's first chunk, taken from the cursor by its workgroup for all its waves at once (all
    // waves start at the same moment: one atomic per wave on one address would queue up)
    uint32_t pool_next = 0, pool_end = 0;     // wave-uniform
    if (first_base < count) { pool_next = first_base; pool_end = (first_base + kChunk < count) ? first_base + kChunk : count; }
    bool exhausted = (pool_next == pool_end);
    // INSTANCED: the ray as the object of the last tested triangle sees it (the triangles of a leaf nearly always
    // belong to one object, and a wave stays with a leaf's triangles for several iterations)
    int32_t cached_ob = -1;
    V3 cached_o = mk(0, 0, 0), cached_d = mk(0, 0, 0);
    uint32_t n_records = 0, n_box_hits = 0, n_tri = 0, n_tri_hits = 0, n_rays = 0;
#ifdef BDPT_AB_PHASE_STATS
    // measuring build: lane slots and active lanes of the two kinds of iteration, reported in place of the box / triangle counters
    uint32_t ph_node_slots = 0, ph_node_lanes = 0, ph_leaf_slots = 0, ph_leaf_lanes = 0, ph_iters = 0;
#endif
#ifdef BDPT_AB_COUNT_REDO
    uint32_t n_redo = 0;
#endif

    for (;;) {
#ifdef BDPT_AB_PHASE_STATS
        if (STATS) ph_iters++;
#endif
        const unsigned long long idle = __ballot(!has);
        const int n_idle = __builtin_popcountll(idle);
        if (!exhausted && (n_idle >= kRefillThreshold)) {
            if (pool_next == pool_end) {
                uint32_t base = count;
                if (cursor) {
                    if (lane_id() == 0) base = atomicAdd(cursor, kChunk);
                    base = __shfl(base, 0, 64);
                }
                if (base >= count) { exhausted = true; pool_next = pool_end = 0; }
                else { pool_next = base; pool_end = (base + kChunk < count) ? base + kChunk : count; }
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t give = avail < (uint32_t)n_idle ? avail : (uint32_t)n_idle;
            if (!has) {
                const uint32_t r = rank_in_mask(idle);
                if (r < give) {
                    const uint32_t k = pool_next + r;
                    double tmax0;
                    if (SHADOW) {
                        const uint64_t cap = wb.shadow_cap;
                        o = mk(wb.shadow[k], wb.shadow[cap + k], wb.shadow[2 * cap + k]);
                        d = mk(wb.shadow[3 * cap + k], wb.shadow[4 * cap + k], wb.shadow[5 * cap + k]);
                        tmax0 = wb.shadow[6 * cap + k];
                        entry = k;
                    } else {
                        slot = k;               // (the ray's place in the queue: its state lives there)
                        const uint64_t ns = wb.ns;
                        o = mk(wb.ray[k], wb.ray[ns + k], wb.ray[2 * ns + k]);
                        d = mk(wb.ray[3 * ns + k], wb.ray[4 * ns + k], wb.ray[5 * ns + k]);
                        tmax0 = wb.tmax ? wb.tmax[k] : kInf;
                        // a direction of zero: an entry without a path (kDeadSlot), or a query no triangle can accept (its
                        // determinant is zero for every triangle, triangle.h:49-50)
                        // (... nor does an interval that ends below zero or is not a number hold anything)
                        if ((d.x == 0.0 && d.y == 0.0 && d.z == 0.0) || !(tmax0 >= 0.0)) { wb.hit_tri[k] = -1; tmax0 = -1.0; }
                    }
                    if (tmax0 >= 0.0) {              // a negative max_t marks an unused queue entry
                        rf = ray_f32(o, d, bound_f32);
                        if (INSTANCED) cached_ob = -1;
                        closest = tmax0;
                        cur = 0;
                        leaf_left = 0;
                        sp = 0;
                        // The far end of a shadow ray lies ON a surface at t = max_t, and in most unoccluded pairs
                        // rounding makes that surface's own triangle accept (the reference then calls the pair
                        // blocked, camera.h:437) - after the walk has crossed everything between the two ends.
                        // "Some triangle accepts and is reachable" does not depend on the order of the tests, so the
                        // walk begins as if it stood at a leaf holding that one triangle, the root waiting on the stack.
                        if (SHADOW && wb.shadow_first) {
                            leaf_next = wb.shadow_first[k];
                            leaf_left = 1;
                            push(quad_entry(true, 0.0f, 0));
                        }
                        h.tri = -1;
                        redo = false;
                        has = true;
                        if (STATS) n_rays++;
                    }
                }
            }
            pool_next += give;
        }
        const unsigned long long active = __ballot(has);
        if (active == 0) {
            if (exhausted) break;
            continue;
        }
        const unsigned long long at_leaf = __ballot(has && leaf_left > 0);
        const bool leaf_phase = (at_leaf == active) || (__builtin_popcountll(at_leaf) >= kLeafThreshold);
        bool pop = false, done = false;
#ifdef BDPT_AB_PHASE_STATS
        if (STATS && !leaf_phase) { ph_node_slots++; if (has && leaf_left == 0) ph_node_lanes++; }
#endif
        if (!leaf_phase) {
            if (has && leaf_left == 0) {
                const auto nd = mem.wnode(cur);
                if (STATS) n_records++;
                {
                    // four children: sort the ones the ray enters by entry distance, go to the nearest, push the rest
                    double e[4];
                    int n_in = 0;
                    const float closest_up = float_up(closest);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float near, far;
                        slab_f32(nd.box[k], rf, closest_up, near, far);
                        const bool in = (nd.child[k] != kQuadEmpty) && !(far < near);
                        e[k] = quad_entry(in, near, nd.child[k]);
                        n_in += in ? 1 : 0;
                    }
                    if (STATS) n_box_hits += (uint32_t)n_in;
                    sort2(e[0], e[1]); sort2(e[2], e[3]); sort2(e[0], e[2]); sort2(e[1], e[3]); sort2(e[1], e[2]);
                    // sp + 3 <= kQuadStackDepth: the host checks the worst case of the tree
                    if (sp + 3 <= kLdsStack) {
                        // all in LDS: three stores without a branch, e[k] to row sp + n_in - 1 - k (the nearest
                        // on top); a child the ray misses goes to the spare row kLdsStack
                        lstk[(n_in > 3 ? sp : kLdsStack) * kBlock] = e[3];
                        lstk[(n_in > 2 ? sp + n_in - 3 : kLdsStack) * kBlock] = e[2];
                        lstk[(n_in > 1 ? sp + n_in - 2 : kLdsStack) * kBlock] = e[1];
                        sp += n_in > 1 ? n_in - 1 : 0;
                    } else {
                        if (n_in > 3) push(e[3]);
                        if (n_in > 2) push(e[2]);
                        if (n_in > 1) push(e[1]);
                    }
                    if (n_in > 0) cur = quad_entry_child(e[0]);
                    else pop = true;
                }
            }
        } else {
          for (;;) {
#ifdef BDPT_AB_PHASE_STATS
            if (STATS) { ph_leaf_slots++; if (has && leaf_left > 0 && !done) ph_leaf_lanes++; }
#endif
            if (has && leaf_left > 0 && !done) {
                const GpuTri tr = mem.stri(leaf_next);
                V3 to = o, td = d;
                if (INSTANCED) {
                    const int32_t ob = mem.xobj(leaf_next);
                    if (ob >= 0) {
                        if (ob != cached_ob) {
                            cached_o = o; cached_d = d;
                            cx_to_object(mem.object(ob), cached_o, cached_d);
                            cached_ob = ob;
                        }
                        to = cached_o; td = cached_d;
                    }
                }
                leaf_next++;
                leaf_left--;
                double t, u, v, det;
                // triangle.h:63: accepted when tmin <= t <= closest (inclusive)
                const bool cand = tri_candidate(tr, to, td, 0.001, t, u, v, det);
                const bool inside = cand && t <= closest;
                if (STATS) { n_tri++; if (inside) n_tri_hits++; }
                if (SHADOW) {
                    if (inside) {
                        // Away from the ends of the interval and from the triangle's outline the boxes above an
                        // accepted triangle contain the hit point with a margin, so they pass; a hit at the far end
                        // (the far end point's own surface, t = max_t up to rounding: most unoccluded pairs) or a
                        // grazing one counts only if the reference's own boxes let it through.
                        // (to, td: the ray as the triangle's own boxes see it)
                        int ask = grazes(tr, u, v, t, det, gbits) ? 2 : (nearly_equal(t, closest) ? far_end_tie<true>(tr, to, td, closest) : 0);
#ifdef BDPT_TIE_CHECK
                        if (!grazes(tr, u, v, t, det, gbits)) {
                            const bool reached = ref.reaches(sc, INSTANCED ? (uint32_t)tr.pad : sc.sleaf[tr.pad], o, d, closest);
                            tie_check_record(nearly_equal(t, closest), tr, to, td, closest, reached);
                            ask = reached ? 0 : 1;
                        }
#endif
                        if (ask == 2) ask = ref.reaches(sc, INSTANCED ? (uint32_t)tr.pad : sc.sleaf[tr.pad], o, d, closest) ? 0 : 1;
                        if (ask == 0) { h.tri = tr.pad; done = true; leaf_left = 0; }
                    }
                } else {
                    // (the walk goes on after an exact tie: rare, and the loop keeps its shape - no branch here)
                    const unsigned tied = (unsigned)cand & ((unsigned)nearly_equal(t, closest) | ((unsigned)inside & (unsigned)grazes(tr, u, v, t, det, gbits)));
                    redo = ((unsigned)redo | tied) != 0u;
                    // equal t: the reference keeps the triangle it visits later = larger position in
                    // its leaf order (tr.pad)
                    const bool better = ((unsigned)inside & ((unsigned)(t < closest) | (unsigned)(h.tri < 0) | (unsigned)(tr.pad > h.tri))) != 0u;
                    closest = better ? t : closest;           // the hit distance IS the running bound
                    h.tri = better ? tr.pad : h.tri;
                }
                if (leaf_left == 0 && !done) pop = true;
            }
            // go on with the next triangles of the same leaves while enough lanes have one
            if (__builtin_popcountll(__ballot(has && leaf_left > 0 && !done)) < kLeafInner) break;
          }
        }
        if (pop) {
            done = true;
            while (sp > 0) {
                sp--;
                {
                    const double e = (sp < kLdsStack) ? lstk[sp * kBlock] : stk_quad[sp - kLdsStack];
                    // strict: a triangle at exactly the running distance may still win the tie
                    if (!(closest < (double)quad_entry_near(e))) { cur = quad_entry_child(e); done = false; break; }
                }
            }
        }
        if (has && !done && leaf_left == 0 && cur < 0) {      // arrived at a leaf: ~cur = (first << 2) | (count - 1)
            const uint32_t code = (uint32_t)~cur;
            leaf_next = code >> 2;
            leaf_left = 1u + (code & 3u);
        }
        if (done) {
            if (SHADOW) {
                if (h.tri < 0) wb.sh_vis[entry] = 1;       // nothing in the way: the pair is visible
            } else {
#ifdef BDPT_AB_COUNT_REDO
                if (STATS && redo) n_redo++;
#endif
                wb.hit_tri[slot] = redo ? kHitRedo : h.tri;      // position in the reference's leaf order (k_shade turns it into the triangle id)
                // an exact tie: the ray goes on the list of k_redo, which traces it in the reference's own order; the
                // distance found here (if any) tells that walk where to look
                if (redo) { redo_list[atomicAdd(redo_count, 1u)] = slot; wb.hit[slot] = (h.tri >= 0) ? closest : kInf; }
                // the barycentric coordinates of the hit are not kept: nothing on this path reads them (the
                // reference's hit_record carries u, v for textures, and every texture here is a solid colour)
                else if (h.tri >= 0) wb.hit[slot] = closest;
            }
            has = false;
        }
    }
#ifdef BDPT_AB_PHASE_STATS
    if (STATS) {
        flush_wave_counter(&stats[SHADOW ? 6 : 0], n_rays);
        flush_wave_counter(&stats[1], ph_node_slots);
        flush_wave_counter(&stats[2], ph_node_lanes);
        flush_wave_counter(&stats[3], ph_iters);
        flush_wave_counter(&stats[4], ph_leaf_slots);
        flush_wave_counter(&stats[5], ph_leaf_lanes);
        return;
    }
#endif
    if (STATS) {
        flush_wave_counter(&stats[SHADOW ? 6 : 0], n_rays);
        flush_wave_counter(&stats[1], 4ull * n_records);
        flush_wave_counter(&stats[2], n_box_hits);
        flush_wave_counter(&stats[3], 2ull * n_records);   // in units of 64 bytes fetched (both record types are 128 bytes)
        flush_wave_counter(&stats[4], n_tri);
#ifdef BDPT_AB_COUNT_REDO
        flush_wave_counter(&stats[5], n_redo);
#else
        flush_wave_counter(&stats[5], n_tri_hits);
#endif
    }
}

// Rays a wave takes from the cursor at a time: the configured chunk while the queue is long, an even
// share (at least 16) when it is short, so that the last rays of a short queue are spread over all
// waves instead of waiting their turn inside a few.
BDPT_DEV uint32_t fair_chunk(uint32_t count, uint32_t chunk) {
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    uint32_t share = ((count + n_waves - 1u) / n_waves + 15u) & ~15u;      // n_waves * share >= count
    if (share < 16u) share = 16u;
    return share < chunk ? share : chunk;
}

template <class NodeT>
BDPT_DEV void stage_tree(const NodeT* nodes, uint32_t n_nodes, const GpuTri* tris, uint32_t n_tris, NodeT* l_nodes, GpuTri* l_tris) {
    const uint4* src = reinterpret_cast<const uint4*>(nodes);
    uint4* dst = reinterpret_cast<uint4*>(l_nodes);
    const uint32_t n16 = n_nodes * (uint32_t)(sizeof(NodeT) / 16);
    for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
    src = reinterpret_cast<const uint4*>(tris);
    dst = reinterpret_cast<uint4*>(l_tris);
    const uint32_t t16 = n_tris * (uint32_t)(sizeof(GpuTri) / 16);
    for (uint32_t k = threadIdx.x; k < t16; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}

template <bool SHADOW, bool STATS, bool LDS_SCENE, bool INSTANCED = false>
__global__ __launch_bounds__(kBlock) void k_traverse(SceneDev sc, WfBuffers wb,
                                                    const uint32_t* count_ptr, uint32_t count_imm, uint32_t* cursor,
                                                    unsigned long long* stats, int refill_threshold, uint32_t chunk,
                                                    uint32_t top_nodes, int leaf_threshold, int leaf_inner, uint32_t scene_lds_bytes,
                                                    int lds_stack, uint32_t* redo_list, uint32_t* redo_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t count = count_ptr ? *count_ptr : count_imm;
    double* const l_stack = reinterpret_cast<double*>(smem + scene_lds_bytes);     // [lds_stack + 1][kBlock]
    const uint32_t wave_chunk = fair_chunk(count, chunk);
    // The first chunk of every wave.  A short queue (at most 64 entries per wave) is simply dealt out, wave w owns
    // chunk w, and the launch touches no counter at all; of a longer one each workgroup takes the first chunks of
    // all its waves with ONE atomic (all waves start at the same moment: one atomic per wave on one address would
    // queue up), and the waves come back to the cursor for more.  Dealing out a long queue as well would tie
    // entries to workgroups that start late because kernels of the other batches hold their place.
    const uint32_t waves_per_block = blockDim.x >> 6;
    uint32_t first_base;
    if ((uint64_t)count <= (uint64_t)gridDim.x * waves_per_block * 64u &&
        (uint64_t)count <= (uint64_t)gridDim.x * waves_per_block * wave_chunk) {      // every entry has an owner
        if ((uint64_t)blockIdx.x * waves_per_block * wave_chunk >= count) return;       // the whole workgroup
        first_base = (blockIdx.x * waves_per_block + (threadIdx.x >> 6)) * wave_chunk;
        cursor = nullptr;
    } else {
        __shared__ uint32_t s_first;
        if (threadIdx.x == 0) s_first = atomicAdd(cursor, waves_per_block * wave_chunk);
        __syncthreads();
        if (s_first >= count) return;          // the whole workgroup: nothing left for it
        first_base = s_first + (threadIdx.x >> 6) * wave_chunk;
    }
    {
        if (LDS_SCENE) {
            GpuQuadNode* l_nodes = reinterpret_cast<GpuQuadNode*>(smem);
            GpuTri* l_tris = reinterpret_cast<GpuTri*>(l_nodes + sc.n_xnodes);
            // Entries of the hierarchy's triangle table: with instances there is one per (object, triangle) pair, and behind
            // the table stand the objects (their transforms) and the object of every entry - the loads a triangle test waits for.
            const uint32_t n_entries = !INSTANCED ? sc.n_tris
                : (scene_lds_bytes - sc.n_xnodes * (uint32_t)sizeof(GpuQuadNode) - sc.n_objs * (uint32_t)sizeof(CxObject)) / (uint32_t)(sizeof(GpuTri) + sizeof(int32_t));
            CxObject* l_objs = reinterpret_cast<CxObject*>(l_tris + n_entries);
            int32_t* l_xobj = reinterpret_cast<int32_t*>(l_objs + (INSTANCED ? sc.n_objs : 0u));
            if (INSTANCED) {
                const uint4* src = reinterpret_cast<const uint4*>(sc.objs);
                uint4* dst = reinterpret_cast<uint4*>(l_objs);
                for (uint32_t k = threadIdx.x; k < sc.n_objs * (uint32_t)(sizeof(CxObject) / 16); k += blockDim.x) dst[k] = src[k];
                for (uint32_t k = threadIdx.x; k < n_entries; k += blockDim.x) l_xobj[k] = sc.xobj[k];
            }
            stage_tree(sc.xnodes, sc.n_xnodes, sc.xtris, n_entries, l_nodes, l_tris);       // (ends with a barrier)
            TreeMem<GpuQuadNode> mem{l_nodes, l_tris, l_xobj, l_objs};
            if (INSTANCED) {        // (the reference's two-level tree stays in global memory: only ties ask for it)
                const InstancedRefTree ref;
                traverse_queue<SHADOW, STATS, true>(mem, ref, sc, wb, count, cursor, stats, refill_threshold, wave_chunk, leaf_threshold, leaf_inner, sc.coord_bound, l_stack, lds_stack, first_base, redo_list, redo_count);
                return;
            }
            // (the copy of the reference's tree lies behind the stack part; only visibility rays consult it)
            if (SHADOW) {
                const RefTree ref = stage_ref_tree(sc, smem + scene_lds_bytes + (size_t)(lds_stack + 1) * kBlock * sizeof(double));
                traverse_queue<SHADOW, STATS, false>(mem, ref, sc, wb, count, cursor, stats, refill_threshold, wave_chunk, leaf_threshold, leaf_inner, sc.coord_bound, l_stack, lds_stack, first_base, redo_list, redo_count);
                return;
            }
            const GlobalRefTree ref;
            traverse_queue<SHADOW, STATS, false>(mem, ref, sc, wb, count, cursor, stats, refill_threshold, wave_chunk, leaf_threshold, leaf_inner, sc.coord_bound, l_stack, lds_stack, first_base, redo_list, redo_count);
        } else {
            GpuQuadNode* l_nodes = reinterpret_cast<GpuQuadNode*>(smem);
            const uint32_t n_top = sc.n_xnodes < top_nodes ? sc.n_xnodes : top_nodes;
            stage_tree(sc.xnodes, n_top, sc.xtris, 0u, l_nodes, (GpuTri*)nullptr);
            TopLdsMem<GpuQuadNode> mem{l_nodes, sc.xnodes, sc.xtris, (int32_t)n_top, sc.xobj, sc.objs};
            if (INSTANCED) {
                const InstancedRefTree ref;
                traverse_queue<SHADOW, STATS, true>(mem, ref, sc, wb, count, cursor, stats, refill_threshold, wave_chunk, leaf_threshold, leaf_inner, sc.coord_bound, l_stack, lds_stack, first_base, redo_list, redo_count);
                return;
            }
            const GlobalRefTree ref;
            traverse_queue<SHADOW, STATS, false>(mem, ref, sc, wb, count, cursor, stats, refill_threshold, wave_chunk, leaf_threshold, leaf_inner, sc.coord_bound, l_stack, lds_stack, first_base, redo_list, redo_count);
        }
    }
}

// Triangle id (or -1) and distance of the closest hit the traversal kernel recorded for `slot`; a ray it marked
// kHitRedo is traced here with the reference's own traversal (bvh_node::hit over the reference's tree).
__device__ __attribute__((noinline)) int32_t redo_closest(const SceneDev& sc, V3 o, V3 d, double tmin, double tmax, double& t) {
    GlobalScene mem{sc.nodes, sc.tris};
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    HitLite h;
    h.tri = -1; h.t = 0; h.u = 0; h.v = 0;
    if (!walk<kStatsOff, false>(mem, o, d, tmin, tmax, h, c)) return -1;
    t = h.t;
    return h.tri;
}

// The same for a scene with instances: the walk over the two-level table; returns the POSITION of the hit entry in
// the leaf order (or -1) and which object it belongs to.
BDPT_DEV int32_t redo_closest_instanced(const SceneDev& sc, V3 o, V3 d, double tmin, double tmax, CxHit& h) {
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Rng unused;
    unused.key = 0; unused.counter = 0;            // (only a medium draws, and scenes with media do not run here)
    h.tri = -1; h.obj = -1; h.t = 0; h.u = 0; h.v = 0;
    if (!cx_walk<kStatsOff, false, false>(sc, o, d, tmin, tmax, h, c, unused)) return -1;
    return (int32_t)((h.obj >= 0 ? sc.obj_pos0[h.obj] : 0u) + sc.tri_pos[h.tri]);
}
// hit_record of an entry of the leaf order that belongs to an object: position and normal as translate::hit /
// rotate_y::hit return them (hittable.h:63,135-145)
BDPT_DEV void instanced_surface(const SceneDev& sc, int32_t ob, V3 outward, V3 o, V3 d, double t_hit, V3& p, V3& normal, bool& front_face) {
    const CxObject obj = sc.objs[ob];
    V3 oo = o, od = d;
    cx_to_object(obj, oo, od);
    p = add(oo, scl(t_hit, od));
    front_face = dot(od, outward) < 0;
    normal = front_face ? outward : neg(outward);
    if (obj.flags & kCxRotate) {
        const double c = obj.cos_theta, sn = obj.sin_theta;
        p = mk((c * p.x) + (sn * p.z), p.y, (-sn * p.x) + (c * p.z));
        normal = mk((c * normal.x) + (sn * normal.z), normal.y, (-sn * normal.x) + (c * normal.z));
    }
    if (obj.flags & kCxTranslate) p = add(p, ld3(obj.offset));
}

// The closest-hit rays of one traversal launch that met an exact tie (hit_tri == kHitRedo; the traversal kernel put
// their slots on a list) are traced here in the reference's own order, densely: one listed ray per lane.  Inside the
// kernels that read the hits a lane with such a ray kept the other 63 of its wave waiting for a walk over the
// reference's median-split tree (13 % of the frame of a scene where two rays in a thousand tie), and the walk's registers
// counted against kernels that have to fit beside the resident traversal waves.
template <bool INSTANCED>
__global__ __launch_bounds__(kBlock) void k_redo(SceneDev sc, WfBuffers wb, const uint32_t* list, const uint32_t* count_ptr) {
    const uint32_t count = *count_ptr;
    const uint64_t ns = wb.ns;
    // Entry k goes to lane k / n_waves of wave k % n_waves: a short list is spread one ray per wave (such a walk is
    // long and of its own shape, and the stream waits for the slowest wave), a long one fills the lanes.
    const uint32_t n_waves = gridDim.x * (kBlock >> 6);
    const uint32_t wave = blockIdx.x * (kBlock >> 6) + (threadIdx.x >> 6);
    for (uint64_t k = wave + (uint64_t)n_waves * lane_id(); k < count; k += (uint64_t)n_waves * 64u) {
        const uint32_t slot = list[k];              // (the ray's place in the live queue)
        const V3 o = mk(wb.ray[slot], wb.ray[ns + slot], wb.ray[2 * ns + slot]);
        const V3 d = mk(wb.ray[3 * ns + slot], wb.ray[4 * ns + slot], wb.ray[5 * ns + slot]);
        const double tmax = wb.tmax ? wb.tmax[slot] : kInf;
        // First with the interval cut off just behind the distance the traversal kernel found.  The reference's walk
        // and this one visit the same nodes in the same order and accept the same triangles up to that distance (a box
        // that only the shorter interval misses starts behind it, and so does everything inside), so a hit found this way
        // IS the reference's; the walk over the unordered median-split tree, which otherwise wanders through
        // everything the ray's line meets, stays near the hit.  Nothing found (the triangle the kernel saw is one the
        // reference does not reach): once more with the interval as it was.
        // The first walk's interval also BEGINS just in front of that distance: the traversal kernel has tested every
        // triangle on the ray that the reference could accept, and none lies nearer than what it found, so the
        // reference accepts nothing in front of the cut either; a box that holds one of the triangles in question
        // reaches into the shortened interval and is passed or missed as before, the others are not looked into.
        const double hint = wb.hit[slot];
        // (distances that tie lie within 2^-19 of each other; a hit distance is good to ~1e-6 of itself even for a ray
        // that meets its triangle at the reference's smallest accepted determinant)
        const double bound = hint * 1.0009765625;                          // 1 + 2^-10
        const double from = __builtin_fmax(0.001, hint * 0.9990234375);    // 1 - 2^-10
        int32_t pos = -1;
        double t = 0.0;
        for (int pass = (bound < tmax) ? 0 : 1; pass < 2 && pos < 0; pass++) {
            const double upto = pass == 0 ? bound : tmax;
            const double lower = pass == 0 ? from : 0.001;
            if (INSTANCED) {
                CxHit rh;
                pos = redo_closest_instanced(sc, o, d, lower, upto, rh);
                t = rh.t;
            } else {
                const int32_t tri = redo_closest(sc, o, d, lower, upto, t);
                pos = tri < 0 ? -1 : (int32_t)sc.tri_pos[tri];
            }
        }
        wb.hit_tri[slot] = pos;
        if (pos >= 0) wb.hit[slot] = t;
    }
}

// ------------------------------------------------------------------ scenes with media: the walk in the reference's order
// constant_medium::hit draws a random number inside bvh_node::hit (constant_medium.h:40), from the sample's stream, so
// the rays of such a scene are traced over the reference's own two-level tree in the reference's order, one ray per
// lane, the sample's stream position read and written back.  Result codes as k_traverse writes them, plus the codes
// <= kHitMedium0 of a hit inside a medium.
// LDS: the tables the walk reads (the two-level tree, the triangles, the objects) are small enough to be copied into LDS by
// every workgroup first: the walk is a chain of dependent fetches, one ray per lane.
struct StagedWalkScene {
    const GpuNode* nodes;
    const GpuTri* tris;
    const CxObject* objs;
};
BDPT_DEV size_t staged_walk_bytes(uint32_t n_nodes, uint32_t n_tris, uint32_t n_objs) {
    return (size_t)n_nodes * sizeof(GpuNode) + (size_t)n_tris * sizeof(GpuTri) + (size_t)n_objs * sizeof(CxObject);
}
// all threads of the workgroup; `dst` 16-byte aligned; ends with a barrier
BDPT_DEV StagedWalkScene stage_walk_scene(const SceneDev& sc, unsigned char* dst) {
    GpuNode* l_nodes = reinterpret_cast<GpuNode*>(dst);
    GpuTri* l_tris = reinterpret_cast<GpuTri*>(l_nodes + sc.n_nodes);
    CxObject* l_objs = reinterpret_cast<CxObject*>(l_tris + sc.n_tris);
    const uint32_t n16 = (uint32_t)(staged_walk_bytes(sc.n_nodes, sc.n_tris, sc.n_objs) / 16);
    const uint32_t a16 = sc.n_nodes * (uint32_t)(sizeof(GpuNode) / 16), b16 = a16 + sc.n_tris * (uint32_t)(sizeof(GpuTri) / 16);
    uint4* out = reinterpret_cast<uint4*>(dst);
    for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x)
        out[k] = k < a16 ? reinterpret_cast<const uint4*>(sc.nodes)[k] : (k < b16 ? reinterpret_cast<const uint4*>(sc.tris)[k - a16] : reinterpret_cast<const uint4*>(sc.objs)[k - b16]);
    __syncthreads();
    return StagedWalkScene{l_nodes, l_tris, l_objs};
}
template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_walk_ordered(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, const uint32_t* queue,
                                                         const uint32_t* count_ptr, uint32_t count_imm) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t count = count_ptr ? *count_ptr : count_imm;
    const uint64_t ns = wb.ns;
    StagedWalkScene staged = {nullptr, nullptr, nullptr};
    if (LDS) staged = stage_walk_scene(sc, smem);
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < count; k += gridDim.x * kBlock) {
        const uint32_t slot = queue ? queue[k] : k;
        int pi, pj, ps;
        if (slot == kDeadSlot || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
        const V3 o = mk(wb.ray[k], wb.ray[ns + k], wb.ray[2 * ns + k]);
        const V3 d = mk(wb.ray[3 * ns + k], wb.ray[4 * ns + k], wb.ray[5 * ns + k]);
        const uint32_t ctr0 = wb.ctrq[k];
        Rng rng = slot_rng(cam, b, slot, ctr0);
        Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        CxHit h;
        int32_t code = -1;
        if (LDS ? cx_walk<kStatsOff>(staged, o, d, 0.001, kInf, h, c, rng) : cx_walk<kStatsOff>(sc, o, d, 0.001, kInf, h, c, rng)) {
            code = (h.tri < 0) ? kHitMedium0 - h.obj : (int32_t)((h.obj >= 0 ? sc.obj_pos0[h.obj] : 0u) + sc.tri_pos[h.tri]);
            wb.hit[k] = h.t;
        }
        wb.hit_tri[k] = code;
        if (rng.counter != ctr0) wb.ctrq[k] = rng.counter;
    }
}

// camera::bidirectional_color's connection loop (camera.h:316-320) for such a scene: the pairs of a sample one after
// the other, each with the reference's full closest-hit query behind camera::visible (camera.h:437) - a medium on the
// way draws from the sample's stream, so neither the order of a sample's pairs nor a pair that will contribute nothing
// can be left out.  Samples have between 1 and depth^2 pairs: a workgroup takes a tile of samples, orders them by their
// number of pairs (counting sort in LDS) and works in rounds - round r is pair r of every sample that has one, which
// is a prefix of the ordered tile, one sample per thread - with stream position and radiance of the samples in LDS
// between the rounds.  (One sample per lane with a loop over its pairs: a wave then takes as long as its longest
// sample, twice as long in a Cornell box with smoke.)
constexpr uint32_t kOrderedMaxPairs = 1024;       // histogram size: samples with more pairs share the last bin
template <bool LDS>
__global__ __launch_bounds__(kStageBlock) void k_conn_ordered(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, const uint32_t* conn_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    StagedWalkScene staged = {nullptr, nullptr, nullptr};
    if (LDS) staged = stage_walk_scene(sc, smem);
    __shared__ uint32_t s_slot[kStageTile], s_cnt[kStageTile], s_ctr[kStageTile];
    __shared__ unsigned long long s_key[kStageTile];
    __shared__ double s_res[3][kStageTile];
    __shared__ uint16_t s_order[kStageTile];
    __shared__ uint32_t s_hist[kOrderedMaxPairs + 1];
    __shared__ uint32_t s_max;
    const uint32_t count = *conn_count;
    const int depth = cam.max_depth;
    const uint64_t ns = wb.ns;
    // bins in use: a sample has at most depth x depth pairs
    const uint32_t nb = ((uint64_t)depth * (uint64_t)depth < kOrderedMaxPairs) ? (uint32_t)(depth * depth) : kOrderedMaxPairs;
    for (uint32_t tile = blockIdx.x * kStageTile; tile < count; tile += gridDim.x * kStageTile) {
        const uint32_t n_el = (count - tile < kStageTile) ? count - tile : kStageTile;
        for (uint32_t i = threadIdx.x; i <= nb; i += kStageBlock) s_hist[i] = 0u;
        if (threadIdx.x == 0) s_max = 0u;
        __syncthreads();
        // state of the tile's samples; histogram of their pair counts
        for (uint32_t el = threadIdx.x; el < n_el; el += kStageBlock) {
            const uint32_t slot = wb.conn_queue[tile + el];
            const uint32_t cnt = wb.counts[slot];
            s_slot[el] = slot;
            s_cnt[el] = cnt;
            s_ctr[el] = wb.ctr[slot];
            int pi, pj, ps;
            slot_to_sample(cam, b, slot, pi, pj, ps);
            s_key[el] = stream_key(b.seed, (uint64_t)pj * (uint64_t)cam.width + (uint64_t)pi, (uint64_t)ps);
            s_res[0][el] = wb.result[slot]; s_res[1][el] = wb.result[ns + slot]; s_res[2][el] = wb.result[2 * ns + slot];
            const uint32_t pairs = (cnt & 0xFFFu) * ((cnt >> 12) & 0xFFFu);
            atomicAdd(&s_hist[pairs < nb ? pairs : nb], 1u);
            atomicMax(&s_max, pairs);
        }
        __syncthreads();
        // s_hist[p] := number of samples with MORE than p pairs (= the samples of round p); the ordered tile
        if (threadIdx.x == 0) {
            uint32_t above = 0;
            for (int p = (int)nb; p >= 0; p--) { const uint32_t here = s_hist[p]; s_hist[p] = above; above += here; }
        }
        __syncthreads();
        for (uint32_t el = threadIdx.x; el < n_el; el += kStageBlock) {
            const uint32_t cnt = s_cnt[el];
            const uint32_t pairs = (cnt & 0xFFFu) * ((cnt >> 12) & 0xFFFu);
            // samples with more pairs first: those with exactly `pairs` stand behind the s_hist[pairs] that have more
            const uint32_t at = atomicAdd(&s_hist[pairs < nb ? pairs : nb], 1u);
            s_order[at] = (uint16_t)el;
        }
        __syncthreads();
        // (s_hist[p] now counts the samples with p pairs or more; round r takes the first s_hist[r + 1] of the order)
        const uint32_t rounds = s_max;
        for (uint32_t r = 0; r < rounds; r++) {
            uint32_t n_r;
            if (r + 1 <= nb) n_r = s_hist[r + 1];
            else n_r = s_hist[nb];                            // (beyond the histogram: checked per sample below)
            for (uint32_t i = threadIdx.x; i < n_r; i += kStageBlock) {
                const uint32_t el = s_order[i];
                const uint32_t cnt = s_cnt[el];
                const uint32_t n_light = (cnt >> 12) & 0xFFFu;
                if (r >= (cnt & 0xFFFu) * n_light) continue;
                const uint32_t a = r / n_light, bi = r - a * n_light;
                const uint32_t slot = s_slot[el];
                const VertexStore store = slot_store(wb, slot);
                const Vertex cv = store.get<true>((int)a);
                const Vertex lv = store.get<true>(depth + (int)bi);
                V3 term = mk(0, 0, 0);
                PairGeom g;
                if (pair_geometry(cv, lv, g)) {                                       // camera.h:444-453, 426-436
                    Rng rng;
                    rng.key = s_key[el];
                    rng.counter = s_ctr[el];
                    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    CxHit tmp;
                    const bool blocked = LDS ? cx_walk<kStatsOff>(staged, g.origin, g.dir_unit, 0.001, g.max_t, tmp, c, rng)
                                             : cx_walk<kStatsOff>(sc, g.origin, g.dir_unit, 0.001, g.max_t, tmp, c, rng);
                    s_ctr[el] = rng.counter;
                    if (!blocked) {
                        const V3 f_cam = (cv.info & kVtxIsLight) ? mk(0, 0, 0) : cv.f;     // material.h:35-37
                        if (!(len2(f_cam) <= 0) && len2(lv.f) > 0) term = pair_contribution(cv, lv, g, f_cam, lv.f);
                    }
                }
                s_res[0][el] = s_res[0][el] + term.x; s_res[1][el] = s_res[1][el] + term.y; s_res[2][el] = s_res[2][el] + term.z;
            }
            __syncthreads();
        }
        for (uint32_t el = threadIdx.x; el < n_el; el += kStageBlock) {
            const uint32_t slot = s_slot[el];
            wb.result[slot] = s_res[0][el]; wb.result[ns + slot] = s_res[1][el]; wb.result[2 * ns + slot] = s_res[2][el];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ texture values of one bounce
// texture::value(u, v, p) (texture.h) of the hits whose material carries a texture, as a pass of its own in front of
// k_shade: a workgroup takes a tile of the queue, collects the entries with such a hit in LDS - those whose texture is a
// noise_texture (perlin::turb, seven octaves of eight corners each: long, and hungry for registers) from the front of the
// list, the others from its back - and evaluates the list densely, one entry per lane.  Inside k_shade the same call cost
// every wave the whole evaluation for the handful of its lanes that stand on a textured surface, and its registers
// (217 against 118) cost the kernel half of its waves.  The value goes to WfBuffers::tex_rgb, by queue entry.
template <bool MEDIA>
__global__ __launch_bounds__(kStageBlock) void k_tex_eval(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, const uint32_t* queue_in,
                                                           const uint32_t* count_in_ptr, uint32_t count_in_imm) {
    __shared__ uint32_t s_list[kStageTile];
    __shared__ uint32_t s_front, s_back;
    const uint32_t count = count_in_ptr ? *count_in_ptr : count_in_imm;
    const uint64_t ns = wb.ns;
    for (uint32_t tile = blockIdx.x * kStageTile; tile < count; tile += gridDim.x * kStageTile) {
        if (threadIdx.x == 0) { s_front = 0u; s_back = 0u; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t k = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            if (k >= count) continue;
            const uint32_t slot = queue_in ? queue_in[k] : k;
            int pi, pj, ps;
            if (slot == kDeadSlot || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
            const int32_t pos = wb.hit_tri[k];
            int32_t mat_id = -1;
            if (MEDIA && pos <= kHitMedium0) mat_id = sc.objs[kHitMedium0 - pos].phase_material;
            else if (pos >= 0) mat_id = (int32_t)(sc.shade[pos].mat & kVtxMatMask);
            if (mat_id < 0) continue;
            const int32_t tex1 = sc.mats[mat_id].pad;
            if (tex1 <= 0) continue;
            if (sc.tex[tex1 - 1].type == BDPT_TEX_NOISE) s_list[atomicAdd(&s_front, 1u)] = k;
            else s_list[kStageTile - 1u - atomicAdd(&s_back, 1u)] = k;
        }
        __syncthreads();
        const uint32_t n_front = s_front, n_all = s_front + s_back;
        for (uint32_t i = threadIdx.x; i < n_all; i += kStageBlock) {
            const uint32_t k = (i < n_front) ? s_list[i] : s_list[kStageTile - 1u - (i - n_front)];
            const int32_t pos = wb.hit_tri[k];
            const double t_hit = wb.hit[k];
            const V3 o = mk(wb.ray[k], wb.ray[ns + k], wb.ray[2 * ns + k]);
            const V3 d = mk(wb.ray[3 * ns + k], wb.ray[4 * ns + k], wb.ray[5 * ns + k]);
            V3 value;
            if (MEDIA && pos <= kHitMedium0) {                                   // constant_medium.h:45-50: u = v = 0
                const int32_t mat_id = sc.objs[kHitMedium0 - pos].phase_material;
                value = cx_texture(sc, sc.mats[mat_id].pad - 1, 0.0, 0.0, add(o, scl(t_hit, d)));
            } else {
                // the hit_record as k_shade forms it: the point through the object's wrappers, (u, v) by the operations of the
                // triangle test that found the hit
                const GpuShadeRec r = sc.shade[pos];
                const int32_t ob = sc.pos_obj[pos];
                V3 p, normal;
                bool front_face;
                V3 to = o, td = d;
                if (ob >= 0) {
                    instanced_surface(sc, ob, ld3(r.normal), o, d, t_hit, p, normal, front_face);
                    cx_to_object(sc.objs[ob], to, td);
                } else {
                    p = add(o, scl(t_hit, d));
                }
                double t, u, v, det;
                (void)tri_candidate(sc.tris[r.tri], to, td, 0.001, t, u, v, det);
                value = cx_texture(sc, sc.mats[r.mat & kVtxMatMask].pad - 1, u, v, p);
            }
            wb.tex_rgb[k] = value.x; wb.tex_rgb[ns + k] = value.y; wb.tex_rgb[2 * ns + k] = value.z;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ shading of one bounce
// TEX (with INSTANCED): materials may carry a texture.  Its value at the hit (texture::value(u, v, p), material.h:45,135)
// takes the place of the material's solid colour, and the vertex record keeps the colour the vertex can contribute
// with; u and v are formed again from the ray and the hit triangle, by the operations that formed them in the
// triangle test (the traversal kernels keep no u, v).
// MEDIA (with TEX): the hits come from k_walk_ordered; a negative code <= kHitMedium0 stands for a point inside the
// medium -(code - kHitMedium0) (constant_medium.h:45-50: normal (1, 0, 0), front face, the medium's phase function).
// Queue order.  A path's ray, throughput and stream position are read from its entry k of the live queue and, when it goes
// on, written to its entry of the next queue, so both are contiguous accesses.  For that the entry of the next queue has
// to be known before the path is shaded: every hit on a surface that scatters (anything but an emitter) gets one, a tile
// of entries at a time with one atomic; whether such a path really goes on is then all but certain (a lambertian sample
// with a pdf of zero or a NaN is the exception), and the exception leaves kDeadSlot and a direction of zero behind.
// keep_draws: a path that ends leaves its number of draws with its slot (the light subpath starts there, the connection
// stage of a scene with media goes on from there, the counting modes add them up).
template <bool EYE, bool STATS, bool LAMBERT, bool INSTANCED = false, bool TEX = false, bool MEDIA = false>
__global__ __launch_bounds__(kStageBlock) void k_shade(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, const uint32_t* queue_in,
                                                  const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out,
                                                  uint32_t* count_out, int last_bounce, int store_vertices, int keep_draws, int sort_shift,
                                                  unsigned long long* stats) {
    __shared__ uint32_t s_wave[kSortBins > kStageWaves ? kSortBins : kStageWaves];
    __shared__ uint32_t s_base;
    const uint32_t count = count_in_ptr ? *count_in_ptr : count_in_imm;
    const uint64_t ns = wb.ns;
    const int depth = cam.max_depth;
    uint32_t n_stored = 0;
    // whole workgroups iterate together so that the append sees every thread
    for (uint32_t tile = blockIdx.x * kStageTile; tile < count; tile += gridDim.x * kStageTile) {
        uint32_t slots[kStagePerThread], at[kStagePerThread], key[kStagePerThread];
        bool may_go_on[kStagePerThread];
#pragma unroll
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t k = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            slots[j] = kDeadSlot;
            at[j] = 0;
            key[j] = 0;
            may_go_on[j] = false;
            if (k >= count) continue;
            const uint32_t slot = queue_in ? queue_in[k] : k;
            int pi, pj, ps;
            if (slot == kDeadSlot || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
            slots[j] = slot;
            if (last_bounce) continue;
            const int32_t pos = wb.hit_tri[k];
            if (MEDIA && pos <= kHitMedium0) may_go_on[j] = true;                   // the phase function of a medium always scatters
            else if (pos >= 0) {
                may_go_on[j] = (sc.shade[pos].mat >> 30) != 3u;                     // every material but an emitter may
                // paths that leave from the same part of the scene become neighbours in the next queue (the positions of
                // the reference's leaf order follow its spatial median splits)
                const uint32_t cell = (uint32_t)pos >> (sort_shift < 0 ? 0 : sort_shift);
                key[j] = cell < kSortBins ? cell : kSortBins - 1u;
            }
        }
        if (sort_shift >= 0) tile_append_sorted(may_go_on, slots, key, queue_out, count_out, s_wave, &s_base, at);
        else tile_append_at(may_go_on, slots, queue_out, count_out, s_wave, &s_base, at);
#pragma unroll 1
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t k = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            const uint32_t slot = slots[j];
            if (slot == kDeadSlot) continue;
            int32_t pos = wb.hit_tri[k];           // position of the hit triangle in the reference's leaf order
            const bool medium = MEDIA && pos <= kHitMedium0;
            double t_hit = 0.0;
            if (pos >= 0 || medium) t_hit = wb.hit[k];        // (k_redo has resolved the rays the traversal kernel left open)
            // (no queue: the camera rays themselves, throughput 1)
            V3 T = (EYE && !queue_in) ? mk(1.0, 1.0, 1.0) : mk(wb.thr[k], wb.thr[ns + k], wb.thr[2 * ns + k]);
            const uint32_t ctr0 = wb.ctrq[k];
            if (pos < 0 && !medium) {
                if (EYE) {   // camera.h:337-338: background_contrib (0) += throughput * background
                    V3 bg = add(mk(0, 0, 0), mulv(T, ld3(cam.background)));
                    wb.result[slot] = bg.x; wb.result[ns + slot] = bg.y; wb.result[2 * ns + slot] = bg.z;
                }
                if (keep_draws) wb.ctr[slot] = ctr0;
                continue;
            }
            V3 o = mk(wb.ray[k], wb.ray[ns + k], wb.ray[2 * ns + k]);
            V3 d = mk(wb.ray[3 * ns + k], wb.ray[4 * ns + k], wb.ray[5 * ns + k]);
            int pi, pj, ps;
            slot_to_sample(cam, b, slot, pi, pj, ps);
            Rng rng;
            rng.key = stream_key(b.seed, (uint64_t)pj * (uint64_t)cam.width + (uint64_t)pi, (uint64_t)ps);
            rng.counter = ctr0;
            Bounce bo;
            if (MEDIA && medium) {
                const int32_t mat_id = sc.objs[kHitMedium0 - pos].phase_material;
                const GpuMaterial gm = sc.mats[mat_id];
                ShadeMat m;
                m.normal = mk(1, 0, 0); m.rgb = ld3(gm.rgb); m.fuzz = gm.fuzz; m.ior = gm.ior; m.type = gm.type; m.id = mat_id;
                const V3 p = add(o, scl(t_hit, d));
                if (gm.pad > 0) m.rgb = mk(wb.tex_rgb[k], wb.tex_rgb[ns + k], wb.tex_rgb[2 * ns + k]);      // (k_tex_eval)
                shade_surface<EYE, LAMBERT, MEDIA>(m, 0, p, mk(1, 0, 0), true, d, T, rng, bo);
                bo.vx.f = (m.type == 4) ? divs(m.rgb, 4 * kPi) : mk(0, 0, 0);
            } else if (INSTANCED) {
                // one copy of the material code: the hit_record first (through the object's wrappers or not), then shading
                const GpuShadeRec r = sc.shade[pos];
                ShadeMat m;
                m.normal = ld3(r.normal); m.rgb = ld3(r.rgb); m.fuzz = r.param; m.ior = r.param;
                m.type = (int32_t)(r.mat >> 30); m.id = (int32_t)(r.mat & kVtxMatMask);
                const int32_t ob = sc.pos_obj[pos];
                V3 p, normal;
                bool front_face;
                if (ob >= 0) {
                    instanced_surface(sc, ob, m.normal, o, d, t_hit, p, normal, front_face);
                } else {
                    p = add(o, scl(t_hit, d));
                    front_face = dot(d, m.normal) < 0;
                    normal = front_face ? m.normal : neg(m.normal);
                }
                // (the vertex of such a scene names its entry of `stris`, not its triangle: SceneDev::tri_xpos)
                const int32_t vertex_tri = (int32_t)pos;
                // (the value of the material's texture at the hit: k_tex_eval has left it with the queue entry)
                if (TEX && sc.mats[m.id].pad > 0) m.rgb = mk(wb.tex_rgb[k], wb.tex_rgb[ns + k], wb.tex_rgb[2 * ns + k]);
                shade_surface<EYE, LAMBERT, MEDIA>(m, vertex_tri, p, normal, front_face, d, T, rng, bo);
                if (TEX) bo.vx.f = (m.type == 3) ? (front_face ? m.rgb : mk(0, 0, 0)) : ((m.type == 0) ? divs(m.rgb, kPi) : mk(0, 0, 0));
            } else {
                shade_bounce_at<EYE, LAMBERT>(sc, (uint32_t)pos, o, d, t_hit, T, rng, bo);
            }
            if (bo.keep && store_vertices) {       // (no vertex store in a scene that cannot form a connection)
                uint32_t cnt = wb.counts[slot];
                const uint32_t ne = cnt & 0xFFFu, nl = (cnt >> 12) & 0xFFFu;
                slot_store(wb, slot).put<TEX>(EYE ? (int)ne : depth + (int)nl, bo.vx);
                cnt += EYE ? 1u : (1u << 12);
                if (EYE && !(bo.vx.info & kVtxIsLight)) cnt |= kCntEyeConnectable;
                if (STATS) n_stored++;
                wb.counts[slot] = cnt;
            }
            if (EYE && bo.emitted) {   // result = background_contrib (0) + throughput * emission
                V3 e = add(mk(0, 0, 0), bo.emitted_term);
                wb.result[slot] = e.x; wb.result[ns + slot] = e.y; wb.result[2 * ns + slot] = e.z;
            }
            if (may_go_on[j]) {
                const uint32_t to = at[j];
                if (bo.cont) {
                    wb.ray_next[to] = bo.o.x; wb.ray_next[ns + to] = bo.o.y; wb.ray_next[2 * ns + to] = bo.o.z;
                    wb.ray_next[3 * ns + to] = bo.d.x; wb.ray_next[4 * ns + to] = bo.d.y; wb.ray_next[5 * ns + to] = bo.d.z;
                    wb.thr_next[to] = bo.T.x; wb.thr_next[ns + to] = bo.T.y; wb.thr_next[2 * ns + to] = bo.T.z;
                    wb.ctrq_next[to] = rng.counter;
                } else {               // it ends after all: nobody's entry
                    queue_out[to] = kDeadSlot;
                    wb.ray_next[3 * ns + to] = 0.0; wb.ray_next[4 * ns + to] = 0.0; wb.ray_next[5 * ns + to] = 0.0;
                }
            }
            if (keep_draws && !(may_go_on[j] && bo.cont)) wb.ctr[slot] = rng.counter;
        }
    }
    if (STATS) flush_wave_counter(&stats[10], n_stored);
}

// ------------------------------------------------------------------ one level of the unidirectional estimator
// camera::path_trace_color (camera.h:255-292) as a pipeline stage: the level's record goes to the sample's vertex
// workspace (slot `level`), a sample whose recursion ends here folds its records back to front into its radiance.
template <bool STATS>
__global__ __launch_bounds__(kStageBlock) void k_shade_unidirectional(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, const uint32_t* queue_in,
                                                                      const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out,
                                                                      uint32_t* count_out, int level, int last_level, unsigned long long* stats) {
    __shared__ uint32_t s_wave[kStageWaves];
    __shared__ uint32_t s_base;
    const uint32_t count = count_in_ptr ? *count_in_ptr : count_in_imm;
    const uint64_t ns = wb.ns;
    unsigned long long draws = 0, samples = 0;
    for (uint32_t tile = blockIdx.x * kStageTile; tile < count; tile += gridDim.x * kStageTile) {
        // (queue order, as in k_shade: every hit may go on and gets its entry of the next queue before it is shaded)
        bool may_go_on[kStagePerThread];
        uint32_t slots[kStagePerThread], at[kStagePerThread];
#pragma unroll
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t k = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            slots[j] = kDeadSlot;
            at[j] = 0;
            may_go_on[j] = false;
            if (k >= count) continue;
            const uint32_t slot = queue_in ? queue_in[k] : k;
            int pi, pj, ps;
            if (slot == kDeadSlot || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
            slots[j] = slot;
            may_go_on[j] = !last_level && wb.hit_tri[k] >= 0;
        }
        tile_append_at(may_go_on, slots, queue_out, count_out, s_wave, &s_base, at);
#pragma unroll 1
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t k = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            const uint32_t slot = slots[j];
            if (slot == kDeadSlot) continue;
            bool goes_on = false;
            const V3 o = mk(wb.ray[k], wb.ray[ns + k], wb.ray[2 * ns + k]);
            const V3 d = mk(wb.ray[3 * ns + k], wb.ray[4 * ns + k], wb.ray[5 * ns + k]);
            const int32_t pos = wb.hit_tri[k];
            HitLite h;
            h.tri = pos < 0 ? -1 : sc.stris[pos].pad; h.t = 0; h.u = 0; h.v = 0;
            if (pos >= 0) h.t = wb.hit[k];
            V3 value = mk(0, 0, 0);           // what the innermost level returns
            int n_records = level;
            Rng rng = slot_rng(cam, b, slot, wb.ctrq[k]);
            if (h.tri < 0) {
                value = ld3(cam.background);                                    // camera.h:264-265
            } else {
                Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                UniBounce ub;
                unidirectional_bounce<kStatsOff>(sc, o, d, h, rng, c, ub);
                if (!ub.cont) {
                    value = ub.value;
                } else {
                    slot_store(wb, slot).put(level, ub.rec);
                    n_records = level + 1;
                    if (!last_level) {
                        goes_on = true;
                        const uint32_t to = at[j];
                        wb.ray_next[to] = ub.o.x; wb.ray_next[ns + to] = ub.o.y; wb.ray_next[2 * ns + to] = ub.o.z;
                        wb.ray_next[3 * ns + to] = ub.d.x; wb.ray_next[4 * ns + to] = ub.d.y; wb.ray_next[5 * ns + to] = ub.d.z;
                        wb.ctrq_next[to] = rng.counter;
                    }
                    // (at the last level the next call has no depth left and returns zero, camera.h:258-259)
                }
            }
            if (!goes_on) {
                if (may_go_on[j]) {           // the recursion ends at this hit after all: nobody's entry
                    const uint32_t to = at[j];
                    queue_out[to] = kDeadSlot;
                    wb.ray_next[3 * ns + to] = 0.0; wb.ray_next[4 * ns + to] = 0.0; wb.ray_next[5 * ns + to] = 0.0;
                }
                for (int q = n_records - 1; q >= 0; q--) value = unidirectional_fold(slot_store(wb, slot).get(q), value);
                wb.result[slot] = value.x; wb.result[ns + slot] = value.y; wb.result[2 * ns + slot] = value.z;
                if (STATS) { draws += rng.counter; samples++; }
            }
        }
    }
    if (STATS) { flush_wave_counter(&stats[8], draws); flush_wave_counter(&stats[9], samples); }
}

// ------------------------------------------------------------------ light subpath start
// the colour of an emitter's material at a point of it, textures included (u = v = 0 there, camera.h:385-391)
struct TexturedEmitterColour {
    const SceneDev& sc;
    BDPT_DEV V3 operator()(const GpuMaterial& m, V3 position) const { return cx_colour(sc, m, 0.0, 0.0, position); }
};
template <bool STATS, bool TEX = false>
__global__ __launch_bounds__(kStageBlock) void k_gen_light(SceneDev sc, GpuCamera cam, WfBuffers wb, WfBatch b, uint32_t* queue_out,
                                                      uint32_t* count_out, unsigned long long* stats) {
    const uint32_t spp = (uint32_t)(cam.sqrt_spp * cam.sqrt_spp);
    const uint32_t n = (uint32_t)b.pixels_stride * spp;
    const uint64_t ns = wb.ns;
    const int depth = cam.max_depth;
    __shared__ uint32_t s_wave[kStageWaves];
    __shared__ uint32_t s_base;
    uint32_t n_stored = 0;
    for (uint32_t tile = blockIdx.x * kStageTile; tile < n; tile += gridDim.x * kStageTile) {
        // (queue order, as in k_shade: every sample that gets a light subpath also gets an entry of the first light queue
        // before its emitter is sampled; the rare one without a first ray leaves the entry to nobody)
        bool may_go_on[kStagePerThread];
        uint32_t slots[kStagePerThread], at[kStagePerThread];
#pragma unroll
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t slot = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            slots[j] = slot;
            at[j] = 0;
            int pi, pj, ps;
            may_go_on[j] = slot < n && slot_to_sample(cam, b, slot, pi, pj, ps) && (wb.counts[slot] & kCntEyeConnectable) && depth - 1 > 0;
        }
        tile_append_at(may_go_on, slots, queue_out, count_out, s_wave, &s_base, at);
#pragma unroll 1
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t slot = slots[j];
            int pi, pj, ps;
            if (slot >= n || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
            bool goes_on = false;
            // A light subpath only acts through its connections (camera.h:316-320), and a pair contributes zero
            // when the eye vertex is delta (camera.h:441) or on an emitter (zero BSDF, material.h:35-37): without
            // a connectable eye vertex nothing of the light subpath is sampled, traced or stored.  Its random
            // draws are the last ones of the sample's stream, so nothing else shifts.  (The reference still adds
            // the all-zero pair colours in that case; x + 0 changes only a -0, and the radiance so far is never
            // -0: every term enters it as 0 + term.)
            const uint32_t cnt = wb.counts[slot];
            if (!(cnt & kCntEyeConnectable)) continue;
            Rng rng = slot_rng(cam, b, slot, wb.ctr[slot]);
            Emitter em;
            if (TEX) sample_emitter(sc, depth, rng, em, TexturedEmitterColour{sc});
            else sample_emitter(sc, depth, rng, em);
            wb.ctr[slot] = rng.counter;
            if (em.valid) {
                if (sc.objs) em.ev.tri = sc.tri_pos[em.ev.tri];      // (an emitter is a triangle of the world's list: its entry of `stris`)
                slot_store(wb, slot).put<TEX>(depth, em.ev);
                if (STATS) n_stored++;
                wb.counts[slot] = cnt | (1u << 12);
                if (em.has_ray && may_go_on[j]) {
                    goes_on = true;
                    const uint32_t to = at[j];
                    wb.ray_next[to] = em.o.x; wb.ray_next[ns + to] = em.o.y; wb.ray_next[2 * ns + to] = em.o.z;
                    wb.ray_next[3 * ns + to] = em.d.x; wb.ray_next[4 * ns + to] = em.d.y; wb.ray_next[5 * ns + to] = em.d.z;
                    wb.thr_next[to] = em.T.x; wb.thr_next[ns + to] = em.T.y; wb.thr_next[2 * ns + to] = em.T.z;
                    wb.ctrq_next[to] = rng.counter;
                }
            }
            if (may_go_on[j] && !goes_on) {
                const uint32_t to = at[j];
                queue_out[to] = kDeadSlot;
                wb.ray_next[3 * ns + to] = 0.0; wb.ray_next[4 * ns + to] = 0.0; wb.ray_next[5 * ns + to] = 0.0;
            }
        }
    }
    if (STATS) flush_wave_counter(&stats[10], n_stored);
}

// ------------------------------------------------------------------ connections
// samples that have at least one stored vertex on both sides (a light vertex exists only where the emitter
// sample was valid and the eye subpath is connectable, see k_gen_light)
template <bool STATS>
__global__ __launch_bounds__(kStageBlock) void k_conn_prepare(GpuCamera cam, WfBuffers wb, WfBatch b, uint32_t* conn_count,
                                                         unsigned long long* stats) {
    const uint32_t spp = (uint32_t)(cam.sqrt_spp * cam.sqrt_spp);
    const uint32_t n = (uint32_t)b.pixels_stride * spp;
    __shared__ uint32_t s_wave[kStageWaves];
    __shared__ uint32_t s_base;
    unsigned long long draws = 0, samples = 0;
    for (uint32_t tile = blockIdx.x * kStageTile; tile < n; tile += gridDim.x * kStageTile) {
        bool want[kStagePerThread];
        uint32_t slots[kStagePerThread];
#pragma unroll
        for (int j = 0; j < kStagePerThread; j++) {
            const uint32_t slot = tile + (uint32_t)j * kStageBlock + threadIdx.x;
            want[j] = false;
            slots[j] = slot;
            int pi, pj, ps;
            if (slot >= n || !slot_to_sample(cam, b, slot, pi, pj, ps)) continue;
            const uint32_t cnt = wb.counts[slot];
            if (STATS) { draws += wb.ctr[slot]; samples++; }
            want[j] = (cnt & 0xFFFu) > 0 && ((cnt >> 12) & 0xFFFu) > 0;
        }
        tile_append(want, slots, wb.conn_queue, conn_count, s_wave, &s_base);
    }
    if (STATS) { flush_wave_counter(&stats[8], draws); flush_wave_counter(&stats[9], samples); }
}

// ------------------------------------------------------------------ connection stage (camera.h:316-320, 440-475)
// Candidate pairs (eye vertices x light vertices) of every tile of kConnTile slots.  A sample has them when it stored
// vertices on both sides (a light vertex exists only where the emitter sample was valid and the eye subpath is
// connectable, see k_gen_light).  Counting modes: the draws and samples of the batch as well.
static_assert(kConnTile == 64 && kConnTile == kVtxGroup, "one wave counts one tile, and k_connect takes one group of vertex records at a time");
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_conn_count(WfBuffers wb, uint32_t n_slots, unsigned long long* stats) {
    unsigned long long draws = 0, samples = 0, all_pairs = 0;
    const uint32_t n_tiles = (n_slots + kConnTile - 1) / kConnTile;
    const uint32_t n_waves = gridDim.x * (kBlock >> 6);
    for (uint32_t tile = blockIdx.x * (kBlock >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
        const uint32_t slot = tile * kConnTile + lane_id();
        unsigned long long sum = 0;
        if (slot < n_slots) {
            const uint32_t cnt = wb.counts[slot];
            sum = (unsigned long long)(cnt & 0xFFFu) * ((cnt >> 12) & 0xFFFu);
            if (STATS) { draws += wb.ctr[slot]; samples++; }
        }
        if (STATS) all_pairs += sum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if (lane_id() == 0) wb.tile_pairs[tile] = sum;
    }
    if (STATS) { flush_wave_counter(&stats[8], draws); flush_wave_counter(&stats[9], samples); flush_wave_counter(&stats[7], all_pairs); }
}

// The passes of the connection stage: the tiles are dealt out in order, as many to a pass as the shadow queue holds for
// certain (`pass_pairs` candidate pairs; one tile never has more).  One workgroup walks over the tile counts, 8192 at a
// time (eight coalesced loads in flight per thread); a pass ends in front of the first tile that no longer fits, which
// thread 0 looks for only in a stretch whose total says that it holds one.  Passes beyond the last tile are empty.
constexpr int kPlanBlock = 1024;
constexpr int kPlanPerThread = 8;
__global__ __launch_bounds__(kPlanBlock) void k_conn_plan(WfBuffers wb, uint32_t n_tiles, unsigned long long pass_pairs, uint32_t n_passes) {
    __shared__ unsigned long long s_wave[kPlanBlock / 64];
    __shared__ unsigned long long s_vals[kPlanBlock];
    __shared__ int s_search;
    // (thread 0 only)
    unsigned long long running = 0, base = 0;
    uint32_t next_pass = 1;                   // pass_bounds[next_pass] is the next end to be written
    if (threadIdx.x == 0) wb.pass_bounds[0] = 0;
    for (uint32_t blk = 0; blk < n_tiles; blk += kPlanBlock * kPlanPerThread) {
        unsigned long long v[kPlanPerThread];
        unsigned long long sum = 0;
#pragma unroll
        for (int j = 0; j < kPlanPerThread; j++) {
            const uint32_t t = blk + (uint32_t)j * kPlanBlock + threadIdx.x;
            v[j] = t < n_tiles ? wb.tile_pairs[t] : 0ull;
            sum += v[j];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if (lane_id() == 0) s_wave[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int w = 0; w < kPlanBlock / 64; w++) tot += s_wave[w];
            s_search = (running + tot - base > pass_pairs) ? 1 : 0;
            if (!s_search) running += tot;
        }
        __syncthreads();
        if (s_search) {                           // (rare: once per pass) a pass ends inside this stretch
#pragma unroll 1
            for (int j = 0; j < kPlanPerThread; j++) {
                s_vals[threadIdx.x] = v[j];
                __syncthreads();
                if (threadIdx.x == 0) {
                    for (int i = 0; i < kPlanBlock; i++) {
                        const unsigned long long r = running + s_vals[i];
                        if (r - base > pass_pairs) {          // tile blk + j * kPlanBlock + i opens the next pass
                            if (next_pass <= n_passes) wb.pass_bounds[next_pass] = blk + (uint32_t)j * kPlanBlock + (uint32_t)i;
                            next_pass++;
                            base = running;
                        }
                        running = r;
                    }
                }
                __syncthreads();
            }
        }
    }
    if (threadIdx.x == 0)
        for (uint32_t p = next_pass; p <= n_passes; p++) wb.pass_bounds[p] = n_tiles;
}

// What a connection needs of the materials (camera.h:458-469), worked out once per workgroup into LDS instead of once
// per pair from global memory (two dependent loads in the middle of every pair): the BSDF value of each material and
// its colour (the radiance of an emitter).  Scenes with more materials than kPairMats read the global table per pair.
constexpr uint32_t kPairMats = 128;
struct PairMats {
    const double* table;        // LDS: [material][6] = BSDF value, colour; nullptr: read sc.mats
    BDPT_DEV V3 bsdf(const SceneDev& sc, uint32_t mat) const {
        if (table) return mk(table[mat * 6u], table[mat * 6u + 1u], table[mat * 6u + 2u]);
        return evaluate_bsdf(sc.mats[mat]);
    }
    // emission or BSDF value of a light-subpath vertex
    BDPT_DEV V3 light_factor(const SceneDev& sc, uint32_t info) const {
        const uint32_t mat = info & kVtxMatMask;
        if (info & kVtxIsLight) {
            if (!(info & kVtxFrontFace)) return mk(0, 0, 0);
            if (table) return mk(table[mat * 6u + 3u], table[mat * 6u + 4u], table[mat * 6u + 5u]);
            return ld3(sc.mats[mat].rgb);
        }
        return bsdf(sc, mat);
    }
};
// all threads of the workgroup; a barrier must follow before the table is read
BDPT_DEV PairMats stage_pair_mats(const SceneDev& sc, double* lds) {
    if (sc.n_mats > kPairMats) return PairMats{nullptr};
    for (uint32_t k = threadIdx.x; k < sc.n_mats; k += blockDim.x) {
        const GpuMaterial m = sc.mats[k];
        const V3 f = evaluate_bsdf(m);
        lds[k * 6u] = f.x; lds[k * 6u + 1u] = f.y; lds[k * 6u + 2u] = f.z;
        lds[k * 6u + 3u] = m.rgb[0]; lds[k * 6u + 4u] = m.rgb[1]; lds[k * 6u + 5u] = m.rgb[2];
    }
    return PairMats{lds};
}

// A vertex record out of a wave's staging area: lds[field * budget + index]
template <bool TEX>
BDPT_DEV Vertex staged_vertex(const double* lds, uint32_t budget, uint32_t idx) {
    const double* b = lds + idx;
    Vertex v;
    v.p = mk(b[0], b[budget], b[2 * budget]);
    v.n = mk(b[3 * budget], b[4 * budget], b[5 * budget]);
    v.T = mk(b[6 * budget], b[7 * budget], b[8 * budget]);
    const unsigned long long w = (unsigned long long)__double_as_longlong(b[9 * budget]);
    v.info = (uint32_t)w;
    v.tri = (uint32_t)(w >> 32);
    v.f = TEX ? mk(b[10 * budget], b[11 * budget], b[12 * budget]) : mk(0, 0, 0);
    return v;
}

// The shadow rays of ALL pairs (eye vertex a, light vertex b) of the samples of one pass.
//
// Memory.  A wave works on one group of 64 neighbouring slots at a time (the vertex records are laid out in such
// groups).  It copies the records of a run of those slots into LDS with row loads - one field of one record of up to
// 64 neighbouring slots is one contiguous piece of memory - and every pair then reads its two vertices from LDS.  A
// vertex is fetched once however many pairs it takes part in, and no load is a gather.  (Before, a launch per eye
// vertex read both vertices of every pair with 8-byte gathers from global memory - the light vertices of a sample once
// per eye vertex - and the texture addresser was busy 84 % of the kernel's time: 150 of the 375 ms a Cornell-box frame
// spends in kernels.)
//
// Pairs.  The pairs of a run of WHOLE samples that together have at most 64 are dealt to the lanes (a sample with more
// than 64 takes rounds of its own), numbered sample by sample, eye vertex by eye vertex: the reference's order
// (camera.h:316-320).  Only pairs that need a ray get a queue entry; a round's entries are consecutive, and a wave takes
// the queue `chunk` entries at a time and never lets the entries of one sample straddle two such pieces (what is left
// of a piece gets a negative max_t, which the traversal skips).  So the entries of a sample are contiguous and in pair
// order: ent_first / ent_count are all k_conn_add needs.
//
// OWN_TEST (scenes whose tables are staged in LDS): the far end point of a shadow ray lies ON a surface at t = max_t, and
// in most unoccluded pairs rounding makes that surface's own triangle accept (t <= max_t): the reference then calls the
// pair blocked (camera.h:437).  That triangle is tested here, before any ray exists - it is the light vertex's own
// triangle, known from the vertex - with the traversal kernel's rules (a hit at the end of the interval counts only if
// the reference's boxes let it through), so most blocked pairs never become a ray; most rays that remain come back
// clear, so what the pair adds (camera.h:471-474) is kept with the entry.  "Some triangle accepts and is reachable"
// does not depend on which such triangle is found first.  On a large scene the test is two dependent loads from random
// places of global memory per pair, and most of its pairs are blocked by something the traversal finds sooner (measured
// on a 0.96 M-triangle lattice: 2 361 -> 1 446 Msamples/s with the test): those scenes start the ray's walk at that
// triangle instead (WfBuffers::shadow_first) and evaluate the few clear pairs when they are added.
// TEX: the factors of a pair come from its vertices (Vertex::f), not from the material table.
// INST (with OWN_TEST): a scene with instances.  A vertex names an entry of the leaf order, the test reads the entry's
// record of the traversal hierarchy's table (triangle, thresholds, face bits) and takes the ray as the entry's object sees it.
template <bool STATS, bool OWN_TEST, bool TEX, bool INST = false>
__global__ __launch_bounds__(kBlock) void k_connect(SceneDev sc, GpuCamera cam, WfBuffers wb, uint32_t n_slots, int pass,
                                                   uint32_t* shadow_count, uint32_t budget, uint32_t chunk, uint32_t n_entries, unsigned long long* stats) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_pair_mats[kPairMats * 6];
    const uint32_t tile0 = wb.pass_bounds[pass], tile1 = wb.pass_bounds[pass + 1];
    if (tile0 >= tile1) return;                  // an empty pass: the whole workgroup, before anything is staged
    constexpr uint32_t W = TEX ? kVtxDoublesTextured : kVtxDoubles;
    const uint32_t n_waves_block = blockDim.x >> 6;
    const PairMats pmats = stage_pair_mats(sc, s_pair_mats);
    RefTree ref{sc.nodes, sc.node_parent};
    const GpuTri* own_tris = sc.tris;
    const int32_t* own_graze = sc.tri_graze;
    const uint32_t* own_leaf = sc.tri_leaf;
    const CxObject* own_objs = sc.objs;
    const int32_t* own_xobj = sc.xobj;
    const uint32_t* own_xpos = sc.tri_xpos;
    if (OWN_TEST && INST) {     // the hierarchy's triangle table, the objects, and per entry its object and its place in that table
        unsigned char* at0 = smem + (size_t)n_waves_block * budget * W * sizeof(double);
        GpuTri* l_tris = reinterpret_cast<GpuTri*>(at0);
        CxObject* l_objs = reinterpret_cast<CxObject*>(l_tris + n_entries);
        int32_t* l_xobj = reinterpret_cast<int32_t*>(l_objs + sc.n_objs);
        uint32_t* l_xpos = reinterpret_cast<uint32_t*>(l_xobj + n_entries);
        const uint4* src = reinterpret_cast<const uint4*>(sc.xtris);
        uint4* dst = reinterpret_cast<uint4*>(l_tris);
        for (uint32_t k = threadIdx.x; k < n_entries * (uint32_t)(sizeof(GpuTri) / 16); k += blockDim.x) dst[k] = src[k];
        src = reinterpret_cast<const uint4*>(sc.objs);
        dst = reinterpret_cast<uint4*>(l_objs);
        for (uint32_t k = threadIdx.x; k < sc.n_objs * (uint32_t)(sizeof(CxObject) / 16); k += blockDim.x) dst[k] = src[k];
        for (uint32_t k = threadIdx.x; k < n_entries; k += blockDim.x) { l_xobj[k] = sc.xobj[k]; l_xpos[k] = sc.tri_xpos[k]; }
        own_tris = l_tris; own_objs = l_objs; own_xobj = l_xobj; own_xpos = l_xpos;
    } else if (OWN_TEST) {         // small scene: the reference's tree and the triangle tables the test reads, in LDS behind the staging areas
        unsigned char* at0 = smem + (size_t)n_waves_block * budget * W * sizeof(double);
        unsigned char* at = at0 + ((ref_tree_lds_bytes(sc.n_nodes) + 15) & ~(size_t)15);
        GpuTri* l_tris = reinterpret_cast<GpuTri*>(at);
        int32_t* l_graze = reinterpret_cast<int32_t*>(l_tris + sc.n_tris);
        uint32_t* l_leaf = reinterpret_cast<uint32_t*>(l_graze + sc.n_tris);
        const uint4* src = reinterpret_cast<const uint4*>(sc.tris);
        uint4* dst = reinterpret_cast<uint4*>(l_tris);
        for (uint32_t k = threadIdx.x; k < sc.n_tris * (uint32_t)(sizeof(GpuTri) / 16); k += blockDim.x) dst[k] = src[k];
        for (uint32_t k = threadIdx.x; k < sc.n_tris; k += blockDim.x) { l_graze[k] = sc.tri_graze[k]; l_leaf[k] = sc.tri_leaf[k]; }
        ref = stage_ref_tree(sc, at0);        // (ends with a barrier)
        own_tris = l_tris; own_graze = l_graze; own_leaf = l_leaf;
    }
    __syncthreads();
    const GrazeBits gbits = graze_bits(sc.bary_margin, 1.0001f);
    const uint32_t depth = (uint32_t)cam.max_depth;
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x >> 6;
    double* const lds = reinterpret_cast<double*>(smem) + (size_t)wave_in_block * budget * W;
    const uint64_t cap = wb.shadow_cap;
    const uint32_t slot_end = (uint64_t)tile1 * kConnTile < n_slots ? tile1 * kConnTile : n_slots;
    const uint32_t n_waves = gridDim.x * n_waves_block;
    uint32_t chunk_next = 0, chunk_end = 0;     // this wave's piece of the shadow queue (the same in all lanes)

    for (uint32_t group = tile0 * (kConnTile / kVtxGroup) + blockIdx.x * n_waves_block + wave_in_block; (uint64_t)group * kVtxGroup < slot_end;
         group += n_waves) {
        const uint32_t slot = group * kVtxGroup + lane;
        uint32_t ne = 0, nl = 0;
        if (slot < slot_end) {
            const uint32_t cnt = wb.counts[slot];
            ne = cnt & 0xFFFu; nl = (cnt >> 12) & 0xFFFu;
            if (ne == 0u || nl == 0u) ne = nl = 0u;
        }
        const uint32_t recs = ne + nl, pairs = ne * nl;
        uint32_t rsum = recs, psum = pairs;          // inclusive running sums over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t r = __shfl_up(rsum, off, 64), p = __shfl_up(psum, off, 64);
            if ((int)lane >= off) { rsum += r; psum += p; }
        }
        uint32_t my_first = 0, my_count = 0;
        if (__shfl(psum, 63, 64) != 0u) {
            const double* const ws = wb.vtx + (uint64_t)group * ((uint64_t)kVtxGroup * wb.vtx_words) + lane;
            uint32_t s0 = 0;
            while (s0 < 64u) {
                // the slots [s0, s1) whose records fit the staging area together (one slot always does)
                const uint32_t r0 = s0 ? __shfl(rsum, (int)s0 - 1, 64) : 0u;
                const uint32_t s1 = s0 + (uint32_t)__builtin_popcountll(__ballot(lane >= s0 && rsum - r0 <= budget));
                const uint32_t p_before = s0 ? __shfl(psum, (int)s0 - 1, 64) : 0u;
                const bool staged = lane >= s0 && lane < s1;
                const uint32_t base = rsum - recs - r0;           // index of the slot's first record (eye vertices, then light vertices)
                if (__shfl(psum, (int)s1 - 1, 64) != p_before) {
                    uint32_t most = staged ? recs : 0u;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { const uint32_t v = __shfl_xor(most, off, 64); most = v > most ? v : most; }
                    for (uint32_t r = 0; r < most; r++) {
                        if (staged && r < recs) {
                            const uint32_t rec = r < ne ? r : depth + (r - ne);
                            const double* src = ws + (uint64_t)rec * W * kVtxGroup;
#pragma unroll
                            for (uint32_t f = 0; f < W; f++) lds[f * budget + base + r] = src[f * kVtxGroup];
                        }
                    }
                    // rounds: runs [q0, q1) of whole slots with at most 64 pairs together
                    uint32_t q0 = s0;
                    while (q0 < s1) {
                        const uint32_t pq = q0 ? __shfl(psum, (int)q0 - 1, 64) : 0u;
                        uint32_t q1 = q0 + (uint32_t)__builtin_popcountll(__ballot(lane >= q0 && lane < s1 && psum - pq <= 64u));
                        if (q1 == q0) q1 = q0 + 1u;                 // a slot with more than 64 pairs: rounds of its own
                        const uint32_t n_run = __shfl(psum, (int)q1 - 1, 64) - pq;
                        if (n_run) {
                            if (chunk_end - chunk_next < n_run) {   // the entries of a slot stay in one piece of the queue
                                for (uint32_t e = chunk_next + lane; e < chunk_end; e += 64u) wb.shadow[6 * cap + e] = -1.0;
                                uint32_t got = 0;
                                if (lane == 0) got = atomicAdd(shadow_count, chunk);
                                chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                                chunk_end = chunk_next + chunk;
                            }
                            // ends of the run's slots in run coordinates, for the owner search: 0 in front of the run, "never" behind it
                            const uint32_t ends = lane < q0 ? 0u : (lane >= q1 ? 0xFFFFFFFFu : psum - pq);
                            for (uint32_t first = 0; first < n_run; first += 64u) {
                                const uint32_t q = first + lane;
                                const bool active = q < n_run;
                                uint32_t owner = 0;             // number of lanes whose slot ends at or before q = the slot pair q belongs to
#pragma unroll
                                for (uint32_t step = 32; step; step >>= 1) {
                                    const uint32_t v = __shfl(ends, (int)(owner + step - 1u), 64);
                                    if (v <= q) owner += step;
                                }
                                owner = owner > 63u ? 63u : owner;
                                const uint32_t o_ne = __shfl(ne, (int)owner, 64), o_nl = __shfl(nl, (int)owner, 64);
                                const uint32_t o_base = __shfl(base, (int)owner, 64);
                                const uint32_t o_start = __shfl(psum - pairs, (int)owner, 64) - pq;
                                bool emit = false;
                                PairGeom g;
                                uint32_t a = 0, bi = 0, lv_tri = 0;
                                V3 kept = mk(0, 0, 0);
                                if (active) {
                                    const uint32_t r = q - o_start;
                                    a = r / (o_nl ? o_nl : 1u);
                                    bi = r - a * o_nl;
                                    const Vertex cv = staged_vertex<TEX>(lds, budget, o_base + a);
                                    // an eye vertex on an emitter has a zero BSDF (material.h:35-37): every pair it forms contributes
                                    // zero whatever the visibility, so no ray is spent on it
                                    if (!(cv.info & kVtxIsLight)) {
                                        const Vertex lv = staged_vertex<TEX>(lds, budget, o_base + o_ne + bi);
                                        lv_tri = lv.tri;
                                        const V3 f_cam = TEX ? cv.f : pmats.bsdf(sc, cv.info & kVtxMatMask);
                                        V3 f_light = mk(0, 0, 0);
                                        emit = !(len2(f_cam) <= 0) && pair_geometry(cv, lv, g);                                // camera.h:458-460
                                        if (emit) { f_light = TEX ? lv.f : pmats.light_factor(sc, lv.info); emit = len2(f_light) > 0; }      // camera.h:462-469
                                        if (OWN_TEST && emit) {
                                            GpuTri own;
                                            V3 to = g.origin, td = g.dir_unit;       // the ray as the triangle's own boxes see it
                                            if (INST) {
                                                const uint32_t x = own_xpos[lv.tri];
                                                own = own_tris[x];
                                                const int32_t ob = own_xobj[x];
                                                if (ob >= 0) cx_to_object(own_objs[ob], to, td);
                                            } else {
                                                own = own_tris[lv.tri];
                                                own.material = own_graze[lv.tri];
                                            }
                                            double t, u, v, det;
                                            if (tri_candidate(own, to, td, 0.001, t, u, v, det) && t <= g.max_t) {
                                                auto reaches = [&]() {
                                                    if (INST) return instanced_reaches(sc, lv.tri, g.origin, g.dir_unit, g.max_t);
                                                    return reference_reaches_inline(ref.nodes, ref.parents, sc.ref_nested != 0u, own_leaf[lv.tri], g.origin, g.dir_unit, g.max_t);
                                                };
                                                int ask = grazes(own, u, v, t, det, gbits) ? 2 : (nearly_equal(t, g.max_t) ? far_end_tie<true>(own, to, td, g.max_t) : 0);
#ifdef BDPT_TIE_CHECK
                                                if (!grazes(own, u, v, t, det, gbits)) {
                                                    const bool reached = reaches();
                                                    tie_check_record(nearly_equal(t, g.max_t), own, to, td, g.max_t, reached);
                                                    ask = reached ? 0 : 1;
                                                }
#endif
                                                if (ask == 2) ask = reaches() ? 0 : 1;
                                                if (ask == 0) emit = false;
                                            }
                                        }
                                        if (OWN_TEST && emit) kept = pair_contribution(cv, lv, g, f_cam, f_light);
                                    }
                                }
                                const unsigned long long writers = __ballot(emit);
                                if (emit) {
                                    const uint64_t e = chunk_next + rank_in_mask(writers);
                                    wb.shadow[e] = g.origin.x; wb.shadow[cap + e] = g.origin.y; wb.shadow[2 * cap + e] = g.origin.z;
                                    wb.shadow[3 * cap + e] = g.dir_unit.x; wb.shadow[4 * cap + e] = g.dir_unit.y; wb.shadow[5 * cap + e] = g.dir_unit.z;
                                    wb.shadow[6 * cap + e] = g.max_t;
                                    wb.sh_vis[e] = 0;
                                    if (OWN_TEST) { wb.contrib[e] = kept.x; wb.contrib[cap + e] = kept.y; wb.contrib[2 * cap + e] = kept.z; }
                                    else {
                                        wb.sh_pair[e] = (a << 12) | bi;
                                        if (wb.shadow_first) wb.shadow_first[e] = sc.tri_xpos[lv_tri];
                                    }
                                }
                                // the slots of the run count their entries of this round
                                if (lane >= q0 && lane < q1) {
                                    const uint32_t lo = psum - pairs - pq, hi = psum - pq;          // the slot's pairs in run coordinates
                                    const uint32_t lo_c = lo > first ? lo - first : 0u;
                                    const uint32_t hi_c = hi > first ? (hi - first < 64u ? hi - first : 64u) : 0u;
                                    if (hi_c > lo_c) {
                                        const unsigned long long below_hi = hi_c >= 64u ? ~0ull : ((1ull << hi_c) - 1ull);
                                        const unsigned long long below_lo = (1ull << lo_c) - 1ull;
                                        const uint32_t c = (uint32_t)__builtin_popcountll(writers & below_hi & ~below_lo);
                                        if (c && my_count == 0u) my_first = chunk_next + (uint32_t)__builtin_popcountll(writers & below_lo);
                                        my_count += c;
                                    }
                                }
                                chunk_next += (uint32_t)__builtin_popcountll(writers);
                            }
                        }
                        q0 = q1;
                    }
                }
                s0 = s1;
            }
        }
        if (slot < slot_end) { wb.ent_first[slot] = my_first; wb.ent_count[slot] = my_count; }
    }
    // the unused rest of the wave's last piece: entries the traversal skips (negative max_t)
    for (uint32_t e = chunk_next + lane; e < chunk_end; e += 64u) wb.shadow[6 * cap + e] = -1.0;
}

// Adds what the clear pairs of the pass's samples contribute, per sample in the order of its queue entries = the
// reference's pair order (camera.h:316-320, 471-474).  The entries carry the contributions (k_connect<., OWN_TEST>).
__global__ __launch_bounds__(kBlock) void k_conn_add_kept(WfBuffers wb, uint32_t n_slots, int pass) {
    const uint32_t tile0 = wb.pass_bounds[pass], tile1 = wb.pass_bounds[pass + 1];
    if (tile0 >= tile1) return;
    const uint32_t slot_end = (uint64_t)tile1 * kConnTile < n_slots ? tile1 * kConnTile : n_slots;
    const uint64_t ns = wb.ns, cap = wb.shadow_cap;
    for (uint32_t slot = tile0 * kConnTile + blockIdx.x * kBlock + threadIdx.x; slot < slot_end; slot += gridDim.x * kBlock) {
        const uint32_t cnt = wb.ent_count[slot];
        if (!cnt) continue;
        const uint32_t first = wb.ent_first[slot];
        V3 result = mk(0, 0, 0);
        bool changed = false;
        for (uint32_t e = first; e < first + cnt; e++) {
            if (!wb.sh_vis[e]) continue;
            // most samples have no clear pair: their radiance is not even read
            if (!changed) { result = mk(wb.result[slot], wb.result[ns + slot], wb.result[2 * ns + slot]); changed = true; }
            result = add(result, mk(wb.contrib[e], wb.contrib[cap + e], wb.contrib[2 * cap + e]));
        }
        if (changed) { wb.result[slot] = result.x; wb.result[ns + slot] = result.y; wb.result[2 * ns + slot] = result.z; }
    }
}

// The same for scenes whose clear pairs are evaluated here, one PAIR per lane.  Few samples have a clear pair, so with
// one sample per lane nearly every wave would run the whole pair evaluation for a handful of its lanes.  A wave numbers
// the clear entries of its 64 samples (prefix sum of the bit counts), lane l evaluates clear entry c0 + l whatever
// sample it belongs to (owner found by a search over the prefix sums, all through lane shuffles), and each sample's lane
// then collects its own pairs' values in entry order - the additions are the reference's, in the reference's order.
// A sample's entries are looked at 64 at a time.
template <bool TEX>
__global__ __launch_bounds__(kBlock) void k_conn_add_pairs(SceneDev sc, GpuCamera cam, WfBuffers wb, uint32_t n_slots, int pass) {
    const uint32_t tile0 = wb.pass_bounds[pass], tile1 = wb.pass_bounds[pass + 1];
    if (tile0 >= tile1) return;
    const uint32_t slot_end = (uint64_t)tile1 * kConnTile < n_slots ? tile1 * kConnTile : n_slots;
    const int depth = cam.max_depth;
    const uint64_t ns = wb.ns;
    const uint32_t lane = lane_id();
    for (uint32_t k0 = tile0 * kConnTile + blockIdx.x * kBlock + (threadIdx.x & ~63u); k0 < slot_end; k0 += gridDim.x * kBlock) {     // (the same for all lanes of a wave)
        const uint32_t slot = k0 + lane;
        uint32_t cnt = 0, first = 0;
        if (slot < slot_end) { cnt = wb.ent_count[slot]; first = wb.ent_first[slot]; }
        uint32_t most = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t v = __shfl_xor(most, off, 64); most = v > most ? v : most; }
        if (!most) continue;
        V3 result = mk(0, 0, 0);
        bool loaded = false, changed = false;
        for (uint32_t part = 0; part < most; part += 64u) {
            unsigned long long bits = 0ull;
            const uint32_t mine_n = cnt > part ? (cnt - part < 64u ? cnt - part : 64u) : 0u;
            for (uint32_t j = 0; j < mine_n; j++) bits |= (unsigned long long)(wb.sh_vis[first + part + j] != 0) << j;
            const uint32_t np = (uint32_t)__builtin_popcountll(bits);
            uint32_t incl = np;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if ((int)lane >= off) incl += v;
            }
            const uint32_t excl = incl - np;
            const uint32_t n_clear = __shfl(incl, 63, 64);
            if (n_clear == 0) continue;
            uint32_t max_np = np;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t v = __shfl_xor(max_np, off, 64); max_np = v > max_np ? v : max_np; }
            if (np && !loaded) { result = mk(wb.result[slot], wb.result[ns + slot], wb.result[2 * ns + slot]); loaded = true; }
            const uint32_t bits_lo = (uint32_t)bits, bits_hi = (uint32_t)(bits >> 32);
            const uint32_t part_first = first + part;
            for (uint32_t c0 = 0; c0 < n_clear; c0 += 64u) {
                const uint32_t q = c0 + lane;
                // the sample clear entry q belongs to: the first lane whose inclusive sum exceeds q
                uint32_t owner = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    const uint32_t v = __shfl(incl, (int)(owner + step - 1u), 64);
                    if (v <= q) owner += step;
                }
                owner = owner > 63u ? 63u : owner;
                const uint32_t o_excl = __shfl(excl, (int)owner, 64);
                const uint32_t o_slot = __shfl(slot, (int)owner, 64);
                const uint32_t o_first = __shfl(part_first, (int)owner, 64);
                unsigned long long o_bits = ((unsigned long long)__shfl(bits_hi, (int)owner, 64) << 32) | __shfl(bits_lo, (int)owner, 64);
                V3 c = mk(0, 0, 0);
                uint32_t valid = 0;
                if (q < n_clear) {
                    for (uint32_t r = q - o_excl; r > 0; r--) o_bits &= o_bits - 1ull;      // its (q - o_excl)-th clear entry
                    const uint32_t ab = wb.sh_pair[o_first + (uint32_t)__builtin_ctzll(o_bits)];
                    const VertexStore store = slot_store(wb, o_slot);
                    const Vertex cv = store.get<TEX>((int)(ab >> 12));
                    const Vertex lv = store.get<TEX>(depth + (int)(ab & 0xFFFu));
                    PairGeom g;
                    V3 f_cam, f_light;
                    bool ok = pair_geometry(cv, lv, g);
                    if (ok) {
                        if (TEX) {
                            f_cam = (cv.info & kVtxIsLight) ? mk(0, 0, 0) : cv.f;
                            f_light = lv.f;
                            ok = !(len2(f_cam) <= 0) && len2(f_light) > 0;
                        } else {
                            ok = pair_factors(sc, cv, lv, f_cam, f_light);
                        }
                    }
                    if (ok) { c = pair_contribution(cv, lv, g, f_cam, f_light); valid = 1u; }
                }
                // every sample's lane takes its own pairs of this round, in entry order
                for (uint32_t j = 0; j < max_np; j++) {
                    const uint32_t mine = excl + j;
                    const uint32_t src = (mine - c0) & 63u;
                    const double cx = __shfl(c.x, (int)src, 64), cy = __shfl(c.y, (int)src, 64), cz = __shfl(c.z, (int)src, 64);
                    const uint32_t v = __shfl(valid, (int)src, 64);
                    if (j < np && mine >= c0 && mine < c0 + 64u && v) { result = add(result, mk(cx, cy, cz)); changed = true; }
                }
            }
        }
        if (changed) { wb.result[slot] = result.x; wb.result[ns + slot] = result.y; wb.result[2 * ns + slot] = result.z; }
    }
}

// per-pixel sum over the samples in the reference's order (s_j outer, s_i inner, camera.h:118-123)
__global__ __launch_bounds__(kBlock) void k_resolve(GpuCamera cam, WfBuffers wb, WfBatch b, double* fb) {
    const int spp = cam.sqrt_spp * cam.sqrt_spp;
    const uint32_t n = (uint32_t)b.n_pixels * 3u;
    for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < n; t += gridDim.x * kBlock) {
        const int ch = (int)(t / (uint32_t)b.n_pixels);
        const int p = (int)(t - (uint32_t)ch * (uint32_t)b.n_pixels);
        const uint32_t q = (uint32_t)b.first_pixel + (uint32_t)p;
        if (q >= (uint32_t)b.n_owned) continue;
        const bool by_pixel = b.sample_stride > 1;
        const double* src = wb.result + (uint64_t)ch * wb.ns + (by_pixel ? (uint64_t)p * (uint64_t)spp : (uint64_t)p);
        const uint64_t step = by_pixel ? 1u : (uint64_t)b.pixels_stride;
        // the additions are sequential (the reference's order); the loads are not: sixteen in flight at a time
        double acc = 0.0;
        int s = 0;
        for (; s + 16 <= spp; s += 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = src[(uint64_t)(s + q) * step];
#pragma unroll
            for (int q = 0; q < 16; q++) acc = acc + v[q];
        }
        for (; s < spp; s++) acc = acc + src[(uint64_t)s * step];
        const uint32_t row = q / (uint32_t)b.row_w;
        const int j = b.rows[row];
        const int i = b.x0 + (int)(q - row * (uint32_t)b.row_w);
        fb[((size_t)j * cam.width + i) * 3 + ch] = acc;
    }
}

// ------------------------------------------------------------------ owned rows of a render call
BDPT_DEV int64_t owned_rows_before(int64_t y, int shard_rows, int shard_count, int shard_index) {
    const int64_t cycle = (int64_t)shard_rows * shard_count;
    int64_t part = y % cycle - (int64_t)shard_index * shard_rows;
    part = part < 0 ? 0 : (part > shard_rows ? shard_rows : part);
    return (y / cycle) * shard_rows + part;
}
__global__ __launch_bounds__(kBlock) void k_fill_rows(int32_t* rows, int y0, int y1, int shard_rows, int shard_count, int shard_index) {
    const int j = y0 + (int)(blockIdx.x * kBlock + threadIdx.x);
    if (j >= y1 || (j / shard_rows) % shard_count != shard_index) return;
    rows[owned_rows_before(j, shard_rows, shard_count, shard_index) - owned_rows_before(y0, shard_rows, shard_count, shard_index)] = j;
}

// ------------------------------------------------------------------ ray batches through the pipeline's traversal kernels
__global__ __launch_bounds__(kBlock) void k_rays_in(const double* rays, uint32_t n, WfBuffers wb, double* tmax_out, float* ray_len_out) {
    float longest = 0.0f;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const double* in = rays + 8 * (uint64_t)k;
#pragma unroll
        for (int c = 0; c < 6; c++) wb.ray[(uint64_t)c * wb.ns + k] = in[c];
        tmax_out[k] = in[7];
        const float l = __builtin_sqrtf((float)(in[3] * in[3] + in[4] * in[4] + in[5] * in[5])) * 1.0001f;
        longest = (l > longest || l != l) ? (l != l ? __builtin_inff() : l) : longest;      // a NaN direction: re-check every hit
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) longest = __builtin_fmaxf(longest, __shfl_down(longest, off, 64));
    // non-negative floats order like their bit patterns
    if (lane_id() == 0 && longest > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(ray_len_out), __float_as_uint(longest));
}

// hit_record of the closest hit (hittable.h:10-27): the traversal kernel reports the triangle and t; u, v are the
// same triangle's test evaluated once more (identical arithmetic, identical bits).  The traversal kernel's lower
// bound is the 0.001 of trace_path (camera.h:336); a ray with another tmin takes the reference's own walk here.
template <bool INSTANCED>
__global__ __launch_bounds__(kBlock) void k_hits_out(SceneDev sc, const double* rays, uint32_t n, WfBuffers wb, double* hits) {
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const double* in = rays + 8 * (uint64_t)k;
        const V3 o = mk(in[0], in[1], in[2]), d = mk(in[3], in[4], in[5]);
        HitLite h;
        h.tri = -1; h.t = 0; h.u = 0; h.v = 0;
        if (INSTANCED) {
            int32_t pos = (in[6] == 0.001) ? wb.hit_tri[k] : kHitRedo;      // (k_redo has resolved the traversal kernel's own marks)
            CxHit ch;
            ch.tri = -1; ch.obj = -1; ch.t = 0; ch.u = 0; ch.v = 0;
            if (pos == kHitRedo) {
                pos = redo_closest_instanced(sc, o, d, in[6], in[7], ch);
            } else if (pos >= 0) {
                ch.tri = sc.stris[pos].pad;
                ch.obj = sc.pos_obj[pos];
                V3 oo = o, od = d;
                if (ch.obj >= 0) cx_to_object(sc.objs[ch.obj], oo, od);
                double t, u, v;
                if (tri_test(sc.tris[ch.tri], oo, od, 0.001, kInf, t, u, v)) { ch.t = t; ch.u = u; ch.v = v; }
            }
            double* out = hits + 11 * (uint64_t)k;
            if (pos < 0) {
                out[0] = -1.0;
#pragma unroll
                for (int q = 1; q < 11; q++) out[q] = 0.0;
            } else {
                CxSurf sf;
                cx_surface(sc, o, d, ch, sf);
                out[0] = (double)ch.tri; out[1] = ch.t; out[2] = ch.u; out[3] = ch.v; out[4] = sf.front ? 1.0 : 0.0;
                out[5] = sf.p.x; out[6] = sf.p.y; out[7] = sf.p.z; out[8] = sf.n.x; out[9] = sf.n.y; out[10] = sf.n.z;
            }
            continue;
        }
        if (in[6] == 0.001) {
            const int32_t pos = wb.hit_tri[k];      // (exact ties already resolved by k_redo)
            if (pos >= 0) {
                h.tri = sc.stris[pos].pad;
                double t, u, v;
                if (tri_test(sc.tris[h.tri], o, d, 0.001, kInf, t, u, v)) { h.t = t; h.u = u; h.v = v; }
            }
        } else {
            GlobalScene mem{sc.nodes, sc.tris};
            if (!walk<kStatsOff, false>(mem, o, d, in[6], in[7], h, c)) h.tri = -1;
        }
        double* out = hits + 11 * (uint64_t)k;
        if (h.tri < 0) {
            out[0] = -1.0;
#pragma unroll
            for (int q = 1; q < 11; q++) out[q] = 0.0;
        } else {
            const V3 outward = ld3(sc.aux[h.tri].normal);
            const bool front = dot(d, outward) < 0;
            const V3 nrm = front ? outward : neg(outward);
            const V3 p = add(o, scl(h.t, d));
            out[0] = (double)h.tri; out[1] = h.t; out[2] = h.u; out[3] = h.v; out[4] = front ? 1.0 : 0.0;
            out[5] = p.x; out[6] = p.y; out[7] = p.z; out[8] = nrm.x; out[9] = nrm.y; out[10] = nrm.z;
        }
    }
}

// camera::visible (camera.h:425-436) up to the traversal: the shadow ray of the pair (a, b) as queue entry k
__global__ __launch_bounds__(kBlock) void k_pairs_in(const double* pairs, uint32_t n, WfBuffers wb) {
    const uint64_t cap = wb.shadow_cap;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const double* in = pairs + 6 * (uint64_t)k;
        const V3 a = mk(in[0], in[1], in[2]), b = mk(in[3], in[4], in[5]);
        wb.sh_vis[k] = 0;
        const V3 dir = sub(b, a);
        const double dist = len(dir);
        double max_t = -1.0;                    // a negative max_t marks an entry without a ray: "not visible"
        V3 origin = a, dir_unit = dir;
        if (dist > 0) {
            dir_unit = divs(dir, dist);
            origin = add(a, scl(0.001, dir_unit));
            if (dist - 0.001 > 0) max_t = dist - 0.001;
        }
        wb.shadow[k] = origin.x; wb.shadow[cap + k] = origin.y; wb.shadow[2 * cap + k] = origin.z;
        wb.shadow[3 * cap + k] = dir_unit.x; wb.shadow[4 * cap + k] = dir_unit.y; wb.shadow[5 * cap + k] = dir_unit.z;
        wb.shadow[6 * cap + k] = max_t;
    }
}
__global__ __launch_bounds__(kBlock) void k_visible_out(WfBuffers wb, uint32_t n, uint8_t* out) {
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) out[k] = wb.sh_vis[k];
}

}  // namespace

// ------------------------------------------------------------------------------ launchers
hipError_t wf_tie_counts(unsigned long long* out8, bool reset) {
#ifdef BDPT_TIE_CHECK
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tie_counts), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_tie_counts), zero, sizeof(zero));
    }
    return e;
#else
    (void)out8; (void)reset;
    return hipErrorNotSupported;
#endif
}

int64_t wf_owned_rows_before(int64_t y, int shard_rows, int shard_count, int shard_index) {
    const int64_t cycle = (int64_t)shard_rows * shard_count;
    int64_t part = y % cycle - (int64_t)shard_index * shard_rows;
    part = part < 0 ? 0 : (part > shard_rows ? shard_rows : part);
    return (y / cycle) * shard_rows + part;
}

hipError_t wf_fill_rows(int32_t* rows, int y0, int y1, int shard_rows, int shard_count, int shard_index, hipStream_t st) {
    if (y1 <= y0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((y1 - y0 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, y0, y1, shard_rows,
                       shard_count, shard_index);
    return hipGetLastError();
}

static unsigned batch_grid(uint32_t n) {
    const uint32_t b = (n + kBlock - 1) / kBlock;
    return b > 4096u ? 4096u : (b ? b : 1u);
}
hipError_t wf_rays_in(const double* rays, uint32_t n, const WfBuffers& wb, double* tmax_out, float* ray_len_out, hipStream_t st) {
    hipLaunchKernelGGL(k_rays_in, dim3(batch_grid(n)), dim3(kBlock), 0, st, rays, n, wb, tmax_out, ray_len_out);
    return hipGetLastError();
}
hipError_t wf_hits_out(const SceneDev& sc, const double* rays, uint32_t n, const WfBuffers& wb, double* hits, hipStream_t st) {
    if (sc.objs) hipLaunchKernelGGL(k_hits_out<true>, dim3(batch_grid(n)), dim3(kBlock), 0, st, sc, rays, n, wb, hits);
    else hipLaunchKernelGGL(k_hits_out<false>, dim3(batch_grid(n)), dim3(kBlock), 0, st, sc, rays, n, wb, hits);
    return hipGetLastError();
}
hipError_t wf_pairs_in(const double* pairs, uint32_t n, const WfBuffers& wb, hipStream_t st) {
    hipLaunchKernelGGL(k_pairs_in, dim3(batch_grid(n)), dim3(kBlock), 0, st, pairs, n, wb);
    return hipGetLastError();
}
hipError_t wf_visible_out(const WfBuffers& wb, uint32_t n, uint8_t* out, hipStream_t st) {
    hipLaunchKernelGGL(k_visible_out, dim3(batch_grid(n)), dim3(kBlock), 0, st, wb, n, out);
    return hipGetLastError();
}

// Sets the LDS limit of one traversal kernel (more than 64 KiB of dynamic LDS has to be requested) and reports
// how many of its workgroups fit a CU.
template <bool SHADOW, bool STATS, bool LDS_SCENE, bool INSTANCED = false>
static hipError_t prepare_traverse(size_t shm, int* blocks_per_cu) {
    auto fn = k_traverse<SHADOW, STATS, LDS_SCENE, INSTANCED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    if (blocks_per_cu) {
        int n = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kBlock, shm);
        *blocks_per_cu = n < 1 ? 1 : n;
    }
    return e;
}

static size_t scene_lds_bytes(const WfLaunchCfg& cfg) {
    return cfg.lds_scene ? cfg.lds_bytes : (size_t)cfg.top_nodes * sizeof(GpuQuadNode);
}
static size_t traverse_lds_bytes(const WfLaunchCfg& cfg) {
    // + the spare row of the stack; + the copy of the reference's tree of a scene staged in LDS (visibility rays)
    return scene_lds_bytes(cfg) + (size_t)(cfg.lds_stack + 1) * kBlock * sizeof(double) + (cfg.lds_scene ? cfg.ref_tree_bytes : 0);
}

hipError_t wf_prepare_traverse(const WfLaunchCfg& cfg, int* blocks_per_cu) {
    const size_t shm = traverse_lds_bytes(cfg);
    hipError_t e;
    if (cfg.conn_lds_bytes > ((size_t)64 << 10)) {      // more than 64 KiB of dynamic LDS has to be asked for
        const int bytes = (int)cfg.conn_lds_bytes;
        const void* fns[] = {reinterpret_cast<const void*>(k_connect<false, true, false>), reinterpret_cast<const void*>(k_connect<true, true, false>),
                             reinterpret_cast<const void*>(k_connect<false, true, false, true>), reinterpret_cast<const void*>(k_connect<true, true, false, true>),
                             reinterpret_cast<const void*>(k_connect<false, true, true, true>), reinterpret_cast<const void*>(k_connect<true, true, true, true>),
                             reinterpret_cast<const void*>(k_connect<false, false, true>), reinterpret_cast<const void*>(k_connect<true, false, true>),
                             reinterpret_cast<const void*>(k_connect<false, false, false>), reinterpret_cast<const void*>(k_connect<true, false, false>)};
        for (const void* fn : fns) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    }
    if (cfg.instanced && cfg.lds_scene) {
        e = prepare_traverse<false, false, true, true>(shm, blocks_per_cu);
        if (e == hipSuccess) e = prepare_traverse<false, true, true, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, false, true, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, true, true, true>(shm, nullptr);
    } else if (cfg.instanced) {
        e = prepare_traverse<false, false, false, true>(shm, blocks_per_cu);
        if (e == hipSuccess) e = prepare_traverse<false, true, false, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, false, false, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, true, false, true>(shm, nullptr);
    } else if (cfg.lds_scene) {
        e = prepare_traverse<false, false, true>(shm, blocks_per_cu);
        if (e == hipSuccess) e = prepare_traverse<false, true, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, false, true>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, true, true>(shm, nullptr);
    } else {
        e = prepare_traverse<false, false, false>(shm, blocks_per_cu);
        if (e == hipSuccess) e = prepare_traverse<false, true, false>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, false, false>(shm, nullptr);
        if (e == hipSuccess) e = prepare_traverse<true, true, false>(shm, nullptr);
    }
    return e;
}

hipError_t wf_gen_eye(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const WfLaunchCfg& cfg,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_gen_eye, dim3(cfg.stream_blocks), dim3(kBlock), 0, st, cam, wb, b);
    return hipGetLastError();
}

template <bool SHADOW, bool STATS, bool LDS_SCENE, bool INSTANCED = false>
static void launch_traverse_as(const SceneDev& sc, const WfBuffers& wb, const uint32_t* count_ptr,
                               uint32_t count_imm, uint32_t* cursor, const WfLaunchCfg& cfg, unsigned long long* stats,
                               hipStream_t st, uint32_t* redo_list, uint32_t* redo_count) {
    hipLaunchKernelGGL((k_traverse<SHADOW, STATS, LDS_SCENE, INSTANCED>), dim3(cfg.persistent_blocks), dim3(kBlock), traverse_lds_bytes(cfg), st,
                       sc, wb, count_ptr, count_imm, cursor, stats, cfg.refill_threshold, cfg.chunk, cfg.top_nodes,
                       cfg.leaf_threshold, cfg.leaf_inner, (uint32_t)scene_lds_bytes(cfg), cfg.lds_stack, redo_list, redo_count);
}

template <bool SHADOW>
static hipError_t launch_traverse(const SceneDev& sc, const WfBuffers& wb, const uint32_t* count_ptr,
                                  uint32_t count_imm, uint32_t* cursor, const WfLaunchCfg& cfg, unsigned long long* stats,
                                  hipStream_t st, uint32_t* redo_list = nullptr, uint32_t* redo_count = nullptr) {
    if (cfg.instanced && cfg.lds_scene) {
        if (stats) launch_traverse_as<SHADOW, true, true, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
        else launch_traverse_as<SHADOW, false, true, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
    } else if (cfg.instanced) {
        if (stats) launch_traverse_as<SHADOW, true, false, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
        else launch_traverse_as<SHADOW, false, false, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
    } else if (stats) {
        if (cfg.lds_scene) launch_traverse_as<SHADOW, true, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
        else launch_traverse_as<SHADOW, true, false>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
    } else {
        if (cfg.lds_scene) launch_traverse_as<SHADOW, false, true>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
        else launch_traverse_as<SHADOW, false, false>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
    }
    return hipGetLastError();
}

hipError_t wf_extend(const SceneDev& sc, const WfBuffers& wb, const uint32_t* count_ptr,
                     uint32_t count_imm, uint32_t* cursor, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st,
                     uint32_t* redo_list, uint32_t* redo_count) {
    return launch_traverse<false>(sc, wb, count_ptr, count_imm, cursor, cfg, stats, st, redo_list, redo_count);
}

hipError_t wf_redo(const SceneDev& sc, const WfBuffers& wb, const WfLaunchCfg& cfg, hipStream_t st, uint32_t* redo_list, uint32_t* redo_count) {
#ifdef BDPT_AB_SKIP_REDO
    return hipSuccess;      // timing experiment only (wrong pixels): what the frame would gain if k_redo cost nothing
#endif
    // few rays, if any: a small grid that returns at once when the list is empty
    if (cfg.instanced) hipLaunchKernelGGL(k_redo<true>, dim3(cfg.redo_blocks), dim3(kBlock), 0, st, sc, wb, redo_list, redo_count);
    else hipLaunchKernelGGL(k_redo<false>, dim3(cfg.redo_blocks), dim3(kBlock), 0, st, sc, wb, redo_list, redo_count);
    return hipGetLastError();
}

hipError_t wf_walk_ordered(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue,
                           const uint32_t* count_ptr, uint32_t count_imm, const WfLaunchCfg& cfg, hipStream_t st) {
    if (cfg.walk_lds_bytes) hipLaunchKernelGGL(k_walk_ordered<true>, dim3(cfg.stream_blocks), dim3(kBlock), cfg.walk_lds_bytes, st, sc, cam, wb, b, queue, count_ptr, count_imm);
    else hipLaunchKernelGGL(k_walk_ordered<false>, dim3(cfg.stream_blocks), dim3(kBlock), 0, st, sc, cam, wb, b, queue, count_ptr, count_imm);
    return hipGetLastError();
}

hipError_t wf_conn_ordered(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* conn_count,
                           const WfLaunchCfg& cfg, hipStream_t st) {
    if (cfg.walk_lds_bytes) hipLaunchKernelGGL(k_conn_ordered<true>, dim3(cfg.stream_blocks), dim3(kStageBlock), cfg.walk_lds_bytes, st, sc, cam, wb, b, conn_count);
    else hipLaunchKernelGGL(k_conn_ordered<false>, dim3(cfg.stream_blocks), dim3(kStageBlock), 0, st, sc, cam, wb, b, conn_count);
    return hipGetLastError();
}

hipError_t wf_occlude(const SceneDev& sc, const WfBuffers& wb, const uint32_t* shadow_count, uint32_t count_imm, uint32_t* cursor,
                      const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    return launch_traverse<true>(sc, wb, shadow_count, count_imm, cursor, cfg, stats, st);
}

hipError_t wf_tex_eval(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue_in,
                       const uint32_t* count_in_ptr, uint32_t count_in_imm, const WfLaunchCfg& cfg, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kStageBlock);
    if (cfg.ordered) hipLaunchKernelGGL((k_tex_eval<true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm);
    else hipLaunchKernelGGL((k_tex_eval<false>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm);
    return hipGetLastError();
}

hipError_t wf_shade(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, bool eye,
                    const uint32_t* queue_in, const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out,
                    uint32_t* count_out, bool last_bounce, bool store_vertices, bool keep_draw_counts, const WfLaunchCfg& cfg,
                    unsigned long long* stats, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kStageBlock);
    const int last = last_bounce ? 1 : 0, keep = store_vertices ? 1 : 0, draws = keep_draw_counts ? 1 : 0;
    if (cfg.ordered) {
        if (eye) hipLaunchKernelGGL((k_shade<true, false, true, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        else hipLaunchKernelGGL((k_shade<false, false, true, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        return hipGetLastError();
    }
    if (cfg.textured) {
        if (eye) {
            if (stats) hipLaunchKernelGGL((k_shade<true, true, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
            else hipLaunchKernelGGL((k_shade<true, false, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        } else {
            if (stats) hipLaunchKernelGGL((k_shade<false, true, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
            else hipLaunchKernelGGL((k_shade<false, false, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        }
        return hipGetLastError();
    }
    if (cfg.instanced) {
        if (eye) {
            if (stats) hipLaunchKernelGGL((k_shade<true, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
            else hipLaunchKernelGGL((k_shade<true, false, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        } else {
            if (stats) hipLaunchKernelGGL((k_shade<false, true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
            else hipLaunchKernelGGL((k_shade<false, false, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        }
        return hipGetLastError();
    }
    if (eye) {
        if (stats) hipLaunchKernelGGL((k_shade<true, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        else if (store_vertices) hipLaunchKernelGGL((k_shade<true, false, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        // no vertex is stored <=> the scene has no lambertian triangle: the variant without that material's code
        else hipLaunchKernelGGL((k_shade<true, false, false>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
    } else {
        if (stats) hipLaunchKernelGGL((k_shade<false, true, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
        else hipLaunchKernelGGL((k_shade<false, false, true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, last, keep, draws, cfg.sort_shift, stats);
    }
    return hipGetLastError();
}

hipError_t wf_shade_unidirectional(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, const uint32_t* queue_in,
                                   const uint32_t* count_in_ptr, uint32_t count_in_imm, uint32_t* queue_out, uint32_t* count_out, int level,
                                   bool last_level, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kStageBlock);
    if (stats) hipLaunchKernelGGL((k_shade_unidirectional<true>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, level, last_level ? 1 : 0, stats);
    else hipLaunchKernelGGL((k_shade_unidirectional<false>), grid, block, 0, st, sc, cam, wb, b, queue_in, count_in_ptr, count_in_imm, queue_out, count_out, level, last_level ? 1 : 0, stats);
    return hipGetLastError();
}

hipError_t wf_gen_light(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, uint32_t* queue_out,
                        uint32_t* count_out, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kStageBlock);
    if (cfg.textured) {
        if (stats) hipLaunchKernelGGL((k_gen_light<true, true>), grid, block, 0, st, sc, cam, wb, b, queue_out, count_out, stats);
        else hipLaunchKernelGGL((k_gen_light<false, true>), grid, block, 0, st, sc, cam, wb, b, queue_out, count_out, stats);
    } else if (stats) hipLaunchKernelGGL((k_gen_light<true>), grid, block, 0, st, sc, cam, wb, b, queue_out, count_out, stats);
    else hipLaunchKernelGGL((k_gen_light<false>), grid, block, 0, st, sc, cam, wb, b, queue_out, count_out, stats);
    return hipGetLastError();
}

hipError_t wf_conn_prepare(const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, uint32_t* conn_count,
                           const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kStageBlock);
    if (stats) hipLaunchKernelGGL((k_conn_prepare<true>), grid, block, 0, st, cam, wb, b, conn_count, stats);
    else hipLaunchKernelGGL((k_conn_prepare<false>), grid, block, 0, st, cam, wb, b, conn_count, stats);
    return hipGetLastError();
}

hipError_t wf_conn_count(const WfBuffers& wb, uint32_t n_slots, const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    dim3 grid(cfg.stream_blocks), block(kBlock);
    if (stats) hipLaunchKernelGGL((k_conn_count<true>), grid, block, 0, st, wb, n_slots, stats);
    else hipLaunchKernelGGL((k_conn_count<false>), grid, block, 0, st, wb, n_slots, stats);
    return hipGetLastError();
}

hipError_t wf_conn_plan(const WfBuffers& wb, uint32_t n_slots, const WfLaunchCfg& cfg, hipStream_t st) {
    hipLaunchKernelGGL(k_conn_plan, dim3(1), dim3(kPlanBlock), 0, st, wb, (n_slots + kConnTile - 1) / kConnTile, cfg.conn_pass_pairs, cfg.conn_passes);
    return hipGetLastError();
}

template <bool STATS, bool OWN_TEST, bool TEX, bool INST = false>
static void launch_connect(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, uint32_t n_slots, int pass, uint32_t* shadow_count,
                           const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    hipLaunchKernelGGL((k_connect<STATS, OWN_TEST, TEX, INST>), dim3(cfg.conn_blocks), dim3(64u * cfg.conn_waves), cfg.conn_lds_bytes, st, sc, cam, wb, n_slots,
                       pass, shadow_count, cfg.conn_rec_budget, cfg.conn_chunk, cfg.n_entries, stats);
}

hipError_t wf_connect(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, uint32_t n_slots, int pass, uint32_t* shadow_count,
                      const WfLaunchCfg& cfg, unsigned long long* stats, hipStream_t st) {
    if (cfg.own_test && cfg.instanced) {
        if (cfg.textured) {
            if (stats) launch_connect<true, true, true, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
            else launch_connect<false, true, true, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
        } else {
            if (stats) launch_connect<true, true, false, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
            else launch_connect<false, true, false, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
        }
    } else if (cfg.own_test) {
        if (stats) launch_connect<true, true, false>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
        else launch_connect<false, true, false>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
    } else if (cfg.textured) {
        if (stats) launch_connect<true, false, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
        else launch_connect<false, false, true>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
    } else {
        if (stats) launch_connect<true, false, false>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
        else launch_connect<false, false, false>(sc, cam, wb, n_slots, pass, shadow_count, cfg, stats, st);
    }
    return hipGetLastError();
}

hipError_t wf_conn_add(const SceneDev& sc, const GpuCamera& cam, const WfBuffers& wb, uint32_t n_slots, int pass, const WfLaunchCfg& cfg,
                       hipStream_t st) {
    if (cfg.own_test) hipLaunchKernelGGL(k_conn_add_kept, dim3(cfg.stream_blocks), dim3(kBlock), 0, st, wb, n_slots, pass);
    else if (cfg.textured) hipLaunchKernelGGL((k_conn_add_pairs<true>), dim3(cfg.stream_blocks), dim3(kBlock), 0, st, sc, cam, wb, n_slots, pass);
    else hipLaunchKernelGGL((k_conn_add_pairs<false>), dim3(cfg.stream_blocks), dim3(kBlock), 0, st, sc, cam, wb, n_slots, pass);
    return hipGetLastError();
}

hipError_t wf_resolve(const GpuCamera& cam, const WfBuffers& wb, const WfBatch& b, double* fb, const WfLaunchCfg& cfg,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3(cfg.stream_blocks), dim3(kBlock), 0, st, cam, wb, b, fb);
    return hipGetLastError();
}

}  // namespace bdpt
