// HIP kernels for gfx950 (wave64) and their launchers.
#include <hip/hip_runtime.h>

#include "device_core.h"
#include "kernels.h"

namespace bdpt {

constexpr int kBlock = 256;

// Cooperative copy of the node and triangle tables into LDS (16-byte words, coalesced).
__device__ __forceinline__ void stage_scene_to_lds(const SceneDev& sc, GpuNode* l_nodes, GpuTri* l_tris) {
    const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
    uint4* dst = reinterpret_cast<uint4*>(l_nodes);
    const uint32_t n16 = sc.n_nodes * (sizeof(GpuNode) / 16);
    for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
    src = reinterpret_cast<const uint4*>(sc.tris);
    dst = reinterpret_cast<uint4*>(l_tris);
    const uint32_t t16 = sc.n_tris * (sizeof(GpuTri) / 16);
    for (uint32_t k = threadIdx.x; k < t16; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}

__device__ __forceinline__ void flush_counters(const Counters& c, uint32_t draws, uint32_t samples,
                                               unsigned long long* stats) {
    // one atomic per counter per lane would be wasteful: reduce across the wave first
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

// Persistent workgroups; one work unit = `pixels_per_group` consecutive owned pixels.  A thread
// renders one sample; the per-pixel SUM is formed in the reference's order (s_j outer, s_i inner,
// camera.h:118-123) by one lane per colour channel reading the samples from LDS, so the FP64 sum
// is bit-identical to the sequential loop.
template <int STATS, bool LDS_SCENE, bool UNIDIRECTIONAL = false>
__global__ __launch_bounds__(kBlock) void render_kernel(SceneDev sc, GpuCamera cam, RenderLaunch rl) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* l_samples = reinterpret_cast<double*>(smem);                 // kBlock * 3
    double* l_carry = l_samples + kBlock * 3;                            // pixels_per_group * 3 (<= kBlock*3)
    __shared__ uint32_t l_unit;

    GpuNode* l_nodes = reinterpret_cast<GpuNode*>(smem + sizeof(double) * kBlock * 6);
    GpuTri* l_tris = reinterpret_cast<GpuTri*>(l_nodes + sc.n_nodes);
    if (LDS_SCENE) stage_scene_to_lds(sc, l_nodes, l_tris);

    const int spp = cam.sqrt_spp * cam.sqrt_spp;
    const int ppg = rl.pixels_per_group;
    const int row_w = rl.x1 - rl.x0;
    const long long n_owned = (long long)rl.n_rows * row_w;
    const int chunks = (spp + kBlock - 1) / kBlock;                      // > 1 only when ppg == 1

    VertexStore store;
    store.ws = rl.vertex_ws;
    store.stride = rl.ws_stride;
    store.g = (uint64_t)blockIdx.x * kBlock + threadIdx.x;

    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t total_draws = 0, total_samples = 0;

    for (;;) {
        __syncthreads();                                                 // l_unit / l_samples reuse
        if (threadIdx.x == 0) l_unit = atomicAdd(rl.unit_counter, 1u);
        __syncthreads();
        const uint32_t unit = l_unit;
        if (unit >= (uint32_t)rl.n_units) break;

        for (int chunk = 0; chunk < chunks; chunk++) {
            // which (pixel, sample) this thread renders
            int g, s;
            if (chunks == 1) { g = threadIdx.x / spp; s = threadIdx.x - g * spp; }
            else { g = 0; s = chunk * kBlock + threadIdx.x; }
            const long long q = (long long)unit * ppg + g;               // owned-pixel ordinal
            const bool active = (g < ppg) && (s < spp) && (q < n_owned);
            V3 col = mk(0, 0, 0);
            if (active) {
                const int j = rl.rows[q / row_w];
                const int i = rl.x0 + (int)(q % row_w);
                const int s_j = s / cam.sqrt_spp, s_i = s - s_j * cam.sqrt_spp;
                uint32_t draws = 0;
                if (LDS_SCENE) {
                    LdsScene mem{l_nodes, l_tris};
                    col = UNIDIRECTIONAL ? sample_radiance_unidirectional<STATS>(mem, sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws)
                                         : sample_radiance<STATS>(mem, sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws);
                } else {
                    GlobalScene mem{sc.nodes, sc.tris};
                    col = UNIDIRECTIONAL ? sample_radiance_unidirectional<STATS>(mem, sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws)
                                         : sample_radiance<STATS>(mem, sc, cam, i, j, s_i, s_j, rl.seed, store, c, draws);
                }
                total_draws += draws;
                total_samples++;
            }
            l_samples[threadIdx.x * 3 + 0] = col.x;
            l_samples[threadIdx.x * 3 + 1] = col.y;
            l_samples[threadIdx.x * 3 + 2] = col.z;
            __syncthreads();
            // ordered per-pixel accumulation: one lane per (pixel, channel) walks that pixel's samples in order
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
    if (STATS) flush_counters(c, total_draws, total_samples, rl.stats);
}

template <int STATS, bool LDS_SCENE>
__global__ __launch_bounds__(kBlock) void trace_closest_kernel(SceneDev sc, const double* rays, uint64_t n, double* hits,
                                                                unsigned long long* stats) {
    extern __shared__ __align__(16) unsigned char smem[];
    GpuNode* l_nodes = reinterpret_cast<GpuNode*>(smem);
    GpuTri* l_tris = reinterpret_cast<GpuTri*>(l_nodes + sc.n_nodes);
    if (LDS_SCENE) stage_scene_to_lds(sc, l_nodes, l_tris);
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kBlock) {
        const double* in = rays + 8 * r;
        V3 o = mk(in[0], in[1], in[2]), d = mk(in[3], in[4], in[5]);
        HitLite h;
        bool hit;
        if (LDS_SCENE) { LdsScene mem{l_nodes, l_tris}; hit = walk<STATS, false>(mem, o, d, in[6], in[7], h, c); }
        else { GlobalScene mem{sc.nodes, sc.tris}; hit = walk<STATS, false>(mem, o, d, in[6], in[7], h, c); }
        double* out = hits + 11 * r;
        if (!hit) {
            out[0] = -1.0;
#pragma unroll
            for (int k = 1; k < 11; k++) out[k] = 0.0;
        } else {
            V3 outward = ld3(sc.aux[h.tri].normal);
            bool front = dot(d, outward) < 0;
            V3 nrm = front ? outward : neg(outward);
            V3 p = add(o, scl(h.t, d));
            out[0] = (double)h.tri; out[1] = h.t; out[2] = h.u; out[3] = h.v; out[4] = front ? 1.0 : 0.0;
            out[5] = p.x; out[6] = p.y; out[7] = p.z; out[8] = nrm.x; out[9] = nrm.y; out[10] = nrm.z;
        }
    }
    if (STATS) flush_counters(c, 0, 0, stats);
}

__global__ __launch_bounds__(kBlock) void visible_kernel(SceneDev sc, const double* pairs, uint64_t n, uint8_t* out) {
    Counters c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    GlobalScene mem{sc.nodes, sc.tris};
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kBlock) {
        const double* in = pairs + 6 * r;
        out[r] = visible<kStatsOff>(mem, mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]), c) ? 1 : 0;
    }
}

__global__ void sincos_kernel(const double* x, uint64_t n, double* s, double* c) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double a, b; det_sincos(x[i], a, b); s[i] = a; c[i] = b; }
}

__global__ void random_kernel(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* out) {
    // every thread replays the stream up to its own index: exercises the counter semantics
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.key = stream_key(seed, pixel, sample);
    rng.counter = (uint32_t)i;
    out[i] = rng.next();
}

// IEEE building blocks the parity argument rests on: / and sqrt must be correctly rounded on the device
__global__ void arith_kernel(int op, const double* a, const double* b, uint64_t n, double* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
        case 0: r = a[i] / b[i]; break;
        case 1: r = __builtin_sqrt(a[i]); break;
        case 2: r = __builtin_fma(a[i], b[i], a[i]); break;
        case 3: r = a[i] * b[i] + a[i]; break;          // must NOT be contracted into an fma
        case 4: r = __builtin_fmin(a[i], b[i]); break;
        case 5: r = __builtin_fmax(a[i], b[i]); break;
        case 6: r = det_pow5(a[i]); break;
        case 8: r = det_log(a[i]); break;
        default: r = 1.0 / a[i]; break;
    }
    out[i] = r;
}

// colors_to_rgb8, src/image/wpng.h:14-35
__global__ void rgb8_kernel(const double* fb, uint64_t n_values, double scale, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    double c = fb[i] * scale;
    double cl = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
    double v = 256.0 * __builtin_sqrt(cl);
    out[i] = (v != v) ? 0 : (uint8_t)v;
}

// ------------------------------------------------------------------------------ launchers
static size_t scene_lds_bytes(const SceneDev& sc) {
    return (size_t)sc.n_nodes * sizeof(GpuNode) + (size_t)sc.n_tris * sizeof(GpuTri);
}

bool scene_fits_lds(const SceneDev& sc) { return scene_lds_bytes(sc) <= kSceneLdsLimit; }

// The render kernel keeps 12 KiB of per-sample colours in LDS beside the tables; together they stay within the
// 64 KiB a kernel may request without raising its dynamic LDS limit.
constexpr size_t kRenderSampleLds = sizeof(double) * kBlock * 6;
static bool render_scene_fits_lds(const SceneDev& sc) { return scene_lds_bytes(sc) + kRenderSampleLds <= kSceneLdsLimit; }

// Resident workgroups per CU of the render kernel variant a launch would pick (sizes the persistent grid).
template <int STATS, bool LDS>
static hipError_t occupancy_of(size_t shmem, int* n) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, render_kernel<STATS, LDS>, kBlock, shmem);
}

template <int STATS, bool LDS>
static hipError_t occupancy_of_unidirectional(size_t shmem, int* n) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, render_kernel<STATS, LDS, true>, kBlock, shmem);
}

hipError_t render_blocks_per_cu(const SceneDev& sc, int stats_mode, bool unidirectional, int* out) {
    const bool lds = render_scene_fits_lds(sc);
    const size_t shmem = kRenderSampleLds + (lds ? scene_lds_bytes(sc) : 0);
    int n = 0;
    hipError_t e;
    if (unidirectional) {
        if (stats_mode != kStatsOff) e = lds ? occupancy_of_unidirectional<kStatsReference, true>(shmem, &n) : occupancy_of_unidirectional<kStatsReference, false>(shmem, &n);
        else e = lds ? occupancy_of_unidirectional<kStatsOff, true>(shmem, &n) : occupancy_of_unidirectional<kStatsOff, false>(shmem, &n);
        *out = n < 1 ? 1 : n;
        return e;
    }
    if (stats_mode == kStatsReference) e = lds ? occupancy_of<kStatsReference, true>(shmem, &n) : occupancy_of<kStatsReference, false>(shmem, &n);
    else if (stats_mode == kStatsWork) e = lds ? occupancy_of<kStatsWork, true>(shmem, &n) : occupancy_of<kStatsWork, false>(shmem, &n);
    else e = lds ? occupancy_of<kStatsOff, true>(shmem, &n) : occupancy_of<kStatsOff, false>(shmem, &n);
    *out = n < 1 ? 1 : n;
    return e;
}

hipError_t launch_render(const SceneDev& sc, const GpuCamera& cam, const RenderLaunch& rl, int grid_blocks, int stats_mode,
                         bool unidirectional, hipStream_t stream) {
    const bool lds = render_scene_fits_lds(sc);
    size_t shmem = kRenderSampleLds + (lds ? scene_lds_bytes(sc) : 0);
    dim3 grid(grid_blocks), block(kBlock);
    if (unidirectional) {
        if (stats_mode != kStatsOff) {
            if (lds) hipLaunchKernelGGL((render_kernel<kStatsReference, true, true>), grid, block, shmem, stream, sc, cam, rl);
            else hipLaunchKernelGGL((render_kernel<kStatsReference, false, true>), grid, block, shmem, stream, sc, cam, rl);
        } else {
            if (lds) hipLaunchKernelGGL((render_kernel<kStatsOff, true, true>), grid, block, shmem, stream, sc, cam, rl);
            else hipLaunchKernelGGL((render_kernel<kStatsOff, false, true>), grid, block, shmem, stream, sc, cam, rl);
        }
        return hipGetLastError();
    }
    if (stats_mode == kStatsReference) {
        if (lds) hipLaunchKernelGGL((render_kernel<kStatsReference, true>), grid, block, shmem, stream, sc, cam, rl);
        else hipLaunchKernelGGL((render_kernel<kStatsReference, false>), grid, block, shmem, stream, sc, cam, rl);
    } else if (stats_mode == kStatsWork) {
        if (lds) hipLaunchKernelGGL((render_kernel<kStatsWork, true>), grid, block, shmem, stream, sc, cam, rl);
        else hipLaunchKernelGGL((render_kernel<kStatsWork, false>), grid, block, shmem, stream, sc, cam, rl);
    } else {
        if (lds) hipLaunchKernelGGL((render_kernel<kStatsOff, true>), grid, block, shmem, stream, sc, cam, rl);
        else hipLaunchKernelGGL((render_kernel<kStatsOff, false>), grid, block, shmem, stream, sc, cam, rl);
    }
    return hipGetLastError();
}

hipError_t launch_trace_closest(const SceneDev& sc, const double* d_rays, uint64_t n, double* d_hits,
                                unsigned long long* d_stats, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const bool lds = scene_fits_lds(sc);
    size_t shmem = lds ? scene_lds_bytes(sc) : 0;
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(kBlock);
    if (d_stats) {
        if (lds) hipLaunchKernelGGL((trace_closest_kernel<kStatsReference, true>), grid, block, shmem, stream, sc, d_rays, n, d_hits, d_stats);
        else hipLaunchKernelGGL((trace_closest_kernel<kStatsReference, false>), grid, block, shmem, stream, sc, d_rays, n, d_hits, d_stats);
    } else {
        if (lds) hipLaunchKernelGGL((trace_closest_kernel<kStatsOff, true>), grid, block, shmem, stream, sc, d_rays, n, d_hits, d_stats);
        else hipLaunchKernelGGL((trace_closest_kernel<kStatsOff, false>), grid, block, shmem, stream, sc, d_rays, n, d_hits, d_stats);
    }
    return hipGetLastError();
}

hipError_t launch_visible(const SceneDev& sc, const double* d_pairs, uint64_t n, uint8_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(visible_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, sc, d_pairs, n, d_out);
    return hipGetLastError();
}

hipError_t launch_sincos(const double* d_x, uint64_t n, double* d_s, double* d_c, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sincos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_x, n, d_s, d_c);
    return hipGetLastError();
}

hipError_t launch_random(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(random_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seed, pixel, sample, n, d_out);
    return hipGetLastError();
}

hipError_t launch_arith(int op, const double* d_a, const double* d_b, uint64_t n, double* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(arith_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, d_a, d_b, n, d_out);
    return hipGetLastError();
}

hipError_t launch_rgb8(const double* d_fb, uint64_t n_values, double scale, uint8_t* d_out, hipStream_t stream) {
    if (n_values == 0) return hipSuccess;
    hipLaunchKernelGGL(rgb8_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, stream, d_fb, n_values, scale, d_out);
    return hipGetLastError();
}

}  // namespace bdpt
