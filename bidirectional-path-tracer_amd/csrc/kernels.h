// Launchers of the HIP kernels (implemented in kernels.hip).
#ifndef BDPT_KERNELS_H
#define BDPT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bdpt_types.h"

namespace bdpt {

// Node + triangle tables up to this size are staged into LDS by every workgroup (160 KiB per CU;
// the limit leaves room for two resident workgroups of the render kernel).
constexpr size_t kSceneLdsLimit = 64 * 1024;

bool scene_fits_lds(const SceneDev& sc);

hipError_t render_blocks_per_cu(const SceneDev& sc, int stats_mode, bool unidirectional, int* out);
// stats_mode: 0 none, 1 reference-equivalent counters, 2 counters of the work really done
// unidirectional: camera::path_trace_color instead of bidirectional_color (counters then always have the reference's meaning)
hipError_t launch_render(const SceneDev& sc, const GpuCamera& cam, const RenderLaunch& rl, int grid_blocks, int stats_mode,
                         bool unidirectional, hipStream_t stream);
hipError_t launch_trace_closest(const SceneDev& sc, const double* d_rays, uint64_t n, double* d_hits,
                                unsigned long long* d_stats, hipStream_t stream);
hipError_t launch_visible(const SceneDev& sc, const double* d_pairs, uint64_t n, uint8_t* d_out, hipStream_t stream);
hipError_t launch_sincos(const double* d_x, uint64_t n, double* d_s, double* d_c, hipStream_t stream);
hipError_t launch_random(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* d_out, hipStream_t stream);
hipError_t launch_arith(int op, const double* d_a, const double* d_b, uint64_t n, double* d_out, hipStream_t stream);
hipError_t launch_rgb8(const double* d_fb, uint64_t n_values, double scale, uint8_t* d_out, hipStream_t stream);

}  // namespace bdpt
#endif
