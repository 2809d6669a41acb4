// Per-GPU statically allocated workspace (see static_scratch.hip).
#ifndef BDPT_STATIC_SCRATCH_H
#define BDPT_STATIC_SCRATCH_H

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace bdpt {

// Size of the static workspace every GPU holds while the library is loaded.
constexpr size_t kStaticScratchBytes = (size_t)256 << 20;

// Address of the workspace on `device`, which must be the current device of the calling thread.  Looked up once
// per device and kept; allocates nothing.
hipError_t static_scratch_address(int device, void** out);

}  // namespace bdpt
#endif
