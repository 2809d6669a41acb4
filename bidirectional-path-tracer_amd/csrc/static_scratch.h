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

// Calls that use the workspace of one GPU must not overlap on it.  static_scratch_begin makes `stream` wait for the
// previous such call on that GPU (whatever context or stream it came from) and keeps other host threads out until
// static_scratch_end, which marks the end of this call's work on `stream`.  The event behind it is created with the
// address lookup (when a context is made); neither call allocates.  `device` must be current.
hipError_t static_scratch_begin(int device, hipStream_t stream);
hipError_t static_scratch_end(int device, hipStream_t stream);

}  // namespace bdpt
#endif
