// Statically allocated device workspace for callers that must not allocate (bdpt_render_device_static).
//
// The array below is part of the library's code object: every GPU the library runs on holds its own copy, at
// its own address, from the moment the code object is loaded there until the process ends.  In host code the
// array's NAME is only a stand-in for the symbol, not a device address; the address on a given GPU comes from
// hipGetSymbolAddress with that GPU current, and that pointer is what kernels, memsets and copies receive.
#include <hip/hip_runtime.h>

#include <mutex>

#include "static_scratch.h"

namespace bdpt {

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_lock;
void* g_ptr[kMaxDevices] = {};
bool g_known[kMaxDevices] = {};
// one user of a GPU's workspace at a time: the host side of a call holds g_busy[device], the device side of the next
// call waits for g_done[device], recorded at the end of the previous one
std::mutex g_busy[kMaxDevices];
hipEvent_t g_done[kMaxDevices] = {};
bool g_recorded[kMaxDevices] = {};
}  // namespace

hipError_t static_scratch_address(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_lock);
    if (!g_known[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;      // the caller makes `device` current first
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        e = hipEventCreateWithFlags(&g_done[device], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_ptr[device] = ptr;
        g_known[device] = true;
    }
    *out = g_ptr[device];
    return hipSuccess;
}

hipError_t static_scratch_begin(int device, hipStream_t stream) {
    if (device < 0 || device >= kMaxDevices || !g_known[device]) return hipErrorInvalidDevice;
    g_busy[device].lock();
    const hipError_t e = g_recorded[device] ? hipStreamWaitEvent(stream, g_done[device], 0) : hipSuccess;
    if (e != hipSuccess) g_busy[device].unlock();
    return e;
}

hipError_t static_scratch_end(int device, hipStream_t stream) {
    const hipError_t e = hipEventRecord(g_done[device], stream);
    if (e == hipSuccess) g_recorded[device] = true;
    g_busy[device].unlock();
    return e;
}

}  // namespace bdpt
