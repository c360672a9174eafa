// Shared host-side helpers for libmaskrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "maskrcnn_hip.h"

namespace mrcnn {

// thread-local error message behind mrcnn_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(mrcnn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Check the launch that was just issued (no synchronisation).
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MRCNN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MRCNN_OK;
}

constexpr int kWave = 64;  // gfx950 wavefront

// Per-device launch state. A process may drive several GPUs (one torch device guard per call); nothing below may be
// cached per process. Both helpers are keyed by hipGetDevice() of the calling thread.
//   ensure_dynamic_lds: hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) — kernels that
//                       use more than 64 KB of dynamic LDS need it on EVERY device they are launched on;
//   device_cu_count:    multiProcessorCount of the current device (persistent-grid sizing), 0 on failure.
int ensure_dynamic_lds(const void* kernel, size_t lds_bytes, const char* who);
int device_cu_count();

}  // namespace mrcnn

#define MRCNN_REQUIRE(cond, ...)                                              \
    do {                                                                      \
        if (!(cond)) return ::mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)
