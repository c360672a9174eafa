// Shared host-side helpers for libmaskrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

// Tuning switches of the launch paths (tile overrides, kernel A/B routes): environment variables honoured ONLY in -DMRCNN_TUNING
// builds (`build.py --variant tuning -DMRCNN_TUNING`, which tools/ load through MRCNN_LIB); the product library ignores them —
// it reads three variables in all: MRCNN_CROP_STAGED, MRCNN_WINO_SPATIAL (documented modes) and MRCNN_CONV_NO_SPLIT (numerics A/B).
inline const char* tuning_env(const char* name) {
#ifdef MRCNN_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace mrcnn

// Schedule fuzzing (-DMRCNN_SYNC_FUZZ: an experiment build, `build.py --variant sync_fuzz -DMRCNN_SYNC_FUZZ`; never the product).
// Behind EVERY workgroup barrier each wave sleeps for a pseudo-random 0 .. 24 576 cycles. A kernel whose LDS hand-offs are
// ordered by its barriers and counted waits computes the same bits at any skew; one that is protected only by "the other
// waves cannot be that far ahead" (round 5's conv3x3_wino4_f32: ten MFMA slots stood in for a barrier) goes wrong on every
// launch instead of one in 5 000. tests/test_gpu_sync_fuzz.py runs the kernels' parity tests on such a build.
// MRCNN_SYNC_FUZZ_POINT() follows the hand-written barriers (asm / __builtin_amdgcn_s_barrier); __syncthreads() is wrapped.
#ifdef MRCNN_SYNC_FUZZ
namespace mrcnn {
__device__ __forceinline__ void sync_fuzz_point() {
    // ONE asm statement, three scalar registers, no control flow the compiler can see (the hand-scheduled kernels have no
    // vector register to spare, and a visible loop in their epilogues changes their register allocation): six bits of the
    // shader clock, a different bit window per SIMD (HW_ID.simd_id: the waves leave the barrier at the same instant), are
    // shifted out one by one with 4 096 cycles of sleep each — 0 .. 24 576 cycles, half of the waves more than 20 000.
    unsigned long long t;
    unsigned r;
    asm volatile("s_memtime %0\n\t"
                 "s_getreg_b32 %1, hwreg(HW_REG_HW_ID, 4, 2)\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_lshl_b32 %1, %1, 1\n\t"
                 "s_add_u32 %1, %1, 3\n\t"
                 "s_lshr_b64 %0, %0, %1\n\t"
                 "s_and_b64 %0, %0, 63\n\t"
                 "1:\n\t"
                 "s_cmp_eq_u64 %0, 0\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_sleep 64\n\t"
                 "s_lshr_b64 %0, %0, 1\n\t"
                 "s_branch 1b\n\t"
                 "2:"
                 : "=&s"(t), "=&s"(r)
                 :
                 : "scc", "memory");
}
__device__ __forceinline__ void fuzzed_syncthreads() {
    __syncthreads();
    sync_fuzz_point();
}
}  // namespace mrcnn
#define MRCNN_SYNC_FUZZ_POINT() ::mrcnn::sync_fuzz_point()
#define __syncthreads() ::mrcnn::fuzzed_syncthreads()
#else
#define MRCNN_SYNC_FUZZ_POINT() ((void)0)
#endif

#define MRCNN_REQUIRE(cond, ...)                                              \
    do {                                                                      \
        if (!(cond)) return ::mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)
