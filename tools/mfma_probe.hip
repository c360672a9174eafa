// Tuning aid (not part of the product): peak issue rate of v_mfma_f32_32x32x2_f32 on gfx950 under the
// conditions of conv.hip — 4 independent accumulators per wave, 1..3 waves per SIMD, optionally with the
// kernel's LDS fragment reads in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREADS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0 * i;
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const float4* lp = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        float4 fa = make_float4(a, a, a, a), fb = make_float4(b, b, b, b), fc = fa, fd = fb;
        if (LDSREADS) { fa = lp[(it * 4) & 511]; fb = lp[(it * 4 + 64) & 511]; fc = lp[(it * 4 + 128) & 511]; fd = lp[(it * 4 + 192) & 511]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float x = s == 0 ? fa.x : s == 1 ? fa.y : s == 2 ? fa.z : fa.w;
            const float y = s == 0 ? fb.x : s == 1 ? fb.y : s == 2 ? fb.z : fb.w;
            const float z = s == 0 ? fc.x : s == 1 ? fc.y : s == 2 ? fc.z : fc.w;
            const float w = s == 0 ? fd.x : s == 1 ? fd.y : s == 2 ? fd.z : fd.w;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, y, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, w, acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int L>
void run(int blocks_per_cu, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<L>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<L>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1.0 * grid * 4 /*waves*/ * iters * 16 * 4096.0;
    printf("lds_reads=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", L, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 3 * 256);
    for (int b = 1; b <= 3; ++b) { run<0>(b, 20000, out); run<1>(b, 20000, out); }
    return 0;
}
