#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NVALU>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NVALU; ++q) v[q & 7] = v[q & 7] * 1.0001f + b;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int N>
void run(int blocks_per_cu, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<N>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<N>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1.0 * grid * 4 * iters * 16 * 4096.0;
    printf("valu_per_mfma=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", N, blocks_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 3 * 256);
    for (int b = 1; b <= 2; ++b) { run<0>(b, 10000, out); run<4>(b, 10000, out); run<8>(b, 10000, out); run<12>(b, 10000, out); run<16>(b, 10000, out); }
    return 0;
}
