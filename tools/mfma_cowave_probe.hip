// Does VALU work of ONE wave issue under the fp32 MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// The round-3 review proposed an F(4x4) Winograd tile with two waves per SIMD "so one's 48 packed transform ops issue under the
// other's MFMAs". tools/mfma_valu_probe.hip showed that VALU work in the SAME instruction stream adds its full issue time to
// v_mfma_f32_32x32x2_f32; this probe separates the two streams: a 512-thread workgroup per CU (waves w and w + 4 share SIMD w),
// waves 0-3 run a pure MFMA loop, waves 4-7 a pure loop of one kind of other instruction. Three launches per kind: MFMA waves
// alone, the other waves alone, both together. together ~= max(alone) -> the two co-issue; together ~= sum -> they share a pipe.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cowave tools/mfma_cowave_probe.hip && /tmp/cowave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { PK_FMA = 0, PK_ADD = 1, FMA = 2, ADD_U32 = 3, DS_READ = 4, PK_FMA_INDEP = 5, MFMA_16 = 6, NKINDS = 7 };
static const char* kind_name[] = {"v_pk_fma_f32 (4 chains)", "v_pk_add_f32 (4 chains)", "v_fma_f32 (8 chains)", "v_add_u32 (8 chains)",
                                  "ds_read_b64", "v_pk_fma_f32 (16 chains)", "v_mfma_f32_32x32x2_f32 (second MFMA wave)"};

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int mfma_iters, int other_iters, int run_mfma, int run_other) {
    extern __shared__ float lds[];   // 100 KB requested: one workgroup per CU
    const int wave = threadIdx.x >> 6;
    float result = 0.f;
    if (wave < 4) {
        if (run_mfma) {
            f32x16 acc[4];
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
            float a = 1.0f + threadIdx.x, b = 2.0f - threadIdx.x;
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s & 3], 0, 0, 0);
            }
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) result += acc[k][r];
        }
    } else if (run_other) {
        if (KIND == PK_FMA || KIND == PK_ADD || KIND == PK_FMA_INDEP) {
            constexpr int NC = KIND == PK_FMA_INDEP ? 16 : 4;
            f32x2 v[NC];
            for (int i = 0; i < NC; ++i) v[i] = f32x2{1.0f + i, 2.0f + threadIdx.x};
            f32x2 m = f32x2{1.0001f, 0.9999f}, c = f32x2{0.5f, 0.25f};
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[s % NC]) : "v"(c));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[s % NC]) : "v"(m), "v"(c));
                }
            }
            for (int i = 0; i < NC; ++i) result += v[i].x + v[i].y;
        } else if (KIND == FMA) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = 1.0f + i + threadIdx.x;
            float m = 1.0001f, c = 0.5f;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[s & 7]) : "v"(m), "v"(c));
            }
            for (int i = 0; i < 8; ++i) result += v[i];
        } else if (KIND == ADD_U32) {
            unsigned v[8];
            for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
            unsigned c = 3;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[s & 7]) : "v"(c));
            }
            for (int i = 0; i < 8; ++i) result += v[i];
        } else if (KIND == DS_READ) {
            f32x2 acc = f32x2{0.f, 0.f};
            const unsigned addr = (threadIdx.x & 255) * 8;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    f32x2 t;
                    asm volatile("ds_read_b64 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr), "n"((s & 15) * 2048));
                    acc += t;
                }
            }
            result = acc.x + acc.y;
        } else if (KIND == MFMA_16) {
            f32x16 acc[4];
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
            float a = 1.0f + threadIdx.x, b = 2.0f - threadIdx.x;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s & 3], 0, 0, 0);
            }
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) result += acc[k][r];
        }
    }
    if (threadIdx.x == 0) lds[0] = 0.f;
    out[blockIdx.x * 512 + threadIdx.x] = result;
}

template <int KIND>
float time_one(float* out, int mi, int oi, int rm, int ro) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 100 * 1024, 0, out, mi, oi, rm, ro);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

template <int KIND>
void run(float* out) {
    const int mi = 4000;                       // 64 000 MFMAs per wave = 4.1 M cycles at 64 cycles each
    int oi = 4000;
    float other = time_one<KIND>(out, mi, oi, 0, 1);
    const float mfma = time_one<KIND>(out, mi, oi, 1, 0);
    oi = static_cast<int>(oi * mfma / other * 0.8f);   // the other waves alone take ~0.8 of the MFMA waves' time
    other = time_one<KIND>(out, mi, oi, 0, 1);
    const float both = time_one<KIND>(out, mi, oi, 1, 1);
    const int per = KIND == MFMA_16 ? 16 : 32;
    printf("{\"other\": \"%s\", \"mfma_alone_ms\": %.3f, \"other_alone_ms\": %.3f, \"together_ms\": %.3f, \"sum_ms\": %.3f, "
           "\"overlap\": %.3f, \"other_instr_per_mfma\": %.2f, \"other_cycles_each_alone\": %.1f}\n",
           kind_name[KIND], mfma, other, both, mfma + other, (mfma + other - both) / (other < mfma ? other : mfma),
           1.0 * oi * per / (mi * 16.0), other * 1e-3 * 2.4e9 / (1.0 * oi * per));
}

int main() {
    float* out; hipMalloc(&out, sizeof(float) * 512 * 256);
    run<PK_FMA>(out); run<PK_ADD>(out); run<PK_FMA_INDEP>(out); run<FMA>(out); run<ADD_U32>(out); run<DS_READ>(out); run<MFMA_16>(out);
    return 0;
}
