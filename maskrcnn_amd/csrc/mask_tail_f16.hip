// The tail of Mask.forward (model.py:906-914) in the plain-fp16 path (BASELINE configs[4]) as ONE launch, gfx950 only:
//     y = sigmoid(conv5(relu(deconv2x2_s2(x) + b_de)) + b_5)
// x fp16 NHWC [R][h][w][256] (the fourth 3x3 conv's output), deconv 256 -> 256 (kernel 2, stride 2: four independent 1x1
// convs, one per output sub-pixel (dy, dx)), conv5 1x1 256 -> classes (81), y fp32 NHWC [R][2h][2w][classes].
// The per-layer path writes the deconv's fp16 map (R * 4hw * 256: 160 MB at 400 RoIs) and reads it back for a 13 GFLOP GEMM whose
// 81 output channels fit no tile well: 0.109 + 0.075 ms for 40 MB in and 102 MB out. Here the deconv's output never leaves the
// registers: with the weight rows permuted as in bottleneck_f16.hip (mrcnn_pack_afrags_f16), two accumulators of
// v_mfma_f32_16x16x32_f16 are the lane's 8 consecutive channels = its B fragment of the next 1x1 conv — rounded to fp16 where
// the per-layer path rounds (its HBM tensor).
//
// Work item: (256 pixels, one sub-pixel). Eight waves, two 16-pixel tiles each (x fragments straight from global memory into
// B operands: 64 registers). The sub-pixel's deconv weights (256 x 256 fp16 = 128 KB) pass through ONE 64 KB LDS buffer as two
// halves of 128 output channels; the next half is fetched into registers (8 x 16 bytes per thread) while the current one is
// multiplied, and written to LDS between two barriers. Per half and 64-channel group: 32 A fragments x 2 pixel tiles of deconv
// MFMAs, bias + ReLU + v_cvt_pk_f16_f32 in registers, then 12 A fragments x 2 of conv5 (its weights, 96 rows = 81 + 15 zero,
// 48 KB, stay in LDS for the workgroup's life), accumulated over the four groups. Epilogue: bias, sigmoid (the expression of
// conv_common.hpp's epilogue), 32-byte runs per lane into the scattered output pixel.
// Persistent workgroups, one per CU (LDS 117 KB), items dealt round-robin (consecutive items = the four sub-pixels of one pixel
// set: its x is re-read from the L2).
#include <algorithm>

#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MT_T = 2;                       // pixel tiles per wave
constexpr int MT_PX = 8 * MT_T * 16;          // pixels per work item
constexpr int L_WDE = 0, WDE_BYTES = 64 * 1024;             // one half of a sub-pixel's deconv weights: [kc 8][j 8] fragments
constexpr int L_W5 = L_WDE + WDE_BYTES, W5_BYTES = 48 * 1024;   // [kc 8][cb 6] fragments
constexpr int L_BDE = L_W5 + W5_BYTES;        // deconv bias [4][256] fp32
constexpr int L_B5 = L_BDE + 4096;            // conv5 bias [96] fp32 (zero beyond `classes`)
constexpr int MT_LDS = L_B5 + 512;
constexpr unsigned OOB = 0xFFFFFFF0u;

struct MTParams {
    const _Float16* x;     // [M][256], M = rois * h * w
    const _Float16* wde;   // A fragments [8][64][64][8]
    const _Float16* w5;    // A fragments [8][6][64][8]
    const float* bde;      // [4 * 256]
    const float* b5;       // [classes]
    float* y;              // [rois][2h][2w][classes]
    int M, h, w, classes, sets, items;
    unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    MRCNN_SYNC_FUZZ_POINT();
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__global__ __launch_bounds__(512, 1) void mask_tail_f16(const MTParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, q = lane >> 4;

    // the staging registers of a weight half: thread i moves the 16 bytes i of each of the eight 8 KB pieces (kc = 0..7)
    u32x4 stage[8];
    auto fetch_half = [&](int sub, int half) {
#pragma unroll
        for (int kc = 0; kc < 8; ++kc)
            stage[kc] = *(reinterpret_cast<const u32x4*>(p.wde + static_cast<long long>((kc * 64 + sub * 16 + half * 8) * 64) * 8) + threadIdx.x);
    };
    auto write_half = [&]() {
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) reinterpret_cast<u32x4*>(smem + L_WDE + kc * 8192)[threadIdx.x] = stage[kc];
    };

    int item = blockIdx.x;
    if (item >= p.items) return;
    fetch_half(item & 3, 0);
    {   // resident: conv5's weights and the biases
        const u32x4* s5 = reinterpret_cast<const u32x4*>(p.w5);
        u32x4* d5 = reinterpret_cast<u32x4*>(smem + L_W5);
        for (int i = threadIdx.x; i < W5_BYTES / 16; i += 512) d5[i] = s5[i];
        float* bde = reinterpret_cast<float*>(smem + L_BDE);
        for (int i = threadIdx.x; i < 1024; i += 512) bde[i] = p.bde[i];
        float* b5 = reinterpret_cast<float*>(smem + L_B5);
        if (threadIdx.x < 96) b5[threadIdx.x] = threadIdx.x < p.classes ? p.b5[threadIdx.x] : 0.0f;
    }
    write_half();
    __syncthreads();

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int hw = p.h * p.w;

    for (; item < p.items; item += gridDim.x) {
        const int set = item >> 2, sub = item & 3, dy = sub >> 1, dx = sub & 1;
        // ---- the wave's two pixel tiles: x fragments, output rows ----------------------------------------------------
        f16x8 xf[MT_T][8];
        unsigned yoff[MT_T];
#pragma unroll
        for (int t = 0; t < MT_T; ++t) {
            const int m = set * MT_PX + (wave * MT_T + t) * 16 + l16;
            const bool ok = m < p.M;
            const unsigned xo = ok ? static_cast<unsigned>(m) * 512u + q * 16u : OOB;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc)
                xf[t][kc] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(xo), kc * 64, 0));
            const int mm = ok ? m : 0;
            const int r = mm / hw, rem = mm - r * hw, yy = rem / p.w, xx = rem - yy * p.w;
            const unsigned opx = static_cast<unsigned>((r * 2 * p.h + 2 * yy + dy) * 2 * p.w + 2 * xx + dx);
            yoff[t] = ok ? opx * static_cast<unsigned>(p.classes) * 4u : OOB;
        }
        f32x4 acc5[MT_T][6];
#pragma unroll
        for (int t = 0; t < MT_T; ++t)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc5[t][c] = zero4();

#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // the next weight half (of this item, or the first of the workgroup's next item) on its way into registers
            const int nitem = item + static_cast<int>(gridDim.x);
            if (half == 0) fetch_half(sub, 1);
            else if (nitem < p.items) fetch_half(nitem & 3, 0);
#pragma unroll
            for (int g = 0; g < 2; ++g) {   // 64 deconv channels: half*128 + g*64 ..
                f32x4 acc[MT_T][4];
#pragma unroll
                for (int t = 0; t < MT_T; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = zero4();
                {   // A fragments one k chunk AHEAD in a second register set, a scheduling barrier between reads and MFMAs: left to
                    // itself the compiler puts each ds_read one MFMA in front of its use and the wave waits out the LDS latency
                    // per fragment (two MFMAs): first version 0.108 ms for 56 GFLOP
                    f16x8 af[2][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[0][j] = *reinterpret_cast<const f16x8*>(smem + L_WDE + (g * 4 + j) * 1024 + lane * 16);
#pragma unroll
                    for (int kc = 0; kc < 8; ++kc) {
                        if (kc + 1 < 8) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                af[(kc + 1) & 1][j] = *reinterpret_cast<const f16x8*>(smem + L_WDE + ((kc + 1) * 8 + g * 4 + j) * 1024 + lane * 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int t = 0; t < MT_T; ++t)
                                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kc & 1][j], xf[t][kc], acc[t][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // bias + ReLU -> fp16: the lane's B fragments of conv5 for k chunks half*4 + g*2 + {0, 1}
                f16x8 d[MT_T][2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float* bp = reinterpret_cast<const float*>(smem + L_BDE) + sub * 256 + half * 128 + g * 64 + hh * 32 + q * 8;
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                    for (int t = 0; t < MT_T; ++t) {
                        f16x2 hp[4];
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            const f32x4& a = acc[t][2 * hh + (e2 >> 1)];
                            const f32x4& b = (e2 >> 1) ? b1 : b0;
                            f32x2 v = (e2 & 1) ? f32x2{a[2] + b[2], a[3] + b[3]} : f32x2{a[0] + b[0], a[1] + b[1]};
                            asm volatile("" : "+v"(v));
                            hp[e2] = __builtin_elementwise_max(__builtin_convertvector(v, f16x2), f16x2{0, 0});
                        }
                        d[t][hh] = __builtin_bit_cast(f16x8, u32x4{__builtin_bit_cast(unsigned, hp[0]), __builtin_bit_cast(unsigned, hp[1]),
                                                                   __builtin_bit_cast(unsigned, hp[2]), __builtin_bit_cast(unsigned, hp[3])});
                    }
                }
                {
                    f16x8 a5[2][6];
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        a5[0][c] = *reinterpret_cast<const f16x8*>(smem + L_W5 + ((half * 4 + g * 2) * 6 + c) * 1024 + lane * 16);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (hh == 0) {
#pragma unroll
                            for (int c = 0; c < 6; ++c)
                                a5[1][c] = *reinterpret_cast<const f16x8*>(smem + L_W5 + ((half * 4 + g * 2 + 1) * 6 + c) * 1024 + lane * 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c = 0; c < 6; ++c)
#pragma unroll
                            for (int t = 0; t < MT_T; ++t)
                                acc5[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a5[hh][c], d[t][hh], acc5[t][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            lds_barrier();        // every wave has finished reading this half
            if (half == 0 || nitem < p.items) write_half();
            lds_barrier();
        }

        // ---- epilogue: bias, sigmoid, 32-byte runs (8 consecutive classes per lane and 32-class group) ------------------------
#pragma unroll
        for (int t = 0; t < MT_T; ++t) {
#pragma unroll
            for (int g32 = 0; g32 < 3; ++g32) {
                const int ch0 = g32 * 32 + q * 8;
                const float* bp = reinterpret_cast<const float*>(smem + L_B5) + ch0;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float s = acc5[t][2 * g32 + (e >> 2)][e & 3] + bp[e];
                    v[e] = 1.0f / (1.0f + expf(-s));
                }
                const unsigned off = yoff[t] == OOB ? OOB : yoff[t] + static_cast<unsigned>(ch0) * 4u;
                if (ch0 + 8 <= p.classes) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), y_rsrc, static_cast<int>(off), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), y_rsrc, static_cast<int>(off), 16, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ch0 + e < p.classes)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), y_rsrc, static_cast<int>(off), e * 4, 0);
                }
            }
        }
    }
}

}  // namespace

extern "C" int mrcnn_mask_tail_f16_supported(int32_t rois, int32_t height, int32_t width, int32_t cin, int32_t cout_deconv,
                                             int32_t classes) {
    if (rois < 1 || height < 1 || width < 1 || cin != 256 || cout_deconv != 256 || classes < 1 || classes > 96) return 0;
    const long long m = static_cast<long long>(rois) * height * width;
    if (m * 512 >= (1LL << 31) || m * 4 * classes * 4 >= (1LL << 32) - 65536) return 0;   // 32-bit byte offsets
    return 1;
}

extern "C" int mrcnn_mask_tail_f16(const void* x_f16, int32_t rois, int32_t height, int32_t width, int32_t cin,
                                   const void* wde_frags, const float* bias_de4, int32_t cout_deconv, const void* w5_frags,
                                   const float* bias5, int32_t classes, float* y_f32, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_f16 && wde_frags && bias_de4 && w5_frags && bias5 && y_f32, "mask_tail_f16: null pointer");
    if (!mrcnn_mask_tail_f16_supported(rois, height, width, cin, cout_deconv, classes))
        return mrcnn::fail(MRCNN_ERR_UNSUPPORTED,
                           "mask_tail_f16: needs Cin 256, deconv Cout 256, classes <= 96, 32-bit byte offsets (got %d x %d x %d x %d "
                           "-> %d -> %d)", rois, height, width, cin, cout_deconv, classes);
    MTParams p{};
    p.x = static_cast<const _Float16*>(x_f16);
    p.wde = static_cast<const _Float16*>(wde_frags);
    p.w5 = static_cast<const _Float16*>(w5_frags);
    p.bde = bias_de4;
    p.b5 = bias5;
    p.y = y_f32;
    p.M = rois * height * width;
    p.h = height; p.w = width; p.classes = classes;
    p.sets = (p.M + MT_PX - 1) / MT_PX;
    p.items = p.sets * 4;
    p.x_bytes = static_cast<unsigned>(static_cast<long long>(p.M) * 512);
    p.y_bytes = static_cast<unsigned>(static_cast<long long>(p.M) * 4 * classes * 4);
    const int cus = mrcnn::device_cu_count() > 0 ? mrcnn::device_cu_count() : 256;
    const int grid = std::max(1, std::min(cus, p.items));
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(mask_tail_f16), MT_LDS, "mask_tail_f16")) return rc;
    hipLaunchKernelGGL(mask_tail_f16, dim3(grid), dim3(512), MT_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("mask_tail_f16");
}
