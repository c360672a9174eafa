// Whole-block fused Bottleneck (reference model.py:190-211) for the stride-1 identity blocks with planes = 64
// (ResNet C2 blocks 1..): ONE launch computes
//     out = relu( bn3(conv3_1x1( relu(bn2(conv2_3x3_same( relu(bn1(conv1_1x1(x))) ))) )) + x )
// with both planes-channel intermediates kept in LDS — x is read once (+ the 1-pixel halo of each tile, mostly L2 hits)
// and re-read once as the residual, out is written once; the two 64-channel maps (2 x 134 MB at batch 8, written
// AND read back by the three-launch path) never reach HBM. All arithmetic is fp32 on v_mfma_f32_32x32x2_f32; the
// summation orders, the Winograd F(2x2,3x3) transforms and the affine/ReLU expressions are those of conv.hip /
// conv_wino.hip, so the result equals the unfused path bit for bit.
//
//   workgroup   512 threads = 8 waves (two per SIMD), one per CU (150 KB of LDS), persistent over 16 x 16 output tiles.
//   phase 1     conv1 on the tile + 1-pixel halo (18 x 18 = 324 pixels, padded to 352 GEMM rows): a double-buffered
//               implicit GEMM M=352, N=64, K=Cin (k tiles of 32 through LDS); its epilogue writes T1 = relu(affine) into
//               an LDS image [18][18][64 (+4)], zero outside the picture (SAME padding pads conv2's INPUT).
//   phase 2     conv2 as Winograd F(2x2,3x3) straight out of T1: 64 tile positions x 64 channels x 16 components,
//               wave w owns components 2w, 2w+1 (128 accumulator registers). No staging of the transformed input:
//               an MFMA A-operand register is one position x one channel per lane, so every lane transforms its OWN
//               position from six 16-byte LDS reads of T1 (its wave's two components need 2 rows x 3 columns of the
//               4 x 4 patch): 10 packed adds per 16 MFMAs. The transformed filter streams through LDS (32 KB per
//               k tile of 8 channels, double-buffered) exactly as in conv_wino.hip.
//   A^T M A     four rounds of 16 positions through LDS (per-wave partial sums of M A, then the row combination),
//               affine + ReLU → T2 [256 pixels][64 (+4)] in LDS.
//   phase 3     conv3: GEMM M=256, N=256, K=64 with the A operand in LDS (T2) and the B fragments read from global
//               (64 KB of weights, L2-resident), wave w owns 32 pixels x 256 channels in two passes of 128;
//               epilogue relu(acc*s3 + t3 + x) → out, 128-byte channel runs.
#include "conv_common.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace mrcnn_conv;

struct BottleneckParams {
    const float* x;    // [B][H][W][Cin] NHWC
    const float* w1;   // [64][Cin]
    const float* s1;
    const float* t1;
    const float* u2;   // [64/8][16][64][8] (mrcnn_winograd_weights_f32)
    const float* s2;
    const float* t2;
    const float* w3;   // [256][64]
    const float* s3;
    const float* t3;
    float* y;          // [B][H][W][256]
    int B, H, W, Cin;
    int tiles_y, tiles_x, total;
    int debug;         // timing probes only (MRCNN_BNECK_DEBUG): bit 0/1/2/3 skips phase 1 / phase 2 / the A^T M A rounds / phase 3
    unsigned x_bytes, w1_bytes, u2_bytes, w3_bytes, y_bytes;
};

constexpr int P = 64;             // planes
constexpr int CO = 4 * P;         // block output channels (== Cin: identity residual)
constexpr int TS = 16;            // output tile side
constexpr int HS = TS + 2;        // halo tile side
constexpr int HP = HS * HS;       // 324 halo pixels
constexpr int M1 = 352;           // halo pixels padded to 11 MFMA row tiles
constexpr int BK = 32;            // phase-1 k tile
constexpr int LS = BK + 4;        // phase-1 LDS row stride (conflict-free ds_read_b128, as conv.hip)
constexpr int S = P + 4;          // floats per T1 / T2 pixel
constexpr int RP = HS * S + 8;    // floats per T1 row (the +8 keeps the patch reads at the 2-way conflict minimum)
constexpr int T1_FLOATS = HS * RP;                 // 22176
constexpr int UPLANE = 64 * 8;                     // floats per component plane of a U k tile
constexpr int U_FLOATS = 16 * UPLANE;              // 8192 per buffer
constexpr int T2_FLOATS = TS * TS * S;             // 17408
constexpr int Z_FLOATS = 16 * 4 * 64 * 4;          // 16 (wave, partial) planes x 4 position quads x 64 ch x 4 positions
constexpr size_t PH1_BYTES = sizeof(float) * 2 * (M1 + P) * LS;
constexpr size_t PH2_BYTES = sizeof(float) * (T1_FLOATS + 2 * U_FLOATS);
constexpr size_t PH3_BYTES = sizeof(float) * (T2_FLOATS + Z_FLOATS);
constexpr size_t FUSED_LDS = PH2_BYTES > PH1_BYTES ? (PH2_BYTES > PH3_BYTES ? PH2_BYTES : PH3_BYTES)
                                                   : (PH1_BYTES > PH3_BYTES ? PH1_BYTES : PH3_BYTES);
static_assert(FUSED_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ float4 f4(const u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 operator+(const float4 a, const float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 operator-(const float4 a, const float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
// a + sgn * b with sgn = +-1: one fma per element, bitwise a + b / a - b (the product is exact)
__device__ __forceinline__ float4 addsub(const float4 a, const float4 b, const float sgn) {
    return make_float4(fmaf(b.x, sgn, a.x), fmaf(b.y, sgn, a.y), fmaf(b.z, sgn, a.z), fmaf(b.w, sgn, a.w));
}
__device__ __forceinline__ float comp(const float4 v, const int s) {
    return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
}

__global__ __launch_bounds__(512, 1) void bottleneck_fused_f32(const BottleneckParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u2), 0, p.u2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, p.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // XCD-aware persistent tile walk: workgroup b (b % 8 = its XCD under round-robin placement) takes the virtual tiles
    // b, b + gridDim, ...; XCD x owns the contiguous tile range [x*per, (x+1)*per), so neighbouring tiles (shared halo
    // rows of x) meet in one L2.
    const int per_xcd = (p.total + 7) >> 3;
    const int tpi = p.tiles_y * p.tiles_x;
    // virtual tile → (image, tile origin); false past the end of this workgroup's XCD range (and then for every later one)
    auto decode = [&](int vt, int& b, int& y0, int& x0) -> bool {
        const int tile = (vt & 7) * per_xcd + (vt >> 3);
        if (vt >= 8 * per_xcd || tile >= p.total) return false;
        b = tile / tpi;
        const int trem = tile - b * tpi;
        const int tyi = trem / p.tiles_x;
        y0 = tyi * TS;
        x0 = (trem - tyi * p.tiles_x) * TS;
        return true;
    };
    float* As = smem;                  // [2][M1][LS]
    float* Bs = smem + 2 * M1 * LS;    // [2][P][LS]
    const int kq = tid & 7, r0 = tid >> 3;  // phase-1 staging: 16-byte slot kq of tile rows r0 + 64 i
    const unsigned b_voff = static_cast<unsigned>(r0 * p.Cin + kq * 4) * 4u;  // r0 < 64 = P
    const bool a5 = r0 < M1 - 320;  // rows 320 + r0 exist only up to 351 (wave-uniform: waves 0-3)
    unsigned a_voff[6];
    u32x4 ra[6], rb;
    auto setup1 = [&](int b, int y0, int x0) {  // halo pixel of each staged row → byte offset in x, or out of range
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int hp = r0 + 64 * i;
            const int hy = (hp * 57) >> 10, hx = hp - hy * HS;  // hp / 18 for hp < 512
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = hp < HP && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                            static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
            a_voff[i] = ok ? static_cast<unsigned>(((b * p.H + iy) * p.W + ix) * p.Cin + kq * 4) * 4u : OOB;
        }
    };
    auto load1 = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(a_voff[i]), kt * BK * 4, 0);
        rb = __builtin_amdgcn_raw_buffer_load_b128(w1_rsrc, static_cast<int>(b_voff), kt * BK * 4, 0);
    };
    auto store1 = [&](int buf) {
        float* a = As + buf * M1 * LS + r0 * LS + kq * 4;
#pragma unroll
        for (int i = 0; i < 5; ++i) *reinterpret_cast<u32x4*>(a + 64 * i * LS) = ra[i];
        if (a5) *reinterpret_cast<u32x4*>(a + 320 * LS) = ra[5];
        *reinterpret_cast<u32x4*>(Bs + buf * P * LS + r0 * LS + kq * 4) = rb;
    };

    // PREFETCH_NEXT: fetch the next tile's first phase-1 k tile behind this tile's phase 3 (measured: the 28 registers it
    // keeps live across the loop edge cost more in spills than the hidden latency returns — off)
    constexpr bool PREFETCH_NEXT = false;
    int vt = blockIdx.x, b, y0, x0;
    bool live = decode(vt, b, y0, x0);
    if (PREFETCH_NEXT && live) {
        setup1(b, y0, x0);
        load1(0);
    }
    while (live) {
        if (!PREFETCH_NEXT) setup1(b, y0, x0);
        // =================================================================================================
        // phase 1: T1 = relu(s1 * (X_halo W1^T) + t1), zero outside the picture
        // =================================================================================================
        // the first two k tiles of the transformed conv2 filter are fetched now and ride in registers through phase 1
        // (their LDS buffers alias the GEMM buffers): phase 2 starts without a global-load wait
        float* Us = smem + T1_FLOATS;  // [2][16][2 quads][64][4]
        unsigned u_voff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 512 * i;  // float4 id inside the 16 x 64 x 8 block, LDS order
            const int xi = id >> 7, r = id & 127, q = r >> 6, ch = r & 63;
            u_voff[i] = static_cast<unsigned>((xi * P + ch) * 8 + q * 4) * 4u;
        }
        u32x4 ru0[4], ru[4];
        auto load_u = [&](u32x4 (&dst)[4], int kt) {
            const int kk = kt < P / 8 ? kt : P / 8 - 1;  // past the end: reload, unused
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dst[i] = __builtin_amdgcn_raw_buffer_load_b128(u2_rsrc, static_cast<int>(u_voff[i]), kk * U_FLOATS * 4, 0);
        };
        auto store_u = [&](int buf, const u32x4 (&src)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Us + buf * U_FLOATS + (tid + 512 * i) * 4) = src[i];
        };
        const bool has2 = wave + 8 < M1 / 32;  // row tiles wave and wave + 8 (11 row tiles over 8 waves)
        f32x16 acc1[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[t][c][r] = 0.f;
        const int nk1 = p.Cin / BK;
        __syncthreads();  // the previous tile's phase 3 no longer reads T2
        if (!PREFETCH_NEXT) load1(0);
        store1(0);
        load_u(ru0, 0);
        load_u(ru, 1);
        __syncthreads();
        const int frag = ln * LS + lh * 4;
        // two copies of the loop (wave-uniform choice): waves 0-2 own two row tiles, the others one
        auto gemm1 = [&](auto two) {
            constexpr bool TWO = decltype(two)::value;
            for (int kt = 0; kt < ((p.debug & 1) ? 0 : nk1); ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nk1) load1(kt + 1);
                const float* Ab = As + buf * M1 * LS + frag + wave * 32 * LS;
                const float* Bb = Bs + buf * P * LS + frag;
                float4 fa[2][2], fb[2][2];  // [slot][tile]: the next chunk's fragments are fetched ahead of this chunk's MFMAs
                auto frags = [&](int slot, int j) {
                    fa[slot][0] = *reinterpret_cast<const float4*>(Ab + j * 8);
                    if constexpr (TWO) fa[slot][1] = *reinterpret_cast<const float4*>(Ab + 8 * 32 * LS + j * 8);
                    fb[slot][0] = *reinterpret_cast<const float4*>(Bb + j * 8);
                    fb[slot][1] = *reinterpret_cast<const float4*>(Bb + 32 * LS + j * 8);
                };
                frags(0, 0);
#pragma unroll
                for (int j = 0; j < BK / 8; ++j) {
                    const int cur = j & 1;
                    if (j + 1 < BK / 8) frags(cur ^ 1, j + 1);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[cur][0], s), comp(fb[cur][0], s), acc1[0][0], 0, 0, 0);
                        acc1[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[cur][0], s), comp(fb[cur][1], s), acc1[0][1], 0, 0, 0);
                        if constexpr (TWO) {
                            acc1[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[cur][1], s), comp(fb[cur][0], s), acc1[1][0], 0, 0, 0);
                            acc1[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[cur][1], s), comp(fb[cur][1], s), acc1[1][1], 0, 0, 0);
                        }
                    }
                }
                if (kt + 1 < nk1) store1(buf ^ 1);
                __syncthreads();
            }
        };
        if (has2) gemm1(std::true_type{});
        else gemm1(std::false_type{});
        // epilogue → T1 (aliases the GEMM buffers: the loop's last barrier is behind every read of them)
        float* T1 = smem;
        int lh_e = lh;  // an opaque copy: keeps the 64 row decodes below behind the main loop (hoisted, they spill)
        asm volatile("" : "+v"(lh_e));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !has2) continue;
            const int rt = t == 0 ? wave : wave + 8;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ch = c * 32 + ln;
                const float sc = p.s1 ? p.s1[ch] : 1.0f, sh = p.t1 ? p.t1[ch] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hp = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e;
                    const int hy = (hp * 57) >> 10, hx = hp - hy * HS;  // hp / 18 for hp < 512
                    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                    const bool in = static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                    static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                    float v = acc1[t][c][r] * sc + sh;
                    v = v > 0.f ? v : 0.f;
                    if (hp < HP) T1[hy * RP + hx * S + ch] = in ? v : 0.f;
                }
            }
        }

        // =================================================================================================
        // phase 2: M_xi = V_xi U_xi for the wave's two components, V computed per lane from T1
        // =================================================================================================
        store_u(0, ru0);   // k tile 0 (loaded at the top of the tile); ru = k tile 1 is already in flight / landed
        __syncthreads();  // T1 and U k tile 0 are complete

        // wave = (i, jh): components xi = 4 i + 2 jh + {0, 1}. B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:
        // transform row i combines patch rows (ra, rb) as ra + rs * rb
        const int wi = wave >> 1, jh = wave & 1;
        const int row_a = wi == 0 ? 0 : wi == 2 ? 2 : 1;
        const int row_b = wi == 0 ? 2 : wi == 1 ? 2 : wi == 2 ? 1 : 3;
        const float rs = wi == 1 ? 1.0f : -1.0f;
        int t1_base[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int pos = a * 32 + ln, py = pos >> 3, px = pos & 7;
            t1_base[a] = 2 * py * RP + (2 * px + jh) * S + 4 * lh;  // columns jh, jh+1, jh+2 of the 4 x 4 patch
        }
        const int offA = row_a * RP, offB = row_b * RP;
        f32x16 acc2[2][2][2];  // [component of the pair][position half][channel half]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[j][a][c][r] = 0.f;
        const float* Bw = Us + (wave * 2) * UPLANE + lh * (64 * 4) + ln * 4;
        for (int kt = 0; kt < ((p.debug & 2) ? 0 : P / 8); ++kt) {
            const int buf = kt & 1;
            float4 av[2][2];  // [component][position half]: four channels (this lane half's quad) of V
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* base = T1 + t1_base[a] + kt * 8;
                float4 l[3];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const float4 da = *reinterpret_cast<const float4*>(base + offA + cc * S);
                    const float4 db = *reinterpret_cast<const float4*>(base + offB + cc * S);
                    l[cc] = addsub(da, db, rs);
                }
                if (jh == 0) {  // columns 0,1,2: components j = 0: t0 - t2, j = 1: t1 + t2
                    av[0][a] = l[0] - l[2];
                    av[1][a] = l[1] + l[2];
                } else {        // columns 1,2,3: components j = 2: t2 - t1, j = 3: t1 - t3
                    av[0][a] = l[1] - l[0];
                    av[1][a] = l[0] - l[2];
                }
            }
            float4 fb[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    fb[j][c] = *reinterpret_cast<const float4*>(Bw + buf * U_FLOATS + j * UPLANE + c * 32 * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc2[j][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(av[j][a], s), comp(fb[j][c], s),
                                                                                 acc2[j][a][c], 0, 0, 0);
            store_u(buf ^ 1, ru);   // k tile kt+1 (its loads were issued one k tile ago)
            load_u(ru, kt + 2);
            __syncthreads();
        }

        // =================================================================================================
        // A^T M A → T2 = relu(s2 * conv2 + t2): four rounds of 16 positions through LDS
        //   (M A)_i0 = M_i0 + M_i1 + M_i2,  (M A)_i1 = M_i1 - M_i2 - M_i3: jh = 0 contributes (M_i0 + M_i1, M_i1),
        //                                                                   jh = 1 contributes (M_i2, -M_i2 - M_i3)
        // =================================================================================================
        float* T2 = smem;               // [256][S]
        float* Z = smem + T2_FLOATS;    // [wave][2 partials][4 position quads][64 channels][4 positions]
        constexpr int ZQ = 4 * 64 * 4;
        {
            const int chn = tid & 63, pql = (tid >> 6) & 3, cc = tid >> 8;
            const float sc = p.s2 ? p.s2[chn] : 1.0f, sh = p.t2 ? p.t2[chn] : 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (p.debug & 4) continue;
                const int a = g >> 1, gq = g & 1;
                if (g) __syncthreads();  // the previous round's Z is read out
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        float4 zp, zq;
                        float* zpp = &zp.x;
                        float* zqp = &zq.x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = (2 * gq + q2) * 4 + e;  // rows 16 gq + 4 (2 q2 + lh) + e of position half a
                            const float ma = acc2[0][a][c][r], mb = acc2[1][a][c][r];
                            zpp[e] = jh == 0 ? ma + mb : ma;
                            zqp[e] = jh == 0 ? mb : -ma - mb;
                        }
                        const int o = ((2 * q2 + lh) * 64 + c * 32 + ln) * 4;
                        *reinterpret_cast<float4*>(Z + (wave * 2 + 0) * ZQ + o) = zp;
                        *reinterpret_cast<float4*>(Z + (wave * 2 + 1) * ZQ + o) = zq;
                    }
                __syncthreads();
                // thread = (channel, position quad, output column cc): the partials of column cc, rows i = 0..3
                float4 yi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 p0 = *reinterpret_cast<const float4*>(Z + ((2 * i) * 2 + cc) * ZQ + (pql * 64 + chn) * 4);
                    const float4 p1 = *reinterpret_cast<const float4*>(Z + ((2 * i + 1) * 2 + cc) * ZQ + (pql * 64 + chn) * 4);
                    yi[i] = p0 + p1;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pos = a * 32 + gq * 16 + pql * 4 + e, py = pos >> 3, px = pos & 7;
                    const float z0 = comp(yi[0], e), z1 = comp(yi[1], e), z2 = comp(yi[2], e), z3 = comp(yi[3], e);
                    const float yv[2] = {z0 + z1 + z2, z1 - z2 - z3};
#pragma unroll
                    for (int ar = 0; ar < 2; ++ar) {
                        float v = yv[ar] * sc + sh;
                        v = v > 0.f ? v : 0.f;
                        T2[((2 * py + ar) * TS + 2 * px + cc) * S + chn] = v;
                    }
                }
            }
        }
        __syncthreads();  // T2 complete

        // =================================================================================================
        // phase 3: out = relu(s3 * (T2 W3^T) + t3 + x); wave w: pixels 32 w .. 32 w + 31, all 256 channels
        // =================================================================================================
        const float* A3 = T2 + (wave * 32 + ln) * S + lh * 4;
        const unsigned w3_lane = static_cast<unsigned>(ln * P + lh * 4) * 4u;
        unsigned rowoff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            rowoff[r] = static_cast<unsigned>((b * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15)) * (CO * 4u);
        }
        // two passes of 128 output channels (64 accumulator registers, B fragments of one k chunk in flight)
#pragma unroll 1
        for (int nh = 0; nh < ((p.debug & 8) ? 0 : 2); ++nh) {
            f32x16 acc3[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[ct][r] = 0.f;
            float res[4][16];  // the residual x of this pass: 64 loads in flight behind the MFMA loop
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[ct][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        x_rsrc, static_cast<int>(rowoff[r] + (nh * 128 + ct * 32 + ln) * 4u), 0, 0));
            const int w3_half = nh * 128 * P * 4;  // byte offset of this half's first weight row
            u32x4 wb[2][4];
            auto load_w3 = [&](int slot, int j) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    wb[slot][ct] = __builtin_amdgcn_raw_buffer_load_b128(w3_rsrc, static_cast<int>(w3_lane),
                                                                         w3_half + (ct * 32 * P + j * 8) * 4, 0);
            };
            load_w3(0, 0);
#pragma unroll
            for (int j = 0; j < P / 8; ++j) {
                const int cur = j & 1;
                if (j + 1 < P / 8) load_w3(cur ^ 1, j + 1);
                const float4 fa = *reinterpret_cast<const float4*>(A3 + j * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc3[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa, s), comp(f4(wb[cur][ct]), s), acc3[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PREFETCH_NEXT && nh == 1) {
                // the NEXT tile's first phase-1 k tile, issued ahead of this pass's stores: vmcnt counts in issue order,
                // so behind 64 stores the next tile's first LDS write would wait for every one of them
                vt += gridDim.x;
                live = decode(vt, b, y0, x0);
                if (live) {
                    setup1(b, y0, x0);
                    load1(0);
                }
            }
            // epilogue: the residual (fetched ahead of the MFMA loop), affine, ReLU, 128-byte channel runs
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = nh * 128 + ct * 32 + ln;
                const float sc = p.s3 ? p.s3[n] : 1.0f, sh = p.t3 ? p.t3[n] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc3[ct][r] * sc + sh;
                    v += res[ct][r];
                    v = v > 0.f ? v : 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(rowoff[r] + n * 4u), 0, 0);
                }
            }
        }
        if (!PREFETCH_NEXT || (p.debug & 8)) {
            vt += gridDim.x;
            live = decode(vt, b, y0, x0);
            if (PREFETCH_NEXT && live) {
                setup1(b, y0, x0);
                load1(0);
            }
        }
    }  // tiles
}

}  // namespace

extern "C" int mrcnn_bottleneck_fused_supported(int32_t height, int32_t width, int32_t cin, int32_t planes) {
    return (planes == P && cin == CO && height >= TS && width >= TS && height % TS == 0 && width % TS == 0) ? 1 : 0;
}

extern "C" int mrcnn_bottleneck_fused_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                          const float* w1, const float* scale1, const float* shift1, const float* u2,
                                          const float* scale2, const float* shift2, const float* w3,
                                          const float* scale3, const float* shift3, int32_t planes, float* y,
                                          mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && w1 && u2 && w3 && y, "bottleneck_fused: null pointer");
    MRCNN_REQUIRE(batch >= 1 && mrcnn_bottleneck_fused_supported(height, width, cin, planes),
                  "bottleneck_fused: this kernel covers the stride-1 identity blocks with planes = 64, Cin = 256 and H, W "
                  "multiples of 16 (got B=%d H=%d W=%d Cin=%d planes=%d); use the per-layer path otherwise",
                  batch, height, width, cin, planes);
    MRCNN_REQUIRE(x != y, "bottleneck_fused: in-place operation is not supported (halo reads)");
    const long long px = 1LL * batch * height * width;
    MRCNN_REQUIRE(px * CO < (1LL << 30), "bottleneck_fused: tensor too large (32-bit buffer byte offsets)");
    BottleneckParams p;
    p.x = x; p.w1 = w1; p.s1 = scale1; p.t1 = shift1; p.u2 = u2; p.s2 = scale2; p.t2 = shift2;
    p.w3 = w3; p.s3 = scale3; p.t3 = shift3; p.y = y;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin;
    p.tiles_y = height / TS; p.tiles_x = width / TS; p.total = batch * p.tiles_y * p.tiles_x;
    static const int debug = mrcnn::tuning_env("MRCNN_BNECK_DEBUG") ? atoi(mrcnn::tuning_env("MRCNN_BNECK_DEBUG")) : 0;
    p.debug = debug;
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.y_bytes = static_cast<unsigned>(4LL * px * CO);
    p.w1_bytes = static_cast<unsigned>(4LL * P * cin);
    p.u2_bytes = static_cast<unsigned>(4LL * 16 * P * P);
    p.w3_bytes = static_cast<unsigned>(4LL * CO * P);
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_fused_f32), FUSED_LDS, "bottleneck_fused"))
        return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "bottleneck_fused: cannot query the device");
    const int ncu = cus >= 8 ? (cus / 8) * 8 : 8;
    const int virt = 8 * ((p.total + 7) / 8);
    const int grid = virt < ncu ? virt : ncu;
    hipLaunchKernelGGL(bottleneck_fused_f32, dim3(grid), dim3(512), FUSED_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("bottleneck_fused_f32");
}
