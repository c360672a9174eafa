// Pipelined fp16 implicit-GEMM convolution for gfx950 (BASELINE configs[4]'s "fp16 MFMA path", the large layers):
// stride-1 1x1 / 3x3 (any KH x KW up to 25 taps) conv + affine + fp16 residual + ReLU, fp16 NHWC in, fp16 and / or fp32 NHWC out,
// v_mfma_f32_16x16x32_f16 with fp32 accumulation. Same arithmetic per output element as conv_f16.hip's PRODUCTS = 1 path
// (the k order of the accumulation chain differs in the grouping of 8 -> bitwise equality is not promised, the tolerance is).
//
// What conv_f16.hip cannot do at one workgroup per CU (its 128x128 tile with a barrier every 8 MFMAs runs the 3x3 layers at
// 600-770 TFLOP/s): keep global -> LDS traffic in flight ACROSS barriers. Here
//   * a workgroup is EIGHT waves (two per SIMD), tile (32*TMW) pixels x 256 channels, wave (wr, wc) owns 16*TMW pixels x 64
//     channels (TMW x 4 accumulators of 16x16); the two wave groups wr = 0 / 1 run one barrier apart, so that one group's
//     LDS reads and DMA issue sit beside the other group's MFMAs on the same SIMD;
//   * operands reach LDS by LDS-DMA only (buffer_load_dwordx4 ... lds, 1 KB per wave instruction = 16 rows x 64 B), never
//     through registers; a k tile is 64 channels of one filter tap, staged as four "half tiles" (A / B x the two 32-channel
//     halves), one half tile per phase, each 16 KB = two DMA instructions per wave;
//   * a k tile is four phases (k half ks = phase >> 1, pixel half mh = phase & 1): ds_read the fragments the phase needs,
//     issue one half tile of a later k tile, s_waitcnt vmcnt(8), barrier, MFMAs, barrier. Two LDS buffers (k tile parity).
//     Phase p = 4 t + s issues: s = 0  B-half-1 of tile t+1 | 1  A-half-1 of t+1 | 2  B-half-0 of t+2 | 3  A-half-0 of t+2.
//     Every half tile is therefore issued >= 2 phases after the last read of the buffer it overwrites (the two groups are one
//     barrier apart: a DMA may only be issued two phases after the read it must not overtake), is waited for 4 phases later
//     (all but the wave's 8 youngest DMA instructions = 4 half tiles) and first read 5 or 6 phases later — one phase after
//     the wait, behind a barrier every wave has passed. ~2.5k cycles of flight time per load, none of it exposed.
//   * LDS image of a half tile: [rows][64 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 3): the ds_read_b128 of a
//     16x16x32 fragment (lane l: row l & 15, chunk l >> 4) is conflict-free; LDS-DMA writes lane-linear, so the permutation
//     is applied to the SOURCE address (lane = (row r, position p) fetches chunk p ^ ((r >> 1) & 3)).
//   * SAME padding without a branch: the A descriptor's base is moved back by (pad_t * W + pad_l) pixels, a lane's offset
//     is its output pixel's, the tap is the scalar offset, and a tap that falls outside the image for that pixel replaces
//     the lane's offset by one beyond num_records (the DMA then moves nothing for that lane and LDS gets zeros).
//   * D = W_frag x X_frag: the accumulator's ROWS are channels, so a lane holds 4 consecutive channels of one pixel per
//     accumulator; the weight rows of a wave's slab are permuted so that a lane's 16 values per pixel tile are channels
//     {h*32 + q*8 .. +7 : h = 0, 1} (q = lane >> 4): 16-byte fp16 stores, four lanes = 64 contiguous bytes.
#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int P8_LDS = 131072;  // [parity][k half][A 16 KB | B 16 KB]
constexpr int P8_MAX_TAPS = 25;

struct P8Params {
    const _Float16* x;         // [B][H][W][Cin]
    const _Float16* w;         // [Cout][KH][KW][Cin]
    const float* scale;        // [Cout] or null
    const float* shift;        // [Cout] or null
    const _Float16* residual;  // [B][OH][OW][Cout] fp16 or null
    _Float16* y16;             // fp16 output or null
    float* y32;                // fp32 output or null
    int B, H, W, Cin, Cout, KH, KW, pad_t, pad_l, OH, OW;
    int M, K, nk;              // M = B*OH*OW, K = KH*KW*Cin, nk = K / 64
    int relu;
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes, x_bias;  // x_bias = (pad_t*W + pad_l)*Cin*2: how far the A descriptor's base is moved back
    unsigned y_elems;                   // M * Cout
};

// (tap, channel block) cursor of a k tile: scalar offset of the tap's first channel of that block, in bytes
struct KCursor {
    int tap, c0, kx, soff, tapoff;
    __device__ __forceinline__ void init() { tap = 0; c0 = 0; kx = 0; soff = 0; tapoff = 0; }
    __device__ __forceinline__ void advance(const P8Params& p) {
        c0 += 64;
        soff += 128;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            if (++kx == p.KW) {
                kx = 0;
                tapoff += (p.W - p.KW + 1) * p.Cin * 2;
            } else {
                tapoff += p.Cin * 2;
            }
            soff = tapoff;
        }
    }
};

template <int OFF>
__device__ __forceinline__ f16x8 lds_rd(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <int TMW>
__global__ __launch_bounds__(512) void conv_f16p(const P8Params p) {
    constexpr int BM = 32 * TMW;
    constexpr int T0 = (TMW + 1) / 2, T1 = TMW - T0;  // pixel tiles of a wave's two phases
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_u8* smem = (lds_u8*)smem_raw;

    // XCD-aware tile order (as conv_common.hpp): XCD x owns a contiguous range of M tiles and walks N fastest
    int m0, n0;
    {
        const int t = blockIdx.x, xcd = t & 7, seq = t >> 3;
        const int mt_lo = (xcd * p.tiles_m) >> 3, mt_hi = ((xcd + 1) * p.tiles_m) >> 3;
        const int mt = mt_lo + seq / p.tiles_n;
        if (mt >= mt_hi) return;
        m0 = mt * BM;
        n0 = (seq % p.tiles_n) * 256;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- staging: lane = (row rr of a 16-row piece, position pp), fetches chunk cc --------------------------------
    const int rr = lane >> 2, pp = lane & 3, cc = pp ^ ((rr >> 1) & 3);
    unsigned avoff[2], amask[2], bvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave + 8 * i, row = piece * 16 + rr, m = m0 + row;
        const bool valid = row < BM && m < p.M;
        const int ohw = p.OH * p.OW;
        const int mm = valid ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
        avoff[i] = valid ? static_cast<unsigned>(((b * p.H + oy) * p.W + ox) * p.Cin) * 2u + cc * 16u : OOB;
        unsigned mask = 0;
        for (int ky = 0, tap = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int iy = oy - p.pad_t + ky, ix = ox - p.pad_l + kx;
                if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << tap;
            }
        amask[i] = mask;
        // weights: LDS row rho of the 256-row tile = (slab wcb, accumulator j, local row 4q + r) <- channel
        // n0 + wcb*64 + (j >> 1)*32 + q*8 + (j & 1)*4 + r
        const int rho = piece * 16 + rr, wcb = rho >> 6, j = (rho >> 4) & 3, q = (rho >> 2) & 3, r = rho & 3;
        const int ch = n0 + wcb * 64 + (j >> 1) * 32 + q * 8 + (j & 1) * 4 + r;
        bvoff[i] = static_cast<unsigned>(ch) * static_cast<unsigned>(p.K) * 2u + cc * 16u;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(const_cast<_Float16*>(p.x)) - p.x_bias, 0, p.x_bytes + p.x_bias, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

    const int nk = p.nk;
    auto dma_a = [&](int kt, int ks, const KCursor& cur) {
        lds_u8* dst = smem + ((kt & 1) * 2 + ks) * 32768 + wave * 1024;
        const int sh = kt < nk ? cur.tap : 31;  // past the end of K: bit 31 is never set, the DMA moves nothing
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned v = ((amask[i] >> sh) & 1u) ? avoff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst + i * 8192, 16, static_cast<int>(v), cur.soff + ks * 64, 0, 0);
        }
    };
    auto dma_b = [&](int kt, int ks) {
        lds_u8* dst = smem + ((kt & 1) * 2 + ks) * 32768 + 16384 + wave * 1024;
        const bool live = kt < nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned v = live ? bvoff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst + i * 8192, 16, static_cast<int>(v), kt * 128 + ks * 64, 0, 0);
        }
    };

    // ---- fragment addressing --------------------------------------------------------------------------------------
    const int l16 = lane & 15, lq = lane >> 4;
    const unsigned lane_off = static_cast<unsigned>(l16 * 64 + ((lq ^ ((l16 >> 1) & 3)) << 4));
    // per parity (the instruction's offset field is 16 bits): + ks*32768 + tile*1024 as the immediate
    const unsigned xa0 = lane_off + static_cast<unsigned>(wr * TMW * 1024), xa1 = xa0 + 65536u;
    const unsigned wa0 = lane_off + 16384u + static_cast<unsigned>(wc * 4096), wa1 = wa0 + 65536u;

    f32x4 acc[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: k tile 0 whole, half 0 of k tile 1 ---------------------------------------------------------------
    KCursor c1, c2;  // cursors of the k tiles t+1 and t+2 of the running tile t
    c1.init();
    dma_b(0, 0);
    dma_a(0, 0, c1);
    dma_b(0, 1);
    dma_a(0, 1, c1);
    c1.advance(p);
    dma_b(1, 0);
    dma_a(1, 0, c1);
    c2 = c1;
    c2.advance(p);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second group runs one barrier behind

    f16x8 wf[4], xf[T0];
    // one phase: S = 0..3 of k tile kt with LDS parity PAR
#define P8_PHASE(PAR, S)                                                                                             \
    {                                                                                                                \
        constexpr int KS = (S) >> 1, MH = (S) & 1;                                                                   \
        constexpr int BASE = KS * 32768;                                                                             \
        const unsigned xa = (PAR) ? xa1 : xa0, wa = (PAR) ? wa1 : wa0;                                               \
        if constexpr (MH == 0) {                                                                                     \
            wf[0] = lds_rd<BASE + 0 * 1024>(wa);                                                                     \
            wf[1] = lds_rd<BASE + 1 * 1024>(wa);                                                                     \
            wf[2] = lds_rd<BASE + 2 * 1024>(wa);                                                                     \
            wf[3] = lds_rd<BASE + 3 * 1024>(wa);                                                                     \
            xf[0] = lds_rd<BASE + 0 * 1024>(xa);                                                                     \
            if constexpr (T0 > 1) xf[1] = lds_rd<BASE + 1 * 1024>(xa);                                               \
            if constexpr (T0 > 2) xf[2] = lds_rd<BASE + 2 * 1024>(xa);                                               \
            if constexpr (T0 > 3) xf[3] = lds_rd<BASE + 3 * 1024>(xa);                                               \
        } else {                                                                                                     \
            if constexpr (T1 > 0) xf[0] = lds_rd<BASE + (T0 + 0) * 1024>(xa);                                        \
            if constexpr (T1 > 1) xf[1] = lds_rd<BASE + (T0 + 1) * 1024>(xa);                                        \
            if constexpr (T1 > 2) xf[2] = lds_rd<BASE + (T0 + 2) * 1024>(xa);                                        \
            if constexpr (T1 > 3) xf[3] = lds_rd<BASE + (T0 + 3) * 1024>(xa);                                        \
        }                                                                                                            \
        if constexpr ((S) == 0) dma_b(kt + 1, 1);                                                                    \
        if constexpr ((S) == 1) dma_a(kt + 1, 1, c1);                                                                \
        if constexpr ((S) == 2) dma_b(kt + 2, 0);                                                                    \
        if constexpr ((S) == 3) dma_a(kt + 2, 0, c2);                                                                \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                             \
        __builtin_amdgcn_s_barrier();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int i = 0; i < (MH == 0 ? T0 : T1); ++i) {                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
                acc[MH * T0 + i][j] =                                                                                \
                    __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[MH * T0 + i][j], 0, 0, 0);             \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
    }

    for (int kt = 0; kt < nk; kt += 2) {
        P8_PHASE(0, 0)
        P8_PHASE(0, 1)
        P8_PHASE(0, 2)
        P8_PHASE(0, 3)
        c1.advance(p);
        c2.advance(p);
        if (kt + 1 < nk) {
            ++kt;
            P8_PHASE(1, 0)
            P8_PHASE(1, 1)
            P8_PHASE(1, 2)
            P8_PHASE(1, 3)
            c1.advance(p);
            c2.advance(p);
            --kt;
        }
    }
#undef P8_PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: lane (pixel l16 of a pixel tile, channel group q) holds channels cb + {0..7} and cb + 32 + {0..7} -------
    const int cb = n0 + wc * 64 + lq * 8;
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[h][e] = p.scale ? p.scale[cb + h * 32 + e] : 1.0f;
            sh[h][e] = p.shift ? p.shift[cb + h * 32 + e] : 0.0f;
        }
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.residual), 0, p.residual ? p.y_elems * 2u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t y16_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y16, 0, p.y16 ? p.y_elems * 2u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t y32_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y32, 0, p.y32 ? p.y_elems * 4u : 0u, 0x00020000);
    unsigned erow[TMW];  // element offset of (pixel, cb), or OOB-ish marker
    bool eok[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int m = m0 + wr * 16 * TMW + i * 16 + l16;
        eok[i] = m < p.M;
        erow[i] = static_cast<unsigned>(eok[i] ? m : 0) * static_cast<unsigned>(p.Cout) + static_cast<unsigned>(cb);
    }
    u32x4 rv[TMW][2];
    if (p.residual) {  // every residual word in one burst, ahead of the stores
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                rv[i][h] = __builtin_amdgcn_raw_buffer_load_b128(
                    r_rsrc, static_cast<int>(eok[i] ? (erow[i] + h * 32) * 2u : OOB), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = acc[i][2 * h + (e >> 2)][e & 3] * sc[h][e] + sh[h][e];
            }
            if (p.residual) {
                const f16x8 r8 = __builtin_bit_cast(f16x8, rv[i][h]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += static_cast<float>(r8[e]);
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (p.y16) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = static_cast<_Float16>(v[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y16_rsrc,
                                                       static_cast<int>(eok[i] ? (erow[i] + h * 32) * 2u : OOB), 0, 0);
            }
            if (p.y32) {
                const unsigned off = eok[i] ? (erow[i] + h * 32) * 4u : OOB;
                const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), y32_rsrc, static_cast<int>(off), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), y32_rsrc,
                                                       static_cast<int>(eok[i] ? off + 16u : OOB), 0, 0);
            }
        }
    }
}

// Tile height for (M, Cout): BM = 32*TMW in {128, 160, 192, 256}; the choice that needs the fewest pixel rows per CU when the
// tiles are dealt out in rounds of `cus` workgroups (one workgroup per CU).
int pick_tmw(long long m, int cout, int cus) {
    const int cand[4] = {8, 6, 5, 4};
    int best = 0;
    long long best_cost = 0;
    for (int c : cand) {
        const long long bm = 32LL * c, tiles = ((m + bm - 1) / bm) * (cout / 256);
        const long long rounds = (tiles + cus - 1) / cus;
        // per-tile cost: the rows plus a fixed part (prologue / epilogue / the first loads) worth ~24 rows
        const long long cost = rounds * (bm + 24);
        if (!best || cost < best_cost) {
            best = c;
            best_cost = cost;
        }
    }
    return best;
}

template <int TMW>
int launch_p8(P8Params p, hipStream_t s) {
    p.tiles_m = (p.M + 32 * TMW - 1) / (32 * TMW);
    p.tiles_n = p.Cout / 256;
    const long long grid = 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n;
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16p: grid too large");
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_f16p<TMW>), P8_LDS, "conv_f16p")) return rc;
    hipLaunchKernelGGL(conv_f16p<TMW>, dim3(static_cast<unsigned>(grid)), dim3(512), P8_LDS, s, p);
    return mrcnn::check_launch("conv_f16p");
}

}  // namespace

extern "C" int mrcnn_conv_f16_pipelined_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t cout,
                                                  int32_t kh, int32_t kw, int32_t pad_top, int32_t pad_left,
                                                  int32_t pad_bottom, int32_t pad_right) {
    if (batch < 1 || height < 1 || width < 1 || cin < 64 || cin % 64 || cout < 256 || cout % 256) return 0;
    if (kh < 1 || kw < 1 || kh * kw > P8_MAX_TAPS) return 0;
    const long long oh = height + pad_top + pad_bottom - kh + 1, ow = width + pad_left + pad_right - kw + 1;
    if (oh < 1 || ow < 1 || pad_top < 0 || pad_left < 0 || pad_top >= kh || pad_left >= kw) return 0;
    const long long m = static_cast<long long>(batch) * oh * ow;
    const long long lim = 1LL << 31;  // byte offsets are 32-bit, fp32 output included
    if (m * cout * 4 >= lim || static_cast<long long>(batch) * height * width * cin * 2 >= lim) return 0;
    if (static_cast<long long>(cout) * kh * kw * cin * 2 >= lim) return 0;
    return 1;
}

extern "C" int mrcnn_conv_f16_pipelined(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                        const void* w_f16, int32_t cout, int32_t kh, int32_t kw, int32_t pad_top,
                                        int32_t pad_left, int32_t pad_bottom, int32_t pad_right, const float* scale,
                                        const float* shift, const void* residual_f16, int32_t activation, void* y_f16,
                                        float* y_f32, int32_t tile_rows, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_f16 && w_f16 && (y_f16 || y_f32), "conv_f16_pipelined: null pointer");
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv_f16_pipelined: activation must be 0 (none) or 1 (ReLU)");
    if (!mrcnn_conv_f16_pipelined_supported(batch, height, width, cin, cout, kh, kw, pad_top, pad_left, pad_bottom, pad_right))
        return mrcnn::fail(MRCNN_ERR_UNSUPPORTED,
                           "conv_f16_pipelined: needs stride 1, Cin %% 64 == 0, Cout %% 256 == 0, <= 25 taps, pads < kernel, "
                           "32-bit byte offsets (got %dx%dx%dx%d -> %d, %dx%d)", batch, height, width, cin, cout, kh, kw);
    P8Params p{};
    p.x = static_cast<const _Float16*>(x_f16);
    p.w = static_cast<const _Float16*>(w_f16);
    p.scale = scale;
    p.shift = shift;
    p.residual = static_cast<const _Float16*>(residual_f16);
    p.y16 = static_cast<_Float16*>(y_f16);
    p.y32 = y_f32;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw;
    p.pad_t = pad_top; p.pad_l = pad_left;
    p.OH = height + pad_top + pad_bottom - kh + 1;
    p.OW = width + pad_left + pad_right - kw + 1;
    p.M = batch * p.OH * p.OW;
    p.K = kh * kw * cin;
    p.nk = p.K / 64;
    p.relu = activation;
    p.x_bytes = static_cast<unsigned>(static_cast<long long>(batch) * height * width * cin * 2);
    p.w_bytes = static_cast<unsigned>(static_cast<long long>(cout) * p.K * 2);
    p.x_bias = static_cast<unsigned>((pad_top * width + pad_left) * cin * 2);
    p.y_elems = static_cast<unsigned>(static_cast<long long>(p.M) * cout);
    int tmw = tile_rows / 32;
    if (tile_rows == 0) {
        const int cus = mrcnn::device_cu_count();
        tmw = pick_tmw(p.M, cout, cus > 0 ? cus : 256);
    }
    hipStream_t s = mrcnn::as_stream(stream);
    switch (tmw) {
        case 4: return launch_p8<4>(p, s);
        case 5: return launch_p8<5>(p, s);
        case 6: return launch_p8<6>(p, s);
        case 8: return launch_p8<8>(p, s);
        default: return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "conv_f16_pipelined: tile_rows must be 0 (auto), 128, 160, 192 or 256");
    }
}
