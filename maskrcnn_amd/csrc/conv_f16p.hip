// Pipelined fp16 implicit-GEMM convolution for gfx950 (BASELINE configs[4]'s "fp16 MFMA path"):
// 1x1 / 3x3 (any KH x KW up to 25 taps, any stride) conv + affine + fp16 residual (same size, or half size = FPN
// nearest-upsample-add) + ReLU, fp16 NHWC in, fp16 and / or fp32 NHWC out, v_mfma_f32_16x16x32_f16 with fp32 accumulation.
// Same operands and the same k order per output element as conv_f16.hip's PRODUCTS = 1 path.
//
// What conv_f16.hip cannot do at one workgroup per CU (its 128x128 tile with a barrier every 8 MFMAs runs the 3x3 layers at
// 600-770 TFLOP/s): keep global -> LDS traffic in flight ACROSS barriers. Here
//   * a workgroup is EIGHT waves (two per SIMD), tile (32*TMW) pixels x (64*NWT) channels; wave (wr, wc) owns 16*TMW pixels
//     x 16*NWT channels (TMW x NWT accumulators of 16x16); the two wave groups wr = 0 / 1 run one barrier apart, so that one
//     group's LDS reads and DMA issue sit beside the other group's MFMAs on the same SIMD;
//   * operands reach LDS by LDS-DMA only (buffer_load_dwordx4 ... lds, 1 KB per wave instruction = 16 rows x 64 B), never
//     through registers; a k tile is 64 channels of one filter tap, staged as four "half tiles" (A / B x the two 32-channel
//     halves), one half tile per phase;
//   * a k tile is four phases (k half ks = phase >> 1, pixel half mh = phase & 1): ds_read the fragments the phase needs,
//     issue one half tile of a later k tile, s_waitcnt vmcnt(N), barrier, MFMAs, barrier. Two LDS buffers (k tile parity).
//     Phase p = 4 t + s issues: s = 0  B-half-1 of tile t+1 | 1  A-half-1 of t+1 | 2  B-half-0 of t+2 | 3  A-half-0 of t+2.
//     Every half tile is therefore issued >= 2 phases after the last read of the buffer it overwrites (the two groups are one
//     barrier apart: a DMA may only be issued two phases after the read it must not overtake), is waited for 4 phases later
//     (all but the wave's youngest 4 half tiles' instructions) and first read 5 or 6 phases later — one phase after the
//     wait, behind a barrier every wave has passed. ~2.5k cycles of flight time per load, none of it exposed.
//   * LDS image of a half tile: [rows][64 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 3): the ds_read_b128 of a
//     16x16x32 fragment (lane l: row l & 15, chunk l >> 4) is conflict-free; LDS-DMA writes lane-linear, so the permutation
//     is applied to the SOURCE address (lane = (row r, position p) fetches chunk p ^ ((r >> 1) & 3)).
//   * SAME padding without a branch: the A descriptor's base is moved back by (pad_t * W + pad_l) pixels, a lane's offset
//     is its output pixel's (times the stride), the tap is the scalar offset, and a tap that falls outside the image for that
//     pixel replaces the lane's offset by one beyond num_records (the DMA moves nothing for that lane, LDS gets zeros).
//   * D = W_frag x X_frag: the accumulator's ROWS are channels, so a lane holds 4 consecutive channels of one pixel per
//     accumulator; the weight rows of a wave's slab are permuted so that a lane's values per pixel tile are 8 consecutive
//     channels per 32-channel half (q = lane >> 4: channels h*32 + q*8 .. +7): 16-byte fp16 stores, four lanes = 64
//     contiguous bytes (NWT = 1: 4 channels, 8-byte stores).
//   * Small tiles (TMW * NWT <= 8: 64 KB of LDS, <= 128 registers) run two workgroups per CU: the short-K expansion layers live
//     in their prologue and epilogue, and a second workgroup is what covers them.
#include "common.hpp"

// every workgroup barrier of this file (schedule-fuzz builds sleep a pseudo-random time behind each: common.hpp)
#define P8_BARRIER() do { __builtin_amdgcn_s_barrier(); MRCNN_SYNC_FUZZ_POINT(); } while (0)

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int P8_MAX_TAPS = 25;

struct P8Params {
    const _Float16* x;         // [B][H][W][Cin]
    const _Float16* w;         // [Cout][KH][KW][Cin]
    const float* scale;        // [Cout] or null
    const float* shift;        // [Cout] or null
    const _Float16* residual;  // [B][OH/res_div][OW/res_div][Cout] fp16 or null
    _Float16* y16;             // fp16 output or null
    float* y32;                // fp32 output or null
    const _Float16* w_head;    // HEADS: [32][Cout] fp16 (rows 0-17: the RPN's conv_class then conv_bbox weights, the rest zero)
    float* head_part;          // HEADS: [Cout / 256][M][32] fp32 head sums, one plane per 256-channel tile
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW;
    int M, K, nk;              // M = B*OH*OW, K = KH*KW*Cin, nk = K / 64
    int relu, res_div;
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes, x_bias;  // x_bias = (pad_t*W + pad_l)*Cin*2: how far the A descriptor's base is moved back
    unsigned y_elems, r_elems;          // M * Cout; residual elements
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

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TMW, int NWT>
struct P8Geom {
    static constexpr int BM = 32 * TMW, BN = 64 * NWT;
    static constexpr int A_SZ = BM * 64, B_SZ = BN * 64;        // bytes of one half tile
    static constexpr int SLOT = A_SZ + B_SZ;                    // (parity, k half) slot: A then B
    static constexpr int LDS = 4 * SLOT + 1024;                 // + the dump piece of the DMA instructions that carry nothing
    static constexpr int NA = (2 * TMW + 7) / 8, NB = (4 * NWT + 7) / 8;  // DMA instructions per wave per half tile
    static constexpr int WAITN = 2 * NA + 2 * NB;               // the youngest four half tiles (A, B, A, B)
    static constexpr int WGS = (2 * LDS <= 160 * 1024 && TMW * NWT <= 8) ? 2 : 1;  // workgroups per CU (LDS and <= 128 registers)
};

// HEADS (the RPN's shared conv with its two 1x1 heads, model.py:605-607,624-641, in one launch): the ReLU'd fp16 tile is not
// stored. A lane's eight consecutive channels of a pixel ARE a B-operand fragment of v_mfma_f32_16x16x32_f16 (lane = (pixel
// column l16, k chunk q)), so each wave multiplies its 64 channels by the [32][64] slice of the head weights straight out of
// registers (4 MFMAs per pixel tile), the four waves of a pixel row add their partial sums through LDS in a fixed order, and
// the workgroup writes [256 pixels][32] fp32 sums of its 256-channel tile; the consumer (mrcnn_rpn_scores_deltas_v2_f32, form
// 4) adds the Cout / 256 planes and the bias. The 512-channel activation never reaches HBM.
template <int TMW, int NWT, bool RES, bool HEADS = false>
__global__ __launch_bounds__(512, (P8Geom<TMW, NWT>::WGS)) void conv_f16p(const P8Params p) {
    static_assert(!HEADS || (NWT == 4 && !RES), "the heads epilogue belongs to the 256-channel tile without a residual");
    using G = P8Geom<TMW, NWT>;
    constexpr int BM = G::BM;
    constexpr int T0 = (TMW + 1) / 2, T1 = TMW - T0;  // pixel tiles of a wave's two phases
    constexpr int SLOT = G::SLOT, A_SZ = G::A_SZ;
    static_assert(SLOT + A_SZ + 4 * NWT * 1024 <= 65536, "ds_read immediates are 16 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_u8* smem = (lds_u8*)smem_raw;
    lds_u8* dump = smem + 4 * SLOT;

    // XCD-aware tile order (as conv_common.hpp): XCD x owns a contiguous range of M tiles and walks N fastest
    int m0, n0;
    {
        const int t = blockIdx.x, xcd = t & 7, seq = t >> 3;
        const int mt_lo = (xcd * p.tiles_m) >> 3, mt_hi = ((xcd + 1) * p.tiles_m) >> 3;
        const int mt = mt_lo + seq / p.tiles_n;
        if (mt >= mt_hi) return;
        m0 = mt * BM;
        n0 = (seq % p.tiles_n) * G::BN;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- staging: lane = (row rr of a 16-row piece, position pp), fetches chunk cc --------------------------------
    const int rr = lane >> 2, pp = lane & 3, cc = pp ^ ((rr >> 1) & 3);
    unsigned avoff[G::NA], amask[G::NA], bvoff[G::NB];
#pragma unroll
    for (int i = 0; i < G::NA; ++i) {
        const int piece = wave + 8 * i, row = piece * 16 + rr, m = m0 + row;
        const bool valid = row < BM && m < p.M;
        const int ohw = p.OH * p.OW;
        const int mm = valid ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
        avoff[i] = valid ? static_cast<unsigned>(((b * p.H + oy * p.stride) * p.W + ox * p.stride) * p.Cin) * 2u + cc * 16u : OOB;
        unsigned mask = 0;
        for (int ky = 0, tap = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
                if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << tap;
            }
        amask[i] = mask;
    }
#pragma unroll
    for (int i = 0; i < G::NB; ++i) {
        // weights: LDS row rho of the tile = (slab wcb, accumulator j, local row 4q + r) <- channel
        // n0 + wcb*16*NWT + (j >> 1)*32 + q*8 + (j & 1)*4 + r   (NWT = 1: n0 + wcb*16 + q*4 + r)
        const int rho = (wave + 8 * i) * 16 + rr;
        const int wcb = rho / (16 * NWT), j = (rho >> 4) % NWT, q = (rho >> 2) & 3, r = rho & 3;
        const int ch = n0 + wcb * 16 * NWT + (NWT == 1 ? q * 4 + r : (j >> 1) * 32 + q * 8 + (j & 1) * 4 + r);
        bvoff[i] = rho < G::BN ? static_cast<unsigned>(ch) * static_cast<unsigned>(p.K) * 2u + cc * 16u : OOB;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(const_cast<_Float16*>(p.x)) - p.x_bias, 0, p.x_bytes + p.x_bias, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

    const int nk = p.nk;
    // a piece beyond the tile's rows carries nothing: it is pointed at the dump piece (its lanes are out of range: zeros)
    auto dma_a = [&](int kt, int ks, const KCursor& cur) {
        lds_u8* slot = smem + ((kt & 1) * 2 + ks) * SLOT;
        const int sh = kt < nk ? cur.tap : 31;  // past the end of K: bit 31 is never set, the DMA moves nothing
#pragma unroll
        for (int i = 0; i < G::NA; ++i) {
            const int piece = wave + 8 * i;
            const unsigned v = ((amask[i] >> sh) & 1u) ? avoff[i] : OOB;
            lds_u8* dst = piece < 2 * TMW ? slot + piece * 1024 : dump;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, static_cast<int>(v), cur.soff + ks * 64, 0, 0);
        }
    };
    auto dma_b = [&](int kt, int ks) {
        lds_u8* slot = smem + ((kt & 1) * 2 + ks) * SLOT + A_SZ;
        const bool live = kt < nk;
#pragma unroll
        for (int i = 0; i < G::NB; ++i) {
            const int piece = wave + 8 * i;
            const unsigned v = live ? bvoff[i] : OOB;
            lds_u8* dst = piece < 4 * NWT ? slot + piece * 1024 : dump;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, static_cast<int>(v), kt * 128 + ks * 64, 0, 0);
        }
    };

    // ---- fragment addressing --------------------------------------------------------------------------------------
    const int l16 = lane & 15, lq = lane >> 4;
    const unsigned lane_off = static_cast<unsigned>(l16 * 64 + ((lq ^ ((l16 >> 1) & 3)) << 4));
    // one base per parity (the instruction's offset field is 16 bits): + ks*SLOT + tile*1024 as the immediate
    const unsigned xa0 = lane_off + static_cast<unsigned>(wr * TMW * 1024), xa1 = xa0 + 2u * SLOT;
    const unsigned wa0 = lane_off + static_cast<unsigned>(A_SZ + wc * NWT * 1024), wa1 = wa0 + 2u * SLOT;

    f32x4 acc[TMW][NWT];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < NWT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    wait_vm<G::WAITN>();
    P8_BARRIER();
    if (wr == 1) P8_BARRIER();  // the second group runs one barrier behind

    f16x8 wf[NWT], xf[T0];
    // one phase: S = 0..3 of k tile kt with LDS parity PAR
#define P8_PHASE(PAR, S)                                                                                             \
    {                                                                                                                \
        constexpr int KS = (S) >> 1, MH = (S) & 1;                                                                   \
        constexpr int BASE = KS * SLOT;                                                                              \
        const unsigned xa = (PAR) ? xa1 : xa0, wa = (PAR) ? wa1 : wa0;                                               \
        if constexpr (MH == 0) {                                                                                     \
            wf[0] = lds_rd<BASE + 0 * 1024>(wa);                                                                     \
            if constexpr (NWT > 1) wf[1] = lds_rd<BASE + 1 * 1024>(wa);                                              \
            if constexpr (NWT > 2) wf[2] = lds_rd<BASE + 2 * 1024>(wa);                                              \
            if constexpr (NWT > 3) wf[3] = lds_rd<BASE + 3 * 1024>(wa);                                              \
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
        wait_vm<G::WAITN>();                                                                                         \
        P8_BARRIER();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int i = 0; i < (MH == 0 ? T0 : T1); ++i) {                                            \
            _Pragma("unroll") for (int j = 0; j < NWT; ++j) {                                                        \
                acc[MH * T0 + i][j] =                                                                                \
                    __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[MH * T0 + i][j], 0, 0, 0);             \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        P8_BARRIER();                                                                                \
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
    if (wr == 0) P8_BARRIER();  // pairs with the second group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: lane (pixel l16 of a pixel tile, channel group lq) holds, per 32-channel half h, the channels ----------
    // cb + 32 h + {0..7} (NWT = 1: cb + {0..3}). The residual is fetched here, in one burst ahead of the stores: fetching it
    // BEFORE the first operand DMA (registers allow it up to 192-row tiles) was measured on the short-K expansion layers it was
    // meant for — C4 conv3 of configs[4] 47.6 -> 51.6 us: the burst competes with the first k tiles' loads, which every wave
    // then waits for.
    constexpr int NH = NWT >= 2 ? NWT / 2 : 1;   // 32-channel halves of the wave's slab
    constexpr int CW = NWT >= 2 ? 8 : 4;         // channels per lane and half
    const int cb = n0 + wc * 16 * NWT + lq * CW;
    float sc[NH][CW], sh[NH][CW];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            sc[h][e] = p.scale ? p.scale[cb + h * 32 + e] : 1.0f;
            sh[h][e] = p.shift ? p.shift[cb + h * 32 + e] : 0.0f;
        }
    if constexpr (HEADS) {
        const __amdgpu_buffer_rsrc_t wh_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16*>(p.w_head), 0, 32u * static_cast<unsigned>(p.Cout) * 2u, 0x00020000);
        f16x8 wh[2][2];  // [head tile][32-channel half]: lane = (head l16 of the tile, k chunk lq)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                wh[ht][h] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    wh_rsrc, static_cast<int>(((ht * 16 + l16) * p.Cout + cb + h * 32) * 2), 0, 0));
        // this wave's partial sums go to LDS as [pixel 0 .. 16 TMW)[32 heads] fp32 (the operand buffers are dead: every read
        // of them was retired before the barriers above)
        lds_u8* red = smem + wave * (TMW * 16 * 128);
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 xb;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[i][2 * h + (e >> 2)][e & 3] * sc[h][e] + sh[h][e];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    asm volatile("" : "+v"(v));  // an fp32 value rounded ONCE to fp16, exactly as the storing epilogue does
                    xb[e] = static_cast<_Float16>(v);
                }
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) hacc[ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ht][h], xb, hacc[ht], 0, 0, 0);
            }
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)  // heads ht*16 + 4 lq + {0..3} of pixel i*16 + l16
                *(lds_f32x4*)(red + ((i * 16 + l16) * 32 + ht * 16 + lq * 4) * 4) = hacc[ht];
        }
        __syncthreads();
        const int nt = n0 >> 8;
        for (int item = threadIdx.x; item < BM * 8; item += 512) {
            const int px = item >> 3, hq = item & 7;
            const int g = px / (16 * TMW), pl = px - g * (16 * TMW);
            const lds_u8* src = smem + (g * 4) * (TMW * 16 * 128) + (pl * 32 + hq * 4) * 4;
            const f32x4 s0 = *(const lds_f32x4*)(src), s1 = *(const lds_f32x4*)(src + TMW * 16 * 128);
            const f32x4 s2 = *(const lds_f32x4*)(src + 2 * TMW * 16 * 128), s3 = *(const lds_f32x4*)(src + 3 * TMW * 16 * 128);
            const f32x4 sum = ((s0 + s1) + s2) + s3;  // fixed order: deterministic
            const int m = m0 + px;
            if (m < p.M)
                *reinterpret_cast<f32x4*>(p.head_part + (static_cast<size_t>(nt) * p.M + m) * 32 + hq * 4) = sum;
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.residual), 0, RES ? p.r_elems * 2u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t y16_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y16, 0, p.y16 ? p.y_elems * 2u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t y32_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y32, 0, p.y32 ? p.y_elems * 4u : 0u, 0x00020000);
    unsigned erow[TMW], rrow[RES ? TMW : 1];  // element offset of (pixel, cb) in the output / in the residual
    bool eok[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int m = m0 + wr * 16 * TMW + i * 16 + l16;
        eok[i] = m < p.M;
        const int mm = eok[i] ? m : 0;
        erow[i] = static_cast<unsigned>(mm) * static_cast<unsigned>(p.Cout) + static_cast<unsigned>(cb);
        if constexpr (RES) {
            if (p.res_div == 1) {
                rrow[i] = erow[i];
            } else {  // the residual pixel (oy / 2, ox / 2) of a [B][OH/2][OW/2][Cout] map (model.py:150-152)
                const int ohw = p.OH * p.OW;
                const int b = mm / ohw, rem = mm - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
                rrow[i] = static_cast<unsigned>((b * (p.OH >> 1) + (oy >> 1)) * (p.OW >> 1) + (ox >> 1)) *
                              static_cast<unsigned>(p.Cout) + static_cast<unsigned>(cb);
            }
        }
    }
    u32x4 rv[RES ? TMW : 1][NH];
    if constexpr (RES) {  // every residual word in one burst, ahead of the stores
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const unsigned off = eok[i] ? (rrow[i] + h * 32) * 2u : OOB;
                if constexpr (CW == 8) {
                    rv[i][h] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, static_cast<int>(off), 0, 0);
                } else {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, static_cast<int>(off), 0, 0);
                    rv[i][h] = u32x4{t.x, t.y, 0u, 0u};
                }
            }
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float v[CW];
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] = acc[i][NWT >= 2 ? 2 * h + (e >> 2) : 0][e & 3] * sc[h][e] + sh[h][e];
            if constexpr (RES) {
                const f16x8 r8 = __builtin_bit_cast(f16x8, rv[i][h]);
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] += static_cast<float>(r8[e]);
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            // the fp32 value is materialised before it is rounded to fp16: without this the compiler folds multiply-add and
            // conversion into v_fma_mixlo_f16 in SOME instantiations (one rounding instead of two: 1 fp16 ulp apart on a few
            // values per 100 000), and the heads epilogue / conv_igemm_f16 would not see the same activation bits
#pragma unroll
            for (int e = 0; e < CW; ++e) asm volatile("" : "+v"(v[e]));
            const unsigned eoff = erow[i] + h * 32;
            if (p.y16) {
                const unsigned off = eok[i] ? eoff * 2u : OOB;
                if constexpr (CW == 8) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = static_cast<_Float16>(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y16_rsrc, static_cast<int>(off), 0, 0);
                } else {
                    const f16x4 o = {static_cast<_Float16>(v[0]), static_cast<_Float16>(v[1]), static_cast<_Float16>(v[2]),
                                     static_cast<_Float16>(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), y16_rsrc, static_cast<int>(off), 0, 0);
                }
            }
            if (p.y32) {
                const unsigned off = eok[i] ? eoff * 4u : OOB;
                const f32x4 lo = {v[0], v[1], v[2], v[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), y32_rsrc, static_cast<int>(off), 0, 0);
                if constexpr (CW == 8) {
                    const f32x4 hi = {v[4], v[5], v[6], v[7]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), y32_rsrc,
                                                           static_cast<int>(eok[i] ? off + 16u : OOB), 0, 0);
                }
            }
        }
    }
}

// Rows of the 256-channel tile for (M, Cout): 32*TMW in {128, 160, 192, 256}, the choice that needs the fewest pixel rows per
// CU when the tiles are dealt out to `cus` CUs as they free up (one workgroup per CU).
int pick_rows(long long m, int cout, int cus) {
    const int cand[4] = {8, 6, 5, 4};
    double best_cost = 0;
    int tmw = 0;
    for (int c : cand) {
        const long long bm = 32LL * c, tiles = ((m + bm - 1) / bm) * (cout / 256);
        const double r = static_cast<double>(tiles) / cus, rc = static_cast<double>((tiles + cus - 1) / cus);
        // between whole rounds and the fractional count; a tile costs its rows plus a fixed part (prologue, epilogue, the
        // weight tile's traffic) worth ~64 rows. Measured on the configs[4] layers: 256 rows for the P2 / P3 3x3 layers, 160
        // for C4 conv1 / conv2, 128 for C5.
        const double cost = (0.5 * r + 0.5 * rc) * static_cast<double>(bm + 64);
        if (!tmw || cost < best_cost) {
            tmw = c;
            best_cost = cost;
        }
    }
    return tmw;
}

template <int TMW, int NWT, bool RES>
int launch_p8r(P8Params p, hipStream_t s) {
    using G = P8Geom<TMW, NWT>;
    p.tiles_m = (p.M + G::BM - 1) / G::BM;
    p.tiles_n = p.Cout / G::BN;
    const long long grid = 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n;
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16p: grid too large");
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_f16p<TMW, NWT, RES>), G::LDS, "conv_f16p")) return rc;
    hipLaunchKernelGGL((conv_f16p<TMW, NWT, RES>), dim3(static_cast<unsigned>(grid)), dim3(512), G::LDS, s, p);
    return mrcnn::check_launch("conv_f16p");
}

template <int TMW>
int launch_p8_heads(P8Params p, hipStream_t s) {
    using G = P8Geom<TMW, 4>;
    p.tiles_m = (p.M + G::BM - 1) / G::BM;
    p.tiles_n = p.Cout / G::BN;
    const long long grid = 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n;
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16p: grid too large");
    static_assert(8 * TMW * 16 * 128 <= G::LDS, "the partial head sums fit the operand buffers");
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_f16p<TMW, 4, false, true>), G::LDS, "conv_f16p")) return rc;
    hipLaunchKernelGGL((conv_f16p<TMW, 4, false, true>), dim3(static_cast<unsigned>(grid)), dim3(512), G::LDS, s, p);
    return mrcnn::check_launch("conv_f16p<heads>");
}

template <int TMW, int NWT>
int launch_p8(const P8Params& p, hipStream_t s) {
    return p.residual ? launch_p8r<TMW, NWT, true>(p, s) : launch_p8r<TMW, NWT, false>(p, s);
}

}  // namespace

extern "C" int mrcnn_conv_f16_pipelined_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t cout,
                                                  int32_t kh, int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                                                  int32_t pad_bottom, int32_t pad_right) {
    if (batch < 1 || height < 1 || width < 1 || cin < 64 || cin % 64 || cout < 64 || cout % 64) return 0;
    if (kh < 1 || kw < 1 || kh * kw > P8_MAX_TAPS || stride < 1 || stride > 8) return 0;
    if (pad_top < 0 || pad_left < 0 || pad_bottom < 0 || pad_right < 0 || pad_top >= kh || pad_left >= kw) return 0;
    if (height + pad_top + pad_bottom < kh || width + pad_left + pad_right < kw) return 0;
    const long long oh = (height + pad_top + pad_bottom - kh) / stride + 1, ow = (width + pad_left + pad_right - kw) / stride + 1;
    const long long m = static_cast<long long>(batch) * oh * ow;
    const long long lim = 1LL << 31;  // byte offsets are 32-bit, fp32 output included
    if (m * cout * 4 >= lim || static_cast<long long>(batch) * height * width * cin * 2 >= lim) return 0;
    if (static_cast<long long>(cout) * kh * kw * cin * 2 >= lim) return 0;
    return 1;
}

extern "C" int mrcnn_conv_f16_pipelined(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                        const void* w_f16, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                                        int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                                        const float* scale, const float* shift, const void* residual_f16, int32_t res_div,
                                        int32_t activation, void* y_f16, float* y_f32, int32_t tile_rows, int32_t tile_cols,
                                        mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_f16 && w_f16 && (y_f16 || y_f32), "conv_f16_pipelined: null pointer");
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv_f16_pipelined: activation must be 0 (none) or 1 (ReLU)");
    if (!mrcnn_conv_f16_pipelined_supported(batch, height, width, cin, cout, kh, kw, stride, pad_top, pad_left, pad_bottom,
                                            pad_right))
        return mrcnn::fail(MRCNN_ERR_UNSUPPORTED,
                           "conv_f16_pipelined: needs Cin %% 64 == 0, Cout %% 64 == 0, <= 25 taps, 0 <= pads < kernel, "
                           "32-bit byte offsets (got %dx%dx%dx%d -> %d, %dx%d stride %d)", batch, height, width, cin, cout,
                           kh, kw, stride);
    P8Params p{};
    p.x = static_cast<const _Float16*>(x_f16);
    p.w = static_cast<const _Float16*>(w_f16);
    p.scale = scale;
    p.shift = shift;
    p.residual = static_cast<const _Float16*>(residual_f16);
    p.y16 = static_cast<_Float16*>(y_f16);
    p.y32 = y_f32;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride;
    p.pad_t = pad_top; p.pad_l = pad_left;
    p.OH = (height + pad_top + pad_bottom - kh) / stride + 1;
    p.OW = (width + pad_left + pad_right - kw) / stride + 1;
    p.M = batch * p.OH * p.OW;
    p.K = kh * kw * cin;
    p.nk = p.K / 64;
    p.relu = activation;
    p.res_div = residual_f16 ? res_div : 1;
    MRCNN_REQUIRE(p.res_div == 1 || (p.res_div == 2 && p.OH % 2 == 0 && p.OW % 2 == 0),
                  "conv_f16_pipelined: res_div must be 1, or 2 with even output sizes (got %d, %dx%d)", res_div, p.OH, p.OW);
    p.x_bytes = static_cast<unsigned>(static_cast<long long>(batch) * height * width * cin * 2);
    p.w_bytes = static_cast<unsigned>(static_cast<long long>(cout) * p.K * 2);
    p.x_bias = static_cast<unsigned>((pad_top * width + pad_left) * cin * 2);
    p.y_elems = static_cast<unsigned>(static_cast<long long>(p.M) * cout);
    p.r_elems = p.y_elems / static_cast<unsigned>(p.res_div * p.res_div);
    // tile: 0 = automatic. Columns: 256 when Cout allows it — except for a short K (<= 256: the expansion layers) on a map
    // with fewer than three 256 x 256 tiles per CU, where 128 x 128 at two workgroups per CU covers the prologue / epilogue
    // better (C4 conv3 of configs[4]: 51.6 -> 49 us; C2 / C3 conv3 with 4-8 tiles per CU: the large tile wins) —, else 128 / 64.
    // Rows: pick_rows for 256 columns, 128 otherwise.
    const int cus = mrcnn::device_cu_count() > 0 ? mrcnn::device_cu_count() : 256;
    const bool few_tiles = static_cast<long long>(p.M) * cout < 3LL * cus * 65536;
    // tuning (in-pipeline sweeps, read once per process): MRCNN_F16P_TILE="rows x cols[:Mmin-Mmax]" forces a tile for the automatic
    // choices (optionally only for layers with Mmin <= M <= Mmax), e.g. "192x256:30000-40000"
    struct ForcedTile {   // parsed once per process (a getenv per launch walks the whole environment)
        int got = 0, r = 0, c = 0;
        long long mlo = 0, mhi = 1LL << 40;
        ForcedTile() {
            if (const char* e = mrcnn::tuning_env("MRCNN_F16P_TILE")) got = sscanf(e, "%dx%d:%lld-%lld", &r, &c, &mlo, &mhi);
        }
    };
    static const ForcedTile forced;
    if (forced.got >= 2) {
        const int r = forced.r, c = forced.c, got = forced.got;
        const long long mlo = forced.mlo, mhi = forced.mhi;
        if (got >= 2 && c > 0 && tile_rows == 0 && tile_cols == 0 && p.M >= mlo && p.M <= mhi && cout % c == 0 &&
            ((c == 256 && (r == 128 || r == 160 || r == 192 || r == 256)) || ((c == 128 || c == 64) && (r == 128 || r == 256)))) {
            tile_rows = r;
            tile_cols = c;
        }
    }
    int nwt = tile_cols ? tile_cols / 64 : (cout % 256 == 0 && !(p.K <= 256 && few_tiles)) ? 4 : (cout % 128 == 0 ? 2 : 1);
    // Small maps (P5 / P6 of configs[4]: M = 8 736 / 2 184 rows): when even 128-row tiles of this width leave more than half the CUs
    // without a workgroup, narrower tiles — two workgroups per CU — fill the chip (round 5, tools/f16_small_probe.py: P5 lateral
    // 31.5 -> 23.0 us, P5 smoothing 33.5 -> 25.5, RPN P6 32.2 -> 23.0; layers with >= cus / 2 tiles are best where they are).
    // The tile never changes a result (same operands, same k order per output element).
    if (!tile_cols && !tile_rows) {
        while (nwt > 1 && ((p.M + 127) / 128) * static_cast<long long>(cout / (64 * nwt)) < cus / 2) nwt >>= 1;
    }
    int tmw = tile_rows ? tile_rows / 32 : 0;
    if (!tmw) tmw = nwt == 4 ? pick_rows(p.M, cout, cus) : 4;
    MRCNN_REQUIRE((tile_cols == 0 || tile_cols == 64 * nwt) && (tile_rows == 0 || tile_rows == 32 * tmw) && nwt >= 1 &&
                      cout % (64 * nwt) == 0,
                  "conv_f16_pipelined: tile %d x %d does not fit Cout %d", tile_rows, tile_cols, cout);
    hipStream_t s = mrcnn::as_stream(stream);
    if (nwt == 4) {
        switch (tmw) {
            case 4: return launch_p8<4, 4>(p, s);
            case 5: return launch_p8<5, 4>(p, s);
            case 6: return launch_p8<6, 4>(p, s);
            case 8: return launch_p8<8, 4>(p, s);
            default: break;
        }
    } else if (nwt == 2) {
        switch (tmw) {
            case 4: return launch_p8<4, 2>(p, s);
            case 8: return launch_p8<8, 2>(p, s);
            default: break;
        }
    } else if (nwt == 1) {
        switch (tmw) {
            case 4: return launch_p8<4, 1>(p, s);
            case 8: return launch_p8<8, 1>(p, s);
            default: break;
        }
    }
    return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT,
                       "conv_f16_pipelined: tile must be 0 (auto), rows 128 / 160 / 192 / 256 x 256 columns, or rows 128 / 256 x "
                       "128 / 64 columns (got %d x %d)", tile_rows, tile_cols);
}

extern "C" int mrcnn_conv_f16_pipelined_heads(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                              const void* w_f16, int32_t cout, int32_t kh, int32_t kw, int32_t pad_top,
                                              int32_t pad_left, int32_t pad_bottom, int32_t pad_right, const float* scale,
                                              const float* shift, int32_t activation, const void* w_head_f16,
                                              float* head_part, int32_t tile_rows, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_f16 && w_f16 && w_head_f16 && head_part, "conv_f16_pipelined_heads: null pointer");
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv_f16_pipelined_heads: activation must be 0 (none) or 1 (ReLU)");
    if (cout % 256 != 0 || !mrcnn_conv_f16_pipelined_supported(batch, height, width, cin, cout, kh, kw, 1, pad_top, pad_left,
                                                               pad_bottom, pad_right))
        return mrcnn::fail(MRCNN_ERR_UNSUPPORTED,
                           "conv_f16_pipelined_heads: needs Cin %% 64 == 0, Cout %% 256 == 0, <= 25 taps, 0 <= pads < kernel, "
                           "32-bit byte offsets (got %dx%dx%dx%d -> %d, %dx%d)", batch, height, width, cin, cout, kh, kw);
    P8Params p{};
    p.x = static_cast<const _Float16*>(x_f16);
    p.w = static_cast<const _Float16*>(w_f16);
    p.scale = scale;
    p.shift = shift;
    p.w_head = static_cast<const _Float16*>(w_head_f16);
    p.head_part = head_part;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = 1;
    p.pad_t = pad_top; p.pad_l = pad_left;
    p.OH = height + pad_top + pad_bottom - kh + 1;
    p.OW = width + pad_left + pad_right - kw + 1;
    p.M = batch * p.OH * p.OW;
    p.K = kh * kw * cin;
    p.nk = p.K / 64;
    p.relu = activation;
    p.res_div = 1;
    p.x_bytes = static_cast<unsigned>(static_cast<long long>(batch) * height * width * cin * 2);
    p.w_bytes = static_cast<unsigned>(static_cast<long long>(cout) * p.K * 2);
    p.x_bias = static_cast<unsigned>((pad_top * width + pad_left) * cin * 2);
    p.y_elems = static_cast<unsigned>(static_cast<long long>(p.M) * cout);
    p.r_elems = p.y_elems;
    const int cus = mrcnn::device_cu_count() > 0 ? mrcnn::device_cu_count() : 256;
    const int tmw = tile_rows ? tile_rows / 32 : pick_rows(p.M, cout, cus);
    hipStream_t s = mrcnn::as_stream(stream);
    switch (tmw) {
        case 4: return launch_p8_heads<4>(p, s);
        case 5: return launch_p8_heads<5>(p, s);
        case 6: return launch_p8_heads<6>(p, s);
        case 8: return launch_p8_heads<8>(p, s);
        default: break;
    }
    return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "conv_f16_pipelined_heads: tile_rows must be 0 (auto), 128, 160, 192 or 256");
}
