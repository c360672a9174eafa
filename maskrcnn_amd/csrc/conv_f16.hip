// Fused convolution + affine + residual + ReLU on fp16-operand MFMA (v_mfma_f32_32x32x16_f16, fp32
// accumulate) for gfx950, channels-last. Same GEMM view, tile order, im2col predication and epilogue as
// conv.hip (the exact-fp32 kernel); what changes is the contraction:
//
//   PRODUCTS = 1   plain fp16 operands (BASELINE config 5's "fp16 MFMA path"): x_hi * w_hi.
//   PRODUCTS = 3   error-compensated split, fp32-grade accuracy at the fp16 MFMA rate / 3:
//                      x = x_hi + x_lo,  w = w_hi + w_lo   (hi = fp16(v), lo = fp16(v - hi): 22 mantissa bits)
//                      x*w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi        (dropped x_lo*w_lo <= 2^-22 |x*w|)
//                  every fp16 x fp16 product is exact in fp32 and the sum accumulates in fp32, so the result
//                  differs from an fp32 fmaf chain by ~2^-21 relative per term (valid for |x|,|w| < 65504).
//
// By default activations stay fp32 in HBM (so RoIAlign / residuals / the boundary see the same tensors as the fp32
// path); they are split into hi/lo fp16 planes while being staged to LDS. Weights are split once by the caller.
// IN16 / OUT16 (PRODUCTS = 1 only: BASELINE config 5's "fp16 MFMA path"): the input / the output and residual are fp16
// in HBM. An fp16 input is staged by plain 16-byte copies (half the bytes, no conversion); an fp16 output is written as
// 4-byte channel pairs (adjacent lanes swap one value by DPP, so a wave-level store is still 64-byte channel runs and
// there are half as many of them). A conv that reads an fp16 tensor sees exactly the operand it would have rounded
// an fp32 tensor to, so only residual adds, pooling and the final stores see the narrower storage.
//   LDS       per plane [rows][32 halves] = 64-byte rows, 16-byte chunks XOR-swizzled by (row>>2)&3 so that the
//             ds_read_b128 fragment reads (16 rows x one chunk per lane group) hit 16 distinct slots — no padding,
//             64 KiB per workgroup (PRODUCTS = 3, 128x128 tile) → two workgroups per CU.
//   MFMA      lane l holds A[row = l&31][k = 8*(l>>5) .. +7] (8 halves = one ds_read_b128) and the same for B;
//             a k tile of 32 is two 16-deep steps; per step and 32x32 tile: PRODUCTS MFMAs of 32 cycles.
#include "conv_common.hpp"

namespace {

using namespace mrcnn_conv;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct ConvParamsH : ConvCommon {
    const _Float16* w_hi;  // [Cout][KH][KW][Cin] fp16 planes; w_lo only for PRODUCTS = 3
    const _Float16* w_lo;
};

constexpr int BK = 32;          // k elements per tile
constexpr int ROW_BYTES = 64;   // 32 halves

template <int BM, int BN, int PRODUCTS>
constexpr size_t lds_bytes() {
    return static_cast<size_t>(2) * (BM + BN) * ROW_BYTES * (PRODUCTS == 3 ? 2 : 1);
}

__device__ __forceinline__ unsigned pack2(_Float16 a, _Float16 b) {
    f16x2 v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}

// fp16 output epilogue: act(acc*scale + shift (+ fp16 residual)) → fp16, RES 0 / 1 / 2 (half-size residual) / 4 (2x2
// transposed-conv scatter). A lane holds ONE channel of 16 rows; lanes 2j and 2j+1 (adjacent channels) pair up: for each
// pair of consecutive rows the even lane ends up with both channels of the first row, the odd lane with both of the
// second (one DPP swap), and each stores 4 bytes. The residual is fetched the same way in reverse.
__device__ __forceinline__ unsigned swap_pair(unsigned v) {
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
}

template <int TM, int TN, int WTM, int WTN, int RES>
__device__ __forceinline__ void epilogue16(const ConvCommon& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                           int lane) {
    static_assert(RES == 0 || RES == 1 || RES == 2 || RES == 4, "fp16 output: plain, residual, half-size residual, deconv");
    const int ln = lane & 31, lh = lane >> 5;
    const bool odd = ln & 1;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0,
                                                                            (RES == 1 || RES == 2) ? p.r_bytes : 0u, 0x00020000);
    const int rh = p.OH >> 1, rw_ = p.OW >> 1;
    float sc[TN], sh[TN];
    unsigned ncol[TN], rcol[TN];  // byte offset of the lane pair's two channels, or OOB
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WTN + jn * 32 + ln;
        const bool n_ok = n < p.Cout;  // Cout is even: both channels of a pair are in or out together
        sc[jn] = (n_ok && p.scale) ? p.scale[n] : 1.0f;
        sh[jn] = (n_ok && p.shift) ? p.shift[n] : 0.0f;
        const int np = n & ~1;
        rcol[jn] = n_ok ? static_cast<unsigned>(np) * 2u : OOB;
        if constexpr (RES != 4) {
            ncol[jn] = rcol[jn];
        } else {  // n = (dy*2 + dx)*cq + co → pixel (2i+dy, 2j+dx), channel co of the [B][2*OH][2*OW][cq] output
            const int cq = p.Cout >> 2, q = np / cq, co = np - q * cq;
            ncol[jn] = n_ok ? static_cast<unsigned>(((q >> 1) * 2 * p.OW + (q & 1)) * cq + co) * 2u : OOB;
        }
    }
    // row offsets of the lane's own row of every row pair, then ALL residual words in one burst ahead of the stores
    // (the short-K expansion layers live in this epilogue: a load → use → store chain per row pair serialises it)
    unsigned yrow[TM][8];
    unsigned rw[(RES == 1 || RES == 2) ? TM : 1][8][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WTM + i * 32 + 4 * lh;
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const int rmine = 2 * rp + (odd ? 1 : 0);  // the row this lane loads the residual of and stores
            const int m = mb + (rmine & 3) + 8 * (rmine >> 2);
            const bool ok = m < p.M;
            unsigned rrow = OOB;
            if constexpr (RES == 4) {
                int b, oy, ox;
                decode_pixel(p, ok ? m : 0, b, oy, ox);
                yrow[i][rp] = ok ? static_cast<unsigned>((b * 2 * p.OH + 2 * oy) * (2 * p.OW) + 2 * ox) * (static_cast<unsigned>(p.Cout >> 2) * 2u) : OOB;
            } else {
                yrow[i][rp] = ok ? static_cast<unsigned>(m) * (static_cast<unsigned>(p.Cout) * 2u) : OOB;
            }
            if constexpr (RES == 1) rrow = yrow[i][rp];
            if constexpr (RES == 2) {
                int b, oy, ox;
                decode_pixel(p, ok ? m : 0, b, oy, ox);
                rrow = ok ? static_cast<unsigned>((b * rh + (oy >> 1)) * rw_ + (ox >> 1)) * (static_cast<unsigned>(p.Cout) * 2u) : OOB;
            }
            if constexpr (RES == 1 || RES == 2) {
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const unsigned off = oob_add(rrow, rcol[jn]);
                    rw[i][rp][jn] = __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, static_cast<int>(off), 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const int r0 = 2 * rp, r1 = r0 + 1;
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                float v0 = acc[i][jn][r0] * sc[jn] + sh[jn], v1 = acc[i][jn][r1] * sc[jn] + sh[jn];
                if constexpr (RES == 1 || RES == 2) {
                    const unsigned w = rw[i][rp][jn];
                    // even lane: (row r0: own, neighbour's) — odd lane: (row r1: neighbour's, own)
                    const unsigned mine = odd ? (w >> 16) : (w & 0xFFFFu), theirs = odd ? (w & 0xFFFFu) : (w >> 16);
                    const unsigned got = swap_pair(theirs);  // my channel at the OTHER row
                    const float rm = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(mine)));
                    const float ro = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(got)));
                    v0 += odd ? ro : rm;
                    v1 += odd ? rm : ro;
                }
                if (p.act) {
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                }
                // fp32 first, then ONE rounding to fp16 (the compiler otherwise folds multiply-add and conversion into
                // v_fma_mixlo_f16 in some instantiations and not in others: 1 fp16 ulp apart on a few values per 100 000 —
                // conv_f16p.hip pins the same sequence, so the two kernels agree bit for bit)
                asm volatile("" : "+v"(v0), "+v"(v1));
                const unsigned h0 = __builtin_bit_cast(unsigned short, static_cast<_Float16>(v0));
                const unsigned h1 = __builtin_bit_cast(unsigned short, static_cast<_Float16>(v1));
                const unsigned keep = odd ? h1 : h0, got = swap_pair(odd ? h0 : h1);
                const unsigned word = odd ? (got | (keep << 16)) : (keep | (got << 16));
                const unsigned off = oob_add(yrow[i][rp], ncol[jn]);
                __builtin_amdgcn_raw_buffer_store_b32(word, y_rsrc, static_cast<int>(off), 0, 0);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int PRODUCTS, bool GENERIC, int RES, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(256, (BM * BN > 128 * 128) ? 1 : 2) void conv_igemm_f16(const ConvParamsH p) {
    static_assert(!(IN16 || OUT16) || PRODUCTS == 1, "fp16 storage is the plain-fp16 mode's");
    static_assert(!(IN16 && GENERIC), "an fp16 input has Cin % 32 == 0");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int AROWS = IN16 ? 64 : 32;  // tile rows covered per staging pass (IN16: 4 threads per row, else 8)
    constexpr int AEL = IN16 ? 8 : 4;      // k elements per staging slot
    constexpr int AEB = IN16 ? 2 : 4;      // bytes per element of x
    constexpr int PA = BM / AROWS;        // 16-byte slots per thread per k tile
    constexpr int PB = BN / 64;           // fp16 16-byte chunks per thread per plane per k tile (4 threads per row)
    constexpr int NPL = PRODUCTS == 3 ? 2 : 1;  // planes per operand (hi [, lo])
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [buf][plane][rows][64 B]
    constexpr int A_PLANE = BM * ROW_BYTES, B_PLANE = BN * ROW_BYTES;
    constexpr int A_BUF = A_PLANE * NPL, B_BUF = B_PLANE * NPL;
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * A_BUF;

    int m0, n0, nt;
    if (!tile_origin(p, BM, BN, m0, n0, nt)) return;  // XCD-aware tile order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ln = lane & 31, lh = lane >> 5;

    // ---- A (activations): fp32 in HBM → 8 threads per row, 4 consecutive k each; fp16 → 4 threads, 8 k ---------
    const int kq = IN16 ? (tid & 3) : (tid & 7), ar0 = IN16 ? (tid >> 2) : (tid >> 3);
    int a_off[PA], a_iy[PA], a_ix[PA];
    row_setup<PA, AROWS>(p, m0, ar0, a_off, a_iy, a_ix);
    // ---- B (weights, fp16 planes): 4 threads per row, 8 consecutive k (16 B) each ----------------------
    const int bc = tid & 3, br0 = tid >> 2;
    int b_off[PB];
    bool b_ok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + br0 + 64 * i;
        b_ok[i] = n < p.Cout;
        b_off[i] = (b_ok[i] ? n : 0) * p.K + bc * 8;  // halves
    }

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wh_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wl_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(PRODUCTS == 3 ? p.w_lo : p.w_hi), 0, p.w_bytes, 0x00020000);

    u32x4 ra[PA];        // 4 fp32 activations per slot (raw bits)
    u32x4 rb[NPL][PB];   // 8 fp16 weights per slot per plane
    // ---- staging in pieces (see conv.hip): PA activation pieces + NPL*PB weight pieces per k tile, issued
    // between the MFMAs of the main loop. Past the end of K the offsets are out of range (zeros, no traffic).
    int t_c0 = 0, t_ky = 0, t_kx = 0, t_k0 = 0;
    int cur_tap_off = 0, cur_ky = 0, cur_kx = 0, cur_k0 = 0;
    bool cur_kvalid = true;
    auto begin_load = [&]() {
        cur_k0 = t_k0;
        cur_kvalid = t_k0 < p.K;
        if constexpr (!GENERIC) {
            cur_ky = t_ky;
            cur_kx = t_kx;
            cur_tap_off = (t_ky * p.W + t_kx) * p.Cin + t_c0 + kq * AEL;
            t_c0 += BK;
            if (t_c0 >= p.Cin) {
                t_c0 = 0;
                if (++t_kx == p.KW) { t_kx = 0; ++t_ky; }
            }
        }
        t_k0 += BK;
    };
    constexpr int NPB = NPL * PB, NP = PA + NPB;
    auto load_piece = [&](int pc) {
        if (pc < PA) {
            const int i = pc;
            unsigned off;
            if constexpr (!GENERIC) {
                const bool ok = cur_kvalid &&
                                static_cast<unsigned>(a_iy[i] + cur_ky) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(a_ix[i] + cur_kx) < static_cast<unsigned>(p.W);
                off = ok ? static_cast<unsigned>(a_off[i] + cur_tap_off) * static_cast<unsigned>(AEB) : OOB;
            } else {
                const int kk = cur_k0 + kq * 4;
                const bool kin = kk < p.K;
                const int tap = kk / p.Cin, c = kk - tap * p.Cin;
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                const int tap_off = (ky * p.W + kx) * p.Cin + c;
                const bool ok = kin &&
                                static_cast<unsigned>(a_iy[i] + ky) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(a_ix[i] + kx) < static_cast<unsigned>(p.W);
                off = ok ? static_cast<unsigned>(a_off[i] + tap_off) * 4u : OOB;
            }
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(off), 0, 0);
        } else {
            const int q = pc - PA, pl = q / PB, i = q % PB;
            const bool ok = b_ok[i] && cur_kvalid && (cur_k0 + bc * 8 < p.K);
            const unsigned off = ok ? static_cast<unsigned>(b_off[i] + cur_k0) * 2u : OOB;
            rb[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(pl == 0 ? wh_rsrc : wl_rsrc, static_cast<int>(off), 0, 0);
        }
    };
    // swizzled byte offset of 16-byte chunk c of row r inside a plane
    auto chunk_off = [](int r, int c) { return r * ROW_BYTES + ((c ^ ((r >> 2) & 3)) << 4); };
    auto store_piece = [&](int pc, int buf) {
        if (pc < PA) {
            const int i = pc, r = ar0 + AROWS * i;
            if constexpr (IN16) {  // already fp16: a plain 16-byte copy
                *reinterpret_cast<u32x4*>(As + buf * A_BUF + chunk_off(r, kq)) = ra[i];
                return;
            }
            unsigned char* a = As + buf * A_BUF + chunk_off(r, kq >> 1) + (kq & 1) * 8;
            if constexpr (PRODUCTS == 3) {
                // hi = the top 11 significand bits (a mask: exactly an fp16 value), lo = fp16(v - hi):
                // 3 VALU per element instead of convert / convert back / subtract / convert
                const unsigned b0 = ra[i].x & 0xFFFFE000u, b1 = ra[i].y & 0xFFFFE000u;
                const unsigned b2 = ra[i].z & 0xFFFFE000u, b3 = ra[i].w & 0xFFFFE000u;
                const float h0 = __uint_as_float(b0), h1 = __uint_as_float(b1);
                const float h2 = __uint_as_float(b2), h3 = __uint_as_float(b3);
                const float l0 = __uint_as_float(ra[i].x) - h0, l1 = __uint_as_float(ra[i].y) - h1;
                const float l2 = __uint_as_float(ra[i].z) - h2, l3 = __uint_as_float(ra[i].w) - h3;
                u32x2 hi = {pack2(static_cast<_Float16>(h0), static_cast<_Float16>(h1)),
                            pack2(static_cast<_Float16>(h2), static_cast<_Float16>(h3))};
                u32x2 lo = {pack2(static_cast<_Float16>(l0), static_cast<_Float16>(l1)),
                            pack2(static_cast<_Float16>(l2), static_cast<_Float16>(l3))};
                *reinterpret_cast<u32x2*>(a) = hi;
                *reinterpret_cast<u32x2*>(a + A_PLANE) = lo;
            } else {
                u32x2 hi = {pack2(static_cast<_Float16>(__uint_as_float(ra[i].x)),
                                  static_cast<_Float16>(__uint_as_float(ra[i].y))),
                            pack2(static_cast<_Float16>(__uint_as_float(ra[i].z)),
                                  static_cast<_Float16>(__uint_as_float(ra[i].w)))};
                *reinterpret_cast<u32x2*>(a) = hi;
            }
        } else {
            const int q = pc - PA, pl = q / PB, i = q % PB;
            *reinterpret_cast<u32x4*>(Bs + buf * B_BUF + pl * B_PLANE + chunk_off(br0 + 64 * i, bc)) = rb[pl][i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: rows are multiples of 32 plus ln, so the swizzle term depends on the lane only
    const int sw = (ln >> 2) & 3;
    const unsigned char* Aw = As + (wm * WTM + ln) * ROW_BYTES;
    const unsigned char* Bw = Bs + (wn * WTN + ln) * ROW_BYTES;
    f16x8 fa[2][NPL][TM], fb[2][NPL][TN];
    auto read_frags = [&](int slot, int buf, int ks) {
        const int co = ((ks * 2 + lh) ^ sw) << 4;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[slot][pl][i] = *reinterpret_cast<const f16x8*>(Aw + buf * A_BUF + pl * A_PLANE +
                                                                  i * 32 * ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fb[slot][pl][i] = *reinterpret_cast<const f16x8*>(Bw + buf * B_BUF + pl * B_PLANE +
                                                                  i * 32 * ROW_BYTES + co);
        }
    };

    // ---- main loop: two 16-deep steps per k tile, order pinned (see conv.hip) -------------------------
    //   step 0: prefetch the fragments of step 1; MFMAs of step 0 with the LDS-write pieces of k tile kt+1;
    //   step 1: barrier; prefetch step 0 of k tile kt+1; MFMAs of step 1 with the load pieces of k tile kt+2.
    const int nk = (p.K + BK - 1) / BK;
    constexpr int NM = PRODUCTS * TM * TN;  // MFMAs per step
    begin_load();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, 0);
    __syncthreads();
    begin_load();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    read_frags(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) {
                read_frags(1, buf, 1);
            } else {
                __syncthreads();
                read_frags(0, buf ^ 1, 0);
                begin_load();
            }
            __builtin_amdgcn_sched_barrier(0);
            int m = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
                    for (int pr = 0; pr < PRODUCTS; ++pr) {
                        // small terms first: (lo,hi), (hi,lo), then (hi,hi)
                        const int pa = (PRODUCTS == 3 && pr == 0) ? 1 : 0;
                        const int pb = (PRODUCTS == 3 && pr == 1) ? 1 : 0;
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][pa][i], fb[ks][pb][jn],
                                                                            acc[i][jn], 0, 0, 0);
#pragma unroll
                        for (int pc = m * NP / NM; pc < (m + 1) * NP / NM; ++pc) {
                            if (ks == 0) store_piece(pc, buf ^ 1);
                            else load_piece(pc);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        ++m;
                    }
                }
            }
        }
    }

    // ---- epilogue: affine + residual + activation (conv_common.hpp) -------------------------------------
    if constexpr (OUT16) {
        epilogue16<TM, TN, WTM, WTN, RES>(p, acc, m0, n0, wm, wn, lane);
    } else {
        ResidualRegs<TM, TN, RES> rv;  // fetched in one burst, ahead of every store
        load_residual<TM, TN, WTM, WTN, RES>(p, m0, n0, wm, wn, lane, rv);
        epilogue<TM, TN, WTM, WTN, RES>(p, acc, rv, m0, n0, wm, wn, lane);
    }
}

template <int BM, int BN, int WM, int WN, int PRODUCTS>
int launch(ConvParamsH p, bool generic, hipStream_t stream) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long long grid = tile_grid(p);
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16: grid too large");
    constexpr size_t lds = lds_bytes<BM, BN, PRODUCTS>();
    const int res = epilogue_variant(p, false);
    auto go = [&](auto kern) -> int {
        if (int rc2 = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv_f16")) return rc2;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, stream, p);
        return MRCNN_OK;
    };
    int rc;
    if (generic)
        rc = res == 0 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, true, 0>)
           : res == 1 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, true, 1>)
           : res == 2 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, true, 2>)
           : res == 3 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, true, 3>)
                      : go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, true, 4>);
    else
        rc = res == 0 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, false, 0>)
           : res == 1 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, false, 1>)
           : res == 2 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, false, 2>)
           : res == 3 ? go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, false, 3>)
                      : go(conv_igemm_f16<BM, BN, WM, WN, PRODUCTS, false, 4>);
    if (rc) return rc;
    return mrcnn::check_launch("conv_igemm_f16");
}

// fp16 storage variants (PRODUCTS = 1). in16/out16 choose the instantiation; only the combinations the pipeline uses
// exist: fp32 → fp16 (RES 0; the stem is GENERIC), fp16 → fp16 (RES 0, 1, 2, 4), fp16 → fp32 (RES 0, 3).
template <int BM, int BN, int WM, int WN>
int launch16(ConvParamsH p, bool generic, bool in16, bool out16, hipStream_t stream) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long long grid = tile_grid(p);
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16: grid too large");
    constexpr size_t lds = lds_bytes<BM, BN, 1>();
    const int res = epilogue_variant(p, false);
    auto go = [&](auto kern) -> int {
        if (int rc2 = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv_f16")) return rc2;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, stream, p);
        return MRCNN_OK;
    };
    int rc = MRCNN_ERR_UNSUPPORTED;
    if (!in16 && out16) {
        if (res != 0) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16: fp32 → fp16 supports no residual / sigmoid / scatter");
        rc = generic ? go(conv_igemm_f16<BM, BN, WM, WN, 1, true, 0, false, true>)
                     : go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 0, false, true>);
    } else if (in16 && out16) {
        if (generic || res == 3) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16: fp16 → fp16 needs Cin %% 32 == 0, no sigmoid");
        rc = res == 0 ? go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 0, true, true>)
           : res == 1 ? go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 1, true, true>)
           : res == 2 ? go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 2, true, true>)
                      : go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 4, true, true>);
    } else if (in16 && !out16) {
        if (generic || !(res == 0 || res == 3))
            return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv_f16: fp16 → fp32 needs Cin %% 32 == 0 and no residual / scatter");
        rc = res == 0 ? go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 0, true, false>)
                      : go(conv_igemm_f16<BM, BN, WM, WN, 1, false, 3, true, false>);
    }
    if (rc) return rc;
    return mrcnn::check_launch("conv_igemm_f16<io16>");
}

template <int PRODUCTS>
int dispatch(const ConvParamsH& p, bool generic, hipStream_t s) {
    if (p.Cout <= 64) return launch<256, 64, 4, 1, PRODUCTS>(p, generic, s);
    return launch<128, 128, 2, 2, PRODUCTS>(p, generic, s);
}

}  // namespace

static int run_conv_f16(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                        const void* w_hi, const void* w_lo, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                        int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                        const float* scale, const float* shift, const float* residual, int32_t res_div,
                        int32_t relu, int32_t products, int32_t out_mode, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && w_hi && y, "conv_f16: null pointer");
    MRCNN_REQUIRE(products == 1 || (products == 3 && w_lo), "conv_f16: products must be 1, or 3 with w_lo");
    ConvParamsH p;
    if (int rc = fill_common(p, "conv_f16", x, batch, height, width, cin, 8, cout, kh, kw, stride, pad_top, pad_left,
                             pad_bottom, pad_right, scale, shift, residual, res_div, relu, out_mode, y, 2))
        return rc;
    p.w_hi = static_cast<const _Float16*>(w_hi);
    p.w_lo = static_cast<const _Float16*>(w_lo);
    const bool generic = (cin % BK) != 0;
    hipStream_t s = mrcnn::as_stream(stream);
    return products == 3 ? dispatch<3>(p, generic, s) : dispatch<1>(p, generic, s);
}

static int run_conv_f16io(const void* x, int32_t x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                          const void* w_hi, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad_top,
                          int32_t pad_left, int32_t pad_bottom, int32_t pad_right, const float* scale, const float* shift,
                          const void* residual, int32_t res_div, int32_t relu, int32_t out_mode, void* y, int32_t y_f16,
                          mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && w_hi && y, "conv_f16io: null pointer");
    MRCNN_REQUIRE(x_f16 || y_f16, "conv_f16io: fp32 in and out is mrcnn_conv_bn_act_nhwc_f16mfma");
    MRCNN_REQUIRE(!y_f16 || cout % 2 == 0, "conv_f16io: an fp16 output needs an even Cout (channel pairs), got %d", cout);
    MRCNN_REQUIRE(!x_f16 || cin % 32 == 0, "conv_f16io: an fp16 input needs Cin %% 32 == 0, got %d", cin);
    MRCNN_REQUIRE(!(out_mode == 1 && (cout / 4) % 2 != 0), "conv_f16io: deconv scatter needs an even channel count");
    ConvParamsH p;
    if (int rc = fill_common(p, "conv_f16io", static_cast<const float*>(x), batch, height, width, cin, 8, cout, kh, kw,
                             stride, pad_top, pad_left, pad_bottom, pad_right, scale, shift,
                             static_cast<const float*>(residual), res_div, relu, out_mode, static_cast<float*>(y), 2))
        return rc;
    if (x_f16) p.x_bytes /= 2;
    if (y_f16) {  // the residual has the output's type
        p.y_bytes /= 2;
        p.r_bytes /= 2;
    }
    p.w_hi = static_cast<const _Float16*>(w_hi);
    p.w_lo = nullptr;
    const bool generic = (cin % BK) != 0;
    hipStream_t s = mrcnn::as_stream(stream);
    if (p.Cout <= 64) return launch16<256, 64, 4, 1>(p, generic, x_f16 != 0, y_f16 != 0, s);
    static const int big = [] { const char* e = mrcnn::tuning_env("MRCNN_F16_BIG"); return e ? atoi(e) : 0; }();
    if (big == 1 && p.Cout % 256 == 0 && p.K >= 1024) return launch16<256, 256, 2, 2>(p, generic, x_f16 != 0, y_f16 != 0, s);
    if (big == 2 && p.Cout % 128 == 0 && p.K >= 1024) return launch16<256, 128, 2, 2>(p, generic, x_f16 != 0, y_f16 != 0, s);
    return launch16<128, 128, 2, 2>(p, generic, x_f16 != 0, y_f16 != 0, s);
}

extern "C" int mrcnn_conv_bn_act_nhwc_f16io(const void* x, int32_t x_is_f16, int32_t batch, int32_t height,
                                            int32_t width, int32_t cin, const void* w_f16, int32_t cout, int32_t kh,
                                            int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                                            int32_t pad_bottom, int32_t pad_right, const float* scale,
                                            const float* shift, const void* residual, int32_t res_div,
                                            int32_t activation, void* y, int32_t y_is_f16, mrcnn_stream_t stream) {
    return run_conv_f16io(x, x_is_f16, batch, height, width, cin, w_f16, cout, kh, kw, stride, pad_top, pad_left,
                          pad_bottom, pad_right, scale, shift, residual, res_div, activation, 0, y, y_is_f16, stream);
}

extern "C" int mrcnn_deconv2x2_bias_act_nhwc_f16io(const void* x_f16, int32_t batch, int32_t height, int32_t width,
                                                   int32_t cin, const void* w_f16, int32_t cout, const float* bias4,
                                                   int32_t activation, void* y_f16, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(cout >= 1, "deconv2x2: cout=%d", cout);
    MRCNN_REQUIRE(activation == 0 || activation == 1, "deconv2x2: activation must be 0 or 1");
    return run_conv_f16io(x_f16, 1, batch, height, width, cin, w_f16, 4 * cout, 1, 1, 1, 0, 0, 0, 0, nullptr, bias4,
                          nullptr, 1, activation, 1, y_f16, 1, stream);
}

extern "C" int mrcnn_conv_bn_act_nhwc_f16mfma(const float* x, int32_t batch, int32_t height, int32_t width,
                                              int32_t cin, const void* w_hi, const void* w_lo, int32_t cout,
                                              int32_t kh, int32_t kw, int32_t stride, int32_t pad_top,
                                              int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                                              const float* scale, const float* shift, const float* residual,
                                              int32_t res_div, int32_t activation, int32_t products, float* y,
                                              mrcnn_stream_t stream) {
    return run_conv_f16(x, batch, height, width, cin, w_hi, w_lo, cout, kh, kw, stride, pad_top, pad_left,
                        pad_bottom, pad_right, scale, shift, residual, res_div, activation, products, 0, y, stream);
}

extern "C" int mrcnn_deconv2x2_bias_act_nhwc_f16mfma(const float* x, int32_t batch, int32_t height, int32_t width,
                                                     int32_t cin, const void* w_hi, const void* w_lo,
                                                     int32_t cout, const float* bias4, int32_t activation,
                                                     int32_t products, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(cout >= 1, "deconv2x2: cout=%d", cout);
    return run_conv_f16(x, batch, height, width, cin, w_hi, w_lo, 4 * cout, 1, 1, 1, 0, 0, 0, 0, nullptr, bias4,
                        nullptr, 1, activation, products, 1, y, stream);
}
