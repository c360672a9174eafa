// Pieces shared by the two implicit-GEMM convolution kernels (conv.hip: exact-fp32 MFMA, conv_f16.hip:
// fp16-operand MFMA): the problem description, tile order, im2col row bookkeeping and the epilogue.
#pragma once
#include "common.hpp"

namespace mrcnn_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond every buffer descriptor's num_records

// row offset + column offset where either may be OOB: a saturating add (one v_add_u32 ... clamp) keeps the sum out of range
// — the or / compare / add / select it replaces are VALU instructions of an epilogue that runs beside the other workgroup's
// MFMAs on the same SIMD, and VALU work does not co-execute with this MFMA.
// The saturated sum 0xFFFFFFFF is clamped back to OOB (0xFFFFFFF0): 8-byte accesses add their size to the offset in the range
// check, and offset + 8 must not wrap past 2^32 into the buffer (one v_min_u32; the 4-byte paths are unaffected either way).
__device__ __forceinline__ unsigned oob_add(unsigned a, unsigned b) {
    const unsigned s = __builtin_elementwise_add_sat(a, b);
    return s < OOB ? s : OOB;
}

// Everything about the convolution that does not depend on how the weights are stored.
struct ConvCommon {
    const float* x;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW;
    int M, K;          // GEMM sizes: M = B*OH*OW, K = KH*KW*Cin
    int res_div, act;  // act: 0 none, 1 ReLU (2 = sigmoid is the compile-time epilogue variant 3)
    int out_mode;      // 0: y[m][n]; 1: 2x2 stride-2 transposed-conv scatter, n = (dy*2+dx)*Cout/4 + co;
                       // 2: k-blocked y[n/8][m][n%8] (what the Winograd kernel reads; variants 0-3 only)
    int ow_shift, ohw_shift;  // log2(OW), log2(OH*OW) when both are powers of two, else -1 (pixel decode without division)
    int res_kblocked;  // the residual tensor is k-blocked [Cout/8][residual pixels][8] instead of NHWC
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes, y_bytes, r_bytes;  // buffer-descriptor ranges (< 4 GiB each)
    // Optional row groups (the heads' GEMMs over [image][RoI slot] rows, model.py:1366-1374: an image's rois tensor has only the
    // proposals that survived NMS — here every image has rows_per_group slots of which the first row_counts[image] hold one):
    // an M tile none of whose rows holds a RoI is skipped (its output rows are left untouched). Device memory; NULL = all rows.
    const int* row_counts;
    int rows_per_group;
};

// Does the M tile [m0, m0 + BM) hold a valid row? Valid rows are a PREFIX of every group, so only the group of the tile's
// first row (from that row on) and the group of its last row (from slot 0 on) can contribute; groups in between are whole.
__device__ __forceinline__ bool tile_has_rows(const ConvCommon& p, int m0, int BM) {
    const int last = min(m0 + BM, p.M) - 1;
    const int g0 = m0 / p.rows_per_group, g1 = last / p.rows_per_group;
    if (m0 - g0 * p.rows_per_group < p.row_counts[g0]) return true;
    for (int g = g0 + 1; g <= g1; ++g)
        if (p.row_counts[g] > 0) return true;
    return false;
}

// Epilogue variants (template parameter RES of the kernels):
//   0 plain, 1 residual of the output's size, 2 residual at half size (FPN nearest-upsample-add),
//   3 sigmoid (no residual), 4 2x2 stride-2 transposed-conv scatter (no residual), 5 fused 1x1 heads (conv.hip)
inline int epilogue_variant(const ConvCommon& p, bool fused_heads) {
    return fused_heads ? 5 : p.out_mode == 1 ? 4 : (p.act == 2 ? 3 : (p.residual ? p.res_div : 0));
}

// Output pixel m -> (image b, row oy, column ox). Every layer of the 1024^2 / 832x1344 pyramids has power-of-two or small
// sizes; with powers of two this is two shifts instead of two integer divisions (~40 VALU instructions each, and the
// half-size-residual epilogue decodes 16 rows per accumulator tile).
__device__ __forceinline__ void decode_pixel(const ConvCommon& p, int m, int& b, int& oy, int& ox) {
    if (p.ow_shift >= 0) {
        b = m >> p.ohw_shift;
        const int rem = m & ((1 << p.ohw_shift) - 1);
        oy = rem >> p.ow_shift;
        ox = rem & ((1 << p.ow_shift) - 1);
    } else {
        const int ohw = p.OH * p.OW;
        b = m / ohw;
        const int rem = m - b * ohw;
        oy = rem / p.OW;
        ox = rem - oy * p.OW;
    }
}

// XCD-aware tile order: XCD x (= blockIdx % 8) owns the M tiles [x*tiles_m/8, (x+1)*tiles_m/8) and walks N
// fastest, so an activation tile is fetched into exactly one XCD's L2 and reused for every N tile / tap.
__device__ __forceinline__ bool tile_origin(const ConvCommon& p, int BM, int BN, int& m0, int& n0, int& nt) {
    const int t = blockIdx.x;
    const int xcd = t & 7, seq = t >> 3;
    const int mt_lo = (xcd * p.tiles_m) >> 3, mt_hi = ((xcd + 1) * p.tiles_m) >> 3;
    const int mt = mt_lo + seq / p.tiles_n;
    nt = seq % p.tiles_n;
    if (mt >= mt_hi) return false;
    m0 = mt * BM;
    n0 = nt * BN;
    return true;
}
inline long long tile_grid(const ConvCommon& p) { return 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n; }

// im2col bookkeeping of the PA tile rows a thread stages: element offset of pixel (b, iy0, ix0) and the
// (iy0, ix0) themselves for the zero-padding predicate; rows beyond M can never be in range.
template <int PA, int RPP>
__device__ __forceinline__ void row_setup(const ConvCommon& p, int m0, int r0, int (&a_off)[PA], int (&a_iy)[PA],
                                          int (&a_ix)[PA]) {
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + r0 + RPP * i;
        if (m < p.M) {
            int b, oy, ox;
            decode_pixel(p, m, b, oy, ox);
            a_iy[i] = oy * p.stride - p.pad_t;
            a_ix[i] = ox * p.stride - p.pad_l;
            a_off[i] = ((b * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin;
        } else {
            a_iy[i] = -(1 << 24);
            a_ix[i] = 0;
            a_off[i] = 0;
        }
    }
}

// The residual operand of one wave's output tile, held in registers: rv[i][jn][r] pairs with acc[i][jn][r].
template <int TM, int TN, int RES>
struct ResidualRegs {
    static constexpr bool USED = RES == 1 || RES == 2;
    float v[USED ? TM : 1][USED ? TN : 1][16];
};

// Issue ALL of the wave's residual loads (TM*TN*16 per lane, branch-free through a buffer descriptor) in one burst at
// the top of the epilogue, ahead of every store: the 1x1 expansion layers are short-K and HBM-bound on residual read +
// output write, and a load-16 / store-16 ping-pong keeps 4x fewer bytes in flight (C2 conv3: 0.29 -> 0.26 ms).
// (Issuing them before the main loop was measured too: no further gain, and it costs registers there.)
template <int TM, int TN, int WTM, int WTN, int RES>
__device__ __forceinline__ void load_residual(const ConvCommon& p, int m0, int n0, int wm, int wn, int lane,
                                              ResidualRegs<TM, TN, RES>& rv) {
    if constexpr (RES == 1 || RES == 2) {
        const int ln = lane & 31, lh = lane >> 5;
        const int ohw = p.OH * p.OW;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.residual), 0, p.r_bytes, 0x00020000);
        const int rh = p.OH >> 1, rw = p.OW >> 1;
        const unsigned row_bytes = p.res_kblocked ? 32u : static_cast<unsigned>(p.Cout) * 4u;
        const unsigned rplane = static_cast<unsigned>(RES == 1 ? p.M : p.M >> 2) * 32u;  // bytes per 8-channel plane
        unsigned ncol[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + wn * WTN + jn * 32 + ln;
            ncol[jn] = n >= p.Cout ? OOB
                       : p.res_kblocked ? static_cast<unsigned>(n >> 3) * rplane + static_cast<unsigned>(n & 7) * 4u
                                        : static_cast<unsigned>(n) * 4u;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WTM + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const bool ok = m < p.M;
                unsigned rrow;
                if constexpr (RES == 1) {
                    rrow = ok ? static_cast<unsigned>(m) * row_bytes : OOB;
                } else {  // residual at half size: (oy/2, ox/2) (FPN nearest-neighbour upsample + add)
                    const int mm = ok ? m : 0;
                    int b, oy, ox;
                    decode_pixel(p, mm, b, oy, ox);
                    rrow = ok ? static_cast<unsigned>((b * rh + (oy >> 1)) * rw + (ox >> 1)) * row_bytes : OOB;
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    // OOB in either term must stay OOB
                    const unsigned off = oob_add(rrow, ncol[jn]);
                    rv.v[i][jn][r] =
                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, static_cast<int>(off), 0, 0));
                }
            }
        }
    }
}

// The lane's per-channel affine, fetched by the caller BEFORE its main loop when it can spare the registers: loaded at
// the top of the epilogue the two loads sit in front of the first store with their whole memory latency exposed, once per
// tile.
template <int TN>
struct AffineRegs {
    float sc[TN], sh[TN];
};
template <int TN, int WTN>
__device__ __forceinline__ void load_affine(const ConvCommon& p, int n0, int wn, int lane, AffineRegs<TN>& a) {
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WTN + jn * 32 + (lane & 31);
        const bool n_ok = n < p.Cout;
        a.sc[jn] = (n_ok && p.scale) ? p.scale[n] : 1.0f;
        a.sh[jn] = (n_ok && p.shift) ? p.shift[n] : 0.0f;
    }
}

// Epilogue: act(acc*scale[c] + shift[c] (+ residual)) → y, 128-byte channel runs per half-wave. Branch-free:
// stores go through a buffer descriptor; out-of-tile rows/channels get an out-of-range offset (dropped).
template <int TM, int TN, int WTM, int WTN, int RES>
__device__ __forceinline__ void epilogue(const ConvCommon& p, f32x16 (&acc)[TM][TN],
                                         const ResidualRegs<TM, TN, RES>& rv, int m0, int n0, int wm, int wn,
                                         int lane, const AffineRegs<TN>* pre = nullptr) {
    static_assert(RES >= 0 && RES <= 4, "variant 5 (fused heads) lives in conv.hip");
    const int ln = lane & 31, lh = lane >> 5;
    const int ohw = p.OH * p.OW;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    float sc[TN], sh[TN];
    unsigned ncol[TN];  // byte offset of the lane's channel, or OOB
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WTN + jn * 32 + ln;
        const bool n_ok = n < p.Cout;
        if (pre) {
            sc[jn] = pre->sc[jn];
            sh[jn] = pre->sh[jn];
        } else {
            sc[jn] = (n_ok && p.scale) ? p.scale[n] : 1.0f;
            sh[jn] = (n_ok && p.shift) ? p.shift[n] : 0.0f;
        }
        if constexpr (RES != 4) {
            ncol[jn] = !n_ok ? OOB
                       : p.out_mode == 2 ? static_cast<unsigned>(n >> 3) * (static_cast<unsigned>(p.M) * 32u) +
                                               static_cast<unsigned>(n & 7) * 4u
                                         : static_cast<unsigned>(n) * 4u;
        } else {  // n = (dy*2 + dx)*cq + co  →  pixel (2i+dy, 2j+dx), channel co of the [B][2*OH][2*OW][cq] output
            const int cq = p.Cout >> 2, q = n / cq, co = n - q * cq;
            ncol[jn] = n_ok ? static_cast<unsigned>(((q >> 1) * 2 * p.OW + (q & 1)) * cq + co) * 4u : OOB;
        }
    }
    const unsigned row_bytes = p.out_mode == 2 ? 32u : static_cast<unsigned>(p.Cout) * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WTM + i * 32 + 4 * lh;  // rows mb + (r&3) + 8*(r>>2)
        unsigned yrow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            const bool ok = m < p.M;
            if constexpr (RES != 4) {
                yrow[r] = ok ? static_cast<unsigned>(m) * row_bytes : OOB;
            } else {
                const int mm = ok ? m : 0;
                int b, oy, ox;
                decode_pixel(p, mm, b, oy, ox);
                yrow[r] = ok ? static_cast<unsigned>((b * 2 * p.OH + 2 * oy) * (2 * p.OW) + 2 * ox) * (row_bytes >> 2)
                             : OOB;
            }
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][jn][r] * sc[jn] + sh[jn];
                if constexpr (RES == 1 || RES == 2) v += rv.v[i][jn][r];
                if constexpr (RES == 3) v = 1.0f / (1.0f + expf(-v));
                else if (p.act) asm("v_max_f32 %0, 0, %0" : "+v"(v));   // NaN -> 0 like (v > 0 ? v : 0), one instruction
                const unsigned off = oob_add(yrow[r], ncol[jn]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(off), 0, 0);
            }
        }
    }
}

// ---- the same epilogue with 8-byte accesses (Cout even) -----------------------------------------------------------
// A lane holds ONE channel of 16 rows, so the plain epilogue moves 4 bytes per lane and instruction — 64 + 64 vector-memory
// instructions per 32 x 32 block with a residual, and in-kernel stamps put the epilogue at a quarter of a tile's time on the
// 1x1 expansion layers. Lanes 2j and 2j+1 (adjacent channels) pair up instead: the even lane ends up with both channels of
// rows 0..7 of its 16, the odd lane with both channels of rows 8..15 (one DPP swap per value pair), and each moves 8 bytes
// per instruction: half the loads and stores. rv holds the pair layout: v[i][jn][2k], [2k+1] = the two channels of the lane's
// k-th row.
__device__ __forceinline__ float swap_lane_pair(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
}

template <int TM, int TN, int WTM, int WTN, int RES>
__device__ __forceinline__ void load_residual_pairs(const ConvCommon& p, int m0, int n0, int wm, int wn, int lane,
                                                    ResidualRegs<TM, TN, RES>& rv) {
    if constexpr (RES == 1 || RES == 2) {
        const int ln = lane & 31, lh = lane >> 5, odd = ln & 1;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.residual), 0, p.r_bytes, 0x00020000);
        const int rh = p.OH >> 1, rw = p.OW >> 1;
        const unsigned row_bytes = p.res_kblocked ? 32u : static_cast<unsigned>(p.Cout) * 4u;
        const unsigned rplane = static_cast<unsigned>(RES == 1 ? p.M : p.M >> 2) * 32u;  // bytes per 8-channel plane
        unsigned ncol[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + wn * WTN + jn * 32 + ln, np = n & ~1;
            ncol[jn] = n >= p.Cout ? OOB
                       : p.res_kblocked ? static_cast<unsigned>(np >> 3) * rplane + static_cast<unsigned>(np & 7) * 4u
                                        : static_cast<unsigned>(np) * 4u;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WTM + i * 32 + 4 * lh + 16 * odd;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int m = mb + (k & 3) + 8 * (k >> 2);
                const bool ok = m < p.M;
                unsigned rrow;
                if constexpr (RES == 1) {
                    rrow = ok ? static_cast<unsigned>(m) * row_bytes : OOB;
                } else {  // residual at half size: (oy/2, ox/2) (FPN nearest-neighbour upsample + add)
                    const int mm = ok ? m : 0;
                    int b, oy, ox;
                    decode_pixel(p, mm, b, oy, ox);
                    rrow = ok ? static_cast<unsigned>((b * rh + (oy >> 1)) * rw + (ox >> 1)) * row_bytes : OOB;
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const unsigned off = oob_add(rrow, ncol[jn]);
                    const auto w = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, static_cast<int>(off), 0, 0);
                    rv.v[i][jn][2 * k] = __uint_as_float(w[0]);
                    rv.v[i][jn][2 * k + 1] = __uint_as_float(w[1]);
                }
            }
        }
    }
}

template <int TM, int TN, int WTM, int WTN, int RES>
__device__ __forceinline__ void epilogue_pairs(const ConvCommon& p, f32x16 (&acc)[TM][TN],
                                               const ResidualRegs<TM, TN, RES>& rv, int m0, int n0, int wm, int wn,
                                               int lane, const AffineRegs<TN>* pre = nullptr) {
    static_assert(RES == 0 || RES == 1 || RES == 2 || RES == 4, "sigmoid (odd channel counts) keeps the 4-byte epilogue");
    const int ln = lane & 31, lh = lane >> 5;
    const bool odd = ln & 1;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    float sc[TN], sh[TN];
    unsigned ncol[TN];  // byte offset of the lane pair's first channel, or OOB
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WTN + jn * 32 + ln, np = n & ~1;
        const bool n_ok = n < p.Cout;  // Cout is even: both channels of a pair are in or out together
        if (pre) {
            sc[jn] = pre->sc[jn];
            sh[jn] = pre->sh[jn];
        } else {
            sc[jn] = (n_ok && p.scale) ? p.scale[n] : 1.0f;
            sh[jn] = (n_ok && p.shift) ? p.shift[n] : 0.0f;
        }
        if constexpr (RES != 4) {
            ncol[jn] = !n_ok ? OOB
                       : p.out_mode == 2 ? static_cast<unsigned>(np >> 3) * (static_cast<unsigned>(p.M) * 32u) +
                                               static_cast<unsigned>(np & 7) * 4u
                                         : static_cast<unsigned>(np) * 4u;
        } else {  // n = (dy*2 + dx)*cq + co  →  pixel (2i+dy, 2j+dx), channel co of the [B][2*OH][2*OW][cq] output (cq even)
            const int cq = p.Cout >> 2, q = np / cq, co = np - q * cq;
            ncol[jn] = n_ok ? static_cast<unsigned>(((q >> 1) * 2 * p.OW + (q & 1)) * cq + co) * 4u : OOB;
        }
    }
    const unsigned row_bytes = p.out_mode == 2 ? 32u : static_cast<unsigned>(p.Cout) * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WTM + i * 32 + 4 * lh + (odd ? 16 : 0);  // the lane's rows: mb + (k&3) + 8*(k>>2)
        unsigned yrow[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = mb + (k & 3) + 8 * (k >> 2);
            const bool ok = m < p.M;
            if constexpr (RES != 4) {
                yrow[k] = ok ? static_cast<unsigned>(m) * row_bytes : OOB;
            } else {
                const int mm = ok ? m : 0;
                int b, oy, ox;
                decode_pixel(p, mm, b, oy, ox);
                yrow[k] = ok ? static_cast<unsigned>((b * 2 * p.OH + 2 * oy) * (2 * p.OW) + 2 * ox) * (row_bytes >> 2)
                             : OOB;
            }
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float va = acc[i][jn][k] * sc[jn] + sh[jn];          // own channel, row k
                const float vb = acc[i][jn][k + 8] * sc[jn] + sh[jn];      // own channel, row k + 8
                const float got = swap_lane_pair(odd ? va : vb);           // the partner's channel at the row this lane keeps
                float lo = odd ? got : va, hi = odd ? vb : got;            // channels np, np + 1
                if constexpr (RES == 1 || RES == 2) {
                    lo += rv.v[i][jn][2 * k];
                    hi += rv.v[i][jn][2 * k + 1];
                }
                if (p.act) asm("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(lo), "+v"(hi));
                const unsigned off = oob_add(yrow[k], ncol[jn]);
                typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(lo), __float_as_uint(hi)}, y_rsrc,
                                                      static_cast<int>(off), 0, 0);
            }
        }
    }
}

// Host-side validation + fill of the common part; returns MRCNN_OK or sets the error message.
inline int fill_common(ConvCommon& p, const char* who, const float* x, int batch, int height, int width, int cin,
                       int cin_multiple, int cout, int kh, int kw, int stride, int pad_top, int pad_left,
                       int pad_bottom, int pad_right, const float* scale, const float* shift, const float* residual,
                       int res_div, int activation, int out_mode, float* y, int weight_elem_bytes,
                       int res_kblocked = 0) {
    if (!(activation >= 0 && activation <= 2))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: activation must be 0 (none), 1 (ReLU) or 2 (sigmoid)", who);
    if (residual && (activation == 2 || out_mode == 1))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: a residual cannot be combined with sigmoid or the deconv scatter", who);
    if (out_mode == 1 && activation == 2)
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: deconv scatter supports activation 0 or 1", who);
    if (out_mode == 2 && cout % 8 != 0)
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: a k-blocked output needs Cout %% 8 == 0", who);
    if (!(batch >= 1 && height >= 1 && width >= 1 && cin >= cin_multiple && cin % cin_multiple == 0 && cout >= 1))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: bad shape B=%d H=%d W=%d Cin=%d (Cin %% %d == 0 required) Cout=%d",
                           who, batch, height, width, cin, cin_multiple, cout);
    if (!(kh >= 1 && kw >= 1 && stride >= 1 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: bad kernel/stride/pad", who);
    if (!(residual == nullptr || res_div == 1 || res_div == 2))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: res_div must be 1 or 2", who);
    p.x = x; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.row_counts = nullptr; p.rows_per_group = 0;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw;
    p.stride = stride; p.pad_t = pad_top; p.pad_l = pad_left;
    p.OH = (height + pad_top + pad_bottom - kh) / stride + 1;
    p.OW = (width + pad_left + pad_right - kw) / stride + 1;
    if (!(p.OH >= 1 && p.OW >= 1)) return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: empty output", who);
    if (!(residual == nullptr || res_div == 1 || (p.OH % 2 == 0 && p.OW % 2 == 0)))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: res_div=2 needs even output size", who);
    const long long M = 1LL * batch * p.OH * p.OW;
    const long long K = 1LL * kh * kw * cin;
    if (!(1LL * batch * height * width * cin < (1LL << 30) && M * cout < (1LL << 30) && K * cout < (1LL << 30) &&
          M < (1LL << 31)))
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT,
                           "%s: tensor too large (each tensor < 2^30 elements: 32-bit buffer byte offsets)", who);
    p.M = static_cast<int>(M);
    p.K = static_cast<int>(K);
    {
        auto log2_exact = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
        const int sw = log2_exact(p.OW), sh = log2_exact(p.OH);
        p.ow_shift = (sw >= 0 && sh >= 0) ? sw : -1;
        p.ohw_shift = (sw >= 0 && sh >= 0) ? sw + sh : -1;
    }
    p.res_div = residual ? res_div : 1;
    p.act = activation;
    p.out_mode = out_mode;
    p.res_kblocked = (residual && res_kblocked) ? 1 : 0;
    if (p.res_kblocked && cout % 8 != 0)
        return mrcnn::fail(MRCNN_ERR_INVALID_ARGUMENT, "%s: a k-blocked residual needs Cout %% 8 == 0", who);
    p.x_bytes = static_cast<unsigned>(4LL * batch * height * width * cin);
    p.w_bytes = static_cast<unsigned>(1LL * weight_elem_bytes * K * cout);
    p.y_bytes = static_cast<unsigned>(4LL * M * cout);
    p.r_bytes = residual ? static_cast<unsigned>(4LL * M * cout / (p.res_div * p.res_div)) : 0u;
    return MRCNN_OK;
}

}  // namespace mrcnn_conv
