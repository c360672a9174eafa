// Fused convolution + per-channel affine (bias/BatchNorm) + residual + ReLU as an fp32 MFMA implicit GEMM
// for gfx950, channels-last. The compute core of Bottleneck.forward (model.py:190-211), the stem
// (model.py:223-226), the FPN lateral/smoothing convs (model.py:145-157), RPN and head convs.
// The reference has no kernel for this (it chains nn.Conv2d / BatchNorm2d / ReLU / F.pad modules); this is
// new work, designed for CDNA4:
//   GEMM view   M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin with k = (ky*KW + kx)*Cin + ci.
//   MFMA        v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD.
//               A operand = pixels (rows), B operand = weights (cols): an accumulator register holds one
//               output channel per lane, so epilogue loads/stores are 128-byte channel runs.
//   LDS         A[BM][32+4] and B[BN][32+4] fp32 tiles, double buffered. Each lane fetches its operand
//               with ONE ds_read_b128 per 8-deep k chunk: lane half h reads k = 8j+4h..+3 and MFMA step s
//               contracts k = 8j+s (lanes 0-31) with k = 8j+4+s (lanes 32-63) — a k permutation applied
//               identically to A and B. The +4 pad makes those reads bank-conflict free.
//   im2col      on the fly from NHWC: a thread owns fixed tile rows and a fixed 16-byte k slot; zero
//               padding (SamePad2d, model.py:64-87) is a predicate, never a padded copy.
//   pipeline    global -> registers for tile t+1 issued before the MFMAs of tile t, written to the other
//               LDS buffer after them; one barrier per k tile.
//   XCD         consecutive tiles of one XCD (blockIdx % 8) walk N first inside a contiguous M range, so
//               an activation tile is fetched into exactly one XCD's L2 and reused for every N tile/tap.
//   epilogue    y = relu(acc*scale[c] + shift[c] + residual), residual optionally read at (oy/2, ox/2)
//               (FPN nearest-neighbour upsample + add, model.py:150-152).
#include "conv_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace mrcnn_conv;

struct ConvParams : ConvCommon {
    const float* w;  // [Cout][KH][KW][Cin]
    // RES == 5 (fused 1x1 heads): y is the per-N-tile partial [tiles_n][M][head_n]; w_head is [32][Cout]
    const float* w_head;
    int head_n;
};

template <int BM, int BN, int BK>
constexpr size_t conv_lds_bytes() {
    return sizeof(float) * 2 * (BM + BN) * (BK == 16 ? BK : BK + 4);   // LDS_STRIDE of the kernel
}

// MODE 0: Cin % 32 == 0, any kernel size / stride / padding: a k tile lies inside one tap.
// MODE 1 (GENERIC): Cin % 32 != 0 (stem, Cin = 4): a k tile may straddle taps, each 16-byte slot decodes its own tap.
// MODE 2 (POINTWISE): 1x1, no padding, Cin % 32 == 0 — every conv the direct kernel still runs in the Winograd mode
//   except the stem. No address arithmetic in the loop: a thread's byte offsets are fixed for the whole tile (out of
//   range for rows beyond M), the k tile is the scalar soffset of the buffer loads. (VALU instructions do not
//   co-execute with this MFMA — tools/mfma_valu_probe.hip — so each one removed is four cycles per wave and k tile.)
// SPLIT (round 5): the sum over K runs as TWO chains — the 32-wide k blocks alternate between two accumulator sets, added once
// at the end — for the long-K layers (K >= 1024: the Bottleneck conv1 layers of C4 / C5, the C5 downsample, the P4 / P5 laterals,
// the classifier's GEMMs). A float64 evaluation of every conv unit of the trunk on identical inputs (tools/fp64_truth.py
// --attribute, profiles/r05_fp64_attribution.jsonl) put exactly those at the top of the HIP path's excess over torch-CPU's
// blocked sums (rms 0.7 - 1.6 ulps of the output range against 0.2 - 0.4): the error of a sequential fp32 sum grows with its
// length, and the second accumulator set is free — 151 + 64 registers stay inside the 256 of two workgroups per CU, the MFMAs
// are the same, one packed add per accumulator register at the end. Which chain a k block belongs to depends on its position in K
// only ((k0 >> 5) & 1), never on the tile (BK = 16 tiles give two consecutive k tiles to a chain), so every tile the launcher may
// pick for a layer — by the batch — produces the same bits.
template <int BM, int BN, int WM, int WN, int BK, int MODE, int RES, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvParams p) {
    constexpr bool GENERIC = MODE == 1, PW = MODE == 2;
    // LDS row pitch in floats. BK = 32: 36 — the +4 pad makes the fragment reads (16 consecutive rows, one 16-byte chunk each)
    // conflict-free, and a 16-lane ds_write_b128 pass (two rows of eight chunks) overlaps in one chunk only. BK = 16: a pad
    // cannot serve both — a write pass is FOUR rows of four chunks, and no odd chunk pitch keeps those sixteen chunks apart
    // (pitch 20: SQ_LDS_BANK_CONFLICT = 0.34 of the LDS-active cycles on the BK = 16 instantiations,
    // profiles/r03_cache_lds_counters.json) — so the rows are unpadded (pitch 16) and chunk c of row r sits at c ^ ((r >> 2) & 3):
    // sixteen consecutive rows of one logical chunk, and four consecutive rows of all four chunks, both cover the 64 banks once.
    constexpr bool SWZ = BK == 16;
    constexpr int LDS_STRIDE = SWZ ? BK : BK + 4;
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    constexpr int TPR = BK / 4;                  // threads (16-byte slots) per tile row
    constexpr int RPP = 256 / TPR;               // tile rows covered per pass of the 256 threads
    constexpr int PA = BM / RPP, PB = BN / RPP;  // 16-byte slots per thread per k tile
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_STRIDE;

    int m0, n0, nt;
    if (!tile_origin(p, BM, BN, m0, n0, nt)) return;  // XCD-aware tile order
    if (p.row_counts && !tile_has_rows(p, m0, BM)) return;   // an M tile of empty RoI slots (uniform: scalar loads)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kq = tid % TPR, r0 = tid / TPR;
    // scale / shift of the lane's channels in flight under the whole main loop — except on the residual layers, short-K
    // and HBM-bound, whose residual burst they would only delay (measured: C4/C5 1x1 layers -3..5 %, C2 conv3 +10 %)
    constexpr bool EARLY_AFFINE = RES == 0 || RES == 3 || RES == 4;
    AffineRegs<TN> affine;
    if constexpr (EARLY_AFFINE) load_affine<TN, WTN>(p, n0, wn, lane, affine);

    // ---- per-thread row bookkeeping for the im2col gather ---------------------------------------------
    int a_off[PA], a_iy[PA], a_ix[PA];
    row_setup<PA, RPP>(p, m0, r0, a_off, a_iy, a_ix);
    int b_off[PB];
    bool b_ok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + r0 + RPP * i;
        b_ok[i] = n < p.Cout;
        b_off[i] = (b_ok[i] ? n : 0) * p.K + kq * 4;
    }

    unsigned a_voff[PW ? PA : 1], b_voff[PW ? PB : 1];  // POINTWISE: fixed byte offsets (k0 = 0), OOB where zero
    if constexpr (PW) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            a_voff[i] = (m0 + r0 + RPP * i) < p.M ? static_cast<unsigned>(a_off[i] + kq * 4) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < PB; ++i) b_voff[i] = b_ok[i] ? static_cast<unsigned>(b_off[i]) * 4u : OOB;
    }
    float4 ra[PA], rb[PB];
    // Predication without branches: loads go through raw buffer descriptors (hardware range check); an
    // out-of-image tap / out-of-range row gets a byte offset beyond num_records and returns zeros. The
    // compiler therefore issues every load of a tile back to back and nothing waits on them until
    // store_tile, after the MFMAs of the current tile.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                           __uint_as_float(v.w));
    };
    // ---- staging in pieces ---------------------------------------------------------------------------
    // One k tile = PA + PB pieces (one 16-byte slot each). Loads and LDS writes are issued piece by piece
    // BETWEEN the MFMAs of the main loop (below), so their issue cycles hide in the 64-cycle shadow of the
    // wave's own MFMAs instead of forming an MFMA-free block around the barrier.
    int t_c0 = 0, t_ky = 0, t_kx = 0, t_k0 = 0;  // (tap, channel, k) of the NEXT k tile to load — wave-uniform
    int cur_tap_off = 0, cur_ky = 0, cur_kx = 0, cur_k0 = 0;
    bool cur_kvalid = true;
    // GENERIC only: per-thread (channel, ky, kx) of the slot k = k0 + 4*kq, no division in the loop
    int n_c = 0, n_ky = 0, n_kx = 0;      // state of the NEXT k tile
    int g_ky = 0, g_kx = 0, g_tap_off = 0;  // state of the k tile being loaded
    bool g_kin = true;
    if constexpr (GENERIC) {
        const int tap = (kq * 4) / p.Cin;
        n_c = kq * 4 - tap * p.Cin;
        n_ky = tap / p.KW;
        n_kx = tap - n_ky * p.KW;
    }
    auto begin_load = [&]() {  // scalar bookkeeping for the k tile about to be loaded
        cur_k0 = t_k0;
        cur_kvalid = t_k0 < p.K;
        if constexpr (GENERIC) {
            g_ky = n_ky;
            g_kx = n_kx;
            g_tap_off = (n_ky * p.W + n_kx) * p.Cin + n_c;
            g_kin = (t_k0 + kq * 4) < p.K;
            n_c += BK;
            while (n_c >= p.Cin) {  // at most BK / Cin + 1 rounds
                n_c -= p.Cin;
                if (++n_kx == p.KW) { n_kx = 0; ++n_ky; }
            }
        }
        if constexpr (!GENERIC) {
            cur_ky = t_ky;
            cur_kx = t_kx;
            cur_tap_off = (t_ky * p.W + t_kx) * p.Cin + t_c0 + kq * 4;
            t_c0 += BK;
            if (t_c0 >= p.Cin) {
                t_c0 = 0;
                if (++t_kx == p.KW) { t_kx = 0; ++t_ky; }
            }
        }
        t_k0 += BK;
    };
    auto load_piece = [&](int pc) {  // pc < PA: activation slot pc; else weight slot pc - PA
        if constexpr (PW) {
            // past the end of K the last k tile is loaded again (into a buffer nobody reads): soffset is not range-checked
            const int kk = cur_k0 < p.K ? cur_k0 : p.K - BK;
            const bool isa = pc < PA;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(isa ? x_rsrc : w_rsrc,
                                                                  static_cast<int>(isa ? a_voff[isa ? pc : 0] : b_voff[isa ? 0 : pc - PA]),
                                                                  kk * 4, 0);
            const float4 f = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                                         __uint_as_float(v.w));
            if (isa) ra[isa ? pc : 0] = f;
            else rb[isa ? 0 : pc - PA] = f;
            return;
        }
        if (pc < PA) {
            const int i = pc;
            if constexpr (!GENERIC) {
                const bool ok = cur_kvalid &&
                                static_cast<unsigned>(a_iy[i] + cur_ky) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(a_ix[i] + cur_kx) < static_cast<unsigned>(p.W);
                ra[i] = bload(x_rsrc, ok ? static_cast<unsigned>(a_off[i] + cur_tap_off) * 4u : OOB);
            } else {  // this thread's slot has its own (tap, channel), advanced incrementally in begin_load
                const bool ok = g_kin &&
                                static_cast<unsigned>(a_iy[i] + g_ky) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(a_ix[i] + g_kx) < static_cast<unsigned>(p.W);
                ra[i] = bload(x_rsrc, ok ? static_cast<unsigned>(a_off[i] + g_tap_off) * 4u : OOB);
            }
        } else {
            const int i = pc - PA;
            bool ok = b_ok[i] && cur_kvalid;
            if constexpr (GENERIC) ok = ok && (cur_k0 + kq * 4 < p.K);
            rb[i] = bload(w_rsrc, ok ? static_cast<unsigned>(b_off[i] + cur_k0) * 4u : OOB);
        }
    };
    // (RPP is a multiple of 4 * 4 rows whenever SWZ: the swizzle term of a thread's rows is that of r0)
    static_assert(!SWZ || RPP % 16 == 0, "swizzle term must not depend on the piece");
    const int kqs = SWZ ? (kq ^ ((r0 >> 2) & 3)) : kq;   // physical 16-byte chunk of this thread's staging slot
    auto store_piece = [&](int pc, int buf) {
        if (pc < PA)
            *reinterpret_cast<float4*>(As + buf * BM * LDS_STRIDE + (r0 + RPP * pc) * LDS_STRIDE + kqs * 4) = ra[pc];
        else
            *reinterpret_cast<float4*>(Bs + buf * BN * LDS_STRIDE + (r0 + RPP * (pc - PA)) * LDS_STRIDE + kqs * 4) =
                rb[pc - PA];
    };
    constexpr int NP = PA + PB;  // pieces per k tile

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop ------------------------------------------------------------------------------------
    // Per k tile and wave: NCH chunks of NM = 4*TM*TN MFMAs. The order of everything is pinned
    // (sched_barrier after every MFMA):
    //   chunk j < NCH-1 : prefetch the fragments of chunk j+1, then its MFMAs;
    //   chunk NCH-2     : additionally one LDS-write piece of k tile kt+1 after every NM/NP-th MFMA;
    //   chunk NCH-1     : barrier, prefetch chunk 0 of k tile kt+1, then its MFMAs with one global-load piece
    //                     of k tile kt+2 after every NM/NP-th MFMA.
    // Past the end of K the loads carry out-of-range offsets (zeros, no memory traffic) and the LDS writes
    // fill a buffer nobody reads, so the loop body is free of conditionals.
    const int nk = (p.K + BK - 1) / BK;
    constexpr int NCH = BK / 8;
    constexpr int NM = 4 * TM * TN;
    static_assert(NCH >= 2, "needs at least two chunks per k tile");
    begin_load();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, 0);
    __syncthreads();
    begin_load();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);

    // lane (row ln, half h) reads logical chunk 2 j + h; every row this lane reads is ln + a multiple of 32, so its swizzle
    // term is (ln >> 2) & 3 and (SWZ: two chunks per k tile) the lane needs just two chunk offsets
    const int fsw = SWZ ? (((lane & 31) >> 2) & 3) : 0;
    const int frow = (lane & 31) * LDS_STRIDE;
    const int fch0 = (((lane >> 5)) ^ fsw) * 4, fch1 = ((2 + (lane >> 5)) ^ fsw) * 4;   // SWZ: floats, chunk j = 0 / 1
    const int frag = frow + (lane >> 5) * 4;
    const float* Aw = As + wm * WTM * LDS_STRIDE + (SWZ ? frow : frag);
    const float* Bw = Bs + wn * WTN * LDS_STRIDE + (SWZ ? frow : frag);
    float4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int slot, int buf, int j) {
        const int jo = SWZ ? (j == 0 ? fch0 : fch1) : j * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[slot][i] = *reinterpret_cast<const float4*>(Aw + buf * BM * LDS_STRIDE + i * 32 * LDS_STRIDE + jo);
#pragma unroll
        for (int i = 0; i < TN; ++i)
            fb[slot][i] = *reinterpret_cast<const float4*>(Bw + buf * BN * LDS_STRIDE + i * 32 * LDS_STRIDE + jo);
    };
    read_frags(0, 0, 0);
    f32x16 acc2[SPLIT ? TM : 1][SPLIT ? TN : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    auto ktile = [&](auto chain_tag, int kt) {
        constexpr int CH = decltype(chain_tag)::value;
        const int buf = kt & 1;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int cur = j & 1;
            if (j + 1 < NCH) {
                read_frags(cur ^ 1, buf, j + 1);
            } else {
                __syncthreads();  // every wave has written its pieces of k tile kt+1 and read k tile kt
                read_frags(cur ^ 1, buf ^ 1, 0);
                begin_load();     // k tile kt+2
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 a = fa[cur][i];
                    const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        const float4 b = fb[cur][jn];
                        const float bv = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
                        if constexpr (SPLIT && CH == 1)
                            acc2[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2[i][jn], 0, 0, 0);
                        else
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jn], 0, 0, 0);
                        const int m = (s * TM + i) * TN + jn;  // MFMA index in the chunk, 0..NM-1
#pragma unroll
                        for (int pc = m * NP / NM; pc < (m + 1) * NP / NM; ++pc) {
                            if (j == NCH - 2) store_piece(pc, buf ^ 1);
                            if (j == NCH - 1) load_piece(pc);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };
    if constexpr (!SPLIT) {
        for (int kt = 0; kt < nk; ++kt) ktile(std::integral_constant<int, 0>{}, kt);
    } else {
        constexpr int TPB = 32 / BK;   // k tiles per 32-wide k block: 1 (BK = 32) or 2 (BK = 16)
        static_assert(BK == 32 || BK == 16, "SPLIT: chains are 32-wide k blocks");
        for (int kt = 0; kt < nk; kt += 2 * TPB) {
#pragma unroll
            for (int u = 0; u < TPB; ++u)
                if (kt + u < nk) ktile(std::integral_constant<int, 0>{}, kt + u);
#pragma unroll
            for (int u = 0; u < TPB; ++u)
                if (kt + TPB + u < nk) ktile(std::integral_constant<int, 1>{}, kt + TPB + u);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
    }

    if constexpr (RES == 5) {
        // ---- fused 1x1 heads (RPN conv_class + conv_bbox, model.py:605-607,624-641): the ReLU'd tile of the
        // shared 3x3 conv never goes to HBM. It is transposed through LDS (accumulators hold one channel per
        // lane; the next MFMA needs one pixel per lane) and multiplied by the [32][Cout] head weights; each
        // N tile writes its partial [BM][head_n] sums, a small kernel adds the tiles_n partials in fixed order.
        static_assert(BM == 128 && BN == 128 && WM == 2 && WN == 2, "heads epilogue is written for the 128x128 tile");
        constexpr int TS = BN + 4;  // row pitch of the transposed tile, floats (conflict-free b128 reads)
        const int ln = lane & 31, lh = lane >> 5;
        __syncthreads();  // nobody still reads the main loop's LDS tiles
        float* T = smem;  // [BM][TS] = 67.6 KB <= the 73.7 KB the main loop used
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + wn * WTN + jn * 32 + ln;
            const float sc = p.scale ? p.scale[n] : 1.0f, sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][jn][r] * sc + sh;
                    v = v > 0.f ? v : 0.f;
                    T[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS + wn * WTN + jn * 32 + ln] = v;
                }
        }
        __syncthreads();
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        const float* Trow = T + (wave * 32 + ln) * TS + lh * 4;                       // A: pixel rows of this wave
        const float* Wrow = p.w_head + static_cast<int64_t>(ln) * p.Cout + n0 + lh * 4;  // B: head channel ln
#pragma unroll 4
        for (int j = 0; j < BN / 8; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(Trow + j * 8);
            const float4 b = *reinterpret_cast<const float4*>(Wrow + j * 8);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, hacc, 0, 0, 0);
        }
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const unsigned tile_base = static_cast<unsigned>(nt) * static_cast<unsigned>(p.M) * p.head_n * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool ok = m < p.M && ln < p.head_n;
            const unsigned off = ok ? tile_base + (static_cast<unsigned>(m) * p.head_n + ln) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hacc[r]), o_rsrc, static_cast<int>(off), 0, 0);
        }
        return;
    }

    // ---- epilogue: affine + residual + activation (conv_common.hpp) -------------------------------------
    ResidualRegs<TM, TN, RES> rv;  // fetched in one burst, ahead of every store
    // 8-byte accesses on channel pairs: measured per layer class — the deconv scatter gains 7 % (its 4-byte stores hit four
    // output rows per instruction), the C5 residual layers 3..6 %, but the short-K HBM-bound layers (C2 / C3 conv3, the
    // downsample convs) LOSE 5..12 % and dominate: only the scatter uses it
    if constexpr (RES == 4) {
        if ((p.Cout & 1) == 0 && ((p.Cout >> 2) & 1) == 0) {  // never across a deconv quadrant
            load_residual_pairs<TM, TN, WTM, WTN, RES>(p, m0, n0, wm, wn, lane, rv);
            epilogue_pairs<TM, TN, WTM, WTN, RES>(p, acc, rv, m0, n0, wm, wn, lane, EARLY_AFFINE ? &affine : nullptr);
            return;
        }
    }
    if constexpr (RES != 5) load_residual<TM, TN, WTM, WTN, RES>(p, m0, n0, wm, wn, lane, rv);
    if constexpr (RES != 5) epilogue<TM, TN, WTM, WTN, RES>(p, acc, rv, m0, n0, wm, wn, lane, EARLY_AFFINE ? &affine : nullptr);
}


// ---- streaming 1x1 kernel: ResNet C2's conv1 / downsample layers (model.py:179-180,254-262 at 256 x 256 per image) -----
// K = Cin <= 256 and Cout <= 256 with short K: those layers move 0.67 GB for 17 GFLOP — HBM floor == MFMA floor — and the
// tiled kernel above pays a prologue (two memory round trips before the first MFMA), one barrier per k tile and an epilogue per
// tile, none of which a second workgroup hides completely (3.2-3.8 TB/s). Here NOTHING is shared between waves except the
// weights:
//   B   the whole [Cout][K] weight matrix sits in LDS for the life of a persistent workgroup ([32*TN][K+4] floats, 65-70 KB:
//       two workgroups per CU), read with the same conflict-free ds_read_b128 fragments as above;
//   A   a wave owns blocks of 32 pixels and loads ITS OWN A fragments straight from global memory into registers — lane
//       (row r, half h) reads k = 8j+4h..+3 of row r, exactly the operand the MFMA wants — no LDS staging, no barrier in the
//       loop. The four loads that consume one 128-byte line of a row are issued back to back; a register group is reloaded
//       for the wave's NEXT block right after its last MFMA, so a whole block (8-32 KB per wave) is always in flight;
//   epilogue: conv_common.hpp's, unchanged. Same MFMA sequence per output as the tiled kernel: results are bit-identical.
template <int KC, int TN, int TNP>
__global__ __launch_bounds__(256, 2) void conv_pw_stream_f32(const ConvParams p) {
    constexpr int K = KC * 8, LDSB = K + 4, NPASS = TN / TNP;
    static_assert(KC % 4 == 0 && TN % TNP == 0, "whole 128-byte lines; whole passes");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // Bs[32*TN][LDSB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    for (int idx = tid; idx < 32 * TN * (K / 4); idx += 256) {  // rows beyond Cout: beyond w_bytes, i.e. zeros
        const int n = idx / (K / 4), c = idx - n * (K / 4);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (n * K + 4 * c) * 4, 0, 0);
        *reinterpret_cast<u32x4*>(smem + n * LDSB + 4 * c) = v;
    }
    AffineRegs<TN> affine;
    load_affine<TN, 32 * TN>(p, 0, 0, lane, affine);
    __syncthreads();

    const int nblocks = (p.M + 31) >> 5;
    const int nw = static_cast<int>(gridDim.x) * 4;
    int blk = static_cast<int>(blockIdx.x) * 4 + wave;
    if (blk >= nblocks) return;
    const unsigned lane_off = static_cast<unsigned>(ln * K + 4 * lh) * 4u;
    auto voff = [&](int b) -> unsigned {  // the lane's row of block b; beyond M (or past the last block): out of range, zeros
        return (b < nblocks && b * 32 + ln < p.M) ? static_cast<unsigned>(b) * (32u * K * 4u) + lane_off : OOB;
    };
    u32x4 a[KC];
    {
        const unsigned vo = voff(blk);
#pragma unroll
        for (int j = 0; j < KC; ++j)
            a[j] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(vo == OOB ? OOB : vo + j * 32u), 0, 0);
    }
    const float* Bw = smem + ln * LDSB + lh * 4;
    for (; blk < nblocks; blk += nw) {
        const unsigned vo_next = voff(blk + nw);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            f32x16 acc[1][TNP];
#pragma unroll
            for (int jn = 0; jn < TNP; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][jn][r] = 0.f;
            float4 fb[2][TNP];
            auto read_b = [&](int slot, int j) {
#pragma unroll
                for (int jn = 0; jn < TNP; ++jn)
                    fb[slot][jn] = *reinterpret_cast<const float4*>(Bw + (ps * TNP + jn) * 32 * LDSB + j * 8);
            };
            read_b(0, 0);
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                if (j + 1 < KC) read_b((j + 1) & 1, j + 1);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 av = a[j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float af = __uint_as_float(s == 0 ? av.x : s == 1 ? av.y : s == 2 ? av.z : av.w);
#pragma unroll
                    for (int jn = 0; jn < TNP; ++jn) {
                        const float4 b = fb[j & 1][jn];
                        const float bf = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
                        acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[0][jn], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ps == NPASS - 1 && (j & 3) == 3) {  // the line's four fragments are dead: fetch the next block's
#pragma unroll
                    for (int jj = j - 3; jj <= j; ++jj)
                        a[jj] = __builtin_amdgcn_raw_buffer_load_b128(
                            x_rsrc, static_cast<int>(vo_next == OOB ? OOB : vo_next + jj * 32u), 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            AffineRegs<TNP> af;
#pragma unroll
            for (int jn = 0; jn < TNP; ++jn) {
                af.sc[jn] = affine.sc[ps * TNP + jn];
                af.sh[jn] = affine.sh[ps * TNP + jn];
            }
            ResidualRegs<1, TNP, 0> rv;
            epilogue<1, TNP, 32, 32 * TNP, 0>(p, acc, rv, blk * 32, ps * TNP * 32, 0, 0, lane, &af);
        }
    }
}

template <int KC, int TN, int TNP>
int launch_pw_stream(ConvParams p, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * 32 * TN * (KC * 8 + 4);
    auto kern = conv_pw_stream_f32<KC, TN, TNP>;
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv_pw_stream")) return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv_pw_stream: no device");
    const int nblocks = (p.M + 31) / 32;
    const int grid = std::min(2 * cus, (nblocks + 3) / 4);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, stream, p);
    return mrcnn::check_launch("conv_pw_stream_f32");
}

// K >= this and one of the three tiles the long-K layers take (128x128 BK32, 128x64 BK16, 128x32 BK32), aligned-tap or
// pointwise mode, epilogues 0-2: the two-chain sum (see SPLIT at the kernel)
constexpr int SPLIT_MIN_K = 1024;
template <int BM, int BN, int WM, int WN, int BK>
constexpr bool split_tile() {
    return (BM == 128 && BN == 128 && WM == 2 && WN == 2 && BK == 32) || (BM == 128 && BN == 64 && WM == 2 && WN == 2 && BK == 16) ||
           (BM == 128 && BN == 32 && WM == 4 && WN == 1 && BK == 32);
}

template <int BM, int BN, int WM, int WN, int BK>
int launch_conv(ConvParams p, int mode, hipStream_t stream) {  // mode: 0 aligned taps, 1 generic, 2 pointwise
    const bool generic = mode == 1;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long long grid = tile_grid(p);
    if (grid > 0x7fffffffLL) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv: grid too large");
    constexpr size_t lds = conv_lds_bytes<BM, BN, BK>();
    const int res = epilogue_variant(p, p.w_head != nullptr);
    auto go = [&](auto kern) -> int {
        if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv")) return rc;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, stream, p);
        return MRCNN_OK;
    };
    int rc;
#ifdef MRCNN_ABLATIONS
    if constexpr (BM == 128 && BN == 128 && WM == 2 && WN == 2 && BK == 32) {
        if (res == 5) {
            if (generic) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv: fused heads need Cin %% 32 == 0");
            if ((rc = go(conv_igemm_f32<BM, BN, WM, WN, BK, 0, 5>))) return rc;
            return mrcnn::check_launch("conv_igemm_f32<heads>");
        }
    }
#else
    (void)generic;
#endif
    if (res == 5) return mrcnn::fail(MRCNN_ERR_UNSUPPORTED, "conv: the fused-heads epilogue exists in MRCNN_ABLATIONS builds only");
    if constexpr (split_tile<BM, BN, WM, WN, BK>()) {
        static const bool no_split = getenv("MRCNN_CONV_NO_SPLIT") != nullptr;   // A/B switch (numerics), read once per process
        if (p.K >= SPLIT_MIN_K && mode != 1 && res <= 2 && !no_split) {
            auto by_res_split = [&](auto mode_tag) -> int {
                constexpr int MD = decltype(mode_tag)::value;
                return res == 0 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 0, true>)
                     : res == 1 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 1, true>)
                                : go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 2, true>);
            };
            rc = mode == 2 ? by_res_split(std::integral_constant<int, 2>{}) : by_res_split(std::integral_constant<int, 0>{});
            if (rc) return rc;
            return mrcnn::check_launch("conv_igemm_f32<split>");
        }
    }
    auto by_res = [&](auto mode_tag) -> int {
        constexpr int MD = decltype(mode_tag)::value;
        return res == 0 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 0>)
             : res == 1 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 1>)
             : res == 2 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 2>)
             : res == 3 ? go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 3>)
                        : go(conv_igemm_f32<BM, BN, WM, WN, BK, MD, 4>);
    };
    rc = mode == 1 ? by_res(std::integral_constant<int, 1>{})
       : mode == 2 ? by_res(std::integral_constant<int, 2>{})
                   : by_res(std::integral_constant<int, 0>{});
    if (rc) return rc;
    return mrcnn::check_launch("conv_igemm_f32");
}

}  // namespace

static int run_conv_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                        const float* w, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad_top,
                        int32_t pad_left, int32_t pad_bottom, int32_t pad_right, const float* scale,
                        const float* shift, const float* residual, int32_t res_div, int32_t relu,
                        int32_t out_mode, float* y, mrcnn_stream_t stream, int32_t res_kblocked = 0,
                        const int32_t* row_counts = nullptr, int32_t rows_per_group = 0) {
    MRCNN_REQUIRE(x && w && y, "conv: null pointer");
    ConvParams p;
    if (int rc = fill_common(p, "conv", x, batch, height, width, cin, 4, cout, kh, kw, stride, pad_top, pad_left,
                             pad_bottom, pad_right, scale, shift, residual, res_div, relu, out_mode, y, 4, res_kblocked))
        return rc;
    if (row_counts) {
        MRCNN_REQUIRE(rows_per_group >= 1 && p.M % rows_per_group == 0, "conv: %d output rows are not whole groups of %d",
                      p.M, rows_per_group);
        p.row_counts = row_counts;
        p.rows_per_group = rows_per_group;
    }
    p.w = w;
    p.w_head = nullptr;
    p.head_n = 0;
    const long long M = p.M;
    const bool generic = (cin % 32) != 0;
    static const bool no_pw = mrcnn::tuning_env("MRCNN_CONV_NO_PW") != nullptr;   // tuning aid, read once per process
    const bool pointwise = !generic && kh == 1 && kw == 1 && pad_top == 0 && pad_left == 0 && pad_bottom == 0 &&
                           pad_right == 0 && !no_pw;
    const int mode = generic ? 1 : pointwise ? 2 : 0;
    hipStream_t s = mrcnn::as_stream(stream);
    static const int force = mrcnn::tuning_env("MRCNN_CONV_TILE") ? atoi(mrcnn::tuning_env("MRCNN_CONV_TILE")) : 0;  // tuning aid
    // ResNet C2's 1x1 layers without a residual (conv1 256 -> 64 / 64 -> 64, downsample 64 -> 256) on large maps: the streaming
    // kernel (bit-identical results; chosen by size only because a persistent wave needs several blocks to pipeline)
    static const bool no_stream = mrcnn::tuning_env("MRCNN_CONV_NO_STREAM") != nullptr;
    if (pointwise && !no_stream && !row_counts && stride == 1 && !residual && relu <= 1 && out_mode != 1 && p.M >= 131072) {
        if (cin == 256 && cout > 32 && cout <= 64) return launch_pw_stream<32, 2, 2>(p, s);
        if (cin == 64 && cout > 32 && cout <= 64) return launch_pw_stream<8, 2, 2>(p, s);
        if (cin == 64 && cout > 128 && cout <= 256) return launch_pw_stream<8, 8, 4>(p, s);
    }
    if (cout <= 32) return launch_conv<128, 32, 4, 1, 32>(p, mode, s);
    // Cout <= 64: 256x64 tile. BK = 16 keeps its LDS at 51 KB (two workgroups per CU; BK = 32 needs 92 KB = one):
    // measured 3-20 % faster on the C2 layers. The stem (generic K) keeps BK = 32.
    if (cout <= 64) return (generic || force == 4) ? launch_conv<256, 64, 4, 1, 32>(p, mode, s)
                                                   : launch_conv<256, 64, 4, 1, 16>(p, mode, s);
    // 64 < Cout <= 96 (the mask head's conv5, 256 -> 81 + sigmoid on 313 600 pixels, model.py:913-914): a 128x96 tile — four
    // waves of 32 rows x 96 columns — instead of a 128-column tile of which 37 % would be padding (MFMA-bound at the padded
    // width: 0.189 -> 0.151 ms). Same products in the same order: bit-identical (MRCNN_CONV_TILE=1 keeps the 128x128 tile).
    if (!generic && force == 0 && cout > 64 && cout <= 96) return launch_conv<128, 96, 4, 1, 32>(p, mode, s);
    // (a 256x128 BK16 tile — 25% fewer LDS/global bytes per MFMA — was measured in round 1: no gain over 128x128 even on
    // the largest layers, 133.5 vs 133.2 TFLOP/s, and a loss on mid-size ones; removed)
    // 128x128 with BK = 16 (41 KB of LDS: three workgroups per CU) for the shortest K (<= 128) with wide outputs — C3 conv3
    // (K = 128 -> 512 + residual) and the C2 downsample (K = 64 -> 256): measured in the pipeline against the 128x64 / 128x128 BK32
    // tiles they used to take, 0.766 -> 0.723 ms over the four C3 layers, 0.199 -> 0.190; every longer-K layer is slower on it
    // round 4 (in-pipeline per-launch events, MRCNN_CONV_TILE sweeps): the same tile for the mask head's transposed conv (K = 256
    // -> 4 x 256 channels scattered 2x2: its epilogue is store-bound, three workgroups per CU overlap it better) 0.371 -> 0.351 ms
    if (!generic && (force == 3 || (force == 0 && ((p.K <= 128 && cout >= 256) || (out_mode == 1 && p.K <= 256 && cout >= 256)))))
        return launch_conv<128, 128, 2, 2, 16>(p, mode, s);
    // Bottleneck conv3 (1x1 expansion + residual, K = planes <= 256): latency-bound on load -> MFMA -> residual -> store
    // per tile; a 128x64 BK16 tile (30 KB of LDS, 32 accumulator registers) keeps five workgroups per CU in flight
    // instead of two: 7 % faster on those layers, slower on everything else (measured per layer, round 1)
    // The same tile when 128x128 tiles would not even give every CU one workgroup (P5 lateral at batch 8: 128 tiles; 131 ->
    // 94 us). Same products in the same order per output: the choice never changes a result.
    const long long tiles128 = ((p.M + 127) / 128) * static_cast<long long>((cout + 127) / 128);
    const bool underfilled = cout >= 128 && tiles128 < mrcnn::device_cu_count();
    // ... and when even 128x64 tiles give a CU only one workgroup with a long K to walk alone (P5 lateral at batch 8: 256 tiles,
    // K = 2048: 0.104 ms at 0.56 of its MFMA floor), 128x32 tiles put two on every CU: 0.072 ms (MRCNN_CONV_TILE=5: 128x64)
    if (!generic && force == 0 && underfilled && p.K >= 1024 &&
        ((p.M + 127) / 128) * static_cast<long long>((cout + 63) / 64) <= mrcnn::device_cu_count())
        return launch_conv<128, 32, 4, 1, 32>(p, mode, s);
    if (!generic && force != 1 &&
        (force == 5 || underfilled || (p.K <= 256 && p.residual && p.res_div == 1 && cout >= 128)))
        return launch_conv<128, 64, 2, 2, 16>(p, mode, s);
    return launch_conv<128, 128, 2, 2, 32>(p, mode, s);
}

extern "C" int mrcnn_conv_bn_act_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                          int32_t cin, const float* w, int32_t cout, int32_t kh,
                                          int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                                          int32_t pad_bottom, int32_t pad_right, const float* scale,
                                          const float* shift, const float* residual, int32_t res_div,
                                          int32_t activation, float* y, mrcnn_stream_t stream) {
    return run_conv_f32(x, batch, height, width, cin, w, cout, kh, kw, stride, pad_top, pad_left, pad_bottom,
                        pad_right, scale, shift, residual, res_div, activation, 0, y, stream);
}

extern "C" int mrcnn_conv_bn_act_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                     const float* w, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                                     int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                                     const float* scale, const float* shift, const float* residual, int32_t res_div,
                                     int32_t residual_layout, int32_t activation, float* y, int32_t y_layout,
                                     mrcnn_stream_t stream) {
    MRCNN_REQUIRE(y_layout == MRCNN_LAYOUT_NHWC || y_layout == MRCNN_LAYOUT_KBLOCKED, "conv: bad y_layout");
    MRCNN_REQUIRE(residual_layout == MRCNN_LAYOUT_NHWC || residual_layout == MRCNN_LAYOUT_KBLOCKED,
                  "conv: bad residual_layout");
    return run_conv_f32(x, batch, height, width, cin, w, cout, kh, kw, stride, pad_top, pad_left, pad_bottom,
                        pad_right, scale, shift, residual, res_div, activation,
                        y_layout == MRCNN_LAYOUT_KBLOCKED ? 2 : 0, y, stream,
                        residual_layout == MRCNN_LAYOUT_KBLOCKED ? 1 : 0);
}

extern "C" int mrcnn_conv_bn_act_rows_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                          const float* w, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                                          int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                                          const float* scale, const float* shift, int32_t activation, float* y,
                                          const int32_t* row_counts, int32_t rows_per_group, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(row_counts == nullptr || rows_per_group >= 1, "conv_rows: rows_per_group=%d", rows_per_group);
    return run_conv_f32(x, batch, height, width, cin, w, cout, kh, kw, stride, pad_top, pad_left, pad_bottom, pad_right, scale,
                        shift, nullptr, 1, activation, 0, y, stream, 0, row_counts, rows_per_group);
}

// BM of the tile run_conv_f32 picks for a row-group call (row_counts != NULL: never the streaming kernel): 256 x 64 tiles for
// 32 < Cout <= 64, 128-row tiles for every other width (128x32 / 128x64 / 128x96 / 128x128 above).
extern "C" int32_t mrcnn_conv_rows_tile_m(int32_t cout) { return (cout > 32 && cout <= 64) ? 256 : 128; }

extern "C" int mrcnn_deconv2x2_bias_act_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                                 int32_t cin, const float* w, int32_t cout, const float* bias4,
                                                 int32_t activation, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(cout >= 1, "deconv2x2: cout=%d", cout);
    return run_conv_f32(x, batch, height, width, cin, w, 4 * cout, 1, 1, 1, 0, 0, 0, 0, nullptr, bias4, nullptr, 1,
                        activation, 1, y, stream);
}

#ifdef MRCNN_ABLATIONS   // the direct-kernel fused RPN level (round 1; the default path fuses the heads into the Winograd kernels)
namespace {
// y[m][c] = bias[c] + sum over N tiles (fixed order: deterministic) of partial[t][m][c]
__global__ __launch_bounds__(256) void heads_reduce(const float* __restrict__ partial, const float* __restrict__ bias,
                                                    int64_t MC, int head_n, int tiles, float* __restrict__ y) {
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < MC;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float v = bias ? bias[e % head_n] : 0.f;
        for (int t = 0; t < tiles; ++t) v += partial[t * MC + e];
        y[e] = v;
    }
}
}  // namespace

extern "C" size_t mrcnn_rpn_level_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cout,
                                                  int32_t head_n) {
    if (batch < 1 || height < 1 || width < 1 || cout < 1 || head_n < 1) return 0;
    return sizeof(float) * static_cast<size_t>((cout + 127) / 128) * batch * height * width * head_n;
}

extern "C" int mrcnn_rpn_level_fused_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                         const float* w_shared, int32_t cout, const float* b_shared,
                                         const float* w_head32, const float* b_head, int32_t head_n,
                                         void* workspace, size_t workspace_bytes, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && w_shared && w_head32 && workspace && y, "rpn_level: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 1 && width >= 1 && cin % 32 == 0 && cin >= 32,
                  "rpn_level: bad shape (Cin %% 32 == 0 required)");
    MRCNN_REQUIRE(cout % 128 == 0 && cout >= 128, "rpn_level: Cout=%d must be a multiple of 128", cout);
    MRCNN_REQUIRE(head_n >= 1 && head_n <= 32, "rpn_level: head_n=%d must be in [1,32]", head_n);
    MRCNN_REQUIRE(workspace_bytes >= mrcnn_rpn_level_workspace_bytes(batch, height, width, cout, head_n),
                  "rpn_level: workspace too small");
    const long long M = 1LL * batch * height * width;
    const int tiles_n = cout / 128;
    MRCNN_REQUIRE(M * cin < (1LL << 30) && 9LL * cin * cout < (1LL << 30) && M * head_n * tiles_n < (1LL << 30),
                  "rpn_level: tensor too large for 32-bit buffer offsets");
    ConvParams p;
    p.x = x; p.w = w_shared; p.scale = nullptr; p.shift = b_shared; p.residual = nullptr;
    p.y = static_cast<float*>(workspace);
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = 3; p.KW = 3;
    p.stride = 1; p.pad_t = 1; p.pad_l = 1; p.OH = height; p.OW = width;
    p.M = static_cast<int>(M);
    p.K = 9 * cin;
    p.res_div = 1; p.act = 1; p.out_mode = 0; p.res_kblocked = 0;
    p.ow_shift = p.ohw_shift = -1;  // (pixel decode by division: this entry point is not on the default path)
    p.x_bytes = static_cast<unsigned>(4LL * M * cin);
    p.w_bytes = static_cast<unsigned>(4LL * p.K * cout);
    p.y_bytes = static_cast<unsigned>(4LL * tiles_n * M * head_n);
    p.r_bytes = 0;
    p.w_head = w_head32;
    p.head_n = head_n;
    hipStream_t s = mrcnn::as_stream(stream);
    if (int rc = launch_conv<128, 128, 2, 2, 32>(p, 0, s)) return rc;
    const int64_t MC = M * head_n;
    int64_t blocks = (MC + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(heads_reduce, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), b_head, MC, head_n, tiles_n, y);
    return mrcnn::check_launch("heads_reduce");
}
#endif  // MRCNN_ABLATIONS
