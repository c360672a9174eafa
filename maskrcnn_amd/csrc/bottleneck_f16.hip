// Bottleneck.forward (model.py:190-211) for the ResNet C2 blocks (planes = 64, stride 1) of the plain-fp16 path
// (BASELINE configs[4]) in ONE launch, gfx950 only:
//     y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + residual),  residual = x  or  bn_d(conv_d(x))  (model.py:254-262)
// fp16 NHWC in, fp16 NHWC out, v_mfma_f32_16x16x32_f16 with fp32 accumulation, both 64-channel intermediates rounded to fp16 exactly
// where the three (four) launches of the per-layer path round them (their HBM tensors) — so the result differs from that path only
// by the summation order inside an MFMA of another shape.
//
// Why one launch in THIS mode and not in fp32 (§5.1c of DESIGN.md: the fp32 whole-block kernel is MFMA-bound and lost): at the
// fp16 MFMA rate the block's arithmetic is a tenth of its memory time. The per-layer path moves, per identity block, x in (conv1),
// the 64-channel maps out and in twice, x in again as the residual and y out: 1.14 GB at 832 x 1344, batch 8, in 0.325 ms; here
// x is read once (with a one-pixel halo) and y written once.
//
// Tile: 8 x 16 output pixels per workgroup step, eight waves, persistent workgroups (one per CU), XCD-aware tile order.
//   * MFMA orientation as conv_f16p.hip: D = W_frag x X_frag — a lane (pixel l16 = lane & 15, k chunk q = lane >> 4) of a
//     B operand holds 8 consecutive channels of its pixel, and an accumulator holds 4 channel rows of that pixel. The rows of
//     every weight block are permuted (row rho of 16-channel block cb <-> channel (cb>>1)*32 + (rho>>2)*8 + (cb&1)*4 + (rho&3)) so
//     that two accumulators (2h, 2h+1) are the lane's channels h*32 + q*8 .. +7 — which IS the lane's B fragment for k chunk h
//     of the next 1x1 conv, and IS the lane's 16 bytes of x / y for 32-channel group h. Hence: conv2's output never leaves the
//     registers (it is conv3's B operand), the identity residual is the lane's own conv1 operand (x is not read twice), and the
//     stores are 16 bytes per lane, 64 contiguous bytes per pixel and instruction.
//   * phase 1, conv1 on the tile + halo (10 x 18 pixels): 12 pixel tiles of 16 — the ten halo rows' columns 1..16 and the two
//     edge columns — wave w owns row w + 1 (its OUTPUT row) and waves 0-3 one of the other four; x fragments straight from global
//     memory into B operands (no LDS), w1's A fragments from LDS (identity: 28 of its 32; the other 4 in registers) or all 8 in registers
//     (first block).
//     relu(bn1(.)) -> fp16 -> LDS image T1 [180 pixels][64 ch], 16-byte chunk c of pixel P at c ^ ((P >> 1) & 7) (the ds_read_b128
//     of a B fragment is conflict-free), ZERO outside the picture (SamePad2d pads conv2's input, not conv1's).
//   * phase 2, conv2 out of T1: per tap and 32-channel chunk one B fragment (the wave's row shifted by the tap) and four A
//     fragments from LDS (w2 resident for the workgroup's life: 72 KB).
//   * phase 3, conv3 (w3 resident in LDS, 32 KB) and, in the first block, the downsample conv on the SAME x fragments (wd: 24
//     fragments in LDS, 8 in registers), epilogue relu(acc * s3 + t3 + residual) as conv_f16p.hip's.
//   * the next tile's x fragments are requested as soon as phase 1 has consumed the current ones; two LDS-only barriers per tile.
// LDS: w2 73 728 + w3 32 768 + w1 | wd 28 672 + T1 23 040 + affine tables 5 120 = 163 328 B of the 163 840: every operand but x is
// resident — inside the tile loop only x loads and y stores touch memory (w1 streamed from L2 per tile, the first version, put an
// exposed L2 round trip in front of each of its eight k chunks).
#include <algorithm>

#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16, HWD = TW + 2, NPIX = (TH + 2) * HWD;   // 10 x 18 halo'd tile
constexpr int L_W2 = 0, W2_BYTES = 72 * 1024;
constexpr int L_W3 = L_W2 + W2_BYTES, W3_BYTES = 32 * 1024;
// the fourth weight set, minus the fragments that live in registers: identity block — w1 (32 fragments, 4 in registers);
// first block — the downsample conv's wd (32 fragments, 8 in registers; its w1 is 8 fragments, all in registers)
constexpr int L_WX = L_W3 + W3_BYTES, WX_BYTES = 28 * 1024;
constexpr int WX_REG_ID = 4, WX_REG_FIRST = 8;
constexpr int L_T1 = L_WX + WX_BYTES, T1_BYTES = NPIX * 128;
constexpr int L_TAB = L_T1 + T1_BYTES;
constexpr int TAB_S1 = 0, TAB_T1 = 256, TAB_S2 = 512, TAB_T2 = 768, TAB_S3 = 1024, TAB_T3 = 2048, TAB_SD = 3072, TAB_TD = 4096;
constexpr int LDS_BYTES = L_TAB + 5120;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int BAND = 4;   // tile rows per band of the tile order
// timing-only ablations (MRCNN_BF16_ABL=<mask> python maskrcnn_amd/build.py; tools/c2_f16_ablate.sh): 1 no x loads after the first
// tile, 2 no stores, 4 no conv2 MFMAs / fragment reads, 8 no conv3 (+ downsample) MFMAs, 16 no conv1 MFMAs. Results are wrong.
#ifndef MRCNN_BF16_ABL
#define MRCNN_BF16_ABL 0
#endif

struct BParams {
    const _Float16* x;      // [B][H][W][Cin]
    const _Float16* w1;     // A fragments [Cin/32][4][64][8]
    const _Float16* w2;     // [18][4][64][8]   (k chunk = tap * 2 + 32-channel half)
    const _Float16* w3;     // [2][16][64][8]
    const _Float16* wd;     // [2][16][64][8] or null
    const float *s1, *t1, *s2, *t2, *s3, *t3, *sd, *td;
    _Float16* y;            // [B][H][W][256]
    int B, H, W;
    int tiles_x, tiles_y, tiles;
    unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ void lds_barrier() {   // LDS-only: outstanding global loads / stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    MRCNN_SYNC_FUZZ_POINT();
}

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Epilogue arithmetic on PAIRS (v_pk_fma_f32, v_pk_add_f32, v_cvt_pk_f16_f32, v_pk_max_f16: three VALU instructions per element
// instead of six — at two waves per SIMD the element-wise epilogues were as long as the MFMAs). Same values as the per-layer kernels'
// epilogues: the fp32 result is materialised, then rounded to fp16 (two roundings); ReLU after the rounding (rounding is monotonic
// and keeps zero: max(fp16(v), 0) == fp16(max(v, 0))).
__device__ __forceinline__ f32x2 pair_of(const f32x4& lo, const f32x4& hi, int j) {   // elements 2j, 2j+1 of the lane's 8 channels
    const f32x4& a = j < 2 ? lo : hi;
    return (j & 1) ? f32x2{a[2], a[3]} : f32x2{a[0], a[1]};
}
// the lane's 8 consecutive table entries (channels ch .. ch + 7) as four pairs
__device__ __forceinline__ void tab8(const unsigned char* tab, int ch, f32x2 (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(tab + ch * 4), b = *reinterpret_cast<const f32x4*>(tab + ch * 4 + 16);
    v[0] = f32x2{a[0], a[1]}; v[1] = f32x2{a[2], a[3]}; v[2] = f32x2{b[0], b[1]}; v[3] = f32x2{b[2], b[3]};
}
__device__ __forceinline__ f16x8 pack8(const f16x2 (&h)[4]) {
    return __builtin_bit_cast(f16x8, u32x4{__builtin_bit_cast(unsigned, h[0]), __builtin_bit_cast(unsigned, h[1]),
                                           __builtin_bit_cast(unsigned, h[2]), __builtin_bit_cast(unsigned, h[3])});
}
__device__ __forceinline__ f16x2 round_relu(f32x2 v) {
    asm volatile("" : "+v"(v));   // the fp32 value exists before it is rounded (no fused multiply-add-convert)
    return __builtin_elementwise_max(__builtin_convertvector(v, f16x2), f16x2{0, 0});
}
// relu(acc * s + t) of the accumulator pair (lo, hi) = the lane's 8 channels, as fp16
__device__ __forceinline__ f16x8 affine_relu_f16(const f32x4& lo, const f32x4& hi, const unsigned char* tab_s, const unsigned char* tab_t,
                                                 int ch) {
    f32x2 s[4], t[4];
    tab8(tab_s, ch, s);
    tab8(tab_t, ch, t);
    f16x2 h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = round_relu(__builtin_elementwise_fma(pair_of(lo, hi, j), s[j], t[j]));
    return pack8(h);
}

template <bool FIRST>
__global__ __launch_bounds__(512, 1) void bottleneck_c2_f16(const BParams p) {
    constexpr int KC1 = FIRST ? 2 : 8;   // 32-channel chunks of x
    constexpr int CIN = KC1 * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, q = lane >> 4;

    // ---- resident operands: w2, w3, the affine tables ---------------------------------------------------------------
    {
        const u32x4* s2 = reinterpret_cast<const u32x4*>(p.w2);
        u32x4* d2 = reinterpret_cast<u32x4*>(smem + L_W2);
        for (int i = threadIdx.x; i < W2_BYTES / 16; i += 512) d2[i] = s2[i];
        const u32x4* s3 = reinterpret_cast<const u32x4*>(p.w3);
        u32x4* d3 = reinterpret_cast<u32x4*>(smem + L_W3);
        for (int i = threadIdx.x; i < W3_BYTES / 16; i += 512) d3[i] = s3[i];
        {
            constexpr int skip = FIRST ? WX_REG_FIRST : WX_REG_ID;   // fragments that stay in registers
            const u32x4* sx = reinterpret_cast<const u32x4*>(FIRST ? p.wd : p.w1) + skip * 64;
            u32x4* dx = reinterpret_cast<u32x4*>(smem + L_WX);
            for (int i = threadIdx.x; i < (32 - skip) * 64; i += 512) dx[i] = sx[i];
        }
        float* tab = reinterpret_cast<float*>(smem + L_TAB);
        const int i = threadIdx.x;
        if (i < 64) {
            tab[TAB_S1 / 4 + i] = p.s1[i]; tab[TAB_T1 / 4 + i] = p.t1[i];
            tab[TAB_S2 / 4 + i] = p.s2[i]; tab[TAB_T2 / 4 + i] = p.t2[i];
        }
        if (i < 256) {
            tab[TAB_S3 / 4 + i] = p.s3[i]; tab[TAB_T3 / 4 + i] = p.t3[i];
            if constexpr (FIRST) { tab[TAB_SD / 4 + i] = p.sd[i]; tab[TAB_TD / 4 + i] = p.td[i]; }
        }
    }
    // register-resident A fragments: w1 of the first block (Cin = 64: all 8) / the k chunk 0 of w1 (identity: 4 of 32), and the
    // first 8 of the downsample conv's 32 (first block) — what the 160 KB of LDS do not hold
    f16x8 w1r[FIRST ? 8 : WX_REG_ID], wdr[FIRST ? WX_REG_FIRST : 1];
#pragma unroll
    for (int f = 0; f < (FIRST ? 8 : WX_REG_ID); ++f) w1r[f] = *reinterpret_cast<const f16x8*>(p.w1 + (f * 64 + lane) * 8);
    if constexpr (FIRST) {
#pragma unroll
        for (int f = 0; f < WX_REG_FIRST; ++f) wdr[f] = *reinterpret_cast<const f16x8*>(p.wd + (f * 64 + lane) * 8);
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // ---- the lane's pixels of the halo'd tile -----------------------------------------------------------------------
    // own tile: halo row wave + 1, columns 1..16; extra tile (waves 0-3): rows 0 / 9 columns 1..16, or column 0 / 17 rows 0..9
    const int hyA = wave + 1, hxA = 1 + l16;
    const bool has_b = wave < 4;
    const int hyB = wave == 0 ? 0 : wave == 1 ? TH + 1 : l16;
    const int hxB = wave < 2 ? 1 + l16 : wave == 2 ? 0 : TW + 1;
    const bool lane_b = has_b && (wave < 2 || l16 < TH + 2);
    const int PA = hyA * HWD + hxA, PB = hyB * HWD + hxB;

    // XCD-aware persistent order: XCD x owns a contiguous range of tiles, its workgroups stride through it side by side
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int lo = static_cast<int>((static_cast<long long>(xcd) * p.tiles) >> 3);
    const int hi = static_cast<int>((static_cast<long long>(xcd + 1) * p.tiles) >> 3);
    int t = lo + local;
    if (t >= hi) return;

    const int tiles_img = p.tiles_x * p.tiles_y;
    struct Geo { int b, ty0, tx0; unsigned offA, offB; bool inA, inB; };
    auto geometry = [&](int tile) {
        Geo g;
        g.b = tile / tiles_img;
        // tiles of an image in bands of BAND tile rows, column by column inside a band: the 32 tiles an XCD works on at one time
        // are a BAND x 8 block (32 x 128 pixels), so most of a tile's halo is a neighbour's interior in the same L2 at the same time
        const int r = tile - g.b * tiles_img, band_tiles = BAND * p.tiles_x;
        const int band = r / band_tiles, rem = r - band * band_tiles;
        const int rows = min(BAND, p.tiles_y - band * BAND);
        const int tx = rem / rows, ty = band * BAND + (rem - tx * rows);
        g.ty0 = ty * TH;
        g.tx0 = tx * TW;
        const int iyA = g.ty0 - 1 + hyA, ixA = g.tx0 - 1 + hxA, iyB = g.ty0 - 1 + hyB, ixB = g.tx0 - 1 + hxB;
        g.inA = iyA >= 0 && iyA < p.H && ixA >= 0 && ixA < p.W;
        g.inB = lane_b && iyB >= 0 && iyB < p.H && ixB >= 0 && ixB < p.W;
        g.offA = g.inA ? static_cast<unsigned>((g.b * p.H + iyA) * p.W + ixA) * (CIN * 2u) + q * 16u : OOB;
        g.offB = g.inB ? static_cast<unsigned>((g.b * p.H + iyB) * p.W + ixB) * (CIN * 2u) + q * 16u : OOB;
        return g;
    };
    f16x8 xa[KC1], xb[KC1];
    auto load_x = [&](const Geo& g) {
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc)
            xa[kc] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(g.offA), kc * 64, 0));
        // (waves 4-7 have no second tile: their offsets are out of range, the loads move nothing — but EVERY wave issues the same
        // memory instructions in the same order, so the compiler's vmcnt bookkeeping is exact; see the dummy stores below)
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc)
            xb[kc] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(g.offB), kc * 64, 0));
    };
    auto write_t1 = [&](const f32x4 (&acc)[4], int P, bool inside) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f16x8 o = affine_relu_f16(acc[2 * h], acc[2 * h + 1], smem + L_TAB + TAB_S1, smem + L_TAB + TAB_T1, h * 32 + q * 8);
            if (!inside) o = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            const int chunk = h * 4 + q;
            *reinterpret_cast<f16x8*>(smem + L_T1 + P * 128 + ((chunk ^ ((P >> 1) & 7)) << 4)) = o;
        }
    };

    Geo g = geometry(t);
    load_x(g);
    // Eight stores that store nothing (out of range), so that the loop is ENTERED with the memory instructions in flight it is
    // re-entered with: x loads of the tile, then the previous tile's 8 stores. The compiler merges the wait counts of both entries
    // to the stricter one; without these the first use of an x fragment waited for "all but 7 - kc" operations — i.e., on the
    // back edge, for the previous tile's STORES to complete (their latency exposed once per tile: ISA, first version).
#pragma unroll
    for (int hp = 0; hp < 8; ++hp)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, y_rsrc, static_cast<int>(OOB), hp * 64, 0);
    for (;;) {
        // ---- phase 1: conv1 on the wave's pixel tiles ---------------------------------------------------------------
        // (every MFMA group of the three phases reads its A / B fragments one group AHEAD, into a second register set, with a
        // scheduling barrier between the reads and the MFMAs: left to itself the compiler sinks each ds_read to one MFMA in front of
        // its use, and the wave waits out the LDS latency once per MFMA — SQ_WAIT_INST 36 % of the wave cycles, first version)
        f32x4 a1[4] = {zero4(), zero4(), zero4(), zero4()}, b1[4] = {zero4(), zero4(), zero4(), zero4()};
        auto w1_frag = [&](int kc, int cb) -> f16x8 {
            if (FIRST || kc * 4 + cb < WX_REG_ID) return w1r[(FIRST || kc * 4 + cb < WX_REG_ID) ? kc * 4 + cb : 0];
            return *reinterpret_cast<const f16x8*>(smem + L_WX + (kc * 4 + cb - WX_REG_ID) * 1024 + lane * 16);
        };
        {
            f16x8 wf[2][4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wf[0][cb] = w1_frag(0, cb);
#pragma unroll
            for (int kc = 0; kc < ((MRCNN_BF16_ABL & 16) ? 1 : KC1); ++kc) {
                if (kc + 1 < KC1) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) wf[(kc + 1) & 1][cb] = w1_frag(kc + 1, cb);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) a1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kc & 1][cb], xa[kc], a1[cb], 0, 0, 0);
                if (has_b) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) b1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kc & 1][cb], xb[kc], b1[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f16x8 xres[KC1];   // the wave's own row of x: the identity residual / the downsample conv's operand
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc) xres[kc] = xa[kc];
        const Geo gc = g;
        const int tn = t + per;
        if (tn < hi) {     // the next tile's x, in flight through phases 2 and 3
            g = geometry(tn);
            if (!(MRCNN_BF16_ABL & 1)) load_x(g);
        }
        lds_barrier();     // every wave has finished reading the previous tile's T1
        write_t1(a1, PA, gc.inA);
        if (lane_b) write_t1(b1, PB, gc.inB);
        lds_barrier();

        // ---- phase 2: conv2 out of T1 ------------------------------------------------------------------------------
        f32x4 a2[4] = {zero4(), zero4(), zero4(), zero4()};
        {
            f16x8 wf[2][8], xf[2][2];
            auto read_tap = [&](int tap, int set) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const int P = (wave + dy) * HWD + l16 + dx;
                const unsigned char* px = smem + L_T1 + P * 128;
                const int sw = (P >> 1) & 7;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    xf[set][kc] = *reinterpret_cast<const f16x8*>(px + (((kc * 4 + q) ^ sw) << 4));
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        wf[set][kc * 4 + cb] = *reinterpret_cast<const f16x8*>(smem + L_W2 + ((tap * 2 + kc) * 4 + cb) * 1024 + lane * 16);
                }
            };
            read_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < ((MRCNN_BF16_ABL & 4) ? 1 : 9); ++tap) {
                if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        a2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tap & 1][kc * 4 + cb], xf[tap & 1][kc], a2[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f16x8 t2f[2];      // relu(bn2(conv2)) in fp16: the lane's B fragments of conv3
#pragma unroll
        for (int h = 0; h < 2; ++h)
            t2f[h] = affine_relu_f16(a2[2 * h], a2[2 * h + 1], smem + L_TAB + TAB_S2, smem + L_TAB + TAB_T2, h * 32 + q * 8);

        // ---- phase 3: conv3 (+ the downsample conv of the first block), residual, ReLU, store --------------------------
        const int oy = gc.ty0 + wave, ox = gc.tx0 + l16;
        const unsigned yoff = (oy < p.H && ox < p.W) ? static_cast<unsigned>((gc.b * p.H + oy) * p.W + ox) * 512u + q * 16u : OOB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // two passes of 128 output channels: 8 (16) accumulators live at a time
            f32x4 a3[8], ad[FIRST ? 8 : 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) a3[j] = zero4();
            if constexpr (FIRST) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ad[j] = zero4();
            }
            {
                // groups of 8 MFMAs: (conv3, kc 0), (conv3, kc 1) and, in the first block, (downsample, kc 0), (downsample, kc 1)
                constexpr int NG = FIRST ? 4 : 2;
                f16x8 wf[2][8];
                auto read_group = [&](int grp, int set) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kc = grp & 1, cb = half * 8 + j, f = kc * 16 + cb;
                        if (grp < 2) wf[set][j] = *reinterpret_cast<const f16x8*>(smem + L_W3 + f * 1024 + lane * 16);
                        else if (f < WX_REG_FIRST) wf[set][j] = wdr[f < WX_REG_FIRST ? f : 0];
                        else wf[set][j] = *reinterpret_cast<const f16x8*>(smem + L_WX + (f - WX_REG_FIRST) * 1024 + lane * 16);
                    }
                };
                read_group(0, 0);
#pragma unroll
                for (int grp = 0; grp < ((MRCNN_BF16_ABL & 8) ? 1 : NG); ++grp) {
                    if (grp + 1 < NG) read_group(grp + 1, (grp + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp < 2) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) a3[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[grp & 1][j], t2f[grp & 1], a3[j], 0, 0, 0);
                    } else if constexpr (FIRST) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) ad[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[grp & 1][j], xres[grp & 1], ad[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const int hp = half * 4 + hh;   // 32-channel group: the lane's channels hp*32 + q*8 .. +7
                f32x2 s[4], sh[4], res[4];
                tab8(smem + L_TAB + TAB_S3, hp * 32 + q * 8, s);
                tab8(smem + L_TAB + TAB_T3, hp * 32 + q * 8, sh);
                if constexpr (FIRST) {
                    f32x2 sdv[4], tdv[4];
                    tab8(smem + L_TAB + TAB_SD, hp * 32 + q * 8, sdv);
                    tab8(smem + L_TAB + TAB_TD, hp * 32 + q * 8, tdv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {   // the downsample branch as the per-layer path leaves it in HBM: fp16
                        f32x2 r = __builtin_elementwise_fma(pair_of(ad[2 * hh], ad[2 * hh + 1], j), sdv[j], tdv[j]);
                        asm volatile("" : "+v"(r));
                        res[j] = __builtin_convertvector(__builtin_convertvector(r, f16x2), f32x2);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[j] = f32x2{static_cast<float>(xres[hp][2 * j]), static_cast<float>(xres[hp][2 * j + 1])};
                }
                f16x2 h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    h[j] = round_relu(__builtin_elementwise_fma(pair_of(a3[2 * hh], a3[2 * hh + 1], j), s[j], sh[j]) + res[j]);
                const f16x8 o = pack8(h);
                if (!(MRCNN_BF16_ABL & 2) || o[0] == static_cast<_Float16>(12345.0f))
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y_rsrc, static_cast<int>(yoff), hp * 64, 0);
            }
        }
        if (tn >= hi) break;
        t = tn;
    }
}

// [cout][k] fp16 (OHWI flattened) -> A fragments [k/32][cout/16][64 lanes][8]: lane (rho = lane & 15, q = lane >> 4) of
// fragment (kk, cb) holds w[channel(cb, rho)][kk*32 + q*8 .. +7], channel(cb, rho) = (cb>>1)*32 + (rho>>2)*8 + (cb&1)*4 + (rho&3)
__global__ __launch_bounds__(256) void pack_afrags_f16(const _Float16* __restrict__ w, int cout, int k, _Float16* __restrict__ dst) {
    const int nb = cout / 16, nk = k / 32;
    const int total = nk * nb * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, f = i >> 6, cb = f % nb, kk = f / nb;
        const int rho = lane & 15, q = lane >> 4;
        const int ch = (cb >> 1) * 32 + (rho >> 2) * 8 + (cb & 1) * 4 + (rho & 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(w + static_cast<long long>(ch) * k + kk * 32 + q * 8);
        *reinterpret_cast<u32x4*>(dst + static_cast<long long>(i) * 8) = v;
    }
}

}  // namespace

extern "C" int mrcnn_pack_afrags_f16(const void* w_f16, int32_t cout, int32_t k, void* frags_f16, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(w_f16 && frags_f16, "pack_afrags_f16: null pointer");
    MRCNN_REQUIRE(cout >= 32 && cout % 32 == 0 && k >= 32 && k % 32 == 0 && static_cast<long long>(cout) * k < (1LL << 30),
                  "pack_afrags_f16: needs cout %% 32 == 0 and k %% 32 == 0 (got %d x %d)", cout, k);
    const int total = (k / 32) * (cout / 16) * 64;
    hipLaunchKernelGGL(pack_afrags_f16, dim3((total + 255) / 256), dim3(256), 0, mrcnn::as_stream(stream),
                       static_cast<const _Float16*>(w_f16), cout, k, static_cast<_Float16*>(frags_f16));
    return mrcnn::check_launch("pack_afrags_f16");
}

extern "C" int mrcnn_bottleneck_c2_f16_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes,
                                                 int32_t has_downsample) {
    if (batch < 1 || height < 1 || width < 1 || planes != 64) return 0;
    if (cin != (has_downsample ? 64 : 256)) return 0;
    const long long px = static_cast<long long>(batch) * height * width;
    if (px * 256 * 2 >= (1LL << 31)) return 0;   // 32-bit byte offsets into y (and x)
    const long long tiles = static_cast<long long>(batch) * ((height + TH - 1) / TH) * ((width + TW - 1) / TW);
    return tiles < (1LL << 28) ? 1 : 0;
}

extern "C" int mrcnn_bottleneck_c2_f16(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                       const void* w1_frags, const float* s1, const float* t1, const void* w2_frags,
                                       const float* s2, const float* t2, const void* w3_frags, const float* s3, const float* t3,
                                       const void* wd_frags, const float* sd, const float* td, void* y_f16,
                                       mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_f16 && w1_frags && w2_frags && w3_frags && y_f16 && s1 && t1 && s2 && t2 && s3 && t3,
                  "bottleneck_c2_f16: null pointer");
    MRCNN_REQUIRE(!wd_frags || (sd && td), "bottleneck_c2_f16: the downsample branch needs its scale and shift");
    if (!mrcnn_bottleneck_c2_f16_supported(batch, height, width, cin, 64, wd_frags ? 1 : 0))
        return mrcnn::fail(MRCNN_ERR_UNSUPPORTED,
                           "bottleneck_c2_f16: needs planes 64, stride 1, Cin 256 (identity) or 64 (with the downsample branch), "
                           "32-bit byte offsets (got %d x %d x %d x %d)", batch, height, width, cin);
    BParams p{};
    p.x = static_cast<const _Float16*>(x_f16);
    p.w1 = static_cast<const _Float16*>(w1_frags);
    p.w2 = static_cast<const _Float16*>(w2_frags);
    p.w3 = static_cast<const _Float16*>(w3_frags);
    p.wd = static_cast<const _Float16*>(wd_frags);
    p.s1 = s1; p.t1 = t1; p.s2 = s2; p.t2 = t2; p.s3 = s3; p.t3 = t3; p.sd = sd; p.td = td;
    p.y = static_cast<_Float16*>(y_f16);
    p.B = batch; p.H = height; p.W = width;
    p.tiles_x = (width + TW - 1) / TW;
    p.tiles_y = (height + TH - 1) / TH;
    p.tiles = batch * p.tiles_x * p.tiles_y;
    p.x_bytes = static_cast<unsigned>(static_cast<long long>(batch) * height * width * cin * 2);
    p.y_bytes = static_cast<unsigned>(static_cast<long long>(batch) * height * width * 256 * 2);
    const int cus = mrcnn::device_cu_count() > 0 ? mrcnn::device_cu_count() : 256;
    const int per_xcd = std::max(1, std::min(cus / 8, (p.tiles + 7) / 8));
    const dim3 grid(8 * per_xcd);
    hipStream_t s = mrcnn::as_stream(stream);
    if (wd_frags) {
        if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_c2_f16<true>), LDS_BYTES, "bottleneck_c2_f16")) return rc;
        hipLaunchKernelGGL((bottleneck_c2_f16<true>), grid, dim3(512), LDS_BYTES, s, p);
    } else {
        if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_c2_f16<false>), LDS_BYTES, "bottleneck_c2_f16")) return rc;
        hipLaunchKernelGGL((bottleneck_c2_f16<false>), grid, dim3(512), LDS_BYTES, s, p);
    }
    return mrcnn::check_launch("bottleneck_c2_f16");
}
