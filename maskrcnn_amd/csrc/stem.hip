// The ResNet stem (model.py:223-226): conv 7x7 stride 2 pad 3, 3 (padded to 4) -> 64 channels, + BN + ReLU, as its own
// kernel. In the generic implicit-GEMM kernel this layer gathers one 16-byte tap per lane straight from global memory
// with per-slot tap decoding and runs at 53 TFLOP/s (0.75 ms per batch of eight 1024^2 images, 3 % of the step).
// Here a persistent workgroup keeps the whole 64 x 7 x 7 x 4 filter in LDS (50 KB, loaded once), stages the 37 x 37
// pixel input patch of a 16 x 16 output tile (22 KB) per tile, and the main loop is 49 fully unrolled taps of
// { 4 ds_read_b64 with compile-time offsets, 8 MFMAs } with no address arithmetic at all.
//   MFMA      v_mfma_f32_32x32x2_f32, exact fp32. A tap contributes k = 4 channels: lane half h holds channels
//             (2h, 2h+1) of both operands, MFMA step s contracts channel s (lanes 0-31) with channel 2+s (lanes 32-63).
//   tile      16 x 16 output pixels x 64 channels per workgroup; wave w owns output rows 4w..4w+3 (two 32-pixel MFMA
//             tiles) x two 32-channel tiles: 64 accumulator registers. 72 KB of LDS: two workgroups per CU.
//   epilogue  scale/shift (folded BN) + ReLU, 128-byte channel runs.
#include "conv_common.hpp"

namespace {

using namespace mrcnn_conv;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StemParams {
    const float* x;      // [B][H][W][4], or the molded image itself [B][3][H][W] (NCHW template variant)
    const float* w;      // [64][7][7][4] (OHWI, channel 3 zero)
    const float* scale;  // [64] or null
    const float* shift;  // [64] or null
    float* y;            // [B][OH][OW][64]
    int B, H, W, OH, OW, tiles_x, tiles_y, tiles, act;
    unsigned x_bytes, y_bytes;
};

constexpr int TS = 16;                // output tile side
constexpr int PS = (TS - 1) * 2 + 7;  // input patch side: 37
constexpr int PITCH = PS;             // pixels per patch row
constexpr int PATCH_FLOATS = PS * PITCH * 4;
constexpr int W_FLOATS = 49 * 64 * 4;
constexpr size_t STEM_LDS = sizeof(float) * (PATCH_FLOATS + W_FLOATS);

// NCHW: the input is the molded image [B][3][H][W] as the boundary hands it over (model.py:1102-1110): the patch is staged
// plane by plane with 4-byte loads (consecutive lanes = consecutive x of one plane) into the same [37][37][4] LDS image, whose
// fourth channel is zeroed once — the separate NCHW -> NHWC4 pass over the image (0.23 ms per batch of eight 1024^2 images)
// is gone; the values the MFMAs see, and therefore the results, are the same bit for bit.
// OUT16: the output is stored as fp16 (the "f16" mode: the products stay exact fp32, one rounding at the store): adjacent lanes
// (adjacent channels) swap one value per pixel pair by DPP and each stores a 4-byte channel pair.
template <bool NCHW, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void stem7x7_s2_f32(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;              // [49 taps][64 channels][4]
    float* Pl = smem + W_FLOATS;   // [37][37][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;

    // the filter, once per workgroup: global [n][tap][4] -> LDS [tap][n][4]
    for (int i = tid; i < 49 * 64; i += 256) {
        const int n = i / 49, tap = i - n * 49;
        *reinterpret_cast<float4*>(Wl + (tap * 64 + n) * 4) = reinterpret_cast<const float4*>(p.w)[i];
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    float sc[2], sh[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        sc[ct] = p.scale ? p.scale[ct * 32 + ln] : 1.0f;
        sh[ct] = p.shift ? p.shift[ct * 32 + ln] : 0.0f;
    }
    // operand base addresses (floats): pixel tile pt = rows 4*wave + 2*pt, +1; lane -> (row ln >> 4, column ln & 15)
    int a_base[2], b_base[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int orow = 4 * wave + 2 * pt + (ln >> 4), ocol = ln & 15;
        a_base[pt] = ((orow * 2) * PITCH + ocol * 2) * 4 + lh * 2;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) b_base[ct] = (ct * 32 + ln) * 4 + lh * 2;

    if constexpr (NCHW) {
        for (int i = tid; i < PS * PS; i += 256) Pl[i * 4 + 3] = 0.f;
    }
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int b = tile / (p.tiles_y * p.tiles_x), rem = tile - b * p.tiles_y * p.tiles_x;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * TS, ox0 = tx * TS;
        const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
        __syncthreads();  // the previous tile's patch is no longer read (also orders the filter stores on the first trip)
        if constexpr (!NCHW) {
            for (int i = tid; i < PS * PS; i += 256) {
                const int py = i / PS, px = i - py * PS;
                const int iy = iy0 + py, ix = ix0 + px;
                const bool ok = static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                const unsigned off = ok ? static_cast<unsigned>((b * p.H + iy) * p.W + ix) * 16u : OOB;
                *reinterpret_cast<u32x4*>(Pl + (py * PITCH + px) * 4) =
                    __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(off), 0, 0);
            }
        } else {
            constexpr int NL = (3 * PS * PS + 255) / 256;  // 17 loads per thread, all issued before the first LDS store
            unsigned v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = tid + 256 * j;
                const int c = i / (PS * PS), r = i - c * (PS * PS), py = r / PS, px = r - py * PS;
                const int iy = iy0 + py, ix = ix0 + px;
                const bool ok = i < 3 * PS * PS && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                const unsigned off = ok ? static_cast<unsigned>(((b * 3 + c) * p.H + iy) * p.W + ix) * 4u : OOB;
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = tid + 256 * j;
                const int c = i / (PS * PS), r = i - c * (PS * PS);
                if (i < 3 * PS * PS) Pl[r * 4 + c] = __uint_as_float(v[j]);
            }
        }
        __syncthreads();

        f32x16 acc[2][2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][ct][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 49; ++tap) {
            const int ky = tap / 7, kx = tap - ky * 7;
            f32x2 a[2], bw[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) a[pt] = *reinterpret_cast<const f32x2*>(Pl + a_base[pt] + (ky * PITCH + kx) * 4);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bw[ct] = *reinterpret_cast<const f32x2*>(Wl + b_base[ct] + tap * 64 * 4);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[pt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s == 0 ? a[pt].x : a[pt].y,
                                                                          s == 0 ? bw[ct].x : bw[ct].y, acc[pt][ct], 0, 0, 0);
        }
        // epilogue: accumulator row r of pixel tile pt = pixel (r&3) + 8*(r>>2) + 4*lh of the tile's 32
        if constexpr (!OUT16) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int oy = oy0 + 4 * wave + 2 * pt + (q >> 4), ox = ox0 + (q & 15);
                    const bool ok = oy < p.OH && ox < p.OW;
                    const unsigned row = static_cast<unsigned>((b * p.OH + oy) * p.OW + ox) * 256u;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        float v = acc[pt][ct][r] * sc[ct] + sh[ct];
                        if (p.act) v = v > 0.f ? v : 0.f;
                        const unsigned o = ok ? row + static_cast<unsigned>(ct * 32 + ln) * 4u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(o), 0, 0);
                    }
                }
        } else {
            const bool odd = ln & 1;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    // rows 2 rp and 2 rp + 1 are neighbouring pixels; the even lane stores the first, the odd lane the second
                    const int r0 = 2 * rp, rmine = r0 + (odd ? 1 : 0);
                    const int q = (rmine & 3) + 8 * (rmine >> 2) + 4 * lh;
                    const int oy = oy0 + 4 * wave + 2 * pt + (q >> 4), ox = ox0 + (q & 15);
                    const bool ok = oy < p.OH && ox < p.OW;
                    const unsigned row = static_cast<unsigned>((b * p.OH + oy) * p.OW + ox) * 128u;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        float v0 = acc[pt][ct][r0] * sc[ct] + sh[ct], v1 = acc[pt][ct][r0 + 1] * sc[ct] + sh[ct];
                        if (p.act) {
                            v0 = v0 > 0.f ? v0 : 0.f;
                            v1 = v1 > 0.f ? v1 : 0.f;
                        }
                        asm volatile("" : "+v"(v0), "+v"(v1));  // fp32 first, then one rounding to fp16
                        const unsigned h0 = __builtin_bit_cast(unsigned short, static_cast<_Float16>(v0));
                        const unsigned h1 = __builtin_bit_cast(unsigned short, static_cast<_Float16>(v1));
                        const unsigned keep = odd ? h1 : h0;
                        const unsigned got = static_cast<unsigned>(
                            __builtin_amdgcn_update_dpp(0, static_cast<int>(odd ? h0 : h1), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                        const unsigned word = odd ? (got | (keep << 16)) : (keep | (got << 16));
                        const unsigned o = ok ? row + static_cast<unsigned>(ct * 32 + (ln & ~1)) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(word, y_rsrc, static_cast<int>(o), 0, 0);
                    }
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The "f16" mode's stem (BASELINE configs[4], "fp16 MFMA path"): conv 7x7 s2 + BN + ReLU + SamePad(3,2) + MaxPool 3x3 s2
// (model.py:223-229) in ONE kernel on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) — round 4. The exact-fp32 stem
// above plus the separate max-pool cost 0.33 + 0.09 ms of that mode's ~10 ms step, nearly all of it fp32-MFMA time for a
// layer whose fp16-MFMA time is ~16x smaller; with fp16 operands the layer is bound by its bytes: the fp32 NCHW image in
// (read once) and the POOLED fp16 map out — the 64-channel full-resolution map (the largest tensor of the step) never exists.
//   tile      a persistent workgroup of NINE waves (two of the 18 MFMA row tiles each) owns 8 x 16 POOLED pixels = 17 x 33 conv outputs (one row / column of
//             overlap with the next tile: 1.10x the MFMA work, which does not matter here) x all 64 channels
//   K         per filter row ky: (kx, c) with kx padded 7 -> 8 and c padded 3 -> 4: 32 halves = two MFMA k steps. A conv
//             pixel's A fragment for (ky, step, lane half h) is the 16 contiguous bytes of patch pixels 2 ox + 4 step + 2 h, + 1
//             (8 bytes per pixel) — one ds_read_b128, conflict-free for 32 consecutive ox
//   patch     39 x 72 pixels x 4 halves staged from the NCHW fp32 image (4-byte loads, consecutive lanes = consecutive x),
//             rounded to fp16 once; the weights ([ky][n][32 halves], 16-byte chunks XOR-swizzled by (n >> 2) & 3) are converted
//             once per workgroup
//   epilogue  affine + ReLU in fp32, one rounding to fp16 into an LDS image [conv pixel][64] with zeros where the conv pixel
//             lies outside the conv map (the pool's zero padding; ReLU'd values are >= 0); then every thread takes a channel
//             pair of eight pooled pixels: nine 4-byte LDS reads and v_pk_max_f16 each, 128-byte output rows
// Requires the ReLU (zero padding is only neutral for non-negative values) and H, W multiples of 4 (even conv sizes: SamePad2d
// (3, 2) then pads bottom / right only).
typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sp_f16x4 __attribute__((ext_vector_type(4)));

struct StemPoolParams {
    const float* x;      // [B][3][H][W]
    const float* w;      // [64][7][7][4] fp32 OHWI (channel 3 zero)
    const float* scale;
    const float* shift;
    _Float16* y;         // [B][POH][POW][64]
    int B, H, W, OH, OW, POH, POW, tiles_x, tiles_y, tiles;
    unsigned x_bytes, y_bytes;
};

constexpr int SP_PH = 8, SP_PW = 16;                       // pooled pixels per tile
constexpr int SP_CH = 2 * SP_PH + 1, SP_CW = 2 * SP_PW + 1;  // conv outputs per tile: 17 x 33
constexpr int SP_NPX = SP_CH * SP_CW;                      // 561
constexpr int SP_NRT = (SP_NPX + 31) / 32;                 // 18 MFMA row tiles
constexpr int SP_IH = (SP_CH - 1) * 2 + 7;                 // 39 patch rows
constexpr int SP_IW = 72;                                  // patch columns: (33 - 1) * 2 + 7 = 71, + the zero-weight tap kx = 7
constexpr int SP_W_HALVES = 7 * 64 * 32;
constexpr int SP_P_HALVES = SP_IH * SP_IW * 4;
constexpr int SP_C_HALVES = SP_NRT * 32 * 64;
constexpr size_t STEM_POOL_LDS = sizeof(_Float16) * (SP_W_HALVES + SP_P_HALVES + SP_C_HALVES);
static_assert(STEM_POOL_LDS <= 160 * 1024, "LDS");

constexpr int SP_THREADS = 576;   // nine waves: two of the tile's 18 MFMA row tiles each
__global__ __launch_bounds__(SP_THREADS, 1) void stem7x7_s2_pool_f16(const StemPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* Wl = reinterpret_cast<_Float16*>(smem_raw);    // [7][64][32] (chunk-swizzled)
    _Float16* Pl = Wl + SP_W_HALVES;                          // [39][72][4]
    _Float16* Cl = Pl + SP_P_HALVES;                          // [576][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;

    // the filter, once per workgroup: fp32 [n][ky][kx][4] -> fp16 [ky][n][kx 0..7][4], kx = 7 zero
    for (int i = tid; i < 7 * 64 * 8; i += SP_THREADS) {
        const int kx = i & 7, n = (i >> 3) & 63, ky = i >> 9;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kx < 7) v = reinterpret_cast<const float4*>(p.w)[(n * 7 + ky) * 7 + kx];
        const int chunk = (kx >> 1) ^ ((n >> 2) & 3);   // 16-byte chunk = two taps
        sp_f16x4 h4 = {static_cast<_Float16>(v.x), static_cast<_Float16>(v.y), static_cast<_Float16>(v.z), static_cast<_Float16>(v.w)};
        *reinterpret_cast<sp_f16x4*>(Wl + (ky * 64 + n) * 32 + chunk * 8 + (kx & 1) * 4) = h4;
    }
    // channel 3 of every patch pixel and the last patch column (only ever multiplied by zero weights) stay zero
    for (int i = tid; i < SP_IH * SP_IW; i += SP_THREADS) *reinterpret_cast<sp_f16x4*>(Pl + i * 4) = sp_f16x4{0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    float sc[2], sh[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        sc[ct] = p.scale ? p.scale[ct * 32 + ln] : 1.0f;
        sh[ct] = p.shift ? p.shift[ct * 32 + ln] : 0.0f;
    }
    // this wave's row tiles: wave and wave + 9. A-fragment base (halves) of conv pixel q = 32 rt + ln: patch pixel
    // (2 oy, 2 ox + 2 lh); pixels beyond the 561 of the tile read pixel 560's (discarded)
    constexpr int NRW = 2;
    int a_base[NRW];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        const int q = min((wave + 9 * i) * 32 + ln, SP_NPX - 1);
        const int oy = q / SP_CW, ox = q - oy * SP_CW;
        a_base[i] = ((2 * oy) * SP_IW + 2 * ox + 2 * lh) * 4;
    }
    // B fragment of channel ct * 32 + ln, k step s: chunk 2 s + lh, swizzled by the channel
    int b_off[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int n = ct * 32 + ln;
            b_off[ct][s2] = n * 32 + ((2 * s2 + lh) ^ ((n >> 2) & 3)) * 8;
        }

    // the patch of a tile: 39 x 71 pixels x 3 planes as 4-byte loads (consecutive lanes = consecutive x of one plane), 15 per
    // thread. They are issued ONE TILE AHEAD — right after the current tile's patch has been written to LDS — and ride in
    // registers through its MFMAs, conv image and pool. Which element a thread's j-th load is never changes: its offset relative
    // to the tile's origin and its place in the LDS patch are computed once (30 registers) — recomputed per tile, the divisions
    // by 39 * 71 and 71 made the kernel VALU-bound.
    // barriers are LDS-only: __syncthreads() also waits for vector memory (vmcnt(0)), i.e. for the prefetched patch and for the
    // pool's stores
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MRCNN_SYNC_FUZZ_POINT();
    };
    constexpr int NE = 3 * SP_IH * (SP_IW - 1);
    constexpr int NL = (NE + SP_THREADS - 1) / SP_THREADS;   // 15
    int s_rel[NL];        // (c * H + py) * W + px
    unsigned s_pk[NL];    // LDS index (py * 72 + px) * 4 + c | py << 14 | px << 20; element beyond the patch: all ones
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + SP_THREADS * j;
        const int c = i / (SP_IH * (SP_IW - 1)), r = i - c * (SP_IH * (SP_IW - 1));
        const int py = r / (SP_IW - 1), px = r - py * (SP_IW - 1);
        s_rel[j] = (c * p.H + py) * p.W + px;
        s_pk[j] = i < NE ? static_cast<unsigned>((py * SP_IW + px) * 4 + c) | (py << 14) | (px << 20) : 0xFFFFFFFFu;
    }
    unsigned pv[NL];
    auto load_patch = [&](int tile_) {
        const bool live = tile_ < p.tiles;
        const int b_ = tile_ / (p.tiles_y * p.tiles_x), rem_ = tile_ - b_ * p.tiles_y * p.tiles_x;
        const int ty_ = rem_ / p.tiles_x, tx_ = rem_ - ty_ * p.tiles_x;
        const int iy0 = ty_ * 2 * SP_PH * 2 - 3, ix0 = tx_ * 2 * SP_PW * 2 - 3;
        const int base = (b_ * 3 * p.H + iy0) * p.W + ix0;                    // wave-uniform; may be negative at the borders
        // a tile whose whole patch lies inside the image needs no per-element test (most tiles)
        const bool inside = live && iy0 >= 0 && ix0 >= 0 && iy0 + SP_IH <= p.H && ix0 + SP_IW - 1 <= p.W;
        if (inside) {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const unsigned off = s_pk[j] != 0xFFFFFFFFu ? static_cast<unsigned>(base + s_rel[j]) * 4u : OOB;
                pv[j] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int iy = iy0 + static_cast<int>((s_pk[j] >> 14) & 63), ix = ix0 + static_cast<int>((s_pk[j] >> 20) & 127);
                const bool ok = live && s_pk[j] != 0xFFFFFFFFu && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                const unsigned off = ok ? static_cast<unsigned>(base + s_rel[j]) * 4u : OOB;
                pv[j] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), 0, 0);
            }
        }
    };
    load_patch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int b = tile / (p.tiles_y * p.tiles_x), rem = tile - b * p.tiles_y * p.tiles_x;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int cy0 = ty * 2 * SP_PH, cx0 = tx * 2 * SP_PW;   // first conv output of the tile
        lds_barrier();  // the previous tile's patch and conv image are no longer read (first trip: the filter / zero stores)
        // an opaque copy of the thread id per tile: the epilogue / pool index arithmetic below is invariant across tiles, and
        // computed once ahead of this loop it costs ~100 registers that the allocator then spills
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int ln_ = t_ & 31, lh_ = (t_ >> 5) & 1, wave_ = t_ >> 6;
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (s_pk[j] != 0xFFFFFFFFu) Pl[s_pk[j] & 0x3FFF] = static_cast<_Float16>(__uint_as_float(pv[j]));
        load_patch(tile + gridDim.x);   // the next tile's, in flight from here
        lds_barrier();

        f32x16 acc[NRW][2];
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.f;
        // (one filter row per trip: fully unrolled, the compiler hoists all 70 fragment reads — 280 registers — to the top)
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                sp_f16x8 bf[2], af[NRW];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bf[ct] = *reinterpret_cast<const sp_f16x8*>(Wl + ky * 64 * 32 + b_off[ct][s2]);
#pragma unroll
                for (int i = 0; i < NRW; ++i)
                    af[i] = *reinterpret_cast<const sp_f16x8*>(Pl + a_base[i] + (ky * SP_IW + 4 * s2) * 4);
#pragma unroll
                for (int i = 0; i < NRW; ++i)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[ct], acc[i][ct], 0, 0, 0);
            }
        // conv image: affine + ReLU, zero outside the conv map, one rounding to fp16. Row r of a 32-pixel row tile is pixel
        // q0 + d_r with d_r a compile-time constant (<= 27 < 33: at most one wrap into the next conv row); the LDS address is
        // one base + immediates
        const bool interior = cy0 + SP_CH <= p.OH && cx0 + SP_CW <= p.OW;   // wave-uniform: no per-pixel test but q < 561
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int q0 = (wave_ + 9 * i) * 32 + 4 * lh_;
            const int oy0 = (q0 * 1986) >> 16, ox0 = q0 - oy0 * SP_CW;   // q0 / 33 for q0 < 576
            _Float16* crow = Cl + q0 * 64 + ln_;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2);
                bool in = q0 + d < SP_NPX;
                if (!interior) {
                    const int oxr = ox0 + d, wrap = oxr >= SP_CW ? 1 : 0;
                    in = in && cy0 + oy0 + wrap < p.OH && cx0 + oxr - wrap * SP_CW < p.OW;
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float v = acc[i][ct][r] * sc[ct] + sh[ct];
                    v = v > 0.f ? v : 0.f;
                    v = in ? v : 0.f;
                    asm volatile("" : "+v"(v));   // fp32 first, then ONE rounding to fp16 (no fused mixed-precision fma)
                    crow[d * 64 + ct * 32] = static_cast<_Float16>(v);
                }
            }
        }
        lds_barrier();
        // pool: thread = (channel pair, 7 or 8 pooled pixels)
        {
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int cp = t2 & 31, g = t2 >> 5;   // 18 groups of 32 lanes over the 128 pooled pixels
#pragma unroll 1
            for (int pp = g; pp < SP_PH * SP_PW; pp += SP_THREADS / 32) {
                const int py = pp >> 4, px = pp & 15;
                sp_f16x2 m = {0, 0};   // the values are >= 0
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const sp_f16x2 v = *reinterpret_cast<const sp_f16x2*>(Cl + ((2 * py + dy) * SP_CW + 2 * px + dx) * 64 + 2 * cp);
                        m = __builtin_elementwise_max(m, v);
                    }
                const int gy = ty * SP_PH + py, gx = tx * SP_PW + px;
                const bool ok = gy < p.POH && gx < p.POW;
                const unsigned off = ok ? static_cast<unsigned>(((b * p.POH + gy) * p.POW + gx) * 64 + 2 * cp) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), y_rsrc, static_cast<int>(off), 0, 0);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The exact-fp32 stem WITH its max-pool (round 5): conv 7x7 s2 p3 + BN + ReLU + SamePad2d(3, 2) + MaxPool2d(3, 2)
// (model.py:223-229) in ONE launch on the fp32 MFMA. The two-launch form writes the 64-channel full-resolution map (537 MB per
// batch of eight 1024^2 images: the largest tensor of the step) and reads it back in the pool; here it never exists.
//   K         THREE real channels: per filter row ky the 21 values (kx, c) are contiguous in a patch row stored [py][px][3], so a
//             conv pixel's A operand for k quad j of row ky is the 8 bytes at patch float 6 ox + 4 j + 2 h — five quads of two MFMAs
//             (k = 4 j + s | 4 j + 2 + s) + one single MFMA (k = 20 | 21, the latter with a zero weight) = 77 MFMAs per 32 x 32
//             output block instead of the 98 the four-channel taps of stem7x7_s2_f32 need.
//   tile      a persistent four-wave workgroup (one wave per SIMD, all 512 registers) owns 7 x 16 POOLED pixels = 15 x 33 conv
//             outputs (one row / column of overlap with the next tile) = 495 pixels = 16 MFMA row tiles, four per wave, and walks
//             the two 32-channel halves one after the other (pass ct): 64 accumulator registers, the conv image of a pass goes to
//             LDS as fp32 [512 pixels][32 channels], zero where the conv pixel lies outside the conv map (the pool's zero padding:
//             ReLU'd values are >= 0), and every thread pools (pooled pixel, four channels): nine 16-byte LDS reads, 16-byte stores.
//   LDS       filter [7][6][64][4] 43 KB (once per workgroup) + patch [35][240] floats 33.6 KB (pitch = 16 mod 32 floats: the
//             8-byte operand reads of 32 consecutive conv pixels, 24 bytes apart, are conflict-free) + conv image 64 KB.
//   patch     staged from the NCHW fp32 image with 4-byte loads (consecutive lanes = consecutive x of one plane), issued ONE TILE
//             AHEAD and riding in registers through the current tile's MFMAs and pools (as stem7x7_s2_pool_f16).
// Requires the ReLU and H, W multiples of 4 (even conv sizes: SamePad2d(3, 2) then pads bottom / right only).
struct StemPool32Params {
    const float* x;      // [B][3][H][W]
    const float* w;      // [64][7][7][4] fp32 OHWI (channel 3 zero)
    const float* scale;
    const float* shift;
    float* y;            // [B][POH][POW][64]
    int B, H, W, OH, OW, POH, POW, tiles_x, tiles_y, tiles;
    unsigned x_bytes, y_bytes;
};

constexpr int S3_PH = 7, S3_PW = 16;                          // pooled pixels per tile
constexpr int S3_CH = 2 * S3_PH + 1, S3_CW = 2 * S3_PW + 1;   // conv outputs per tile: 15 x 33
constexpr int S3_NPX = S3_CH * S3_CW;                         // 495
constexpr int S3_NRT = (S3_NPX + 31) / 32;                    // 16 MFMA row tiles
constexpr int S3_IH = (S3_CH - 1) * 2 + 7;                    // 35 patch rows
constexpr int S3_IW = (S3_CW - 1) * 2 + 7;                    // 71 patch pixels per row
constexpr int S3_PITCH = 240;                                 // floats per patch row (213 used; = 16 mod 32)
constexpr int S3_KQ = 6;                                      // k quads per filter row (the sixth holds k = 20 only)
constexpr int S3_W_FLOATS = 7 * S3_KQ * 64 * 4;
constexpr int S3_P_FLOATS = S3_IH * S3_PITCH;
constexpr int S3_CP = 32;                                     // floats per conv-image pixel (one 32-channel half)
constexpr int S3_C_FLOATS = S3_NRT * 32 * S3_CP;
constexpr size_t STEM_POOL32_LDS = sizeof(float) * (S3_W_FLOATS + S3_P_FLOATS + S3_C_FLOATS);
static_assert(S3_NRT == 16 && STEM_POOL32_LDS <= 160 * 1024, "tile / LDS");
static_assert(S3_PITCH % 32 == 16 && S3_PITCH >= 3 * S3_IW + 3, "patch pitch");

__global__ __launch_bounds__(256, 1) void stem7x7_s2_pool_f32(const StemPool32Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                   // [7 rows][6 quads][64 channels][4]
    float* Pl = Wl + S3_W_FLOATS;       // [35][240]: pixel px, channel c at float 3 px + c
    float* Cl = Pl + S3_P_FLOATS;       // [512 conv pixels][32 channels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;

    // the filter, once per workgroup: global [n][ky][kx][4] -> LDS [ky][quad][n][e], k = 4 quad + e = 3 kx + c (k >= 21: zero)
    for (int i = tid; i < S3_W_FLOATS; i += 256) {
        const int e = i & 3, n = (i >> 2) & 63, slot = i >> 8;
        const int ky = slot / S3_KQ, j = slot - ky * S3_KQ, k = 4 * j + e;
        float v = 0.f;
        if (k < 21) {
            const int kx = k / 3, c = k - 3 * kx;
            v = p.w[((n * 7 + ky) * 7 + kx) * 4 + c];
        }
        Wl[i] = v;
    }
    // the pad floats of every patch row (read by the last quads of the last conv column against zero weights) stay zero
    for (int i = tid; i < S3_P_FLOATS; i += 256) Pl[i] = 0.f;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // this wave's row tiles: wave + 4 i. A-operand base (floats) of conv pixel q = 32 rt + ln: patch row 2 oy, float 6 ox + 2 lh;
    // pixels beyond the 495 of the tile read the last pixel's (discarded)
    constexpr int NRW = 4;
    int a_base[NRW];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        const int q = min((wave + 4 * i) * 32 + ln, S3_NPX - 1);
        const int oy = q / S3_CW, ox = q - oy * S3_CW;
        a_base[i] = (2 * oy) * S3_PITCH + 6 * ox + 2 * lh;
    }
    const int b_base = ln * 4 + lh * 2;   // + 32 * 4 for the second channel half

    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MRCNN_SYNC_FUZZ_POINT();
    };
    // patch elements of a thread: fixed offsets relative to the tile's origin and fixed places in the LDS patch (computed once)
    constexpr int NE = 3 * S3_IH * S3_IW;                     // 7455
    constexpr int NL = (NE + 255) / 256;                      // 30
    // Only the last of a thread's NL elements can lie beyond the patch (NE = 29 * 256 + 31): it then loads nothing (offset out of
    // range -> 0) and writes that zero to a pad float of patch row 0 — no per-element branch anywhere.
    static_assert((NL - 1) * 256 < NE, "every element but a thread's last is inside the patch");
    const bool tail_ok = tid + 256 * (NL - 1) < NE;
    int s_rel[NL];        // (c * H + py) * W + px
    unsigned s_pk[NL];    // LDS float index py * 240 + 3 px + c | py << 14 | px << 20
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + 256 * j;
        const int c = i / (S3_IH * S3_IW), r = i - c * (S3_IH * S3_IW);
        const int py = r / S3_IW, px = r - py * S3_IW;
        s_rel[j] = (c * p.H + py) * p.W + px;
        s_pk[j] = i < NE ? static_cast<unsigned>(py * S3_PITCH + 3 * px + c) | (py << 14) | (px << 20)
                         : static_cast<unsigned>(S3_PITCH - 1);
    }
    unsigned pv[NL];
    auto load_patch = [&](int tile_) {
        // the per-element tables are made opaque once per call: everything derived from them (the (py, px) bit fields, LDS
        // addresses) is otherwise hoisted out of the tile loop — ~150 loop-invariant values that the allocator then spills
        // (SGPR pairs to VGPR lanes: 1 400 v_readlane / v_writelane in the first build)
#pragma unroll
        for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(s_pk[j]), "+v"(s_rel[j]));
        const bool live = tile_ < p.tiles;
        const int b_ = tile_ / (p.tiles_y * p.tiles_x), rem_ = tile_ - b_ * p.tiles_y * p.tiles_x;
        const int ty_ = rem_ / p.tiles_x, tx_ = rem_ - ty_ * p.tiles_x;
        const int iy0 = ty_ * 2 * S3_PH * 2 - 3, ix0 = tx_ * 2 * S3_PW * 2 - 3;
        const int base = (b_ * 3 * p.H + iy0) * p.W + ix0;                    // wave-uniform; may be negative at the borders
        const bool inside = live && iy0 >= 0 && ix0 >= 0 && iy0 + S3_IH <= p.H && ix0 + S3_IW <= p.W;
        if (inside) {   // a tile whose whole patch lies inside the image needs no per-element test (most tiles)
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                unsigned off = static_cast<unsigned>(base + s_rel[j]) * 4u;
                if (j == NL - 1) off = tail_ok ? off : OOB;
                pv[j] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int iy = iy0 + static_cast<int>((s_pk[j] >> 14) & 63), ix = ix0 + static_cast<int>((s_pk[j] >> 20) & 127);
                bool ok = static_cast<int>(live) & static_cast<int>(static_cast<unsigned>(iy) < static_cast<unsigned>(p.H)) &
                          static_cast<int>(static_cast<unsigned>(ix) < static_cast<unsigned>(p.W));
                if (j == NL - 1) ok = static_cast<int>(ok) & static_cast<int>(tail_ok);
                const unsigned off = ok ? static_cast<unsigned>(base + s_rel[j]) * 4u : OOB;
                pv[j] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), 0, 0);
            }
        }
    };
    load_patch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int b = tile / (p.tiles_y * p.tiles_x), rem = tile - b * p.tiles_y * p.tiles_x;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        lds_barrier();  // the previous tile's patch and conv image are no longer read (first trip: the filter / zero stores)
#pragma unroll
        for (int j = 0; j < NL; ++j) Pl[s_pk[j] & 0x3FFF] = __uint_as_float(pv[j]);
        load_patch(tile + gridDim.x);   // the next tile's, in flight from here
        lds_barrier();

#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            f32x16 acc[NRW];
#pragma unroll
            for (int i = 0; i < NRW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const float* wl = Wl + b_base + ct * 32 * 4;
            // One wave per SIMD: nothing else covers an LDS round trip, so the operands of step s + 1 are read BEFORE the MFMAs of
            // step s (two register sets; the sched_barriers pin the order the source states).
            f32x2 a[2][NRW], bw[2];
            auto fetch = [&](int step, int set) {
                const int ky = step / S3_KQ, j = step - ky * S3_KQ;
                bw[set] = *reinterpret_cast<const f32x2*>(wl + step * 64 * 4);
#pragma unroll
                for (int i = 0; i < NRW; ++i) a[set][i] = *reinterpret_cast<const f32x2*>(Pl + a_base[i] + ky * S3_PITCH + 4 * j);
            };
            fetch(0, 0);
#pragma unroll
            for (int step = 0; step < 7 * S3_KQ; ++step) {
                const int cur = step & 1;
                if (step + 1 < 7 * S3_KQ) fetch(step + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NRW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].x, bw[cur].x, acc[i], 0, 0, 0);
                if (step % S3_KQ < S3_KQ - 1) {
#pragma unroll
                    for (int i = 0; i < NRW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].y, bw[cur].y, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // conv image of this channel half: affine + ReLU. Row r of a 32-pixel row tile is pixel q0 + d_r, d_r a compile-time
            // constant: one LDS base + immediates. (Conv pixels beyond the conv map hold whatever the zero-filled patch gives; the
            // pool below never lets them through.)
            const float sc = p.scale ? p.scale[ct * 32 + ln] : 1.0f, sh = p.shift ? p.shift[ct * 32 + ln] : 0.0f;
#pragma unroll
            for (int i = 0; i < NRW; ++i) {
                float* crow = Cl + ((wave + 4 * i) * 32 + 4 * lh) * S3_CP + ln;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][r] * sc + sh;
                    crow[((r & 3) + 8 * (r >> 2)) * S3_CP] = v > 0.f ? v : 0.f;
                }
            }
            lds_barrier();
            // pool: thread = (pooled pixel, four channels). SamePad2d(3, 2) on an even conv size pads ONE zero row / column at the
            // bottom / right (model.py:64-87): the only window taps outside the conv map are dy = 2 of the last pooled row and
            // dx = 2 of the last pooled column — they count as 0 (a multiplication by 0 or 1: the values are finite and >= 0).
#pragma unroll 1
            for (int it = tid; it < S3_PH * S3_PW * 8; it += 256) {
                const int cq = it & 7, pp = it >> 3;
                const int py = pp >> 4, px = pp & 15;
                const int gy = ty * S3_PH + py, gx = tx * S3_PW + px;
                const float fy = 2 * gy + 2 < p.OH ? 1.f : 0.f, fx = 2 * gx + 2 < p.OW ? 1.f : 0.f;
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);   // the values are >= 0
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        float4 v = *reinterpret_cast<const float4*>(Cl + ((2 * py + dy) * S3_CW + 2 * px + dx) * S3_CP + 4 * cq);
                        if (dy == 2 || dx == 2) {
                            const float f = dy == 2 && dx == 2 ? fy * fx : dy == 2 ? fy : fx;
                            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                        }
                        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                    }
                const bool ok = gy < p.POH && gx < p.POW;
                const unsigned off = ok ? static_cast<unsigned>(((b * p.POH + gy) * p.POW + gx) * 64 + ct * 32 + 4 * cq) * 4u : OOB;
                const u32x4 o = {__float_as_uint(m.x), __float_as_uint(m.y), __float_as_uint(m.z), __float_as_uint(m.w)};
                __builtin_amdgcn_raw_buffer_store_b128(o, y_rsrc, static_cast<int>(off), 0, 0);
            }
            if (ct == 0) lds_barrier();   // the second half's conv image overwrites the first's
        }
    }
}

}  // namespace

namespace {
int run_stem(bool nchw, const float* x, int32_t batch, int32_t height, int32_t width, const float* w, const float* scale,
             const float* shift, int32_t activation, float* y, mrcnn_stream_t stream, bool out16 = false) {
    MRCNN_REQUIRE(!out16 || nchw, "stem: the fp16-output form reads the NCHW image");
    MRCNN_REQUIRE(x && w && y, "stem: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 2 && width >= 2 && height % 2 == 0 && width % 2 == 0,
                  "stem: B=%d H=%d W=%d (even sizes required)", batch, height, width);
    MRCNN_REQUIRE(activation == 0 || activation == 1, "stem: activation must be 0 or 1");
    MRCNN_REQUIRE(1LL * batch * height * width * 4 < (1LL << 30) && 1LL * batch * (height / 2) * (width / 2) * 64 < (1LL << 30),
                  "stem: tensor too large (32-bit buffer byte offsets)");
    StemParams p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.y = y;
    p.B = batch; p.H = height; p.W = width; p.OH = height / 2; p.OW = width / 2;
    p.tiles_x = (p.OW + TS - 1) / TS;
    p.tiles_y = (p.OH + TS - 1) / TS;
    p.tiles = batch * p.tiles_x * p.tiles_y;
    p.act = activation;
    p.x_bytes = static_cast<unsigned>((nchw ? 12LL : 16LL) * batch * height * width);
    p.y_bytes = static_cast<unsigned>((out16 ? 128LL : 256LL) * batch * p.OH * p.OW);
    const void* kern = out16 ? reinterpret_cast<const void*>(stem7x7_s2_f32<true, true>)
                     : nchw ? reinterpret_cast<const void*>(stem7x7_s2_f32<true>) : reinterpret_cast<const void*>(stem7x7_s2_f32<false>);
    if (int rc = mrcnn::ensure_dynamic_lds(kern, STEM_LDS, "stem")) return rc;
    const int num_cu = mrcnn::device_cu_count();
    if (num_cu <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "stem: cannot query the device");
    const int grid = p.tiles < 2 * num_cu ? p.tiles : 2 * num_cu;  // persistent: two workgroups per CU
    if (out16) hipLaunchKernelGGL((stem7x7_s2_f32<true, true>), dim3(grid), dim3(256), STEM_LDS, mrcnn::as_stream(stream), p);
    else if (nchw) hipLaunchKernelGGL(stem7x7_s2_f32<true>, dim3(grid), dim3(256), STEM_LDS, mrcnn::as_stream(stream), p);
    else hipLaunchKernelGGL(stem7x7_s2_f32<false>, dim3(grid), dim3(256), STEM_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("stem7x7_s2_f32");
}
}  // namespace

extern "C" int mrcnn_stem_conv7x7_s2_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                              const float* w, const float* scale, const float* shift,
                                              int32_t activation, float* y, mrcnn_stream_t stream) {
    return run_stem(false, x, batch, height, width, w, scale, shift, activation, y, stream);
}

extern "C" int mrcnn_stem_conv7x7_s2_nchw_f32(const float* x_nchw, int32_t batch, int32_t height, int32_t width,
                                              const float* w, const float* scale, const float* shift,
                                              int32_t activation, float* y, mrcnn_stream_t stream) {
    return run_stem(true, x_nchw, batch, height, width, w, scale, shift, activation, y, stream);
}

extern "C" int mrcnn_stem_conv7x7_s2_nchw_f16out(const float* x_nchw, int32_t batch, int32_t height, int32_t width,
                                                 const float* w, const float* scale, const float* shift,
                                                 int32_t activation, void* y_f16, mrcnn_stream_t stream) {
    return run_stem(true, x_nchw, batch, height, width, w, scale, shift, activation, static_cast<float*>(y_f16), stream, true);
}

extern "C" int mrcnn_stem_conv7x7_s2_pool_f16(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                              const float* scale, const float* shift, void* y_f16, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_nchw && w && y_f16, "stem_pool: null pointer");
    // (an odd conv size would give SamePad2d(3, 2) a top / left component — model.py:64-87 — and shift the pooling windows)
    MRCNN_REQUIRE(batch >= 1 && height >= 4 && width >= 4 && height % 4 == 0 && width % 4 == 0,
                  "stem_pool: B=%d H=%d W=%d (multiples of 4 required)", batch, height, width);
    MRCNN_REQUIRE(1LL * batch * height * width * 3 < (1LL << 30), "stem_pool: tensor too large (32-bit buffer byte offsets)");
    StemPoolParams p;
    p.x = x_nchw; p.w = w; p.scale = scale; p.shift = shift; p.y = static_cast<_Float16*>(y_f16);
    p.B = batch; p.H = height; p.W = width; p.OH = height / 2; p.OW = width / 2;
    p.POH = (p.OH + 1) / 2; p.POW = (p.OW + 1) / 2;   // SamePad2d(3, 2) + MaxPool2d(3, 2): ceil(n / 2) (model.py:64-87,227-228)
    p.tiles_x = (p.POW + SP_PW - 1) / SP_PW;
    p.tiles_y = (p.POH + SP_PH - 1) / SP_PH;
    p.tiles = batch * p.tiles_x * p.tiles_y;
    p.x_bytes = static_cast<unsigned>(12LL * batch * height * width);
    p.y_bytes = static_cast<unsigned>(128LL * batch * p.POH * p.POW);
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(stem7x7_s2_pool_f16), STEM_POOL_LDS, "stem_pool")) return rc;
    const int num_cu = mrcnn::device_cu_count();
    if (num_cu <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "stem_pool: cannot query the device");
    const int grid = p.tiles < num_cu ? p.tiles : num_cu;   // persistent: one eight-wave workgroup per CU
    hipLaunchKernelGGL(stem7x7_s2_pool_f16, dim3(grid), dim3(SP_THREADS), STEM_POOL_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("stem7x7_s2_pool_f16");
}

extern "C" int mrcnn_stem_conv7x7_s2_pool_f32(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                              const float* scale, const float* shift, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_nchw && w && y, "stem_pool_f32: null pointer");
    // (an odd conv size would give SamePad2d(3, 2) a top / left component — model.py:64-87 — and shift the pooling windows)
    MRCNN_REQUIRE(batch >= 1 && height >= 4 && width >= 4 && height % 4 == 0 && width % 4 == 0,
                  "stem_pool_f32: B=%d H=%d W=%d (multiples of 4 required)", batch, height, width);
    // input 12 B and output 16 B per input pixel (64 fp32 channels per 4 x 4 pixels): the output is the larger tensor
    MRCNN_REQUIRE(1LL * batch * height * width * 3 < (1LL << 30) && 16LL * batch * height * width < (1LL << 31),
                  "stem_pool_f32: tensor too large (32-bit buffer byte offsets: B*H*W < 2^27 pixels)");
    StemPool32Params p;
    p.x = x_nchw; p.w = w; p.scale = scale; p.shift = shift; p.y = y;
    p.B = batch; p.H = height; p.W = width; p.OH = height / 2; p.OW = width / 2;
    p.POH = (p.OH + 1) / 2; p.POW = (p.OW + 1) / 2;   // SamePad2d(3, 2) + MaxPool2d(3, 2): ceil(n / 2) (model.py:64-87,227-228)
    p.tiles_x = (p.POW + S3_PW - 1) / S3_PW;
    p.tiles_y = (p.POH + S3_PH - 1) / S3_PH;
    p.tiles = batch * p.tiles_x * p.tiles_y;
    p.x_bytes = static_cast<unsigned>(12LL * batch * height * width);
    p.y_bytes = static_cast<unsigned>(256LL * batch * p.POH * p.POW);
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(stem7x7_s2_pool_f32), STEM_POOL32_LDS, "stem_pool_f32")) return rc;
    const int num_cu = mrcnn::device_cu_count();
    if (num_cu <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "stem_pool_f32: cannot query the device");
    const int grid = p.tiles < num_cu ? p.tiles : num_cu;   // persistent: one four-wave workgroup per CU
    hipLaunchKernelGGL(stem7x7_s2_pool_f32, dim3(grid), dim3(256), STEM_POOL32_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("stem7x7_s2_pool_f32");
}
