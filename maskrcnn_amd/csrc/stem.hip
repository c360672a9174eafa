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
