// Small HBM-bound helpers around the conv stack (gfx950): zero-padded max-pool on channels-last data
// (stem: SamePad2d(3,2) + MaxPool2d(3,2), model.py:227-228; P6 = MaxPool2d(1,2), model.py:109,161) and the
// NCHW <-> NHWC conversions at the reference boundary (model.py:1109 hands over NCHW).
#pragma clang fp contract(off)

#include "common.hpp"

namespace {

// y[b,oy,ox,c] = max over k x k window of zero-padded x. One thread = one output pixel x 4 channels.
__global__ __launch_bounds__(256) void maxpool_nhwc(const float* __restrict__ x, int B, int H, int W,
                                                    int C, int k, int stride, int pt, int pl, int OH,
                                                    int OW, float* __restrict__ y, int y_kblocked) {
    const int c4 = C >> 2;
    const int64_t total = static_cast<int64_t>(B) * OH * OW * c4;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(e % c4) * 4;
        int64_t pix = e / c4;
        const int ox = static_cast<int>(pix % OW);
        pix /= OW;
        const int oy = static_cast<int>(pix % OH);
        const int b = static_cast<int>(pix / OH);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride + ky - pt;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride + kx - pl;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // F.pad(..., 'constant', 0)
                if (static_cast<unsigned>(iy) < static_cast<unsigned>(H) &&
                    static_cast<unsigned>(ix) < static_cast<unsigned>(W))
                    v = *reinterpret_cast<const float4*>(
                        x + ((static_cast<int64_t>(b) * H + iy) * W + ix) * C + c);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        const int64_t opix = (static_cast<int64_t>(b) * OH + oy) * OW + ox;
        if (y_kblocked)  // [C/8][B*OH*OW][8]
            *reinterpret_cast<float4*>(y + ((c >> 3) * (static_cast<int64_t>(B) * OH * OW) + opix) * 8 + (c & 7)) = m;
        else
            *reinterpret_cast<float4*>(y + opix * C + c) = m;
    }
}

// The same on fp16 NHWC (the fp16-activation mode of BASELINE config 5): one thread = one output pixel x 8 channels.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void maxpool_nhwc_f16(const _Float16* __restrict__ x, int B, int H, int W, int C, int k,
                                                        int stride, int pt, int pl, int OH, int OW,
                                                        _Float16* __restrict__ y) {
    const int c8 = C >> 3;
    const int64_t total = static_cast<int64_t>(B) * OH * OW * c8;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(e % c8) * 8;
        int64_t pix = e / c8;
        const int ox = static_cast<int>(pix % OW);
        pix /= OW;
        const int oy = static_cast<int>(pix % OH);
        const int b = static_cast<int>(pix / OH);
        h16x8 m;
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = static_cast<_Float16>(-INFINITY);
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride + ky - pt;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride + kx - pl;
                h16x8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = static_cast<_Float16>(0.0f);  // F.pad(..., 'constant', 0)
                if (static_cast<unsigned>(iy) < static_cast<unsigned>(H) &&
                    static_cast<unsigned>(ix) < static_cast<unsigned>(W))
                    v = *reinterpret_cast<const h16x8*>(x + ((static_cast<int64_t>(b) * H + iy) * W + ix) * C + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
            }
        }
        *reinterpret_cast<h16x8*>(y + ((static_cast<int64_t>(b) * OH + oy) * OW + ox) * C + c) = m;
    }
}

// [B][C][HW] -> [B][HW][Cp] (channels zero-padded to Cp), 32x32 LDS tiles.
__global__ __launch_bounds__(256) void nchw_to_nhwc(const float* __restrict__ x, int C, int HW, int Cp,
                                                    float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* xb = x + static_cast<int64_t>(b) * C * HW;
    float* yb = y + static_cast<int64_t>(b) * HW * Cp;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, hw = hw0 + tx;
        tile[i][tx] = (c < C && hw < HW) ? xb[static_cast<int64_t>(c) * HW + hw] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int hw = hw0 + i, c = c0 + tx;
        if (hw < HW && c < Cp) yb[static_cast<int64_t>(hw) * Cp + c] = tile[tx][i];
    }
}

// Image boundary case: C <= 4 planes -> 4-channel pixels (C zero-padded). One thread per pixel: three coalesced plane
// reads, one 16-byte store (the generic tiled transpose wastes 7/8 of each 32x32 tile here).
__global__ __launch_bounds__(256) void nchw_to_nhwc4(const float* __restrict__ x, int C, int64_t HW, int64_t total,
                                                     float* __restrict__ y) {
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t b = e / HW, hw = e - b * HW;
        const float* p = x + b * C * HW + hw;
        float4 v;
        v.x = p[0];
        v.y = C > 1 ? p[HW] : 0.f;
        v.z = C > 2 ? p[2 * HW] : 0.f;
        v.w = C > 3 ? p[3 * HW] : 0.f;
        *reinterpret_cast<float4*>(y + e * 4) = v;
    }
}

// [B][HW][C] -> [B][C][HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw(const float* __restrict__ x, int C, int HW,
                                                    float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xb = x + static_cast<int64_t>(b) * HW * C;
    float* yb = y + static_cast<int64_t>(b) * C * HW;
    for (int i = ty; i < 32; i += 8) {
        const int hw = hw0 + i, c = c0 + tx;
        tile[i][tx] = (hw < HW && c < C) ? xb[static_cast<int64_t>(hw) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, hw = hw0 + tx;
        if (c < C && hw < HW) yb[static_cast<int64_t>(c) * HW + hw] = tile[tx][i];
    }
}

}  // namespace

extern "C" int mrcnn_maxpool_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t channels,
                                 int32_t kernel, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom,
                                 int32_t pad_right, float* y, int32_t y_layout, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "maxpool: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 1 && width >= 1 && channels >= 4 && channels % 4 == 0,
                  "maxpool: bad shape (channels %% 4 == 0 required)");
    MRCNN_REQUIRE(kernel >= 1 && stride >= 1 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 &&
                      pad_right >= 0, "maxpool: bad kernel/stride/pad");
    MRCNN_REQUIRE(y_layout == MRCNN_LAYOUT_NHWC || (y_layout == MRCNN_LAYOUT_KBLOCKED && channels % 8 == 0),
                  "maxpool: y_layout must be NHWC, or k-blocked with channels %% 8 == 0");
    const int OH = (height + pad_top + pad_bottom - kernel) / stride + 1;
    const int OW = (width + pad_left + pad_right - kernel) / stride + 1;
    MRCNN_REQUIRE(OH >= 1 && OW >= 1, "maxpool: empty output");
    const int64_t total = static_cast<int64_t>(batch) * OH * OW * (channels / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(maxpool_nhwc, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       mrcnn::as_stream(stream), x, batch, height, width, channels, kernel, stride,
                       pad_top, pad_left, OH, OW, y, y_layout == MRCNN_LAYOUT_KBLOCKED ? 1 : 0);
    return mrcnn::check_launch("maxpool_nhwc");
}

extern "C" int mrcnn_maxpool_nhwc_f16(const void* x, int32_t batch, int32_t height, int32_t width, int32_t channels,
                                      int32_t kernel, int32_t stride, int32_t pad_top, int32_t pad_left,
                                      int32_t pad_bottom, int32_t pad_right, void* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "maxpool_f16: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 1 && width >= 1 && channels >= 8 && channels % 8 == 0,
                  "maxpool_f16: bad shape (channels %% 8 == 0 required)");
    MRCNN_REQUIRE(kernel >= 1 && stride >= 1 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0,
                  "maxpool_f16: bad kernel/stride/pad");
    const int OH = (height + pad_top + pad_bottom - kernel) / stride + 1;
    const int OW = (width + pad_left + pad_right - kernel) / stride + 1;
    MRCNN_REQUIRE(OH >= 1 && OW >= 1, "maxpool_f16: empty output");
    const int64_t total = static_cast<int64_t>(batch) * OH * OW * (channels / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(maxpool_nhwc_f16, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, mrcnn::as_stream(stream),
                       static_cast<const _Float16*>(x), batch, height, width, channels, kernel, stride, pad_top, pad_left,
                       OH, OW, static_cast<_Float16*>(y));
    return mrcnn::check_launch("maxpool_nhwc_f16");
}

extern "C" int mrcnn_maxpool_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                      int32_t channels, int32_t kernel, int32_t stride, int32_t pad_top,
                                      int32_t pad_left, int32_t pad_bottom, int32_t pad_right, float* y,
                                      mrcnn_stream_t stream) {
    return mrcnn_maxpool_f32(x, batch, height, width, channels, kernel, stride, pad_top, pad_left, pad_bottom, pad_right,
                             y, MRCNN_LAYOUT_NHWC, stream);
}

extern "C" int mrcnn_nchw_to_nhwc_f32(const float* x, int32_t batch, int32_t channels, int32_t height,
                                      int32_t width, int32_t channels_padded, float* y,
                                      mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "nchw_to_nhwc: null pointer");
    MRCNN_REQUIRE(batch >= 1 && batch <= 65535 && channels >= 1 && height >= 1 && width >= 1 &&
                      channels_padded >= channels, "nchw_to_nhwc: bad shape");
    const int HW = height * width;
    if (channels <= 4 && channels_padded == 4) {
        const int64_t total = static_cast<int64_t>(batch) * HW;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(nchw_to_nhwc4, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                           mrcnn::as_stream(stream), x, channels, static_cast<int64_t>(HW), total, y);
        return mrcnn::check_launch("nchw_to_nhwc4");
    }
    dim3 grid((HW + 31) / 32, (channels_padded + 31) / 32, batch);
    MRCNN_REQUIRE(grid.y <= 65535, "nchw_to_nhwc: too many channels");
    hipLaunchKernelGGL(nchw_to_nhwc, grid, dim3(256), 0, mrcnn::as_stream(stream), x, channels, HW,
                       channels_padded, y);
    return mrcnn::check_launch("nchw_to_nhwc");
}

extern "C" int mrcnn_nhwc_to_nchw_f32(const float* x, int32_t batch, int32_t channels, int32_t height,
                                      int32_t width, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "nhwc_to_nchw: null pointer");
    MRCNN_REQUIRE(batch >= 1 && batch <= 65535 && channels >= 1 && height >= 1 && width >= 1,
                  "nhwc_to_nchw: bad shape");
    const int HW = height * width;
    dim3 grid((HW + 31) / 32, (channels + 31) / 32, batch);
    MRCNN_REQUIRE(grid.y <= 65535, "nhwc_to_nchw: too many channels");
    hipLaunchKernelGGL(nhwc_to_nchw, grid, dim3(256), 0, mrcnn::as_stream(stream), x, channels, HW, y);
    return mrcnn::check_launch("nhwc_to_nchw");
}

// ------------------------------------------------------------------------------------------------------
// RPN glue: the per-level permute/view/softmax/cat of rpn_detect (model.py:627-641,
// 1294-1304) and the gather / delta decode / clip of rpn_refine (model.py:1336-1358, data.py:124-148,86-92)
// as two launches instead of ~30 elementwise ones.
// ------------------------------------------------------------------------------------------------------
namespace {

struct RpnLevels {
    const float* y[5];   // mode 0: fused head output per level, NHWC [B][H][W][18]: 0-5 (bg,fg) logits x 3, 6-17 deltas
                         // mode 1: the Winograd kernel's head sums [2 k halves][rows][32] in position-major pixel order
                         //         (row = ((b*H/2 + y/2)*W/2 + x/2)*4 + (y&1)*2 + (x&1)); bias not yet added
    int hw[5];           // H*W per level
    int w[5];            // W per level
    int first[5];        // first anchor index of the level
    int mode[5];
    int rows[5];         // mode 1: rows per k half
    const float* bias;   // mode 1: [18] head bias (conv_class then conv_bbox)
};

// one thread per (image, anchor): fg = softmax(bg, fg)[1] as exp(x - max) / sum (the formula torch uses)
__global__ __launch_bounds__(256) void rpn_scores_deltas(RpnLevels lv, int B, int A,
                                                         float* __restrict__ scores,
                                                         float* __restrict__ deltas) {
    const int64_t total = static_cast<int64_t>(B) * A;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int b = static_cast<int>(e / A), a = static_cast<int>(e - static_cast<int64_t>(b) * A);
        int l = 0;
#pragma unroll
        for (int i = 1; i < 5; ++i) l += (a >= lv.first[i]) ? 1 : 0;
        const int local = a - lv.first[l];
        const int pix = local / 3, r = local - pix * 3;
        float l0, l1;
        float4 d;
        if (lv.mode[l] == 0) {
            const float* p = lv.y[l] + (static_cast<int64_t>(b) * lv.hw[l] + pix) * 18;
            l0 = p[2 * r];
            l1 = p[2 * r + 1];
            d = make_float4(p[6 + 4 * r], p[7 + 4 * r], p[8 + 4 * r], p[9 + 4 * r]);
        } else {
            const int W = lv.w[l], H = lv.hw[l] / W;
            const int y = pix / W, x = pix - y * W;
            int64_t row;
            if (lv.mode[l] == 1) {  // 64 consecutive positions per M tile: plain position-major order
                row = ((static_cast<int64_t>(b) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * 4 + (y & 1) * 2 + (x & 1);
            } else if (lv.mode[l] == 4) {  // plain pixel order (the fp16 pipelined conv's head sums), two planes
                row = static_cast<int64_t>(b) * lv.hw[l] + pix;
            } else if (lv.mode[l] == 2) {  // 8 x 8 position blocks: M-tile-major
                const int ty = y >> 1, tx = x >> 1, tyb = ((H >> 1) + 7) >> 3, txb = ((W >> 1) + 7) >> 3;
                const int64_t mt = (static_cast<int64_t>(b) * tyb + (ty >> 3)) * txb + (tx >> 3);
                row = mt * 256 + (((ty & 7) << 3) + (tx & 7)) * 4 + (y & 1) * 2 + (x & 1);
            } else {                       // F(4x4): 4 x 8 blocks of 4 x 4-pixel positions, M-tile-major, one part
                const int ty = y >> 2, tx = x >> 2, tyb = ((H >> 2) + 3) >> 2, txb = ((W >> 2) + 7) >> 3;
                const int64_t mt = (static_cast<int64_t>(b) * tyb + (ty >> 2)) * txb + (tx >> 3);
                row = mt * 512 + (((ty & 3) << 3) + (tx & 7)) * 16 + (y & 3) * 4 + (x & 3);
            }
            const float* p0 = lv.y[l] + row * 32;
            const float* bs = lv.bias;
            if (lv.mode[l] == 3) {
                l0 = p0[2 * r] + bs[2 * r];
                l1 = p0[2 * r + 1] + bs[2 * r + 1];
                d = make_float4(p0[6 + 4 * r] + bs[6 + 4 * r], p0[7 + 4 * r] + bs[7 + 4 * r], p0[8 + 4 * r] + bs[8 + 4 * r],
                                p0[9 + 4 * r] + bs[9 + 4 * r]);
            } else {
            const float* p1 = p0 + static_cast<int64_t>(lv.rows[l]) * 32;
            // fixed order: (k half 0 + k half 1) + bias
            l0 = (p0[2 * r] + p1[2 * r]) + bs[2 * r];
            l1 = (p0[2 * r + 1] + p1[2 * r + 1]) + bs[2 * r + 1];
            d = make_float4((p0[6 + 4 * r] + p1[6 + 4 * r]) + bs[6 + 4 * r], (p0[7 + 4 * r] + p1[7 + 4 * r]) + bs[7 + 4 * r],
                            (p0[8 + 4 * r] + p1[8 + 4 * r]) + bs[8 + 4 * r], (p0[9 + 4 * r] + p1[9 + 4 * r]) + bs[9 + 4 * r]);
            }
        }
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        scores[e] = e1 / (e0 + e1);
        *reinterpret_cast<float4*>(deltas + e * 4) = d;
    }
}

// one thread per (image, top-k slot): boxes_refine(anchor, delta*std) then clamp, data.py op order
__global__ __launch_bounds__(256) void proposal_decode(const float* __restrict__ anchors,
                                                       const float* __restrict__ deltas,
                                                       const int64_t* __restrict__ order,
                                                       const float* __restrict__ top_scores, int B, int A,
                                                       int K, float s0, float s1, float s2, float s3,
                                                       float img_h, float img_w, float* __restrict__ dets) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * K) return;
    const int b = e / K;
    const int64_t idx = order[e];
    const float4 an = *reinterpret_cast<const float4*>(anchors + idx * 4);
    const float4 dl = *reinterpret_cast<const float4*>(deltas + (static_cast<int64_t>(b) * A + idx) * 4);
    const float dy = dl.x * s0, dx = dl.y * s1, dh = dl.z * s2, dw = dl.w * s3;  // boxes_scale (:1341)
    float height = an.z - an.x;
    float width = an.w - an.y;
    float cy = an.x + 0.5f * height;
    float cx = an.y + 0.5f * width;
    cy = cy + dy * height;
    cx = cx + dx * width;
    height = height * expf(dh);
    width = width * expf(dw);
    float y1 = cy - 0.5f * height;
    float x1 = cx - 0.5f * width;
    float y2 = y1 + height;
    float x2 = x1 + width;
    y1 = fminf(fmaxf(y1, 0.f), img_h);
    x1 = fminf(fmaxf(x1, 0.f), img_w);
    y2 = fminf(fmaxf(y2, 0.f), img_h);
    x2 = fminf(fmaxf(x2, 0.f), img_w);
    float* o = dets + static_cast<int64_t>(e) * 5;
    o[0] = y1; o[1] = x1; o[2] = y2; o[3] = x2; o[4] = top_scores[e];
}

}  // namespace

extern "C" int mrcnn_rpn_scores_deltas_v2_f32(const float* const heads[5], const int32_t level_h[5],
                                              const int32_t level_w[5], const int32_t level_mode[5],
                                              const float* head_bias, int32_t batch, float* scores, float* deltas,
                                              mrcnn_stream_t stream) {
    MRCNN_REQUIRE(heads && level_h && level_w && level_mode && scores && deltas && batch >= 1,
                  "rpn_scores_deltas: bad arguments");
    RpnLevels lv;
    int a = 0;
    lv.bias = head_bias;
    for (int l = 0; l < 5; ++l) {
        MRCNN_REQUIRE(heads[l] && level_h[l] >= 1 && level_w[l] >= 1, "rpn_scores_deltas: bad level %d", l);
        MRCNN_REQUIRE(level_mode[l] == 0 ||
                          ((level_mode[l] == 1 || level_mode[l] == 2) && head_bias && level_h[l] % 2 == 0 && level_w[l] % 2 == 0) ||
                          (level_mode[l] == 3 && head_bias && level_h[l] % 4 == 0 && level_w[l] % 4 == 0) ||
                          (level_mode[l] == 4 && head_bias),
                      "rpn_scores_deltas: level %d: mode must be 0 (NHWC heads), 1 or 2 (head sums of "
                      "mrcnn_conv3x3_winograd_heads_f32 in tile mode 1 / 2: even H, W and a bias vector) or 3 (head sums "
                      "of mrcnn_conv3x3_winograd4_heads_f32: H, W multiples of 4 and a bias vector) or 4 (the two planes of "
                      "mrcnn_conv_f16_pipelined_heads for 512 channels, pixel order; a bias vector)", l);
        lv.y[l] = heads[l];
        lv.hw[l] = level_h[l] * level_w[l];
        lv.w[l] = level_w[l];
        lv.mode[l] = level_mode[l];
        lv.rows[l] = 0;
        if (level_mode[l] == 1) {
            const int64_t t = static_cast<int64_t>(batch) * (level_h[l] / 2) * (level_w[l] / 2);
            lv.rows[l] = static_cast<int>(((t + 63) / 64) * 256);  // == mrcnn_conv3x3_winograd_heads_rows(.., 1)
        } else if (level_mode[l] == 2) {
            lv.rows[l] = batch * ((level_h[l] / 2 + 7) / 8) * ((level_w[l] / 2 + 7) / 8) * 256;
        } else if (level_mode[l] == 4) {
            lv.rows[l] = batch * level_h[l] * level_w[l];
        }
        lv.first[l] = a;
        a += lv.hw[l] * 3;
    }
    const int64_t total = static_cast<int64_t>(batch) * a;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(rpn_scores_deltas, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       mrcnn::as_stream(stream), lv, batch, a, scores, deltas);
    return mrcnn::check_launch("rpn_scores_deltas");
}

extern "C" int mrcnn_rpn_scores_deltas_f32(const float* const heads[5], const int32_t level_hw[5],
                                           int32_t batch, float* scores, float* deltas,
                                           mrcnn_stream_t stream) {
    MRCNN_REQUIRE(level_hw, "rpn_scores_deltas: bad arguments");
    const int32_t one[5] = {1, 1, 1, 1, 1}, zero[5] = {0, 0, 0, 0, 0};
    return mrcnn_rpn_scores_deltas_v2_f32(heads, level_hw, one, zero, nullptr, batch, scores, deltas, stream);
}

extern "C" int mrcnn_proposal_decode_f32(const float* anchors, const float* deltas, const int64_t* order,
                                         const float* top_scores, int32_t batch, int32_t num_anchors,
                                         int32_t k, const float std_dev[4], float image_height,
                                         float image_width, float* dets, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(anchors && deltas && order && top_scores && std_dev && dets, "proposal_decode: null pointer");
    MRCNN_REQUIRE(batch >= 1 && num_anchors >= 1 && k >= 1, "proposal_decode: bad sizes");
    const int total = batch * k;
    hipLaunchKernelGGL(proposal_decode, dim3((total + 255) / 256), dim3(256), 0, mrcnn::as_stream(stream),
                       anchors, deltas, order, top_scores, batch, num_anchors, k, std_dev[0], std_dev[1],
                       std_dev[2], std_dev[3], image_height, image_width, dets);
    return mrcnn::check_launch("proposal_decode");
}

// ------------------------------------------------------------------------------------------------------
// Detection decode: the first half of MaskRCNN.mrn_refine (model.py:1405-1443) for a whole batch in one
// launch — softmax + argmax over classes, class-specific delta gather, boxes_refine(rois, delta*std)
// (data.py:124-148), scale to pixels, clip to the image window, round, and the validity rule
// (class > 0, inside the image's live RoI slots, optional score threshold).
// One wavefront per RoI: lanes stride over the classes, wave reductions for max / sum / argmax.
// ------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void detection_decode(
    const float* __restrict__ logits, int64_t logit_stride, const float* __restrict__ bbox,
    int64_t bbox_stride, const float* __restrict__ rois, const int* __restrict__ roi_counts,
    const float* __restrict__ windows, int B, int P, int C, float s0, float s1, float s2, float s3,
    float img_h, float img_w, float min_conf, float* __restrict__ dets, int* __restrict__ nms_cls,
    int64_t* __restrict__ class_ids) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);  // RoI index in [0, B*P)
    if (r >= B * P) return;
    const int b = r / P, slot = r - b * P;
    if (slot >= roi_counts[b]) {
        // no RoI in this slot (the reference's tensors simply end before it, model.py:1366-1374). Its logits / bbox rows may
        // never have been written (the head skips row tiles of empty slots): nothing of them is read; a fixed, excluded record
        if (lane == 0) {
            float* o = dets + static_cast<int64_t>(r) * 5;
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f;
            class_ids[r] = 0;
            nms_cls[r] = -(slot + 1);
        }
        return;
    }
    const float* lg = logits + static_cast<int64_t>(r) * logit_stride;
    // max and first argmax over classes
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = lg[c];
        if (v > best) { best = v; besti = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(lg[c] - best);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane != 0) return;
    const float score = 1.0f / sum;  // exp(0) / sum: the probability of the arg-max class
    const float* dl = bbox + static_cast<int64_t>(r) * bbox_stride + besti * 4;
    const float dy = dl[0] * s0, dx = dl[1] * s1, dh = dl[2] * s2, dw = dl[3] * s3;
    const float* ro = rois + static_cast<int64_t>(r) * 4;
    float height = ro[2] - ro[0];
    float width = ro[3] - ro[1];
    float cy = ro[0] + 0.5f * height;
    float cx = ro[1] + 0.5f * width;
    cy = cy + dy * height;
    cx = cx + dx * width;
    height = height * expf(dh);
    width = width * expf(dw);
    float y1 = cy - 0.5f * height;
    float x1 = cx - 0.5f * width;
    float y2 = y1 + height;
    float x2 = x1 + width;
    y1 = y1 * img_h; x1 = x1 * img_w; y2 = y2 * img_h; x2 = x2 * img_w;  // boxes_scale (:1426)
    const float* w = windows + b * 4;
    y1 = rintf(fminf(fmaxf(y1, w[0]), w[2]));
    x1 = rintf(fminf(fmaxf(x1, w[1]), w[3]));
    y2 = rintf(fminf(fmaxf(y2, w[0]), w[2]));
    x2 = rintf(fminf(fmaxf(x2, w[1]), w[3]));
    float* o = dets + static_cast<int64_t>(r) * 5;
    o[0] = y1; o[1] = x1; o[2] = y2; o[3] = x2; o[4] = score;
    bool valid = besti > 0 && slot < roi_counts[b];
    if (min_conf > 0.f) valid = valid && score >= min_conf;
    class_ids[r] = besti;
    nms_cls[r] = valid ? besti : -(slot + 1);  // excluded slots: unique negative class, never interact in NMS
}

}  // namespace

extern "C" int mrcnn_detection_decode_f32(const float* logits, int64_t logit_stride, const float* bbox,
                                          int64_t bbox_stride, const float* rois, const int32_t* roi_counts,
                                          const float* windows, int32_t batch, int32_t rois_per_image,
                                          int32_t num_classes, const float std_dev[4], float image_height,
                                          float image_width, float min_confidence, float* dets,
                                          int32_t* nms_class_ids, int64_t* class_ids, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(logits && bbox && rois && roi_counts && windows && std_dev && dets && nms_class_ids && class_ids,
                  "detection_decode: null pointer");
    MRCNN_REQUIRE(batch >= 1 && rois_per_image >= 1 && num_classes >= 1, "detection_decode: bad sizes");
    const int total = batch * rois_per_image;
    hipLaunchKernelGGL(detection_decode, dim3((total + 3) / 4), dim3(256), 0, mrcnn::as_stream(stream), logits,
                       logit_stride, bbox, bbox_stride, rois, roi_counts, windows, batch, rois_per_image,
                       num_classes, std_dev[0], std_dev[1], std_dev[2], std_dev[3], image_height, image_width,
                       min_confidence, dets, nms_class_ids, class_ids);
    return mrcnn::check_launch("detection_decode");
}
