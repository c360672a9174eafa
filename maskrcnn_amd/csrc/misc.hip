// Small HBM-bound helpers around the conv stack (gfx950): zero-padded max-pool on channels-last data
// (stem: SamePad2d(3,2) + MaxPool2d(3,2), model.py:227-228; P6 = MaxPool2d(1,2), model.py:109,161) and the
// NCHW <-> NHWC conversions at the reference boundary (model.py:1109 hands over NCHW).
#pragma clang fp contract(off)

#include "common.hpp"

namespace {

// y[b,oy,ox,c] = max over k x k window of zero-padded x. One thread = one output pixel x 4 channels.
__global__ __launch_bounds__(256) void maxpool_nhwc(const float* __restrict__ x, int B, int H, int W,
                                                    int C, int k, int stride, int pt, int pl, int OH,
                                                    int OW, float* __restrict__ y) {
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
        *reinterpret_cast<float4*>(y + ((static_cast<int64_t>(b) * OH + oy) * OW + ox) * C + c) = m;
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

extern "C" int mrcnn_maxpool_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                      int32_t channels, int32_t kernel, int32_t stride, int32_t pad_top,
                                      int32_t pad_left, int32_t pad_bottom, int32_t pad_right, float* y,
                                      mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "maxpool: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 1 && width >= 1 && channels >= 4 && channels % 4 == 0,
                  "maxpool: bad shape (channels %% 4 == 0 required)");
    MRCNN_REQUIRE(kernel >= 1 && stride >= 1 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 &&
                      pad_right >= 0, "maxpool: bad kernel/stride/pad");
    const int OH = (height + pad_top + pad_bottom - kernel) / stride + 1;
    const int OW = (width + pad_left + pad_right - kernel) / stride + 1;
    MRCNN_REQUIRE(OH >= 1 && OW >= 1, "maxpool: empty output");
    const int64_t total = static_cast<int64_t>(batch) * OH * OW * (channels / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(maxpool_nhwc, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       mrcnn::as_stream(stream), x, batch, height, width, channels, kernel, stride,
                       pad_top, pad_left, OH, OW, y);
    return mrcnn::check_launch("maxpool_nhwc");
}

extern "C" int mrcnn_nchw_to_nhwc_f32(const float* x, int32_t batch, int32_t channels, int32_t height,
                                      int32_t width, int32_t channels_padded, float* y,
                                      mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "nchw_to_nhwc: null pointer");
    MRCNN_REQUIRE(batch >= 1 && batch <= 65535 && channels >= 1 && height >= 1 && width >= 1 &&
                      channels_padded >= channels, "nchw_to_nhwc: bad shape");
    const int HW = height * width;
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
