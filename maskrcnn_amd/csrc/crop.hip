// crop_and_resize ("RoIAlign" of this Mask R-CNN) for gfx950.
//   forward : restates /root/reference/c++ext/maskrcnn/csrc/cpu/crop_cpu.cpp:13-116 arithmetic op for op
//             (scale :52-55, sample coordinate :59-61/:82-84, strict outside test :63/:85,
//              floorf/ceilf taps :76-77/:94-95, a+(b-a)*t lerps :107-110), FP contraction off.
//   backward: crop_cpu.cpp:167-265 with fp32 atomics.
//   pyramid : model.py:276-393 (level assignment + per-level crop + order restore) in ONE launch on
//             channels-last feature maps.
// Not a port of crop_cuda.cu (one thread per output scalar, NCHW-innermost-x, 4 scattered loads each):
//   * sample coordinates are computed once per (box,row) / (box,column) into LDS, not per output;
//   * NCHW entry point: a workgroup owns (box, channel slab); lanes run along x then y of the crop so
//     stores are fully coalesced and the 4 taps of neighbouring lanes share cache lines;
//   * NHWC pyramid kernel: a wavefront owns one sample point and reads each tap as 64 lanes x 16 B =
//     1 KiB of consecutive channels (perfectly coalesced HBM gathers), lerps in registers, and writes
//     1 KiB of consecutive channels of the NHWC output.
#pragma clang fp contract(off)

#include "common.hpp"

namespace {

struct Sample {   // one crop row or column
    int lo, hi;   // tap indices (floorf / ceilf)
    float lerp;   // in - lo
    int inside;   // 0 → extrapolation_value
};

// crop_cpu.cpp:52-61 (y) and :82-84 (x): identical formulas with (c1,c2,size,crop) swapped in.
__device__ __forceinline__ Sample make_sample(float c1, float c2, int size, int crop, int t) {
    float in;
    if (crop > 1) {
        float s = c2 - c1;
        s = s * static_cast<float>(size - 1);
        const float scale = s / static_cast<float>(crop - 1);
        const float a = c1 * static_cast<float>(size - 1);
        const float b = static_cast<float>(t) * scale;
        in = a + b;
    } else {  // 0.5 * (c1 + c2) * (size - 1): the literal is double in the reference
        const float sum = c1 + c2;
        in = static_cast<float>(0.5 * static_cast<double>(sum) * static_cast<double>(size - 1));
    }
    Sample r;
    r.inside = !(in < 0.0f || in > static_cast<float>(size - 1));
    r.lo = r.inside ? static_cast<int>(floorf(in)) : 0;
    r.hi = r.inside ? static_cast<int>(ceilf(in)) : 0;
    r.lerp = in - static_cast<float>(r.lo);
    return r;
}

__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float xl, float yl) {
    float t = tr - tl;
    t = t * xl;
    const float top = tl + t;
    float u = br - bl;
    u = u * xl;
    const float bot = bl + u;
    float v = bot - top;
    v = v * yl;
    return top + v;
}

// ------------------------------------------------------------------------------------------------
// NCHW forward (the drop-in signature). grid = (num_boxes, channel slabs), block = 256.
// ------------------------------------------------------------------------------------------------
struct Tap {  // one crop position (y, x): the 4 tap offsets inside a channel plane and the lerp weights
    int tl, tr, bl, br;
    float xl, yl;
    int inside, pad;
};

__global__ __launch_bounds__(256) void crop_forward_nchw(
    const float* __restrict__ image, int batch, int depth, int H, int W,
    const float* __restrict__ boxes, const int* __restrict__ box_index, float extrap, int ch, int cw,
    int slab, float* __restrict__ crops) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + ch;
    Tap* taps = reinterpret_cast<Tap*>(sx + cw);  // [ch*cw] when it fits (use_table), else unused
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * slab;
    const int c1 = min(depth, c0 + slab);
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
    const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    const int b_in = box_index[b];
    const bool bad = (b_in < 0 || b_in >= batch);
    for (int t = threadIdx.x; t < ch + cw; t += blockDim.x) {
        if (t < ch) sy[t] = make_sample(y1, y2, H, ch, t);
        else sx[t - ch] = make_sample(x1, x2, W, cw, t - ch);
    }
    __syncthreads();
    const int plane = ch * cw;
    const bool use_table = plane <= 1024;  // host sizes the LDS accordingly
    if (use_table) {  // per-position record: no index arithmetic left in the channel loop
        for (int r = threadIdx.x; r < plane; r += blockDim.x) {
            const Sample Y = sy[r / cw], X = sx[r % cw];
            Tap t;
            t.tl = Y.lo * W + X.lo; t.tr = Y.lo * W + X.hi;
            t.bl = Y.hi * W + X.lo; t.br = Y.hi * W + X.hi;
            t.xl = X.lerp; t.yl = Y.lerp;
            t.inside = (!bad && Y.inside && X.inside) ? 1 : 0;
            t.pad = 0;
            taps[r] = t;
        }
        __syncthreads();
    }
    const int64_t HW = static_cast<int64_t>(H) * W;
    const float* img = image + (bad ? 0 : static_cast<int64_t>(b_in) * depth * HW);
    float* out = crops + static_cast<int64_t>(b) * depth * plane;
    const int total = (c1 - c0) * plane;
    if (use_table) {
        // e = c*plane + r advances by blockDim.x: carry (c, r) incrementally instead of dividing
        int c = c0 + threadIdx.x / plane, r = threadIdx.x % plane;
        const int step_c = blockDim.x / plane, step_r = blockDim.x % plane;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const Tap t = taps[r];
            float v = extrap;
            if (t.inside) {
                const float* p = img + c * HW;
                v = bilerp(p[t.tl], p[t.tr], p[t.bl], p[t.br], t.xl, t.yl);
            }
            out[static_cast<int64_t>(c) * plane + r] = v;
            c += step_c;
            r += step_r;
            if (r >= plane) { r -= plane; ++c; }
        }
        return;
    }
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int c = c0 + e / plane;
        const int r = e % plane;
        const int y = r / cw, x = r % cw;
        const Sample Y = sy[y], X = sx[x];
        float v = extrap;
        if (!bad && Y.inside && X.inside) {
            const float* p = img + c * HW;
            const float tl = p[Y.lo * W + X.lo], tr = p[Y.lo * W + X.hi];
            const float bl = p[Y.hi * W + X.lo], br = p[Y.hi * W + X.hi];
            v = bilerp(tl, tr, bl, br, X.lerp, Y.lerp);
        }
        out[static_cast<int64_t>(c) * plane + r] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// NCHW backward: one thread per grad element, 4 atomics (crop_cpu.cpp:244-260).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_backward_nchw(
    const float* __restrict__ grads, const float* __restrict__ boxes,
    const int* __restrict__ box_index, int batch, int depth, int H, int W, int ch, int cw, int slab,
    float* __restrict__ gimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + ch;
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * slab;
    const int c1 = min(depth, c0 + slab);
    const int b_in = box_index[b];
    if (b_in < 0 || b_in >= batch) return;  // uniform per block
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
    const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    for (int t = threadIdx.x; t < ch + cw; t += blockDim.x) {
        if (t < ch) sy[t] = make_sample(y1, y2, H, ch, t);
        else sx[t - ch] = make_sample(x1, x2, W, cw, t - ch);
    }
    __syncthreads();
    const int plane = ch * cw;
    const int64_t HW = static_cast<int64_t>(H) * W;
    float* img = gimg + static_cast<int64_t>(b_in) * depth * HW;
    const float* g = grads + static_cast<int64_t>(b) * depth * plane;
    const int total = (c1 - c0) * plane;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int c = c0 + e / plane;
        const int r = e % plane;
        const Sample Y = sy[r / cw], X = sx[r % cw];
        if (!(Y.inside && X.inside)) continue;
        const float gv = g[static_cast<int64_t>(c) * plane + r];
        float* p = img + c * HW;
        const float omy = 1.0f - Y.lerp, omx = 1.0f - X.lerp;
        const float dtop = omy * gv, dbot = Y.lerp * gv;
        atomicAdd(p + Y.lo * W + X.lo, omx * dtop);
        atomicAdd(p + Y.lo * W + X.hi, X.lerp * dtop);
        atomicAdd(p + Y.hi * W + X.lo, omx * dbot);
        atomicAdd(p + Y.hi * W + X.hi, X.lerp * dbot);
    }
}

// ------------------------------------------------------------------------------------------------
// NHWC pyramid forward: grid = num_rois, block = 256 (4 waves). Wave w takes sample points
// w, w+4, ...; a lane owns 4 consecutive channels (float4), looping over depth in 256-channel steps.
// ------------------------------------------------------------------------------------------------
struct PyramidArgs {
    const float* fm[4];
    int h[4], w[4];
};

__device__ __forceinline__ int roi_level(float y1, float x1, float y2, float x2, float image_area) {
    // model.py:324-338 in fp32: 4 + log2(sqrt(h*w) / (224 / sqrt(area))), round half to even, clamp.
    // Every fp32 operation of that formula is evaluated CORRECTLY ROUNDED: sqrt, the two divisions and log2 go through
    // double and are rounded to float once (exact for sqrt and division, 53 >= 2*24 + 2; for log2 up to double's own
    // error, ~1e-8 of the inputs). torch-CPU's log2 (MKL VML, high-accuracy mode) is correctly rounded on all but
    // ~1e-4 of its inputs, the device's log2f on far fewer: with log2f 7-11 % of the boxes within a few ulp of a level
    // boundary k = 2.5 / 3.5 / 4.5 landed one level off (tests/test_gpu_fullsize.py::test_level_boundaries_ulp_sweep).
    const float h = y2 - y1, w = x2 - x1;
    const float hw = h * w;
    const float denom = static_cast<float>(224.0 / static_cast<double>(static_cast<float>(sqrt(static_cast<double>(image_area)))));
    const float ratio = static_cast<float>(static_cast<double>(static_cast<float>(sqrt(static_cast<double>(hw)))) /
                                           static_cast<double>(denom));
    const float k = 4.0f + static_cast<float>(log2(static_cast<double>(ratio)));
    // NaN (negative area) / -inf (zero area) → the reference's int cast is UB; it lands on level 2
    if (!(k >= 2.0f)) return 2;
    if (k >= 5.0f) return 5;
    return static_cast<int>(rintf(k));
}

__global__ __launch_bounds__(256) void roi_align_pyramid_nhwc(
    PyramidArgs a, int batch, int depth, const float* __restrict__ rois,
    const int* __restrict__ roi_batch, int rois_per_image, int pool, float image_area,
    float* __restrict__ out, int* __restrict__ levels_out, int out_kblocked, int64_t out_pixels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + pool;
    const int r = blockIdx.x;
    const float y1 = rois[r * 4 + 0], x1 = rois[r * 4 + 1];
    const float y2 = rois[r * 4 + 2], x2 = rois[r * 4 + 3];
    const int level = roi_level(y1, x1, y2, x2, image_area);
    const int li = level - 2;
    const int H = a.h[li], W = a.w[li];
    int b_in = roi_batch ? roi_batch[r] : r / rois_per_image;
    const bool bad = (b_in < 0 || b_in >= batch);
    if (threadIdx.x == 0 && levels_out) levels_out[r] = level;
    for (int t = threadIdx.x; t < 2 * pool; t += blockDim.x) {
        if (t < pool) sy[t] = make_sample(y1, y2, H, pool, t);
        else sx[t - pool] = make_sample(x1, x2, W, pool, t - pool);
    }
    __syncthreads();
    const float* img = a.fm[li] + (bad ? 0 : static_cast<int64_t>(b_in) * H * W * depth);
    float* o = out + static_cast<int64_t>(r) * pool * pool * depth;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int points = pool * pool;
    for (int pt = wave; pt < points; pt += 4) {
        const Sample Y = sy[pt / pool], X = sx[pt % pool];
        const bool inside = !bad && Y.inside && X.inside;
        const int64_t otl = (static_cast<int64_t>(Y.lo) * W + X.lo) * depth;
        const int64_t otr = (static_cast<int64_t>(Y.lo) * W + X.hi) * depth;
        const int64_t obl = (static_cast<int64_t>(Y.hi) * W + X.lo) * depth;
        const int64_t obr = (static_cast<int64_t>(Y.hi) * W + X.hi) * depth;
        for (int c = lane * 4; c < depth; c += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // pyramid call sites use extrapolation 0
            if (inside) {
                const float4 tl = *reinterpret_cast<const float4*>(img + otl + c);
                const float4 tr = *reinterpret_cast<const float4*>(img + otr + c);
                const float4 bl = *reinterpret_cast<const float4*>(img + obl + c);
                const float4 br = *reinterpret_cast<const float4*>(img + obr + c);
                v.x = bilerp(tl.x, tr.x, bl.x, br.x, X.lerp, Y.lerp);
                v.y = bilerp(tl.y, tr.y, bl.y, br.y, X.lerp, Y.lerp);
                v.z = bilerp(tl.z, tr.z, bl.z, br.z, X.lerp, Y.lerp);
                v.w = bilerp(tl.w, tr.w, bl.w, br.w, X.lerp, Y.lerp);
            }
            if (out_kblocked)  // [depth/8][num_rois*pool*pool][8]: what the Winograd kernel (mask head conv1) reads
                *reinterpret_cast<float4*>(out + ((c >> 3) * out_pixels + static_cast<int64_t>(r) * points + pt) * 8 + (c & 7)) = v;
            else
                *reinterpret_cast<float4*>(o + static_cast<int64_t>(pt) * depth + c) = v;
        }
    }
}

}  // namespace

extern "C" int mrcnn_crop_forward_f32(const float* image, int32_t batch, int32_t depth,
                                      int32_t height, int32_t width, const float* boxes,
                                      const int32_t* box_index, int32_t num_boxes,
                                      float extrapolation_value, int32_t crop_height,
                                      int32_t crop_width, float* crops, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(batch >= 1 && depth >= 1 && height >= 1 && width >= 1,
                  "crop_forward: bad image shape [%d,%d,%d,%d]", batch, depth, height, width);
    MRCNN_REQUIRE(crop_height >= 1 && crop_width >= 1 && crop_height + crop_width <= 4096,
                  "crop_forward: bad crop size %dx%d", crop_height, crop_width);
    MRCNN_REQUIRE(num_boxes >= 0, "crop_forward: num_boxes=%d", num_boxes);
    if (num_boxes == 0) return MRCNN_OK;
    MRCNN_REQUIRE(image && boxes && box_index && crops, "crop_forward: null pointer");
    const int plane = crop_height * crop_width;
    // a channel slab gives each workgroup ~2k outputs (measured best of 1k/2k/4k/8k: thousands of workgroups keep every CU gathering); grid.y <= 65535
    int slab = (2048 + plane - 1) / plane;
    if (slab < 1) slab = 1;
    if (slab > depth) slab = depth;
    int gy = (depth + slab - 1) / slab;
    if (gy > 65535) { gy = 65535; slab = (depth + gy - 1) / gy; gy = (depth + slab - 1) / slab; }
    const size_t lds = sizeof(Sample) * (crop_height + crop_width) + (plane <= 1024 ? sizeof(Tap) * plane : 0);
    hipLaunchKernelGGL(crop_forward_nchw, dim3(num_boxes, gy), dim3(256), lds,
                       mrcnn::as_stream(stream), image, batch, depth, height, width, boxes,
                       box_index, extrapolation_value, crop_height, crop_width, slab, crops);
    return mrcnn::check_launch("crop_forward_nchw");
}

extern "C" int mrcnn_crop_backward_f32(const float* grads, const float* boxes,
                                       const int32_t* box_index, int32_t num_boxes, int32_t batch,
                                       int32_t depth, int32_t height, int32_t width,
                                       int32_t crop_height, int32_t crop_width, float* grads_image,
                                       mrcnn_stream_t stream) {
    MRCNN_REQUIRE(batch >= 1 && depth >= 1 && height >= 1 && width >= 1,
                  "crop_backward: bad image shape [%d,%d,%d,%d]", batch, depth, height, width);
    MRCNN_REQUIRE(crop_height >= 1 && crop_width >= 1 && crop_height + crop_width <= 4096,
                  "crop_backward: bad crop size %dx%d", crop_height, crop_width);
    MRCNN_REQUIRE(grads_image, "crop_backward: null grads_image");
    hipStream_t s = mrcnn::as_stream(stream);
    hipError_t e = hipMemsetAsync(grads_image, 0,
                                  sizeof(float) * static_cast<size_t>(batch) * depth * height * width, s);
    if (e != hipSuccess)
        return mrcnn::fail(MRCNN_ERR_LAUNCH, "crop_backward: memset: %s", hipGetErrorString(e));
    if (num_boxes <= 0) return MRCNN_OK;
    MRCNN_REQUIRE(grads && boxes && box_index, "crop_backward: null pointer");
    const int plane = crop_height * crop_width;
    int slab = (8192 + plane - 1) / plane;
    if (slab > depth) slab = depth;
    int gy = (depth + slab - 1) / slab;
    if (gy > 65535) { gy = 65535; slab = (depth + gy - 1) / gy; gy = (depth + slab - 1) / slab; }
    const size_t lds = sizeof(Sample) * (crop_height + crop_width);
    hipLaunchKernelGGL(crop_backward_nchw, dim3(num_boxes, gy), dim3(256), lds, s, grads, boxes,
                       box_index, batch, depth, height, width, crop_height, crop_width, slab,
                       grads_image);
    return mrcnn::check_launch("crop_backward_nchw");
}

extern "C" int mrcnn_roi_align_pyramid_f32(const float* const fm[4], const int32_t fm_h[4], const int32_t fm_w[4],
                                           int32_t batch, int32_t depth, const float* rois, const int32_t* roi_batch,
                                           int32_t num_rois, int32_t rois_per_image, int32_t pool, float image_area,
                                           float* out, int32_t out_layout, int32_t* levels_out, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(fm && fm_h && fm_w && rois && out, "roi_align_pyramid: null pointer");
    MRCNN_REQUIRE(batch >= 1 && depth >= 4 && depth % 4 == 0, "roi_align_pyramid: depth=%d must be a multiple of 4", depth);
    MRCNN_REQUIRE(pool >= 1 && pool <= 1024, "roi_align_pyramid: pool=%d", pool);
    MRCNN_REQUIRE(roi_batch || rois_per_image >= 1, "roi_align_pyramid: need roi_batch or rois_per_image");
    MRCNN_REQUIRE(out_layout == MRCNN_LAYOUT_NHWC || (out_layout == MRCNN_LAYOUT_KBLOCKED && depth % 8 == 0),
                  "roi_align_pyramid: out_layout must be NHWC, or k-blocked with depth %% 8 == 0");
    if (num_rois <= 0) return MRCNN_OK;
    PyramidArgs a;
    for (int l = 0; l < 4; ++l) {
        MRCNN_REQUIRE(fm[l] && fm_h[l] >= 1 && fm_w[l] >= 1, "roi_align_pyramid: bad level %d", l);
        a.fm[l] = fm[l];
        a.h[l] = fm_h[l];
        a.w[l] = fm_w[l];
    }
    const size_t lds = sizeof(Sample) * 2 * pool;
    hipLaunchKernelGGL(roi_align_pyramid_nhwc, dim3(num_rois), dim3(256), lds,
                       mrcnn::as_stream(stream), a, batch, depth, rois, roi_batch, rois_per_image,
                       pool, image_area, out, levels_out, out_layout == MRCNN_LAYOUT_KBLOCKED ? 1 : 0,
                       static_cast<int64_t>(num_rois) * pool * pool);
    return mrcnn::check_launch("roi_align_pyramid_nhwc");
}

extern "C" int mrcnn_roi_align_pyramid_nhwc_f32(const float* const fm[4], const int32_t fm_h[4],
                                                const int32_t fm_w[4], int32_t batch, int32_t depth,
                                                const float* rois, const int32_t* roi_batch,
                                                int32_t num_rois, int32_t rois_per_image,
                                                int32_t pool, float image_area, float* out,
                                                int32_t* levels_out, mrcnn_stream_t stream) {
    return mrcnn_roi_align_pyramid_f32(fm, fm_h, fm_w, batch, depth, rois, roi_batch, num_rois, rois_per_image, pool,
                                       image_area, out, MRCNN_LAYOUT_NHWC, levels_out, stream);
}
