// Image pre-/post-processing either side of the hot path, on the GPU:
//   resize_image + mold_image  (utils.py:42-90, model.py:1750-1754, :1102-1110)  uint8 HWC -> resized, centre-padded,
//                              mean-subtracted fp32 CHW
//   full_masks                 (data.py:287-314)  28x28 sigmoid masks -> full-size binary masks pasted at their boxes
// Both reduce to Pillow's 8-bit BILINEAR resample (scipy.misc.imresize / torchvision Resize are thin wrappers over
// Image.resize): double-precision triangle-filter coefficients with the support stretched by the scale factor when
// shrinking, normalised, rounded to 22-bit fixed point; a horizontal pass into an 8-bit intermediate, then a vertical
// pass, each (2^21 + sum in*k) >> 22 clamped to [0,255]. The coefficient arithmetic below follows that definition
// operation by operation in fp64 (this file is compiled with -ffp-contract=off), so results are bit-identical to
// Pillow's, which is what the tests check. All of it is byte/integer work bound by HBM traffic (and tiny).
#include "common.hpp"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Axis {  // one resample direction
    int in_size, out_size, ksize;
    double scale, support, ss;  // ss = 1 / filterscale
};

Axis make_axis(int in_size, int out_size) {
    Axis a;
    a.in_size = in_size;
    a.out_size = out_size;
    double filterscale = a.scale = static_cast<double>(static_cast<float>(in_size)) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    a.support = 1.0 * filterscale;  // BILINEAR support = 1.0
    a.ksize = static_cast<int>(ceil(a.support)) * 2 + 1;
    a.ss = 1.0 / filterscale;
    return a;
}

__device__ __forceinline__ double tri(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Coefficients of output sample xx: first input tap, tap count, and k[0..ksize) (stride kstride ints).
__device__ __forceinline__ void axis_coeffs(const Axis& a, int xx, int& xmin, int& cnt, int* k, int kstride) {
    const double center = (xx + 0.5) * a.scale;
    xmin = static_cast<int>(center - a.support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = static_cast<int>(center + a.support + 0.5);
    if (xmax > a.in_size) xmax = a.in_size;
    cnt = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < cnt; ++x) ww += tri((x + xmin - center + 0.5) * a.ss);
    for (int x = 0; x < cnt; ++x) {
        double w = tri((x + xmin - center + 0.5) * a.ss);
        if (ww != 0.0) w /= ww;
        k[x * kstride] = static_cast<int>(0.5 + w * static_cast<double>(1 << PRECISION_BITS));
    }
    for (int x = cnt; x < a.ksize; ++x) k[x * kstride] = 0;
}

__device__ __forceinline__ unsigned clip8(int v) {
    v >>= PRECISION_BITS;
    return static_cast<unsigned>(v < 0 ? 0 : v > 255 ? 255 : v);
}
// clip8 for values that are PACKED into a word (b0 | b1 << 8 | ...): the compiler fuses two shift + clamp + pack steps into
// gfx950's v_ashr_pk_u8_i32, whose result it then ORs the next bytes onto as if bits 16-31 were zero — on the MI355X they are
// not (they keep what the destination register held: measured, round 5: bytes 2 and 3 of every packed word came out OR-ed with
// stale bits). Making each clamped value opaque before it is shifted into place keeps the plain shift / clamp / or sequence.
__device__ __forceinline__ unsigned clip8_for_packing(int v) {
    unsigned c = clip8(v);
    asm volatile("" : "+v"(c));
    return c;
}

__global__ __launch_bounds__(256) void coeffs_kernel(const Axis a, int* __restrict__ bounds, int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= a.out_size) return;
    int xmin, cnt;
    axis_coeffs(a, xx, xmin, cnt, kk + static_cast<int64_t>(xx) * a.ksize, 1);
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = cnt;
}

// src n x [in_h][in_w][c] (image / row strides in bytes) -> tmp [n*in_h][out_w][c]
__global__ __launch_bounds__(256) void resample_h_u8(const uint8_t* __restrict__ src, int n, int in_h, int c,
                                                     int64_t image_stride, int64_t row_stride, int out_w,
                                                     const int* __restrict__ bounds, const int* __restrict__ kk,
                                                     int ksize, uint8_t* __restrict__ tmp) {
    const int64_t row = static_cast<int64_t>(out_w) * c, total = row * in_h * n;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int yg = static_cast<int>(e / row), r = static_cast<int>(e - yg * row);
        const int img = yg / in_h, y = yg - img * in_h;
        const int xx = r / c, ch = r - xx * c;
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int* k = kk + static_cast<int64_t>(xx) * ksize;
        const uint8_t* s = src + img * image_stride + y * row_stride + static_cast<int64_t>(xmin) * c + ch;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) ss += s[static_cast<int64_t>(x) * c] * k[x];
        tmp[e] = static_cast<uint8_t>(clip8(ss));
    }
}

__device__ __forceinline__ unsigned resample_v_at(const uint8_t* __restrict__ tmp, int64_t row, int xe, int ymin,
                                                  int cnt, const int* __restrict__ k) {
    const uint8_t* s = tmp + ymin * row + xe;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) ss += s[y * row] * k[y];
    return clip8(ss);
}

// tmp n x [in_h][row] -> dst n x [out_h][row], row = out_w*c
__global__ __launch_bounds__(256) void resample_v_u8(const uint8_t* __restrict__ tmp, int64_t row, int n, int in_h,
                                                     int out_h, const int* __restrict__ bounds,
                                                     const int* __restrict__ kk, int ksize,
                                                     uint8_t* __restrict__ dst) {
    const int64_t total = row * out_h * n;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int yg = static_cast<int>(e / row), xe = static_cast<int>(e - yg * row);
        const int img = yg / out_h, yy = yg - img * out_h;
        dst[e] = static_cast<uint8_t>(resample_v_at(tmp + img * row * in_h, row, xe, bounds[2 * yy],
                                                    bounds[2 * yy + 1], kk + static_cast<int64_t>(yy) * ksize));
    }
}

// ---- the same two passes, shaped for bandwidth (round 5: detect() decodes hundreds of full-size masks per batch) -------------
// Horizontal, single-channel images: a thread owns FOUR consecutive output columns — their taps and coefficients live in
// registers for all the rows it walks — and writes them as one 32-bit store (a wave: 256 contiguous bytes per row). Taps beyond
// a column's count carry coefficient 0 and a clamped index, so the loads of a row are independent of the counts.
// grid (column groups of 1024, row chunks of ROWS), block 256. Needs out_w % 4 == 0 and ksize <= 3 (not shrinking by > 1... the
// BILINEAR support is 1 when enlarging: at most three taps); the generic kernel above takes every other case.
constexpr int RH_ROWS = 16;
__global__ __launch_bounds__(256) void resample_h1_u8_fast(const uint8_t* __restrict__ src, int total_rows, int in_h, int in_w,
                                                           int64_t image_stride, int64_t row_stride, int out_w,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk,
                                                           uint8_t* __restrict__ tmp) {
    const int x4 = (blockIdx.y * 256 + threadIdx.x) * 4;   // (rows on grid.x: hundreds of thousands of them; columns on grid.y)
    if (x4 >= out_w) return;
    int i0[4], i1[4], i2[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = x4 + j;
        const int xmin = bounds[2 * xx];
        i0[j] = xmin;
        i1[j] = min(xmin + 1, in_w - 1);
        i2[j] = min(xmin + 2, in_w - 1);
        k0[j] = kk[xx * 3];
        k1[j] = kk[xx * 3 + 1];
        k2[j] = kk[xx * 3 + 2];
    }
    const int y_lo = blockIdx.x * RH_ROWS, y_hi = min(y_lo + RH_ROWS, total_rows);
    for (int yg = y_lo; yg < y_hi; ++yg) {
        const int img = yg / in_h, y = yg - img * in_h;
        const uint8_t* s = src + img * image_stride + y * row_stride;
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ss = (1 << (PRECISION_BITS - 1)) + s[i0[j]] * k0[j] + s[i1[j]] * k1[j] + s[i2[j]] * k2[j];
            packed |= clip8_for_packing(ss) << (8 * j);
        }
        *reinterpret_cast<unsigned*>(tmp + static_cast<int64_t>(yg) * out_w + x4) = packed;
    }
}

// Vertical, any channel count (a row is just `row` bytes): a thread owns SIXTEEN consecutive bytes of an output row — one 16-byte
// load per tap row, one 16-byte store; the row's taps and coefficients are wave-uniform. grid (16-byte groups / 256, output
// rows of all images), block 256. Needs row % 16 == 0 and 16-byte aligned buffers.
__global__ __launch_bounds__(256) void resample_v_u8_fast(const uint8_t* __restrict__ tmp, int64_t row, int total_rows, int in_h, int out_h,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          int ksize, uint8_t* __restrict__ dst) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // short rows: several output rows per block (rpb of them, G = groups per row each), long rows: column blocks on grid.y
    const int G = static_cast<int>((row / 16 < 256 ? row / 16 : 256)), rpb = 256 / G;
    const int r = threadIdx.x / G, gi = threadIdx.x - r * G;
    const int64_t xe = (static_cast<int64_t>(blockIdx.y) * 256 + gi) * 16;
    const int yg = blockIdx.x * rpb + r;
    if (r >= rpb || xe >= row || yg >= total_rows) return;
    const int img = yg / out_h, yy = yg - img * out_h;
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + static_cast<int64_t>(yy) * ksize;
    const uint8_t* s = tmp + (static_cast<int64_t>(img) * in_h + ymin) * row + xe;
    int acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(s + t * row);
        const int kt = k[t];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += static_cast<int>((v[j >> 2] >> (8 * (j & 3))) & 255u) * kt;
    }
    u32x4 o = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j >> 2] |= clip8_for_packing(acc[j]) << (8 * (j & 3));
    *reinterpret_cast<u32x4*>(dst + static_cast<int64_t>(yg) * row + xe) = o;
}

// Both passes in ONE launch for the decode of full-size masks (single channel, enlarging: <= 3 taps each way, out_w % 16 == 0): a
// workgroup owns a tile of FV_ROWS x FV_COLS output pixels of one image. It stages the input footprint of the tile (<= FV_IN rows
// of <= FV_COLS + 4 bytes) in LDS with coalesced byte loads; if every byte of it is zero, so is the tile — the resample of zeros is
// exactly zero — and the workgroup stores zeros and is done (a pasted mask is zero outside its box: most tiles of most masks). Otherwise
// the horizontal pass runs from LDS into LDS ([rows][FV_COLS] bytes — the 8-bit intermediate of Pillow's two-pass resample, never written
// to memory: for 400 masks of 640 x 1024 -> 1200 x 1920 that is 491 MB written and read back) and the vertical pass from there. Same
// integer arithmetic, same intermediate rounding: bit-identical to the two launches.
constexpr int FV_ROWS = 16;     // output rows per workgroup
constexpr int FV_COLS = 256;    // output columns per workgroup
constexpr int FV_IN = 20;       // input rows a workgroup may need (enlarging: <= FV_ROWS + 2)
constexpr int FV_IN_W = FV_COLS + 8;   // input columns a workgroup may need (enlarging: <= FV_COLS + 3)
__global__ __launch_bounds__(256) void resample_hv1_u8_fused(const uint8_t* __restrict__ src, int in_h, int in_w,
                                                             int64_t image_stride, int64_t row_stride, int out_h, int out_w,
                                                             const int* __restrict__ bh, const int* __restrict__ kh,
                                                             const int* __restrict__ bv, const int* __restrict__ kv,
                                                             uint8_t* __restrict__ dst) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char in_lds[FV_IN * FV_IN_W];    // [rows_in][FV_IN_W]
    __shared__ __attribute__((aligned(16))) unsigned char mid_lds[FV_IN * FV_COLS];   // [rows_in][FV_COLS]
    const int tiles_x = (out_w + FV_COLS - 1) / FV_COLS, tiles_y = (out_h + FV_ROWS - 1) / FV_ROWS;
    const int tx = blockIdx.x % tiles_x, rest = blockIdx.x / tiles_x;
    const int ty = rest % tiles_y, img = rest / tiles_y;
    const int y0 = ty * FV_ROWS, y1 = min(y0 + FV_ROWS, out_h);
    const int x0 = tx * FV_COLS, x1 = min(x0 + FV_COLS, out_w);
    // the input rows and columns the tile reads: the bounds are monotone in the output coordinate
    const int in0 = bv[2 * y0], rows_in = min(min(bv[2 * (y1 - 1)] + 3, in_h) - in0, FV_IN);
    const int c0 = bh[2 * x0], cols_in = min(min(bh[2 * (x1 - 1)] + 3, in_w) - c0, FV_IN_W);
    const uint8_t* simg = src + img * image_stride + in0 * row_stride + c0;
    unsigned any = 0;
    for (int i = threadIdx.x; i < rows_in * cols_in; i += 256) {
        const int r = i / cols_in, c = i - r * cols_in;
        const unsigned v = simg[r * row_stride + c];
        in_lds[r * FV_IN_W + c] = static_cast<unsigned char>(v);
        any |= v;
    }
    const int groups = (x1 - x0) / 16;   // 16-byte store groups per tile row
    uint8_t* dtile = dst + (static_cast<int64_t>(img) * out_h + y0) * out_w + x0;
    if (!__syncthreads_or(static_cast<int>(any))) {
        for (int it = threadIdx.x; it < (y1 - y0) * groups; it += 256) {
            const int ry = it / groups, gx = it - ry * groups;
            *reinterpret_cast<u32x4*>(dtile + static_cast<int64_t>(ry) * out_w + gx * 16) = u32x4{0, 0, 0, 0};
        }
        return;
    }
    // ---- horizontal pass, LDS -> LDS: a thread owns four output columns; the four waves take the input rows in turn
    {
        const int q = threadIdx.x & 63, rg = threadIdx.x >> 6, x4 = x0 + q * 4;
        if (x4 < x1) {
            int i0[4], i1[4], i2[4], k0[4], k1[4], k2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = x4 + j, xmin = bh[2 * xx];
                i0[j] = xmin - c0; i1[j] = min(xmin + 1, in_w - 1) - c0; i2[j] = min(xmin + 2, in_w - 1) - c0;
                k0[j] = kh[xx * 3]; k1[j] = kh[xx * 3 + 1]; k2[j] = kh[xx * 3 + 2];
            }
            for (int r = rg; r < rows_in; r += 4) {
                const unsigned char* sr = in_lds + r * FV_IN_W;
                unsigned packed = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ss = (1 << (PRECISION_BITS - 1)) + sr[i0[j]] * k0[j] + sr[i1[j]] * k1[j] + sr[i2[j]] * k2[j];
                    packed |= clip8_for_packing(ss) << (8 * j);
                }
                *reinterpret_cast<unsigned*>(mid_lds + r * FV_COLS + q * 4) = packed;
            }
        }
    }
    __syncthreads();
    // ---- vertical pass from LDS: a thread owns sixteen bytes of an output row
    for (int it = threadIdx.x; it < (y1 - y0) * groups; it += 256) {
        const int ry = it / groups, gx = it - ry * groups, yy = y0 + ry;
        const int ymin = bv[2 * yy], cnt = bv[2 * yy + 1];
        const int* k = kv + yy * 3;
        int acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < cnt; ++t) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(mid_lds + (ymin - in0 + t) * FV_COLS + gx * 16);
            const int kt = k[t];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += static_cast<int>((v[j >> 2] >> (8 * (j & 3))) & 255u) * kt;
        }
        u32x4 o = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j >> 2] |= clip8_for_packing(acc[j]) << (8 * (j & 3));
        *reinterpret_cast<u32x4*>(dtile + static_cast<int64_t>(ry) * out_w + gx * 16) = o;
    }
}

// Vertical pass (or a plain read when RESIZE is false) fused with centre padding, mean subtraction and HWC -> CHW:
// dst[ch][oy][ox] = float(double(pixel) - mean[ch]); pad pixels are 0 before the subtraction (utils.py:86,
// model.py:1754: MEAN_PIXEL is a float64 array, so the reference subtracts in double and narrows afterwards).
template <bool RESIZE>
__global__ __launch_bounds__(256) void mold_kernel(const uint8_t* __restrict__ img, int new_h, int new_w, int top,
                                                   int left, int out_h, int out_w, const int* __restrict__ bounds,
                                                   const int* __restrict__ kk, int ksize, double m0, double m1,
                                                   double m2, float* __restrict__ dst, int64_t img_stride) {
    const int64_t plane = static_cast<int64_t>(out_h) * out_w;
    const int64_t row = static_cast<int64_t>(new_w) * 3;
    img += blockIdx.y * img_stride;            // image blockIdx.y of a batch of equally sized images
    dst += blockIdx.y * 3 * plane;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < plane;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int oy = static_cast<int>(e / out_w), ox = static_cast<int>(e - static_cast<int64_t>(oy) * out_w);
        const int yy = oy - top, xx = ox - left;
        unsigned p0 = 0, p1 = 0, p2 = 0;
        if (static_cast<unsigned>(yy) < static_cast<unsigned>(new_h) &&
            static_cast<unsigned>(xx) < static_cast<unsigned>(new_w)) {
            if constexpr (RESIZE) {
                const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
                const int* k = kk + static_cast<int64_t>(yy) * ksize;
                p0 = resample_v_at(img, row, xx * 3 + 0, ymin, cnt, k);
                p1 = resample_v_at(img, row, xx * 3 + 1, ymin, cnt, k);
                p2 = resample_v_at(img, row, xx * 3 + 2, ymin, cnt, k);
            } else {
                const uint8_t* s = img + yy * row + xx * 3;
                p0 = s[0]; p1 = s[1]; p2 = s[2];
            }
        }
        dst[e] = static_cast<float>(static_cast<double>(p0) - m0);
        dst[plane + e] = static_cast<float>(static_cast<double>(p1) - m1);
        dst[2 * plane + e] = static_cast<float>(static_cast<double>(p2) - m2);
    }
}

struct ResizePlan {
    Axis ah, av;
    size_t off_bh, off_kh, off_bv, off_kv, total;  // tmp at offset 0
};

ResizePlan plan_resize(int n, int in_h, int in_w, int c, int out_h, int out_w) {
    ResizePlan p;
    p.ah = make_axis(in_w, out_w);
    p.av = make_axis(in_h, out_h);
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    size_t o = up(static_cast<size_t>(n) * in_h * out_w * c);
    p.off_bh = o; o += up(sizeof(int) * 2 * out_w);
    p.off_kh = o; o += up(sizeof(int) * static_cast<size_t>(p.ah.ksize) * out_w);
    p.off_bv = o; o += up(sizeof(int) * 2 * out_h);
    p.off_kv = o; o += up(sizeof(int) * static_cast<size_t>(p.av.ksize) * out_h);
    p.total = o;
    return p;
}

unsigned blocks_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return static_cast<unsigned>(b < 1 ? 1 : b > 256 * 32 ? 256 * 32 : b);
}

// coefficient tables + horizontal pass; leaves the intermediate at workspace[0..)
int run_horizontal(const uint8_t* src, int n, int in_h, int in_w, int c, int64_t image_stride, int64_t row_stride,
                   int out_h, int out_w, const ResizePlan& p, unsigned char* ws, hipStream_t s) {
    (void)out_h;
    int* bh = reinterpret_cast<int*>(ws + p.off_bh);
    int* kh = reinterpret_cast<int*>(ws + p.off_kh);
    int* bv = reinterpret_cast<int*>(ws + p.off_bv);
    int* kv = reinterpret_cast<int*>(ws + p.off_kv);
    hipLaunchKernelGGL(coeffs_kernel, dim3((out_w + 255) / 256), dim3(256), 0, s, p.ah, bh, kh);
    hipLaunchKernelGGL(coeffs_kernel, dim3((out_h + 255) / 256), dim3(256), 0, s, p.av, bv, kv);
    if (c == 1 && p.ah.ksize == 3 && out_w % 4 == 0) {
        const dim3 grid(static_cast<unsigned>((static_cast<int64_t>(n) * in_h + RH_ROWS - 1) / RH_ROWS),
                        static_cast<unsigned>((out_w / 4 + 255) / 256));
        hipLaunchKernelGGL(resample_h1_u8_fast, grid, dim3(256), 0, s, src, n * in_h, in_h, in_w, image_stride, row_stride, out_w,
                           bh, kh, ws);
        return mrcnn::check_launch("resample_h1_u8_fast");
    }
    hipLaunchKernelGGL(resample_h_u8, dim3(blocks_for(static_cast<int64_t>(n) * in_h * out_w * c)), dim3(256), 0, s,
                       src, n, in_h, c, image_stride, row_stride, out_w, bh, kh, p.ah.ksize, ws);
    return mrcnn::check_launch("resample_h_u8");
}

bool resize_args_ok(int in_h, int in_w, int c, int out_h, int out_w) {
    return in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1 && c >= 1 && c <= 4 && in_h <= 16384 && in_w <= 16384 &&
           out_h <= 16384 && out_w <= 16384;
}

// ---------------------------------------------------------------------------------------------------------------
// full_masks: the canvases are cleared with one memset (a pure HBM write stream: the bulk of the bytes), then one
// workgroup per (detection, TILE_ROWS output rows) that touches the box stages the 8-bit mask, runs the horizontal
// pass for the box's columns into LDS and the vertical pass for its rows, thresholds (> 127) and overwrites the
// 16-byte groups that overlap the box. Workgroups whose rows miss the box exit at once.
constexpr int TILE_ROWS = 32;
constexpr int CHUNK = 256;  // box columns staged per pass (== the workgroup size: a thread per column)

struct PasteParams {
    const float* masks;        // element (det, y, x, class) at det*sn + y*sy + x*sx + class*sc
    int64_t sn, sy, sx, sc;
    const int64_t* class_ids;  // [n]
    const float* boxes;        // [n][4]
    uint8_t* out;              // [n][H][W]
    int n, mh, mw, C, H, W;
    unsigned on_value;  // byte written where the resized mask > 127 (1: boolean view, 255: an 'L' image)
    int off_hk, off_tmp, off_vb, off_vk;  // byte offsets into dynamic LDS (mask bytes at 0)
};

template <int PX>  // pixels per thread in the store loops: 16 (one 16-byte store; needs W % 16 == 0) or 4
__global__ __launch_bounds__(256) void paste_masks_kernel(const PasteParams p) {
    typedef unsigned int store_t __attribute__((ext_vector_type(PX / 4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // consecutive workgroups (which the dispatcher deals round-robin to XCDs and CUs) take the same row tile of
    // consecutive detections: tiles that hit their boxes then spread over the chip instead of piling onto the few
    // CUs a (tile fastest) order would send every detection's tile t to
    const int det = blockIdx.x % p.n, row0 = (blockIdx.x / p.n) * TILE_ROWS;
    const int row1 = min(row0 + TILE_ROWS, p.H);
    const int tid = threadIdx.x;
    const float* b = p.boxes + 4 * static_cast<int64_t>(det);
    const double y1 = b[0], x1 = b[1], y2 = b[2], x2 = b[3];
    const int bh = static_cast<int>(y2 - y1), bw = static_cast<int>(x2 - x1);  // data.py:295 int(box.height()), ...
    const int top = static_cast<int>(y1), left = static_cast<int>(x1);         // :298,300
    const int64_t cls = p.class_ids[det];
    // boxes the reference cannot paste (empty, or not inside the canvas: PIL raises) give an all-zero mask
    const bool valid = bh > 0 && bw > 0 && top >= 0 && left >= 0 && top + bh <= p.H && left + bw <= p.W &&
                       cls >= 0 && cls < p.C;
    uint8_t* out = p.out + static_cast<int64_t>(det) * p.H * p.W;
    if (!valid || row0 >= top + bh || row1 <= top) return;  // the canvas is already zero
    uint8_t* m8 = lds;
    int* hk = reinterpret_cast<int*>(lds + p.off_hk);
    uint8_t* tmp = lds + p.off_tmp;  // [mh][bw]
    int* vb = reinterpret_cast<int*>(lds + p.off_vb);
    int* vk = reinterpret_cast<int*>(lds + p.off_vk);

    // mask channel * 255.0 (fp32, data.py:291) -> 'L' (Pillow f2l: clamp, truncate)
    const float* msrc = p.masks + det * p.sn + cls * p.sc;
    for (int i = tid; i < p.mh * p.mw; i += 256) {
        const int my = i / p.mw, mx = i - my * p.mw;
        const float v = msrc[my * p.sy + mx * p.sx] * 255.0f;
        m8[i] = v <= 0.0f ? 0 : v >= 255.0f ? 255 : static_cast<uint8_t>(v);
    }
    auto axis = [](int in_size, int out_size) {
        Axis a;
        a.in_size = in_size; a.out_size = out_size;
        double fs = a.scale = static_cast<double>(static_cast<float>(in_size)) / out_size;
        if (fs < 1.0) fs = 1.0;
        a.support = 1.0 * fs; a.ksize = static_cast<int>(ceil(a.support)) * 2 + 1; a.ss = 1.0 / fs;
        return a;
    };
    const Axis av = axis(p.mh, bh), ah = axis(p.mw, bw);
    // vertical coefficients of this tile's rows that lie in the box, packed from the first such row
    const int rfirst = max(top - row0, 0), rlast = min(top + bh, row1) - row0;  // tile rows [rfirst, rlast)
    if (tid >= rfirst && tid < rlast) {
        int ymin, cnt;
        axis_coeffs(av, row0 + tid - top, ymin, cnt, vk + (tid - rfirst) * av.ksize, 1);
        vb[2 * tid] = ymin;
        vb[2 * tid + 1] = cnt;
    }
    // Column chunks of CHUNK canvas pixels, aligned to the store groups. Per chunk: horizontal pass for its box
    // columns (a thread per column; its coefficients live in its own LDS slot) into tmp [mh][CHUNK], then the
    // vertical pass + threshold (data.py:308 mask > 127) for the tile's box rows, PX pixels and one store per thread.
    const int cb = (left / PX) * PX;
    for (int c0 = cb; c0 < left + bw; c0 += CHUNK) {
        __syncthreads();  // mask / coefficients staged; previous chunk's tmp consumed
        {
            const int xx = c0 + tid - left;
            if (xx >= 0 && xx < bw) {
                int* k = hk + (tid - max(left - c0, 0)) * ah.ksize;  // slots packed from the first box column
                int xmin, cnt;
                axis_coeffs(ah, xx, xmin, cnt, k, 1);
                if (ah.ksize == 3) {
                    // not shrinking (the usual case): at most 3 taps. Taps beyond cnt have coefficient 0 and a clamped
                    // index, so the rows' LDS reads are independent of each other and of cnt and can all be in flight
                    const int k0 = k[0], k1 = k[1], k2 = k[2];
                    const int i0 = xmin, i1 = min(xmin + 1, p.mw - 1), i2 = min(xmin + 2, p.mw - 1);
#pragma unroll 4
                    for (int r = 0; r < p.mh; ++r) {
                        const uint8_t* s = m8 + r * p.mw;
                        const int ss = (1 << (PRECISION_BITS - 1)) + s[i0] * k0 + s[i1] * k1 + s[i2] * k2;
                        tmp[r * CHUNK + tid] = static_cast<uint8_t>(clip8(ss));
                    }
                } else {
                    for (int r = 0; r < p.mh; ++r) {
                        const uint8_t* s = m8 + r * p.mw + xmin;
                        int ss = 1 << (PRECISION_BITS - 1);
                        for (int x = 0; x < cnt; ++x) ss += s[x] * k[x];
                        tmp[r * CHUNK + tid] = static_cast<uint8_t>(clip8(ss));
                    }
                }
            }
        }
        __syncthreads();
        constexpr int GPC = CHUNK / PX;  // store groups per chunk row
        const int ngroups = min(GPC, (left + bw - c0 + PX - 1) / PX);
        for (int i = tid; i < (rlast - rfirst) * ngroups; i += 256) {
            const int ri = i / ngroups, gi = i - ri * ngroups, rr = rfirst + ri;
            const int ymin = vb[2 * rr], cnt = vb[2 * rr + 1];
            const int* k = vk + ri * av.ksize;
            store_t packed = {};
            if (av.ksize == 3) {  // at most 3 taps (see the horizontal pass): three PX-byte LDS reads per store group
                const int k0 = k[0], k1 = k[1], k2 = k[2];
                const store_t r0 = *reinterpret_cast<const store_t*>(tmp + ymin * CHUNK + gi * PX);
                const store_t r1 = *reinterpret_cast<const store_t*>(tmp + min(ymin + 1, p.mh - 1) * CHUNK + gi * PX);
                const store_t r2 = *reinterpret_cast<const store_t*>(tmp + min(ymin + 2, p.mh - 1) * CHUNK + gi * PX);
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const int xx = c0 + gi * PX + j - left;
                    const unsigned w0 = r0[j / 4], w1 = r1[j / 4], w2 = r2[j / 4];
                    const int sh = 8 * (j & 3);
                    const int ss = (1 << (PRECISION_BITS - 1)) + static_cast<int>((w0 >> sh) & 255u) * k0 +
                                   static_cast<int>((w1 >> sh) & 255u) * k1 + static_cast<int>((w2 >> sh) & 255u) * k2;
                    if (xx >= 0 && xx < bw && clip8(ss) > 127u) packed[j / 4] |= p.on_value << sh;
                }
            } else {
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const int cx = gi * PX + j, xx = c0 + cx - left;  // chunk column, box column
                    if (xx >= 0 && xx < bw) {
                        const uint8_t* s = tmp + ymin * CHUNK + cx;
                        int ss = 1 << (PRECISION_BITS - 1);
                        for (int y = 0; y < cnt; ++y) ss += s[y * CHUNK] * k[y];
                        if (clip8(ss) > 127u) packed[j / 4] |= p.on_value << (8 * (j & 3));
                    }
                }
            }
            reinterpret_cast<store_t*>(out + static_cast<int64_t>(row0 + rr) * p.W + c0)[gi] = packed;
        }
    }
}

}  // namespace

extern "C" size_t mrcnn_resize_u8_workspace_bytes(int32_t n, int32_t in_h, int32_t in_w, int32_t channels,
                                                  int32_t out_h, int32_t out_w) {
    if (n < 1 || !resize_args_ok(in_h, in_w, channels, out_h, out_w)) return 0;
    return plan_resize(n, in_h, in_w, channels, out_h, out_w).total;
}

extern "C" int mrcnn_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t in_h, int32_t in_w, int32_t channels,
                                        int64_t src_image_stride, int64_t src_row_stride, uint8_t* dst,
                                        int32_t out_h, int32_t out_w, void* workspace, size_t workspace_bytes,
                                        mrcnn_stream_t stream) {
    MRCNN_REQUIRE(src && dst && workspace, "resize_u8: null pointer");
    MRCNN_REQUIRE(n >= 1 && n <= 65536, "resize_u8: n=%d", n);
    MRCNN_REQUIRE(src_row_stride >= static_cast<int64_t>(in_w) * channels && src_image_stride >= 0,
                  "resize_u8: row stride %lld shorter than a row", static_cast<long long>(src_row_stride));
    MRCNN_REQUIRE(resize_args_ok(in_h, in_w, channels, out_h, out_w),
                  "resize_u8: bad shape %dx%dx%d -> %dx%d (sizes 1..16384, channels 1..4)", in_h, in_w, channels, out_h,
                  out_w);
    const ResizePlan p = plan_resize(n, in_h, in_w, channels, out_h, out_w);
    MRCNN_REQUIRE(workspace_bytes >= p.total, "resize_u8: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    hipStream_t s = mrcnn::as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    // single-channel enlargement with aligned rows (decode_masks): both passes in one launch, the intermediate stays in LDS
    const int64_t fv_tiles = static_cast<int64_t>(n) * ((out_h + FV_ROWS - 1) / FV_ROWS) * ((out_w + FV_COLS - 1) / FV_COLS);
    if (channels == 1 && p.ah.ksize == 3 && p.av.ksize == 3 && out_w % 16 == 0 && out_h >= in_h && out_w >= in_w &&
        (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && fv_tiles <= 0x7FFFFFFF) {
        int* bh = reinterpret_cast<int*>(ws + p.off_bh);
        int* kh = reinterpret_cast<int*>(ws + p.off_kh);
        int* bv = reinterpret_cast<int*>(ws + p.off_bv);
        int* kv = reinterpret_cast<int*>(ws + p.off_kv);
        hipLaunchKernelGGL(coeffs_kernel, dim3((out_w + 255) / 256), dim3(256), 0, s, p.ah, bh, kh);
        hipLaunchKernelGGL(coeffs_kernel, dim3((out_h + 255) / 256), dim3(256), 0, s, p.av, bv, kv);
        hipLaunchKernelGGL(resample_hv1_u8_fused, dim3(static_cast<unsigned>(fv_tiles)), dim3(256), 0, s, src, in_h, in_w,
                           src_image_stride, src_row_stride, out_h, out_w, bh, kh, bv, kv, dst);
        return mrcnn::check_launch("resample_hv1_u8_fused");
    }
    if (int rc = run_horizontal(src, n, in_h, in_w, channels, src_image_stride, src_row_stride, out_h, out_w, p, ws, s))
        return rc;
    const int64_t row = static_cast<int64_t>(out_w) * channels;
    if (row % 16 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
        const int G = static_cast<int>(row / 16 < 256 ? row / 16 : 256), rpb = 256 / G;
        const int64_t total_rows = static_cast<int64_t>(n) * out_h;
        const dim3 grid(static_cast<unsigned>((total_rows + rpb - 1) / rpb), static_cast<unsigned>((row / 16 + 255) / 256));
        hipLaunchKernelGGL(resample_v_u8_fast, grid, dim3(256), 0, s, ws, row, static_cast<int>(total_rows), in_h, out_h,
                           reinterpret_cast<const int*>(ws + p.off_bv), reinterpret_cast<const int*>(ws + p.off_kv), p.av.ksize, dst);
        return mrcnn::check_launch("resample_v_u8_fast");
    }
    hipLaunchKernelGGL(resample_v_u8, dim3(blocks_for(row * out_h * n)), dim3(256), 0, s, ws, row, n, in_h, out_h,
                       reinterpret_cast<const int*>(ws + p.off_bv), reinterpret_cast<const int*>(ws + p.off_kv),
                       p.av.ksize, dst);
    return mrcnn::check_launch("resample_v_u8");
}

extern "C" int mrcnn_mold_images_u8(const uint8_t* src, int32_t n, int64_t src_image_stride, int32_t in_h, int32_t in_w,
                                    int32_t new_h, int32_t new_w, int32_t top, int32_t left, int32_t out_h, int32_t out_w,
                                    const double mean_pixel[3], float* dst, void* workspace, size_t workspace_bytes,
                                    mrcnn_stream_t stream) {
    MRCNN_REQUIRE(src && dst && mean_pixel, "mold_image: null pointer");
    MRCNN_REQUIRE(n >= 1 && n <= 65535 && src_image_stride >= static_cast<int64_t>(in_h) * in_w * 3, "mold_image: n=%d / image stride", n);
    MRCNN_REQUIRE(resize_args_ok(in_h, in_w, 3, new_h, new_w) && out_h >= 1 && out_w >= 1 && out_h <= 16384 &&
                      out_w <= 16384, "mold_image: bad shape");
    MRCNN_REQUIRE(top >= 0 && left >= 0 && top + new_h <= out_h && left + new_w <= out_w,
                  "mold_image: the resized image (%dx%d at %d,%d) does not fit the %dx%d output", new_h, new_w, top,
                  left, out_h, out_w);
    hipStream_t s = mrcnn::as_stream(stream);
    const dim3 grid(blocks_for(static_cast<int64_t>(out_h) * out_w), static_cast<unsigned>(n));
    if (new_h == in_h && new_w == in_w) {  // scale == 1 (utils.py:72): no resample at all
        hipLaunchKernelGGL(mold_kernel<false>, grid, dim3(256), 0, s, src, new_h, new_w, top, left, out_h, out_w,
                           nullptr, nullptr, 0, mean_pixel[0], mean_pixel[1], mean_pixel[2], dst, src_image_stride);
        return mrcnn::check_launch("mold_kernel");
    }
    const ResizePlan p = plan_resize(n, in_h, in_w, 3, new_h, new_w);
    MRCNN_REQUIRE(workspace && workspace_bytes >= p.total, "mold_image: workspace too small (%zu < %zu)",
                  workspace_bytes, p.total);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    if (int rc = run_horizontal(src, n, in_h, in_w, 3, src_image_stride, static_cast<int64_t>(in_w) * 3, new_h, new_w, p, ws, s))
        return rc;
    // the horizontal pass left image i's rows at ws + i * in_h * new_w * 3
    hipLaunchKernelGGL(mold_kernel<true>, grid, dim3(256), 0, s, ws, new_h, new_w, top, left, out_h, out_w,
                       reinterpret_cast<const int*>(ws + p.off_bv), reinterpret_cast<const int*>(ws + p.off_kv),
                       p.av.ksize, mean_pixel[0], mean_pixel[1], mean_pixel[2], dst,
                       static_cast<int64_t>(in_h) * new_w * 3);
    return mrcnn::check_launch("mold_kernel");
}

extern "C" int mrcnn_mold_image_u8(const uint8_t* src, int32_t in_h, int32_t in_w, int32_t new_h, int32_t new_w,
                                   int32_t top, int32_t left, int32_t out_h, int32_t out_w, const double mean_pixel[3],
                                   float* dst, void* workspace, size_t workspace_bytes, mrcnn_stream_t stream) {
    return mrcnn_mold_images_u8(src, 1, static_cast<int64_t>(in_h) * in_w * 3, in_h, in_w, new_h, new_w, top, left, out_h, out_w,
                                mean_pixel, dst, workspace, workspace_bytes, stream);
}

extern "C" int mrcnn_paste_masks_u8(const float* masks, int64_t stride_n, int64_t stride_y, int64_t stride_x,
                                    int64_t stride_c, int32_t n, int32_t mask_h, int32_t mask_w,
                                    int32_t num_classes, const int64_t* class_ids, const float* boxes, int32_t height,
                                    int32_t width, int32_t on_value, uint8_t* out, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(n >= 0, "paste_masks: n=%d", n);
    MRCNN_REQUIRE(on_value >= 1 && on_value <= 255, "paste_masks: on_value=%d must be in [1,255]", on_value);
    if (n == 0) return MRCNN_OK;
    MRCNN_REQUIRE(masks && class_ids && boxes && out, "paste_masks: null pointer");
    MRCNN_REQUIRE(mask_h >= 1 && mask_w >= 1 && mask_h <= 64 && mask_w <= 64 && num_classes >= 1,
                  "paste_masks: mask %dx%d (1..64), classes %d", mask_h, mask_w, num_classes);
    MRCNN_REQUIRE(height >= 1 && width >= 4 && width % 4 == 0 && height <= 16384 && width <= 16384 && n <= 65535,
                  "paste_masks: canvas %dx%d (width %% 4 == 0 required), n=%d (<= 65535)", height, width, n);
    PasteParams p;
    p.masks = masks; p.sn = stride_n; p.sy = stride_y; p.sx = stride_x; p.sc = stride_c; p.class_ids = class_ids; p.boxes = boxes; p.out = out;
    p.n = n; p.mh = mask_h; p.mw = mask_w; p.C = num_classes; p.H = height; p.W = width;
    p.on_value = static_cast<unsigned>(on_value);
    auto up = [](int v) { return (v + 15) & ~15; };
    int o = up(mask_h * mask_w);
    p.off_hk = o;
    // per-thread coefficient slots of one 256-column chunk: ksize 3 when enlarging (<= 256*3 ints); when shrinking
    // there are bw < mask_w columns of ksize <= 2*ceil(mask_w/bw)+1 taps: bw*ksize <= 5*mask_w ints
    o += up(4 * (256 * 3 > 5 * mask_w ? 256 * 3 : 5 * mask_w));
    p.off_tmp = o; o += up(mask_h * CHUNK);
    p.off_vb = o; o += up(4 * 2 * TILE_ROWS);
    // vertical coefficients of the tile's box rows: ksize 3 when enlarging (<= TILE_ROWS*3 ints); when shrinking there
    // are bh < mask_h rows of <= 2*ceil(mask_h/bh)+1 taps: <= 5*mask_h ints
    p.off_vk = o; o += up(4 * (TILE_ROWS * 3 > 5 * mask_h ? TILE_ROWS * 3 : 5 * mask_h));
    hipStream_t s = mrcnn::as_stream(stream);
    hipError_t e = hipMemsetAsync(out, 0, static_cast<size_t>(n) * height * width, s);
    if (e != hipSuccess) return mrcnn::fail(MRCNN_ERR_LAUNCH, "paste_masks: hipMemsetAsync: %s", hipGetErrorString(e));
    const dim3 grid(static_cast<unsigned>((height + TILE_ROWS - 1) / TILE_ROWS) * n);
    if (width % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0)
        hipLaunchKernelGGL(paste_masks_kernel<16>, grid, dim3(256), o, s, p);
    else
        hipLaunchKernelGGL(paste_masks_kernel<4>, grid, dim3(256), o, s, p);
    return mrcnn::check_launch("paste_masks_kernel");
}
