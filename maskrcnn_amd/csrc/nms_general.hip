// Greedy NMS for ANY number of boxes and for float64 boxes — the rest of the drop-in surface of
//   at::Tensor nms(const at::Tensor& dets, const float threshold)        c++ext/maskrcnn/csrc/nms.h:15-30
// whose CPU path dispatches over the floating types (cpu/nms_cpu.cpp:73-79, AT_DISPATCH_FLOATING_TYPES) and takes any N.
// nms.hip serves what model.py asks for (fp32, <= 16384 boxes per segment, LDS-resident or pair-mask paths); this file is the
// general single-segment path: no O(N^2) memory, arithmetic in the boxes' own type, same visiting order → same keep set.
//   1. score order: stable LSD radix sort (hipCUB) of (monotone key of the score, input index) — descending score, NaN
//      first, -0 == +0, ties by ascending index (the order nms.hip's bitonic sort produces)
//   2. gather the boxes into that order; areas as nms_cpu.cpp:26
//   3. per 64-box chunk of the order, ONE launch: every workgroup resolves the chunk (a wave ballots the chunk's 64 x 64
//      IoU >= thr matrix and walks its alive boxes), then tests its own 256 later boxes against the chunk's survivors.
//      Chunks are ordered by the kernel boundary; the chunk's alive flags may be rewritten by workgroup 0 while another
//      workgroup still reads them — harmless: in every intermediate state each suppressed box's suppressor is still alive,
//      so the walk returns the same survivors.
//   4. survivors → flags by input index → ascending indices (ballot / popcount compaction)
// IoU arithmetic is the reference's, op for op, FP contraction off (see nms.hip).
#pragma clang fp contract(off)

#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace {

using u64 = unsigned long long;
using u32 = unsigned int;

template <typename T>
struct BoxT {
    T y1, x1, y2, x2, area;
};

template <typename T>
__device__ __forceinline__ bool iou_ge(const BoxT<T>& i, const BoxT<T>& j, float thr) {
    const T xx1 = (i.x1 < j.x1) ? j.x1 : i.x1;  // std::max(ix1, x1[j])          nms_cpu.cpp:54-65
    const T yy1 = (i.y1 < j.y1) ? j.y1 : i.y1;
    const T xx2 = (j.x2 < i.x2) ? j.x2 : i.x2;  // std::min(ix2, x2[j])
    const T yy2 = (j.y2 < i.y2) ? j.y2 : i.y2;
    T tw = xx2 - xx1;
    tw = tw + static_cast<T>(1);
    T th = yy2 - yy1;
    th = th + static_cast<T>(1);
    const T w = (static_cast<T>(0) < tw) ? tw : static_cast<T>(0);
    const T h = (static_cast<T>(0) < th) ? th : static_cast<T>(0);
    const T inter = w * h;
    T uni = i.area + j.area;
    uni = uni - inter;
    const T ovr = inter / uni;
    return ovr >= static_cast<T>(thr);  // `ovr >= threshold` with a float threshold: the usual arithmetic conversion
}

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = u32;
    __device__ static u32 make(float s) {
        u32 u = __float_as_uint(s);
        if (s != s) u = 0x7FC00000u;   // any NaN → +qNaN (largest)
        if (s == 0.0f) u = 0u;         // -0 → +0
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return ~u;                     // ascending key == descending score
    }
};
template <> struct KeyOf<double> {
    using type = u64;
    __device__ static u64 make(double s) {
        u64 u = static_cast<u64>(__double_as_longlong(s));
        if (s != s) u = 0x7FF8000000000000ull;
        if (s == 0.0) u = 0ull;
        u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
        return ~u;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void nmsg_keys_kernel(const T* __restrict__ dets, int64_t n, int64_t row_stride,
                                                        int64_t col_stride, typename KeyOf<T>::type* __restrict__ keys,
                                                        u32* __restrict__ vals) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = KeyOf<T>::make(dets[i * row_stride + 4 * col_stride]);
    vals[i] = static_cast<u32>(i);
}

template <typename T>
__global__ __launch_bounds__(256) void nmsg_gather_kernel(const T* __restrict__ dets, int64_t n, int64_t row_stride,
                                                          int64_t col_stride, const u32* __restrict__ order,
                                                          T* __restrict__ box, T* __restrict__ area,
                                                          unsigned char* __restrict__ alive,
                                                          unsigned char* __restrict__ keepf) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const T* r = dets + static_cast<int64_t>(order[p]) * row_stride;
    const T y1 = r[0], x1 = r[col_stride], y2 = r[2 * col_stride], x2 = r[3 * col_stride];
    box[p * 4 + 0] = y1; box[p * 4 + 1] = x1; box[p * 4 + 2] = y2; box[p * 4 + 3] = x2;
    T w = x2 - x1;
    w = w + static_cast<T>(1);
    T h = y2 - y1;
    h = h + static_cast<T>(1);
    area[p] = w * h;  // nms_cpu.cpp:26
    alive[p] = 1;
    keepf[p] = 0;
}

// chunk c = sorted positions [64 c, 64 c + 64). grid.x = max(1, ceil((n - 64 (c + 1)) / 256)), block = 256.
template <typename T>
__global__ __launch_bounds__(256) void nmsg_chunk_kernel(const T* __restrict__ box, const T* __restrict__ area,
                                                         unsigned char* __restrict__ alive, u64* __restrict__ keptw,
                                                         int64_t n, int64_t c, float thr) {
    __shared__ BoxT<T> cb[64];
    __shared__ u64 kept_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t c0 = c * 64;
    const int64_t pj = c0 + lane;
    const bool valid = pj < n;
    BoxT<T> bj = {0, 0, 0, 0, 0};
    if (tid < 64) {
        if (valid) bj = {box[pj * 4], box[pj * 4 + 1], box[pj * 4 + 2], box[pj * 4 + 3], area[pj]};
        cb[lane] = bj;
    }
    __syncthreads();
    if (tid < 64) {
        const bool al = valid && alive[pj] != 0;
        u64 mine = 0;  // lane i: later boxes of the chunk that box i suppresses
        for (int i = 0; i < 64; ++i) {
            const BoxT<T> bi = cb[i];
            const bool hit = (lane > i) && valid && iou_ge(bi, bj, thr);
            const u64 m = __ballot(hit);
            if (lane == i) mine = m;
        }
        const u32 cm_lo = static_cast<u32>(mine), cm_hi = static_cast<u32>(mine >> 32);
        u64 av = __ballot(al);
        u64 kept = 0, rem = av;
        while (rem) {
            const int i = __builtin_ctzll(rem);
            kept |= 1ull << i;
            const u32 m_hi = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(cm_hi), i));
            const u32 m_lo = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(cm_lo), i));
            av &= ~((static_cast<u64>(m_hi) << 32) | m_lo);
            rem = av & ~((2ull << i) - 1ull);
        }
        if (lane == 0) kept_s = kept;
        if (blockIdx.x == 0) {
            if (valid) alive[pj] = ((kept >> lane) & 1ull) ? 1 : 0;
            if (lane == 0) keptw[c] = kept;
        }
    }
    __syncthreads();
    const u64 kept = kept_s;
    const int64_t p = c0 + 64 + static_cast<int64_t>(blockIdx.x) * 256 + tid;
    if (p >= n || kept == 0 || alive[p] == 0) return;
    const BoxT<T> bp = {box[p * 4], box[p * 4 + 1], box[p * 4 + 2], box[p * 4 + 3], area[p]};
    u64 m = kept;
    while (m) {
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        if (iou_ge(cb[i], bp, thr)) {
            alive[p] = 0;
            break;
        }
    }
}

__global__ __launch_bounds__(256) void nmsg_flags_kernel(const u64* __restrict__ keptw, const u32* __restrict__ order,
                                                         int64_t n, unsigned char* __restrict__ keepf) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    if ((keptw[p >> 6] >> (p & 63)) & 1ull) keepf[order[p]] = 1;
}

// one wave: ascending-index compaction, 64 indices per step
__global__ __launch_bounds__(64) void nmsg_compact_kernel(const unsigned char* __restrict__ keepf, int64_t n,
                                                          int64_t* __restrict__ keep_out, int64_t* __restrict__ count_out) {
    const int lane = threadIdx.x;
    int64_t pos = 0;
    for (int64_t w0 = 0; w0 < n; w0 += 64) {
        const int64_t i = w0 + lane;
        const bool f = i < n && keepf[i] != 0;
        const u64 m = __ballot(f);
        if (f) keep_out[pos + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        pos += __builtin_popcountll(m);
    }
    for (int64_t i = pos + lane; i < n; i += 64) keep_out[i] = -1;
    if (lane == 0) *count_out = pos;
}

template <typename T>
struct Layout {
    size_t keys_in, keys_out, vals_in, vals_out, box, area, alive, keepf, keptw, cub, total, cub_bytes;
};

template <typename T>
int layout(int64_t n, Layout<T>* L) {
    using K = typename KeyOf<T>::type;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L->keys_in = take(sizeof(K) * n); L->keys_out = take(sizeof(K) * n);
    L->vals_in = take(sizeof(u32) * n); L->vals_out = take(sizeof(u32) * n);
    L->box = take(sizeof(T) * 4 * n); L->area = take(sizeof(T) * n);
    L->alive = take(n); L->keepf = take(n);
    L->keptw = take(sizeof(u64) * ((n + 63) / 64));
    size_t cub_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, static_cast<const K*>(nullptr), static_cast<K*>(nullptr),
                                                      static_cast<const u32*>(nullptr), static_cast<u32*>(nullptr),
                                                      static_cast<int>(n), 0, static_cast<int>(sizeof(K) * 8), nullptr);
    if (e != hipSuccess) return mrcnn::fail(MRCNN_ERR_LAUNCH, "nms_general: radix sort sizing: %s", hipGetErrorString(e));
    L->cub_bytes = cub_bytes;
    L->cub = take(cub_bytes);
    L->total = off;
    return MRCNN_OK;
}

template <typename T>
int run(const T* dets, int64_t n, int64_t row_stride, int64_t col_stride, float thr, int64_t* keep_out,
        int64_t* count_out, void* workspace, size_t workspace_bytes, hipStream_t s) {
    using K = typename KeyOf<T>::type;
    Layout<T> L;
    if (int rc = layout<T>(n, &L)) return rc;
    MRCNN_REQUIRE(workspace_bytes >= L.total, "nms_general: workspace too small (%zu < %zu bytes)", workspace_bytes, L.total);
    unsigned char* b = static_cast<unsigned char*>(workspace);
    K* keys_in = reinterpret_cast<K*>(b + L.keys_in);
    K* keys_out = reinterpret_cast<K*>(b + L.keys_out);
    u32* vals_in = reinterpret_cast<u32*>(b + L.vals_in);
    u32* order = reinterpret_cast<u32*>(b + L.vals_out);
    T* box = reinterpret_cast<T*>(b + L.box);
    T* area = reinterpret_cast<T*>(b + L.area);
    unsigned char* alive = b + L.alive;
    unsigned char* keepf = b + L.keepf;
    u64* keptw = reinterpret_cast<u64*>(b + L.keptw);
    const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
    hipLaunchKernelGGL(nmsg_keys_kernel<T>, dim3(blocks), dim3(256), 0, s, dets, n, row_stride, col_stride, keys_in, vals_in);
    if (int rc = mrcnn::check_launch("nmsg_keys_kernel")) return rc;
    size_t cub_bytes = L.cub_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(b + L.cub, cub_bytes, keys_in, keys_out, vals_in, order, static_cast<int>(n),
                                                      0, static_cast<int>(sizeof(K) * 8), s);
    if (e != hipSuccess) return mrcnn::fail(MRCNN_ERR_LAUNCH, "nms_general: radix sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(nmsg_gather_kernel<T>, dim3(blocks), dim3(256), 0, s, dets, n, row_stride, col_stride, order, box, area,
                       alive, keepf);
    if (int rc = mrcnn::check_launch("nmsg_gather_kernel")) return rc;
    const int64_t nchunks = (n + 63) / 64;
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t later = n - 64 * (c + 1);
        const unsigned grid = later > 0 ? static_cast<unsigned>((later + 255) / 256) : 1u;
        hipLaunchKernelGGL(nmsg_chunk_kernel<T>, dim3(grid), dim3(256), 0, s, box, area, alive, keptw, n, c, thr);
    }
    if (int rc = mrcnn::check_launch("nmsg_chunk_kernel")) return rc;
    hipLaunchKernelGGL(nmsg_flags_kernel, dim3(blocks), dim3(256), 0, s, keptw, order, n, keepf);
    hipLaunchKernelGGL(nmsg_compact_kernel, dim3(1), dim3(64), 0, s, keepf, n, keep_out, count_out);
    return mrcnn::check_launch("nmsg_compact_kernel");
}

}  // namespace

// dtype: 0 = float32, 1 = float64. n <= 2^31 - 64 (32-bit sort length and input indices).
extern "C" size_t mrcnn_nms_general_workspace_bytes(int64_t n, int32_t dtype) {
    if (n < 1 || n > 0x7FFFFFBFll || (dtype != 0 && dtype != 1)) return 0;
    if (dtype == 0) {
        Layout<float> L;
        return layout<float>(n, &L) ? 0 : L.total;
    }
    Layout<double> L;
    return layout<double>(n, &L) ? 0 : L.total;
}

extern "C" int mrcnn_nms_general(const void* dets, int32_t dtype, int64_t n, int64_t row_stride, int64_t col_stride,
                                 float threshold, int64_t* keep_out, int64_t* count_out, void* workspace,
                                 size_t workspace_bytes, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(dets && keep_out && count_out && workspace, "nms_general: null pointer");
    MRCNN_REQUIRE(dtype == 0 || dtype == 1, "nms_general: dtype=%d (0 = float32, 1 = float64)", dtype);
    MRCNN_REQUIRE(n >= 1 && n <= 0x7FFFFFBFll, "nms_general: n=%lld outside [1, 2^31 - 65]", (long long)n);
    MRCNN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "nms_general: workspace must be 256-byte aligned");
    hipStream_t s = mrcnn::as_stream(stream);
    if (dtype == 0)
        return run<float>(static_cast<const float*>(dets), n, row_stride, col_stride, threshold, keep_out, count_out, workspace,
                          workspace_bytes, s);
    return run<double>(static_cast<const double*>(dets), n, row_stride, col_stride, threshold, keep_out, count_out, workspace,
                       workspace_bytes, s);
}
