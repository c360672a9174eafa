// Greedy NMS for gfx950 — bit-exact with the reference CPU path
//   /root/reference/c++ext/maskrcnn/csrc/cpu/nms_cpu.cpp:11-70 (areas :26, sort :28, loop :42-68, `>=` :65,
//   ascending-index output :69) and nms.h:15-30.
//
// Design (not a port of the reference's nms_cuda.cu, which builds an N x N/64 mask in global memory,
// copies it to the host with a blocking hipMemcpy and scans it there):
//   * one workgroup per segment (image, or image x class set), everything LDS-resident, no global
//     scratch, no host round trip, graph-capturable;
//   * score order by an in-LDS bitonic sort of 64-bit keys (~score | index) — ties resolve to the
//     lower input index, NaN scores first (ATen's convention);
//   * suppression in 64-box chunks of the sorted order, wave64-shaped:
//       A. every wave ballots one row of the chunk's 64x64 IoU>=thr matrix per step (lane = column),
//       B. wave 0 resolves the chunk serially over the *alive* boxes only (readlane + scalar ops),
//       C. all threads test their own (register-resident) later boxes against the chunk's survivors,
//          broadcasting survivor boxes from LDS;
//   * ascending-index compaction with wave ballots + a workgroup prefix sum.
// IoU arithmetic is the reference's, op for op, compiled with FP contraction off: separately rounded
// (x2-x1+1)*(y2-y1+1), std::max/min NaN semantics via ternaries, correctly rounded IEEE division.
#pragma clang fp contract(off)

#include "common.hpp"

namespace {

using u64 = unsigned long long;
using u32 = unsigned int;

struct Box {
    float y1, x1, y2, x2, area;
};

// ovr(i, j) >= thr with i = the kept (higher-score) box, j = candidate; nms_cpu.cpp:54-65
__device__ __forceinline__ bool iou_ge(const Box& i, const Box& j, float thr) {
    const float xx1 = (i.x1 < j.x1) ? j.x1 : i.x1;  // std::max(ix1, x1[j])
    const float yy1 = (i.y1 < j.y1) ? j.y1 : i.y1;
    const float xx2 = (j.x2 < i.x2) ? j.x2 : i.x2;  // std::min(ix2, x2[j])
    const float yy2 = (j.y2 < i.y2) ? j.y2 : i.y2;
    float tw = xx2 - xx1;
    tw = tw + 1.0f;
    float th = yy2 - yy1;
    th = th + 1.0f;
    const float w = (0.0f < tw) ? tw : 0.0f;  // std::max(0, tw)
    const float h = (0.0f < th) ? th : 0.0f;
    const float inter = w * h;
    float uni = i.area + j.area;
    uni = uni - inter;
    const float ovr = inter / uni;  // correctly rounded (hipcc default for fp32 '/')
    return ovr >= thr;
}

// descending score, NaN first, -0 == +0, ties by ascending index  →  ascending u64 key
__device__ __forceinline__ u64 make_key(float score, u32 idx) {
    u32 u = __float_as_uint(score);
    if (score != score) u = 0x7FC00000u;  // any NaN → +qNaN (largest)
    if (score == 0.0f) u = 0u;            // -0 → +0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float → uint
    return (static_cast<u64>(~u) << 32) | idx;
}

template <int CAP, int T>
__global__ __launch_bounds__(T) void nms_kernel(const float* __restrict__ dets, int64_t n_max,
                                                int64_t seg_stride, int64_t row_stride,
                                                int64_t col_stride,
                                                const int32_t* __restrict__ seg_counts,
                                                const int32_t* __restrict__ class_ids, float thr,
                                                int64_t* __restrict__ keep_out,
                                                int32_t* __restrict__ counts_out) {
    constexpr int PER = CAP / T;  // sorted positions owned per thread (p = tid + k*T)
    constexpr int NW = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);             // [CAP]
    float* by1 = reinterpret_cast<float*>(keys + CAP);    // [CAP] each, sorted order
    float* bx1 = by1 + CAP;
    float* by2 = bx1 + CAP;
    float* bx2 = by2 + CAP;
    float* bar = bx2 + CAP;
    int* bcl = reinterpret_cast<int*>(bar + CAP);         // [CAP]
    u64* colmask = reinterpret_cast<u64*>(bcl + CAP);     // [64]
    u64* misc = colmask + 64;                             // [0] = chunk survivors
    int* wsum = reinterpret_cast<int*>(misc + 2);         // [NW + 1]
    unsigned char* sup = reinterpret_cast<unsigned char*>(wsum + 32);  // [CAP] sorted order
    unsigned char* keepf = sup + CAP;                                  // [CAP] input order

    const int seg = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int n = seg_counts ? seg_counts[seg] : static_cast<int>(n_max);
    n = n < 0 ? 0 : (n > n_max ? static_cast<int>(n_max) : n);
    const float* d = dets + static_cast<int64_t>(seg) * seg_stride;
    const int32_t* cls = class_ids ? class_ids + static_cast<int64_t>(seg) * n_max : nullptr;
    int64_t* keep = keep_out + static_cast<int64_t>(seg) * n_max;

    int np2 = 64;  // sort width: next power of two >= n (>= 64 keeps chunk logic uniform)
    while (np2 < n) np2 <<= 1;

    // ---- 1. keys ------------------------------------------------------------------------------
    for (int i = tid; i < np2; i += T) {
        keys[i] = (i < n) ? make_key(d[i * row_stride + 4 * col_stride], static_cast<u32>(i))
                          : ~0ull;
        keepf[i] = 0;
    }
    for (int i = np2 + tid; i < CAP; i += T) keepf[i] = 0;
    __syncthreads();

    // ---- 2. bitonic sort (ascending key == descending score, stable by index) -------------------
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const u64 a = keys[i], b = keys[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[i + j] = a;
                }
            }
            __syncthreads();
        }
    }

    // ---- 3. gather boxes into sorted order; areas as nms_cpu.cpp:26 -----------------------------
    Box mine[PER];
    int mycls[PER];
    bool dead[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int p = tid + k * T;
        Box b = {0.f, 0.f, 0.f, 0.f, 0.f};
        int c = 0;
        const bool valid = p < n;
        if (valid) {
            const int64_t i = static_cast<int64_t>(keys[p] & 0xFFFFFFFFu);
            const float* r = d + i * row_stride;
            b.y1 = r[0];
            b.x1 = r[col_stride];
            b.y2 = r[2 * col_stride];
            b.x2 = r[3 * col_stride];
            float w = b.x2 - b.x1;
            w = w + 1.0f;
            float h = b.y2 - b.y1;
            h = h + 1.0f;
            b.area = w * h;
            c = cls ? cls[i] : 0;
        }
        if (p < np2) {
            by1[p] = b.y1; bx1[p] = b.x1; by2[p] = b.y2; bx2[p] = b.x2; bar[p] = b.area;
            bcl[p] = c;
            sup[p] = valid ? 0 : 1;
        }
        mine[k] = b;
        mycls[k] = c;
        dead[k] = !valid;
    }
    __syncthreads();

    // ---- 4. chunked suppression ---------------------------------------------------------------
    for (int c0 = 0; c0 < n; c0 += 64) {
        // A. rows of the chunk's 64x64 matrix: wave w ballots rows w, w+NW, ...; lane = column
        {
            const int pj = c0 + lane;  // < np2 always (np2 multiple of 64)
            const Box bj = {by1[pj], bx1[pj], by2[pj], bx2[pj], bar[pj]};
            const int cj = bcl[pj];
            for (int i = wave; i < 64; i += NW) {
                const int pi = c0 + i;
                const Box bi = {by1[pi], bx1[pi], by2[pi], bx2[pi], bar[pi]};
                const bool hit = (lane > i) && (pj < n) && (bcl[pi] == cj) && iou_ge(bi, bj, thr);
                const u64 m = __ballot(hit);
                if (lane == 0) colmask[i] = m;
            }
        }
        __syncthreads();
        // B. serial resolve over alive boxes (wave 0)
        if (wave == 0) {
            const u64 cm = colmask[lane];
            const u32 cm_lo = static_cast<u32>(cm), cm_hi = static_cast<u32>(cm >> 32);
            u64 alive = __ballot(sup[c0 + lane] == 0);
            u64 kept = 0, rem = alive;
            while (rem) {
                const int i = __builtin_ctzll(rem);
                kept |= 1ull << i;
                // readlane returns int: go through u32 so the low word is not sign-extended
                const u32 m_hi = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(cm_hi), i));
                const u32 m_lo = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(cm_lo), i));
                const u64 m = (static_cast<u64>(m_hi) << 32) | m_lo;
                alive &= ~m;
                rem = alive & ~((2ull << i) - 1ull);  // i == 63 → 2<<63 == 0 → mask = ~(-1) = 0
            }
            sup[c0 + lane] = ((kept >> lane) & 1ull) ? 0 : 1;
            if (lane == 0) misc[0] = kept;
        }
        __syncthreads();
        // C. survivors of this chunk suppress every later box (owner threads, boxes in registers)
        const u64 kept = misc[0];
        if (kept != 0 && c0 + 64 < n) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int p = tid + k * T;
                if (p >= c0 + 64 && !dead[k]) {
                    u64 m = kept;
                    while (m) {
                        const int i = __builtin_ctzll(m);
                        m &= m - 1;
                        const int pi = c0 + i;
                        const Box bi = {by1[pi], bx1[pi], by2[pi], bx2[pi], bar[pi]};
                        if (bcl[pi] == mycls[k] && iou_ge(bi, mine[k], thr)) {
                            dead[k] = true;
                            sup[p] = 1;
                            break;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- 5. survivors by input index ------------------------------------------------------------
    for (int p = tid; p < n; p += T)
        if (!sup[p]) keepf[keys[p] & 0xFFFFFFFFu] = 1;
    __syncthreads();

    // ---- 6. ascending-index compaction: thread t owns indices [t*PER, t*PER+PER) -----------------
    int local = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) local += keepf[tid * PER + k];
    int incl = local;  // wave inclusive scan
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < NW; ++w) {
            const int v = wsum[w];
            wsum[w] = acc;
            acc += v;
        }
        wsum[NW] = acc;
    }
    __syncthreads();
    const int total = wsum[NW];
    int pos = wsum[wave] + incl - local;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid * PER + k;
        if (keepf[i]) keep[pos++] = i;
    }
    for (int64_t i = total + tid; i < n_max; i += T) keep[i] = -1;
    if (tid == 0) counts_out[seg] = total;
}

template <int CAP>
constexpr size_t nms_lds_bytes() {
    return sizeof(u64) * CAP + sizeof(float) * 5 * CAP + sizeof(int) * CAP + sizeof(u64) * 64 +
           sizeof(u64) * 2 + sizeof(int) * 32 + 2 * CAP;
}

template <int CAP, int T>
int launch(const float* dets, int32_t S, int64_t n_max, int64_t seg_stride, int64_t row_stride,
           int64_t col_stride, const int32_t* seg_counts, const int32_t* class_ids, float thr,
           int64_t* keep_out, int32_t* counts_out, hipStream_t stream) {
    constexpr size_t lds = nms_lds_bytes<CAP>();
    auto kern = nms_kernel<CAP, T>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess)
            return mrcnn::fail(MRCNN_ERR_LAUNCH, "nms: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(S), dim3(T), lds, stream, dets, n_max, seg_stride, row_stride,
                       col_stride, seg_counts, class_ids, thr, keep_out, counts_out);
    return mrcnn::check_launch("nms_kernel");
}

}  // namespace

extern "C" int64_t mrcnn_nms_max_boxes(void) { return 4096; }

extern "C" int mrcnn_nms_batched_f32(const float* dets, int32_t num_segments, int64_t n_max,
                                     int64_t seg_stride, int64_t row_stride, int64_t col_stride,
                                     const int32_t* seg_counts, const int32_t* class_ids,
                                     float threshold, int64_t* keep_out, int32_t* counts_out,
                                     mrcnn_stream_t stream) {
    MRCNN_REQUIRE(dets && keep_out && counts_out, "nms: null pointer");
    MRCNN_REQUIRE(num_segments >= 1, "nms: num_segments=%d must be >= 1", num_segments);
    MRCNN_REQUIRE(n_max >= 1 && n_max <= mrcnn_nms_max_boxes(),
                  "nms: n_max=%lld outside [1, %lld] (on-chip path)", (long long)n_max,
                  (long long)mrcnn_nms_max_boxes());
    hipStream_t s = mrcnn::as_stream(stream);
    if (n_max <= 256)
        return launch<256, 256>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                seg_counts, class_ids, threshold, keep_out, counts_out, s);
    if (n_max <= 1024)
        return launch<1024, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                  seg_counts, class_ids, threshold, keep_out, counts_out, s);
    if (n_max <= 2048)
        return launch<2048, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                  seg_counts, class_ids, threshold, keep_out, counts_out, s);
    return launch<4096, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                              seg_counts, class_ids, threshold, keep_out, counts_out, s);
}
