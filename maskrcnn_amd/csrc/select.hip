// Selection steps of the two refine stages, so that the step has no library sort/top-k/gather in it:
//   topk_desc          rpn_refine's "sort scores descending, keep the first pre_nms_limit" (model.py:1345-1350) over
//                      the 261 888 anchor scores of an image
//   proposal_select    keep[:proposal_count] -> gather boxes -> normalise (model.py:1366-1374), zero-padded slots
//   detection_select   mrn_refine's tail: keep ∩ per-class NMS keep, top detection_max_instances by score, gathers
//                      (model.py:1475-1487), plus the normalised boxes the mask head pools (model.py:1188)
// All three are latency-bound integer/compare work on tiny outputs (top-k: two chip-wide passes over the scores,
// the other two one workgroup per image). Ordering is total and
// deterministic: descending score, ties by ascending index (the reference's ATen sort leaves ties unspecified).
#include "common.hpp"

namespace {

constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_WAVES = TOPK_THREADS / 64;
constexpr int SORT_CAP = 4096;  // elements the in-LDS bitonic sort handles (== mrcnn_nms_max_boxes())

// float -> uint32 whose unsigned order is the float order (NaN with the sign bit clear sorts above +inf, as in ATen)
__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Descending bitonic sort of n (a power of two) keys in LDS by all threads of the workgroup.
template <typename T>
__device__ __forceinline__ void bitonic_sort_desc(T* keys, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = tid; i < n; i += nthreads) {
                const int l = i ^ j;
                if (l > i) {
                    const T a = keys[i], b = keys[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) {
                        keys[i] = b;
                        keys[l] = a;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// Descending sort of one 64-bit key per thread across a 1024-thread workgroup: thread t ends up with the t-th
// largest. Compare-exchange distances below 64 stay inside a wave (shuffles, no barrier, no LDS); the ten steps at
// distance >= 64 go through a double-buffered LDS exchange with one barrier each. buf: 2 x 1024 keys of LDS.
__device__ __forceinline__ uint64_t sort1024_desc(uint64_t v, uint64_t* buf, int tid) {
    int flip = 0;
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            uint64_t other;
            if (j >= 64) {
                uint64_t* b = buf + flip * 1024;
                flip ^= 1;
                b[tid] = v;
                __syncthreads();
                other = b[tid ^ j];
            } else {
                const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), j, 64);
                const uint32_t hi = __shfl_xor(static_cast<uint32_t>(v >> 32), j, 64);
                other = (static_cast<uint64_t>(hi) << 32) | lo;
            }
            // in a descending pair ((tid & k) == 0) the lower index keeps the larger key
            const bool keep_max = ((tid & k) == 0) == ((tid & j) == 0);
            v = keep_max ? (v > other ? v : other) : (v < other ? v : other);
        }
    }
    return v;
}

// One histogram increment per lane with the wave's duplicates merged first: scores cluster (most anchors are
// background), so a plain LDS atomic per lane serialises up to 64 ways on a few bins. Up to four rounds of "the
// first pending lane's bin: everyone in it votes, one lane adds the vote count"; lanes still pending after that
// (spread-out data, where contention is not the problem) add individually. The votes see the lanes that are active
// at the call, so it may be called under divergence.
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active) {
    uint64_t pending = __ballot(active);
#pragma unroll 1
    for (int round = 0; round < 4 && pending; ++round) {
        const int leader = __ffsll(static_cast<long long>(pending)) - 1;
        const uint32_t lbin = __shfl(bin, leader, 64);
        const uint64_t same = __ballot(active && bin == lbin);
        if (static_cast<int>(threadIdx.x & 63) == leader) atomicAdd(&hist[lbin], static_cast<uint32_t>(__popcll(same)));
        if (bin == lbin) active = false;
        pending &= ~same;
    }
    if (active) atomicAdd(&hist[bin], 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// top-k: radix select of the k-th largest key (11 + 11 + 10 bits, LDS histograms), then an index-ordered compaction
// of the keys above it plus the lowest-index keys equal to it, then a bitonic sort of the k survivors.
// fn(value, index) for every element of a row, by all threads of the workgroup; 16 elements per thread and trip
// (four 16-byte loads in flight before the first use) when the row is 16-byte aligned.
template <typename F>
__device__ __forceinline__ void stream_row(const float* __restrict__ s, int64_t n, int tid, int nthreads, F&& fn) {
    const bool vec_ok = (reinterpret_cast<uintptr_t>(s) & 15) == 0;
    const int64_t nvec = vec_ok ? n / 4 : 0;  // float4s
    const float4* s4 = reinterpret_cast<const float4*>(s);
    int64_t v = tid;
    for (; v + 3 * static_cast<int64_t>(nthreads) < nvec; v += 4 * static_cast<int64_t>(nthreads)) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = s4[v + u * static_cast<int64_t>(nthreads)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = 4 * (v + u * static_cast<int64_t>(nthreads));
            fn(q[u].x, i); fn(q[u].y, i + 1); fn(q[u].z, i + 2); fn(q[u].w, i + 3);
        }
    }
    for (; v < nvec; v += nthreads) {
        const float4 q = s4[v];
        fn(q.x, 4 * v); fn(q.y, 4 * v + 1); fn(q.z, 4 * v + 2); fn(q.w, 4 * v + 3);
    }
    for (int64_t i = nvec * 4 + tid; i < n; i += nthreads) fn(s[i], i);
}

constexpr int TOPK_SPLIT = 8;                           // workgroups per image in the two chip-wide passes
constexpr int TOPK_NMAX = TOPK_SPLIT * TOPK_THREADS;    // per-thread maxima per image

struct TopkShared {
    uint32_t hist[2048];
    uint32_t wave_sum[TOPK_WAVES];
    uint32_t wave_gt[TOPK_WAVES], wave_eq[TOPK_WAVES];
    uint32_t prefix, remaining;  // selected high bits so far; how many of the keys matching them are still wanted
    uint64_t keys[SORT_CAP];
};

struct TopkWorkspace {  // device memory, per call
    uint32_t* ncand;      // [batch] candidates appended (zeroed by the first kernel)
    uint32_t* bound;      // [batch] lower bound of the k-th largest key
    uint32_t* maxima;     // [batch][TOPK_NMAX]
    uint64_t* cand;       // [batch][SORT_CAP] (key << 32 | ~index)
};

// Radix select (11 + 11 + 10 bits, LDS histograms) of the want-th largest of the keys `each` enumerates
// (each(f) calls f(key) for every key the calling thread owns; all TOPK_THREADS threads take part).
// → thr = that key, need_eq = how many keys equal to it belong to the `want` largest.
template <bool MERGE, typename Each>  // MERGE: wave-merged histogram adds (clustered keys) or plain LDS atomics
__device__ __forceinline__ void radix_select(TopkShared& sh, int tid, uint32_t want_total, Each&& each, uint32_t& thr,
                                             uint32_t& need_eq) {
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    if (tid == 0) {
        sh.prefix = 0;
        sh.remaining = want_total;
    }
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    uint32_t mask_hi = 0;  // bits already decided
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = 1 << bits[pass];
        for (int i = tid; i < 2048; i += TOPK_THREADS) sh.hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = sh.prefix;
        each([&](uint32_t key) {
            const uint32_t bin = (key >> shift) & (nb - 1);
            const bool in = (key & mask_hi) == prefix;
            if constexpr (MERGE) hist_add(sh.hist, bin, in);
            else if (in) atomicAdd(&sh.hist[bin], 1u);
        });
        __syncthreads();
        // suffix sums over the bins: thread t owns bins 2t, 2t+1 (bins >= nb are empty)
        const uint32_t h0 = sh.hist[2 * tid], h1 = sh.hist[2 * tid + 1];
        uint32_t v = h0 + h1;  // inclusive suffix scan of v over threads (higher thread = higher bins)
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_down(v, off, 64);
            if (lane + off < 64) v += o;
        }
        if (lane == 0) sh.wave_sum[wave] = v;
        __syncthreads();
        uint32_t above_waves = 0;
        for (int w = wave + 1; w < TOPK_WAVES; ++w) above_waves += sh.wave_sum[w];
        const uint32_t incl = v + above_waves;   // keys in bins >= 2*tid
        const uint32_t above1 = incl - h0 - h1;  // keys in bins > 2*tid + 1
        const uint32_t above0 = above1 + h1;     // keys in bins > 2*tid
        const uint32_t want = sh.remaining;
        __syncthreads();
        // the bin holding the want-th largest: above < want <= above + count. Exactly one (thread, bin) matches.
        if (above1 < want && want <= above1 + h1) {
            sh.prefix = prefix | (static_cast<uint32_t>(2 * tid + 1) << shift);
            sh.remaining = want - above1;
        } else if (above0 < want && want <= above0 + h0) {
            sh.prefix = prefix | (static_cast<uint32_t>(2 * tid) << shift);
            sh.remaining = want - above0;
        }
        mask_hi |= static_cast<uint32_t>(nb - 1) << shift;
        __syncthreads();
    }
    thr = sh.prefix;
    need_eq = sh.remaining;
}

// Sort the `count` (>= k) composites in sh.keys (descending key, ascending index) and write the first k.
__device__ __forceinline__ void emit_topk(TopkShared& sh, int tid, int count, int k, const float* __restrict__ s,
                                          float* __restrict__ top, int64_t* __restrict__ order) {
    int cp = 1;
    while (cp < count) cp <<= 1;
    for (int i = count + tid; i < cp; i += TOPK_THREADS) sh.keys[i] = 0;  // padding sorts last
    bitonic_sort_desc(sh.keys, cp, tid, TOPK_THREADS);
    for (int i = tid; i < k; i += TOPK_THREADS) {
        const uint32_t idx = 0xFFFFFFFFu - static_cast<uint32_t>(sh.keys[i]);
        top[i] = s[idx];
        order[i] = idx;
    }
}

// Exact top-k of one row by one workgroup, whatever the data: radix select of the k-th largest key over the whole
// row, then an index-ordered compaction of the keys above it plus the lowest-index keys equal to it. Slow (one CU
// streams the row five times): only used when the two-pass scheme below meets more than SORT_CAP candidates, i.e.
// thousands of exactly equal scores around the cut.
__device__ __forceinline__ void exact_topk_row(TopkShared& sh, int tid, const float* __restrict__ s, int64_t n, int k,
                                               float* __restrict__ top, int64_t* __restrict__ order) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t thr, need_eq;
    radix_select<true>(sh, tid, static_cast<uint32_t>(k), [&](auto&& f) {
        stream_row(s, n, tid, TOPK_THREADS, [&](float v, int64_t) { f(order_key(v)); });
    }, thr, need_eq);
    // ---- compaction in index order: wave w owns the contiguous range [w*per, (w+1)*per) ------------------------
    const int64_t per = (n + TOPK_WAVES - 1) / TOPK_WAVES;
    const int64_t lo = wave * per, hi = (lo + per < n) ? lo + per : n;
    uint32_t cgt = 0, ceq = 0;
#pragma unroll 8
    for (int64_t i = lo + lane; i < hi; i += 64) {
        const uint32_t key = order_key(s[i]);
        cgt += key > thr;
        ceq += key == thr;
    }
    for (int off = 32; off > 0; off >>= 1) {
        cgt += __shfl_down(cgt, off, 64);
        ceq += __shfl_down(ceq, off, 64);
    }
    if (lane == 0) {
        sh.wave_gt[wave] = cgt;
        sh.wave_eq[wave] = ceq;
    }
    __syncthreads();
    uint32_t gt_before = 0, eq_before = 0;
    for (int w = 0; w < wave; ++w) {
        gt_before += sh.wave_gt[w];
        eq_before += sh.wave_eq[w];
    }
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t base = lo; base < hi; base += 4 * 64) {
        float f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four loads in flight, then index order
            const int64_t i = base + u * 64 + lane;
            f[u] = i < hi ? s[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + u * 64 + lane;
            const uint32_t key = order_key(f[u]);
            const bool gt = i < hi && key > thr, eq = i < hi && key == thr;
            const uint64_t bgt = __ballot(gt), beq = __ballot(eq);
            if (bgt | beq) {
                const uint32_t eq_rank = eq_before + __popcll(beq & lt_mask);
                if (gt || (eq && eq_rank < need_eq)) {
                    const uint32_t eq_taken_before = eq_rank < need_eq ? eq_rank : need_eq;  // eq taken so far
                    const uint32_t slot = gt_before + __popcll(bgt & lt_mask) + eq_taken_before;
                    // descending (key, then ascending index): index stored complemented
                    sh.keys[slot] = (static_cast<uint64_t>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(i));
                }
                gt_before += __popcll(bgt);
                eq_before += __popcll(beq);
            }
        }
    }
    __syncthreads();
    emit_topk(sh, tid, k, k, s, top, order);
}

// The slice of a row that workgroup g of TOPK_SPLIT streams: element range [lo, hi), lo a multiple of 4.
__device__ __forceinline__ void row_slice(int64_t n, int g, int64_t& lo, int64_t& hi) {
    const int64_t per = (((n + TOPK_SPLIT - 1) / TOPK_SPLIT) + 3) & ~static_cast<int64_t>(3);
    lo = g * per;
    hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = hi = n;
}

// Four launches, each image spread over TOPK_SPLIT workgroups where the whole row is streamed:
//   maxima   every thread keeps the largest key it sees. These TOPK_NMAX values are distinct elements, so their k-th
//            largest is a lower bound of the row's k-th largest, and for data in no particular order only about
//            k + k^2/(2*TOPK_NMAX) elements reach it — no histogram over the 262k scores is ever built.
//   bound    one workgroup per image selects that k-th largest maximum (radix select over 8192 keys in registers).
//   collect  every element whose key reaches the bound is appended (unordered) to the image's candidate list.
//   finish   one workgroup per image: if needed trims the candidates to exactly k (radix select in LDS), sorts
//            them by (key, index) in registers and writes the result. More than SORT_CAP candidates (thousands of
//            equal scores around the cut) → exact_topk_row.
__global__ __launch_bounds__(TOPK_THREADS) void topk_maxima_kernel(const float* __restrict__ scores, int64_t n,
                                                                   const TopkWorkspace ws) {
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    int64_t lo, hi;
    row_slice(n, g, lo, hi);
    uint32_t m = 0;
    stream_row(scores + b * n + lo, hi - lo, tid, TOPK_THREADS, [&](float f, int64_t) {
        const uint32_t key = order_key(f);
        m = key > m ? key : m;
    });
    ws.maxima[static_cast<int64_t>(b) * TOPK_NMAX + g * TOPK_THREADS + tid] = m;
    if (g == 0 && tid == 0) ws.ncand[b] = 0;
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_bound_kernel(int k, const TopkWorkspace ws) {
    __shared__ TopkShared sh;
    const int tid = threadIdx.x, b = blockIdx.x;
    const uint32_t* maxima = ws.maxima + static_cast<int64_t>(b) * TOPK_NMAX;
    uint32_t mine[TOPK_SPLIT];
#pragma unroll
    for (int u = 0; u < TOPK_SPLIT; ++u) mine[u] = maxima[u * TOPK_THREADS + tid];
    uint32_t thr, need_eq;
    radix_select<false>(sh, tid, static_cast<uint32_t>(k), [&](auto&& f) {
#pragma unroll
        for (int u = 0; u < TOPK_SPLIT; ++u) f(mine[u]);
    }, thr, need_eq);
    if (tid == 0) ws.bound[b] = thr;
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_collect_kernel(const float* __restrict__ scores, int64_t n,
                                                                    const TopkWorkspace ws) {
    // candidates are gathered in LDS first and handed to the image's list with ONE global atomic per workgroup
    // (a returning global atomic per candidate costs ~80 ns each on one address: 88 us for ~1000 of them)
    __shared__ uint64_t local[SORT_CAP];
    __shared__ uint32_t nlocal, base;
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    int64_t lo, hi;
    row_slice(n, g, lo, hi);
    const uint32_t bound = ws.bound[b];
    if (tid == 0) nlocal = 0;
    __syncthreads();
    stream_row(scores + b * n + lo, hi - lo, tid, TOPK_THREADS, [&](float f, int64_t i) {
        const uint32_t key = order_key(f);
        if (key >= bound) {
            const uint32_t slot = atomicAdd(&nlocal, 1u);
            if (slot < SORT_CAP)
                local[slot] = (static_cast<uint64_t>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(lo + i));
        }
    });
    __syncthreads();
    const uint32_t mine = nlocal;  // may exceed SORT_CAP: the total then does too and the finish kernel falls back
    if (tid == 0) base = mine ? atomicAdd(ws.ncand + b, mine) : 0u;
    __syncthreads();
    uint64_t* cand = ws.cand + static_cast<int64_t>(b) * SORT_CAP;
    const uint32_t first = base, stored = mine < SORT_CAP ? mine : SORT_CAP;
    for (uint32_t i = tid; i < stored; i += TOPK_THREADS)
        if (first + i < SORT_CAP) cand[first + i] = local[i];
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_finish_kernel(const float* __restrict__ scores, int64_t n, int k,
                                                                   const TopkWorkspace ws, float* __restrict__ top,
                                                                   int64_t* __restrict__ order) {
    __shared__ TopkShared sh;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* s = scores + b * n;
    float* top_b = top + static_cast<int64_t>(b) * k;
    int64_t* order_b = order + static_cast<int64_t>(b) * k;
    const uint32_t ncand = ws.ncand[b];
    if (ncand > SORT_CAP) {
        exact_topk_row(sh, tid, s, n, k, top_b, order_b);
        return;
    }
    const uint64_t* cand = ws.cand + static_cast<int64_t>(b) * SORT_CAP;
    if (k > TOPK_THREADS) {  // more results than threads: sort all candidates in LDS
        for (uint32_t i = tid; i < ncand; i += TOPK_THREADS) sh.keys[i] = cand[i];
        __syncthreads();
        emit_topk(sh, tid, static_cast<int>(ncand), k, s, top_b, order_b);
        return;
    }
    uint64_t mine = 0;  // this thread's key for the register sort (0 = padding, sorts last)
    if (ncand <= TOPK_THREADS) {
        if (tid < static_cast<int>(ncand)) mine = cand[tid];
    } else {
        // trim to exactly k: the k-th largest key among the candidates, then everything above it plus the keys equal
        // to it. With more equal keys than wanted (ties across the cut) the lowest indices must win: sort them all.
        constexpr int PER = SORT_CAP / TOPK_THREADS;
        uint64_t c[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t i = u * TOPK_THREADS + tid;
            c[u] = i < ncand ? cand[i] : 0;
        }
        uint32_t thr, need_eq;
        radix_select<false>(sh, tid, static_cast<uint32_t>(k), [&](auto&& f) {
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (static_cast<uint32_t>(u * TOPK_THREADS + tid) < ncand) f(static_cast<uint32_t>(c[u] >> 32));
        }, thr, need_eq);
        // count the keys equal to thr (wave_gt[0] doubles as the counter, wave_eq[0] as the output cursor)
        if (tid == 0) sh.wave_gt[0] = sh.wave_eq[0] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (static_cast<uint32_t>(u * TOPK_THREADS + tid) < ncand && static_cast<uint32_t>(c[u] >> 32) == thr)
                atomicAdd(&sh.wave_gt[0], 1u);
        __syncthreads();
        if (sh.wave_gt[0] != need_eq) {  // uniform
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const uint32_t i = u * TOPK_THREADS + tid;
                if (i < ncand) sh.keys[i] = c[u];
            }
            __syncthreads();
            emit_topk(sh, tid, static_cast<int>(ncand), k, s, top_b, order_b);
            return;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (static_cast<uint32_t>(u * TOPK_THREADS + tid) < ncand && static_cast<uint32_t>(c[u] >> 32) >= thr)
                sh.keys[atomicAdd(&sh.wave_eq[0], 1u)] = c[u];  // exactly k of them
        __syncthreads();
        if (tid < k) mine = sh.keys[tid];
        __syncthreads();
    }
    const uint64_t v = sort1024_desc(mine, sh.keys, tid);
    if (tid < k) {
        const uint32_t idx = 0xFFFFFFFFu - static_cast<uint32_t>(v);
        top_b[tid] = s[idx];
        order_b[tid] = idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proposal_select_kernel(const float* __restrict__ dets,
                                                              const int64_t* __restrict__ keep,
                                                              const int32_t* __restrict__ keep_counts, int k, int p,
                                                              float norm_h, float norm_w, float* __restrict__ rois,
                                                              int32_t* __restrict__ counts) {
    const int b = blockIdx.y;
    const int cnt = min(keep_counts[b], p);  // keep[:proposal_count], model.py:1366
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[b] = cnt;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < cnt) {
        const int64_t idx = keep[static_cast<int64_t>(b) * k + i];
        const float* d = dets + (static_cast<int64_t>(b) * k + idx) * 5;
        r = make_float4(d[0] / norm_h, d[1] / norm_w, d[2] / norm_h, d[3] / norm_w);  // model.py:1371-1374
    }
    reinterpret_cast<float4*>(rois)[static_cast<int64_t>(b) * p + i] = r;
}

// ---------------------------------------------------------------------------------------------------------------
struct DetSelShared {
    uint64_t keys[SORT_CAP];
    uint32_t kept[SORT_CAP / 32];
};

__global__ __launch_bounds__(1024) void detection_select_kernel(
    const float* __restrict__ dets, const int32_t* __restrict__ nms_cls, const int64_t* __restrict__ class_ids,
    const int64_t* __restrict__ keep, const int32_t* __restrict__ keep_counts, int p, int d, float norm_h,
    float norm_w, int64_t* __restrict__ out_ids, float* __restrict__ out_scores, float* __restrict__ out_boxes,
    float* __restrict__ out_rois, int32_t* __restrict__ out_counts) {
    __shared__ DetSelShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* db = dets + static_cast<int64_t>(b) * p * 5;
    for (int i = tid; i < SORT_CAP / 32; i += 1024) sh.kept[i] = 0;
    __syncthreads();
    const int nk = keep_counts[b];
    for (int i = tid; i < nk; i += 1024) {
        const int64_t idx = keep[static_cast<int64_t>(b) * p + i];
        atomicOr(&sh.kept[idx >> 5], 1u << (idx & 31));
    }
    __syncthreads();
    int pp = 1;
    while (pp < p) pp <<= 1;
    // candidates: survived the class-aware NMS and carry a foreground class (model.py:1437-1443,1475);
    // key = (score, lowest index first); everything else sorts last with key 0
    auto key_of = [&](int i) -> uint64_t {
        if (i < p && ((sh.kept[i >> 5] >> (i & 31)) & 1u) && nms_cls[static_cast<int64_t>(b) * p + i] > 0)
            return (static_cast<uint64_t>(order_key(db[i * 5 + 4])) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(i));
        return 0;
    };
    auto emit = [&](int i, uint64_t e) {  // output slot i gets the i-th largest candidate (or zeros)
        const bool ok = e != 0;
        const int64_t o = static_cast<int64_t>(b) * d + i;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        float score = 0.f;
        int64_t id = 0;
        if (ok) {
            const uint32_t idx = 0xFFFFFFFFu - static_cast<uint32_t>(e);
            box = make_float4(db[idx * 5 + 0], db[idx * 5 + 1], db[idx * 5 + 2], db[idx * 5 + 3]);
            score = db[idx * 5 + 4];
            id = class_ids[static_cast<int64_t>(b) * p + idx];
        }
        out_ids[o] = id;
        out_scores[o] = score;
        reinterpret_cast<float4*>(out_boxes)[o] = box;
        reinterpret_cast<float4*>(out_rois)[o] =
            make_float4(box.x / norm_h, box.y / norm_w, box.z / norm_h, box.w / norm_w);
        return ok;
    };
    int cnt = 0;
    if (p <= 1024) {  // one key per thread: register sort
        const uint64_t v = sort1024_desc(key_of(tid), sh.keys, tid);
        if (tid < d) cnt += emit(tid, v);
    } else {
        for (int i = tid; i < pp; i += 1024) sh.keys[i] = key_of(i);
        bitonic_sort_desc(sh.keys, pp, tid, 1024);
        for (int i = tid; i < d; i += 1024) cnt += emit(i, sh.keys[i]);
    }
    // count of valid slots (they are a prefix of the sorted order)
    __syncthreads();
    uint32_t* total = sh.kept;
    if (tid == 0) total[0] = 0;
    __syncthreads();
    if (cnt) atomicAdd(&total[0], static_cast<uint32_t>(cnt));
    __syncthreads();
    if (tid == 0) out_counts[b] = static_cast<int32_t>(total[0]);
}

}  // namespace

static size_t topk_layout(int batch, TopkWorkspace* ws, unsigned char* base) {
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    size_t o = 0;
    const size_t counters = o; o += up(sizeof(uint32_t) * 2 * batch);  // ncand, bound
    const size_t maxima = o;   o += up(sizeof(uint32_t) * static_cast<size_t>(batch) * TOPK_NMAX);
    const size_t cand = o;     o += up(sizeof(uint64_t) * static_cast<size_t>(batch) * SORT_CAP);
    if (ws) {
        uint32_t* c = reinterpret_cast<uint32_t*>(base + counters);
        ws->ncand = c;
        ws->bound = c + batch;
        ws->maxima = reinterpret_cast<uint32_t*>(base + maxima);
        ws->cand = reinterpret_cast<uint64_t*>(base + cand);
    }
    return o;
}

extern "C" size_t mrcnn_topk_workspace_bytes(int32_t batch) {
    return batch >= 1 ? topk_layout(batch, nullptr, nullptr) : 0;
}

extern "C" int mrcnn_topk_desc_f32(const float* scores, int32_t batch, int64_t n, int32_t k, float* top_scores,
                                   int64_t* order, void* workspace, size_t workspace_bytes, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(scores && top_scores && order && workspace, "topk: null pointer");
    MRCNN_REQUIRE(batch >= 1 && batch <= 65535 && n >= 1 && n < (1LL << 32) - 1, "topk: batch=%d n=%lld", batch,
                  static_cast<long long>(n));
    MRCNN_REQUIRE(k >= 1 && k <= SORT_CAP && k <= n, "topk: k=%d must be in [1, min(n, %d)]", k, SORT_CAP);
    TopkWorkspace ws;
    const size_t need = topk_layout(batch, &ws, static_cast<unsigned char*>(workspace));
    MRCNN_REQUIRE(workspace_bytes >= need, "topk: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t s = mrcnn::as_stream(stream);
    hipLaunchKernelGGL(topk_maxima_kernel, dim3(TOPK_SPLIT, batch), dim3(TOPK_THREADS), 0, s, scores, n, ws);
    hipLaunchKernelGGL(topk_bound_kernel, dim3(batch), dim3(TOPK_THREADS), 0, s, k, ws);
    hipLaunchKernelGGL(topk_collect_kernel, dim3(TOPK_SPLIT, batch), dim3(TOPK_THREADS), 0, s, scores, n, ws);
    hipLaunchKernelGGL(topk_finish_kernel, dim3(batch), dim3(TOPK_THREADS), 0, s, scores, n, k, ws, top_scores, order);
    return mrcnn::check_launch("topk kernels");
}

extern "C" int mrcnn_proposal_select_f32(const float* dets, const int64_t* keep, const int32_t* keep_counts,
                                         int32_t batch, int32_t k, int32_t proposal_count, float image_height,
                                         float image_width, float* rois, int32_t* counts, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(dets && keep && keep_counts && rois && counts, "proposal_select: null pointer");
    MRCNN_REQUIRE(batch >= 1 && batch <= 65535 && k >= 1 && proposal_count >= 1 && proposal_count <= k,
                  "proposal_select: batch=%d k=%d proposal_count=%d (1 <= proposal_count <= k)", batch, k,
                  proposal_count);
    hipLaunchKernelGGL(proposal_select_kernel, dim3((proposal_count + 255) / 256, batch), dim3(256), 0,
                       mrcnn::as_stream(stream), dets, keep, keep_counts, k, proposal_count, image_height, image_width,
                       rois, counts);
    return mrcnn::check_launch("proposal_select_kernel");
}

extern "C" int mrcnn_detection_select_f32(const float* dets, const int32_t* nms_class_ids, const int64_t* class_ids,
                                          const int64_t* keep, const int32_t* keep_counts, int32_t batch,
                                          int32_t rois_per_image, int32_t max_instances, float image_height,
                                          float image_width, int64_t* out_class_ids, float* out_scores,
                                          float* out_boxes, float* out_rois, int32_t* out_counts,
                                          mrcnn_stream_t stream) {
    MRCNN_REQUIRE(dets && nms_class_ids && class_ids && keep && keep_counts && out_class_ids && out_scores &&
                      out_boxes && out_rois && out_counts, "detection_select: null pointer");
    MRCNN_REQUIRE(batch >= 1 && rois_per_image >= 1 && rois_per_image <= SORT_CAP,
                  "detection_select: batch=%d rois_per_image=%d (<= %d)", batch, rois_per_image, SORT_CAP);
    MRCNN_REQUIRE(max_instances >= 1 && max_instances <= rois_per_image,
                  "detection_select: max_instances=%d must be in [1, rois_per_image]", max_instances);
    hipLaunchKernelGGL(detection_select_kernel, dim3(batch), dim3(1024), 0, mrcnn::as_stream(stream), dets,
                       nms_class_ids, class_ids, keep, keep_counts, rois_per_image, max_instances, image_height,
                       image_width, out_class_ids, out_scores, out_boxes, out_rois, out_counts);
    return mrcnn::check_launch("detection_select_kernel");
}
