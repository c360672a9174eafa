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


// =====================================================================================================
// Path 2 (needs a caller-provided workspace): the O(N^2) pair tests leave the single workgroup and spread
// over the whole chip; only the inherently serial greedy scan stays on one wave per segment.
//   K1 nms_sort_kernel   1 workgroup / segment : LDS bitonic sort, sorted boxes + areas + classes + input
//                                                indices to the workspace
//   K2 nms_mask_kernel   1 wave / 64x64 tile   : lane = column box j, the tile's 64 row boxes broadcast by readlane: bit i of
//                                                the lane's word = "earlier box i of row block rb suppresses j" →
//                                                col[rb][j] (column-oriented, upper triangle of tiles; no ballot)
//   K3 nms_scan_kernel   1 workgroup / segment : the column words of the segment pulled into LDS (N <= 1024) or read
//                                                from L2; wave 0 walks the 64-box chunks in score order: lane j of chunk c
//                                                is removed iff col[rb][j] & kept[rb] != 0 for an earlier chunk rb (c word
//                                                reads per lane — round 5; rounds 1-4 OR-ed the 64 ROWS of every chunk's
//                                                survivors into a removed set: 64 reads per lane and chunk, 48 us per
//                                                8 x 1000 boxes), then resolves the chunk itself as a fixpoint of ballots on
//                                                the diagonal word; then ballot/popcount compaction to ascending input indices.
// Identical arithmetic (iou_ge) and identical visiting order → identical keep set to path 1 and the CPU path.
// =====================================================================================================
struct NmsWs {
    float4* box;    // [S][Np] sorted (y1,x1,y2,x2)
    float* area;    // [S][Np]
    int* cls;       // [S][Np]
    int* idx;       // [S][Np] input index, -1 for padding
    u64* col;       // [S][NB][Np]   column-oriented: word rb of box p = earlier boxes of chunk rb that suppress p (rb <= p / 64)
    int np, nb;
};

template <int CAP, int T>
__global__ __launch_bounds__(T) void nms_sort_kernel(const float* __restrict__ dets, int64_t n_max,
                                                     int64_t seg_stride, int64_t row_stride,
                                                     int64_t col_stride,
                                                     const int32_t* __restrict__ seg_counts,
                                                     const int32_t* __restrict__ class_ids, NmsWs ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);  // [CAP]
    const int seg = blockIdx.x, tid = threadIdx.x;
    int n = seg_counts ? seg_counts[seg] : static_cast<int>(n_max);
    n = n < 0 ? 0 : (n > n_max ? static_cast<int>(n_max) : n);
    const float* d = dets + static_cast<int64_t>(seg) * seg_stride;
    const int32_t* cls = class_ids ? class_ids + static_cast<int64_t>(seg) * n_max : nullptr;
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += T)
        keys[i] = (i < n) ? make_key(d[i * row_stride + 4 * col_stride], static_cast<u32>(i)) : ~0ull;
    __syncthreads();
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
    const int64_t base = static_cast<int64_t>(seg) * ws.np;
    for (int p = tid; p < ws.np; p += T) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float area = 0.f;
        int c = 0, idx = -1;
        if (p < n) {
            idx = static_cast<int>(keys[p] & 0xFFFFFFFFu);
            const float* r = d + static_cast<int64_t>(idx) * row_stride;
            b.x = r[0]; b.y = r[col_stride]; b.z = r[2 * col_stride]; b.w = r[3 * col_stride];
            float w = b.w - b.y;
            w = w + 1.0f;
            float h = b.z - b.x;
            h = h + 1.0f;
            area = w * h;  // nms_cpu.cpp:26
            c = cls ? cls[idx] : 0;
        }
        ws.box[base + p] = b;
        ws.area[base + p] = area;
        ws.cls[base + p] = c;
        ws.idx[base + p] = idx;
    }
}

// grid = (nb*(nb+1)/2 upper-triangle tiles, S), block = 64
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsWs ws, float thr) {
    // decode (rb <= cb) from the linear upper-triangle index
    int t = blockIdx.x, rb = 0;
    while (t >= ws.nb - rb) { t -= ws.nb - rb; ++rb; }
    const int cb = rb + t;
    const int seg = blockIdx.y, lane = threadIdx.x;
    const int64_t base = static_cast<int64_t>(seg) * ws.np;
    const int pj = cb * 64 + lane, pi0 = rb * 64;
    const float4 bj4 = ws.box[base + pj];
    const Box bj = {bj4.x, bj4.y, bj4.z, bj4.w, ws.area[base + pj]};
    const int cj = ws.cls[base + pj];
    const bool jvalid = ws.idx[base + pj] >= 0;
    // row boxes: lane i holds row i; broadcast with readlane
    const float4 bi4 = ws.box[base + pi0 + lane];
    const float ai = ws.area[base + pi0 + lane];
    const int ci = ws.cls[base + pi0 + lane];
    u64 col = 0;   // lane j: rows of this tile that suppress column box j
    for (int i = 0; i < 64; ++i) {
        Box bi;
        bi.y1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bi4.x), i));
        bi.x1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bi4.y), i));
        bi.y2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bi4.z), i));
        bi.x2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bi4.w), i));
        bi.area = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ai), i));
        const int cri = __builtin_amdgcn_readlane(ci, i);
        const bool hit = jvalid && (pj > pi0 + i) && (cri == cj) && iou_ge(bi, bj, thr);
        col |= hit ? (1ull << i) : 0ull;
    }
    ws.col[(static_cast<int64_t>(seg) * ws.nb + rb) * ws.np + pj] = col;
}

// grid = S, block = 256. LDS: the segment's column words (when they fit) + keep flags + the chunks' survivor words.
template <bool MASK_IN_LDS>
__global__ __launch_bounds__(256) void nms_scan_kernel(NmsWs ws, int64_t n_max,
                                                       const int32_t* __restrict__ seg_counts,
                                                       int64_t* __restrict__ keep_out,
                                                       int32_t* __restrict__ counts_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = ws.np, nb = ws.nb;
    int n = seg_counts ? seg_counts[seg] : static_cast<int>(n_max);
    n = n < 0 ? 0 : (n > n_max ? static_cast<int>(n_max) : n);
    const int64_t base = static_cast<int64_t>(seg) * np;
    const u64* gcol = ws.col + base * nb;                                       // [nb][np]
    u64* lcol = reinterpret_cast<u64*>(smem);                                   // [nb][np] if MASK_IN_LDS
    unsigned char* keepf = smem + (MASK_IN_LDS ? sizeof(u64) * np * nb : 0);    // [np] by input index
    u64* keptw = reinterpret_cast<u64*>(keepf + np);                            // [nb] survivors per chunk (+ 2 spare words)
    const int nchunks = (n + 63) >> 6;
    if (MASK_IN_LDS) {
        // K2 wrote the words rb <= (column chunk) of columns < 64 * ceil(n / 64): copy exactly those (the upper triangle)
        for (int rb = 0; rb < nchunks; ++rb)
            for (int p = rb * 64 + tid; p < nchunks * 64; p += 256) lcol[rb * np + p] = gcol[static_cast<int64_t>(rb) * np + p];
    }
    for (int i = tid; i < np; i += 256) keepf[i] = 0;
    __syncthreads();
    const u64* ck = MASK_IN_LDS ? lcol : gcol;
    if (wave == 0) {
        for (int c = 0; c < nchunks; ++c) {
            const int p = c * 64 + lane;
            // removed by a survivor of an EARLIER chunk? (keptw[rb]: written by lane 0 below, read back by the same wave — LDS
            // operations of one wave complete in order; one address for all lanes: a broadcast read)
            const u64 dc = ck[static_cast<int64_t>(c) * np + p];  // earlier boxes of this chunk that would suppress box p
            u64 rem = 0;
            int rb = 0;
            for (; rb + 8 <= c; rb += 8) {   // eight independent reads in flight (they come from L2 when the words are not in LDS)
                u64 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ck[static_cast<int64_t>(rb + u) * np + p];
#pragma unroll
                for (int u = 0; u < 8; ++u) rem |= v[u] & keptw[rb + u];
            }
            for (; rb < c; ++rb) rem |= ck[static_cast<int64_t>(rb) * np + p] & keptw[rb];
            const bool me_alive = p < n && rem == 0ull;
            // Greedy resolution of the chunk as a fixpoint: K[j] = alive[j] && no kept EARLIER box suppresses j.
            // Starting from K = alive, iteration t fixes (at least) the first t+1 positions, and a fixpoint
            // satisfies the greedy recurrence, whose solution is unique — typically 2-4 rounds of one AND +
            // one ballot instead of a 64-step serial walk.
            u64 kept = __ballot(me_alive);
            for (int round = 0; round < 65; ++round) {
                const u64 next = __ballot(me_alive && (dc & kept) == 0ull);
                if (next == kept) break;
                kept = next;
            }
            if (lane == 0) keptw[c] = kept;
        }
    }
    __syncthreads();
    // survivors → flags by input index
    for (int p = tid; p < n; p += 256)
        if ((keptw[p >> 6] >> (p & 63)) & 1ull) keepf[ws.idx[base + p]] = 1;
    __syncthreads();
    // ascending-index compaction by wave 0 (ballot + popcount), then -1 padding by everyone
    int64_t* keep = keep_out + static_cast<int64_t>(seg) * n_max;
    int* totalp = reinterpret_cast<int*>(keptw + nb);  // one of the two spare words after keptw
    if (wave == 0) {
        int basepos = 0;
        for (int w0 = 0; w0 < n; w0 += 64) {
            const int i = w0 + lane;
            const bool f = (i < n) && keepf[i];
            const u64 m = __ballot(f);
            if (f) keep[basepos + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
            basepos += __builtin_popcountll(m);
        }
        if (lane == 0) { *totalp = basepos; counts_out[seg] = basepos; }
    }
    __syncthreads();
    const int total = *totalp;
    for (int64_t i = total + tid; i < n_max; i += 256) keep[i] = -1;
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
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "nms")) return rc;
    hipLaunchKernelGGL(kern, dim3(S), dim3(T), lds, stream, dets, n_max, seg_stride, row_stride,
                       col_stride, seg_counts, class_ids, thr, keep_out, counts_out);
    return mrcnn::check_launch("nms_kernel");
}

template <int CAP, int T>
int launch_sort(const float* dets, int32_t S, int64_t n_max, int64_t seg_stride, int64_t row_stride,
                int64_t col_stride, const int32_t* seg_counts, const int32_t* class_ids, NmsWs ws,
                hipStream_t stream) {
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(nms_sort_kernel<CAP, T>), sizeof(u64) * CAP, "nms"))
        return rc;
    hipLaunchKernelGGL((nms_sort_kernel<CAP, T>), dim3(S), dim3(T), sizeof(u64) * CAP, stream, dets, n_max,
                       seg_stride, row_stride, col_stride, seg_counts, class_ids, ws);
    return mrcnn::check_launch("nms_sort_kernel");
}

size_t ws_layout(int32_t S, int64_t n_max, void* base, NmsWs* ws) {
    const int np = static_cast<int>((n_max + 63) / 64 * 64), nb = np / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_box = take(sizeof(float4) * S * np), o_area = take(sizeof(float) * S * np);
    const size_t o_cls = take(sizeof(int) * S * np), o_idx = take(sizeof(int) * S * np);
    const size_t o_col = take(sizeof(u64) * S * np * nb);
    if (ws) {
        unsigned char* b = static_cast<unsigned char*>(base);
        ws->box = reinterpret_cast<float4*>(b + o_box);
        ws->area = reinterpret_cast<float*>(b + o_area);
        ws->cls = reinterpret_cast<int*>(b + o_cls);
        ws->idx = reinterpret_cast<int*>(b + o_idx);
        ws->col = reinterpret_cast<u64*>(b + o_col);
        ws->np = np;
        ws->nb = nb;
    }
    return off;
}

}  // namespace

// With a workspace: 16384 boxes per segment (256 chunks of 64: the column words live in the workspace; the sort holds
// 8-byte keys in LDS: 128 KB). Without: the single-launch path, 4096.
constexpr int64_t NMS_MAX_WS = 16384, NMS_MAX_LDS = 4096;
extern "C" int64_t mrcnn_nms_max_boxes(void) { return NMS_MAX_WS; }

extern "C" size_t mrcnn_nms_workspace_bytes(int32_t num_segments, int64_t n_max) {
    if (num_segments < 1 || n_max < 1 || n_max > NMS_MAX_WS) return 0;
    return ws_layout(num_segments, n_max, nullptr, nullptr);
}

extern "C" int mrcnn_nms_batched_f32(const float* dets, int32_t num_segments, int64_t n_max,
                                     int64_t seg_stride, int64_t row_stride, int64_t col_stride,
                                     const int32_t* seg_counts, const int32_t* class_ids,
                                     float threshold, int64_t* keep_out, int32_t* counts_out,
                                     void* workspace, size_t workspace_bytes, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(dets && keep_out && counts_out, "nms: null pointer");
    MRCNN_REQUIRE(num_segments >= 1, "nms: num_segments=%d must be >= 1", num_segments);
    MRCNN_REQUIRE(n_max >= 1 && n_max <= ((workspace && n_max > 128) ? NMS_MAX_WS : NMS_MAX_LDS),
                  "nms: n_max=%lld outside [1, %lld] (%lld with a workspace)", (long long)n_max, (long long)NMS_MAX_LDS,
                  (long long)NMS_MAX_WS);
    hipStream_t s = mrcnn::as_stream(stream);
    if (workspace && n_max > 128) {
        // ---- path 2: sort → pair mask over the whole chip → serial scan ------------------------------
        MRCNN_REQUIRE(workspace_bytes >= mrcnn_nms_workspace_bytes(num_segments, n_max),
                      "nms: workspace too small (%zu < %zu bytes)", workspace_bytes,
                      mrcnn_nms_workspace_bytes(num_segments, n_max));
        MRCNN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "nms: workspace must be 16-byte aligned");
        NmsWs ws;
        ws_layout(num_segments, n_max, workspace, &ws);
        int rc;
        if (n_max <= 1024)
            rc = launch_sort<1024, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                         seg_counts, class_ids, ws, s);
        else if (n_max <= 2048)
            rc = launch_sort<2048, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                         seg_counts, class_ids, ws, s);
        else if (n_max <= 4096)
            rc = launch_sort<4096, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                         seg_counts, class_ids, ws, s);
        else if (n_max <= 8192)
            rc = launch_sort<8192, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                         seg_counts, class_ids, ws, s);
        else
            rc = launch_sort<16384, 1024>(dets, num_segments, n_max, seg_stride, row_stride, col_stride,
                                          seg_counts, class_ids, ws, s);
        if (rc) return rc;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(ws.nb * (ws.nb + 1) / 2, num_segments), dim3(64), 0, s, ws,
                           threshold);
        if ((rc = mrcnn::check_launch("nms_mask_kernel"))) return rc;
        const size_t tail = (ws.np + 7) / 8 * 8 + sizeof(u64) * (ws.nb + 2);
        const size_t lds_full = sizeof(u64) * ws.np * ws.nb + tail;
        if (lds_full <= 150 * 1024) {
            auto k = nms_scan_kernel<true>;
            if ((rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds_full, "nms"))) return rc;
            hipLaunchKernelGGL(k, dim3(num_segments), dim3(256), lds_full, s, ws, n_max, seg_counts, keep_out,
                               counts_out);
        } else {
            hipLaunchKernelGGL(nms_scan_kernel<false>, dim3(num_segments), dim3(256), tail, s, ws, n_max,
                               seg_counts, keep_out, counts_out);
        }
        return mrcnn::check_launch("nms_scan_kernel");
    }
    // ---- path 1: one LDS-resident workgroup per segment, no workspace --------------------------------
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
