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

#include <algorithm>
#include <cstdlib>

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

// One (box, channel slab) on the whole workgroup, any crop size: per-position tap records in LDS, one thread per output
// scalar, 4 scalar gathers each. `smem` holds Sample[ch + cw] (already filled by the caller) followed by room for
// Tap[plane] when plane <= 1024.
__device__ __forceinline__ void crop_slab_generic(
    unsigned char* smem, const float* __restrict__ image, int depth, int H, int W, int b, int b_in, bool bad,
    float extrap, int ch, int cw, int c0, int c1, float* __restrict__ crops) {
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + ch;
    Tap* taps = reinterpret_cast<Tap*>(sx + cw);  // [ch*cw] when it fits (use_table), else unused
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

// The general entry: any crop size, any W, any alignment. grid = (num_boxes, channel slabs), block = 256.
__global__ __launch_bounds__(256) void crop_forward_nchw(
    const float* __restrict__ image, int batch, int depth, int H, int W,
    const float* __restrict__ boxes, const int* __restrict__ box_index, float extrap, int ch, int cw,
    int slab, float* __restrict__ crops) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + ch;
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
    crop_slab_generic(smem, image, depth, H, W, b, b_in, bad, extrap, ch, cw, c0, c1, crops);
}

// ------------------------------------------------------------------------------------------------
// NCHW forward, staged (the fast path: plane = ch*cw a multiple of 4 and <= 256 — 14x14 is BASELINE configs[1] —, ch + cw <= 64,
// W % 4 == 0, 16-byte aligned tensors).
// What the counters say about the gather kernel above on configs[1] (profiles/r03_crop_counters.json): it pulls 186 MB over
// the fabric for a 67 MB map — all slabs of a box run on the XCD (box % 8), so a cache line that several boxes touch is
// fetched by several L2s, 2.8x on average — and 186 + 51 MB in 40 us IS the ~6 TB/s the fabric delivers; on the small
// levels, where nothing is fetched, it still takes 27 us: every 256-thread workgroup builds a tap table for 8 outputs per
// thread, and a thread has one output (4 scalar taps) in flight at a time.
// Here a WAVE owns (box, a run of channels) and moves each channel's FOOTPRINT — the rows ymin..ymax x the 16-byte-aligned
// column segments xa..xmax the box's samples touch, S = rows x segments "slots" — into LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: one 16-byte segment per lane, no registers), G channels per group, through a per-wave ring
// of 4 x 2 KB (2 x 4 KB for footprints over 128 slots) so that up to three groups are in flight: no workgroup barrier
// anywhere. The LDS image of a channel is [rows][4 * segments] floats in slot order, which is the order the DMA writes.
// A lane owns 4 consecutive crop positions for EVERY channel: its tap offsets and lerp weights are registers, the taps of a
// position are two ds_read2_b32 (the (lo, lo+1) column pair of the top and of the bottom row), and the 4 results leave as
// one 16-byte store — a channel's 196 outputs are one 784-byte run. Arithmetic and its order are those of bilerp() above.
// Workgroups are numbered so that an XCD (workgroup id % 8: round-robin dispatch, speed only) works on its OWN channel slabs
// for all boxes: a map line is then fetched by one L2 (or by 8 / slabs of them when there are fewer than 8 slabs).
// Boxes whose footprint exceeds half the ring (S > 256 slots: larger than ~28 x 28 pixels on P2) take the gather path in the
// same launch.
// ------------------------------------------------------------------------------------------------
constexpr int CS_RING_SLOTS = 512;                   // 16-byte slots in a wave's ring (a multiple of 256; MRCNN_CROP_RING)
constexpr int CS_KMAX = 4;                           // most LDS-DMA wave-instructions per group (a 256-slot buffer)
typedef float cs_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char cs_lds_u8;

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: the largest immediate of a fixed set that does not exceed n (waiting for
// more than asked is always safe)
__device__ __forceinline__ void cs_wait_vmcnt_at_most(int n) {
    if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Small wave-uniform quotients without the ~35-instruction integer division sequence: a < 1024, 1 <= b <= a: the float
// quotient (a + 0.5) / b is at least 0.5 / b >= 1/2048 away from an integer, the 1-ulp reciprocal's error is below 1e-4.
__device__ __forceinline__ int cs_div_small(int a, int b) {
    return static_cast<int>((static_cast<float>(a) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(b)));
}

// map_mode 0: grid (box, slab). 1: 1-D grid, slabs % 8 == 0: XCD x takes slabs x, x + 8, ... for every box. 2: 1-D grid,
// 8 % slabs == 0: XCD x takes slab x % slabs for the boxes with box % (8 / slabs) == x / slabs.
// A workgroup is ONE wave (a slab = cpw channels of one box): nothing is shared between waves, a wave's slot on the CU is
// free again the moment it ends, and 8.2 KB of LDS per wave leave the register file (5 waves per SIMD) as the occupancy limit.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 5))) void crop_forward_nchw_staged(
    const float* __restrict__ image, int batch, int depth, int H, int W,
    const float* __restrict__ boxes, const int* __restrict__ box_index, int num_boxes, float extrap, int ch, int cw,
    int cpw, int slabs, int map_mode, int ring_slots, float* __restrict__ crops) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef MRCNN_CROP_STAMPS   // tuning build (tools/crop_stamp.py): s_memtime at the phase boundaries, written over the crop
    unsigned long long stamp[8];
    int nstamp = 0;
#define CS_STAMP() do { if (nstamp < 8) stamp[nstamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define CS_STAMP() do {} while (0)
#endif
    CS_STAMP();
    int b, slab;
    if (map_mode == 0) {
        b = blockIdx.x;
        slab = blockIdx.y;
    } else {
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
        if (map_mode == 1) {
            const int sl = q / num_boxes;
            b = q - sl * num_boxes;
            slab = sl * 8 + x;
        } else {
            const int rep = 8 / slabs;          // XCDs per slab
            slab = x % slabs;
            b = q * rep + x / slabs;
        }
        if (b >= num_boxes || slab >= slabs) return;
    }
    const int lane = threadIdx.x;
    const int c_lo = slab * cpw;
    const int nchan = min(depth, c_lo + cpw) - c_lo;
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
    const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    const int b_in = box_index[b];
    const bool bad = (b_in < 0 || b_in >= batch);
    const int plane = ch * cw;
    // The box's samples (lane t: row sample t and column sample t; ch + cw <= 64), with a copy in the LAST KB of the ring for
    // the per-position look-ups — the part of the ring no DMA touches before those are done. Samples advance linearly with
    // t, so the ones inside the image are a contiguous run of t: the footprint's bounds are at its two ends.
    Sample* tab = reinterpret_cast<Sample*>(smem + (ring_slots - 64) * 16);
    const Sample my_y = make_sample(y1, y2, H, ch, min(lane, ch - 1));
    const Sample my_x = make_sample(x1, x2, W, cw, min(lane, cw - 1));
    if (lane < ch) tab[lane] = my_y;
    if (lane < cw) tab[ch + lane] = my_x;
    const unsigned long long in_y = __ballot(lane < ch && my_y.inside), in_x = __ballot(lane < cw && my_x.inside);
    CS_STAMP();
    const bool any = !bad && in_y != 0 && in_x != 0;
    int ymin = 0, ymax = 0, xmin = 0, xmax = 0;
    if (any) {
        const int yf = __builtin_ctzll(in_y), yl_ = 63 - __builtin_clzll(in_y);
        const int xf = __builtin_ctzll(in_x), xl_ = 63 - __builtin_clzll(in_x);
        ymin = min(__builtin_amdgcn_readlane(my_y.lo, yf), __builtin_amdgcn_readlane(my_y.lo, yl_));
        ymax = max(__builtin_amdgcn_readlane(my_y.hi, yf), __builtin_amdgcn_readlane(my_y.hi, yl_));
        xmin = min(__builtin_amdgcn_readlane(my_x.lo, xf), __builtin_amdgcn_readlane(my_x.lo, xl_));
        xmax = max(__builtin_amdgcn_readlane(my_x.hi, xf), __builtin_amdgcn_readlane(my_x.hi, xl_));
    }
    const int xa = xmin & ~3;
    const int nseg = ((xmax - xa) >> 2) + 1;
    const int nrows = ymax - ymin + 1;
    const int S = nrows * nseg;
    if (S > 256) {   // nothing has been staged yet: this (box, slab) takes the gather path
        Sample* sy = reinterpret_cast<Sample*>(smem);   // which wants the sample table at the start of LDS
        if (lane < ch) sy[lane] = my_y;
        if (lane < cw) sy[ch + lane] = my_x;
        __syncthreads();
        crop_slab_generic(smem, image, depth, H, W, b, b_in, bad, extrap, ch, cw, c_lo, c_lo + nchan, crops);
        return;
    }
    const int p0 = lane * 4;
    const bool active = p0 < plane;
    float* out = crops + (static_cast<int64_t>(b) * depth + c_lo) * plane + p0;
    if (!any) {   // every sample of the box is outside the image (or box_index is bad)
        if (active)
            for (int c = 0; c < nchan; ++c)
                *reinterpret_cast<float4*>(out + static_cast<int64_t>(c) * plane) = make_float4(extrap, extrap, extrap, extrap);
        return;
    }
    // ring geometry: NB buffers of BS slots; G channels and K LDS-DMA instructions per group; D groups in flight ahead of
    // the one being interpolated
    const int BS = S <= 128 ? 128 : 256;
    const int NB = ring_slots / BS;
    const int G = min(min(8, cs_div_small(BS, S)), nchan);
    const int K = (G * S + 63) >> 6;
    const int D = NB - 1;
    const int ngroups = cs_div_small(nchan + G - 1, G);
    const unsigned chan_bytes = static_cast<unsigned>(H) * static_cast<unsigned>(W) * 4u;
    const unsigned chan_lds = static_cast<unsigned>(S) * 16u;
    // the lane's source offset for each DMA instruction of a group, relative to the group's first channel.
    // slot -> (channel, row, segment) by float reciprocal (1 ulp): slot < 256, so the quotient's distance from an integer
    // (>= 1/512) exceeds the rounding error (< 1e-4)
    const float rS = __builtin_amdgcn_rcpf(static_cast<float>(S)), rseg = __builtin_amdgcn_rcpf(static_cast<float>(nseg));
    unsigned voff[CS_KMAX];
#pragma unroll
    for (int k = 0; k < CS_KMAX; ++k) {
        const int slot = lane + 64 * k;
        const int cg = static_cast<int>((static_cast<float>(slot) + 0.5f) * rS);
        const int t = slot - cg * S;
        const int r = static_cast<int>((static_cast<float>(t) + 0.5f) * rseg);
        const int sg = t - r * nseg;
        const bool ok = cg < G;
        voff[k] = (ok ? static_cast<unsigned>(cg) * chan_bytes : 0u) +
                  static_cast<unsigned>(((ymin + (ok ? r : 0)) * W + xa + (ok ? 4 * sg : 0)) * 4);
    }
    const float* img = image + static_cast<int64_t>(b_in) * depth * H * W;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(img), 0, static_cast<unsigned>(depth) * chan_bytes, 0x00020000);
    cs_lds_u8* ring = (cs_lds_u8*)smem;
    const unsigned ring_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring));
    // a group always moves G channels; the last group of the image's last channels starts early enough to stay inside the
    // image (first_of) and the interpolation skips the channels in front (c + shift)
    auto first_of = [&](int g) { return min(c_lo + g * G, depth - G); };
    auto dma = [&](int g, int buf) {
        const int soff = static_cast<int>(static_cast<unsigned>(first_of(g)) * chan_bytes);
        cs_lds_u8* dst = ring + buf * (BS * 16);
#pragma unroll
        for (int k = 0; k < CS_KMAX; ++k)
            if (k < K)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + k * 1024, 16, static_cast<int>(voff[k]), soff, 0, 0);
    };
    CS_STAMP();
    for (int g = 0; g < min(D, ngroups); ++g) dma(g, g);   // buffers 0 .. NB - 2: the sample table (in the last one) is still intact
    CS_STAMP();
    // while those fly: the lane's 4 crop positions (p < 256, cw < 64: the float quotient is exact enough to truncate)
    unsigned top[4], bot[4];
    float xl[4], yl[4];
    bool selx[4];
    unsigned ins[4], outside[4];   // all-ones / 0 for a position inside the image; 0 / the extrapolation value's bits
    const float rcw = __builtin_amdgcn_rcpf(static_cast<float>(cw));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = min(p0 + i, plane - 1);
        const int y = static_cast<int>((static_cast<float>(p) + 0.5f) * rcw), x = p - y * cw;
        const Sample Y = tab[y], X = tab[ch + x];
        const bool in = Y.inside && X.inside;
        ins[i] = in ? 0xFFFFFFFFu : 0u;
        outside[i] = in ? 0u : __float_as_uint(extrap);
        top[i] = in ? static_cast<unsigned>(((Y.lo - ymin) * nseg * 4 + (X.lo - xa)) * 4) : 0u;
        bot[i] = in ? top[i] + static_cast<unsigned>((Y.hi - Y.lo) * nseg * 16) : 0u;
        selx[i] = X.hi != X.lo;
        xl[i] = X.lerp;
        yl[i] = Y.lerp;
    }
    // the look-ups above must have returned before a DMA may overwrite the table
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int buf = 0, buf_ahead = D;   // g % NB, (g + D) % NB
    for (int g = 0; g < ngroups; ++g) {
        // buffer (g + D) % NB is the one group g - 1 was interpolated from: every read of it has been waited for
        if (g + D < ngroups) dma(g + D, buf_ahead);
        // in issue order behind group g's pieces: the pieces of the groups ahead of it and the stores (G each) of the groups
        // interpolated since it was issued — at most that many operations may still be pending
        const int ahead = min(D, ngroups - 1 - g);
        cs_wait_vmcnt_at_most(ahead * K + min(D, g) * G);
        if (g == 0) CS_STAMP();
        const int gc = min(G, nchan - g * G);
        const int shift = c_lo + g * G - first_of(g);
        const unsigned base = ring_addr + static_cast<unsigned>(buf * (BS * 16)) + static_cast<unsigned>(shift) * chan_lds;
        buf = buf + 1 == NB ? 0 : buf + 1;
        buf_ahead = buf_ahead + 1 == NB ? 0 : buf_ahead + 1;
        float* o = out + static_cast<int64_t>(g) * G * plane;
        // The taps of two channels are in flight together (LDS reads return in order: a counted lgkmcnt retires one channel's
        // eight reads; the counter has 4 bits). The reads are asm: the compiler orders an LDS read it knows of behind EVERY
        // LDS-DMA in flight (vmcnt(0)), which would drain the prefetch. And because it does not know that an asm read's
        // result is pending, a read and its wait sit in ONE straight-line block: a result carried over a loop edge or a
        // branch merge may be copied to another register there — before the data has arrived.
        struct Taps { cs_f32x2 t[4], u[4]; };
        auto read = [&](int c, Taps& r) {
            const unsigned cb = base + static_cast<unsigned>(c) * chan_lds;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r.t[i]) : "v"(cb + top[i]));
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r.u[i]) : "v"(cb + bot[i]));
            }
        };
#define CS_WAIT(n, r)                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r.t[0]), "+v"(r.t[1]), "+v"(r.t[2]), "+v"(r.t[3]), "+v"(r.u[0]), \
                     "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]))
        auto finish = [&](int c, const Taps& r) {
            unsigned v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tl = r.t[i].x, tr = selx[i] ? r.t[i].y : tl;
                const float bl = r.u[i].x, br = selx[i] ? r.u[i].y : bl;
                // positions outside the image computed something from slot 0: keep the extrapolation value's bits instead
                v[i] = (__float_as_uint(bilerp(tl, tr, bl, br, xl[i], yl[i])) & ins[i]) | outside[i];
            }
            if (active)
                *reinterpret_cast<uint4*>(o + static_cast<int64_t>(c) * plane) = make_uint4(v[0], v[1], v[2], v[3]);
        };
        int c = 0;
        for (; c + 2 <= gc; c += 2) {
            Taps r0, r1;
            read(c, r0); read(c + 1, r1);
            CS_WAIT(8, r0); finish(c, r0);
            CS_WAIT(0, r1); finish(c + 1, r1);
        }
        if (c < gc) {
            Taps r0;
            read(c, r0);
            CS_WAIT(0, r0); finish(c, r0);
        }
#undef CS_WAIT
        if (g == 0) CS_STAMP();
    }
    CS_STAMP();
#ifdef MRCNN_CROP_STAMPS
    if (slab == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(crops + static_cast<int64_t>(b) * depth * plane);
            for (int i = 0; i < nstamp; ++i) d[i] = stamp[i];
            d[8] = (static_cast<unsigned long long>(S) << 32) | static_cast<unsigned>(G * 256 + K);
        }
    }
#endif
#undef CS_STAMP
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
    float* __restrict__ out, int* __restrict__ levels_out, int out_kblocked, int64_t out_pixels,
    const int* __restrict__ roi_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample* sy = reinterpret_cast<Sample*>(smem);
    Sample* sx = sy + pool;
    const int r = blockIdx.x;
    if (roi_counts) {   // slot r of its image holds no RoI (model.py:1366-1374: the reference's rois tensor ends there): no output
        const int img = r / rois_per_image;
        if (r - img * rois_per_image >= roi_counts[img]) return;
    }
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
            if (out_kblocked == 1)  // [depth/8][num_rois*pool*pool][8]: what the Winograd kernel (mask head conv1) reads
                *reinterpret_cast<float4*>(out + ((c >> 3) * out_pixels + static_cast<int64_t>(r) * points + pt) * 8 + (c & 7)) = v;
            else if (out_kblocked == 2) {  // fp16 NHWC (the "f16" mode's heads): the rounding their first conv would apply
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 hv = {static_cast<_Float16>(v.x), static_cast<_Float16>(v.y), static_cast<_Float16>(v.z),
                               static_cast<_Float16>(v.w)};
                *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(out) + (static_cast<int64_t>(r) * points + pt) * depth + c) = hv;
            } else
                *reinterpret_cast<float4*>(o + static_cast<int64_t>(pt) * depth + c) = v;
        }
    }
}

// (Round 5: a MAP-stationary kernel — a workgroup owns a band of 16 map rows x 4 channels streamed into LDS as whole rows, bins
// the boxes' crop rows into the band and writes 14-float runs — was built behind MRCNN_CROP_BAND, bit-identical to the kernels
// here, and lost the hipGraph-timed A/B on BASELINE configs[1] by 4x: 118 us against 30.3 on P2, 98 / 145 / 242 against 21.7 /
// 17.1 / 16.3 on P3 - P5 (profiles/r05_crop_band_probe.jsonl). Removed; the box-stationary kernels stay.)
// MRCNN_CROP_STAGED=0 keeps every call on the gather kernel (tuning / A-B measurements; results are identical)
// The tuning switches of the launch path, read ONCE per process (a getenv walks the whole environment: not per launch).
struct CropTuning {
    bool staged;      // MRCNN_CROP_STAGED=0: gather kernel only
    int cpw;          // MRCNN_CROP_CPW=n: channels per wave (0 = automatic)
    bool linear_map;  // MRCNN_CROP_MAP=0: plain (box, slab) grid
    int ring_slots;   // MRCNN_CROP_RING=n: DMA ring size (0 = default)
    CropTuning() {
        const char* e = getenv("MRCNN_CROP_STAGED");
        staged = !(e && e[0] == '0');
        e = mrcnn::tuning_env("MRCNN_CROP_CPW");
        cpw = (e && atoi(e) > 0) ? atoi(e) : 0;
        e = mrcnn::tuning_env("MRCNN_CROP_MAP");
        linear_map = e && atoi(e) == 0;
        e = mrcnn::tuning_env("MRCNN_CROP_RING");
        ring_slots = (e && atoi(e) >= 512 && atoi(e) % 256 == 0 && atoi(e) <= 4096) ? atoi(e) : 0;
    }
};
const CropTuning& crop_tuning() {
    static const CropTuning t;
    return t;
}
bool crop_staged_enabled() { return crop_tuning().staged; }

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
    hipStream_t s = mrcnn::as_stream(stream);
    // the staged kernel: 4 positions per lane, 16-byte segments and stores, 32-bit byte offsets inside one image
    const bool staged = plane % 4 == 0 && plane <= 256 && crop_height + crop_width <= 64 && width % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(image) & 15) == 0 && (reinterpret_cast<uintptr_t>(crops) & 15) == 0 &&
                        static_cast<uint64_t>(depth) * height * width * 4 < (1ull << 32) && crop_staged_enabled();
    if (staged) {
        // channels per wave (= per workgroup): 16 — 8 on large maps, where the footprints are big, a channel is its own DMA
        // group and the waves of the largest boxes would otherwise run long after the others (measured on 256 boxes x 256
        // channels: P2 39 -> 30.5 us, P3 21.9 vs 21.1) —, fewer when that leaves the chip short of waves
        int cpw = static_cast<int64_t>(height) * width >= 256 * 256 ? 8 : 16;
        while (cpw > 2 && static_cast<int64_t>(num_boxes) * ((depth + cpw - 1) / cpw) < 4096) cpw >>= 1;
        if (crop_tuning().cpw) cpw = crop_tuning().cpw;
        const int slabs = (depth + cpw - 1) / cpw;
        // workgroup numbering that keeps a channel slab on one XCD (or on 8 / slabs of them)
        int mode = slabs % 8 == 0 ? 1 : (slabs < 8 && 8 % slabs == 0) ? 2 : 0;
        if (crop_tuning().linear_map) mode = 0;
        int64_t wgs = mode == 1 ? static_cast<int64_t>(num_boxes) * slabs
                    : mode == 2 ? 8ll * ((num_boxes + 8 / slabs - 1) / (8 / slabs)) : 0;
        if (mode != 0 && wgs > 0x7FFFFFFF) mode = 0;
        if (mode != 0 || slabs <= 65535) {
            const dim3 grid = mode == 0 ? dim3(num_boxes, slabs) : dim3(static_cast<unsigned>(wgs));
            int ring_slots = CS_RING_SLOTS;
            if (crop_tuning().ring_slots) ring_slots = crop_tuning().ring_slots;
            // + the dword a (lo, lo+1) pair may read past the last slot; and never less than the in-launch gather fallback
            // (footprints of more than 256 slots) writes from the start of LDS: its sample table + one Tap per crop position
            // (16 x 16 crops: 512 + 8192 B, more than the default ring)
            const size_t lds = std::max(static_cast<size_t>(ring_slots) * 16 + 16,
                                        sizeof(Sample) * (crop_height + crop_width) + sizeof(Tap) * plane);
            hipLaunchKernelGGL(crop_forward_nchw_staged, grid, dim3(64), lds, s, image, batch, depth,
                               height, width, boxes, box_index, num_boxes, extrapolation_value, crop_height, crop_width, cpw,
                               slabs, mode, ring_slots, crops);
            return mrcnn::check_launch("crop_forward_nchw_staged");
        }
    }
    // a channel slab gives each workgroup ~2k outputs (measured best of 1k/2k/4k/8k: thousands of workgroups keep every CU gathering); grid.y <= 65535
    int slab = (2048 + plane - 1) / plane;
    if (slab < 1) slab = 1;
    if (slab > depth) slab = depth;
    int gy = (depth + slab - 1) / slab;
    if (gy > 65535) { gy = 65535; slab = (depth + gy - 1) / gy; gy = (depth + slab - 1) / slab; }
    const size_t lds = sizeof(Sample) * (crop_height + crop_width) + (plane <= 1024 ? sizeof(Tap) * plane : 0);
    hipLaunchKernelGGL(crop_forward_nchw, dim3(num_boxes, gy), dim3(256), lds,
                       s, image, batch, depth, height, width, boxes,
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

extern "C" int mrcnn_roi_align_pyramid_counted_f32(const float* const fm[4], const int32_t fm_h[4], const int32_t fm_w[4],
                                                   int32_t batch, int32_t depth, const float* rois, const int32_t* roi_batch,
                                                   int32_t num_rois, int32_t rois_per_image, const int32_t* roi_counts,
                                                   int32_t pool, float image_area, void* out, int32_t out_layout,
                                                   int32_t* levels_out, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(roi_counts == nullptr || (roi_batch == nullptr && rois_per_image >= 1 && num_rois % rois_per_image == 0),
                  "roi_align_pyramid: roi_counts needs rois_per_image (no roi_batch) and whole images of slots");
    MRCNN_REQUIRE(fm && fm_h && fm_w && rois && out, "roi_align_pyramid: null pointer");
    MRCNN_REQUIRE(batch >= 1 && depth >= 4 && depth % 4 == 0, "roi_align_pyramid: depth=%d must be a multiple of 4", depth);
    MRCNN_REQUIRE(pool >= 1 && pool <= 1024, "roi_align_pyramid: pool=%d", pool);
    MRCNN_REQUIRE(roi_batch || rois_per_image >= 1, "roi_align_pyramid: need roi_batch or rois_per_image");
    MRCNN_REQUIRE(out_layout == MRCNN_LAYOUT_NHWC || out_layout == MRCNN_LAYOUT_NHWC_F16 ||
                      (out_layout == MRCNN_LAYOUT_KBLOCKED && depth % 8 == 0),
                  "roi_align_pyramid: out_layout must be NHWC, NHWC_F16, or k-blocked with depth %% 8 == 0");
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
                       pool, image_area, static_cast<float*>(out), levels_out,
                       out_layout == MRCNN_LAYOUT_KBLOCKED ? 1 : out_layout == MRCNN_LAYOUT_NHWC_F16 ? 2 : 0,
                       static_cast<int64_t>(num_rois) * pool * pool, roi_counts);
    return mrcnn::check_launch("roi_align_pyramid_nhwc");
}

extern "C" int mrcnn_roi_align_pyramid_f32(const float* const fm[4], const int32_t fm_h[4], const int32_t fm_w[4],
                                           int32_t batch, int32_t depth, const float* rois, const int32_t* roi_batch,
                                           int32_t num_rois, int32_t rois_per_image, int32_t pool, float image_area,
                                           void* out, int32_t out_layout, int32_t* levels_out, mrcnn_stream_t stream) {
    return mrcnn_roi_align_pyramid_counted_f32(fm, fm_h, fm_w, batch, depth, rois, roi_batch, num_rois, rois_per_image, nullptr,
                                               pool, image_area, out, out_layout, levels_out, stream);
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
