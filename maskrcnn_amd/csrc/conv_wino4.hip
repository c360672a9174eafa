// 3x3 stride-1 SAME convolution + per-channel affine + ReLU as a fused Winograd F(4x4, 3x3) on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32): 4x fewer multiply-adds than the direct implicit GEMM (F(2x2,3x3) of conv_wino.hip: 2.25x)
// for the large-map 3x3 layers — FPN smoothing (model.py:154-157), the RPN's shared conv (model.py:605,624; with the two
// 1x1 heads fused, model.py:606-607,627-641) and the Bottleneck conv2 layers (model.py:182). All arithmetic is fp32
// (filter transform in double, once); the transforms' larger coefficients cost about half a decimal digit against
// F(2x2): max |err| 1.8e-5 .. 3.3e-5 at |act| ~ 8 on the full-size layers, inside the 1e-4 parity bar
// (tests/test_gpu_conv.py, tests/test_gpu_fullsize.py).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A       g: 3x3 filter, d: 6x6 input patch, Y: 4x4 outputs
//   interpolation points 0, +-3/4, +-3/2, inf: the usual 0, +-1, +-2 scaled by 3/4, which cuts the fp32 error of the
//   transforms to a third (measured; every matrix entry stays a dyadic rational, exact in fp32)
//
//   GEMM view   36 independent products, one per transform component (xi, nu): M[T x N] = V[T x C] U[C x N], T = tile
//               positions (one per 4x4 output pixels), N = Cout, C = Cin.
//   tile        a workgroup = 4 waves, ONE per SIMD, owns 32 positions (a 4 x 8 block of one image: 16 x 32 output pixels)
//               x 64 output channels x all 36 components: 288 accumulator registers per lane — the whole 512-entry
//               register file (256 AGPRs + 32 VGPRs hold the accumulators; the MFMAs are inline asm so that each
//               accumulator's register class is fixed). Wave (qa, qb) owns the 3 x 3 quadrant xi in 3qa..3qa+2, nu in
//               3qb..3qb+2 of the component grid: its A operands need only a 5 x 5 part of the 6 x 6 patch.
//   staging     per k tile of 4 input channels: the raw 18 x 34 x 4 input region through registers into LDS (two
//               channel-pair planes; a row's base is rotated by row/4 so that the 32 lanes of a half, which sit 4 pixels
//               apart in both directions, read 32 different bank pairs), and the 36 x 64 x 4 block of U by LDS-DMA
//               (buffer_load ... lds, no registers: a wave moves 9 components, 1 KB each). Double-buffered, one barrier
//               per k tile.
//   transform   an MFMA A-operand register is one position x one channel per lane, so every lane transforms its OWN
//               position out of the raw region (25 8-byte reads, 48 packed VALU ops) straight into the 9 A operands of
//               its wave's components — V is never stored. Each A operand feeds two MFMAs (the two 32-channel halves
//               of the N tile).
//   epilogue    the 36 components of 8 positions x 64 channels go through LDS per round; every thread then applies
//               A^T . A to one position x one channel pair (the pair rides in packed operations), affine, ReLU, 8-byte
//               stores NHWC and/or k-blocked. HEADS variant: instead of the stores the round's 128 pixels x 64 channels
//               feed the RPN's 1x1 heads by MFMA (details at wino4_wave).
#include "conv_common.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace mrcnn_conv;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// LDS pointers stay in their address space: arithmetic on generic pointers costs a null check per conversion (and the
// LDS-DMA destination / asm read addresses are 32-bit LDS offsets anyway)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

struct Wino4Params {
    const float* x;      // k-blocked input  [Cin/8][B*H*W][8]
    const float* u;      // transformed filter [Cin/4][36][2 channel pairs][Cout][2]
    const float* scale;  // [Cout] or null
    const float* shift;  // [Cout] or null
    float* y;            // NHWC output [B][H][W][Cout], or null
    float* yk;           // k-blocked output [Cout/8][B*H*W][8], or null
    int B, H, W, Cin, Cout, TH, TW, act;  // TH, TW: tile positions per image (H/4, W/4)
    int tyb, txb;                         // 4 x 8-position blocks per image along y / x
    int tiles_m, tiles_n;
    unsigned x_plane, u_ktile, yk_plane;  // bytes per 8-channel plane of x, per k tile of u, per 8-channel output plane
    unsigned x_bytes, u_bytes, y_bytes;
    // fused 1x1 heads (HEADS): w_head [32][Cout] (rows >= the real head count are zero); head_part [tiles_m * 512 pixel
    // rows, M-tile-major: row = mt * 512 + position * 16 + i * 4 + j][32] sums over all channels, without the bias
    const float* w_head;
    float* head_part;
    unsigned head_bytes;
    // fused 1x1 expansion (CONV3; Cout == 64): w3 [C3][64] (OHWI of a 1x1 conv), scale3 / shift3 [C3] or null, residual and
    // y3 NHWC [B][H][W][C3]: y3 = relu(scale3 * (relu(scale * conv3x3 + shift) W3^T) + shift3 + residual)
    const float* w3;
    const float* scale3;
    const float* shift3;
    const float* res3;
    float* y3;
    int C3;
    unsigned w3_bytes, y3_bytes;
    int debug;  // MRCNN_W4_DEBUG: timing ablations (wrong results): 1 no DMA, 2 no transform, 4 no patch reads, 8 no B reads, 16 no raw staging
};

// The barriers of the kernel, all by hand (a __syncthreads() would also drain the vector-memory queue: the raw loads that are
// meant to stay in flight, the previous round's global stores). STAGING (end of the prologue and of every k tile that staged
// something): every LDS-DMA this wave issued has landed — `vm` is the number of YOUNGER vector-memory operations, the raw
// loads to registers, that may stay in flight —, every LDS access of the wave is retired, barrier. EPILOGUE: LDS only.
#define W4_STAGE_WAIT_AND_BARRIER(vm) \
    do { asm volatile("s_waitcnt vmcnt(" #vm ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); MRCNN_SYNC_FUZZ_POINT(); } while (0)
#define W4_EPILOGUE_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); MRCNN_SYNC_FUZZ_POINT(); } while (0)
constexpr int W4_N = 64;                              // output channels per workgroup
#ifndef MRCNN_W4_WALK_SHIFT
#define MRCNN_W4_WALK_SHIFT 4
#endif
constexpr int W4_WALK_SHIFT = MRCNN_W4_WALK_SHIFT;    // HEADS: runs of 16 consecutive M tiles start their N-tile walk together
constexpr int W4_TSLOT = 6;                           // MFMA slot of the k loop that carries the input transform
constexpr int W4_RW = 40;                             // channel pairs per raw row in LDS (34 used; == 0 mod 8)
constexpr int W4_RPLANE = 18 * W4_RW + 8;             // pairs per (buffer, channel pair) plane, rotation included
constexpr int W4_RS_FLOATS = 2 * 2 * W4_RPLANE * 2;   // raw region: [2 buffers][2 channel pairs][plane][2]
constexpr int W4_UBUF = 36 * 256;                     // floats per U buffer: [36][2 channel pairs][64][2]
constexpr int W4_Z_FLOATS = 36 * 8 * 64;              // epilogue exchange of a round: [36 components][8 positions][64 channels]
constexpr int W4_DUMP_FLOATS = 256;                   // where the LDS-DMAs of k tiles past the last land (dma_u1): behind everything else
constexpr size_t WINO4_LDS = sizeof(float) * (W4_RS_FLOATS + 2 * W4_UBUF + W4_DUMP_FLOATS);
// HEADS, epilogue: behind Z the round's transposed tile T (128 pixels x 64 channels; it overlaps U buffer 1, dead then), the
// N tile's head weights (the 8 B-operand pieces [j][lane][4]) and — alive through the whole N-tile walk of an M tile, beyond
// the staging area — the M tile's head sums [512 pixels][20] (18 real columns)
constexpr int W4_T_FLOATS = 128 * W4_N;
constexpr int W4_WH_FLOATS = 8 * 256;
constexpr int W4_HP = 20;
constexpr int W4_T_OFF = W4_Z_FLOATS;
constexpr int W4_WH_OFF = W4_T_OFF + W4_T_FLOATS;
constexpr int W4_H_OFF = W4_WH_OFF + W4_WH_FLOATS;
constexpr size_t WINO4_HEADS_LDS = sizeof(float) * (W4_H_OFF + 512 * W4_HP + W4_DUMP_FLOATS);
static_assert(W4_H_OFF >= W4_RS_FLOATS + 2 * W4_UBUF && WINO4_HEADS_LDS <= 160 * 1024, "HEADS LDS map");
// CONV3: the 1x1 expansion's weights W3 [<= 256][64] stay in LDS for the workgroup's whole life, ABOVE the staging area (the
// k loop never touches them), as 64 B-operand pieces [column block][j][lane][4]; the epilogue's transposed tile T has no
// room of its own then — it ALIASES Z: row (position p8, pixel r) of T lies in Z's chunk (component r, position p8), which
// only the half-wave that owns position p8 reads (all 36 of its chunks, into registers) before it writes T there
constexpr int W4_W3_OFF = W4_RS_FLOATS + 2 * W4_UBUF;
constexpr int W4_W3_FLOATS = 256 * W4_N;
constexpr size_t WINO4_CONV3_LDS = sizeof(float) * (W4_W3_OFF + W4_W3_FLOATS + W4_DUMP_FLOATS);
static_assert(WINO4_CONV3_LDS <= 160 * 1024 && W4_Z_FLOATS <= W4_W3_OFF, "CONV3 LDS map");
static_assert(W4_Z_FLOATS <= W4_RS_FLOATS + 2 * W4_UBUF, "the exchange buffer aliases the staging buffers");

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {  // a * b + c
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 b, f32x2 c) {  // c - a * b
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The MFMA with the accumulator's register class fixed by the constraint: 16 of a wave's 18 accumulators live in AGPRs,
// 2 in VGPRs (left to itself the register allocator shuttles 288 accumulator registers between the two files inside the
// loop). No wait state in front: the compiler pads only its own instructions, but in this kernel an MFMA's A operand was
// written by VALU code a whole k tile earlier (the prologue's at least a barrier earlier) and its B operand by an LDS read
// that an s_waitcnt has retired — never by the instruction just before.
__device__ __forceinline__ void mfma_a(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// One ds_read_b64, as asm: left to the compiler, pairs of these reads become ds_read2_b64 (half rate, banked mod 32 — the
// raw region's layout is conflict-free for ds_read_b64 only), and a volatile access becomes a flat load with a full wait.
// The compiler does not count an asm load: every consumer sits behind an explicit s_waitcnt lgkmcnt.
__device__ __forceinline__ f32x2 lds_read_b64(unsigned addr, int offset) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(offset));
    return v;
}

// Three rows of B^T applied to five consecutive samples. With the points 0, +-a, +-b, inf (a = 3/4, b = 3/2) the rows are
//   [a2b2 0 -(a2+b2) 0 1 0] [0 -ab2 -b2 a 1 0] [0 ab2 -b2 -a 1 0] [0 -a2b -a2 b 1 0] [0 a2b -a2 -b 1 0] [0 a2b2 0 -(a2+b2) 0 1]
// (a2 = a^2 ...; a = 1, b = 2 gives the textbook matrix). Q = 0: rows 0..2 of samples d0..d4 (= e0..e4); Q = 1: rows 3..5
// of samples d1..d5 (= e0..e4).
constexpr float W4_A = 0.75f, W4_B = 1.5f;
struct W4Consts { f32x2 a2b2, sab, b2, a, a2, b; };  // sab = a2 + b2
__device__ __forceinline__ W4Consts w4_consts() {
    constexpr float a2 = W4_A * W4_A, b2 = W4_B * W4_B;
    return {{a2 * b2, a2 * b2}, {a2 + b2, a2 + b2}, {b2, b2}, {W4_A, W4_A}, {a2, a2}, {W4_B, W4_B}};
}
// One asm statement per call: every statement boundary costs a compiler-inserted wait state in this loop.
template <int Q>
__device__ __forceinline__ void bt3(const f32x2 e0, const f32x2 e1, const f32x2 e2, const f32x2 e3, const f32x2 e4,
                                    const W4Consts& k, f32x2& r0, f32x2& r1, f32x2& r2) {
    f32x2 t1, t2;
    if constexpr (Q == 0) {
        asm("v_pk_fma_f32 %3, %10, %7, %9 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // t1 = d4 - b2 d2
            "v_pk_fma_f32 %4, %10, %6, %8 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // t2 = d3 - b2 d1
            "v_pk_fma_f32 %0, %11, %7, %9 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // r0 = d4 - (a2+b2) d2
            "v_pk_fma_f32 %1, %13, %4, %3\n\t"                                   // r1 = t1 + a t2
            "v_pk_fma_f32 %2, %13, %4, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // r2 = t1 - a t2
            "v_pk_fma_f32 %0, %12, %5, %0"                                        // r0 += a2b2 d0
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(t1), "=&v"(t2)
            : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(k.b2), "v"(k.sab), "v"(k.a2b2), "v"(k.a));
    } else {
        asm("v_pk_fma_f32 %3, %10, %6, %8 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // u1 = d4 - a2 d2   (e3 - a2 e1)
            "v_pk_fma_f32 %4, %10, %5, %7 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // u2 = d3 - a2 d1   (e2 - a2 e0)
            "v_pk_fma_f32 %2, %11, %7, %9 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // r2 = d5 - (a2+b2) d3
            "v_pk_fma_f32 %0, %13, %4, %3\n\t"                                   // r0 = u1 + b u2
            "v_pk_fma_f32 %1, %13, %4, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"    // r1 = u1 - b u2
            "v_pk_fma_f32 %2, %12, %5, %2"                                        // r2 += a2b2 d1
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(t1), "=&v"(t2)
            : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(k.a2), "v"(k.sab), "v"(k.a2b2), "v"(k.b));
    }
}

// A^T (4 x 6) applied to six samples: rows [1 1 1 1 1 0] [0 1 -1 2 -2 0] [0 1 1 4 4 0] [0 1 -1 8 -8 1]
// A^T (4 x 6) applied to six samples: [1 1 1 1 1 0] [0 a -a b -b 0] [0 a2 a2 b2 b2 0] [0 a3 -a3 b3 -b3 1] — dyadic rationals
// like B^T's entries, so the kernel applies both transforms without rounding a coefficient
struct W4OutConsts { f32x2 a, b, a2, b2, a3, b3; };
__device__ __forceinline__ W4OutConsts w4_out_consts() {
    constexpr float a2 = W4_A * W4_A, b2 = W4_B * W4_B;
    return {{W4_A, W4_A}, {W4_B, W4_B}, {a2, a2}, {b2, b2}, {a2 * W4_A, a2 * W4_A}, {b2 * W4_B, b2 * W4_B}};
}
__device__ __forceinline__ void at4p(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                     const W4OutConsts& k, f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3) {
    f32x2 s12, d12, s34, d34;
    asm("v_pk_add_f32 %4, %9, %10\n\t"                                   // s12 = m1 + m2
        "v_pk_add_f32 %5, %9, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"         // d12 = m1 - m2
        "v_pk_add_f32 %6, %11, %12\n\t"                                  // s34 = m3 + m4
        "v_pk_add_f32 %7, %11, %12 neg_lo:[0,1] neg_hi:[0,1]\n\t"        // d34 = m3 - m4
        "v_pk_add_f32 %0, %8, %4\n\t"                                    // y0 = m0 + s12
        "v_pk_mul_f32 %1, %14, %5\n\t"                                   // y1 = a d12
        "v_pk_mul_f32 %2, %16, %4\n\t"                                   // y2 = a2 s12
        "v_pk_fma_f32 %3, %18, %5, %13\n\t"                              // y3 = a3 d12 + m5
        "v_pk_add_f32 %0, %0, %6\n\t"                                    // y0 += s34
        "v_pk_fma_f32 %1, %15, %7, %1\n\t"                               // y1 += b d34
        "v_pk_fma_f32 %2, %17, %6, %2\n\t"                               // y2 += b2 s34
        "v_pk_fma_f32 %3, %19, %7, %3"                                    // y3 += b3 d34
        : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(s12), "=&v"(d12), "=&v"(s34), "=&v"(d34)
        : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(k.a), "v"(k.b), "v"(k.a2), "v"(k.b2), "v"(k.a3), "v"(k.b3));
}
// One wave's share of the kernel; QA, QB = its quadrant of the component grid (compile-time: the transform's operations
// differ per quadrant; the four waves of a workgroup run four instances of this code and meet at the same barriers).
// HEADS: the output tile is not stored; it feeds the RPN's two 1x1 heads on chip (as conv3x3_wino8s_f32<true> of
// conv_wino.hip): a workgroup owns whole M tiles and walks their N tiles, adding each N tile's contribution to the M tile's
// head sums, which stay in LDS until the last N tile writes them out.
// DBG: compile-time tuning variants (MRCNN_W4_ABLATIONS builds): bits 1..1024 leave parts out (wrong results, timing only),
// 2048 records s_memtime stamps of a tile's phases (tools/w4_stamp.py); 4096 delays wave 0 behind the prologue's staging
// barrier and 8192 leaves out the barrier behind the prologue's operand reads (4096 alone: results unchanged; 4096 + 8192: round
// 5's rare wrong tiles on every tile — tools/w4_war_demo.py). DBG = 0 is the product.
template <int QA, int QB, int DBG, bool HEADS, bool ACT, bool CONV3 = false>
__device__ __forceinline__ void wino4_wave(const Wino4Params& p, lds_f32* smem) {
    static_assert(!(HEADS && CONV3), "one fused consumer at a time");
    constexpr int W4_DUMP_OFF = HEADS ? W4_H_OFF + 512 * W4_HP : CONV3 ? W4_W3_OFF + W4_W3_FLOATS : W4_RS_FLOATS + 2 * W4_UBUF;
    lds_f32x2* Rs = (lds_f32x2*)smem;            // [2][2][W4_RPLANE] channel pairs
    lds_f32* Us = smem + W4_RS_FLOATS;           // [2][36][2][64][2]
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int wave = QA * 2 + QB;
    const int ln = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
    const int nk = p.Cin >> 2;
    const int total_tiles = 8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
    const W4Consts kin = w4_consts();

    unsigned long long stamp[16];
    int nstamp = 0;
    auto STAMP = [&]() { if constexpr ((DBG & 2048) != 0) { if (nstamp < 16) stamp[nstamp++] = __builtin_readcyclecounter(); } };
    if constexpr (CONV3) {
        // W3 → LDS once per workgroup, as B-operand pieces: piece (cb, j) = what lane (column ln of block cb, half lh) multiplies
        // in k steps 4 j .. 4 j + 3, i.e. W3[cb * 32 + ln][8 j + 4 lh .. + 3]; wave w moves pieces 16 w .. 16 w + 15. The wait
        // is the vmcnt(0) behind the first tile's k loop.
        const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, p.w3_bytes, 0x00020000);
        const int npieces = (p.C3 >> 5) * 8;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int piece = wave * 16 + q, cb = piece >> 3, j = piece & 7;
            if (piece < npieces)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3_rsrc, smem + W4_W3_OFF + piece * 256, 16,
                                                         static_cast<int>(((cb * 32 + ln) * W4_N + lh * 4) * 4), j * 32, 0, 0);
        }
    }
    for (int it = 0;; ++it) {
        if ((DBG & 2048) && it == 1) nstamp = 0;
        STAMP();  // 0: tile start
        // virtual tile b, b + grid, ... in the XCD-aware order of conv_wino.hip: the workgroups of one XCD walk the N
        // tiles of neighbouring M tiles, so the raw input region is shared in that XCD's L2
        // HEADS: M-tile units b, b + grid, ..., each walked over all its N tiles by this workgroup
        const int tile = HEADS ? (blockIdx.x + (it / p.tiles_n) * gridDim.x) * p.tiles_n + it % p.tiles_n
                               : blockIdx.x + it * gridDim.x;
        if (tile >= total_tiles) break;
        int xcd, seq;
        if constexpr (HEADS) {
            const int unit = tile / p.tiles_n;
            xcd = unit & 7;
            seq = (unit >> 3) * p.tiles_n + (tile - unit * p.tiles_n);
        } else {
            xcd = tile & 7;
            seq = tile >> 3;
        }
        const int mt_lo = (xcd * p.tiles_m) >> 3, mt_hi = ((xcd + 1) * p.tiles_m) >> 3;
        int mt = mt_lo + seq / p.tiles_n;
        if (mt >= mt_hi) continue;  // uniform
        if constexpr (HEADS) {
            // XCD x walks its M range from an offset of x runs: the eight XCDs are then on different N-tile slices of U at any
            // time (which M tile a workgroup takes when has no bearing on the numbers)
            const int span = mt_hi - mt_lo;
            int o = (mt - mt_lo) + (xcd << W4_WALK_SHIFT);
            while (o >= span) o -= span;
            mt = mt_lo + o;
        }
        const int per_img = p.tyb * p.txb;
        const int b = mt / per_img, trem = mt - b * per_img;
        const int tby = trem / p.txb, tbx = trem - tby * p.txb;
        // HEADS: step `walk` of the M tile's N-tile walk. The walk starts at an N tile that depends on the M tile's place IN ITS
        // IMAGE (never on the batch: image i alone == slice i of a batch, bit for bit): with every workgroup on the same N
        // tile at the same time all 256 stream the same 36 KB of U per k tile and the loop runs 10 % slower (3 200 against
        // 2 900 cycles per k tile, in-kernel stamps) than when the eight slices are in use side by side. Round 3: runs of
        // 2^W4_WALK_SHIFT consecutive M tiles — which an XCD works on at the same time — start at the SAME N tile, so an XCD's L2
        // fetches a slice of U once for all of them instead of all eight slices per step (fabric reads of the heads launches:
        // profiles/r03_hbm_traffic.json)
        const int walk = seq % p.tiles_n;
        const int nt = HEADS ? (walk + (trem >> W4_WALK_SHIFT)) % p.tiles_n : walk;
        const int n0 = nt * W4_N;
        const int TY0 = tby * 4, TX0 = tbx * 8;          // first tile position of the block
        const int iy0 = 4 * TY0 - 1, ix0 = 4 * TX0 - 1;  // first raw input pixel (may be -1: zero padding)

        // ---- staging: raw pixels tid, tid + 256, tid + 512 of the 18 x 34 region (16 bytes = the k tile's 4 channels)
        unsigned r_off[3];
        int r_lds[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + 256 * i;
            const int r = f / 34, c = f - r * 34;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = f < 612 && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                            static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
            r_off[i] = ok ? static_cast<unsigned>((b * p.H + iy) * p.W + ix) * 32u : OOB;
            r_lds[i] = f < 612 ? r * W4_RW + (r >> 2) + c : W4_RPLANE - 1;  // the plane's last pair is never read
        }
        // U by LDS-DMA: wave w moves components 9w..9w+8; lane = (channel pair lh, channels n0 + 2 ln, +1)
        const unsigned u_voff = static_cast<unsigned>((lh * p.Cout + n0 + 2 * ln) * 2) * 4u;
        const unsigned u_comp = static_cast<unsigned>(p.Cout) * 16u;  // bytes per component of a k tile

        u32x4 rr[3];
        // The k loop runs one instruction stream for every k tile, the last two included — they stage "k tiles" nk and nk + 1.
        // Raw pixels: the last k tile again, into registers and the raw buffers nobody reads any more (plain LDS stores,
        // retired by the wave's own lgkmcnt(0) in front of the next barrier). U: see dma_u1.
        auto kclamp = [&](int kt) { return kt < nk ? kt : nk - 1; };
        auto load_raw1 = [&](int kt, int i) {
            const int k = kclamp(kt);
            const int soff = static_cast<int>(static_cast<unsigned>(k >> 1) * p.x_plane + static_cast<unsigned>(k & 1) * 16u);
            rr[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(r_off[i]), soff, 0);
        };
        auto write_raw1 = [&](int buf, int i) {
            lds_f32x2* base = Rs + buf * 2 * W4_RPLANE;
            const float4 v = __builtin_bit_cast(float4, rr[i]);
            base[r_lds[i]] = f32x2{v.x, v.y};
            base[W4_RPLANE + r_lds[i]] = f32x2{v.z, v.w};
        };
        // An LDS-DMA for a k tile that does not exist (kt >= nk: the last two k tiles of the loop) lands in the DUMP — 1 KB that
        // nothing ever reads and nothing aliases — never in the U buffers: those are the bytes the epilogue's exchange area Z
        // (and HEADS' T) is about to occupy, and no byte that a DMA targets may be one that somebody else writes or reads
        // without a covering vmcnt + barrier in between. (Round 5 let these dummies land in the U buffers.) The load itself
        // stays, so that the k tile's vector-memory count (9 DMAs, then 3 raw loads) is the same in every k tile.
        lds_f32* const dump = smem + W4_DUMP_OFF;
        auto dma_u1 = [&](int kt, int buf, int j) {
            const int c = wave * 9 + j;
            const int soff = static_cast<int>(static_cast<unsigned>(kclamp(kt)) * p.u_ktile + static_cast<unsigned>(c) * u_comp);
            lds_f32* dst = kt < nk ? Us + buf * W4_UBUF + c * 256 : dump;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, dst, 16, static_cast<int>(u_voff), soff, 0, 0);
        };

        // the lane's position: px = ln & 7, py = ln >> 3; raw rows 4 py + dy, columns 4 px + dx
        const int lp_x = ln & 7, lp_y = ln >> 3;
        const int rbase = lh * W4_RPLANE + 4 * lp_y * W4_RW + lp_y + 4 * lp_x;
        constexpr int cg0 = (3 * QA) * 6 + 3 * QB;  // the quadrant's first component
        // LDS byte addresses of the lane's patch origin / B column in both buffers (the reads are asm: see lds_read_b64)
        const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
        const unsigned rp_addr[2] = {lds0 + static_cast<unsigned>(rbase) * 8u,
                                     lds0 + static_cast<unsigned>(2 * W4_RPLANE + rbase) * 8u};
        const unsigned up_addr[2] = {lds0 + static_cast<unsigned>(W4_RS_FLOATS * 4 + (lh * 64 + ln) * 8),
                                     lds0 + static_cast<unsigned>((W4_RS_FLOATS + W4_UBUF) * 4 + (lh * 64 + ln) * 8)};

        f32x16 acc[18];
#pragma unroll
        for (int i = 0; i < 18; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // acc[0] is in its register class BEFORE the k loop: left to the loop's first MFMA, the HEADS and CONV3 instantiations
        // copied a[0:15] to VGPRs and back at the head of every loop trip (32 v_accvgpr_* per two k tiles) — and a build whose
        // register allocation moved that copy INTO the loop read the accumulator 6 wait states behind an asm MFMA, which the
        // compiler cannot see: wrong sums (tools/isa_audit.py checks C and D pin both)
        asm volatile("" : "+a"(acc[0]));

        // A: operand sets of the k tile in flight and the next one. B: ONE set, refreshed in place — the MFMAs run in
        // groups of four (two accumulators x two k steps, alternating, so that no MFMA waits for the one before it), Bv[2g]
        // and Bv[2g+1] are dead after slot 4g+3 and the next k tile's values are read into them at slots 4g+4, 4g+5; only
        // the last group (dead after the last slot) has a second register set.
        f32x2 A[2][9], Bv[16], Bl[2][2];
        f32x2 e[5][5], t[3][5];
        auto read_col = [&](int buf, int c) {  // column QB + c of the patch, rows QA..QA+4
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int dy = QA + r, dx = QB + c;
                e[c][r] = lds_read_b64(rp_addr[buf], (dy * W4_RW + (dy >= 4 ? 1 : 0) + dx) * 8);
            }
        };
        // newer LDS reads than the column's own may stay in flight (wait_n of them: the schedule below knows the count)
        auto row_stage = [&](int c, int wait_n) {
            if (wait_n < 0) {}
            else if (wait_n >= 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(e[c][0]), "+v"(e[c][1]), "+v"(e[c][2]), "+v"(e[c][3]), "+v"(e[c][4]));
            else if (wait_n == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(e[c][0]), "+v"(e[c][1]), "+v"(e[c][2]), "+v"(e[c][3]), "+v"(e[c][4]));
            else if (wait_n == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(e[c][0]), "+v"(e[c][1]), "+v"(e[c][2]), "+v"(e[c][3]), "+v"(e[c][4]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[c][0]), "+v"(e[c][1]), "+v"(e[c][2]), "+v"(e[c][3]), "+v"(e[c][4]));
            bt3<QA>(e[c][0], e[c][1], e[c][2], e[c][3], e[c][4], kin, t[0][c], t[1][c], t[2][c]);
        };
        auto col_stage = [&](int set, int i) {
            bt3<QB>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], kin, A[set][i * 3 + 0], A[set][i * 3 + 1], A[set][i * 3 + 2]);
        };
        auto read_b = [&](int set, int buf, int q) {  // q = component * 2 + channel half of the N tile
            const int ci = q >> 1, nb = q & 1;
            const int off = (cg0 + (ci / 3) * 6 + ci % 3) * 128 + nb * 32;
            if (q < 16) Bv[q] = lds_read_b64(up_addr[buf], off * 8);
            else Bl[set][q - 16] = lds_read_b64(up_addr[buf], off * 8);
        };
        auto mfma = [&](int set, int m) {  // slot m: group m / 4 = component; accumulator (channel half) m & 1, k step (m >> 1) & 1
            const int s = (m >> 1) & 1, q = (m >> 2) * 2 + (m & 1);
            const f32x2 bq = q < 16 ? Bv[q] : Bl[set][q - 16];
            const float a = s == 0 ? A[set][q >> 1].x : A[set][q >> 1].y;
            const float bb = s == 0 ? bq.x : bq.y;
            if (q < 16) mfma_a(acc[q], a, bb);
            else mfma_v(acc[q], a, bb);
        };

        auto tie_b = [&](int set) {  // the asm reads of B have landed (lgkmcnt(0) just before): order their consumers behind
            asm volatile("" : "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]), "+v"(Bv[4]), "+v"(Bv[5]), "+v"(Bv[6]), "+v"(Bv[7]), "+v"(Bv[8]));
            asm volatile("" : "+v"(Bv[9]), "+v"(Bv[10]), "+v"(Bv[11]), "+v"(Bv[12]), "+v"(Bv[13]), "+v"(Bv[14]), "+v"(Bv[15]), "+v"(Bl[set][0]), "+v"(Bl[set][1]));
        };

        STAMP();  // 1: setup done
        // ---- prologue: k tiles 0 and 1 staged, operands of k tile 0 in registers, raw k tile 2 in flight. All loads of k tiles
        // 0 and 1 go out together (one memory round trip, not two)
        {
            u32x4 r0[3], r1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r0[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(r_off[i]), 0, 0);
                r1[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(r_off[i]), 16, 0);
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) dma_u1(0, 0, j);
#pragma unroll
            for (int j = 0; j < 9; ++j) dma_u1(1, 1, j);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rr[i] = r0[i];
                write_raw1(0, i);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rr[i] = r1[i];
                write_raw1(1, i);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) load_raw1(2, i);
        W4_STAGE_WAIT_AND_BARRIER(3);
        if constexpr ((DBG & 4096) != 0) {  // ablation builds: wave 0 falls about 8 000 cycles behind the others here
            if (wave == 0) __builtin_amdgcn_s_sleep(127);
        }
        // The 18 B reads first, then the patch one column AHEAD of the column being transformed (LDS returns in order: with
        // the next column's five reads in flight, lgkmcnt(5) retires this column and every B read): round 6 — read, wait for
        // all, transform, five times over and the B reads last, left six LDS latencies exposed in every tile's prologue
#pragma unroll
        for (int q = 0; q < 18; ++q) read_b(0, 0, q);
        read_col(0, 0);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (c + 1 < 5) read_col(0, c + 1);
            row_stage(c, c + 1 < 5 ? 5 : 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) col_stage(0, i);
        // THE barrier round 5's rare wrong tiles were missing (round 6, tools/w4_forensics.py): the reads above are
        // of buffer 0 (raw k tile 0, U(0)) AFTER the staging barrier, and k tile 0 below re-stages buffer 0 — U(2) by LDS-DMA
        // from MFMA slot 10 on, raw k tile 2 in slots 27-29. In steady state a k tile's operands are read during the k tile
        // before, in front of ITS closing barrier; here nothing separated a slow wave's reads from the other waves' writes
        // except those 10 slots (about 700 cycles). Every captured wrong tile (50 of 50) was wave 0's quadrant of k tile 0
        // computed from another wave's U(2) pieces (exactly) or raw k tile 2 pixels. All reads retired, then the barrier.
        if constexpr ((DBG & 8192) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ablation builds: round 5's kernel
        else W4_EPILOGUE_BARRIER();
        tie_b(0);

        // ---- one k tile. 36 MFMA slots, the order pinned (sched_barrier after every slot); beside the MFMAs of k tile kt:
        //   slots 10,11,14,15,..,26  LDS-DMA of U(kt+2) into the buffer U(kt) has left (its B reads completed before the last
        //                barrier): the slots that carry nothing else — a DMA beside LDS reads and VALU work delays the next MFMA
        //   slots 0-4    the lane's patch of k tile kt+1, one column per slot;  slot 6 its whole transform -> A(kt+1), all 48
        //                packed operations behind one MFMA (a counted wait in front). Spread over eight slots, six operations
        //                each, the same work cost 3 % more of the kernel: every switch between the MFMA stream and VALU
        //                work costs, so VALU work is bunched — LDS reads and DMA pieces, the opposite, are spread (bunched
        //                they stall their queues: all 25 patch reads in one slot +2.5 %, all 9 DMA pieces +4 %)
        //   slots 4-35   B(kt+1): slots 4g+4, 4g+5 refill the registers group g has left; slots 34, 35 the last group's
        //   slots 27-29  raw k tile kt+2: registers -> LDS (the buffer of kt, last read a k tile ago); reload with kt+3
        // then vmcnt(3) (the three raw loads may stay in flight, every DMA has landed), lgkmcnt(0), barrier.
        auto b_in_slot = [](int sl) { return (sl >= 4 && sl <= 33 && (sl & 3) < 2) || sl == 34 || sl == 35 ? 1 : 0; };
        auto ktile = [&](auto par, int kt) {
            constexpr int CUR = decltype(par)::value, NXT = CUR ^ 1;
#pragma unroll
            for (int slot = 0; slot < 36; ++slot) {
                mfma(CUR, slot);
                if (slot >= 10 && slot <= 26 && (slot & 3) >= 2 && !(DBG & 1)) dma_u1(kt + 2, CUR, ((slot - 10) >> 2) * 2 + (slot & 1));
                if (slot == W4_TSLOT && !(DBG & 2)) {   // the whole transform behind ONE MFMA (see the slot table above)
#pragma unroll
                    for (int c = 0; c < 5; ++c) row_stage(c, c == 0 ? ((DBG & 12) ? 0 : b_in_slot(4) + b_in_slot(5)) : -1);
#pragma unroll
                    for (int i = 0; i < 3; ++i) col_stage(NXT, i);
                }
                if (slot < 5 && !(DBG & 4)) read_col(NXT, slot);
                if (slot >= 4 && slot <= 33 && (slot & 3) < 2 && !(DBG & 8)) read_b(NXT, NXT, ((slot - 4) >> 2) * 2 + (slot & 1));
                if (slot >= 34 && !(DBG & 8)) read_b(NXT, NXT, slot - 18);
                if (slot >= 27 && slot < 30 && !(DBG & 16)) {
                    write_raw1(CUR, slot - 27);
                    load_raw1(kt + 3, slot - 27);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            W4_STAGE_WAIT_AND_BARRIER(3);
            tie_b(NXT);
        };
        STAMP();  // 2: prologue done
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(std::integral_constant<int, 0>{}, kt);
            ktile(std::integral_constant<int, 1>{}, kt + 1);
        }
        // The MFMAs are inline asm: the compiler's hazard recogniser does not see them. Guarantee — rather than rely on the
        // instructions that happen to sit in between — the wait states a 16-pass MFMA needs before its accumulator is read
        // (18 for a VALU / vector-memory read of the result): 20 here, once per tile.
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        STAMP();  // 3: k loop done

        // ---- epilogue: four rounds of 8 positions (accumulator registers 4g..4g+3 of both lane halves)
        if (DBG & 64) {  // ablation: no epilogue at all (keeps the accumulators alive through one store)
            float keep = 0.f;
#pragma unroll
            for (int i = 0; i < 18; ++i) keep += acc[i][0];
            if (keep == 12345.678f) p.y[0] = keep;
            continue;
        }
        lds_f32* Z = smem;
        // epilogue thread = (position p8 of the round, channel pair n, n + 1): the pair rides in packed operations and leaves
        // in 8-byte stores
        const int n = 2 * (tid & 31), p8 = tid >> 5;
        const int ng = n0 + n;
        f32x2 sc2 = {1.0f, 1.0f}, sh2 = {0.0f, 0.0f};
        if (p.scale) sc2 = *reinterpret_cast<const f32x2*>(p.scale + ng);
        if (p.shift) sh2 = *reinterpret_cast<const f32x2*>(p.shift + ng);
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y ? p.y_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t yk_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.yk, 0, p.yk ? p.y_bytes : 0u, 0x00020000);
        const unsigned ncol = static_cast<unsigned>(ng) * 4u;
        const unsigned kcol = static_cast<unsigned>(ng >> 3) * p.yk_plane + static_cast<unsigned>(ng & 7) * 4u;
        // HEADS: the round's 128 pixels x 64 channels, [pixel][channel ^ swizzle], behind the exchange buffer
        lds_f32* Tt = smem + W4_T_OFF;
        // row pxl (= position * 16 + pixel of the 4 x 4 tile) of the round's transposed tile
        auto t_row = [&](int pxl) -> lds_f32* {
            if constexpr (CONV3) return Z + ((pxl & 15) * 8 + (pxl >> 4)) * 64;   // inside Z: see W4_W3_OFF
            else return Tt + pxl * W4_N;
        };
        // the N tile's head weights: piece j = what lane (head ln, half lh) multiplies in k steps 4j..4j+3 (channels
        // n0 + 8 j + 4 lh + e), by LDS-DMA behind T — in registers they are 32 VGPRs the transform below has not got
        lds_f32* Wh = smem + W4_WH_OFF;
        if constexpr (HEADS) {
            const __amdgpu_buffer_rsrc_t wh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_head), 0, static_cast<unsigned>(p.Cout) * 128u, 0x00020000);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = wave * 2 + jj;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wh_rsrc, Wh + j * 256, 16, static_cast<int>((ln * p.Cout + n0 + lh * 4) * 4), j * 32, 0, 0);
            }
        }
        const __amdgpu_buffer_rsrc_t hp_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.head_part, 0, HEADS ? p.head_bytes : 0u, 0x00020000);
        const int pixstep_y = p.Cout * 4, rowstep_y = p.W * pixstep_y, rowstep_k = p.W * 32;  // scalar store offsets
        const W4OutConsts kout = w4_out_consts();
        // exchange-area write addresses of the lane: [channel half nb][components >= 32] — position rows 4 lh .. + 3 of a component
        // start (4 lh * 64 + nb * 32 + ln) floats into its 2 KB chunk; the component and the position row are the immediates
        const unsigned z_lane[2][2] = {{lds0 + static_cast<unsigned>((4 * lh * 64 + ln) * 4), lds0 + static_cast<unsigned>((4 * lh * 64 + ln) * 4 + 32 * 2048)},
                                       {lds0 + static_cast<unsigned>((4 * lh * 64 + 32 + ln) * 4), lds0 + static_cast<unsigned>((4 * lh * 64 + 32 + ln) * 4 + 32 * 2048)}};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // Z[component][position][channel]: a lane's four positions (registers 4g..4g+3) are four 4-byte writes (32
            // consecutive channels per half-wave), a thread's channel pair one 8-byte read: both conflict-free
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        // two positions per instruction (ds_write2st64_b32: both dwords 256 B = one position row apart): 36 LDS
                        // instructions per round instead of 72 — the phase is bound by their issue, not by LDS bandwidth
                        constexpr int q = 0;
                        const int acc_i = (i * 3 + j) * 2 + nb;
                        const int comp = cg0 + i * 6 + j;
                        const unsigned zaddr = z_lane[nb][comp >= 32 ? 1 : 0];
                        const int unit = (comp & 31) * 8;   // 256-byte units: component stride 2 KB
                        (void)q;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            if (acc_i < 16)
                                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"
                                             :: "v"(zaddr), "a"(acc[acc_i][4 * g + e]), "a"(acc[acc_i][4 * g + e + 1]), "n"(unit + e), "n"(unit + e + 1) : "memory");
                            else
                                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"
                                             :: "v"(zaddr), "v"(acc[acc_i][4 * g + e]), "v"(acc[acc_i][4 * g + e + 1]), "n"(unit + e), "n"(unit + e + 1) : "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);  // or all 72 accumulator registers of the round are copied out at once
                    }
            // barriers of the epilogue: LDS only. __syncthreads() would also wait (vmcnt(0)) for the previous round's global
            // stores — output pixels or head sums — to complete: their whole latency, once per round
            if (g == 0 || g == 1) STAMP();  // after the Z writes
            W4_EPILOGUE_BARRIER();
            if (g == 0 || g == 1) STAMP();  // after the first barrier
            // the six reads of row xi + 1 are issued BEFORE row xi is transformed (round 6: read, wait for all, transform, six times
            // over, left five LDS latencies of ~150 cycles exposed per round at one wave per SIMD)
            f32x2 w[6][4], m[2][6];
            auto read_row = [&](int xi) {
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) m[xi & 1][nu] = *(const lds_f32x2*)(Z + ((xi * 6 + nu) * 8 + p8) * 64 + n);
            };
            read_row(0);
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                if (xi + 1 < 6) read_row(xi + 1);
                __builtin_amdgcn_sched_barrier(0);   // (or the scheduler sinks the reads behind the transform again)
                at4p(m[xi & 1][0], m[xi & 1][1], m[xi & 1][2], m[xi & 1][3], m[xi & 1][4], m[xi & 1][5], kout, w[xi][0], w[xi][1],
                     w[xi][2], w[xi][3]);
            }
            // second stage per output column j: its four pixels (i = 0..3) leave right away
            const int pos = 8 * g + p8;
            const int TY = TY0 + (pos >> 3), TX = TX0 + (pos & 7);
            const bool valid = TY < p.TH && TX < p.TW;
            const unsigned pix = static_cast<unsigned>((b * p.H + 4 * TY) * p.W + 4 * TX);
            const unsigned base_y = valid ? pix * static_cast<unsigned>(p.Cout) * 4u + ncol : OOB;
            const unsigned base_k = valid ? pix * 32u + kcol : OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 yv[4];
                at4p(w[0][j], w[1][j], w[2][j], w[3][j], w[4][j], w[5][j], kout, yv[0], yv[1], yv[2], yv[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = pk_fma(yv[i], sc2, sh2);
                if constexpr (ACT) {  // one v_max per value (fmaxf() costs a canonicalising v_max, the v_max and a select each)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        asm("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(yv[i].x), "+v"(yv[i].y));
                }
                if (DBG & 32) { if (yv[0].x == 12345.678f) p.y[0] = yv[0].x; continue; }
                if constexpr (HEADS || CONV3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // pixel row of the round: position p8, pixel i * 4 + j
                        const int pxl = p8 * 16 + i * 4 + j;
                        *(lds_f32x2*)(t_row(pxl) + (n ^ ((pxl & 15) << 2))) = yv[i];
                    }
                } else {
                    if (p.y) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yv[i]), y_rsrc, static_cast<int>(base_y),
                                                                  i * rowstep_y + j * pixstep_y, 0);
                    }
                    if (p.yk) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yv[i]), yk_rsrc, static_cast<int>(base_k),
                                                                  i * rowstep_k + j * 32, 0);
                    }
                }
            }
            if (g == 0 || g == 1) STAMP();  // after transform + stores
            if (HEADS && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the head-weight DMA has landed
            W4_EPILOGUE_BARRIER();  // Z is read out: the next round / tile may overwrite it
            if (g == 0 || g == 1) STAMP();  // round end
            if constexpr (HEADS && !(DBG & 128)) {
                // wave w: pixels 32 w .. 32 w + 31 of the round x 32 heads x all 64 channels. The M tile's sums over the N
                // tiles walked so far stay in LDS (this workgroup walks all of them back to back); only the last N tile
                // writes them out
                const int prow = g * 128 + wave * 32 + 4 * lh;
                lds_f32* hrow = smem + W4_H_OFF + prow * W4_HP + ln;
                const bool hcol = ln < W4_HP;
                f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
                if (walk != 0 && hcol && !(DBG & 512)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) hacc[r] = hrow[((r & 3) + 8 * (r >> 2)) * W4_HP];
                }
                const int pxl = wave * 32 + ln;
                const lds_f32* trow = Tt + pxl * W4_N;
                const int swz = (pxl & 15) << 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 a4 = *(const lds_f32x4*)(trow + ((j * 8 + lh * 4) ^ swz));
                    const f32x4 b4 = *(const lds_f32x4*)(Wh + j * 256 + lane * 4);
                    if (DBG & 1024) { hacc[j] += a4.x * b4.x; continue; }
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, hacc, 0, 0, 0);
                }
                if (walk != p.tiles_n - 1) {
                    if (hcol) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) hrow[((r & 3) + 8 * (r >> 2)) * W4_HP] = hacc[r];
                    }
                } else {
                    const unsigned hbase = static_cast<unsigned>(mt * 512 + prow) * 128u + static_cast<unsigned>(ln) * 4u;
#pragma unroll
                    for (int r = 0; r < (DBG & 256 ? 1 : 16); ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hacc[r]), hp_rsrc,
                                                              static_cast<int>(hbase + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * 128u), 0, 0);
                }
                // T overlaps U buffer 1: the next tile's prologue must not start before every wave has read it
                if (g == 3) W4_EPILOGUE_BARRIER();
                if (g == 0 || g == 1) STAMP();  // after the heads block
            }
            if constexpr (CONV3) {
                // The Bottleneck's 1x1 expansion on the round's 128 pixels while they are on chip (model.py:203-209: conv3 +
                // bn3 + residual + ReLU): wave w takes pixels 32 w .. 32 w + 31 x all C3 output channels, 32 at a time:
                // A = the transposed tile T (k = the 64 channels of conv2), B = the W3 pieces resident in LDS, the same k
                // order and the same epilogue expression as the direct kernel runs for this layer (conv.hip /
                // conv_common.hpp): equal bit for bit. Only the residual loads and the stores are vector-memory traffic; the
                // residual of column block cb + 1 is in flight while block cb is multiplied and stored (one wave per SIMD:
                // nobody else covers its latency).
                const __amdgpu_buffer_rsrc_t r3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res3), 0, p.y3_bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t y3_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y3, 0, p.y3_bytes, 0x00020000);
                const int pxl = wave * 32 + ln;
                const lds_f32* trow = t_row(pxl);
                const int swz = (pxl & 15) << 2;
                const lds_f32* W3s = smem + W4_W3_OFF + lane * 4;
                // 8-byte accesses on channel pairs (conv_common.hpp, epilogue_pairs): lanes 2c / 2c + 1 hold channels n, n + 1 of
                // the same 16 rows; after one DPP swap per row pair the even lane owns both channels of rows 0..7 of the 16, the
                // odd lane both of rows 8..15 — half the residual loads and stores. The value arithmetic is unchanged.
                const int odd = ln & 1;
                unsigned rowoff[8];   // byte offset of the lane's k-th output pixel in y3 / the residual, or OOB
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int pr = wave * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh + 16 * odd;   // pixel of the round: position pr >> 4, (i, j)
                    const int TY = TY0 + g, TX = TX0 + (pr >> 4);
                    const bool ok = TY < p.TH && TX < p.TW;
                    const unsigned pix = static_cast<unsigned>((b * p.H + 4 * TY + ((pr >> 2) & 3)) * p.W + 4 * TX + (pr & 3));
                    rowoff[k] = ok ? pix * static_cast<unsigned>(p.C3) * 4u : OOB;
                }
                const int ncb = p.C3 >> 5;
                f32x4 afr[8];   // the wave's A fragments (its 32 pixels x all 64 k): read once per round, used by every column block
#pragma unroll
                for (int j = 0; j < 8; ++j) afr[j] = *(const lds_f32x4*)(trow + ((j * 8 + lh * 4) ^ swz));
                auto fetch_res = [&](int cb, float (&res)[16]) {
                    const unsigned ccol = static_cast<unsigned>(cb * 32 + (ln & ~1)) * 4u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const auto w = __builtin_amdgcn_raw_buffer_load_b64(r3_rsrc, static_cast<int>(oob_add(rowoff[k], ccol)), 0, 0);
                        res[2 * k] = __uint_as_float(w[0]);
                        res[2 * k + 1] = __uint_as_float(w[1]);
                    }
                };
                auto block = [&](int cb, const float (&res)[16], float (&res_next)[16]) {
                    const unsigned ccol = static_cast<unsigned>(cb * 32 + (ln & ~1)) * 4u;
                    const float sc3 = p.scale3 ? p.scale3[cb * 32 + ln] : 1.0f, sh3 = p.shift3 ? p.shift3[cb * 32 + ln] : 0.0f;
                    if (cb + 1 < ncb) fetch_res(cb + 1, res_next);
                    f32x16 hacc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
                    // the B fragment of step j + 1 is read while the four MFMAs of step j run (read-then-wait in front of
                    // every group of four left the LDS latency exposed eight times per column block)
                    f32x4 bq[2];
                    bq[0] = *(const lds_f32x4*)(W3s + (cb * 8) * 256);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4 a4 = afr[j];
                        if (j + 1 < 8) bq[(j + 1) & 1] = *(const lds_f32x4*)(W3s + (cb * 8 + j + 1) * 256);
                        __builtin_amdgcn_sched_barrier(0);
                        const f32x4 b4 = bq[j & 1];
                        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, hacc, 0, 0, 0);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, hacc, 0, 0, 0);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, hacc, 0, 0, 0);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, hacc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float va = hacc[k] * sc3 + sh3;          // own channel, row k
                        const float vb = hacc[k + 8] * sc3 + sh3;      // own channel, row k + 8
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? va : vb),
                                                                                                 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                        float lo = odd ? got : va, hi = odd ? vb : got;   // channels n & ~1, (n & ~1) + 1 of the row this lane keeps
                        lo += res[2 * k];
                        hi += res[2 * k + 1];
                        asm("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(lo), "+v"(hi));
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(lo), __float_as_uint(hi)}, y3_rsrc,
                                                              static_cast<int>(oob_add(rowoff[k], ccol)), 0, 0);
                    }
                };
                float r0[16], r1[16];
                fetch_res(0, r0);
                for (int cb = 0; cb < ncb; cb += 2) {
                    block(cb, r0, r1);
                    if (cb + 1 < ncb) block(cb + 1, r1, r0);
                }
                // T lies inside Z: the next round's Z writes (and, after the last round, the next tile's prologue) must not
                // start before every wave has read its rows
                W4_EPILOGUE_BARRIER();
            }
        }
        if constexpr ((DBG & 2048) != 0) {
            // plain: into the (second) output tensor; HEADS: over the shift vector (a tuning build: results are void anyway)
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(HEADS ? const_cast<float*>(p.shift) : p.yk);
            if (it == 1 && blockIdx.x == 0 && threadIdx.x == 0 && dbg)
                for (int i = 0; i < nstamp; ++i) dbg[i] = stamp[i];
        }
    }  // tiles
}

// DBG: timing ablations for tuning (wrong results): 1 no DMA, 2 no transform, 4 no patch reads, 8 no B reads, 16 no raw
// staging. Only DBG = 0 is built unless MRCNN_W4_ABLATIONS is defined.
template <int DBG, bool HEADS, bool ACT = true, bool CONV3 = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino4_f32(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds_f32* lds = (lds_f32*)smem;
    if (wave == 0) wino4_wave<0, 0, DBG, HEADS, ACT, CONV3>(p, lds);
    else if (wave == 1) wino4_wave<0, 1, DBG, HEADS, ACT, CONV3>(p, lds);
    else if (wave == 2) wino4_wave<1, 0, DBG, HEADS, ACT, CONV3>(p, lds);
    else wino4_wave<1, 1, DBG, HEADS, ACT, CONV3>(p, lds);
}

// G g G^T in double, stored as the kernel's [Cin/4][36][2][Cout][2]. G (6 x 3): row of point p = [1 p p^2] / N_p, N_p the
// product of (p - q) over the other finite points; last row [0 0 1]
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ w, int cout, int cin,
                                                            float* __restrict__ u) {
    const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (e >= static_cast<int64_t>(cout) * cin) return;
    const int n = static_cast<int>(e / cin), c = static_cast<int>(e - static_cast<int64_t>(n) * cin);
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((static_cast<int64_t>(n) * 3 + ky) * 3 + kx) * cin + c];
    auto g6 = [](double x0, double x1, double x2, double (&r)[6]) {
        const double a = W4_A, b = W4_B;
        const double pts[5] = {0.0, a, -a, b, -b};
        for (int j = 0; j < 5; ++j) {
            double nj = 1.0;
            for (int l = 0; l < 5; ++l)
                if (l != j) nj *= pts[j] - pts[l];
            r[j] = (x0 + pts[j] * x1 + pts[j] * pts[j] * x2) / nj;
        }
        r[5] = x2;
    };
    double t[6][3];
    for (int kx = 0; kx < 3; ++kx) {
        double r[6];
        g6(g[0][kx], g[1][kx], g[2][kx], r);
        for (int i = 0; i < 6; ++i) t[i][kx] = r[i];
    }
    const int kt = c >> 2, h = (c >> 1) & 1, s = c & 1;
    for (int i = 0; i < 6; ++i) {
        double r[6];
        g6(t[i][0], t[i][1], t[i][2], r);
        for (int j = 0; j < 6; ++j)
            u[((static_cast<int64_t>(kt) * 36 + (i * 6 + j)) * 2 + h) * cout * 2 + n * 2 + s] = static_cast<float>(r[j]);
    }
}

}  // namespace


extern "C" int mrcnn_winograd4_weights_f32(const float* w_ohwi, int32_t cout, int32_t cin, float* u,
                                           mrcnn_stream_t stream) {
    MRCNN_REQUIRE(w_ohwi && u, "winograd4_weights: null pointer");
    MRCNN_REQUIRE(cout >= 1 && cin >= 4 && cin % 4 == 0, "winograd4_weights: Cout=%d Cin=%d (Cin %% 4 == 0 required)", cout, cin);
    const int64_t e = static_cast<int64_t>(cout) * cin;
    hipLaunchKernelGGL(wino4_weights_kernel, dim3(static_cast<unsigned>((e + 255) / 256)), dim3(256), 0,
                       mrcnn::as_stream(stream), w_ohwi, cout, cin, u);
    return mrcnn::check_launch("wino4_weights_kernel");
}

// shapes the kernel takes, the 32-bit byte-offset limits included (B*H*W*Cin, B*H*W*Cout < 2^30 elements): a caller that
// asks first can fall back to the F(2x2) or the direct kernel instead of being refused by the launch
extern "C" int32_t mrcnn_conv3x3_winograd4_supported(int32_t batch, int32_t height, int32_t width, int32_t cin,
                                                     int32_t cout) {
    if (!(batch >= 1 && height >= 4 && width >= 4 && height % 4 == 0 && width % 4 == 0 && cin >= 8 && cin % 8 == 0 &&
          cout >= W4_N && cout % W4_N == 0))
        return 0;
    const long long px = 1LL * batch * height * width;
    return px * cin < (1LL << 30) && px * cout < (1LL << 30) && 36LL * cin * cout < (1LL << 30);
}

extern "C" int mrcnn_conv3x3_winograd4_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width,
                                           int32_t cin, const float* u, int32_t cout, const float* scale,
                                           const float* shift, int32_t activation, float* y_nhwc, float* y_kblocked,
                                           mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_kblocked && u && (y_nhwc || y_kblocked), "conv3x3_winograd4: null pointer");
    MRCNN_REQUIRE(batch >= 1 && mrcnn_conv3x3_winograd4_supported(batch, height, width, cin, cout),
                  "conv3x3_winograd4: B=%d H=%d W=%d (%% 4 == 0) Cin=%d (%% 8 == 0) Cout=%d (%% 64 == 0), B*H*W*C < 2^30 required",
                  batch, height, width, cin, cout);
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv3x3_winograd4: activation must be 0 or 1");
    const long long px = 1LL * batch * height * width;
    MRCNN_REQUIRE(px * cin < (1LL << 30) && px * cout < (1LL << 30) && 36LL * cin * cout < (1LL << 30),
                  "conv3x3_winograd4: tensor too large (32-bit buffer byte offsets)");
    Wino4Params p;
    p.x = x_kblocked; p.u = u; p.scale = scale; p.shift = shift; p.y = y_nhwc; p.yk = y_kblocked;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout;
    p.TH = height / 4; p.TW = width / 4; p.act = activation;
    p.tyb = (p.TH + 3) / 4; p.txb = (p.TW + 7) / 8;
    p.tiles_m = batch * p.tyb * p.txb;
    p.tiles_n = cout / W4_N;
    p.x_plane = static_cast<unsigned>(4LL * px * 8);
    p.u_ktile = static_cast<unsigned>(4LL * 36 * 4 * cout);
    p.yk_plane = static_cast<unsigned>(4LL * px * 8);
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.u_bytes = static_cast<unsigned>(4LL * 36 * cin * cout);
    p.y_bytes = static_cast<unsigned>(4LL * px * cout);
    p.w_head = nullptr; p.head_part = nullptr; p.head_bytes = 0;
    p.w3 = nullptr; p.scale3 = nullptr; p.shift3 = nullptr; p.res3 = nullptr; p.y3 = nullptr; p.C3 = 0; p.w3_bytes = 0; p.y3_bytes = 0;
    p.debug = 0;
#ifdef MRCNN_W4_ABLATIONS
    p.debug = getenv("MRCNN_W4_DEBUG") ? atoi(getenv("MRCNN_W4_DEBUG")) : 0;
#endif
    const long long grid = 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n;
    MRCNN_REQUIRE(grid <= 0x7fffffffLL, "conv3x3_winograd4: grid too large");
    void (*kern)(const Wino4Params) = activation ? conv3x3_wino4_f32<0, false, true> : conv3x3_wino4_f32<0, false, false>;
#ifdef MRCNN_W4_ABLATIONS
    switch (p.debug) {
        case 1: kern = conv3x3_wino4_f32<1, false>; break;
        case 2: kern = conv3x3_wino4_f32<2, false>; break;
        case 4: kern = conv3x3_wino4_f32<4, false>; break;
        case 8: kern = conv3x3_wino4_f32<8, false>; break;
        case 16: kern = conv3x3_wino4_f32<16, false>; break;
        case 31: kern = conv3x3_wino4_f32<31, false>; break;
        case 2048: kern = conv3x3_wino4_f32<2048, false>; break;
        case 4096: kern = conv3x3_wino4_f32<4096, false, false>; break;
        case 8192: kern = conv3x3_wino4_f32<8192, false, false>; break;
        case 12288: kern = conv3x3_wino4_f32<12288, false, false>; break;
        case 63: kern = conv3x3_wino4_f32<63, false>; break;
        case 95: kern = conv3x3_wino4_f32<95, false>; break;
        default: break;
    }
#endif
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), WINO4_LDS, "conv3x3_winograd4"))
        return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd4: cannot query the device");
    const int ncu = cus >= 8 ? (cus / 8) * 8 : 8;
    const long long launch = grid > ncu ? ncu : grid;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(launch)), dim3(256), WINO4_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("conv3x3_wino4_f32");
}

extern "C" int64_t mrcnn_conv3x3_winograd4_heads_rows(int32_t batch, int32_t height, int32_t width) {
    if (batch < 1 || height < 4 || width < 4 || height % 4 || width % 4) return 0;
    return static_cast<int64_t>(batch) * ((height / 4 + 3) / 4) * ((width / 4 + 7) / 8) * 512;
}

extern "C" int mrcnn_conv3x3_winograd4_heads_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width,
                                                 int32_t cin, const float* u, int32_t cout, const float* scale,
                                                 const float* shift, int32_t activation, const float* w_head32,
                                                 float* head_part, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_kblocked && u && w_head32 && head_part, "conv3x3_winograd4_heads: null pointer");
    MRCNN_REQUIRE(batch >= 1 && mrcnn_conv3x3_winograd4_supported(batch, height, width, cin, cout),
                  "conv3x3_winograd4_heads: B=%d H=%d W=%d (%% 4 == 0) Cin=%d (%% 8 == 0) Cout=%d (%% 64 == 0), B*H*W*C < 2^30 "
                  "required", batch, height, width, cin, cout);
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv3x3_winograd4_heads: activation must be 0 or 1");
    const long long px = 1LL * batch * height * width;
    const long long rows = mrcnn_conv3x3_winograd4_heads_rows(batch, height, width);
    MRCNN_REQUIRE(px * cin < (1LL << 30) && 36LL * cin * cout < (1LL << 30) && rows * 32 < (1LL << 30),
                  "conv3x3_winograd4_heads: tensor too large (32-bit buffer byte offsets)");
    Wino4Params p;
    p.x = x_kblocked; p.u = u; p.scale = scale; p.shift = shift; p.y = nullptr; p.yk = nullptr;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout;
    p.TH = height / 4; p.TW = width / 4; p.act = activation;
    p.tyb = (p.TH + 3) / 4; p.txb = (p.TW + 7) / 8;
    p.tiles_m = batch * p.tyb * p.txb;
    p.tiles_n = cout / W4_N;
    p.x_plane = static_cast<unsigned>(4LL * px * 8);
    p.u_ktile = static_cast<unsigned>(4LL * 36 * 4 * cout);
    p.yk_plane = 0;
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.u_bytes = static_cast<unsigned>(4LL * 36 * cin * cout);
    p.y_bytes = 0;
    p.w_head = w_head32; p.head_part = head_part;
    p.head_bytes = static_cast<unsigned>(4LL * rows * 32);
    p.w3 = nullptr; p.scale3 = nullptr; p.shift3 = nullptr; p.res3 = nullptr; p.y3 = nullptr; p.C3 = 0; p.w3_bytes = 0; p.y3_bytes = 0;
    p.debug = 0;
#ifdef MRCNN_W4_ABLATIONS
    p.debug = getenv("MRCNN_W4_DEBUG") ? atoi(getenv("MRCNN_W4_DEBUG")) : 0;
#endif
    void (*kern)(const Wino4Params) = activation ? conv3x3_wino4_f32<0, true, true> : conv3x3_wino4_f32<0, true, false>;
#ifdef MRCNN_W4_ABLATIONS
    switch (p.debug) {
        case 32: kern = conv3x3_wino4_f32<32, true>; break;
        case 128: kern = conv3x3_wino4_f32<128, true>; break;
        case 160: kern = conv3x3_wino4_f32<160, true>; break;
        case 256: kern = conv3x3_wino4_f32<256, true>; break;
        case 512: kern = conv3x3_wino4_f32<512, true>; break;
        case 1024: kern = conv3x3_wino4_f32<1024, true>; break;
        case 1792: kern = conv3x3_wino4_f32<1792, true>; break;
        case 2048: kern = conv3x3_wino4_f32<2048, true>; break;
        default: break;
    }
#endif
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), WINO4_HEADS_LDS, "conv3x3_winograd4_heads"))
        return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd4_heads: cannot query the device");
    const int ncu = cus >= 8 ? (cus / 8) * 8 : 8;
    const long long units = 8LL * ((p.tiles_m + 7) / 8);  // M-tile units; a workgroup walks the N tiles of its units
    const long long launch = units < ncu ? units : ncu;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(launch)), dim3(256), WINO4_HEADS_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("conv3x3_wino4_f32<heads>");
}

// conv2 (3x3, planes -> planes = 64) + BN + ReLU, then conv3 (1x1, 64 -> c3) + BN + residual + ReLU of a ResNet Bottleneck
// (model.py:197-209) in ONE launch: the 64-channel map between them never reaches HBM.
extern "C" int mrcnn_conv3x3_winograd4_conv3_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width,
                                                 int32_t cin, const float* u, const float* scale, const float* shift,
                                                 const float* w3, int32_t c3, const float* scale3, const float* shift3,
                                                 const float* residual, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_kblocked && u && w3 && residual && y, "conv3x3_winograd4_conv3: null pointer");
    MRCNN_REQUIRE(batch >= 1 && mrcnn_conv3x3_winograd4_supported(batch, height, width, cin, W4_N),
                  "conv3x3_winograd4_conv3: B=%d H=%d W=%d (%% 4 == 0) Cin=%d (%% 8 == 0), B*H*W*C < 2^30 required", batch,
                  height, width, cin);
    MRCNN_REQUIRE(c3 >= 32 && c3 % 32 == 0 && c3 <= 256, "conv3x3_winograd4_conv3: c3=%d must be a multiple of 32, at most 256", c3);
    MRCNN_REQUIRE(residual != y, "conv3x3_winograd4_conv3: in-place operation is not supported");
    const long long px = 1LL * batch * height * width;
    MRCNN_REQUIRE(px * c3 < (1LL << 30), "conv3x3_winograd4_conv3: tensor too large (32-bit buffer byte offsets)");
    Wino4Params p;
    p.x = x_kblocked; p.u = u; p.scale = scale; p.shift = shift; p.y = nullptr; p.yk = nullptr;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = W4_N;
    p.TH = height / 4; p.TW = width / 4; p.act = 1;
    p.tyb = (p.TH + 3) / 4; p.txb = (p.TW + 7) / 8;
    p.tiles_m = batch * p.tyb * p.txb;
    p.tiles_n = 1;
    p.x_plane = static_cast<unsigned>(4LL * px * 8);
    p.u_ktile = static_cast<unsigned>(4LL * 36 * 4 * W4_N);
    p.yk_plane = 0;
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.u_bytes = static_cast<unsigned>(4LL * 36 * cin * W4_N);
    p.y_bytes = 0;
    p.w_head = nullptr; p.head_part = nullptr; p.head_bytes = 0;
    p.w3 = w3; p.scale3 = scale3; p.shift3 = shift3; p.res3 = residual; p.y3 = y; p.C3 = c3;
    p.w3_bytes = static_cast<unsigned>(4LL * c3 * W4_N);
    p.y3_bytes = static_cast<unsigned>(4LL * px * c3);
    p.debug = 0;
    void (*kern)(const Wino4Params) = conv3x3_wino4_f32<0, false, true, true>;
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), WINO4_CONV3_LDS, "conv3x3_winograd4_conv3"))
        return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd4_conv3: cannot query the device");
    const int ncu = cus >= 8 ? (cus / 8) * 8 : 8;
    const long long grid = 8LL * ((p.tiles_m + 7) / 8);
    const long long launch = grid > ncu ? ncu : grid;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(launch)), dim3(256), WINO4_CONV3_LDS, mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("conv3x3_wino4_f32<conv3>");
}
