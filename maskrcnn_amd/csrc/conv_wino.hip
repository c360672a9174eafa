// 3x3 stride-1 SAME convolution + per-channel affine + ReLU as a fused Winograd F(2x2, 3x3) on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), channels-last. 2.25x fewer multiply-adds than the direct implicit GEMM of conv.hip for the
// layers that hold three quarters of the step's FLOPs (FPN smoothing model.py:154-157, RPN conv_shared :605,624,
// Bottleneck conv2 :182, the mask head's four 3x3 convs :880-903). All arithmetic is fp32; the result differs from
// the direct kernel by the transforms' rounding (a few 1e-7 relative per term), well inside the 1e-4 parity bar.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A       g: 3x3 filter, d: 4x4 input patch, Y: 2x2 outputs
//
//   GEMM view   16 independent products, one per transform component xi: M_xi[T x N] = V_xi[T x C] U_xi[C x N],
//               T = B*(H/2)*(W/2) tile positions, N = Cout, C = Cin. U is transformed once by wino_weights_kernel.
//   tile        a workgroup owns 64 consecutive tile positions x 64 output channels (128 KB of LDS: V and U, 16 component
//               planes each, double-buffered; one workgroup per CU). Two kernels share everything but the wave layout:
//               conv3x3_wino8_f32 (default): 512 threads, wave w owns components 2w, 2w+1 for the whole tile (128
//               accumulator registers, two waves per SIMD), persistent workgroups;
//               conv3x3_wino_f32 (MRCNN_WINO_WAVES=4): 256 threads, wave w owns components 4w..4w+3 (256 accumulator
//               registers in AGPRs, one wave per SIMD).
//   layouts     both operands are k-blocked: x as [Cin/8][B*H*W][8] (an NHWC input costs one kblock_kernel pass; producers
//               can write it directly), U as [Cin/8][16][Cout][8], so that a k tile's loads use whole cache lines.
//   staging     per k tile of 8 input channels a thread fetches a position's 4x4 patch for one channel pair (16 x 8-byte
//               loads, zero padding by descriptor range check), applies B^T . B with packed adds and writes 16 components
//               to LDS; the 16 x 64 x 8 block of U is copied through registers. No address arithmetic in the loop: fixed
//               per-thread voffsets, the k tile's plane as scalar soffset, immediate LDS offsets — VALU instructions do
//               not co-execute with this MFMA (tools/mfma_valu_probe.hip), every one of them costs MFMA time.
//               Double-buffered, one barrier per k tile; loads for k tile t+2 are in flight during the MFMAs of t+1.
//   MFMA        per component and k tile: two ds_read_b128 per operand (lane half h holds channels 4h..4h+3, the k
//               permutation of conv.hip) feed sixteen MFMAs.
//   epilogue    M A in registers, A^T (M A) across waves through LDS, affine, ReLU, 256-byte channel runs, NHWC and/or
//               k-blocked.
#include "conv_common.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace mrcnn_conv;

struct WinoParams {
    const float* x;      // k-blocked input  [Cin/8][B*H*W][8]
    const float* u;      // k-blocked filter [Cin/8][16][Cout][8]
    const float* scale;  // [Cout] or null
    const float* shift;  // [Cout] or null
    float* y;            // NHWC output [B][H][W][Cout], or null
    float* yk;           // k-blocked output [Cout/8][B*H*W][8], or null
    int B, H, W, Cin, Cout, TH, TW, T, act;
    unsigned x_plane, u_plane, yk_plane;  // bytes per 8-channel plane of x / u / the k-blocked output
    int tiles_m, tiles_n;
    unsigned x_bytes, u_bytes, y_bytes;
    // fused 1x1 heads (conv3x3_wino8_f32<true>): w_head [32][Cout] (rows >= the real head count are zero);
    // head_part [2 k halves][tiles_m * 256 pixel rows in position-major order][32] partial sums
    const float* w_head;
    float* head_part;
    unsigned head_bytes;
};

constexpr int WT = 64;   // tile positions per workgroup
constexpr int WN = 64;   // output channels per workgroup
constexpr int WK = 8;    // input channels per k tile
constexpr int PLANE = WT * WK;  // floats per component plane (V and U alike: WT == WN)
constexpr size_t WINO_LDS = sizeof(float) * 2 * 2 * 16 * PLANE;  // {V,U} x 2 buffers x 16 components = 128 KiB
constexpr size_t WINO_HEADS_LDS = WINO_LDS + sizeof(float) * 128 * WN;  // + the transposed tile of a round: 160 KiB

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Two-lane packed fp32 add / subtract: ONE VALU issue for both channels of the pair (the compiler scalarises a float2
// subtraction into two v_add_f32 with a negate modifier; on this MFMA every VALU instruction in the loop is paid for).
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

#ifdef MRCNN_ABLATIONS   // the four-wave kernel of round 1 (one wave per SIMD): an ablation build only (MRCNN_WINO_WAVES=4)
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                    // [2 buffers][16 components][2 quads][WT][4]
    float* Us = smem + 2 * 16 * PLANE;   // [2 buffers][16 components][2 quads][WN][4]

    ConvCommon tc;  // only the fields tile_origin reads
    tc.tiles_m = p.tiles_m;
    tc.tiles_n = p.tiles_n;
    int m0, n0, nt;
    if (!tile_origin(tc, WT, WN, m0, n0, nt)) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- staging bookkeeping. On this MFMA every VALU instruction costs its four issue cycles on top of the MFMA
    // time (measured: tools/mfma_valu_probe.hip — no co-execution, within a wave or across waves), so the loop
    // carries no address arithmetic at all: per-thread byte offsets are fixed for the whole tile (voffset, out of
    // range where the load must return zeros), the k tile's plane is the scalar soffset, LDS addresses are one base
    // register plus instruction-immediate offsets, and the transform is spread evenly over all four waves.
    //   V: thread = (position tid >> 2, channel pair tid & 3): sixteen 8-byte patch loads; a wave-level load covers 16
    //      pixels x 32 contiguous bytes = 8 cache lines (a position per lane would touch 32)
    //   U: eight 16-byte slots of the 16 x 64 x 8 block
    unsigned voff[16], uoff[8];
    {
        const int pos = tid >> 2, pair = tid & 3;  // four lanes = the 32 contiguous bytes of one pixel's k tile
        const int P = m0 + pos;
        const bool pv = P < p.T;
        const int PP = pv ? P : 0;
        const int b = PP / (p.TH * p.TW), rem = PP - b * p.TH * p.TW;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int iy = iy0 + dy, ix = ix0 + dx;
                const bool ok = pv && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                voff[dy * 4 + dx] = ok ? static_cast<unsigned>(((b * p.H + iy) * p.W + ix) * WK + pair * 2) * 4u : OOB;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + 256 * i;  // float4 id inside the 16 x 64 x 8 block
            const int xi = id >> 7, r = id & 127, q = r >> 6, ch = r & 63;
            const int n = n0 + ch;
            uoff[i] = n < p.Cout ? static_cast<unsigned>((xi * p.Cout + n) * WK + q * 4) * 4u : OOB;
        }
    }
    // LDS float offsets inside a buffer: V component xi at v_lds + xi*PLANE; U slot i at u_lds + i*2*PLANE
    // (id = tid + 256*i: xi = 2i + (tid >> 7), so consecutive slots are two planes apart)
    const int v_lds = ((tid & 3) >> 1) * (WT * 4) + (tid >> 2) * 4 + (tid & 1) * 2;
    const int u_lds = (tid >> 7) * PLANE + ((tid & 127) >> 6) * (WN * 4) + (tid & 63) * 4;

    const int nk = (p.Cin + WK - 1) / WK;
    f32x2 vl[16];  // raw patch of the k tile in flight
    u32x4 ul[8];
    auto plane_of = [&](int kt) { return static_cast<unsigned>(kt < nk ? kt : nk - 1); };  // past the end: reload, unused
    auto v_load = [&](int i, int kt) {
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, static_cast<int>(voff[i]),
                                                             static_cast<int>(plane_of(kt) * p.x_plane), 0);
        vl[i] = f32x2{__uint_as_float(r.x), __uint_as_float(r.y)};
    };
    auto u_load = [&](int i, int kt) {
        ul[i] = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, static_cast<int>(uoff[i]),
                                                      static_cast<int>(plane_of(kt) * p.u_plane), 0);
    };
    // V = B^T d B on two channels at a time, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: first down the columns
    // (t), then along the rows (one output component at a time)
    f32x2 t[4][4];
    auto column_pass = [&](int dx) {
        const f32x2 d0 = vl[0 * 4 + dx], d1 = vl[1 * 4 + dx], d2 = vl[2 * 4 + dx], d3 = vl[3 * 4 + dx];
        t[0][dx] = pk_sub(d0, d2);
        t[1][dx] = pk_add(d1, d2);
        t[2][dx] = pk_sub(d2, d1);
        t[3][dx] = pk_sub(d1, d3);
    };
    auto v_comp = [&](int xi) {
        const int i = xi >> 2, j = xi & 3;
        return j == 0 ? pk_sub(t[i][0], t[i][2]) : j == 1 ? pk_add(t[i][1], t[i][2])
             : j == 2 ? pk_sub(t[i][2], t[i][1]) : pk_sub(t[i][1], t[i][3]);
    };
    auto v_store = [&](int xi, float* vbuf) { *reinterpret_cast<f32x2*>(vbuf + xi * PLANE) = v_comp(xi); };
    f32x2 vv[16];   // main loop: the 16 components, computed in one go (see the slot comment there)
    auto u_store = [&](int i, float* ubuf) { *reinterpret_cast<u32x4*>(ubuf + i * 2 * PLANE) = ul[i]; };

    // wave w owns transform row i = w: components xi = 4w + j, j = 0..3, for the WHOLE 64 x 64 workgroup tile
    // (2 x 2 MFMA tiles per component: an operand fragment is reused twice, 0.25 LDS reads per MFMA as in conv.hip)
    f32x16 acc[4][2][2];  // [j][position half][channel half]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][a][c][r] = 0.f;

    // prologue: k tile 0 into buffer 0, loads of k tile 1 in flight
#pragma unroll
    for (int i = 0; i < 16; ++i) v_load(i, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) u_load(i, 0);
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) column_pass(dx);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) v_store(xi, Vs + v_lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) u_store(i, Us + u_lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) v_load(i, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) u_load(i, 1);
    __syncthreads();

    // a component plane is [2 channel quads][64 rows][4 channels]: staging writes (a lane per row) and operand reads
    // (a lane per row, lane half = quad) are both contiguous 16-byte runs across lanes — no LDS bank conflicts
    const float* Aw = Vs + (wave * 4) * PLANE + lh * (WT * 4) + ln * 4;
    const float* Bw = Us + (wave * 4) * PLANE + lh * (WN * 4) + ln * 4;
    float4 fa[2][2], fb[2][2];  // [slot][half]
    auto read_frags = [&](int slot, int buf, int j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fa[slot][h] = *reinterpret_cast<const float4*>(Aw + buf * 16 * PLANE + j * PLANE + h * 32 * 4);
            fb[slot][h] = *reinterpret_cast<const float4*>(Bw + buf * 16 * PLANE + j * PLANE + h * 32 * 4);
        }
    };
    read_frags(0, 0, 0);
    // ---- main loop: 64 MFMAs per k tile and wave (4 components x 2 x 2 tiles x 4 k steps), order pinned. Staging of
    // k tile kt+1 (its loads were issued during k tile kt-1) is spread over MFMA slots 8..39 of every wave; the barrier
    // sits before the last component, whose operands are already in registers, so that the first operands of k tile
    // kt+1 are fetched behind it, not after it.
    //   slot 8-11 column passes | 12-27 reissue the patch loads for k tile kt+2 | 16-31 one V component out each
    //   slot 32-39 one U slot to LDS + reissue its load
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        float* vnext = Vs + (buf ^ 1) * 16 * PLANE + v_lds;
        float* unext = Us + (buf ^ 1) * 16 * PLANE + u_lds;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            if (j + 1 < 4) {
                read_frags(cur ^ 1, buf, j + 1);
            } else {
                __syncthreads();  // k tile kt+1 is complete in buf^1; everyone is done reading buf
                read_frags(cur ^ 1, buf ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float4 fav = fa[cur][a], fbv = fb[cur][c];
                        const float av = st == 0 ? fav.x : st == 1 ? fav.y : st == 2 ? fav.z : fav.w;
                        const float bv = st == 0 ? fbv.x : st == 1 ? fbv.y : st == 2 ? fbv.z : fbv.w;
                        acc[j][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j][a][c], 0, 0, 0);
                        const int m = j * 16 + st * 4 + a * 2 + c;  // MFMA slot 0..63
                        if (m >= 8 && m < 12) column_pass(m - 8);
                        if (m >= 12 && m < 28) v_load(m - 12, kt + 2);
                        if (m >= 16 && m < 32) v_store(m - 16, vnext);
                        if (m >= 32 && m < 40) {
                            u_store(m - 32, unext);
                            u_load(m - 32, kt + 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    // ---- epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1] ---------------------------------------------------------
    // The wave holds row i = wave of M (4 columns j): the column combination (M A) happens in registers, the row
    // combination crosses waves through LDS: Z[i][c][position][channel], 4 x 2 x 64 x 64 floats = the 128 KB the
    // operand buffers occupied. Then every wave finishes 16 positions: affine, ReLU, 256-byte channel runs.
    __syncthreads();  // (the behind-the-barrier operand prefetch of a k tile that does not exist is still in flight)
    // Z[i][c][position / 4][channel][position % 4]: an accumulator's registers r..r+3 are four consecutive positions
    // of one channel, so writer and reader both move 16 bytes per lane, contiguous across lanes
    float* Z = smem;
    constexpr int ZP = (WT / 4) * WN * 4;  // floats per (i, c) plane
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pq = a * 8 + 2 * rq + lh, ch = c * 32 + ln;  // positions 4*pq .. 4*pq+3
                float4 z0, z1;
                float* z0p = &z0.x;
                float* z1p = &z1.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rq * 4 + e;
                    const float m0v = acc[0][a][c][r], m1v = acc[1][a][c][r], m2v = acc[2][a][c][r], m3v = acc[3][a][c][r];
                    z0p[e] = m0v + m1v + m2v;
                    z1p[e] = m1v - m2v - m3v;
                }
                *reinterpret_cast<float4*>(Z + (wave * 2 + 0) * ZP + (pq * WN + ch) * 4) = z0;
                *reinterpret_cast<float4*>(Z + (wave * 2 + 1) * ZP + (pq * WN + ch) * 4) = z1;
            }
    __syncthreads();
    const int ch = lane, n = n0 + ch;
    const bool n_ok = n < p.Cout;
    const float sc = (n_ok && p.scale) ? p.scale[n] : 1.0f, sh = (n_ok && p.shift) ? p.shift[n] : 0.0f;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const unsigned ncol = n_ok ? static_cast<unsigned>(n) * 4u : OOB;
    const __amdgpu_buffer_rsrc_t yk_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.yk, 0, p.y_bytes, 0x00020000);
    const unsigned kcol = static_cast<unsigned>(n >> 3) * p.yk_plane + static_cast<unsigned>(n & 7) * 4u;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {  // this wave finishes positions 16*wave + 4*g .. +3
        const int pq = wave * 4 + g;
        float4 z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) z[i][c] = *reinterpret_cast<const float4*>(Z + (i * 2 + c) * ZP + (pq * WN + ch) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int P = __builtin_amdgcn_readfirstlane(m0 + pq * 4 + e);
            const bool pv = P < p.T;
            const int PP = pv ? P : 0;
            const int b = PP / (p.TH * p.TW), rem = PP - b * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float z0 = (&z[0][c].x)[e], z1 = (&z[1][c].x)[e], z2 = (&z[2][c].x)[e], z3 = (&z[3][c].x)[e];
                const float yv[2] = {z0 + z1 + z2, z1 - z2 - z3};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v = yv[a] * sc + sh;
                    if (p.act) v = v > 0.f ? v : 0.f;
                    const unsigned px = static_cast<unsigned>((b * p.H + 2 * ty + a) * p.W + 2 * tx + c);
                    if (p.y) {
                        const unsigned o = (pv && n_ok) ? px * (static_cast<unsigned>(p.Cout) * 4u) + ncol : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(o), 0, 0);
                    }
                    if (p.yk) {  // the layout the next Winograd conv reads: no transposition pass in between
                        const unsigned o = (pv && n_ok) ? kcol + px * 32u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yk_rsrc, static_cast<int>(o), 0, 0);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
#endif  // MRCNN_ABLATIONS

// Eight-wave variant: the same workgroup tile (64 positions x 64 channels, 128 KB of LDS) and k-blocked operands, but
// 512 threads — wave w owns TWO components (2w, 2w+1) for the whole tile: 128 accumulator registers, so TWO waves per
// SIMD. With one wave per SIMD every LDS / VMEM / scalar instruction and every wait sits on the MFMA critical path
// (there is nobody else to issue); with two, one wave's staging and stalls overlap the other's MFMAs (VALU still does
// not: tools/mfma_valu_probe.hip). Waves 0-3 stage V (patch loads + input transform), waves 4-7 stage U; wave w and
// w+4 share a SIMD, so each SIMD carries one of each. The output transform needs all 16 components of a position:
// M.A is split into partial sums per wave (columns j = 0,1 or 2,3), exchanged through LDS in two rounds of 32
// positions (8 waves x 2 partials x 32 x 64 floats = the 128 KB the operand buffers occupied).
// HEADS = true (the RPN's conv_shared, model.py:605-607,624-641): the ReLU'd 64-channel output tile is not stored;
// it is transposed through LDS and multiplied by the [32][Cout] weights of both 1x1 heads while on chip. A workgroup
// then owns whole M tiles and walks their N tiles itself, adding each N tile's contribution to the M tile's
// [256 pixels][32] head sums (a read-modify-write of 32 KB that stays in its XCD's L2; waves w and w + 4 split K and
// keep separate sums). The 512-channel activation (1.43 GB written and read back per step) never reaches HBM.
template <bool HEADS>
__global__ __launch_bounds__(512, 1) void conv3x3_wino8_f32(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;
    float* Us = smem + 2 * 16 * PLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Persistent workgroups (p.persistent: one per CU, gridDim a multiple of 8): workgroup b takes the tiles b,
    // b + gridDim, ... of the XCD-aware virtual grid, so it stays on its XCD and the CUs of an XCD work on neighbouring
    // tiles at any time; this saves a workgroup launch (128 KB of LDS, eight waves) per tile.
    const int total_tiles = 8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
    for (int it = 0;; ++it) {
    int m0, n0, nt;
    {
        // plain: virtual tile b, b + grid, ... in the XCD-aware (M tile, N tile) order. HEADS: M-tile units b, b + grid,
        // ..., each walked over all its N tiles by this workgroup
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
        const int mt = mt_lo + seq / p.tiles_n;
        if (mt >= mt_hi) continue;  // uniform
        m0 = mt * WT;
        nt = seq % p.tiles_n;
        n0 = nt * WN;
    }
    const int ln = lane & 31, lh = lane >> 5;
    const bool v_role = wave < 4;
    const int st = tid & 255;  // index inside the staging group (V: waves 0-3, U: waves 4-7)

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

    // fixed per-thread byte offsets (see the four-wave kernel): V role 16 patch taps, U role 8 slots (in off[0..7])
    unsigned off[16];
    if (v_role) {
        const int pos = st >> 2, pair = st & 3;
        const int P = m0 + pos;
        const bool pv = P < p.T;
        const int PP = pv ? P : 0;
        const int b = PP / (p.TH * p.TW), rem = PP - b * p.TH * p.TW;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int iy = iy0 + dy, ix = ix0 + dx;
                const bool ok = pv && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                                static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
                off[dy * 4 + dx] = ok ? static_cast<unsigned>(((b * p.H + iy) * p.W + ix) * WK + pair * 2) * 4u : OOB;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = st + 256 * i;
            const int xi = id >> 7, r = id & 127, q = r >> 6, ch = r & 63;
            const int n = n0 + ch;
            off[i] = n < p.Cout ? static_cast<unsigned>((xi * p.Cout + n) * WK + q * 4) * 4u : OOB;
        }
#pragma unroll
        for (int i = 8; i < 16; ++i) off[i] = OOB;
    }
    const int v_lds = ((st & 3) >> 1) * (WT * 4) + (st >> 2) * 4 + (st & 1) * 2;
    const int u_lds = (st >> 7) * PLANE + ((st & 127) >> 6) * (WN * 4) + (st & 63) * 4;

    const int nk = (p.Cin + WK - 1) / WK;
    auto plane_of = [&](int kt) { return static_cast<unsigned>(kt < nk ? kt : nk - 1); };
    // one register file for both roles: V uses ld as 16 x (2 floats), U as 8 x (4 dwords)
    u32x2 ld[16];
    auto v_load = [&](int i, int kt) {
        ld[i] = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, static_cast<int>(off[i]),
                                                     static_cast<int>(plane_of(kt) * p.x_plane), 0);
    };
    auto u_load = [&](int i, int kt) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, static_cast<int>(off[i]),
                                                              static_cast<int>(plane_of(kt) * p.u_plane), 0);
        ld[2 * i] = u32x2{r.x, r.y};
        ld[2 * i + 1] = u32x2{r.z, r.w};
    };
    auto f2 = [](u32x2 v) { return f32x2{__uint_as_float(v.x), __uint_as_float(v.y)}; };
    f32x2 t[4][4];
    auto column_pass = [&](int dx) {
        const f32x2 d0 = f2(ld[0 * 4 + dx]), d1 = f2(ld[1 * 4 + dx]), d2 = f2(ld[2 * 4 + dx]), d3 = f2(ld[3 * 4 + dx]);
        t[0][dx] = pk_sub(d0, d2);
        t[1][dx] = pk_add(d1, d2);
        t[2][dx] = pk_sub(d2, d1);
        t[3][dx] = pk_sub(d1, d3);
    };
    auto v_comp = [&](int xi) {
        const int i = xi >> 2, j = xi & 3;
        return j == 0 ? pk_sub(t[i][0], t[i][2]) : j == 1 ? pk_add(t[i][1], t[i][2])
             : j == 2 ? pk_sub(t[i][2], t[i][1]) : pk_sub(t[i][1], t[i][3]);
    };
    auto v_store = [&](int xi, float* vbuf) { *reinterpret_cast<f32x2*>(vbuf + xi * PLANE) = v_comp(xi); };
    f32x2 vv[16];   // main loop: the 16 components, computed in one go (see the slot comment there)
    auto u_store = [&](int i, float* ubuf) {
        *reinterpret_cast<u32x4*>(ubuf + i * 2 * PLANE) = u32x4{ld[2 * i].x, ld[2 * i].y, ld[2 * i + 1].x, ld[2 * i + 1].y};
    };

    f32x16 acc[2][2][2];  // [component of the pair][position half][channel half]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][a][c][r] = 0.f;

    // prologue: k tile 0 into buffer 0, loads of k tile 1 in flight
    if (v_role) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v_load(i, 0);
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) column_pass(dx);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) v_store(xi, Vs + v_lds);
#pragma unroll
        for (int i = 0; i < 16; ++i) v_load(i, 1);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) u_load(i, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) u_store(i, Us + u_lds);
#pragma unroll
        for (int i = 0; i < 8; ++i) u_load(i, 1);
    }
    __syncthreads();

    const float* Aw = Vs + (wave * 2) * PLANE + lh * (WT * 4) + ln * 4;
    const float* Bw = Us + (wave * 2) * PLANE + lh * (WN * 4) + ln * 4;
    float4 fa[2], fb[2];  // operand fragments of the component being multiplied (the partner wave covers their latency)
    auto read_frags = [&](int buf, int j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fa[h] = *reinterpret_cast<const float4*>(Aw + buf * 16 * PLANE + j * PLANE + h * 32 * 4);
            fb[h] = *reinterpret_cast<const float4*>(Bw + buf * 16 * PLANE + j * PLANE + h * 32 * 4);
        }
    };
    // main loop: 32 MFMAs per k tile and wave. Staging of k tile kt+1 is spread over slots 0..23 (V: 4 column passes,
    // 16 loads for kt+2, 16 component stores; U: 8 stores + 8 loads), one barrier at the end of the k tile — the stalls
    // (operand reads, the barrier) are covered by the other wave of the SIMD.
    // the two staging roles run separate copies of the loop (wave-uniform, decided once): no per-slot branches, and each
    // copy gets its own register allocation
    auto main_loop = [&](auto role) {
        constexpr bool VROLE = decltype(role)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            float* vnext = Vs + (buf ^ 1) * 16 * PLANE + v_lds;
            float* unext = Us + (buf ^ 1) * 16 * PLANE + u_lds;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                read_frags(buf, j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int stp = 0; stp < 4; ++stp) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float4 fav = fa[a], fbv = fb[c];
                            const float av = stp == 0 ? fav.x : stp == 1 ? fav.y : stp == 2 ? fav.z : fav.w;
                            const float bv = stp == 0 ? fbv.x : stp == 1 ? fbv.y : stp == 2 ? fbv.z : fbv.w;
                            acc[j][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j][a][c], 0, 0, 0);
                            const int m = j * 16 + stp * 4 + a * 2 + c;  // MFMA slot 0..31
                            if constexpr (VROLE) {
                                // the whole transform (32 packed adds) behind ONE MFMA: every switch between the MFMA
                                // stream and VALU work costs (measured on the F(4x4) kernel: -3 %); the LDS writes and
                                // the loads stay one per slot
                                if (m == 0) {
#pragma unroll
                                    for (int dx = 0; dx < 4; ++dx) column_pass(dx);
#pragma unroll
                                    for (int xi = 0; xi < 16; ++xi) vv[xi] = v_comp(xi);
                                }
                                if (m >= 4 && m < 20) v_load(m - 4, kt + 2);
                                if (m >= 8 && m < 24) *reinterpret_cast<f32x2*>(vnext + (m - 8) * PLANE) = vv[m - 8];
                            } else {
                                if (m >= 4 && m < 12) {
                                    u_store(m - 4, unext);
                                    u_load(m - 4, kt + 2);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            __syncthreads();  // k tile kt+1 complete in buf^1, everyone done reading buf
        }
    };
    if (v_role) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    // ---- epilogue: wave = (i = wave >> 1, jh = wave & 1) holds M_i,2jh and M_i,2jh+1 ------------------------------
    //   (M A)_i0 = M_i0 + M_i1 + M_i2,  (M A)_i1 = M_i1 - M_i2 - M_i3:  jh = 0 contributes (M_i0 + M_i1,  M_i1),
    //                                                                    jh = 1 contributes (M_i2, -M_i2 - M_i3)
    const int n = n0 + (tid & 63);
    const bool n_ok = n < p.Cout;
    const float sc = (n_ok && p.scale) ? p.scale[n] : 1.0f, sh = (n_ok && p.shift) ? p.shift[n] : 0.0f;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yk_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.yk, 0, p.y_bytes, 0x00020000);
    const unsigned ncol = n_ok ? static_cast<unsigned>(n) * 4u : OOB;
    const unsigned kcol = static_cast<unsigned>(n >> 3) * p.yk_plane + static_cast<unsigned>(n & 7) * 4u;
    float* Z = smem;  // [wave][2 partials][8 position quads][64 channels][4 positions]
    constexpr int ZQ = 8 * WN * 4;  // floats per (wave, partial) plane
    const int jh = wave & 1;
    // HEADS: the round's 128 pixels x 64 channels, transposed for the head MFMA; 16-byte chunks XOR-swizzled by the pixel
    // (a 64-float pitch would put every row on the same banks), exactly the 32 KB above the operand buffers
    float* Tt = smem + 2 * 2 * 16 * PLANE;
    const int hrt = wave & 3, hkh = wave >> 2;  // head MFMA: pixel rows 32 hrt .. +31 of the round, channels 32 hkh .. +31
    float4 wh[4];
    // opaque copies of the lane coordinates: everything the heads epilogue derives from them (LDS and global offsets,
    // the head-weight loads) stays behind the main loop — hoisted out of the tile loop those ~25 values spill inside it
    int ln_e = ln, lh_e = lh, tid_e = tid;
    if constexpr (HEADS) {
        asm volatile("" : "+v"(ln_e), "+v"(lh_e), "+v"(tid_e));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wh[j] = *reinterpret_cast<const float4*>(p.w_head + static_cast<int64_t>(ln_e) * p.Cout + n0 + hkh * 32 + j * 8 + lh_e * 4);
    }
    const __amdgpu_buffer_rsrc_t hp_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.head_part, 0, HEADS ? p.head_bytes : 0u, 0x00020000);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {  // position half
        __syncthreads();  // operand buffers (or the previous round's Z) are no longer read
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pq = 2 * rq + lh, ch = c * 32 + ln;
                float4 zp, zq;
                float* zpp = &zp.x;
                float* zqp = &zq.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rq * 4 + e;
                    const float ma = h == 0 ? acc[0][0][c][r] : acc[0][1][c][r];
                    const float mb = h == 0 ? acc[1][0][c][r] : acc[1][1][c][r];
                    zpp[e] = jh == 0 ? ma + mb : ma;
                    zqp[e] = jh == 0 ? mb : -ma - mb;
                }
                *reinterpret_cast<float4*>(Z + (wave * 2 + 0) * ZQ + (pq * WN + ch) * 4) = zp;
                *reinterpret_cast<float4*>(Z + (wave * 2 + 1) * ZQ + (pq * WN + ch) * 4) = zq;
            }
        __syncthreads();
        // 512 threads finish 32 positions: thread = (channel tid & 63, position quad tid >> 6)
        const int pq = tid >> 6, ch = tid & 63;
        float4 y0[4], y1[4];  // (M A)_i0, (M A)_i1 for i = 0..3, four positions each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 p0 = *reinterpret_cast<const float4*>(Z + ((2 * i) * 2 + 0) * ZQ + (pq * WN + ch) * 4);
            const float4 p1 = *reinterpret_cast<const float4*>(Z + ((2 * i + 1) * 2 + 0) * ZQ + (pq * WN + ch) * 4);
            const float4 q0 = *reinterpret_cast<const float4*>(Z + ((2 * i) * 2 + 1) * ZQ + (pq * WN + ch) * 4);
            const float4 q1 = *reinterpret_cast<const float4*>(Z + ((2 * i + 1) * 2 + 1) * ZQ + (pq * WN + ch) * 4);
            y0[i] = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
            y1[i] = make_float4(q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, q0.w + q1.w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int P = __builtin_amdgcn_readfirstlane(m0 + h * 32 + pq * 4 + e);
            const bool pv = P < p.T;
            const int PP = pv ? P : 0;
            const int b = PP / (p.TH * p.TW), rem = PP - b * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float z0 = c == 0 ? (&y0[0].x)[e] : (&y1[0].x)[e], z1 = c == 0 ? (&y0[1].x)[e] : (&y1[1].x)[e];
                const float z2 = c == 0 ? (&y0[2].x)[e] : (&y1[2].x)[e], z3 = c == 0 ? (&y0[3].x)[e] : (&y1[3].x)[e];
                const float yv[2] = {z0 + z1 + z2, z1 - z2 - z3};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v = yv[a] * sc + sh;
                    if (p.act) v = v > 0.f ? v : 0.f;
                    if constexpr (HEADS) {
                        const int pxl = ((tid_e >> 6) * 4 + e) * 4 + a * 2 + c;  // position-major pixel of the round
                        Tt[pxl * WN + ((tid_e & 63) ^ ((pxl & 15) << 2))] = v;
                        continue;
                    }
                    const unsigned px = static_cast<unsigned>((b * p.H + 2 * ty + a) * p.W + 2 * tx + c);
                    if (p.y) {
                        const unsigned o = (pv && n_ok) ? px * (static_cast<unsigned>(p.Cout) * 4u) + ncol : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(o), 0, 0);
                    }
                    if (p.yk) {
                        const unsigned o = (pv && n_ok) ? kcol + px * 32u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yk_rsrc, static_cast<int>(o), 0, 0);
                    }
                }
            }
        }
        if constexpr (HEADS) {
            // [128 pixels x 32 channels of this wave] x [32 channels x 32 head outputs] on top of the M tile's running
            // sums (N tile 0 starts them): 16 MFMAs per wave and round
            __syncthreads();
            const int row0 = (m0 * 4 + h * 128 + hrt * 32 + 4 * lh_e);  // + (r & 3) + 8 (r >> 2)
            const unsigned hbase = (static_cast<unsigned>(hkh) * static_cast<unsigned>(p.tiles_m) * 256u + static_cast<unsigned>(row0)) * 128u +
                                   static_cast<unsigned>(ln_e) * 4u;
            f32x16 hacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hacc[r] = 0.f;
                if (nt != 0)
                    hacc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        hp_rsrc, static_cast<int>(hbase + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * 128u), 0, 0));
            }
            const int pxl = hrt * 32 + ln_e;
            const float* trow = Tt + pxl * WN;
            const int swz = (pxl & 15) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a4 = *reinterpret_cast<const float4*>(trow + ((hkh * 32 + j * 8 + lh_e * 4) ^ swz));
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, wh[j].x, hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, wh[j].y, hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, wh[j].z, hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, wh[j].w, hacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hacc[r]), hp_rsrc,
                                                      static_cast<int>(hbase + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * 128u), 0, 0);
        }
    }
    __syncthreads();  // Z is read out: the next tile may overwrite the operand buffers
    }  // tiles
}

// ---------------------------------------------------------------------------------------------------------------
// Spatial-tile variant (round 2): the same 64 positions x 64 channels x 16 components per workgroup, eight waves, two
// components per wave — but the 64 positions are an 8 x 8 block of ONE image (16 x 16 output pixels), and the
// transformed input V is never staged: two thirds of the linear-tile kernel's patch loads fetch pixels that neighbouring
// positions fetch again (64 x 16 patch pixels for an 18 x 18 = 324-pixel footprint).
//   staging     per k tile of 8 channels the RAW 18 x 18 x 8 input region (10 KB: at most two 16-byte loads per thread)
//               and the 32 KB block of U (four per thread) go through registers into LDS, double-buffered: 6 loads per
//               thread and k tile instead of 16 (V role) / 8 (U role).
//   transform   an MFMA A-operand register is one position x one channel per lane, so every lane transforms its OWN
//               position out of the raw region: its wave's two components need 2 rows x 3 columns of the 4 x 4 patch —
//               six 16-byte LDS reads (the lane half picks the channel quad) and 10 packed adds per 16 MFMAs. The pixel
//               pitch is 12 floats: the reads are 2-way bank-conflicted at best (positions sit on even pixels).
//   pipeline    order pinned as in the kernels above; see the slot table at the main loop.
//   everything else (U layout, MFMA order, accumulators, A^T M A rounds, stores) is the linear kernel's, so the result is
//   bit-identical to it.
constexpr int RPIX = 12;                       // floats per raw pixel in LDS (8 channels + pad)
constexpr int RROW = 18 * RPIX;                // floats per raw row
constexpr int RAW_FLOATS = 18 * RROW + 8;      // one buffer (3896 floats, 16-byte multiple)
constexpr size_t WINOS_LDS = WINO_LDS;         // the epilogue's Z exchange (128 KiB) is the high-water mark

struct WinoSParams : WinoParams {
    int tyb, txb;  // 8 x 8-position blocks per image along y / x
};

// HEADS as in conv3x3_wino8_f32<true>: the output tile feeds the 1x1 heads on chip; a workgroup owns whole M tiles (8 x 8
// position blocks) and walks their N tiles; head sums in M-tile-major rows: row = mt * 256 + (py * 8 + px) * 4 + a * 2 + c.
template <bool HEADS>
__global__ __launch_bounds__(512, 1) void conv3x3_wino8s_f32(const WinoSParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Rs = smem;                         // [2][RAW_FLOATS]
    float* Us = smem + 2 * RAW_FLOATS;        // [2][16][2 quads][64][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
    const int nk = (p.Cin + WK - 1) / WK;
    const int total_tiles = 8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
    for (int it = 0;; ++it) {
        int mt, n0, nt;
        {
            // plain: virtual tile b, b + grid, ... in the XCD-aware (M tile, N tile) order. HEADS: M-tile units b, b + grid,
            // ..., each walked over all its N tiles by this workgroup
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
            mt = mt_lo + seq / p.tiles_n;
            if (mt >= mt_hi) continue;  // uniform
            nt = seq % p.tiles_n;
            n0 = nt * WN;
        }
        const int per_img = p.tyb * p.txb;
        const int b = mt / per_img, trem = mt - b * per_img;
        const int tyb = trem / p.txb, txb = trem - tyb * p.txb;
        const int ty0 = tyb * 8, tx0 = txb * 8;          // first tile position of the block
        const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;  // first raw input pixel (may be -1: zero padding)

        // ---- staging offsets: raw float4 ids tid and tid + 512 (648 in all: 324 pixels x 2 channel quads), U slots
        unsigned r_off[2], u_off[4];
        int r_lds[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 512 * i, px = id >> 1, q = id & 1;
            const int hy = (px * 57) >> 10, hx = px - hy * 18;  // px / 18 for px < 512
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = id < 648 && static_cast<unsigned>(iy) < static_cast<unsigned>(p.H) &&
                            static_cast<unsigned>(ix) < static_cast<unsigned>(p.W);
            r_off[i] = ok ? static_cast<unsigned>(((b * p.H + iy) * p.W + ix) * WK + q * 4) * 4u : OOB;
            r_lds[i] = hy * RROW + hx * RPIX + q * 4;
        }
        const bool r1 = tid + 512 < 648;  // wave-uniform for waves 0 and 1, false for waves 3-7; wave 2 mixed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 512 * i;
            const int xi = id >> 7, r = id & 127, q = r >> 6, ch = r & 63;
            const int n = n0 + ch;
            u_off[i] = n < p.Cout ? static_cast<unsigned>((xi * p.Cout + n) * WK + q * 4) * 4u : OOB;
        }
        auto plane_of = [&](int kt) { return static_cast<unsigned>(kt < nk ? kt : nk - 1); };
        u32x4 rr[2], ru[4];
        // pieces of a k tile: 0,1 = raw region, 2..5 = U
        auto load_piece = [&](int pc, int kt) {
            const unsigned pl = plane_of(kt);
            if (pc < 2) rr[pc] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, static_cast<int>(r_off[pc]), static_cast<int>(pl * p.x_plane), 0);
            else ru[pc - 2] = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, static_cast<int>(u_off[pc - 2]), static_cast<int>(pl * p.u_plane), 0);
        };
        auto store_piece = [&](int pc, int buf) {
            if (pc == 0) *reinterpret_cast<u32x4*>(Rs + buf * RAW_FLOATS + r_lds[0]) = rr[0];
            else if (pc == 1) { if (r1) *reinterpret_cast<u32x4*>(Rs + buf * RAW_FLOATS + r_lds[1]) = rr[1]; }
            else *reinterpret_cast<u32x4*>(Us + buf * 16 * PLANE + (tid + 512 * (pc - 2)) * 4) = ru[pc - 2];
        };

        // wave = (i, jh): components xi = 4 i + 2 jh + {0, 1}; transform row i combines patch rows (ra, rb) as ra + rs rb
        const int wi = wave >> 1, jh = wave & 1;
        const int row_a = wi == 0 ? 0 : wi == 2 ? 2 : 1;
        const int row_b = wi == 0 ? 2 : wi == 1 ? 2 : wi == 2 ? 1 : 3;
        const float rs = wi == 1 ? 1.0f : -1.0f;
        int pbase[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int pos = a * 32 + ln, py = pos >> 3, px = pos & 7;
            pbase[a] = 2 * py * RROW + (2 * px + jh) * RPIX + 4 * lh;  // columns jh, jh+1, jh+2 of the patch
        }
        const int offA = row_a * RROW, offB = row_b * RROW;
        const float* Bw = Us + (wave * 2) * PLANE + lh * (WN * 4) + ln * 4;

        f32x16 acc[2][2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][a][c][r] = 0.f;

        // patch pieces of position half a: d[0..2] = row A columns 0..2, d[3..5] = row B
        float4 d[6], av[2][2], fb[2][2];
        auto patch_read = [&](int i, int buf, int a) {
            d[i] = *reinterpret_cast<const float4*>(Rs + buf * RAW_FLOATS + pbase[a] + (i < 3 ? offA : offB) + (i % 3) * RPIX);
        };
        // packed arithmetic: one VALU issue per channel pair (VALU does not co-execute with this MFMA: every issue counts)
        const f32x2 rs2 = {rs, rs};
        auto lo2 = [](const float4& v) { return f32x2{v.x, v.y}; };
        auto hi2 = [](const float4& v) { return f32x2{v.z, v.w}; };
        auto pk_fma = [](f32x2 a, f32x2 b, f32x2 c) {
            f32x2 r;
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            return r;
        };
        auto join = [](f32x2 lo, f32x2 hi) { return make_float4(lo.x, lo.y, hi.x, hi.y); };
        auto row_pass = [&](int cc) {  // l[cc] = dA + rs * dB, in place of d[cc]
            d[cc] = join(pk_fma(lo2(d[3 + cc]), rs2, lo2(d[cc])), pk_fma(hi2(d[3 + cc]), rs2, hi2(d[cc])));
        };
        auto col_pass = [&](int a) {
            if (jh == 0) {  // columns 0,1,2: components j = 0: t0 - t2, j = 1: t1 + t2
                av[0][a] = join(pk_sub(lo2(d[0]), lo2(d[2])), pk_sub(hi2(d[0]), hi2(d[2])));
                av[1][a] = join(pk_add(lo2(d[1]), lo2(d[2])), pk_add(hi2(d[1]), hi2(d[2])));
            } else {        // columns 1,2,3: components j = 2: t2 - t1, j = 3: t1 - t3
                av[0][a] = join(pk_sub(lo2(d[1]), lo2(d[0])), pk_sub(hi2(d[1]), hi2(d[0])));
                av[1][a] = join(pk_sub(lo2(d[0]), lo2(d[2])), pk_sub(hi2(d[0]), hi2(d[2])));
            }
        };
        auto fb_read = [&](int j, int buf) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fb[j][c] = *reinterpret_cast<const float4*>(Bw + buf * 16 * PLANE + j * PLANE + c * 32 * 4);
        };

        // prologue: k tile 0 in buffer 0, k tile 1 in flight, the first operands (position half 0, component pair j = 0)
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) load_piece(pc, 0);
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) store_piece(pc, 0);
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) load_piece(pc, 1);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; ++i) patch_read(i, 0, 0);
        fb_read(0, 0);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) row_pass(cc);
        col_pass(0);
        // ---- main loop, order pinned (sched_barrier after every MFMA); 32 MFMAs per k tile and wave:
        //   slots  0-15  position half 0 (j = 0: 0-7, j = 1: 8-15): fb[1] of this k tile, the six patch reads of half 1
        //                (one per slot), their transform (slot 8), the six LDS writes of k tile kt+1 (10-15)
        //   slots 16-31  position half 1: the six global loads of k tile kt+2 (16-21); the barrier sits after slot 23, then
        //                fb[0] and the patch reads of half 0 of k tile kt+1 (24-29) and their transform (30) — the
        //                next k tile starts with its operands in registers.
        // the k loop is unrolled by two so that the buffer index is a compile-time constant: every LDS address is then a
        // per-thread base plus an instruction-immediate offset (no address VALU in the loop)
        auto ktile = [&](auto bufc, int kt) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int stp = 0; stp < 4; ++stp)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float4 fav = av[j][a], fbv = fb[j][c];
                            const float avv = stp == 0 ? fav.x : stp == 1 ? fav.y : stp == 2 ? fav.z : fav.w;
                            const float bvv = stp == 0 ? fbv.x : stp == 1 ? fbv.y : stp == 2 ? fbv.z : fbv.w;
                            acc[j][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv, bvv, acc[j][a][c], 0, 0, 0);
                            const int m = a * 16 + j * 8 + stp * 2 + c;  // slot 0..31
                            if (m == 0) fb_read(1, buf);
                            if (m < 6) patch_read(m, buf, 1);
                            if (m == 8) {   // each transform behind ONE MFMA: switching between MFMAs and VALU work costs
                                row_pass(0); row_pass(1); row_pass(2);
                                col_pass(1);
                            }
                            if (m >= 10 && m < 16) store_piece(m - 10, buf ^ 1);
                            if (m >= 16 && m < 22) load_piece(m - 16, kt + 2);
                            if (m == 23) {
                                __syncthreads();  // k tile kt+1 is complete in buf^1; nobody reads buf any more
                                fb_read(0, buf ^ 1);
                            }
                            if (m >= 24 && m < 30) patch_read(m - 24, buf ^ 1, 0);
                            if (m == 30) {
                                row_pass(0); row_pass(1); row_pass(2);
                                col_pass(0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(std::integral_constant<int, 0>{}, kt);
            ktile(std::integral_constant<int, 1>{}, kt + 1);
        }
        if (kt < nk) ktile(std::integral_constant<int, 0>{}, kt);  // odd number of k tiles
        __syncthreads();  // the behind-the-barrier operand reads of a k tile that does not exist are done

        // ---- epilogue: as the linear kernel's, positions decoded from the 8 x 8 block
        const int n = n0 + (tid & 63);
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.0f, sh = (n_ok && p.shift) ? p.shift[n] : 0.0f;
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yk_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.yk, 0, p.y_bytes, 0x00020000);
        const unsigned ncol = n_ok ? static_cast<unsigned>(n) * 4u : OOB;
        const unsigned kcol = static_cast<unsigned>(n >> 3) * p.yk_plane + static_cast<unsigned>(n & 7) * 4u;
        float* Z = smem;  // [wave][2 partials][8 position quads][64 channels][4 positions]
        constexpr int ZQ = 8 * WN * 4;
        // HEADS (see conv3x3_wino8_f32<true>): the round's 128 pixels x 64 channels transposed above the Z exchange
        float* Tt = smem + 2 * 2 * 16 * PLANE;
        const int hrt = wave & 3, hkh = wave >> 2;
        float4 wh[4];
        int ln_e = ln, lh_e = lh, tid_e = tid;  // opaque copies: keep the heads epilogue's address arithmetic behind the loop
        if constexpr (HEADS) {
            asm volatile("" : "+v"(ln_e), "+v"(lh_e), "+v"(tid_e));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wh[j] = *reinterpret_cast<const float4*>(p.w_head + static_cast<int64_t>(ln_e) * p.Cout + n0 + hkh * 32 + j * 8 + lh_e * 4);
        }
        const __amdgpu_buffer_rsrc_t hp_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.head_part, 0, HEADS ? p.head_bytes : 0u, 0x00020000);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            if (h) __syncthreads();  // (h = 0: the barrier behind the main loop)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int pq = 2 * rq + lh, ch = c * 32 + ln;
                    float4 zp, zq;
                    float* zpp = &zp.x;
                    float* zqp = &zq.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = rq * 4 + e;
                        const float ma = h == 0 ? acc[0][0][c][r] : acc[0][1][c][r];
                        const float mb = h == 0 ? acc[1][0][c][r] : acc[1][1][c][r];
                        zpp[e] = jh == 0 ? ma + mb : ma;
                        zqp[e] = jh == 0 ? mb : -ma - mb;
                    }
                    *reinterpret_cast<float4*>(Z + (wave * 2 + 0) * ZQ + (pq * WN + ch) * 4) = zp;
                    *reinterpret_cast<float4*>(Z + (wave * 2 + 1) * ZQ + (pq * WN + ch) * 4) = zq;
                }
            __syncthreads();
            const int pq = tid >> 6, ch = tid & 63;
            float4 y0[4], y1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 p0 = *reinterpret_cast<const float4*>(Z + ((2 * i) * 2 + 0) * ZQ + (pq * WN + ch) * 4);
                const float4 p1 = *reinterpret_cast<const float4*>(Z + ((2 * i + 1) * 2 + 0) * ZQ + (pq * WN + ch) * 4);
                const float4 q0 = *reinterpret_cast<const float4*>(Z + ((2 * i) * 2 + 1) * ZQ + (pq * WN + ch) * 4);
                const float4 q1 = *reinterpret_cast<const float4*>(Z + ((2 * i + 1) * 2 + 1) * ZQ + (pq * WN + ch) * 4);
                y0[i] = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
                y1[i] = make_float4(q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, q0.w + q1.w);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pos = h * 32 + pq * 4 + e;  // uniform per wave
                const int ty = ty0 + (pos >> 3), tx = tx0 + (pos & 7);
                const bool pv = ty < p.TH && tx < p.TW;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float z0 = c == 0 ? (&y0[0].x)[e] : (&y1[0].x)[e], z1 = c == 0 ? (&y0[1].x)[e] : (&y1[1].x)[e];
                    const float z2 = c == 0 ? (&y0[2].x)[e] : (&y1[2].x)[e], z3 = c == 0 ? (&y0[3].x)[e] : (&y1[3].x)[e];
                    const float yv[2] = {z0 + z1 + z2, z1 - z2 - z3};
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float v = yv[a] * sc + sh;
                        if (p.act) v = v > 0.f ? v : 0.f;
                        if constexpr (HEADS) {
                            const int pxl = ((tid_e >> 6) * 4 + e) * 4 + a * 2 + c;  // position-major pixel of the round
                            Tt[pxl * WN + ((tid_e & 63) ^ ((pxl & 15) << 2))] = v;
                            continue;
                        }
                        const unsigned px = static_cast<unsigned>((b * p.H + 2 * ty + a) * p.W + 2 * tx + c);
                        if (p.y) {
                            const unsigned o = (pv && n_ok) ? px * (static_cast<unsigned>(p.Cout) * 4u) + ncol : OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, static_cast<int>(o), 0, 0);
                        }
                        if (p.yk) {
                            const unsigned o = (pv && n_ok) ? kcol + px * 32u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yk_rsrc, static_cast<int>(o), 0, 0);
                        }
                    }
                }
            }
            if constexpr (HEADS) {
                __syncthreads();
                const int row0 = mt * 256 + h * 128 + hrt * 32 + 4 * lh_e;  // + (r & 3) + 8 (r >> 2)
                const unsigned hbase = (static_cast<unsigned>(hkh) * static_cast<unsigned>(p.tiles_m) * 256u + static_cast<unsigned>(row0)) * 128u +
                                       static_cast<unsigned>(ln_e) * 4u;
                f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hacc[r] = 0.f;
                    if (nt != 0)
                        hacc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            hp_rsrc, static_cast<int>(hbase + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * 128u), 0, 0));
                }
                const int pxl = hrt * 32 + ln_e;
                const float* trow = Tt + pxl * WN;
                const int swz = (pxl & 15) << 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 a4 = *reinterpret_cast<const float4*>(trow + ((hkh * 32 + j * 8 + lh_e * 4) ^ swz));
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, wh[j].x, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, wh[j].y, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, wh[j].z, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, wh[j].w, hacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hacc[r]), hp_rsrc,
                                                          static_cast<int>(hbase + static_cast<unsigned>((r & 3) + 8 * (r >> 2)) * 128u), 0, 0);
            }
        }
        __syncthreads();  // Z is read out: the next tile may overwrite the staging buffers
    }
}

// U_xi[n][c] = (G g G^T)[i][j], xi = 4i + j, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; evaluated in double; stored
// k-blocked [Cin/8][16][Cout][8] so that a k tile's 16 x 64 x 8 block is sixteen contiguous 2 KB runs.
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, int cout, int cin,
                                                           float* __restrict__ u) {
    const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (e >= static_cast<int64_t>(cout) * cin) return;
    const int n = static_cast<int>(e / cin), c = static_cast<int>(e - static_cast<int64_t>(n) * cin);
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((static_cast<int64_t>(n) * 3 + ky) * 3 + kx) * cin + c];
    double t[4][3];
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
    for (int i = 0; i < 4; ++i) {
        const double r[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
        for (int j = 0; j < 4; ++j)
            u[((static_cast<int64_t>(c >> 3) * 16 + (i * 4 + j)) * cout + n) * 8 + (c & 7)] = static_cast<float>(r[j]);
    }
}

// NHWC [M][C] -> k-blocked [C/8][M][8] through LDS: reads are full rows, writes 32-byte pieces that are contiguous
// across consecutive pixels. The Winograd kernel's patch loads then use every byte of the cache lines they touch
// (in NHWC an 8-channel k tile is 32 bytes out of every 128-byte line: 4x the L2 -> L1 traffic, and the kernel was
// bound by it).
__global__ __launch_bounds__(256) void kblock_kernel(const float* __restrict__ x, int64_t M, int C,
                                                     float* __restrict__ y) {
    __shared__ float tile[32][64 + 1];  // 32 pixels x 64 channels
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * 32;
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    for (int i = tid; i < 32 * 16; i += 256) {  // float4 reads: 16 per pixel row
        const int r = i >> 4, q = i & 15;
        const int64_t m = m0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && c0 + q * 4 < C) v = *reinterpret_cast<const float4*>(x + m * C + c0 + q * 4);
        tile[r][q * 4 + 0] = v.x; tile[r][q * 4 + 1] = v.y; tile[r][q * 4 + 2] = v.z; tile[r][q * 4 + 3] = v.w;
    }
    __syncthreads();
    for (int i = tid; i < 8 * 32 * 2; i += 256) {  // (channel group, pixel, half): float4 writes
        const int g = i >> 6, r = (i >> 1) & 31, h = i & 1;
        const int64_t m = m0 + r;
        const int c = c0 + g * 8 + h * 4;
        if (m < M && c < C) {
            const float4 v = make_float4(tile[r][g * 8 + h * 4 + 0], tile[r][g * 8 + h * 4 + 1],
                                         tile[r][g * 8 + h * 4 + 2], tile[r][g * 8 + h * 4 + 3]);
            *reinterpret_cast<float4*>(y + (static_cast<int64_t>(c >> 3) * M + m) * 8 + h * 4) = v;
        }
    }
}

}  // namespace

static int g_spatial = -1;  // -1: not decided yet (environment), 0 / 1: linear / spatial tiles for large maps

extern "C" int mrcnn_winograd_set_spatial(int32_t on) {
    g_spatial = on < 0 ? -1 : (on ? 1 : 0);
    return MRCNN_OK;
}

extern "C" int mrcnn_winograd_weights_f32(const float* w, int32_t cout, int32_t cin, float* u, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(w && u, "winograd_weights: null pointer");
    MRCNN_REQUIRE(cout >= 1 && cin >= 1 && 16LL * cout * cin < (1LL << 30), "winograd_weights: cout=%d cin=%d", cout, cin);
    const int64_t n = static_cast<int64_t>(cout) * cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       mrcnn::as_stream(stream), w, cout, cin, u);
    return mrcnn::check_launch("wino_weights_kernel");
}

extern "C" size_t mrcnn_conv3x3_winograd_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cin) {
    if (batch < 1 || height < 1 || width < 1 || cin < 1) return 0;
    return sizeof(float) * static_cast<size_t>(batch) * height * width * cin;
}

extern "C" int mrcnn_conv3x3_winograd_f32(const float* x, int32_t x_layout, int32_t batch, int32_t height,
                                          int32_t width, int32_t cin, const float* u, int32_t cout,
                                          const float* scale, const float* shift, int32_t activation, float* y_nhwc,
                                          float* y_kblocked, void* workspace, size_t workspace_bytes,
                                          mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && u && (y_nhwc || y_kblocked), "conv3x3_winograd: null pointer");
    MRCNN_REQUIRE(x_layout == MRCNN_LAYOUT_NHWC || x_layout == MRCNN_LAYOUT_KBLOCKED, "conv3x3_winograd: bad x_layout");
    MRCNN_REQUIRE(x_layout == MRCNN_LAYOUT_KBLOCKED ||
                      (workspace && workspace_bytes >= mrcnn_conv3x3_winograd_workspace_bytes(batch, height, width, cin)),
                  "conv3x3_winograd: an NHWC input needs a workspace of mrcnn_conv3x3_winograd_workspace_bytes()");
    MRCNN_REQUIRE(batch >= 1 && height >= 2 && width >= 2 && height % 2 == 0 && width % 2 == 0,
                  "conv3x3_winograd: B=%d H=%d W=%d (even sizes required)", batch, height, width);
    MRCNN_REQUIRE(cin >= 8 && cin % 8 == 0 && cout >= 1, "conv3x3_winograd: Cin=%d (%% 8 == 0 required) Cout=%d", cin, cout);
    MRCNN_REQUIRE(y_kblocked == nullptr || cout % 8 == 0, "conv3x3_winograd: a k-blocked output needs Cout %% 8 == 0");
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv3x3_winograd: activation must be 0 or 1");
    const long long px = 1LL * batch * height * width;
    MRCNN_REQUIRE(px * cin < (1LL << 30) && px * cout < (1LL << 30) && 16LL * cin * cout < (1LL << 30),
                  "conv3x3_winograd: tensor too large (32-bit buffer byte offsets)");
    hipStream_t st = mrcnn::as_stream(stream);
    const float* x8 = x;
    if (x_layout == MRCNN_LAYOUT_NHWC) {
        float* t = static_cast<float*>(workspace);
        hipLaunchKernelGGL(kblock_kernel, dim3(static_cast<unsigned>((px + 31) / 32), (cin + 63) / 64), dim3(256), 0, st,
                           x, static_cast<int64_t>(px), cin, t);
        x8 = t;
    }
    WinoParams p;
    p.x = x8; p.u = u; p.scale = scale; p.shift = shift; p.y = y_nhwc; p.yk = y_kblocked;
    p.w_head = nullptr; p.head_part = nullptr; p.head_bytes = 0;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout;
    p.TH = height / 2; p.TW = width / 2; p.T = batch * p.TH * p.TW; p.act = activation;
    p.tiles_m = (p.T + WT - 1) / WT;
    p.tiles_n = (cout + WN - 1) / WN;
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.u_bytes = static_cast<unsigned>(4LL * 16 * cin * cout);
    p.x_plane = static_cast<unsigned>(4LL * px * WK);
    p.u_plane = static_cast<unsigned>(4LL * 16 * cout * WK);
    p.yk_plane = static_cast<unsigned>(4LL * px * 8);
    p.y_bytes = static_cast<unsigned>(4LL * px * cout);
    const long long grid = 8LL * ((p.tiles_m + 7) / 8) * p.tiles_n;
    MRCNN_REQUIRE(grid <= 0x7fffffffLL, "conv3x3_winograd: grid too large");
    if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wino8_f32<false>), WINO_LDS, "conv3x3_winograd"))
        return rc;
    // spatial-tile kernel (8 x 8 position blocks of one image) for maps of at least 8 x 8 positions — the default;
    // MRCNN_WINO_SPATIAL=0 (or mrcnn_winograd_set_spatial(0)) keeps the linear-tile kernel everywhere. Same results bit for bit.
    if (g_spatial < 0) g_spatial = (getenv("MRCNN_WINO_SPATIAL") && atoi(getenv("MRCNN_WINO_SPATIAL")) == 0) ? 0 : 1;
    if (g_spatial == 1 && p.TH >= 8 && p.TW >= 8) {
        WinoSParams q;
        static_cast<WinoParams&>(q) = p;
        q.tyb = (p.TH + 7) / 8;
        q.txb = (p.TW + 7) / 8;
        q.tiles_m = batch * q.tyb * q.txb;
        const long long sgrid = 8LL * ((q.tiles_m + 7) / 8) * q.tiles_n;
        MRCNN_REQUIRE(sgrid <= 0x7fffffffLL, "conv3x3_winograd: grid too large");
        if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wino8s_f32<false>), WINOS_LDS, "conv3x3_winograd"))
            return rc;
        const int cus = mrcnn::device_cu_count();
        if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd: cannot query the device");
        const int ncu = cus >= 8 ? (cus / 8) * 8 : 8;
        const long long launch = sgrid > ncu ? ncu : sgrid;
        hipLaunchKernelGGL(conv3x3_wino8s_f32<false>, dim3(static_cast<unsigned>(launch)), dim3(512), WINOS_LDS, st, q);
        return mrcnn::check_launch("conv3x3_wino8s_f32");
    }
#ifdef MRCNN_ABLATIONS
    static const bool four_waves = mrcnn::tuning_env("MRCNN_WINO_WAVES") && atoi(mrcnn::tuning_env("MRCNN_WINO_WAVES")) == 4;
    if (four_waves) {
        if (int rc = mrcnn::ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wino_f32), WINO_LDS, "conv3x3_winograd"))
            return rc;
        hipLaunchKernelGGL(conv3x3_wino_f32, dim3(static_cast<unsigned>(grid)), dim3(256), WINO_LDS, st, p);
        return mrcnn::check_launch("conv3x3_wino_f32");
    }
#endif
    {
        const int cus = mrcnn::device_cu_count();
        if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd: cannot query the device");
        const int num_cu = cus >= 8 ? (cus / 8) * 8 : 8;
        static const bool persistent = !(mrcnn::tuning_env("MRCNN_WINO_PERSISTENT") && atoi(mrcnn::tuning_env("MRCNN_WINO_PERSISTENT")) == 0);
        const long long launch = (persistent && grid > num_cu) ? num_cu : grid;
        hipLaunchKernelGGL(conv3x3_wino8_f32<false>, dim3(static_cast<unsigned>(launch)), dim3(512), WINO_LDS, st, p);
    }
    return mrcnn::check_launch("conv3x3_wino_f32");
}

// tile_mode: 1 = 64 consecutive positions per M tile (rows in position-major order), 2 = 8 x 8 position blocks
extern "C" int64_t mrcnn_conv3x3_winograd_heads_rows(int32_t batch, int32_t height, int32_t width, int32_t tile_mode) {
    if (batch < 1 || height < 2 || width < 2 || height % 2 || width % 2) return 0;
    const int64_t th = height / 2, tw = width / 2;
    if (tile_mode == 2) return static_cast<int64_t>(batch) * ((th + 7) / 8) * ((tw + 7) / 8) * 256;
    if (tile_mode != 1) return 0;
    return ((static_cast<int64_t>(batch) * th * tw + WT - 1) / WT) * 256;
}

extern "C" int32_t mrcnn_conv3x3_winograd_heads_tile_mode(int32_t height, int32_t width) {
    if (g_spatial < 0) g_spatial = (getenv("MRCNN_WINO_SPATIAL") && atoi(getenv("MRCNN_WINO_SPATIAL")) == 0) ? 0 : 1;
    return (g_spatial == 1 && height / 2 >= 8 && width / 2 >= 8) ? 2 : 1;
}

extern "C" int mrcnn_conv3x3_winograd_heads_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width,
                                                int32_t cin, const float* u, int32_t cout, const float* scale,
                                                const float* shift, int32_t activation, const float* w_head32,
                                                int32_t tile_mode, float* head_part, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x_kblocked && u && w_head32 && head_part, "conv3x3_winograd_heads: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 2 && width >= 2 && height % 2 == 0 && width % 2 == 0,
                  "conv3x3_winograd_heads: B=%d H=%d W=%d (even sizes required)", batch, height, width);
    MRCNN_REQUIRE(cin >= 8 && cin % 8 == 0 && cout >= WN && cout % WN == 0,
                  "conv3x3_winograd_heads: Cin=%d (%% 8 == 0) Cout=%d (%% 64 == 0) required", cin, cout);
    MRCNN_REQUIRE(activation == 0 || activation == 1, "conv3x3_winograd_heads: activation must be 0 or 1");
#ifdef MRCNN_ABLATIONS
    MRCNN_REQUIRE(tile_mode == 1 || (tile_mode == 2 && height / 2 >= 8 && width / 2 >= 8),
                  "conv3x3_winograd_heads: tile_mode must be 1, or 2 on maps of at least 8 x 8 tile positions");
#else   // the linear-tile heads variant (tile_mode 1) exists in MRCNN_ABLATIONS builds only: same numbers, slower
    MRCNN_REQUIRE(tile_mode == 2 && height / 2 >= 8 && width / 2 >= 8,
                  "conv3x3_winograd_heads: tile_mode must be 2 (8 x 8 position blocks), on maps of at least 8 x 8 tile positions");
#endif
    const long long px = 1LL * batch * height * width;
    const long long rows = mrcnn_conv3x3_winograd_heads_rows(batch, height, width, tile_mode);
    MRCNN_REQUIRE(px * cin < (1LL << 30) && 16LL * cin * cout < (1LL << 30) && 2 * rows * 32 < (1LL << 30),
                  "conv3x3_winograd_heads: tensor too large (32-bit buffer byte offsets)");
    WinoSParams p;
    p.x = x_kblocked; p.u = u; p.scale = scale; p.shift = shift; p.y = nullptr; p.yk = nullptr;
    p.B = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout;
    p.TH = height / 2; p.TW = width / 2; p.T = batch * p.TH * p.TW; p.act = activation;
    p.tyb = (p.TH + 7) / 8; p.txb = (p.TW + 7) / 8;
    p.tiles_m = tile_mode == 2 ? batch * p.tyb * p.txb : (p.T + WT - 1) / WT;
    p.tiles_n = cout / WN;
    p.x_bytes = static_cast<unsigned>(4LL * px * cin);
    p.u_bytes = static_cast<unsigned>(4LL * 16 * cin * cout);
    p.x_plane = static_cast<unsigned>(4LL * px * WK);
    p.u_plane = static_cast<unsigned>(4LL * 16 * cout * WK);
    p.yk_plane = 0; p.y_bytes = 0;
    p.w_head = w_head32; p.head_part = head_part;
    p.head_bytes = static_cast<unsigned>(4LL * 2 * rows * 32);
    const void* kern = reinterpret_cast<const void*>(conv3x3_wino8s_f32<true>);
#ifdef MRCNN_ABLATIONS
    if (tile_mode != 2) kern = reinterpret_cast<const void*>(conv3x3_wino8_f32<true>);
#endif
    if (int rc = mrcnn::ensure_dynamic_lds(kern, WINO_HEADS_LDS, "conv3x3_winograd_heads")) return rc;
    const int cus = mrcnn::device_cu_count();
    if (cus <= 0) return mrcnn::fail(MRCNN_ERR_LAUNCH, "conv3x3_winograd_heads: cannot query the device");
    const int num_cu = cus >= 8 ? (cus / 8) * 8 : 8;
    const long long units = 8LL * ((p.tiles_m + 7) / 8);  // M-tile units; a workgroup walks the N tiles of its units
    const long long launch = units < num_cu ? units : num_cu;
#ifdef MRCNN_ABLATIONS
    if (tile_mode != 2) {
        hipLaunchKernelGGL(conv3x3_wino8_f32<true>, dim3(static_cast<unsigned>(launch)), dim3(512), WINO_HEADS_LDS,
                           mrcnn::as_stream(stream), static_cast<const WinoParams&>(p));
        return mrcnn::check_launch("conv3x3_wino8_f32<heads>");
    }
#endif
    hipLaunchKernelGGL(conv3x3_wino8s_f32<true>, dim3(static_cast<unsigned>(launch)), dim3(512), WINO_HEADS_LDS,
                       mrcnn::as_stream(stream), p);
    return mrcnn::check_launch("conv3x3_wino8s_f32<heads>");
}

extern "C" int mrcnn_conv3x3_winograd_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                                               int32_t cin, const float* u, int32_t cout, const float* scale,
                                               const float* shift, int32_t activation, float* y, void* workspace,
                                               size_t workspace_bytes, mrcnn_stream_t stream) {
    return mrcnn_conv3x3_winograd_f32(x, MRCNN_LAYOUT_NHWC, batch, height, width, cin, u, cout, scale, shift,
                                      activation, y, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mrcnn_nhwc_to_kblocked_f32(const float* x, int64_t pixels, int32_t channels, float* y,
                                          mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && y, "nhwc_to_kblocked: null pointer");
    MRCNN_REQUIRE(pixels >= 1 && channels >= 8 && channels % 8 == 0 && pixels * channels < (1LL << 31),
                  "nhwc_to_kblocked: pixels=%lld channels=%d", static_cast<long long>(pixels), channels);
    hipLaunchKernelGGL(kblock_kernel, dim3(static_cast<unsigned>((pixels + 31) / 32), (channels + 63) / 64), dim3(256), 0,
                       mrcnn::as_stream(stream), x, pixels, channels, y);
    return mrcnn::check_launch("kblock_kernel");
}
