/* maskrcnn_hip.h — C ABI of libmaskrcnn_hip.so: the MI355X (gfx950) implementation of the
 * Mask R-CNN inference hot path of delldu/MaskRCNN.
 *
 * This is the drop-in boundary. Every entry point takes plain device pointers, sizes and a HIP stream
 * (no torch types), launches asynchronously on that stream, performs no allocation and no host
 * synchronisation (safe inside hipGraph capture), and returns MRCNN_OK or a negative error code with
 * a thread-local message in mrcnn_last_error(). Nothing here ever calls exit().
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference tree).
 * The Python face (package `maskrcnn`: nms, CropFunction, _C.{nms,crop_forward,crop_backward}, and
 * torch.ops.maskrcnn.*) is a thin binding over these symbols: see INTEGRATION.md.
 */
#ifndef MASKRCNN_HIP_H
#define MASKRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCNN_ABI_VERSION 17

#define MRCNN_OK 0
#define MRCNN_ERR_INVALID_ARGUMENT (-1) /* bad shape / null pointer / unsupported size          */
#define MRCNN_ERR_UNSUPPORTED (-2)      /* valid request this build has no kernel for            */
#define MRCNN_ERR_LAUNCH (-3)           /* hipLaunch / runtime error (message has hipGetErrorString) */

typedef void* mrcnn_stream_t; /* a hipStream_t; NULL = the null stream */

/* ABI version of the loaded library (== MRCNN_ABI_VERSION it was built with). */
int mrcnn_abi_version(void);
/* Message for the last non-OK return on this thread ("" if none). Never NULL. */
const char* mrcnn_last_error(void);
/* gfx target the device code was compiled for ("gfx950"). */
const char* mrcnn_arch(void);

/* ------------------------------------------------------------------------------------------------
 * NMS — replaces  at::Tensor nms(const at::Tensor& dets, const float threshold)
 *       c++ext/maskrcnn/csrc/nms.h:15-30, cpu path csrc/cpu/nms_cpu.cpp:11-79.
 *
 * Greedy NMS with the CPU path's exact arithmetic: areas (x2-x1+1)*(y2-y1+1), boxes visited in
 * descending score order (ties: lower input index first), box j suppressed by a kept box i when
 * inter/(area_i+area_j-inter) >= threshold (IEEE fp32, no FMA contraction; `>=`, not the CUDA
 * path's `>`). Survivors are reported as ASCENDING INPUT INDICES (nms_cpu.cpp:69).
 *
 * S independent segments are processed by one launch (one workgroup per segment).
 *   dets        [S][n_max] rows (y1,x1,y2,x2,score); element (s,i,c) at
 *               dets[s*seg_stride + i*row_stride + c*col_stride]   (strides in elements)
 *   seg_counts  int32[S] boxes in each segment (<= n_max), device memory; NULL = all n_max
 *   class_ids   int32, element (s,i) at class_ids[s*n_max + i]; NULL = single class. When given,
 *               a box only suppresses boxes of the same class: the one-pass form of the per-class
 *               loop in MaskRCNN.mrn_refine (model.py:1454-1475).
 *   keep_out    int64[S][n_max]: first counts_out[s] entries = kept indices ascending, rest = -1
 *   counts_out  int32[S]
 *   workspace   optional device scratch of >= mrcnn_nms_workspace_bytes(S, n_max) bytes, 16-byte aligned.
 *               NULL: one LDS-resident workgroup per segment does everything (one launch, no scratch).
 *               Given (and n_max > 128): the pair tests are spread over the whole chip (sort → 64x64 pair-mask
 *               tiles → serial scan), 3 launches, several times faster at n_max >= 500. Same results.
 * Limits: 1 <= n_max <= mrcnn_nms_max_boxes() = 16384 with a workspace, 4096 without.  S >= 1.
 * ---------------------------------------------------------------------------------------------- */
int64_t mrcnn_nms_max_boxes(void);
size_t mrcnn_nms_workspace_bytes(int32_t num_segments, int64_t n_max);
int mrcnn_nms_batched_f32(const float* dets, int32_t num_segments, int64_t n_max, int64_t seg_stride,
                          int64_t row_stride, int64_t col_stride, const int32_t* seg_counts,
                          const int32_t* class_ids, float threshold, int64_t* keep_out,
                          int32_t* counts_out, void* workspace, size_t workspace_bytes,
                          mrcnn_stream_t stream);

/* The rest of the reference's nms surface: ANY number of boxes, and float64 boxes (cpu/nms_cpu.cpp:73-79 dispatches over the
 * floating types; nms.h:15 takes a float threshold either way). One segment; arithmetic in the boxes' own type; the same
 * visiting order (descending score, ties by ascending index) and `>=` rule → the same keep set as the CPU path. No O(N^2)
 * memory: radix sort (hipCUB), then one launch per 64-box chunk of the order (csrc/nms_general.hip).
 *   dets        dtype 0: float32, 1: float64; box r = (y1,x1,y2,x2,score) at dets[r*row_stride + c*col_stride] (elements)
 *   keep_out    int64[n]: first *count_out entries = kept input indices ascending, rest = -1;  count_out: int64[1] (device)
 *   workspace   >= mrcnn_nms_general_workspace_bytes(n, dtype) bytes, 256-byte aligned.   1 <= n <= 2^31 - 65. */
size_t mrcnn_nms_general_workspace_bytes(int64_t n, int32_t dtype);
int mrcnn_nms_general(const void* dets, int32_t dtype, int64_t n, int64_t row_stride, int64_t col_stride, float threshold,
                      int64_t* keep_out, int64_t* count_out, void* workspace, size_t workspace_bytes,
                      mrcnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * crop_and_resize ("RoIAlign") forward — replaces
 *   void crop_forward(image, boxes, box_index, extrapolation_value, crop_height, crop_width, crops)
 *   c++ext/maskrcnn/csrc/crop.h:14-34, cpu path csrc/cpu/crop_cpu.cpp:13-164.
 *
 *   image      fp32 [batch][depth][H][W]   (NCHW, contiguous)
 *   boxes      fp32 [num_boxes][4] normalised (y1,x1,y2,x2)
 *   box_index  int32[num_boxes] in [0,batch). An out-of-range index fills that box's crop with
 *              extrapolation_value (the reference CPU path printf()s and exit(-1)s, crop_cpu.cpp:47-50;
 *              its CUDA path silently skips, crop_cuda.cu:41-44).
 *   crops      fp32 [num_boxes][depth][crop_height][crop_width], fully overwritten.
 * Arithmetic is the reference's, op for op (one bilinear sample per bin, endpoint-inclusive grid,
 * strict outside test, floorf/ceilf taps, a+(b-a)*t lerps, crop size 1 → box centre in double).
 * ---------------------------------------------------------------------------------------------- */
int mrcnn_crop_forward_f32(const float* image, int32_t batch, int32_t depth, int32_t height,
                           int32_t width, const float* boxes, const int32_t* box_index,
                           int32_t num_boxes, float extrapolation_value, int32_t crop_height,
                           int32_t crop_width, float* crops, mrcnn_stream_t stream);

/* crop_and_resize backward — replaces
 *   void crop_backward(grads, boxes, box_index, grads_image)
 *   c++ext/maskrcnn/csrc/crop.h:36-53, cpu path csrc/cpu/crop_cpu.cpp:167-265.
 *   grads [num_boxes][depth][crop_h][crop_w] → grads_image [batch][depth][H][W] (zeroed here, then
 *   accumulated with fp32 atomics: summation order differs from the serial CPU loop). */
int mrcnn_crop_backward_f32(const float* grads, const float* boxes, const int32_t* box_index,
                            int32_t num_boxes, int32_t batch, int32_t depth, int32_t height,
                            int32_t width, int32_t crop_height, int32_t crop_width,
                            float* grads_image, mrcnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pyramid RoIAlign, channels-last — replaces the Python dispatcher
 *   roi_align(inputs, pool_size, image_shape)   model.py:276-393
 * (level = clamp(round_half_even(4 + log2(sqrt(h*w) / (224/sqrt(H_img*W_img)))), 2, 5), one
 * crop_forward per level, concat, restore order) with one launch and no host synchronisation.
 *
 *   fm[l]      fp32 [batch][H_l][W_l][depth]  (NHWC), l = 0..3 for P2..P5; depth % 4 == 0
 *   rois       fp32 [num_rois][4] normalised (y1,x1,y2,x2); roi r reads image roi_batch[r]
 *              (roi_batch NULL → image r / rois_per_image).
 *   out        fp32 [num_rois][pool][pool][depth] (NHWC), original roi order
 *   levels_out int32[num_rois] (optional, may be NULL): the level 2..5 chosen per roi
 * ---------------------------------------------------------------------------------------------- */
int mrcnn_roi_align_pyramid_nhwc_f32(const float* const fm[4], const int32_t fm_h[4],
                                     const int32_t fm_w[4], int32_t batch, int32_t depth,
                                     const float* rois, const int32_t* roi_batch, int32_t num_rois,
                                     int32_t rois_per_image, int32_t pool, float image_area,
                                     float* out, int32_t* levels_out, mrcnn_stream_t stream);
/* The same with a selectable output layout: out_layout = MRCNN_LAYOUT_NHWC (as above), MRCNN_LAYOUT_KBLOCKED
 * ([depth/8][num_rois*pool*pool][8], depth % 8 == 0 — what mrcnn_conv3x3_winograd_f32 reads: the mask head's first
 * conv then needs no layout pass), or MRCNN_LAYOUT_NHWC_F16 (NHWC, each value rounded to fp16 — `out` then points to
 * fp16 storage: the "f16" mode's heads, whose first conv would round the fp32 values the same way). */
int mrcnn_roi_align_pyramid_f32(const float* const fm[4], const int32_t fm_h[4], const int32_t fm_w[4], int32_t batch,
                                int32_t depth, const float* rois, const int32_t* roi_batch, int32_t num_rois,
                                int32_t rois_per_image, int32_t pool, float image_area, void* out, int32_t out_layout,
                                int32_t* levels_out, mrcnn_stream_t stream);
/* The same for the fixed-shape pipeline, whose rois tensor holds rois_per_image SLOTS per image of which only the first
 * roi_counts[image] carry a proposal (model.py:1366-1374: the reference's rois tensor simply ends after the boxes NMS kept):
 * slots beyond the count are skipped — nothing is read, their output rows are left untouched. roi_counts int32 [num_rois /
 * rois_per_image] device memory, or NULL (= mrcnn_roi_align_pyramid_f32); needs roi_batch == NULL. */
int mrcnn_roi_align_pyramid_counted_f32(const float* const fm[4], const int32_t fm_h[4], const int32_t fm_w[4], int32_t batch,
                                        int32_t depth, const float* rois, const int32_t* roi_batch, int32_t num_rois,
                                        int32_t rois_per_image, const int32_t* roi_counts, int32_t pool, float image_area,
                                        void* out, int32_t out_layout, int32_t* levels_out, mrcnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused convolution + bias/BatchNorm affine + residual + ReLU, channels-last, fp32 MFMA implicit
 * GEMM — the building block of Bottleneck.forward (model.py:190-211), the stem (model.py:223-226),
 * FPN laterals/smoothing (model.py:145-157), RPN and head convolutions. No reference kernel exists
 * for it (the reference calls nn.Conv2d / BatchNorm2d / ReLU / F.pad as separate modules).
 *
 *   y[b,oy,ox,co] = act( scale[co] * sum_{ky,kx,ci} x[b, oy*stride+ky-pad_top, ox*stride+kx-pad_left, ci]
 *                                                  * w[co,ky,kx,ci]  + shift[co]  (+ residual) )
 *   x        fp32 [batch][H][W][cin]           (NHWC)   cin % 4 == 0
 *   w        fp32 [cout][kh][kw][cin]          (OHWI)
 *   scale    fp32 [cout] or NULL (=1);  shift fp32 [cout] or NULL (=0)
 *            (BN eval + conv bias folded by the caller: scale = g/sqrt(var+eps),
 *             shift = (bias-mean)*scale + beta; kept as an fp32 epilogue, not folded into w)
 *   residual fp32 [batch][OH/res_div][OW/res_div][cout] or NULL, added before the activation;
 *            res_div = 1 (Bottleneck `out += residual`) or 2 (FPN top-down: nearest 2x upsample,
 *            model.py:150-152)
 *   activation 0 none, 1 ReLU, 2 sigmoid (1/(1+exp(-v)), Mask.forward's last op, model.py:914)
 *   y        fp32 [batch][OH][OW][cout],  OH = (H + pad_top + pad_bottom - kh)/stride + 1, same for OW
 * Zero padding is applied on the fly (SamePad2d, model.py:64-87, never materialised).
 * ---------------------------------------------------------------------------------------------- */
int mrcnn_conv_bn_act_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width,
                               int32_t cin, const float* w, int32_t cout, int32_t kh, int32_t kw,
                               int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom,
                               int32_t pad_right, const float* scale, const float* shift,
                               const float* residual, int32_t res_div, int32_t activation, float* y,
                               mrcnn_stream_t stream);

/* Same contract as mrcnn_conv_bn_act_nhwc_f32, but the contraction runs on fp16-operand MFMA
 * (v_mfma_f32_32x32x16_f16, fp32 accumulate). x, residual and y stay fp32 in HBM; x is converted while it is
 * staged to LDS. The caller prepares the weights once as fp16 planes [cout][kh][kw][cin]:
 *     w_hi = fp16(w),   w_lo = fp16(w - fp32(w_hi))
 *   products = 1   plain fp16 operands, x_hi*w_hi (BASELINE config 5's "fp16 MFMA path"); w_lo may be NULL.
 *                  Error ~2^-11 relative per term: does NOT meet the 1e-4 bar of the fp32 path.
 *   products = 3   error-compensated split x_hi*w_hi + x_hi*w_lo + x_lo*w_hi: every product exact in fp32,
 *                  fp32 accumulation, ~2^-21 relative per term (fp32-grade) for |x|, |w| < 65504.
 * cin % 8 == 0 (the stem pads 3 -> 8 channels). */
int mrcnn_conv_bn_act_nhwc_f16mfma(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                   const void* w_hi, const void* w_lo, int32_t cout, int32_t kh, int32_t kw,
                                   int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom,
                                   int32_t pad_right, const float* scale, const float* shift,
                                   const float* residual, int32_t res_div, int32_t activation, int32_t products,
                                   float* y, mrcnn_stream_t stream);

/* 2x2 stride-2 transposed convolution + bias + activation, NHWC — Mask.forward's `deconv` + ReLU
 * (nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2), model.py:864,906-912) as ONE GEMM whose epilogue scatters
 * straight into the up-sampled tensor (no pixel-shuffle copy):
 *     y[b, 2i+dy, 2j+dx, co] = act( sum_ci x[b,i,j,ci] * w[(dy*2+dx)*cout + co][ci] + bias4[(dy*2+dx)*cout + co] )
 *   x [batch][H][W][cin];  w fp32 [4*cout][1][1][cin] (the caller repacks the [cin][cout][2][2] weight once);
 *   bias4 fp32 [4*cout] (the bias repeated for the four taps);  y [batch][2H][2W][cout].
 * The _f16mfma form takes the split fp16 weight planes and `products` like mrcnn_conv_bn_act_nhwc_f16mfma. */
int mrcnn_deconv2x2_bias_act_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                      const float* w, int32_t cout, const float* bias4, int32_t activation,
                                      float* y, mrcnn_stream_t stream);
int mrcnn_deconv2x2_bias_act_nhwc_f16mfma(const float* x, int32_t batch, int32_t height, int32_t width,
                                          int32_t cin, const void* w_hi, const void* w_lo, int32_t cout,
                                          const float* bias4, int32_t activation, int32_t products, float* y,
                                          mrcnn_stream_t stream);

/* The plain-fp16 mode (products = 1) with fp16 ACTIVATIONS in HBM — BASELINE configs[4]'s "fp16 MFMA path":
 *   x_is_f16 / y_is_f16 choose the storage type of the input and of the output (and of the residual, which has the
 *   output's type); at least one of them is set (fp32 both ways is mrcnn_conv_bn_act_nhwc_f16mfma). fp16 tensors are
 *   NHWC like their fp32 counterparts. w_f16 = the fp16 weight plane [cout][kh][kw][cin]. Supported combinations:
 *     fp32 → fp16   no residual, activation 0/1 (the stem, the first conv behind RoIAlign)
 *     fp16 → fp16   residual (res_div 1 or 2) or none, activation 0/1; cin % 32 == 0
 *     fp16 → fp32   no residual, activation 0/1/2 (FPN smoothing, RPN heads, the FC layers, the mask sigmoid)
 *   cout must be even for an fp16 output (channel pairs are stored as 4-byte words). A conv that reads an fp16 tensor
 *   sees exactly the operand the fp32-storage path rounds to, so only residual adds, pooling and stores differ.
 * mrcnn_deconv2x2_bias_act_nhwc_f16io: the 2x2 stride-2 transposed conv, fp16 in and out.
 * mrcnn_maxpool_nhwc_f16: the zero-padded max-pool on fp16 NHWC (channels % 8 == 0). */
int mrcnn_conv_bn_act_nhwc_f16io(const void* x, int32_t x_is_f16, int32_t batch, int32_t height, int32_t width,
                                 int32_t cin, const void* w_f16, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                                 int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right,
                                 const float* scale, const float* shift, const void* residual, int32_t res_div,
                                 int32_t activation, void* y, int32_t y_is_f16, mrcnn_stream_t stream);
int mrcnn_deconv2x2_bias_act_nhwc_f16io(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                        const void* w_f16, int32_t cout, const float* bias4, int32_t activation,
                                        void* y_f16, mrcnn_stream_t stream);
/* The pipelined form of the plain-fp16 conv (csrc/conv_f16p.hip: eight waves, LDS-DMA staging kept in flight across barriers,
 * v_mfma_f32_16x16x32_f16) — what the "f16" mode runs wherever the shape allows. fp16 NHWC input, cin % 64 == 0, cout % 64 == 0,
 * kh*kw <= 25, any stride, 0 <= pad_top < kh, 0 <= pad_left < kw;
 *   y = act(conv(x, w) * scale + shift + residual)
 * with an optional fp16 residual [batch][OH/res_div][OW/res_div][cout], res_div 1 or 2 (2 = FPN nearest-upsample-add, even OH / OW),
 * written as fp16 (y_f16) and / or fp32 (y_f32) NHWC — at least one of them non-null (the FPN smoothing convs write both: fp32
 * for RoIAlign, fp16 for the RPN). activation 0 none / 1 ReLU.
 * tile_rows x tile_cols: 0 = automatic (from M, cout, K and the CU count), or 128 / 160 / 192 / 256 pixels x 256 channels (one
 * workgroup per CU), or 128 / 256 pixels x 128 / 64 channels (128-pixel ones: two workgroups per CU).
 * Same operands and k order per output element as mrcnn_conv_bn_act_nhwc_f16io.
 * mrcnn_conv_f16_pipelined_supported: 1 when the shape is in range (incl. the 32-bit byte-offset limits), else 0. */
int mrcnn_conv_f16_pipelined_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t cout, int32_t kh,
                                       int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom,
                                       int32_t pad_right);
int mrcnn_conv_f16_pipelined(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin, const void* w_f16,
                             int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                             int32_t pad_bottom, int32_t pad_right, const float* scale, const float* shift,
                             const void* residual_f16, int32_t res_div, int32_t activation, void* y_f16, float* y_f32,
                             int32_t tile_rows, int32_t tile_cols, mrcnn_stream_t stream);
/* The RPN's shared conv with its two 1x1 heads (model.py:605-607,624-641) on the pipelined fp16 kernel, one launch: stride 1,
 * cout % 256 == 0; act(conv(x, w) * scale + shift) is rounded to fp16 and multiplied, while still in registers, by w_head_f16
 * [32][cout] (rows 0-17: conv_class then conv_bbox weights, rows 18-31 zero). head_part fp32 [cout / 256][M][32], M = batch * OH *
 * OW pixels in image order: the sums over each 256-channel tile, without the head bias — for cout = 512 the input form 4 of
 * mrcnn_rpn_scores_deltas_v2_f32, which adds the two planes and the bias. Fully overwritten. The activation itself is not stored. */
int mrcnn_conv_f16_pipelined_heads(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                   const void* w_f16, int32_t cout, int32_t kh, int32_t kw, int32_t pad_top, int32_t pad_left,
                                   int32_t pad_bottom, int32_t pad_right, const float* scale, const float* shift,
                                   int32_t activation, const void* w_head_f16, float* head_part, int32_t tile_rows,
                                   mrcnn_stream_t stream);
int mrcnn_maxpool_nhwc_f16(const void* x, int32_t batch, int32_t height, int32_t width, int32_t channels,
                           int32_t kernel, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom,
                           int32_t pad_right, void* y, mrcnn_stream_t stream);

/* Zero-padded max-pool, NHWC fp32: [batch][H][W][C] -> [batch][OH][OW][C], OH = (H+pad_top+pad_bottom-k)/s+1.
 * Covers the stem's SamePad2d(3,2) + MaxPool2d(3,2) (model.py:227-228; pads (0,1,0,1) on even sizes — the
 * caller computes the pads with the reference's formula, model.py:75-87) and P6 = MaxPool2d(1,2)
 * (model.py:109,161). Out-of-range taps read 0, exactly as F.pad(..., 'constant', 0) then max. C % 4 == 0. */
int mrcnn_maxpool_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t channels,
                           int32_t kernel, int32_t stride, int32_t pad_top, int32_t pad_left,
                           int32_t pad_bottom, int32_t pad_right, float* y, mrcnn_stream_t stream);
/* The same with a selectable output layout (MRCNN_LAYOUT_NHWC / MRCNN_LAYOUT_KBLOCKED [C/8][batch*OH*OW][8], C % 8 == 0:
 * P6 for the RPN's Winograd conv). */
int mrcnn_maxpool_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t channels, int32_t kernel,
                      int32_t stride, int32_t pad_top, int32_t pad_left, int32_t pad_bottom, int32_t pad_right, float* y,
                      int32_t y_layout, mrcnn_stream_t stream);

/* Bottleneck.forward (/root/reference/model.py:190-211, residual branch :254-262) for the ResNet C2 blocks of the plain-fp16
 * path in ONE launch (csrc/bottleneck_f16.hip): planes = 64, stride 1, fp16 NHWC in and out,
 *   y = relu(conv3(relu(conv2(relu(conv1(x)*s1+t1))*s2+t2))*s3+t3 + residual),   conv2 = SamePad2d(3,1) + 3x3,
 * residual = x (cin = 256, wd_frags null) or fp16(conv_d(x)*sd+td) (cin = 64: the stage's first block, 1x1 downsample conv + BN).
 * Both 64-channel intermediates are rounded to fp16 where the per-layer path (mrcnn_conv_bn_act_nhwc_f16io /
 * mrcnn_conv_f16_pipelined, one launch per conv) rounds them; the two paths differ by the summation order inside an MFMA only.
 * Weights arrive as A fragments made by mrcnn_pack_afrags_f16 from the fp16 OHWI tensors flattened to [cout][k]
 * (k = kh*kw*cin): w1 [64][cin], w2 [64][576], w3 [256][64], wd [256][64]; s / t: the folded BN scale and shift (fp32, per output
 * channel). Image i of a batch == image i alone (a tile never spans images, one kernel for every batch size).
 * _supported: 1 when the shape is in range (planes 64, cin 256 / 64 as above, 32-bit byte offsets), else 0. */
int mrcnn_pack_afrags_f16(const void* w_f16, int32_t cout, int32_t k, void* frags_f16, mrcnn_stream_t stream);
int mrcnn_bottleneck_c2_f16_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes,
                                      int32_t has_downsample);
int mrcnn_bottleneck_c2_f16(const void* x_f16, int32_t batch, int32_t height, int32_t width, int32_t cin, const void* w1_frags,
                            const float* s1, const float* t1, const void* w2_frags, const float* s2, const float* t2,
                            const void* w3_frags, const float* s3, const float* t3, const void* wd_frags, const float* sd,
                            const float* td, void* y_f16, mrcnn_stream_t stream);

/* The tail of Mask.forward (/root/reference/model.py:906-914) of the plain-fp16 path in ONE launch (csrc/mask_tail_f16.hip):
 *   y = sigmoid(conv5(relu(deconv(x) + bias_de)) + bias5)
 * x fp16 NHWC [rois][height][width][256]; deconv = ConvTranspose2d(256, 256, kernel 2, stride 2) given as the A fragments
 * (mrcnn_pack_afrags_f16) of its GEMM form [4*256][256], output row (dy*2 + dx)*256 + co (the weight mrcnn_deconv2x2_bias_act_nhwc_f16io
 * takes), bias_de4 [4*256] (the bias repeated per sub-pixel); conv5 1x1 256 -> classes (<= 96) given as the A fragments of its
 * weight zero-padded to [96][256], bias5 [classes]; y fp32 NHWC [rois][2*height][2*width][classes], fully overwritten.
 * The deconv's output is rounded to fp16 where the two-launch path rounds it (its HBM tensor) but never stored.
 * _supported: 1 when the shape is in range (cin 256, cout_deconv 256, classes <= 96, 32-bit byte offsets), else 0. */
int mrcnn_mask_tail_f16_supported(int32_t rois, int32_t height, int32_t width, int32_t cin, int32_t cout_deconv, int32_t classes);
int mrcnn_mask_tail_f16(const void* x_f16, int32_t rois, int32_t height, int32_t width, int32_t cin, const void* wde_frags,
                        const float* bias_de4, int32_t cout_deconv, const void* w5_frags, const float* bias5, int32_t classes,
                        float* y_f32, mrcnn_stream_t stream);

/* RPN glue, two launches.
 * mrcnn_rpn_scores_deltas_f32 — replaces the per-level permute/view/softmax/cat of RPN.forward + rpn_detect
 *   (model.py:627-641,1294-1304): heads[l] = fused head output of level l, NHWC [batch][H_l][W_l][18]
 *   (channels 0-5: (bg,fg) logits of the 3 anchor ratios, 6-17: their 4 deltas); level_hw[l] = H_l*W_l.
 *   scores [batch][A] = softmax(bg,fg)[1], deltas [batch][A][4], A = 3*sum(level_hw), anchor index
 *   = first(l) + (y*W_l + x)*3 + ratio — the reference's order (utils.py:154-160).
 * mrcnn_proposal_decode_f32 — replaces the gather + boxes_refine + boxes_clamp_ of rpn_refine
 *   (model.py:1341-1358, data.py:124-148,86-92): for the top-k anchor indices `order` [batch][k] (int64) and
 *   their scores, dets [batch][k][5] = (clip(refine(anchor, delta*std_dev), [0,H]x[0,W]), score), in the
 *   reference's fp32 op order. */
int mrcnn_rpn_scores_deltas_f32(const float* const heads[5], const int32_t level_hw[5], int32_t batch,
                                float* scores, float* deltas, mrcnn_stream_t stream);
/* The same with a per-level input form (level_mode[l]): 0 = NHWC [batch][H_l][W_l][18] head outputs as above;
 * 1 / 2 = the head sums of mrcnn_conv3x3_winograd_heads_f32 in tile mode 1 / 2: [2][rows][32] fp32 in that function's row
 * order, bias not yet added: logits/deltas = (sum of the two k halves) + head_bias[c]; 3 = the head sums of
 * mrcnn_conv3x3_winograd4_heads_f32: [rows][32], logits/deltas = sum + head_bias[c]; 4 = the two planes of
 * mrcnn_conv_f16_pipelined_heads (cout = 512): [2][batch*H_l*W_l][32] in pixel order, (plane 0 + plane 1) + head_bias[c].
 * level_h/level_w: H_l, W_l. */
int mrcnn_rpn_scores_deltas_v2_f32(const float* const heads[5], const int32_t level_h[5], const int32_t level_w[5],
                                   const int32_t level_mode[5], const float* head_bias, int32_t batch, float* scores,
                                   float* deltas, mrcnn_stream_t stream);
int mrcnn_proposal_decode_f32(const float* anchors, const float* deltas, const int64_t* order,
                              const float* top_scores, int32_t batch, int32_t num_anchors, int32_t k,
                              const float std_dev[4], float image_height, float image_width, float* dets,
                              mrcnn_stream_t stream);

/* Detection decode — the first half of MaskRCNN.mrn_refine (model.py:1405-1443) for a whole batch in one launch:
 * softmax + arg-max over classes, class-specific delta gather, boxes_refine(rois, delta*std_dev) (data.py:124-148),
 * scale to pixels, clip to each image's window, round (half to even), validity = class > 0 and slot <
 * roi_counts[image] (and score >= min_confidence when min_confidence > 0; config.py:204 sets 0 = no filter).
 *   logits [batch*P][num_classes] (row stride logit_stride elements), bbox [batch*P][num_classes][4] (row stride
 *   bbox_stride), rois [batch*P][4] normalised, windows [batch][4] pixels.
 *   dets [batch*P][5] = (y1,x1,y2,x2,score), class_ids int64 [batch*P] = arg-max class,
 *   nms_class_ids int32 [batch*P] = class for valid slots, a unique negative value otherwise (so that excluded
 *   slots neither suppress nor are suppressed in the class-aware NMS that follows). */
int mrcnn_detection_decode_f32(const float* logits, int64_t logit_stride, const float* bbox, int64_t bbox_stride,
                               const float* rois, const int32_t* roi_counts, const float* windows, int32_t batch,
                               int32_t rois_per_image, int32_t num_classes, const float std_dev[4],
                               float image_height, float image_width, float min_confidence, float* dets,
                               int32_t* nms_class_ids, int64_t* class_ids, mrcnn_stream_t stream);

/* 3x3 stride-1 SAME convolution + affine + ReLU as a fused Winograd F(2x2,3x3) on the fp32 MFMA: 2.25x fewer
 * multiply-adds than mrcnn_conv_bn_act_nhwc_f32 for the same layer, all arithmetic in fp32 (the result differs by the
 * transforms' rounding, a few 1e-7 relative per term). x [batch][H][W][cin] -> y [batch][H][W][cout], H and W even,
 * cin % 8 == 0. u = the filter transformed once by mrcnn_winograd_weights_f32: w [cout][3][3][cin] (OHWI) ->
 * u (16*cout*cin floats, k-blocked [cin/8][16][cout][8]). activation 0 (none) or 1 (ReLU). workspace: device memory of
 * mrcnn_conv3x3_winograd_workspace_bytes(...) bytes (the k-blocked copy of x the kernel reads). */
int mrcnn_winograd_weights_f32(const float* w, int32_t cout, int32_t cin, float* u, mrcnn_stream_t stream);
/* The same convolution with explicit tensor layouts, so that chains of 3x3 convs skip the transposition pass:
 * x_layout MRCNN_LAYOUT_NHWC (workspace needed) or MRCNN_LAYOUT_KBLOCKED = [cin/8][batch][H][W][8] (what the kernel
 * reads; no workspace). Outputs: y_nhwc and/or y_kblocked ([cout/8][batch][H][W][8], cout % 8 == 0); either may be
 * NULL, not both. mrcnn_conv_bn_act_f32 is mrcnn_conv_bn_act_nhwc_f32 with selectable output and residual layouts
 * (a k-blocked output feeds a Winograd conv directly; the FPN laterals read their half-size residual k-blocked); mrcnn_nhwc_to_kblocked_f32 is the standalone transposition. */
#define MRCNN_LAYOUT_NHWC 0
#define MRCNN_LAYOUT_KBLOCKED 1
#define MRCNN_LAYOUT_NHWC_F16 2 /* mrcnn_roi_align_pyramid_f32's output only */
int mrcnn_conv3x3_winograd_f32(const float* x, int32_t x_layout, int32_t batch, int32_t height, int32_t width,
                               int32_t cin, const float* u, int32_t cout, const float* scale, const float* shift,
                               int32_t activation, float* y_nhwc, float* y_kblocked, void* workspace,
                               size_t workspace_bytes, mrcnn_stream_t stream);
/* Winograd F(4x4, 3x3): the same convolution with 4x (instead of 2.25x) fewer multiply-adds, for maps whose height and
 * width are multiples of 4 (csrc/conv_wino4.hip; fp32 arithmetic, max |err| about 2e-5 at unit scale against the direct
 * convolution — ten times F(2x2)'s, inside the 1e-4 parity bar). Replaces the same reference lines as
 * mrcnn_conv3x3_winograd_f32 (model.py:154-157, :605,624).
 *   mrcnn_winograd4_weights_f32   w_ohwi [Cout][3][3][Cin] -> u = G g G^T in the kernel's order [Cin/4][36][2][Cout][2]
 *                                 (36 * Cout * Cin floats; evaluated in double); Cin % 4 == 0
 *   mrcnn_conv3x3_winograd4_supported   1 when H % 4 == 0, W % 4 == 0, Cin % 8 == 0, Cout % 64 == 0 and the tensors stay
 *                                 inside the kernel's 32-bit byte offsets (B*H*W*Cin, B*H*W*Cout < 2^30 elements)
 *   mrcnn_conv3x3_winograd4_f32   x k-blocked [Cin/8][B*H*W][8] -> relu?(conv * scale + shift) as NHWC and/or k-blocked
 *                                 [Cout/8][B*H*W][8] (either output pointer may be null, not both) */
int mrcnn_winograd4_weights_f32(const float* w_ohwi, int32_t cout, int32_t cin, float* u, mrcnn_stream_t stream);
int32_t mrcnn_conv3x3_winograd4_supported(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t cout);
int mrcnn_conv3x3_winograd4_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                const float* u, int32_t cout, const float* scale, const float* shift, int32_t activation,
                                float* y_nhwc, float* y_kblocked, mrcnn_stream_t stream);
/* conv2 + conv3 of a ResNet Bottleneck in one launch (model.py:197-209): relu(scale * conv3x3_same(x) + shift) with 64 output
 * channels stays on chip (the F(4x4) kernel's epilogue) and is multiplied by the 1x1 expansion's weights there:
 *     y = relu(scale3 * (that W3^T) + shift3 + residual)
 *   x k-blocked [Cin/8][B*H*W][8]; u = mrcnn_winograd4_weights_f32 of the [64][3][3][Cin] conv2 weight; w3 fp32 [c3][64]
 *   (OHWI of the 1x1 conv), c3 % 32 == 0; residual and y NHWC [B][H][W][c3], residual != y. The conv3 part runs the
 *   direct kernel's k order and epilogue expression: equal bit for bit to mrcnn_conv3x3_winograd4_f32 followed by
 *   mrcnn_conv_bn_act_f32 with the same residual. Same shape limits as mrcnn_conv3x3_winograd4_f32 with Cout = 64. */
int mrcnn_conv3x3_winograd4_conv3_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                      const float* u, const float* scale, const float* shift, const float* w3, int32_t c3,
                                      const float* scale3, const float* shift3, const float* residual, float* y,
                                      mrcnn_stream_t stream);
/* The F(4x4) form of mrcnn_conv3x3_winograd_heads_f32 (RPN shared conv + its two 1x1 heads, model.py:605-641, in one
 * launch): head_part fp32 [rows][32], rows = mrcnn_conv3x3_winograd4_heads_rows(batch, H, W); row of pixel (b,y,x) =
 * mt*512 + ((y/4 & 3)*8 + (x/4 & 7))*16 + (y&3)*4 + (x&3), mt = (b*ceil(H/16) + y/16)*ceil(W/32) + x/32; the sums over
 * all cout channels without the head bias — input form 3 of mrcnn_rpn_scores_deltas_v2_f32. Fully overwritten. */
int64_t mrcnn_conv3x3_winograd4_heads_rows(int32_t batch, int32_t height, int32_t width);
int mrcnn_conv3x3_winograd4_heads_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                      const float* u, int32_t cout, const float* scale, const float* shift,
                                      int32_t activation, const float* w_head32, float* head_part, mrcnn_stream_t stream);

/* Tile shape of mrcnn_conv3x3_winograd_f32 on maps of at least 8 x 8 tile positions: 1 (default) = 8 x 8 position blocks of
 * one image with the input transform done per lane out of a raw LDS region (conv3x3_wino8s_f32), 0 = 64 consecutive
 * positions with a staged transform (conv3x3_wino8_f32), -1 = back to the default / MRCNN_WINO_SPATIAL. Same results bit
 * for bit; a process-wide tuning switch, not per stream. */
int mrcnn_winograd_set_spatial(int32_t on);
int mrcnn_conv_bn_act_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin, const float* w,
                          int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                          int32_t pad_bottom, int32_t pad_right, const float* scale, const float* shift,
                          const float* residual, int32_t res_div, int32_t residual_layout, int32_t activation,
                          float* y, int32_t y_layout, mrcnn_stream_t stream);
/* mrcnn_conv_bn_act_nhwc_f32 for the heads' GEMMs over [image][RoI slot] rows (Classifier.forward, model.py:782-794, runs on
 * the rois that survived NMS only — model.py:1366-1374): the output rows (batch * OH * OW of them) come in groups of
 * rows_per_group slots of which the first row_counts[group] are valid; an M tile without a valid row is skipped — nothing
 * is read, its output rows are left untouched. Rows are independent in a GEMM, so the valid rows' results are bit-identical to
 * the unmasked call. row_counts int32 device memory, or NULL (= every row). */
int mrcnn_conv_bn_act_rows_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin, const float* w,
                               int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad_top, int32_t pad_left,
                               int32_t pad_bottom, int32_t pad_right, const float* scale, const float* shift,
                               int32_t activation, float* y, const int32_t* row_counts, int32_t rows_per_group,
                               mrcnn_stream_t stream);
/* Rows of the M tile mrcnn_conv_bn_act_rows_f32 gives a layer with `cout` output channels (the skip rule's granularity: a caller
 * that accounts for the work that ran — bench.py's roofline pass — counts tiles of this many rows). */
int32_t mrcnn_conv_rows_tile_m(int32_t cout);
int mrcnn_nhwc_to_kblocked_f32(const float* x, int64_t pixels, int32_t channels, float* y, mrcnn_stream_t stream);
size_t mrcnn_conv3x3_winograd_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cin);
int mrcnn_conv3x3_winograd_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                    const float* u, int32_t cout, const float* scale, const float* shift,
                                    int32_t activation, float* y, void* workspace, size_t workspace_bytes,
                                    mrcnn_stream_t stream);

/* The ResNet stem as its own kernel (model.py:223-226): conv 7x7 stride 2 pad 3, 4 input channels (RGB + one zero
 * channel) -> 64, + affine + ReLU. x [batch][H][W][4], w [64][7][7][4] (OHWI), y [batch][H/2][W/2][64]; H, W even.
 * Same arithmetic as mrcnn_conv_bn_act_nhwc_f32 on this layer (exact fp32 MFMA; the accumulation order over k is
 * identical), 3x faster: filter resident in LDS, input patch staged per 16x16 output tile, no addressing in the loop. */
int mrcnn_stem_conv7x7_s2_nhwc_f32(const float* x, int32_t batch, int32_t height, int32_t width, const float* w,
                                   const float* scale, const float* shift, int32_t activation, float* y,
                                   mrcnn_stream_t stream);
/* The same reading the molded image itself, x_nchw [batch][3][H][W] (model.py:1102-1110): no NCHW -> NHWC4 pass over the image
 * beforehand. w stays [64][7][7][4] (channel 3 zero). Bit-identical to the NHWC form on the converted image. */
int mrcnn_stem_conv7x7_s2_nchw_f32(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                   const float* scale, const float* shift, int32_t activation, float* y,
                                   mrcnn_stream_t stream);
/* The same with the output stored as fp16 NHWC (the "f16" mode's trunk input): fp32 products and accumulation as above, one
 * rounding at the store. */
int mrcnn_stem_conv7x7_s2_nchw_f16out(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                      const float* scale, const float* shift, int32_t activation, void* y_f16,
                                      mrcnn_stream_t stream);
/* The exact-fp32 stem with its max-pool in ONE launch (round 5): conv 7x7 s2 p3 + affine + ReLU + SamePad2d(3, 2) + MaxPool2d(3, 2)
 * (model.py:223-229) on the fp32 MFMA — x_nchw fp32 [batch][3][H][W] (H, W multiples of 4), w fp32 OHWI [64][7][7][4] (channel 3
 * zero, never multiplied: K = 7 x 21 real values), y fp32 NHWC [batch][ceil(H/4)][ceil(W/4)][64]. Exact fp32 products, fp32
 * accumulation in the order (ky, kx, c); the max is exact, so the result equals conv-then-pool of the same sums. The
 * full-resolution 64-channel map never reaches memory. ReLU is part of the contract (zero padding of the pool). */
int mrcnn_stem_conv7x7_s2_pool_f32(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                   const float* scale, const float* shift, float* y, mrcnn_stream_t stream);
/* The "f16" mode's stem in ONE launch on the fp16 MFMA: conv 7x7 s2 p3 + affine + ReLU + SamePad2d(3, 2) + MaxPool2d(3, 2)
 * (model.py:223-229) — x_nchw fp32 [batch][3][H][W] (H, W multiples of 4), w fp32 OHWI [64][7][7][4] (channel 3 zero; rounded to fp16
 * inside), y_f16 fp16 NHWC [batch][ceil(H/4)][ceil(W/4)][64]. Image and weights are rounded to fp16 once, products accumulate
 * in fp32, the conv output is rounded to fp16 once before the max (as the two-launch form stores it). The full-resolution
 * 64-channel map never reaches memory. ReLU is part of the contract (the pool's zero padding is neutral only for values >= 0). */
int mrcnn_stem_conv7x7_s2_pool_f16(const float* x_nchw, int32_t batch, int32_t height, int32_t width, const float* w,
                                   const float* scale, const float* shift, void* y_f16, mrcnn_stream_t stream);

/* ---- selection steps of the two refine stages (no library sort / top-k / gather in the step) --------------------
 * Total, deterministic order everywhere: descending score, ties by ascending index (ATen's sort, which the
 * reference calls at model.py:1346,1478, leaves ties unspecified).
 * mrcnn_topk_desc_f32 — replaces `scores.sort(descending=True)` + `[:pre_nms_limit]` of rpn_refine
 *   (model.py:1345-1350): scores [batch][n] -> top_scores [batch][k], order int64 [batch][k]. k <= 4096, k <= n;
 *   workspace >= mrcnn_topk_workspace_bytes(batch) bytes of device memory (contents irrelevant).
 * mrcnn_proposal_select_f32 — replaces keep[:proposal_count] + gather + normalise (model.py:1366-1374):
 *   dets [batch][k][5], keep int64 [batch][k] (NMS output, ascending, padded) with keep_counts [batch] ->
 *   rois [batch][proposal_count][4] = box / (H,W,H,W), zero beyond counts[b] = min(keep_counts[b], proposal_count).
 * mrcnn_detection_select_f32 — replaces the tail of mrn_refine (model.py:1475-1487) and the mask-head box
 *   normalisation (model.py:1188): among the RoIs the class-aware NMS kept (keep / keep_counts, rows of
 *   rois_per_image) that carry a foreground class (nms_class_ids > 0), the max_instances highest scores ->
 *   out_class_ids int64, out_scores, out_boxes [batch][max_instances][4] (pixels), out_rois = boxes / (H,W,H,W),
 *   out_counts [batch]; unused slots are zero. rois_per_image <= 4096. */
size_t mrcnn_topk_workspace_bytes(int32_t batch);
int mrcnn_topk_desc_f32(const float* scores, int32_t batch, int64_t n, int32_t k, float* top_scores, int64_t* order,
                        void* workspace, size_t workspace_bytes, mrcnn_stream_t stream);
int mrcnn_proposal_select_f32(const float* dets, const int64_t* keep, const int32_t* keep_counts, int32_t batch,
                              int32_t k, int32_t proposal_count, float image_height, float image_width, float* rois,
                              int32_t* counts, mrcnn_stream_t stream);
int mrcnn_detection_select_f32(const float* dets, const int32_t* nms_class_ids, const int64_t* class_ids,
                               const int64_t* keep, const int32_t* keep_counts, int32_t batch, int32_t rois_per_image,
                               int32_t max_instances, float image_height, float image_width, int64_t* out_class_ids,
                               float* out_scores, float* out_boxes, float* out_rois, int32_t* out_counts,
                               mrcnn_stream_t stream);

/* Layout conversions at the boundary (reference tensors are NCHW, model.py:1109). */
int mrcnn_nchw_to_nhwc_f32(const float* x, int32_t batch, int32_t channels, int32_t height,
                           int32_t width, int32_t channels_padded, float* y, mrcnn_stream_t stream);
int mrcnn_nhwc_to_nchw_f32(const float* x, int32_t batch, int32_t channels, int32_t height,
                           int32_t width, float* y, mrcnn_stream_t stream);

/* ---- image pre-/post-processing around the hot path --------------------------------------
 * The reference resizes 8-bit images through Pillow (scipy.misc.imresize, utils.py:73; torchvision Resize on a PIL
 * image, data.py:277,295): BILINEAR with the support stretched by the scale factor when shrinking, 22-bit fixed-point
 * coefficients, horizontal then vertical pass with an 8-bit intermediate. These entry points reproduce that
 * arithmetic bit for bit.
 *
 * mrcnn_resize_bilinear_u8: n images of one size, src[i] [in_h][in_w][channels] interleaved uint8 at
 *   src + i*src_image_stride with src_row_stride bytes between rows (so a crop of a larger image — transform.CenterCrop
 *   before Resize in decode_masks, data.py:272-277 — needs no copy) -> dst n x [out_h][out_w][channels] contiguous,
 *   each == numpy.array(Image.fromarray(src[i]).resize((out_w, out_h), Image.BILINEAR)). channels 1..4, sizes <= 16384. */
size_t mrcnn_resize_u8_workspace_bytes(int32_t n, int32_t in_h, int32_t in_w, int32_t channels, int32_t out_h,
                                       int32_t out_w);
int mrcnn_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t in_h, int32_t in_w, int32_t channels,
                             int64_t src_image_stride, int64_t src_row_stride, uint8_t* dst, int32_t out_h,
                             int32_t out_w, void* workspace, size_t workspace_bytes, mrcnn_stream_t stream);
/* Replaces utils.resize_image's resize + pad (utils.py:72-88), mold_image (model.py:1750-1754) and the HWC -> CHW
 * float conversion of detect() (model.py:1108-1110) for one RGB image: src uint8 [in_h][in_w][3] is resized to
 * new_h x new_w (no resample when that equals the input size), placed at (top, left) of a zero canvas out_h x out_w,
 * and dst[c][y][x] = float(double(pixel) - mean_pixel[c]) (the reference subtracts a float64 MEAN_PIXEL array).
 * The caller computes scale / new size / pads exactly as utils.py:56-85 does (host integers).
 * workspace: mrcnn_resize_u8_workspace_bytes(1, in_h, in_w, 3, new_h, new_w) bytes; may be NULL when no resize. */
int mrcnn_mold_image_u8(const uint8_t* src, int32_t in_h, int32_t in_w, int32_t new_h, int32_t new_w, int32_t top,
                        int32_t left, int32_t out_h, int32_t out_w, const double mean_pixel[3], float* dst,
                        void* workspace, size_t workspace_bytes, mrcnn_stream_t stream);
/* The same for n images of ONE size (a batch from one camera / dataset: detect() on a list, model.py:1097-1110): src image i
 * at src + i * src_image_stride, dst n x [3][out_h][out_w]; one set of coefficient tables, one horizontal pass and one
 * vertical + mold pass for the whole batch. workspace: mrcnn_resize_u8_workspace_bytes(n, in_h, in_w, 3, new_h, new_w). */
int mrcnn_mold_images_u8(const uint8_t* src, int32_t n, int64_t src_image_stride, int32_t in_h, int32_t in_w, int32_t new_h,
                         int32_t new_w, int32_t top, int32_t left, int32_t out_h, int32_t out_w, const double mean_pixel[3],
                         float* dst, void* workspace, size_t workspace_bytes, mrcnn_stream_t stream);
/* Replaces datalib.full_masks (data.py:287-314). For detection i: the class_ids[i] channel of its mask_h x mask_w
 * sigmoid mask (element (i,y,x,c) at masks[i*stride_n + y*stride_y + x*stride_x + c*stride_c], so both the
 * reference's [N,C,h,w] and this library's [N,h,w,C] layouts are accepted) is multiplied by 255, converted to 8 bits
 * (clamp, truncate: Image.fromarray(F).convert('L')), resized to the box's int(y2-y1) x int(x2-x1) and pasted at
 * (int(y1), int(x1)) of a zero height x width canvas; out[i][y][x] = on_value where the 8-bit value > 127, else 0
 * (on_value 1: the boolean mask of data.py:308; 255: the same mask as the 'L' image decode_masks starts from).
 * Boxes the reference cannot paste (empty, or not inside the canvas: PIL raises) give an all-zero mask, as do
 * class ids outside [0, num_classes) — so padded detection slots (class 0, zero box) cost nothing and stay empty.
 * mask_h, mask_w <= 64; width % 4 == 0. */
int mrcnn_paste_masks_u8(const float* masks, int64_t stride_n, int64_t stride_y, int64_t stride_x, int64_t stride_c,
                         int32_t n, int32_t mask_h, int32_t mask_w, int32_t num_classes, const int64_t* class_ids,
                         const float* boxes, int32_t height, int32_t width, int32_t on_value, uint8_t* out,
                         mrcnn_stream_t stream);

/* RPN conv_shared + both 1x1 heads in one launch on the Winograd kernel (RPN.forward, model.py:605-607,624-641):
 * relu(conv3x3_same(x) * scale + shift) is never stored — each 64-channel output tile is transposed through LDS and
 * multiplied by w_head32 ([32][cout]: rows 0..17 = conv_class (6) then conv_bbox (12) weights, other rows zero) while
 * on chip; a workgroup owns whole M tiles and adds the contribution of each of its cout/64 N tiles to the M tile's sums.
 *   x_kblocked  fp32 [cin/8][batch][H][W][8];  u = mrcnn_winograd_weights_f32 of the [cout][3][3][cin] filter
 *   tile_mode   1: an M tile is 64 consecutive tile positions; 2: an 8 x 8 block of positions of one image (the faster
 *               kernel; maps of at least 8 x 8 positions). mrcnn_conv3x3_winograd_heads_tile_mode(H, W) returns the
 *               recommended one (follows mrcnn_winograd_set_spatial).
 *   head_part   fp32 [2][rows][32], rows = mrcnn_conv3x3_winograd_heads_rows(batch, H, W, tile_mode): the two k halves of
 *               the sums (without the head bias); row of pixel (b,y,x): mode 1 = ((b*H/2 + y/2)*W/2 + x/2)*4 + (y&1)*2 +
 *               (x&1); mode 2 = mt*256 + ((y/2 & 7)*8 + (x/2 & 7))*4 + (y&1)*2 + (x&1) with mt the block index
 *               (b*ceil(H/16) + y/16)*ceil(W/16) + x/16 — the input forms 1 / 2 of mrcnn_rpn_scores_deltas_v2_f32.
 *               Fully overwritten; no zero-fill needed.
 *   H, W even; cin % 8 == 0; cout % 64 == 0. Deterministic (fixed summation order). */
int64_t mrcnn_conv3x3_winograd_heads_rows(int32_t batch, int32_t height, int32_t width, int32_t tile_mode);
int32_t mrcnn_conv3x3_winograd_heads_tile_mode(int32_t height, int32_t width);
int mrcnn_conv3x3_winograd_heads_f32(const float* x_kblocked, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                     const float* u, int32_t cout, const float* scale, const float* shift,
                                     int32_t activation, const float* w_head32, int32_t tile_mode, float* head_part,
                                     mrcnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-block fused Bottleneck — replaces the module composite  Bottleneck.forward  (model.py:190-211):
 *   out = relu( bn3(conv3(relu(bn2(conv2_same(relu(bn1(conv1(x)))))))) + x )
 * for the stride-1 IDENTITY blocks (no downsample branch) with planes = 64 and Cin = 4 * planes = 256 (ResNet C2 blocks
 * after the first): ONE launch, both 64-channel intermediates stay in LDS (csrc/bottleneck.hip). BatchNorm (eval) and
 * the conv biases are the folded fp32 (scale, shift) epilogues; conv2 runs as Winograd F(2x2,3x3) on the fp32 MFMA.
 * The result equals the three-launch path (mrcnn_conv_bn_act_f32 → mrcnn_conv3x3_winograd_f32 → mrcnn_conv_bn_act_f32
 * with residual) bit for bit.
 *   x   fp32 [batch][height][width][cin] NHWC;  y  fp32 [batch][height][width][4*planes], y != x
 *   w1  [planes][cin] (conv1, OHWI 1x1);  u2 = mrcnn_winograd_weights_f32(conv2 weights [planes][3][3][planes]);
 *   w3  [4*planes][planes] (conv3, OHWI 1x1);  scaleN, shiftN: per output channel of conv N (NULL = 1 / 0)
 * mrcnn_bottleneck_fused_supported() tells whether a shape is covered (planes == 64, cin == 256, height and width
 * multiples of 16); other blocks run the per-layer entry points.
 * ---------------------------------------------------------------------------------------------- */
int mrcnn_bottleneck_fused_supported(int32_t height, int32_t width, int32_t cin, int32_t planes);
int mrcnn_bottleneck_fused_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                               const float* w1, const float* scale1, const float* shift1, const float* u2,
                               const float* scale2, const float* shift2, const float* w3, const float* scale3,
                               const float* shift3, int32_t planes, float* y, mrcnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Bottleneck.forward as ONE call — replaces the module composite  Bottleneck.forward  (model.py:174-211, downsample branch
 * :254-262) for EVERY block of the trunk; the implementation of torch.ops.maskrcnn.bottleneck_forward (csrc/bottleneck_op.hip).
 * The call plans the block and enqueues its launches on `stream`:
 *   ResNet C2 (planes = 64, maps of >= winograd4_min_tiles tiles of 16 x 32 pixels per image, u4 given, fuse_conv3 != 0):
 *       conv1 (+ downsample) on the direct kernel, then conv2 + conv3 + residual + ReLU in ONE launch
 *       (mrcnn_conv3x3_winograd4_conv3_f32): 2 launches per identity block;
 *   elsewhere: conv1 (+ downsample), conv2 by F(4x4) (u4, same size rule) / F(2x2) (u2, even maps) Winograd or, with neither
 *       transform given, the exact direct kernel, then conv3 + residual + ReLU: 3 (4) launches.
 * The plan depends on the shape per image and on which transforms are passed, never on the batch.
 *   x   fp32 NHWC [batch][height][width][cin];  y  fp32 NHWC [batch][ceil(H/s)][ceil(W/s)][4*planes], y != x
 *   weights[14] (device pointers, fp32; scale / shift = folded BatchNorm + bias per output channel, NULL = 1 / 0):
 *       0 w1 [planes][cin]          1 scale1   2 shift1
 *       3 w2 [planes][3][3][planes] (OHWI)     4 u2 = mrcnn_winograd_weights_f32(w2) or NULL     5 u4 = mrcnn_winograd4_weights_f32(w2) or NULL
 *       6 scale2   7 shift2         8 w3 [4*planes][planes]   9 scale3   10 shift3
 *       11 wd [4*planes][cin] or NULL (identity block: stride 1, cin == 4*planes)   12 scale_d   13 shift_d
 *   workspace: >= mrcnn_bottleneck_workspace_bytes(...) bytes of device memory, 256-byte aligned (the block's intermediates).
 * ---------------------------------------------------------------------------------------------- */
size_t mrcnn_bottleneck_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes,
                                        int32_t stride, int32_t has_downsample);
/* The plan mrcnn_bottleneck_forward_f32 follows for these arguments, as bits: 1 = conv2 on the F(4x4) kernel, 2 = conv2 on the
 * F(2x2) kernel (neither: direct kernel), 4 = conv2 + conv3 + residual in one launch; -1 = bad shape. */
int32_t mrcnn_bottleneck_plan(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes, int32_t stride,
                              int32_t have_u2, int32_t have_u4, int32_t winograd4_min_tiles, int32_t fuse_conv3);
int mrcnn_bottleneck_forward_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes,
                                 int32_t stride, const float* const weights[14], int32_t winograd4_min_tiles,
                                 int32_t fuse_conv3, void* workspace, size_t workspace_bytes, float* y, mrcnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKRCNN_HIP_H */
