/* Entry points that exist only in MRCNN_ABLATIONS builds of libmaskrcnn_hip.so (MRCNN_ABLATIONS=1 python maskrcnn_amd/build.py):
 * kernels that were measured against what the default library ships and lost — kept for A/B measurements, not part of the
 * product ABI (include/maskrcnn_hip.h). Also behind the same flag, without an entry point of their own: the four-wave
 * F(2x2) Winograd kernel (MRCNN_WINO_WAVES=4) and the linear-tile heads variant (tile_mode 1 of
 * mrcnn_conv3x3_winograd_heads_f32). */
#ifndef MASKRCNN_HIP_ABLATIONS_H
#define MASKRCNN_HIP_ABLATIONS_H
#include "maskrcnn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One RPN level in two launches — RPN.forward (model.py:609-649) without ever writing the 512-channel shared
 * activation to HBM: SamePad + conv_shared 3x3 (cin -> cout) + bias + ReLU, then BOTH 1x1 heads (conv_class 6 +
 * conv_bbox 12 = head_n 18 channels) applied to the tile while it is still on chip (transposed through LDS); each
 * 128-channel slice of the shared conv writes a partial [M][head_n] to `workspace`, and a small second kernel adds
 * the cout/128 partials in fixed order (deterministic) plus the head bias.
 *   x [batch][H][W][cin] NHWC;  w_shared fp32 [cout][3][3][cin];  b_shared [cout];
 *   w_head32 fp32 [32][cout]: rows 0..head_n-1 = the 1x1 head weights (class rows first), remaining rows zero;
 *   b_head [head_n];  y [batch][H][W][head_n] — the layout mrcnn_rpn_scores_deltas_f32 consumes.
 *   cin % 32 == 0, cout % 128 == 0, head_n <= 32; workspace >= mrcnn_rpn_level_workspace_bytes(...). */
size_t mrcnn_rpn_level_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cout,
                                       int32_t head_n);
int mrcnn_rpn_level_fused_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                              const float* w_shared, int32_t cout, const float* b_shared, const float* w_head32,
                              const float* b_head, int32_t head_n, void* workspace, size_t workspace_bytes, float* y,
                              mrcnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
