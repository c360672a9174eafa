// Bottleneck.forward (model.py:190-211) as ONE call of the C ABI: what torch.ops.maskrcnn.bottleneck_forward runs.
// Host code only — it plans the block and enqueues this library's own entry points on the caller's stream:
//     residual   = x                                   (identity)          or  bn_d(conv1x1_stride_s(x))   (model.py:254-262)
//     h          = relu(bn1(conv1x1_stride_s(x)))      written k-blocked when conv2 takes a Winograd kernel
//     planes 64 on a large map:   y = relu(bn3(conv3(relu(bn2(conv2(h))))) + residual)  in ONE launch
//                                 (mrcnn_conv3x3_winograd4_conv3_f32: conv2 + conv3 + residual, csrc/conv_wino4.hip)
//     otherwise: h2 = relu(bn2(conv2(h))) by F(4x4) / F(2x2) Winograd or the direct kernel, then
//                y  = relu(bn3(conv1x1(h2)) + residual)                     (mrcnn_conv_bn_act_f32)
// i.e. two launches for the ResNet C2 blocks (three with a downsample branch), three (four) elsewhere. The plan is a function
// of the block's shape PER IMAGE and of which conv2 transforms the caller supplies — never of the batch: image i of a batch equals
// image i alone bit for bit. Every intermediate lives in the caller's workspace; nothing is allocated or synchronised here.
#include "common.hpp"

namespace {

constexpr size_t kAlign = 256;
inline size_t aligned(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

struct Plan {
    int oh, ow;
    long long m;           // output pixels
    bool use4, use2, fused3;
    size_t res_off, h_off, h2_off, total;
};

// tiles of 16 x 32 output pixels per image: the F(4x4) kernel's M tiles (the rule modules.py applies to every F(4x4) layer)
inline int wino4_tiles_per_image(int h, int w) { return ((h / 4 + 3) / 4) * ((w / 4 + 7) / 8); }

Plan make_plan(int batch, int height, int width, int cin, int planes, int stride, bool has_ds, bool have_u2, bool have_u4,
               int min_tiles4, bool fuse_conv3) {
    Plan p{};
    p.oh = (height + stride - 1) / stride;   // 1x1, no padding: floor((H - 1) / s) + 1
    p.ow = (width + stride - 1) / stride;
    p.m = 1LL * batch * p.oh * p.ow;
    const bool even = p.oh % 2 == 0 && p.ow % 2 == 0 && planes % 8 == 0;
    p.use4 = have_u4 && have_u2 && even && mrcnn_conv3x3_winograd4_supported(batch, p.oh, p.ow, planes, planes) &&
             wino4_tiles_per_image(p.oh, p.ow) >= min_tiles4;
    p.use2 = !p.use4 && have_u2 && even;
    p.fused3 = p.use4 && fuse_conv3 && planes == 64;
    size_t off = 0;
    p.res_off = off;
    if (has_ds) off += aligned(sizeof(float) * p.m * 4 * planes);
    p.h_off = off;
    off += aligned(sizeof(float) * p.m * planes);
    p.h2_off = off;
    if (!p.fused3) off += aligned(sizeof(float) * p.m * planes);
    p.total = off;
    (void)cin;
    return p;
}

}  // namespace

extern "C" size_t mrcnn_bottleneck_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes,
                                                   int32_t stride, int32_t has_downsample) {
    if (batch < 1 || height < 1 || width < 1 || cin < 1 || planes < 1 || stride < 1) return 0;
    // the largest plan: no fused conv3 (the workspace then serves whichever transforms the caller passes)
    return make_plan(batch, height, width, cin, planes, stride, has_downsample != 0, false, false, 0, false).total;
}

extern "C" int32_t mrcnn_bottleneck_plan(int32_t batch, int32_t height, int32_t width, int32_t cin, int32_t planes, int32_t stride,
                                         int32_t have_u2, int32_t have_u4, int32_t winograd4_min_tiles, int32_t fuse_conv3) {
    if (batch < 1 || height < 1 || width < 1 || cin < 1 || planes < 1 || stride < 1) return -1;
    const Plan p = make_plan(batch, height, width, cin, planes, stride, false, have_u2 != 0, have_u4 != 0, winograd4_min_tiles,
                             fuse_conv3 != 0);
    return (p.use4 ? 1 : 0) | (p.use2 ? 2 : 0) | (p.fused3 ? 4 : 0);
}

extern "C" int mrcnn_bottleneck_forward_f32(const float* x, int32_t batch, int32_t height, int32_t width, int32_t cin,
                                            int32_t planes, int32_t stride, const float* const weights[14],
                                            int32_t winograd4_min_tiles, int32_t fuse_conv3, void* workspace,
                                            size_t workspace_bytes, float* y, mrcnn_stream_t stream) {
    MRCNN_REQUIRE(x && weights && y && workspace, "bottleneck_forward: null pointer");
    MRCNN_REQUIRE(batch >= 1 && height >= 1 && width >= 1 && cin >= 1 && planes >= 1 && stride >= 1,
                  "bottleneck_forward: bad shape B=%d H=%d W=%d Cin=%d planes=%d stride=%d", batch, height, width, cin, planes,
                  stride);
    const float *w1 = weights[0], *s1 = weights[1], *t1 = weights[2], *w2 = weights[3], *u2 = weights[4], *u4 = weights[5],
                *s2 = weights[6], *t2 = weights[7], *w3 = weights[8], *s3 = weights[9], *t3 = weights[10], *wd = weights[11],
                *sd = weights[12], *td = weights[13];
    MRCNN_REQUIRE(w1 && w2 && w3, "bottleneck_forward: conv1 / conv2 / conv3 weights are required");
    const int c3 = 4 * planes;
    MRCNN_REQUIRE(wd != nullptr || (stride == 1 && cin == c3),
                  "bottleneck_forward: an identity block needs stride 1 and Cin == 4 * planes (Cin=%d planes=%d stride=%d)", cin,
                  planes, stride);
    MRCNN_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % kAlign == 0, "bottleneck_forward: workspace must be 256-byte aligned");
    const Plan p = make_plan(batch, height, width, cin, planes, stride, wd != nullptr, u2 != nullptr, u4 != nullptr,
                             winograd4_min_tiles, fuse_conv3 != 0);
    MRCNN_REQUIRE(workspace_bytes >= p.total, "bottleneck_forward: workspace of %zu bytes, %zu needed", workspace_bytes, p.total);
    char* ws = static_cast<char*>(workspace);
    float* h = reinterpret_cast<float*>(ws + p.h_off);
    float* h2 = reinterpret_cast<float*>(ws + p.h2_off);
    const float* res = x;
    if (wd) {  // model.py:254-262: 1x1 stride-s conv + BN, no activation
        float* r = reinterpret_cast<float*>(ws + p.res_off);
        if (int rc = mrcnn_conv_bn_act_f32(x, batch, height, width, cin, wd, c3, 1, 1, stride, 0, 0, 0, 0, sd, td, nullptr, 1,
                                           MRCNN_LAYOUT_NHWC, 0, r, MRCNN_LAYOUT_NHWC, stream))
            return rc;
        res = r;
    }
    const bool kb = p.use4 || p.use2;
    if (int rc = mrcnn_conv_bn_act_f32(x, batch, height, width, cin, w1, planes, 1, 1, stride, 0, 0, 0, 0, s1, t1, nullptr, 1,
                                       MRCNN_LAYOUT_NHWC, 1, h, kb ? MRCNN_LAYOUT_KBLOCKED : MRCNN_LAYOUT_NHWC, stream))
        return rc;
    if (p.fused3)
        return mrcnn_conv3x3_winograd4_conv3_f32(h, batch, p.oh, p.ow, planes, u4, s2, t2, w3, c3, s3, t3, res, y, stream);
    int rc;
    if (p.use4)
        rc = mrcnn_conv3x3_winograd4_f32(h, batch, p.oh, p.ow, planes, u4, planes, s2, t2, 1, h2, nullptr, stream);
    else if (p.use2)
        rc = mrcnn_conv3x3_winograd_f32(h, MRCNN_LAYOUT_KBLOCKED, batch, p.oh, p.ow, planes, u2, planes, s2, t2, 1, h2, nullptr,
                                        nullptr, 0, stream);
    else  // SamePad2d(3, 1) = (1, 1, 1, 1) as a load predicate (model.py:64-87,195)
        rc = mrcnn_conv_bn_act_f32(h, batch, p.oh, p.ow, planes, w2, planes, 3, 3, 1, 1, 1, 1, 1, s2, t2, nullptr, 1,
                                   MRCNN_LAYOUT_NHWC, 1, h2, MRCNN_LAYOUT_NHWC, stream);
    if (rc) return rc;
    return mrcnn_conv_bn_act_f32(h2, batch, p.oh, p.ow, planes, w3, c3, 1, 1, 1, 0, 0, 0, 0, s3, t3, res, 1, MRCNN_LAYOUT_NHWC, 1,
                                 y, MRCNN_LAYOUT_NHWC, stream);
}
