/* ORACLE — TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * The image pre-/post-processing either side of the hot path resizes 8-bit images:
 *   utils.py:73-74   resize_image:  scipy.misc.imresize(image, (round(h*scale), round(w*scale)))
 *   data.py:293-295  full_masks:    Image.fromarray(mask*255.0).convert('L'); transform.Resize((bh, bw))
 *   data.py:277      decode_masks:  transform.Resize((nh, nw))
 * Both wrappers are absent from /root/reference AND from this image (scipy.misc.imresize was removed in
 * scipy 1.3; torchvision is not installed). They are thin: imresize == Image.fromarray(a).resize((w, h), BILINEAR)
 * (scipy 1.0 misc/pilutil.py:imresize, interp='bilinear' default), Resize((h, w)) on a PIL image ==
 * img.resize((w, h), Image.BILINEAR) (torchvision 0.2 transforms/functional.py:resize). The arithmetic lives in
 * Pillow (third-party, un-vendored; installed here: Pillow 12.2.0). This file restates Pillow's published
 * algorithm — src/libImaging/Resample.c: precompute_coeffs(), normalize_coeffs_8bpc(),
 * ImagingResampleHorizontal_8bpc(), ImagingResampleVertical_8bpc(), BILINEAR filter (support 1.0), and
 * src/libImaging/Convert.c f2l() for the float -> 'L' conversion — and is pinned bit-exactly against Pillow
 * itself (tests/test_oracle_image.py live, tests/golden/image_*.npz committed).
 *
 *   coefficients (double):  scale = in/out; filterscale = max(scale, 1); support = 1.0*filterscale;
 *                           ksize = (int)ceil(support)*2 + 1
 *       for each output xx: center = (xx + 0.5)*scale; xmin = max((int)(center - support + 0.5), 0);
 *                           xmax = min((int)(center + support + 0.5), in) - xmin;
 *                           w[x] = tri((x + xmin - center + 0.5) / filterscale), normalised by their sum
 *   8-bit fixed point:      k[x] = (int)(0.5 + w[x]*2^22)          (PRECISION_BITS = 32 - 8 - 2)
 *   pass:                   out = clip8((2^21 + sum in[xmin + x]*k[x]) >> 22)
 *   order:                  horizontal pass into an 8-bit intermediate, then vertical pass.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION_BITS (32 - 8 - 2)

static double tri(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

int oracle_resample_ksize(int in_size, int out_size) {
    double scale = (double)((float)in_size - 0.0f) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = 1.0 * filterscale;
    return (int)ceil(support) * 2 + 1;
}

/* bounds[2*xx] = xmin, bounds[2*xx+1] = tap count; kk[xx*ksize + x] fixed-point coefficients. Returns ksize. */
int oracle_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
    double scale, filterscale, support, center, ww, ss;
    int xx, x, ksize, xmin, xmax;
    filterscale = scale = (double)((float)in_size - 0.0f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    support = 1.0 * filterscale;
    ksize = (int)ceil(support) * 2 + 1;
    double* k = (double*)malloc(sizeof(double) * (size_t)ksize);
    for (xx = 0; xx < out_size; xx++) {
        center = 0.0f + (xx + 0.5) * scale;
        ww = 0.0;
        ss = 1.0 / filterscale;
        xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (x = 0; x < xmax; x++) {
            double w = tri((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        bounds[xx * 2 + 0] = xmin;
        bounds[xx * 2 + 1] = xmax;
        for (x = 0; x < ksize; x++) {
            if (k[x] < 0) kk[xx * ksize + x] = (int)(-0.5 + k[x] * (1 << PRECISION_BITS));
            else kk[xx * ksize + x] = (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        }
    }
    free(k);
    return ksize;
}

static uint8_t clip8(int in) {
    int v = in >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

/* src [in_h][in_w][c] -> dst [out_h][out_w][c], interleaved channels, each channel independently. */
int oracle_resize_bilinear_u8(const uint8_t* src, int in_h, int in_w, int c, uint8_t* dst, int out_h, int out_w) {
    if (in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || c < 1) return -1;
    int ksh = oracle_resample_ksize(in_w, out_w), ksv = oracle_resample_ksize(in_h, out_h);
    int32_t* bh = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)out_w);
    int32_t* kh = (int32_t*)malloc(sizeof(int32_t) * (size_t)ksh * out_w);
    int32_t* bv = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)out_h);
    int32_t* kv = (int32_t*)malloc(sizeof(int32_t) * (size_t)ksv * out_h);
    uint8_t* tmp = (uint8_t*)malloc((size_t)in_h * out_w * c);
    oracle_resample_coeffs(in_w, out_w, bh, kh);
    oracle_resample_coeffs(in_h, out_h, bv, kv);
    for (int y = 0; y < in_h; y++)
        for (int xx = 0; xx < out_w; xx++) {
            const int xmin = bh[2 * xx], xmax = bh[2 * xx + 1];
            const int32_t* k = kh + (size_t)xx * ksh;
            for (int ch = 0; ch < c; ch++) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int x = 0; x < xmax; x++) ss += src[((size_t)y * in_w + x + xmin) * c + ch] * k[x];
                tmp[((size_t)y * out_w + xx) * c + ch] = clip8(ss);
            }
        }
    for (int yy = 0; yy < out_h; yy++) {
        const int ymin = bv[2 * yy], ymax = bv[2 * yy + 1];
        const int32_t* k = kv + (size_t)yy * ksv;
        for (int xx = 0; xx < out_w * c; xx++) {
            int ss = 1 << (PRECISION_BITS - 1);
            for (int y = 0; y < ymax; y++) ss += tmp[((size_t)(y + ymin) * out_w * c) + xx] * k[y];
            dst[(size_t)yy * out_w * c + xx] = clip8(ss);
        }
    }
    free(bh); free(kh); free(bv); free(kv); free(tmp);
    return 0;
}

/* Convert.c f2l: float -> 8-bit, clamp then truncate */
void oracle_f32_to_l8(const float* src, int64_t n, uint8_t* dst) {
    for (int64_t i = 0; i < n; i++) {
        float v = src[i];
        dst[i] = v <= 0.0f ? 0 : v >= 255.0f ? 255 : (uint8_t)v;
    }
}
