/* ORACLE — TEST INFRASTRUCTURE ONLY. Never imported, linked or executed by the product path
 * (maskrcnn_amd/, maskrcnn/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * Plain-C restatement of the reference CPU crop-and-resize ("RoIAlign"):
 *   forward : /root/reference/c++ext/maskrcnn/csrc/cpu/crop_cpu.cpp:13-116 (crop_per_box),
 *             :119-164 (crop_cpu_forward: output [N, C, ph, pw], zero-initialised)
 *   backward: /root/reference/c++ext/maskrcnn/csrc/cpu/crop_cpu.cpp:167-265 (crop_cpu_backward)
 *
 * Semantics restated (Appendix A of SURVEY.md): tf.image.crop_and_resize with ONE bilinear sample per
 * output bin on an endpoint-inclusive grid over normalised (y1,x1,y2,x2) boxes scaled by (H-1),(W-1);
 * samples with in_y<0 || in_y>H-1 (strict) take extrapolation_value; taps are floorf/ceilf; lerp is
 * a + (b-a)*t; crop_height==1 samples the box centre, evaluated in double then narrowed to float.
 *
 * Pinned by tests/golden/crop_*.npz (generated from the reference's compiled sources, oracle/_ref).
 * Build without FMA contraction (Makefile).
 *
 * Deviation: the reference printf()s and exit(-1)s on a box_index outside [0,batch) (:47-50); this
 * restatement returns -1 instead of killing the process (tests assert the error path separately).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

int oracle_crop_forward_f32(const float* image_data, int batch_size, int depth, int image_height,
                            int image_width, const float* boxes_data, const int32_t* box_index_data,
                            int num_boxes, float extrapolation_value, int crop_height,
                            int crop_width, float* crops_data) {
    const int image_channel_elements = image_height * image_width;
    const int64_t image_elements = (int64_t)depth * image_channel_elements;
    const int crop_channel_elements = crop_height * crop_width;
    const int64_t crop_elements = (int64_t)depth * crop_channel_elements;

    memset(crops_data, 0, sizeof(float) * (size_t)num_boxes * crop_elements); /* :141-143 */

    for (int b = 0; b < num_boxes; ++b) {
        const float* box = boxes_data + b * 4;
        const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
        const int b_in = box_index_data[b];
        if (b_in < 0 || b_in >= batch_size) return -1; /* reference: exit(-1), :47-50 */

        /* :52-55  (y2 - y1) * (H-1) / (ph-1): float*int -> float, float/int -> float */
        float height_scale = 0, width_scale = 0;
        if (crop_height > 1) {
            float t = y2 - y1; t = t * (float)(image_height - 1);
            height_scale = t / (float)(crop_height - 1);
        }
        if (crop_width > 1) {
            float t = x2 - x1; t = t * (float)(image_width - 1);
            width_scale = t / (float)(crop_width - 1);
        }

        for (int y = 0; y < crop_height; ++y) {
            float in_y;
            if (crop_height > 1) { /* :59-60 */
                float a = y1 * (float)(image_height - 1);
                float s = (float)y * height_scale;
                in_y = a + s;
            } else { /* :61  0.5 * (y1 + y2) * (H-1): the literal 0.5 is double */
                float sum = y1 + y2;
                in_y = (float)(0.5 * (double)sum * (double)(image_height - 1));
            }
            float* out_row = crops_data + crop_elements * b + (int64_t)y * crop_width;
            if (in_y < 0 || in_y > (float)(image_height - 1)) { /* :63-74 */
                for (int x = 0; x < crop_width; ++x)
                    for (int d = 0; d < depth; ++d)
                        out_row[(int64_t)crop_channel_elements * d + x] = extrapolation_value;
                continue;
            }
            const int top_y_index = (int)floorf(in_y);
            const int bottom_y_index = (int)ceilf(in_y);
            const float y_lerp = in_y - (float)top_y_index;

            for (int x = 0; x < crop_width; ++x) {
                float in_x;
                if (crop_width > 1) { /* :82-83 */
                    float a = x1 * (float)(image_width - 1);
                    float s = (float)x * width_scale;
                    in_x = a + s;
                } else { /* :84 */
                    float sum = x1 + x2;
                    in_x = (float)(0.5 * (double)sum * (double)(image_width - 1));
                }
                if (in_x < 0 || in_x > (float)(image_width - 1)) { /* :85-92 */
                    for (int d = 0; d < depth; ++d)
                        out_row[(int64_t)crop_channel_elements * d + x] = extrapolation_value;
                    continue;
                }
                const int left_x_index = (int)floorf(in_x);
                const int right_x_index = (int)ceilf(in_x);
                const float x_lerp = in_x - (float)left_x_index;

                for (int d = 0; d < depth; ++d) { /* :98-111 */
                    const float* pimage =
                        image_data + b_in * image_elements + (int64_t)d * image_channel_elements;
                    const float top_left = pimage[top_y_index * image_width + left_x_index];
                    const float top_right = pimage[top_y_index * image_width + right_x_index];
                    const float bottom_left = pimage[bottom_y_index * image_width + left_x_index];
                    const float bottom_right = pimage[bottom_y_index * image_width + right_x_index];
                    float t = top_right - top_left; t = t * x_lerp;
                    const float top = top_left + t;
                    float u = bottom_right - bottom_left; u = u * x_lerp;
                    const float bottom = bottom_left + u;
                    float v = bottom - top; v = v * y_lerp;
                    out_row[(int64_t)crop_channel_elements * d + x] = top + v;
                }
            }
        }
    }
    return 0;
}

int oracle_crop_backward_f32(const float* grads_data, const float* boxes_data,
                             const int32_t* box_index_data, int num_boxes, int batch_size,
                             int depth, int image_height, int image_width, int crop_height,
                             int crop_width, float* grads_image_data) {
    const int image_channel_elements = image_height * image_width;
    const int64_t image_elements = (int64_t)depth * image_channel_elements;
    const int crop_channel_elements = crop_height * crop_width;
    const int64_t crop_elements = (int64_t)depth * crop_channel_elements;

    memset(grads_image_data, 0, sizeof(float) * (size_t)batch_size * image_elements); /* :197 */

    for (int b = 0; b < num_boxes; ++b) {
        const float* box = boxes_data + b * 4;
        const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
        const int b_in = box_index_data[b];
        if (b_in < 0 || b_in >= batch_size) return -1; /* :213-216 */

        float height_scale = 0, width_scale = 0;
        if (crop_height > 1) {
            float t = y2 - y1; t = t * (float)(image_height - 1);
            height_scale = t / (float)(crop_height - 1);
        }
        if (crop_width > 1) {
            float t = x2 - x1; t = t * (float)(image_width - 1);
            width_scale = t / (float)(crop_width - 1);
        }
        for (int y = 0; y < crop_height; ++y) {
            float in_y;
            if (crop_height > 1) {
                float a = y1 * (float)(image_height - 1);
                float s = (float)y * height_scale;
                in_y = a + s;
            } else {
                float sum = y1 + y2;
                in_y = (float)(0.5 * (double)sum * (double)(image_height - 1));
            }
            if (in_y < 0 || in_y > (float)(image_height - 1)) continue;
            const int top_y_index = (int)floorf(in_y);
            const int bottom_y_index = (int)ceilf(in_y);
            const float y_lerp = in_y - (float)top_y_index;
            for (int x = 0; x < crop_width; ++x) {
                float in_x;
                if (crop_width > 1) {
                    float a = x1 * (float)(image_width - 1);
                    float s = (float)x * width_scale;
                    in_x = a + s;
                } else {
                    float sum = x1 + x2;
                    in_x = (float)(0.5 * (double)sum * (double)(image_width - 1));
                }
                if (in_x < 0 || in_x > (float)(image_width - 1)) continue;
                const int left_x_index = (int)floorf(in_x);
                const int right_x_index = (int)ceilf(in_x);
                const float x_lerp = in_x - (float)left_x_index;
                for (int d = 0; d < depth; ++d) { /* :244-260, serial += in box order */
                    float* pimage = grads_image_data + b_in * image_elements +
                                    (int64_t)d * image_channel_elements;
                    const float grad_val = grads_data[crop_elements * b +
                                                      (int64_t)crop_channel_elements * d +
                                                      y * crop_width + x];
                    const float one_m_y = 1.0f - y_lerp;
                    const float one_m_x = 1.0f - x_lerp;
                    const float dtop = one_m_y * grad_val;
                    pimage[top_y_index * image_width + left_x_index] += one_m_x * dtop;
                    pimage[top_y_index * image_width + right_x_index] += x_lerp * dtop;
                    const float dbottom = y_lerp * grad_val;
                    pimage[bottom_y_index * image_width + left_x_index] += one_m_x * dbottom;
                    pimage[bottom_y_index * image_width + right_x_index] += x_lerp * dbottom;
                }
            }
        }
    }
    return 0;
}
