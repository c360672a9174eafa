// The reference's pybind module, on the MI355X library: `maskrcnn._C_native` exports nms / crop_forward / crop_backward with the
// names, argument order and doc strings of c++ext/maskrcnn/csrc/vision.cpp:11-15, implemented on the C ABI of
// libmaskrcnn_hip.so (include/maskrcnn_hip.h) — what a maintainer of the reference would compile in place of csrc/vision.cpp +
// csrc/cuda/*.cu. The same three functions are also registered with the dispatcher as maskrcnn_native::{nms, crop_forward,
// crop_backward} (TORCH_LIBRARY), so a C++ or TorchScript caller reaches them without Python. (The shipped Python path,
// maskrcnn/_C.py -> torch.ops.maskrcnn.*, registers its ops from Python over ctypes and also takes CPU tensors by staging; this
// module is the GPU-only, Python-free form of the same calls — tests/test_gpu_native_ext.py holds the two equal bit for bit.)
//
// Semantics kept from the reference: nms -> int64 indices into dets, ascending (nms_cpu.cpp:69), float32 or float64 boxes
// (nms_cpu.cpp:73-79), `>=` rule; crop_forward resizes `crops` to [N, C, h, w] and overwrites it (crop_cpu.cpp:141-143);
// crop_backward zeroes and accumulates grads_image (crop_cpu.cpp:197). CPU tensors raise "Not compiled with CPU support" — the
// mirror of nms.h:24 / crop.h:28,47.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm's device type is "cuda": its guard and stream types
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>
#include <torch/library.h>

#include "maskrcnn_hip.h"

namespace {

mrcnn_stream_t current_stream() {
    return reinterpret_cast<mrcnn_stream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream());
}

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", mrcnn_last_error()); }

at::Tensor nms(const at::Tensor& dets, double threshold) {
    TORCH_CHECK(dets.is_cuda(), "Not compiled with CPU support");
    TORCH_CHECK(dets.dim() == 2 && dets.size(1) == 5, "nms: dets must be [N, 5] (y1, x1, y2, x2, score)");
    TORCH_CHECK(dets.scalar_type() == at::kFloat || dets.scalar_type() == at::kDouble, "nms: float32 or float64 boxes");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dets.device());
    const int64_t n = dets.size(0);
    if (n == 0) return at::empty({0}, dets.options().dtype(at::kLong));
    const int32_t dtype = dets.scalar_type() == at::kDouble ? 1 : 0;
    auto keep = at::empty({n}, dets.options().dtype(at::kLong));
    auto count = at::empty({1}, dets.options().dtype(at::kLong));
    auto ws = at::empty({static_cast<int64_t>(mrcnn_nms_general_workspace_bytes(n, dtype)) + 256}, dets.options().dtype(at::kByte));
    auto* wp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws.data_ptr()) + 255) & ~uintptr_t(255));
    check(mrcnn_nms_general(dets.data_ptr(), dtype, n, dets.stride(0), dets.stride(1), static_cast<float>(threshold),
                            keep.data_ptr<int64_t>(), count.data_ptr<int64_t>(), wp,
                            mrcnn_nms_general_workspace_bytes(n, dtype), current_stream()),
          "nms");
    return keep.narrow(0, 0, count.item<int64_t>());   // one host synchronisation, as the reference's D2H copy (nms_cuda.cu:108)
}

void check_crop_inputs(const at::Tensor& image, const at::Tensor& boxes, const at::Tensor& box_index) {
    TORCH_CHECK(image.is_cuda() && boxes.is_cuda() && box_index.is_cuda(), "Not compiled with CPU support");
    TORCH_CHECK(image.scalar_type() == at::kFloat && boxes.scalar_type() == at::kFloat, "crop: expected scalar type Float for image and boxes");
    TORCH_CHECK(box_index.scalar_type() == at::kInt, "crop: expected scalar type Int for box_index");
    TORCH_CHECK(image.dim() == 4 && boxes.dim() == 2 && boxes.size(1) == 4 && box_index.numel() == boxes.size(0),
                "crop: image [B,C,H,W], boxes [N,4], box_index [N] expected");
}

void crop_forward(const at::Tensor& image, const at::Tensor& boxes, const at::Tensor& box_index, double extrapolation_value,
                  int64_t crop_height, int64_t crop_width, at::Tensor crops) {
    check_crop_inputs(image, boxes, box_index);
    TORCH_CHECK(crops.is_cuda() && crops.scalar_type() == at::kFloat, "crop_forward: crops must be a float tensor on the GPU");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(image.device());
    const auto im = image.contiguous(), bx = boxes.contiguous(), bi = box_index.contiguous();
    const int64_t n = bx.size(0);
    crops.resize_({n, im.size(1), crop_height, crop_width});
    check(mrcnn_crop_forward_f32(im.data_ptr<float>(), static_cast<int32_t>(im.size(0)), static_cast<int32_t>(im.size(1)),
                                 static_cast<int32_t>(im.size(2)), static_cast<int32_t>(im.size(3)), bx.data_ptr<float>(),
                                 bi.data_ptr<int32_t>(), static_cast<int32_t>(n), static_cast<float>(extrapolation_value),
                                 static_cast<int32_t>(crop_height), static_cast<int32_t>(crop_width), crops.data_ptr<float>(),
                                 current_stream()),
          "crop_forward");
}

void crop_backward(const at::Tensor& grads, const at::Tensor& boxes, const at::Tensor& box_index, at::Tensor grads_image) {
    TORCH_CHECK(grads.is_cuda() && boxes.is_cuda() && box_index.is_cuda() && grads_image.is_cuda(), "Not compiled with CPU support");
    TORCH_CHECK(grads.scalar_type() == at::kFloat && boxes.scalar_type() == at::kFloat && grads_image.scalar_type() == at::kFloat,
                "crop_backward: expected scalar type Float");
    TORCH_CHECK(box_index.scalar_type() == at::kInt, "crop_backward: expected scalar type Int for box_index");
    TORCH_CHECK(grads_image.is_contiguous() && grads_image.dim() == 4 && grads.dim() == 4 && grads.size(1) == grads_image.size(1),
                "crop_backward: grads [N,C,h,w], grads_image a contiguous [B,C,H,W] tensor");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(grads.device());
    const auto g = grads.contiguous(), bx = boxes.contiguous(), bi = box_index.contiguous();
    check(mrcnn_crop_backward_f32(g.data_ptr<float>(), bx.data_ptr<float>(), bi.data_ptr<int32_t>(), static_cast<int32_t>(g.size(0)),
                                  static_cast<int32_t>(grads_image.size(0)), static_cast<int32_t>(grads_image.size(1)),
                                  static_cast<int32_t>(grads_image.size(2)), static_cast<int32_t>(grads_image.size(3)),
                                  static_cast<int32_t>(g.size(2)), static_cast<int32_t>(g.size(3)), grads_image.data_ptr<float>(),
                                  current_stream()),
          "crop_backward");
}

}  // namespace

// Python-free callers: torch.ops.maskrcnn_native.* from C++ (c10::Dispatcher) or TorchScript
TORCH_LIBRARY(maskrcnn_native, m) {
    m.def("nms(Tensor dets, float threshold) -> Tensor");
    m.def("crop_forward(Tensor image, Tensor boxes, Tensor box_index, float extrapolation_value, int crop_height, "
          "int crop_width, Tensor(a!) crops) -> ()");
    m.def("crop_backward(Tensor grads, Tensor boxes, Tensor box_index, Tensor(a!) grads_image) -> ()");
}
TORCH_LIBRARY_IMPL(maskrcnn_native, CUDA, m) {
    m.impl("nms", &nms);
    m.impl("crop_forward", &crop_forward);
    m.impl("crop_backward", &crop_backward);
}

// the reference's module (vision.cpp:11-15)
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("nms", &nms, "non-maximum suppression");
    m.def("crop_forward", &crop_forward, "crop forward");
    m.def("crop_backward", &crop_backward, "crop backward");
}
