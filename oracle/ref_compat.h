/* TEST INFRASTRUCTURE ONLY (oracle/): force-included (-include) when oracle/build_ref.py
 * compiles the reference's own c++ext CPU sources *in place* from /root/reference.
 *
 * The reference was written for torch 1.0 and calls
 *     AT_DISPATCH_FLOATING_TYPES(dets.type(), ...)          (csrc/cpu/nms_cpu.cpp:75)
 * torch >= 2.x's dispatch macro resolves the type through ::detail::scalar_type(), which no
 * longer has an overload for at::DeprecatedTypeProperties (what Tensor::type() returns).
 * This header restores that one overload so the reference translation units compile
 * unmodified. It contains no reference code and no arithmetic.
 */
#pragma once
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>
namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) {
  return t.scalarType();
}
}  // namespace detail
