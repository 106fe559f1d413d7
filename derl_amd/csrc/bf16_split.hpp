// Exact three-term bf16 split of fp32 values (x == hi + mid + lo): shared by the plane-splitting
// launch of dx_cnn_pack (planes.hip) and the opt-in split-bf16 GEMM experiment.
#pragma once
#include "igemm_dev.hpp"

namespace dx {
namespace {

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint32_t pack2(float a, float b) {  // round-to-nearest-even
  bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f32(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

struct Split4 {
  uint2 hi, mid, lo;  // 4 bf16 each
};

__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = pack2(x0, x1);
  const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);  // exact
  m = pack2(r0, r1);
  l = pack2(r0 - lo_f32(m), r1 - hi_f32(m));             // exact residual, exact conversion
}

__device__ __forceinline__ Split4 split4(f32x4 v) {
  Split4 s;
  split2(v.x, v.y, s.hi.x, s.mid.x, s.lo.x);
  split2(v.z, v.w, s.hi.y, s.mid.y, s.lo.y);
  return s;
}

}  // namespace
}  // namespace dx
