// Split-bf16 arithmetic shared by kernels that take an f32 contraction to the bf16 matrix pipe without giving up f32 accuracy:
// every f32 operand is split EXACTLY into three bf16 values (x = x1 + x2 + x3, round-to-nearest-even at each step, barring
// underflow) and a product a*b is taken as the six largest of the nine cross terms, a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1
// (smallest first), each an exact bf16 x bf16 product accumulated in f32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
// <= 2^-24 relative.  (csrc/deepfm_l1_sb.hip and csrc/softmax_ce.hip carry their own copies with profiling switches.)
#pragma once
#include "common.hpp"

namespace lr {
namespace sb {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ uint32_t pack2(float a, float b) {          // one v_cvt_pk_bf16_f32: element 0 in the low half
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
__device__ __forceinline__ float lo_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// four f32 -> three planes of four bf16 (two packed words each)
__device__ __forceinline__ void split4(float4 x, uint2& p1, uint2& p2, uint2& p3) {
  p1.x = pack2(x.x, x.y); p1.y = pack2(x.z, x.w);
  const float r0 = x.x - lo_f32(p1.x), r1 = x.y - hi_f32(p1.x), r2 = x.z - lo_f32(p1.y), r3 = x.w - hi_f32(p1.y);
  p2.x = pack2(r0, r1); p2.y = pack2(r2, r3);
  const float s0 = r0 - lo_f32(p2.x), s1 = r1 - hi_f32(p2.x), s2 = r2 - lo_f32(p2.y), s3 = r3 - hi_f32(p2.y);
  p3.x = pack2(s0, s1); p3.y = pack2(s2, s3);
}
// eight f32 (two float4: elements 0-3, 4-7) -> three bf16x8 operand fragments
__device__ __forceinline__ void split8(float4 lo, float4 hi, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  uint2 l1, l2, l3, h1, h2, h3;
  split4(lo, l1, l2, l3);
  split4(hi, h1, h2, h3);
  const u32x4 v1 = {l1.x, l1.y, h1.x, h1.y}, v2 = {l2.x, l2.y, h2.x, h2.y}, v3 = {l3.x, l3.y, h3.x, h3.y};
  __builtin_memcpy(&a1, &v1, 16);
  __builtin_memcpy(&a2, &v2, 16);
  __builtin_memcpy(&a3, &v3, 16);
}
// acc += a * b with a = a1 + a2 + a3, b = b1 + b2 + b3: the six largest cross terms, smallest first
__device__ __forceinline__ void mfma6(f32x16& acc, bf16x8 a1, bf16x8 a2, bf16x8 a3, bf16x8 b1, bf16x8 b2, bf16x8 b3) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}

}  // namespace sb
}  // namespace lr
