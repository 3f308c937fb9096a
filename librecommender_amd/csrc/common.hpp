// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/libreco_hip.h"

namespace lr {

constexpr int kWave = 64;
constexpr int kBlock = 256;        // 4 waves = one per SIMD of a CU
constexpr int kNumCU = 256;        // MI355X
constexpr int kPersistentBlocks = kNumCU * 8;  // memory-bound grid cap (guide §6 G11)

inline hipStream_t as_stream(lr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Zero a small device buffer with a KERNEL on the launch stream.  hipMemsetAsync becomes a memset NODE when the call
// is captured into a hipGraph; on this stack (ROCm 7.0 runtime inside PyTorch 2.10) replays of a captured training step
// that mixed memset / memcpy nodes with kernel nodes raced with their neighbouring kernel nodes once the device was kept
// busy by another stream (device-side batch loader): device memory faults at varying addresses, gone with
// AMD_SERIALIZE_KERNEL=3 and gone with kernel-only graphs (profiles/r03_graph_fault.md).  Kernel nodes of one captured
// stream are ordered among themselves; everything a captured step does is therefore a kernel.
static __global__ void lr_zero_words_kernel(uint32_t* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
inline void zero_words_async(void* p, int n_words, hipStream_t s) {
  hipLaunchKernelGGL(lr_zero_words_kernel, dim3(1), dim3(64), 0, s, static_cast<uint32_t*>(p), n_words);
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LR_OK : static_cast<int>(e);
}

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int grid_for(int64_t work_items, int items_per_block, int cap = kPersistentBlocks) {
  int64_t g = ceil_div(work_items, items_per_block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
// streaming (non-temporal) 16-byte store: outputs that are written once and not re-read by
// this kernel should not evict table rows from L2/MALL.
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  __builtin_nontemporal_store(v.x, p + 0);
  __builtin_nontemporal_store(v.y, p + 1);
  __builtin_nontemporal_store(v.z, p + 2);
  __builtin_nontemporal_store(v.w, p + 3);
}
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z),
                     fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int mask) {
  return make_float4(__shfl_xor(v.x, mask), __shfl_xor(v.y, mask), __shfl_xor(v.z, mask),
                     __shfl_xor(v.w, mask));
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// One Adam update of a single element; returns the new weight.  Mirrors lr_adam_hp docs.
//   tf_style (tf.train.AdamOptimizer._apply_sparse_shared): m = m*b1 + g*(1-b1) with (1-b1)
//     formed in fp32; w -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t).
//   torch (torch.optim.Adam, single-tensor path): g += wd*w; m = m + (g-m)*(1-b1) [lerp_];
//     v = v*b2 + (1-b2)*g*g with (1-b) formed in double; w -= lr/(1-b1^t) * m /
//     (sqrt(v)/sqrt(1-b2^t) + eps).
struct AdamCoef {
  float b1, b2, omb1, omb2, eps, wd;
  float step_size;  // tf: lr*sqrt(1-b2^t)/(1-b1^t); torch: lr/(1-b1^t)
  float bc2_sqrt;   // torch: sqrt(1-b2^t); tf: unused
  int tf_style;
};
inline AdamCoef make_adam_coef(const lr_adam_hp& hp) {
  AdamCoef c;
  c.b1 = static_cast<float>(hp.beta1);
  c.b2 = static_cast<float>(hp.beta2);
  c.eps = static_cast<float>(hp.eps);
  c.wd = static_cast<float>(hp.weight_decay);
  c.tf_style = hp.tf_style;
  const double t = static_cast<double>(hp.step < 1 ? 1 : hp.step);
  const double bc1 = 1.0 - pow(hp.beta1, t);
  const double bc2 = 1.0 - pow(hp.beta2, t);
  if (hp.tf_style) {
    c.omb1 = 1.f - c.b1;
    c.omb2 = 1.f - c.b2;
    c.step_size = static_cast<float>(hp.lr * sqrt(bc2) / bc1);
    c.bc2_sqrt = 1.f;
  } else {
    c.omb1 = static_cast<float>(1.0 - hp.beta1);
    c.omb2 = static_cast<float>(1.0 - hp.beta2);
    c.step_size = static_cast<float>(hp.lr / bc1);
    c.bc2_sqrt = static_cast<float>(sqrt(bc2));
  }
  return c;
}
__device__ __forceinline__ float adam_elem(float w, float g, float& m, float& v,
                                           const AdamCoef& c) {
  float denom;
  if (c.tf_style) {
    m = fmaf(m, c.b1, g * c.omb1);
    v = fmaf(v, c.b2, (g * g) * c.omb2);
    denom = sqrtf(v) + c.eps;
  } else {
    g = fmaf(c.wd, w, g);
    m = fmaf(g - m, c.omb1, m);
    v = fmaf(v, c.b2, (c.omb2 * g) * g);
    denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  }
  return w - c.step_size * (m / denom);
}
__device__ __forceinline__ float4 adam_vec(float4 w, float4 g, float4& m, float4& v,
                                           const AdamCoef& c) {
  float4 r;
  r.x = adam_elem(w.x, g.x, m.x, v.x, c);
  r.y = adam_elem(w.y, g.y, m.y, v.y, c);
  r.z = adam_elem(w.z, g.z, m.z, v.z, c);
  r.w = adam_elem(w.w, g.w, m.w, v.w, c);
  return r;
}

// out[c] = sum_k partial[k*stride + c] in a fixed order.  Workgroup = 16 columns x 16 k-slices: thread (cx, ky)
// sums k = ky, ky+16, ... (double), the 16 slices are combined in slice order through LDS.
// (`out2`, nullable: a second copy of the result — the fused tail keeps one in LDS and lets one workgroup write the global one)
__device__ __forceinline__ void reduce_partials_body(int cb0, int cb_stride, const float* __restrict__ partial, int nblk,
                                                     int64_t n, int64_t stride, float* __restrict__ out,
                                                     float* __restrict__ out2) {
  __shared__ double red[16][17];
  const int cx = threadIdx.x & 15, ky = threadIdx.x >> 4;
  for (int64_t c0 = static_cast<int64_t>(cb0) * 16; c0 < n; c0 += static_cast<int64_t>(cb_stride) * 16) {
    const int64_t c = c0 + cx;
    double t = 0.0;
    if (c < n)
      for (int k = ky; k < nblk; k += 16) t += static_cast<double>(partial[static_cast<int64_t>(k) * stride + c]);
    red[ky][cx] = t;
    __syncthreads();
    if (ky == 0 && c < n) {
      double tot = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) tot += red[g][cx];
      if (out != nullptr) out[c] = static_cast<float>(tot);
      if (out2 != nullptr) out2[c] = static_cast<float>(tot);
    }
    __syncthreads();
  }
}

}  // namespace lr

#define LR_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return LR_EINVAL; \
  } while (0)
