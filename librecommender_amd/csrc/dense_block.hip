// First Dense layer of `dense_nn` (layers/dense.py:12-49) over a MATERIALISED block of rows — the general feature
// nets (DIN: [user | item | attention output], algorithms/din.py:165-192): the block is addressed like a table through
// an id map, so the fused lookup + first-layer MFMA kernels of csrc/deepfm_l1.hip (forward, weight gradient, row
// gradient) and the BatchNorm-fold kernels of csrc/deepfm_fold.hip serve it unchanged.  What is left is HBM-bound
// glue over [B, n_in] floats:
//   lr_table_colstats_f32  per-(field, chunk) partial column sums / sums of squares of the gathered block
//                          table[idx[b, f], :] — the batch statistics of the input BatchNorm
//                          (tf.layers.batch_normalization(training=True), layers/dense.py:30-31)
//   lr_bn_remainder_f32    G[r, :] -= a[f(r), :] + c[f(r), :] * x[r, :]  — the BatchNorm-backward terms that do not go
//                          through the GEMM (dx = G - a - c * x, see layers/dense.py:_FoldedBNDense); f(r) = r / rows_per_plane
//                          for a block stored plane by plane, r % period for a row-major [B, period * Kp] block
// Fixed summation orders: results are run-to-run identical.
#include "common.hpp"

namespace lr {

// grid (F, C): workgroup (f, c) sums samples b = c, c + C, ... of field f.  Row group = K/4 lanes x 16-byte pieces;
// kBlock / LPR row groups walk the chunk's samples side by side and are folded through LDS in group order.
template <int LPR>
__global__ __launch_bounds__(kBlock) void table_colstats_kernel(const float* __restrict__ table, int64_t V,
                                                               const int32_t* __restrict__ idx, int64_t B, int F,
                                                               int C, float* __restrict__ partial) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  __shared__ float4 red[2][NG][LPR];
  const int f = blockIdx.x, c = blockIdx.y;
  const int lane = threadIdx.x % LPR, g = threadIdx.x / LPR;
  // contiguous sample range of chunk c (fixed, independent of the launch)
  const int64_t per = (B + C - 1) / C;
  const int64_t b0 = c * per, b1 = (b0 + per) < B ? (b0 + per) : B;
  float4 s = f4_zero(), q = f4_zero();
  for (int64_t b = b0 + g; b < b1; b += NG) {
    const int32_t id = idx[b * F + f];
    if (id >= 0 && id < V) {
      const float4 x = ld4(table + static_cast<int64_t>(id) * K + lane * 4);
      s = f4_add(s, x);
      q = f4_fma(x, x, q);
    }
  }
  red[0][g][lane] = s;
  red[1][g][lane] = q;
  __syncthreads();
  if (g == 0) {
    float4 ts = red[0][0][lane], tq = red[1][0][lane];
    for (int k = 1; k < NG; ++k) {
      ts = f4_add(ts, red[0][k][lane]);
      tq = f4_add(tq, red[1][k][lane]);
    }
    float* out = partial + ((static_cast<int64_t>(f) * C + c) * 2) * K;
    st4(out + lane * 4, ts);
    st4(out + K + lane * 4, tq);
  }
}

__global__ __launch_bounds__(kBlock) void bn_remainder_kernel(float* __restrict__ G, const float* __restrict__ x,
                                                             const float* __restrict__ a, const float* __restrict__ c,
                                                             int64_t rows, int64_t rows_per_plane, int64_t period, int Kp) {
  const int q4 = Kp / 4;
  const int64_t total = rows * q4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / q4;
    const int k = static_cast<int>(e - r * q4) * 4;
    const int64_t p = period > 0 ? r % period : r / rows_per_plane;
    const float4 av = ld4(a + p * Kp + k), cv = ld4(c + p * Kp + k);
    const float4 xv = ld4(x + r * Kp + k);
    float4 g = ld4(G + r * Kp + k);
    g.x = g.x - av.x - cv.x * xv.x;
    g.y = g.y - av.y - cv.y * xv.y;
    g.z = g.z - av.z - cv.z * xv.z;
    g.w = g.w - av.w - cv.w * xv.w;
    st4(G + r * Kp + k, g);
  }
}

}  // namespace lr

using namespace lr;

extern "C" int lr_table_colstats_f32(const float* table, int64_t V, int K, const int32_t* idx, int64_t B, int F,
                                     int C, float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(table && idx && partial && V >= 1 && B >= 1 && F >= 1 && C >= 1 && C <= 65535);
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(table) % 16 == 0 && reinterpret_cast<uintptr_t>(partial) % 16 == 0);
  hipStream_t s = as_stream(stream);
#define LR_TCS(LPR)                                                                                          \
  {                                                                                                          \
    hipLaunchKernelGGL((table_colstats_kernel<LPR>), dim3(F, C), dim3(kBlock), 0, s, table, V, idx, B, F, C, \
                       partial);                                                                             \
    return launch_status();                                                                                  \
  }
  if (K == 16) LR_TCS(4)
  if (K == 32) LR_TCS(8)
  if (K == 64) LR_TCS(16)
  if (K == 128) LR_TCS(32)
#undef LR_TCS
  return LR_ESHAPE;
}

extern "C" int lr_bn_remainder_f32(float* G, const float* x, const float* a, const float* c, int64_t rows,
                                   int64_t rows_per_plane, int64_t period, int Kp, lr_stream_t stream) {
  LR_CHECK_ARG(G && x && a && c && rows >= 0 && rows_per_plane >= 1 && period >= 0 && Kp >= 4 && Kp % 4 == 0);
  if (rows == 0) return LR_OK;
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(G) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(a) % 16 == 0 && reinterpret_cast<uintptr_t>(c) % 16 == 0);
  hipLaunchKernelGGL(bn_remainder_kernel, dim3(grid_for(rows * (Kp / 4), kBlock)), dim3(kBlock), 0, as_stream(stream),
                     G, x, a, c, rows, rows_per_plane, period, Kp);
  return launch_status();
}
