// CSR SpMM for LightGCN propagation: Y = A * X with A in CSR (int64 rowptr, int32 col, fp32
// val), X/Y row-major [rows, K].  HBM-bound: per nonzero 8 B of (col,val) + one K*4-byte
// gathered row.  One row group of LPR = K/4 lanes per output row, 4 nonzeros in flight.
// Optional fused accumulation acc += Y implements the running layer sum of
// lightgcn_module.py:83-84 without re-reading Y.
#include "common.hpp"

namespace lr {

template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_vec_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int64_t rows, const float* __restrict__ X,
    float* __restrict__ Y, float* __restrict__ acc) {
  constexpr int K = LPR * 4;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int c4 = static_cast<int>(gtid % LPR) * 4;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t r = gtid / LPR; r < rows; r += ngroups) {
    const int64_t j0 = rowptr[r], j1 = rowptr[r + 1];
    float4 y = f4_zero();
    int64_t j = j0;
    for (; j + 4 <= j1; j += 4) {
      const int32_t c0 = col[j], c1 = col[j + 1], c2 = col[j + 2], c3 = col[j + 3];
      const float a0 = val[j], a1 = val[j + 1], a2 = val[j + 2], a3 = val[j + 3];
      const float4 x0 = ld4(X + static_cast<int64_t>(c0) * K + c4);
      const float4 x1 = ld4(X + static_cast<int64_t>(c1) * K + c4);
      const float4 x2 = ld4(X + static_cast<int64_t>(c2) * K + c4);
      const float4 x3 = ld4(X + static_cast<int64_t>(c3) * K + c4);
      y = f4_fma(make_float4(a0, a0, a0, a0), x0, y);
      y = f4_fma(make_float4(a1, a1, a1, a1), x1, y);
      y = f4_fma(make_float4(a2, a2, a2, a2), x2, y);
      y = f4_fma(make_float4(a3, a3, a3, a3), x3, y);
    }
    for (; j < j1; ++j) {
      const float a = val[j];
      y = f4_fma(make_float4(a, a, a, a), ld4(X + static_cast<int64_t>(col[j]) * K + c4), y);
    }
    st4(Y + r * K + c4, y);
    if (acc != nullptr) st4(acc + r * K + c4, f4_add(ld4(acc + r * K + c4), y));
  }
}

__global__ __launch_bounds__(kBlock) void spmm_scalar_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int64_t rows, const float* __restrict__ X, int K,
    float* __restrict__ Y, float* __restrict__ acc) {
  const int64_t total = rows * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += stride) {
    const int64_t r = t / K;
    const int c = static_cast<int>(t - r * K);
    float y = 0.f;
    for (int64_t j = rowptr[r]; j < rowptr[r + 1]; ++j)
      y = fmaf(val[j], X[static_cast<int64_t>(col[j]) * K + c], y);
    Y[t] = y;
    if (acc) acc[t] += y;
  }
}

}  // namespace lr

using namespace lr;

extern "C" int lr_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                               int64_t rows, const float* X, int K, float* Y, float* acc,
                               lr_stream_t stream) {
  LR_CHECK_ARG(rows >= 0 && K >= 1);
  if (rows == 0) return LR_OK;
  LR_CHECK_ARG(rowptr && X && Y);
  hipStream_t s = as_stream(stream);
  const bool aligned = reinterpret_cast<uintptr_t>(X) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(Y) % 16 == 0 &&
                       (!acc || reinterpret_cast<uintptr_t>(acc) % 16 == 0);
#define LR_SPMM(LPR)                                                                        \
  {                                                                                         \
    const int grid = grid_for(rows, kBlock / LPR);                                          \
    hipLaunchKernelGGL((spmm_vec_kernel<LPR>), dim3(grid), dim3(kBlock), 0, s, rowptr, col, \
                       val, rows, X, Y, acc);                                               \
    return launch_status();                                                                 \
  }
  if (aligned) {
    if (K == 16) LR_SPMM(4)
    if (K == 32) LR_SPMM(8)
    if (K == 64) LR_SPMM(16)
    if (K == 128) LR_SPMM(32)
  }
#undef LR_SPMM
  hipLaunchKernelGGL(spmm_scalar_kernel, dim3(grid_for(rows * K, kBlock)), dim3(kBlock), 0, s,
                     rowptr, col, val, rows, X, K, Y, acc);
  return launch_status();
}
