// BatchNorm-fold algebra of the fused DeepFM first layer (layers/dense.py:30-31 + the first Dense of
// `dense_nn`, algorithms/deepfm.py:165-170) as four small kernels instead of ~30 elementwise launches:
//   fold_stats : batch statistics of the gathered block from the per-field partial sums
//                (lr_fm_field_stats_f32) -> mean, inv = rsqrt(var + eps), s = gamma * inv,
//                t = beta - mean * s, moving averages (momentum)                      [F*K] each
//   pack_scaled: Wp = diag(s) W written straight into the two MFMA fragment orders (deepfm_l1.hip)
//   fold_bias  : bp = b + t^T W  (per-slab partial column sums; summed by lr_reduce_partials_f32)
//   fold_bwd   : from the weight-gradient slabs of lr_deepfm_l1_wgrad_f32 (sum in slab order) and
//                sgz = column sums of gz:   dW, dgamma, dbeta, db and the row-side remainder terms
//                bn_a, bn_c  (d x = G - a - c x, consumed by lr_fm_rows_adam_f32)
// Same formulas as layers/dense.py `fused_l1_forward/backward`; every reduction has a fixed order.
#include "common.hpp"

namespace lr {

struct FoldStats {
  const float* partial; int F, C, K; int64_t B; float eps, momentum;
  const float* gamma; const float* beta; float* moving_mean; float* moving_var;
  float* mean_out; float* inv_out; float* s_out; float* t_out;
};
// statistics, scale / shift and moving averages of ONE input column r; returns t[r]
__device__ __forceinline__ float l1_fold_stats_row(const FoldStats& A, int r) {
  const float* __restrict__ partial = A.partial;
  const int C = A.C, K = A.K;
  const int64_t B = A.B;
  const float eps = A.eps, momentum = A.momentum;
  const float* __restrict__ gamma = A.gamma;
  const float* __restrict__ beta = A.beta;
  float* __restrict__ moving_mean = A.moving_mean;
  float* __restrict__ moving_var = A.moving_var;
  float* __restrict__ mean_out = A.mean_out;
  float* __restrict__ inv_out = A.inv_out;
  float* __restrict__ s_out = A.s_out;
  float* __restrict__ t_out = A.t_out;
  {
    const int f = r / K, k = r - f * K;
    double s1 = 0.0, s2 = 0.0;
    // fixed order; fp64 combine (E[x^2] - mean^2 cancels in fp32).  Eight chunks' loads are issued before their adds:
    // the adds stay in chunk order (same bits), the chain no longer waits one memory latency per chunk
    const float* p0 = partial + static_cast<int64_t>(f) * C * 2 * K + k;
    int c = 0;
    for (; c + 8 <= C; c += 8) {
      float a[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = p0[static_cast<int64_t>(c + u) * 2 * K];
        q[u] = p0[static_cast<int64_t>(c + u) * 2 * K + K];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s1 += static_cast<double>(a[u]);
        s2 += static_cast<double>(q[u]);
      }
    }
    for (; c < C; ++c) {
      s1 += static_cast<double>(p0[static_cast<int64_t>(c) * 2 * K]);
      s2 += static_cast<double>(p0[static_cast<int64_t>(c) * 2 * K + K]);
    }
    const double m = s1 / static_cast<double>(B);
    double v = s2 / static_cast<double>(B) - m * m;
    v = v < 0.0 ? 0.0 : v;
    const float mean = static_cast<float>(m), var = static_cast<float>(v);
    moving_mean[r] = moving_mean[r] * momentum + mean * (1.f - momentum);
    moving_var[r] = moving_var[r] * momentum + var * (1.f - momentum);
    const float inv = rsqrtf(var + eps);
    const float s = gamma[r] * inv;
    mean_out[r] = mean;
    inv_out[r] = inv;
    s_out[r] = s;
    const float tr = beta[r] - mean * s;
    t_out[r] = tr;
    return tr;
  }
}
__global__ __launch_bounds__(kBlock) void l1_fold_stats_kernel(FoldStats A) {
  const int n = A.F * A.K;
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) l1_fold_stats_row(A, r);
}

// partial[blk][h] = sum over the slab's rows of t[r] * W[r][h]; slab `nblk` (the extra one) = b
// FUSED (round 6): the slab's workgroup first finalises the statistics of ITS rows (l1_fold_stats_row: the arithmetic of
// l1_fold_stats_kernel, one thread per row) and takes t from LDS — one launch instead of two, the same bits
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void l1_fold_bias_kernel(const float* __restrict__ t, const float* __restrict__ W,
                                                              const float* __restrict__ b, int n_rows, int H1,
                                                              int rows_per_blk, float* __restrict__ partial, FoldStats A) {
  const int nblk = gridDim.x - 1;
  if (static_cast<int>(blockIdx.x) == nblk) {
    for (int h = threadIdx.x; h < H1; h += kBlock) partial[static_cast<int64_t>(nblk) * H1 + h] = b[h];
    return;
  }
  __shared__ float red[kBlock];
  __shared__ float s_t[kBlock];
  // thread = (row lane rl, column h): kBlock / H1 rows in flight
  const int h = threadIdx.x % H1, rl = threadIdx.x / H1, RL = kBlock / H1;
  const int r0 = blockIdx.x * rows_per_blk;
  const int r1 = r0 + rows_per_blk < n_rows ? r0 + rows_per_blk : n_rows;
  if (FUSED) {
    const int r = r0 + static_cast<int>(threadIdx.x);
    if (static_cast<int>(threadIdx.x) < rows_per_blk && r < r1) s_t[threadIdx.x] = l1_fold_stats_row(A, r);
    __syncthreads();
  }
  float acc = 0.f;
  for (int r = r0 + rl; r < r1; r += RL) acc = fmaf(FUSED ? s_t[r - r0] : t[r], W[static_cast<int64_t>(r) * H1 + h], acc);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0) {
    float s = 0.f;
    for (int q = 0; q < RL; ++q) s += red[q * H1 + h];   // fixed order
    partial[static_cast<int64_t>(blockIdx.x) * H1 + h] = s;
  }
}

template <int LPR>   // lanes per row = H1 / 4
__global__ __launch_bounds__(kBlock) void l1_fold_bwd_kernel(
    const float* __restrict__ part, int n_slabs, int n_rows, int64_t B, const float* __restrict__ sgz,
    const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ inv, float* __restrict__ dW,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db, float* __restrict__ bn_a,
    float* __restrict__ bn_c) {
  constexpr int H1 = LPR * 4, RPB = kBlock / LPR;
  const int gl = threadIdx.x % LPR, c4 = gl * 4;
  const int r = blockIdx.x * RPB + threadIdx.x / LPR;
  if (blockIdx.x == 0 && threadIdx.x < H1) db[threadIdx.x] = sgz[threadIdx.x];
  if (r >= n_rows) return;      // whole row groups leave together: the shuffles below stay inside a group
  const int64_t off = static_cast<int64_t>(r) * H1 + c4;
  float4 g = f4_zero();
  for (int sl = 0; sl < n_slabs; ++sl)   // slab order
    g = f4_add(g, ld4(part + static_cast<int64_t>(sl) * n_rows * H1 + off));
  if (gamma == nullptr) {
    st4(dW + off, g);
    return;
  }
  const float4 sg = ld4(sgz + c4), w = ld4(W + off);
  const float mu = mean[r], iv = inv[r], ga = gamma[r], be = beta[r];
  float4 xh;
  xh.x = (g.x - mu * sg.x) * iv; xh.y = (g.y - mu * sg.y) * iv;
  xh.z = (g.z - mu * sg.z) * iv; xh.w = (g.w - mu * sg.w) * iv;
  float4 o;
  o.x = fmaf(ga, xh.x, be * sg.x); o.y = fmaf(ga, xh.y, be * sg.y);
  o.z = fmaf(ga, xh.z, be * sg.z); o.w = fmaf(ga, xh.w, be * sg.w);
  st4(dW + off, o);
  float dg = xh.x * w.x + xh.y * w.y + xh.z * w.z + xh.w * w.w;
  float dbt = w.x * sg.x + w.y * sg.y + w.z * sg.z + w.w * sg.w;
#pragma unroll
  for (int o2 = 1; o2 < LPR; o2 <<= 1) {   // butterfly over the row group: the same order every run
    dg += __shfl_xor(dg, o2);
    dbt += __shfl_xor(dbt, o2);
  }
  if (gl == 0) {
    const float s = ga * iv;
    const float c = s * iv * (dg / static_cast<float>(B));
    dgamma[r] = dg;
    dbeta[r] = dbt;
    bn_c[r] = c;
    bn_a[r] = s * (dbt / static_cast<float>(B)) - c * mu;
  }
}

static inline bool al16f(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace lr

using namespace lr;

extern "C" int lr_deepfm_l1_fold_stats_f32(const float* partial, int F, int C, int K, int64_t B, float eps,
                                           float momentum, const float* gamma, const float* beta,
                                           float* moving_mean, float* moving_var, float* mean, float* inv,
                                           float* s, float* t, lr_stream_t stream) {
  LR_CHECK_ARG(F >= 1 && C >= 1 && K >= 1 && B >= 1);
  LR_CHECK_ARG(partial && gamma && beta && moving_mean && moving_var && mean && inv && s && t);
  const FoldStats A{partial, F, C, K, B, eps, momentum, gamma, beta, moving_mean, moving_var, mean, inv, s, t};
  hipLaunchKernelGGL(l1_fold_stats_kernel, dim3(grid_for(static_cast<int64_t>(F) * K, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), A);
  return launch_status();
}

// lr_deepfm_l1_fold_stats_f32 + lr_deepfm_l1_fold_bias_f32 in ONE launch (same results, bit for bit): the workgroup of a
// 64-row slab finalises the statistics of its own rows before it forms the slab's bias partial
extern "C" int lr_deepfm_l1_fold_stats_bias_f32(const float* partial, int F, int C, int K, int64_t B, float eps,
                                                float momentum, const float* gamma, const float* beta,
                                                float* moving_mean, float* moving_var, float* mean, float* inv,
                                                float* s, float* t, const float* W, const float* b, int H1,
                                                float* bias_partial, lr_stream_t stream) {
  LR_CHECK_ARG(F >= 1 && C >= 1 && K >= 1 && B >= 1 && H1 >= 1);
  LR_CHECK_ARG(partial && gamma && beta && moving_mean && moving_var && mean && inv && s && t && W && b && bias_partial);
  if (H1 > kBlock || kBlock % H1 != 0) return LR_ESHAPE;
  const int n_rows = F * K;
  const int nblk = static_cast<int>(ceil_div(n_rows, 64));
  const FoldStats A{partial, F, C, K, B, eps, momentum, gamma, beta, moving_mean, moving_var, mean, inv, s, t};
  hipLaunchKernelGGL(l1_fold_bias_kernel<true>, dim3(nblk + 1), dim3(kBlock), 0, as_stream(stream),
                     static_cast<const float*>(nullptr), W, b, n_rows, H1, 64, bias_partial, A);
  return launch_status();
}

extern "C" int lr_deepfm_l1_fold_bias_slabs(int n_rows) { return static_cast<int>(ceil_div(n_rows, 64)) + 1; }

extern "C" int lr_deepfm_l1_fold_bias_f32(const float* t, const float* W, const float* b, int n_rows, int H1,
                                          float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(n_rows >= 1 && H1 >= 1 && t && W && b && partial);
  if (H1 > kBlock || kBlock % H1 != 0) return LR_ESHAPE;
  const int nblk = lr_deepfm_l1_fold_bias_slabs(n_rows) - 1;
  hipLaunchKernelGGL(l1_fold_bias_kernel<false>, dim3(nblk + 1), dim3(kBlock), 0, as_stream(stream), t, W, b, n_rows, H1,
                     64, partial, FoldStats{});
  return launch_status();
}

extern "C" int lr_deepfm_l1_fold_bwd_f32(const float* part, int n_slabs, int n_rows, int H1, int64_t B,
                                         const float* sgz, const float* W, const float* gamma,
                                         const float* beta, const float* mean, const float* inv, float* dW,
                                         float* dgamma, float* dbeta, float* db, float* bn_a, float* bn_c,
                                         lr_stream_t stream) {
  LR_CHECK_ARG(n_slabs >= 1 && n_rows >= 1 && B >= 1 && part && sgz && dW && db);
  LR_CHECK_ARG(al16f(part) && al16f(sgz) && al16f(dW));
  if (gamma != nullptr) {
    LR_CHECK_ARG(W && beta && mean && inv && dgamma && dbeta && bn_a && bn_c && al16f(W));
  }
  hipStream_t s = as_stream(stream);
#define LR_FOLDB(LPR)                                                                                     \
  {                                                                                                       \
    hipLaunchKernelGGL((l1_fold_bwd_kernel<LPR>), dim3(static_cast<int>(ceil_div(n_rows, kBlock / LPR))), \
                       dim3(kBlock), 0, s, part, n_slabs, n_rows, B, sgz, W, gamma, beta, mean, inv, dW,  \
                       dgamma, dbeta, db, bn_a, bn_c);                                                    \
    return launch_status();                                                                               \
  }
  if (H1 == 64) LR_FOLDB(16)
  if (H1 == 128) LR_FOLDB(32)
  if (H1 == 256) LR_FOLDB(64)
#undef LR_FOLDB
  return LR_ESHAPE;
}
