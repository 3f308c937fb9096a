// DeepFM "tail": everything of dense_nn after the first Dense layer (layers/dense.py:33-49), the output
// layer over [linear term | pairwise term | deep term] (algorithms/deepfm.py:158-172), the sigmoid
// cross-entropy loss (tfops/loss.py:14-16) and the backward of all of it, as a handful of small kernels.
//
// The tensors here are tiny ([B, <=256] activations, <=64 KB weight matrices) but a framework executes the
// chain as ~100 separate launches per step (0.66 ms of the 3.85 ms cfg 2 step, profiles/r02_*).  Batch-
// statistics BatchNorm forces a grid-wide reduction between consecutive layers, so the chain is cut at those
// points and nowhere else:
//
//   forward   colstats(z_1) | bn_finalize | layer_fwd(1->2)+colstats | bn_finalize | ... | head
//   backward  layer_bwd(n->n-1) | bn_bwd_finalize | ... | first_bwd | reduce_partials
//
// Workgroup = 64 samples; the small matrix products run on the VALU from LDS tiles (4 x NC register tile per
// thread); every batch reduction is a per-workgroup partial + a fixed-order second pass (no atomics:
// run-to-run bit identical).  Widths: multiples of 16, <= 256.
#include "common.hpp"

namespace lr {

constexpr int kTT = 64;          // samples per workgroup
constexpr int kTP = kTT + 4;     // padded sample stride of transposed LDS tiles

// acc[r][c] += sum_k At[k][row0 + 4*ty + r] * Wl[k][tx*NC + c]      (At: [Kd][lda], Wl: [Kd][ldw])
template <int NC>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ At, int lda, const float* __restrict__ Wl,
                                          int ldw, int Kd, int row0, float (&acc)[4][NC]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const float* ap = At + row0 + 4 * ty;
  const float* wp = Wl + tx * NC;
#pragma unroll 4
  for (int k = 0; k < Kd; ++k) {
    const float4 a = ld4(ap + k * lda);
    float w[NC];
    if constexpr (NC % 4 == 0) {
#pragma unroll
      for (int c = 0; c < NC; c += 4) {
        const float4 x = ld4(wp + k * ldw + c);
        w[c] = x.x; w[c + 1] = x.y; w[c + 2] = x.z; w[c + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) w[c] = wp[k * ldw + c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      acc[0][c] = fmaf(a.x, w[c], acc[0][c]);
      acc[1][c] = fmaf(a.y, w[c], acc[1][c]);
      acc[2][c] = fmaf(a.z, w[c], acc[2][c]);
      acc[3][c] = fmaf(a.w, w[c], acc[3][c]);
    }
  }
}

struct BnRef {            // batch-statistics BatchNorm of one layer's activation (all nullable together)
  const float* mean; const float* inv; const float* gamma; const float* beta;
};

// Dropout after a hidden layer's BatchNorm (layers/dense.py:44-47: tf.layers.dropout(net, rate, training): kept entries scaled
// by 1 / keep).  The mask is a counter-based function of (seed, layer, sample, column) — splitmix64's finaliser — so the backward
// kernels regenerate it instead of storing it, and a test can restate it (tests/test_tail_dropout_gpu.py).  keep >= 1: off.
struct DropRef {
  uint32_t seed; float keep; int layer;
};
__device__ __forceinline__ float drop_scale(const DropRef& d, int64_t sample, int c) {
  if (d.keep >= 1.f) return 1.f;
  uint64_t x = (static_cast<uint64_t>(sample) * 4096ull + static_cast<uint64_t>(c)) ^
               (static_cast<uint64_t>(d.seed) * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(d.layer) * 0xD1B54A32D192ED03ull);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  const float u = static_cast<float>(x & 0xFFFFFFull) * (1.f / 16777216.f);
  return u < d.keep ? 1.f / d.keep : 0.f;
}

// h = BN(relu(z)) (or relu(z)); also returns x_hat for the backward
__device__ __forceinline__ float bn_act(float z, const BnRef& bn, int c, float& xhat) {
  const float a = fmaxf(z, 0.f);
  if (bn.mean == nullptr) { xhat = 0.f; return a; }
  xhat = (a - bn.mean[c]) * bn.inv[c];
  return fmaf(bn.gamma[c], xhat, bn.beta[c]);
}

// ---------------------------------------------------------------------------------------------------
// column sums of relu(z) and relu(z)^2 per workgroup of 64 samples: partial[blk][{0,1}][d]
// ---------------------------------------------------------------------------------------------------
// Round 4: all 256 threads move 16-byte pieces — row group rg = tid / (d/4) takes rows rg, rg + RP, ... of the tile (RP =
// 256 / (d/4) rows per pass), the RP group sums of a column are added in group order through LDS.  (One thread per column walking
// the 64 rows serially left half the workgroup idle at d = 128 and took 18 us for 8 MB.)
template <bool kGrad>
__device__ __forceinline__ void tile_colsum_store(float4 s, float4 q, int rg, int c4, int RP, int d, bool active, float* red,
                                                  float* __restrict__ out0, float* __restrict__ out1) {
  // red: [2][RP][d]
  if (active) {
    st4(red + rg * d + c4, s);
    if (!kGrad) st4(red + (RP + rg) * d + c4, q);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < (kGrad ? d : 2 * d); c += kBlock) {
    const int which = c / d, col = c - which * d;
    float t = 0.f;
    for (int g = 0; g < RP; ++g) t += red[(which * RP + g) * d + col];        // fixed order
    (which == 0 ? out0 : out1)[col] = t;
  }
}

__global__ __launch_bounds__(kBlock) void mlp_colstats_kernel(const float* __restrict__ z, int64_t B, int d,
                                                             float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[2 * 1024 + 2 * 256];
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  const int cq = d / 4, RP = kBlock / cq;
  const int rg = threadIdx.x / cq, c4 = (threadIdx.x % cq) * 4;
  const bool active = rg < RP;
  float4 s = f4_zero(), q = f4_zero();
  if (active)
    for (int r = rg; r < nb; r += RP) {
      float4 a = ld4(z + (b0 + r) * d + c4);
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      s = f4_add(s, a);
      q = f4_fma(a, a, q);
    }
  float* out = partial + static_cast<int64_t>(blockIdx.x) * 2 * d;
  tile_colsum_store<false>(s, q, rg, c4, RP, d, active, red, out, out + d);
}

// mean / rsqrt(var + eps) from the partials (fixed order, double), moving averages (momentum m):
// tf.layers.batch_normalization(training=True) + UPDATE_OPS (layers/dense.py:31-41, tf_trainer.py:122-123)
__global__ __launch_bounds__(kBlock) void mlp_bn_finalize_kernel(const float* __restrict__ partial, int nblk, int d,
                                                                int64_t B, float eps, float momentum,
                                                                float* __restrict__ moving_mean,
                                                                float* __restrict__ moving_var,
                                                                float* __restrict__ mean_out,
                                                                float* __restrict__ inv_out) {
  __shared__ double red[2][16][17];
  const int cx = threadIdx.x & 15, ky = threadIdx.x >> 4;
  for (int c0 = blockIdx.x * 16; c0 < d; c0 += gridDim.x * 16) {
    const int c = c0 + cx;
    double s = 0.0, q = 0.0;
    if (c < d)
      for (int k = ky; k < nblk; k += 16) {
        s += static_cast<double>(partial[(static_cast<int64_t>(k) * 2 + 0) * d + c]);
        q += static_cast<double>(partial[(static_cast<int64_t>(k) * 2 + 1) * d + c]);
      }
    red[0][ky][cx] = s;
    red[1][ky][cx] = q;
    __syncthreads();
    if (ky == 0 && c < d) {
      s = 0.0; q = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) { s += red[0][g][cx]; q += red[1][g][cx]; }
      const double mean = s / static_cast<double>(B);
      double var = q / static_cast<double>(B) - mean * mean;
      if (var < 0.0) var = 0.0;
      const float mf = static_cast<float>(mean), vf = static_cast<float>(var);
      mean_out[c] = mf;
      inv_out[c] = 1.0f / sqrtf(vf + eps);
      if (moving_mean != nullptr) {
        moving_mean[c] = fmaf(moving_mean[c], momentum, mf * (1.f - momentum));
        moving_var[c] = fmaf(moving_var[c], momentum, vf * (1.f - momentum));
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// z_out = BN(relu(z_in)) @ W + b   (+ column statistics of relu(z_out) for the next BatchNorm)
//   LDS: At [d_in][kTP] (h, transposed) | Wl [d_in][d_out] | red [16][d_out] x 2
// ---------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(kBlock) void mlp_layer_fwd_kernel(
    const float* __restrict__ z_in, int64_t B, int d_in, BnRef bn, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ z_out, float* __restrict__ partial_out, DropRef drop) {
  constexpr int d_out = NC * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* At = reinterpret_cast<float*>(smem);
  float* Wl = At + d_in * kTP;
  float* red = Wl + d_in * d_out;                      // [2][16][d_out]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  for (int q = tid; q < kTT * d_in; q += kBlock) {
    const int r = q / d_in, c = q - r * d_in;
    float xh;
    At[c * kTP + r] = r < nb ? bn_act(z_in[(b0 + r) * d_in + c], bn, c, xh) * drop_scale(drop, b0 + r, c) : 0.f;
  }
  for (int q = tid; q < d_in * d_out / 4; q += kBlock) st4(Wl + q * 4, ld4(W + q * 4));
  __syncthreads();
  float acc[4][NC];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = 0.f;
  gemm_tile<NC>(At, kTP, Wl, d_out, d_in, 0, acc);
  float s[NC], q2[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { s[c] = 0.f; q2[c] = 0.f; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * ty + r;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float v = acc[r][c] + bias[tx * NC + c];
      if (row < nb) {
        z_out[(b0 + row) * d_out + tx * NC + c] = v;
        const float a = fmaxf(v, 0.f);
        s[c] += a;
        q2[c] = fmaf(a, a, q2[c]);
      }
    }
  }
  if (partial_out != nullptr) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      red[ty * d_out + tx * NC + c] = s[c];
      red[(16 + ty) * d_out + tx * NC + c] = q2[c];
    }
    __syncthreads();
    for (int c = tid; c < 2 * d_out; c += kBlock) {
      const int which = c / d_out, col = c - which * d_out;
      float t = 0.f;
      for (int g = 0; g < 16; ++g) t += red[(which * 16 + g) * d_out + col];     // fixed order
      partial_out[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d_out + col] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Output layer + loss (deepfm.py:158, 171-172; tfops/loss.py:14-16, mean over the batch):
//   lt = lin_out @ wl + bl ; logit = wo[0]*lt + pair @ wo[1:1+K] + z_n @ wo[1+K:] + bo
//   loss_b = max(x,0) - x*y + log1p(exp(-|x|)) ; gl = (sigmoid(x) - y) / B
// Gradient partials per workgroup, layout [wo (1+K+dn) | bo | wl (F) | bl]; loss partial at the end.
// Plain form (F == 0: no linear term, wl / bl / lin_out NULL; K == 0: no pairwise term, pair NULL) — the
// output layer of DIN / YouTubeRanking (algorithms/din.py:190-192): logit = z_n @ wo + bo, partials
// [wo (K+dn) | bo], loss partial at the end.
// ---------------------------------------------------------------------------------------------------
// `kStage` (round 4): the tile's rows of zn / pair / lin_out — each a CONTIGUOUS block of the [B, *] arrays — are first copied
// into LDS with 16-byte loads by all 256 threads; the per-sample dot products and the per-column gradient sums then read LDS.
// (Reading lin_out [B, 202] column by column straight from memory, 64 dependent steps per thread, took 27 us for 19 MB.)
// Same arithmetic, same summation order.  Without room in LDS (F in the thousands) the direct form runs.
template <bool kStage>
__global__ __launch_bounds__(kBlock) void mlp_head_kernel(
    const float* __restrict__ zn, int dn, const float* __restrict__ pair, int K, const float* __restrict__ lin_out,
    int F, const float* __restrict__ labels, const float* __restrict__ wl, const float* __restrict__ bl,
    const float* __restrict__ wo, const float* __restrict__ bo, int64_t B, float* __restrict__ logits,
    float* __restrict__ gl, float* __restrict__ partial) {
  __shared__ float s_lt[kTT], s_gl[kTT], s_loss[kTT];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  const int off = F > 0 ? 1 : 0;                  // wo[0] weighs the linear term when there is one
  const int G = off + K + dn + 1 + F + off;
  // row blocks of the tile: global pointers, or their LDS copies
  const float* t_zn = zn + b0 * dn;
  const float* t_pair = K > 0 ? pair + b0 * K : nullptr;
  const float* t_lin = F > 0 ? lin_out + b0 * F : nullptr;
  if (kStage) {
    float* l_zn = reinterpret_cast<float*>(smem);
    float* l_pair = l_zn + kTT * dn;
    float* l_lin = l_pair + kTT * K;                         // (kTT * dn and kTT * K are multiples of 4 floats)
    auto copy = [&](float* dst, const float* src, int n) {    // n floats, contiguous; 16-byte pieces where aligned
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (n & 3) == 0) {
        for (int q = tid; q < n / 4; q += kBlock) st4(dst + q * 4, ld4(src + q * 4));
      } else {
        for (int q = tid; q < n; q += kBlock) dst[q] = src[q];
      }
    };
    copy(l_zn, t_zn, nb * dn);
    if (K > 0) copy(l_pair, t_pair, nb * K);
    if (F > 0) copy(l_lin, t_lin, nb * F);
    __syncthreads();
    t_zn = l_zn; t_pair = l_pair; t_lin = l_lin;
  }
  {   // 4 threads per sample split the three dot products; partner lanes are adjacent
    const int r = tid >> 2, part = tid & 3;
    float lt = 0.f, acc = 0.f;
    if (r < nb) {
      for (int f = part; f < F; f += 4) lt = fmaf(t_lin[r * F + f], wl[f], lt);
      for (int k = part; k < K; k += 4) acc = fmaf(t_pair[r * K + k], wo[off + k], acc);
      for (int j = part; j < dn; j += 4) acc = fmaf(t_zn[r * dn + j], wo[off + K + j], acc);
    }
    lt += __shfl_xor(lt, 1); lt += __shfl_xor(lt, 2);
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (part == 0) {
      float g = 0.f, l = 0.f;
      if (off) lt += bl[0];
      if (r < nb) {
        const float x = (off ? fmaf(wo[0], lt, acc) : acc) + bo[0];
        const float y = labels[b0 + r];
        l = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        const float sg = 1.f / (1.f + expf(-x));
        g = (sg - y) / static_cast<float>(B);
        if (logits != nullptr) logits[b0 + r] = x;
        gl[b0 + r] = g;
      }
      s_lt[r] = lt; s_gl[r] = g; s_loss[r] = l;
    }
  }
  __syncthreads();
  float* out = partial + static_cast<int64_t>(blockIdx.x) * (G + 1);
  const float wo0 = off ? wo[0] : 0.f;
  for (int c = tid; c < G + 1; c += kBlock) {
    float t = 0.f;
    if (c == G) {
      for (int r = 0; r < nb; ++r) t += s_loss[r];
    } else if (c < off) {
      for (int r = 0; r < nb; ++r) t = fmaf(s_gl[r], s_lt[r], t);
    } else if (c < off + K) {
      for (int r = 0; r < nb; ++r) t = fmaf(s_gl[r], t_pair[r * K + (c - off)], t);
    } else if (c < off + K + dn) {
      for (int r = 0; r < nb; ++r) t = fmaf(s_gl[r], t_zn[r * dn + (c - off - K)], t);
    } else if (c == off + K + dn) {
      for (int r = 0; r < nb; ++r) t += s_gl[r];
    } else if (c < off + K + dn + 1 + F) {
      for (int r = 0; r < nb; ++r) t = fmaf(s_gl[r], t_lin[r * F + (c - off - 1 - K - dn)], t);
      t *= wo0;
    } else {      // c == G - 1 with a linear term: d bl
      for (int r = 0; r < nb; ++r) t += s_gl[r];
      t *= wo0;
    }
    out[c] = t;
  }
}

// gradient w.r.t. the layer's pre-activation z from the gradient w.r.t. its BatchNorm output h:
//   ga = gamma*inv*(gh - dbeta/B - xhat*dgamma/B) ; gz = ga * (z > 0)        (no BN: gz = gh * (z > 0))
struct BnBwdRef {
  BnRef bn; const float* dgamma; const float* dbeta;
};
__device__ __forceinline__ float act_bwd(float gh, float z, const BnBwdRef& r, int c, float invB) {
  if (z <= 0.f) return 0.f;
  if (r.bn.mean == nullptr) return gh;
  const float xhat = (z - r.bn.mean[c]) * r.bn.inv[c];          // z > 0: relu(z) = z
  return r.bn.gamma[c] * r.bn.inv[c] * (gh - r.dbeta[c] * invB - xhat * r.dgamma[c] * invB);
}

// ---------------------------------------------------------------------------------------------------
// Backward through one Dense layer  z_out = h_in @ W + b,  h_in = BN_in(relu(z_in)):
//   upstream gz_out: mode 0: gl[s] * wd[o] (z_out is the last layer) ; mode 1: act_bwd(gh_out, z_out, ...)
//   dW partial [d_in][d_out] = h_in^T gz_out ; db partial [d_out] ; gh_in = gz_out @ W^T (stored) ;
//   BN_in sums: sum_s gh_in, sum_s gh_in * xhat_in
//   LDS: G [kTT][d_out] | Gt [d_out][kTP] | H [kTT][d_in] | Wt [d_out][d_in] | red [2][16][d_in]
// ---------------------------------------------------------------------------------------------------
template <int NCI, int NCO>
__global__ __launch_bounds__(kBlock) void mlp_layer_bwd_kernel(
    int mode, const float* __restrict__ gl, const float* __restrict__ wd, const float* __restrict__ gh_out,
    const float* __restrict__ z_out, BnBwdRef up, const float* __restrict__ z_in, BnRef bn_in,
    const float* __restrict__ W, int64_t B, float* __restrict__ gh_in, float* __restrict__ dW_partial,
    float* __restrict__ db_partial, float* __restrict__ bn_partial, DropRef drop_in) {
  constexpr int d_in = NCI * 16, d_out = NCO * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* G = reinterpret_cast<float*>(smem);            // [kTT][d_out]
  float* Gt = G + kTT * d_out;                          // [d_out][kTP]
  float* H = Gt + d_out * kTP;                          // [kTT][d_in]   (h_in, natural layout)
  float* Wt = H + kTT * d_in;                           // [d_out][d_in]
  float* red = Wt + d_out * d_in;                       // [2][16][d_in]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  const float invB = 1.f / static_cast<float>(B);
  for (int q = tid; q < kTT * d_out; q += kBlock) {
    const int r = q / d_out, o = q - r * d_out;
    float g = 0.f;
    if (r < nb) {
      if (mode == 0) g = gl[b0 + r] * wd[o];
      else g = act_bwd(gh_out[(b0 + r) * d_out + o], z_out[(b0 + r) * d_out + o], up, o, invB);
    }
    G[q] = g;
    Gt[o * kTP + r] = g;
  }
  for (int q = tid; q < kTT * d_in; q += kBlock) {
    const int r = q / d_in, c = q - r * d_in;
    float xh = 0.f, h = 0.f;
    if (r < nb) h = bn_act(z_in[(b0 + r) * d_in + c], bn_in, c, xh) * drop_scale(drop_in, b0 + r, c);   // the Dense saw the dropped h
    H[q] = h;
  }
  for (int q = tid; q < d_in * d_out; q += kBlock) {      // W [d_in][d_out] -> Wt [d_out][d_in]
    const int i = q / d_out, o = q - i * d_out;
    Wt[o * d_in + i] = W[q];
  }
  __syncthreads();
  // ---- dW partial: rows i of h_in^T, cols o; reduction over the 64 samples ------------------------
  float* dWp = dW_partial + static_cast<int64_t>(blockIdx.x) * d_in * d_out;
  for (int row0 = 0; row0 < d_in; row0 += 64) {
    float acc[4][NCO];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < NCO; ++c) acc[r][c] = 0.f;
    gemm_tile<NCO>(H, d_in, G, d_out, kTT, row0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = row0 + 4 * ty + r;
      if (i < d_in) {
#pragma unroll
        for (int c = 0; c < NCO; ++c) dWp[i * d_out + tx * NCO + c] = acc[r][c];
      }
    }
  }
  for (int o = tid; o < d_out; o += kBlock) {
    float t = 0.f;
    for (int r = 0; r < kTT; ++r) t += G[r * d_out + o];
    db_partial[static_cast<int64_t>(blockIdx.x) * d_out + o] = t;
  }
  // ---- gh_in = gz_out @ W^T ------------------------------------------------------------------------
  float acc[4][NCI];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < NCI; ++c) acc[r][c] = 0.f;
  gemm_tile<NCI>(Gt, kTP, Wt, d_in, d_out, 0, acc);
  float s1[NCI], s2[NCI];
#pragma unroll
  for (int c = 0; c < NCI; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * ty + r;
    if (row < nb) {
#pragma unroll
      for (int c = 0; c < NCI; ++c) {
        const int col = tx * NCI + c;
        const float g = acc[r][c] * drop_scale(drop_in, b0 + row, col);     // through the dropout: gradient w.r.t. the BatchNorm output
        gh_in[(b0 + row) * d_in + col] = g;
        s1[c] += g;
        float xh = 0.f;
        if (bn_in.mean != nullptr) bn_act(z_in[(b0 + row) * d_in + col], bn_in, col, xh);
        s2[c] = fmaf(g, xh, s2[c]);
      }
    }
  }
  if (bn_partial != nullptr) {
#pragma unroll
    for (int c = 0; c < NCI; ++c) {
      red[ty * d_in + tx * NCI + c] = s1[c];
      red[(16 + ty) * d_in + tx * NCI + c] = s2[c];
    }
    __syncthreads();
    for (int c = tid; c < 2 * d_in; c += kBlock) {
      const int which = c / d_in, col = c - which * d_in;
      float t = 0.f;
      for (int g = 0; g < 16; ++g) t += red[(which * 16 + g) * d_in + col];
      bn_partial[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d_in + col] = t;
    }
  }
}

// out[c] = sum_k partial[k*stride + c] in a fixed order.  Workgroup = 16 columns x 16 k-slices: thread (cx, ky)
// sums k = ky, ky+16, ... (double), the 16 slices are combined in slice order through LDS.
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float* __restrict__ partial, int nblk,
                                                                int64_t n, int64_t stride, float* __restrict__ out) {
  __shared__ double red[16][17];
  const int cx = threadIdx.x & 15, ky = threadIdx.x >> 4;
  for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * 16; c0 < n; c0 += static_cast<int64_t>(gridDim.x) * 16) {
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
      out[c] = static_cast<float>(tot);
    }
    __syncthreads();
  }
}

// several independent reductions in ONE launch (blockIdx.y = job): the tail's parameter gradients are only
// needed by the optimiser, so their ten reductions are deferred to the end of the backward
struct ReduceJob {
  const float* partial; float* out;
  int64_t n, stride;
  int nblk, pad;
};
__global__ __launch_bounds__(kBlock) void reduce_partials_multi_kernel(const ReduceJob* __restrict__ jobs) {
  __shared__ double red[16][17];
  const ReduceJob J = jobs[blockIdx.y];
  const int cx = threadIdx.x & 15, ky = threadIdx.x >> 4;
  for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * 16; c0 < J.n; c0 += static_cast<int64_t>(gridDim.x) * 16) {
    const int64_t c = c0 + cx;
    double t = 0.0;
    if (c < J.n)
      for (int k = ky; k < J.nblk; k += 16) t += static_cast<double>(J.partial[static_cast<int64_t>(k) * J.stride + c]);
    red[ky][cx] = t;
    __syncthreads();
    if (ky == 0 && c < J.n) {
      double tot = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) tot += red[g][cx];
      J.out[c] = static_cast<float>(tot);
    }
    __syncthreads();
  }
}

// gz_1 = act_bwd(gh_1, z_1) and its column sums (partial [nblk][d]); 16-byte pieces, all threads (see mlp_colstats_kernel)
__global__ __launch_bounds__(kBlock) void mlp_first_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ z,
                                                              BnBwdRef up, int64_t B, int d, float* __restrict__ gz,
                                                              float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[1024 + 256];
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  const float invB = 1.f / static_cast<float>(B);
  const int cq = d / 4, RP = kBlock / cq;
  const int rg = threadIdx.x / cq, c4 = (threadIdx.x % cq) * 4;
  const bool active = rg < RP;
  float4 t = f4_zero();
  if (active)
    for (int r = rg; r < nb; r += RP) {
      const int64_t q = (b0 + r) * d + c4;
      const float4 g4 = ld4(gh + q), z4 = ld4(z + q);
      float4 o;
      o.x = act_bwd(g4.x, z4.x, up, c4 + 0, invB);
      o.y = act_bwd(g4.y, z4.y, up, c4 + 1, invB);
      o.z = act_bwd(g4.z, z4.z, up, c4 + 2, invB);
      o.w = act_bwd(g4.w, z4.w, up, c4 + 3, invB);
      st4(gz + q, o);
      t = f4_add(t, o);
    }
  if (partial != nullptr) {
    float* out = partial + static_cast<int64_t>(blockIdx.x) * d;
    tile_colsum_store<true>(t, f4_zero(), rg, c4, RP, d, active, red, out, out);
  }
}

static inline bool tail_width_ok(int d) { return d >= 16 && d <= 256 && d % 16 == 0; }
static inline bool al16t(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace lr

using namespace lr;

extern "C" int lr_mlp_tail_supported(int d_in, int d_out) {
  if (!tail_width_ok(d_in) || !tail_width_ok(d_out)) return 0;
  // backward LDS: G + Gt + H + Xh + Wt + red
  const size_t lds = static_cast<size_t>(kTT) * d_out * 4 + static_cast<size_t>(d_out) * kTP * 4 +
                     static_cast<size_t>(kTT) * d_in * 4 + static_cast<size_t>(d_out) * d_in * 4 +
                     2 * 16 * static_cast<size_t>(d_in) * 4;
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int lr_mlp_colstats_f32(const float* z, int64_t B, int d, float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(z && partial && B >= 1 && d >= 1);
  if (d % 4 != 0 || d > 256 || !al16t(z)) return LR_ESHAPE;      // 16-byte pieces; <= 64 column quads per row pass
  hipLaunchKernelGGL(mlp_colstats_kernel, dim3(static_cast<int>(ceil_div(B, kTT))), dim3(kBlock), 0,
                     as_stream(stream), z, B, d, partial);
  return launch_status();
}

extern "C" int lr_mlp_bn_finalize_f32(const float* partial, int nblk, int d, int64_t B, float eps, float momentum,
                                      float* moving_mean, float* moving_var, float* mean_out, float* inv_out,
                                      lr_stream_t stream) {
  LR_CHECK_ARG(partial && mean_out && inv_out && nblk >= 1 && d >= 1 && B >= 1);
  LR_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
  hipLaunchKernelGGL(mlp_bn_finalize_kernel, dim3((d + 15) / 16), dim3(kBlock), 0, as_stream(stream), partial, nblk, d, B,
                     eps, momentum, moving_mean, moving_var, mean_out, inv_out);
  return launch_status();
}

#define LR_NC_SWITCH(nc, ...)                                                              \
  switch (nc) {                                                                             \
    case 1: { constexpr int NCV = 1; __VA_ARGS__ } break;                                          \
    case 2: { constexpr int NCV = 2; __VA_ARGS__ } break;                                          \
    case 3: { constexpr int NCV = 3; __VA_ARGS__ } break;                                          \
    case 4: { constexpr int NCV = 4; __VA_ARGS__ } break;                                          \
    case 6: { constexpr int NCV = 6; __VA_ARGS__ } break;                                          \
    case 8: { constexpr int NCV = 8; __VA_ARGS__ } break;                                          \
    case 12: { constexpr int NCV = 12; __VA_ARGS__ } break;                                        \
    case 16: { constexpr int NCV = 16; __VA_ARGS__ } break;                                        \
    default: return LR_ESHAPE;                                                              \
  }

template <typename Kern>
static int tail_lds(Kern kern, size_t bytes) {
  if (bytes > 160 * 1024) return LR_ESHAPE;
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return LR_OK;
}

extern "C" int lr_mlp_layer_fwd_f32(const float* z_in, int64_t B, int d_in, const float* mean, const float* inv,
                                    const float* gamma, const float* beta, const float* W, const float* bias,
                                    int d_out, float* z_out, float* partial_out, uint32_t drop_seed, float drop_keep,
                                    int drop_layer, lr_stream_t stream) {
  LR_CHECK_ARG(z_in && W && bias && z_out && B >= 1);
  LR_CHECK_ARG((mean == nullptr) == (inv == nullptr) && (mean == nullptr) == (gamma == nullptr) &&
               (mean == nullptr) == (beta == nullptr));
  LR_CHECK_ARG(al16t(W));
  if (!tail_width_ok(d_in) || !tail_width_ok(d_out)) return LR_ESHAPE;
  const size_t lds = static_cast<size_t>(d_in) * kTP * 4 + static_cast<size_t>(d_in) * d_out * 4 +
                     2 * 16 * static_cast<size_t>(d_out) * 4;
  const BnRef bn{mean, inv, gamma, beta};
  LR_CHECK_ARG(drop_keep > 0.f && d_in <= 4096);
  const DropRef drop{drop_seed, drop_keep, drop_layer};
  const int grid = static_cast<int>(ceil_div(B, kTT));
  LR_NC_SWITCH(d_out / 16, {
    int rc = tail_lds(mlp_layer_fwd_kernel<NCV>, lds);
    if (rc != LR_OK) return rc;
    hipLaunchKernelGGL((mlp_layer_fwd_kernel<NCV>), dim3(grid), dim3(kBlock), lds, as_stream(stream), z_in, B,
                       d_in, bn, W, bias, z_out, partial_out, drop);
  })
  return launch_status();
}

extern "C" int lr_mlp_head_f32(const float* zn, int dn, const float* pair, int K, const float* lin_out, int F,
                               const float* labels, const float* wl, const float* bl, const float* wo,
                               const float* bo, int64_t B, float* logits, float* gl, float* partial,
                               lr_stream_t stream) {
  LR_CHECK_ARG(zn && labels && wo && bo && gl && partial);
  LR_CHECK_ARG(B >= 1 && dn >= 1 && K >= 0 && F >= 0);
  LR_CHECK_ARG((K > 0) == (pair != nullptr));
  LR_CHECK_ARG((F > 0) == (lin_out != nullptr) && (F > 0) == (wl != nullptr) && (F > 0) == (bl != nullptr));
  const dim3 grid(static_cast<int>(ceil_div(B, kTT)));
  const size_t lds = static_cast<size_t>(kTT) * (static_cast<size_t>(dn) + K + F) * 4;
  if (lds <= 120 * 1024 && dn % 4 == 0 && K % 4 == 0) {
    int rc = tail_lds(mlp_head_kernel<true>, lds);
    if (rc != LR_OK) return rc;
    hipLaunchKernelGGL(mlp_head_kernel<true>, grid, dim3(kBlock), lds, as_stream(stream), zn, dn, pair, K, lin_out, F, labels, wl,
                       bl, wo, bo, B, logits, gl, partial);
  } else {
    hipLaunchKernelGGL(mlp_head_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), zn, dn, pair, K, lin_out, F, labels, wl,
                       bl, wo, bo, B, logits, gl, partial);
  }
  return launch_status();
}

extern "C" int lr_mlp_layer_bwd_f32(int mode, const float* gl, const float* wd, const float* gh_out,
                                    const float* z_out, const float* up_mean, const float* up_inv,
                                    const float* up_gamma, const float* up_dgamma, const float* up_dbeta,
                                    const float* z_in, const float* in_mean, const float* in_inv,
                                    const float* in_gamma, const float* in_beta, const float* W, int d_in,
                                    int d_out, int64_t B, float* gh_in, float* dW_partial, float* db_partial,
                                    float* bn_partial, uint32_t drop_seed, float drop_keep, int drop_layer,
                                    lr_stream_t stream) {
  LR_CHECK_ARG(B >= 1 && z_in && W && gh_in && dW_partial && db_partial);
  LR_CHECK_ARG(mode == 0 ? (gl && wd) : (gh_out && z_out));
  LR_CHECK_ARG((in_mean == nullptr) == (bn_partial == nullptr));
  if (!lr_mlp_tail_supported(d_in, d_out)) return LR_ESHAPE;
  const size_t lds = static_cast<size_t>(kTT) * d_out * 4 + static_cast<size_t>(d_out) * kTP * 4 +
                     static_cast<size_t>(kTT) * d_in * 4 + static_cast<size_t>(d_out) * d_in * 4 +
                     2 * 16 * static_cast<size_t>(d_in) * 4;
  const BnBwdRef up{BnRef{up_mean, up_inv, up_gamma, nullptr}, up_dgamma, up_dbeta};
  const BnRef bin{in_mean, in_inv, in_gamma, in_beta};
  LR_CHECK_ARG(drop_keep > 0.f);
  const DropRef drop_in{drop_seed, drop_keep, drop_layer};
  const int grid = static_cast<int>(ceil_div(B, kTT));
  const int nci = d_in / 16, nco = d_out / 16;
#define LR_BWD(NI, NO)                                                                                  \
  if (nci == NI && nco == NO) {                                                                         \
    int rc = tail_lds(mlp_layer_bwd_kernel<NI, NO>, lds);                                               \
    if (rc != LR_OK) return rc;                                                                         \
    hipLaunchKernelGGL((mlp_layer_bwd_kernel<NI, NO>), dim3(grid), dim3(kBlock), lds, as_stream(stream), \
                       mode, gl, wd, gh_out, z_out, up, z_in, bin, W, B, gh_in, dW_partial, db_partial,  \
                       bn_partial, drop_in);                                                            \
    return launch_status();                                                                             \
  }
  // (d_in, d_out) pairs of the usual pyramids: 256/128/64/32/16 halvings and equal widths
  LR_BWD(16, 8) LR_BWD(16, 16) LR_BWD(8, 4) LR_BWD(8, 8) LR_BWD(8, 2) LR_BWD(4, 2) LR_BWD(4, 4) LR_BWD(4, 1)
  LR_BWD(2, 1) LR_BWD(2, 2) LR_BWD(1, 1) LR_BWD(16, 4) LR_BWD(8, 1) LR_BWD(16, 2)
#undef LR_BWD
  return LR_ESHAPE;
}

extern "C" int lr_reduce_partials_f32(const float* partial, int nblk, int64_t n, int64_t stride, float* out,
                                      lr_stream_t stream) {
  LR_CHECK_ARG(partial && out && nblk >= 1 && n >= 1 && stride >= n);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid_for(n, 16)), dim3(kBlock), 0, as_stream(stream), partial,
                     nblk, n, stride, out);
  return launch_status();
}

extern "C" size_t lr_reduce_job_bytes(void) { return sizeof(ReduceJob); }

extern "C" int lr_reduce_partials_multi_f32(const void* jobs_dev, int n_jobs, int64_t max_n, lr_stream_t stream) {
  LR_CHECK_ARG(jobs_dev != nullptr && n_jobs >= 1 && n_jobs <= 65535 && max_n >= 1);
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(grid_for(max_n, 16, 512), n_jobs), dim3(kBlock), 0,
                     as_stream(stream), static_cast<const ReduceJob*>(jobs_dev));
  return launch_status();
}

extern "C" int lr_mlp_first_bwd_f32(const float* gh, const float* z, const float* mean, const float* inv,
                                    const float* gamma, const float* dgamma, const float* dbeta, int64_t B, int d,
                                    float* gz, float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(gh && z && gz && B >= 1 && d >= 1);
  if (d % 4 != 0 || d > 256 || !al16t(gh) || !al16t(z) || !al16t(gz)) return LR_ESHAPE;
  LR_CHECK_ARG((mean == nullptr) == (inv == nullptr) && (mean == nullptr) == (gamma == nullptr) &&
               (mean == nullptr) == (dgamma == nullptr) && (mean == nullptr) == (dbeta == nullptr));
  const BnBwdRef up{BnRef{mean, inv, gamma, nullptr}, dgamma, dbeta};
  hipLaunchKernelGGL(mlp_first_bwd_kernel, dim3(static_cast<int>(ceil_div(B, kTT))), dim3(kBlock), 0,
                     as_stream(stream), gh, z, up, B, d, gz, partial);
  return launch_status();
}
