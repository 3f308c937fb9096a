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
#include <cstring>

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


// One 32 x 32 output tile on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32: per output element the same k-ordered f32 fma chain as
// gemm_tile's loop — identical bits): acc[r] (row (r & 3) + 8 (r >> 2) + 4 h, column j of the tile) +=
// sum_k A[k][row0 + row] * Bm[k][col0 + j], both operands k-major in LDS (32 consecutive floats per lane half: conflict-free).
// Round 5: the tail's products ran on the VALU from LDS register tiles — 48 us for the 128 -> 64 backward, 0.19 ms for the chain
// (profiles/r04_deepfm_kernel_trace.md); a 64-sample tile is 64-128 MFMAs per wave.
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ void mfma_tile(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb, int Kd,
                                          int row0, int col0, f32x16& acc) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const float* ap = A + h * lda + row0 + j;
  const float* bp = Bm + h * ldb + col0 + j;
  for (int k = 0; k < Kd; k += 8) {                   // (Kd is a multiple of 16 here: four reduction pairs per round, operands read ahead)
    const float a0 = ap[k * lda], a1 = ap[(k + 2) * lda], a2 = ap[(k + 4) * lda], a3 = ap[(k + 6) * lda];
    const float b0 = bp[k * ldb], b1 = bp[(k + 2) * ldb], b2 = bp[(k + 4) * ldb], b3 = bp[(k + 6) * ldb];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ f32x16 tile_zero() { return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }


// Staging loops: q = tid, tid + 256, ... < n; `load(q)` reads global memory, `store(q, v)` consumes the value.  U loads are issued
// before the first value is used — written as a plain loop (load, use, load, use ...) every iteration pays a full memory round
// trip: 8-32 serialized L2 / HBM latencies per staging loop were what the tail's kernels spent their time on (round 5:
// scripts/lab/r05/tail_marks.py — the 128 <- 64 backward tile took 52 us with 4 us of MFMA work in it).
template <int U, typename T, typename L, typename S>
__device__ __forceinline__ void staged_loop(int n, L load, S store) {
  for (int base = threadIdx.x; base < n; base += U * kBlock) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = base + u * kBlock;
      const int qc = q < n ? q : base;                 // (clamped: every load is unconditional)
      v[u] = load(qc);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = base + u * kBlock;
      if (q < n) store(q, v[u]);
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

// The same with the column's parameters in registers: the staging loops below give a thread ONE column whenever the width
// divides the workgroup size (q = tid + 256 k -> column tid % d), so the four / five parameter loads per ELEMENT (flat loads
// from LDS in the fused tail) become loads per thread.  Same arithmetic, same bits.  (Round 5: the elementwise staging, not the
// products, is what the tail's time was made of — scripts/lab/r05/tail_marks.py.)
struct BnCol { float mean, inv, gamma, beta; bool on; };
__device__ __forceinline__ BnCol bn_col(const BnRef& bn, int c) {
  BnCol b{0.f, 0.f, 0.f, 0.f, bn.mean != nullptr};
  if (b.on) { b.mean = bn.mean[c]; b.inv = bn.inv[c]; b.gamma = bn.gamma[c]; b.beta = bn.beta[c]; }
  return b;
}
__device__ __forceinline__ float bn_act_col(float z, const BnCol& b, float& xhat) {
  const float a = fmaxf(z, 0.f);
  if (!b.on) { xhat = 0.f; return a; }
  xhat = (a - b.mean) * b.inv;
  return fmaf(b.gamma, xhat, b.beta);
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

__device__ __forceinline__ void mlp_colstats_body(int blk, const float* __restrict__ z, int64_t B, int d,
                                                  float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[2 * 1024 + 2 * 256];
  const int64_t b0 = static_cast<int64_t>(blk) * kTT;
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
  float* out = partial + static_cast<int64_t>(blk) * 2 * d;
  tile_colsum_store<false>(s, q, rg, c4, RP, d, active, red, out, out + d);
  __syncthreads();                                   // (`red` is reused by the caller's next tile)
}
__global__ __launch_bounds__(kBlock) void mlp_colstats_kernel(const float* __restrict__ z, int64_t B, int d,
                                                             float* __restrict__ partial) {
  mlp_colstats_body(blockIdx.x, z, B, d, partial);
}

// mean / rsqrt(var + eps) from the partials (fixed order, double), moving averages (momentum m):
// tf.layers.batch_normalization(training=True) + UPDATE_OPS (layers/dense.py:31-41, tf_trainer.py:122-123)
__device__ __forceinline__ void mlp_bn_finalize_body(int cb0, int cb_stride, const float* __restrict__ partial, int nblk, int d,
                                                     int64_t B, float eps, float momentum, float* __restrict__ moving_mean,
                                                     float* __restrict__ moving_var, float* __restrict__ mean_out,
                                                     float* __restrict__ inv_out) {
  __shared__ double red[2][16][17];
  const int cx = threadIdx.x & 15, ky = threadIdx.x >> 4;
  for (int c0 = cb0 * 16; c0 < d; c0 += cb_stride * 16) {
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
__global__ __launch_bounds__(kBlock) void mlp_bn_finalize_kernel(const float* __restrict__ partial, int nblk, int d,
                                                                int64_t B, float eps, float momentum,
                                                                float* __restrict__ moving_mean,
                                                                float* __restrict__ moving_var,
                                                                float* __restrict__ mean_out,
                                                                float* __restrict__ inv_out) {
  mlp_bn_finalize_body(blockIdx.x, gridDim.x, partial, nblk, d, B, eps, momentum, moving_mean, moving_var, mean_out, inv_out);
}

// ---------------------------------------------------------------------------------------------------
// z_out = BN(relu(z_in)) @ W + b   (+ column statistics of relu(z_out) for the next BatchNorm)
//   LDS: At [d_in][kTP] (h, transposed) | Wl [d_in][d_out] | red [16][d_out] x 2
// ---------------------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void mlp_layer_fwd_body(
    int blk, char* smem, const float* __restrict__ z_in, int64_t B, int d_in, BnRef bn, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ z_out, float* __restrict__ partial_out, DropRef drop) {
  constexpr int d_out = NC * 16;
  float* At = reinterpret_cast<float*>(smem);
  float* Wl = At + d_in * kTP;
  float* red = Wl + d_in * d_out;                      // [2][16][d_out]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t b0 = static_cast<int64_t>(blk) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  if (kBlock % d_in == 0) {                            // one column per thread: its BatchNorm parameters in registers
    const int c = tid % d_in;
    const BnCol bc = bn_col(bn, c);
    const int64_t last = (b0 + nb - 1) * d_in;
    staged_loop<8, float>(kTT * d_in,
                          [&](int q) { const int64_t o = b0 * d_in + q; return z_in[o <= last + c ? o : last + c]; },
                          [&](int q, float z) {
                            const int r = q / d_in;
                            float xh;
                            At[c * kTP + r] = r < nb ? bn_act_col(z, bc, xh) * drop_scale(drop, b0 + r, c) : 0.f;
                          });
  } else {
    for (int q = tid; q < kTT * d_in; q += kBlock) {
      const int r = q / d_in, c = q - r * d_in;
      float xh;
      At[c * kTP + r] = r < nb ? bn_act(z_in[(b0 + r) * d_in + c], bn, c, xh) * drop_scale(drop, b0 + r, c) : 0.f;
    }
  }
  staged_loop<8, float4>(d_in * d_out / 4, [&](int q) { return ld4(W + q * 4); }, [&](int q, float4 w) { st4(Wl + q * 4, w); });
  __syncthreads();
  if constexpr (NC % 2 == 0) {
    // 2 sample tiles x d_out / 32 column tiles of 32 x 32, dealt to the 4 waves; column statistics: every lane sums its 16 rows
    // in register order, the (lane half, sample tile) partial sums are added in that order through LDS
    constexpr int CTn = d_out / 32, TILES = 2 * CTn;
    const int wid = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    if (partial_out != nullptr)
      for (int c = tid; c < 2 * 4 * d_out; c += kBlock) red[c] = 0.f;      // [2][4][d_out]: (which, 2 rt + h, column)
    __syncthreads();
    for (int t = wid; t < TILES; t += 4) {
      const int rt = t / CTn, ct = t % CTn;
      f32x16 acc = tile_zero();
      mfma_tile(At, kTP, Wl, d_out, d_in, rt * 32, ct * 32, acc);
      const int col = ct * 32 + j;
      const float bv = bias[col];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[r] + bv;
        if (row < nb) {
          z_out[(b0 + row) * d_out + col] = v;
          const float a = fmaxf(v, 0.f);
          s1 += a;
          s2 = fmaf(a, a, s2);
        }
      }
      if (partial_out != nullptr) {
        red[(0 * 4 + 2 * rt + h) * d_out + col] = s1;
        red[(1 * 4 + 2 * rt + h) * d_out + col] = s2;
      }
    }
    if (partial_out != nullptr) {
      __syncthreads();
      for (int c = tid; c < 2 * d_out; c += kBlock) {
        const int which = c / d_out, col = c - which * d_out;
        float t = 0.f;
        for (int g = 0; g < 4; ++g) t += red[(which * 4 + g) * d_out + col];      // fixed order
        partial_out[(static_cast<int64_t>(blk) * 2 + which) * d_out + col] = t;
      }
    }
    __syncthreads();
    return;
  }
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
      partial_out[(static_cast<int64_t>(blk) * 2 + which) * d_out + col] = t;
    }
  }
  __syncthreads();                                   // (At / Wl / red are reused by the caller's next tile / phase)
}

template <int NC>
__global__ __launch_bounds__(kBlock) void mlp_layer_fwd_kernel(
    const float* __restrict__ z_in, int64_t B, int d_in, BnRef bn, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ z_out, float* __restrict__ partial_out, DropRef drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_layer_fwd_body<NC>(blockIdx.x, smem, z_in, B, d_in, bn, W, bias, z_out, partial_out, drop);
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
__device__ __forceinline__ void mlp_head_body(
    int blk, char* smem, const float* __restrict__ zn, int dn, const float* __restrict__ pair, int K,
    const float* __restrict__ lin_out, int F, const float* __restrict__ labels, const float* wl,
    const float* __restrict__ bl, const float* wo, const float* __restrict__ bo, int64_t B,
    float* __restrict__ logits, float* __restrict__ gl, float* __restrict__ partial) {
  __shared__ float s_lt[kTT], s_gl[kTT], s_loss[kTT];
  const int tid = threadIdx.x;
  const int64_t b0 = static_cast<int64_t>(blk) * kTT;
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
        staged_loop<8, float4>(n / 4, [&](int q) { return ld4(src + q * 4); }, [&](int q, float4 v) { st4(dst + q * 4, v); });
      } else {
        for (int q = tid; q < n; q += kBlock) dst[q] = src[q];
      }
    };
    copy(l_zn, t_zn, nb * dn);
    if (K > 0) copy(l_pair, t_pair, nb * K);
    if (F > 0) copy(l_lin, t_lin, nb * F);
    // the output / linear weights too: read from memory inside the dot-product loops they were one dependent L1 / L2 round
    // trip per term (74 terms per thread at cfg 2: most of the kernel's 20 us)
    float* l_wo = l_lin + kTT * F;                          // [off + K + dn]
    float* l_wl = l_wo + (off + K + dn);                    // [F]
    for (int q = tid; q < off + K + dn; q += kBlock) l_wo[q] = wo[q];
    for (int q = tid; q < F; q += kBlock) l_wl[q] = wl[q];
    __syncthreads();
    t_zn = l_zn; t_pair = l_pair; t_lin = l_lin;
    wo = l_wo; wl = l_wl;
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
  float* out = partial + static_cast<int64_t>(blk) * (G + 1);
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
  __syncthreads();                                   // (s_lt / s_gl / s_loss and the staged tile are reused by the caller's next tile)
}
template <bool kStage>
__global__ __launch_bounds__(kBlock) void mlp_head_kernel(
    const float* __restrict__ zn, int dn, const float* __restrict__ pair, int K, const float* __restrict__ lin_out,
    int F, const float* __restrict__ labels, const float* __restrict__ wl, const float* __restrict__ bl,
    const float* __restrict__ wo, const float* __restrict__ bo, int64_t B, float* __restrict__ logits,
    float* __restrict__ gl, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_head_body<kStage>(blockIdx.x, smem, zn, dn, pair, K, lin_out, F, labels, wl, bl, wo, bo, B, logits, gl, partial);
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

struct BnBwdCol { float mean, inv, gamma, dgamma, dbeta; bool on; };
__device__ __forceinline__ BnBwdCol bn_bwd_col(const BnBwdRef& r, int c) {
  BnBwdCol b{0.f, 0.f, 0.f, 0.f, 0.f, r.bn.mean != nullptr};
  if (b.on) { b.mean = r.bn.mean[c]; b.inv = r.bn.inv[c]; b.gamma = r.bn.gamma[c]; b.dgamma = r.dgamma[c]; b.dbeta = r.dbeta[c]; }
  return b;
}
__device__ __forceinline__ float act_bwd_col(float gh, float z, const BnBwdCol& b, float invB) {
  if (z <= 0.f) return 0.f;
  if (!b.on) return gh;
  const float xhat = (z - b.mean) * b.inv;
  return b.gamma * b.inv * (gh - b.dbeta * invB - xhat * b.dgamma * invB);
}

// ---------------------------------------------------------------------------------------------------
// Backward through one Dense layer  z_out = h_in @ W + b,  h_in = BN_in(relu(z_in)):
//   upstream gz_out: mode 0: gl[s] * wd[o] (z_out is the last layer) ; mode 1: act_bwd(gh_out, z_out, ...)
//   dW partial [d_in][d_out] = h_in^T gz_out ; db partial [d_out] ; gh_in = gz_out @ W^T (stored) ;
//   BN_in sums: sum_s gh_in, sum_s gh_in * xhat_in
//   LDS: G [kTT][d_out] | Gt [d_out][kTP] | H [kTT][d_in] | Wt [d_out][d_in] | red [2][16][d_in]
// ---------------------------------------------------------------------------------------------------
template <int NCI, int NCO>
__device__ __forceinline__ void mlp_layer_bwd_body(
    int blk, char* smem, int mode, const float* __restrict__ gl, const float* __restrict__ wd,
    const float* __restrict__ gh_out, const float* __restrict__ z_out, BnBwdRef up, const float* __restrict__ z_in,
    BnRef bn_in, const float* __restrict__ W, int64_t B, float* __restrict__ gh_in, float* __restrict__ dW_partial,
    float* __restrict__ db_partial, float* __restrict__ bn_partial, DropRef drop_in) {
  constexpr int d_in = NCI * 16, d_out = NCO * 16;
  float* G = reinterpret_cast<float*>(smem);            // [kTT][d_out]
  float* Gt = G + kTT * d_out;                          // [d_out][kTP]
  float* H = Gt + d_out * kTP;                          // [kTT][d_in]   (h_in, natural layout)
  float* Wt = H + kTT * d_in;                           // [d_out][d_in]
  float* red = Wt + d_out * d_in;                       // [2][16][d_in]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t b0 = static_cast<int64_t>(blk) * kTT;
  const int nb = (B - b0) < kTT ? static_cast<int>(B - b0) : kTT;
  const float invB = 1.f / static_cast<float>(B);
  if constexpr (kBlock % d_out == 0) {                 // one column per thread: its parameters in registers
    const int o = tid % d_out;
    const float wdo = mode == 0 ? wd[o] : 0.f;
    const BnBwdCol uc = mode == 0 ? BnBwdCol{0.f, 0.f, 0.f, 0.f, 0.f, false} : bn_bwd_col(up, o);
    const int64_t lastq = (b0 + nb - 1) * d_out + o;
    if (mode == 0) {
      staged_loop<8, float>(kTT * d_out, [&](int q) { const int r = q / d_out; return gl[b0 + (r < nb ? r : nb - 1)]; },
                            [&](int q, float v) {
                              const int r = q / d_out;
                              const float g = r < nb ? v * wdo : 0.f;
                              G[q] = g;
                              Gt[o * kTP + r] = g;
                            });
    } else {
      staged_loop<8, float2>(kTT * d_out,
                             [&](int q) {
                               const int64_t a_ = b0 * d_out + q, ac = a_ <= lastq ? a_ : lastq;
                               return make_float2(gh_out[ac], z_out[ac]);
                             },
                             [&](int q, float2 v) {
                               const int r = q / d_out;
                               const float g = r < nb ? act_bwd_col(v.x, v.y, uc, invB) : 0.f;
                               G[q] = g;
                               Gt[o * kTP + r] = g;
                             });
    }
  } else {
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
  }
  if constexpr (kBlock % d_in == 0) {
    const int c = tid % d_in;
    const BnCol bc = bn_col(bn_in, c);
    const int64_t lastq = (b0 + nb - 1) * d_in + c;
    staged_loop<16, float>(kTT * d_in,
                          [&](int q) { const int64_t a_ = b0 * d_in + q; return z_in[a_ <= lastq ? a_ : lastq]; },
                          [&](int q, float z) {
                            const int r = q / d_in;
                            float xh = 0.f;
                            H[q] = r < nb ? bn_act_col(z, bc, xh) * drop_scale(drop_in, b0 + r, c) : 0.f;   // the Dense saw the dropped h
                          });
  } else {
    for (int q = tid; q < kTT * d_in; q += kBlock) {
      const int r = q / d_in, c = q - r * d_in;
      float xh = 0.f, h = 0.f;
      if (r < nb) h = bn_act(z_in[(b0 + r) * d_in + c], bn_in, c, xh) * drop_scale(drop_in, b0 + r, c);   // the Dense saw the dropped h
      H[q] = h;
    }
  }
  staged_loop<8, float4>(d_in * d_out / 4, [&](int q) { return ld4(W + q * 4); },      // W [d_in][d_out] -> Wt [d_out][d_in]
                         [&](int q, float4 w) {
                           const int i = (q * 4) / d_out, o = q * 4 - i * d_out;
                           Wt[o * d_in + i] = w.x;
                           Wt[(o + 1) * d_in + i] = w.y;
                           Wt[(o + 2) * d_in + i] = w.z;
                           Wt[(o + 3) * d_in + i] = w.w;
                         });
  __syncthreads();
  // ---- dW partial: rows i of h_in^T, cols o; reduction over the 64 samples ------------------------
  float* dWp = dW_partial + static_cast<int64_t>(blk) * d_in * d_out;
  if constexpr (NCI % 2 == 0 && NCO % 2 == 0) {
    // both products as 32 x 32 MFMA tiles dealt to the 4 waves (see mfma_tile: the same fma chains, the same bits)
    constexpr int IT = d_in / 32, OT = d_out / 32;
    const int wid = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    for (int t = wid; t < IT * OT; t += 4) {
      const int it = t / OT, ot = t % OT;
      f32x16 acc = tile_zero();
      mfma_tile(H, d_in, G, d_out, kTT, it * 32, ot * 32, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) dWp[(it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * d_out + ot * 32 + j] = acc[r];
    }
    for (int o = tid; o < d_out; o += kBlock) {
      float t = 0.f;
      for (int r = 0; r < kTT; ++r) t += G[r * d_out + o];
      db_partial[static_cast<int64_t>(blk) * d_out + o] = t;
    }
    if (bn_partial != nullptr)
      for (int c = tid; c < 2 * 4 * d_in; c += kBlock) red[c] = 0.f;        // [2][4][d_in]: (which, 2 rt + h, column)
    __syncthreads();
    for (int t = wid; t < 2 * IT; t += 4) {              // gh_in = gz_out @ W^T: 2 sample tiles x d_in / 32 column tiles
      const int rt = t / IT, ct = t % IT;
      f32x16 acc = tile_zero();
      mfma_tile(Gt, kTP, Wt, d_in, d_out, rt * 32, ct * 32, acc);
      const int col = ct * 32 + j;
      const BnCol bc = bn_col(bn_in, col);
      float zz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        zz[r] = bc.on ? z_in[(b0 + (row < nb ? row : nb - 1)) * d_in + col] : 0.f;
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < nb) {
          const float g = acc[r] * drop_scale(drop_in, b0 + row, col);      // through the dropout: gradient w.r.t. the BatchNorm output
          gh_in[(b0 + row) * d_in + col] = g;
          s1 += g;
          float xh = 0.f;
          if (bc.on) bn_act_col(zz[r], bc, xh);
          s2 = fmaf(g, xh, s2);
        }
      }
      if (bn_partial != nullptr) {
        red[(0 * 4 + 2 * rt + h) * d_in + col] = s1;
        red[(1 * 4 + 2 * rt + h) * d_in + col] = s2;
      }
    }
    if (bn_partial != nullptr) {
      __syncthreads();
      for (int c = tid; c < 2 * d_in; c += kBlock) {
        const int which = c / d_in, col = c - which * d_in;
        float t = 0.f;
        for (int g = 0; g < 4; ++g) t += red[(which * 4 + g) * d_in + col];        // fixed order
        bn_partial[(static_cast<int64_t>(blk) * 2 + which) * d_in + col] = t;
      }
    }
    __syncthreads();
    return;
  }
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
    db_partial[static_cast<int64_t>(blk) * d_out + o] = t;
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
      bn_partial[(static_cast<int64_t>(blk) * 2 + which) * d_in + col] = t;
    }
  }
  __syncthreads();                                   // (the LDS tiles are reused by the caller's next tile / phase)
}
template <int NCI, int NCO>
__global__ __launch_bounds__(kBlock) void mlp_layer_bwd_kernel(
    int mode, const float* __restrict__ gl, const float* __restrict__ wd, const float* __restrict__ gh_out,
    const float* __restrict__ z_out, BnBwdRef up, const float* __restrict__ z_in, BnRef bn_in,
    const float* __restrict__ W, int64_t B, float* __restrict__ gh_in, float* __restrict__ dW_partial,
    float* __restrict__ db_partial, float* __restrict__ bn_partial, DropRef drop_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_layer_bwd_body<NCI, NCO>(blockIdx.x, smem, mode, gl, wd, gh_out, z_out, up, z_in, bn_in, W, B, gh_in, dW_partial,
                               db_partial, bn_partial, drop_in);
}

// (reduce_partials_body: csrc/common.hpp — shared with the weight-pack launches that carry the folded bias's reduction)
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float* __restrict__ partial, int nblk,
                                                                int64_t n, int64_t stride, float* __restrict__ out) {
  reduce_partials_body(blockIdx.x, gridDim.x, partial, nblk, n, stride, out, nullptr);
}

// several independent reductions in ONE launch (blockIdx.y = job): the tail's parameter gradients are only
// needed by the optimiser, so their ten reductions are deferred to the end of the backward
struct ReduceJob {
  const float* partial; float* out;
  int64_t n, stride;
  int nblk;
  float div;        // != 0: out = float(sum) / div (the loss's 1 / B: no elementwise launch behind the reduction); 0: the plain sum
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
      const float r = static_cast<float>(tot);
      J.out[c] = J.div != 0.f ? r / J.div : r;
    }
    __syncthreads();
  }
}

// gz_1 = act_bwd(gh_1, z_1) and its column sums (partial [nblk][d]); 16-byte pieces, all threads (see mlp_colstats_kernel)
__device__ __forceinline__ void mlp_first_bwd_body(int blk, const float* __restrict__ gh, const float* __restrict__ z,
                                                   BnBwdRef up, int64_t B, int d, float* __restrict__ gz,
                                                   float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[1024 + 256];
  const int64_t b0 = static_cast<int64_t>(blk) * kTT;
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
    float* out = partial + static_cast<int64_t>(blk) * d;
    tile_colsum_store<true>(t, f4_zero(), rg, c4, RP, d, active, red, out, out);
    __syncthreads();
  }
}
__global__ __launch_bounds__(kBlock) void mlp_first_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ z,
                                                              BnBwdRef up, int64_t B, int d, float* __restrict__ gz,
                                                              float* __restrict__ partial) {
  mlp_first_bwd_body(blockIdx.x, gh, z, up, B, d, gz, partial);
}


// ---------------------------------------------------------------------------------------------------
// The whole tail of a three-layer dense_nn (z0 [B, d0] -> d1 -> d2 -> output layer -> loss -> backward down to
// gz0) as ONE persistent launch (round 5).  The chain above is ~15 launches of 3-45 us cut at every batch-wide
// BatchNorm reduction: 0.31 ms of the 2.4 ms cfg 2 step and 0.15 ms of the 0.8 ms cfg 3 step, mostly launch
// boundaries and half-empty grids.  Here min(tiles, CUs) workgroups stay resident, run the SAME per-tile bodies
// (bit-identical results) and meet at four grid barriers where a BatchNorm needs the whole batch:
//   colstats(z0) | B | finalize -> layer 0->1 (+ stats of z1) | B | finalize -> layer 1->2, head, backward of layer 2
//   | B | reduce (d beta, d gamma)_1 -> backward of layer 1 | B | reduce (d beta, d gamma)_0 -> gz0 + its column sums
// Every workgroup finalises the statistics / reduces the backward sums for itself (same fixed order -> same bits,
// kept in LDS); workgroup 0 also writes the global copies (moving averages, parameter gradients).  The weight /
// bias gradient partials are summed by the caller's one multi-job reduction launch as before.
// Grid barrier: per-wave drain of its stores, workgroup barrier, lane 0: agent-scope release, arrive on a monotonic
// counter, relaxed poll with s_sleep, agent-scope acquire (MI355X_MICROARCH.md "barrier-counter").  The poll is
// BOUNDED: a workgroup that is not resident after ~2 s sets an error word and the launch finishes (with garbage)
// instead of hanging — the grid never exceeds one workgroup per CU and needs nothing else to be scheduled.
// ---------------------------------------------------------------------------------------------------
struct Tail3Args {
  int64_t B;
  int K, F;
  const float* z0; const float* pair; const float* lin_out; const float* labels;
  // BatchNorm over relu(z0) / relu(z1): all pointers of one are NULL together
  float eps0, mom0; float* mm0; float* mv0; const float* gamma0; const float* beta0; float* dgamma0; float* dbeta0;
  float eps1, mom1; float* mm1; float* mv1; const float* gamma1; const float* beta1; float* dgamma1; float* dbeta1;
  const float* W1; const float* b1; const float* W2; const float* b2;
  const float* wl; const float* bl; const float* wo; const float* bo;
  float* z1; float* z2; float* gh0; float* gh1;
  float* stat0; float* stat1; float* bnp0; float* bnp1;          // [tiles][2][d] partials
  float* mean0; float* inv0; float* mean1; float* inv1;          // global copies of the statistics
  float* dW1p; float* db1p; float* dW2p; float* db2p; float* headp;
  float* gl; float* gz0; float* sgzp;
  uint32_t drop_seed; float keep;
  unsigned* sync;                                                // 18 words: [0] arrivals (zeroed by the launcher), [1] error, [2..] phase marks
};


// Column sums of a [nblk][ncols] partial array for ONE workgroup that needs them all (the fused tail: every workgroup reduces
// for itself): thread c < ncols (<= 256) walks its column with 16 double accumulators — slice g takes the blocks k = g (mod 16)
// in ascending order, the slices are then added in slice order: exactly the sums (and bits) of mlp_bn_finalize_body /
// reduce_partials_body, whose 16 x 16 thread layout leaves a lone workgroup with d / 16 serial latency-bound rounds (measured:
// the first fused tail spent 36 such rounds of ~4 us).  Sixteen independent loads per thread are in flight instead.
__device__ __forceinline__ void tail_colsum_body(const float* __restrict__ partial, int nblk, int ncols, double* __restrict__ out) {
  const int c = threadIdx.x;
  if (c < ncols) {
    double acc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0;
    int k = 0;
    for (; k + 32 <= nblk; k += 32) {                  // 32 loads in flight, added in the same order
      float v[32];
#pragma unroll
      for (int g = 0; g < 32; ++g) v[g] = partial[static_cast<int64_t>(k + g) * ncols + c];
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] += static_cast<double>(v[g]);
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] += static_cast<double>(v[16 + g]);
    }
    for (; k + 16 <= nblk; k += 16) {
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] += static_cast<double>(partial[static_cast<int64_t>(k + g) * ncols + c]);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g)
      if (k + g < nblk) acc[g] += static_cast<double>(partial[static_cast<int64_t>(k + g) * ncols + c]);
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += acc[g];
    out[c] = tot;
  }
  __syncthreads();
}
// mean / rsqrt(var + eps) (+ moving averages) from the sums of tail_colsum_body over a [nblk][2][d] statistics array
__device__ __forceinline__ void tail_bn_finalize(const double* __restrict__ sums, int d, int64_t B, float eps, float momentum,
                                                 float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                 float* __restrict__ mean_out, float* __restrict__ inv_out) {
  const int c = threadIdx.x;
  if (c < d) {
    const double mean = sums[c] / static_cast<double>(B);
    double var = sums[d + c] / static_cast<double>(B) - mean * mean;
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

constexpr unsigned kTailSpinLimit = 2000000u;
constexpr int kTailSticky = 18, kTailLimitWord = 19, kTailDoneWord = 20, kTailSyncWords = 24;   // layout of `sync`: see lr_mlp_tail3_args

// Grid barrier of the one-launch tail.  The launch is a plain one, so co-residency of its workgroups is NOT guaranteed by the
// runtime: the launcher sizes the grid from the device's own occupancy figure (tail3_resident_blocks), and the poll below is
// bounded.  A workgroup whose poll runs out sets sync[1] (this launch) and the STICKY word sync[18] (never cleared by the
// library); every workgroup that sees either (sync[1] while polling, sync[18] at the kernel's entry on any later launch) stops before the next phase — nothing is computed from statistics that did not see the
// whole batch — and writes NaN into its loss partial.  The host reads sync[18] wherever it reads the loss back and raises.
// Returns true when the barrier completed.
__device__ __forceinline__ bool tail_grid_barrier(unsigned* sync, unsigned target, unsigned limit, int* s_fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int fail = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 63u) == 0u && __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        fail = 1;
        break;
      }
      if (spins > limit) {
        __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + kTailSticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_fail = fail;
  }
  __syncthreads();
  return *s_fail == 0;
}

template <int NC0, int NC1, int NC2>
__global__ __launch_bounds__(kBlock) void mlp_tail3_kernel(Tail3Args a) {
  constexpr int d0 = NC0 * 16, d1 = NC1 * 16, d2 = NC2 * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_mean[2][256], s_inv[2][256], s_dg[2][256], s_db[2][256];
  __shared__ double s_sum[256];
  __shared__ int s_fail;
  const int wg = blockIdx.x, G = gridDim.x;
  const int64_t B = a.B;
  const int tiles = static_cast<int>(ceil_div(B, kTT));
  // a barrier that did not complete (now, or in an earlier launch on these buffers): this workgroup's loss partial becomes
  // NaN and it computes nothing further
  const int head_cols = (a.F > 0 ? 2 : 0) + a.K + NC2 * 16 + 1 + a.F;
  auto give_up = [&]() {
    if (threadIdx.x == 0) a.headp[static_cast<int64_t>(wg) * (head_cols + 1) + head_cols] = __builtin_nanf("");
  };
  if (__hip_atomic_load(a.sync + kTailSticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up(); return; }
  unsigned limit = a.sync[kTailLimitWord];
  if (limit == 0u) limit = kTailSpinLimit;
  const bool bn0 = a.gamma0 != nullptr, bn1 = a.gamma1 != nullptr;
  const bool lead = wg == 0;
  unsigned phase = 0;
  const DropRef drop0{a.drop_seed, a.keep, 0}, drop1{a.drop_seed, a.keep, 1};
  const BnRef none{nullptr, nullptr, nullptr, nullptr};
  // profiling aid: workgroup 0 leaves the shader clock (low 32 bits) at every phase boundary in sync[2 ..] (16 marks)
  int mark_i = 2;
  auto mark = [&]() {
    if (lead && threadIdx.x == 0 && mark_i < 18) a.sync[mark_i] = static_cast<unsigned>(__builtin_amdgcn_s_memtime());
    ++mark_i;
  };
  mark();

  if (bn0) {
    for (int t = wg; t < tiles; t += G) mlp_colstats_body(t, a.z0, B, d0, a.stat0);
    mark();
    if (!tail_grid_barrier(a.sync, ++phase * G, limit, &s_fail)) { give_up(); return; }
    mark();
    tail_colsum_body(a.stat0, tiles, 2 * d0, s_sum);
    tail_bn_finalize(s_sum, d0, B, a.eps0, a.mom0, lead ? a.mm0 : nullptr, lead ? a.mv0 : nullptr, s_mean[0], s_inv[0]);
    if (lead)
      for (int c = threadIdx.x; c < d0; c += kBlock) { a.mean0[c] = s_mean[0][c]; a.inv0[c] = s_inv[0][c]; }
  }
  const BnRef r0 = bn0 ? BnRef{s_mean[0], s_inv[0], a.gamma0, a.beta0} : none;
  mark();
  for (int t = wg; t < tiles; t += G)
    mlp_layer_fwd_body<NC1>(t, smem, a.z0, B, d0, r0, a.W1, a.b1, a.z1, bn1 ? a.stat1 : nullptr, drop0);
  mark();
  if (bn1) {
    if (!tail_grid_barrier(a.sync, ++phase * G, limit, &s_fail)) { give_up(); return; }
    mark();
    tail_colsum_body(a.stat1, tiles, 2 * d1, s_sum);
    tail_bn_finalize(s_sum, d1, B, a.eps1, a.mom1, lead ? a.mm1 : nullptr, lead ? a.mv1 : nullptr, s_mean[1], s_inv[1]);
    if (lead)
      for (int c = threadIdx.x; c < d1; c += kBlock) { a.mean1[c] = s_mean[1][c]; a.inv1[c] = s_inv[1][c]; }
  }
  const BnRef r1 = bn1 ? BnRef{s_mean[1], s_inv[1], a.gamma1, a.beta1} : none;
  const int off = a.F > 0 ? 1 : 0;
  const float* wd = a.wo + off + a.K;                 // the deep term's output weights
  mark();
  for (int t = wg; t < tiles; t += G) {
    mlp_layer_fwd_body<NC2>(t, smem, a.z1, B, d1, r1, a.W2, a.b2, a.z2, nullptr, drop1);
    if (t == wg) mark();
    mlp_head_body<true>(t, smem, a.z2, d2, a.pair, a.K, a.lin_out, a.F, a.labels, a.wl, a.bl, a.wo, a.bo, B, nullptr, a.gl,
                        a.headp);
    if (t == wg) mark();
    mlp_layer_bwd_body<NC1, NC2>(t, smem, 0, a.gl, wd, nullptr, nullptr, BnBwdRef{none, nullptr, nullptr}, a.z1, r1, a.W2, B,
                                 a.gh1, a.dW2p, a.db2p, bn1 ? a.bnp1 : nullptr, drop1);
  }
  mark();
  if (bn1) {
    if (!tail_grid_barrier(a.sync, ++phase * G, limit, &s_fail)) { give_up(); return; }
    mark();
    tail_colsum_body(a.bnp1, tiles, 2 * d1, s_sum);     // [0, d): sum gh = d beta; [d, 2 d): sum gh * xhat = d gamma
    if (threadIdx.x < d1) {
      const float db = static_cast<float>(s_sum[threadIdx.x]), dg = static_cast<float>(s_sum[d1 + threadIdx.x]);
      s_db[1][threadIdx.x] = db; s_dg[1][threadIdx.x] = dg;
      if (lead) { a.dbeta1[threadIdx.x] = db; a.dgamma1[threadIdx.x] = dg; }
    }
    __syncthreads();
  }
  const BnBwdRef up1 = bn1 ? BnBwdRef{BnRef{s_mean[1], s_inv[1], a.gamma1, nullptr}, s_dg[1], s_db[1]}
                           : BnBwdRef{none, nullptr, nullptr};
  mark();
  for (int t = wg; t < tiles; t += G)
    mlp_layer_bwd_body<NC0, NC1>(t, smem, 1, nullptr, nullptr, a.gh1, a.z1, up1, a.z0, r0, a.W1, B, a.gh0, a.dW1p, a.db1p,
                                 bn0 ? a.bnp0 : nullptr, drop0);
  mark();
  if (bn0) {
    if (!tail_grid_barrier(a.sync, ++phase * G, limit, &s_fail)) { give_up(); return; }
    mark();
    tail_colsum_body(a.bnp0, tiles, 2 * d0, s_sum);
    if (threadIdx.x < d0) {
      const float db = static_cast<float>(s_sum[threadIdx.x]), dg = static_cast<float>(s_sum[d0 + threadIdx.x]);
      s_db[0][threadIdx.x] = db; s_dg[0][threadIdx.x] = dg;
      if (lead) { a.dbeta0[threadIdx.x] = db; a.dgamma0[threadIdx.x] = dg; }
    }
    __syncthreads();
  }
  const BnBwdRef up0 = bn0 ? BnBwdRef{BnRef{s_mean[0], s_inv[0], a.gamma0, nullptr}, s_dg[0], s_db[0]}
                           : BnBwdRef{none, nullptr, nullptr};
  mark();
  for (int t = wg; t < tiles; t += G) mlp_first_bwd_body(t, a.gh0, a.z0, up0, B, d0, a.gz0, a.sgzp);
  mark();
  // the last workgroup to get here (every other one has passed the last barrier) clears the arrival counter for the next
  // launch on these words: no zeroing launch in front of the kernel
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(a.sync + kTailDoneWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == static_cast<unsigned>(G) - 1u) {
      __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.sync + kTailDoneWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
  const size_t lds = static_cast<size_t>(kTT) * (static_cast<size_t>(dn) + K + F) * 4 + (static_cast<size_t>(1) + K + dn + F) * 4;
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

// ---- the fused three-layer tail -----------------------------------------------------------------------------------
static_assert(2 * 128 <= kBlock, "one thread per statistics column");
static_assert(sizeof(Tail3Args) == sizeof(lr_mlp_tail3_args), "Tail3Args mirrors lr_mlp_tail3_args");

extern "C" int lr_mlp_tail3_supported(int d0, int d1, int d2, int K, int F) {
  if (!(d0 == 128 && d1 == 64 && d2 == 32)) return 0;
  if (K < 0 || F < 0 || K % 4 != 0) return 0;
  const size_t head = static_cast<size_t>(kTT) * (static_cast<size_t>(d2) + K + F) * 4 + (static_cast<size_t>(1) + K + d2 + F) * 4;
  return head <= 100 * 1024 ? 1 : 0;
}

// Workgroups of the fused tail that can be resident at once on the current device (cached per device; 0 on error).
static int tail3_resident_blocks(const void* kern, size_t lds) {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  const int c = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (c > 0) return c;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds) != hipSuccess) return 0;
  if (per_cu > 1) per_cu = 1;                          // one per CU: the phases are sized for a CU's whole LDS bandwidth
  const int r = cus * per_cu;
  if (r > 0) __atomic_store_n(&cached[dev], r, __ATOMIC_RELAXED);
  return r;
}

extern "C" int lr_mlp_tail3_resident_blocks(void) {
  auto kern = mlp_tail3_kernel<8, 4, 2>;
  // the largest LDS request the launcher can make (K = F = 0 leaves the backward tile of the widest layer pair)
  const size_t lds = static_cast<size_t>(kTT) * 64 * 4 + static_cast<size_t>(64) * kTP * 4 + static_cast<size_t>(kTT) * 128 * 4 +
                     static_cast<size_t>(64) * 128 * 4 + 2 * 16 * static_cast<size_t>(128) * 4;
  if (tail_lds(kern, lds) != LR_OK) return 0;
  return tail3_resident_blocks(reinterpret_cast<const void*>(kern), lds);
}

extern "C" int lr_mlp_tail3_f32(const lr_mlp_tail3_args* args, lr_stream_t stream) {
  LR_CHECK_ARG(args != nullptr);
  Tail3Args a;
  memcpy(&a, args, sizeof(a));
  LR_CHECK_ARG(a.B >= 1 && a.z0 && a.labels && a.W1 && a.b1 && a.W2 && a.b2 && a.wo && a.bo && a.sync);
  LR_CHECK_ARG(a.z1 && a.z2 && a.gh0 && a.gh1 && a.dW1p && a.db1p && a.dW2p && a.db2p && a.headp && a.gl && a.gz0 && a.sgzp);
  LR_CHECK_ARG((a.K > 0) == (a.pair != nullptr));
  LR_CHECK_ARG((a.F > 0) == (a.lin_out != nullptr) && (a.F > 0) == (a.wl != nullptr) && (a.F > 0) == (a.bl != nullptr));
  LR_CHECK_ARG((a.gamma0 == nullptr) == (a.beta0 == nullptr) && (a.gamma0 == nullptr) == (a.stat0 == nullptr) &&
               (a.gamma0 == nullptr) == (a.bnp0 == nullptr) && (a.gamma0 == nullptr) == (a.dgamma0 == nullptr) &&
               (a.gamma0 == nullptr) == (a.dbeta0 == nullptr) && (a.gamma0 == nullptr) == (a.mean0 == nullptr) &&
               (a.gamma0 == nullptr) == (a.inv0 == nullptr));
  LR_CHECK_ARG((a.gamma1 == nullptr) == (a.beta1 == nullptr) && (a.gamma1 == nullptr) == (a.stat1 == nullptr) &&
               (a.gamma1 == nullptr) == (a.bnp1 == nullptr) && (a.gamma1 == nullptr) == (a.dgamma1 == nullptr) &&
               (a.gamma1 == nullptr) == (a.dbeta1 == nullptr) && (a.gamma1 == nullptr) == (a.mean1 == nullptr) &&
               (a.gamma1 == nullptr) == (a.inv1 == nullptr));
  LR_CHECK_ARG(a.keep > 0.f);
  if (!lr_mlp_tail3_supported(128, 64, 32, a.K, a.F)) return LR_ESHAPE;
  for (const void* p : {static_cast<const void*>(a.z0), static_cast<const void*>(a.W1), static_cast<const void*>(a.W2),
                        static_cast<const void*>(a.z1), static_cast<const void*>(a.z2), static_cast<const void*>(a.gh0),
                        static_cast<const void*>(a.gh1), static_cast<const void*>(a.gz0)})
    if (!al16t(p)) return LR_EINVAL;
  constexpr int d0 = 128, d1 = 64, d2 = 32;
  const int tiles = static_cast<int>(ceil_div(a.B, kTT));
  // dynamic LDS: the largest of the phases' tiles
  auto fwd = [](int di, int dq) {
    return static_cast<size_t>(di) * kTP * 4 + static_cast<size_t>(di) * dq * 4 + 2 * 16 * static_cast<size_t>(dq) * 4;
  };
  auto bwd = [](int di, int dq) {
    return static_cast<size_t>(kTT) * dq * 4 + static_cast<size_t>(dq) * kTP * 4 + static_cast<size_t>(kTT) * di * 4 +
           static_cast<size_t>(dq) * di * 4 + 2 * 16 * static_cast<size_t>(di) * 4;
  };
  size_t lds = static_cast<size_t>(kTT) * (static_cast<size_t>(d2) + a.K + a.F) * 4 + (static_cast<size_t>(1) + a.K + d2 + a.F) * 4;
  for (size_t x : {fwd(d0, d1), fwd(d1, d2), bwd(d1, d2), bwd(d0, d1)}) lds = x > lds ? x : lds;
  auto kern = mlp_tail3_kernel<8, 4, 2>;
  int rc = tail_lds(kern, lds);
  if (rc != LR_OK) return rc;
  // every workgroup must be resident at once: no more of them than the device itself says fit (CUs of THIS device x the
  // kernel's own occupancy at this LDS size), never a constant
  const int resident = tail3_resident_blocks(reinterpret_cast<const void*>(kern), lds);
  if (resident < 1) return LR_ESHAPE;
  const int grid = tiles < resident ? tiles : resident;
  hipStream_t s = as_stream(stream);
  // (no zeroing launch: the caller hands `sync` over zeroed once, the kernel leaves the arrival counter at 0 — see the header)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, s, a);
  return launch_status();
}
