// DIN attention pooling (layers/attention.py:28-64 of the reference), forward and backward,
// with the key/query rows gathered straight from the item table (no [N+1,K'] materialisation,
// no [B,L,4K'] cross tensor).
//
// Mapping: ONE WAVEFRONT PER SAMPLE.  A row group of LPR = K/4 lanes holds 4 dims of q and of
// one key; the 64/LPR groups of the wave process different keys concurrently.  The first MLP
// layer is folded per sample:  z_l = b1 + (W1a+W1c)^T q + (W1b - W1c + diag(q) W1d)^T key_l,
// so the per-key work is a [4 x 16] register tile per lane (64 FMAs) followed by a butterfly
// reduce-scatter over the group (15-16 shuffles) that leaves hidden unit j on lane j.  Softmax
// is computed online per group and merged across groups; only positions l < len are touched
// (masked positions get weight exactly 0 in the reference: exp(-(2^32)+1 - max) == 0 in fp32).
// H (hidden units of the attention MLP) is 16, the reference's fixed value.
#include <stdlib.h>

#include "common.hpp"
#include "din_mfma.hpp"

namespace lr {

constexpr int kH = 16;

template <int LPR>
struct DinCfg {
  static constexpr int K = LPR * 4;
  static constexpr int SLOTS = kWave / LPR;
  static constexpr int RS = LPR < 16 ? LPR : 16;  // lanes taking part in the reduce-scatter
  static constexpr int NV = kH / RS;              // hidden units left per lane afterwards
};

// ---- butterfly helpers over a row group -------------------------------------------------
// in: v[16] partial sums per lane.  out: v[0..NV) = group totals of hidden units jidx(i).
template <int LPR>
__device__ __forceinline__ void reduce_scatter16(float (&v)[kH], int gl) {
  constexpr int RS = DinCfg<LPR>::RS;
#pragma unroll
  for (int s = 0; (1 << s) < RS; ++s) {
    const int w = 8 >> s;
    const bool bit = (gl >> s) & 1;
#pragma unroll
    for (int i = 0; i < w; ++i) {
      const float keep = bit ? v[i + w] : v[i];
      const float send = bit ? v[i] : v[i + w];
      v[i] = keep + __shfl_xor(send, 1 << s);
    }
  }
  if constexpr (LPR >= 32) v[0] += __shfl_xor(v[0], 16);
  if constexpr (LPR >= 64) v[0] += __shfl_xor(v[0], 32);
}
// hidden-unit index of v[i] after reduce_scatter16 on lane gl
template <int LPR>
__device__ __forceinline__ int hidden_index(int i, int gl) {
  constexpr int RS = DinCfg<LPR>::RS;
  int j = i;
#pragma unroll
  for (int s = 0; (1 << s) < RS; ++s) j += ((gl >> s) & 1) * (8 >> s);
  return j;
}
// inverse: v[0..NV) per lane -> all 16 values on every lane of the group
template <int LPR>
__device__ __forceinline__ void all_gather16(float (&v)[kH], int gl) {
  constexpr int RS = DinCfg<LPR>::RS;
  constexpr int S = RS == 16 ? 4 : RS == 8 ? 3 : 2;
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    const int w = 8 >> s;
    const bool bit = (gl >> s) & 1;
#pragma unroll
    for (int i = 0; i < w; ++i) {
      const float r = __shfl_xor(v[i], 1 << s);
      const float lo = bit ? r : v[i];
      const float hi = bit ? v[i] : r;
      v[i] = lo;
      v[i + w] = hi;
    }
  }
}
template <int LPR>
__device__ __forceinline__ float group_sum_rs(float x) {  // all-reduce over the RS lanes
  constexpr int RS = DinCfg<LPR>::RS;
#pragma unroll
  for (int o = 1; o < RS; o <<= 1) x += __shfl_xor(x, o);
  return x;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {  // all-reduce over the LPR lanes
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) x += __shfl_xor(x, o);
  return x;
}

// ---- W1 in LDS: [4 terms][4 d][LPR l][16], 16-byte slots rotated by (l>>2) ---------------
// (lanes of one ds_read_b128 group then cover all 64 banks: conflict-free)
template <int LPR>
__device__ __forceinline__ void stage_w1(const float* __restrict__ W1, float* __restrict__ Wl) {
  constexpr int K = LPR * 4;
  for (int q = threadIdx.x; q < 4 * K * 4; q += kBlock) {  // one 16-byte slot per iteration
    const int k = q >> 2, s = q & 3;
    const int t = k / K, kk = k - t * K, l = kk >> 2, d = kk & 3;
    const int phys = ((t * 4 + d) * LPR + l) * kH + (((s + (l >> 2)) & 3) << 2);
    st4(Wl + phys, ld4(W1 + k * kH + s * 4));
  }
}
template <int LPR>
__device__ __forceinline__ void load_w1_row(const float* __restrict__ Wl, int t, int d, int gl,
                                            float (&w)[kH]) {
  const float* row = Wl + ((t * 4 + d) * LPR + gl) * kH;
  const int rot = gl >> 2;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 x = ld4(row + (((s + rot) & 3) << 2));
    w[4 * s] = x.x; w[4 * s + 1] = x.y; w[4 * s + 2] = x.z; w[4 * s + 3] = x.w;
  }
}

// hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp) — the attention kernels are instruction-bound
// and libm's expf / IEEE division cost ~40 instructions per call; tolerance of the tests: 1e-5
__device__ __forceinline__ float sigmoidf_(float z) { return __frcp_rn(1.f + __expf(-z)); }

template <int LPR, bool GATHER>
__device__ __forceinline__ float4 load_row4(const float* __restrict__ src, int64_t V,
                                            const int32_t* __restrict__ ids, int64_t pos, int c4) {
  constexpr int K = LPR * 4;
  if constexpr (GATHER) {
    const int32_t id = ids[pos];
    return (id >= 0 && id < V) ? ld4(src + static_cast<int64_t>(id) * K + c4) : f4_zero();
  } else {
    return ld4(src + pos * K + c4);
  }
}

// folded first layer for one sample: Wk[d][j] and the q-part A (reduce-scattered, + b1)
template <int LPR>
__device__ __forceinline__ void fold_layer1(const float* __restrict__ Wl, float4 q4, int gl,
                                            const float* __restrict__ b1, float (&Wk)[4][kH],
                                            float (&A)[kH]) {
  const float qd[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
  for (int j = 0; j < kH; ++j) A[j] = 0.f;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float wa[kH], wb[kH], wc[kH], wd[kH];
    load_w1_row<LPR>(Wl, 0, d, gl, wa);
    load_w1_row<LPR>(Wl, 1, d, gl, wb);
    load_w1_row<LPR>(Wl, 2, d, gl, wc);
    load_w1_row<LPR>(Wl, 3, d, gl, wd);
#pragma unroll
    for (int j = 0; j < kH; ++j) {
      Wk[d][j] = fmaf(qd[d], wd[j], wb[j] - wc[j]);
      A[j] = fmaf(qd[d], wa[j] + wc[j], A[j]);
    }
  }
  reduce_scatter16<LPR>(A, gl);
#pragma unroll
  for (int i = 0; i < DinCfg<LPR>::NV; ++i) A[i] += b1[hidden_index<LPR>(i, gl)];
}

template <int LPR>
__device__ __forceinline__ void key_preact(const float (&Wk)[4][kH], float4 k4, float (&p)[kH]) {
#pragma unroll
  for (int j = 0; j < kH; ++j)
    p[j] = fmaf(k4.w, Wk[3][j], fmaf(k4.z, Wk[2][j], fmaf(k4.y, Wk[1][j], k4.x * Wk[0][j])));
}

// =========================================================================================
// forward
// =========================================================================================
template <int LPR, bool GATHER>
__global__ __launch_bounds__(kBlock) void din_fwd_kernel(
    const float* __restrict__ qsrc, const float* __restrict__ ksrc, int64_t V,
    const int32_t* __restrict__ item, const int32_t* __restrict__ seq,
    const int32_t* __restrict__ len, int64_t B, int L, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    float* __restrict__ out, float* __restrict__ attn) {
  using Cfg = DinCfg<LPR>;
  constexpr int K = Cfg::K, SLOTS = Cfg::SLOTS, NV = Cfg::NV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wl = reinterpret_cast<float*>(smem);
  float* sc_all = Wl + 4 * K * kH;  // [4 waves][L]
  stage_w1<LPR>(W1, Wl);
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  const int gl = lane % LPR, slot = lane / LPR, c4 = gl * 4;
  float* sc = sc_all + wid * L;
  float w2v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) w2v[i] = W2[hidden_index<LPR>(i, gl)];
  const float b2v = b2[0];
  const float rsK = 1.0f / sqrtf(static_cast<float>(K));

  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / kWave);
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + wid; b < B; b += nwaves) {
    int n = len[b];
    n = n < 0 ? 0 : (n > L ? L : n);
    const float4 q4 = load_row4<LPR, GATHER>(qsrc, V, item, b, c4);
    float Wk[4][kH], A[kH];
    fold_layer1<LPR>(Wl, q4, gl, b1, Wk, A);

    float m = -INFINITY, den = 0.f;
    float4 acc = f4_zero();
    const int iters = (n + SLOTS - 1) / SLOTS;
    for (int it = 0; it < iters; ++it) {
      const int l = it * SLOTS + slot;
      const bool act = l < n;
      const float4 k4 = act ? load_row4<LPR, GATHER>(ksrc, V, seq, b * L + l, c4) : f4_zero();
      float p[kH];
      key_preact<LPR>(Wk, k4, p);
      reduce_scatter16<LPR>(p, gl);
      float sl = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) sl = fmaf(w2v[i], sigmoidf_(A[i] + p[i]), sl);
      const float s = (group_sum_rs<LPR>(sl) + b2v) * rsK;
      if (act) {
        const float mn = fmaxf(m, s);
        const float f = __expf(m - mn);  // m = -inf first time -> 0
        const float e = __expf(s - mn);
        den = fmaf(den, f, e);
        acc = f4_fma(make_float4(e, e, e, e), k4, f4_scale(acc, f));
        m = mn;
        if (gl == 0) sc[l] = s;
      }
    }
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {  // merge the groups' online-softmax states
      const float m2 = __shfl_xor(m, o), den2 = __shfl_xor(den, o);
      const float4 acc2 = f4_shfl_xor(acc, o);
      const float mn = fmaxf(m, m2);
      const float f1 = (m == -INFINITY) ? 0.f : __expf(m - mn);
      const float f2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
      den = den * f1 + den2 * f2;
      acc = f4_add(f4_scale(acc, f1), f4_scale(acc2, f2));
      m = mn;
    }
    const float inv = den > 0.f ? 1.f / den : 0.f;
    if (slot == 0) st4(out + b * K + c4, f4_scale(acc, inv));
    for (int l = lane; l < L; l += kWave)
      attn[b * L + l] = (l < n) ? __expf(sc[l] - m) * inv : 0.f;
  }
}

// =========================================================================================
// backward: per-position gradients + per-block partials of the MLP parameter gradients
// partial layout per block: GA[K*16] | GB[K*16] | GD[K*16] | db1[16] | dW2[16] | db2[1] (+pad)
// =========================================================================================
__host__ __device__ inline int din_partial_floats(int K) { return 3 * K * kH + 2 * kH + 4; }

template <int LPR, bool GATHER>
__global__ __launch_bounds__(kBlock) void din_bwd_kernel(
    const float* __restrict__ qsrc, const float* __restrict__ ksrc, int64_t V,
    const int32_t* __restrict__ item, const int32_t* __restrict__ seq,
    const int32_t* __restrict__ len, int64_t B, int L, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ attn, const float* __restrict__ gout, float* __restrict__ gq,
    float* __restrict__ gkey, float* __restrict__ partial) {
  using Cfg = DinCfg<LPR>;
  constexpr int K = Cfg::K, SLOTS = Cfg::SLOTS, NV = Cfg::NV;
  constexpr int NW = kBlock / kWave;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wl = reinterpret_cast<float*>(smem);          // [4K*16]
  float* accum = Wl + 4 * K * kH;                      // [NW][3][K*16]
  float* small = accum + NW * 3 * K * kH;              // [NW][2*16+4]
  float* da_all = small + NW * (2 * kH + 4);           // [NW][L]
  stage_w1<LPR>(W1, Wl);
  for (int q = threadIdx.x; q < NW * 3 * K * kH + NW * (2 * kH + 4); q += kBlock) accum[q] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  const int gl = lane % LPR, slot = lane / LPR, c4 = gl * 4;
  float* da = da_all + wid * L;
  float* GA = accum + wid * 3 * K * kH;
  float* GB = GA + K * kH;
  float* GD = GB + K * kH;
  float w2v[NV], dW2acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    w2v[i] = W2[hidden_index<LPR>(i, gl)];
    dW2acc[i] = 0.f;
  }
  float db1acc[kH];
#pragma unroll
  for (int j = 0; j < kH; ++j) db1acc[j] = 0.f;
  float db2acc = 0.f;
  const float rsK = 1.0f / sqrtf(static_cast<float>(K));

  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * NW;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * NW + wid; b < B; b += nwaves) {
    int n = len[b];
    n = n < 0 ? 0 : (n > L ? L : n);
    const float4 q4 = load_row4<LPR, GATHER>(qsrc, V, item, b, c4);
    const float4 go = ld4(gout + b * K + c4);
    float Wk[4][kH], A[kH];
    fold_layer1<LPR>(Wl, q4, gl, b1, Wk, A);
    const int iters = (n + SLOTS - 1) / SLOTS;

    // pass 1: da_l = <gout, key_l>, dot = sum_l a_l da_l
    float dot = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int l = it * SLOTS + slot;
      const bool act = l < n;
      const float4 k4 = act ? load_row4<LPR, GATHER>(ksrc, V, seq, b * L + l, c4) : f4_zero();
      const float d = group_sum<LPR>(go.x * k4.x + go.y * k4.y + go.z * k4.z + go.w * k4.w);
      if (act) {
        dot = fmaf(attn[b * L + l], d, dot);
        if (gl == 0) da[l] = d;
      }
    }
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) dot += __shfl_xor(dot, o);

    // pass 2
    float Kz[4][kH], Dz[kH];
#pragma unroll
    for (int j = 0; j < kH; ++j) {
      Dz[j] = 0.f;
      Kz[0][j] = Kz[1][j] = Kz[2][j] = Kz[3][j] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
      const int l = it * SLOTS + slot;
      const bool act = l < n;
      const float4 k4 = act ? load_row4<LPR, GATHER>(ksrc, V, seq, b * L + l, c4) : f4_zero();
      float p[kH];
      key_preact<LPR>(Wk, k4, p);
      reduce_scatter16<LPR>(p, gl);
      const float a_l = act ? attn[b * L + l] : 0.f;
      const float draw = act ? a_l * (da[l] - dot) * rsK : 0.f;  // d loss / d (pre-scale score)
      float dz[kH];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float h = sigmoidf_(A[i] + p[i]);
        dW2acc[i] = fmaf(draw, h, dW2acc[i]);
        dz[i] = draw * w2v[i] * h * (1.f - h);
      }
      if (gl == 0) db2acc += draw;  // one lane per group (each group owns different keys)
      all_gather16<LPR>(dz, gl);
      const float kd[4] = {k4.x, k4.y, k4.z, k4.w};
      float g[4] = {a_l * go.x, a_l * go.y, a_l * go.z, a_l * go.w};
#pragma unroll
      for (int j = 0; j < kH; ++j) {
        Dz[j] += dz[j];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          g[d] = fmaf(dz[j], Wk[d][j], g[d]);
          Kz[d][j] = fmaf(kd[d], dz[j], Kz[d][j]);
        }
      }
      if (act) st4(gkey + (b * L + l) * K + c4, make_float4(g[0], g[1], g[2], g[3]));
    }
    for (int l = n + slot; l < L; l += SLOTS) st4(gkey + (b * L + l) * K + c4, f4_zero());

    // combine the groups' partial sums over keys
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {
#pragma unroll
      for (int j = 0; j < kH; ++j) {
        Dz[j] += __shfl_xor(Dz[j], o);
#pragma unroll
        for (int d = 0; d < 4; ++d) Kz[d][j] += __shfl_xor(Kz[d][j], o);
      }
    }
    // dq_d = sum_j (W1a+W1c)[d][j] Dz_j + W1d[d][j] Kz[d][j]
    const float qd[4] = {q4.x, q4.y, q4.z, q4.w};
    float dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float wa[kH], wc[kH], wd[kH];
      load_w1_row<LPR>(Wl, 0, d, gl, wa);
      load_w1_row<LPR>(Wl, 2, d, gl, wc);
      load_w1_row<LPR>(Wl, 3, d, gl, wd);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < kH; ++j) s = fmaf(wa[j] + wc[j], Dz[j], fmaf(wd[j], Kz[d][j], s));
      dq[d] = s;
    }
    if (slot == 0) {
      st4(gq + b * K + c4, make_float4(dq[0], dq[1], dq[2], dq[3]));
      // parameter-gradient accumulators of this wave (LDS, private to the wave)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int j = 0; j < kH; ++j) {
          const int o = (c4 + d) * kH + j;
          GA[o] = fmaf(qd[d], Dz[j], GA[o]);
          GB[o] += Kz[d][j];
          GD[o] = fmaf(qd[d], Kz[d][j], GD[o]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kH; ++j) db1acc[j] += Dz[j];
  }

  // ---- block epilogue: fold the 4 waves, emit this block's partial ------------------------
  float* sm = small + wid * (2 * kH + 4);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < kH; ++j) sm[j] = db1acc[j];
  }
  {  // db2: one contribution per group-leader lane; sum them over the wave
    float x = db2acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) sm[2 * kH] = x;
  }
  {  // dW2 lives reduce-scattered on the first RS lanes; every group saw only its own keys
    float tmp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      tmp[i] = dW2acc[i];
#pragma unroll
      for (int o = LPR; o < kWave; o <<= 1) tmp[i] += __shfl_xor(tmp[i], o);
    }
    if (slot == 0 && gl < Cfg::RS) {
#pragma unroll
      for (int i = 0; i < NV; ++i) sm[kH + hidden_index<LPR>(i, gl)] = tmp[i];
    }
  }
  __syncthreads();
  float* dst = partial + static_cast<int64_t>(blockIdx.x) * din_partial_floats(K);
  for (int q = threadIdx.x; q < 3 * K * kH; q += kBlock) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += accum[w * 3 * K * kH + q];
    dst[q] = s;
  }
  for (int q = threadIdx.x; q < 2 * kH + 1; q += kBlock) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += small[w * (2 * kH + 4) + q];
    dst[3 * K * kH + q] = s;
  }
}

// final, deterministic reduction over blocks; emits gW1 [4K,16] (= GA | GB | GA-GB | GD)
__global__ __launch_bounds__(kBlock) void din_reduce_kernel(const float* __restrict__ partial,
                                                            int nblocks, int K,
                                                            float* __restrict__ gW1,
                                                            float* __restrict__ gb1,
                                                            float* __restrict__ gW2,
                                                            float* __restrict__ gb2) {
  const int P = din_partial_floats(K);
  const int KH = K * kH;
  const int total = KH + 2 * kH + 1;
  for (int q = blockIdx.x * kBlock + threadIdx.x; q < total; q += gridDim.x * kBlock) {
    if (q < KH) {
      float a = 0.f, bsum = 0.f, d = 0.f;
      for (int blk = 0; blk < nblocks; ++blk) {
        const float* p = partial + static_cast<int64_t>(blk) * P;
        a += p[q];
        bsum += p[KH + q];
        d += p[2 * KH + q];
      }
      gW1[q] = a;
      gW1[KH + q] = bsum;
      gW1[2 * KH + q] = a - bsum;
      gW1[3 * KH + q] = d;
    } else {
      const int r = q - KH;
      float s = 0.f;
      for (int blk = 0; blk < nblocks; ++blk)
        s += partial[static_cast<int64_t>(blk) * P + 3 * KH + r];
      if (r < kH) gb1[r] = s;
      else if (r < 2 * kH) gW2[r - kH] = s;
      else gb2[0] = s;
    }
  }
}

#ifndef LR_DIN_GRID_MULT
#define LR_DIN_GRID_MULT 2     // workgroups per CU the attention kernels' persistent grids are capped at (profiling switch)
#endif
static inline int din_grid(int64_t B) { return grid_for(B, kBlock / kWave, kNumCU * LR_DIN_GRID_MULT); }

static inline size_t din_fwd_lds(int K, int L) { return (size_t(4) * K * kH + 4 * size_t(L)) * 4; }
static inline size_t din_bwd_lds(int K, int L) {
  const int NW = kBlock / kWave;
  return (size_t(4) * K * kH + size_t(NW) * 3 * K * kH + NW * (2 * kH + 4) + size_t(NW) * L) * 4;
}

template <typename Kern>
static int set_lds(Kern kern, size_t lds) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return LR_OK;
}

// ---- MFMA path (din_mfma.hpp): every K the shuffle kernels take, sequences up to 2048 keys -----------
static inline bool din_use_mfma(int K, int L) {
  return (K == 16 || K == 32 || K == 64 || K == 128) && L <= 2048;
}
static inline size_t din_mfma_fwd_lds(int K, int L) { return size_t(3) * (K / 16) * 64 * 16 + size_t(4) * L * 4; }
static inline size_t din_mfma_data_lds(int K, int L) {      // weight images, [4 waves][L] scores, [4 waves][2][K / 16][4] row pieces
  return size_t(6) * (K / 16) * 64 * 16 + size_t(4) * ((L + 3) & ~3) * 4 + size_t(4) * 2 * (K / 16) * 4 * 16;
}
static inline size_t din_mfma_param_lds(int K) {
  const size_t tile = size_t(4) * 16 * (K + 4) * 4, fold = size_t(3) * K * kDH * 4;
  return tile > fold ? tile : fold;
}
static inline size_t din_mfma_ws_floats(int64_t B, int L, int K) {
  const int grid = din_grid(B);
  return static_cast<size_t>(B) * L * kDH + static_cast<size_t>(B) * kDH + static_cast<size_t>(grid) * 4 * kDinSmall +
         static_cast<size_t>(grid) * 3 * K * kDH;
}

template <bool GATHER>
static int din_fwd_dispatch(const float* qsrc, const float* ksrc, int64_t V, int K,
                            const int32_t* item, const int32_t* seq, const int32_t* len, int64_t B,
                            int L, const float* W1, const float* b1, const float* W2,
                            const float* b2, float* out, float* attn, hipStream_t s, float* hid = nullptr,
                            int32_t* order_out = nullptr) {
  const int grid = din_grid(B);
  if ((hid != nullptr || order_out != nullptr) && !din_use_mfma(K, L)) return LR_ESHAPE;      // MFMA kernels only
  if (din_use_mfma(K, L)) {
    const size_t lds_m = din_mfma_fwd_lds(K, L);
#define LR_DINFM(NT)                                                                          \
  {                                                                                           \
    auto kern = din_fwd_mfma_kernel<NT, GATHER>;                                              \
    int rc = set_lds(kern, lds_m);                                                            \
    if (rc != LR_OK) return rc;                                                               \
    hipLaunchKernelGGL(kern, dim3(order_out != nullptr && grid < 2 ? 2 : grid), dim3(kBlock), lds_m, s, qsrc, ksrc, V, item, seq, \
                       len, B, L, W1, b1, W2, b2, out, attn, hid, order_out);                 \
    return launch_status();                                                                   \
  }
    if (K == 16) LR_DINFM(1)
    if (K == 32) LR_DINFM(2)
    if (K == 64) LR_DINFM(4)
    if (K == 128) LR_DINFM(8)
#undef LR_DINFM
  }
  const size_t lds = din_fwd_lds(K, L);
  if (lds > 160 * 1024) return LR_ESHAPE;
#define LR_DINF(LPR)                                                                         \
  {                                                                                          \
    auto kern = din_fwd_kernel<LPR, GATHER>;                                                 \
    int rc = set_lds(kern, lds);                                                             \
    if (rc != LR_OK) return rc;                                                              \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, s, qsrc, ksrc, V, item, seq, len, \
                       B, L, W1, b1, W2, b2, out, attn);                                     \
    return launch_status();                                                                  \
  }
  if (K == 16) LR_DINF(4)
  if (K == 32) LR_DINF(8)
  if (K == 64) LR_DINF(16)
  if (K == 128) LR_DINF(32)
#undef LR_DINF
  return LR_ESHAPE;
}

template <bool GATHER>
static int din_bwd_dispatch(const float* qsrc, const float* ksrc, int64_t V, int K,
                            const int32_t* item, const int32_t* seq, const int32_t* len, int64_t B,
                            int L, const float* W1, const float* b1, const float* W2,
                            const float* b2, const float* attn, const float* gout, float* gq,
                            float* gkey, float* gW1, float* gb1, float* gW2, float* gb2, void* ws,
                            size_t ws_bytes, hipStream_t s, int parts = 3, int keep_pad_rows = 0,
                            const float* hid = nullptr, const int32_t* order = nullptr) {
  const int grid = din_grid(B);
  if ((hid != nullptr || order != nullptr) && !din_use_mfma(K, L)) return LR_ESHAPE;
  if (din_use_mfma(K, L)) {
    if (ws == nullptr || ws_bytes < din_mfma_ws_floats(B, L, K) * 4) return LR_EWORKSPACE;
    float* dzbuf = static_cast<float*>(ws);
    float* Dzbuf = dzbuf + static_cast<size_t>(B) * L * kDH;
    float* small = Dzbuf + static_cast<size_t>(B) * kDH;
    float* partial_m = small + static_cast<size_t>(grid) * 4 * kDinSmall;
    const size_t lds_d = din_mfma_data_lds(K, L), lds_p = din_mfma_param_lds(K);
#define LR_DINBM(NT)                                                                            \
  {                                                                                             \
    auto kd = din_bwd_data_kernel<NT, GATHER, false>;                                           \
    auto kh = din_bwd_data_kernel<NT, GATHER, true>;                                            \
    auto kp = din_bwd_param_kernel<NT, GATHER>;                                                 \
    int rc = set_lds(kd, lds_d);                                                                \
    if (rc != LR_OK) return rc;                                                                 \
    rc = set_lds(kh, lds_d);                                                                    \
    if (rc != LR_OK) return rc;                                                                 \
    rc = set_lds(kp, lds_p);                                                                    \
    if (rc != LR_OK) return rc;                                                                 \
    if ((parts & 1) && hid == nullptr)                                                          \
      hipLaunchKernelGGL(kd, dim3(grid), dim3(kBlock), lds_d, s, qsrc, ksrc, V, item, seq, len, \
                         B, L, W1, b1, W2, attn, gout, gq, gkey, dzbuf, Dzbuf, small,           \
                         keep_pad_rows, hid, order);                                            \
    if ((parts & 1) && hid != nullptr)                                                          \
      hipLaunchKernelGGL(kh, dim3(grid), dim3(kBlock), lds_d, s, qsrc, ksrc, V, item, seq, len, \
                         B, L, W1, b1, W2, attn, gout, gq, gkey, dzbuf, Dzbuf, small,           \
                         keep_pad_rows, hid, order);                                            \
    if (parts & 2)                                                                              \
      hipLaunchKernelGGL(kp, dim3(grid), dim3(kBlock), lds_p, s, qsrc, ksrc, V, item, seq, len, \
                         B, L, dzbuf, Dzbuf, partial_m, order);                                 \
    break;                                                                                      \
  }
    switch (K) {
      case 16: LR_DINBM(1)
      case 32: LR_DINBM(2)
      case 64: LR_DINBM(4)
      default: LR_DINBM(8)
    }
#undef LR_DINBM
    int rcm = launch_status();
    if (rcm != LR_OK || !(parts & 2)) return rcm;
    hipLaunchKernelGGL(din_reduce2_kernel, dim3((K * kDH + 2 * kDH + 1 + 15) / 16), dim3(16 * kDinRedSlices), 0, s,
                       partial_m, grid, small, grid * 4, K, gW1, gb1, gW2, gb2);
    return launch_status();
  }
  if (!(parts & 1)) return LR_OK;     // the shuffle kernel computes everything in the data part
  const size_t lds = din_bwd_lds(K, L);
  if (lds > 160 * 1024) return LR_ESHAPE;
  const size_t need = static_cast<size_t>(grid) * din_partial_floats(K) * 4;
  if (ws == nullptr || ws_bytes < need) return LR_EWORKSPACE;
  float* partial = static_cast<float*>(ws);
#define LR_DINB(LPR)                                                                          \
  {                                                                                           \
    auto kern = din_bwd_kernel<LPR, GATHER>;                                                  \
    int rc = set_lds(kern, lds);                                                              \
    if (rc != LR_OK) return rc;                                                               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, s, qsrc, ksrc, V, item, seq, len,  \
                       B, L, W1, b1, W2, b2, attn, gout, gq, gkey, partial);                  \
    break;                                                                                    \
  }
  switch (K) {
    case 16: LR_DINB(4)
    case 32: LR_DINB(8)
    case 64: LR_DINB(16)
    case 128: LR_DINB(32)
    default: return LR_ESHAPE;
  }
#undef LR_DINB
  int rc = launch_status();
  if (rc != LR_OK) return rc;
  hipLaunchKernelGGL(din_reduce_kernel, dim3(grid_for(K * kH + 2 * kH + 1, kBlock)), dim3(kBlock),
                     0, s, partial, grid, K, gW1, gb1, gW2, gb2);
  return launch_status();
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_din_attn_ws_bytes(int64_t B, int L, int K, int H) {
  if (B < 0 || H != kH || K < 1 || L < 1) return 0;
  const size_t shuffle = static_cast<size_t>(din_grid(B)) * din_partial_floats(K) * 4;
  const size_t mfma = din_use_mfma(K, L) ? din_mfma_ws_floats(B, L, K) * 4 : 0;
  return shuffle > mfma ? shuffle : mfma;
}

#define LR_DIN_COMMON_CHECK()                                                          \
  LR_CHECK_ARG(B >= 0 && L >= 1 && K >= 1);                                            \
  if (H != kH) return LR_ESHAPE;                                                       \
  if (B == 0) return LR_OK;                                                            \
  LR_CHECK_ARG(len && W1 && b1 && W2 && b2);

extern "C" int lr_din_attn_pool_fwd_f32(const float* item_table, int64_t V, int K,
                                        const int32_t* item, const int32_t* seq,
                                        const int32_t* len, int64_t B, int L, const float* W1,
                                        const float* b1, const float* W2, const float* b2, int H,
                                        float* out, float* attn, float* hid, int32_t* order_out,
                                        lr_stream_t stream) {
  LR_DIN_COMMON_CHECK();
  LR_CHECK_ARG(item_table && item && seq && out && attn && V >= 0);
  LR_CHECK_ARG(hid == nullptr || reinterpret_cast<uintptr_t>(hid) % 16 == 0);
  return din_fwd_dispatch<true>(item_table, item_table, V, K, item, seq, len, B, L, W1, b1, W2, b2,
                                out, attn, as_stream(stream), hid, order_out);
}

extern "C" int lr_din_attn_dense_fwd_f32(const float* q, const float* keys, int K,
                                         const int32_t* len, int64_t B, int L, const float* W1,
                                         const float* b1, const float* W2, const float* b2, int H,
                                         float* out, float* attn, lr_stream_t stream) {
  LR_DIN_COMMON_CHECK();
  LR_CHECK_ARG(q && keys && out && attn);
  return din_fwd_dispatch<false>(q, keys, 0, K, nullptr, nullptr, len, B, L, W1, b1, W2, b2, out,
                                 attn, as_stream(stream));
}

extern "C" int lr_din_attn_pool_bwd_f32(const float* item_table, int64_t V, int K,
                                        const int32_t* item, const int32_t* seq,
                                        const int32_t* len, int64_t B, int L, const float* W1,
                                        const float* b1, const float* W2, const float* b2, int H,
                                        const float* attn, const float* gout, float* gq,
                                        float* gkey, float* gW1, float* gb1, float* gW2,
                                        float* gb2, void* ws, size_t ws_bytes,
                                        lr_stream_t stream) {
  LR_DIN_COMMON_CHECK();
  LR_CHECK_ARG(item_table && item && seq && attn && gout && gq && gkey && gW1 && gb1 && gW2 &&
               gb2 && V >= 0);
  return din_bwd_dispatch<true>(item_table, item_table, V, K, item, seq, len, B, L, W1, b1, W2, b2,
                                attn, gout, gq, gkey, gW1, gb1, gW2, gb2, ws, ws_bytes,
                                as_stream(stream));
}

extern "C" int lr_din_attn_pool_bwd_parts_f32(const float* item_table, int64_t V, int K,
                                              const int32_t* item, const int32_t* seq,
                                              const int32_t* len, int64_t B, int L, const float* W1,
                                              const float* b1, const float* W2, const float* b2, int H,
                                              const float* attn, const float* gout, float* gq,
                                              float* gkey, float* gW1, float* gb1, float* gW2,
                                              float* gb2, void* ws, size_t ws_bytes, int parts,
                                              int keep_pad_rows, const float* hid, const int32_t* order,
                                              lr_stream_t stream) {
  LR_DIN_COMMON_CHECK();
  LR_CHECK_ARG(parts >= 1 && parts <= 3);
  LR_CHECK_ARG(hid == nullptr || reinterpret_cast<uintptr_t>(hid) % 16 == 0);
  LR_CHECK_ARG(item_table && item && seq && attn && gout && gq && gkey && gW1 && gb1 && gW2 &&
               gb2 && V >= 0);
  return din_bwd_dispatch<true>(item_table, item_table, V, K, item, seq, len, B, L, W1, b1, W2, b2,
                                attn, gout, gq, gkey, gW1, gb1, gW2, gb2, ws, ws_bytes,
                                as_stream(stream), parts, keep_pad_rows, hid, order);
}

extern "C" int lr_din_attn_dense_bwd_f32(const float* q, const float* keys, int K,
                                         const int32_t* len, int64_t B, int L, const float* W1,
                                         const float* b1, const float* W2, const float* b2, int H,
                                         const float* attn, const float* gout, float* gq,
                                         float* gkey, float* gW1, float* gb1, float* gW2,
                                         float* gb2, void* ws, size_t ws_bytes,
                                         lr_stream_t stream) {
  LR_DIN_COMMON_CHECK();
  LR_CHECK_ARG(q && keys && attn && gout && gq && gkey && gW1 && gb1 && gW2 && gb2);
  return din_bwd_dispatch<false>(q, keys, 0, K, nullptr, nullptr, len, B, L, W1, b1, W2, b2, attn,
                                 gout, gq, gkey, gW1, gb1, gW2, gb2, ws, ws_bytes,
                                 as_stream(stream));
}

// ---- the id stream of the fused DIN step in ONE launch (was seven elementwise torch kernels inside the captured step) ----------
// ids[n_pos], plane-major: [Fp field planes of B rows | the attention-output plane: -1 | query rows (items) | window rows, pads -1]
//   plane 0 = users + user_off, plane 1 = items + item_off, plane 2 + p = sparse[b][cols[p]] + sparse_off   (tfops/features.py:6-44)
//   window: seqs[b][l] + item_off for l < lens[b]                                                       (sequence.py:56-58)
namespace lr {
__device__ __forceinline__ void din_ids_body(const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                             const int32_t* __restrict__ sparse, int sparse_ld,
                                             const int32_t* __restrict__ cols, int n_plain,
                                             const int32_t* __restrict__ seqs, const int32_t* __restrict__ lens,
                                             int64_t B, int L, int32_t user_off, int32_t item_off,
                                             int32_t sparse_off, int32_t* __restrict__ ids, int64_t nblocks) {
  const int Fp = 2 + n_plain;
  const int64_t n_field = static_cast<int64_t>(Fp) * B, n0 = n_field + B, total = n0 + B + B * L;
  const int64_t stride = nblocks * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    int32_t v;
    if (q < n_field) {
      const int p = static_cast<int>(q / B);
      const int64_t b = q - static_cast<int64_t>(p) * B;
      if (p == 0) v = users[b] + user_off;
      else if (p == 1) v = items[b] + item_off;
      else v = sparse[b * sparse_ld + (cols != nullptr ? cols[p - 2] : p - 2)] + sparse_off;
    } else if (q < n0) {
      v = -1;
    } else if (q < n0 + B) {
      v = items[q - n0] + item_off;
    } else {
      const int64_t w = q - n0 - B, b = w / L;
      const int l = static_cast<int>(w - b * L);
      v = l < lens[b] ? seqs[w] + item_off : -1;
    }
    ids[q] = v;
  }
}
__global__ __launch_bounds__(kBlock) void din_ids_kernel(const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                                         const int32_t* __restrict__ sparse, int sparse_ld,
                                                         const int32_t* __restrict__ cols, int n_plain,
                                                         const int32_t* __restrict__ seqs, const int32_t* __restrict__ lens,
                                                         int64_t B, int L, int32_t user_off, int32_t item_off,
                                                         int32_t sparse_off, int32_t* __restrict__ ids) {
  din_ids_body(users, items, sparse, sparse_ld, cols, n_plain, seqs, lens, B, L, user_off, item_off, sparse_off, ids,
               static_cast<int64_t>(gridDim.x));
}
}  // namespace lr

extern "C" int lr_din_build_ids_i32(const int32_t* users, const int32_t* items, const int32_t* sparse, int sparse_ld,
                                    const int32_t* cols, int n_plain, const int32_t* seqs, const int32_t* lens, int64_t B,
                                    int L, int32_t user_off, int32_t item_off, int32_t sparse_off, int32_t* ids,
                                    lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && L >= 0 && n_plain >= 0);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(users && items && ids && (L == 0 || (seqs && lens)) && (n_plain == 0 || (sparse && sparse_ld >= 1)));
  const int64_t total = (static_cast<int64_t>(2 + n_plain) + 2 + L) * B;
  hipLaunchKernelGGL(lr::din_ids_kernel, dim3(lr::grid_for(total, lr::kBlock)), dim3(lr::kBlock), 0, lr::as_stream(stream),
                     users, items, sparse, sparse_ld, cols, n_plain, seqs, lens, B, L, user_off, item_off, sparse_off, ids);
  return lr::launch_status();
}

#ifdef LR_DIN_MARKS
extern "C" int lr_din_debug_marks(unsigned long long* out64) {
  return static_cast<int>(hipMemcpyFromSymbol(out64, HIP_SYMBOL(lr::lr_din_marks), 64 * sizeof(unsigned long long)));
}
#endif
