// FM pairwise interaction: stand-alone fwd/bwd on a materialised e[B,F,K], and the fused
// gather + interaction forward / interaction-backward + segment-sum + Adam backward that the
// FM / DeepFM training step uses.
//
// Forward mapping: ONE WAVEFRONT PER SAMPLE.  A row group of LPR = K/4 lanes reads one field's
// row (16 B / lane); the 64/LPR groups of the wave read different fields concurrently, UNR
// deep, so 64/LPR*UNR row fetches are in flight per wave.  Sum and sum-of-squares accumulate
// in registers; the groups are combined with xor-shuffles (no LDS, no second pass over e).
#include "common.hpp"

namespace lr {

template <int LPR, bool GATHER>
__global__ __launch_bounds__(kBlock) void fm_fwd_kernel(
    const float* __restrict__ src, int64_t V, const int32_t* __restrict__ idx, int64_t B, int F,
    float* __restrict__ e, float* __restrict__ pair, float* __restrict__ fsum) {
  constexpr int K = LPR * 4;
  constexpr int SLOTS = kWave / LPR;
  constexpr int UNR = (SLOTS >= 4) ? 4 : 8;
  const int lane = threadIdx.x & (kWave - 1);
  const int slot = lane / LPR;
  const int c4 = (lane % LPR) * 4;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / kWave;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / kWave);
  for (int64_t b = wave; b < B; b += nwaves) {
    float4 S = f4_zero(), Q = f4_zero();
    const int64_t row0 = b * F;
    for (int f0 = 0; f0 < F; f0 += SLOTS * UNR) {
      float4 x[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int f = f0 + u * SLOTS + slot;
        x[u] = f4_zero();
        if (f < F) {
          if constexpr (GATHER) {
            const int32_t id = idx[row0 + f];
            if (id >= 0 && id < V) x[u] = ld4(src + static_cast<int64_t>(id) * K + c4);
          } else {
            x[u] = ld4(src + (row0 + f) * K + c4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int f = f0 + u * SLOTS + slot;
        S = f4_add(S, x[u]);
        Q = f4_fma(x[u], x[u], Q);
        if constexpr (GATHER) {
          if (e != nullptr && f < F) st4_nt(e + (row0 + f) * K + c4, x[u]);
        }
      }
    }
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {
      S = f4_add(S, f4_shfl_xor(S, o));
      Q = f4_add(Q, f4_shfl_xor(Q, o));
    }
    if (slot == 0) {
      float4 p;
      p.x = 0.5f * (S.x * S.x - Q.x);
      p.y = 0.5f * (S.y * S.y - Q.y);
      p.z = 0.5f * (S.z * S.z - Q.z);
      p.w = 0.5f * (S.w * S.w - Q.w);
      st4(pair + b * K + c4, p);
      if (fsum != nullptr) st4(fsum + b * K + c4, S);
    }
  }
}

// generic-K stand-alone forward: one thread per (b,k)
__global__ __launch_bounds__(kBlock) void fm_fwd_scalar_kernel(const float* __restrict__ e,
                                                               int64_t B, int F, int K,
                                                               float* __restrict__ pair,
                                                               float* __restrict__ fsum) {
  const int64_t total = B * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += stride) {
    const int64_t b = t / K;
    const int k = static_cast<int>(t - b * K);
    float S = 0.f, Q = 0.f;
    for (int f = 0; f < F; ++f) {
      const float x = e[(b * F + f) * K + k];
      S += x;
      Q = fmaf(x, x, Q);
    }
    pair[t] = 0.5f * (S * S - Q);
    if (fsum) fsum[t] = S;
  }
}

__global__ __launch_bounds__(kBlock) void fm_bwd_kernel(const float* __restrict__ e,
                                                        const float* __restrict__ fsum,
                                                        const float* __restrict__ gpair,
                                                        int64_t B, int F, int K,
                                                        float* __restrict__ ge, int accumulate) {
  const int64_t total = B * F * K;
  const int64_t FK = static_cast<int64_t>(F) * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += stride) {
    const int64_t b = t / FK;
    const int k = static_cast<int>(t % K);
    const float g = gpair[b * K + k] * (fsum[b * K + k] - e[t]);
    ge[t] = accumulate ? ge[t] + g : g;
  }
}

// Fused backward + Adam.  One row group per distinct row r; positions q = b*F + f.
template <int LPR>
__global__ __launch_bounds__(kBlock) void fm_bwd_adam_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ gdeep, const float* __restrict__ gpair,
    const float* __restrict__ fsum, int F, const int32_t* __restrict__ seg_pos,
    const int32_t* __restrict__ seg_rows, const int32_t* __restrict__ seg_start,
    const int32_t* __restrict__ n_seg_ptr, AdamCoef coef) {
  constexpr int K = LPR * 4;
  const int n_seg = *n_seg_ptr;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int c4 = static_cast<int>(gtid % LPR) * 4;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t s = gtid / LPR; s < n_seg; s += ngroups) {
    const int p0 = seg_start[s], p1 = seg_start[s + 1];
    float4 gd = f4_zero();   // sum gdeep[q]
    float4 gps = f4_zero();  // sum gpair[b] * fsum[b]
    float4 gp = f4_zero();   // sum gpair[b]
    for (int p = p0; p < p1; ++p) {
      const int32_t q = seg_pos[p];
      const int64_t b = q / F;
      if (gdeep != nullptr) gd = f4_add(gd, ld4(gdeep + static_cast<int64_t>(q) * K + c4));
      const float4 a = ld4(gpair + b * K + c4);
      gps = f4_fma(a, ld4(fsum + b * K + c4), gps);
      gp = f4_add(gp, a);
    }
    const int64_t off = static_cast<int64_t>(seg_rows[s]) * K + c4;
    const float4 w = ld4(table + off);
    float4 g;  // gd + gps - w*gp
    g.x = gd.x + (gps.x - w.x * gp.x);
    g.y = gd.y + (gps.y - w.y * gp.y);
    g.z = gd.z + (gps.z - w.z * gp.z);
    g.w = gd.w + (gps.w - w.w * gp.w);
    float4 mm = ld4(m + off), vv = ld4(v + off);
    st4(table + off, adam_vec(w, g, mm, vv, coef));
    st4(m + off, mm);
    st4(v + off, vv);
  }
}

template <int LPR, bool GATHER>
static int launch_fm_fwd(const float* src, int64_t V, const int32_t* idx, int64_t B, int F,
                         float* e, float* pair, float* fsum, hipStream_t s) {
  const int grid = grid_for(B, kBlock / kWave, kNumCU * 8);
  hipLaunchKernelGGL((fm_fwd_kernel<LPR, GATHER>), dim3(grid), dim3(kBlock), 0, s, src, V, idx,
                     B, F, e, pair, fsum);
  return launch_status();
}

template <bool GATHER>
static int dispatch_fm_fwd(const float* src, int64_t V, int K, const int32_t* idx, int64_t B,
                           int F, float* e, float* pair, float* fsum, hipStream_t s) {
  switch (K) {
    case 16: return launch_fm_fwd<4, GATHER>(src, V, idx, B, F, e, pair, fsum, s);
    case 32: return launch_fm_fwd<8, GATHER>(src, V, idx, B, F, e, pair, fsum, s);
    case 64: return launch_fm_fwd<16, GATHER>(src, V, idx, B, F, e, pair, fsum, s);
    case 128: return launch_fm_fwd<32, GATHER>(src, V, idx, B, F, e, pair, fsum, s);
    case 256: return launch_fm_fwd<64, GATHER>(src, V, idx, B, F, e, pair, fsum, s);
    default: return LR_ESHAPE;
  }
}

static inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace lr

using namespace lr;

extern "C" int lr_fm_pairwise_fwd_f32(const float* e, int64_t B, int F, int K, float* pair,
                                      float* fsum, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(e && pair);
  hipStream_t s = as_stream(stream);
  if (al16(e) && al16(pair) && (!fsum || al16(fsum))) {
    int rc = dispatch_fm_fwd<false>(e, 0, K, nullptr, B, F, nullptr, pair, fsum, s);
    if (rc != LR_ESHAPE) return rc;
  }
  hipLaunchKernelGGL(fm_fwd_scalar_kernel, dim3(grid_for(B * K, kBlock)), dim3(kBlock), 0, s, e,
                     B, F, K, pair, fsum);
  return launch_status();
}

extern "C" int lr_fm_pairwise_bwd_f32(const float* e, const float* fsum, const float* gpair,
                                      int64_t B, int F, int K, float* ge, int accumulate,
                                      lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(e && fsum && gpair && ge);
  hipLaunchKernelGGL(fm_bwd_kernel, dim3(grid_for(B * F * K, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), e, fsum, gpair, B, F, K, ge, accumulate);
  return launch_status();
}

extern "C" int lr_fm_embed_fwd_f32(const float* table, int64_t V, int K, const int32_t* idx,
                                   int64_t B, int F, float* e, float* pair, float* fsum,
                                   lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && pair);
  LR_CHECK_ARG(al16(table) && al16(pair) && (!e || al16(e)) && (!fsum || al16(fsum)));
  return dispatch_fm_fwd<true>(table, V, K, idx, B, F, e, pair, fsum, as_stream(stream));
}

extern "C" int lr_fm_embed_bwd_adam_f32(float* table, float* m, float* v, int64_t V, int K,
                                        const float* gdeep, const float* gpair,
                                        const float* fsum, int64_t B, int F,
                                        const int32_t* seg_pos, const int32_t* seg_rows,
                                        const int32_t* seg_start, const int32_t* n_seg,
                                        lr_adam_hp hp, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && B >= 0 && F >= 1 && K >= 1 && hp.step >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && gpair && fsum && seg_pos && seg_rows && seg_start && n_seg);
  LR_CHECK_ARG(al16(table) && al16(m) && al16(v) && al16(gpair) && al16(fsum) &&
               (!gdeep || al16(gdeep)));
  hipStream_t s = as_stream(stream);
  const AdamCoef coef = make_adam_coef(hp);
  const int64_t n_max = B * F;
#define LR_FMB(LPR)                                                                          \
  {                                                                                          \
    const int grid = grid_for(n_max, kBlock / LPR);                                          \
    hipLaunchKernelGGL((fm_bwd_adam_kernel<LPR>), dim3(grid), dim3(kBlock), 0, s, table, m, v, \
                       gdeep, gpair, fsum, F, seg_pos, seg_rows, seg_start, n_seg, coef);    \
    return launch_status();                                                                  \
  }
  if (K == 16) LR_FMB(4)
  if (K == 32) LR_FMB(8)
  if (K == 64) LR_FMB(16)
  if (K == 128) LR_FMB(32)
#undef LR_FMB
  return LR_ESHAPE;
}
