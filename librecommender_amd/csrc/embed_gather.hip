// Row gather, fixed-length bag pooling and pointwise dot — HBM-bound row traffic.
//
// Layout: a table row is K fp32 = K*4 contiguous bytes.  A "row group" of LPR = K/4 lanes
// reads one row with one 16-byte load per lane, so a 64-lane wavefront keeps 64/LPR rows in
// flight per load instruction (K=64: 4 rows of 256 B; K=128: 2 rows of 512 B) and every
// fetched 64-B sector is fully used.  Each group additionally unrolls UNR independent rows so
// a wave has >= 8-16 row fetches outstanding (HBM-miss latency ~900 cycles, guide).
#include "common.hpp"

namespace lr {

// ---------------------------------------------------------------------------------------
// gather
// ---------------------------------------------------------------------------------------
template <int LPR, int UNR>
__global__ __launch_bounds__(kBlock) void embed_gather_vec_kernel(
    const float* __restrict__ table, int64_t V, const int32_t* __restrict__ idx, int64_t n,
    float* __restrict__ out) {
  constexpr int K = LPR * 4;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t group = gtid / LPR;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t base = group; base < n; base += ngroups * UNR) {
    float4 v[UNR];
    int32_t id[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = base + u * ngroups;
      id[u] = (r < n) ? idx[r] : -1;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = id[u] >= 0 && id[u] < V;
      v[u] = ok ? ld4(table + static_cast<int64_t>(id[u]) * K + lane * 4) : f4_zero();
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = base + u * ngroups;
      if (r < n) st4_nt(out + r * K + lane * 4, v[u]);
    }
  }
}

// any K (K=1 linear tables, odd sizes): one thread per output element.
__global__ __launch_bounds__(kBlock) void embed_gather_scalar_kernel(
    const float* __restrict__ table, int64_t V, int K, const int32_t* __restrict__ idx,
    int64_t n, float* __restrict__ out) {
  const int64_t total = n * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t r = e / K;
    const int c = static_cast<int>(e - r * K);
    const int32_t id = idx[r];
    out[e] = (id >= 0 && id < V) ? table[static_cast<int64_t>(id) * K + c] : 0.f;
  }
}

template <int LPR>
static int launch_gather_vec(const float* table, int64_t V, const int32_t* idx, int64_t n,
                             float* out, hipStream_t s) {
  constexpr int UNR = 4;
  const int groups_per_block = kBlock / LPR;
  const int grid = grid_for(n, groups_per_block * UNR);
  hipLaunchKernelGGL((embed_gather_vec_kernel<LPR, UNR>), dim3(grid), dim3(kBlock), 0, s,
                     table, V, idx, n, out);
  return launch_status();
}

// ---------------------------------------------------------------------------------------
// bag pooling (fixed bag_len, OOV -> 0)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float pool_scale(int combiner, int cnt) {
  if (combiner == LR_COMBINER_SUM) return 1.f;
  if (cnt <= 0) return 0.f;  // tf.div_no_nan
  return combiner == LR_COMBINER_MEAN ? 1.f / static_cast<float>(cnt)
                                      : 1.f / sqrtf(static_cast<float>(cnt));
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void bag_pool_vec_kernel(
    const float* __restrict__ table, int64_t V, const int32_t* __restrict__ idx,
    int64_t nbags, int bag_len, int combiner, int32_t oov, float* __restrict__ out) {
  constexpr int K = LPR * 4;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t b = gtid / LPR; b < nbags; b += ngroups) {
    float4 acc = f4_zero();
    int cnt = 0;
    const int32_t* ids = idx + b * bag_len;
    for (int j = 0; j < bag_len; ++j) {
      const int32_t id = ids[j];
      if (id != oov && id >= 0 && id < V) {
        acc = f4_add(acc, ld4(table + static_cast<int64_t>(id) * K + lane * 4));
        ++cnt;
      }
    }
    st4(out + b * K + lane * 4, f4_scale(acc, pool_scale(combiner, cnt)));
  }
}

__global__ __launch_bounds__(kBlock) void bag_pool_scalar_kernel(
    const float* __restrict__ table, int64_t V, int K, const int32_t* __restrict__ idx,
    int64_t nbags, int bag_len, int combiner, int32_t oov, float* __restrict__ out) {
  const int64_t total = nbags * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t b = e / K;
    const int c = static_cast<int>(e - b * K);
    float acc = 0.f;
    int cnt = 0;
    for (int j = 0; j < bag_len; ++j) {
      const int32_t id = idx[b * bag_len + j];
      if (id != oov && id >= 0 && id < V) {
        acc += table[static_cast<int64_t>(id) * K + c];
        ++cnt;
      }
    }
    out[e] = acc * pool_scale(combiner, cnt);
  }
}

__global__ __launch_bounds__(kBlock) void bag_pool_bwd_kernel(
    const float* __restrict__ gout, int K, const int32_t* __restrict__ idx, int64_t V,
    int64_t nbags, int bag_len, int combiner, int32_t oov, float* __restrict__ gentry) {
  // one thread per (entry, column); the per-bag count is recomputed (bag_len is tiny).
  const int64_t total = nbags * bag_len * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t entry = e / K;
    const int c = static_cast<int>(e - entry * K);
    const int64_t b = entry / bag_len;
    int cnt = 0;
    for (int j = 0; j < bag_len; ++j) {
      const int32_t id = idx[b * bag_len + j];
      cnt += (id != oov && id >= 0 && id < V) ? 1 : 0;
    }
    const int32_t id = idx[entry];
    const bool live = id != oov && id >= 0 && id < V;
    gentry[e] = live ? gout[b * K + c] * pool_scale(combiner, cnt) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// pointwise <U[user], I[item]>: one 16-lane group per pair (D % 4 == 0) else scalar loop.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pair_dot_kernel(
    const float* __restrict__ U, int64_t nU, const float* __restrict__ I, int64_t nI, int D,
    const int32_t* __restrict__ user, const int32_t* __restrict__ item, int64_t n,
    float* __restrict__ out) {
  constexpr int G = 16;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % G);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / G;
  const int64_t nround = ceil_div(n, ngroups) * ngroups;  // keep groups converged for shfl
  for (int64_t i = gtid / G; i < nround; i += ngroups) {
    float acc = 0.f;
    if (i < n) {
      const int32_t u = user[i], it = item[i];
      if (u >= 0 && u < nU && it >= 0 && it < nI) {
        const float* pu = U + static_cast<int64_t>(u) * D;
        const float* pi = I + static_cast<int64_t>(it) * D;
        for (int c = lane; c < D; c += G) acc = fmaf(pu[c], pi[c], acc);
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (i < n && lane == 0) out[i] = acc;
  }
}

}  // namespace lr

using namespace lr;

extern "C" int lr_embed_gather_f32(const float* table, int64_t V, int K, const int32_t* idx,
                                   int64_t n, float* out, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && K >= 1 && n >= 0);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && out);
  hipStream_t s = as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  if (aligned) {
    switch (K) {
      case 16: return launch_gather_vec<4>(table, V, idx, n, out, s);
      case 32: return launch_gather_vec<8>(table, V, idx, n, out, s);
      case 64: return launch_gather_vec<16>(table, V, idx, n, out, s);
      case 128: return launch_gather_vec<32>(table, V, idx, n, out, s);
      case 256: return launch_gather_vec<64>(table, V, idx, n, out, s);
      default: break;
    }
  }
  const int grid = grid_for(n * K, kBlock);
  hipLaunchKernelGGL(embed_gather_scalar_kernel, dim3(grid), dim3(kBlock), 0, s, table, V, K,
                     idx, n, out);
  return launch_status();
}

extern "C" int lr_embed_bag_pool_f32(const float* table, int64_t V, int K, const int32_t* idx,
                                     int64_t nbags, int bag_len, int combiner, int32_t oov,
                                     float* out, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && K >= 1 && nbags >= 0 && bag_len >= 1);
  LR_CHECK_ARG(combiner >= LR_COMBINER_SUM && combiner <= LR_COMBINER_SQRTN);
  if (nbags == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && out);
  hipStream_t s = as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(out) % 16 == 0);
#define LR_BAG(LPR)                                                                        \
  {                                                                                        \
    const int grid = grid_for(nbags, kBlock / LPR);                                        \
    hipLaunchKernelGGL((bag_pool_vec_kernel<LPR>), dim3(grid), dim3(kBlock), 0, s, table, \
                       V, idx, nbags, bag_len, combiner, oov, out);                        \
    return launch_status();                                                                \
  }
  if (aligned) {
    if (K == 16) LR_BAG(4)
    if (K == 32) LR_BAG(8)
    if (K == 64) LR_BAG(16)
    if (K == 128) LR_BAG(32)
  }
#undef LR_BAG
  const int grid = grid_for(nbags * K, kBlock);
  hipLaunchKernelGGL(bag_pool_scalar_kernel, dim3(grid), dim3(kBlock), 0, s, table, V, K, idx,
                     nbags, bag_len, combiner, oov, out);
  return launch_status();
}

extern "C" int lr_embed_bag_pool_bwd_f32(const float* gout, int K, const int32_t* idx,
                                         int64_t V, int64_t nbags, int bag_len, int combiner,
                                         int32_t oov, float* gentry, lr_stream_t stream) {
  LR_CHECK_ARG(K >= 1 && nbags >= 0 && bag_len >= 1);
  LR_CHECK_ARG(combiner >= LR_COMBINER_SUM && combiner <= LR_COMBINER_SQRTN);
  if (nbags == 0) return LR_OK;
  LR_CHECK_ARG(gout && idx && gentry);
  const int grid = grid_for(nbags * bag_len * K, kBlock);
  hipLaunchKernelGGL(bag_pool_bwd_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), gout,
                     K, idx, V, nbags, bag_len, combiner, oov, gentry);
  return launch_status();
}

extern "C" int lr_pair_dot_f32(const float* U, int64_t nU, const float* I, int64_t nI, int D,
                               const int32_t* user, const int32_t* item, int64_t n, float* out,
                               lr_stream_t stream) {
  LR_CHECK_ARG(D >= 1 && n >= 0 && nU >= 0 && nI >= 0);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(U && I && user && item && out);
  const int grid = grid_for(n, kBlock / 16);
  hipLaunchKernelGGL(pair_dot_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), U, nU, I,
                     nI, D, user, item, n, out);
  return launch_status();
}
