// DIN attention pooling on the f32 MFMA pipe (v_mfma_f32_16x16x4_f32) — layers/attention.py:28-64.
//
// The attention MLP's first layer over concat([q, k, q-k, q*k]) (4K -> 16) splits into products with
// weights shared by every (sample, key) pair:
//   z[key][j] = b1[j] + sum_d (W1a+W1c)[d][j] q[d]              "zq": once per sample
//             + sum_d (W1b-W1c)[d][j] k[d] + sum_d W1d[d][j] (q[d] k[d])
// so 16 keys x 16 hidden units is one MFMA tile with the reduction running over the embedding dims.
// Mapping: one wavefront per sample, keys in tiles of 16.  Lane (i = lane % 16, kq = lane / 16) holds dims
// {16 t + 4 kq + e} of key i of the tile (one 16-byte load per t), which is at the same time
//   * the B operand B[k = kq][n = key i] of MFMA step (t, e)  (A = the weights, lane = hidden unit), so the
//     product comes out TRANSPOSED: lane = key, registers = hidden units 4 kq + r.  The score of a key is
//     then 4 FMAs + two cross-group adds and lands on the very lanes that hold the key's values: the
//     softmax-weighted sum of the keys needs no data movement;
//   * the layout of the output tile of  d key = dz W^T  (A = weights with lane = dim, B = dz as it stands),
//     so the key gradient is combined and stored with 16-byte accesses.
// Softmax is online per lane (keys i, i+16, ...) and merged over the 16 lanes at the end.
// Backward = data kernel (key / query gradients, dz spilled to a [B*L,16] buffer: 26 MB at cfg 3) +
// parameter kernel (W1 gradient = keys^T dz over all B*L rows, per-workgroup partials, fixed-order
// reduction).  All reductions have a fixed order: results are run-to-run identical.
#pragma once
#include "common.hpp"

namespace lr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kDH = 16;   // hidden units of the attention MLP (the reference's fixed value)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float din_sigmoid(float z) { return __frcp_rn(1.f + __expf(-z)); }

// sum over the 16 lanes of a DPP row (lanes with equal lane / 16); every lane gets the total
__device__ __forceinline__ float row_sum16(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
  return x;
}
__device__ __forceinline__ float row_max16(float x) {
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false)));
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false)));
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false)));
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false)));
  return x;
}
__device__ __forceinline__ float group_sum4(float x) {   // over the 4 lane groups (same lane % 16)
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// ---- weight images in LDS ----------------------------------------------------------------------
// "F" image (forward-type operand, lane = hidden unit j): F[m][t][lane] = float4 over e of
//     Wm[d = 16 t + 4 (lane/16) + e][j = lane % 16]
// "T" image (transposed-type operand, lane = dim):         T[m][u][lane] = float4 over r of
//     Wm[d = 16 u + lane % 16][j = 4 (lane/16) + r]
// m: 0 = W1b - W1c, 1 = W1d, 2 = W1a + W1c   (rows of W1: [q | k | q-k | q*k] blocks of K, layers/attention.py:50-52)
__device__ __forceinline__ float din_wm(const float* __restrict__ W1, int K, int m, int d, int j) {
  if (m == 0) return W1[(K + d) * kDH + j] - W1[(2 * K + d) * kDH + j];
  if (m == 1) return W1[(3 * K + d) * kDH + j];
  return W1[d * kDH + j] + W1[(2 * K + d) * kDH + j];
}
template <int NT, bool WITH_T, bool WITH_F = true>
__device__ __forceinline__ void din_stage_weights(const float* __restrict__ W1, float4* __restrict__ F,
                                                  float4* __restrict__ T) {
  constexpr int K = 16 * NT;
  for (int q = threadIdx.x; q < 3 * NT * 64; q += kBlock) {
    const int l = q & 63, t = (q >> 6) % NT, m = q / (64 * NT);
    const int j = l & 15, d0 = 16 * t + 4 * (l >> 4);
    if (WITH_F)
      F[q] = make_float4(din_wm(W1, K, m, d0, j), din_wm(W1, K, m, d0 + 1, j), din_wm(W1, K, m, d0 + 2, j),
                         din_wm(W1, K, m, d0 + 3, j));
    if (WITH_T) {
      const int d = 16 * t + (l & 15), j0 = 4 * (l >> 4);
      T[q] = make_float4(din_wm(W1, K, m, d, j0), din_wm(W1, K, m, d, j0 + 1), din_wm(W1, K, m, d, j0 + 2),
                         din_wm(W1, K, m, d, j0 + 3));
    }
  }
}

// 16-byte piece [c, c+4) of row `pos` (GATHER: of table row ids[pos]; ids outside [0, V) read as zeros)
template <bool GATHER>
__device__ __forceinline__ const float* din_row_ptr(const float* __restrict__ src, int64_t V,
                                                    const int32_t* __restrict__ ids, int64_t pos, int K,
                                                    bool& ok) {
  if (GATHER) {
    const int32_t id = ids[pos];
    ok = ok && id >= 0 && id < V;
    return src + static_cast<int64_t>(ok ? id : 0) * K;
  }
  return src + pos * K;
}

// z tile (transposed: lane = key, regs = hidden 4 kq + r) from the lane's key pieces; weights from LDS
template <int NT>
__device__ __forceinline__ f32x4 din_z_tile(const float4* __restrict__ F, int lane, const float4 (&k4)[NT],
                                            const float4 (&q4)[NT]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 wb = F[(0 * NT + t) * 64 + lane], wd = F[(1 * NT + t) * 64 + lane];
    acc = mfma16(wb.x, k4[t].x, acc);
    acc = mfma16(wb.y, k4[t].y, acc);
    acc = mfma16(wb.z, k4[t].z, acc);
    acc = mfma16(wb.w, k4[t].w, acc);
    acc = mfma16(wd.x, k4[t].x * q4[t].x, acc);
    acc = mfma16(wd.y, k4[t].y * q4[t].y, acc);
    acc = mfma16(wd.z, k4[t].z * q4[t].z, acc);
    acc = mfma16(wd.w, k4[t].w * q4[t].w, acc);
  }
  return acc;
}
template <int NT>
__device__ __forceinline__ f32x4 din_zq(const float4* __restrict__ F, int lane, const float4 (&q4)[NT]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 wa = F[(2 * NT + t) * 64 + lane];
    acc = mfma16(wa.x, q4[t].x, acc);
    acc = mfma16(wa.y, q4[t].y, acc);
    acc = mfma16(wa.z, q4[t].z, acc);
    acc = mfma16(wa.w, q4[t].w, acc);
  }
  return acc;
}

// The samples in the order the BACKWARD attention kernels' waves take them: a STABLE partition by descending tile count
// (class = min(ceil(len / 16), 16)), so that wave w's samples w, w + nwaves, ... come one from each length class.  One workgroup;
// chunks of 256 samples in index order, per class a ballot prefix inside a wave + wave totals through LDS: deterministic.
constexpr int kOrdClasses = 16;
constexpr int kOrdSlots = 4096;                   // (class, 64-sample chunk) counters in LDS
__device__ __forceinline__ int din_len_class(int n, int L, int nclass) {
  n = n < 0 ? 0 : (n > L ? L : n);
  int tiles = n > 0 ? (n + 15) >> 4 : 1;
  if (tiles > nclass) tiles = nclass;
  return nclass - tiles;                          // 0 = the longest class
}
__device__ __forceinline__ void din_order_body(const int32_t* __restrict__ lens, int64_t B, int L, int32_t* __restrict__ order) {
  // counts per (class, chunk of 64 samples) by ballots, ONE exclusive scan over them in class-major order, then every wave
  // places its chunks' samples: two barriers in all (a first version walked 256-sample chunks with three barriers each and
  // cost more than the balance gave back)
  __shared__ int cnt[kOrdSlots];
  __shared__ int part[kBlock];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const int nchunks = static_cast<int>((B + kWave - 1) / kWave);
  int nclass = (L + 15) >> 4;
  if (nclass > kOrdClasses) nclass = kOrdClasses;
  if (nclass * nchunks > kOrdSlots) nclass = kOrdSlots / nchunks;        // very large batches: fewer classes
  if (nclass < 1) {                                                     // (B > 262,144: the identity order)
    for (int64_t q = tid; q < B; q += kBlock) order[q] = static_cast<int32_t>(q);
    return;
  }
  // (eight chunks' lengths are requested together: one dependent global load per chunk was ~1.3 us each, 85 us for a batch)
  constexpr int NWV = kBlock / kWave, UB = 8;
  for (int ch0 = wid; ch0 < nchunks; ch0 += NWV * UB) {
    int ln[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t q = static_cast<int64_t>(ch0 + u * NWV) * kWave + lane;
      ln[u] = q < B ? lens[q] : -1;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int ch = ch0 + u * NWV;
      if (ch >= nchunks) break;
      const int64_t q = static_cast<int64_t>(ch) * kWave + lane;
      const int c = q < B ? din_len_class(ln[u], L, nclass) : -1;
      for (int k = 0; k < nclass; ++k) {
        const uint64_t m = __ballot(c == k);
        if (lane == 0) cnt[k * nchunks + ch] = __popcll(m);
      }
    }
  }
  __syncthreads();
  const int total = nclass * nchunks, per = (total + kBlock - 1) / kBlock;
  const int lo = tid * per < total ? tid * per : total, hi = lo + per < total ? lo + per : total;
  int sum = 0;
  for (int e = lo; e < hi; ++e) sum += cnt[e];
  part[tid] = sum;
  __syncthreads();
  int run = 0;
#pragma unroll 8
  for (int e = 0; e < tid; ++e) run += part[e];                          // (<= 255 LDS reads per thread, pipelined)
  for (int e = lo; e < hi; ++e) {
    const int v = cnt[e];
    cnt[e] = run;
    run += v;
  }
  __syncthreads();
  for (int ch0 = wid; ch0 < nchunks; ch0 += NWV * UB) {
    int ln[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t q = static_cast<int64_t>(ch0 + u * NWV) * kWave + lane;
      ln[u] = q < B ? lens[q] : -1;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int ch = ch0 + u * NWV;
      if (ch >= nchunks) break;
      const int64_t q = static_cast<int64_t>(ch) * kWave + lane;
      const int c = q < B ? din_len_class(ln[u], L, nclass) : -1;
      for (int k = 0; k < nclass; ++k) {
        const uint64_t m = __ballot(c == k);
        if (c == k) order[cnt[k * nchunks + ch] + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<int32_t>(q);
      }
    }
  }
}

// =================================================================================================
// forward
//   Per sample the first layer is folded over the query:  z = (Wb + diag(q) Wd)^T k + zq  — W_eff [K,16] is formed
//   once per sample in registers (one FMA per weight), so a 16-key tile costs 4 NT MFMAs with both operands in
//   registers (half the MFMAs of the unfolded form, no weight reads and no q*k products in the tile loop), on two
//   independent accumulators (a 16x16x4 MFMA has 40 cycles of dependent latency against 32 of issue).  The key rows of
//   tile T + 1 are requested before tile T is computed (row ids one tile further ahead): the loop is bound by the
//   latency of random 4 K-byte-strided row reads from a multi-GB table, and the prefetch keeps two tiles per wave in flight.
// =================================================================================================
// row pointer from an id that is ALREADY in a register (GATHER) or from the position (dense rows); `ok` is cleared for
// ids outside [0, V); the pointer is always valid (row 0 stands in for a masked row)
template <bool GATHER>
__device__ __forceinline__ const float* din_row_of(const float* __restrict__ src, int64_t V, int32_t id, int64_t pos,
                                                   int K, bool& ok) {
  if (GATHER) {
    ok = ok && id >= 0 && id < V;
    return src + static_cast<int64_t>(ok ? id : 0) * K;
  }
  return src + pos * K;
}

template <int NT, bool GATHER>
__global__ __launch_bounds__(kBlock, 2) void din_fwd_mfma_kernel(
    const float* __restrict__ qsrc, const float* __restrict__ ksrc, int64_t V,
    const int32_t* __restrict__ item, const int32_t* __restrict__ seq, const int32_t* __restrict__ len,
    int64_t B, int L, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out,
    float* __restrict__ attn, float* __restrict__ hid, int32_t* __restrict__ order_out) {
  // `order_out` (nullable, [B]): ONE extra workgroup of this launch writes the samples as a stable partition by descending
  // key-tile count — the walk order of the two backward kernels (their `order`), which then hand every wave one sample of each
  // length class instead of whatever lengths its fixed stride happens to hit: their last third used to run with a fraction of
  // the waves (PMC, round 6).  Computed HERE because it costs ~10 us of dependent loads in one workgroup: beside the forward's
  // 50 us it is free, in front of it (in the id-stream launch) it cost the step more than the balance gave back.
  // It is workgroup 0 of a grid that is NOT enlarged (the launcher's persistent grid fills the chip exactly: a 513th workgroup
  // would start when the first of the others ends and put its 20 us behind the kernel instead of beside it).
  const bool extra = order_out != nullptr;
  if (extra && blockIdx.x == 0) {
    din_order_body(len, B, L, order_out);
    // with `hid` too: the transposed-type weight images of the backward's data kernel, built ONCE here (behind the hidden
    // activations, at hid + B * L * 16) instead of by every one of its workgroups (strided reads of W1: 11 us of that kernel)
    if (hid != nullptr)
      din_stage_weights<NT, true, false>(W1, nullptr, reinterpret_cast<float4*>(hid + static_cast<int64_t>(B) * L * kDH));
    return;
  }
  const int wg = static_cast<int>(blockIdx.x) - (extra ? 1 : 0);
  // `hid` (nullable, [B*L, 16]): the hidden activations h = sigmoid(z) of every live key are kept for the backward, which
  // then does not recompute the first layer (64 of its 128 MFMAs per tile and the forward weight images in LDS)
  constexpr int K = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* F = reinterpret_cast<float4*>(smem);                         // [3][NT][64]
  float* sc_all = reinterpret_cast<float*>(F + 3 * NT * 64);           // [4 waves][L]
  din_stage_weights<NT, false>(W1, F, nullptr);
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int i = lane & 15, kq = lane >> 4;
  float* sc = sc_all + wid * L;
  float w2r[4], b1r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w2r[r] = W2[4 * kq + r];
    b1r[r] = b1[4 * kq + r];
  }
  const float b2v = b2[0];
  const float rsK = 1.0f / sqrtf(static_cast<float>(K));
  const int64_t nwaves = static_cast<int64_t>(gridDim.x - (extra ? 1 : 0)) * (kBlock / kWave);
  auto clampn = [&](int n) { return n < 0 ? 0 : (n > L ? L : n); };
  auto seq_id = [&](int64_t b, int l) -> int32_t {        // always in bounds; masked by the caller's `ok`
    return GATHER ? seq[b * L + (l < L ? l : L - 1)] : 0;
  };

  int64_t slot = static_cast<int64_t>(wg) * (kBlock / kWave) + wid;
  if (slot >= B) return;
  auto sample_of = [&](int64_t sl) -> int64_t { return sl; };       // (the forward walks the batch in index order)
  int64_t b = sample_of(slot);
  // ---- prologue: the first sample's scalars, then its query row and first key tile (exposed once per wave) ----
  int n = clampn(len[b]);
  int32_t idn = seq_id(b, 16 + i);                                      // id of this lane's key in tile 1
  float4 q4[NT], kc[NT], kn[NT];
  {
    bool okq = true, okc = i < n;
    const float* qp = din_row_of<GATHER>(qsrc, V, GATHER ? item[b] : 0, b, K, okq);
    const float* kp = din_row_of<GATHER>(ksrc, V, seq_id(b, i), b * L + (okc ? i : 0), K, okc);
#pragma unroll
    for (int t = 0; t < NT; ++t) q4[t] = ld4(qp + 16 * t + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) kc[t] = ld4(kp + 16 * t + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      q4[t] = okq ? q4[t] : f4_zero();
      kc[t] = okc ? kc[t] : f4_zero();
    }
  }
  while (true) {
    // ---- scalars of the NEXT sample of this wave: requested a whole sample ahead of their use ----
    const int64_t nslot = slot + nwaves;
    const bool has_next = nslot < B;                                    // wave-uniform
    const int64_t nb = has_next ? sample_of(nslot) : b;
    const int64_t nbc = nb;
    const int n2 = clampn(len[nbc]);
    const int32_t idq2 = GATHER ? item[nbc] : 0;
    const int32_t idk2 = seq_id(nbc, i), idk2b = seq_id(nbc, 16 + i);

    f32x4 zq = din_zq<NT>(F, lane, q4);
#pragma unroll
    for (int r = 0; r < 4; ++r) zq[r] += b1r[r];
    float4 we[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 wb = F[(0 * NT + t) * 64 + lane], wd = F[(1 * NT + t) * 64 + lane];
      we[t] = f4_fma(q4[t], wd, wb);
    }
    float m = -INFINITY, den = 0.f;
    float4 o4[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o4[t] = f4_zero();
    const int tiles = n > 0 ? (n + 15) >> 4 : 1;                        // n == 0: one all-masked tile carries the prefetch
    bool okq2 = true;
    for (int T = 0; T < tiles; ++T) {
      const int l = 16 * T + i;
      const bool act = l < n;
      const bool more = T + 1 < tiles;                                  // wave-uniform
      // ---- rows in flight while this tile is computed: the sample's next tile, or the next sample's query + tile 0 ----
      bool okn = false;
      if (more) {
        okn = l + 16 < n;
        const float* np_ = din_row_of<GATHER>(ksrc, V, idn, b * L + (okn ? l + 16 : 0), K, okn);
#pragma unroll
        for (int t = 0; t < NT; ++t) kn[t] = ld4(np_ + 16 * t + 4 * kq);
        idn = seq_id(b, l + 32);                                        // id two tiles ahead
      } else if (has_next) {
        okn = i < n2;
        const float* qp2 = din_row_of<GATHER>(qsrc, V, idq2, nb, K, okq2);
        const float* np_ = din_row_of<GATHER>(ksrc, V, idk2, nb * L + (okn ? i : 0), K, okn);
#pragma unroll
        for (int t = 0; t < NT; ++t) q4[t] = ld4(qp2 + 16 * t + 4 * kq);      // q4 is dead by now (zq / we are formed)
#pragma unroll
        for (int t = 0; t < NT; ++t) kn[t] = ld4(np_ + 16 * t + 4 * kq);
      }
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        a0 = mfma16(we[t].x, kc[t].x, a0);
        a1 = mfma16(we[t].y, kc[t].y, a1);
        a0 = mfma16(we[t].z, kc[t].z, a0);
        a1 = mfma16(we[t].w, kc[t].w, a1);
      }
      float s = 0.f, hh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hh[r] = din_sigmoid((a0[r] + a1[r]) + zq[r]);
        s = fmaf(w2r[r], hh[r], s);
      }
      s = (group_sum4(s) + b2v) * rsK;
      if (act && hid != nullptr) st4(hid + (b * L + l) * kDH + 4 * kq, make_float4(hh[0], hh[1], hh[2], hh[3]));
      if (act) {
        const float mn = fmaxf(m, s);
        const float f = __expf(m - mn);   // m = -inf the first time -> 0
        const float e = __expf(s - mn);
        den = fmaf(den, f, e);
#pragma unroll
        for (int t = 0; t < NT; ++t) o4[t] = f4_fma(make_float4(e, e, e, e), kc[t], f4_scale(o4[t], f));
        m = mn;
        if (kq == 0) sc[l] = s;
      }
      if (more || has_next) {
#pragma unroll
        for (int t = 0; t < NT; ++t) kc[t] = okn ? kn[t] : f4_zero();
      }
    }
    // merge the 16 lanes' online-softmax states (fixed order: DPP butterfly)
    const float mt = row_max16(m);
    const float f = (m == -INFINITY) ? 0.f : __expf(m - mt);
    den = row_sum16(den * f);
    const float inv = den > 0.f ? 1.f / den : 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 o;
      o.x = row_sum16(o4[t].x * f) * inv;
      o.y = row_sum16(o4[t].y * f) * inv;
      o.z = row_sum16(o4[t].z * f) * inv;
      o.w = row_sum16(o4[t].w * f) * inv;
      if (i == 0) st4(out + b * K + 16 * t + 4 * kq, o);
    }
    for (int l = lane; l < L; l += kWave) attn[b * L + l] = (l < n) ? __expf(sc[l] - mt) * inv : 0.f;
    if (!has_next) break;
    b = nb;
    slot = nslot;
    n = n2;
    idn = idk2b;
#pragma unroll
    for (int t = 0; t < NT; ++t) q4[t] = okq2 ? q4[t] : f4_zero();
  }
}

// =================================================================================================
// backward, data part: gq, gkey, dz [B*L,16] (rows l >= len zeroed), Dz [B,16], per-wave partials of
// (db1[16] | dW2[16] | db2) -> small[nblocks*4][36]
// =================================================================================================
constexpr int kDinSmall = 2 * kDH + 4;

// Lab builds (-DLR_DIN_MARKS, scripts/lab/r06/din_marks.sh): wave 0 of workgroup 1 of the data kernel leaves shader-clock time
// stamps of its first two samples' phases in lr_din_marks (read back by lr_din_debug_marks).  The product build has none of it.
#ifdef LR_DIN_MARKS
__device__ unsigned long long lr_din_marks[64];
#define LR_DIN_MARK(i) do { if (blockIdx.x == 1 && threadIdx.x == 0 && (i) < 64) lr_din_marks[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LR_DIN_MARK(i) do { } while (0)
#endif
#ifndef LR_DIN_BWD_WAVES
#define LR_DIN_BWD_WAVES 2     // waves per SIMD the attention backward kernels are compiled for (profiling: 3 / 4 spill)
#endif
#ifndef LR_DIN_P1_TILES
#define LR_DIN_P1_TILES 2      // key tiles of the first backward pass in flight together (4, same box: 0.1605 vs 0.1576 ms)
#endif
#ifndef LR_DIN_BWD_WAVES_H
#define LR_DIN_BWD_WAVES_H 2   // the same for the form that reads the saved hidden activations
#endif
// SAVED_H: h = sigmoid(z) comes from the forward's `hid` buffer instead of being recomputed (no forward weight images, no zq)
template <int NT, bool GATHER, bool SAVED_H = false>
__global__ __launch_bounds__(kBlock, SAVED_H ? LR_DIN_BWD_WAVES_H : LR_DIN_BWD_WAVES) void din_bwd_data_kernel(
    const float* __restrict__ qsrc, const float* __restrict__ ksrc, int64_t V,
    const int32_t* __restrict__ item, const int32_t* __restrict__ seq, const int32_t* __restrict__ len,
    int64_t B, int L, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ attn, const float* __restrict__ gout,
    float* __restrict__ gq, float* __restrict__ gkey, float* __restrict__ dzbuf, float* __restrict__ Dzbuf,
    float* __restrict__ small, int keep_pad_rows, const float* __restrict__ hid, const int32_t* __restrict__ order) {
  constexpr int K = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* F = reinterpret_cast<float4*>(smem);                         // [3][NT][64]
  float4* Tw = F + 3 * NT * 64;                                        // [3][NT][64]
  float* sda_all = reinterpret_cast<float*>(Tw + 3 * NT * 64);         // [4 waves][L]
  // SAVED_H: the sample's query and output-gradient rows live in LDS during the second pass (one copy per lane group: every
  // key lane of a group holds the same 16-byte piece) instead of in 64 VGPRs
  float4* sqg_all = reinterpret_cast<float4*>(sda_all + 4 * ((L + 3) & ~3));   // [4 waves][2][NT][4]
  LR_DIN_MARK(0);
  if (SAVED_H && order != nullptr) {       // the forward's extra workgroup left the images behind `hid`: a coalesced 24 KB copy
    const float4* timg = reinterpret_cast<const float4*>(hid + static_cast<int64_t>(B) * L * kDH);
    for (int q = threadIdx.x; q < 3 * NT * 64; q += kBlock) Tw[q] = timg[q];
  } else {
    din_stage_weights<NT, true, !SAVED_H>(W1, F, Tw);    // (SAVED_H never reads the forward-type images)
  }
  __syncthreads();
  LR_DIN_MARK(1);
  int mark_s = 0;

  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int i = lane & 15, kq = lane >> 4;
  float* sda = sda_all + wid * L;
  float4* sqw = sqg_all + wid * 2 * NT * 4;
  float4* sgw = sqw + NT * 4;
  float w2r[4], b1r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w2r[r] = W2[4 * kq + r];
    b1r[r] = b1[4 * kq + r];
  }
  const float rsK = 1.0f / sqrtf(static_cast<float>(K));
  float dW2acc[4] = {0.f, 0.f, 0.f, 0.f}, db1acc[4] = {0.f, 0.f, 0.f, 0.f};
  float db2acc = 0.f;

  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / kWave);
  for (int64_t slot = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + wid; slot < B; slot += nwaves) {
    const int64_t b = order != nullptr ? order[slot] : slot;
    LR_DIN_MARK(2 + 8 * mark_s);
    int n = len[b];
    n = n < 0 ? 0 : (n > L ? L : n);
    float4 q4[NT], go4[NT];
    {
      bool ok = true;
      const float* qp = din_row_ptr<GATHER>(qsrc, V, item, b, K, ok);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        q4[t] = ok ? ld4(qp + 16 * t + 4 * kq) : f4_zero();
        go4[t] = ld4(gout + b * K + 16 * t + 4 * kq);
      }
    }
    if constexpr (SAVED_H) {
      if (i == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          sqw[t * 4 + kq] = q4[t];
          sgw[t * 4 + kq] = go4[t];
        }
      }
    }
    f32x4 zq = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!SAVED_H) {
      zq = din_zq<NT>(F, lane, q4);
#pragma unroll
      for (int r = 0; r < 4; ++r) zq[r] += b1r[r];
    }
    const int tiles = (n + 15) >> 4;
    LR_DIN_MARK(3 + 8 * mark_s);

    // pass 1: da_l = <gout, key_l>, dot = sum_l a_l da_l
    // (LR_DIN_P1_TILES tiles per trip: their rows are requested together — the walk is bound by the latency of random row reads,
    // one wave holds one sample, and a sample of 17 - 32 keys exposes that latency once instead of twice; per lane the sums are
    // taken in the same ascending key order as before.  Four tiles per trip measured 3 us slower than two.)
    float dotp = 0.f;
    constexpr int TP = SAVED_H ? LR_DIN_P1_TILES : 2;   // tiles whose rows are requested together (the recomputing form keeps q4 live: 2)
    for (int T = 0; T < tiles; T += TP) {
      bool okv[TP], actv[TP];
      const float* kpv[TP];
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const int l = 16 * (T + p) + i;
        actv[p] = l < n;
        okv[p] = actv[p];
        kpv[p] = din_row_ptr<GATHER>(ksrc, V, seq, b * L + (actv[p] ? l : 0), K, okv[p]);
      }
      float4 kv[TP][NT];
#pragma unroll
      for (int p = 0; p < TP; ++p) {
#pragma unroll
        for (int t = 0; t < NT; ++t) kv[p][t] = ld4(kpv[p] + 16 * t + 4 * kq);     // (the pointer is always valid: row 0 stands in)
      }
      float dv[TP];
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 k = okv[p] ? kv[p][t] : f4_zero();
          d = fmaf(go4[t].x, k.x, fmaf(go4[t].y, k.y, fmaf(go4[t].z, k.z, fmaf(go4[t].w, k.w, d))));
        }
        dv[p] = group_sum4(d);
      }
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const int l = 16 * (T + p) + i;
        if (actv[p]) {
          dotp = fmaf(attn[b * L + l], dv[p], dotp);
          if (kq == 0) sda[l] = dv[p];
        }
      }
    }
    const float dot = row_sum16(dotp);
    LR_DIN_MARK(4 + 8 * mark_s);

    // pass 2
    float4 dq4[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dq4[t] = f4_zero();
    float Dz[4] = {0.f, 0.f, 0.f, 0.f};
    // (the rows of tile T + 1 are requested before tile T is computed: they come out of L2 — pass 1 has just read them —
    // and that round trip now runs beside the tile's MFMAs instead of in front of them)
    float4 k4[NT], kn[SAVED_H ? NT : 1];       // (the recomputing form has no registers to spare for the look-ahead)
    bool okn = false;
    if constexpr (SAVED_H) {
      bool ok0 = i < n;
      const float* kp = din_row_ptr<GATHER>(ksrc, V, seq, b * L + (ok0 ? i : 0), K, ok0);
#pragma unroll
      for (int t = 0; t < NT; ++t) kn[t] = ld4(kp + 16 * t + 4 * kq);
      okn = ok0;
    }
    for (int T = 0; T < tiles; ++T) {
      const int l = 16 * T + i;
      const bool act = l < n;
      const int64_t pos = b * L + (act ? l : 0);
      if constexpr (!SAVED_H) {
        bool ok = act;
        const float* kp = din_row_ptr<GATHER>(ksrc, V, seq, pos, K, ok);
#pragma unroll
        for (int t = 0; t < NT; ++t) k4[t] = ok ? ld4(kp + 16 * t + 4 * kq) : f4_zero();
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) k4[t] = okn ? kn[t] : f4_zero();
      }
      if (SAVED_H && T + 1 < tiles) {                                            // wave-uniform
        bool ok1 = l + 16 < n;
        const float* kp = din_row_ptr<GATHER>(ksrc, V, seq, b * L + (ok1 ? l + 16 : 0), K, ok1);
#pragma unroll
        for (int t = 0; t < NT; ++t) kn[t] = ld4(kp + 16 * t + 4 * kq);
        okn = ok1;
      }
      const float a_l = act ? attn[pos] : 0.f;
      const float draw = act ? a_l * (sda[l] - dot) * rsK : 0.f;     // d loss / d (pre-scale score)
      // the weight images are re-read from LDS for every tile: an opaque lane index keeps the compiler from
      // hoisting those (loop-invariant) reads into ~200 registers
      int lw = lane;
      asm volatile("" : "+v"(lw));
      float hv[4];
      if constexpr (SAVED_H) {
        const float4 h4 = act ? ld4(hid + pos * kDH + 4 * kq) : f4_zero();
        hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
      } else {
        const f32x4 acc = din_z_tile<NT>(F, lw, k4, q4);
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = din_sigmoid(acc[r] + zq[r]);
      }
      float dz[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = hv[r];
        dW2acc[r] = fmaf(draw, h, dW2acc[r]);
        dz[r] = draw * w2r[r] * h * (1.f - h);
        db1acc[r] += dz[r];
        Dz[r] += dz[r];
      }
      db2acc += draw;
      if (act) st4(dzbuf + pos * kDH + 4 * kq, make_float4(dz[0], dz[1], dz[2], dz[3]));
      // d key = a_l gout + dz (W1b-W1c)^T + q * (dz W1d^T); d q += k * (dz W1d^T)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const float4 wb = Tw[(0 * NT + u) * 64 + lw], wd = Tw[(1 * NT + u) * 64 + lw];
        f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
        g1 = mfma16(wb.x, dz[0], g1);
        g2 = mfma16(wd.x, dz[0], g2);
        g1 = mfma16(wb.y, dz[1], g1);
        g2 = mfma16(wd.y, dz[1], g2);
        g1 = mfma16(wb.z, dz[2], g1);
        g2 = mfma16(wd.z, dz[2], g2);
        g1 = mfma16(wb.w, dz[3], g1);
        g2 = mfma16(wd.w, dz[3], g2);
        float4 qv, gv;
        if constexpr (SAVED_H) {
          qv = sqw[u * 4 + kq];
          gv = sgw[u * 4 + kq];
        } else {
          qv = q4[u];
          gv = go4[u];
        }
        float4 gk;
        gk.x = fmaf(a_l, gv.x, fmaf(qv.x, g2[0], g1[0]));
        gk.y = fmaf(a_l, gv.y, fmaf(qv.y, g2[1], g1[1]));
        gk.z = fmaf(a_l, gv.z, fmaf(qv.z, g2[2], g1[2]));
        gk.w = fmaf(a_l, gv.w, fmaf(qv.w, g2[3], g1[3]));
        if (act) st4(gkey + pos * K + 16 * u + 4 * kq, gk);
        dq4[u].x = fmaf(k4[u].x, g2[0], dq4[u].x);
        dq4[u].y = fmaf(k4[u].y, g2[1], dq4[u].y);
        dq4[u].z = fmaf(k4[u].z, g2[2], dq4[u].z);
        dq4[u].w = fmaf(k4[u].w, g2[3], dq4[u].w);
      }
    }
    // rows past the sequence end: zero gradient — unless the caller drops those positions from the table update
    // anyway (`keep_pad_rows`: 103 MB of zeros per launch at cfg 3); the parameter kernel masks dz by `len` itself
    if (!keep_pad_rows) {
      for (int l = n + i; l < L; l += 16) {
        const int64_t pos = b * L + l;
#pragma unroll
        for (int u = 0; u < NT; ++u) st4(gkey + pos * K + 16 * u + 4 * kq, f4_zero());
      }
    }
    LR_DIN_MARK(5 + 8 * mark_s);
#ifdef LR_DIN_MARKS
    if (blockIdx.x == 1 && threadIdx.x == 0) lr_din_marks[7 + 8 * mark_s] = static_cast<unsigned long long>(n);
#endif
    // Dz_j = sum over the sample's keys; d q += (W1a+W1c) Dz
#pragma unroll
    for (int r = 0; r < 4; ++r) Dz[r] = row_sum16(Dz[r]);
    if (i == 0) st4(Dzbuf + b * kDH + 4 * kq, make_float4(Dz[0], Dz[1], Dz[2], Dz[3]));
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float4 wa = Tw[(2 * NT + u) * 64 + lane];
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      g = mfma16(wa.x, Dz[0], g);
      g = mfma16(wa.y, Dz[1], g);
      g = mfma16(wa.z, Dz[2], g);
      g = mfma16(wa.w, Dz[3], g);
      float4 o;
      o.x = row_sum16(dq4[u].x) + g[0];
      o.y = row_sum16(dq4[u].y) + g[1];
      o.z = row_sum16(dq4[u].z) + g[2];
      o.w = row_sum16(dq4[u].w) + g[3];
      if (i == 0) st4(gq + b * K + 16 * u + 4 * kq, o);
    }
    LR_DIN_MARK(6 + 8 * mark_s);
    ++mark_s;
  }
  LR_DIN_MARK(60);
  // per-wave partials of db1 | dW2 | db2 (sum over this wave's keys = lanes of a row)
  float* sm = small + (static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + wid) * kDinSmall;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float x = row_sum16(db1acc[r]), y = row_sum16(dW2acc[r]);
    if (i == 0) {
      sm[4 * kq + r] = x;
      sm[kDH + 4 * kq + r] = y;
    }
  }
  const float z = row_sum16(db2acc);     // every lane group counted every key once
  if (lane == 0) sm[2 * kDH] = z;
}

// =================================================================================================
// backward, parameter part: per-workgroup partials of
//   GA[d][j] = sum_b q_b[d] Dz_b[j],  GB[d][j] = sum_{b,l} k[d] dz[j],  GD[d][j] = sum_{b,l} q[d] k[d] dz[j]
// (gW1 = [GA | GB | GA-GB | GD], layers/attention.py:50-52).  One wavefront per sample; the key tile goes
// through LDS to get the reduction index (the key) off the lane axis.
// =================================================================================================
template <int NT, bool GATHER>
__global__ __launch_bounds__(kBlock, LR_DIN_BWD_WAVES) void din_bwd_param_kernel(
    const float* __restrict__ qsrc, const float* __restrict__ ksrc, int64_t V,
    const int32_t* __restrict__ item, const int32_t* __restrict__ seq, const int32_t* __restrict__ len,
    int64_t B, int L, const float* __restrict__ dzbuf, const float* __restrict__ Dzbuf,
    float* __restrict__ partial, const int32_t* __restrict__ order) {
  constexpr int K = 16 * NT, LDK = K + 4, NW = kBlock / kWave;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile_all = reinterpret_cast<float*>(smem);                    // [NW][16][LDK]; reused for the fold
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int i = lane & 15, kq = lane >> 4;
  float* tile = tile_all + wid * 16 * LDK;

  f32x4 GA[NT], GB[NT], GD[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    GA[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    GB[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    GD[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * NW;
  for (int64_t slot = static_cast<int64_t>(blockIdx.x) * NW + wid; slot < B; slot += nwaves) {
    const int64_t b = order != nullptr ? order[slot] : slot;
    int n = len[b];
    n = n < 0 ? 0 : (n > L ? L : n);
    bool qok = true;
    const float* qp = din_row_ptr<GATHER>(qsrc, V, item, b, K, qok);
    float qd[NT];      // q[16 u + lane % 16]: scales the A operand of GD
    const float Dzj = Dzbuf[b * kDH + i];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      qd[u] = qok ? qp[16 * u + i] : 0.f;
      const float4 qc = qok ? ld4(qp + 16 * u + 4 * kq) : f4_zero();     // rows 16 u + 4 kq + r of the C tile
      GA[u][0] = fmaf(qc.x, Dzj, GA[u][0]);
      GA[u][1] = fmaf(qc.y, Dzj, GA[u][1]);
      GA[u][2] = fmaf(qc.z, Dzj, GA[u][2]);
      GA[u][3] = fmaf(qc.w, Dzj, GA[u][3]);
    }
    const int tiles = (n + 15) >> 4;
    for (int T = 0; T < tiles; ++T) {
      const int l = 16 * T + i;
      const bool act = l < n;
      bool ok = act;
      const float* kp = din_row_ptr<GATHER>(ksrc, V, seq, b * L + (act ? l : 0), K, ok);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        st4(tile + i * LDK + 16 * t + 4 * kq, ok ? ld4(kp + 16 * t + 4 * kq) : f4_zero());
      // B operand: dz[key = 4 kq + r][j = lane % 16]  (zero rows past the sequence end were written by the data kernel)
      float dzv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lk = 16 * T + 4 * kq + r;
        dzv[r] = lk < n ? dzbuf[(b * L + lk) * kDH + i] : 0.f;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);      // my LDS writes have landed (the tile is private to the wave)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float kv = tile[(4 * kq + r) * LDK + 16 * u + i];
          GB[u] = mfma16(kv, dzv[r], GB[u]);
          GD[u] = mfma16(kv * qd[u], dzv[r], GD[u]);
        }
      }
      asm volatile("" ::: "memory");
    }
  }
  // fold the 4 waves in LDS in wave order ((w0 + w1) + w2) + w3 and emit the workgroup's partial [3][K][16]
  __syncthreads();
  float* fold = tile_all;                       // 3 * K * 16 floats (host sizes the LDS for the larger of the two uses)
  for (int w = 0; w < NW; ++w) {
    if (wid == w) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (16 * u + 4 * kq + r) * kDH + i;
          if (w == 0) {
            fold[o] = GA[u][r];
            fold[K * kDH + o] = GB[u][r];
            fold[2 * K * kDH + o] = GD[u][r];
          } else {
            fold[o] += GA[u][r];
            fold[K * kDH + o] += GB[u][r];
            fold[2 * K * kDH + o] += GD[u][r];
          }
        }
    }
    __syncthreads();
  }
  float* dst = partial + static_cast<int64_t>(blockIdx.x) * (3 * K * kDH);
  for (int q = threadIdx.x; q < 3 * K * kDH; q += kBlock) dst[q] = fold[q];
}

// final, fixed-order reduction over workgroups: gW1 [4K,16] = GA | GB | GA-GB | GD, gb1, gW2, gb2.
// 16 outputs x 64 slices per 1,024-thread block: slice s adds contributions s, s+64, ... in order (eight independent
// loads per array at 512 workgroups: one latency round instead of a 32-long chain), then the 64 slice sums are added
// in slice order — the same order every run.
constexpr int kDinRedSlices = 64;
__global__ __launch_bounds__(16 * kDinRedSlices) void din_reduce2_kernel(const float* __restrict__ partial, int nblocks,
                                                                         const float* __restrict__ small, int nsmall,
                                                                         int K, float* __restrict__ gW1,
                                                                         float* __restrict__ gb1, float* __restrict__ gW2,
                                                                         float* __restrict__ gb2) {
  __shared__ float red[3][kDinRedSlices][17];
  const int KH = K * kDH;
  const int total = KH + 2 * kDH + 1;
  const int qo = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + qo;
  float a = 0.f, bsum = 0.f, d = 0.f;
  if (q < KH) {
#pragma unroll 4
    for (int blk = sl; blk < nblocks; blk += kDinRedSlices) {
      const float* p = partial + static_cast<int64_t>(blk) * 3 * KH;
      a += p[q];
      bsum += p[KH + q];
      d += p[2 * KH + q];
    }
  } else if (q < total) {
#pragma unroll 4
    for (int w = sl; w < nsmall; w += kDinRedSlices) a += small[static_cast<int64_t>(w) * kDinSmall + (q - KH)];
  }
  red[0][sl][qo] = a;
  red[1][sl][qo] = bsum;
  red[2][sl][qo] = d;
  __syncthreads();
  if (sl == 0 && q < total) {
    float ta = 0.f, tb = 0.f, td = 0.f;
#pragma unroll
    for (int s = 0; s < kDinRedSlices; ++s) {
      ta += red[0][s][qo];
      tb += red[1][s][qo];
      td += red[2][s][qo];
    }
    if (q < KH) {
      gW1[q] = ta;
      gW1[KH + q] = tb;
      gW1[2 * KH + q] = ta - tb;
      gW1[3 * KH + q] = td;
    } else {
      const int r = q - KH;
      if (r < kDH) gb1[r] = ta;
      else if (r < 2 * kDH) gW2[r - kDH] = ta;
      else gb2[0] = ta;
    }
  }
}

}  // namespace lr
