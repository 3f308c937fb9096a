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
    float* __restrict__ e, float* __restrict__ pair, float* __restrict__ fsum,
    const float* __restrict__ lin, float* __restrict__ lin_out) {
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
            const bool ok = id >= 0 && id < V;
            if (ok) x[u] = ld4(src + static_cast<int64_t>(id) * K + c4);
            if (lin != nullptr && c4 == 0) lin_out[row0 + f] = ok ? lin[id] : 0.f;
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

// ---------------------------------------------------------------------------------------
// Fused backward + Adam ("wavefront-bucketed" over the CSR-by-row of the batch).
//   per distinct row r with positions P(r) = {q = b*F + f}:
//     g_r  = sum_q ( gdeep[q] - bn_a[f] )                       (deep path, BN-fold offset)
//          + sum_q gpair[b]*fsum[b]  -  w_r * sum_q ( gpair[b] + bn_c[f] )
//     glin = sum_q glin[q]                                       (linear weight, optional)
//   then one Adam update of (w,m,v)[r] and (lin,lin_m,lin_v)[r].
// Bucketing: runs of <= kLongSeg positions are summed by ONE row group (LPR lanes, 2 positions
// in flight); longer runs (the Zipf head: thousands of positions on one row) are appended to
// a device list and summed by a WHOLE workgroup (256/LPR groups striding the run, LDS tree in
// fixed group order) so the kernel's critical path is not one hot row.  Results do not depend
// on the order of the list: every row's summation order is fixed by its own run.
// ---------------------------------------------------------------------------------------
constexpr int kLongSeg = 32;

struct FmBwdArgs {
  float* table; float* m; float* v;
  float* lin; float* lin_m; float* lin_v;          // nullable (all or none)
  const float* gdeep; const float* gpair; const float* fsum;
  const float* glin;                               // nullable, FIELD-MAJOR [F,B]
  const float* bn_a; const float* bn_c;            // nullable, [F*K] each
  const int32_t* seg_pos; const int32_t* seg_rows; const int32_t* seg_start;
  const int32_t* n_seg;
  int32_t* long_count; int32_t* long_list;         // workspace
  float* grows_out; float* glin_out;               // != NULL: "rows" mode (see lr_fm_embed_bwd_rows_f32)
  int F;
  int64_t B;
};

template <int LPR>
struct FmAcc {
  float4 gd, gps, gp;
  float gl;
};

template <int LPR>
__device__ __forceinline__ void fm_acc_pos(const FmBwdArgs& A, int32_t q, int c4, FmAcc<LPR>& acc) {
  constexpr int K = LPR * 4;
  const int64_t b = q / A.F;
  const int f = q - static_cast<int32_t>(b) * A.F;
  const float4 a = ld4(A.gpair + b * K + c4);
  const float4 fs = ld4(A.fsum + b * K + c4);
  if (A.gdeep != nullptr) acc.gd = f4_add(acc.gd, ld4(A.gdeep + static_cast<int64_t>(q) * K + c4));
  acc.gps = f4_fma(a, fs, acc.gps);
  acc.gp = f4_add(acc.gp, a);
  if (A.bn_a != nullptr) {
    acc.gd = f4_sub(acc.gd, ld4(A.bn_a + f * K + c4));
    acc.gp = f4_add(acc.gp, ld4(A.bn_c + f * K + c4));
  }
  // field-major: the positions of a run share f and ascend in b, and runs are walked in row
  // (= field) order, so these 4-byte reads stay inside one 4*B-byte strip that lives in L2
  if (A.glin != nullptr) acc.gl += A.glin[static_cast<int64_t>(f) * A.B + b];
}

// `s` = run (segment) number, `row` = table row.  In "rows" mode the table is the per-step row
// cache addressed by run number and the summed gradient is written out instead of applied.
template <int LPR>
__device__ __forceinline__ void fm_apply(const FmBwdArgs& A, int64_t s, int32_t row, int c4,
                                         const FmAcc<LPR>& acc, const AdamCoef& coef) {
  constexpr int K = LPR * 4;
  const bool rows_mode = A.grows_out != nullptr;
  const int64_t off = (rows_mode ? s : static_cast<int64_t>(row)) * K + c4;
  const float4 w = ld4(A.table + off);
  float4 g;
  g.x = acc.gd.x + (acc.gps.x - w.x * acc.gp.x);
  g.y = acc.gd.y + (acc.gps.y - w.y * acc.gp.y);
  g.z = acc.gd.z + (acc.gps.z - w.z * acc.gp.z);
  g.w = acc.gd.w + (acc.gps.w - w.w * acc.gp.w);
  if (rows_mode) {
    st4(A.grows_out + off, g);
    if (A.glin_out != nullptr && c4 == 0) A.glin_out[s] = acc.gl;
    return;
  }
  float4 mm = ld4(A.m + off), vv = ld4(A.v + off);
  st4(A.table + off, adam_vec(w, g, mm, vv, coef));
  st4(A.m + off, mm);
  st4(A.v + off, vv);
  if (A.lin != nullptr && c4 == 0) {
    float lm = A.lin_m[row], lv = A.lin_v[row];
    A.lin[row] = adam_elem(A.lin[row], acc.gl, lm, lv, coef);
    A.lin_m[row] = lm;
    A.lin_v[row] = lv;
  }
}

// Short runs: one run per row group.  The run's position ids are fetched by the group's lanes in
// ONE load and handed round with ds_bpermute, so the per-position loads (gdeep row + the sample's
// gpair/fsum) are address-ready and the compiler can keep two positions in flight; 64 VGPRs ->
// 8 waves/SIMD of independent runs cover the latencies.
template <int LPR>
__device__ __forceinline__ void fm_bwd_short_runs(const FmBwdArgs& A, const AdamCoef& coef, int bid,
                                                  int nblocks) {
  constexpr int K = LPR * 4;
  const int n_seg = *A.n_seg;
  const int64_t gtid = static_cast<int64_t>(bid) * kBlock + threadIdx.x;
  const int gl = static_cast<int>(gtid % LPR);
  const int c4 = gl * 4;
  const int64_t ngroups = static_cast<int64_t>(nblocks) * kBlock / LPR;
  const bool rows_mode = A.grows_out != nullptr;
  for (int64_t s = gtid / LPR; s < n_seg; s += ngroups) {
    const int a0 = A.seg_start[s], a1 = A.seg_start[s + 1];
    if (a1 - a0 > kLongSeg) continue;       // on the long-run list (fm_bwd_classify_kernel)
    const int32_t row = A.seg_rows ? A.seg_rows[s] : 0;
    const int64_t off = (rows_mode ? s : static_cast<int64_t>(row)) * K + c4;
    FmAcc<LPR> acc{f4_zero(), f4_zero(), f4_zero(), 0.f};
    for (int base = a0; base < a1; base += LPR) {
      const int nq = (a1 - base) < LPR ? (a1 - base) : LPR;
      const int32_t qmine = A.seg_pos[base + (gl < nq ? gl : 0)];
#pragma unroll 2
      for (int i = 0; i < nq; ++i) fm_acc_pos<LPR>(A, __shfl(qmine, i, LPR), c4, acc);
    }
    // The row is read-modify-written back to back: its lines are still in L2 when the stores
    // arrive, so every line goes to HBM once.  (Requesting w/m/v before the walk hides their
    // latency but lets the lines fall out of L2 first — measured: +60 % fabric traffic.)
    const float4 w = ld4(A.table + off);
    float4 mm = f4_zero(), vv = f4_zero();
    float lw = 0.f, lm = 0.f, lv = 0.f;
    if (!rows_mode) {
      mm = ld4(A.m + off);
      vv = ld4(A.v + off);
      if (A.lin != nullptr && gl == 0) {
        lw = A.lin[row]; lm = A.lin_m[row]; lv = A.lin_v[row];
      }
    }
    float4 g;
    g.x = acc.gd.x + (acc.gps.x - w.x * acc.gp.x);
    g.y = acc.gd.y + (acc.gps.y - w.y * acc.gp.y);
    g.z = acc.gd.z + (acc.gps.z - w.z * acc.gp.z);
    g.w = acc.gd.w + (acc.gps.w - w.w * acc.gp.w);
    if (rows_mode) {
      st4(A.grows_out + off, g);
      if (A.glin_out != nullptr && gl == 0) A.glin_out[s] = acc.gl;
      continue;
    }
    st4(A.table + off, adam_vec(w, g, mm, vv, coef));
    st4(A.m + off, mm);
    st4(A.v + off, vv);
    if (A.lin != nullptr && gl == 0) {
      A.lin[row] = adam_elem(lw, acc.gl, lm, lv, coef);
      A.lin_m[row] = lm;
      A.lin_v[row] = lv;
    }
  }
}

template <int LPR>
__device__ __forceinline__ void fm_bwd_long_runs(const FmBwdArgs& A, const AdamCoef& coef, int bid,
                                                 int nblocks) {
  constexpr int NG = kBlock / LPR;  // row groups per workgroup
  __shared__ float4 red[NG][LPR][3];
  __shared__ float redl[NG];
  const int n_long = *A.long_count;
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR, c4 = gl * 4;
  for (int li = bid; li < n_long; li += nblocks) {
    const int32_t s = A.long_list[li];
    const int p0 = A.seg_start[s], p1 = A.seg_start[s + 1];
    FmAcc<LPR> acc{f4_zero(), f4_zero(), f4_zero(), 0.f};
    // group g takes chunks g, g+NG, ... of LPR consecutive positions: one load fetches a chunk's
    // ids (one per lane), ds_bpermute hands them round (as in the short-run kernel)
    for (int base = p0 + grp * LPR; base < p1; base += NG * LPR) {
      const int nq = (p1 - base) < LPR ? (p1 - base) : LPR;
      const int32_t qmine = A.seg_pos[base + (gl < nq ? gl : 0)];
#pragma unroll 2
      for (int i = 0; i < nq; ++i) fm_acc_pos<LPR>(A, __shfl(qmine, i, LPR), c4, acc);
    }
    red[grp][gl][0] = acc.gd;
    red[grp][gl][1] = acc.gps;
    red[grp][gl][2] = acc.gp;
    if (gl == 0) redl[grp] = acc.gl;
    __syncthreads();
    if (grp == 0) {
      FmAcc<LPR> t{f4_zero(), f4_zero(), f4_zero(), 0.f};
#pragma unroll 4
      for (int g = 0; g < NG; ++g) {  // fixed order
        t.gd = f4_add(t.gd, red[g][gl][0]);
        t.gps = f4_add(t.gps, red[g][gl][1]);
        t.gp = f4_add(t.gp, red[g][gl][2]);
        t.gl += redl[g];
      }
      fm_apply<LPR>(A, s, A.seg_rows ? A.seg_rows[s] : 0, c4, t, coef);
    }
    __syncthreads();
  }
}

// Long runs are listed by a tiny pre-pass over the run lengths; then ONE launch serves both kinds:
// the first kLongBlocks workgroups walk the long-run list (a whole workgroup per run), all the
// others the short runs (a row group per run).  The latency-bound long runs (the Zipf head: a few
// thousand rows holding ~20 % of the positions) thereby execute underneath the bandwidth-bound
// short-run traffic instead of after it.
constexpr int kLongBlocks = kNumCU;

__global__ __launch_bounds__(kBlock) void fm_bwd_classify_kernel(
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ n_seg_ptr,
    int32_t* __restrict__ long_count, int32_t* __restrict__ long_list) {
  const int n_seg = *n_seg_ptr;
  const int stride = gridDim.x * kBlock;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n_seg; s += stride)
    if (seg_start[s + 1] - seg_start[s] > kLongSeg) long_list[atomicAdd(long_count, 1)] = s;
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void fm_bwd_adam_kernel(FmBwdArgs A, AdamCoef coef) {
  if (blockIdx.x < kLongBlocks)
    fm_bwd_long_runs<LPR>(A, coef, blockIdx.x, kLongBlocks);
  else
    fm_bwd_short_runs<LPR>(A, coef, blockIdx.x - kLongBlocks, gridDim.x - kLongBlocks);
}

// ---------------------------------------------------------------------------------------
// Batch statistics of the gathered block e[B,F,K] WITHOUT reading e: column (f,k) of e holds
// table[row, k] once per position of `row`, so
//     sum_b e[b,f,k]   = sum over the field's runs of  len(run) * table[row,k]
//     sum_b e[b,f,k]^2 = sum over the field's runs of  len(run) * table[row,k]^2 .
// The runs of a field are contiguous in segment order (segments are sorted by global row and a
// field owns a contiguous row range).  Workgroup (f, c) reduces chunk c of field f's runs
// (row group per run, LDS tree in fixed order) into partial[f][c][{sum,sumsq}][K]; the caller
// adds the C partials.  Reads the distinct rows once (~62 % of the positions on Zipf ids).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_rows(const int32_t* __restrict__ a, int n, int32_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// `slots` != NULL (row-sharded tables): the rows live in a per-step row cache addressed through the
// position -> cache-slot map; the run's row is cache[slots[seg_pos[seg_start[r]]]].
template <int LPR>
__global__ __launch_bounds__(kBlock) void fm_field_stats_kernel(
    const float* __restrict__ table, const int32_t* __restrict__ seg_rows,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ n_seg_ptr,
    const int32_t* __restrict__ field_row_start, int C, float* __restrict__ partial,
    const int32_t* __restrict__ seg_pos, const int32_t* __restrict__ slots) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  __shared__ float4 red[NG][LPR][2];
  const int f = blockIdx.x / C, c = blockIdx.x % C;
  const int n_seg = *n_seg_ptr;
  const int lo = lower_bound_rows(seg_rows, n_seg, field_row_start[f]);
  const int hi = lower_bound_rows(seg_rows, n_seg, field_row_start[f + 1]);
  const int64_t span = hi - lo;
  const int beg = lo + static_cast<int>(span * c / C), end = lo + static_cast<int>(span * (c + 1) / C);
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR, c4 = gl * 4;
  float4 s = f4_zero(), q = f4_zero();
  auto run_row = [&](int r, float& cnt) -> int32_t {
    const int a0 = seg_start[r];
    cnt = static_cast<float>(seg_start[r + 1] - a0);
    return slots != nullptr ? slots[seg_pos[a0]] : seg_rows[r];
  };
  int r = beg + grp;
  for (; r + 3 * NG < end; r += 4 * NG) {      // four rows in flight per row group (same summation order as below)
    float c0, c1, c2, c3;
    const int32_t r0 = run_row(r, c0), r1 = run_row(r + NG, c1), r2 = run_row(r + 2 * NG, c2), r3 = run_row(r + 3 * NG, c3);
    const float4 w0 = ld4(table + static_cast<int64_t>(r0) * K + c4);
    const float4 w1 = ld4(table + static_cast<int64_t>(r1) * K + c4);
    const float4 w2 = ld4(table + static_cast<int64_t>(r2) * K + c4);
    const float4 w3 = ld4(table + static_cast<int64_t>(r3) * K + c4);
    s = f4_fma(make_float4(c0, c0, c0, c0), w0, s);
    q = f4_fma(make_float4(c0, c0, c0, c0), f4_mul(w0, w0), q);
    s = f4_fma(make_float4(c1, c1, c1, c1), w1, s);
    q = f4_fma(make_float4(c1, c1, c1, c1), f4_mul(w1, w1), q);
    s = f4_fma(make_float4(c2, c2, c2, c2), w2, s);
    q = f4_fma(make_float4(c2, c2, c2, c2), f4_mul(w2, w2), q);
    s = f4_fma(make_float4(c3, c3, c3, c3), w3, s);
    q = f4_fma(make_float4(c3, c3, c3, c3), f4_mul(w3, w3), q);
  }
  for (; r < end; r += NG) {
    float cnt;
    const int32_t row = run_row(r, cnt);
    const float4 w = ld4(table + static_cast<int64_t>(row) * K + c4);
    s = f4_fma(make_float4(cnt, cnt, cnt, cnt), w, s);
    q = f4_fma(make_float4(cnt, cnt, cnt, cnt), f4_mul(w, w), q);
  }
  red[grp][gl][0] = s;
  red[grp][gl][1] = q;
  __syncthreads();
  if (grp == 0) {
    float4 ts = f4_zero(), tq = f4_zero();
#pragma unroll 4
    for (int g = 0; g < NG; ++g) {  // fixed order
      ts = f4_add(ts, red[g][gl][0]);
      tq = f4_add(tq, red[g][gl][1]);
    }
    float* dst = partial + static_cast<int64_t>(blockIdx.x) * 2 * K;
    st4(dst + c4, ts);
    st4(dst + K + c4, tq);
  }
}

// ---------------------------------------------------------------------------------------
// Adam over per-position row gradients that were written IN RUN ORDER (lr_deepfm_l1_dgrad_f32):
// run s owns rows [seg_start[s], seg_start[s+1]) of `ge`, so a run is one contiguous stream.
//   E   = sum_p ge[p]                       (ascending p: fixed order)
//   sgl = sum_p gl[seg_pos[p] / F]          (d loss / d logit of the position's sample; L2-resident)
//   g   = E - n * bn_a[f] - w * (n * bn_c[f] + sgl * wp)        n = run length, f = field of the run
//   lin: g_lin = sgl * lin_scale[f]
// Short runs: one row group per run; long runs (> kLongSeg): whole workgroup + LDS tree, listed
// by fm_bwd_classify_kernel — the same bucketing as fm_bwd_adam_kernel above.
// ---------------------------------------------------------------------------------------
struct FmRowsArgs {
  float* table; float* m; float* v;
  float* lin; float* lin_m; float* lin_v;          // nullable (all or none)
  const float* ge; const float* gl; const float* wp;
  const float* bn_a; const float* bn_c;            // nullable, [F*K]
  const float* lin_scale;                          // [F], with lin
  const int32_t* seg_pos; const int32_t* seg_rows; const int32_t* seg_start; const int32_t* n_seg;
  const int32_t* long_count; const int32_t* long_list;
  const AdamCoef* coef_dev;                        // != NULL: coefficients are read from device memory
  // gradient mode (row-sharded tables): `table` / `lin` are the per-step row caches (read only), the run's row
  // is cache[slots[first position]] and the per-row gradients are WRITTEN to grows / glin_rows at that slot
  const int32_t* slots; float* grows; float* glin_rows;
  // compact gradient mode (`slots` == NULL, `grows` != NULL): `table` / `lin` are the tables themselves (read only) and run s
  // writes its gradient at grows[s] / glin_rows[s] — the operand of the dense table pass lr_adam_dense_rows_f32
  int compact;
  int F;
};

// the row's own operands: requested as soon as the row is known (in front of the walk over the run's
// positions), consumed by fm_rows_finish
struct FmRowOperands {
  float4 w, mm, vv;
  float lw, lm, lv;
};
template <int LPR>
__device__ __forceinline__ FmRowOperands fm_rows_load(const FmRowsArgs& A, int32_t row, int c4, int gl_lane) {
  constexpr int K = LPR * 4;
  const int64_t off = static_cast<int64_t>(row) * K + c4;
  FmRowOperands o;
  o.w = ld4(A.table + off);
  const bool adam = A.grows == nullptr;
  o.mm = adam ? ld4(A.m + off) : f4_zero();
  o.vv = adam ? ld4(A.v + off) : f4_zero();
  o.lw = o.lm = o.lv = 0.f;
  if (adam && A.lin != nullptr && gl_lane == 0) { o.lw = A.lin[row]; o.lm = A.lin_m[row]; o.lv = A.lin_v[row]; }
  return o;
}

template <int LPR>
__device__ __forceinline__ void fm_rows_finish(const FmRowsArgs& A, int32_t row, int f, int n, int c4,
                                               int gl_lane, float4 E, float sgl, const AdamCoef& coef,
                                               const FmRowOperands& o, int32_t run) {
  constexpr int K = LPR * 4;
  const int64_t off = static_cast<int64_t>(row) * K + c4;
  const float4 w = o.w;
  float4 mm = o.mm, vv = o.vv;
  float lw = o.lw, lm = o.lm, lv = o.lv;
  const float fn = static_cast<float>(n);
  float4 cw = A.wp != nullptr ? ld4(A.wp + c4) : f4_zero();
  cw.x *= sgl; cw.y *= sgl; cw.z *= sgl; cw.w *= sgl;
  float4 g = E;
  if (A.bn_a != nullptr) {
    const float4 a = ld4(A.bn_a + f * K + c4), c = ld4(A.bn_c + f * K + c4);
    g.x -= fn * a.x; g.y -= fn * a.y; g.z -= fn * a.z; g.w -= fn * a.w;
    cw.x = fmaf(fn, c.x, cw.x); cw.y = fmaf(fn, c.y, cw.y);
    cw.z = fmaf(fn, c.z, cw.z); cw.w = fmaf(fn, c.w, cw.w);
  }
  g.x -= w.x * cw.x; g.y -= w.y * cw.y; g.z -= w.z * cw.z; g.w -= w.w * cw.w;
  if (A.grows != nullptr) {      // gradient mode: hand the per-row gradient to the exchange / the dense table pass
    const int32_t out = A.compact ? run : row;
    st4(A.grows + static_cast<int64_t>(out) * K + c4, g);
    if (A.glin_rows != nullptr && gl_lane == 0) A.glin_rows[out] = sgl * A.lin_scale[f];
    return;
  }
  st4(A.table + off, adam_vec(w, g, mm, vv, coef));
  st4(A.m + off, mm);
  st4(A.v + off, vv);
  if (A.lin != nullptr && gl_lane == 0) {
    A.lin[row] = adam_elem(lw, sgl * A.lin_scale[f], lm, lv, coef);
    A.lin_m[row] = lm;
    A.lin_v[row] = lv;
  }
}

template <int LPR>
__device__ __forceinline__ void fm_rows_apply(const FmRowsArgs& A, int32_t row, int f, int n, int c4,
                                              int gl_lane, float4 E, float sgl, const AdamCoef& coef, int32_t run) {
  fm_rows_finish<LPR>(A, row, f, n, c4, gl_lane, E, sgl, coef, fm_rows_load<LPR>(A, row, c4, gl_lane), run);
}

template <int LPR, bool EARLY>
__device__ __forceinline__ void fm_rows_short(const FmRowsArgs& A, const AdamCoef& coef, int bid,
                                              int nblocks) {
  constexpr int K = LPR * 4;
  const int n_seg = *A.n_seg;
  const int64_t gtid = static_cast<int64_t>(bid) * kBlock + threadIdx.x;
  const int gl = static_cast<int>(gtid % LPR);
  const int c4 = gl * 4;
  const int64_t ngroups = static_cast<int64_t>(nblocks) * kBlock / LPR;
  for (int64_t s = gtid / LPR; s < n_seg; s += ngroups) {
    const int a0 = A.seg_start[s], a1 = A.seg_start[s + 1];
    if (a1 - a0 > kLongSeg) continue;
    float4 E = f4_zero();
    float sgl = 0.f;
    // first chunk of positions; the row and its operands are requested before the walk so that their
    // latency runs beside the walk's instead of behind it
    const int n0 = (a1 - a0) < LPR ? (a1 - a0) : LPR;
    int32_t qmine = A.seg_pos[a0 + (gl < n0 ? gl : 0)];
    const int32_t q_first = __shfl(qmine, 0, LPR);
    const int32_t row = A.slots != nullptr ? A.slots[q_first] : A.seg_rows[s];
    const int f = q_first % A.F;
    FmRowOperands ops_;
    if (EARLY) ops_ = fm_rows_load<LPR>(A, row, c4, gl);
    for (int base = a0; base < a1; base += LPR) {
      const int nq = (a1 - base) < LPR ? (a1 - base) : LPR;
      if (base != a0) qmine = A.seg_pos[base + (gl < nq ? gl : 0)];
      const float glm = A.gl != nullptr ? A.gl[qmine / A.F] : 0.f;
#pragma unroll 4
      for (int i = 0; i < nq; ++i) {
        E = f4_add(E, ld4(A.ge + static_cast<int64_t>(base + i) * K + c4));
        sgl += __shfl(glm, i, LPR);
      }
    }
    if (!EARLY) ops_ = fm_rows_load<LPR>(A, row, c4, gl);
    fm_rows_finish<LPR>(A, row, f, a1 - a0, c4, gl, E, sgl, coef, ops_, static_cast<int32_t>(s));
  }
}

template <int LPR>
__device__ __forceinline__ void fm_rows_long(const FmRowsArgs& A, const AdamCoef& coef, int bid,
                                             int nblocks) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  __shared__ float4 red[NG][LPR];
  __shared__ float redl[NG];
  const int n_long = *A.long_count;
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR, c4 = gl * 4;
  for (int li = bid; li < n_long; li += nblocks) {
    const int32_t s = A.long_list[li];
    const int p0 = A.seg_start[s], p1 = A.seg_start[s + 1];
    float4 E = f4_zero();
    float sgl = 0.f;
    for (int base = p0 + grp * LPR; base < p1; base += NG * LPR) {
      const int nq = (p1 - base) < LPR ? (p1 - base) : LPR;
      const int32_t qmine = A.seg_pos[base + (gl < nq ? gl : 0)];
      const float glm = A.gl != nullptr ? A.gl[qmine / A.F] : 0.f;
#pragma unroll 4
      for (int i = 0; i < nq; ++i) {
        E = f4_add(E, ld4(A.ge + static_cast<int64_t>(base + i) * K + c4));
        sgl += __shfl(glm, i, LPR);
      }
    }
    red[grp][gl] = E;
    if (gl == 0) redl[grp] = sgl;
    __syncthreads();
    if (grp == 0) {
      float4 t = f4_zero();
      float tl = 0.f;
#pragma unroll 4
      for (int g = 0; g < NG; ++g) {  // fixed order
        t = f4_add(t, red[g][gl]);
        tl += redl[g];
      }
      const int32_t row = A.slots != nullptr ? A.slots[A.seg_pos[p0]] : A.seg_rows[s];
      fm_rows_apply<LPR>(A, row, A.seg_pos[p0] % A.F, p1 - p0, c4, gl, t, tl, coef, s);
    }
    __syncthreads();
  }
}

template <int LPR, bool EARLY>
__global__ __launch_bounds__(kBlock) void fm_rows_adam_kernel(FmRowsArgs A, AdamCoef coef_arg) {
  // hipGraph replays freeze kernel arguments: a captured training step reads the step-dependent
  // coefficients (bias corrections, decayed learning rate) from a device buffer instead
  const AdamCoef coef = A.coef_dev != nullptr ? *A.coef_dev : coef_arg;
  if (blockIdx.x < kLongBlocks)
    fm_rows_long<LPR>(A, coef, blockIdx.x, kLongBlocks);
  else
    fm_rows_short<LPR, EARLY>(A, coef, blockIdx.x - kLongBlocks, gridDim.x - kLongBlocks);
}

template <int LPR, bool GATHER>
static int launch_fm_fwd(const float* src, int64_t V, const int32_t* idx, int64_t B, int F,
                         float* e, float* pair, float* fsum, const float* lin, float* lin_out,
                         hipStream_t s) {
  const int grid = grid_for(B, kBlock / kWave, kNumCU * 8);
  hipLaunchKernelGGL((fm_fwd_kernel<LPR, GATHER>), dim3(grid), dim3(kBlock), 0, s, src, V, idx,
                     B, F, e, pair, fsum, lin, lin_out);
  return launch_status();
}

template <bool GATHER>
static int dispatch_fm_fwd(const float* src, int64_t V, int K, const int32_t* idx, int64_t B,
                           int F, float* e, float* pair, float* fsum, const float* lin,
                           float* lin_out, hipStream_t s) {
  switch (K) {
    case 16: return launch_fm_fwd<4, GATHER>(src, V, idx, B, F, e, pair, fsum, lin, lin_out, s);
    case 32: return launch_fm_fwd<8, GATHER>(src, V, idx, B, F, e, pair, fsum, lin, lin_out, s);
    case 64: return launch_fm_fwd<16, GATHER>(src, V, idx, B, F, e, pair, fsum, lin, lin_out, s);
    case 128: return launch_fm_fwd<32, GATHER>(src, V, idx, B, F, e, pair, fsum, lin, lin_out, s);
    case 256: return launch_fm_fwd<64, GATHER>(src, V, idx, B, F, e, pair, fsum, lin, lin_out, s);
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
    int rc = dispatch_fm_fwd<false>(e, 0, K, nullptr, B, F, nullptr, pair, fsum, nullptr, nullptr, s);
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

extern "C" int lr_fm_embed_fwd_f32(const float* table, const float* lin, int64_t V, int K,
                                   const int32_t* idx, int64_t B, int F, float* e, float* pair,
                                   float* fsum, float* lin_out, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && pair);
  LR_CHECK_ARG(al16(table) && al16(pair) && (!e || al16(e)) && (!fsum || al16(fsum)));
  LR_CHECK_ARG((lin == nullptr) == (lin_out == nullptr));
  return dispatch_fm_fwd<true>(table, V, K, idx, B, F, e, pair, fsum, lin, lin_out,
                               as_stream(stream));
}

extern "C" size_t lr_fm_embed_bwd_ws_bytes(int64_t B, int F) {
  if (B < 0 || F < 1) return 0;
  // counter (padded) + one int32 per possible long run
  return 256 + static_cast<size_t>(B * F / lr::kLongSeg + 1) * sizeof(int32_t);
}

static int fm_bwd_launch(FmBwdArgs A, int K, int64_t B, int F, const AdamCoef& coef, void* ws,
                         size_t ws_bytes, hipStream_t s) {
  if (B * F >= (int64_t(1) << 31)) return LR_ESHAPE;
  if (ws == nullptr || ws_bytes < lr_fm_embed_bwd_ws_bytes(B, F)) return LR_EWORKSPACE;
  A.long_count = static_cast<int32_t*>(ws);
  A.long_list = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + 256);
  A.F = F;
  A.B = B;
  zero_words_async(A.long_count, 1, s);
  const int64_t n_max = B * F;
#define LR_FMB(LPR)                                                                            \
  {                                                                                            \
    const int grid = grid_for(n_max, kBlock / LPR);                                            \
    hipLaunchKernelGGL(fm_bwd_classify_kernel, dim3(grid_for(n_max, kBlock, kNumCU * 4)),      \
                       dim3(kBlock), 0, s, A.seg_start, A.n_seg, A.long_count, A.long_list);   \
    hipLaunchKernelGGL((fm_bwd_adam_kernel<LPR>), dim3(grid + kLongBlocks), dim3(kBlock), 0, s, \
                       A, coef);                                                               \
    return launch_status();                                                                    \
  }
  if (K == 16) LR_FMB(4)
  if (K == 32) LR_FMB(8)
  if (K == 64) LR_FMB(16)
  if (K == 128) LR_FMB(32)
#undef LR_FMB
  return LR_ESHAPE;
}

extern "C" int lr_fm_embed_bwd_adam_f32(float* table, float* m, float* v, float* lin, float* lin_m,
                                        float* lin_v, int64_t V, int K, const float* gdeep,
                                        const float* gpair, const float* fsum, const float* glin,
                                        const float* bn_a, const float* bn_c, int64_t B, int F,
                                        const int32_t* seg_pos, const int32_t* seg_rows,
                                        const int32_t* seg_start, const int32_t* n_seg,
                                        lr_adam_hp hp, void* ws, size_t ws_bytes,
                                        lr_stream_t stream) {
  LR_CHECK_ARG(V >= 0 && B >= 0 && F >= 1 && K >= 1 && hp.step >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && gpair && fsum && seg_pos && seg_rows && seg_start && n_seg);
  LR_CHECK_ARG(al16(table) && al16(m) && al16(v) && al16(gpair) && al16(fsum) &&
               (!gdeep || al16(gdeep)) && (!bn_a || al16(bn_a)) && (!bn_c || al16(bn_c)));
  LR_CHECK_ARG((lin == nullptr) == (lin_m == nullptr) && (lin == nullptr) == (lin_v == nullptr));
  LR_CHECK_ARG((bn_a == nullptr) == (bn_c == nullptr));
  LR_CHECK_ARG(glin == nullptr || lin != nullptr);
  FmBwdArgs A{};
  A.table = table; A.m = m; A.v = v; A.lin = lin; A.lin_m = lin_m; A.lin_v = lin_v;
  A.gdeep = gdeep; A.gpair = gpair; A.fsum = fsum; A.glin = glin; A.bn_a = bn_a; A.bn_c = bn_c;
  A.seg_pos = seg_pos; A.seg_rows = seg_rows; A.seg_start = seg_start; A.n_seg = n_seg;
  A.grows_out = nullptr; A.glin_out = nullptr;
  return fm_bwd_launch(A, K, B, F, make_adam_coef(hp), ws, ws_bytes, as_stream(stream));
}

extern "C" int lr_fm_embed_bwd_rows_f32(const float* row_cache, int K, const float* gdeep,
                                        const float* gpair, const float* fsum, const float* glin,
                                        const float* bn_a, const float* bn_c, int64_t B, int F,
                                        const int32_t* seg_pos, const int32_t* seg_start,
                                        const int32_t* n_seg, float* grows_out, float* glin_out,
                                        void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(row_cache && gpair && fsum && seg_pos && seg_start && n_seg && grows_out);
  LR_CHECK_ARG(al16(row_cache) && al16(gpair) && al16(fsum) && al16(grows_out) &&
               (!gdeep || al16(gdeep)) && (!bn_a || al16(bn_a)) && (!bn_c || al16(bn_c)));
  LR_CHECK_ARG((bn_a == nullptr) == (bn_c == nullptr));
  LR_CHECK_ARG((glin == nullptr) == (glin_out == nullptr));
  FmBwdArgs A{};
  A.table = const_cast<float*>(row_cache);  // read-only in rows mode
  A.gdeep = gdeep; A.gpair = gpair; A.fsum = fsum; A.glin = glin; A.bn_a = bn_a; A.bn_c = bn_c;
  A.seg_pos = seg_pos; A.seg_rows = nullptr; A.seg_start = seg_start; A.n_seg = n_seg;
  A.grows_out = grows_out; A.glin_out = glin_out;
  lr_adam_hp hp{};
  hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999; hp.tf_style = 1;
  return fm_bwd_launch(A, K, B, F, make_adam_coef(hp), ws, ws_bytes, as_stream(stream));
}

static int fm_field_stats_impl(const float* table, int K, const int32_t* seg_rows,
                               const int32_t* seg_start, const int32_t* n_seg,
                               const int32_t* field_row_start, int F, int C, float* partial,
                               const int32_t* seg_pos, const int32_t* slots, lr_stream_t stream);

extern "C" int lr_fm_field_stats_f32(const float* table, int K, const int32_t* seg_rows,
                                     const int32_t* seg_start, const int32_t* n_seg,
                                     const int32_t* field_row_start, int F, int C, float* partial,
                                     lr_stream_t stream) {
  return fm_field_stats_impl(table, K, seg_rows, seg_start, n_seg, field_row_start, F, C, partial, nullptr, nullptr,
                             stream);
}

extern "C" int lr_fm_field_stats_slots_f32(const float* cache, int K, const int32_t* seg_rows,
                                           const int32_t* seg_start, const int32_t* n_seg,
                                           const int32_t* field_row_start, int F, int C, float* partial,
                                           const int32_t* seg_pos, const int32_t* slots, lr_stream_t stream) {
  LR_CHECK_ARG(seg_pos && slots);
  return fm_field_stats_impl(cache, K, seg_rows, seg_start, n_seg, field_row_start, F, C, partial, seg_pos, slots,
                             stream);
}

static int fm_field_stats_impl(const float* table, int K, const int32_t* seg_rows,
                               const int32_t* seg_start, const int32_t* n_seg,
                               const int32_t* field_row_start, int F, int C, float* partial,
                               const int32_t* seg_pos, const int32_t* slots, lr_stream_t stream) {
  LR_CHECK_ARG(F >= 1 && C >= 1 && K >= 1);
  LR_CHECK_ARG(table && seg_rows && seg_start && n_seg && field_row_start && partial);
  LR_CHECK_ARG(al16(table) && al16(partial));
  hipStream_t s = as_stream(stream);
#define LR_FST(LPR)                                                                            \
  {                                                                                            \
    hipLaunchKernelGGL((fm_field_stats_kernel<LPR>), dim3(F * C), dim3(kBlock), 0, s, table,   \
                       seg_rows, seg_start, n_seg, field_row_start, C, partial, seg_pos, slots); \
    return launch_status();                                                                    \
  }
  if (K == 16) LR_FST(4)
  if (K == 32) LR_FST(8)
  if (K == 64) LR_FST(16)
  if (K == 128) LR_FST(32)
#undef LR_FST
  return LR_ESHAPE;
}

static int fm_rows_adam_impl(float* table, float* m, float* v, float* lin, float* lin_m,
                                   float* lin_v, int64_t V, int K, const float* ge, const float* gl,
                                   const float* wp, const float* bn_a, const float* bn_c,
                                   const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                                   const int32_t* seg_rows, const int32_t* seg_start,
                                   const int32_t* n_seg, lr_adam_hp hp, const void* coef_dev, void* ws,
                                   size_t ws_bytes, lr_stream_t stream, const int32_t* slots = nullptr,
                                   float* grows = nullptr, float* glin_rows = nullptr, bool compact = false) {
  const bool grad_mode = grows != nullptr;
  LR_CHECK_ARG(V >= 0 && B >= 0 && F >= 1 && K >= 1 && (grad_mode || coef_dev != nullptr || hp.step >= 1));
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && (grad_mode || (m && v)) && ge && seg_pos && seg_rows && seg_start && n_seg);
  if (grad_mode) {   // placeholders that pass the pointer checks below; never dereferenced in gradient mode
    m = v = table;
    if (lin != nullptr) lin_m = lin_v = lin;
    hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999;
    LR_CHECK_ARG((slots != nullptr) != compact && al16(grows) && (glin_rows == nullptr) == (lin == nullptr));
  }
  LR_CHECK_ARG(al16(table) && al16(m) && al16(v) && al16(ge) && (!wp || al16(wp)) &&
               (!bn_a || al16(bn_a)) && (!bn_c || al16(bn_c)));
  LR_CHECK_ARG((lin == nullptr) == (lin_m == nullptr) && (lin == nullptr) == (lin_v == nullptr));
  LR_CHECK_ARG((bn_a == nullptr) == (bn_c == nullptr) && (gl == nullptr) == (wp == nullptr));
  LR_CHECK_ARG(lin == nullptr || (lin_scale != nullptr && gl != nullptr));
  if (B * F >= (int64_t(1) << 31)) return LR_ESHAPE;
  if (ws == nullptr || ws_bytes < lr_fm_embed_bwd_ws_bytes(B, F)) return LR_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  FmRowsArgs A{};
  A.table = table; A.m = m; A.v = v; A.lin = lin; A.lin_m = lin_m; A.lin_v = lin_v;
  A.ge = ge; A.gl = gl; A.wp = wp; A.bn_a = bn_a; A.bn_c = bn_c; A.lin_scale = lin_scale;
  A.seg_pos = seg_pos; A.seg_rows = seg_rows; A.seg_start = seg_start; A.n_seg = n_seg;
  int32_t* long_count = static_cast<int32_t*>(ws);
  int32_t* long_list = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + 256);
  A.long_count = long_count; A.long_list = long_list; A.F = F;
  A.coef_dev = static_cast<const AdamCoef*>(coef_dev);
  A.slots = slots; A.grows = grows; A.glin_rows = glin_rows; A.compact = compact ? 1 : 0;
  zero_words_async(long_count, 1, s);
  const int64_t n_max = B * F;
  if (coef_dev != nullptr) { hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999; }
  const AdamCoef coef = make_adam_coef(hp);
  // Row operands are requested BEHIND the position walk.  Requesting them in front of it (the <LPR, true>
  // instantiation, a compile-time switch) shortens the dependent chain but measured slower in both modes (GPU call
  // r02m: Adam 1.05 vs 1.01 ms, gradient mode 0.72 vs 0.66 ms): the kernel is bound by the number of random
  // 128-byte requests in flight, and the early loads only compete with the walk's.
  constexpr bool early = false;
#define LR_FMR(LPR)                                                                            \
  {                                                                                            \
    const int grid = grid_for(n_max, kBlock / LPR);                                            \
    hipLaunchKernelGGL(fm_bwd_classify_kernel, dim3(grid_for(n_max, kBlock, kNumCU * 4)),      \
                       dim3(kBlock), 0, s, seg_start, n_seg, long_count, long_list);           \
    hipLaunchKernelGGL((fm_rows_adam_kernel<LPR, early>), dim3(grid + kLongBlocks), dim3(kBlock), 0, s, \
                       A, coef);                                                               \
    return launch_status();                                                                    \
  }
  if (K == 16) LR_FMR(4)
  if (K == 32) LR_FMR(8)
  if (K == 64) LR_FMR(16)
  if (K == 128) LR_FMR(32)
#undef LR_FMR
  return LR_ESHAPE;
}

extern "C" int lr_fm_rows_adam_f32(float* table, float* m, float* v, float* lin, float* lin_m,
                                   float* lin_v, int64_t V, int K, const float* ge, const float* gl,
                                   const float* wp, const float* bn_a, const float* bn_c,
                                   const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                                   const int32_t* seg_rows, const int32_t* seg_start,
                                   const int32_t* n_seg, lr_adam_hp hp, void* ws, size_t ws_bytes,
                                   lr_stream_t stream) {
  return fm_rows_adam_impl(table, m, v, lin, lin_m, lin_v, V, K, ge, gl, wp, bn_a, bn_c, lin_scale, B, F,
                           seg_pos, seg_rows, seg_start, n_seg, hp, nullptr, ws, ws_bytes, stream);
}

extern "C" int lr_fm_rows_adam_dc_f32(float* table, float* m, float* v, float* lin, float* lin_m,
                                      float* lin_v, int64_t V, int K, const float* ge, const float* gl,
                                      const float* wp, const float* bn_a, const float* bn_c,
                                      const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                                      const int32_t* seg_rows, const int32_t* seg_start,
                                      const int32_t* n_seg, const void* coef_dev, void* ws,
                                      size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(coef_dev != nullptr);
  lr_adam_hp hp{};
  return fm_rows_adam_impl(table, m, v, lin, lin_m, lin_v, V, K, ge, gl, wp, bn_a, bn_c, lin_scale, B, F,
                           seg_pos, seg_rows, seg_start, n_seg, hp, coef_dev, ws, ws_bytes, stream);
}

extern "C" int lr_fm_rows_grad_f32(const float* cache, const float* lin_cache, int64_t n_cache, int K,
                                   const float* ge, const float* gl, const float* wp, const float* bn_a,
                                   const float* bn_c, const float* lin_scale, int64_t B, int F,
                                   const int32_t* seg_pos, const int32_t* seg_rows,
                                   const int32_t* seg_start, const int32_t* n_seg, const int32_t* slots,
                                   float* grows, float* glin_rows, void* ws, size_t ws_bytes,
                                   lr_stream_t stream) {
  LR_CHECK_ARG(grows != nullptr && cache != nullptr);
  lr_adam_hp hp{};
  return fm_rows_adam_impl(const_cast<float*>(cache), nullptr, nullptr, const_cast<float*>(lin_cache), nullptr,
                           nullptr, n_cache, K, ge, gl, wp, bn_a, bn_c, lin_scale, B, F, seg_pos, seg_rows,
                           seg_start, n_seg, hp, nullptr, ws, ws_bytes, stream, slots, grows, glin_rows);
}

extern "C" int lr_fm_rows_grad_compact_f32(const float* table, const float* lin, int64_t V, int K, const float* ge,
                                           const float* gl, const float* wp, const float* bn_a, const float* bn_c,
                                           const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                                           const int32_t* seg_rows, const int32_t* seg_start, const int32_t* n_seg,
                                           float* grows, float* glin_rows, void* ws, size_t ws_bytes,
                                           lr_stream_t stream) {
  LR_CHECK_ARG(grows != nullptr && table != nullptr);
  lr_adam_hp hp{};
  return fm_rows_adam_impl(const_cast<float*>(table), nullptr, nullptr, const_cast<float*>(lin), nullptr, nullptr, V, K,
                           ge, gl, wp, bn_a, bn_c, lin_scale, B, F, seg_pos, seg_rows, seg_start, n_seg, hp, nullptr,
                           ws, ws_bytes, stream, nullptr, grows, glin_rows, true);
}
