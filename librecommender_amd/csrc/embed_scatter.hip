// Gradient scatter onto embedding rows: deterministic segment sums over the "CSR by touched
// row" built by lr_segments_build, optionally fused with a row-wise Adam update.
//
// Work decomposition: one row group of LPR = K/4 lanes owns one distinct row.  It walks the
// row's run of positions in ascending order (fixed summation order => run-to-run identical
// results, no fp32 atomics), 4 gradient rows in flight per step, then read-modify-writes
// w, m, v of that row exactly once.  n_seg is read from device memory: no host sync.
#include "common.hpp"

namespace lr {

template <int LPR>
__device__ __forceinline__ float4 seg_sum_rows(const float* __restrict__ grad,
                                               const int32_t* __restrict__ seg_pos, int p0,
                                               int p1, int lane) {
  constexpr int K = LPR * 4;
  float4 acc = f4_zero();
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    const int32_t q0 = seg_pos[p], q1 = seg_pos[p + 1], q2 = seg_pos[p + 2], q3 = seg_pos[p + 3];
    const float4 a = ld4(grad + static_cast<int64_t>(q0) * K + lane * 4);
    const float4 b = ld4(grad + static_cast<int64_t>(q1) * K + lane * 4);
    const float4 c = ld4(grad + static_cast<int64_t>(q2) * K + lane * 4);
    const float4 d = ld4(grad + static_cast<int64_t>(q3) * K + lane * 4);
    acc = f4_add(f4_add(f4_add(f4_add(acc, a), b), c), d);  // strictly ascending order
  }
  for (; p < p1; ++p)
    acc = f4_add(acc, ld4(grad + static_cast<int64_t>(seg_pos[p]) * K + lane * 4));
  return acc;
}

enum class SegMode { kSum, kAdd, kAdam };
constexpr int kLongRun = 32;      // runs longer than this are summed by whole workgroups (see seg_long_* below)
constexpr int kLongChunk = 512;

template <int LPR, SegMode MODE>
__global__ __launch_bounds__(kBlock) void seg_vec_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ grad, const int32_t* __restrict__ seg_pos,
    const int32_t* __restrict__ seg_rows, const int32_t* __restrict__ seg_start,
    const int32_t* __restrict__ n_seg_ptr, float* __restrict__ grows, float alpha,
    AdamCoef coef_arg, const AdamCoef* __restrict__ coef_dev, int skip_long) {
  constexpr int K = LPR * 4;
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;   // device-resident form: hipGraph-captured steps
  const int n_seg = *n_seg_ptr;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t s = gtid / LPR; s < n_seg; s += ngroups) {
    const int p0 = seg_start[s], p1 = seg_start[s + 1];
    if (skip_long && p1 - p0 > kLongRun) continue;      // summed by whole workgroups (seg_long_* kernels)
    const float4 g = seg_sum_rows<LPR>(grad, seg_pos, p0, p1, lane);
    if constexpr (MODE == SegMode::kSum) {
      st4(grows + s * K + lane * 4, g);
    } else {
      const int64_t off = static_cast<int64_t>(seg_rows[s]) * K + lane * 4;
      if constexpr (MODE == SegMode::kAdd) {
        st4(table + off, f4_fma(make_float4(alpha, alpha, alpha, alpha), g, ld4(table + off)));
      } else {
        float4 mm = ld4(m + off), vv = ld4(v + off);
        const float4 w = adam_vec(ld4(table + off), g, mm, vv, coef);
        st4(table + off, w);
        st4(m + off, mm);
        st4(v + off, vv);
      }
    }
  }
}

// ---- long runs (Zipf heads: one row receiving thousands of positions) -------------------------------------------
// A run of more than kLongRun positions is cut into chunks of kLongChunk positions; one workgroup per chunk sums its
// positions (row groups take positions g, g + NG, ... and are folded through LDS in group order), the chunk partials
// of a run are added in chunk order by one row group, which then applies the update.  Every order is fixed: results
// stay run-to-run identical.  Workspace layout (lr_embed_scatter_ws_bytes):
//   int32 long_count | int32 chunk_count | pad to 256 B | long_seg[NL] | long_base[NL] | chunk_slot[NC] | partial[NC][K]
struct LongWs {
  int32_t* counts;      // [0] = long runs, [1] = chunks
  int32_t* long_seg; int32_t* long_base; int32_t* chunk_slot; float* partial;
  int n_long_max, n_chunk_max;
};

__global__ __launch_bounds__(kBlock) void seg_long_classify_kernel(const int32_t* __restrict__ seg_start,
                                                                   const int32_t* __restrict__ n_seg_ptr, LongWs w) {
  const int n_seg = *n_seg_ptr;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; s < n_seg; s += stride) {
    const int len = seg_start[s + 1] - seg_start[s];
    if (len <= kLongRun) continue;
    const int nch = (len + kLongChunk - 1) / kLongChunk;
    const int slot = atomicAdd(&w.counts[0], 1);
    const int base = atomicAdd(&w.counts[1], nch);
    w.long_seg[slot] = static_cast<int32_t>(s);
    w.long_base[slot] = base;
    for (int j = 0; j < nch; ++j) w.chunk_slot[base + j] = slot;
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void seg_long_chunk_kernel(const float* __restrict__ grad,
                                                                const int32_t* __restrict__ seg_pos,
                                                                const int32_t* __restrict__ seg_start, LongWs w) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  __shared__ float4 red[NG][LPR];
  const int lane = threadIdx.x % LPR, g = threadIdx.x / LPR;
  const int n_chunks = w.counts[1];
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int slot = w.chunk_slot[c];
    const int s = w.long_seg[slot];
    const int j = c - w.long_base[slot];
    const int p0 = seg_start[s] + j * kLongChunk;
    const int pe = seg_start[s + 1];
    const int p1 = (p0 + kLongChunk) < pe ? (p0 + kLongChunk) : pe;
    // positions g, g + NG, ...: four independent rows in flight, added in ascending order
    float4 acc = f4_zero();
    int p = p0 + g;
    for (; p + 3 * NG < p1; p += 4 * NG) {
      const int32_t q0 = seg_pos[p], q1 = seg_pos[p + NG], q2 = seg_pos[p + 2 * NG], q3 = seg_pos[p + 3 * NG];
      const float4 a = ld4(grad + static_cast<int64_t>(q0) * K + lane * 4);
      const float4 b = ld4(grad + static_cast<int64_t>(q1) * K + lane * 4);
      const float4 c2 = ld4(grad + static_cast<int64_t>(q2) * K + lane * 4);
      const float4 d = ld4(grad + static_cast<int64_t>(q3) * K + lane * 4);
      acc = f4_add(f4_add(f4_add(f4_add(acc, a), b), c2), d);
    }
    for (; p < p1; p += NG)
      acc = f4_add(acc, ld4(grad + static_cast<int64_t>(seg_pos[p]) * K + lane * 4));
    red[g][lane] = acc;
    __syncthreads();
    if (g == 0) {
      float4 t = red[0][lane];
      for (int k = 1; k < NG; ++k) t = f4_add(t, red[k][lane]);
      st4(w.partial + static_cast<int64_t>(c) * K + lane * 4, t);
    }
    __syncthreads();
  }
}

template <int LPR, SegMode MODE>
__global__ __launch_bounds__(kBlock) void seg_long_finish_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v, const int32_t* __restrict__ seg_rows,
    const int32_t* __restrict__ seg_start, float* __restrict__ grows, float alpha, AdamCoef coef_arg,
    const AdamCoef* __restrict__ coef_dev, LongWs w) {
  constexpr int K = LPR * 4;
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;
  const int n_long = w.counts[0];
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t q = gtid / LPR; q < n_long; q += ngroups) {
    const int s = w.long_seg[q];
    const int base = w.long_base[q];
    const int nch = (seg_start[s + 1] - seg_start[s] + kLongChunk - 1) / kLongChunk;
    float4 g = f4_zero();
    const float* pp = w.partial + static_cast<int64_t>(base) * K + lane * 4;
    int j = 0;
    for (; j + 4 <= nch; j += 4) {         // four partials in flight, added in chunk order
      const float4 a = ld4(pp + static_cast<int64_t>(j) * K), b = ld4(pp + static_cast<int64_t>(j + 1) * K);
      const float4 c2 = ld4(pp + static_cast<int64_t>(j + 2) * K), d = ld4(pp + static_cast<int64_t>(j + 3) * K);
      g = f4_add(f4_add(f4_add(f4_add(g, a), b), c2), d);
    }
    for (; j < nch; ++j) g = f4_add(g, ld4(pp + static_cast<int64_t>(j) * K));
    if constexpr (MODE == SegMode::kSum) {
      st4(grows + static_cast<int64_t>(s) * K + lane * 4, g);
    } else {
      const int64_t off = static_cast<int64_t>(seg_rows[s]) * K + lane * 4;
      if constexpr (MODE == SegMode::kAdd) {
        st4(table + off, f4_fma(make_float4(alpha, alpha, alpha, alpha), g, ld4(table + off)));
      } else {
        float4 mm = ld4(m + off), vv = ld4(v + off);
        const float4 wn = adam_vec(ld4(table + off), g, mm, vv, coef);
        st4(table + off, wn);
        st4(m + off, mm);
        st4(v + off, vv);
      }
    }
  }
}

static inline size_t sc_align(size_t x) { return (x + 255) / 256 * 256; }
static inline int long_max(int64_t n_max) { return static_cast<int>(n_max / kLongRun + 1); }
static inline int chunk_max(int64_t n_max) { return static_cast<int>(n_max / kLongChunk + n_max / kLongRun + 2); }
static LongWs make_long_ws(void* ws, int64_t n_max) {
  LongWs w;
  char* p = static_cast<char*>(ws);
  w.n_long_max = long_max(n_max);
  w.n_chunk_max = chunk_max(n_max);
  w.counts = reinterpret_cast<int32_t*>(p); p += 256;
  w.long_seg = reinterpret_cast<int32_t*>(p); p += sc_align(static_cast<size_t>(w.n_long_max) * 4);
  w.long_base = reinterpret_cast<int32_t*>(p); p += sc_align(static_cast<size_t>(w.n_long_max) * 4);
  w.chunk_slot = reinterpret_cast<int32_t*>(p); p += sc_align(static_cast<size_t>(w.n_chunk_max) * 4);
  w.partial = reinterpret_cast<float*>(p);
  return w;
}

// Row-wise Adam on a table AND its per-row linear weight from one pass over the segments (the owner-side
// update of the row-sharded tables: `embed` [V,K] + `lin` [V,1] share the received row ids).  Lane 0 of the
// row group sums the run's scalar gradients in the same ascending order.
struct LinAdam {
  float* lin; float* lin_m; float* lin_v;
  const float* glin;
};
template <int LPR>
__global__ __launch_bounds__(kBlock) void seg_adam_lin_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ grad,
    const int32_t* __restrict__ seg_pos, const int32_t* __restrict__ seg_rows,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ n_seg_ptr, LinAdam L, AdamCoef coef_arg,
    const AdamCoef* __restrict__ coef_dev) {
  constexpr int K = LPR * 4;
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;
  const int n_seg = *n_seg_ptr;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t s = gtid / LPR; s < n_seg; s += ngroups) {
    const int p0 = seg_start[s], p1 = seg_start[s + 1];
    const float4 g = seg_sum_rows<LPR>(grad, seg_pos, p0, p1, lane);
    const int32_t row = seg_rows[s];
    const int64_t off = static_cast<int64_t>(row) * K + lane * 4;
    float4 mm = ld4(m + off), vv = ld4(v + off);
    const float4 w = adam_vec(ld4(table + off), g, mm, vv, coef);
    st4(table + off, w);
    st4(m + off, mm);
    st4(v + off, vv);
    // the run's scalar gradients: LPR positions per step loaded side by side, added in ascending order
    float gs = 0.f;
    for (int base = p0; base < p1; base += LPR) {
      const int nq = (p1 - base) < LPR ? (p1 - base) : LPR;
      const float x = lane < nq ? L.glin[seg_pos[base + lane]] : 0.f;
#pragma unroll 4
      for (int i = 0; i < nq; ++i) gs += __shfl(x, i, LPR);
    }
    if (lane == 0) {
      float lm = L.lin_m[row], lv = L.lin_v[row];
      L.lin[row] = adam_elem(L.lin[row], gs, lm, lv, coef);
      L.lin_m[row] = lm;
      L.lin_v[row] = lv;
    }
  }
}

// Generic K: one thread per (segment, column).
template <SegMode MODE>
__global__ __launch_bounds__(kBlock) void seg_scalar_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v, int K,
    const float* __restrict__ grad, const int32_t* __restrict__ seg_pos,
    const int32_t* __restrict__ seg_rows, const int32_t* __restrict__ seg_start,
    const int32_t* __restrict__ n_seg_ptr, float* __restrict__ grows, float alpha,
    AdamCoef coef_arg, const AdamCoef* __restrict__ coef_dev) {
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;
  const int64_t total = static_cast<int64_t>(*n_seg_ptr) * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t s = e / K;
    const int c = static_cast<int>(e - s * K);
    float g = 0.f;
    for (int p = seg_start[s]; p < seg_start[s + 1]; ++p)
      g += grad[static_cast<int64_t>(seg_pos[p]) * K + c];
    if constexpr (MODE == SegMode::kSum) {
      grows[e] = g;
    } else {
      const int64_t off = static_cast<int64_t>(seg_rows[s]) * K + c;
      if constexpr (MODE == SegMode::kAdd) {
        table[off] = fmaf(alpha, g, table[off]);
      } else {
        float mm = m[off], vv = v[off];
        table[off] = adam_elem(table[off], g, mm, vv, coef);
        m[off] = mm;
        v[off] = vv;
      }
    }
  }
}

template <SegMode MODE>
static int launch_seg(float* table, float* m, float* v, int K, const float* grad,
                      const int32_t* seg_pos, const int32_t* seg_rows, const int32_t* seg_start,
                      const int32_t* n_seg, int64_t n_max, float* grows, float alpha,
                      AdamCoef coef, hipStream_t s, const AdamCoef* coef_dev = nullptr, void* ws = nullptr,
                      size_t ws_bytes = 0) {
  bool aligned = reinterpret_cast<uintptr_t>(grad) % 16 == 0;
  if (table) aligned = aligned && reinterpret_cast<uintptr_t>(table) % 16 == 0;
  if (m) aligned = aligned && reinterpret_cast<uintptr_t>(m) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(v) % 16 == 0;
  if (grows) aligned = aligned && reinterpret_cast<uintptr_t>(grows) % 16 == 0;
  const bool use_long = ws != nullptr && aligned && n_max > kLongRun;
  if (ws != nullptr && ws_bytes < lr_embed_scatter_ws_bytes(n_max, K)) return LR_EWORKSPACE;
#define LR_SEG(LPR)                                                                          \
  {                                                                                          \
    const int grid = grid_for(n_max, kBlock / LPR);                                          \
    LongWs w{};                                                                              \
    if (use_long) {                                                                          \
      w = make_long_ws(ws, n_max);                                                           \
      zero_words_async(w.counts, 2, s);                                                      \
      hipLaunchKernelGGL(seg_long_classify_kernel, dim3(grid_for(n_max, kBlock, kNumCU * 2)), dim3(kBlock), 0, s, \
                         seg_start, n_seg, w);                                               \
    }                                                                                        \
    hipLaunchKernelGGL((seg_vec_kernel<LPR, MODE>), dim3(grid), dim3(kBlock), 0, s, table, m, \
                       v, grad, seg_pos, seg_rows, seg_start, n_seg, grows, alpha, coef, coef_dev, use_long ? 1 : 0); \
    if (use_long) {                                                                          \
      const int gc = w.n_chunk_max < kNumCU * 4 ? w.n_chunk_max : kNumCU * 4;                \
      hipLaunchKernelGGL((seg_long_chunk_kernel<LPR>), dim3(gc), dim3(kBlock), 0, s, grad, seg_pos, seg_start, w); \
      hipLaunchKernelGGL((seg_long_finish_kernel<LPR, MODE>), dim3(grid_for(w.n_long_max, kBlock / LPR)), dim3(kBlock), \
                         0, s, table, m, v, seg_rows, seg_start, grows, alpha, coef, coef_dev, w); \
    }                                                                                        \
    return launch_status();                                                                  \
  }
  if (aligned) {
    if (K == 16) LR_SEG(4)
    if (K == 32) LR_SEG(8)
    if (K == 64) LR_SEG(16)
    if (K == 128) LR_SEG(32)
  }
#undef LR_SEG
  const int grid = grid_for(n_max * K, kBlock);
  hipLaunchKernelGGL((seg_scalar_kernel<MODE>), dim3(grid), dim3(kBlock), 0, s, table, m, v, K,
                     grad, seg_pos, seg_rows, seg_start, n_seg, grows, alpha, coef, coef_dev);
  return launch_status();
}

// ---- dense (TF1-semantics) Adam --------------------------------------------------------
__global__ __launch_bounds__(kBlock) void mark_slots_kernel(const int32_t* __restrict__ seg_rows,
                                                            const int32_t* __restrict__ n_seg_ptr,
                                                            int32_t* __restrict__ row_slot) {
  const int n_seg = *n_seg_ptr;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; s < n_seg; s += stride)
    row_slot[seg_rows[s]] = static_cast<int32_t>(s);
}

__global__ __launch_bounds__(kBlock) void adam_dense_kernel(
    float* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ vmax, int64_t V, int K, const float* __restrict__ grows,
    int32_t* __restrict__ row_slot, int dense_grad, float l2, AdamCoef coef_arg,
    const AdamCoef* __restrict__ coef_dev) {
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;   // see fm_rows_adam_kernel
  const int64_t total = V * K;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t r = e / K;
    const int c = static_cast<int>(e - r * K);
    const int32_t slot = row_slot ? row_slot[r] : -1;
    const float w = table[e];
    float g = dense_grad ? grows[e] : (slot >= 0 ? grows[static_cast<int64_t>(slot) * K + c] : 0.f);
    g = fmaf(2.f * l2, w, g);
    float mm = m[e], vv = v[e];
    if (vmax == nullptr) {
      table[e] = adam_elem(w, g, mm, vv, coef);
    } else {  // amsgrad (torch.optim.Adam(amsgrad=True)): the denominator uses max_t v_t
      float unused = adam_elem(w, g, mm, vv, coef);
      (void)unused;
      const float vm = fmaxf(vmax[e], vv);
      vmax[e] = vm;
      const float denom = coef.tf_style ? sqrtf(vm) + coef.eps : sqrtf(vm) / coef.bc2_sqrt + coef.eps;
      table[e] = w - coef.step_size * (mm / denom);
    }
    m[e] = mm;
    v[e] = vv;
  }
}

// The TF1 table pass of the fused FM / DeepFM step: EVERY row of the embedding table and of the linear table decays its moments
// and moves (training/tf_trainer.py:120: tf.train.AdamOptimizer applies IndexedSlices gradients through _apply_sparse_shared, which
// still updates m, v and the variable of every row); the rows of this batch take their gradient from the compact per-run arrays
// grows [n_seg, K] / glin_rows [n_seg] through row_slot.  One launch, 16 bytes per lane and array, pure streaming.
template <int LPR>
__global__ __launch_bounds__(kBlock) void adam_rows_kernel(float* __restrict__ table, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ lin,
                                                           float* __restrict__ lin_m, float* __restrict__ lin_v, int64_t V,
                                                           const float* __restrict__ grows,
                                                           const float* __restrict__ glin_rows,
                                                           const int32_t* __restrict__ row_slot, AdamCoef coef_arg,
                                                           const AdamCoef* __restrict__ coef_dev) {
  constexpr int K = LPR * 4;
  constexpr int U = 2;                        // quads per thread and trip: 6 U streaming requests in flight per lane
  const AdamCoef coef = coef_dev != nullptr ? *coef_dev : coef_arg;
  const int64_t total = V * LPR;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  // every byte of (table, m, v) is read once and written once per step: streaming (non-temporal) accesses keep the pass out of
  // the way of what the rest of the step holds in L2 / Infinity Cache
  auto ldnt = [](const float* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    return make_float4(__builtin_nontemporal_load(&q->x), __builtin_nontemporal_load(&q->y),
                       __builtin_nontemporal_load(&q->z), __builtin_nontemporal_load(&q->w));
  };
  auto stnt = [](float* p, float4 x) {
    float4* q = reinterpret_cast<float4*>(p);
    __builtin_nontemporal_store(x.x, &q->x); __builtin_nontemporal_store(x.y, &q->y);
    __builtin_nontemporal_store(x.z, &q->z); __builtin_nontemporal_store(x.w, &q->w);
  };
  for (int64_t q0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q0 < total; q0 += U * stride) {
    int64_t off[U], r[U];
    int32_t slot[U];
    float4 w[U], mm[U], vv[U], g[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = q0 + u * stride;
      ok[u] = q < total;
      r[u] = ok[u] ? q / LPR : 0;
      const int c4 = ok[u] ? static_cast<int>(q - r[u] * LPR) * 4 : 0;
      off[u] = r[u] * K + c4;
      slot[u] = row_slot[r[u]];
      w[u] = ldnt(table + off[u]);
      mm[u] = ldnt(m + off[u]);
      vv[u] = ldnt(v + off[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      g[u] = slot[u] >= 0 ? ld4(grows + static_cast<int64_t>(slot[u]) * K + (off[u] - r[u] * K)) : f4_zero();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      stnt(table + off[u], adam_vec(w[u], g[u], mm[u], vv[u], coef));
      stnt(m + off[u], mm[u]);
      stnt(v + off[u], vv[u]);
      if (lin != nullptr && off[u] == r[u] * K) {
        float lm = lin_m[r[u]], lv = lin_v[r[u]];
        lin[r[u]] = adam_elem(lin[r[u]], slot[u] >= 0 ? glin_rows[slot[u]] : 0.f, lm, lv, coef);
        lin_m[r[u]] = lm;
        lin_v[r[u]] = lv;
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void clear_slots_kernel(const int32_t* __restrict__ seg_rows,
                                                             const int32_t* __restrict__ n_seg_ptr,
                                                             int32_t* __restrict__ row_slot) {
  const int n_seg = *n_seg_ptr;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; s < n_seg; s += stride)
    row_slot[seg_rows[s]] = -1;
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_embed_scatter_ws_bytes(int64_t n_max, int K) {
  if (n_max < 0 || K < 1) return 0;
  return 256 + 2 * sc_align(static_cast<size_t>(long_max(n_max)) * 4) + sc_align(static_cast<size_t>(chunk_max(n_max)) * 4) +
         sc_align(static_cast<size_t>(chunk_max(n_max)) * K * 4);
}

extern "C" int lr_embed_segment_sum_f32(const float* grad, int K, const int32_t* seg_pos,
                                        const int32_t* seg_start, const int32_t* n_seg,
                                        int64_t n_max, float* grows, void* ws, size_t ws_bytes,
                                        lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && K >= 1 && n_max >= 0);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(grad && seg_pos && grows);
  return launch_seg<SegMode::kSum>(nullptr, nullptr, nullptr, K, grad, seg_pos, nullptr,
                                   seg_start, n_seg, n_max, grows, 0.f, AdamCoef{},
                                   as_stream(stream), nullptr, ws, ws_bytes);
}

extern "C" int lr_embed_scatter_add_f32(float* table, int64_t V, int K, const float* grad,
                                        const int32_t* seg_pos, const int32_t* seg_rows,
                                        const int32_t* seg_start, const int32_t* n_seg,
                                        int64_t n_max, float alpha, void* ws, size_t ws_bytes,
                                        lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && K >= 1 && n_max >= 0 && V >= 0);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(table && grad && seg_pos && seg_rows);
  return launch_seg<SegMode::kAdd>(table, nullptr, nullptr, K, grad, seg_pos, seg_rows,
                                   seg_start, n_seg, n_max, nullptr, alpha, AdamCoef{},
                                   as_stream(stream), nullptr, ws, ws_bytes);
}

extern "C" int lr_embed_scatter_adam_f32(float* table, float* m, float* v, int64_t V, int K,
                                         const float* grad, const int32_t* seg_pos,
                                         const int32_t* seg_rows, const int32_t* seg_start,
                                         const int32_t* n_seg, int64_t n_max, lr_adam_hp hp,
                                         void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && K >= 1 && n_max >= 0 && V >= 0 && hp.step >= 1);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && grad && seg_pos && seg_rows);
  return launch_seg<SegMode::kAdam>(table, m, v, K, grad, seg_pos, seg_rows, seg_start, n_seg,
                                    n_max, nullptr, 0.f, make_adam_coef(hp), as_stream(stream), nullptr, ws, ws_bytes);
}

static int scatter_adam_lin_impl(float* table, float* m, float* v, int64_t V, int K, const float* grad, float* lin,
                                 float* lin_m, float* lin_v, const float* glin, const int32_t* seg_pos,
                                 const int32_t* seg_rows, const int32_t* seg_start, const int32_t* n_seg,
                                 int64_t n_max, lr_adam_hp hp, const AdamCoef* coef_dev, lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && K >= 1 && n_max >= 0 && V >= 0 && hp.step >= 1);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && grad && seg_pos && seg_rows && lin && lin_m && lin_v && glin);
  const bool aligned = reinterpret_cast<uintptr_t>(grad) % 16 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(m) % 16 == 0 && reinterpret_cast<uintptr_t>(v) % 16 == 0;
  const AdamCoef coef = make_adam_coef(hp);
  hipStream_t s = as_stream(stream);
  if (!aligned || !(K == 16 || K == 32 || K == 64 || K == 128)) {     // two launches of the general kernels
    int rc = launch_seg<SegMode::kAdam>(table, m, v, K, grad, seg_pos, seg_rows, seg_start, n_seg, n_max, nullptr, 0.f,
                                        coef, s, coef_dev);
    if (rc != LR_OK) return rc;
    return launch_seg<SegMode::kAdam>(lin, lin_m, lin_v, 1, glin, seg_pos, seg_rows, seg_start, n_seg, n_max, nullptr,
                                      0.f, coef, s, coef_dev);
  }
  const LinAdam L{lin, lin_m, lin_v, glin};
#define LR_SAL(LPR)                                                                                   \
  {                                                                                                   \
    hipLaunchKernelGGL((seg_adam_lin_kernel<LPR>), dim3(grid_for(n_max, kBlock / LPR)), dim3(kBlock), 0, s, \
                       table, m, v, grad, seg_pos, seg_rows, seg_start, n_seg, L, coef, coef_dev);    \
    return launch_status();                                                                           \
  }
  if (K == 16) LR_SAL(4)
  if (K == 32) LR_SAL(8)
  if (K == 64) LR_SAL(16)
  LR_SAL(32)
#undef LR_SAL
}

// ---- owner-side update from the peers' de-duplicated lists (row-sharded tables) ------------------------------
// After the gradient all-to-all the owner holds W lists back to back, list p = the rows peer p touched, each row at most
// ONCE per list.  A row therefore collects at most W gradient rows, one per peer, and "group by row" needs no sort:
//   peer_mark_kernel  : tab[row * W + p] = 1 + (index of the row in the concatenated lists)        (W > 1 only)
//   peer_adam_kernel  : one lane group per received entry; the entry of the LOWEST peer holding the row leads: it adds
//                       the row's gradients in ascending peer order (the order the sorted-segment path adds them:
//                       same bits), applies Adam to w / m / v and to the row's linear weight
//   peer_clear_kernel : tab entries back to 0                                                     (W > 1 only)
// With one list (W == 1) every entry leads and no table is touched.  Replaces a 4-pass radix sort + scan per step.
struct PeerLists {
  int32_t begin[65];            // list p = entries [begin[p], begin[p + 1])
  int W;
};
__device__ __forceinline__ int peer_of(const PeerLists& P, int i) {
  int p = 0;
  while (p + 1 < P.W && i >= P.begin[p + 1]) ++p;
  return p;
}
__global__ __launch_bounds__(kBlock) void peer_mark_kernel(const int32_t* __restrict__ ids, PeerLists P, int32_t* __restrict__ tab) {
  const int n = P.begin[P.W];
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    tab[static_cast<int64_t>(ids[i]) * P.W + peer_of(P, i)] = i + 1;
}
__global__ __launch_bounds__(kBlock) void peer_clear_kernel(const int32_t* __restrict__ ids, PeerLists P, int32_t* __restrict__ tab) {
  const int n = P.begin[P.W];
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    tab[static_cast<int64_t>(ids[i]) * P.W + peer_of(P, i)] = 0;
}
template <int LPR, bool kLin>
__global__ __launch_bounds__(kBlock) void peer_adam_kernel(float* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
                                                           const float* __restrict__ grad, const int32_t* __restrict__ ids,
                                                           PeerLists P, const int32_t* __restrict__ tab, LinAdam L,
                                                           AdamCoef coef) {
  constexpr int K = LPR * 4;
  const int n = P.begin[P.W];
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int lane = static_cast<int>(gtid % LPR);
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * kBlock / LPR;
  for (int64_t i = gtid / LPR; i < n; i += ngroups) {
    const int32_t row = ids[i];
    float4 g;
    float gs = 0.f;
    if (P.W == 1) {
      g = ld4(grad + i * K + lane * 4);
      if (kLin) gs = L.glin[i];
    } else {
      const int32_t* t = tab + static_cast<int64_t>(row) * P.W;
      const int p = peer_of(P, static_cast<int>(i));
      bool lead = true;
      for (int q = 0; q < p; ++q) lead = lead && t[q] == 0;
      if (!lead) continue;
      g = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = p; q < P.W; ++q) {
        const int j = t[q];
        if (j == 0) continue;
        const float4 x = ld4(grad + static_cast<int64_t>(j - 1) * K + lane * 4);
        g.x += x.x; g.y += x.y; g.z += x.z; g.w += x.w;
        if (kLin) gs += L.glin[j - 1];
      }
    }
    const int64_t off = static_cast<int64_t>(row) * K + lane * 4;
    float4 mm = ld4(m + off), vv = ld4(v + off);
    const float4 w = adam_vec(ld4(table + off), g, mm, vv, coef);
    st4(table + off, w);
    st4(m + off, mm);
    st4(v + off, vv);
    if (kLin && lane == 0) {
      float lm = L.lin_m[row], lv = L.lin_v[row];
      L.lin[row] = adam_elem(L.lin[row], gs, lm, lv, coef);
      L.lin_m[row] = lm;
      L.lin_v[row] = lv;
    }
  }
}

extern "C" int lr_embed_peer_adam_f32(float* table, float* m, float* v, int64_t V, int K, const float* grad, float* lin,
                                      float* lin_m, float* lin_v, const float* glin, const int32_t* ids,
                                      const int64_t* peer_counts, int W, int32_t* peer_tab, lr_adam_hp hp,
                                      lr_stream_t stream) {
  LR_CHECK_ARG(peer_counts && W >= 1 && V >= 0 && hp.step >= 1);
  if (W > 64 || !(K == 16 || K == 32 || K == 64 || K == 128)) return LR_ESHAPE;
  PeerLists P;
  P.W = W;
  int64_t n = 0;
  for (int p = 0; p < W; ++p) {
    LR_CHECK_ARG(peer_counts[p] >= 0);
    P.begin[p] = static_cast<int32_t>(n);
    n += peer_counts[p];
    if (n >= (int64_t(1) << 31) - 1) return LR_ESHAPE;
  }
  for (int p = W; p <= 64; ++p) P.begin[p] = static_cast<int32_t>(n);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && grad && ids && (W == 1 || peer_tab));
  const bool with_lin = lin != nullptr;
  LR_CHECK_ARG(!with_lin || (lin_m && lin_v && glin));
  if (reinterpret_cast<uintptr_t>(grad) % 16 || reinterpret_cast<uintptr_t>(table) % 16 || reinterpret_cast<uintptr_t>(m) % 16 ||
      reinterpret_cast<uintptr_t>(v) % 16)
    return LR_ESHAPE;
  const AdamCoef coef = make_adam_coef(hp);
  const LinAdam L{lin, lin_m, lin_v, glin};
  hipStream_t s = as_stream(stream);
  const dim3 g1(grid_for(n, kBlock));
  if (W > 1) hipLaunchKernelGGL(peer_mark_kernel, g1, dim3(kBlock), 0, s, ids, P, peer_tab);
#define LR_PA(LPR)                                                                                                        \
  {                                                                                                                       \
    if (with_lin)                                                                                                         \
      hipLaunchKernelGGL((peer_adam_kernel<LPR, true>), dim3(grid_for(n, kBlock / LPR)), dim3(kBlock), 0, s, table, m, v, grad, \
                         ids, P, peer_tab, L, coef);                                                                      \
    else                                                                                                                  \
      hipLaunchKernelGGL((peer_adam_kernel<LPR, false>), dim3(grid_for(n, kBlock / LPR)), dim3(kBlock), 0, s, table, m, v, grad, \
                         ids, P, peer_tab, L, coef);                                                                      \
  }
  if (K == 16) LR_PA(4)
  else if (K == 32) LR_PA(8)
  else if (K == 64) LR_PA(16)
  else LR_PA(32)
#undef LR_PA
  if (W > 1) hipLaunchKernelGGL(peer_clear_kernel, g1, dim3(kBlock), 0, s, ids, P, peer_tab);
  return launch_status();
}

extern "C" int lr_embed_scatter_adam_lin_f32(float* table, float* m, float* v, int64_t V, int K,
                                             const float* grad, float* lin, float* lin_m, float* lin_v,
                                             const float* glin, const int32_t* seg_pos,
                                             const int32_t* seg_rows, const int32_t* seg_start,
                                             const int32_t* n_seg, int64_t n_max, lr_adam_hp hp,
                                             lr_stream_t stream) {
  return scatter_adam_lin_impl(table, m, v, V, K, grad, lin, lin_m, lin_v, glin, seg_pos, seg_rows, seg_start, n_seg,
                               n_max, hp, nullptr, stream);
}

static lr_adam_hp dc_placeholder_hp() {     // the kernel reads the coefficients from device memory
  lr_adam_hp hp{};
  hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999; hp.tf_style = 1;
  return hp;
}

extern "C" int lr_embed_scatter_adam_lin_dc_f32(float* table, float* m, float* v, int64_t V, int K,
                                                const float* grad, float* lin, float* lin_m, float* lin_v,
                                                const float* glin, const int32_t* seg_pos,
                                                const int32_t* seg_rows, const int32_t* seg_start,
                                                const int32_t* n_seg, int64_t n_max, const void* coef_dev,
                                                lr_stream_t stream) {
  LR_CHECK_ARG(coef_dev != nullptr);
  return scatter_adam_lin_impl(table, m, v, V, K, grad, lin, lin_m, lin_v, glin, seg_pos, seg_rows, seg_start, n_seg,
                               n_max, dc_placeholder_hp(), static_cast<const AdamCoef*>(coef_dev), stream);
}

extern "C" int lr_embed_scatter_adam_dc_f32(float* table, float* m, float* v, int64_t V, int K,
                                            const float* grad, const int32_t* seg_pos,
                                            const int32_t* seg_rows, const int32_t* seg_start,
                                            const int32_t* n_seg, int64_t n_max, const void* coef_dev,
                                            void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && K >= 1 && n_max >= 0 && V >= 0 && coef_dev != nullptr);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && grad && seg_pos && seg_rows);
  return launch_seg<SegMode::kAdam>(table, m, v, K, grad, seg_pos, seg_rows, seg_start, n_seg, n_max, nullptr, 0.f,
                                    make_adam_coef(dc_placeholder_hp()), as_stream(stream),
                                    static_cast<const AdamCoef*>(coef_dev), ws, ws_bytes);
}

extern "C" int lr_adam_dense_f32(float* table, float* m, float* v, float* vmax, int64_t V, int K,
                                 const float* grows, const int32_t* seg_rows,
                                 const int32_t* n_seg, int64_t n_max, int32_t* row_slot,
                                 float l2, lr_adam_hp hp, lr_stream_t stream) {
  LR_CHECK_ARG(table && m && v && V >= 0 && K >= 1 && n_max >= 0 && hp.step >= 1);
  if (V == 0) return LR_OK;
  hipStream_t s = as_stream(stream);
  const bool dense_grad = grows != nullptr && seg_rows == nullptr;  // grows is a full [V,K] gradient
  const bool sparse = n_max > 0 && !dense_grad;
  if (sparse) {
    LR_CHECK_ARG(grows && seg_rows && n_seg && row_slot);
    hipLaunchKernelGGL(mark_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s,
                       seg_rows, n_seg, row_slot);
  }
  hipLaunchKernelGGL(adam_dense_kernel, dim3(grid_for(V * K, kBlock)), dim3(kBlock), 0, s, table,
                     m, v, vmax, V, K, grows, sparse ? row_slot : nullptr, dense_grad ? 1 : 0, l2,
                     make_adam_coef(hp), static_cast<const AdamCoef*>(nullptr));
  if (sparse) {
    hipLaunchKernelGGL(clear_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s,
                       seg_rows, n_seg, row_slot);
  }
  return launch_status();
}

static int adam_dense_rows_impl(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v, int64_t V, int K,
                                const float* grows, const float* glin_rows, const int32_t* seg_rows, const int32_t* n_seg,
                                int64_t n_max, int32_t* row_slot, lr_adam_hp hp, const void* coef_dev, lr_stream_t stream) {
  LR_CHECK_ARG(table && m && v && V >= 0 && n_max >= 0 && (coef_dev != nullptr || hp.step >= 1));
  if (V == 0) return LR_OK;
  LR_CHECK_ARG(grows && seg_rows && n_seg && row_slot);
  LR_CHECK_ARG((lin == nullptr) == (lin_m == nullptr) && (lin == nullptr) == (lin_v == nullptr) &&
               (lin == nullptr) == (glin_rows == nullptr));
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(table) % 16 == 0 && reinterpret_cast<uintptr_t>(m) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(v) % 16 == 0 && reinterpret_cast<uintptr_t>(grows) % 16 == 0);
  if (K != 16 && K != 32 && K != 64 && K != 128) return LR_ESHAPE;
  hipStream_t s = as_stream(stream);
  if (coef_dev != nullptr) { hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999; }
  const AdamCoef coef = make_adam_coef(hp);
  if (n_max > 0)
    hipLaunchKernelGGL(mark_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s, seg_rows, n_seg, row_slot);
#define LR_ADR(LPR)                                                                                               \
  hipLaunchKernelGGL((adam_rows_kernel<LPR>), dim3(grid_for(V * LPR, kBlock, kNumCU * 32)), dim3(kBlock), 0, s, table, m, v, \
                     lin, lin_m, lin_v, V, grows, glin_rows, row_slot, coef, static_cast<const AdamCoef*>(coef_dev))
  if (K == 16) LR_ADR(4);
  else if (K == 32) LR_ADR(8);
  else if (K == 64) LR_ADR(16);
  else LR_ADR(32);
#undef LR_ADR
  if (n_max > 0)
    hipLaunchKernelGGL(clear_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s, seg_rows, n_seg, row_slot);
  return launch_status();
}

extern "C" int lr_row_slots_i32(const int32_t* seg_rows, const int32_t* n_seg, int64_t n_max, int32_t* row_slot, int set,
                                lr_stream_t stream) {
  LR_CHECK_ARG(n_max >= 0);
  if (n_max == 0) return LR_OK;
  LR_CHECK_ARG(seg_rows && n_seg && row_slot);
  hipStream_t s = as_stream(stream);
  if (set) hipLaunchKernelGGL(mark_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s, seg_rows, n_seg, row_slot);
  else hipLaunchKernelGGL(clear_slots_kernel, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s, seg_rows, n_seg, row_slot);
  return launch_status();
}

extern "C" int lr_adam_dense_rows_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v, int64_t V,
                                      int K, const float* grows, const float* glin_rows, const int32_t* seg_rows,
                                      const int32_t* n_seg, int64_t n_max, int32_t* row_slot, lr_adam_hp hp,
                                      lr_stream_t stream) {
  return adam_dense_rows_impl(table, m, v, lin, lin_m, lin_v, V, K, grows, glin_rows, seg_rows, n_seg, n_max, row_slot, hp,
                              nullptr, stream);
}

extern "C" int lr_adam_dense_rows_dc_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v, int64_t V,
                                         int K, const float* grows, const float* glin_rows, const int32_t* seg_rows,
                                         const int32_t* n_seg, int64_t n_max, int32_t* row_slot, const void* coef_dev,
                                         lr_stream_t stream) {
  LR_CHECK_ARG(coef_dev != nullptr);
  lr_adam_hp hp{};
  return adam_dense_rows_impl(table, m, v, lin, lin_m, lin_v, V, K, grows, glin_rows, seg_rows, n_seg, n_max, row_slot, hp,
                              coef_dev, stream);
}

// ---- step-dependent Adam coefficients in device memory (hipGraph-captured training steps) ----------
extern "C" size_t lr_adam_coef_bytes(void) { return sizeof(lr::AdamCoef); }

namespace lr {
__global__ void adam_coef_store_kernel(AdamCoef c, AdamCoef* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = c;
}
}  // namespace lr

extern "C" int lr_adam_coef_store(lr_adam_hp hp, void* coef_dev, lr_stream_t stream) {
  LR_CHECK_ARG(coef_dev != nullptr && hp.step >= 1);
  hipLaunchKernelGGL(lr::adam_coef_store_kernel, dim3(1), dim3(64), 0, lr::as_stream(stream),
                     lr::make_adam_coef(hp), static_cast<lr::AdamCoef*>(coef_dev));
  return lr::launch_status();
}

extern "C" int lr_adam_dense_dc_f32(float* table, float* m, float* v, int64_t n, const float* grad,
                                    const void* coef_dev, lr_stream_t stream) {
  LR_CHECK_ARG(n >= 0 && coef_dev != nullptr);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(table && m && v && grad);
  lr_adam_hp hp{};
  hp.step = 1; hp.beta1 = 0.9; hp.beta2 = 0.999; hp.tf_style = 1;
  hipLaunchKernelGGL(lr::adam_dense_kernel, dim3(lr::grid_for(n, lr::kBlock)), dim3(lr::kBlock), 0,
                     lr::as_stream(stream), table, m, v, static_cast<float*>(nullptr), n, 1, grad,
                     static_cast<int32_t*>(nullptr), 1, 0.f, lr::make_adam_coef(hp),
                     static_cast<const lr::AdamCoef*>(coef_dev));
  return lr::launch_status();
}
