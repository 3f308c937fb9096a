// CSR SpMM for LightGCN propagation: Y = A * X with A in CSR (int64 rowptr, int32 col, fp32
// val), X/Y row-major [rows, K].  HBM-bound: per nonzero 8 B of (col,val) + one K*4-byte
// gathered row.  One row group of LPR = K/4 lanes per output row, 4 nonzeros in flight.
// Optional fused accumulation acc += Y implements the running layer sum of
// lightgcn_module.py:83-84 without re-reading Y.  Y == nullptr (with acc): accumulate only — the
// column-blocked products of the row-sharded net add block after block into one table.
#include <type_traits>

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
    if (Y != nullptr) st4(Y + r * K + c4, y);
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
    if (Y) Y[t] = y;
    if (acc) acc[t] += y;
  }
}

// ---------------------------------------------------------------------------------------
// Degree-bucketed variant.  A recommender graph's degrees are Zipf-distributed: the head items have
// 10^5..10^6 neighbours, and one 16-lane row group walking such a row alone is the kernel's critical
// path.  Rows of more than kSpLong nonzeros are cut into chunks of kSpChunk nonzeros by a pre-pass
// (device lists, no host sync) and every chunk is summed by a WHOLE workgroup (256/LPR row groups
// striding the chunk, 4 nonzeros in flight each, LDS fold in group order); rows of several chunks are
// finished by a second pass that adds the chunk sums in chunk order.  The first kLongBlocks workgroups
// of the main launch serve the chunk list while the others serve the short rows, so the long rows run
// underneath the bandwidth-bound bulk.  Every row's summation order is fixed by the row alone (not by the
// order of the lists): results are run-to-run identical.
// ---------------------------------------------------------------------------------------
constexpr int kSpLong = 128;
constexpr int kSpChunk = 2048;
#ifndef LR_SP_LONGBLOCKS          // profiling builds (scripts/lab/r04/m.sh) vary these two
#define LR_SP_LONGBLOCKS 256
#endif
#ifndef LR_SP_CHUNKWIDE
#define LR_SP_CHUNKWIDE 0
#endif
#ifndef LR_SP_EPI_NT              // non-temporal accesses to (w, m, v) in the optimiser epilogue
#define LR_SP_EPI_NT 1
#endif
#ifndef LR_SP_MASKWIDE            // nonzeros in flight per row group where the operand's rows are filtered by a bitmap
#define LR_SP_MASKWIDE 8          // (measured at cfg 5, GPU call r05ab: 8 -> 16.5 ms, 16 -> 19.2 ms, 32 -> 32 ms per product)
#endif
constexpr int kSpLongBlocks = LR_SP_LONGBLOCKS;
// Cache-policy experiments (lab builds, scripts/lab/r06/spmm_nt.sh; the product build defines none of them):
//   LR_SP_NT_STREAM 1: the (col, val) stream is read with non-temporal loads (read once: should not displace gathered rows in L2)
//   LR_SP_NT_COLD H  : gathered rows of columns outside [0, H) and [LR_SP_SPLIT, LR_SP_SPLIT + H) are read non-temporally — on a
//                      Zipf graph whose ids are popularity ranks those ranges are the hot rows of the two sides
#ifndef LR_SP_NT_STREAM
#define LR_SP_NT_STREAM 0
#endif
#ifndef LR_SP_NT_COLD
#define LR_SP_NT_COLD 0
#endif
#ifndef LR_SP_SPLIT
#define LR_SP_SPLIT 10000000
#endif
typedef float sp_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sp_ld4_nt(const float* p) {
  const sp_v4f v = __builtin_nontemporal_load(reinterpret_cast<const sp_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bool sp_hot(int32_t c) {
  return static_cast<uint32_t>(c) < static_cast<uint32_t>(LR_SP_NT_COLD) ||
         static_cast<uint32_t>(c - LR_SP_SPLIT) < static_cast<uint32_t>(LR_SP_NT_COLD);
}
template <int K>
__device__ __forceinline__ float4 sp_ldrow(const float* __restrict__ X, int32_t c, int c4) {
  const float* p = X + static_cast<int64_t>(c) * K + c4;
  if (LR_SP_NT_COLD > 0) return sp_hot(c) ? ld4(p) : sp_ld4_nt(p);
  return ld4(p);
}
__device__ __forceinline__ int32_t sp_ldc(const int32_t* p) { return LR_SP_NT_STREAM ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float sp_lda(const float* p) { return LR_SP_NT_STREAM ? __builtin_nontemporal_load(p) : *p; }

struct SpmmLists {
  int32_t* counters;      // [0] chunks, [1] partial slots, [2] multi-chunk rows
  int64_t* chunk_row;     // [max_chunks]
  int32_t* chunk_idx;     // [max_chunks] chunk number inside its row
  int32_t* chunk_slot;    // [max_chunks] partial slot, or -1: the row is this single chunk
  int64_t* multi_row;     // [max_multi]
  int32_t* multi_slot;    // [max_multi] first partial slot
  int32_t* multi_nc;      // [max_multi]
  float* partial;         // [max_partial][K]
};

static inline int64_t sp_max_long(int64_t nnz) { return nnz / kSpLong + 1; }
static inline int64_t sp_max_chunks(int64_t nnz) { return nnz / kSpChunk + sp_max_long(nnz) + 1; }
static inline int64_t sp_max_multi(int64_t nnz) { return nnz / kSpChunk + 1; }
static inline int64_t sp_max_partial(int64_t nnz) { return 2 * (nnz / kSpChunk) + 2; }
static inline size_t sp_al(size_t x) { return (x + 255) / 256 * 256; }

static SpmmLists sp_carve(void* ws, int64_t nnz, int K, size_t* total) {
  char* p = static_cast<char*>(ws);
  size_t o = 0;
  SpmmLists L{};
  auto take = [&](size_t bytes) { char* q = p ? p + o : nullptr; o += sp_al(bytes); return q; };
  L.counters = reinterpret_cast<int32_t*>(take(256));
  L.chunk_row = reinterpret_cast<int64_t*>(take(sp_max_chunks(nnz) * 8));
  L.chunk_idx = reinterpret_cast<int32_t*>(take(sp_max_chunks(nnz) * 4));
  L.chunk_slot = reinterpret_cast<int32_t*>(take(sp_max_chunks(nnz) * 4));
  L.multi_row = reinterpret_cast<int64_t*>(take(sp_max_multi(nnz) * 8));
  L.multi_slot = reinterpret_cast<int32_t*>(take(sp_max_multi(nnz) * 4));
  L.multi_nc = reinterpret_cast<int32_t*>(take(sp_max_multi(nnz) * 4));
  L.partial = reinterpret_cast<float*>(take(static_cast<size_t>(sp_max_partial(nnz)) * K * 4));
  if (total) *total = o;
  return L;
}

__global__ __launch_bounds__(kBlock) void spmm_classify_kernel(const int64_t* __restrict__ rowptr,
                                                               int64_t rows, SpmmLists L) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride) {
    const int64_t deg = rowptr[r + 1] - rowptr[r];
    if (deg <= kSpLong) continue;
    const int nc = static_cast<int>((deg + kSpChunk - 1) / kSpChunk);
    const int base = atomicAdd(&L.counters[0], nc);
    int slot = -1;
    if (nc > 1) {
      slot = atomicAdd(&L.counters[1], nc);
      const int m = atomicAdd(&L.counters[2], 1);
      L.multi_row[m] = r;
      L.multi_slot[m] = slot;
      L.multi_nc[m] = nc;
    }
    for (int c = 0; c < nc; ++c) {
      L.chunk_row[base + c] = r;
      L.chunk_idx[base + c] = c;
      L.chunk_slot[base + c] = nc > 1 ? slot + c : -1;
    }
  }
}

// bit r of a row bitmap (uint32 words): set = the row takes part
__device__ __forceinline__ bool sp_bit(const uint32_t* __restrict__ bm, int64_t r) {
  return (bm[r >> 5] >> (r & 31)) & 1u;
}

// `MASKED`: X rows whose bit in `xm` is clear are known to be zero and are not read (y + a * 0 == y: the same bits) — the first
// backward product of a LightGCN step multiplies by a gradient that is nonzero on the batch's rows only
template <int LPR, bool MASKED = false>
__device__ __forceinline__ float4 spmm_walk(const int32_t* __restrict__ col, const float* __restrict__ val,
                                            const float* __restrict__ X, int64_t j, int64_t j1, int64_t step,
                                            int c4, const uint32_t* __restrict__ xm = nullptr) {
  constexpr int K = LPR * 4;
  float4 y = f4_zero();
  auto xrow = [&](int32_t c) {
    if (MASKED && !sp_bit(xm, c)) return f4_zero();
    return ld4(X + static_cast<int64_t>(c) * K + c4);
  };
  // eight nonzeros in flight where a row has them (same ascending fma order as the four-wide body: bit-identical sums;
  // one dependent col -> row round less per eight nonzeros); only the contiguous walk of a short row (step == 4)
  if (step == 4 || LR_SP_CHUNKWIDE) {
    auto wide = [&](auto width) {
      constexpr int Wd = decltype(width)::value;
      for (; j + (step == 4 ? Wd : (Wd / 4 - 1) * step + 4) <= j1; j += (step == 4 ? Wd : (Wd / 4) * step)) {
        int32_t c[Wd];
        float a[Wd];
#pragma unroll
        for (int q = 0; q < Wd; ++q) {       // contiguous (short row) or quads `step` apart (a chunk shared by NG row groups)
          const int64_t jq = step == 4 ? j + q : j + (q / 4) * step + (q % 4);
          c[q] = sp_ldc(col + jq); a[q] = sp_lda(val + jq);
        }
        float4 x[Wd];
        if (MASKED) {       // all bitmap words first, then the rows that are there: one round trip each, not one per nonzero
          uint32_t wq[Wd];
#pragma unroll
          for (int q = 0; q < Wd; ++q) wq[q] = xm[c[q] >> 5];
#pragma unroll
          for (int q = 0; q < Wd; ++q)
            x[q] = ((wq[q] >> (c[q] & 31)) & 1u) ? ld4(X + static_cast<int64_t>(c[q]) * K + c4) : f4_zero();
        } else {
#pragma unroll
          for (int q = 0; q < Wd; ++q) x[q] = sp_ldrow<K>(X, c[q], c4);
        }
#pragma unroll
        for (int q = 0; q < Wd; ++q) y = f4_fma(make_float4(a[q], a[q], a[q], a[q]), x[q], y);
      }
    };
    // (16 in flight measured no better for the plain product, GPU call r03ag, and worse for the bitmap-filtered one, r05ab)
    if (MASKED && LR_SP_MASKWIDE != 8) wide(std::integral_constant<int, LR_SP_MASKWIDE>{});
    wide(std::integral_constant<int, 8>{});
  }
  for (; j < j1; j += step) {
    const int64_t rem = j1 - j;
    if (rem >= 4) {
      const int32_t c0 = col[j], c1 = col[j + 1], c2 = col[j + 2], c3 = col[j + 3];
      const float a0 = val[j], a1 = val[j + 1], a2 = val[j + 2], a3 = val[j + 3];
      float4 x0, x1, x2, x3;
      if (MASKED) {
        const uint32_t w0 = xm[c0 >> 5], w1 = xm[c1 >> 5], w2 = xm[c2 >> 5], w3 = xm[c3 >> 5];
        x0 = ((w0 >> (c0 & 31)) & 1u) ? ld4(X + static_cast<int64_t>(c0) * K + c4) : f4_zero();
        x1 = ((w1 >> (c1 & 31)) & 1u) ? ld4(X + static_cast<int64_t>(c1) * K + c4) : f4_zero();
        x2 = ((w2 >> (c2 & 31)) & 1u) ? ld4(X + static_cast<int64_t>(c2) * K + c4) : f4_zero();
        x3 = ((w3 >> (c3 & 31)) & 1u) ? ld4(X + static_cast<int64_t>(c3) * K + c4) : f4_zero();
      } else {
        x0 = sp_ldrow<K>(X, c0, c4);
        x1 = sp_ldrow<K>(X, c1, c4);
        x2 = sp_ldrow<K>(X, c2, c4);
        x3 = sp_ldrow<K>(X, c3, c4);
      }
      y = f4_fma(make_float4(a0, a0, a0, a0), x0, y);
      y = f4_fma(make_float4(a1, a1, a1, a1), x1, y);
      y = f4_fma(make_float4(a2, a2, a2, a2), x2, y);
      y = f4_fma(make_float4(a3, a3, a3, a3), x3, y);
    } else {
      for (int64_t q = j; q < j1; ++q) {
        const float a = val[q];
        y = f4_fma(make_float4(a, a, a, a), xrow(col[q]), y);
      }
    }
  }
  return y;
}

// Optional epilogue of a product whose rows are the gradient of a parameter table (the LAST backward product of a LightGCN step,
// lightgcn_module.py:66-88 under autograd + training/torch_trainer.py:63-69): instead of writing row r of Y, the optimiser step
// of row r of (w, m, v [, vmax]) is applied to g = y (+ alpha * gsum[slot] for the batch's rows: the loss's own gradient rows,
// summed per distinct row) — the arithmetic of lr_embed_scatter_add_f32 followed by lr_adam_dense_f32, without the 5 GB
// gradient table written and read in between.
struct SpmmAdam {
  float* w; float* m; float* v; float* vmax;     // [rows, K]; vmax nullable (AMSGrad)
  const int32_t* row_slot;                       // [rows]: -1 or the row's index in gsum (nullable: no batch rows)
  const float* gsum;                             // [n_seg, K]
  float alpha;
  AdamCoef coef;
};

template <int LPR, bool FUSED>
__device__ __forceinline__ void sp_store(int64_t r, int c4, float4 y, float* __restrict__ Y, float* __restrict__ acc,
                                         const SpmmAdam& A) {
  constexpr int K = LPR * 4;
  const int64_t off = r * K + c4;
  if (!FUSED) {
    if (Y != nullptr) st4(Y + off, y);          // (Y == nullptr: accumulate only, `acc += A X`)
    if (acc != nullptr) st4(acc + off, f4_add(ld4(acc + off), y));
    return;
  }
  float4 g = y;
  if (A.row_slot != nullptr) {
    const int32_t slot = A.row_slot[r];
    if (slot >= 0) g = f4_fma(make_float4(A.alpha, A.alpha, A.alpha, A.alpha), ld4(A.gsum + static_cast<int64_t>(slot) * K + c4), y);
  }
  // (w, m, v are touched once per step: streaming accesses keep them out of the L2 lines the gathered rows of X live in)
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
  const float4 w = LR_SP_EPI_NT ? ldnt(A.w + off) : ld4(A.w + off);
  float4 mm = LR_SP_EPI_NT ? ldnt(A.m + off) : ld4(A.m + off), vv = LR_SP_EPI_NT ? ldnt(A.v + off) : ld4(A.v + off);
  const float gg[4] = {g.x, g.y, g.z, g.w}, ww[4] = {w.x, w.y, w.z, w.w};
  float mq[4] = {mm.x, mm.y, mm.z, mm.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w}, out[4], vm[4];
  float4 vmx = f4_zero();
  if (A.vmax != nullptr) vmx = ld4(A.vmax + off);
  const float vmq[4] = {vmx.x, vmx.y, vmx.z, vmx.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {                  // the element arithmetic of adam_dense_kernel (csrc/embed_scatter.hip)
    out[e] = adam_elem(ww[e], gg[e], mq[e], vq[e], A.coef);
    if (A.vmax != nullptr) {
      vm[e] = fmaxf(vmq[e], vq[e]);
      const float denom = A.coef.tf_style ? sqrtf(vm[e]) + A.coef.eps : sqrtf(vm[e]) / A.coef.bc2_sqrt + A.coef.eps;
      out[e] = ww[e] - A.coef.step_size * (mq[e] / denom);
    }
  }
  if (LR_SP_EPI_NT) {
    stnt(A.w + off, make_float4(out[0], out[1], out[2], out[3]));
    stnt(A.m + off, make_float4(mq[0], mq[1], mq[2], mq[3]));
    stnt(A.v + off, make_float4(vq[0], vq[1], vq[2], vq[3]));
  } else {
    st4(A.w + off, make_float4(out[0], out[1], out[2], out[3]));
    st4(A.m + off, make_float4(mq[0], mq[1], mq[2], mq[3]));
    st4(A.v + off, make_float4(vq[0], vq[1], vq[2], vq[3]));
  }
  if (A.vmax != nullptr) st4(A.vmax + off, make_float4(vm[0], vm[1], vm[2], vm[3]));
}

// `xm` / `ym` (MASKED instantiation, each nullable): bitmaps over the rows of X that may be nonzero / over the rows of Y that are
// wanted (the others are left untouched) — the last forward product of a training step is only read at the batch's rows.
// `n_long` workgroups serve the chunk list (the rows a `ym` leaves are mostly the long ones).
template <int LPR, bool MASKED = false, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void spmm_bucketed_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int64_t rows, const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ acc, SpmmLists L,
    const uint32_t* __restrict__ xm, const uint32_t* __restrict__ ym, int n_long, SpmmAdam A) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  const bool xmask = MASKED && xm != nullptr, ymask = MASKED && ym != nullptr;
  if (static_cast<int>(blockIdx.x) < n_long) {
    __shared__ float4 red[NG][LPR];
    const int n_chunks = L.counters[0];
    const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR, c4 = gl * 4;
    for (int ci = blockIdx.x; ci < n_chunks; ci += n_long) {
      const int64_t r = L.chunk_row[ci];
      if (ymask && !sp_bit(ym, r)) continue;          // (uniform over the workgroup)
      const int64_t j0 = rowptr[r] + static_cast<int64_t>(L.chunk_idx[ci]) * kSpChunk;
      const int64_t jr = rowptr[r + 1];
      const int64_t j1 = j0 + kSpChunk < jr ? j0 + kSpChunk : jr;
      red[grp][gl] = xmask ? spmm_walk<LPR, true>(col, val, X, j0 + grp * 4, j1, NG * 4, c4, xm)
                           : spmm_walk<LPR>(col, val, X, j0 + grp * 4, j1, NG * 4, c4);
      __syncthreads();
      if (grp == 0) {
        float4 t = f4_zero();
#pragma unroll 4
        for (int g = 0; g < NG; ++g) t = f4_add(t, red[g][gl]);   // fixed order
        const int slot = L.chunk_slot[ci];
        if (slot >= 0) {
          st4(L.partial + static_cast<int64_t>(slot) * K + c4, t);
        } else {
          sp_store<LPR, FUSED>(r, c4, t, Y, acc, A);
        }
      }
      __syncthreads();
    }
    return;
  }
  const int64_t gtid = static_cast<int64_t>(blockIdx.x - n_long) * kBlock + threadIdx.x;
  const int c4 = static_cast<int>(gtid % LPR) * 4;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x - n_long) * kBlock / LPR;
  for (int64_t r = gtid / LPR; r < rows; r += ngroups) {
    if (ymask && !sp_bit(ym, r)) continue;
    const int64_t j0 = rowptr[r], j1 = rowptr[r + 1];
    if (j1 - j0 > kSpLong) continue;
    const float4 y = xmask ? spmm_walk<LPR, true>(col, val, X, j0, j1, 4, c4, xm) : spmm_walk<LPR>(col, val, X, j0, j1, 4, c4);
    sp_store<LPR, FUSED>(r, c4, y, Y, acc, A);
  }
}

// Rows of more than one chunk: their chunk sums are added up here.  One WORKGROUP per such row: row group g adds chunks
// g, g + NG, ... in ascending order, the NG group sums are then added in group order through LDS — a fixed order, and
// the head row of a Zipf graph (thousands of chunks) no longer sets the launch's tail with one serial chain.
template <int LPR, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void spmm_finish_kernel(float* __restrict__ Y, float* __restrict__ acc,
                                                             SpmmLists L, const uint32_t* __restrict__ ym, SpmmAdam A) {
  constexpr int K = LPR * 4, NG = kBlock / LPR;
  __shared__ float4 red[NG][LPR];
  const int n_multi = L.counters[2];
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR, c4 = gl * 4;
  for (int m = blockIdx.x; m < n_multi; m += gridDim.x) {
    const int64_t r = L.multi_row[m];
    if (ym != nullptr && !sp_bit(ym, r)) continue;
    const int slot = L.multi_slot[m], nc = L.multi_nc[m];
    float4 y = f4_zero();
    for (int c = grp; c < nc; c += NG) y = f4_add(y, ld4(L.partial + static_cast<int64_t>(slot + c) * K + c4));
    red[grp][gl] = y;
    __syncthreads();
    if (grp == 0) {
      float4 t = red[0][gl];
#pragma unroll 4
      for (int g = 1; g < NG; ++g) t = f4_add(t, red[g][gl]);      // group order
      sp_store<LPR, FUSED>(r, c4, t, Y, acc, A);
    }
    __syncthreads();
  }
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_spmm_csr_ws_bytes(int64_t rows, int64_t nnz, int K) {
  if (rows < 0 || nnz < 0 || K < 1) return 0;
  size_t total = 0;
  sp_carve(nullptr, nnz, K, &total);
  return total;
}

extern "C" int lr_spmm_csr_bucketed_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                                        int64_t rows, int64_t nnz, const float* X, int K, float* Y,
                                        float* acc, void* ws, size_t ws_bytes, int lists_ready, lr_stream_t stream) {
  return lr_spmm_csr_masked_f32(rowptr, col, val, rows, nnz, X, K, Y, acc, nullptr, nullptr, ws, ws_bytes, lists_ready, stream);
}

// row bitmaps (uint32 words, bit r of word r / 32): set / cleared by the listed ids (ids < 0 or >= n_bits are skipped)
namespace lr {
__global__ __launch_bounds__(kBlock) void bitmap_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int64_t n_bits,
                                                            uint32_t* __restrict__ bm, int set) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t r = ids[i];
    if (r < 0 || r >= n_bits) continue;
    if (set) atomicOr(&bm[r >> 5], 1u << (r & 31));
    else bm[r >> 5] = 0u;                      // clearing: every listed id's whole word (only listed ids were ever set)
  }
}
}  // namespace lr

extern "C" int lr_bitmap_ids_i32(const int32_t* ids, int64_t n, int64_t n_bits, uint32_t* bitmap, int set,
                                 lr_stream_t stream) {
  LR_CHECK_ARG(n >= 0 && n_bits >= 0);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(ids && bitmap);
  hipLaunchKernelGGL(lr::bitmap_ids_kernel, dim3(lr::grid_for(n, lr::kBlock)), dim3(lr::kBlock), 0, lr::as_stream(stream), ids, n,
                     n_bits, bitmap, set);
  return lr::launch_status();
}

static int spmm_bucketed_impl(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows,
                              int64_t nnz, const float* X, int K, float* Y, float* acc, const uint32_t* xmask,
                              const uint32_t* ymask, void* ws, size_t ws_bytes, int lists_ready,
                              lr_stream_t stream, const SpmmAdam* epi) {
  LR_CHECK_ARG(rows >= 0 && nnz >= 0 && K >= 1);
  if (rows == 0) return LR_OK;
  LR_CHECK_ARG(rowptr && X && (Y || acc || epi));      // Y == nullptr with acc: accumulate only
  const bool aligned = reinterpret_cast<uintptr_t>(X) % 16 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0 &&
                       (!acc || reinterpret_cast<uintptr_t>(acc) % 16 == 0);
  const bool masked = xmask != nullptr || ymask != nullptr;
  if (!aligned || !(K == 16 || K == 32 || K == 64 || K == 128)) {
    if (masked || epi) return LR_ESHAPE;    // (the plain kernels take no bitmaps and no epilogue)
    return lr_spmm_csr_f32(rowptr, col, val, rows, X, K, Y, acc, stream);
  }
  if (epi != nullptr && masked) return LR_EINVAL;
  const SpmmAdam A = epi != nullptr ? *epi : SpmmAdam{};
  size_t need = 0;
  SpmmLists L = sp_carve(ws, nnz, K, &need);
  if (ws == nullptr || ws_bytes < need) return LR_EWORKSPACE;
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(ws) % 16 == 0);
  hipStream_t s = as_stream(stream);
  if (!lists_ready) {     // the chunk lists depend on rowptr alone: a caller that multiplies by the same graph again (six
                          // products per LightGCN step, every step) keeps `ws` and passes lists_ready = 1
    zero_words_async(L.counters, 64, s);
    hipLaunchKernelGGL(spmm_classify_kernel, dim3(grid_for(rows, kBlock, kNumCU * 4)), dim3(kBlock), 0, s, rowptr, rows, L);
  }
  // with a row bitmap the bulk of the work is the chunk list of the (few, long) wanted rows: give it the chip
  const int n_long = ymask != nullptr ? kNumCU * 8 : kSpLongBlocks;
#define LR_SPMMB(LPR)                                                                                  \
  {                                                                                                    \
    const int grid = grid_for(rows, kBlock / LPR) + n_long;                                            \
    if (epi != nullptr) {                                                                              \
      hipLaunchKernelGGL((spmm_bucketed_kernel<LPR, false, true>), dim3(grid), dim3(kBlock), 0, s, rowptr, col, val, \
                         rows, X, Y, acc, L, xmask, ymask, n_long, A);                                 \
      hipLaunchKernelGGL((spmm_finish_kernel<LPR, true>), dim3(kNumCU), dim3(kBlock), 0, s, Y, acc, L, ymask, A); \
      return launch_status();                                                                          \
    }                                                                                                  \
    if (masked)                                                                                        \
      hipLaunchKernelGGL((spmm_bucketed_kernel<LPR, true>), dim3(grid), dim3(kBlock), 0, s, rowptr, col, val, \
                         rows, X, Y, acc, L, xmask, ymask, n_long, A);                                 \
    else                                                                                               \
      hipLaunchKernelGGL((spmm_bucketed_kernel<LPR>), dim3(grid), dim3(kBlock), 0, s, rowptr, col, val, \
                         rows, X, Y, acc, L, xmask, ymask, n_long, A);                                 \
    hipLaunchKernelGGL((spmm_finish_kernel<LPR>), dim3(kNumCU), dim3(kBlock), 0, s, Y, acc, L, ymask, A); \
    return launch_status();                                                                            \
  }
  if (K == 16) LR_SPMMB(4)
  if (K == 32) LR_SPMMB(8)
  if (K == 64) LR_SPMMB(16)
  LR_SPMMB(32)
#undef LR_SPMMB
}

extern "C" int lr_spmm_csr_masked_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows,
                                      int64_t nnz, const float* X, int K, float* Y, float* acc, const uint32_t* xmask,
                                      const uint32_t* ymask, void* ws, size_t ws_bytes, int lists_ready,
                                      lr_stream_t stream) {
  return spmm_bucketed_impl(rowptr, col, val, rows, nnz, X, K, Y, acc, xmask, ymask, ws, ws_bytes, lists_ready, stream, nullptr);
}

extern "C" int lr_spmm_csr_adam_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows, int64_t nnz,
                                    const float* X, int K, float* w, float* m, float* v, float* vmax,
                                    const int32_t* row_slot, const float* gsum, float alpha, lr_adam_hp hp, void* ws,
                                    size_t ws_bytes, int lists_ready, lr_stream_t stream) {
  LR_CHECK_ARG(w && m && v && hp.step >= 1 && (row_slot == nullptr) == (gsum == nullptr));
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(w) % 16 == 0 && reinterpret_cast<uintptr_t>(m) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(v) % 16 == 0 && (!vmax || reinterpret_cast<uintptr_t>(vmax) % 16 == 0) &&
               (!gsum || reinterpret_cast<uintptr_t>(gsum) % 16 == 0));
  SpmmAdam A{};
  A.w = w; A.m = m; A.v = v; A.vmax = vmax; A.row_slot = row_slot; A.gsum = gsum; A.alpha = alpha;
  A.coef = make_adam_coef(hp);
  return spmm_bucketed_impl(rowptr, col, val, rows, nnz, X, K, w, nullptr, nullptr, nullptr, ws, ws_bytes, lists_ready, stream, &A);
}

extern "C" int lr_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                               int64_t rows, const float* X, int K, float* Y, float* acc,
                               lr_stream_t stream) {
  LR_CHECK_ARG(rows >= 0 && K >= 1);
  if (rows == 0) return LR_OK;
  LR_CHECK_ARG(rowptr && X && (Y || acc));
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
