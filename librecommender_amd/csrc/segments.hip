// Segment construction: group the positions of a batch's row ids by row ("CSR by touched
// row").  Integer-only work; result is fully deterministic (stable LSD radix sort).
//
// Device-wide sort and scan come from the rocPRIM headers shipped with ROCm (templates
// compiled into this object for gfx950 — no runtime library dependency); everything that
// touches embedding bytes is hand-written in the other translation units.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace lr {

__global__ __launch_bounds__(kBlock) void seg_keys_kernel(const int32_t* __restrict__ idx,
                                                          int64_t n, uint32_t V,
                                                          uint32_t* __restrict__ keys) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int32_t id = idx[i];
    keys[i] = (id >= 0 && static_cast<uint32_t>(id) < V) ? static_cast<uint32_t>(id) : V;
  }
}

__global__ __launch_bounds__(kBlock) void seg_heads_kernel(const uint32_t* __restrict__ keys,
                                                           int64_t n, uint32_t V,
                                                           int32_t* __restrict__ heads,
                                                           int32_t* __restrict__ n_seg,
                                                           int32_t* __restrict__ seg_start) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t0 == 0) {  // defaults for "no valid entry"
    n_seg[0] = 0;
    seg_start[0] = 0;
  }
  for (int64_t i = t0; i < n; i += stride) {
    const uint32_t k = keys[i];
    heads[i] = (k < V && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
  }
}

__global__ __launch_bounds__(kBlock) void seg_emit_kernel(
    const uint32_t* __restrict__ keys, const int32_t* __restrict__ rank, int64_t n, uint32_t V,
    const int32_t* __restrict__ seg_pos, int32_t* __restrict__ seg_rows,
    int32_t* __restrict__ seg_start, int32_t* __restrict__ n_seg,
    int32_t* __restrict__ pos_to_seg) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t k = keys[i];
    if (k >= V) {
      if (pos_to_seg != nullptr) pos_to_seg[seg_pos[i]] = -1;
      continue;
    }
    const int32_t r = rank[i];  // inclusive scan of heads: 1-based segment number
    if (pos_to_seg != nullptr) pos_to_seg[seg_pos[i]] = r - 1;
    if (i == 0 || keys[i - 1] != k) {
      seg_rows[r - 1] = static_cast<int32_t>(k);
      seg_start[r - 1] = static_cast<int32_t>(i);
    }
    if (i == n - 1 || keys[i + 1] >= V) {  // last valid entry closes the CSR
      seg_start[r] = static_cast<int32_t>(i + 1);
      n_seg[0] = r;
    }
  }
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static inline int key_bits(int64_t V) {
  int bits = 1;
  while (bits < 32 && (static_cast<uint64_t>(1) << bits) <= static_cast<uint64_t>(V)) ++bits;
  return bits;  // keys take values in [0, V] (V = "dropped" sentinel)
}

struct SegWs {
  uint32_t* keys_in;
  uint32_t* keys_out;
  int32_t* rank;
  void* prim;
  size_t prim_bytes;
};

static inline size_t seg_fixed_bytes(int64_t n) {
  return align_up(static_cast<size_t>(n) * 4) * 3;
}

// -----------------------------------------------------------------------------------------
// Field-partitioned segment build.  The feature models address ONE table by global rows
// [user | item | sparse field 0 | sparse field 1 ...] and column f of idx only holds rows of field
// f, so the batch's keys are already partitioned into F disjoint ascending ranges: sorting each
// column on its own (16384 keys, local id < 2^20) in LDS replaces the global radix sort.
//
//   seg_field_sort_kernel : one workgroup per field.  Elements (local id << 32 | sample) live in
//     registers (64 per thread, wave w owns samples [w*4096, (w+1)*4096)); each 8-bit LSD pass
//     ranks them with a wave-level match (8 ballots), per-wave digit counters in LDS (histogram
//     sweep with fire-and-forget atomics, ranking sweep with returning atomics that pipeline
//     across the unrolled steps) and scatters through ONE 128 KB LDS buffer.  Stable, so the
//     samples of a run stay ascending.  Ids outside the field's row range sort last and are dropped.
//   seg_field_emit_kernel : per field, adds the preceding fields' counts (block reduction over <= F
//     values) and emits seg_pos / seg_rows / seg_start / n_seg in the layout of lr_segments_build,
//     plus slotT[f][b] = index of position (b, f) in seg_pos (-1 if dropped).
// -----------------------------------------------------------------------------------------
constexpr int kFS = 16384;             // samples per field handled in LDS
constexpr int kSortThreads = 1024;     // 16 waves: 4 per SIMD hide the LDS-atomic latency of the ranking
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kFSE = kFS / kSortThreads;   // elements per thread
constexpr int kFSW = kFS / kSortWaves;     // contiguous samples per wave
constexpr uint32_t kKeyInvalid = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t match_digit(uint32_t d) {
  uint64_t mask = ~0ull;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool bit = (d >> k) & 1u;
    const uint64_t bk = __ballot(bit);
    mask &= bit ? bk : ~bk;
  }
  return mask;
}

__global__ __launch_bounds__(kSortThreads) void seg_field_sort_kernel(
    const int32_t* __restrict__ idxT, int B, const int32_t* __restrict__ frs,
    uint64_t* __restrict__ sorted, int32_t* __restrict__ fcount) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* buf = reinterpret_cast<uint64_t*>(smem);       // [kFS]
  int* cnt = reinterpret_cast<int*>(buf + kFS);            // [kSortWaves][256]
  int* wsum = cnt + kSortWaves * 256;                      // [4] scan scratch, then [kSortWaves][2] totals
  const int f = blockIdx.x;
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int32_t lo = frs[f], hi = frs[f + 1];
  const uint64_t lt = (1ull << lane) - 1ull;
  const int32_t* col = idxT + static_cast<int64_t>(f) * B;
  // local ids are < hi - lo and the dropped-entry key is all ones: passes covering
  // bits(hi - lo) keep every valid key below it in the examined digits
  int bits = 0;
  while (bits < 32 && ((1ull << bits) - 1ull) < static_cast<uint64_t>(hi > lo ? hi - lo : 0)) ++bits;
  const int n_pass = (bits + 7) / 8;

  uint64_t x[kFSE];
#pragma unroll
  for (int s = 0; s < kFSE; ++s) {
    const int p = wid * kFSW + s * 64 + lane;
    const int32_t id = p < B ? col[p] : -1;
    const bool ok = id >= lo && id < hi;
    const uint32_t key = ok ? static_cast<uint32_t>(id - lo) : kKeyInvalid;
    x[s] = (static_cast<uint64_t>(key) << 32) | static_cast<uint32_t>(p);
  }

  int* my = cnt + wid * 256;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int shift = 32 + 8 * pass;
    for (int i = tid; i < kSortWaves * 256; i += kSortThreads) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kFSE; ++s) {               // histogram of this wave's elements
      const uint32_t d = static_cast<uint32_t>(x[s] >> shift) & 255u;
      const uint64_t mk = match_digit(d);
      if ((mk & lt) == 0ull)                       // lowest lane of its digit group
        __hip_atomic_fetch_add(my + d, __popcll(mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    // threads 0..255: exclusive offsets of (digit d, wave w) in digit-major, wave-minor order
    int tot = 0, inc = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < kSortWaves; ++w) tot += cnt[w * 256 + tid];
      inc = tot;                                    // inclusive scan over the 256 digits
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
      }
      if (lane == 63) wsum[wid] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      int run = inc - tot;
      for (int w = 0; w < wid; ++w) run += wsum[w];
#pragma unroll
      for (int w = 0; w < kSortWaves; ++w) {
        const int c = cnt[w * 256 + tid];
        cnt[w * 256 + tid] = run;
        run += c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kFSE; ++s) {               // rank + scatter
      uint32_t d = static_cast<uint32_t>(x[s] >> shift) & 255u;
      asm volatile("" : "+v"(d));   // opaque: recompute the digit group here instead of keeping the
                                    // histogram sweep's per-bit masks alive across the scan (spills)
      const uint64_t mk = match_digit(d);
      const int leader = __ffsll(static_cast<long long>(mk)) - 1;
      int old = 0;
      if (lane == leader)
        old = __hip_atomic_fetch_add(my + d, __popcll(mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = __shfl(old, leader);
      buf[old + __popcll(mk & lt)] = x[s];
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the live ranges of the unrolled steps
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kFSE; ++s) x[s] = buf[wid * kFSW + s * 64 + lane];
    // nothing writes `buf` again before two further barriers of the next pass
  }
  if (n_pass == 0) {   // degenerate (empty row range): everything is dropped, keep the sample order
#pragma unroll
    for (int s = 0; s < kFSE; ++s) buf[wid * kFSW + s * 64 + lane] = x[s];
  }
  __syncthreads();

  // sorted elements out (valid ones first) + the field's counts: valid entries, distinct rows
  int nv = 0, nh = 0;
  uint64_t* dst = sorted + static_cast<int64_t>(f) * B;
#pragma unroll
  for (int s = 0; s < kFSE; ++s) {
    const int i = wid * kFSW + s * 64 + lane;
    const uint32_t key = static_cast<uint32_t>(x[s] >> 32);
    const bool valid = key != kKeyInvalid;
    const bool head = valid && (i == 0 || static_cast<uint32_t>(buf[i > 0 ? i - 1 : 0] >> 32) != key);
    nv += __popcll(__ballot(valid));
    nh += __popcll(__ballot(head));
    if (i < B) dst[i] = x[s];
  }
  __syncthreads();                                  // wsum[0..3] (scan scratch) is free again
  if (lane == 0) { wsum[wid * 2] = nv; wsum[wid * 2 + 1] = nh; }
  __syncthreads();
  if (tid == 0) {
    int tv = 0, th = 0;
    for (int w = 0; w < kSortWaves; ++w) { tv += wsum[2 * w]; th += wsum[2 * w + 1]; }
    fcount[2 * f] = tv;
    fcount[2 * f + 1] = th;
  }
}

constexpr int kEmitThreads = 1024;     // 16 waves per field: each walks 1/16 of the field's sorted entries
constexpr int kEmitWaves = kEmitThreads / 64;

__global__ __launch_bounds__(kEmitThreads) void seg_field_emit_kernel(
    const uint64_t* __restrict__ sorted, const int32_t* __restrict__ fcount, int B, int F,
    const int32_t* __restrict__ frs, int32_t* __restrict__ seg_pos, int32_t* __restrict__ seg_rows,
    int32_t* __restrict__ seg_start, int32_t* __restrict__ n_seg, int32_t* __restrict__ slotT) {
  __shared__ int red[2][kEmitWaves];
  __shared__ int wh[kEmitWaves];
  const int f = blockIdx.x;
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
  // bases = counts of the preceding fields
  int bv = 0, bh = 0;
  for (int g = tid; g < f; g += kEmitThreads) { bv += fcount[2 * g]; bh += fcount[2 * g + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { bv += __shfl_xor(bv, o); bh += __shfl_xor(bh, o); }
  if (lane == 0) { red[0][wid] = bv; red[1][wid] = bh; }
  __syncthreads();
  int base_v = 0, base_h = 0;
#pragma unroll
  for (int w = 0; w < kEmitWaves; ++w) { base_v += red[0][w]; base_h += red[1][w]; }
  const int nv = fcount[2 * f];
  const int32_t lo = frs[f];
  const uint64_t* src = sorted + static_cast<int64_t>(f) * B;
  const int chunk = ((B + kEmitWaves - 1) / kEmitWaves + 63) / 64 * 64;   // per-wave contiguous range, whole steps
  const int i0 = wid * chunk, i1 = (i0 + chunk) < B ? (i0 + chunk) : B;
  auto is_head = [&](int i, uint32_t key) {
    return i < nv && (i == 0 || static_cast<uint32_t>(src[i - 1] >> 32) != key);
  };
  int heads = 0;                                          // distinct rows in this wave's range
  for (int i = i0 + lane; i < i0 + chunk; i += 64) {
    const bool h = i < i1 && is_head(i, static_cast<uint32_t>(src[i < B ? i : 0] >> 32));
    heads += __popcll(__ballot(h));
  }
  if (lane == 0) wh[wid] = heads;
  __syncthreads();
  int run = base_h;
  for (int w = 0; w < wid; ++w) run += wh[w];
  for (int i = i0 + lane; i < i0 + chunk; i += 64) {
    const bool in = i < i1;
    const uint64_t e = src[in ? i : 0];
    const uint32_t key = static_cast<uint32_t>(e >> 32);
    const int p = static_cast<int>(static_cast<uint32_t>(e));
    const bool h = in && is_head(i, key);
    const uint64_t hb = __ballot(h);
    if (in) {
      const bool valid = i < nv;
      if (valid) seg_pos[base_v + i] = p * F + f;
      if (slotT != nullptr) slotT[static_cast<int64_t>(f) * B + p] = valid ? base_v + i : -1;
      if (h) {
        const int r = run + __popcll(hb & lt);
        seg_rows[r] = lo + static_cast<int32_t>(key);
        seg_start[r] = base_v + i;
      }
    }
    run += __popcll(hb);
  }
  if (f == F - 1 && tid == 0) {
    const int tot_h = base_h + fcount[2 * f + 1];
    seg_start[tot_h] = base_v + nv;
    n_seg[0] = tot_h;
  }
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_segments_ws_bytes(int64_t n, int64_t V) {
  (void)V;
  if (n < 0) n = 0;
  // 3 int32 arrays + rocPRIM scratch (ping-pong key/value buffers + histograms), bounded.
  return seg_fixed_bytes(n) + align_up(static_cast<size_t>(n) * 8) * 2 + (size_t(8) << 20);
}

extern "C" int lr_segments_build(const int32_t* idx, int64_t n, int64_t V, int32_t* seg_pos,
                                 int32_t* seg_rows, int32_t* seg_start, int32_t* n_seg,
                                 int32_t* pos_to_seg, void* ws, size_t ws_bytes,
                                 lr_stream_t stream) {
  LR_CHECK_ARG(seg_start && n_seg && n >= 0 && V >= 0 && V < (int64_t(1) << 31));
  LR_CHECK_ARG(n < (int64_t(1) << 31));
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    hipLaunchKernelGGL(seg_heads_kernel, dim3(1), dim3(kBlock), 0, s, nullptr, int64_t(0), 0u,
                       nullptr, n_seg, seg_start);
    return launch_status();
  }
  LR_CHECK_ARG(idx && seg_pos && seg_rows && ws);
  if (ws_bytes < lr_segments_ws_bytes(n, V)) return LR_EWORKSPACE;

  char* p = static_cast<char*>(ws);
  const size_t a = align_up(static_cast<size_t>(n) * 4);
  SegWs w;
  w.keys_in = reinterpret_cast<uint32_t*>(p);
  w.keys_out = reinterpret_cast<uint32_t*>(p + a);
  w.rank = reinterpret_cast<int32_t*>(p + 2 * a);
  w.prim = p + 3 * a;
  w.prim_bytes = ws_bytes - 3 * a;

  const uint32_t Vu = static_cast<uint32_t>(V);
  const int grid = grid_for(n, kBlock);
  hipLaunchKernelGGL(seg_keys_kernel, dim3(grid), dim3(kBlock), 0, s, idx, n, Vu, w.keys_in);

  const int bits = key_bits(V);
  size_t need = 0;
  rocprim::counting_iterator<int32_t> iota(0);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, w.keys_in, w.keys_out, iota, seg_pos,
                                           static_cast<size_t>(n), 0u,
                                           static_cast<unsigned>(bits), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (need > w.prim_bytes) return LR_EWORKSPACE;
  e = rocprim::radix_sort_pairs(w.prim, need, w.keys_in, w.keys_out, iota, seg_pos,
                                static_cast<size_t>(n), 0u, static_cast<unsigned>(bits), s);
  if (e != hipSuccess) return static_cast<int>(e);

  hipLaunchKernelGGL(seg_heads_kernel, dim3(grid), dim3(kBlock), 0, s, w.keys_out, n, Vu,
                     w.rank, n_seg, seg_start);
  need = 0;
  e = rocprim::inclusive_scan(nullptr, need, w.rank, w.rank, static_cast<size_t>(n),
                              rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (need > w.prim_bytes) return LR_EWORKSPACE;
  e = rocprim::inclusive_scan(w.prim, need, w.rank, w.rank, static_cast<size_t>(n),
                              rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);

  hipLaunchKernelGGL(seg_emit_kernel, dim3(grid), dim3(kBlock), 0, s, w.keys_out, w.rank, n,
                     Vu, seg_pos, seg_rows, seg_start, n_seg, pos_to_seg);
  return launch_status();
}

extern "C" size_t lr_segments_fields_ws_bytes(int64_t B, int F) {
  if (B < 0 || F < 1) return 0;
  return align_up(static_cast<size_t>(B) * F * 8) + align_up(static_cast<size_t>(F) * 8);
}

extern "C" int lr_segments_build_fields(const int32_t* idxT, int64_t B, int F,
                                        const int32_t* field_row_start,
                                        int32_t* seg_pos, int32_t* seg_rows, int32_t* seg_start,
                                        int32_t* n_seg, int32_t* slotT, void* ws, size_t ws_bytes,
                                        lr_stream_t stream) {
  LR_CHECK_ARG(B >= 1 && F >= 1);
  LR_CHECK_ARG(idxT && field_row_start && seg_pos && seg_rows && seg_start && n_seg && ws);
  if (B > kFS || B * F >= (int64_t(1) << 31)) return LR_ESHAPE;
  if (ws_bytes < lr_segments_fields_ws_bytes(B, F)) return LR_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  uint64_t* sorted = static_cast<uint64_t*>(ws);
  int32_t* fcount = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + align_up(static_cast<size_t>(B) * F * 8));
  const size_t lds = static_cast<size_t>(kFS) * 8 + kSortWaves * 256 * 4 + kSortWaves * 2 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(seg_field_sort_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    attr_set = true;
  }
  hipLaunchKernelGGL(seg_field_sort_kernel, dim3(F), dim3(kSortThreads), lds, s, idxT, static_cast<int>(B),
                     field_row_start, sorted, fcount);
  hipLaunchKernelGGL(seg_field_emit_kernel, dim3(F), dim3(kEmitThreads), 0, s, sorted, fcount,
                     static_cast<int>(B), F, field_row_start, seg_pos, seg_rows, seg_start, n_seg, slotT);
  return launch_status();
}
