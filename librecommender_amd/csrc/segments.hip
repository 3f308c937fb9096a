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
    int32_t* __restrict__ seg_start, int32_t* __restrict__ n_seg, int32_t* __restrict__ slotT,
    int32_t* __restrict__ runT) {
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
      // run number of the entry = heads at or before it, less one (a run that began in an earlier wave's range continues)
      if (runT != nullptr) runT[static_cast<int64_t>(f) * B + p] = valid ? run + __popcll(hb & (lt | (1ull << lane))) - 1 : -1;
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

// -----------------------------------------------------------------------------------------
// Own LSD radix sort of (key, position) pairs for id streams up to kRsMaxN entries — the DIN / TwoTower / feature-layer
// steps sort 0.1 - 3 M ids per step, where rocPRIM picks a block sort + ~18 merge passes (0.15 ms of 5 us launches at
// 442 k keys) or, above ~1 M keys, its onesweep form whose memsets cannot be captured in a hipGraph.  8-bit digits,
// ceil(bits(V) / 8) passes of three launches (kernels only, no memset):
//   rs_hist_kernel    : digit counts of each 2,048-key chunk                      -> hist[digit][chunk]
//   rs_scan_kernel    : one wave per digit: exclusive scan of its counts over the chunks, in place, + the digit total (the
//                       scatter adds the digits' bases itself: 256 totals scanned in LDS)
//   rs_scatter_kernel : each wave owns 512 contiguous keys of the chunk and ranks them in order (wave-level digit match,
//                       running per-(wave, digit) bases in LDS): stable, so equal rows keep ascending positions
// -----------------------------------------------------------------------------------------
constexpr int kRsKPT = 8;                       // keys per thread
constexpr int kRsKPB = kBlock * kRsKPT;         // keys per workgroup
constexpr int kRsWaves = kBlock / 64;
constexpr int kRsKPW = kRsKPB / kRsWaves;       // contiguous keys per wave
constexpr int64_t kRsMaxN = int64_t(1) << 22;

__global__ __launch_bounds__(kBlock) void rs_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, int nblk,
                                                         int32_t* __restrict__ hist) {
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  cnt[tid] = 0;                                  // kBlock == 256 digits
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kRsKPB;
#pragma unroll
  for (int st = 0; st < kRsKPT; ++st) {
    const int64_t i = base + st * kBlock + tid;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1);
  }
  __syncthreads();
  hist[static_cast<int64_t>(tid) * nblk + blockIdx.x] = cnt[tid];
}

// one wave per digit: exclusive scan of the digit's counts over the chunks (in place) + the digit's total
__global__ __launch_bounds__(64) void rs_scan_kernel(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ tot) {
  const int d = blockIdx.x, lane = threadIdx.x;
  int32_t* row = hist + static_cast<int64_t>(d) * nblk;
  int carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const int v = b < nblk ? row[b] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (b < nblk) row[b] = carry + x - v;
    carry += __shfl(x, 63);
  }
  if (lane == 0) tot[d] = carry;
}

__global__ __launch_bounds__(kBlock) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ pos_in,
                                                            int64_t n, int shift, int nblk, const int32_t* __restrict__ hist,
                                                            const int32_t* __restrict__ tot, uint32_t* __restrict__ keys_out,
                                                            int32_t* __restrict__ pos_out) {
  __shared__ int run[kRsWaves][256];
  __shared__ int dsum[256];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
  const int my_tot = tot[tid];
  dsum[tid] = my_tot;
#pragma unroll
  for (int w = 0; w < kRsWaves; ++w) run[w][tid] = 0;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {              // inclusive scan of the 256 digit totals
    const int v = tid >= off ? dsum[tid - off] : 0;
    __syncthreads();
    dsum[tid] += v;
    __syncthreads();
  }
  const int dbase = dsum[tid] - my_tot;                  // keys of smaller digits, all chunks
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kRsKPB + wid * kRsKPW;
  uint32_t k[kRsKPT];
#pragma unroll
  for (int st = 0; st < kRsKPT; ++st) {
    const int64_t i = base + st * 64 + lane;
    k[st] = i < n ? keys_in[i] : 0u;
    if (i < n) atomicAdd(&run[wid][(k[st] >> shift) & 255u], 1);
  }
  __syncthreads();
  {   // counts -> first output slot of every (wave, digit): chunk base, then the waves in order
    int g = dbase + hist[static_cast<int64_t>(tid) * nblk + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const int c = run[w][tid];
      run[w][tid] = g;
      g += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < kRsKPT; ++st) {
    const int64_t i = base + st * 64 + lane;
    const bool ok = i < n;
    const uint32_t d = (k[st] >> shift) & 255u;
    const uint64_t okb = __ballot(ok);
    const uint64_t mask = match_digit(d) & okb;            // live lanes of this step holding my digit
    const int rank = __popcll(mask & lt);
    int dst = 0;
    if (ok) dst = run[wid][d] + rank;
    asm volatile("" ::: "memory");                          // every lane has read its base before a leader moves it
    if (ok && rank == 0) run[wid][d] = dst + __popcll(mask);
    asm volatile("" ::: "memory");
    if (ok) {
      keys_out[dst] = k[st];
      pos_out[dst] = pos_in != nullptr ? pos_in[i] : static_cast<int32_t>(i);
    }
  }
}

// sorts (keys_in, position) by key into (keys_out, pos_out); tmp_keys / tmp_pos are the ping-pong partners of the outputs,
// hist holds 256 * ceil(n / kRsKPB) ints.  keys_in is only read.
static int own_radix_sort(const uint32_t* keys_in, uint32_t* keys_out, int32_t* pos_out, uint32_t* tmp_keys, int32_t* tmp_pos,
                          int32_t* hist, int64_t n, int bits, hipStream_t s) {
  const int passes = (bits + 7) / 8;
  const int nblk = static_cast<int>((n + kRsKPB - 1) / kRsKPB);
  int32_t* tot = hist + static_cast<int64_t>(nblk) * 256;       // 256 digit totals behind the table
  // the last pass must write (keys_out, pos_out): walk back from there
  const uint32_t* src_k = keys_in;
  const int32_t* src_p = nullptr;                 // first pass: positions are 0 .. n-1
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    uint32_t* dst_k = to_out ? keys_out : tmp_keys;
    int32_t* dst_p = to_out ? pos_out : tmp_pos;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(kBlock), 0, s, src_k, n, 8 * p, nblk, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(64), 0, s, hist, nblk, tot);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblk), dim3(kBlock), 0, s, src_k, src_p, n, 8 * p, nblk, hist, tot, dst_k, dst_p);
    src_k = dst_k;
    src_p = dst_p;
  }
  return launch_status();
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
  hipError_t e = hipSuccess;
  if (n <= kRsMaxN) {
    // own sort: `rank` doubles as the positions' ping-pong buffer (it is only written after the sort), the histogram
    // table sits at the head of the scratch area
    const size_t hist_bytes = align_up((static_cast<size_t>((n + kRsKPB - 1) / kRsKPB) + 1) * 256 * 4);
    if (hist_bytes + a > w.prim_bytes) return LR_EWORKSPACE;
    const int rc = own_radix_sort(w.keys_in, w.keys_out, seg_pos, reinterpret_cast<uint32_t*>(static_cast<char*>(w.prim) + hist_bytes),
                                  w.rank, static_cast<int32_t*>(w.prim), n, bits, s);
    if (rc != LR_OK) return rc;
  } else {
    rocprim::counting_iterator<int32_t> iota(0);
    e = rocprim::radix_sort_pairs(nullptr, need, w.keys_in, w.keys_out, iota, seg_pos, static_cast<size_t>(n), 0u,
                                  static_cast<unsigned>(bits), s);
    if (e != hipSuccess) return static_cast<int>(e);
    if (need > w.prim_bytes) return LR_EWORKSPACE;
    e = rocprim::radix_sort_pairs(w.prim, need, w.keys_in, w.keys_out, iota, seg_pos, static_cast<size_t>(n), 0u,
                                  static_cast<unsigned>(bits), s);
    if (e != hipSuccess) return static_cast<int>(e);
  }

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

static int build_fields_impl(const int32_t* idxT, int64_t B, int F, const int32_t* field_row_start, int32_t* seg_pos,
                             int32_t* seg_rows, int32_t* seg_start, int32_t* n_seg, int32_t* slotT, int32_t* runT, void* ws,
                             size_t ws_bytes, lr_stream_t stream) {
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
                     static_cast<int>(B), F, field_row_start, seg_pos, seg_rows, seg_start, n_seg, slotT, runT);
  return launch_status();
}

extern "C" int lr_segments_build_fields(const int32_t* idxT, int64_t B, int F, const int32_t* field_row_start,
                                        int32_t* seg_pos, int32_t* seg_rows, int32_t* seg_start, int32_t* n_seg,
                                        int32_t* slotT, void* ws, size_t ws_bytes, lr_stream_t stream) {
  return build_fields_impl(idxT, B, F, field_row_start, seg_pos, seg_rows, seg_start, n_seg, slotT, nullptr, ws, ws_bytes,
                           stream);
}

extern "C" int lr_segments_build_fields_runs(const int32_t* idxT, int64_t B, int F, const int32_t* field_row_start,
                                             int32_t* seg_pos, int32_t* seg_rows, int32_t* seg_start, int32_t* n_seg,
                                             int32_t* slotT, int32_t* runT, void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(runT != nullptr);
  return build_fields_impl(idxT, B, F, field_row_start, seg_pos, seg_rows, seg_start, n_seg, slotT, runT, ws, ws_bytes,
                           stream);
}

// -----------------------------------------------------------------------------------------
// Owner partition of a batch's distinct rows (row-sharded tables, round-robin: owner = row % W, the owner's local row =
// row / W).  Input: the distinct rows in ANY order (here: ascending, from the field-wise sort).  Output: the STABLE
// owner-major order the all-to-all needs — perm[r] = place of row r in it, send_ids[perm[r]] = local row, counts[o] = rows
// asked from owner o (int64, what the exchange reads on the host), counts[W] = their total.  Three small launches over
// chunks of kOwnChunk rows: per-chunk owner counts, one block turning them into bases (owner-major, then chunk), placement
// with the same in-chunk ranking.  No sort: with the field-wise segment build this replaces the device-wide radix sort of
// the owner-major keys (3.3 M keys, 0.27 ms) by ~20 us.  n_seg is read on the device; chunks past it do nothing.
// -----------------------------------------------------------------------------------------
constexpr int kOwnChunk = 1024;         // rows per workgroup (256 threads, 4 sweeps)
constexpr int kOwnMaxW = 64;

// in-chunk counts per (sweep, wave, owner) in LDS; returns this thread's rank among earlier rows of the same owner in its wave
template <bool kPlace>
__device__ __forceinline__ void owner_chunk(const int32_t* __restrict__ rows, int n, int W, int chunk, int (*cnt)[4][kOwnMaxW],
                                            int* my_owner, int* my_rank) {
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = chunk * kOwnChunk + it * kBlock + tid;
    const int o = i < n ? static_cast<int>(static_cast<uint32_t>(rows[i]) % static_cast<uint32_t>(W)) : -1;
    int rank = 0;
    for (int w = 0; w < W; ++w) {             // uniform loop: one ballot per owner
      const uint64_t mk = __ballot(o == w);
      if (o == w) rank = __popcll(mk & lt);
      if (lane == 0) cnt[it][wid][w] = __popcll(mk);
    }
    if (kPlace) { my_owner[it] = o; my_rank[it] = rank; }
  }
}

__global__ __launch_bounds__(kBlock) void owner_count_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ n_seg,
                                                             int W, int32_t* __restrict__ chunk_cnt) {
  __shared__ int cnt[4][4][kOwnMaxW];
  const int n = n_seg[0];
  const int chunk = blockIdx.x;
  if (chunk * kOwnChunk >= n) {               // nothing here (the scan still reads the zeros)
    for (int w = threadIdx.x; w < W; w += kBlock) chunk_cnt[static_cast<int64_t>(chunk) * W + w] = 0;
    return;
  }
  owner_chunk<false>(rows, n, W, chunk, cnt, nullptr, nullptr);
  __syncthreads();
  for (int w = threadIdx.x; w < W; w += kBlock) {
    int t = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int v = 0; v < 4; ++v) t += cnt[it][v][w];
    chunk_cnt[static_cast<int64_t>(chunk) * W + w] = t;
  }
}

// one block: chunk_cnt [nchunk][W] -> exclusive bases in owner-major order (in place); counts[0..W) and the total
__global__ __launch_bounds__(1024) void owner_scan_kernel(int32_t* __restrict__ chunk_cnt, int nchunk, int W,
                                                          int64_t* __restrict__ counts) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  const int per = (nchunk + 1023) / 1024;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int w = 0; w < W; ++w) {
    int loc = 0;
    for (int j = 0; j < per; ++j) {
      const int c = tid * per + j;
      if (c < nchunk) loc += chunk_cnt[static_cast<int64_t>(c) * W + w];
    }
    part[tid] = loc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive Hillis-Steele over the 1024 partial sums
      const int v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    const int base = carry;
    int run = base + part[tid] - loc;
    for (int j = 0; j < per; ++j) {
      const int c = tid * per + j;
      if (c < nchunk) {
        const int v = chunk_cnt[static_cast<int64_t>(c) * W + w];
        chunk_cnt[static_cast<int64_t>(c) * W + w] = run;
        run += v;
      }
    }
    __syncthreads();
    if (tid == 1023) {
      counts[w] = part[1023];
      carry = base + part[1023];
    }
    __syncthreads();
  }
  if (tid == 0) counts[W] = carry;
}

__global__ __launch_bounds__(kBlock) void owner_place_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ n_seg,
                                                             int W, const int32_t* __restrict__ chunk_base,
                                                             int32_t* __restrict__ perm, int32_t* __restrict__ send_ids) {
  __shared__ int cnt[4][4][kOwnMaxW];
  const int n = n_seg[0];
  const int chunk = blockIdx.x;
  if (chunk * kOwnChunk >= n) return;
  int own[4], rank[4];
  owner_chunk<true>(rows, n, W, chunk, cnt, own, rank);
  __syncthreads();
  const int tid = threadIdx.x, wid = tid >> 6;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int o = own[it];
    if (o < 0) continue;
    int place = chunk_base[static_cast<int64_t>(chunk) * W + o] + rank[it];
    for (int jt = 0; jt <= it; ++jt)
      for (int v = 0; v < 4; ++v)
        if (jt < it || v < wid) place += cnt[jt][v][o];
    const int i = chunk * kOwnChunk + it * kBlock + tid;
    perm[i] = place;
    send_ids[place] = static_cast<int32_t>(static_cast<uint32_t>(rows[i]) / static_cast<uint32_t>(W));
  }
}

extern "C" size_t lr_owner_partition_ws_bytes(int64_t n_max, int W) {
  if (n_max < 0 || W < 1) return 0;
  return align_up(static_cast<size_t>((n_max + kOwnChunk - 1) / kOwnChunk + 1) * W * 4);
}

extern "C" int lr_owner_partition_i32(const int32_t* rows, const int32_t* n_seg, int64_t n_max, int W, int32_t* perm,
                                      int32_t* send_ids, int64_t* counts, void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(n_seg && counts && n_max >= 0 && W >= 1);
  if (W > kOwnMaxW || n_max >= (int64_t(1) << 31)) return LR_ESHAPE;
  if (ws_bytes < lr_owner_partition_ws_bytes(n_max, W)) return LR_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  const int nchunk = static_cast<int>((n_max + kOwnChunk - 1) / kOwnChunk);
  int32_t* chunk_cnt = static_cast<int32_t*>(ws);
  if (nchunk > 0) {
    LR_CHECK_ARG(rows && perm && send_ids && ws);
    hipLaunchKernelGGL(owner_count_kernel, dim3(nchunk), dim3(kBlock), 0, s, rows, n_seg, W, chunk_cnt);
  }
  hipLaunchKernelGGL(owner_scan_kernel, dim3(1), dim3(1024), 0, s, chunk_cnt, nchunk, W, counts);
  if (nchunk > 0)
    hipLaunchKernelGGL(owner_place_kernel, dim3(nchunk), dim3(kBlock), 0, s, rows, n_seg, W, chunk_cnt, perm, send_ids);
  return launch_status();
}
