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
