// Device-side construction of LightGCN's normalised bipartite Laplacian  A^ = D^-1/2 A D^-1/2  as CSR —
// replaces LightGCNModel._build_laplacian_matrix (algorithms/torch_modules/lightgcn_module.py:36-61: a scipy dok
// matrix filled user by user, lil / csr conversions, two sparse products; minutes at 10^8 interactions).
//
//   A = [[0, R], [R^T, 0]] over nodes [users | items], R[u, i] = 1 for every distinct interaction (u, i)
//   (dok assignment: repeats collapse);  val(r, c) = deg(r)^-1/2 * deg(c)^-1/2, deg = row sums of A
//   (isolated nodes: deg^-1/2 := 0, lightgcn_module.py:51-53).
//
// Integer work is exact: (u << 32 | i) keys radix-sorted on the device (rocPRIM, the only library code), repeats
// dropped by head flags + scan, the item block obtained by a second sort of (i << 32 | u) whose payload is the
// entry's position in the user block — which is also the transpose map (A^T.val = A.val[tperm], needed for edge
// dropout).  Row pointers come from the sorted keys' boundaries.  Nothing is read by the host except the final
// count.  Values: deg^-1/2 formed in double and rounded once to fp32, product in fp32 (the reference multiplies fp32
// diagonals into an fp32 matrix).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace lr {

static inline size_t lap_align(size_t x) { return (x + 255) / 256 * 256; }

__global__ __launch_bounds__(kBlock) void lap_keys_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ it,
                                                          int64_t E, int64_t nu, int64_t ni, uint64_t* __restrict__ keys) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const int32_t a = u[e], b = it[e];
    const bool ok = a >= 0 && a < nu && b >= 0 && b < ni;
    keys[e] = ok ? (static_cast<uint64_t>(a) << 32) | static_cast<uint32_t>(b) : ~0ull;
  }
}

// head[e] = 1 for the first occurrence of a valid key in the sorted list
__global__ __launch_bounds__(kBlock) void lap_heads_kernel(const uint64_t* __restrict__ keys, int64_t E,
                                                           int32_t* __restrict__ head) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const uint64_t k = keys[e];
    head[e] = (k != ~0ull && (e == 0 || keys[e - 1] != k)) ? 1 : 0;
  }
}

// compact the distinct pairs (rank = inclusive scan of head, 1-based) and form the transposed keys; the last
// valid head publishes the count
__global__ __launch_bounds__(kBlock) void lap_compact_kernel(const uint64_t* __restrict__ keys,
                                                             const int32_t* __restrict__ head,
                                                             const int32_t* __restrict__ rank, int64_t E,
                                                             uint64_t* __restrict__ uniq, uint64_t* __restrict__ tkeys,
                                                             int64_t* __restrict__ n_pairs) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    if (e == E - 1) n_pairs[0] = rank[e];
    if (!head[e]) continue;
    const uint64_t k = keys[e];
    const int64_t p = rank[e] - 1;
    uniq[p] = k;
    tkeys[p] = (k << 32) | (k >> 32);
  }
}

// rowptr entries of one block from its sorted (row << 32 | col) keys: rows in (row of key[p-1], row of key[p]] start
// at p; rows past the last key start at n.  `base` = first entry of the block, `row0` = first node of the block.
__global__ __launch_bounds__(kBlock) void lap_rowptr_kernel(const uint64_t* __restrict__ skeys,
                                                            const int64_t* __restrict__ n_ptr, int64_t n_rows,
                                                            int64_t row0, int64_t base_mul,
                                                            int64_t* __restrict__ rowptr, int write_end) {
  const int64_t n = n_ptr[0];
  const int64_t base = base_mul * n;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  for (int64_t p = t0; p <= n; p += stride) {
    const int64_t hi = p < n ? static_cast<int64_t>(skeys[p] >> 32) : n_rows - 1;
    const int64_t lo = p > 0 ? static_cast<int64_t>(skeys[p - 1] >> 32) + 1 : 0;
    if (p < n) {
      for (int64_t r = lo; r <= hi; ++r) rowptr[row0 + r] = base + p;
    } else {
      for (int64_t r = lo; r < n_rows; ++r) rowptr[row0 + r] = base + n;
      if (write_end) rowptr[row0 + n_rows] = base + n;
    }
  }
}

__device__ __forceinline__ float lap_dinv(int64_t deg) {
  return deg > 0 ? static_cast<float>(1.0 / sqrt(static_cast<double>(deg))) : 0.f;
}

// col / val of both blocks (+ the transpose map).  User block entry p = uniq[p] = (u, i); item block entry q =
// sorted tkeys[q] = (i, u) whose payload src[q] is its user-block position.
__global__ __launch_bounds__(kBlock) void lap_fill_kernel(const uint64_t* __restrict__ uniq,
                                                          const uint64_t* __restrict__ tsorted,
                                                          const int32_t* __restrict__ src,
                                                          const int64_t* __restrict__ n_ptr, int64_t nu,
                                                          const int64_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                          float* __restrict__ val, int32_t* __restrict__ tperm) {
  const int64_t n = n_ptr[0];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < 2 * n; e += stride) {
    const bool ublock = e < n;
    const uint64_t k = ublock ? uniq[e] : tsorted[e - n];
    const int64_t hi = static_cast<int64_t>(k >> 32), lo = static_cast<int64_t>(k & 0xffffffffull);
    const int64_t r = ublock ? hi : nu + hi, c = ublock ? nu + lo : lo;
    col[e] = static_cast<int32_t>(c);
    val[e] = lap_dinv(rowptr[r + 1] - rowptr[r]) * lap_dinv(rowptr[c + 1] - rowptr[c]);
    if (tperm != nullptr && !ublock) {
      const int32_t p = src[e - n];
      tperm[e] = p;
      tperm[p] = static_cast<int32_t>(e);
    }
  }
}

static inline int bits_for(int64_t n) {
  int b = 1;
  while (b < 32 && (int64_t(1) << b) < n) ++b;
  return b;
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_csr_laplacian_ws_bytes(int64_t E) {
  if (E < 0) E = 0;
  const size_t e = static_cast<size_t>(E);
  // keys (2 x u64) + transposed keys (2 x u64) + head / rank / payload in / payload out (4 x i32) + count + rocPRIM
  return 4 * lap_align(e * 8) + 4 * lap_align(e * 4) + 256 + lap_align(e * 16) + (size_t(16) << 20);
}

extern "C" int lr_csr_laplacian_build(const int32_t* users, const int32_t* items, int64_t E, int64_t n_users,
                                      int64_t n_items, int64_t* rowptr, int32_t* col, float* val, int32_t* tperm,
                                      int64_t* n_pairs, void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(E >= 0 && n_users >= 1 && n_items >= 1 && rowptr && n_pairs);
  LR_CHECK_ARG(n_users < (int64_t(1) << 31) && n_items < (int64_t(1) << 31) && n_users + n_items < (int64_t(1) << 31));
  LR_CHECK_ARG(2 * E < (int64_t(1) << 31));           // int32 entry positions (col / tperm)
  hipStream_t s = as_stream(stream);
  const int64_t n_nodes = n_users + n_items;
  if (E == 0) {
    hipError_t e0 = hipMemsetAsync(rowptr, 0, static_cast<size_t>(n_nodes + 1) * 8, s);
    if (e0 != hipSuccess) return static_cast<int>(e0);
    e0 = hipMemsetAsync(n_pairs, 0, 8, s);
    return e0 == hipSuccess ? LR_OK : static_cast<int>(e0);
  }
  LR_CHECK_ARG(users && items && col && val && ws);
  if (ws_bytes < lr_csr_laplacian_ws_bytes(E)) return LR_EWORKSPACE;
  char* p = static_cast<char*>(ws);
  const size_t a8 = lap_align(static_cast<size_t>(E) * 8), a4 = lap_align(static_cast<size_t>(E) * 4);
  uint64_t* k0 = reinterpret_cast<uint64_t*>(p); p += a8;
  uint64_t* k1 = reinterpret_cast<uint64_t*>(p); p += a8;
  uint64_t* t0 = reinterpret_cast<uint64_t*>(p); p += a8;
  uint64_t* t1 = reinterpret_cast<uint64_t*>(p); p += a8;
  int32_t* head = reinterpret_cast<int32_t*>(p); p += a4;
  int32_t* rank = reinterpret_cast<int32_t*>(p); p += a4;
  int32_t* pay0 = reinterpret_cast<int32_t*>(p); p += a4;     // unused slot kept for alignment of the layout
  int32_t* src = reinterpret_cast<int32_t*>(p); p += a4;
  p += 256;
  void* prim = p;
  const size_t prim_bytes = ws_bytes - static_cast<size_t>(p - static_cast<char*>(ws));
  (void)pay0;
  const int grid = grid_for(E, kBlock);
  const size_t n = static_cast<size_t>(E);

  hipLaunchKernelGGL(lap_keys_kernel, dim3(grid), dim3(kBlock), 0, s, users, items, E, n_users, n_items, k0);
  // invalid keys are all-ones: sorting the full 64 bits keeps them last
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, need, k0, k1, n, 0u, 64u, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (need > prim_bytes) return LR_EWORKSPACE;
  e = rocprim::radix_sort_keys(prim, need, k0, k1, n, 0u, 64u, s);
  if (e != hipSuccess) return static_cast<int>(e);

  hipLaunchKernelGGL(lap_heads_kernel, dim3(grid), dim3(kBlock), 0, s, k1, E, head);
  need = 0;
  e = rocprim::inclusive_scan(nullptr, need, head, rank, n, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (need > prim_bytes) return LR_EWORKSPACE;
  e = rocprim::inclusive_scan(prim, need, head, rank, n, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  // uniq -> k0 (the unsorted keys are no longer needed), transposed keys -> t0; entries past the count keep stale
  // bytes: every later kernel stops at the device-side count, and the second sort is told to ignore them by
  // pre-filling t0 with all-ones
  e = hipMemsetAsync(t0, 0xff, static_cast<size_t>(E) * 8, s);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(lap_compact_kernel, dim3(grid), dim3(kBlock), 0, s, k1, head, rank, E, k0, t0, n_pairs);

  rocprim::counting_iterator<int32_t> iota(0);
  need = 0;
  const unsigned tbits = 32u + static_cast<unsigned>(bits_for(n_items));
  (void)tbits;   // all 64 bits are sorted so that the all-ones filler stays behind every real key
  e = rocprim::radix_sort_pairs(nullptr, need, t0, t1, iota, src, n, 0u, 64u, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (need > prim_bytes) return LR_EWORKSPACE;
  e = rocprim::radix_sort_pairs(prim, need, t0, t1, iota, src, n, 0u, 64u, s);
  if (e != hipSuccess) return static_cast<int>(e);

  const int gr = grid_for(E + 1, kBlock);
  hipLaunchKernelGGL(lap_rowptr_kernel, dim3(gr), dim3(kBlock), 0, s, k0, n_pairs, n_users, int64_t(0), int64_t(0),
                     rowptr, 0);
  hipLaunchKernelGGL(lap_rowptr_kernel, dim3(gr), dim3(kBlock), 0, s, t1, n_pairs, n_items, n_users, int64_t(1),
                     rowptr, 1);
  hipLaunchKernelGGL(lap_fill_kernel, dim3(grid_for(2 * E, kBlock)), dim3(kBlock), 0, s, k0, t1, src, n_pairs, n_users,
                     rowptr, col, val, tperm);
  return launch_status();
}
