// Device-side negative sampling: the acceptance rules of the reference's host
// samplers — `negatives_from_random` (sampling/negatives.py:17-31: a negative must differ from
// its positive, <= 10 resampling rounds) and `negatives_from_unconsumed` (:55-82: per (user, item)
// up to 10 tries avoiding the positive, the negatives already drawn for it and the user's consumed
// set, then up to 10 more avoiding only the first two) — driven by a counter-based generator so
// the result is a pure function of (seed, position): deterministic, order-free, no RNG state.
// The reference's own samplers consume numpy / Python RNG streams on the host; those streams
// cannot be reproduced on a device, so this is an opt-in sampler with its own (bit-exact) oracle
// (oracle/ops_np.py:sample_negatives_counter); the host collators stay the reference-exact path.
#include "common.hpp"

namespace lr {

__host__ __device__ inline uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ bool in_sorted(const int32_t* __restrict__ a, int64_t lo, int64_t hi,
                                          int32_t x) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == x) return true;
    if (v < x) lo = mid + 1; else hi = mid;
  }
  return false;
}

constexpr int kTriesStrict = 10, kTriesTotal = 20;

__global__ __launch_bounds__(kBlock) void sample_negatives_kernel(
    const int32_t* __restrict__ users, const int32_t* __restrict__ items_pos, int64_t n,
    int num_neg, uint32_t n_items, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, uint64_t seed, int32_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride) {
    const int32_t pos = items_pos[p];
    int64_t c_lo = 0, c_hi = 0;
    if (cptr != nullptr) {
      const int32_t u = users[p];
      c_lo = cptr[u];
      c_hi = cptr[u + 1];
    }
    for (int j = 0; j < num_neg; ++j) {
      const uint64_t ctr = (static_cast<uint64_t>(p) * num_neg + j) * 32u;
      int32_t cand = 0;
      for (int t = 0; t < kTriesTotal; ++t) {
        const uint64_t z = mix64(seed + 0x9e3779b97f4a7c15ull * (ctr + t + 1));
        cand = static_cast<int32_t>((static_cast<uint64_t>(static_cast<uint32_t>(z >> 32)) * n_items) >> 32);
        bool bad = cand == pos;
        for (int jj = 0; jj < j && !bad; ++jj) bad = out[p * num_neg + jj] == cand;
        if (!bad && t < kTriesStrict && c_lo < c_hi) bad = in_sorted(cidx, c_lo, c_hi, cand);
        if (!bad) break;
      }
      out[p * num_neg + j] = cand;
    }
  }
}

}  // namespace lr

using namespace lr;

extern "C" int lr_sample_negatives_i32(const int32_t* users, const int32_t* items_pos, int64_t n,
                                       int num_neg, int32_t n_items, const int64_t* consumed_ptr,
                                       const int32_t* consumed_idx, uint64_t seed, int32_t* out,
                                       lr_stream_t stream) {
  LR_CHECK_ARG(n >= 0 && num_neg >= 1 && n_items >= 1);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG(items_pos && out);
  LR_CHECK_ARG((consumed_ptr == nullptr) == (consumed_idx == nullptr));
  LR_CHECK_ARG(consumed_ptr == nullptr || users != nullptr);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), users, items_pos, n, num_neg,
                     static_cast<uint32_t>(n_items), consumed_ptr, consumed_idx, seed, out);
  return launch_status();
}
