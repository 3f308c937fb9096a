// Full-catalog ranking of DeepFM without the (user x item) feature cross product (SURVEY 8 row f2;
// reference: recommendation/recommend.py:81-105 + preprocess.py:110-172 run the whole model on one
// materialised feature row per (user, item) pair).  Everything embedding-facing in the model is linear in
// "user-side fields + item-side fields" (recommendation/catalog.py), so a pair costs only the MLP tail:
//
//   deep[u][i] = sum_c relu( sum_k relu(P[u][k] + Q[i][k]) * W2[k][c] + b2[c] ) * v3[c] + c3
//
// with P = user part of the first Dense layer (bias included), Q = item part (cached for the catalogue),
// W2 / b2 the second Dense layer with the first BatchNorm's inference affine folded in, and v3 / c3 the third
// Dense layer (no activation, layers/dense.py:33-49) contracted with the output layer's weights of the deep
// term (deepfm.py:171-172) and the second BatchNorm folded in — a hidden stack (H1, H2, H3) collapses to ONE
// H1 x H2 product per pair plus an H2-wide weighted sum.
//
// Mapping: 4 waves per workgroup, each owning 32 items whose Q rows stay in VGPRs; the workgroup walks a block
// of users whose P rows sit in LDS (broadcast reads); per (user, 32 items): H1/2 x H2/32 v_mfma_f32_32x32x2_f32
// with the A operand relu(p + q) formed on the fly (2 VALU per reduction step) and W2 read from LDS in fragment
// order; epilogue: bias, relu, weight by v3 per lane (= output column), sum over the 32 column lanes (DPP row
// sums + one cross-row exchange).  f32 throughout; fixed summation order.
#include "common.hpp"
#include "split_bf16.hpp"

namespace lr {

using f32x16m = __attribute__((ext_vector_type(16))) float;

constexpr int kPmUB = 64;      // users per workgroup pass (P rows staged in LDS)

__device__ __forceinline__ float pm_row_sum32(float x) {   // sum over the 32 lanes of a lane half, on every lane
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
  x += __shfl_xor(x, 16);
  return x;
}

template <int H1, int H2>
__global__ __launch_bounds__(kBlock, 2) void pair_mlp_kernel(
    const float* __restrict__ P, int64_t B, const float* __restrict__ Q, int64_t N,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ v3, float c3,
    float* __restrict__ out, int64_t ld_out, int accumulate) {
  constexpr int HH = H1 / 2;          // reduction values per lane half
  constexpr int NT = H2 / 32;         // column tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* wl = reinterpret_cast<float4*>(smem);                 // [NT][HH/4][64] W2 fragments, 4 steps per read
  float* pl = reinterpret_cast<float*>(wl + NT * (HH / 4) * 64); // [kPmUB][H1]
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;

  // W2 in fragment order: lane (n = j, half h), step s -> W2[h * HH + s][t * 32 + n]
  for (int q = tid; q < NT * (HH / 4) * 64; q += kBlock) {
    const int l = q & 63, s4 = (q >> 6) % (HH / 4), t = q / (64 * (HH / 4));
    const int n = l & 31, hh = l >> 5;
    const int k0 = hh * HH + s4 * 4;
    wl[q] = make_float4(W2[(k0 + 0) * H2 + t * 32 + n], W2[(k0 + 1) * H2 + t * 32 + n],
                        W2[(k0 + 2) * H2 + t * 32 + n], W2[(k0 + 3) * H2 + t * 32 + n]);
  }
  // this wave's 32 items: my row's half of Q in registers
  const int64_t item = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * 32 + j;
  const bool item_ok = item < N;
  float qr[HH];
#pragma unroll
  for (int s = 0; s < HH; s += 4) {
    const float4 x = item_ok ? ld4(Q + item * H1 + h * HH + s) : f4_zero();
    qr[s] = x.x; qr[s + 1] = x.y; qr[s + 2] = x.z; qr[s + 3] = x.w;
  }
  float bb[NT], vv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bb[t] = b2[t * 32 + j];
    vv[t] = v3[t * 32 + j];
  }
  // rows of the 32x32 accumulator tile held by this lane half: (r&3) + 8 (r>>2) + 4h; lanes 0..15 of each half
  // write one of them each
  const int my_r = lane & 15;
  const int64_t out_item = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * 32 + (my_r & 3) + 8 * (my_r >> 2) + 4 * h;

  const int64_t ub0 = static_cast<int64_t>(blockIdx.y) * kPmUB;
  const int nu = (B - ub0) < kPmUB ? static_cast<int>(B - ub0) : kPmUB;
  for (int q = tid; q < nu * (H1 / 4); q += kBlock)
    reinterpret_cast<float4*>(pl)[q] = ld4(P + ub0 * H1 + static_cast<int64_t>(q) * 4);
  __syncthreads();

  for (int u = 0; u < nu; ++u) {
    f32x16m acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16m{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* pu = pl + u * H1 + h * HH;
#pragma unroll
    for (int s4 = 0; s4 < HH / 4; ++s4) {
      const float4 p4 = ld4(pu + s4 * 4);                         // broadcast inside the lane half
      const float a0 = fmaxf(p4.x + qr[s4 * 4 + 0], 0.f), a1 = fmaxf(p4.y + qr[s4 * 4 + 1], 0.f);
      const float a2 = fmaxf(p4.z + qr[s4 * 4 + 2], 0.f), a3 = fmaxf(p4.w + qr[s4 * 4 + 3], 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 w = wl[(t * (HH / 4) + s4) * 64 + lane];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, w.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, w.w, acc[t], 0, 0, 0);
      }
    }
    // epilogue: deep[item of reg r] = sum over the 64 / 32 column lanes of relu(acc + b2) * v3
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) x = fmaf(fmaxf(acc[t][r] + bb[t], 0.f), vv[t], x);
      x = pm_row_sum32(x);
      mine = (my_r == r) ? x : mine;
    }
    if (j < 16 && out_item < N) {
      float* o = out + (ub0 + u) * ld_out + out_item;
      *o = accumulate ? *o + mine + c3 : mine + c3;
    }
  }
}

// The same contraction as six-term split-bf16 products (split_bf16.hpp): relu(p + q) is formed in f32 and split exactly into
// three bf16 planes per (user, item) pair on the fly (8 values per lane and k-block), W2's planes sit in LDS in B-operand
// fragment order, a 32-item x 32-column x 16 block is six v_mfma_f32_32x32x16_bf16 with f32 accumulation: 96 MFMAs of 32
// matrix-pipe cycles per (user, 32 items) at (128, 64) against 128 of 64 cycles in the f32 form.  Same epilogue, fixed order.
constexpr int kPmUBsb = 32;    // users per workgroup pass of the split-bf16 form (W2's planes take 48 KB of LDS)

template <int H1, int H2>
__global__ __launch_bounds__(kBlock, 2) void pair_mlp_sb_kernel(
    const float* __restrict__ P, int64_t B, const float* __restrict__ Q, int64_t N,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ v3, float c3,
    float* __restrict__ out, int64_t ld_out, int accumulate) {
  constexpr int KB = H1 / 16;         // k-blocks of a 32 x 32 x 16 MFMA
  constexpr int NT = H2 / 32;         // column tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  sb::bf16x8* wl = reinterpret_cast<sb::bf16x8*>(smem);                        // [3 planes][NT][KB][64]
  float* pl = reinterpret_cast<float*>(wl + 3 * NT * KB * 64);                 // [kPmUBsb][H1]
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;

  // W2 planes in B-operand fragment order: lane (n = j, half h), element e of (t, kb) = W2[kb * 16 + h * 8 + e][t * 32 + n]
  for (int q = tid; q < NT * KB * 64; q += kBlock) {
    const int l = q & 63, kb = (q >> 6) % KB, t = q / (64 * KB);
    const int n = l & 31, hh = l >> 5;
    const float* src = W2 + static_cast<int64_t>(kb * 16 + hh * 8) * H2 + t * 32 + n;
    const float4 lo = make_float4(src[0], src[H2], src[2 * H2], src[3 * H2]);
    const float4 hi = make_float4(src[4 * H2], src[5 * H2], src[6 * H2], src[7 * H2]);
    sb::bf16x8 w1, w2, w3;
    sb::split8(lo, hi, w1, w2, w3);
    wl[q] = w1;
    wl[NT * KB * 64 + q] = w2;
    wl[2 * NT * KB * 64 + q] = w3;
  }
  // this wave's 32 items: lane (item j, half h) keeps Q[item][kb * 16 + h * 8 .. + 7] of every k-block
  const int64_t item = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * 32 + j;
  const bool item_ok = item < N;
  float4 qlo[KB], qhi[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    qlo[kb] = item_ok ? ld4(Q + item * H1 + kb * 16 + h * 8) : f4_zero();
    qhi[kb] = item_ok ? ld4(Q + item * H1 + kb * 16 + h * 8 + 4) : f4_zero();
  }
  float bb[NT], vv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bb[t] = b2[t * 32 + j];
    vv[t] = v3[t * 32 + j];
  }
  const int my_r = lane & 15;
  const int64_t out_item = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * 32 + (my_r & 3) + 8 * (my_r >> 2) + 4 * h;

  const int64_t ub0 = static_cast<int64_t>(blockIdx.y) * kPmUBsb;
  const int nu = (B - ub0) < kPmUBsb ? static_cast<int>(B - ub0) : kPmUBsb;
  for (int q = tid; q < nu * (H1 / 4); q += kBlock)
    reinterpret_cast<float4*>(pl)[q] = ld4(P + ub0 * H1 + static_cast<int64_t>(q) * 4);
  __syncthreads();

  auto relu4 = [](float4 a, float4 b) {
    return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
  };
  for (int u = 0; u < nu; ++u) {
    sb::f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = sb::f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* pu = pl + u * H1 + h * 8;
    // W2's fragments are re-read from LDS for every user: an opaque lane index keeps the compiler from hoisting those
    // (loop-invariant) reads out of the user loop into 192 registers (it spilled)
    int lw = lane;
    asm volatile("" : "+v"(lw));
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float4 plo = ld4(pu + kb * 16), phi = ld4(pu + kb * 16 + 4);        // broadcast inside the lane half
      sb::bf16x8 a1, a2, a3;
      sb::split8(relu4(plo, qlo[kb]), relu4(phi, qhi[kb]), a1, a2, a3);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int o = (t * KB + kb) * 64 + lw;
        sb::mfma6(acc[t], a1, a2, a3, wl[o], wl[NT * KB * 64 + o], wl[2 * NT * KB * 64 + o]);
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) x = fmaf(fmaxf(acc[t][r] + bb[t], 0.f), vv[t], x);
      x = pm_row_sum32(x);
      mine = (my_r == r) ? x : mine;
    }
    if (j < 16 && out_item < N) {
      float* o = out + (ub0 + u) * ld_out + out_item;
      *o = accumulate ? *o + mine + c3 : mine + c3;
    }
  }
}

template <int H1, int H2>
static int pair_mlp_sb_launch(const float* P, int64_t B, const float* Q, int64_t N, const float* W2, const float* b2,
                              const float* v3, float c3, float* out, int64_t ld_out, int accumulate, hipStream_t s) {
  const size_t lds = static_cast<size_t>(3) * (H2 / 32) * (H1 / 16) * 64 * 16 + static_cast<size_t>(kPmUBsb) * H1 * 4;
  auto kern = pair_mlp_sb_kernel<H1, H2>;
  static bool set = false;
  if (!set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    set = true;
  }
  const dim3 grid(static_cast<unsigned>(ceil_div(N, 128)), static_cast<unsigned>(ceil_div(B, kPmUBsb)));
  hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate);
  return launch_status();
}

template <int H1, int H2>
static int pair_mlp_launch(const float* P, int64_t B, const float* Q, int64_t N, const float* W2, const float* b2,
                           const float* v3, float c3, float* out, int64_t ld_out, int accumulate, hipStream_t s) {
  const size_t lds = static_cast<size_t>(H2 / 32) * (H1 / 8) * 64 * 16 + static_cast<size_t>(kPmUB) * H1 * 4;
  auto kern = pair_mlp_kernel<H1, H2>;
  static bool set = false;
  if (!set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    set = true;
  }
  const dim3 grid(static_cast<unsigned>(ceil_div(N, 128)), static_cast<unsigned>(ceil_div(B, kPmUB)));
  hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate);
  return launch_status();
}

}  // namespace lr

using namespace lr;

extern "C" int lr_pair_mlp_supported(int H1, int H2) {
  return ((H1 == 128 || H1 == 64) && (H2 == 64 || H2 == 32)) ? 1 : 0;
}

extern "C" int lr_pair_mlp_f32(const float* P, int64_t B, const float* Q, int64_t N, int H1, const float* W2,
                               const float* b2, int H2, const float* v3, float c3, float* out, int64_t ld_out,
                               int accumulate, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && N >= 0 && ld_out >= N);
  if (B == 0 || N == 0) return LR_OK;
  LR_CHECK_ARG(P && Q && W2 && b2 && v3 && out);
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(P) % 16 == 0 && reinterpret_cast<uintptr_t>(Q) % 16 == 0);
  if (!lr_pair_mlp_supported(H1, H2) || ceil_div(B, kPmUB) > 65535) return LR_ESHAPE;
  hipStream_t s = as_stream(stream);
  if (H1 == 128 && H2 == 64) return pair_mlp_launch<128, 64>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  if (H1 == 128 && H2 == 32) return pair_mlp_launch<128, 32>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  if (H1 == 64 && H2 == 64) return pair_mlp_launch<64, 64>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  return pair_mlp_launch<64, 32>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
}

// the same contract with the H1 x H2 product taken as six-term split-bf16 products (f32 accumulation): equal to lr_pair_mlp_f32
// to f32 rounding, not bit for bit
extern "C" int lr_pair_mlp_sb_f32(const float* P, int64_t B, const float* Q, int64_t N, int H1, const float* W2,
                                  const float* b2, int H2, const float* v3, float c3, float* out, int64_t ld_out,
                                  int accumulate, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && N >= 0 && ld_out >= N);
  if (B == 0 || N == 0) return LR_OK;
  LR_CHECK_ARG(P && Q && W2 && b2 && v3 && out);
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(P) % 16 == 0 && reinterpret_cast<uintptr_t>(Q) % 16 == 0);
  if (!lr_pair_mlp_supported(H1, H2) || ceil_div(B, kPmUBsb) > 65535) return LR_ESHAPE;
  hipStream_t s = as_stream(stream);
  if (H1 == 128 && H2 == 64) return pair_mlp_sb_launch<128, 64>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  if (H1 == 128 && H2 == 32) return pair_mlp_sb_launch<128, 32>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  if (H1 == 64 && H2 == 64) return pair_mlp_sb_launch<64, 64>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
  return pair_mlp_sb_launch<64, 32>(P, B, Q, N, W2, b2, v3, c3, out, ld_out, accumulate, s);
}
