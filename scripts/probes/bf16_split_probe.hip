// Evidence for DESIGN section 8 item 6(a) (round 4; NOT part of the library): could the first layer's three f32 contractions run
// as split-bf16 MFMA products at f32-level error?
//  Part A — issue cost: v_mfma_f32_32x32x16_bf16 (32,768 FLOP, floor 32 cycles) with NV v_fma_f32 behind every MFMA, one wave per
//           SIMD: does the VALU work hide behind a bf16 MFMA (it does not behind an f32 MFMA: profiles/r04_mfma_issue_probe.txt)?
//  Part B — accuracy: one 32x32 output tile of A[32 x K] * B[K x 32], K = 12,928 (202 fields x 64, the first layer's reduction),
//           A ~ N(0, 0.1) (embedding rows), B ~ N(0, 0.05) (weights), against an f64 reference:
//             f32   : v_mfma_f32_32x32x2_f32 chain (what the library runs)
//             x3    : a = a1 + a2, b = b1 + b2 (bf16 each): a1 b1 + a1 b2 + a2 b1
//             x6    : three-way splits, the six largest cross terms
//             x9    : all nine
//           The k index of an operand element only has to be the SAME function of (lane half, element) for A and B, so the
//           probe does not depend on the instruction's documented k layout.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// ---------------- Part A ----------------
template <int NV, int NACC = 2>
__global__ __launch_bounds__(256, 1) void issue_probe(unsigned long long* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float c = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (NACC == 1 || (m & 1) == 0)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(c));
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[1] = 1;
  if ((threadIdx.x & 63) == 0) atomicAdd(&out[0], t1 - t0);
}

template <int NV, int NACC = 2>
void run_issue(unsigned long long* out) {
  const int iters = 4000, blocks = 256, threads = 256;
  hipMemset(out, 0, 16);
  issue_probe<NV, NACC><<<blocks, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipMemset(out, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  issue_probe<NV, NACC><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  const double waves = blocks * (threads / 64.0), mf = iters * 16.0;
  // s_memtime ticks at 100 MHz on this part: report ns per MFMA per SIMD from the wall clock as well
  printf("bf16 MFMA (%d accumulator%s) + %2d v_fma each: %7.2f ns per MFMA per SIMD (wall), memtime ticks/MFMA %.3f\n", NACC, NACC == 1 ? ", every MFMA depends on the one before" : "s alternating", NV,
         ms * 1e6 / mf, static_cast<double>(h[0]) / waves / mf);
}

// ---------------- Part B ----------------
static inline uint16_t f2bf(float f) {          // round to nearest even
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(r >> 16);
}
static inline float bf2f(uint16_t h) { uint32_t u = static_cast<uint32_t>(h) << 16; float f; memcpy(&f, &u, 4); return f; }

// planes: [P][K][32] bf16 (row / col index fastest), f32: [K][32]
__global__ __launch_bounds__(64) void tile_f32(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C) {
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(k + h) * 32 + j], B[(k + h) * 32 + j], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}
__global__ __launch_bounds__(64) void tile_bf16(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Bp, int K, int nterms,
                                               const int* __restrict__ ta, const int* __restrict__ tb, float* __restrict__ C) {
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 16) {
    for (int t = nterms - 1; t >= 0; --t) {      // smallest terms first
      bf16x8 a, b;
      uint16_t ua[8], ub[8];
      for (int e = 0; e < 8; ++e) {
        ua[e] = Ap[(static_cast<size_t>(ta[t]) * K + k + 8 * h + e) * 32 + j];
        ub[e] = Bp[(static_cast<size_t>(tb[t]) * K + k + 8 * h + e) * 32 + j];
      }
      memcpy(&a, ua, 16); memcpy(&b, ub, 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
  unsigned long long* out; hipMalloc(&out, 16);
  printf("== Part A: issue cost behind a bf16 MFMA (floor: 32 cycles = 13.3 ns at 2.4 GHz; f32 MFMA: 64 cycles, +13.5 for the first v_fma, +4.4 each further)\n");
  run_issue<0>(out); run_issue<1>(out); run_issue<2>(out); run_issue<4>(out); run_issue<6>(out); run_issue<8>(out); run_issue<12>(out);
  run_issue<0, 1>(out); run_issue<2, 1>(out);

  printf("== Part B: error of one 32x32 tile, K = 12928, against f64\n");
  const int K = 12928;
  std::mt19937 gen(7);
  std::normal_distribution<float> na(0.f, 0.1f), nb(0.f, 0.05f);
  std::vector<float> A(K * 32), B(K * 32);
  for (auto& v : A) v = na(gen);
  for (auto& v : B) v = nb(gen);
  std::vector<double> ref(32 * 32, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) ref[i * 32 + j] += static_cast<double>(A[k * 32 + i]) * static_cast<double>(B[k * 32 + j]);
  double rms = 0; for (double v : ref) rms += v * v; rms = std::sqrt(rms / 1024);
  // host f32 fma chain in k order (what a scalar f32 loop gives)
  std::vector<float> hostf(1024, 0.f);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) hostf[i * 32 + j] = fmaf(A[k * 32 + i], B[k * 32 + j], hostf[i * 32 + j]);
  auto report = [&](const char* name, const float* c) {
    double se = 0, mx = 0;
    for (int q = 0; q < 1024; ++q) { const double d = c[q] - ref[q]; se += d * d; mx = std::fmax(mx, std::fabs(d)); }
    printf("%-34s rms err / rms result %.3e   max abs err %.3e   (rms result %.4f)\n", name, std::sqrt(se / 1024) / rms, mx, rms);
  };
  report("host f32 fma chain", hostf.data());
  // three-way bf16 split
  std::vector<uint16_t> Ap(3 * K * 32), Bp(3 * K * 32);
  auto split = [&](const std::vector<float>& X, std::vector<uint16_t>& P) {
    for (size_t q = 0; q < X.size(); ++q) {
      const float x = X[q];
      const uint16_t h1 = f2bf(x); const float r1 = x - bf2f(h1);
      const uint16_t h2 = f2bf(r1); const float r2 = r1 - bf2f(h2);
      const uint16_t h3 = f2bf(r2);
      P[q] = h1; P[X.size() + q] = h2; P[2 * X.size() + q] = h3;
    }
  };
  split(A, Ap); split(B, Bp);
  float *dA, *dB, *dC; uint16_t *dAp, *dBp; int *dta, *dtb;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  hipMalloc(&dAp, Ap.size() * 2); hipMalloc(&dBp, Bp.size() * 2); hipMalloc(&dta, 64); hipMalloc(&dtb, 64);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> c(1024);
  tile_f32<<<1, 64>>>(dA, dB, K, dC); hipMemcpy(c.data(), dC, 4096, hipMemcpyDeviceToHost);
  report("v_mfma_f32_32x32x2_f32 chain", c.data());
  struct Scheme { const char* name; std::vector<int> a, b; };
  const std::vector<Scheme> schemes = {
      {"bf16 x1 (plain bf16 operands)", {0}, {0}},
      {"bf16 x3 (2-way split)", {0, 0, 1}, {0, 1, 0}},
      {"bf16 x6 (3-way split, 6 terms)", {0, 0, 1, 1, 0, 2}, {0, 1, 0, 1, 2, 0}},
      {"bf16 x9 (3-way split, 9 terms)", {0, 0, 1, 1, 0, 2, 1, 2, 2}, {0, 1, 0, 1, 2, 0, 2, 1, 2}}};
  for (const auto& s : schemes) {
    hipMemcpy(dta, s.a.data(), s.a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dtb, s.b.data(), s.b.size() * 4, hipMemcpyHostToDevice);
    tile_bf16<<<1, 64>>>(dAp, dBp, K, static_cast<int>(s.a.size()), dta, dtb, dC);
    hipMemcpy(c.data(), dC, 4096, hipMemcpyDeviceToHost);
    report(s.name, c.data());
  }
  return 0;
}
