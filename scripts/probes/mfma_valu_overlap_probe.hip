// Do the bf16 MFMAs of one wave overlap with the VALU work of the OTHER wave on its SIMD?  (round 5)
// One workgroup of 512 threads per CU = two waves per SIMD (waves w and w + 4 share one).  Every wave runs `iters` stages;
// a stage = NM v_mfma_f32_32x32x16_bf16 (on NACC accumulators, dealt round-robin: NACC = 1 is one dependent chain) followed by
// NV dependent-free v_fma_f32.  Modes:
//   0  both waves of a SIMD run stages in phase                     (what a kernel whose waves are released together does)
//   1  the second wave starts half a stage late (s_sleep)            (does a phase offset survive, does it help?)
//   2  the first wave runs only the MFMA part, the second only the VALU part  (pure cross-wave overlap)
//   3  one wave per SIMD busy (the other exits at once)              (the single-wave time of a stage)
//   4  in-wave interleave: NV / NM v_fma_f32 dealt behind every MFMA (program order), both waves in phase
// Prints shader cycles per stage and wave (s_memtime of the longest wave of a block, averaged over the blocks).
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NM, int NV, int NACC, int MODE>
__global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, int iters) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 va, vb;
  for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(1.0f + threadIdx.x * 1e-3f); vb[e] = (__bf16)0.5f; }
  float x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float b = 0.5f;
  const bool do_m = !(MODE == 2 && wid >= 4), do_v = !(MODE == 2 && wid < 4);
  if (MODE == 3 && wid >= 4) return;
  __syncthreads();
  if (MODE == 1 && wid >= 4) __builtin_amdgcn_s_sleep((NM * 32 + NV * 4) / 2 / 64);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(va), "v"(vb));
#pragma unroll
        for (int v = 0; v < NV / NM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(b));
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(va), "v"(vb));
      }
      if (do_v) {
#pragma unroll 16
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(b));
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[1] = 1;
  if ((threadIdx.x & 63) == 0) atomicMax(&out[2 + blockIdx.x], t1 - t0);
}

template <int NM, int NV, int NACC, int MODE>
void run(unsigned long long* out) {
  const int iters = 400, blocks = 256;
  hipMemset(out, 0, (2 + blocks) * 8);
  probe<NM, NV, NACC, MODE><<<blocks, 512>>>(out, 10);
  hipDeviceSynchronize();
  hipMemset(out, 0, (2 + blocks) * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<NM, NV, NACC, MODE><<<blocks, 512>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2 + 256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int b = 0; b < blocks; ++b) sum += (double)h[2 + b];
  // s_memtime ticks at 100 MHz on this part: convert through the event time instead
  printf("NM=%3d NV=%4d NACC=%d mode=%d : %8.3f us per stage (event time / iters), memtime ticks/stage %.1f ; MFMA pipe floor per SIMD "
         "(both waves) at 2.4 GHz %.3f us\n",
         NM, NV, NACC, MODE, ms * 1e3 / iters, sum / blocks / iters, (MODE == 2 || MODE == 3 ? 1 : 2) * NM * 32 / 2400.0);
}

int main() {
  unsigned long long* out;
  hipMalloc(&out, (2 + 256) * 8);
  printf("-- stage = 96 MFMA + 480 VALU (the split-bf16 softmax-CE stage)\n");
  run<96, 480, 1, 3>(out);
  run<96, 480, 1, 0>(out);
  run<96, 480, 1, 1>(out);
  run<96, 480, 1, 2>(out);
  run<96, 480, 1, 4>(out);
  run<96, 480, 2, 0>(out);
  run<96, 480, 2, 4>(out);
  run<96, 480, 4, 0>(out);
  run<96, 480, 4, 4>(out);
  printf("-- MFMA only / VALU only\n");
  run<96, 0, 1, 0>(out);
  run<96, 0, 1, 3>(out);
  run<96, 0, 2, 3>(out);
  run<96, 0, 4, 3>(out);
  run<1, 480, 1, 3>(out);
  run<1, 480, 1, 0>(out);
  return 0;
}
