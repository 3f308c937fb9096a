// Probe for DESIGN 8: would a per-RUN backward of the first layer pay?  Instead of lr_deepfm_l1_dgrad_f32 (one
// 128 x 64 product per POSITION, 3.31 M positions, MFMA-bound, writes 847 MB of per-position rows that the row update
// re-reads) one could sum gz[b(p)] (and gl[b] * fsum[b]) over the positions of each RUN first (2.06 M runs) and apply
// one product per run.  The MFMA work shrinks; what it costs is this kernel: random 512 + 256-byte gathers from the
// [B, 128] / [B, 64] activations (12 MB hot set) for every position.  Synthetic cfg-2 shape: B = 16,384, F = 202,
// Zipf(1.05) ids over 50,001 values per field.
// Build: hipcc --offload-arch=gfx950 -O3 rungrad_probe.hip -o rungrad_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

constexpr int H1 = 128, K = 64;

// one 32-lane group per run: lanes hold a float4 of the gz sum, lanes 0..15 also a float4 of the fsum sum
__global__ void runsum_kernel(const float* __restrict__ gz, const float* __restrict__ fsum, const float* __restrict__ gl,
                              const int* __restrict__ seg_pos, const int* __restrict__ seg_start, int n_seg, int F,
                              float* __restrict__ out /* [n_seg][K] */, float* __restrict__ sgl) {
  const int lane = threadIdx.x & 31;
  const int groups = (gridDim.x * blockDim.x) >> 5;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; s < n_seg; s += groups) {
    const int p0 = seg_start[s], p1 = seg_start[s + 1];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    float t = 0.f;
    int p = p0;
    for (; p + 1 < p1; p += 2) {                                  // two positions in flight
      const int b0 = seg_pos[p] / F, b1 = seg_pos[p + 1] / F;
      const float4 x0 = *reinterpret_cast<const float4*>(gz + static_cast<size_t>(b0) * H1 + lane * 4);
      const float4 x1 = *reinterpret_cast<const float4*>(gz + static_cast<size_t>(b1) * H1 + lane * 4);
      const float g0 = gl[b0], g1 = gl[b1];
      if (lane < 16) {
        const float4 y0 = *reinterpret_cast<const float4*>(fsum + static_cast<size_t>(b0) * K + lane * 4);
        const float4 y1 = *reinterpret_cast<const float4*>(fsum + static_cast<size_t>(b1) * K + lane * 4);
        c.x += g0 * y0.x + g1 * y1.x; c.y += g0 * y0.y + g1 * y1.y; c.z += g0 * y0.z + g1 * y1.z; c.w += g0 * y0.w + g1 * y1.w;
      }
      a.x += x0.x + x1.x; a.y += x0.y + x1.y; a.z += x0.z + x1.z; a.w += x0.w + x1.w;
      t += g0 + g1;
    }
    if (p < p1) {
      const int b0 = seg_pos[p] / F;
      const float4 x0 = *reinterpret_cast<const float4*>(gz + static_cast<size_t>(b0) * H1 + lane * 4);
      const float g0 = gl[b0];
      if (lane < 16) {
        const float4 y0 = *reinterpret_cast<const float4*>(fsum + static_cast<size_t>(b0) * K + lane * 4);
        c.x += g0 * y0.x; c.y += g0 * y0.y; c.z += g0 * y0.z; c.w += g0 * y0.w;
      }
      a.x += x0.x; a.y += x0.y; a.z += x0.z; a.w += x0.w;
      t += g0;
    }
    // stand-in for the 128 -> 64 product: fold the two halves of the gz sum (keeps the output volume of the real thing)
    const float4 o = make_float4(a.x + __shfl_xor(a.x, 16), a.y + __shfl_xor(a.y, 16), a.z + __shfl_xor(a.z, 16), a.w + __shfl_xor(a.w, 16));
    if (lane < 16) {
      *reinterpret_cast<float4*>(out + static_cast<size_t>(s) * K + lane * 4) = make_float4(o.x + c.x, o.y + c.y, o.z + c.z, o.w + c.w);
      if (lane == 0) sgl[s] = t;
    }
  }
}

int main() {
  const int B = 16384, F = 202, V = 50001;
  std::mt19937_64 rng(7);
  std::vector<double> cdf(V);
  double acc = 0;
  for (int v = 0; v < V; ++v) { acc += 1.0 / std::pow(v + 1.0, 1.05); cdf[v] = acc; }
  std::vector<int> seg_pos, seg_start{0};
  seg_pos.reserve(static_cast<size_t>(B) * F);
  std::vector<std::pair<int, int>> kv(B);
  std::uniform_real_distribution<double> U(0.0, acc);
  for (int f = 0; f < F; ++f) {
    for (int b = 0; b < B; ++b) kv[b] = {static_cast<int>(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()), b};
    std::sort(kv.begin(), kv.end());
    for (int b = 0; b < B; ++b) {
      if (b > 0 && kv[b].first != kv[b - 1].first) seg_start.push_back(static_cast<int>(seg_pos.size()));
      seg_pos.push_back(kv[b].second * F + f);
    }
    seg_start.push_back(static_cast<int>(seg_pos.size()));
  }
  const int n_seg = static_cast<int>(seg_start.size()) - 1, n_pos = static_cast<int>(seg_pos.size());
  printf("positions %d, runs %d (%.2f positions per run)\n", n_pos, n_seg, static_cast<double>(n_pos) / n_seg);
  float *gz, *fsum, *gl, *out, *sgl; int *dpos, *dstart;
  hipMalloc(&gz, static_cast<size_t>(B) * H1 * 4); hipMalloc(&fsum, static_cast<size_t>(B) * K * 4); hipMalloc(&gl, B * 4);
  hipMalloc(&out, static_cast<size_t>(n_seg) * K * 4); hipMalloc(&sgl, static_cast<size_t>(n_seg) * 4);
  hipMalloc(&dpos, static_cast<size_t>(n_pos) * 4); hipMalloc(&dstart, static_cast<size_t>(n_seg + 1) * 4);
  hipMemset(gz, 0, static_cast<size_t>(B) * H1 * 4); hipMemset(fsum, 0, static_cast<size_t>(B) * K * 4); hipMemset(gl, 0, B * 4);
  hipMemcpy(dpos, seg_pos.data(), static_cast<size_t>(n_pos) * 4, hipMemcpyHostToDevice);
  hipMemcpy(dstart, seg_start.data(), static_cast<size_t>(n_seg + 1) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gathered = static_cast<double>(n_pos) * (H1 + K + 1) * 4, written = static_cast<double>(n_seg) * (K + 1) * 4;
  for (int grid : {2048, 4096, 8192, 16384}) {
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(runsum_kernel, dim3(grid), dim3(256), 0, 0, gz, fsum, gl, dpos, dstart, n_seg, F, out, sgl);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it > 0 && ms < best) best = ms;
    }
    printf("run sums, grid %5d: %.4f ms  (%.2f GB gathered from the 12 MB activations = %.0f GB/s, %.2f GB written)\n", grid, best,
           gathered / 1e9, gathered / best / 1e6, written / 1e9);
  }
  return 0;
}
