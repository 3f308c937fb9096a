// Layout probe for the row-wise Adam update: random 256-byte read-modify-writes of w, m, v held in three
// arrays (SoA, the library's layout) against one interleaved [V][3][K] array (AoS), same sorted row set, same
// arithmetic, 16 lanes x float4 per row.  Build: hipcc --offload-arch=gfx950 -O3 aos_probe.hip -o aos_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

constexpr int K = 64;

__device__ __forceinline__ void adam4(float4& w, float4& m, float4& v, const float4 g) {
  const float b1 = 0.9f, b2 = 0.999f, lr = 1e-3f, eps = 1e-5f;
  float* pw = &w.x; float* pm = &m.x; float* pv = &v.x; const float* pg = &g.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pm[i] = b1 * pm[i] + (1 - b1) * pg[i];
    pv[i] = b2 * pv[i] + (1 - b2) * pg[i] * pg[i];
    pw[i] -= lr * pm[i] / (sqrtf(pv[i]) + eps);
  }
}

__global__ void soa_kernel(float* w, float* m, float* v, const float* g, const int* rows, int n) {
  const int lane = threadIdx.x & 15;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < n; s += (gridDim.x * blockDim.x) >> 4) {
    const size_t o = static_cast<size_t>(rows[s]) * K + lane * 4;
    float4 a = *reinterpret_cast<float4*>(w + o), b = *reinterpret_cast<float4*>(m + o), c = *reinterpret_cast<float4*>(v + o);
    const float4 gg = *reinterpret_cast<const float4*>(g + static_cast<size_t>(s) * K + lane * 4);
    adam4(a, b, c, gg);
    *reinterpret_cast<float4*>(w + o) = a; *reinterpret_cast<float4*>(m + o) = b; *reinterpret_cast<float4*>(v + o) = c;
  }
}

// SoA rows + the three 4-byte linear-weight arrays (lane 0 of the row group), as the library does
__global__ void soa_lin3_kernel(float* w, float* m, float* v, float* l0, float* l1, float* l2, const float* g,
                                const int* rows, int n) {
  const int lane = threadIdx.x & 15;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < n; s += (gridDim.x * blockDim.x) >> 4) {
    const int row = rows[s];
    const size_t o = static_cast<size_t>(row) * K + lane * 4;
    float4 a = *reinterpret_cast<float4*>(w + o), b = *reinterpret_cast<float4*>(m + o), c = *reinterpret_cast<float4*>(v + o);
    float x = 0.f, y = 0.f, z = 0.f;
    if (lane == 0) { x = l0[row]; y = l1[row]; z = l2[row]; }
    const float4 gg = *reinterpret_cast<const float4*>(g + static_cast<size_t>(s) * K + lane * 4);
    adam4(a, b, c, gg);
    *reinterpret_cast<float4*>(w + o) = a; *reinterpret_cast<float4*>(m + o) = b; *reinterpret_cast<float4*>(v + o) = c;
    if (lane == 0) { y = 0.9f * y + 0.1f * gg.x; z = 0.999f * z + 0.001f * gg.x * gg.x; x -= 1e-3f * y / (sqrtf(z) + 1e-5f);
                     l0[row] = x; l1[row] = y; l2[row] = z; }
  }
}

// SoA rows + ONE [V][4] array holding (lin, lin_m, lin_v, pad): one 16-byte access each way
__global__ void soa_lin4_kernel(float* w, float* m, float* v, float4* l, const float* g, const int* rows, int n) {
  const int lane = threadIdx.x & 15;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < n; s += (gridDim.x * blockDim.x) >> 4) {
    const int row = rows[s];
    const size_t o = static_cast<size_t>(row) * K + lane * 4;
    float4 a = *reinterpret_cast<float4*>(w + o), b = *reinterpret_cast<float4*>(m + o), c = *reinterpret_cast<float4*>(v + o);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane == 0) t = l[row];
    const float4 gg = *reinterpret_cast<const float4*>(g + static_cast<size_t>(s) * K + lane * 4);
    adam4(a, b, c, gg);
    *reinterpret_cast<float4*>(w + o) = a; *reinterpret_cast<float4*>(m + o) = b; *reinterpret_cast<float4*>(v + o) = c;
    if (lane == 0) { t.y = 0.9f * t.y + 0.1f * gg.x; t.z = 0.999f * t.z + 0.001f * gg.x * gg.x; t.x -= 1e-3f * t.y / (sqrtf(t.z) + 1e-5f);
                     l[row] = t; }
  }
}

// split form: the row kernel leaves one linear-weight gradient per run, a lane-per-run kernel applies it
__global__ void soa_gout_kernel(float* w, float* m, float* v, float* gout, const float* g, const int* rows, int n) {
  const int lane = threadIdx.x & 15;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < n; s += (gridDim.x * blockDim.x) >> 4) {
    const size_t o = static_cast<size_t>(rows[s]) * K + lane * 4;
    float4 a = *reinterpret_cast<float4*>(w + o), b = *reinterpret_cast<float4*>(m + o), c = *reinterpret_cast<float4*>(v + o);
    const float4 gg = *reinterpret_cast<const float4*>(g + static_cast<size_t>(s) * K + lane * 4);
    adam4(a, b, c, gg);
    *reinterpret_cast<float4*>(w + o) = a; *reinterpret_cast<float4*>(m + o) = b; *reinterpret_cast<float4*>(v + o) = c;
    if (lane == 0) gout[s] = gg.x;
  }
}
__global__ void lin_runs_kernel(float* l0, float* l1, float* l2, const float* gout, const int* rows, int n) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int row = rows[s];
    const float gx = gout[s];
    float x = l0[row], y = l1[row], z = l2[row];
    y = 0.9f * y + 0.1f * gx; z = 0.999f * z + 0.001f * gx * gx; x -= 1e-3f * y / (sqrtf(z) + 1e-5f);
    l0[row] = x; l1[row] = y; l2[row] = z;
  }
}

__global__ void aos_kernel(float* t, const float* g, const int* rows, int n) {
  const int lane = threadIdx.x & 15;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < n; s += (gridDim.x * blockDim.x) >> 4) {
    float* p = t + static_cast<size_t>(rows[s]) * 3 * K + lane * 4;
    float4 a = *reinterpret_cast<float4*>(p), b = *reinterpret_cast<float4*>(p + K), c = *reinterpret_cast<float4*>(p + 2 * K);
    const float4 gg = *reinterpret_cast<const float4*>(g + static_cast<size_t>(s) * K + lane * 4);
    adam4(a, b, c, gg);
    *reinterpret_cast<float4*>(p) = a; *reinterpret_cast<float4*>(p + K) = b; *reinterpret_cast<float4*>(p + 2 * K) = c;
  }
}

int main() {
  const size_t V = 12000202; const int n = 2060000;
  std::mt19937_64 rng(1);
  std::vector<int> rows(V);
  for (size_t i = 0; i < V; ++i) rows[i] = static_cast<int>(i);
  for (int i = 0; i < n; ++i) std::swap(rows[i], rows[i + rng() % (V - i)]);
  rows.resize(n);
  std::sort(rows.begin(), rows.end());
  float *w, *m, *v, *t, *g; int* r;
  hipMalloc(&w, V * K * 4); hipMalloc(&m, V * K * 4); hipMalloc(&v, V * K * 4); hipMalloc(&t, V * 3 * K * 4);
  hipMalloc(&g, static_cast<size_t>(n) * K * 4); hipMalloc(&r, n * 4);
  hipMemset(w, 0, V * K * 4); hipMemset(m, 0, V * K * 4); hipMemset(v, 0, V * K * 4); hipMemset(t, 0, V * 3 * K * 4);
  hipMemset(g, 0, static_cast<size_t>(n) * K * 4);
  hipMemcpy(r, rows.data(), n * 4, hipMemcpyHostToDevice);
  float *l0, *l1, *l2; float4* l4;
  hipMalloc(&l0, V * 4); hipMalloc(&l1, V * 4); hipMalloc(&l2, V * 4); hipMalloc(&l4, V * 16);
  hipMemset(l0, 0, V * 4); hipMemset(l1, 0, V * 4); hipMemset(l2, 0, V * 4); hipMemset(l4, 0, V * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = static_cast<double>(n) * K * 4 * 7 + n * 4.0;
  for (int grid : {2048, 4096, 8192}) {
    for (int which = 0; which < 5; ++which) {
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(soa_kernel, dim3(grid), dim3(256), 0, 0, w, m, v, g, r, n);
        else if (which == 1) hipLaunchKernelGGL(aos_kernel, dim3(grid), dim3(256), 0, 0, t, g, r, n);
        else if (which == 2) hipLaunchKernelGGL(soa_lin3_kernel, dim3(grid), dim3(256), 0, 0, w, m, v, l0, l1, l2, g, r, n);
        else if (which == 3) hipLaunchKernelGGL(soa_lin4_kernel, dim3(grid), dim3(256), 0, 0, w, m, v, l4, g, r, n);
        else {
          hipLaunchKernelGGL(soa_gout_kernel, dim3(grid), dim3(256), 0, 0, w, m, v, reinterpret_cast<float*>(l4), g, r, n);
          hipLaunchKernelGGL(lin_runs_kernel, dim3(2048), dim3(256), 0, 0, l0, l1, l2, reinterpret_cast<const float*>(l4), r, n);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      const char* names[5] = {"SoA w|m|v            ", "AoS [V][3][K]        ", "SoA + lin x3 (4 B)   ", "SoA + lin [V][4] 16 B", "SoA, lin in 2nd kernel"};
      printf("%s grid %5d: %.4f ms  %.0f GB/s (row bytes only)\n", names[which], grid, best, bytes / best / 1e6);
    }
  }
  return 0;
}
