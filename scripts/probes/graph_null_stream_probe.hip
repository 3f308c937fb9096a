// Probe for DESIGN.md 8 "hipGraph replays and the default stream": is a hipGraphLaunch on the legacy NULL stream ordered
// against the eager launches enqueued on the NULL stream right before / after it?
//
//   iteration t:  eager  A: x[i] = t                 (long: many passes over a large buffer)
//                 graph  { B: y[i] = x[i] + 1 ; B2: w[i] = y[i] * 2 }     (captured once, replayed every iteration)
//                 eager  C: z[i] = w[i]              -> expect z == 2 (t + 1) everywhere
//
// run once with every launch on the NULL stream, once on a created (non-blocking) stream, once with the graph on its
// own stream ordered by events against the NULL stream (what nets/din_fused.py:GraphRunner does).  Prints the number of
// wrong words per mode.  Build: hipcc --offload-arch=gfx950 -O2 graph_null_stream_probe.hip -o graph_null_stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);           \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

__global__ void kA(int* x, size_t n, int t, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      x[i] = (p == passes - 1) ? t : -1;
}
__global__ void kB(const int* x, int* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] + 1;
}
__global__ void kB2(const int* y, int* w, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = y[i] * 2;
}
__global__ void kC(const int* w, int* z, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) z[i] = w[i];
}
__global__ void kCheck(const int* z, size_t n, int expect, unsigned long long* bad) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (z[i] != expect) ++c;
  if (c) atomicAdd(bad, c);
}

int main() {
  const size_t n = size_t(1) << 24;      // 64 MB per buffer
  int *x, *y, *w, *z;
  unsigned long long* bad;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&z, n * 4));
  CK(hipMalloc(&bad, 8));
  hipStream_t cap, side, gs;
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(kB, dim3(2048), dim3(256), 0, cap, x, y, n);
  hipLaunchKernelGGL(kB2, dim3(2048), dim3(256), 0, cap, y, w, n);
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const char* names[3] = {"everything on the NULL stream", "everything on a created stream",
                          "graph on its own stream, event-ordered against the NULL stream"};
  for (int mode = 0; mode < 3; ++mode) {
    hipStream_t s = mode == 1 ? side : nullptr;
    CK(hipMemsetAsync(bad, 0, 8, s));
    CK(hipStreamSynchronize(s));
    const int iters = 200;
    for (int t = 1; t <= iters; ++t) {
      hipLaunchKernelGGL(kA, dim3(512), dim3(256), 0, s, x, n, t, 3);
      if (mode == 2) {
        CK(hipEventRecord(e1, s));
        CK(hipStreamWaitEvent(gs, e1, 0));
        CK(hipGraphLaunch(ge, gs));
        CK(hipEventRecord(e2, gs));
        CK(hipStreamWaitEvent(s, e2, 0));
      } else {
        CK(hipGraphLaunch(ge, s));
      }
      hipLaunchKernelGGL(kC, dim3(2048), dim3(256), 0, s, w, z, n);
      hipLaunchKernelGGL(kCheck, dim3(1024), dim3(256), 0, s, z, n, 2 * (t + 1), bad);
    }
    CK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("mode %d (%s): %llu wrong words over %d iterations x %zu words\n", mode, names[mode], h, iters, n);
  }
  return 0;
}
