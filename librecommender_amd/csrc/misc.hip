// Error strings / ABI version.
#include <stdlib.h>

#include "common.hpp"

extern "C" const char* lr_strerror(int code) {
  switch (code) {
    case LR_OK: return "ok";
    case LR_EINVAL: return "invalid argument";
    case LR_ESHAPE: return "shape not supported by the compiled gfx950 kernels";
    case LR_EWORKSPACE: return "workspace too small";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown error";
}

extern "C" int lr_abi_version(void) { return 25; }

// A captured training step must hold KERNEL nodes only: memset / memcpy nodes replayed next to eager work on another stream
// gave memory faults at varying addresses on this stack (round 3, nets/din_fused.py:GraphRunner).  Counts the nodes of a
// captured graph that are neither kernel nor empty (fork / join) nodes; < 0: the negated HIP error.
extern "C" int lr_graph_foreign_nodes(void* graph, int* n_nodes_out) {
  if (graph == nullptr) return -LR_EINVAL;
  hipGraph_t g = static_cast<hipGraph_t>(graph);
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(g, nullptr, &n);
  if (e != hipSuccess) return -static_cast<int>(e);
  if (n_nodes_out != nullptr) *n_nodes_out = static_cast<int>(n);
  if (n == 0) return 0;
  hipGraphNode_t* nodes = static_cast<hipGraphNode_t*>(malloc(n * sizeof(hipGraphNode_t)));
  if (nodes == nullptr) return -LR_EINVAL;
  e = hipGraphGetNodes(g, nodes, &n);
  int foreign = 0;
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e == hipSuccess && t != hipGraphNodeTypeKernel && t != hipGraphNodeTypeEmpty) ++foreign;
  }
  free(nodes);
  return e == hipSuccess ? foreign : -static_cast<int>(e);
}

// ---- measurement probe: sustained f32 MFMA issue rate -----------------------------------------------
// `iters` x 8 back-to-back v_mfma_f32_32x32x2_f32 on four independent accumulators per wave, `waves_per_simd`
// waves on every SIMD of every CU: what the matrix pipe sustains at the clock the chip actually holds under that
// load (the 157.3 TFLOP/s figure assumes the 2.4 GHz peak clock).  scripts/mfma_peak.py times it.
namespace lr {
using f32x16p = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(kBlock) void mfma_probe_kernel(int iters, float seed, float* __restrict__ out) {
  f32x16p a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = seed + static_cast<float>(threadIdx.x) * 1e-6f, y = 1.0f - x;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 123.456f) out[0] = s;     // keeps the chain alive; never true for the seeds used
}
}  // namespace lr

extern "C" int lr_mfma_f32_probe(int iters, int waves_per_simd, float* out, lr_stream_t stream) {
  LR_CHECK_ARG(iters >= 1 && waves_per_simd >= 1 && waves_per_simd <= 8 && out != nullptr);
  hipLaunchKernelGGL(lr::mfma_probe_kernel, dim3(lr::kNumCU * waves_per_simd), dim3(lr::kBlock), 0,
                     lr::as_stream(stream), iters, 0.25f, out);
  return lr::launch_status();
}

// ---- test aid: hold compute units busy -------------------------------------------------------------------
// `grid` workgroups, each claiming `lds_bytes` of LDS (so that at 160 KB one workgroup owns a CU) and spinning for
// `usec` microseconds of the constant 100 MHz counter.  tests/test_tail_fused_gpu.py runs it on a second stream
// beside the one-launch tail to take residency away from the tail's workgroups.
namespace lr {
__global__ __launch_bounds__(kBlock) void occupy_kernel(unsigned long long ticks, unsigned* out) {
  extern __shared__ char occ_smem[];
  if (threadIdx.x == 0) {
    occ_smem[0] = 1;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (out != nullptr && occ_smem[0] == 77) out[0] = 1u;
  }
  __syncthreads();
}
}  // namespace lr

extern "C" int lr_probe_occupy(int grid, size_t lds_bytes, int64_t usec, lr_stream_t stream) {
  LR_CHECK_ARG(grid >= 1 && usec >= 0 && usec <= 20 * 1000 * 1000 && lds_bytes <= 160 * 1024);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lr::occupy_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(lr::occupy_kernel, dim3(grid), dim3(lr::kBlock), lds_bytes, lr::as_stream(stream),
                     static_cast<unsigned long long>(usec) * 100ull, static_cast<unsigned*>(nullptr));
  return lr::launch_status();
}

// ---- measurement aid: the shader clock a timed region actually ran at ----------------------------------------
// One lane per workgroup stores {s_memtime (ticks of the shader clock), s_memrealtime (constant rate)} into the slot of the
// COMPUTE UNIT it runs on (XCC id, shader engine, shader array, CU: s_getreg HW_REG_XCC_ID / HW_REG_HW_ID): 4,096 slots of two
// words, 2,048 workgroups so that nearly every CU is hit.  s_memtime counters of different CUs / engines carry different
// offsets and WHERE a workgroup lands is not repeatable, so only differences of the SAME slot between two calls mean anything
// (a first version compared one workgroup's readings of two launches and printed 4.5 GHz; a per-XCD version still scattered).
// Two calls around a timed region give its mean shader clock: median over the CUs of d(memtime) / d(memrealtime) x the
// real-time rate (bench.py: `config.shader_clock_mhz`) — boxes of this pool differ by ~6 % in step time, this number says how
// much of that is the clock the part sustained.
namespace lr {
constexpr int kClockSlots = 4096;
__global__ void clock_probe_kernel(unsigned long long* out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;          // hwreg(HW_REG_XCC_ID, 0, 4)
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));                 // hwreg(HW_REG_HW_ID, 0, 32)
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    const unsigned slot = (xcc << 9) | (se << 5) | (sh << 4) | cu;                 // < 4096
    out[2 * slot] = __builtin_amdgcn_s_memtime();
    out[2 * slot + 1] = wall_clock64();
  }
}
}  // namespace lr

extern "C" int lr_clock_probe_slots(void) { return lr::kClockSlots; }

extern "C" int lr_clock_probe(uint64_t* out, lr_stream_t stream) {
  LR_CHECK_ARG(out != nullptr);
  hipLaunchKernelGGL(lr::clock_probe_kernel, dim3(2048), dim3(64), 0, lr::as_stream(stream),
                     reinterpret_cast<unsigned long long*>(out));
  return lr::launch_status();
}
