// Streaming in-batch softmax cross-entropy on the f32 MFMA pipe: the B x N logit matrix of the two-tower
// retrieval loss (tfops/loss.py:71-75 `softmax_cross_entropy` over `adjust_logits`, algorithms/two_tower.py:458-479)
// never leaves registers — 17 GB at B = N = 65,536 if materialised.
//
//   logits[i][j] = X_i . Y_j + bias[j]            X = user-tower outputs / temperature, Y = item-tower outputs,
//                                                 bias = -log Q(j) (sampling-bias correction), all optional
//   masked (accidental hits): id_row[i] == id_col[j] and j != pos0 + i   ->  float32.min in the reference,
//                                                 i.e. probability exactly 0 (the positive itself is never masked)
//   loss_i = logsumexp_j logits[i][:] - logits[i][pos0 + i]
//
// Two sweeps, four GEMM-sized contractions in total (the minimum without writing B x N or atomics):
//   MODE 0 (rows stationary)  : S = Y_tile X^T, online softmax (running max / sum per row, flash-style rescale),
//                               W_i = sum_j P_ij Y_j accumulated in the same pass  ->  lse, positive logit, W
//                               d loss / d X_i = g_i (W_i - Y_pos(i))  (host-side elementwise)
//   MODE 1 (columns stationary): S recomputed, P = exp(S - lse_i), V_j = sum_i g_i P_ij X_i
//                               d loss / d Y_j = V_j - [j = pos(i)] g_i X_i
// Both are ONE kernel template: 4 waves per workgroup, each holding 32 stationary vectors in VGPRs as the
// B operand of v_mfma_f32_32x32x2_f32 (one stationary vector per lane column, so its scalars — running max,
// sum, bias, id — are one register per lane); the streamed side goes through LDS in 32-row stages held in a
// 3-deep ring with wave-level full/done counters and no workgroup barrier in the loop (the pipeline of
// score_topk.hip).  The second contraction takes P straight from the accumulator registers: in the 32x32
// accumulator layout lane half h of register r holds streamed row (r&3) + 8 (r>>2) + 4 h, so MFMA step r
// uses exactly that row pair as its reduction pair and no cross-lane movement is needed.  Reductions run in a
// fixed order: results are run-to-run identical.
#include "common.hpp"

namespace lr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kSceNeg = -3.0e38f;               // "float32.min": finite, so no inf - inf
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kScePD = 2;                         // stages of prefetch in flight
constexpr int kSceNB = 3;                         // LDS ring depth

struct SceArgs {
  const float* X; int64_t nX;       // stationary vectors [nX, D]
  const float* Y; int64_t nY;       // streamed vectors  [nY, D]
  int D;
  const float* bias;                // per COLUMN, nullable
  const int32_t* idr; const int32_t* idc;   // accidental-hit ids of rows / columns (both or none)
  int64_t pos0;                     // positive column of row i = pos0 + i
  float* lse; float* pos_logit; float* W;   // MODE 0 outputs (W nullable)
  const float* lse_in; const float* g;      // MODE 1 inputs (per ROW)
  float* V;                                 // MODE 1 output
  // MODE 0 with the streamed range cut into G column ranges (few stationary rows: a rank's share of a global
  // batch): range g writes its running state instead of final values — merged by sce_merge_kernel
  int G;
  float* part_m; float* part_s; float* part_pos; float* part_W;   // [G][nX] (x3), [G][nX][D]
};

template <int DT, int MODE, bool GEMM2>
__global__ __launch_bounds__(kBlock, 2) void softmax_ce_kernel(SceArgs a) {
  constexpr int DH = DT / 2;
  constexpr int LDW = DT + 4;
  constexpr int kTI = 32;
  constexpr int NB = kSceNB;
  constexpr int NQ = kTI * DT / 4;
  constexpr int NLD = NQ / kBlock;
  constexpr int NP = DT / 64;               // 64-wide output blocks of the second contraction
  static_assert(DT == 64 || DT == 128, "compiled widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                     // [NB][32][LDW]
  float* scf = tile + NB * kTI * LDW;                               // [NB][2][32]
  int* sci = reinterpret_cast<int*>(scf + NB * 2 * 32);             // [NB][32]
  int* full_cnt = sci + NB * 32;                                    // [NB]
  int* done_cnt = full_cnt + NB;                                    // [NB]
  if (threadIdx.x < 2 * NB) full_cnt[threadIdx.x] = 0;

  const int tid = threadIdx.x, wid = tid / kWave, lane = tid & (kWave - 1);
  const int j = lane & 31, h = lane >> 5;
  const int D = a.D;
  const bool has_bias = a.bias != nullptr, has_mask = a.idr != nullptr;
  // valid addresses for the disabled operands (values are replaced by selects below)
  const float* bias_p = has_bias ? a.bias : (MODE == 0 ? a.Y : a.X);
  const int32_t* idr_p = has_mask ? a.idr : reinterpret_cast<const int32_t*>(MODE == 0 ? a.X : a.Y);
  const int32_t* idc_p = has_mask ? a.idc : reinterpret_cast<const int32_t*>(MODE == 0 ? a.Y : a.X);

  // ---- stationary vectors: B operand, resident in registers ------------------------------
  const int64_t xi = (static_cast<int64_t>(blockIdx.x) * 4 + wid) * 32 + j;
  const bool x_ok = xi < a.nX;
  const int64_t xc = x_ok ? xi : a.nX - 1;
  float bfrag[DH];
#pragma unroll
  for (int s = 0; s < DH; s += 4) {
    const int d = h * DH + s;
    float4 x = f4_zero();
    if (x_ok && d < D) x = ld4(a.X + xi * D + d);
    bfrag[s] = x.x; bfrag[s + 1] = x.y; bfrag[s + 2] = x.z; bfrag[s + 3] = x.w;
  }
  // per-lane scalars of the stationary vector
  int my_id;
  float my_bias2 = 0.f;              // MODE 1: bias of my column, base-2 scaled
  if (MODE == 0) {
    my_id = has_mask ? idr_p[xc] : -1;
  } else {
    my_id = has_mask ? idc_p[xc] : -1;
    my_bias2 = has_bias ? bias_p[xc] * kLog2e : 0.f;
  }
  // MODE 0: my row's positive column; MODE 1: my column index as seen from row i: i == xi - pos0
  const int64_t my_pos = MODE == 0 ? a.pos0 + xi : xi - a.pos0;
  float run_m = kSceNeg, run_s = 0.f, pos_l2 = 0.f;

  f32x16 dacc[GEMM2 ? NP * 2 : 1];
#pragma unroll
  for (int t = 0; t < (GEMM2 ? NP * 2 : 1); ++t)
    dacc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // ---- staging -----------------------------------------------------------------------------
  float4 pre[NLD];
  float ps0 = 0.f, ps1 = 0.f;
  int pi0 = 0;
  uint32_t pre_ok = 0;
  const uint32_t Nu = static_cast<uint32_t>(a.nY), Du = static_cast<uint32_t>(D);
  auto stage_load = [&](int64_t st) {     // st may lie past the end: addresses are clamped
    pre_ok = 0;
    const uint32_t r0 = static_cast<uint32_t>(st * kTI < a.nY ? st * kTI : a.nY);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = tid + u * kBlock;
      const uint32_t row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
      const uint32_t r = r0 + row;
      if ((r < Nu) && (c4 < Du)) pre_ok |= 1u << u;
      const uint32_t rc = r < Nu ? r : Nu - 1;
      const uint32_t cc = c4 < Du ? c4 : Du - 4;
      pre[u] = ld4(a.Y + (static_cast<uint64_t>(rc) * Du + cc));
    }
    // per-row scalars of the stage (every thread loads; lanes 0..31 of wave 0 publish them)
    const uint32_t rs = r0 + (tid & 31);
    const uint32_t rsc = rs < Nu ? rs : Nu - 1;
    if (rs < Nu) pre_ok |= 1u << 31;
    if (MODE == 0) {
      ps0 = bias_p[rsc];
      pi0 = idc_p[rsc];
    } else {
      ps0 = a.lse_in[rsc];
      ps1 = a.g[rsc];
      pi0 = idr_p[rsc];
    }
  };
  auto stage_write = [&](int buf) {
    float* dst = tile + buf * kTI * LDW;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = tid + u * kBlock;
      const int row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
      st4(dst + row * LDW + c4, ((pre_ok >> u) & 1u) ? pre[u] : f4_zero());
    }
    if (tid < 32) {
      const bool ok = (pre_ok >> 31) & 1u;
      if (MODE == 0) {
        // a column past the end gets bias "float32.min": probability 0, no separate bounds test
        scf[(buf * 2 + 0) * 32 + tid] = ok ? (has_bias ? ps0 * kLog2e : 0.f) : kSceNeg;
        sci[buf * 32 + tid] = (ok && has_mask) ? pi0 : -2;
      } else {
        scf[(buf * 2 + 0) * 32 + tid] = ok ? ps0 * kLog2e : 0.f;
        scf[(buf * 2 + 1) * 32 + tid] = ok ? ps1 : 0.f;      // a row past the end carries no gradient
        sci[buf * 32 + tid] = (ok && has_mask) ? pi0 : -2;
      }
    }
  };
  {  // consume the stationary fragment once: its loads are waited for here, not in the stage loop
    float chk = 0.f;
#pragma unroll
    for (int s = 0; s < DH; ++s) chk += bfrag[s];
    if (chk == 1.2345e30f) run_s = 1.f;
  }
  auto wave_signal = [&](int* c) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto wave_wait = [&](int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
  };

  const int n_st_all = static_cast<int>(ceil_div(a.nY, (int64_t)kTI));
  const int G = (MODE == 0 && a.G > 1) ? a.G : 1, gy = (MODE == 0 && a.G > 1) ? static_cast<int>(blockIdx.y) : 0;
  const int st_lo = static_cast<int>(static_cast<int64_t>(n_st_all) * gy / G);
  const int n_st = static_cast<int>(static_cast<int64_t>(n_st_all) * (gy + 1) / G) - st_lo;
  __syncthreads();
  for (int p = 0; p < kScePD && p < n_st; ++p) {
    stage_load(st_lo + p);
    stage_write(p % NB);
    wave_signal(&full_cnt[p % NB]);
  }

  for (int i = 0; i < n_st; ++i) {
    const int buf = i % NB;
    const bool more = i + kScePD < n_st;
    stage_load(st_lo + i + kScePD);   // unconditional (clamped): in flight during the MFMAs below
    wave_wait(&full_cnt[buf], 4 * (i / NB + 1));

    const float* src = tile + buf * kTI * LDW;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* arow = src + j * LDW + h * DH;
#pragma unroll
    for (int s = 0; s < DH; s += 4) {
      const float4 av = ld4(arow + s);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bfrag[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bfrag[s + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bfrag[s + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bfrag[s + 3], acc, 0, 0, 0);
    }
    // ---- epilogue: register r of lane (j,h) = streamed row y = 8 (r>>2) + 4 h + (r&3) against my vector --
    float pv[16];
    const int y0 = (st_lo + i) * kTI + 4 * h;
    if (MODE == 0) {
      float l2[16];
      float tmax = kSceNeg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = ld4(scf + (buf * 2 + 0) * 32 + 8 * q + 4 * h);
        const int4 idv = *reinterpret_cast<const int4*>(sci + buf * 32 + 8 * q + 4 * h);
        const float bb[4] = {b.x, b.y, b.z, b.w};
        const int ii[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = q * 4 + t;
          const int64_t col = y0 + 8 * q + t;
          float v = fmaf(acc[r], kLog2e, bb[t]);
          const bool is_pos = col == my_pos;
          if (ii[t] == my_id && !is_pos) v = kSceNeg;
          v = fmaxf(v, kSceNeg);                       // bias "float32.min" + score stays finite
          if (is_pos) pos_l2 = v;
          l2[r] = v;
          tmax = fmaxf(tmax, v);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(run_m, tmax);
      const float alpha = __builtin_amdgcn_exp2f(run_m - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(l2[r] - m_new);
        psum += pv[r];
      }
      run_s = fmaf(run_s, alpha, psum);
      run_m = m_new;
      if (GEMM2) {
        if (__ballot(alpha != 1.0f) != 0ull) {        // the running maximum moved: rescale W's partial sums
#pragma unroll
          for (int t = 0; t < NP * 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[t][r] *= alpha;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 ls = ld4(scf + (buf * 2 + 0) * 32 + 8 * q + 4 * h);
        const float4 gg = ld4(scf + (buf * 2 + 1) * 32 + 8 * q + 4 * h);
        const int4 idv = *reinterpret_cast<const int4*>(sci + buf * 32 + 8 * q + 4 * h);
        const float ll[4] = {ls.x, ls.y, ls.z, ls.w};
        const float gq[4] = {gg.x, gg.y, gg.z, gg.w};
        const int ii[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = q * 4 + t;
          const int64_t row = y0 + 8 * q + t;
          const float v = fmaf(acc[r], kLog2e, my_bias2);
          float p = __builtin_amdgcn_exp2f(fminf(v - ll[t], 0.f));
          if (ii[t] == my_id && row != my_pos) p = 0.f;
          pv[r] = gq[t] * p;
        }
      }
    }
    if (GEMM2) {
      // second contraction: out^T[d][my vector] += sum_y Ystage[y][d] * pv[y]; lane m = j supplies the
      // d = 64 P + 2 m + e entries of streamed row y_h(r) — one 8-byte LDS read feeds two MFMAs
      const float* ycol = src + (4 * h) * LDW + 2 * j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int yrow = 8 * (r >> 2) + (r & 3);
#pragma unroll
        for (int P = 0; P < NP; ++P) {
          const float2 y2 = *reinterpret_cast<const float2*>(ycol + yrow * LDW + 64 * P);
          dacc[2 * P] = __builtin_amdgcn_mfma_f32_32x32x2f32(y2.x, pv[r], dacc[2 * P], 0, 0, 0);
          dacc[2 * P + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y2.y, pv[r], dacc[2 * P + 1], 0, 0, 0);
        }
      }
    }
    wave_signal(&done_cnt[buf]);
    if (more) {
      const int b2 = (i + kScePD) % NB;
      wave_wait(&done_cnt[b2], 4 * ((i + kScePD) / NB));
      stage_write(b2);
      wave_signal(&full_cnt[b2]);
    }
  }

  // ---- results ------------------------------------------------------------------------------
  float inv_s = 1.f;
  const bool partial = MODE == 0 && a.G > 1;
  if (MODE == 0) {
    const float s_tot = run_s + __shfl_xor(run_s, 32);
    inv_s = partial ? 1.f : 1.f / s_tot;
    const float p_other = __shfl_xor(pos_l2, 32);
    const int64_t pc = my_pos - 4 * h;                 // which lane half saw the positive column?
    const bool mine = ((pc % 8) + 8) % 8 < 4;          // rows 8q + 4h + t, t < 4
    const float pl2 = mine ? pos_l2 : p_other;
    if (x_ok && h == 0) {
      if (partial) {
        const int64_t o = static_cast<int64_t>(gy) * a.nX + xi;
        a.part_m[o] = run_m;
        a.part_s[o] = s_tot;
        // only the range that holds the row's positive column has seen it
        const int64_t pst = my_pos / kTI;
        a.part_pos[o] = (pst >= st_lo && pst < st_lo + n_st) ? pl2 : kSceNeg;
      } else {
        a.lse[xi] = (run_m + __builtin_amdgcn_logf(s_tot)) * kLn2;
        a.pos_logit[xi] = pl2 * kLn2;
      }
    }
  }
  if (GEMM2) {
    float* out = MODE == 0 ? (partial ? a.part_W + static_cast<int64_t>(gy) * a.nX * D : a.W) : a.V;
    if (x_ok) {
#pragma unroll
      for (int P = 0; P < NP; ++P)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = 64 * P + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h);
          if (d < D) {
            float2 o;
            o.x = dacc[2 * P][r] * inv_s;
            o.y = dacc[2 * P + 1][r] * inv_s;
            *reinterpret_cast<float2*>(out + xi * D + d) = o;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two sweeps with every f32 product formed as a six-term split-bf16 product (f32 accumulate) on the bf16
// MFMA pipe (v_mfma_f32_32x32x16_bf16: 2.67x fewer pipe cycles per f32 product than v_mfma_f32_32x32x2_f32, and it
// leaves issue slots for the epilogue).  x = x1 + x2 + x3 exactly with x1 = bf16(x), x2 = bf16(x - x1), x3 =
// bf16(x - x1 - x2); a*b ~ a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1 (smallest first) — the error against f64 is
// that of the f32 fma chain (tests/test_softmax_ce_gpu.py compares both with the f64 shadow).
//   * 8 waves per workgroup, 32 stationary vectors each (256 per workgroup, one workgroup per CU = two waves per SIMD):
//     the stationary vectors are split ONCE into three bf16 planes held in registers as the B operand
//     (lane (j, g) of k-block kb: d = 16 kb + 8 g + e);
//   * a stage (32 streamed rows) is split by the staging threads as it is written to LDS: three row-major bf16
//     planes [32][DT] (row stride DT*2 + 16 bytes).  First contraction: A fragment = 16 bytes of a row;
//   * second contraction (reduction over the 32 streamed rows): B = P from the accumulator registers — register
//     r of lane (j, g) is streamed row 8 (r>>2) + 4 g + (r&3), so the eight reduction slots of k-block kb2 are
//     registers 8 kb2 .. 8 kb2 + 7 when slot e stands for row 16 kb2 + 8 (e>>2) + 4 g + (e&3); the A operand
//     (Y^T) takes the same rows of the SAME LDS image through the transposing read ds_read_b64_tr_b16 (a 16-lane
//     group reads a [4 rows][16 columns] block, lane i receiving column i).
// ------------------------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;

#ifndef LR_SCE_TR
#define LR_SCE_TR 1        // 0: the second contraction's A fragments by eight 2-byte reads (reference form of the layout)
#endif

#ifndef LR_SCE_WAVES
#define LR_SCE_WAVES 8     // waves per workgroup = waves per CU (8 = two per SIMD, 256 VGPRs each; 4 = one per SIMD, 512 VGPRs)
#endif
constexpr int kSbWaves = LR_SCE_WAVES;
constexpr int kSbThreads = 64 * kSbWaves;
constexpr int kSbRows = 32 * kSbWaves;        // stationary vectors per workgroup
#ifndef LR_SCE_PIPE
#define LR_SCE_PIPE 0      // 1: the second contraction of stage i-1 runs beside the softmax arithmetic of stage i, program order
#endif                     //    pinned MFMA by MFMA (needs ring depth >= 4).  Measured no faster than 0: profiles/r05_softmax_ce_sb.md
#ifndef LR_SCE_NB
#define LR_SCE_NB (LR_SCE_PIPE ? 4 : 3)        // LDS ring depth of the split-bf16 form
#endif
#ifndef LR_SCE_PD
#define LR_SCE_PD 2        // stages of prefetch in flight
#endif
#ifndef LR_SCE_PAD
#define LR_SCE_PAD 16      // bytes behind every plane row (16: conflict-free 16-byte fragment reads, 4-way conflicts of the transposing reads)
#endif
#ifndef LR_SCE_ABL
#define LR_SCE_ABL 0       // profiling only (wrong results): 1 = no LDS reads in the second contraction, 2 = none in the first
#endif
#ifndef LR_SCE_SKEW
#define LR_SCE_SKEW 0      // s_sleep argument (x 64 cycles) the second wave of every SIMD waits before its first stage
#endif
constexpr int kSbNB = LR_SCE_NB, kSbPD = LR_SCE_PD;

__device__ __forceinline__ uint32_t sce_pack2(float a, float b) {
  const f32x2v v = {a, b};
  const bf16x2v h = __builtin_convertvector(v, bf16x2v);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
__device__ __forceinline__ float sce_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float sce_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void sce_split4(float4 x, uint2& p1, uint2& p2, uint2& p3) {
  p1.x = sce_pack2(x.x, x.y); p1.y = sce_pack2(x.z, x.w);
  const float r0 = x.x - sce_lo(p1.x), r1 = x.y - sce_hi(p1.x), r2 = x.z - sce_lo(p1.y), r3 = x.w - sce_hi(p1.y);
  p2.x = sce_pack2(r0, r1); p2.y = sce_pack2(r2, r3);
  const float s0 = r0 - sce_lo(p2.x), s1 = r1 - sce_hi(p2.x), s2 = r2 - sce_lo(p2.y), s3 = r3 - sce_hi(p2.y);
  p3.x = sce_pack2(s0, s1); p3.y = sce_pack2(s2, s3);
}
__device__ __forceinline__ void sce_split8(float4 lo, float4 hi, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  uint2 l1, l2, l3, h1, h2, h3;
  sce_split4(lo, l1, l2, l3);
  sce_split4(hi, h1, h2, h3);
  const u32x4v v1 = {l1.x, l1.y, h1.x, h1.y}, v2 = {l2.x, l2.y, h2.x, h2.y}, v3 = {l3.x, l3.y, h3.x, h3.y};
  __builtin_memcpy(&a1, &v1, 16);
  __builtin_memcpy(&a2, &v2, 16);
  __builtin_memcpy(&a3, &v3, 16);
}
__device__ __forceinline__ void sce_mfma6(f32x16& acc, bf16x8 a1, bf16x8 a2, bf16x8 a3, bf16x8 b1, bf16x8 b2, bf16x8 b3) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
// [4 rows][16 columns] bf16 block per 16-lane group: every lane passes the address of ITS four contiguous source
// elements (row i>>2, columns 4 (i&3)..+3 of the block) and receives column i of the block, rows 0..3
__device__ __forceinline__ s16x4 sce_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p));
}

template <int DT, int MODE, bool GEMM2>
__global__ __launch_bounds__(kSbThreads, kSbWaves / 4) void softmax_ce_sb_kernel(SceArgs a) {
  constexpr int KB = DT / 16;               // k-blocks of the first contraction
  constexpr int NDT = DT / 32;              // 32-wide output tiles of the second contraction
  constexpr int ROWB = DT * 2 + LR_SCE_PAD; // bytes of one row of a plane
  constexpr int PLANE = 32 * ROWB;
  constexpr int STAGE = 3 * PLANE;
  constexpr int kTI = 32;
  constexpr int NB = kSbNB;
  constexpr int NQ = kTI * DT / 4;
  constexpr int NLD = NQ / kSbThreads;
  static_assert(DT == 64 || DT == 128, "compiled widths");
  static_assert(NLD >= 1 && NQ % kSbThreads == 0, "staging shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                     // [NB][3][32][ROWB]
  float* scf = reinterpret_cast<float*>(smem + NB * STAGE);              // [NB][2][32]
  int* sci = reinterpret_cast<int*>(scf + NB * 2 * 32);                  // [NB][32]
  int* full_cnt = sci + NB * 32;                                         // [NB]
  int* done_cnt = full_cnt + NB;                                         // [NB]
  if (threadIdx.x < 2 * NB) full_cnt[threadIdx.x] = 0;

  const int tid = threadIdx.x, wid = tid / kWave, lane = tid & (kWave - 1);
  const int j = lane & 31, h = lane >> 5;
  const int D = a.D;
  const bool has_bias = a.bias != nullptr, has_mask = a.idr != nullptr;
  const float* bias_p = has_bias ? a.bias : (MODE == 0 ? a.Y : a.X);
  const int32_t* idr_p = has_mask ? a.idr : reinterpret_cast<const int32_t*>(MODE == 0 ? a.X : a.Y);
  const int32_t* idc_p = has_mask ? a.idc : reinterpret_cast<const int32_t*>(MODE == 0 ? a.Y : a.X);

  // ---- stationary vectors: three bf16 planes of the B operand, resident in registers ------------------------
  const int64_t xi = (static_cast<int64_t>(blockIdx.x) * kSbWaves + wid) * 32 + j;
  const bool x_ok = xi < a.nX;
  const int64_t xc = x_ok ? xi : a.nX - 1;
  bf16x8 xb[KB][3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int d = 16 * kb + 8 * h;
    float4 lo = f4_zero(), hi = f4_zero();
    if (x_ok && d < D) lo = ld4(a.X + xi * D + d);
    if (x_ok && d + 4 < D) hi = ld4(a.X + xi * D + d + 4);
    sce_split8(lo, hi, xb[kb][0], xb[kb][1], xb[kb][2]);
  }
  int my_id;
  float my_bias2 = 0.f;
  if (MODE == 0) {
    my_id = has_mask ? idr_p[xc] : -1;
  } else {
    my_id = has_mask ? idc_p[xc] : -1;
    my_bias2 = has_bias ? bias_p[xc] * kLog2e : 0.f;
  }
  const int64_t my_pos = MODE == 0 ? a.pos0 + xi : xi - a.pos0;
  float run_m = kSceNeg, run_s = 0.f, pos_l2 = 0.f;

  f32x16 dacc[GEMM2 ? NDT : 1];
#pragma unroll
  for (int t = 0; t < (GEMM2 ? NDT : 1); ++t)
    dacc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // ---- staging -------------------------------------------------------------------------------------------
  float4 pre[NLD];
  float ps0 = 0.f, ps1 = 0.f;
  int pi0 = 0;
  uint32_t pre_ok = 0;
  const uint32_t Nu = static_cast<uint32_t>(a.nY), Du = static_cast<uint32_t>(D);
  auto stage_load = [&](int64_t st) {     // st may lie past the end: addresses are clamped
    pre_ok = 0;
    const uint32_t r0 = static_cast<uint32_t>(st * kTI < a.nY ? st * kTI : a.nY);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = tid + u * kSbThreads;
      const uint32_t row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
      const uint32_t r = r0 + row;
      if ((r < Nu) && (c4 < Du)) pre_ok |= 1u << u;
      const uint32_t rc = r < Nu ? r : Nu - 1;
      const uint32_t cc = c4 < Du ? c4 : Du - 4;
      pre[u] = ld4(a.Y + (static_cast<uint64_t>(rc) * Du + cc));
    }
    const uint32_t rs = r0 + (tid & 31);
    const uint32_t rsc = rs < Nu ? rs : Nu - 1;
    if (rs < Nu) pre_ok |= 1u << 31;
    if (tid < 32) {
      if (MODE == 0) {
        ps0 = bias_p[rsc];
        pi0 = idc_p[rsc];
      } else {
        ps0 = a.lse_in[rsc];
        ps1 = a.g[rsc];
        pi0 = idr_p[rsc];
      }
    }
  };
  auto stage_write = [&](int buf) {
    char* dst = tile + buf * STAGE;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = tid + u * kSbThreads;
      const int row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
      uint2 p1, p2, p3;
      sce_split4(((pre_ok >> u) & 1u) ? pre[u] : f4_zero(), p1, p2, p3);
      char* d0 = dst + row * ROWB + c4 * 2;
      *reinterpret_cast<uint2*>(d0) = p1;
      *reinterpret_cast<uint2*>(d0 + PLANE) = p2;
      *reinterpret_cast<uint2*>(d0 + 2 * PLANE) = p3;
    }
    if (tid < 32) {
      const bool ok = (pre_ok >> 31) & 1u;
      if (MODE == 0) {
        scf[(buf * 2 + 0) * 32 + tid] = ok ? (has_bias ? ps0 * kLog2e : 0.f) : kSceNeg;
        sci[buf * 32 + tid] = (ok && has_mask) ? pi0 : -2;
      } else {
        scf[(buf * 2 + 0) * 32 + tid] = ok ? ps0 * kLog2e : 0.f;
        scf[(buf * 2 + 1) * 32 + tid] = ok ? ps1 : 0.f;
        sci[buf * 32 + tid] = (ok && has_mask) ? pi0 : -2;
      }
    }
  };
  auto wave_signal = [&](int* c) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto wave_wait = [&](int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
  };

  const int n_st_all = static_cast<int>(ceil_div(a.nY, (int64_t)kTI));
  const int G = (MODE == 0 && a.G > 1) ? a.G : 1, gy = (MODE == 0 && a.G > 1) ? static_cast<int>(blockIdx.y) : 0;
  const int st_lo = static_cast<int>(static_cast<int64_t>(n_st_all) * gy / G);
  const int n_st = static_cast<int>(static_cast<int64_t>(n_st_all) * (gy + 1) / G) - st_lo;
  __syncthreads();
  for (int p = 0; p < kSbPD && p < n_st; ++p) {
    stage_load(st_lo + p);
    stage_write(p % NB);
    wave_signal(&full_cnt[p % NB]);
  }

  if (LR_SCE_SKEW > 0 && __builtin_amdgcn_readfirstlane(wid) >= 4) __builtin_amdgcn_s_sleep(LR_SCE_SKEW);
  // transposing-read geometry of this lane: 16-lane group gi reads rows 4 (gi>>1) .. +3, columns 16 (gi&1) .. +15
  const int gi = lane >> 4, li = lane & 15;
  const int tr_off = (4 * (gi >> 1) + (li >> 2)) * ROWB + (16 * (gi & 1) + 4 * (li & 3)) * 2;

  // ---- the pieces of a stage.  Everything below is inlined; a phase (A: first contraction + the split of the previous stage's
  // probabilities, B: second contraction of the PREVIOUS stage + this stage's softmax arithmetic) is one basic block, and the
  // sched_group_barrier sequences at its end deal its VALU instructions behind its MFMAs: a bf16 MFMA hides ~5 VALU instructions
  // issued behind it by the SAME wave, while the other wave of the SIMD hides next to nothing
  // (scripts/probes/mfma_valu_overlap_probe.hip: 96 MFMA + 480 VALU per stage and wave, two waves per SIMD: 8,172 cycles per
  // stage pair in separate phases, 6,215 dealt 5 behind each MFMA; the pipe's floor is 6,144) ----------------------------------
  auto gemm1 = [&](const char* src, f32x16& acc) {
    const char* arow = src + j * ROWB + (8 * h) * 2;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#if LR_SCE_ABL & 2
      const bf16x8 a1 = xb[kb][1], a2 = xb[kb][2], a3 = xb[kb][0];
#else
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow + kb * 32);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(arow + kb * 32 + PLANE);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(arow + kb * 32 + 2 * PLANE);
#endif
      sce_mfma6(acc, a1, a2, a3, xb[kb][0], xb[kb][1], xb[kb][2]);
    }
  };
  // reduction slot e of k-block kb2 of lane half h is streamed row 16 kb2 + 8 (e>>2) + 4 h + (e&3) = accumulator register 8 kb2 + e
  auto psplit = [&](const float (&pv)[16], bf16x8 (&pf)[2][3]) {
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
      sce_split8(make_float4(pv[8 * kb2], pv[8 * kb2 + 1], pv[8 * kb2 + 2], pv[8 * kb2 + 3]),
                 make_float4(pv[8 * kb2 + 4], pv[8 * kb2 + 5], pv[8 * kb2 + 6], pv[8 * kb2 + 7]),
                 pf[kb2][0], pf[kb2][1], pf[kb2][2]);
  };
  // second contraction: out^T[d][my vector] += sum_y Ystage[y][d] * pv[y]
  auto gemm2 = [&](const char* src, const bf16x8 (&pf)[2][3]) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        bf16x8 ya[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#if LR_SCE_ABL & 1
          ya[p] = pf[kb2 ^ 1][p];
          continue;
#endif
#if LR_SCE_TR
          const char* tp = src + p * PLANE + (16 * kb2) * ROWB + dt * 64 + tr_off;
          const s16x4 lo = sce_tr_read(tp), hi = sce_tr_read(tp + 8 * ROWB);
          const short v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#else
          short v8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v8[e] = *reinterpret_cast<const short*>(src + p * PLANE + (16 * kb2 + 8 * (e >> 2) + 4 * h + (e & 3)) * ROWB +
                                                    (32 * dt + j) * 2);
#endif
          __builtin_memcpy(&ya[p], v8, 16);
        }
        sce_mfma6(dacc[GEMM2 ? dt : 0], ya[0], ya[1], ya[2], pf[kb2][0], pf[kb2][1], pf[kb2][2]);
      }
    }
  };
  // register r of lane (j,h) = streamed row y = 8 (r>>2) + 4 h + (r&3) against my vector; returns the rescale factor of the
  // running maximum (MODE 0; 1 otherwise)
  auto epilogue = [&](int buf, int stage, const f32x16& acc, float (&pv)[16]) -> float {
    const int y0 = stage * kTI + 4 * h;
    if (MODE == 0) {
      float l2[16];
      float tmax = kSceNeg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = ld4(scf + (buf * 2 + 0) * 32 + 8 * q + 4 * h);
        const int4 idv = *reinterpret_cast<const int4*>(sci + buf * 32 + 8 * q + 4 * h);
        const float bb[4] = {b.x, b.y, b.z, b.w};
        const int ii[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = q * 4 + t;
          const int64_t col = y0 + 8 * q + t;
          float v = fmaf(acc[r], kLog2e, bb[t]);
          const bool is_pos = col == my_pos;
          if (ii[t] == my_id && !is_pos) v = kSceNeg;
          v = fmaxf(v, kSceNeg);
          if (is_pos) pos_l2 = v;
          l2[r] = v;
          tmax = fmaxf(tmax, v);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(run_m, tmax);
      const float alpha = __builtin_amdgcn_exp2f(run_m - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(l2[r] - m_new);
        psum += pv[r];
      }
      run_s = fmaf(run_s, alpha, psum);
      run_m = m_new;
      return alpha;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 ls = ld4(scf + (buf * 2 + 0) * 32 + 8 * q + 4 * h);
        const float4 gg = ld4(scf + (buf * 2 + 1) * 32 + 8 * q + 4 * h);
        const int4 idv = *reinterpret_cast<const int4*>(sci + buf * 32 + 8 * q + 4 * h);
        const float ll[4] = {ls.x, ls.y, ls.z, ls.w};
        const float gq[4] = {gg.x, gg.y, gg.z, gg.w};
        const int ii[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = q * 4 + t;
          const int64_t row = y0 + 8 * q + t;
          const float v = fmaf(acc[r], kLog2e, my_bias2);
          float p = __builtin_amdgcn_exp2f(fminf(v - ll[t], 0.f));
          if (ii[t] == my_id && row != my_pos) p = 0.f;
          pv[r] = gq[t] * p;
        }
      }
      return 1.f;
    }
  };
  auto rescale = [&](float alpha) {       // the running maximum moved: rescale W's partial sums
    if (MODE == 0 && GEMM2) {
      if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) dacc[t][r] *= alpha;
      }
    }
  };
  auto refill = [&](int i) {              // stage i + PD into its ring slot once every wave is done with the slot's last stage
    if (i + kSbPD < n_st) {
      const int b2 = (i + kSbPD) % NB;
      wave_wait(&done_cnt[b2], kSbWaves * ((i + kSbPD) / NB));
      stage_write(b2);
      wave_signal(&full_cnt[b2]);
    }
  };

#if LR_SCE_PIPE
  if (GEMM2) {
    float pvp[16];                        // the previous stage's probabilities: B operand of ITS second contraction
    {                                     // stage 0: nothing to overlap with yet
      stage_load(st_lo + kSbPD);
      wave_wait(&full_cnt[0], kSbWaves);
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      gemm1(tile, acc);
      epilogue(0, st_lo, acc, pvp);       // dacc is still zero: no rescale
      refill(0);
    }
    for (int i = 1; i < n_st; ++i) {
      const int buf = i % NB, bufp = (i - 1) % NB;
      stage_load(st_lo + i + kSbPD);
      wave_wait(&full_cnt[buf], kSbWaves * (i / NB + 1));
      const char* src = tile + buf * STAGE;
      // ---- phase A: S(i) = Ystage(i) X^T; the split of P(i-1) in twelve steps behind its MFMAs (program order is pinned by
      // a scheduling barrier after every MFMA + slice: the compiler's own order put all VALU work in front of / behind the MFMAs)
      bf16x8 pf[2][3];
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      {
        constexpr int AT[6] = {2, 0, 1, 1, 0, 0}, BT[6] = {0, 2, 1, 0, 1, 0};      // the six products, smallest first
        constexpr int GAP = (6 * KB) / 12;
        const char* arow = src + j * ROWB + (8 * h) * 2;
        bf16x8 an[3], ac[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const bf16x8*>(arow + p * PLANE);
        uint2 pq[3][4];
        float rr[4];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
          for (int p = 0; p < 3; ++p) ac[p] = an[p];
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            const int sl = kb * 6 + t;
            if (t < 3 && kb + 1 < KB) an[t] = *reinterpret_cast<const bf16x8*>(arow + (kb + 1) * 32 + t * PLANE);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[AT[t]], xb[kb][BT[t]], acc, 0, 0, 0);
            if (sl % GAP == 0 && sl / GAP < 12) {
              const int g4 = (sl / GAP) / 3, st = (sl / GAP) % 3;
              if (st == 0) {
                const float x0 = pvp[4 * g4], x1 = pvp[4 * g4 + 1], x2 = pvp[4 * g4 + 2], x3 = pvp[4 * g4 + 3];
                pq[0][g4].x = sce_pack2(x0, x1); pq[0][g4].y = sce_pack2(x2, x3);
                rr[0] = x0 - sce_lo(pq[0][g4].x); rr[1] = x1 - sce_hi(pq[0][g4].x);
                rr[2] = x2 - sce_lo(pq[0][g4].y); rr[3] = x3 - sce_hi(pq[0][g4].y);
              } else if (st == 1) {
                pq[1][g4].x = sce_pack2(rr[0], rr[1]); pq[1][g4].y = sce_pack2(rr[2], rr[3]);
                rr[0] -= sce_lo(pq[1][g4].x); rr[1] -= sce_hi(pq[1][g4].x);
                rr[2] -= sce_lo(pq[1][g4].y); rr[3] -= sce_hi(pq[1][g4].y);
              } else {
                pq[2][g4].x = sce_pack2(rr[0], rr[1]); pq[2][g4].y = sce_pack2(rr[2], rr[3]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u32x4v v = {pq[p][2 * kb2].x, pq[p][2 * kb2].y, pq[p][2 * kb2 + 1].x, pq[p][2 * kb2 + 1].y};
            __builtin_memcpy(&pf[kb2][p], &v, 16);
          }
      }
      // ---- phase B: W += P(i-1)^T Ystage(i-1), the softmax arithmetic of stage i in slices behind its MFMAs ----
      float pv[16];
      float alpha = 1.f;
      {
        constexpr int AT[6] = {2, 0, 1, 1, 0, 0}, BT[6] = {0, 2, 1, 0, 1, 0};
        constexpr int ORD[6] = {4, 5, 0, 1, 2, 3};        // read order of the next chunk: its first MFMA takes plane 3
        constexpr int NS = 12 * NDT, SPS = 48 / NS;       // MFMA slots, softmax slices per slot
        const char* srcp = tile + bufp * STAGE;
        auto tr_at = [&](int c, int k) -> s16x4 {
          const int dt = c >> 1, kb2 = c & 1, pl = k >> 1, t = k & 1;
#if LR_SCE_TR
          return sce_tr_read(srcp + pl * PLANE + (16 * kb2 + 8 * t) * ROWB + dt * 64 + tr_off);
#else
          s16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = *reinterpret_cast<const short*>(srcp + pl * PLANE + (16 * kb2 + 8 * t + 4 * h + e) * ROWB + (32 * dt + j) * 2);
          return v;
#endif
        };
        float4 sb0[2], sb1[2];
        int4 sid[2];
        auto load_q = [&](int q) {
          sb0[q & 1] = ld4(scf + (buf * 2 + 0) * 32 + 8 * q + 4 * h);
          if (MODE == 1) sb1[q & 1] = ld4(scf + (buf * 2 + 1) * 32 + 8 * q + 4 * h);
          sid[q & 1] = *reinterpret_cast<const int4*>(sci + buf * 32 + 8 * q + 4 * h);
        };
        // my positive's position inside this stage as seen from lane half h (anything outside 0..31: not here)
        const int64_t rel = my_pos - static_cast<int64_t>(st_lo + i) * kTI;
        const int relh = ((rel >= 0 && rel < kTI) ? static_cast<int>(rel) : -64) - 4 * h;
        load_q(0);
        s16x4 yn[6], yc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) yn[k] = tr_at(0, k);
        float l2[16];
        float tmax = kSceNeg, m_new = 0.f, psum = 0.f, c_v = 0.f;
        bool c_pos = false, c_hit = false;
        auto eslice = [&](int e) {
          if (MODE == 0) {
            if (e < 32) {
              const int r = e >> 1, q = r >> 2, t = r & 3;
              if ((e & 1) == 0) {
                if (t == 0 && q + 1 < 4) load_q(q + 1);
                const float bq[4] = {sb0[q & 1].x, sb0[q & 1].y, sb0[q & 1].z, sb0[q & 1].w};
                const int iq[4] = {sid[q & 1].x, sid[q & 1].y, sid[q & 1].z, sid[q & 1].w};
                c_v = fmaf(acc[r], kLog2e, bq[t]);
                c_pos = relh == 8 * q + t;
                c_hit = iq[t] == my_id;
              } else {
                float v = (c_hit && !c_pos) ? kSceNeg : c_v;
                v = fmaxf(v, kSceNeg);
                if (c_pos) pos_l2 = v;
                l2[r] = v;
                tmax = fmaxf(tmax, v);
              }
            } else {
              const int r = e - 32;
              if (r == 0) {
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                m_new = fmaxf(run_m, tmax);
                alpha = __builtin_amdgcn_exp2f(run_m - m_new);
              }
              pv[r] = __builtin_amdgcn_exp2f(l2[r] - m_new);
              psum += pv[r];
              asm volatile("" : "+v"(pv[r]));     // keeps the slice in ITS slot (pure arithmetic is otherwise sunk to its use)
            }
          } else {
            const int r = e / 3, q = r >> 2, t = r & 3, part = e % 3;
            if (part == 0) {
              if (t == 0 && q + 1 < 4) load_q(q + 1);
              const float lq[4] = {sb0[q & 1].x, sb0[q & 1].y, sb0[q & 1].z, sb0[q & 1].w};
              c_v = fminf(fmaf(acc[r], kLog2e, my_bias2) - lq[t], 0.f);
            } else if (part == 1) {
              const int iq[4] = {sid[q & 1].x, sid[q & 1].y, sid[q & 1].z, sid[q & 1].w};
              c_v = __builtin_amdgcn_exp2f(c_v);
              c_hit = iq[t] == my_id && relh != 8 * q + t;
            } else {
              const float gq[4] = {sb1[q & 1].x, sb1[q & 1].y, sb1[q & 1].z, sb1[q & 1].w};
              pv[r] = gq[t] * (c_hit ? 0.f : c_v);
              asm volatile("" : "+v"(pv[r]));
            }
          }
        };
#pragma unroll
        for (int c = 0; c < 2 * NDT; ++c) {
#pragma unroll
          for (int k = 0; k < 6; ++k) yc[k] = yn[k];
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            const int sl = c * 6 + t;
            if (c + 1 < 2 * NDT) yn[ORD[t]] = tr_at(c + 1, ORD[t]);
            const short v8[8] = {yc[2 * AT[t]][0], yc[2 * AT[t]][1], yc[2 * AT[t]][2], yc[2 * AT[t]][3],
                                 yc[2 * AT[t] + 1][0], yc[2 * AT[t] + 1][1], yc[2 * AT[t] + 1][2], yc[2 * AT[t] + 1][3]};
            bf16x8 ya;
            __builtin_memcpy(&ya, v8, 16);
            dacc[c >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, pf[c & 1][BT[t]], dacc[c >> 1], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < SPS; ++u) eslice(sl * SPS + u);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (MODE == 0) {
          run_s = fmaf(run_s, alpha, psum);
          run_m = m_new;
        }
      }
      rescale(alpha);
#pragma unroll
      for (int r = 0; r < 16; ++r) pvp[r] = pv[r];
      wave_signal(&done_cnt[bufp]);
      refill(i);
    }
    {                                     // the last stage's second contraction
      bf16x8 pf[2][3];
      psplit(pvp, pf);
      gemm2(tile + ((n_st - 1) % NB) * STAGE, pf);
    }
  } else
#endif
  {
    for (int i = 0; i < n_st; ++i) {
      const int buf = i % NB;
      stage_load(st_lo + i + kSbPD);
      wave_wait(&full_cnt[buf], kSbWaves * (i / NB + 1));
      const char* src = tile + buf * STAGE;
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      gemm1(src, acc);
      float pv[16];
      const float alpha = epilogue(buf, st_lo + i, acc, pv);
      if (GEMM2) {
        rescale(alpha);
        bf16x8 pf[2][3];
        psplit(pv, pf);
        gemm2(src, pf);
      }
      wave_signal(&done_cnt[buf]);
      refill(i);
    }
  }

  // ---- results ---------------------------------------------------------------------------------------------
  float inv_s = 1.f;
  const bool partial = MODE == 0 && a.G > 1;
  if (MODE == 0) {
    const float s_tot = run_s + __shfl_xor(run_s, 32);
    inv_s = partial ? 1.f : 1.f / s_tot;
    const float p_other = __shfl_xor(pos_l2, 32);
    const int64_t pc = my_pos - 4 * h;
    const bool mine = ((pc % 8) + 8) % 8 < 4;
    const float pl2 = mine ? pos_l2 : p_other;
    if (x_ok && h == 0) {
      if (partial) {
        const int64_t o = static_cast<int64_t>(gy) * a.nX + xi;
        a.part_m[o] = run_m;
        a.part_s[o] = s_tot;
        const int64_t pst = my_pos / kTI;
        a.part_pos[o] = (pst >= st_lo && pst < st_lo + n_st) ? pl2 : kSceNeg;
      } else {
        a.lse[xi] = (run_m + __builtin_amdgcn_logf(s_tot)) * kLn2;
        a.pos_logit[xi] = pl2 * kLn2;
      }
    }
  }
  if (GEMM2) {
    float* out = MODE == 0 ? (partial ? a.part_W + static_cast<int64_t>(gy) * a.nX * D : a.W) : a.V;
    if (x_ok) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // registers 4q .. 4q+3 of lane (j, h): d = 32 dt + 8 q + 4 h + 0..3 of vector j
          const int d = 32 * dt + 8 * q + 4 * h;
          if (d < D)
            st4(out + xi * D + d, make_float4(dacc[dt][4 * q] * inv_s, dacc[dt][4 * q + 1] * inv_s,
                                              dacc[dt][4 * q + 2] * inv_s, dacc[dt][4 * q + 3] * inv_s));
        }
    }
  }
}

// combine the G column ranges of a row: m = max m_g, s = sum s_g 2^(m_g - m), W = sum W_g 2^(m_g - m) / s
// (range order: fixed), the positive logit from the one range that saw it
__global__ __launch_bounds__(kBlock) void sce_merge_kernel(SceArgs a) {
  const int D = a.D, D4 = D / 4;
  const int64_t total = a.nX * D4;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total;
       q += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t xi = q / D4;
    const int c4 = static_cast<int>(q - xi * D4) * 4;
    float m = kSceNeg;
    for (int g = 0; g < a.G; ++g) m = fmaxf(m, a.part_m[static_cast<int64_t>(g) * a.nX + xi]);
    float s = 0.f, pos = kSceNeg;
    float4 w = f4_zero();
    for (int g = 0; g < a.G; ++g) {
      const int64_t o = static_cast<int64_t>(g) * a.nX + xi;
      const float f = __builtin_amdgcn_exp2f(a.part_m[o] - m);
      s = fmaf(a.part_s[o], f, s);
      pos = fmaxf(pos, a.part_pos[o]);
      if (a.W != nullptr) w = f4_fma(make_float4(f, f, f, f), ld4(a.part_W + o * D + c4), w);
    }
    if (a.W != nullptr) st4(a.W + xi * D + c4, f4_scale(w, 1.f / s));
    if (c4 == 0) {
      a.lse[xi] = (m + __builtin_amdgcn_logf(s)) * kLn2;
      a.pos_logit[xi] = pos * kLn2;
    }
  }
}

// `arith` — an explicit argument of every entry point (no library-side state: two callers with different settings cannot race
// between the workspace query and the launch): 1 = six-term split-bf16 products with f32 accumulation, 0 = the f32 fma chain
static int sce_ranges(int64_t B, int64_t N, int arith) {
  // fewer workgroups on the stationary side than fill the chip (f32 form: 128 rows per workgroup, two per CU; split-bf16
  // form: 256 rows, one per CU): cut the streamed side, >= 8 stages per range
  const int64_t tiles = ceil_div(B, arith ? kSbRows : 128), stages = ceil_div(N, 32);
  const int64_t want = arith ? kNumCU : 2 * kNumCU;
  if (tiles >= want) return 1;
  int64_t G = ceil_div(want, tiles);
  if (G > stages / 8) G = stages / 8;
  if (G > 16) G = 16;
  return G < 2 ? 1 : static_cast<int>(G);
}

static size_t sce_lds_bytes(int DT) {
  return static_cast<size_t>(kSceNB) * 32 * (DT + 4) * 4 + kSceNB * 3 * 32 * 4 + 2 * kSceNB * 4 + 16;
}

static size_t sce_sb_lds_bytes(int DT) {
  return static_cast<size_t>(kSbNB) * 3 * 32 * (DT * 2 + LR_SCE_PAD) + kSbNB * 3 * 32 * 4 + 2 * kSbNB * 4 + 16;
}

template <int DT, int MODE, bool GEMM2>
static int sce_launch(const SceArgs& a, hipStream_t s, int arith) {
  if (arith) {
    const size_t lds = sce_sb_lds_bytes(DT);
    auto kern = softmax_ce_sb_kernel<DT, MODE, GEMM2>;
    static bool lds_set = false;
    if (!lds_set && lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return static_cast<int>(e);
      lds_set = true;
    }
    const int grid = static_cast<int>(ceil_div(a.nX, kSbRows));
    hipLaunchKernelGGL(kern, dim3(grid, (MODE == 0 && a.G > 1) ? a.G : 1), dim3(kSbThreads), lds, s, a);
  } else {
    const size_t lds = sce_lds_bytes(DT);
    auto kern = softmax_ce_kernel<DT, MODE, GEMM2>;
    static bool lds_set = false;
    if (!lds_set && lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return static_cast<int>(e);
      lds_set = true;
    }
    const int grid = static_cast<int>(ceil_div(a.nX, 128));
    hipLaunchKernelGGL(kern, dim3(grid, (MODE == 0 && a.G > 1) ? a.G : 1), dim3(kBlock), lds, s, a);
  }
  if (MODE == 0 && a.G > 1)
    hipLaunchKernelGGL(sce_merge_kernel, dim3(grid_for(a.nX * (a.D / 4), kBlock)), dim3(kBlock), 0, s, a);
  return launch_status();
}

static inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

static bool sce_shape_ok(int64_t B, int64_t N, int D) {
  return B >= 1 && N >= 1 && D >= 4 && D <= 128 && D % 4 == 0 && B < (int64_t(1) << 31) && N < (int64_t(1) << 31);
}

}  // namespace lr

using namespace lr;

extern "C" int lr_softmax_ce_supported(int64_t B, int64_t N, int D) { return sce_shape_ok(B, N, D) ? 1 : 0; }

extern "C" size_t lr_softmax_ce_fwd_ws_bytes(int64_t B, int64_t N, int D, int arith) {
  if (!sce_shape_ok(B, N, D) || (arith != 0 && arith != 1)) return 0;
  const int G = sce_ranges(B, N, arith);
  return G > 1 ? static_cast<size_t>(G) * B * (3 + D) * sizeof(float) : 0;
}

extern "C" int lr_softmax_ce_fwd_f32(const float* X, int64_t B, const float* Y, int64_t N, int D,
                                     const float* col_bias, const int32_t* row_ids, const int32_t* col_ids,
                                     int64_t pos0, float* lse, float* pos_logit, float* W, void* ws,
                                     size_t ws_bytes, int arith, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && N >= 1 && D >= 1 && (arith == 0 || arith == 1));
  if (B == 0) return LR_OK;
  if (!sce_shape_ok(B, N, D)) return LR_ESHAPE;
  LR_CHECK_ARG(X && Y && lse && pos_logit);
  LR_CHECK_ARG((row_ids == nullptr) == (col_ids == nullptr));
  LR_CHECK_ARG(pos0 >= 0 && pos0 + B <= N);
  LR_CHECK_ARG(al16(X) && al16(Y) && (!W || al16(W)));
  SceArgs a{};
  a.X = X; a.nX = B; a.Y = Y; a.nY = N; a.D = D; a.bias = col_bias; a.idr = row_ids; a.idc = col_ids;
  a.pos0 = pos0; a.lse = lse; a.pos_logit = pos_logit; a.W = W;
  a.G = sce_ranges(B, N, arith);
  if (a.G > 1) {
    if (ws == nullptr || ws_bytes < lr_softmax_ce_fwd_ws_bytes(B, N, D, arith)) return LR_EWORKSPACE;
    LR_CHECK_ARG(al16(ws));
    float* p = static_cast<float*>(ws);
    const int64_t gb = static_cast<int64_t>(a.G) * B;
    a.part_W = p;                        // 16-byte aligned rows first
    a.part_m = p + gb * D;
    a.part_s = a.part_m + gb;
    a.part_pos = a.part_s + gb;
  }
  hipStream_t s = as_stream(stream);
  if (D <= 64) return W ? sce_launch<64, 0, true>(a, s, arith) : sce_launch<64, 0, false>(a, s, arith);
  return W ? sce_launch<128, 0, true>(a, s, arith) : sce_launch<128, 0, false>(a, s, arith);
}

extern "C" int lr_softmax_ce_bwd_cols_f32(const float* X, int64_t B, const float* Y, int64_t N, int D,
                                          const float* col_bias, const int32_t* row_ids,
                                          const int32_t* col_ids, int64_t pos0, const float* lse,
                                          const float* g, float* V, int arith, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 1 && N >= 0 && D >= 1 && (arith == 0 || arith == 1));
  if (N == 0) return LR_OK;
  if (!sce_shape_ok(B, N, D)) return LR_ESHAPE;
  LR_CHECK_ARG(X && Y && lse && g && V);
  LR_CHECK_ARG((row_ids == nullptr) == (col_ids == nullptr));
  LR_CHECK_ARG(pos0 >= 0 && pos0 + B <= N);
  LR_CHECK_ARG(al16(X) && al16(Y) && al16(V));
  SceArgs a{};
  // columns are the stationary side here: X <-> Y swap roles inside the kernel
  a.X = Y; a.nX = N; a.Y = X; a.nY = B; a.D = D; a.bias = col_bias; a.idr = row_ids; a.idc = col_ids;
  a.pos0 = pos0; a.lse_in = lse; a.g = g; a.V = V;
  hipStream_t s = as_stream(stream);
  if (D <= 64) return sce_launch<64, 1, true>(a, s, arith);
  return sce_launch<128, 1, true>(a, s, arith);
}
