// DeepFM first MLP layer fused with the embedding lookup, on the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TF dense peak on MI355X).
//
// The reference concatenates the F gathered rows of a sample into deep_embed [B, F*K]
// (algorithms/deepfm.py:236-247), applies batch-statistics BatchNorm and the first Dense of
// dense_nn (layers/dense.py:30-41, deepfm.py:163-169).  Materialising that block costs 847 MB per
// step at BASELINE cfg 2 and every consumer (BN statistics, three GEMMs, the FM backward) re-reads
// it.  Here it never exists:
//
//   lr_deepfm_l1_fwd_f32    z1 = gather(table, idx) @ Wp + bias  (+ FM sum / pairwise term + linear
//                           weights from the same staged rows).  Wp is the BatchNorm-folded kernel.
//   lr_deepfm_l1_wgrad_f32  dWp = gather(table, idx)^T @ gz      (re-gathers; per-field reduction over
//                           the batch, split into batch chunks -> partial slabs, fixed-order sum)
//   lr_deepfm_l1_dgrad_f32  ge[slot(b,f)] = gz[b] @ Wp_f^T + gl[b] * wp * fsum[b]   per-position row
//                           gradient written straight into RUN ORDER (slot = position of (b,f) in the
//                           batch's CSR-by-row), so that the Adam kernel streams it sequentially.
//
// Common structure: one workgroup = 4 waves (one per SIMD), TS = 32 or 64 samples per tile (32: two
// to three workgroups share a CU, so one workgroup's staging / barrier phases hide behind another's
// MFMA chain — measured on cfg 2: 64-sample tiles with one wave per SIMD reach 32-51 % of the f32
// MFMA peak), the reduction
// index of every MFMA is permuted so that both operands are read 16 bytes at a time (lane half h
// owns a contiguous half of the reduction range).  Gathered rows go HBM -> VGPR -> LDS (rows padded
// by 16 B: conflict-free ds_read_b128), one field / slab ahead of the MFMAs; weights are pre-packed
// in fragment order (lr_deepfm_l1_pack_f32) so that every lane's operand is one coalesced 16-byte
// load per four MFMAs.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace lr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Profiling builds only (scripts/lab/r04/ablate.sh compiles this file with -DLR_L1_ABLATE=<bits> into a separate
// library loaded through LIBRECO_HIP_LIB): parts of the wide forward kernel are switched off to see what its time is
// made of.  1: every gather reads row 0 (no HBM latency)  2: weights never refilled (no L2 stream)  4: no barrier
// 8: no LDS staging traffic  16: no MFMA.  The product build defines nothing: kAblate == 0 folds every test away.
#ifndef LR_L1_ABLATE
#define LR_L1_ABLATE 0
#endif
constexpr int kAblate = LR_L1_ABLATE;


__device__ __forceinline__ f32x16 acc_zero() {
  return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}
// row of accumulator register r inside a 32x32 tile for lane half h (guide §3: C/D layout)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// -----------------------------------------------------------------------------------------
// Packing of the folded kernel Wp [F*KD, H1] (row-major) into fragment order.
//   WpA (forward: B operand, k = embedding dim, j = output column)
//       float4 index ((f*(H1/32) + ct)*(KD/8) + s4)*64 + lane, component c
//         = Wp[f*KD + h*KD/2 + s4*4 + c][ct*32 + j]            lane = h*32 + j
//   WpB (dgrad: B operand, k = output column, j = embedding dim)
//       float4 index ((f*(KD/32) + ni)*(H1/8) + s4)*64 + lane, component c
//         = Wp[f*KD + ni*32 + j][h*H1/2 + s4*4 + c]
// -----------------------------------------------------------------------------------------
// `red_partial` (nullable): the LAST ceil(H1 / 16) workgroups of the launch do not pack — they sum the folded bias's slab partials
// (lr_reduce_partials_f32's arithmetic: reduce_partials_body) into `red_out`, so that reduction needs no launch of its own
__global__ __launch_bounds__(kBlock) void l1_pack_kernel(const float* __restrict__ Wp,
                                                         const float* __restrict__ scale, int F, int KD,
                                                         int H1, float* __restrict__ WpA,
                                                         float* __restrict__ WpB, const float* __restrict__ red_partial,
                                                         int red_nblk, float* __restrict__ red_out) {
  const int n_red = red_partial != nullptr ? (H1 + 15) / 16 : 0;
  const int n_pack = static_cast<int>(gridDim.x) - n_red;
  if (static_cast<int>(blockIdx.x) >= n_pack) {
    reduce_partials_body(static_cast<int>(blockIdx.x) - n_pack, n_red, red_partial, red_nblk, H1, H1, red_out, nullptr);
    return;
  }
  const int64_t total = static_cast<int64_t>(F) * KD * H1 / 4;   // float4 slots per buffer
  const int64_t stride = static_cast<int64_t>(n_pack) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int lane = static_cast<int>(q & 63);
    const int j = lane & 31, h = lane >> 5;
    {
      int64_t t = q >> 6;
      const int s4 = static_cast<int>(t % (KD / 8)); t /= (KD / 8);
      const int ct = static_cast<int>(t % (H1 / 32));
      const int f = static_cast<int>(t / (H1 / 32));
      const int64_t row = static_cast<int64_t>(f) * KD + h * (KD / 2) + s4 * 4;
      const int col = ct * 32 + j;
      float4 v;
      v.x = Wp[(row + 0) * H1 + col];
      v.y = Wp[(row + 1) * H1 + col];
      v.z = Wp[(row + 2) * H1 + col];
      v.w = Wp[(row + 3) * H1 + col];
      if (scale != nullptr) {   // Wp = diag(scale) W (BatchNorm fold)
        v.x *= scale[row + 0]; v.y *= scale[row + 1]; v.z *= scale[row + 2]; v.w *= scale[row + 3];
      }
      st4(WpA + q * 4, v);
    }
    {
      int64_t t = q >> 6;
      const int s4 = static_cast<int>(t % (H1 / 8)); t /= (H1 / 8);
      const int ni = static_cast<int>(t % (KD / 32));
      const int f = static_cast<int>(t / (KD / 32));
      const int64_t row = static_cast<int64_t>(f) * KD + ni * 32 + j;
      float4 w = ld4(Wp + row * H1 + h * (H1 / 2) + s4 * 4);
      if (scale != nullptr) w = f4_scale(w, scale[row]);
      st4(WpB + q * 4, w);
    }
  }
}

// [B, F] -> [F, B] (32x32 tiles through LDS, coalesced both ways)
__global__ __launch_bounds__(kBlock) void idx_transpose_kernel(const int32_t* __restrict__ idx,
                                                               int64_t B, int F,
                                                               int32_t* __restrict__ idxT) {
  __shared__ int32_t tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int64_t tiles_f = (F + 31) / 32, tiles_b = (B + 31) / 32;
  for (int64_t t = blockIdx.x; t < tiles_f * tiles_b; t += gridDim.x) {
    const int64_t b0 = (t / tiles_f) * 32;
    const int f0 = static_cast<int>(t % tiles_f) * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int64_t b = b0 + r;
      const int f = f0 + tx;
      tile[r][tx] = (b < B && f < F) ? idx[b * F + f] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int f = f0 + r;
      const int64_t b = b0 + tx;
      if (b < B && f < F) idxT[static_cast<int64_t>(f) * B + b] = tile[tx][r];
    }
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------------------
// Forward.  grid = ceil(B / TS) workgroups; workgroup tile = TS samples x H1 outputs.
//   LDS: rows[4][TS][KD+4] (two field pairs in flight)  |  idc[2][TS][32] (the tile's row ids in CHUNKS of 32
//   fields, each slot overwritten in place by the gathered linear weight once consumed; a chunk is copied out
//   to lin_out and refilled with the ids of the chunk after next while the MFMAs of the chunk between run).
//   The id tile used to hold all F fields (25.8 KB at F = 202, TS = 32): with the chunks the workgroup needs
//   43 KB whatever F is, and three workgroups share a CU (K <= 64).
//   wave w: output columns {32*(w + 4*c)} (H1 >= 128: both 32-sample tiles, NC = H1/128 column
//   tiles) or, for H1 == 64, column tile w&1 of sample tile w>>1.
// -----------------------------------------------------------------------------------------
template <int KD, int H1, int kTS>
struct L1Fwd {
  static constexpr int LDW = KD + 4;
  static constexpr int CPR = KD / 4;                 // 16-byte chunks per row
  static constexpr int RPP = kBlock / CPR;           // rows staged per pass of the workgroup
  static constexpr int NLD = kTS / RPP;              // float4 per thread per field
  static constexpr int KH = KD / 2;                  // reduction values per lane half and field
  static constexpr bool kWide = H1 >= 128;
  static constexpr int NC = kWide ? H1 / 128 : 1;    // column tiles per wave
  static constexpr int NS = kWide ? kTS / 32 : 1;    // sample tiles per wave
  static constexpr int CT = H1 / 32;                 // column tiles in total
  static_assert(KD % 16 == 0 && KD >= 16 && KD <= 128, "embed size");
  static_assert((H1 == 64 && kTS == 64) || H1 % 128 == 0, "first hidden width");
  static_assert(kTS % RPP == 0 && (kTS == 32 || kTS == 64), "stage passes");
  static constexpr int FC = 32;                      // fields per id chunk
  static constexpr int NCH = kTS * FC / kBlock;      // chunk elements per thread
  static size_t lds_bytes(int) {
    return static_cast<size_t>(4) * kTS * LDW * 4 + static_cast<size_t>(2) * kTS * FC * 4;   // 4 row buffers + 2 id chunks
  }
};

template <int KD, int H1, int kTS>
__global__ __launch_bounds__(kBlock, ((kTS == 32 && KD <= 64) ? 2 : 1)) void l1_fwd_kernel(
    const float* __restrict__ table, const float* __restrict__ lin, int64_t V,
    const int32_t* __restrict__ idx, int64_t B, int F, const float* __restrict__ WpA,
    const float* __restrict__ bias, float* __restrict__ z1, float* __restrict__ pair,
    float* __restrict__ fsum, float* __restrict__ lin_out) {
  using C = L1Fwd<KD, H1, kTS>;
  constexpr int LDW = C::LDW, CPR = C::CPR, RPP = C::RPP, NLD = C::NLD, KH = C::KH;
  constexpr int NC = C::NC, NS = C::NS, CT = C::CT, FC = C::FC, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);                          // [4][kTS][LDW]: field f in buffer f & 3
  int32_t* idc = reinterpret_cast<int32_t*>(smem + 4 * kTS * LDW * 4);   // [2][kTS][FC]: chunk c in buffer c & 1

  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTS;
  const int nb = (B - b0) < kTS ? static_cast<int>(B - b0) : kTS;        // valid samples of the tile

  // ---- id chunks: element e = tid + kBlock * i of a chunk is (sample e / FC, field 32 c + e % FC); a thread
  // flushes / refills the SAME elements, so a refill needs no barrier against its own flush ------------------
  auto chunk_load = [&](int c) {
    int32_t* dst = idc + (c & 1) * kTS * FC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = tid + kBlock * i, r = e / FC, f = c * FC + (e % FC);
      dst[e] = (r < nb && f < F) ? idx[(b0 + r) * F + f] : -1;
    }
  };
  auto chunk_flush = [&](int c) {          // the chunk's slots hold the gathered linear weights by now
    const float* src = reinterpret_cast<const float*>(idc + (c & 1) * kTS * FC);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = tid + kBlock * i, r = e / FC, f = c * FC + (e % FC);
      if (r < nb && f < F) lin_out[(b0 + r) * F + f] = src[e];
    }
  };
  auto id_slot = [&](int row, int f) -> int32_t* { return idc + ((f >> 5) & 1) * kTS * FC + row * FC + (f & (FC - 1)); };
  chunk_load(0);
  if (F > FC) chunk_load(1);

  // staging role of this thread: rows srow + u*RPP, chunk c4 of each
  const int srow = tid / CPR, c4 = (tid % CPR) * 4;
  // two register sets: the fields of the NEXT pair are in flight while the current pair is computed
  float4 pre[2][NLD];
  float prel[2][NLD];
  uint32_t pre_ok[2] = {0u, 0u};
  float4 S[NLD], Q[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) { S[u] = f4_zero(); Q[u] = f4_zero(); prel[0][u] = prel[1][u] = 0.f; }
  const uint32_t Vu = static_cast<uint32_t>(V);

  auto stage_load = [&](int f, auto set_c) {   // global -> registers (rows of field f), branch-free
    constexpr int set = decltype(set_c)::value;
    pre_ok[set] = 0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int32_t id = *id_slot(srow + u * RPP, f);
      const bool ok = static_cast<uint32_t>(id) < Vu;
      const uint32_t idc_ = ok ? static_cast<uint32_t>(id) : 0u;
      if (ok) pre_ok[set] |= 1u << u;
      pre[set][u] = ld4(table + static_cast<uint64_t>(idc_) * KD + c4);
      if (lin != nullptr && c4 == 0) prel[set][u] = lin[idc_];
    }
  };
  auto stage_write = [&](int f, auto set_c) {  // registers -> LDS buffer f & 3 (+ FM sums, linear weights)
    constexpr int set = decltype(set_c)::value;
    float* dst = rows + (f & 3) * kTS * LDW;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = (pre_ok[set] >> u) & 1u;
      const float4 x = ok ? pre[set][u] : f4_zero();
      S[u] = f4_add(S[u], x);
      Q[u] = f4_fma(x, x, Q[u]);
      st4(dst + (srow + u * RPP) * LDW + c4, x);
      if (lin != nullptr && c4 == 0)
        *reinterpret_cast<float*>(id_slot(srow + u * RPP, f)) = ok ? prel[set][u] : 0.f;
    }
  };

  // ---- this wave's output tiles ---------------------------------------------------------
  const int ct0 = C::kWide ? wid : (wid & 1);          // first column tile
  const int st0 = C::kWide ? 0 : (wid >> 1);           // first sample tile
  f32x16 acc[NC][NS];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[c][s] = acc_zero();

  float4 bw0[NC][KH / 4], bw1[NC][KH / 4];              // weight fragments, two fields in flight
  auto load_w = [&](int f, float4 (&bw)[NC][KH / 4]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* p = WpA + ((static_cast<int64_t>(f) * CT + ct0 + 4 * c) * (KD / 8)) * 256 + lane * 4;
#pragma unroll
      for (int s4 = 0; s4 < KH / 4; ++s4) bw[c][s4] = ld4(p + s4 * 256);
    }
  };
  // A fragments of a field are read in ONE burst (a single LDS wait) and the MFMA chain then runs
  // without interruption; left to itself the compiler re-uses one register quad and waits for LDS
  // in front of every group of four MFMAs.
  auto compute = [&](int buf, const float4 (&bw)[NC][KH / 4]) {
    const float* src = rows + buf * kTS * LDW + j * LDW + h * KH;
    float4 a[NS][KH / 4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int s4 = 0; s4 < KH / 4; ++s4) a[s][s4] = ld4(src + (st0 + s) * 32 * LDW + s4 * 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s4 = 0; s4 < KH / 4; ++s4)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[c][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][s4].x, bw[c][s4].x, acc[c][s], 0, 0, 0);
          acc[c][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][s4].y, bw[c][s4].y, acc[c][s], 0, 0, 0);
          acc[c][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][s4].z, bw[c][s4].z, acc[c][s], 0, 0, 0);
          acc[c][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][s4].w, bw[c][s4].w, acc[c][s], 0, 0, 0);
        }
  };

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  __syncthreads();                      // id chunks 0 / 1 visible
  load_w(0, bw0);
  stage_load(0, Set0{});
  if (F > 1) stage_load(1, Set1{});
  stage_write(0, Set0{});
  if (F > 1) stage_write(1, Set1{});
  if (F > 2) stage_load(2, Set0{});
  if (F > 3) stage_load(3, Set1{});
  __syncthreads();
  // TWO fields per barrier: the pair (f, f+1) is computed from LDS buffers f & 3, (f+1) & 3 while the next
  // pair is written into the other two buffers (last read one pair ago: every wave is past that pair's
  // barrier) and the pair after that is requested from HBM.  One field per barrier left the waves with 32
  // MFMAs (2,048 cycles) between barriers.  The steady-state body is free of conditionals: a branch around a
  // load leaves the compiler's wait-count bookkeeping with "maybe pending" registers at the join and it then
  // fences every MFMA group behind loads that were issued for later fields (seen in the ISA of the first
  // version: vmcnt(7)...vmcnt(0) in front of the eight MFMA groups).  The id-chunk rotation therefore sits
  // BETWEEN runs of 16 steady iterations (outer loop over chunks), not inside them: at f = 32 c every slot of chunk
  // c - 1 holds its linear weight (last one written at iteration f - 4) and no wave reads its ids any more (last
  // read at iteration f - 6), chunk c + 1 is first needed at iteration f + 28.
  int f = 0, flushed = 0;
  const int f_steady = F - 5;           // steady iterations: f < f_steady
  for (int c = 0; f < f_steady; ++c) {
    if (c > 0) {
      if (lin != nullptr) chunk_flush(c - 1);
      flushed = c;
      chunk_load(c + 1);                // (ids of fields >= F read as -1: never staged)
    }
    const int f_end = (c + 1) * FC < f_steady ? (c + 1) * FC : f_steady;
    for (; f < f_end; f += 2) {
      load_w(f + 1, bw1);
      compute(f & 3, bw0);
      load_w(f + 2, bw0);
      compute((f + 1) & 3, bw1);
      stage_write(f + 2, Set0{});
      stage_write(f + 3, Set1{});
      stage_load(f + 4, Set0{});
      stage_load(f + 5, Set1{});
      __syncthreads();
    }
  }
  for (; f < F; f += 2) {               // the last pairs
    if (f + 1 < F) load_w(f + 1, bw1);
    compute(f & 3, bw0);
    if (f + 2 < F) load_w(f + 2, bw0);
    if (f + 1 < F) compute((f + 1) & 3, bw1);
    if (f + 2 < F) stage_write(f + 2, Set0{});
    if (f + 3 < F) stage_write(f + 3, Set1{});
    if (f + 4 < F) stage_load(f + 4, Set0{});
    if (f + 5 < F) stage_load(f + 5, Set1{});
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (ct0 + 4 * c) * 32 + j;
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int smp = (st0 + s) * 32 + acc_row(r, h);
        if (smp < nb) z1[(b0 + smp) * H1 + col] = acc[c][s][r] + bv;
      }
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int smp = srow + u * RPP;
    if (smp < nb) {
      float4 p;
      p.x = 0.5f * (S[u].x * S[u].x - Q[u].x);
      p.y = 0.5f * (S[u].y * S[u].y - Q[u].y);
      p.z = 0.5f * (S[u].z * S[u].z - Q[u].z);
      p.w = 0.5f * (S[u].w * S[u].w - Q[u].w);
      st4(pair + (b0 + smp) * KD + c4, p);
      if (fsum != nullptr) st4(fsum + (b0 + smp) * KD + c4, S[u]);
    }
  }
  if (lin != nullptr) {   // the chunks still resident hold linear weights: copy them out
    for (int c = flushed; c * FC < F; ++c) chunk_flush(c);
  }
}

// -----------------------------------------------------------------------------------------
// Forward, 64-sample tiles, ONE wave per SIMD (round 4).
//
// Why: PMC of the 32-sample kernel above on cfg 2 (profiles/r04_pmc_l1.md): the f32 MFMA pipe is busy 65 % of the
// kernel's cycles at an effective clock of 2.0-2.1 GHz, while softmax_ce.hip (stationary operand in VGPRs) holds 81 %
// at 2.3-2.4 GHz.  Every 32-sample workgroup streams the whole packed kernel (6.6 MB) out of L2: 512 workgroups x
// 6.6 MB = 3.4 GB per launch (5.7 TB/s of L2 -> CU traffic, 35.7 M TCP -> TCC requests), and the two waves of a SIMD
// (one per co-resident workgroup) drift into the same phase: both in their MFMA chains, then both in their
// stage / barrier sections.
// Here a workgroup owns 64 samples x H1 outputs (grid = B / 64 = one workgroup per CU at cfg 2's B = 16,384) and each
// of its 4 waves TWO 32x32 accumulators per column tile (sample tiles 0 / 1 against the wave's column tile): a weight
// fragment feeds two MFMAs, so the L2 -> CU weight traffic halves.  With one wave per SIMD nothing hides behind a
// partner wave, so the instruction stream itself is the schedule: a field's 16 NC MFMA groups are the time base
// (8 NC MFMAs = 512 NC cycles of the pipe each) and every other instruction of the pipeline is dealt to a fixed group
// and pinned there with sched_barrier:
//   groups 0 .. NQ/2-1  (first half of field f's chain):  rows of field f+1: registers -> LDS buffer (f+1)&1, one row
//                        per group (+ FM sums); then lgkmcnt(0) + s_barrier in the MIDDLE of the chain
//   groups NQ/2 .. NQ-1 (second half):  A fragments of field f+1: LDS -> the other fragment register set (4 ds_read_b128
//                        per group); ids of field f+3 (group NQ/2), its rows requested from HBM (groups after)
//   every group q:       the weight quad it just consumed is refilled IN PLACE with the same quad of field f+2
// so that the chain never waits: fragments are read 32+ MFMAs before their first use, weights two fields (8,192
// cycles) ahead, table rows 1.5 fields ahead of their LDS write, and the one barrier per field falls between two
// MFMAs of waves that run in lockstep (same work, one wave per SIMD: no partner to lose arbitration to).
// Arithmetic: per output element the same k-ordered fma chain as the 32-sample kernel (bit-identical results).
// -----------------------------------------------------------------------------------------
template <int KD, int H1>
struct L1Fwd64 {
  static constexpr int TS = 64;
  static constexpr int LDW = KD + 4;
  static constexpr int CPR = KD / 4;                 // 16-byte chunks per row
  static constexpr int RPP = kBlock / CPR;           // rows staged per pass of the workgroup
  static constexpr int NLD = TS / RPP;               // rows (float4) per thread and field  (= KD / 16)
  static constexpr int KH = KD / 2;                  // reduction values per lane half and field
  static constexpr int NQ = KH / 4;                  // operand quads per lane and field      (= KD / 8)
  static constexpr int NC = H1 / 128;                // column tiles per wave
  static constexpr int CT = H1 / 32;
  static constexpr int FC = 32;                      // fields per id chunk
  static constexpr int NCH = TS * FC / kBlock;
  static_assert(KD % 16 == 0 && KD >= 32 && KD <= 128 && H1 % 128 == 0 && NLD * 2 == NQ, "shape");
  static size_t lds_bytes() { return static_cast<size_t>(2) * TS * LDW * 4 + static_cast<size_t>(2) * TS * FC * 4; }
};

template <int KD, int H1, bool kLin>
__global__ __launch_bounds__(kBlock, 1) void l1_fwd64_kernel(
    const float* __restrict__ table, const float* __restrict__ lin, int64_t V,
    const int32_t* __restrict__ idx, int64_t B, int F, const float* __restrict__ WpA,
    const float* __restrict__ bias, float* __restrict__ z1, float* __restrict__ pair,
    float* __restrict__ fsum, float* __restrict__ lin_out) {
  using C = L1Fwd64<KD, H1>;
  constexpr int TS = C::TS, LDW = C::LDW, CPR = C::CPR, RPP = C::RPP, NLD = C::NLD, KH = C::KH, NQ = C::NQ;
  constexpr int NC = C::NC, CT = C::CT, FC = C::FC, NCH = C::NCH, HQ = NQ / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);                          // [2][TS][LDW]: field f in buffer f & 1
  int32_t* idc = reinterpret_cast<int32_t*>(smem + 2 * TS * LDW * 4);    // [2][TS][FC]: chunk c in buffer c & 1

  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * TS;
  const int nb = (B - b0) < TS ? static_cast<int>(B - b0) : TS;

  auto chunk_load = [&](int c) {
    int32_t* dst = idc + (c & 1) * TS * FC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = tid + kBlock * i, r = e / FC, f = c * FC + (e % FC);
      dst[e] = (r < nb && f < F) ? idx[(b0 + r) * F + f] : -1;
    }
  };
  auto chunk_flush = [&](int c) {          // the chunk's slots hold the gathered linear weights by now
    const float* src = reinterpret_cast<const float*>(idc + (c & 1) * TS * FC);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = tid + kBlock * i, r = e / FC, f = c * FC + (e % FC);
      if (r < nb && f < F) lin_out[(b0 + r) * F + f] = src[e];
    }
  };
  auto id_slot = [&](int row, int f) -> int32_t* { return idc + ((f >> 5) & 1) * TS * FC + row * FC + (f & (FC - 1)); };

  const int srow = tid / CPR, c4 = (tid % CPR) * 4;
  const uint32_t Vu = static_cast<uint32_t>(V);
  float4 pre[2][NLD];            // rows of field g in set g & 1 (requested 1.5 fields before their LDS write)
  float prel[2][NLD];
  uint32_t pid[NLD];             // clamped ids of the rows being requested
  uint32_t pre_ok[2] = {0u, 0u};
  float4 S[NLD], Q[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) { S[u] = f4_zero(); Q[u] = f4_zero(); prel[0][u] = prel[1][u] = 0.f; pid[u] = 0u; }

  // ids: the raw LDS read and its use sit in DIFFERENT MFMA groups (PMC of the first version: the wave parked ~400 cycles
  // per field on the lgkmcnt of an id it had requested a few instructions earlier)
  auto ids_read = [&](int f, int u) { pid[u] = static_cast<uint32_t>(*id_slot(srow + u * RPP, f)); };
  auto row_load = [&](auto set_c, int u) {                   // HBM -> registers, branch-free (every lane of a row loads `lin`: one request)
    constexpr int set = decltype(set_c)::value;
    const bool ok = pid[u] < Vu;
    const uint32_t id = (ok && !(kAblate & 1)) ? pid[u] : 0u;
    pre_ok[set] = (pre_ok[set] & ~(1u << u)) | (ok ? (1u << u) : 0u);
    pre[set][u] = ld4(table + static_cast<uint64_t>(id) * KD + c4);
    if (kLin) prel[set][u] = lin[id];
  };
  auto row_write = [&](int f, auto set_c, int u) {           // registers -> LDS buffer f & 1 (+ FM sums, linear weight into the id slot)
    constexpr int set = decltype(set_c)::value;
    float* dst = rows + (f & 1) * TS * LDW;
    const bool ok = (pre_ok[set] >> u) & 1u;
    const float4 x = ok ? pre[set][u] : f4_zero();
    S[u] = f4_add(S[u], x);
    Q[u] = f4_fma(x, x, Q[u]);
    if (!(kAblate & 8)) st4(dst + (srow + u * RPP) * LDW + c4, x);
    if (kLin) *reinterpret_cast<float*>(id_slot(srow + u * RPP, f)) = ok ? prel[set][u] : 0.f;
  };

  f32x16 acc[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c) { acc[c][0] = acc_zero(); acc[c][1] = acc_zero(); }
  float4 a[2][2][NQ];            // [set][sample tile][quad]: A fragments of field f in set f & 1
  float4 bw[2][NC][NQ];          // weights of field f in set f & 1
  auto w_ptr = [&](int f, int c) {
    return WpA + ((static_cast<int64_t>(f) * CT + wid + 4 * c) * (KD / 8)) * 256 + lane * 4;
  };
  auto frag_read = [&](int f, auto set_c, int s, int q) {
    constexpr int set = decltype(set_c)::value;
    if ((kAblate & 8) && f > 1) return;
    a[set][s][q] = ld4(rows + (f & 1) * TS * LDW + (s * 32 + j) * LDW + h * KH + q * 4);
  };
  auto mfma_group = [&](auto set_c, int q) {
    constexpr int set = decltype(set_c)::value;
    if (kAblate & 16) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        asm volatile("" ::"v"(a[set][0][q].x), "v"(a[set][1][q].w), "v"(bw[set][c][q].x), "v"(bw[set][c][q].w));
      return;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][0][q].x, bw[set][c][q].x, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][1][q].x, bw[set][c][q].x, acc[c][1], 0, 0, 0);
      acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][0][q].y, bw[set][c][q].y, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][1][q].y, bw[set][c][q].y, acc[c][1], 0, 0, 0);
      acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][0][q].z, bw[set][c][q].z, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][1][q].z, bw[set][c][q].z, acc[c][1], 0, 0, 0);
      acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][0][q].w, bw[set][c][q].w, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][1][q].w, bw[set][c][q].w, acc[c][1], 0, 0, 0);
    }
  };

  // A group's other instructions are dealt into the gaps between its MFMAs (one wave per SIMD: an instruction issues
  // only in program order, so twenty VALU / DS instructions in ONE gap leave the pipe idle behind them): after every
  // MFMA up to four of {VALU, SALU, VMEM, DS}.
  auto spread = [&]() {
    if (kAblate & 32) return;
#pragma unroll
    for (int i = 0; i < 8 * NC; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x096, 4, 0);
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // One field.  P = f & 1 (compile time): a[P], bw[P] hold field f; pre[1-P] holds the rows of field f+1; pre[P] those
  // of field f+2.  kSteady: f + 3 < F, no conditionals in the body.
  auto field_step = [&](int f, auto p_c, auto steady_c) {
    constexpr int P = decltype(p_c)::value;
    constexpr bool kSteady = decltype(steady_c)::value;
    using SP = std::integral_constant<int, P>;
    using SN = std::integral_constant<int, 1 - P>;
    const bool has1 = kSteady || f + 1 < F, has2 = kSteady || f + 2 < F, has3 = kSteady || f + 3 < F;
    const float* wn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) wn[c] = w_ptr(has2 ? f + 2 : f, c);
#pragma unroll
    for (int q = 0; q < HQ; ++q) {                 // ---- first half of the chain
      mfma_group(SP{}, q);
      if (has2 && !(kAblate & 2)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) bw[P][c][q] = ld4(wn[c] + q * 256);
      }
      if (has1) {       // rows dealt to the groups before the last one: their LDS writes have a whole group to land before the barrier
#pragma unroll
        for (int u = 0; u < NLD; ++u)
          if ((u < HQ - 1 ? u : HQ - 2) == q || (HQ == 1 && q == 0)) row_write(f + 1, SN{}, u);
      }
      spread();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(kAblate & 4)) __syncthreads();           // rows of field f+1 visible (buffer (f+1)&1 was last read one field ago)
#pragma unroll
    for (int q = HQ; q < NQ; ++q) {                // ---- second half
      mfma_group(SP{}, q);
      if (has2 && !(kAblate & 2)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) bw[P][c][q] = ld4(wn[c] + q * 256);
      }
      if (has1) {
        constexpr int per = 2 * NQ / HQ;           // 4 fragment reads per group
#pragma unroll
        for (int t = 0; t < per; ++t) {
          const int e = (q - HQ) * per + t;
          frag_read(f + 1, SN{}, e / NQ, e % NQ);
        }
      }
      if (has3) {
        if (q == HQ) {
#pragma unroll
          for (int u = 0; u < NLD; ++u) ids_read(f + 3, u);
        } else {
#pragma unroll
          for (int u = 0; u < NLD; ++u)
            if (1 + u * (NLD - 1) / NLD == q - HQ) row_load(SN{}, u);
        }
      }
      spread();
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue ------------------------------------------------------------------------------
  chunk_load(0);
  if (F > FC) chunk_load(1);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      bw[0][c][q] = ld4(w_ptr(0, c) + q * 256);
      bw[1][c][q] = ld4(w_ptr(F > 1 ? 1 : 0, c) + q * 256);
    }
#pragma unroll
  for (int u = 0; u < NLD; ++u) { ids_read(0, u); row_load(Set0{}, u); }
  if (F > 1) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) { ids_read(1, u); row_load(Set1{}, u); }
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) row_write(0, Set0{}, u);
  if (F > 2) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) { ids_read(2, u); row_load(Set0{}, u); }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < NQ; ++q) frag_read(0, Set0{}, s, q);

  // ---- fields ----------------------------------------------------------------------------------
  // The id-chunk rotation sits between runs of steady steps (as in the 32-sample kernel): at f = 32 c every slot of
  // chunk c - 1 holds its linear weight (last written at step f - 2) and none of its ids is read any more (last read at
  // step f - 4); chunk c + 1 is first read at step f + 29.
  int f = 0, flushed = 0;
  const int f_steady = (F - 3) & ~1;      // steady steps run in pairs: f < f_steady  =>  f + 1 + 3 < F
  for (int c = 0; f < f_steady; ++c) {
    if (c > 0) {
      if (kLin) chunk_flush(c - 1);
      flushed = c;
      chunk_load(c + 1);
    }
    const int f_end = (c + 1) * FC < f_steady ? (c + 1) * FC : f_steady;
    for (; f < f_end; f += 2) {
      field_step(f, Set0{}, std::true_type{});
      field_step(f + 1, Set1{}, std::true_type{});
    }
  }
  for (; f < F; f += 2) {
    field_step(f, Set0{}, std::false_type{});
    if (f + 1 < F) field_step(f + 1, Set1{}, std::false_type{});
  }

  // ---- epilogue --------------------------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (wid + 4 * c) * 32 + j;
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int smp = s * 32 + acc_row(r, h);
        if (smp < nb) z1[(b0 + smp) * H1 + col] = acc[c][s][r] + bv;
      }
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int smp = srow + u * RPP;
    if (smp < nb) {
      float4 p;
      p.x = 0.5f * (S[u].x * S[u].x - Q[u].x);
      p.y = 0.5f * (S[u].y * S[u].y - Q[u].y);
      p.z = 0.5f * (S[u].z * S[u].z - Q[u].z);
      p.w = 0.5f * (S[u].w * S[u].w - Q[u].w);
      st4(pair + (b0 + smp) * KD + c4, p);
      if (fsum != nullptr) st4(fsum + (b0 + smp) * KD + c4, S[u]);
    }
  }
  if (kLin) {
    __syncthreads();      // the last fields' linear weights were written by other threads
    for (int c = flushed; c * FC < F; ++c) chunk_flush(c);
  }
}

// -----------------------------------------------------------------------------------------
// Weight gradient of the folded kernel: partial[ch][f*KD + i][n] = sum over the chunk's samples
// of x[b,f,i] * gz[b,n].  grid = F * n_chunks; workgroup (f, ch) walks its TS-sample slabs.
//   A[i][k=sample] = rows slab (LDS, [TS][KD], read by columns: conflict-free ds_read_b32)
//   B[k=sample][j] = gz slab   (LDS, [TS][H1])
//   wave w: output columns {32*(w + 4c)}, all KD/32 row tiles.
// -----------------------------------------------------------------------------------------
template <int KD, int H1, int kTS>
struct L1Wg {
  // fields per workgroup: the gz slab is the same for every field, so a workgroup that takes FG fields
  // stages it once and runs FG times the MFMAs per barrier interval (K <= 64: two fields fit the LDS
  // budget of two workgroups per CU)
  static constexpr int FG = (KD <= 64 && H1 <= 128) ? 2 : 1;
  static constexpr int CPR = KD / 4, RPP = kBlock / CPR, NLD = kTS / RPP;
  static constexpr int NI = KD / 32;                       // row tiles (embedding dims)
  static constexpr int NCW = (H1 / 32 + 3) / 4;            // column tiles per wave
  static constexpr int NGZ = kTS * H1 / 4 / kBlock;        // float4 of the gz slab per thread
  static_assert(KD % 32 == 0 && KD <= 128, "embed size");
  static_assert(H1 % 32 == 0 && (kTS * H1 / 4) % kBlock == 0, "first hidden width");
  static size_t lds_bytes() { return static_cast<size_t>(2) * kTS * (FG * KD + H1) * 4; }
};

template <int KD, int H1, int kTS>
__global__ __launch_bounds__(kBlock, ((kTS == 32 && H1 <= 128) ? 2 : 1)) void l1_wgrad_kernel(
    const float* __restrict__ table, int64_t V, const int32_t* __restrict__ idxT, int64_t B, int F,
    const float* __restrict__ gz, int n_chunks, float* __restrict__ partial) {
  using C = L1Wg<KD, H1, kTS>;
  constexpr int CPR = C::CPR, RPP = C::RPP, NLD = C::NLD, NI = C::NI, NCW = C::NCW, NGZ = C::NGZ, FG = C::FG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);                    // [2][FG][kTS][KD]
  float* gzt = rows + 2 * FG * kTS * KD;                           // [2][kTS][H1]
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int fg = blockIdx.x / n_chunks, ch = blockIdx.x % n_chunks;
  const int64_t slabs = ceil_div(B, kTS);
  const int64_t s_lo = slabs * ch / n_chunks, s_hi = slabs * (ch + 1) / n_chunks;
  const int n_sl = static_cast<int>(s_hi - s_lo);
  const int srow = tid / CPR, c4 = (tid % CPR) * 4;
  const uint32_t Vu = static_cast<uint32_t>(V);
  const int32_t* ids[FG];
  bool f_ok[FG];
#pragma unroll
  for (int q = 0; q < FG; ++q) {
    const int f = fg * FG + q;
    f_ok[q] = f < F;
    ids[q] = idxT + static_cast<int64_t>(f_ok[q] ? f : F - 1) * B;
  }

  // two register sets: the rows / gz of slab s+2 and s+3 are in flight while slab s is multiplied (random HBM rows
  // need more than one slab period under load)
  float4 pre[2][FG][NLD], pgz[2][NGZ];
  uint32_t pre_ok[2] = {0u, 0u};
  // The ids of a slab are requested ONE `stage_load` before its rows (calls go slab by slab): the rows' addresses then
  // depend on loads issued a whole MFMA phase ago.  Requested in the same call, the wait for them — vmcnt counts in order —
  // also drained the rows of the previous slab still in flight, once per slab (round 4).
  int32_t idn[FG][NLD];
  auto ids_load = [&](int64_t sl) {
    const int64_t b0 = sl * kTS;
#pragma unroll
    for (int q = 0; q < FG; ++q)
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int64_t b = b0 + srow + u * RPP;
        idn[q][u] = ids[q][b < B ? b : B - 1];
        if (b >= B) idn[q][u] = -1;
      }
  };
  ids_load(s_lo);
  auto stage_load = [&](int64_t sl, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const int64_t b0 = sl * kTS;
    pre_ok[set] = 0;
#pragma unroll
    for (int q = 0; q < FG; ++q)
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int32_t id = idn[q][u];
        const bool ok = f_ok[q] && static_cast<uint32_t>(id) < Vu;
        const uint32_t idc = ok ? static_cast<uint32_t>(id) : 0u;
        if (ok) pre_ok[set] |= 1u << (q * NLD + u);
        pre[set][q][u] = ld4(table + static_cast<uint64_t>(idc) * KD + c4);
      }
    ids_load(sl + 1);
#pragma unroll
    for (int u = 0; u < NGZ; ++u) {
      const int q = tid + u * kBlock;                  // float4 slot of the [kTS][H1] slab
      const int64_t b = b0 + q / (H1 / 4);
      const int64_t bc = b < B ? b : B - 1;            // clamped; zeroed at write time
      pgz[set][u] = ld4(gz + bc * H1 + (q % (H1 / 4)) * 4);
    }
  };
  auto stage_write = [&](int64_t sl, int buf, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const int64_t b0 = sl * kTS;
    float* dr = rows + buf * FG * kTS * KD;
    float* dg = gzt + buf * kTS * H1;
#pragma unroll
    for (int q = 0; q < FG; ++q)
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        st4(dr + (q * kTS + srow + u * RPP) * KD + c4,
            ((pre_ok[set] >> (q * NLD + u)) & 1u) ? pre[set][q][u] : f4_zero());
#pragma unroll
    for (int u = 0; u < NGZ; ++u) {
      const int q = tid + u * kBlock;
      const bool ok = b0 + q / (H1 / 4) < B;
      st4(dg + q * 4, ok ? pgz[set][u] : f4_zero());
    }
  };

  f32x16 acc[FG][NCW][NI];
#pragma unroll
  for (int q = 0; q < FG; ++q)
#pragma unroll
    for (int c = 0; c < NCW; ++c)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[q][c][i] = acc_zero();

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // slab s lives in LDS buffer s & 1 and came through register set s & 1
  if (n_sl > 0) {
    stage_load(s_lo, Set0{});
    stage_write(s_lo, 0, Set0{});
    if (n_sl > 1) stage_load(s_lo + 1, Set1{});
    if (n_sl > 2) stage_load(s_lo + 2, Set0{});
  }
  __syncthreads();
  // operands of 8 reduction steps are read in one burst, then 8 * FG * NCW * NI MFMAs run back to back
  auto compute = [&](int buf) {
    const float* xr = rows + buf * FG * kTS * KD;
    const float* gr = gzt + buf * kTS * H1;
#pragma unroll
    for (int t0 = 0; t0 < kTS / 2; t0 += 8) {
      float a[8][FG][NI], b[8][NCW];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = 2 * (t0 + u) + h;                    // sample of the slab
#pragma unroll
        for (int q = 0; q < FG; ++q)
#pragma unroll
          for (int i = 0; i < NI; ++i) a[u][q][i] = xr[(q * kTS + k) * KD + i * 32 + j];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
          const int ct = wid + 4 * c;
          b[u][c] = (ct * 32 < H1) ? gr[k * H1 + ct * 32 + j] : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < FG; ++q)
#pragma unroll
          for (int c = 0; c < NCW; ++c)
#pragma unroll
            for (int i = 0; i < NI; ++i)
              acc[q][c][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q][i], b[u][c], acc[q][c][i], 0, 0, 0);
    }
  };
  // steady state without conditionals (see l1_fwd_kernel): slab s+1 is written from the set that was requested two
  // slabs ago, slab s+3 is requested into the set just freed; then the last slabs
  int s = 0;
  for (; s + 4 < n_sl; s += 2) {
    compute(0);
    stage_write(s_lo + s + 1, 1, Set1{});
    stage_load(s_lo + s + 3, Set1{});
    __syncthreads();
    compute(1);
    stage_write(s_lo + s + 2, 0, Set0{});
    stage_load(s_lo + s + 4, Set0{});
    __syncthreads();
  }
  for (; s < n_sl; s += 2) {
    compute(0);
    if (s + 1 < n_sl) stage_write(s_lo + s + 1, 1, Set1{});
    if (s + 3 < n_sl) stage_load(s_lo + s + 3, Set1{});
    __syncthreads();
    if (s + 1 < n_sl) {
      compute(1);
      if (s + 2 < n_sl) stage_write(s_lo + s + 2, 0, Set0{});
      if (s + 4 < n_sl) stage_load(s_lo + s + 4, Set0{});
      __syncthreads();
    }
  }
#pragma unroll
  for (int q = 0; q < FG; ++q) {
    const int f = fg * FG + q;
    if (f >= F) continue;
    float* out = partial + (static_cast<int64_t>(ch) * F + f) * KD * H1;
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const int ct = wid + 4 * c;
      if (ct * 32 >= H1) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[(i * 32 + acc_row(r, h)) * H1 + ct * 32 + j] = acc[q][c][i][r];
    }
  }
}

// -----------------------------------------------------------------------------------------
// Row gradients in run order.  grid = ceil(B / TS); workgroup tile = TS samples.
//   A[i=sample][k=output column] = gz tile, resident in registers for the whole kernel
//   B[k][j=embedding dim]        = Wp_f^T from the packed WpB, one field (of this wave) ahead
//   the (TS/32 x KD/32) output tiles of a field are dealt to the 4 waves; with fewer than 4 tiles
//   (TS = 32, KD <= 64) the spare waves take every other field instead
//   epilogue per field: + gl[b] * wp[dim] * fsum[b][dim]  (the FM pairwise term's gradient:
//   d pair / d x = fsum - x; the "- x" half is applied per run by lr_fm_rows_adam_f32), then one
//   128-byte segment per (sample, dim tile) is stored at row slot_of_pos[b*F + f] of ge.
// -----------------------------------------------------------------------------------------
template <int KD, int H1, int kTS>
struct L1Dg {
  static constexpr int HH = H1 / 2;                          // reduction values per lane half
  static constexpr int NT = KD / 32;                         // dim tiles
  static constexpr int MS = kTS / 32;                        // sample tiles
  static constexpr int TILES = MS * NT;                      // 32x32 output tiles per field
  static constexpr int TW = TILES >= 4 ? TILES / 4 : 1;      // tiles per wave (same sample tile)
  static constexpr int NG = TILES >= 4 ? 1 : 4 / TILES;      // field groups (waves beyond the tiles take other fields)
  static_assert(KD % 32 == 0 && KD <= 128, "embed size");
  static_assert(H1 % 8 == 0 && H1 <= 256, "first hidden width");
  static_assert((kTS == 32 || kTS == 64) && TW <= NT && NT % TW == 0, "tile mapping");
  static size_t lds_bytes(int F) {
    return static_cast<size_t>(kTS) * KD * 4 + kTS * 4 + KD * 4 + static_cast<size_t>(kTS) * (H1 + 4) * 4 +
           static_cast<size_t>(kTS) * F * 4;
  }
};

template <int KD, int H1, int kTS>
__global__ __launch_bounds__(kBlock, (kTS == 32 ? 2 : 1)) void l1_dgrad_kernel(
    const float* __restrict__ gz, const float* __restrict__ WpB, int F, int64_t B,
    const float* __restrict__ gl, const float* __restrict__ wp, const float* __restrict__ fsum,
    const int32_t* __restrict__ slotT, float* __restrict__ ge) {
  using C = L1Dg<KD, H1, kTS>;
  constexpr int HH = C::HH, NT = C::NT, TW = C::TW, NG = C::NG, TILES = C::TILES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* fs = reinterpret_cast<float*>(smem);                 // [kTS][KD]
  float* glt = fs + kTS * KD;                                 // [kTS]
  float* wpt = glt + kTS;                                     // [KD]
  float* gzt = wpt + KD;                                      // [kTS][H1 + 4]: the gz tile (A operand)
  int32_t* slots = reinterpret_cast<int32_t*>(gzt + kTS * (H1 + 4));           // [F][kTS]
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kTS;
  const int nb = (B - b0) < kTS ? static_cast<int>(B - b0) : kTS;

  // slot tile, field-major [F][TS]: slotT is [F, B], so every field contributes one contiguous piece
  for (int q = tid; q < kTS * F; q += kBlock) {
    const int ff = q / kTS, r = q % kTS;
    slots[q] = (r < nb) ? slotT[static_cast<int64_t>(ff) * B + b0 + r] : -1;
  }
  for (int q = tid; q < kTS * KD / 4; q += kBlock) {
    const int r = q / (KD / 4);
    st4(fs + q * 4, (r < nb && fsum != nullptr) ? ld4(fsum + b0 * KD + q * 4) : f4_zero());
  }
  if (tid < kTS) glt[tid] = (tid < nb && gl != nullptr) ? gl[b0 + tid] : 0.f;
  if (tid < KD) wpt[tid] = wp != nullptr ? wp[tid] : 0.f;

  // this wave: tiles t0 .. t0+TW-1 (one sample tile mi, TW dim tiles) of the fields fg, fg+NG, ...
  const int t0 = (NG == 1) ? wid * TW : (wid % TILES);
  const int fg = (NG == 1) ? 0 : (wid / TILES);
  const int mi = t0 / NT, ni0 = t0 % NT;
  // gz tile -> LDS (rows padded by 16 B: conflict-free ds_read_b128 of the A fragments)
  for (int q = tid; q < kTS * H1 / 4; q += kBlock) {
    const int r = q / (H1 / 4), c4 = (q % (H1 / 4)) * 4;
    st4(gzt + r * (H1 + 4) + c4, r < nb ? ld4(gz + (b0 + r) * H1 + c4) : f4_zero());
  }
  const float* arow = gzt + (mi * 32 + j) * (H1 + 4) + h * HH;   // this lane's A values: sample mi*32+j, columns [h*HH, (h+1)*HH)
  __syncthreads();

  // FM term of this lane's 16 accumulator rows, constant over the fields
  float fm[TW][16];
  int srow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    srow[r] = mi * 32 + acc_row(r, h);
#pragma unroll
    for (int w = 0; w < TW; ++w) {
      const int dim = (ni0 + w) * 32 + j;
      fm[w][r] = glt[srow[r]] * wpt[dim] * fs[srow[r] * KD + dim];
    }
  }

  // Weights: ONE register copy, refilled in place.  As soon as MFMA group s4 of field f has been issued its
  // operand quad is overwritten by the load of the same quad of the wave's NEXT field, which then has a whole
  // chain (64 MFMAs) to arrive — a second copy (128 more VGPRs next to the 64 of the gz fragment) would not
  // fit two waves per SIMD.
  float4 bw[TW][HH / 4];
  auto w_ptr = [&](int f, int w) {
    return WpB + ((static_cast<int64_t>(f) * NT + ni0 + w) * (H1 / 8)) * 256 + lane * 4;
  };
  // Row of ge for every accumulator register of the wave's next field: read from LDS in one burst in
  // front of the chain (dropped positions go to the spare row B*F at the end of ge, so the 16 stores of
  // the epilogue are unconditional: no branch, no LDS wait between them).
  const uint32_t spare = static_cast<uint32_t>(B * F);
  uint32_t dst[16];
  auto load_slots = [&](int f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int32_t sl = slots[f * kTS + srow[r]];
      dst[r] = ((sl >= 0 ? static_cast<uint32_t>(sl) : spare) * KD + j) * 4u;     // BYTE offset: uniform base + 32-bit lane offset addressing
    }
  };
  auto chain = [&](auto has_next, const float* __restrict__ wnext0, const float* __restrict__ wnext1,
                   f32x16 (&acc)[TW]) {
#pragma unroll
    for (int w = 0; w < TW; ++w) acc[w] = acc_zero();
    float4 a = ld4(arow);
#pragma unroll
    for (int s4 = 0; s4 < HH / 4; ++s4) {
      const float4 an = ld4(arow + ((s4 + 1 < HH / 4) ? (s4 + 1) * 4 : 0));     // one group ahead of the MFMAs
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[w][s4].x, acc[w], 0, 0, 0);
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[w][s4].y, acc[w], 0, 0, 0);
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[w][s4].z, acc[w], 0, 0, 0);
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[w][s4].w, acc[w], 0, 0, 0);
        if constexpr (decltype(has_next)::value) bw[w][s4] = ld4(((w == 0) ? wnext0 : wnext1) + s4 * 256);
      }
      // keep "MFMAs, refill" in program order: clustered after the chain (the scheduler's preference) the
      // refills sit behind the epilogue's stores in the memory queue and the next chain waits for both
      __builtin_amdgcn_sched_barrier(0);
      a = an;
    }
  };
  auto store = [&](const f32x16 (&acc)[TW]) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int w = 0; w < TW; ++w)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(ge) + dst[r] + (ni0 + w) * 128) = acc[w][r] + fm[w][r];
  };
  static_assert(TW <= 2, "two weight streams per wave at most");
  if (fg < F) {
#pragma unroll
    for (int w = 0; w < TW; ++w)
#pragma unroll
      for (int s4 = 0; s4 < HH / 4; ++s4) bw[w][s4] = ld4(w_ptr(fg, w) + s4 * 256);
    load_slots(fg);
  }
  int f = fg;
  for (; f + NG < F; f += NG) {               // steady state: a successor field exists, no conditionals
    f32x16 acc[TW];
    chain(std::true_type{}, w_ptr(f + NG, 0), w_ptr(f + NG, TW - 1), acc);
    store(acc);
    load_slots(f + NG);
  }
  if (f < F) {                                // the wave's last field
    f32x16 acc[TW];
    chain(std::false_type{}, nullptr, nullptr, acc);
    store(acc);
  }
}

// -----------------------------------------------------------------------------------------
// Row gradients in run order, 64-sample tiles, ONE wave per SIMD (round 4; see l1_fwd64_kernel for the why).
//   wave w: dim tile w % NT of the fields  w / NT, w / NT + NG, ...  (NG = 4 / NT field groups), BOTH 32-sample tiles:
//   a weight quad feeds two MFMAs (half the L2 -> CU weight traffic of the 32-sample kernel).
//   A = the two gz fragments, resident in registers for the whole kernel (2 x H1/2 values per lane): the chain touches no LDS.
//   B = Wp_f^T quads, ONE register copy refilled in place: quad q is overwritten with the wave's next field as soon as
//   group q has been issued (a whole chain = 8,192 cycles to arrive).
//   Two accumulator sets: while field f is accumulated into one, the other (the wave's previous field) is stored,
//   two 128-byte row segments per MFMA group, and the store offsets of field f (slot map in LDS) replace the two
//   just used.  No barrier after the prologue: the four waves share nothing but read-only LDS.
// Arithmetic per element: the same k-ordered chain + FM term as l1_dgrad_kernel (bit-identical).
// -----------------------------------------------------------------------------------------
template <int KD, int H1>
struct L1Dg64 {
  static constexpr int TS = 64;
  static constexpr int HH = H1 / 2;
  static constexpr int NQ = HH / 4;                          // operand quads per lane
  static constexpr int NT = KD / 32;                         // dim tiles = waves per field
  static constexpr int NG = 4 / NT;                          // field groups
  static_assert((NT == 1 || NT == 2 || NT == 4) && H1 % 32 == 0 && H1 <= 256 && NQ % 16 == 0, "shape");
  static size_t lds_bytes(int F) {
    return static_cast<size_t>(TS) * KD * 4 + TS * 4 + KD * 4 + static_cast<size_t>(TS) * F * 4;
  }
};

template <int KD, int H1>
__global__ __launch_bounds__(kBlock, 1) void l1_dgrad64_kernel(
    const float* __restrict__ gz, const float* __restrict__ WpB, int F, int64_t B,
    const float* __restrict__ gl, const float* __restrict__ wp, const float* __restrict__ fsum,
    const int32_t* __restrict__ slotT, float* __restrict__ ge) {
  using C = L1Dg64<KD, H1>;
  constexpr int TS = C::TS, HH = C::HH, NQ = C::NQ, NT = C::NT, NG = C::NG;
  constexpr int SPG = 32 / NQ;                                // stores per MFMA group (32 row segments per field and lane)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* fs = reinterpret_cast<float*>(smem);                 // [TS][KD]
  float* glt = fs + TS * KD;                                  // [TS]
  float* wpt = glt + TS;                                      // [KD]
  int32_t* slots = reinterpret_cast<int32_t*>(wpt + KD);      // [F][TS]
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * TS;
  const int nb = (B - b0) < TS ? static_cast<int>(B - b0) : TS;
  const int ni = wid % NT, fg = wid / NT;

  // slot map of the tile as BYTE offsets of the ge rows (dropped positions -> the spare row B*F), computed once here:
  // every instruction a wave issues between two f32 MFMAs costs pipe time (profiles/r04_mfma_issue_probe.txt), so the
  // chain's store addresses are one LDS read + one add each
  const uint32_t spare = static_cast<uint32_t>(B * F);
  for (int q = tid; q < TS * F; q += kBlock) {
    const int ff = q / TS, r = q % TS;
    const int32_t sl = (r < nb) ? slotT[static_cast<int64_t>(ff) * B + b0 + r] : -1;
    slots[q] = static_cast<int32_t>((sl >= 0 ? static_cast<uint32_t>(sl) : spare) * (KD * 4u));
  }
  for (int q = tid; q < TS * KD / 4; q += kBlock) {
    const int r = q / (KD / 4);
    st4(fs + q * 4, (r < nb && fsum != nullptr) ? ld4(fsum + b0 * KD + q * 4) : f4_zero());
  }
  if (tid < TS) glt[tid] = (tid < nb && gl != nullptr) ? gl[b0 + tid] : 0.f;
  if (tid < KD) wpt[tid] = wp != nullptr ? wp[tid] : 0.f;

  // gz fragments straight from global memory (once per kernel): lane (j, h) holds samples mi*32 + j, columns [h*HH, (h+1)*HH)
  float4 ag[2][NQ];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = mi * 32 + j;
    const int64_t rc = b0 + (r < nb ? r : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 v = ld4(gz + rc * H1 + h * HH + q * 4);
      ag[mi][q] = r < nb ? v : f4_zero();
    }
  }
  __syncthreads();

  // accumulator register rr = mi * 16 + r of this lane is sample mi*32 + acc_row(r, h), dim ni*32 + j
  float fm[32];
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int sm = (rr >> 4) * 32 + acc_row(rr & 15, h);
    const int dim = ni * 32 + j;
    fm[rr] = glt[sm] * wpt[dim] * fs[sm * KD + dim];
  }
  uint32_t dst[32];
  const uint32_t lane_off = j * 4u + ni * 128u;
  auto slot_off = [&](int f, int rr) {
    const int sm = (rr >> 4) * 32 + acc_row(rr & 15, h);
    return static_cast<uint32_t>(slots[f * TS + sm]) + lane_off;                       // BYTE offset into ge
  };
  float4 bw[NQ];
  auto w_ptr = [&](int f) { return WpB + ((static_cast<int64_t>(f) * NT + ni) * (H1 / 8)) * 256 + lane * 4; };
  f32x16 acc[2][2];

  auto spread = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x096, 3, 0);
    }
  };
  // one field of this wave: accumulate into set P, store set 1-P (the previous field) on the way
  auto chain = [&](int f, auto p_c, bool has_prev, bool has_next) {
    constexpr int P = decltype(p_c)::value;
    acc[P][0] = acc_zero();
    acc[P][1] = acc_zero();
    const float* wn = w_ptr(has_next ? f + NG : f);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[0][q].x, bw[q].x, acc[P][0], 0, 0, 0);
      acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[1][q].x, bw[q].x, acc[P][1], 0, 0, 0);
      acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[0][q].y, bw[q].y, acc[P][0], 0, 0, 0);
      acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[1][q].y, bw[q].y, acc[P][1], 0, 0, 0);
      acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[0][q].z, bw[q].z, acc[P][0], 0, 0, 0);
      acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[1][q].z, bw[q].z, acc[P][1], 0, 0, 0);
      acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[0][q].w, bw[q].w, acc[P][0], 0, 0, 0);
      acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[1][q].w, bw[q].w, acc[P][1], 0, 0, 0);
      if (has_next) bw[q] = ld4(wn + q * 256);
#pragma unroll
      for (int t = 0; t < SPG; ++t) {
        const int rr = q * SPG + t;
        if (has_prev)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(ge) + dst[rr]) = acc[1 - P][rr >> 4][rr & 15] + fm[rr];
        dst[rr] = slot_off(f, rr);
      }
      spread();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  if (fg < F) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) bw[q] = ld4(w_ptr(fg) + q * 256);
    int f = fg;
    bool prev = false;
    // steady state in pairs (both accumulator sets named at compile time): a successor field exists
    for (; f + 2 * NG < F; f += 2 * NG) {
      chain(f, Set0{}, prev, true);
      chain(f + NG, Set1{}, true, true);
      prev = true;
    }
    int last = 0;
    if (f < F) {
      chain(f, Set0{}, prev, f + NG < F);
      prev = true;
      last = 0;
      if (f + NG < F) {
        chain(f + NG, Set1{}, true, false);
        last = 1;
      }
    }
    // the wave's last field
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
      const float v = (last == 0 ? acc[0][rr >> 4][rr & 15] : acc[1][rr >> 4][rr & 15]) + fm[rr];
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ge) + dst[rr]) = v;
    }
  }
}

static inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
constexpr size_t kMaxLds = 160 * 1024;

template <typename Kern>
static int set_lds(Kern kern, size_t bytes) {
  if (bytes > kMaxLds) return LR_ESHAPE;
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return LR_OK;
}

}  // namespace lr

using namespace lr;

// (K, H1, TS) instantiations; TS = 32 needs H1 >= 128 in the forward kernel (4 column tiles for 4 waves)
#define LR_L1_SHAPES(X) \
  X(64, 128, 32) X(64, 128, 64) X(32, 128, 32) X(32, 128, 64) X(128, 128, 32) X(128, 128, 64) \
  X(64, 256, 32) X(64, 256, 64) X(64, 64, 64) X(32, 64, 64)

// tile size: 32 where compiled (more resident workgroups per CU), else the one-workgroup-per-CU 64 variant
static int l1_tile(int K, int H1) {
  (void)K;
  return H1 >= 128 ? 32 : 64;
}

// (K, H1) instantiations of the round-4 kernels (64-sample tiles, one wave per SIMD, two accumulators per column tile)
#define LR_L1_WIDE_SHAPES(X) X(64, 128) X(32, 128) X(64, 256)

// Which forward / backward-data kernel a batch gets: the wide kernels need about one 64-sample workgroup per CU to
// fill the chip (B >= 0.75 * 256 * 64 = 12,288); smaller batches keep the 32-sample kernels (two per CU from
// B = 16,384 / 2).  `lr_deepfm_l1_tile_override` pins the choice (tests run both families on the same inputs).
static int g_l1_tile_override = 0;
extern "C" void lr_deepfm_l1_tile_override(int tile) { g_l1_tile_override = (tile == 32 || tile == 64) ? tile : 0; }

static bool l1_wide_compiled(int K, int H1) {
#define X(KD, HD) if (K == KD && H1 == HD) return true;
  LR_L1_WIDE_SHAPES(X)
#undef X
  return false;
}
static bool l1_use_wide(int K, int H1, int64_t B) {
  if (!l1_wide_compiled(K, H1) || g_l1_tile_override == 32) return false;
  if (g_l1_tile_override == 64) return true;
  return ceil_div(B, 64) >= (3 * kNumCU) / 4;
}

extern "C" int lr_deepfm_l1_supported(int K, int H1) {
#define X(KD, HD, TS) if (K == KD && H1 == HD) return 1;
  LR_L1_SHAPES(X)
#undef X
  return 0;
}

extern "C" int lr_deepfm_l1_pack_f32(const float* Wp, int F, int K, int H1, float* WpA, float* WpB,
                                     lr_stream_t stream) {
  LR_CHECK_ARG(F >= 1 && Wp && WpA && WpB && al16(Wp) && al16(WpA) && al16(WpB));
  if (!lr_deepfm_l1_supported(K, H1)) return LR_ESHAPE;
  const int64_t total = static_cast<int64_t>(F) * K * H1 / 4;
  hipLaunchKernelGGL(l1_pack_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     Wp, static_cast<const float*>(nullptr), F, K, H1, WpA, WpB, static_cast<const float*>(nullptr), 0,
                     static_cast<float*>(nullptr));
  return launch_status();
}

extern "C" int lr_deepfm_l1_pack_scaled_f32(const float* W, const float* scale, int F, int K, int H1,
                                            float* WpA, float* WpB, const float* red_partial, int red_nblk,
                                            float* red_out, lr_stream_t stream) {
  LR_CHECK_ARG(F >= 1 && W && scale && WpA && WpB);
  LR_CHECK_ARG(al16(W) && al16(WpA) && al16(WpB));
  LR_CHECK_ARG((red_partial == nullptr) == (red_out == nullptr) && (red_partial == nullptr || red_nblk >= 1));
  if (!lr_deepfm_l1_supported(K, H1)) return LR_ESHAPE;
  const int64_t total = static_cast<int64_t>(F) * K * H1 / 4;
  const int n_red = red_partial != nullptr ? (H1 + 15) / 16 : 0;
  hipLaunchKernelGGL(l1_pack_kernel, dim3(grid_for(total, kBlock) + n_red), dim3(kBlock), 0, as_stream(stream),
                     W, scale, F, K, H1, WpA, WpB, red_partial, red_nblk, red_out);
  return launch_status();
}

extern "C" int lr_idx_transpose_i32(const int32_t* idx, int64_t B, int F, int32_t* idxT,
                                    lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(idx && idxT);
  const int64_t tiles = ceil_div(B, 32) * ceil_div(F, 32);
  hipLaunchKernelGGL(idx_transpose_kernel, dim3(grid_for(tiles, 1)), dim3(kBlock), 0, as_stream(stream),
                     idx, B, F, idxT);
  return launch_status();
}

extern "C" int lr_deepfm_l1_fwd_f32(const float* table, const float* lin, int64_t V, int K,
                                    const int32_t* idx, int64_t B, int F, const float* WpA,
                                    const float* bias, int H1, float* z1, float* pair, float* fsum,
                                    float* lin_out, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 1 && B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && WpA && z1 && pair);
  LR_CHECK_ARG(al16(table) && al16(WpA) && al16(pair) && (!fsum || al16(fsum)));
  LR_CHECK_ARG((lin == nullptr) == (lin_out == nullptr));
  if (l1_use_wide(K, H1, B)) {
#define X(KD, HD)                                                                                    \
    if (K == KD && H1 == HD) {                                                                       \
      const size_t lds = L1Fwd64<KD, HD>::lds_bytes();                                               \
      static bool lds_set = false;                                                                   \
      if (!lds_set) {                                                                                \
        int rc = set_lds(l1_fwd64_kernel<KD, HD, true>, lds);                                        \
        if (rc == LR_OK) rc = set_lds(l1_fwd64_kernel<KD, HD, false>, lds);                          \
        if (rc != LR_OK) return rc;                                                                  \
        lds_set = true;                                                                              \
      }                                                                                              \
      const dim3 grid(static_cast<int>(ceil_div(B, 64)));                                            \
      if (lin != nullptr)                                                                            \
        hipLaunchKernelGGL((l1_fwd64_kernel<KD, HD, true>), grid, dim3(kBlock), lds, as_stream(stream), table, lin, V, \
                           idx, B, F, WpA, bias, z1, pair, fsum, lin_out);                           \
      else                                                                                           \
        hipLaunchKernelGGL((l1_fwd64_kernel<KD, HD, false>), grid, dim3(kBlock), lds, as_stream(stream), table, lin, V, \
                           idx, B, F, WpA, bias, z1, pair, fsum, lin_out);                           \
      return launch_status();                                                                        \
    }
    LR_L1_WIDE_SHAPES(X)
#undef X
  }
  const int ts = l1_tile(K, H1);
#define X(KD, HD, TS)                                                                               \
  if (K == KD && H1 == HD && ts == TS) {                                                            \
    const size_t lds = L1Fwd<KD, HD, TS>::lds_bytes(F);                                             \
    static bool lds_set = false;   /* once per instantiation (first eager call) */                \
    if (!lds_set) {                                                                                 \
      int rc = set_lds(l1_fwd_kernel<KD, HD, TS>, lds);                                               \
      if (rc != LR_OK) return rc;                                                                   \
      lds_set = true;                                                                               \
    }                                                                                               \
    hipLaunchKernelGGL((l1_fwd_kernel<KD, HD, TS>), dim3(static_cast<int>(ceil_div(B, TS))),        \
                       dim3(kBlock), lds, as_stream(stream), table, lin, V, idx, B, F, WpA, bias,   \
                       z1, pair, fsum, lin_out);                                                    \
    return launch_status();                                                                         \
  }
  LR_L1_SHAPES(X)
#undef X
  return LR_ESHAPE;
}

extern "C" int lr_deepfm_l1_wgrad_chunks(int64_t B, int F) {
  // F * n_chunks workgroups of equal length spread over 256 CUs: pick the split whose busiest CU
  // carries the least work, preferring fewer partial slabs.  The price of a slab (its share of the fixed-order
  // sum in lr_deepfm_l1_fold_bwd_f32) grows with the F * K rows it holds: half a slab time at the 202 fields of
  // BASELINE cfg 2, next to nothing for the three planes of a DIN block — there up to 64 chunks are worth it
  // (cfg 3, B = 8,192: 48 workgroups of 8 slabs -> 192 of 2).
  if (B < 1 || F < 1) return 1;
  const int64_t slabs = ceil_div(B, 64);
  const int cap = F > 16 ? 16 : 64;
  const double slab_price = 0.5 * (F < 202 ? static_cast<double>(F) / 202.0 : 1.0);
  int best = 1;
  double best_cost = 1e30;
  for (int n = 1; n <= cap && n <= slabs; ++n) {
    const double rounds = static_cast<double>(ceil_div(static_cast<int64_t>(F) * n, kNumCU));
    const double cost = rounds * static_cast<double>(ceil_div(slabs, n)) + slab_price * n + (F > 16 ? 0.0 : 0.05 * n);
    if (cost < best_cost) { best_cost = cost; best = n; }
  }
  return best;
}

extern "C" int lr_deepfm_l1_wgrad_f32(const float* table, int64_t V, int K, const int32_t* idxT,
                                      int64_t B, int F, const float* gz, int H1, int n_chunks,
                                      float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 1 && B >= 1 && F >= 1 && n_chunks >= 1);
  LR_CHECK_ARG(table && idxT && gz && partial && al16(table) && al16(gz));
  const int ts = l1_tile(K, 128);      // slab size: independent of H1
#define X(KD, HD, TS)                                                                               \
  if (K == KD && H1 == HD && (ts == TS || HD < 128)) {                                              \
    const size_t lds = L1Wg<KD, HD, TS>::lds_bytes();                                               \
    static bool lds_set = false;   /* once per instantiation (first eager call) */                \
    if (!lds_set) {                                                                                 \
      int rc = set_lds(l1_wgrad_kernel<KD, HD, TS>, lds);                                               \
      if (rc != LR_OK) return rc;                                                                   \
      lds_set = true;                                                                               \
    }                                                                                               \
    const int groups = (F + L1Wg<KD, HD, TS>::FG - 1) / L1Wg<KD, HD, TS>::FG;                       \
    hipLaunchKernelGGL((l1_wgrad_kernel<KD, HD, TS>), dim3(groups * n_chunks), dim3(kBlock), lds,   \
                       as_stream(stream), table, V, idxT, B, F, gz, n_chunks, partial); \
    return launch_status();                                                                         \
  }
  LR_L1_SHAPES(X)
#undef X
  return LR_ESHAPE;
}

extern "C" int lr_deepfm_l1_dgrad_f32(const float* gz, int H1, const float* WpB, int K, int F,
                                      int64_t B, const float* gl, const float* wp, const float* fsum,
                                      const int32_t* slotT, float* ge, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(gz && WpB && slotT && ge && al16(gz) && al16(WpB) && al16(ge));
  LR_CHECK_ARG((gl == nullptr) == (wp == nullptr) && (gl == nullptr) == (fsum == nullptr));
  LR_CHECK_ARG(!fsum || al16(fsum));
  if ((B * F + 1) * K * 4 >= (int64_t(1) << 32)) return LR_ESHAPE;  // 32-bit byte offsets into ge
  if (l1_use_wide(K, H1, B) && H1 == 128) {
#define X(KD, HD)                                                                                    \
    if (K == KD && H1 == HD) {                                                                       \
      if constexpr (HD == 128) {                                                                     \
        const size_t lds = L1Dg64<KD, 128>::lds_bytes(F);                                            \
        if (lds <= kMaxLds) {                                                                        \
          static bool lds_set = false;                                                               \
          if (!lds_set) {                                                                            \
            int rc = set_lds(l1_dgrad64_kernel<KD, 128>, kMaxLds);                                   \
            if (rc != LR_OK) return rc;                                                              \
            lds_set = true;                                                                          \
          }                                                                                          \
          hipLaunchKernelGGL((l1_dgrad64_kernel<KD, 128>), dim3(static_cast<int>(ceil_div(B, 64))),  \
                             dim3(kBlock), lds, as_stream(stream), gz, WpB, F, B, gl, wp, fsum, slotT, ge); \
          return launch_status();                                                                    \
        }                                                                                            \
      }                                                                                              \
    }
    LR_L1_WIDE_SHAPES(X)
#undef X
  }
  const int ts = H1 > 128 ? 64 : l1_tile(K, 128);   // H1 = 256: the gz fragment alone is 128 VGPRs -> one wave per SIMD
#define X(KD, HD, TS)                                                                               \
  if (K == KD && H1 == HD && (ts == TS || HD < 128)) {                                              \
    const size_t lds = L1Dg<KD, HD, TS>::lds_bytes(F);                                              \
    if (lds > kMaxLds) return LR_ESHAPE;                                                            \
    static bool lds_set = false;   /* once per instantiation (first eager call); the need grows with F: allow the CU's LDS */ \
    if (!lds_set) {                                                                                 \
      int rc = set_lds(l1_dgrad_kernel<KD, HD, TS>, kMaxLds);                                           \
      if (rc != LR_OK) return rc;                                                                   \
      lds_set = true;                                                                               \
    }                                                                                               \
    hipLaunchKernelGGL((l1_dgrad_kernel<KD, HD, TS>), dim3(static_cast<int>(ceil_div(B, TS))),      \
                       dim3(kBlock), lds, as_stream(stream), gz, WpB, F, B, gl, wp, fsum, slotT, ge); \
    return launch_status();                                                                         \
  }
  LR_L1_SHAPES(X)
#undef X
  return LR_ESHAPE;
}
