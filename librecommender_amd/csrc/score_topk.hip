// Full-catalog scoring with fused top-k:  scores = users @ items^T on the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TF dense peak on MI355X), with the
// B x N score matrix never leaving registers.
//
// Decomposition
//   grid      : (user tile of WU*32 users) x (item range g of G); block b runs on XCD b%8, and
//               the mapping puts the user tiles of ONE item range on ONE XCD back to back so
//               the range is fetched from HBM once and re-read from that XCD's L2.
//   workgroup : 4 waves = WU user slabs x WI item sub-tiles (WU*WI = 4).  Item rows are staged
//               through LDS in stages of 32*WI rows (register-staged, rows padded by 16 B so
//               ds_read_b128 is conflict-free) held in a ring of NB buffers.  There is NO
//               workgroup barrier in the stage loop: a wave that has written its share of a stage
//               bumps that buffer's "full" counter in LDS, a wave that has finished reading bumps
//               "done"; consumers / producers spin on those counters.  With the ring two stages
//               deep the waits are almost never taken, so the four waves (and the second
//               workgroup of the CU) drift apart and their staging / epilogue phases hide behind
//               each other's MFMA phases.
//   wave      : A = 32 items (from LDS), B = its 32 users (held in VGPRs for the whole kernel),
//               so C[item,user] puts ONE user per lane column: the running top-k threshold is
//               a single register per lane.  The reduction index is permuted (lane half h owns
//               dims [h*D/2,(h+1)*D/2)) so both operands are read with 16-byte accesses.
//   top-k     : a score survives if (score, id) > the user's threshold; survivors are appended
//               to a per-(wave,user) candidate list in global scratch (LDS counter).  When a
//               list is nearly full the owning wave selects its exact k-th key by bisection on
//               an order-preserving 64-bit key (score bits << 32 | ~id), compacts in place and
//               raises the threshold.  After the last tile each list holds <= k entries; a
//               second kernel merges the lists of a user with a bitonic sort in LDS.
//   SB        : the split-bf16 form (lr_score_topk_sb_f32).  Same decomposition; the item stage is split into three bf16 planes
//               ONCE per workgroup when it is written to LDS (every f32 -> x1 + x2 + x3 exactly, split_bf16.hpp), the users'
//               planes are split once into registers, and a 32 x 32 x 16 block of the contraction is six
//               v_mfma_f32_32x32x16_bf16 (192 matrix-pipe cycles against 512 for eight v_mfma_f32_32x32x2_f32).  Scores agree
//               with the f32 chain to f32 rounding (not bit for bit): tests pin both against fp64.
//   Order     : (score desc, id asc) — total order, so results are run-to-run identical and
//               independent of the tiling.  NaN scores are dropped.
#include <stdlib.h>

#include "common.hpp"
#include "split_bf16.hpp"

namespace lr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef LR_TOPK_WIN_ROWS
#define LR_TOPK_WIN_ROWS 256
#endif
#ifndef LR_TOPK_WIN_SLACK
#define LR_TOPK_WIN_SLACK 1
#endif
constexpr int kPD = 2;      // stages of item prefetch in flight
// The one-term filter (AR 2) has 1/6 of the split form's MFMA work per stage: with 32-row stages its loop is bound by the latency
// of the item loads (a stage's loads are issued one iteration before they are written to LDS).  It therefore takes LR_TK_RS
// 32-row sub-tiles per wave and stage (more bytes in flight per workgroup, fewer stage hand-overs per item row).
#ifndef LR_TK_RS
#define LR_TK_RS 2          // (measured at 1,024 users x 100 M x 128, k' = 256: 46.5 ms with 1, 44.6 with 2, 86.9 with 4 — one workgroup per CU)
#endif
#ifndef LR_TK_NB
#define LR_TK_NB 3
#endif
#ifndef LR_TK_RS_NARROW
#define LR_TK_RS_NARROW 0
#endif
#ifndef LR_TK_RS8
#define LR_TK_RS8 4         // the eight-wave form: 128-row stages (four sub-tiles per wave)
#endif
#ifndef LR_TK_NB8
#define LR_TK_NB8 3
#endif
#ifndef LR_TK_W8
#define LR_TK_W8 0          // 1: batches above 128 users run the eight-wave form of the filter
#endif
#ifndef LR_TK_OCC
#define LR_TK_OCC 2         // (its 64-row stages: two workgroups per CU by LDS at a reduction width of 128)
#endif
template <int DT, int WU, int AR, int TU = 1>
struct TkShape {
  // 32-row item sub-tiles per wave and stage.  The filter: 32 LR_TK_RS rows per stage.  The exact forms at narrow reduction
  // widths (a 32-row stage of 16 floats is 2 KB: the hand-over of a stage costs more than its MFMAs): 128 rows at DT = 16, 64 at 32
  // WU == 8 (the filter only): workgroups of EIGHT waves, one user tile each, sharing a staged stage — the registers a wave saves on
  // user fragments (32 instead of 64) hold a second accumulator set: the threshold tests of one sub-tile run under the MFMAs
  // of the next
  static constexpr int NW = WU == 8 ? 8 : 4;                                   // waves per workgroup
  static constexpr int RSW = AR == 2 ? (WU == 8 ? LR_TK_RS8 : LR_TK_RS) : (LR_TK_RS_NARROW && DT <= 32) ? 64 / DT : 1;
  static constexpr int RS = RSW * WU >= NW ? RSW * WU / NW : 1;
  static constexpr int NB = AR == 2 ? (WU == 8 ? LR_TK_NB8 : LR_TK_NB) : (DT <= 128 ? 3 : 2);          // stage buffers in the LDS ring
  static constexpr int OCC = AR == 2 ? (WU == 8 ? 1 : LR_TK_OCC) : AR == 1 ? 2 : DT <= 128 ? 3 : 1;   // workgroups per CU the registers are cut for
  static constexpr int TI = 32 * (NW / WU) * RS;                               // item rows per stage
};
constexpr int kRing = 32;   // per-wave candidate ring entries (LDS)

struct TopkPlan {
  int DT;      // compiled reduction width (16..256), >= D
  int WU, WI;  // waves per workgroup along users / items
  int TU;      // 32-user tiles per wave (1; the one-term filter: 2 when the batch fills them)
  int G;       // item ranges
  int n_ut;    // user tiles
  int C;       // candidate-list capacity per (list, user)
  int lists;   // G * WI
  int gl;      // lists one merge block takes (its keys live in registers); more lists than that: two merge levels,
  int groups;  // ceil(lists / gl) groups merged to k keys each, then merged with one another
  int64_t B_pad;
  size_t key_bytes;
  size_t tmp_off, tmp_bytes;   // [groups][B_pad][k] keys between the two merge levels (inside ws_bytes)
  size_t ws_bytes;   // candidate lists + one shared threshold per user
  bool ok;
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

constexpr int kMergeKeys = 16384;        // keys per user the merge holds in the registers of 256 threads; 512 threads: twice that
static TopkPlan make_plan(int64_t B, int64_t N, int D, int k, int arith = 0) {
  TopkPlan p{};
  p.ok = false;
  if (B < 1 || N < 1 || D < 4 || D > 256 || (D % 4) != 0 || k < 1 || k > 4096) return p;
  p.DT = D <= 16 ? 16 : D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : 256;
  p.WU = B > 64 ? 4 : 2;
  p.WI = 4 / p.WU;
  // the one-term filter: two user tiles per wave (a staged item row then feeds twice the MFMA work: the filter's loop is bound by
  // instruction issue, not by the matrix pipe) and a merge over twice the keys (its k' is ~2.5 k)
#ifndef LR_TK_TU_MAX
#define LR_TK_TU_MAX 2
#endif
  p.TU = (arith == 2 && B > 128) ? LR_TK_TU_MAX : 1;
  if (arith == 2 && B > 128 && LR_TK_W8) {           // eight waves per workgroup, one user tile each
    p.WU = 8;
    p.WI = 1;
    p.TU = 1;
  }
  const int merge_keys = arith == 2 ? 2 * kMergeKeys : kMergeKeys;
  p.n_ut = static_cast<int>(ceil_div(B, 32 * p.WU * p.TU));
  p.B_pad = static_cast<int64_t>(p.n_ut) * 32 * p.WU * p.TU;
  const int64_t stages = ceil_div(N, static_cast<int64_t>(32 * p.WI));
  // ~3 workgroups per CU (the f32 form's LDS + VGPR budget).  The split-bf16 form holds 2 per CU (80 KB of plane stages, 96 VGPRs
  // of user planes) and was measured with its own one-round grid as well (G = 2 * 256 / n_ut: 146.1 ms per 100 M x 1,024 pass
  // against 141.1 ms with this one, GPU calls r06 topk_sb_time): it keeps the same plan, and the same workspace.
  // (the one-term filter holds two workgroups per CU: one round of them)
#ifndef LR_TK_WGS2
#define LR_TK_WGS2 2
#endif
  int64_t G = ceil_div((arith == 2 ? (p.WU == 8 ? 1 : LR_TK_WGS2) : 3) * kNumCU, p.n_ut);
  // One merge block holds gl lists' keys in registers; above that the lists are merged in groups of gl and the groups with one
  // another (two launches): up to gl^2 lists.  Small batches (one or two user tiles) need many more item ranges than gl to put a
  // workgroup on every CU — with G <= gl a batch of <= 64 users streamed 100 M items through 80 workgroups (39 ms; 73 ms in the
  // f32 form) where the catalogue's bytes alone take 8 ms.
  p.gl = static_cast<int>(merge_keys / k);
  if (p.gl < 1) return p;                            // k too large for the merge
  const int64_t max_lists = static_cast<int64_t>(p.gl) * p.gl;
  if (G * p.WI > max_lists) G = max_lists / p.WI;
  if (G > stages / 4) G = stages / 4;                // >= 4 stages per range
  p.C = round_up((2 * k > k + 64 ? 2 * k : k + 64), 64);
  while (G > 8 && static_cast<size_t>(G) * p.WI * p.B_pad * p.C * sizeof(uint64_t) > (size_t(1) << 30)) G /= 2;   // lists within 1 GiB
  if (G >= 8) G = G / 8 * 8;                         // whole ranges per XCD
  if (G < 1) G = 1;
  p.G = static_cast<int>(G);
  p.lists = p.G * p.WI;
  if (p.lists > max_lists) return p;
  p.groups = p.lists > p.gl ? static_cast<int>(ceil_div(p.lists, p.gl)) : 1;
  p.key_bytes = static_cast<size_t>(p.lists) * p.B_pad * p.C * sizeof(uint64_t);
  // behind the lists: one shared threshold per user, then one progress word per workgroup (see "loose lockstep"), then the keys
  // between the two merge levels
  p.tmp_off = (p.key_bytes + static_cast<size_t>(p.B_pad) * sizeof(uint64_t) +
               static_cast<size_t>(round_up(p.G * p.n_ut, 64)) * sizeof(int) + 255) / 256 * 256;
  p.tmp_bytes = p.groups > 1 ? static_cast<size_t>(p.groups) * p.B_pad * k * sizeof(uint64_t) : 0;
  p.ws_bytes = p.tmp_off + p.tmp_bytes;
  p.ok = true;
  return p;
}

// ---- order-preserving keys ---------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float s) {
  const uint32_t b = __float_as_uint(s + 0.0f);  // -0 -> +0
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ uint64_t make_key(float s, uint32_t id) {
  return (static_cast<uint64_t>(fkey(s)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - id);
}

// exact k-th largest key of L[0..cnt) (cnt >= k), wave-cooperative; returns it on all lanes.
__device__ __forceinline__ uint64_t wave_select_kth(const uint64_t* L, int cnt, int k, int lane) {
  uint64_t T = 0;
  for (int bit = 63; bit >= 0; --bit) {
    const uint64_t trial = T | (1ull << bit);
    int c = 0;
    for (int j0 = 0; j0 < cnt; j0 += kWave) {
      const int j = j0 + lane;
      const bool ge = (j < cnt) && (L[j] >= trial);
      c += __popcll(__ballot(ge));
    }
    if (c >= k) T = trial;
  }
  return T;
}

// keep the entries >= T (exactly k of them when cnt >= k), in place, stable.  Returns new cnt.
__device__ __forceinline__ int wave_compact(uint64_t* L, int cnt, uint64_t T, int lane) {
  int base = 0;
  for (int j0 = 0; j0 < cnt; j0 += kWave) {
    const int j = j0 + lane;
    const uint64_t e = (j < cnt) ? L[j] : 0ull;
    const bool keep = (j < cnt) && (e >= T);
    const uint64_t mask = __ballot(keep);
    const int pre = __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) L[base + pre] = e;  // base+pre <= j: never overtakes unread data of this wave
    base += __popcll(mask);
  }
  return base;
}

// Register-resident selection for short lists (cnt <= 64*CM): the list is read ONCE, the k-th
// key is found by bisection over ballots of register values (score bits first; id bits only
// when ties at the threshold would keep too many), and the survivors are written back compacted.
// ~1-3 us per call instead of 64 dependent passes over global memory.
template <int CM>
__device__ __forceinline__ int wave_select_compact_reg(uint64_t* L, int cnt, int k, int keep_max,
                                                       int lane, uint64_t& T_out) {
  uint64_t e[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    const int j = c * kWave + lane;
    e[c] = (j < cnt) ? L[j] : 0ull;
  }
  auto count_ge = [&](uint64_t t) {
    int n = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c) n += __popcll(__ballot(e[c] >= t));
    return n;
  };
  uint64_t T = 0;
  for (int bit = 63; bit >= 32; --bit) {
    const uint64_t trial = T | (1ull << bit);
    if (count_ge(trial) >= k) T = trial;
  }
  if (count_ge(T) > keep_max) {  // many equal scores at the cut: resolve by id as well
    for (int bit = 31; bit >= 0; --bit) {
      const uint64_t trial = T | (1ull << bit);
      if (count_ge(trial) >= k) T = trial;
    }
  }
  int base = 0;
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    const bool keep = e[c] >= T && e[c] != 0ull;
    const uint64_t mask = __ballot(keep);
    const int pre = __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) L[base + pre] = e[c];
    base += __popcll(mask);
  }
  T_out = T;
  return base;
}

// dispatch on the list capacity; returns the new count and the threshold
__device__ __forceinline__ int wave_shrink_list(uint64_t* L, int cnt, int k, int C, int lane,
                                                uint64_t& T) {
  const int keep_max = C - 64;  // leaves room for two more sub-tiles of appends
  if (C <= 128) return wave_select_compact_reg<2>(L, cnt, k, keep_max, lane, T);
  if (C <= 256) return wave_select_compact_reg<4>(L, cnt, k, keep_max, lane, T);
  if (C <= 512) return wave_select_compact_reg<8>(L, cnt, k, keep_max, lane, T);
  T = wave_select_kth(L, cnt, k, lane);
  return wave_compact(L, cnt, T, lane);
}

__device__ __forceinline__ bool is_consumed(const int32_t* __restrict__ ci, int64_t lo, int64_t hi,
                                            int32_t id) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t v = ci[mid];
    if (v == id) return true;
    if (v < id) lo = mid + 1; else hi = mid;
  }
  return false;
}

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ ci, int64_t lo,
                                                   int64_t hi, int64_t v) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(ci[mid]) < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// The filter's bound: |approx - exact| <= kFiltDelta |u| |i| (bf16 rounding 2^-9 per operand, exact bf16 x bf16 products, <= 129 f32
// additions: 2^-8 + 2^-18 + 129 x 2^-24 = 0.003918, 2 % to spare); kNormUp covers the f32 roundings of a norm's sum and root.
constexpr float kFiltDelta = 0.004f;
constexpr float kNormUp = 1.0009765625f;
using s16x4 = __attribute__((ext_vector_type(4))) short;
constexpr float kBf16Up = 1.00390625f;      // >= (1 + 2^-9) (elements rounded to bf16) x (1 + 2^-10)
__device__ __forceinline__ uint32_t bf16_up(float x) {     // x >= 0 (or inf / NaN) rounded UP to bf16, as the 16 high bits
  const uint32_t u = __float_as_uint(x);
  return (u + ((u & 0xffffu) ? 0x10000u : 0u)) >> 16;
}

// Lab builds (-DLR_TK_MARKS): every wave of the filter kernel adds the shader-clock time it spent per phase of the stage loop
// (0 waiting for the stage, 1 MFMAs + epilogues, 2 waiting for the ring slot, 3 waiting for the prefetch + writing the stage,
// 4 stages, 5 the window-edge lockstep) to lr_tk_marks; read back with lr_score_topk_debug_marks.
#ifdef LR_TK_MARKS
__device__ unsigned long long lr_tk_marks[8];
#define LR_TK_T() __builtin_amdgcn_s_memtime()
#else
#define LR_TK_T() 0ull
#endif

// v + (v of the lane the DPP control names; 0 for the rows outside ROWS)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}

// AR: 0 = the f32 fma chain, 1 = six-term split-bf16 products, 2 = ONE bf16 product per f32 product (operands rounded to bf16,
// f32 accumulation) — an APPROXIMATE score with a proven error bound, only used as the filter of lr_score_topk_filter_f32.
// MASKED (its own instantiation, so that profiles keep it apart from the full passes): the exact re-run for the users the filter
// could not certify.  User slot q of the launch is user `umap[q]` of the caller for q < *n_act and empty beyond (their lists stay
// empty); a workgroup all of whose slots are empty returns at once.  Lists, thresholds and outputs are indexed by SLOT.
// AR 2 scores are UPPER BOUNDS of the exact scores: one more k-block carries delta |u| (user side) and |i| (item side: v_dot2 over the
// row's bf16 fragments as they are read for the MFMAs), both rounded UP to bf16, so that the accumulator ends as
// approx + delta |u| |i|  >=  exact  (kFiltDelta).
// `maxn2` (AR 2, nullable): set to 1 when a staged row's norm is not finite (the bound does not hold for it).
template <int DT, int WU, int AR = 0, int TU = 1, bool MASKED = false>
__global__ __launch_bounds__((TkShape<DT, WU, AR, TU>::NW * 64), (TkShape<DT, WU, AR, TU>::OCC)) void score_topk_kernel(
    const float* __restrict__ users, int64_t B, const float* __restrict__ items, int64_t N, int D,
    const int64_t* __restrict__ consumed_ptr, const int32_t* __restrict__ consumed_idx,
    const uint8_t* __restrict__ filter_flag, int k, int64_t item_base, int G, int n_ut, int C,
    int64_t B_pad, uint64_t* __restrict__ keys, int item_stride, int* __restrict__ progress, int mute_ut,
    const int32_t* __restrict__ umap, const int* __restrict__ n_act, unsigned* __restrict__ maxn2) {
  constexpr bool SB = AR != 0;                 // operands as bf16 planes (AR 1: three, AR 2: one)
  constexpr int NP = AR == 1 ? 3 : 1;          // planes
  // Loose lockstep (`progress`, nullable; n_ut <= 64): the n_ut workgroups of an item range share the range through their
  // XCD's L2, which only works while they read the same neighbourhood — left alone they drift apart (different epilogue
  // work per user tile) and each fetches the range from HBM on its own (measured 2.2x the catalogue at 100 M items).
  // Every workgroup publishes the window (kWinRows item rows) it is in; a wave that is more than one window ahead of the
  // slowest workgroup of its range sleeps until that one catches up.  The wait is BOUNDED (a workgroup that is not
  // resident cannot deadlock the others) and only shapes timing: results do not depend on it.
  // item_stride > 1: the threshold pre-pass over every item_stride-th row of the catalogue — row `it` of this
  // launch is catalogue row it * item_stride (N counts the sampled rows)
  constexpr int NW = TkShape<DT, WU, AR, TU>::NW;    // waves per workgroup
  constexpr int NT = NW * 64;
  constexpr int WI = NW / WU;
  constexpr int DH = DT / 2;          // dims per lane half
  constexpr int LDW = DT + 4;         // padded LDS row (floats)
  static_assert(TU == 1 || AR == 2, "several user tiles per wave: the one-term filter only");
  constexpr int kTI = TkShape<DT, WU, AR, TU>::TI;   // item rows per stage: one 32-row sub-tile per wave (AR 2: LR_TK_RS of them)
  constexpr int SUBS = kTI / 32;
  constexpr int NB = TkShape<DT, WU, AR, TU>::NB;    // stage buffers in the LDS ring (3 workgroups/CU fit; SB: 2)
  constexpr int NQ = kTI * DT / 4;            // float4 slots per stage
  constexpr int NLD = (NQ + NT - 1) / NT;     // float4 staging loads per thread
  // SB: a stage is three bf16 planes [3][kTI][DT bf16 + 16 B pad] (the pad keeps ds_read_b128 of 8 consecutive rows on 8 slots)
  constexpr int RSB = DT * 2 + 16;            // padded plane row (bytes)
  constexpr int PLANE = kTI * RSB;
  constexpr int KB = DT / 16;                 // k-blocks of a 32 x 32 x 16 MFMA
  constexpr int kStageBytes = SB ? NP * PLANE : kTI * LDW * 4;
  static_assert(!SB || (DT >= 16 && DT <= 128), "split-bf16 form: 96 VGPRs of user planes at DT = 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                       // [NB][kTI][LDW]   (SB: [NB][3][kTI][RSB bytes])
  int* cnt_lds = reinterpret_cast<int*>(smem + NB * kStageBytes);     // [4 waves][TU][32]
  // per-wave candidate ring (keeps global stores — and the waits they drag in — out of the
  // per-sub-tile epilogue): [4][kRing] keys, [4][kRing] (user<<16 | slot), [4] counters
  uint64_t* ring_keys_all = reinterpret_cast<uint64_t*>(cnt_lds + NW * TU * 32);
  uint32_t* ring_dst_all = reinterpret_cast<uint32_t*>(ring_keys_all + NW * kRing);
  int* ring_cnt_all = reinterpret_cast<int*>(ring_dst_all + NW * kRing);
  int* full_cnt = ring_cnt_all + NW;  // [NB] waves that have written their share of the stage
  int* done_cnt = full_cnt + NB;      // [NB] waves that have finished reading it
  if (threadIdx.x < 2 * NB) full_cnt[threadIdx.x] = 0;
  __shared__ unsigned s_bad;
  if (AR == 2 && threadIdx.x == 0) s_bad = 0u;
  bool bad_norm = false;              // AR 2: this lane has staged a row whose norm is not finite

  // XCD-aware decode: consecutive blocks of one XCD = the user tiles of one item range.
  const int bid = blockIdx.x;
  int g, ut;
  if (G % 8 == 0) {
    const int xcd = bid % 8, within = bid / 8;
    ut = within % n_ut;
    g = (within / n_ut) * 8 + xcd;
  } else {
    ut = bid % n_ut;
    g = bid / n_ut;
  }
  const int tid = threadIdx.x;
  const int wid = tid / kWave;
  const int lane = tid & (kWave - 1);
  const int wu = wid / WI, wi = wid % WI;
  const int j = lane & 31;   // user column / item row inside a 32x32 tile
  const int h = lane >> 5;   // lane half: owns dims [h*DH, (h+1)*DH)

  // item range of this workgroup, in whole stages
  const int64_t stages_total = ceil_div(N, (int64_t)kTI);
  const int64_t st0 = stages_total * g / G, st1 = stages_total * (g + 1) / G;

  // ---- this wave's users (TU tiles of 32): B operand, resident in registers ------------------
  const int64_t tile0 = (static_cast<int64_t>(ut) * WU + wu) * TU;      // first 32-user tile of this wave
  int64_t user[TU];                   // slot of the launch (lists, thresholds)
  int64_t urow[TU];                   // the caller's user behind it (embedding row, consumed list, filter flag)
  bool user_ok[TU];
  bool any_ok = false;
  [[maybe_unused]] int64_t n_slots = B;
  if constexpr (MASKED) n_slots = *n_act < B ? *n_act : B;
#pragma unroll
  for (int t = 0; t < TU; ++t) {
    user[t] = (tile0 + t) * 32 + j;
    user_ok[t] = user[t] < B && (!MASKED || user[t] < n_slots);
    urow[t] = user[t];
    if constexpr (MASKED) urow[t] = user_ok[t] ? umap[user[t]] : 0;
    any_ok |= user_ok[t];
  }
  if constexpr (MASKED) {              // nobody of this user tile is wanted: leave (the partners of the range must not wait for us)
    if (!__syncthreads_or(any_ok ? 1 : 0)) {
      if (progress != nullptr && tid == 0)
        __hip_atomic_store(progress + static_cast<int64_t>(g) * n_ut + ut, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  float bfrag[SB ? 1 : DH];
  sb::bf16x8 ub1[TU][SB ? KB : 1], ub2[SB ? KB : 1], ub3[SB ? KB : 1];    // SB: the users' planes, lane half h owns k = 16 kb + 8 h ..
  s16x4 ubx[TU];                                                          // AR 2: the bound's k-block of 8 (delta |u| in k = 0)
  if constexpr (SB) {
#pragma unroll
    for (int t = 0; t < TU; ++t) {
      [[maybe_unused]] float un2 = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int d = kb * 16 + h * 8;
        float4 lo = f4_zero(), hi = f4_zero();
        if (user_ok[t] && d < D) lo = ld4(users + urow[t] * D + d);
        if (user_ok[t] && d + 4 < D) hi = ld4(users + urow[t] * D + d + 4);
        if constexpr (AR == 1) {
          sb::split8(lo, hi, ub1[t][kb], ub2[kb], ub3[kb]);
        } else {                                            // AR 2: the operand rounded to bf16
          const sb::u32x4 v = {sb::pack2(lo.x, lo.y), sb::pack2(lo.z, lo.w), sb::pack2(hi.x, hi.y), sb::pack2(hi.z, hi.w)};
          __builtin_memcpy(&ub1[t][kb], &v, 16);
          un2 = fmaf(lo.x, lo.x, fmaf(lo.y, lo.y, fmaf(lo.z, lo.z, fmaf(lo.w, lo.w, un2))));
          un2 = fmaf(hi.x, hi.x, fmaf(hi.y, hi.y, fmaf(hi.z, hi.z, fmaf(hi.w, hi.w, un2))));
        }
      }
      if constexpr (AR == 2) {
        un2 += __shfl_xor(un2, 32);                         // the other half of the user's row
        const uint32_t du = bf16_up(kFiltDelta * sqrtf(un2) * kNormUp);
        ubx[t] = s16x4{static_cast<short>(h == 0 ? du : 0u), 0, 0, 0};
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < DH; s += 4) {
      const int d = h * DH + s;
      float4 x = f4_zero();
      if (user_ok[0] && d < D) x = ld4(users + urow[0] * D + d);
      bfrag[s] = x.x; bfrag[s + 1] = x.y; bfrag[s + 2] = x.z; bfrag[s + 3] = x.w;
    }
  }
  // the user's consumed ids are ascending: narrow the list ONCE to this workgroup's item range,
  // so the per-candidate test below usually sees an empty (or 1-2 element) range
  // ...and keep up to four of them in registers (the common case: ~50 consumed ids spread over
  // G item ranges); longer remainders fall back to the binary search
  int64_t c_lo[TU], c_hi[TU];
  int n_c[TU];
  int32_t cr0[TU], cr1[TU], cr2[TU], cr3[TU];
#pragma unroll
  for (int t = 0; t < TU; ++t) {
#ifdef LR_TK_LAB_NOCONS
    const bool filt = false && consumed_ptr != nullptr && consumed_idx != nullptr &&
#else
    const bool filt = user_ok[t] && consumed_ptr != nullptr && consumed_idx != nullptr &&
#endif
                      (filter_flag == nullptr || filter_flag[urow[t]] != 0);
    c_lo[t] = filt ? consumed_ptr[urow[t]] : 0;
    c_hi[t] = filt ? consumed_ptr[urow[t] + 1] : 0;
    if (c_lo[t] < c_hi[t]) {
      c_lo[t] = lower_bound_i32(consumed_idx, c_lo[t], c_hi[t], item_base + st0 * kTI * item_stride);
      c_hi[t] = lower_bound_i32(consumed_idx, c_lo[t], c_hi[t], item_base + st1 * kTI * item_stride);
    }
    n_c[t] = static_cast<int>(c_hi[t] - c_lo[t]);
    cr0[t] = cr1[t] = cr2[t] = cr3[t] = -1;
    if (n_c[t] >= 1 && n_c[t] <= 4) {
      cr0[t] = consumed_idx[c_lo[t]];
      if (n_c[t] > 1) cr1[t] = consumed_idx[c_lo[t] + 1];
      if (n_c[t] > 2) cr2[t] = consumed_idx[c_lo[t] + 2];
      if (n_c[t] > 3) cr3[t] = consumed_idx[c_lo[t] + 3];
    }
  }

  const int list = g * WI + wi;
  int* wave_cnt = cnt_lds + wid * TU * 32;          // [TU][32] entries of each user's list
  if (h == 0) {
#pragma unroll
    for (int t = 0; t < TU; ++t) wave_cnt[t * 32 + j] = 0;
  }
  uint64_t* ring_keys = ring_keys_all + wid * kRing;
  uint32_t* ring_dst = ring_dst_all + wid * kRing;
  int* ring_cnt = ring_cnt_all + wid;
  if (lane == 0) *ring_cnt = 0;
  // the lists of this wave's 32 TU users are contiguous: slab_keys + (t * 32 + j) * C
  uint64_t* slab_keys = keys + (static_cast<int64_t>(list) * B_pad + tile0 * 32) * C;
  auto ring_flush = [&]() {   // wave-uniform call
    const int n = *ring_cnt < kRing ? *ring_cnt : kRing;
    for (int e = lane; e < n; e += kWave) {
      const uint32_t d = ring_dst[e];
      slab_keys[static_cast<int64_t>(d >> 16) * C + (d & 0xffffu)] = ring_keys[e];
    }
    if (lane == 0) *ring_cnt = 0;
  };
  // Threshold shared by ALL lists of a user (one uint64 per user behind the candidate lists):
  // the k-th best key of ANY single list is a lower bound of the user's global k-th best, so
  // every list may filter with the maximum published so far.  Stale or lost updates only make
  // the filter weaker, never wrong, so relaxed agent-scope accesses suffice and the final result
  // does not depend on timing.
  uint64_t* const tau_base = keys + static_cast<int64_t>(G) * WI * B_pad * C;
  uint64_t tau[TU];                                     // composite threshold of my user
  float tau_s[TU];                                      // its score part (fast pre-test)
#pragma unroll
  for (int t = 0; t < TU; ++t) {
    tau[t] = 0;
    tau_s[t] = user_ok[t] ? -INFINITY : INFINITY;
  }

  // ---- staging helpers -------------------------------------------------------------------
  float4 pre[NLD];
  // Prefetch of the next stage: branch-free loads from clamped addresses.  The loaded registers
  // are NOT touched until stage_write (any use — even a select — would make the compiler wait for
  // the loads right here and serialise HBM/L2 latency with the MFMA phase); validity is kept as
  // a bit mask and applied when writing to LDS.
  uint32_t pre_ok = 0;
  const uint32_t Nu = static_cast<uint32_t>(N), Du = static_cast<uint32_t>(D);
  const uint64_t row_stride = static_cast<uint64_t>(D) * static_cast<uint64_t>(item_stride);
  // A stage that lies whole inside the catalogue (all but the last one, and the clamped ones past a range's end) takes the plain
  // path: the stage's base address is uniform (scalar registers), every lane adds ONE precomputed 32-bit offset, validity is a
  // per-lane constant — the generic path costs ~10 vector instructions per 16-byte load in 64-bit multiplies and clamps, which
  // is what bounds the one-term filter (its loop is issue-bound, not MFMA-bound).
  constexpr int RPL = NT / (DT / 4);                // item rows one load step of the workgroup covers
  constexpr bool kEvenStage = (NQ % NT) == 0;       // every thread stages NLD pieces
  const uint32_t lane_c4 = (static_cast<uint32_t>(tid) % (DT / 4)) * 4;
  const uint32_t lane_off = (static_cast<uint32_t>(tid) / (DT / 4)) * static_cast<uint32_t>(row_stride) + (lane_c4 < Du ? lane_c4 : Du - 4);
  uint32_t lane_ok = 0;                             // which of my NLD pieces exist in a whole stage
#pragma unroll
  for (int u = 0; u < NLD; ++u) lane_ok |= ((tid + u * NT < NQ) && lane_c4 < Du) ? (1u << u) : 0u;
  const bool all_cols = Du == static_cast<uint32_t>(DT);       // no padded columns: a whole stage needs no zero fill
  bool pre_plain = false;                                      // the stage in `pre` was loaded by the plain path and needs no zero fill
  auto stage_load = [&](int64_t st) __attribute__((always_inline)) {     // st may lie past the range's end: addresses are clamped
    const uint32_t it0 = static_cast<uint32_t>(st * kTI < N ? st * kTI : N);   // N < 2^31
    if (static_cast<int64_t>(it0) + NLD * RPL <= N) {
      const float* sbase = items + static_cast<uint64_t>(it0) * row_stride;    // uniform
#pragma unroll
      for (int u = 0; u < NLD; ++u) pre[u] = ld4(sbase + static_cast<uint64_t>(u * RPL) * row_stride + lane_off);
      pre_ok = lane_ok;
      pre_plain = kEvenStage && all_cols;
      return;
    }
    pre_ok = 0;
    pre_plain = false;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = tid + u * NT;
      const uint32_t row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
      const uint32_t it = it0 + row;
      if ((q < NQ) && (it < Nu) && (c4 < Du)) pre_ok |= 1u << u;
      const uint32_t itc = it < Nu ? it : Nu - 1;
      const uint32_t cc = c4 < Du ? c4 : Du - 4;
      pre[u] = ld4(items + (static_cast<uint64_t>(itc) * row_stride + cc));
    }
  };
  auto stage_write_as = [&](int buf, [[maybe_unused]] bool norms, auto plain_tag) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;      // every piece valid: no selects
    if constexpr (SB) {       // the split happens HERE, once per item element per workgroup (not once per wave that multiplies it)
      char* dst = smem + buf * kStageBytes;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int q = tid + u * NT;
        const int row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
        float4 x = pre[u];
        if constexpr (!PLAIN) x = ((q < NQ) && ((pre_ok >> u) & 1u)) ? x : f4_zero();
        if (PLAIN || q < NQ) {
          char* d0 = dst + row * RSB + c4 * 2;
          if constexpr (AR == 1) {
            uint2 p1, p2, p3;
            sb::split4(x, p1, p2, p3);
            *reinterpret_cast<uint2*>(d0) = p1;
            *reinterpret_cast<uint2*>(d0 + PLANE) = p2;
            *reinterpret_cast<uint2*>(d0 + 2 * PLANE) = p3;
          } else {
            *reinterpret_cast<uint2*>(d0) = make_uint2(sb::pack2(x.x, x.y), sb::pack2(x.z, x.w));
          }
        }
      }
    } else {
      float* dst = tile + buf * kTI * LDW;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int q = tid + u * NT;
        const int row = q / (DT / 4), c4 = (q % (DT / 4)) * 4;
        if constexpr (PLAIN) st4(dst + row * LDW + c4, pre[u]);
        else if (q < NQ) st4(dst + row * LDW + c4, ((pre_ok >> u) & 1u) ? pre[u] : f4_zero());
      }
    }
  };
  auto stage_write = [&](int buf, bool norms) __attribute__((always_inline)) {
    if (pre_plain) stage_write_as(buf, norms, std::true_type{});
    else stage_write_as(buf, norms, std::false_type{});
  };

  {  // consume every B-fragment register once: the compiler then waits for those loads HERE and
     // the stage loop carries no vmcnt(0) that would serialise the item prefetch
    float chk = 0.f;
    if constexpr (SB) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int t = 0; t < TU; ++t) chk += static_cast<float>(ub1[t][kb][0]);
        if constexpr (AR == 1) chk += static_cast<float>(ub2[kb][7]) + static_cast<float>(ub3[kb][3]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < DH; ++s) chk += bfrag[s];
    }
    if (chk == 1.2345e30f) wave_cnt[j] = -1;
  }
  // wave-level signalling on LDS counters.  LDS operations of one wave are performed in issue
  // order, so "data writes, then counter += 1" / "counter read, then data reads" needs no fence —
  // only the compiler must not reorder (and a release fence would also wait for the global
  // prefetch in flight, which is exactly what must not happen here).
  auto wave_signal = [&](int* c) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my LDS writes / reads have completed
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto wave_wait = [&](int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
  };

  constexpr int kWinRows = LR_TOPK_WIN_ROWS;      // 12 ranges per XCD x (1 + slack) windows x rows x 512 B against the 4 MB L2
  constexpr int kWinSlack = LR_TOPK_WIN_SLACK;    // windows a workgroup may run ahead of the slowest one of its range
  constexpr int kLockSpins = 2000, kLockSleep = 8; // bound of one window-edge wait (see below)
  constexpr int WN = kWinRows / kTI;              // stages per window
  int* my_prog = progress != nullptr ? progress + static_cast<int64_t>(g) * n_ut : nullptr;
  const int n_st = static_cast<int>(st1 - st0);
  __syncthreads();   // counters zeroed; the only workgroup barrier of the kernel
  for (int pstage = 0; pstage < kPD && pstage < n_st; ++pstage) {
    stage_load(st0 + pstage);
    stage_write(pstage % NB, true);
    wave_signal(&full_cnt[pstage % NB]);
  }

  [[maybe_unused]] unsigned long long mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n_st; ++i) {
    [[maybe_unused]] const unsigned long long tm0 = LR_TK_T();
    const int64_t st = st0 + i;
    const int buf = i % NB;
    const bool more = i + kPD < n_st;
    // The shared threshold is requested first and the item prefetch is issued UNCONDITIONALLY
    // (clamped addresses past the end): the number of younger loads in flight is then a compile-
    // time constant and the wait for the threshold in the epilogue does not drain the prefetch.
    uint64_t tau_seen[TU];
#pragma unroll
    for (int t = 0; t < TU; ++t)
      tau_seen[t] = __hip_atomic_load(tau_base + (user_ok[t] ? user[t] : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int prog_seen = 0x7fffffff;                  // the range's workgroups' windows, read with the threshold (used after the MFMAs)
    const bool win_edge = my_prog != nullptr && (i % WN) == 0;
    if (win_edge) {
      // (mute_ut: test hook, -1 in production — that workgroup never publishes, as if it were not running)
      if (tid == 0 && ut != mute_ut) __hip_atomic_store(progress + static_cast<int64_t>(g) * n_ut + ut, i / WN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prog_seen = __hip_atomic_load(my_prog + (lane < n_ut ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stage_load(st + kPD);  // in flight during the MFMAs below
    wave_wait(&full_cnt[buf], NW * (i / NB + 1));
    [[maybe_unused]] const unsigned long long tm1 = LR_TK_T();

    const float* src = tile + buf * kTI * LDW;
    // one 32-row sub-tile of the stage against this wave's user tiles: the MFMA chains ...
    auto chain = [&](int sub, f32x16 (&acc)[TU]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < TU; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (SB) {
        // item planes from LDS: one ds_read_b128 per plane and k-block feeds six MFMAs (16 B = this lane's 8 k of row j)
        const char* arow = smem + buf * kStageBytes + (sub * 32 + j) * RSB + h * 16;
        [[maybe_unused]] float rn2 = 0.f;          // AR 2: squared norm of this lane's half of (bf16) item row j
#ifndef LR_TK_AGRP
#define LR_TK_AGRP 1
#endif
        constexpr int AG = (AR == 2 && KB % LR_TK_AGRP == 0) ? LR_TK_AGRP : 1;     // A fragments read ahead of their MFMAs
        sb::bf16x8 ag[AG];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          if (kb % AG == 0) {
#pragma unroll
            for (int e = 0; e < AG; ++e) ag[e] = *reinterpret_cast<const sb::bf16x8*>(arow + (kb + e) * 32);
          }
          const sb::bf16x8 a1 = ag[kb % AG];
#ifndef LR_TK_LAB_NONORM
          if constexpr (AR == 2) {                 // four v_dot2_f32_bf16 in the shadow of the MFMAs below
            sb::bf16x2 pr[4];
            __builtin_memcpy(pr, &a1, 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) rn2 = __builtin_amdgcn_fdot2_f32_bf16(pr[e], pr[e], rn2, false);
          }
#endif
          if constexpr (AR == 1) {
            const sb::bf16x8 a2 = *reinterpret_cast<const sb::bf16x8*>(arow + PLANE + kb * 32);
            const sb::bf16x8 a3 = *reinterpret_cast<const sb::bf16x8*>(arow + 2 * PLANE + kb * 32);
            sb::mfma6(acc[0], a1, a2, a3, ub1[0][kb], ub2[kb], ub3[kb]);
          } else {
#pragma unroll
            for (int t = 0; t < TU; ++t) {
#ifdef LR_TK_LAB_NOMFMA
              acc[t][kb] += static_cast<float>(a1[0]);
#else
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ub1[t][kb], acc[t], 0, 0, 0);
#endif
            }
          }
        }
        if constexpr (AR == 2) {                 // + delta |u| |i|: the score becomes an upper bound of the exact one
          // |i| of row j: this lane's half + the half of lane j +- 32 (v_permlane32_swap), x kBf16Up for the bf16 rounding of the
          // row's elements and the f32 roundings of sum and root, rounded UP to bf16; lane half 0 holds k = 0 of the block
          const uint32_t rb = __float_as_uint(rn2);
          const auto sw = __builtin_amdgcn_permlane32_swap(rb, rb, false, false);
          const float n2 = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
          bad_norm |= !(n2 < INFINITY);            // inf, NaN, overflow: the bound does not hold
          // (v_sqrt_f32: 1 ulp, inside kBf16Up's spare)
          const s16x4 ax = {static_cast<short>(h == 0 ? bf16_up(__builtin_amdgcn_sqrtf(n2) * kBf16Up) : 0u), 0, 0, 0};
#pragma unroll
          for (int t = 0; t < TU; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ax, ubx[t], acc[t], 0, 0, 0);   // (k = 8: two-register operands)
        }
      } else {
      const float* arow = src + (sub * 32 + j) * LDW + h * DH;
      // A fragments straight from LDS (ds_read_b128 feeds 4 MFMAs); LDS latency is covered by
      // the second wave of the SIMD (measured: pre-loading whole chunks only costs registers)
#pragma unroll
      for (int s = 0; s < DH; s += 4) {
        const float4 a = ld4(arow + s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bfrag[s], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bfrag[s + 1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bfrag[s + 2], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bfrag[s + 3], acc[0], 0, 0, 0);
      }
      }
    };
    // ... and the threshold tests of its scores
    auto epilogue = [&](int sub, f32x16 (&acc)[TU]) __attribute__((always_inline)) {
      // ---- epilogue (per user tile): threshold filter; lane (j,h) holds items (r&3)+8*(r>>2)+4*h of user j
      // 16-bit mask of the accumulator registers that reach the user's threshold; survivors are
      // rare after warm-up, so the per-survivor work runs in a ctz loop over the set bits only
#pragma unroll
      for (int t = 0; t < TU; ++t) {
      // Tie the threshold's first use to the finished accumulator: without the (empty) asm the
      // compiler hoists part of it — and the wait for its load — above the MFMA chain.
      uint32_t ts_lo = static_cast<uint32_t>(tau_seen[t]), ts_hi = static_cast<uint32_t>(tau_seen[t] >> 32);
      {
        float a0 = acc[t][0];
        asm volatile("" : "+v"(ts_lo), "+v"(ts_hi), "+v"(a0));
        acc[t][0] = a0;
      }
      const uint64_t tau_in = (static_cast<uint64_t>(ts_hi) << 32) | ts_lo;
      if (user_ok[t] && tau_in > tau[t]) {       // another list of this user has raised the bar
        tau[t] = tau_in;
        tau_s[t] = fkey_inv(ts_hi);
      }
      // fast reject: the sub-tile's best score per user against the threshold (15 v_max + 1 cmp)
      float best = acc[t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) best = fmaxf(best, acc[t][r]);
      // A score EQUAL to the threshold's only counts with an id below the threshold's (keys order ties by ascending id): a
      // sub-tile whose first row is not below it cannot hold one.  Without this, tied scores (an all-zero user, duplicated rows)
      // send every sub-tile down the per-survivor path (measured: 206 - 522 ms instead of 141 per 1,024 x 100 M pass).
      const uint32_t tau_it = ~static_cast<uint32_t>(tau[t]);
      const bool pass = best > tau_s[t] || (best == tau_s[t] && static_cast<uint32_t>(st * kTI + sub * 32) < tau_it);
#ifdef LR_TK_LAB_NOEPI
      if (AR == 2 ? __ballot(best == 1.2345e30f) != 0ull : __ballot(pass) != 0ull) {
#else
      if (__ballot(pass) != 0ull) {
#endif
        int* my_cnt = wave_cnt + t * 32;
        uint64_t* my_keys = slab_keys + static_cast<int64_t>(t * 32 + j) * C;
        uint32_t hit = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) hit |= (acc[t][r] >= tau_s[t]) ? (1u << r) : 0u;
        hit = pass ? hit : 0u;
        const int64_t row0 = st * kTI + sub * 32 + 4 * h;
        while (hit != 0u) {
          const int r = __builtin_ctz(hit);
          hit &= hit - 1u;
          float s = acc[t][0];
#pragma unroll
          for (int q = 1; q < 16; ++q) s = (r == q) ? acc[t][q] : s;   // register select, no scratch
          const int64_t it = row0 + (r & 3) + 8 * (r >> 2);
          const uint64_t key = make_key(s, static_cast<uint32_t>(it));
          if (it < N && key > tau[t]) {
            const int32_t gid = static_cast<int32_t>(item_base + it * item_stride);
            const bool seen = (n_c[t] <= 4) ? (gid == cr0[t] || gid == cr1[t] || gid == cr2[t] || gid == cr3[t])
                                            : is_consumed(consumed_idx, c_lo[t], c_hi[t], gid);
            if (!seen) {
              const int slot = atomicAdd(&my_cnt[j], 1);
              const int rp = atomicAdd(ring_cnt, 1);
              if (rp < kRing) {
                ring_keys[rp] = key;
                ring_dst[rp] = (static_cast<uint32_t>(t * 32 + j) << 16) | static_cast<uint32_t>(slot);
              } else {
                my_keys[slot] = key;   // ring full (warm-up bursts): straight to the list
              }
            }
          }
        }
        // lists that could overflow on the next sub-tile (32 new entries per user at most).
        // The counter lives in LDS (in-order per wave); the appended keys are only fenced when a
        // compaction is about to read them back — a fence per sub-tile would stall every epilogue
        // for a global-store round trip.
        const bool need = user_ok[t] && (my_cnt[j] > C - 32);
        uint64_t todo = __ballot(need && h == 0);
        if (todo != 0ull || *ring_cnt > kRing / 2) ring_flush();
        if (todo != 0ull) __threadfence_block();
        while (todo != 0ull) {
          const int uj = __builtin_ctzll(todo);
          todo &= todo - 1ull;
          const int64_t u_glob = (tile0 + t) * 32 + uj;
          uint64_t* L = keys + (static_cast<int64_t>(list) * B_pad + u_glob) * C;
          const int cnt = my_cnt[uj];
          uint64_t T;
          const int kept = wave_shrink_list(L, cnt, k, C, lane, T);
          __threadfence_block();
          if (lane == 0) {
            my_cnt[uj] = kept;
            __hip_atomic_fetch_max(tau_base + u_glob, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (j == uj && T > tau[t]) {
            tau[t] = T;
            tau_s[t] = fkey_inv(static_cast<uint32_t>(T >> 32));
          }
        }
      }
      }
    };
#ifndef LR_TK_PIPE
#define LR_TK_PIPE 0
#endif
#ifndef LR_TK_W8_NOPIPE
#define LR_TK_W8_NOPIPE 0
#endif
    if constexpr (AR == 2 && WI == 1 && SUBS % 2 == 0 && ((WU == 8 && !LR_TK_W8_NOPIPE) || LR_TK_PIPE)) {
      // two accumulator sets: the chains of sub-tile s + 1 are issued before the tests of sub-tile s — the tests run while the
      // matrix pipe works
      f32x16 accA[TU], accB[TU];
      chain(0, accA);
#pragma unroll
      for (int sub = 0; sub < SUBS; sub += 2) {
        chain(sub + 1, accB);
        epilogue(sub, accA);
        if (sub + 2 < SUBS) chain(sub + 2, accA);
        epilogue(sub + 1, accB);
      }
    } else
    {
#pragma unroll 1
      for (int sub = wi; sub < SUBS; sub += WI) {
        [[maybe_unused]] const unsigned long long tma = LR_TK_T();
        f32x16 acc[TU];
        chain(sub, acc);
#ifdef LR_TK_MARKS
      {
        float a0 = acc[0][0], a1 = acc[TU - 1][15];
        asm volatile("" : "+v"(a0), "+v"(a1));          // the MFMA chains have delivered
        acc[0][0] = a0; acc[TU - 1][15] = a1;
      }
      const unsigned long long tmb = LR_TK_T();
      mk[6] += tmb - tma;
#endif
        epilogue(sub, acc);
      }
    }
    [[maybe_unused]] const unsigned long long tm2 = LR_TK_T();
    if (win_edge) {                // too far ahead of the slowest workgroup of the range: let it catch up
      const int want = i / WN - kWinSlack;
      int spins = 0;
      for (;;) {
        int mn = prog_seen;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mn, o); mn = x < mn ? x : mn; }
        if (mn >= want) break;
        // The wait is BOUNDED: kLockSpins polls of kLockSleep x 64 clocks each (~0.5 ms at 2.4 GHz, a thousand times a window's
        // 0.4 us) — a workgroup of the range that is not resident (more workgroups than the chip holds, a partner that has
        // not been dispatched yet) must not be waited for.  Giving up only drops the lockstep (my_prog = nullptr: no more
        // window-edge waits for this workgroup), never a result: tests/test_score_topk_gpu.py mutes a workgroup and gets the same ids.
        if (++spins > kLockSpins) { my_prog = nullptr; break; }
        __builtin_amdgcn_s_sleep(kLockSleep);
        prog_seen = __hip_atomic_load(my_prog + (lane < n_ut ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    [[maybe_unused]] const unsigned long long tm3 = LR_TK_T();
    wave_signal(&done_cnt[buf]);
    [[maybe_unused]] unsigned long long tm4 = tm3;
    if (more) {
      const int b2 = (i + kPD) % NB;
      wave_wait(&done_cnt[b2], NW * ((i + kPD) / NB));   // earlier users of that buffer are through
      tm4 = LR_TK_T();
      stage_write(b2, true);
      wave_signal(&full_cnt[b2]);
    }
#ifdef LR_TK_MARKS
    const unsigned long long tm5 = LR_TK_T();
    mk[0] += tm1 - tm0; mk[1] += tm2 - tm1; mk[5] += tm3 - tm2; mk[2] += tm4 - tm3; mk[3] += tm5 - tm4; mk[4] += 1;
#endif
  }
#ifdef LR_TK_MARKS
  if (AR == 2 && item_stride == 1 && lane == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(&lr_tk_marks[q], mk[q]);
#endif

  if (progress != nullptr && tid == 0 && ut != mute_ut)     // done with the range: never hold the others back
    __hip_atomic_store(progress + static_cast<int64_t>(g) * n_ut + ut, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- final: every list is cut to its best min(cnt,k) entries and padded with 0 to k -----
  ring_flush();
  __threadfence_block();
  for (int uj = 0; uj < 32 * TU; ++uj) {
    const int64_t u_glob = tile0 * 32 + uj;
    if (u_glob >= B) break;
    uint64_t* L = keys + (static_cast<int64_t>(list) * B_pad + u_glob) * C;
    int cnt = wave_cnt[uj];
    if (cnt > k) {  // exact cut to k (ids break score ties: total order)
      if (C <= 512) {
        uint64_t T;
        cnt = (C <= 128)   ? wave_select_compact_reg<2>(L, cnt, k, k, lane, T)
              : (C <= 256) ? wave_select_compact_reg<4>(L, cnt, k, k, lane, T)
                           : wave_select_compact_reg<8>(L, cnt, k, k, lane, T);
      } else {
        const uint64_t T = wave_select_kth(L, cnt, k, lane);
        cnt = wave_compact(L, cnt, T, lane);
      }
    }
    for (int q = cnt + lane; q < k; q += kWave) L[q] = 0ull;
  }
  if constexpr (AR == 2) {
    if (bad_norm) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (tid == 0 && maxn2 != nullptr && s_bad != 0u) atomicOr(maxn2, 1u);
  }
}

// ---- merge: per user, exact k-th largest of the lists' keys by bisection on register-resident
// keys (<= 64 per thread), then only the k winners are sorted in LDS ------------------------------
constexpr int kMergeKPT = 64;   // keys per thread: 256 * 64 = 16384 candidate keys per user (NT = 512: 32768)

// blockIdx.y = group of `gl` lists (one group: the whole merge).  `keys_out` (first of two levels): the group's k best keys, sorted,
// to keys_out[(group * B_pad + u) * k ..] instead of scores / ids.
template <int NT = kBlock>
__global__ __launch_bounds__(NT) void topk_merge_keys_kernel(
    const uint64_t* __restrict__ keys_all, int lists_all, int64_t B_pad, int C, int k, int64_t item_base,
    int K2, float* __restrict__ out_scores, int64_t* __restrict__ out_ids, uint64_t* __restrict__ tau_out,
    int gl, uint64_t* __restrict__ keys_out) {
  const int l0 = static_cast<int>(blockIdx.y) * gl;
  const int lists = lists_all - l0 < gl ? lists_all - l0 : gl;
  const uint64_t* __restrict__ keys = keys_all + static_cast<int64_t>(l0) * B_pad * C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);          // [K2] winners
  __shared__ int wave_cnt[2][NT / kWave];
  __shared__ int n_keep;
  const int64_t u = blockIdx.x;
  const int M = lists * k;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  uint64_t e[kMergeKPT];
#pragma unroll
  for (int c = 0; c < kMergeKPT; ++c) {
    const int q = c * NT + tid;
    uint64_t x = 0ull;
    if (q < M) {
      const int l = q / k, r = q - l * k;
      x = keys[(static_cast<int64_t>(l) * B_pad + u) * C + r];
    }
    e[c] = x;
  }
  if (tid == 0) n_keep = 0;
  // bisection: largest T with |{e >= T}| >= k  (T = 0 when fewer than k keys exist)
  uint64_t T = 0;
  for (int bit = 63; bit >= 0; --bit) {
    const uint64_t trial = T | (1ull << bit);
    int c_loc = 0;
#pragma unroll
    for (int c = 0; c < kMergeKPT; ++c) c_loc += (e[c] >= trial) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c_loc += __shfl_xor(c_loc, off);
    const int par = bit & 1;
    if (lane == 0) wave_cnt[par][wid] = c_loc;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) total += wave_cnt[par][w];
    if (total >= k) T = trial;
  }
  if (tau_out != nullptr) {
    // threshold pre-pass: only the score part of the k-th best key is kept (id part 0: a key with the same
    // score still passes the strict comparison of the main pass)
    if (tid == 0) tau_out[u] = T & 0xFFFFFFFF00000000ull;
    return;
  }
  for (int q = tid; q < K2; q += NT) a[q] = 0ull;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kMergeKPT; ++c)
    if (e[c] != 0ull && e[c] >= T) {
      const int slot = atomicAdd(&n_keep, 1);      // keys are distinct: at most k winners
      if (slot < K2) a[slot] = e[c];
    }
  __syncthreads();
  for (int size = 2; size <= K2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < K2 / 2; t += NT) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const uint64_t x = a[lo], y = a[hi];
        if (desc ? (x < y) : (x > y)) {
          a[lo] = y;
          a[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  if (keys_out != nullptr) {
    for (int r = tid; r < k; r += NT) keys_out[(static_cast<int64_t>(blockIdx.y) * B_pad + u) * k + r] = a[r];
    return;
  }
  for (int r = tid; r < k; r += NT) {
    const uint64_t w = a[r];
    if (w == 0ull) {
      out_scores[u * k + r] = -INFINITY;
      out_ids[u * k + r] = -1;
    } else {
      out_scores[u * k + r] = fkey_inv(static_cast<uint32_t>(w >> 32));
      out_ids[u * k + r] = item_base + static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(w));
    }
  }
}

// merge of already-final (score, global id) lists from S shards: re-key, then same sort.
__global__ __launch_bounds__(kBlock) void topk_merge_pairs_kernel(
    const float* __restrict__ scores, const int64_t* __restrict__ ids, int S, int64_t B, int k,
    int M2, float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);        // (fkey << 32) | ~slot
  const int64_t u = blockIdx.x;
  const int M = S * k;
  // Global ids can exceed 32 bits in principle, so the tie-break rank is computed from the id
  // but the payload carried through the sort is the slot index.  Ties on score are broken by
  // id via a second key array.
  uint64_t* idk = a + M2;                                  // ~id (64-bit) for tie-breaks
  for (int q = threadIdx.x; q < M2; q += kBlock) {
    uint64_t e = 0ull, t = 0ull;
    if (q < M) {
      const int sh = q / k, r = q - sh * k;
      const int64_t src = (static_cast<int64_t>(sh) * B + u) * k + r;
      const int64_t id = ids[src];
      const float s = scores[src];
      if (id >= 0 && s == s) {
        e = (static_cast<uint64_t>(fkey(s)) << 32) | static_cast<uint32_t>(q);
        t = ~static_cast<uint64_t>(id);
      }
    }
    a[q] = e;
    idk[q] = t;
  }
  __syncthreads();
  auto less = [&](uint64_t x, uint64_t xt, uint64_t y, uint64_t yt) {
    const uint32_t xs = static_cast<uint32_t>(x >> 32), ys = static_cast<uint32_t>(y >> 32);
    if (xs != ys) return xs < ys;
    return xt < yt;
  };
  for (int size = 2; size <= M2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < M2 / 2; t += kBlock) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const uint64_t x = a[lo], y = a[hi], xt = idk[lo], yt = idk[hi];
        const bool sw = desc ? less(x, xt, y, yt) : less(y, yt, x, xt);
        if (sw) {
          a[lo] = y; a[hi] = x;
          idk[lo] = yt; idk[hi] = xt;
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < k; r += kBlock) {
    const uint64_t e = a[r];
    if (e == 0ull) {
      out_scores[u * k + r] = -INFINITY;
      out_ids[u * k + r] = -1;
    } else {
      out_scores[u * k + r] = fkey_inv(static_cast<uint32_t>(e >> 32));
      out_ids[u * k + r] = static_cast<int64_t>(~idk[r]);
    }
  }
}

static inline int next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

// test hook (lr_score_topk_test_mute): the user-tile workgroup that never publishes its progress word; -1 = none
static int g_topk_mute_ut = -1;
extern "C" void lr_score_topk_test_mute(int ut) { g_topk_mute_ut = ut; }

struct TopkExtra {            // optional arguments of a launch (all zero: the plain kernels)
  const int32_t* umap;        // MASKED launches: slot -> caller's user ...
  const int* n_act;           // ... for the first *n_act slots
  unsigned* maxn2;            // AR 2: raised to the largest squared item-row norm staged
};

template <int DT, int WU, int AR = 0, int TU = 1, bool MASKED = false>
static int launch_score(const TopkPlan& p, const float* users, int64_t B, const float* items,
                        int64_t N, int D, const int64_t* cptr, const int32_t* cidx,
                        const uint8_t* flag, int k, int64_t item_base, uint64_t* keys,
                        hipStream_t s, int item_stride, int* progress, TopkExtra ex = TopkExtra{}) {
  constexpr int NB = TkShape<DT, WU, AR, TU>::NB;
  constexpr int TI = TkShape<DT, WU, AR, TU>::TI;
  constexpr size_t stage = AR ? static_cast<size_t>(AR == 1 ? 3 : 1) * TI * (DT * 2 + 16) : static_cast<size_t>(TI) * (DT + 4) * 4;
  constexpr int NW = TkShape<DT, WU, AR, TU>::NW;
  const size_t lds = NB * stage + NW * TU * 32 * sizeof(int) +
                     NW * kRing * (sizeof(uint64_t) + sizeof(uint32_t)) + (NW + 2 * NB) * sizeof(int) + 16;
  auto kern = score_topk_kernel<DT, WU, AR, TU, MASKED>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  const int grid = p.G * p.n_ut;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, users, B, items, N, D, cptr, cidx,
                     flag, k, item_base, p.G, p.n_ut, p.C, p.B_pad, keys, item_stride, progress, g_topk_mute_ut, ex.umap, ex.n_act, ex.maxn2);
  return launch_status();
}

template <int DT, int AR = 0>
static int dispatch_wu(const TopkPlan& p, const float* users, int64_t B, const float* items,
                       int64_t N, int D, const int64_t* cptr, const int32_t* cidx,
                       const uint8_t* flag, int k, int64_t item_base, uint64_t* keys,
                       hipStream_t s, int item_stride, int* progress, TopkExtra ex = TopkExtra{}) {
  if constexpr (AR == 2) {
#if LR_TK_W8                                 // (lab: the eight-wave form is only compiled into builds that plan it)
    if (p.WU == 8)
      return launch_score<DT, 8, AR, 1>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
#endif
    if (p.TU == 2) {
      if (p.WU == 4)
        return launch_score<DT, 4, AR, 2>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
      return LR_ESHAPE;                    // (two tiles per wave are planned for batches above 128 users: WU == 4)
    }
  }
  if (p.TU != 1) return LR_ESHAPE;
  if (ex.umap != nullptr) {                 // the masked re-run of the filter: compiled for the filter's reduction widths
    if constexpr (AR != 2 && (DT == 64 || DT == 128)) {
      if (p.WU == 4)
        return launch_score<DT, 4, AR, 1, true>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
      return launch_score<DT, 2, AR, 1, true>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
    }
    return LR_ESHAPE;
  }
  if (p.WU == 4)
    return launch_score<DT, 4, AR>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
  return launch_score<DT, 2, AR>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex);
}

static int dispatch_dt(const TopkPlan& p, const float* users, int64_t B, const float* items, int64_t N, int D,
                       const int64_t* cptr, const int32_t* cidx, const uint8_t* flag, int k, int64_t item_base,
                       uint64_t* keys, hipStream_t s, int item_stride, int* progress = nullptr, int arith = 0,
                       TopkExtra ex = TopkExtra{}) {
#define LR_TK(DTV, ARV) return dispatch_wu<DTV, ARV>(p, users, B, items, N, D, cptr, cidx, flag, k, item_base, keys, s, item_stride, progress, ex)
  if (arith == 1) {         // split-bf16 (compiled for reduction widths up to 128; wider: the f32 chain)
    switch (p.DT) {
      case 16: LR_TK(16, 1);
      case 32: LR_TK(32, 1);
      case 64: LR_TK(64, 1);
      case 128: LR_TK(128, 1);
      default: break;
    }
  }
  if (arith == 2) {         // the one-term bf16 filter: reduction widths 64 and 128 only (the caller checks)
    switch (p.DT) {
      case 64: LR_TK(64, 2);
      case 128: LR_TK(128, 2);
      default: return LR_ESHAPE;
    }
  }
  switch (p.DT) {
    case 16: LR_TK(16, 0);
    case 32: LR_TK(32, 0);
    case 64: LR_TK(64, 0);
    case 128: LR_TK(128, 0);
    default: LR_TK(256, 0);
  }
#undef LR_TK
}

// The catalogue-level threshold pre-pass: every kPreStride-th row is scored first (1/32 of the work); the exact
// k-th best unconsumed score of that sample is a lower bound of every user's final k-th best, so the main pass
// starts with a threshold that admits ~k * kPreStride candidates per user instead of ~k ln(N / (k lists)) per
// list.  Results are unchanged (the threshold only filters).
constexpr int kPreStride = 32;
constexpr int64_t kPreMinItems = int64_t(1) << 20;

}  // namespace lr

using namespace lr;

extern "C" size_t lr_score_topk_ws_bytes(int64_t B, int64_t N, int D, int k) {
  const TopkPlan p = make_plan(B, N, D, k);      // (one plan for both arithmetics)
  return p.ok ? p.ws_bytes : 0;
}

static inline const uint64_t* keys_of(const void* ws) { return static_cast<const uint64_t*>(ws); }

static int score_topk_impl(const float* users, int64_t B, const float* items, int64_t N,
                           int D, const int64_t* consumed_ptr, const int32_t* consumed_idx,
                           const uint8_t* filter_flag, int k, int64_t item_base,
                           float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes,
                           lr_stream_t stream, int arith, TopkExtra ex = TopkExtra{}) {
  LR_CHECK_ARG(B >= 0 && N >= 0 && k >= 1 && item_base >= 0);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(users && out_scores && out_ids);
  LR_CHECK_ARG(N < (int64_t(1) << 31) && item_base + N < (int64_t(1) << 31));
  hipStream_t s = as_stream(stream);
  if (N == 0) {  // nothing to score: every slot is empty (id -1, score -inf)
    hipLaunchKernelGGL(topk_merge_keys_kernel<kBlock>, dim3(static_cast<unsigned>(B)), dim3(kBlock),
                       static_cast<size_t>(next_pow2(k < 2 ? 2 : k)) * sizeof(uint64_t), s, nullptr, 0,
                       int64_t(0), 0, k, item_base, next_pow2(k < 2 ? 2 : k), out_scores, out_ids,
                       static_cast<uint64_t*>(nullptr), 1, static_cast<uint64_t*>(nullptr));
    return launch_status();
  }
  LR_CHECK_ARG(items != nullptr);
  const TopkPlan p = make_plan(B, N, D, k, arith);
  if (!p.ok) return LR_ESHAPE;
  if (ws == nullptr || ws_bytes < p.ws_bytes) return LR_EWORKSPACE;
  auto merge = [&](const TopkPlan& q, uint64_t* tau_out) {     // the lists of a pass -> top k per user (or only its k-th key)
    const int K2m = next_pow2(k < 2 ? 2 : k);
    const size_t lds = static_cast<size_t>(K2m) * sizeof(uint64_t);
    auto launch = [&](const uint64_t* src, int lists, int C, int gl, int groups, uint64_t* keys_out) {
      const dim3 grid(static_cast<unsigned>(B), static_cast<unsigned>(groups));
      const int per_block = lists < gl ? lists : gl;
      if (static_cast<int64_t>(per_block) * k > kMergeKeys)
        hipLaunchKernelGGL(topk_merge_keys_kernel<2 * kBlock>, grid, dim3(2 * kBlock), lds, s, src, lists, q.B_pad, C, k, item_base, K2m,
                           out_scores, out_ids, keys_out != nullptr ? nullptr : tau_out, gl, keys_out);
      else
        hipLaunchKernelGGL(topk_merge_keys_kernel<kBlock>, grid, dim3(kBlock), lds, s, src, lists, q.B_pad, C, k, item_base, K2m,
                           out_scores, out_ids, keys_out != nullptr ? nullptr : tau_out, gl, keys_out);
    };
    if (q.groups <= 1) {
      launch(keys_of(ws), q.lists, q.C, q.lists, 1, nullptr);
    } else {                      // groups of gl lists -> k keys each (tmp, behind the main plan's lists), then the groups
      uint64_t* tmp = reinterpret_cast<uint64_t*>(static_cast<char*>(ws) + p.tmp_off);
      launch(keys_of(ws), q.lists, q.C, q.gl, q.groups, tmp);
      launch(tmp, q.groups, k, q.groups, 1, nullptr);
    }
  };
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(users) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(items) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(ws) % 8 == 0);
  uint64_t* keys = static_cast<uint64_t*>(ws);
  {  // shared per-user thresholds start at 0 ("nothing known yet")
    hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(ws) + p.key_bytes, 0, p.tmp_off - p.key_bytes, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  // loose lockstep of the workgroups of an item range (see the kernel): worth it when a range exceeds what L2 holds
  int* progress = (p.n_ut > 1 && p.n_ut <= 64 && p.G % 8 == 0 && N * D * 4 / p.G > (int64_t(1) << 20))
                      ? reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + p.key_bytes + static_cast<size_t>(p.B_pad) * sizeof(uint64_t))
                      : nullptr;
  int rc;
  if (N >= kPreMinItems) {      // catalogue-level threshold pre-pass over a strided sample
#ifndef LR_TK_PRE_STRIDE2
#define LR_TK_PRE_STRIDE2 32
#endif
    const int pre_stride = arith == 2 ? LR_TK_PRE_STRIDE2 : kPreStride;
    const int64_t Ns = (N + pre_stride - 1) / pre_stride;
    const TopkPlan ps = make_plan(B, Ns, D, k, arith);
    if (ps.ok && ps.key_bytes <= p.key_bytes && ps.B_pad == p.B_pad && ps.tmp_bytes <= p.tmp_bytes) {   // the sample's lists fit the main pass's buffers
      uint64_t* tau = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(ws) + p.key_bytes);
      uint64_t* tau_s = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(ws) + ps.key_bytes);
      // the pre-pass kernel publishes its own running thresholds behind ITS lists (inside the main key buffer)
      hipError_t e2 = hipMemsetAsync(tau_s, 0, static_cast<size_t>(ps.B_pad) * sizeof(uint64_t), s);
      if (e2 != hipSuccess) return static_cast<int>(e2);
      rc = dispatch_dt(ps, users, B, items, Ns, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, keys, s,
                       pre_stride, nullptr, arith, ex);
      if (rc != LR_OK) return rc;
      merge(ps, tau);
    }
  }
  rc = dispatch_dt(p, users, B, items, N, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, keys, s, 1, progress, arith, ex);
  if (rc != LR_OK) return rc;
  merge(p, nullptr);
  return launch_status();
}


// ======================================================================================================================
// Filtered scoring: a cheap pass that PROVABLY cannot lose a winner, exact scores for what it keeps.
//
//   1. the fused score + top-k kernel in arithmetic 2 (ONE bf16 MFMA product per f32 product: both operands rounded to bf16,
//      f32 accumulation) ranks every item of a user by an UPPER BOUND of its exact score,
//          b(u, i) = approx(u, i) + delta |u| |i|  >=  exact(u, i)
//      (|approx - exact| <= delta |u| |i|: kFiltDelta; the term rides in the MFMA chain as one more k-block holding delta |u| and
//      |i|, each rounded up to bf16), and keeps the k' = lr_score_topk_filter_kp(k) > k items of largest bound (consumed ids
//      already dropped).  The bound is per item: a few rows of large norm do not loosen it for the others;
//   2. topk_rescore_kernel recomputes those k' scores in f32 (fixed-order fma chains), sorts them by (score desc, id asc) and
//      keeps the k best — and CERTIFIES the result: every item outside the k' has exact <= b <= b_min = the k'-th largest bound;
//      if the k-th best exact score among the candidates is > b_min, nothing outside can belong to the top k;
//   3. users that are not certified (dense near-ties: more than k' items within the bound's width of the k-th score) are re-run
//      by the exact kernel (MASKED instantiation: workgroups none of whose users failed leave at once) and their rows replaced.
//      A catalogue holding a row whose norm is not finite certifies nobody.
// The returned scores are f32 dot products of their pairs and the ids those of the exact ranking, whatever the data; what the
// data decides is only how many users take the slow path (none on the bench's shape: k' = 256 for k = 100 leaves ~60 spare
// candidates beyond the ~195 the bound needs at 10^8 N(0, 1) items x 128).
// ======================================================================================================================
constexpr int64_t kFiltMinItems = int64_t(1) << 20;

static int filt_kp(int k) {                         // candidates per user of the approximate pass; 0: no filter for this k
  if (k < 1 || k > 100) return 0;                   // (k' <= 256: at k = 200, k' = 456 the pass took as long as the exact kernel)
  int kp = 2 * k + 56;
  if (kp < 64) kp = 64;
  return round_up(kp, 8);                           // <= 256
}

__global__ __launch_bounds__(kBlock) void topk_rescore_kernel(
    const float* __restrict__ users, int D, const float* __restrict__ items, int64_t item_base,
    const float* __restrict__ a_scores, const int64_t* __restrict__ a_ids, int kp, int k, int K2,
    const unsigned* __restrict__ maxn2, float* __restrict__ out_scores, int64_t* __restrict__ out_ids,
    uint8_t* __restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);          // [K2] exact keys of the candidates
  const int64_t u = blockIdx.x;
  const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15;
  const float* up = users + u * D;
  auto sum16 = [](float x) {                                 // over the 16 lanes of a group, fixed order, on every lane
    x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
    return x;
  };
  // a user row holding inf / NaN: its bf16 image can turn an exact +-inf into NaN (inf x a flushed denormal) — never certified
  float uz = 0.f;
  for (int f = tid; f < D / 4; f += kBlock) {
    const float4 w = ld4(up + 4 * f);
    uz += w.x * 0.f + w.y * 0.f + w.z * 0.f + w.w * 0.f;      // 0, or NaN if an element is not finite
  }
  const bool user_bad = __syncthreads_or(uz == 0.f ? 0 : 1) != 0;
  for (int c = grp; c < K2; c += kBlock / 16) {
    uint64_t key = 0ull;
    const int64_t id = c < kp ? a_ids[u * kp + c] : -1;
    if (id >= 0) {                                           // (uniform over the group)
      const int64_t row = id - item_base;
      const float* ip = items + row * D;
      float acc = 0.f;
      for (int f = gl; f < D / 4; f += 16) {
        const float4 x = ld4(ip + 4 * f), w = ld4(up + 4 * f);
        acc = fmaf(x.x, w.x, fmaf(x.y, w.y, fmaf(x.z, w.z, fmaf(x.w, w.w, acc))));
      }
      const float sc = sum16(acc);
      if (sc == sc) key = make_key(sc, static_cast<uint32_t>(row));   // NaN scores are dropped, as everywhere
    }
    if (gl == 0) a[c] = key;
  }
  __syncthreads();
  for (int size = 2; size <= K2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < K2 / 2; t += kBlock) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const uint64_t x = a[lo], y = a[hi];
        if (desc ? (x < y) : (x > y)) {
          a[lo] = y;
          a[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < k; r += kBlock) {
    const uint64_t w = a[r];
    if (w == 0ull) {
      out_scores[u * k + r] = -INFINITY;
      out_ids[u * k + r] = -1;
    } else {
      out_scores[u * k + r] = fkey_inv(static_cast<uint32_t>(w >> 32));
      out_ids[u * k + r] = item_base + static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(w));
    }
  }
  if (tid == 0) {
    const bool full = a_ids[u * kp + kp - 1] >= 0;            // the approximate pass filled its list: items exist outside it
    bool ok = *maxn2 == 0u && !user_bad;                      // (a row with a non-finite norm: no bound, nobody is certified)
    if (ok && full) {
      // every item outside the list has  exact <= its bound <= b_min = the k'-th largest bound: nothing outside can reach a k-th
      // exact score above b_min (a tie with it is not certified)
      const float b_min = a_scores[u * kp + kp - 1];
      const uint64_t wk = a[k - 1];
      ok = wk != 0ull && fkey_inv(static_cast<uint32_t>(wk >> 32)) > b_min;
    }
    fail[u] = ok ? 0 : 1;
  }
}

// The users the filter could not certify, gathered to the front: umap[0 .. F) = their ids in ascending order.  counts[0] = F if
// F <= kFiltSmall else 0 (the re-run planned for kFiltSmall users: more item ranges, every CU busy with few users), counts[1] = F
// if F > kFiltSmall else 0 (the re-run planned for the whole batch, only the first ceil(F / tile) user tiles alive).  One block.
constexpr int kFiltSmall = 128;
__global__ __launch_bounds__(kBlock) void topk_compact_failed_kernel(const uint8_t* __restrict__ fail, int64_t B,
                                                                    int32_t* __restrict__ umap, int* __restrict__ counts) {
  __shared__ int wsum[kBlock / kWave];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int64_t q0 = 0; q0 < B; q0 += kBlock) {
    const int64_t q = q0 + tid;
    const bool f = q < B && fail[q] != 0;
    const uint64_t m = __ballot(f);
    if (lane == 0) wsum[wid] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (f) umap[off + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<int32_t>(q);
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) {
    const int F = base;
    const bool small = B > kFiltSmall && F <= kFiltSmall;
    counts[0] = small ? F : 0;
    counts[1] = small ? 0 : F;
    counts[2] = F;
  }
}

// rows of the re-run (slot order) go to the users they belong to
__global__ __launch_bounds__(kBlock) void topk_scatter_rows_kernel(const int32_t* __restrict__ umap, const int* __restrict__ n_act,
                                                                  int64_t B_launch, int k, const float* __restrict__ s2,
                                                                  const int64_t* __restrict__ i2, float* __restrict__ out_scores,
                                                                  int64_t* __restrict__ out_ids) {
  const int64_t n = *n_act < B_launch ? *n_act : B_launch;
  const int64_t total = n * k, stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int64_t slot = q / k, r = q - slot * k;
    const int64_t dst = static_cast<int64_t>(umap[slot]) * k + r;
    out_scores[dst] = s2[q];
    out_ids[dst] = i2[q];
  }
}

struct FiltLayout {
  int kp;
  TopkPlan pa, pe, psm;      // approximate pass, exact pass over the batch, exact pass over kFiltSmall slots
  size_t r0, off_as, off_ai, off_mx, off_fail, off_s2, off_i2, off_umap, off_cnt, total;
  bool ok;
};
static size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static FiltLayout filt_layout(int64_t B, int64_t N, int D, int k) {
  FiltLayout L{};
  L.ok = false;
  L.kp = filt_kp(k);
  L.pe = make_plan(B, N, D, k);
  if (!L.pe.ok) return L;
  L.r0 = al256(L.pe.ws_bytes);
  L.total = L.r0;
  if (L.kp == 0 || L.kp >= N || !(D > 32 && D <= 128)) return L;         // (the filter kernel is compiled for reduction widths 64 and 128)
  L.pa = make_plan(B, N, D, L.kp, 2);
  if (!L.pa.ok) return L;
  if (al256(L.pa.ws_bytes) > L.r0) L.r0 = al256(L.pa.ws_bytes);
  L.psm = make_plan(B > kFiltSmall ? kFiltSmall : B, N, D, k);
  if (!L.psm.ok) return L;
  if (al256(L.psm.ws_bytes) > L.r0) L.r0 = al256(L.psm.ws_bytes);
  size_t o = L.r0;
  L.off_as = o; o += al256(static_cast<size_t>(B) * L.kp * sizeof(float));
  L.off_ai = o; o += al256(static_cast<size_t>(B) * L.kp * sizeof(int64_t));
  L.off_mx = o; o += 256;
  L.off_fail = o; o += al256(static_cast<size_t>(B));
  L.off_s2 = o; o += al256(static_cast<size_t>(B) * k * sizeof(float));
  L.off_i2 = o; o += al256(static_cast<size_t>(B) * k * sizeof(int64_t));
  L.off_umap = o; o += al256(static_cast<size_t>(B) * sizeof(int32_t));
  L.off_cnt = o; o += 256;
  L.total = o;
  L.ok = true;
  return L;
}

extern "C" int lr_score_topk_f32(const float* users, int64_t B, const float* items, int64_t N,
                                 int D, const int64_t* consumed_ptr, const int32_t* consumed_idx,
                                 const uint8_t* filter_flag, int k, int64_t item_base,
                                 float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes,
                                 lr_stream_t stream) {
  return score_topk_impl(users, B, items, N, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, out_scores, out_ids, ws,
                         ws_bytes, stream, 0);
}

// The same contract with the scores taken as split-bf16 products (f32 accumulation): equal to the f32 chain to f32 rounding,
// not bit for bit.  Reduction widths above 128 run the f32 chain.
extern "C" int lr_score_topk_sb_f32(const float* users, int64_t B, const float* items, int64_t N,
                                    int D, const int64_t* consumed_ptr, const int32_t* consumed_idx,
                                    const uint8_t* filter_flag, int k, int64_t item_base,
                                    float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes,
                                    lr_stream_t stream) {
  return score_topk_impl(users, B, items, N, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, out_scores, out_ids, ws,
                         ws_bytes, stream, 1);
}


#ifdef LR_TK_MARKS
extern "C" int lr_score_topk_debug_marks(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(lr::lr_tk_marks), 8 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(lr::lr_tk_marks), z, sizeof(z));
  }
  return static_cast<int>(e);
}
#endif

extern "C" int lr_score_topk_filter_kp(int k) { return filt_kp(k); }

extern "C" size_t lr_score_topk_filter_ws_bytes(int64_t B, int64_t N, int D, int k) {
  if (B < 1 || N < 1) return 0;
  const FiltLayout L = filt_layout(B, N, D, k);
  return L.pe.ok ? L.total : 0;
}

// flags: bit 0 = use the filter below 2^20 items too (tests); `exact_arith` (0 / 1): the arithmetic of the exact kernel that serves
// the shapes the filter does not take and the users it cannot certify.  `n_fallback` (nullable, device int32): users re-run exactly.
extern "C" int lr_score_topk_filter_f32(const float* users, int64_t B, const float* items, int64_t N, int D,
                                        const int64_t* consumed_ptr, const int32_t* consumed_idx,
                                        const uint8_t* filter_flag, int k, int64_t item_base, float* out_scores,
                                        int64_t* out_ids, void* ws, size_t ws_bytes, int exact_arith, int flags,
                                        uint8_t* failed_out, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && N >= 0 && k >= 1 && item_base >= 0 && (exact_arith == 0 || exact_arith == 1));
  if (B == 0) return LR_OK;
  hipStream_t s = as_stream(stream);
  const FiltLayout L = (N >= 1) ? filt_layout(B, N, D, k) : FiltLayout{};
  const bool use = N >= 1 && L.ok && (N >= kFiltMinItems || (flags & 1));
  if (!use) {
    if (failed_out != nullptr) {
      hipError_t e = hipMemsetAsync(failed_out, 0, static_cast<size_t>(B), s);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    return score_topk_impl(users, B, items, N, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, out_scores, out_ids, ws,
                           ws_bytes, stream, exact_arith);
  }
  LR_CHECK_ARG(users && items && out_scores && out_ids);
  if (ws == nullptr || ws_bytes < L.total) return LR_EWORKSPACE;
  LR_CHECK_ARG(reinterpret_cast<uintptr_t>(ws) % 256 == 0);
  char* w = static_cast<char*>(ws);
  float* a_s = reinterpret_cast<float*>(w + L.off_as);
  int64_t* a_i = reinterpret_cast<int64_t*>(w + L.off_ai);
  unsigned* mx = reinterpret_cast<unsigned*>(w + L.off_mx);
  uint8_t* fail = reinterpret_cast<uint8_t*>(w + L.off_fail);
  float* s2 = reinterpret_cast<float*>(w + L.off_s2);
  int64_t* i2 = reinterpret_cast<int64_t*>(w + L.off_i2);
  hipError_t e = hipMemsetAsync(mx, 0, 256, s);
  if (e != hipSuccess) return static_cast<int>(e);
  TopkExtra ex{};
  ex.maxn2 = mx;
  int rc = score_topk_impl(users, B, items, N, D, consumed_ptr, consumed_idx, filter_flag, L.kp, item_base, a_s, a_i, ws, L.r0,
                           stream, 2, ex);
  if (rc != LR_OK) return rc;
  const int K2 = next_pow2(L.kp);
  hipLaunchKernelGGL(topk_rescore_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), static_cast<size_t>(K2) * sizeof(uint64_t), s,
                     users, D, items, item_base, a_s, a_i, L.kp, k, K2, mx, out_scores, out_ids, fail);
  rc = launch_status();
  if (rc != LR_OK) return rc;
  // the users without a proof, gathered to the front, through the exact kernel: planned for kFiltSmall slots when they are few
  // (every CU busy with few users: more item ranges), for the whole batch otherwise (only the leading user tiles alive) — the
  // launch that does not apply finds no slot alive and returns at once
  int32_t* umap = reinterpret_cast<int32_t*>(w + L.off_umap);
  int* cnt = reinterpret_cast<int*>(w + L.off_cnt);
  hipLaunchKernelGGL(topk_compact_failed_kernel, dim3(1), dim3(kBlock), 0, s, fail, B, umap, cnt);
  for (int pass = 0; pass < 2; ++pass) {
    const int64_t Bl = pass == 0 ? kFiltSmall : B;
    if (pass == 0 && B <= kFiltSmall) continue;
    TopkExtra ex2{};
    ex2.umap = umap;
    ex2.n_act = cnt + pass;
    rc = score_topk_impl(users, Bl, items, N, D, consumed_ptr, consumed_idx, filter_flag, k, item_base, s2, i2, ws, L.r0, stream,
                         exact_arith, ex2);
    if (rc != LR_OK) return rc;
    hipLaunchKernelGGL(topk_scatter_rows_kernel, dim3(grid_for(Bl * k, kBlock)), dim3(kBlock), 0, s, umap, cnt + pass, Bl, k, s2, i2,
                       out_scores, out_ids);
  }
  if (failed_out != nullptr) {
    e = hipMemcpyAsync(failed_out, fail, static_cast<size_t>(B), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return launch_status();
}

extern "C" int lr_topk_merge_f32(const float* scores, const int64_t* ids, int S, int64_t B, int k,
                                 float* out_scores, int64_t* out_ids, lr_stream_t stream) {
  LR_CHECK_ARG(S >= 1 && B >= 0 && k >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(scores && ids && out_scores && out_ids);
  if (static_cast<int64_t>(S) * k > 8192) return LR_ESHAPE;
  const int M2 = next_pow2(S * k < 2 ? 2 : S * k);
  const size_t lds = static_cast<size_t>(M2) * sizeof(uint64_t) * 2;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_merge_pairs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(topk_merge_pairs_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), lds,
                     as_stream(stream), scores, ids, S, B, k, M2, out_scores, out_ids);
  return launch_status();
}
