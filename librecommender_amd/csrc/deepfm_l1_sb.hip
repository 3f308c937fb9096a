// DeepFM first MLP layer fused with the embedding lookup — the three contractions as SPLIT-bf16 MFMA products with f32
// accumulation (round 5: the default arithmetic of the layer where the shape is compiled, K = 64, H1 = 128).
//
//   lr_deepfm_l1_fwd_sb_f32    z1[b, :]   = sum_f table[idx[b, f], :] @ Wp[f*K:(f+1)*K, :] + bias   (+ fsum / pair / lin_out)
//   lr_deepfm_l1_wgrad_sb_f32  partial[c] = sum over chunk c of table[idxT[f, b], :]^T gz[b, :]
//   lr_deepfm_l1_dgrad_sb_f32  ge[slotT[f, b], :] = gz[b, :] @ Wp[f*K:(f+1)*K, :]^T + gl[b] * wp[:] * fsum[b, :]
// — the contracts of lr_deepfm_l1_fwd / wgrad / dgrad_f32 (deepfm_l1.hip; reference: algorithms/deepfm.py:155-170,
// layers/dense.py:12-49, training/tf_trainer.py:120-121), same arguments except the packed weights.
//
// Arithmetic.  Every f32 operand is split exactly into three bf16 values (x = x1 + x2 + x3, round-to-nearest-even at each
// step) and a product a*b is taken as the six largest of the nine cross terms, a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1
// (smallest first), each an exact bf16 x bf16 product accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Measured on this
// layer's own reduction (K = 12,928) the result is as close to f64 as the f32 fma chain of deepfm_l1.hip (relative rms error
// 1.88e-6 vs 2.08e-6, profiles/r04_bf16_split_probe.txt); it is NOT bit-identical to it: tests pin both against f64.
// Why: a bf16 MFMA does 32,768 FLOP in 32 cycles and hides ~5 other instructions behind it, an f32 MFMA does 4,096 FLOP
// in 64 cycles on the vector ALU and hides none — six bf16 MFMAs per 32 x 32 x 16 block are 192 cycles against 512.
//
// What bound round 4's first forward kernel (0.44 ms against 0.15 ms of MFMA time): every wave both staged (gather, split,
// LDS writes) and multiplied, and the one barrier per field put all eight waves of a CU into the SAME phase — a field cost
// MFMA time + staging time (profiles/r04_l1_split_bf16_kernel.md); and at 64 samples per workgroup the weight planes were
// re-read from L2 once per 64 samples (4.1 GB per launch).  The kernels here:
//   * 128 samples per workgroup; where 128-sample tiles do not fill the chip the FIELDS are split across workgroups (the
//     forward then writes partial sums that a small kernel adds; the row-gradient kernel has no sum across fields);
//   * forward / weight gradient: wave roles.  Waves 0-3 (one per SIMD) only multiply: operands come out of LDS / registers,
//     nothing else is in their instruction stream; waves 4-7 (the other wave of each SIMD) only stage: rows requested FOUR
//     stages ahead into two register sets, split / written into a ring of three LDS buffers two stages ahead — their VALU
//     and memory instructions issue in the shadow of the partner's bf16 MFMAs;
//   * row gradient: gz is the stationary operand (split once, 96 VGPRs), the weight planes stream through a two-buffer LDS
//     ring filled by LDS-direct loads (global_load_lds_dwordx4: no staging registers, no ds_write), eight multiplying waves.
#include <type_traits>

#include "common.hpp"

namespace lr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int SB_KD = 64, SB_H1 = 128;                 // the compiled shape
constexpr int SB_THREADS = 512;

// Profiling builds only (scripts/lab/r05/ablate_build.sh compiles this file with -DLR_SB_ABLATE=<bits> into a separate library
// loaded through LIBRECO_HIP_LIB): 1: every gather reads row 0 (no HBM latency / traffic)  2: no MFMA  4: the row-gradient
// kernel stores nothing  8: weight planes never refilled (no L2 -> CU weight stream)  16: the multiplying waves read no
// fragments from LDS  32: the staging waves write nothing to LDS  64: no barriers.  The product build defines nothing.
#ifndef LR_SB_ABLATE
#define LR_SB_ABLATE 0
#endif
constexpr int kSbAblate = LR_SB_ABLATE;
__device__ __forceinline__ void sb_sync() {
  if (!(kSbAblate & 64)) __syncthreads();
}

__device__ __forceinline__ void split3(f32x8 x, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  a1 = __builtin_convertvector(x, bf16x8);
  const f32x8 r1 = x - __builtin_convertvector(a1, f32x8);
  a2 = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(a2, f32x8);
  a3 = __builtin_convertvector(r2, bf16x8);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {          // one v_cvt_pk_bf16_f32: element 0 in the low half
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
__device__ __forceinline__ float lo_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// four f32 -> three planes of four bf16 (two packed words each): x = p1 + p2 + p3 exactly (barring underflow)
__device__ __forceinline__ void split4(float4 x, uint2& p1, uint2& p2, uint2& p3) {
  p1.x = pack2(x.x, x.y); p1.y = pack2(x.z, x.w);
  const float r0 = x.x - lo_f32(p1.x), r1 = x.y - hi_f32(p1.x), r2 = x.z - lo_f32(p1.y), r3 = x.w - hi_f32(p1.y);
  p2.x = pack2(r0, r1); p2.y = pack2(r2, r3);
  const float s0 = r0 - lo_f32(p2.x), s1 = r1 - hi_f32(p2.x), s2 = r2 - lo_f32(p2.y), s3 = r3 - hi_f32(p2.y);
  p3.x = pack2(s0, s1); p3.y = pack2(s2, s3);
}
// eight f32 (two float4: elements 0-3, 4-7) -> three bf16x8 operand fragments
__device__ __forceinline__ void split8(float4 lo, float4 hi, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  uint2 l1, l2, l3, h1, h2, h3;
  split4(lo, l1, l2, l3);
  split4(hi, h1, h2, h3);
  const u32x4 v1 = {l1.x, l1.y, h1.x, h1.y}, v2 = {l2.x, l2.y, h2.x, h2.y}, v3 = {l3.x, l3.y, h3.x, h3.y};
  __builtin_memcpy(&a1, &v1, 16);
  __builtin_memcpy(&a2, &v2, 16);
  __builtin_memcpy(&a3, &v3, 16);
}
// acc += a * b with a = a1 + a2 + a3, b = b1 + b2 + b3: the six largest cross terms, smallest first
__device__ __forceinline__ void mfma6(f32x16& acc, bf16x8 a1, bf16x8 a2, bf16x8 a3, bf16x8 b1, bf16x8 b2, bf16x8 b3) {
  if (kSbAblate & 2) {
    asm volatile("" : "+v"(acc) : "v"(a1), "v"(a2), "v"(a3), "v"(b1), "v"(b2), "v"(b3));
    return;
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 acc0() { return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
// LDS-direct load of 64 x 16 bytes: lane l's 16 bytes at `gsrc` (per lane) land at lds_wave_base (wave-uniform) + 16 l
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc), (lds_void_t*)(lds_wave_base), 16, 0, 0);
}

// -------------------------------------------------------------------------------------------------------------------------
// Packing.  All operand planes are stored in MFMA fragment order, 16 bytes (8 bf16) per lane and plane, lane = 32 g + j:
//   WsbA (forward, B operand)        [f][kb = K/16][ct = H1/32][plane 3][lane][8]   element e: Wp[f*K + 16 kb + 8 g + e][32 ct + j]
//   WsbB (row gradient, A operand of the transposed product ge^T = Wp_f gz^T)
//                                    [f][kb = H1/16][nt = K/32][plane 3][lane][8]   element e: Wp[f*K + 32 nt + j][16 kb + 8 g + e]
//   gzp  (weight gradient, B operand) [slab = ceil(B/16)][nt = H1/32][plane 3][lane][8]   element e: gz[16 slab + 8 g + e][32 nt + j]
// (the reduction index of an operand element only has to be the same function of (g, e) for both operands of an MFMA).
// Wp = diag(scale) W (the BatchNorm fold), formed in f32 exactly as lr_deepfm_l1_pack_scaled_f32 does, then split.
// -------------------------------------------------------------------------------------------------------------------------
// `red_partial` (nullable): the LAST ceil(H1 / 16) workgroups sum the folded bias's slab partials into `red_out` instead of packing
// (reduce_partials_body, the arithmetic of lr_reduce_partials_f32): that reduction needs no launch of its own
__global__ __launch_bounds__(kBlock) void l1_sb_pack_kernel(const float* __restrict__ W, const float* __restrict__ scale, int F,
                                                            int K, int H1, bf16x8* __restrict__ outA,
                                                            bf16x8* __restrict__ outB, const float* __restrict__ red_partial,
                                                            int red_nblk, float* __restrict__ red_out) {
  const int n_red = red_partial != nullptr ? (H1 + 15) / 16 : 0;
  const int n_pack = static_cast<int>(gridDim.x) - n_red;
  if (static_cast<int>(blockIdx.x) >= n_pack) {
    reduce_partials_body(static_cast<int>(blockIdx.x) - n_pack, n_red, red_partial, red_nblk, H1, H1, red_out, nullptr);
    return;
  }
  const int64_t total = static_cast<int64_t>(F) * K * H1 / 8;              // one (.., lane) per thread and buffer: 3 x 16 bytes
  const int64_t stride = static_cast<int64_t>(n_pack) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int lane = static_cast<int>(q & 63);
    const int j = lane & 31, g = lane >> 5;
    if (outA != nullptr) {
      const int KB = K / 16, CT = H1 / 32;
      int64_t t = q >> 6;
      const int ct = static_cast<int>(t % CT); t /= CT;
      const int kb = static_cast<int>(t % KB);
      const int f = static_cast<int>(t / KB);
      const int64_t row0 = static_cast<int64_t>(f) * K + kb * 16 + 8 * g;
      const int col = ct * 32 + j;
      f32x8 x;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = W[(row0 + e) * H1 + col];
        if (scale != nullptr) v *= scale[row0 + e];
        x[e] = v;
      }
      bf16x8 p1, p2, p3;
      split3(x, p1, p2, p3);
      bf16x8* dst = outA + (((static_cast<int64_t>(f) * KB + kb) * CT + ct) * 3) * 64 + lane;
      dst[0] = p1; dst[64] = p2; dst[128] = p3;
    }
    if (outB != nullptr) {
      const int KBH = H1 / 16, NT = K / 32;
      int64_t t = q >> 6;
      const int nt = static_cast<int>(t % NT); t /= NT;
      const int kb = static_cast<int>(t % KBH);
      const int f = static_cast<int>(t / KBH);
      const int64_t row = static_cast<int64_t>(f) * K + nt * 32 + j;
      const float s = scale != nullptr ? scale[row] : 1.f;
      const float* src = W + row * H1 + kb * 16 + 8 * g;
      f32x8 x;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = scale != nullptr ? src[e] * s : src[e];
      bf16x8 p1, p2, p3;
      split3(x, p1, p2, p3);
      bf16x8* dst = outB + (((static_cast<int64_t>(f) * KBH + kb) * NT + nt) * 3) * 64 + lane;
      dst[0] = p1; dst[64] = p2; dst[128] = p3;
    }
  }
}

__global__ __launch_bounds__(kBlock) void l1_sb_gz_pack_kernel(const float* __restrict__ gz, int64_t B, int H1,
                                                               bf16x8* __restrict__ out) {
  const int NT = H1 / 32;
  const int64_t slabs = ceil_div(B, 16);
  const int64_t total = slabs * NT * 64;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int lane = static_cast<int>(q & 63);
    const int j = lane & 31, g = lane >> 5;
    int64_t t = q >> 6;
    const int nt = static_cast<int>(t % NT);
    const int64_t slab = t / NT;
    const int64_t s0 = slab * 16 + 8 * g;
    f32x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (s0 + e < B) ? gz[(s0 + e) * H1 + nt * 32 + j] : 0.f;
    bf16x8 p1, p2, p3;
    split3(x, p1, p2, p3);
    bf16x8* dst = out + ((slab * NT + nt) * 3) * 64 + lane;
    dst[0] = p1; dst[64] = p2; dst[128] = p3;
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(B / TS), KS): workgroup (tile, y) multiplies the tile's TS samples with the fields
// [y F / KS, (y + 1) F / KS).  KS == 1: writes z1 (+ bias) / pair / fsum; KS > 1: partial z1 / field sums / sums of
// squares into the workspace, added by l1_sb_combine_kernel.
//   waves 0-3  multiply: wave c owns output columns [32 c, 32 c + 32) of all TS / 32 sample tiles; its weight fragments
//              (12 x 16 bytes per field) come straight from L2 into ONE register copy, refilled in place for the next field
//              as soon as a group of MFMAs has consumed them; the A planes come out of the LDS ring, a k-block ahead.
//   waves 4-7  stage: rows of field i + 4 requested from HBM into one of two register sets, field i + 2 split into planes and
//              written into ring buffer (i + 2) % 3 (rotated slots: conflict-free 8-byte writes), FM sums, linear weights.
//   one barrier per field: buffer i % 3 is read (and (i + 1) % 3 prefetched from) while (i + 2) % 3 is written.
// -------------------------------------------------------------------------------------------------------------------------
template <int TS, bool kLin>
__global__ __launch_bounds__(SB_THREADS, (TS == 64 ? 4 : 2)) void l1_fwd_sb_kernel(
    const float* __restrict__ table, const float* __restrict__ lin, int64_t V, const int32_t* __restrict__ idx, int64_t B,
    int F, const bf16x8* __restrict__ Wsb, const float* __restrict__ bias, float* __restrict__ z1,
    float* __restrict__ pair, float* __restrict__ fsum, float* __restrict__ lin_out, float* __restrict__ ws) {
  constexpr int KD = SB_KD, H1 = SB_H1, KB = KD / 16, CT = H1 / 32, RT = TS / 32;
  constexpr int NLD = TS / 16;                          // rows per staging thread and field (256 staging threads, 16 per row)
  constexpr int AF = RT * KB * 3 * 64;                  // 16-byte slots of one field's A planes
  constexpr int WF = KB * CT * 3 * 64;                  // ... of one field's weight planes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* al = reinterpret_cast<bf16x8*>(smem);         // [3][AF]

  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int j = lane & 31, g = lane >> 5;
  const int KS = gridDim.y, y = blockIdx.y;
  const int f_lo = static_cast<int>(static_cast<int64_t>(F) * y / KS);
  const int nf = static_cast<int>(static_cast<int64_t>(F) * (y + 1) / KS) - f_lo;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * TS;
  const int nb = (B - b0) < TS ? static_cast<int>(B - b0) : TS;
  float* z1p = ws + static_cast<int64_t>(y) * B * H1;                                   // [KS][B][H1]
  float* Sp = ws + static_cast<int64_t>(KS) * B * H1 + static_cast<int64_t>(y) * B * KD;  // [KS][B][KD]
  float* Qp = Sp + static_cast<int64_t>(KS) * B * KD;                                   // [KS][B][KD]

  if (wid < 4) {
    // ================================================= multiplying waves =================================================
    const int ct = wid;
    bf16x8 bq[KB][3];
    auto b_ptr = [&](int i) { return Wsb + static_cast<int64_t>(f_lo + i) * WF + (ct * 3) * 64 + lane; };
    {
      const bf16x8* src = b_ptr(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[kb][p] = src[(kb * CT * 3 + p) * 64];
    }
    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = acc0();
    // A fragments: ONE register copy per sample tile, refilled in place with the tile's next k-block (of the next field after
    // the last one: that ring buffer was completed a step ago) as soon as the tile's six MFMAs are issued — the next use is
    // (RT - 1) x 6 MFMAs away
    bf16x8 afr[RT][3];
    auto a_read = [&](int buf, int kb, int t, bf16x8 (&dst)[3]) {
      const bf16x8* ar = al + buf * AF + g * 32 + ((j + 2 * kb + g) & 31);         // the rotated slot of this lane's row
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (!(kSbAblate & 16)) dst[p] = ar[((t * KB + kb) * 3 + p) * 64];
    };
    sb_sync();                                    // fields 0 / 1 staged
#pragma unroll
    for (int t = 0; t < RT; ++t) a_read(0, 0, t, afr[t]);
    int cur = 0, nxt = 1;                               // ring buffers of field i / i + 1
    for (int i = 0; i < nf; ++i) {
      const bf16x8* bnext = b_ptr(i + 1 < nf ? i + 1 : i);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          mfma6(acc[t], afr[t][0], afr[t][1], afr[t][2], bq[kb][0], bq[kb][1], bq[kb][2]);
          if (kb + 1 < KB) a_read(cur, kb + 1, t, afr[t]);
          else a_read(nxt, 0, t, afr[t]);
          // program order IS the schedule: left to itself the compiler gathers the refills at the end of the field (the first
          // MFMA of the next field then waits for an L2 round trip) and reads fragments right in front of their MFMAs
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!(kSbAblate & 8)) {
#pragma unroll
          for (int p = 0; p < 3; ++p) bq[kb][p] = bnext[(kb * CT * 3 + p) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      sb_sync();
      cur = nxt;
      nxt = (nxt == 2) ? 0 : nxt + 1;
    }
    const int col = ct * 32 + j;
    const float bv = (KS == 1 && bias != nullptr) ? bias[col] : 0.f;
    float* zo = KS == 1 ? z1 : z1p;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int smp = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (smp < nb) zo[(b0 + smp) * H1 + col] = acc[t][r] + bv;
      }
  } else {
    // =================================================== staging waves ===================================================
    const int tid = static_cast<int>(threadIdx.x) - 256;
    const int srow = tid >> 4, c4 = (tid & 15) * 4;
    const uint32_t Vu = static_cast<uint32_t>(V);
    // where this thread's four floats of a row go inside an A plane (bytes from the plane-0 slot): k = c4 .. c4 + 3 ->
    // kb = c4 / 16, lane half g = (c4 / 8) & 1, elements c4 & 7 ..  The slot of row r inside a 32-lane half is ROTATED by
    // 2 kb + g: the 16 chunks of one row (one 16-lane group of an 8-byte store) would otherwise start 256 bytes apart — with
    // the rotation they cover all sixteen 8-byte positions of a 128-byte bank row (round 4's rotation by 8 kb + 4 g left them
    // on four: SQ_LDS_BANK_CONFLICT was half of all LDS cycles); the fragment reads (16 different rows per lane group, 16 bytes
    // each) stay conflict-free for any rotation.  Row r = srow + 16 u lies in sample tile u >> 1 at row srow + 16 (u & 1) of it.
    uint32_t a_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = srow + 16 * h;
      const int kb_ = c4 >> 4, g_ = (c4 >> 3) & 1;
      a_base[h] = static_cast<uint32_t>(((kb_ * 3) * 64 + g_ * 32 + ((r + 2 * kb_ + g_) & 31)) * 16 + (c4 & 7) * 2);
    }
    auto a_off = [&](int u) { return a_base[u & 1] + static_cast<uint32_t>((u >> 1) * (KB * 3 * 64 * 16)); };
    uint32_t rowok = 0;
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (srow + u * 16 < nb) rowok |= 1u << u;
    // (clamped rows: every load is unconditional)
    const int32_t* idt = idx + b0 * F + f_lo;
    auto row_off = [&](int u) { const int r = srow + u * 16; return static_cast<uint32_t>(r < nb ? r : nb - 1) * static_cast<uint32_t>(F); };
    // linear weights: row srow + 16 q of a field is looked after by the thread with chunk index q (one value per thread)
    const int lq = tid & 15;
    const bool l_mine = kLin && lq < NLD && srow + 16 * lq < nb;
    const uint32_t l_off = static_cast<uint32_t>(l_mine ? srow + 16 * lq : 0) * static_cast<uint32_t>(F);
    float4 pre[2][NLD];
    float prel[2] = {0.f, 0.f};
    uint32_t pre_ok[2] = {0u, 0u};
    bool prel_ok[2] = {false, false};
    int32_t idn[2][NLD];
    int32_t idl[2] = {-1, -1};
    float4 S[NLD], Q[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) { S[u] = f4_zero(); Q[u] = f4_zero(); }

    // The ids of a field are requested a whole step BEFORE its rows and, within a step, IN FRONT of that step's row
    // requests: the memory counter counts in order, so the wait for ids issued behind a batch of row requests would also
    // wait for those rows (an HBM round trip per step; deepfm_l1.hip, round 4).
    auto ids_load = [&](int i, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const int ic = i < nf ? i : nf - 1;
#pragma unroll
      for (int u = 0; u < NLD; ++u) idn[set][u] = idt[row_off(u) + ic];
      if (kLin) idl[set] = idt[l_off + ic];
    };
    auto stage_load = [&](int i, auto set_c) {          // rows of field i (ids requested a step earlier) -> register set
      constexpr int set = decltype(set_c)::value;
      pre_ok[set] = 0;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const bool ok = static_cast<uint32_t>(idn[set][u]) < Vu;
        const uint32_t id = (ok && !(kSbAblate & 1)) ? static_cast<uint32_t>(idn[set][u]) : 0u;
        if (ok) pre_ok[set] |= 1u << u;
        pre[set][u] = ld4(table + static_cast<uint64_t>(id) * KD + c4);
      }
      pre_ok[set] &= rowok;
      if (kLin) {
        prel_ok[set] = static_cast<uint32_t>(idl[set]) < Vu;
        prel[set] = lin[prel_ok[set] ? static_cast<uint32_t>(idl[set]) : 0u];
      }
    };
    auto stage_write = [&](int i, int buf, auto set_c) {  // register set -> ring buffer `buf`: planes; FM sums; linear weights out
      constexpr int set = decltype(set_c)::value;
      char* da = reinterpret_cast<char*>(al + buf * AF);
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const bool ok = (pre_ok[set] >> u) & 1u;
        const float4 x = ok ? pre[set][u] : f4_zero();
        S[u] = f4_add(S[u], x);
        Q[u] = f4_fma(x, x, Q[u]);
        uint2 p1, p2, p3;
        split4(x, p1, p2, p3);
        if (kSbAblate & 32) asm volatile("" :: "v"(p1.x), "v"(p2.x), "v"(p3.x), "v"(p1.y), "v"(p2.y), "v"(p3.y));
        else {
          *reinterpret_cast<uint2*>(da + a_off(u)) = p1;
          *reinterpret_cast<uint2*>(da + a_off(u) + 1024) = p2;
          *reinterpret_cast<uint2*>(da + a_off(u) + 2048) = p3;
        }
      }
      if (kLin && l_mine) lin_out[b0 * F + f_lo + l_off + i] = prel_ok[set] ? prel[set] : 0.f;
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    ids_load(0, S0{});
    ids_load(1, S1{});
    stage_load(0, S0{});
    stage_load(1, S1{});
    ids_load(2, S0{});
    ids_load(3, S1{});
    stage_write(0, 0, S0{});
    if (nf > 1) stage_write(1, 1, S1{});
    stage_load(2, S0{});
    stage_load(3, S1{});
    ids_load(4, S0{});
    sb_sync();
    // step i: ids of field i + 5 requested; field i + 2 -> buffer (i + 2) % 3 from set i & 1; rows of field i + 4 requested
    // into the same set.  The steady loop is free of conditionals (fields beyond the range are clamped re-reads that nobody
    // writes out).
    int i = 0, wb = 2;
    for (; i + 3 < nf; i += 2) {
      ids_load(i + 5, S1{});
      stage_write(i + 2, wb, S0{});
      stage_load(i + 4, S0{});
      sb_sync();
      wb = (wb == 2) ? 0 : wb + 1;
      ids_load(i + 6, S0{});
      stage_write(i + 3, wb, S1{});
      stage_load(i + 5, S1{});
      sb_sync();
      wb = (wb == 2) ? 0 : wb + 1;
    }
    for (; i < nf; ++i) {                               // the last (up to three) steps
      if (i + 2 < nf) {
        if (i & 1) stage_write(i + 2, wb, S1{}); else stage_write(i + 2, wb, S0{});
      }
      sb_sync();
      wb = (wb == 2) ? 0 : wb + 1;
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (!((rowok >> u) & 1u)) continue;
      const int64_t b = b0 + srow + u * 16;
      if (KS == 1) {
        float4 p;
        p.x = 0.5f * (S[u].x * S[u].x - Q[u].x);
        p.y = 0.5f * (S[u].y * S[u].y - Q[u].y);
        p.z = 0.5f * (S[u].z * S[u].z - Q[u].z);
        p.w = 0.5f * (S[u].w * S[u].w - Q[u].w);
        st4(pair + b * KD + c4, p);
        if (fsum != nullptr) st4(fsum + b * KD + c4, S[u]);
      } else {
        st4(Sp + b * KD + c4, S[u]);
        st4(Qp + b * KD + c4, Q[u]);
      }
    }
  }
}

// z1 = bias + sum_y z1p[y];  fsum = sum_y Sp[y];  pair = (fsum^2 - sum_y Qp[y]) / 2      (y in ascending order)
__global__ __launch_bounds__(kBlock) void l1_sb_combine_kernel(const float* __restrict__ ws, int KS, int64_t B,
                                                               const float* __restrict__ bias, float* __restrict__ z1,
                                                               float* __restrict__ pair, float* __restrict__ fsum) {
  constexpr int KD = SB_KD, H1 = SB_H1, ZQ = H1 / 4, SQ = KD / 4;
  const float* Sp = ws + static_cast<int64_t>(KS) * B * H1;
  const float* Qp = Sp + static_cast<int64_t>(KS) * B * KD;
  const int64_t total = B * (ZQ + SQ);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int64_t b = q / (ZQ + SQ);
    const int c = static_cast<int>(q % (ZQ + SQ));
    if (c < ZQ) {
      float4 a = bias != nullptr ? ld4(bias + c * 4) : f4_zero();
      for (int y = 0; y < KS; ++y) a = f4_add(a, ld4(ws + (static_cast<int64_t>(y) * B + b) * H1 + c * 4));
      st4(z1 + b * H1 + c * 4, a);
    } else {
      const int k4 = (c - ZQ) * 4;
      float4 s = f4_zero(), qq = f4_zero();
      for (int y = 0; y < KS; ++y) {
        s = f4_add(s, ld4(Sp + (static_cast<int64_t>(y) * B + b) * KD + k4));
        qq = f4_add(qq, ld4(Qp + (static_cast<int64_t>(y) * B + b) * KD + k4));
      }
      float4 p;
      p.x = 0.5f * (s.x * s.x - qq.x);
      p.y = 0.5f * (s.y * s.y - qq.y);
      p.z = 0.5f * (s.z * s.z - qq.z);
      p.w = 0.5f * (s.w * s.w - qq.w);
      st4(pair + b * KD + k4, p);
      if (fsum != nullptr) st4(fsum + b * KD + k4, s);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// Row gradients in run order.  grid = (ceil(B / 128), KS): workgroup (tile, y) writes the rows of its 128 samples for
// the fields [y F / KS, (y + 1) F / KS) — no sum across fields, so the split costs nothing.
//   wave w: sample tile w & 3, dim tile w >> 2 (eight multiplying waves, two per SIMD).
//   A operand = the wave's gz fragments (8 k-blocks x 3 planes, split once, resident in 96 VGPRs),
//   B operand = the field's weight planes out of a two-buffer LDS ring (48 KB per field, staged one field ahead through
//   registers — LDS-direct loads measured slower here: they force a drain of the whole memory counter, stores included, in
//   front of every barrier).
//   An accumulator register holds ONE sample's value of the lane's dim: a store instruction writes the 128 contiguous bytes
//   of a (row, dim tile) per half wave — two full-line requests per instruction.  (The transposed product — four
//   consecutive dims of one sample per lane, 16-byte stores — writes 32 different lines per instruction: measured 0.37 ms,
//   the store path of a CU takes ~4 cycles per line request and all eight waves stored at once behind the barrier.)
//   Two accumulator sets: field f accumulates into one while the other (field f - 1) is stored, two rows per MFMA group —
//   the stores issue in the shadow of the chain and stay in flight across the barrier.
//   The slots of a field (128 per tile) ride with the planes through LDS.
// -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SB_THREADS, 2) void l1_dgrad_sb_kernel(
    const float* __restrict__ gz, const bf16x8* __restrict__ WsbB, int F, int64_t B, const float* __restrict__ gl,
    const float* __restrict__ wp, const float* __restrict__ fsum, const int32_t* __restrict__ slotT, float* __restrict__ ge) {
  constexpr int KD = SB_KD, H1 = SB_H1, KBH = H1 / 16, NT = KD / 32;
  constexpr int WF = KBH * NT * 3 * 64;                 // 16-byte slots of one field's planes (3,072 = 48 KB)
  constexpr int NWL = WF / SB_THREADS;                  // slots per thread and field (6)
  constexpr int BUF = WF * 16 + 128 * 4;                // bytes of one ring buffer: planes + the tile's 128 slots
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int tid = threadIdx.x;
  const int j = lane & 31, g = lane >> 5;
  const int rt = wid & 3, nt = wid >> 2;
  const int KS = gridDim.y, y = blockIdx.y;
  const int f_lo = static_cast<int>(static_cast<int64_t>(F) * y / KS);
  const int f_hi = static_cast<int>(static_cast<int64_t>(F) * (y + 1) / KS);
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * 128;

  // staging of one field: NWL x 16 bytes of planes per thread + one slot for threads 0 .. 127
  u32x4 pw[NWL];
  int32_t psl = -1;
  const bool sl_mine = tid < 128 && b0 + tid < B;
  auto stage_load = [&](int f) {
    const bf16x8* src = WsbB + static_cast<int64_t>(f) * WF + tid;
#pragma unroll
    for (int u = 0; u < NWL; ++u) pw[u] = *reinterpret_cast<const u32x4*>(src + u * SB_THREADS);
    psl = sl_mine ? slotT[static_cast<int64_t>(f) * B + b0 + tid] : -1;
  };
  auto stage_write = [&](int buf) {
    char* d = smem + buf * BUF;
#pragma unroll
    for (int u = 0; u < NWL; ++u) *reinterpret_cast<u32x4*>(d + (tid + u * SB_THREADS) * 16) = pw[u];
    if (tid < 128) *reinterpret_cast<int32_t*>(d + WF * 16 + tid * 4) = psl;
  };
  if (f_lo < f_hi) stage_load(f_lo);

  // gz fragments of this lane's sample (A operand: lane = sample 32 rt + j, columns 16 kb + 8 g .. + 7)
  bf16x8 g1[KBH], g2[KBH], g3[KBH];
  {
    const int64_t smp = b0 + rt * 32 + j;
    const bool s_ok = smp < B;
    const float* p0 = gz + (s_ok ? smp : B - 1) * H1 + 8 * g;
#pragma unroll
    for (int kb = 0; kb < KBH; ++kb) {
      float4 lo = ld4(p0 + kb * 16), hi = ld4(p0 + kb * 16 + 4);
      if (!s_ok) { lo = f4_zero(); hi = f4_zero(); }
      split8(lo, hi, g1[kb], g2[kb], g3[kb]);
    }
  }
  // FM term of this lane's 16 accumulator registers (constant over the fields): sample 32 rt + 8 q + 4 g + i (r = 4 q + i)
  // of the tile, dim 32 nt + j
  float fm[16];
  {
    const int dim = nt * 32 + j;
    const float wv = (wp != nullptr && fsum != nullptr && gl != nullptr) ? wp[dim] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t smp = b0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      fm[r] = (wv != 0.f && smp < B) ? gl[smp] * wv * fsum[smp * KD + dim] : 0.f;
    }
  }
  const uint32_t spare = static_cast<uint32_t>(B * F);
  const uint32_t dim_off = static_cast<uint32_t>((nt * 32 + j) * 4);
  uint32_t dst[16];                                     // BYTE offsets of the rows the pending accumulator set goes to
  auto slots_read = [&](int buf) {                      // the tile's slots of the field in `buf` -> dst (dropped positions -> spare row)
    const int32_t* sp = reinterpret_cast<const int32_t*>(smem + buf * BUF + WF * 16) + rt * 32 + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = *reinterpret_cast<const int4*>(sp + 8 * q);
      dst[4 * q + 0] = (v.x >= 0 ? static_cast<uint32_t>(v.x) : spare) * (KD * 4) + dim_off;
      dst[4 * q + 1] = (v.y >= 0 ? static_cast<uint32_t>(v.y) : spare) * (KD * 4) + dim_off;
      dst[4 * q + 2] = (v.z >= 0 ? static_cast<uint32_t>(v.z) : spare) * (KD * 4) + dim_off;
      dst[4 * q + 3] = (v.w >= 0 ? static_cast<uint32_t>(v.w) : spare) * (KD * 4) + dim_off;
    }
  };
  auto store2 = [&](const f32x16& a, int r0) {          // two rows of the pending set
#pragma unroll
    for (int r = r0; r < r0 + 2; ++r) {
      const float v = a[r] + fm[r];
      if (kSbAblate & 4) asm volatile("" :: "v"(v), "v"(dst[r]));
      else *reinterpret_cast<float*>(reinterpret_cast<char*>(ge) + dst[r]) = v;
    }
  };
  // one field: its chain into `acc`, the pending set `prev` stored two rows per k-block
  auto chain = [&](int buf, f32x16& acc, const f32x16& prev, auto has_prev) {
    const bf16x8* wr = reinterpret_cast<const bf16x8*>(smem + buf * BUF) + (nt * 3) * 64 + lane;
    acc = acc0();
    bf16x8 wf[2][3];                                    // the planes of a k-block are read one block ahead of their MFMAs
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[0][p] = wr[p * 64];
#pragma unroll
    for (int kb = 0; kb < KBH; ++kb) {
      if (kb + 1 < KBH) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (!(kSbAblate & 16)) wf[(kb + 1) & 1][p] = wr[((kb + 1) * NT * 3 + p) * 64];
      }
      mfma6(acc, g1[kb], g2[kb], g3[kb], wf[kb & 1][0], wf[kb & 1][1], wf[kb & 1][2]);
      if constexpr (decltype(has_prev)::value) store2(prev, 2 * kb);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (f_lo < f_hi) stage_write(0);
  sb_sync();

  f32x16 accA = acc0(), accB = acc0();
  // step f: planes / slots of field f + 1 requested; chain of field f (stores of field f - 1 in its shadow); slots of field f
  // read (all reads of the buffer are in front of the barrier); field f + 1 written into the other buffer; barrier.
  auto step = [&](int f, int buf, f32x16& acc, const f32x16& prev, auto has_prev) {
    if (f + 1 < f_hi && !(kSbAblate & 8)) stage_load(f + 1);
    __builtin_amdgcn_sched_barrier(0);
    chain(buf, acc, prev, has_prev);
    slots_read(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (f + 1 < f_hi && !(kSbAblate & 8)) stage_write(buf ^ 1);
    sb_sync();
  };
  int f = f_lo;
  if (f < f_hi) {
    step(f, 0, accA, accB, std::false_type{});
    ++f;
    for (; f + 1 < f_hi; f += 2) {
      step(f, 1, accB, accA, std::true_type{});
      step(f + 1, 0, accA, accB, std::true_type{});
    }
    if (f < f_hi) {                                     // (odd count: one more step, then its own rows)
      step(f, 1, accB, accA, std::true_type{});
#pragma unroll
      for (int r = 0; r < 16; r += 2) store2(accB, r);
    } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) store2(accA, r);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// Weight gradient.  grid = ceil(F / FG) * n_chunks; workgroup (field group, chunk) walks the chunk's samples in stages of
// 32 (two 16-sample MFMA slabs).  partial[ch][f*K + i][n] = sum over the chunk of table[idxT[f, b], i] * gz[b, n].
//   A operand = gathered rows, TRANSPOSED (lane = embedding dim, 8 consecutive samples per lane): the rows lie in LDS as
//   they come ([sample][K] f32), a lane reads its 8 values with ds_read_b32 (32 lanes contiguous: conflict-free) and splits
//   them in registers — 44 VALU per fragment, feeding 24 MFMAs.
//   B operand = gz planes from lr_deepfm_l1_sb_gz_pack (fragment order: copied as they are).
//   waves 0-3 multiply: wave c owns (field q, dim tile mt) = (c >> 1, c & 1) [FG = 2] or field c, both dim tiles [FG = 4],
//   against all four column tiles: 64 accumulator registers per (field, dim tile);
//   waves 4-7 stage: rows of stage s + 4 requested into one of two register sets, stage s + 2 written into a ring of three
//   row buffers; the gz planes of stage s + 2 requested, those of stage s + 1 written into a ring of two.
// -------------------------------------------------------------------------------------------------------------------------
template <int FG, int CW>
__global__ __launch_bounds__((CW + 4) * 64, (CW + 4) / 4) void l1_wgrad_sb_kernel(
    const float* __restrict__ table, int64_t V, const int32_t* __restrict__ idxT, int64_t B, int F,
    const bf16x8* __restrict__ gzp, int n_chunks, float* __restrict__ partial) {
  constexpr int KD = SB_KD, H1 = SB_H1, CT = H1 / 32;
  constexpr int TSW = 32;                               // samples per stage
  constexpr int NP = FG / 2;                            // (field, dim tile) pairs per multiplying wave
  constexpr int CTW = CT * 4 / CW;                      // column tiles per multiplying wave (CW = 8: two waves share a pair)
  static_assert((CW == 4 || CW == 8) && (CW == 4 || FG == 2), "wave layout");
  constexpr int RSZ = FG * TSW * KD;                    // floats of one row buffer
  constexpr int PSL = 2 * CT * 3 * 64;                  // 16-byte slots of one stage's gz planes (1,536 = 24 KB)
  constexpr int NLD = FG * TSW * (KD / 4) / 256;        // float4 per staging thread and stage (FG * 2)
  constexpr int NPL = PSL / 256;                        // plane slots per staging thread and stage (6)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);                             // [3][FG][TSW][KD]
  bf16x8* pl = reinterpret_cast<bf16x8*>(smem + 3 * RSZ * 4);              // [2][PSL]

  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int j = lane & 31, g = lane >> 5;
  const int fg = blockIdx.x / n_chunks, ch = blockIdx.x % n_chunks;
  const int64_t stages = ceil_div(B, TSW);
  const int64_t s_lo = stages * ch / n_chunks, s_hi = stages * (ch + 1) / n_chunks;
  const int n_st = static_cast<int>(s_hi - s_lo);

  if (wid < CW) {
    // ================================================= multiplying waves =================================================
    f32x16 acc[NP][CTW];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int c = 0; c < CTW; ++c) acc[p][c] = acc0();
    const int ct0 = CW == 8 ? (wid & 1) * CTW : 0;
    int pq[NP], pm[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      pq[p] = FG == 4 ? wid : (CW == 8 ? (wid >> 2) : (wid >> 1));
      pm[p] = FG == 4 ? p : (CW == 8 ? ((wid >> 1) & 1) : (wid & 1));
    }
    bf16x8 af[2][NP][3];
    // this lane's 8 samples of slab `sl` (0 / 1) of row buffer rb: rows 16 sl + 8 g + e, element 32 mt + j
    auto a_frag = [&](int rb, int sl, bf16x8 (&dst)[NP][3]) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float* src = rows + rb * RSZ + (pq[p] * TSW + sl * 16 + 8 * g) * KD + pm[p] * 32 + j;
        float4 lo, hi;
        if (!(kSbAblate & 16)) {
        lo.x = src[0 * KD]; lo.y = src[1 * KD]; lo.z = src[2 * KD]; lo.w = src[3 * KD];
        hi.x = src[4 * KD]; hi.y = src[5 * KD]; hi.z = src[6 * KD]; hi.w = src[7 * KD];
        }
        if (kSbAblate & 16) { lo = f4_zero(); hi = f4_zero(); asm volatile("" : "+v"(lo.x), "+v"(hi.x)); }
        split8(lo, hi, dst[p][0], dst[p][1], dst[p][2]);
      }
    };
    auto slab = [&](int pb, int sl, const bf16x8 (&a)[NP][3]) {
      const bf16x8* br = pl + pb * PSL + sl * (CT * 3 * 64) + ct0 * 3 * 64 + lane;
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const bf16x8 b1 = br[(c * 3 + 0) * 64], b2 = br[(c * 3 + 1) * 64], b3 = br[(c * 3 + 2) * 64];
#pragma unroll
        for (int p = 0; p < NP; ++p) mfma6(acc[p][c], a[p][0], a[p][1], a[p][2], b1, b2, b3);
      }
    };
    sb_sync();                                    // stages 0 / 1 staged
    if (n_st > 0) a_frag(0, 0, af[0]);
    int rb = 0, rn = 1;
    for (int s = 0; s < n_st; ++s) {
      a_frag(rb, 1, af[1]);
      slab(s & 1, 0, af[0]);
      a_frag(rn, 0, af[0]);                             // first slab of the next stage: its row buffer was completed a step ago
      slab(s & 1, 1, af[1]);
      sb_sync();
      rb = rn;
      rn = (rn == 2) ? 0 : rn + 1;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f = fg * FG + pq[p];
      if (f >= F) continue;
      float* out = partial + (static_cast<int64_t>(ch) * F + f) * KD * H1;
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[(pm[p] * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * H1 + (ct0 + c) * 32 + j] = acc[p][c][r];
    }
  } else {
    // =================================================== staging waves ===================================================
    const int tid = static_cast<int>(threadIdx.x) - CW * 64;
    const int srow = tid >> 4, c4 = (tid & 15) * 4;     // float4 u of a stage: field u >> 1, sample srow + 16 (u & 1)
    const uint32_t Vu = static_cast<uint32_t>(V);
    const int32_t* ids[NLD];
    bool f_ok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = fg * FG + (u >> 1);
      f_ok[u] = f < F;
      ids[u] = idxT + static_cast<int64_t>(f_ok[u] ? f : F - 1) * B;
    }
    float4 pre[2][NLD];
    uint32_t pre_ok[2] = {0u, 0u};
    int32_t idn[2][NLD];
    u32x4 pgz[NPL];
    const int64_t last_stage = s_hi > s_lo ? s_hi - 1 : s_lo;
    // (ids a step ahead of their rows and in front of that step's row requests: see l1_fwd_sb_kernel)
    auto ids_load = [&](int64_t st, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const int64_t stc = st < s_hi ? st : last_stage;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int64_t b = stc * TSW + srow + 16 * (u & 1);
        idn[set][u] = ids[u][b < B ? b : B - 1];
      }
    };
    auto stage_load = [&](int64_t st, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const int64_t stc = st < s_hi ? st : last_stage;
      pre_ok[set] = 0;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int64_t b = stc * TSW + srow + 16 * (u & 1);
        const bool ok = f_ok[u] && b < B && static_cast<uint32_t>(idn[set][u]) < Vu;
        const uint32_t id = (ok && !(kSbAblate & 1)) ? static_cast<uint32_t>(idn[set][u]) : 0u;
        if (ok) pre_ok[set] |= 1u << u;
        pre[set][u] = ld4(table + static_cast<uint64_t>(id) * KD + c4);
      }
    };
    auto stage_write = [&](int rbuf, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      float* dr = rows + rbuf * RSZ;
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        st4(dr + ((u >> 1) * TSW + srow + 16 * (u & 1)) * KD + c4, ((pre_ok[set] >> u) & 1u) ? pre[set][u] : f4_zero());
    };
    const int64_t slabs_all = ceil_div(B, 16);
    auto planes_load = [&](int64_t st) {                // (slabs beyond the batch hold zeros: lr_deepfm_l1_sb_gz_pack pads to 16;
      const int64_t stc = st < s_hi ? st : last_stage;  //  a stage's second slab may not exist at all: clamped, its rows are zero)
#pragma unroll
      for (int v = 0; v < NPL; ++v) {
        const int slot = tid + 256 * v;
        int64_t sb = stc * 2 + slot / (CT * 3 * 64);
        if (sb >= slabs_all) sb = slabs_all - 1;
        pgz[v] = *reinterpret_cast<const u32x4*>(gzp + sb * (CT * 3 * 64) + slot % (CT * 3 * 64));
      }
    };
    auto planes_write = [&](int pb) {
#pragma unroll
      for (int v = 0; v < NPL; ++v) *reinterpret_cast<u32x4*>(pl + pb * PSL + tid + 256 * v) = pgz[v];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    ids_load(s_lo, S0{});
    ids_load(s_lo + 1, S1{});
    planes_load(s_lo);
    stage_load(s_lo, S0{});
    stage_load(s_lo + 1, S1{});
    ids_load(s_lo + 2, S0{});
    ids_load(s_lo + 3, S1{});
    stage_write(0, S0{});
    stage_write(1, S1{});
    planes_write(0);
    planes_load(s_lo + 1);
    stage_load(s_lo + 2, S0{});
    stage_load(s_lo + 3, S1{});
    ids_load(s_lo + 4, S0{});
    sb_sync();
    // step s: ids of stage s + 5 requested; rows of stage s + 2 -> row buffer (s + 2) % 3 from set s & 1; planes of stage
    // s + 1 -> plane buffer (s + 1) & 1, planes of stage s + 2 requested; rows of stage s + 4 requested into set s & 1.
    // (Stages beyond the chunk are clamped re-reads that nobody multiplies.)
    int wb = 2;
    int s = 0;
    for (; s + 1 < n_st; s += 2) {
      ids_load(s_lo + s + 5, S1{});
      stage_write(wb, S0{});
      planes_write(1);
      planes_load(s_lo + s + 2);
      stage_load(s_lo + s + 4, S0{});
      sb_sync();
      wb = (wb == 2) ? 0 : wb + 1;
      ids_load(s_lo + s + 6, S0{});
      stage_write(wb, S1{});
      planes_write(0);
      planes_load(s_lo + s + 3);
      stage_load(s_lo + s + 5, S1{});
      sb_sync();
      wb = (wb == 2) ? 0 : wb + 1;
    }
    if (s < n_st) {
      stage_write(wb, S0{});
      planes_write(1);
      sb_sync();
    }
  }
}

}  // namespace lr

using namespace lr;

static inline bool sb_al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
template <typename Kern>
static int sb_set_lds(Kern kern, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(bytes));
  return e == hipSuccess ? LR_OK : static_cast<int>(e);
}

extern "C" int lr_deepfm_l1_sb_supported(int K, int H1) { return (K == SB_KD && H1 == SB_H1) ? 1 : 0; }
extern "C" int lr_deepfm_l1_fwd_sb_supported(int K, int H1) { return lr_deepfm_l1_sb_supported(K, H1); }

extern "C" size_t lr_deepfm_l1_sb_pack_bytes(int F, int K, int H1) {
  if (F < 1 || K < 32 || K % 32 != 0 || H1 < 32 || H1 % 32 != 0) return 0;
  return static_cast<size_t>(F) * K * H1 * 6;           // three bf16 planes
}

extern "C" int lr_deepfm_l1_sb_pack(const float* W, const float* scale, int F, int K, int H1, void* outA, void* outB,
                                    const float* red_partial, int red_nblk, float* red_out, lr_stream_t stream) {
  LR_CHECK_ARG(W && (outA || outB) && F >= 1);
  LR_CHECK_ARG((red_partial == nullptr) == (red_out == nullptr) && (red_partial == nullptr || red_nblk >= 1));
  if (lr_deepfm_l1_sb_pack_bytes(F, K, H1) == 0) return LR_ESHAPE;
  if (!sb_al16(outA) || !sb_al16(outB)) return LR_EINVAL;
  const int64_t total = static_cast<int64_t>(F) * K * H1 / 8;
  const int n_red = red_partial != nullptr ? (H1 + 15) / 16 : 0;
  hipLaunchKernelGGL(l1_sb_pack_kernel, dim3(grid_for(total, kBlock) + n_red), dim3(kBlock), 0, as_stream(stream), W, scale, F, K,
                     H1, static_cast<bf16x8*>(outA), static_cast<bf16x8*>(outB), red_partial, red_nblk, red_out);
  return launch_status();
}

extern "C" size_t lr_deepfm_l1_sb_gz_pack_bytes(int64_t B, int H1) {
  if (B < 0 || H1 < 32 || H1 % 32 != 0) return 0;
  return static_cast<size_t>(ceil_div(B, 16)) * 16 * H1 * 6;
}

extern "C" int lr_deepfm_l1_sb_gz_pack(const float* gz, int64_t B, int H1, void* out, lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(gz && out);
  if (H1 < 32 || H1 % 32 != 0) return LR_ESHAPE;
  if (!sb_al16(out)) return LR_EINVAL;
  const int64_t total = ceil_div(B, 16) * (H1 / 32) * 64;
  hipLaunchKernelGGL(l1_sb_gz_pack_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, as_stream(stream), gz, B, H1,
                     static_cast<bf16x8*>(out));
  return launch_status();
}

// ---- profiling / test switches (same results in every mode) ------------------------------------------------------------
static int g_sb_fwd_tile = 0;      // 0: automatic; 64 / 128: samples per workgroup of the forward
static int g_sb_ksplit = 0;        // 0: automatic; >= 1: field groups of the forward / row-gradient grids
static int g_sb_wgrad_cw = 0;      // 0: automatic (8 with 2 fields per workgroup); 4 / 8: multiplying waves of the weight gradient
static int g_sb_wgrad_fg = 0;      // 0: automatic (2); 2 / 4: fields per workgroup of the weight gradient
extern "C" void lr_deepfm_l1_sb_override(int fwd_tile, int ksplit, int wgrad_cw, int wgrad_fg) {
  g_sb_fwd_tile = (fwd_tile == 64 || fwd_tile == 128) ? fwd_tile : 0;
  g_sb_ksplit = ksplit > 0 ? ksplit : 0;
  g_sb_wgrad_cw = (wgrad_cw == 4 || wgrad_cw == 8) ? wgrad_cw : 0;
  g_sb_wgrad_fg = (wgrad_fg == 2 || wgrad_fg == 4) ? wgrad_fg : 0;
}

// field groups of the row-gradient grid: two rounds of one workgroup per CU (measured on cfg 2: 0.35 ms with 512 workgroups
// against 0.41 ms with 256 — a workgroup's prologue (its gz tile split into planes) overlaps other workgroups' chains)
static int sb_ksplit(int64_t tiles, int F) {
  if (g_sb_ksplit > 0) return g_sb_ksplit < F ? g_sb_ksplit : F;
  int ks = 1;
  while (ks < 8 && tiles * ks * 2 <= 2 * kNumCU && ks * 2 <= F) ks *= 2;
  return ks;
}
// forward: 128-sample tiles (the weight planes are re-read once per 128 samples), field groups so that the grid fills the
// chip once; 64-sample tiles (two workgroups per CU at 128 VGPRs) measured the same on cfg 2 and stay selectable
static int sb_fwd_tile(int64_t B) {
  if (g_sb_fwd_tile) return g_sb_fwd_tile;
  return 128;
}
static int sb_fwd_ksplit(int64_t tiles, int F, int ts) {
  if (g_sb_ksplit > 0) return g_sb_ksplit < F ? g_sb_ksplit : F;
  const int64_t target = ts == 64 ? 2 * kNumCU : kNumCU;
  int ks = 1;
  while (ks < 8 && tiles * ks * 2 <= target && ks * 2 <= F) ks *= 2;
  return ks;
}

extern "C" size_t lr_deepfm_l1_fwd_sb_ws_bytes(int64_t B, int F) {
  if (B <= 0 || F < 1) return 0;
  const int ts = sb_fwd_tile(B);
  const int ks = sb_fwd_ksplit(ceil_div(B, ts), F, ts);   // (a test that overrides the split asks again)
  return static_cast<size_t>(ks) * static_cast<size_t>(B) * (SB_H1 + 2 * SB_KD) * 4;
}

extern "C" int lr_deepfm_l1_fwd_sb_f32(const float* table, const float* lin, int64_t V, int K, const int32_t* idx, int64_t B,
                                       int F, const void* Wsb, const float* bias, int H1, float* z1, float* pair, float* fsum,
                                       float* lin_out, void* ws, size_t ws_bytes, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 1 && B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && Wsb && z1 && pair);
  LR_CHECK_ARG((lin == nullptr) == (lin_out == nullptr));
  if (!lr_deepfm_l1_sb_supported(K, H1)) return LR_ESHAPE;
  for (const void* p : {static_cast<const void*>(table), Wsb, static_cast<const void*>(z1), static_cast<const void*>(pair),
                        static_cast<const void*>(fsum), static_cast<const void*>(bias), static_cast<const void*>(ws)})
    if (!sb_al16(p)) return LR_EINVAL;
  const int ts = sb_fwd_tile(B);
  const int64_t tiles = ceil_div(B, ts);
  const int ks = sb_fwd_ksplit(tiles, F, ts);
  if (ks > 1) {
    const size_t need = static_cast<size_t>(ks) * static_cast<size_t>(B) * (SB_H1 + 2 * SB_KD) * 4;
    if (ws == nullptr || ws_bytes < need) return LR_EINVAL;
  }
  hipStream_t s = as_stream(stream);
  const size_t lds = static_cast<size_t>(3) * (ts / 32) * (SB_KD / 16) * 3 * 64 * 16;
  const dim3 grid(static_cast<unsigned>(tiles), static_cast<unsigned>(ks));
  auto launch = [&](auto kern) -> int {
    int rc = sb_set_lds(kern, lds);
    if (rc != LR_OK) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(SB_THREADS), lds, s, table, lin, V, idx, B, F, static_cast<const bf16x8*>(Wsb), bias,
                       z1, pair, fsum, lin_out, static_cast<float*>(ws));
    return launch_status();
  };
  int rc;
  if (ts == 128) rc = lin != nullptr ? launch(l1_fwd_sb_kernel<128, true>) : launch(l1_fwd_sb_kernel<128, false>);
  else rc = lin != nullptr ? launch(l1_fwd_sb_kernel<64, true>) : launch(l1_fwd_sb_kernel<64, false>);
  if (rc != LR_OK || ks == 1) return rc;
  const int64_t total = B * ((SB_H1 + SB_KD) / 4);
  hipLaunchKernelGGL(l1_sb_combine_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, s, static_cast<const float*>(ws), ks,
                     B, bias, z1, pair, fsum);
  return launch_status();
}

extern "C" int lr_deepfm_l1_dgrad_sb_f32(const float* gz, int H1, const void* WsbB, int K, int F, int64_t B, const float* gl,
                                         const float* wp, const float* fsum, const int32_t* slotT, float* ge,
                                         lr_stream_t stream) {
  LR_CHECK_ARG(B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(gz && WsbB && slotT && ge);
  if (!lr_deepfm_l1_sb_supported(K, H1)) return LR_ESHAPE;
  if ((B * static_cast<int64_t>(F) + 1) * SB_KD * 4 > (static_cast<int64_t>(1) << 32)) return LR_EINVAL;   // 32-bit byte offsets into ge
  for (const void* p : {static_cast<const void*>(gz), WsbB, static_cast<const void*>(wp), static_cast<const void*>(fsum),
                        static_cast<const void*>(ge)})
    if (!sb_al16(p)) return LR_EINVAL;
  const int64_t tiles = ceil_div(B, 128);
  const int ks = sb_ksplit(tiles, F);
  const size_t lds = static_cast<size_t>(2) * ((SB_H1 / 16) * (SB_KD / 32) * 3 * 64 * 16 + 128 * 4);
  const dim3 grid(static_cast<unsigned>(tiles), static_cast<unsigned>(ks));
  int rc = sb_set_lds(l1_dgrad_sb_kernel, lds);
  if (rc != LR_OK) return rc;
  hipLaunchKernelGGL(l1_dgrad_sb_kernel, grid, dim3(SB_THREADS), lds, as_stream(stream), gz, static_cast<const bf16x8*>(WsbB), F, B,
                     gl, wp, fsum, slotT, ge);
  return launch_status();
}

// four fields per workgroup: the gz planes of a stage feed twice the MFMAs (0.30 ms against 0.34 ms with two on cfg 2)
static int sb_wgrad_fg() { return g_sb_wgrad_fg ? g_sb_wgrad_fg : 4; }

extern "C" int lr_deepfm_l1_wgrad_sb_chunks(int64_t B, int F) {
  if (B <= 0 || F < 1) return 1;
  const int64_t groups = ceil_div(F, sb_wgrad_fg());
  const int64_t stages = ceil_div(B, 32);
  // two rounds of one workgroup per CU (FG = 2) or one (FG = 4), at least 8 stages per workgroup, at most 64 chunks
  int64_t c = (sb_wgrad_fg() == 2 ? 2 * kNumCU : kNumCU) / groups;
  if (c * 8 > stages) c = stages / 8;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return static_cast<int>(c);
}

extern "C" int lr_deepfm_l1_wgrad_sb_f32(const float* table, int64_t V, int K, const int32_t* idxT, int64_t B, int F,
                                         const void* gzp, int H1, int n_chunks, float* partial, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 1 && B >= 1 && F >= 1 && n_chunks >= 1);
  LR_CHECK_ARG(table && idxT && gzp && partial);
  if (!lr_deepfm_l1_sb_supported(K, H1)) return LR_ESHAPE;
  if (!sb_al16(table) || !sb_al16(gzp)) return LR_EINVAL;
  const int fg = sb_wgrad_fg();
  const int cw = (fg == 2 && g_sb_wgrad_cw != 4) ? 8 : 4;
  const size_t lds = static_cast<size_t>(3) * fg * 32 * SB_KD * 4 + static_cast<size_t>(2) * 2 * (SB_H1 / 32) * 3 * 64 * 16;
  const dim3 grid(static_cast<unsigned>(ceil_div(F, fg) * n_chunks));
  auto launch = [&](auto kern) -> int {
    int rc = sb_set_lds(kern, lds);
    if (rc != LR_OK) return rc;
    hipLaunchKernelGGL(kern, grid, dim3((cw + 4) * 64), lds, as_stream(stream), table, V, idxT, B, F,
                       static_cast<const bf16x8*>(gzp), n_chunks, partial);
    return launch_status();
  };
  if (fg == 4) return launch(l1_wgrad_sb_kernel<4, 4>);
  return cw == 8 ? launch(l1_wgrad_sb_kernel<2, 8>) : launch(l1_wgrad_sb_kernel<2, 4>);
}
