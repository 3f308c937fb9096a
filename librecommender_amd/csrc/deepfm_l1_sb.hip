// EXPERIMENTAL, opt-in (round 4): the first layer's forward contraction as SPLIT-bf16 MFMA products.
//
//   z1[b, :] = sum_f table[idx[b, f], :] @ Wp[f*K:(f+1)*K, :] + bias          (+ fsum / pair / lin_out as lr_deepfm_l1_fwd_f32)
//
// Every f32 operand is split exactly into three bf16 values (x = x1 + x2 + x3, round-to-nearest-even each step) and a product
// a*b is taken as the six largest of the nine cross terms, a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1, each an exact
// bf16 x bf16 product accumulated in f32 by v_mfma_f32_32x32x16_bf16 (smallest terms first).  Measured on this layer's own
// reduction (K = 12,928) the result is as close to f64 as the f32 fma chain of lr_deepfm_l1_fwd_f32 (relative rms error 1.88e-6
// vs 2.08e-6, profiles/r04_bf16_split_probe.txt); it is NOT bit-identical to it, which is why nothing selects this kernel by
// default.  Why bother: a bf16 MFMA does 32,768 FLOP in 32 cycles and hides up to ~5 VALU instructions behind it, an f32 MFMA
// does 4,096 FLOP in 64 cycles and hides none — six bf16 MFMAs per 32 x 32 x 16 block are 192 cycles against 512.
//
// Layout.  Weights are packed once per step by lr_deepfm_l1_sb_pack into MFMA fragment order, three bf16 planes:
//   Wsb[f][kb = K/16][ct = H1/32][plane 3][lane 64][8 bf16]      lane (j = lane & 31, g = lane >> 5), element e:
//                                                                row f*K + kb*16 + 8*g + e, column ct*32 + j
// (the k index of an operand element only has to be the same function of (g, e) for A and B).
// (Kernel layout: see l1_fwd_sb_kernel.)
#include <type_traits>

#include "common.hpp"

namespace lr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ void split3(f32x8 x, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  a1 = __builtin_convertvector(x, bf16x8);
  const f32x8 r1 = x - __builtin_convertvector(a1, f32x8);
  a2 = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(a2, f32x8);
  a3 = __builtin_convertvector(r2, bf16x8);
}

// ---- pack: W [F*K, H1] (optionally row-scaled: the BatchNorm fold) -> three bf16 planes in fragment order --------------
__global__ __launch_bounds__(kBlock) void l1_sb_pack_kernel(const float* __restrict__ W, const float* __restrict__ scale, int F,
                                                            int K, int H1, bf16x8* __restrict__ out) {
  const int KB = K / 16, CT = H1 / 32;
  const int64_t total = static_cast<int64_t>(F) * KB * CT * 64;             // one (f, kb, ct, lane) per thread: 3 x 16 bytes
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < total; q += stride) {
    const int lane = static_cast<int>(q & 63);
    const int j = lane & 31, g = lane >> 5;
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
    bf16x8* dst = out + (((static_cast<int64_t>(f) * KB + kb) * CT + ct) * 3) * 64 + lane;
    dst[0] = p1;
    dst[64] = p2;
    dst[128] = p3;
  }
}

// ---- forward ------------------------------------------------------------------------------------------------------------
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
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

// Workgroup = 64 samples x all H1 columns, NW waves (8: one 32 x 32 tile each, two waves per SIMD; 4: two column tiles each).
// Per field both operands go through LDS as bf16 planes in MFMA fragment order, double buffered, requested one field ahead
// (ids two ahead: the rows' addresses never wait for a load issued in the same step, see deepfm_l1.hip):
//   A: the 64 gathered rows, split ONCE by the threads that stage them (16 f32 per thread and field) — the MFMA phase then
//      issues nothing but ds_read_b128 and MFMAs;   [2 row tiles][K/16][3 planes][64 lanes][8 bf16]  = 24 KB
//   B: the field's weight planes from lr_deepfm_l1_sb_pack, copied as they are                        = 48 KB
// kBDirect: the weight planes do not go through LDS — every wave requests the fragments of ITS column tile(s) for the next field
// straight into registers at the top of a field's step (LDS then carries the A planes only: 120 instead of 264 KB per field).
// kSpec (8 waves): waves 0-3 only multiply (two column tiles each, one wave per SIMD), waves 4-7 only stage (gather, split, LDS
// writes, the next field's requests): with every wave doing both, the per-field barrier put all waves into the SAME phase and a
// field cost MFMA time + VALU time + load-issue time; a staging wave's VALU / VMEM work hides behind its SIMD's bf16 MFMAs.
// TS_ = 128 (8 waves, kBDirect): 128 samples per workgroup, every wave two sample tiles x one column tile — a weight fragment
// feeds two MFMAs and the planes are re-read once per 128 samples instead of 64.
template <int KD, int H1, bool kLin, int NW, bool kBDirect, bool kSpec = false, int TS_ = 64>
__global__ __launch_bounds__(NW * 64, 1) void l1_fwd_sb_kernel(
    const float* __restrict__ table, const float* __restrict__ lin, int64_t V, const int32_t* __restrict__ idx, int64_t B,
    int F, const bf16x8* __restrict__ Wsb, const float* __restrict__ bias, float* __restrict__ z1,
    float* __restrict__ pair, float* __restrict__ fsum, float* __restrict__ lin_out) {
  constexpr int NT = kSpec ? 256 : NW * 64;            // staging threads
  constexpr int TS = TS_, CPR = KD / 4, RPP = NT / CPR, NLD = TS / RPP;
  constexpr int RPW = TS / 64;                         // sample tiles per multiplying wave
  constexpr int KB = KD / 16, CT = H1 / 32;
  constexpr int CPW = kSpec ? 2 : 2 * CT / NW;         // column tiles per multiplying wave
  constexpr int AF = (TS / 32) * KB * 3 * 64;          // 16-byte slots of one field's A planes (TS / 32 row tiles)
  constexpr int WF = KB * CT * 3 * 64;                 // ... of one field's weight planes
  constexpr int NWL = WF / NT;
  static_assert(CT == 4 && (NW == 4 || NW == 8) && WF % NT == 0 && TS % RPP == 0 && (!kSpec || (NW == 8 && !kBDirect)) &&
                    (TS == 64 || (TS == 128 && NW == 8 && kBDirect && !kSpec)), "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* al = reinterpret_cast<bf16x8*>(smem);                                    // [2][AF]
  bf16x8* wl = al + 2 * AF;                                                        // [2][WF]

  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_comp = !kSpec || wid < 4, is_stage = !kSpec || wid >= 4;           // wave-uniform roles
  const int tid = kSpec ? static_cast<int>(threadIdx.x) - (wid >= 4 ? 256 : 0) : static_cast<int>(threadIdx.x);   // index among the staging threads
  const int j = lane & 31, g = lane >> 5;
  const int rt = (wid & 1) * RPW, ct0 = (kSpec ? (wid & 3) >> 1 : wid >> 1) * CPW;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * TS;
  const int nb = (B - b0) < TS ? static_cast<int>(B - b0) : TS;
  const int srow = tid / CPR, c4 = (tid % CPR) * 4;
  const uint32_t Vu = static_cast<uint32_t>(V);
  // where this thread's four floats of a row go inside an A plane (bytes from the plane-0 slot of (row tile, kb)):
  // k = c4 .. c4 + 3 -> kb = c4 / 16, lane half (c4 / 8) & 1, elements c4 & 7 ..
  uint32_t a_off[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int r = srow + u * RPP;
    const int kb_ = c4 >> 4, g_ = (c4 >> 3) & 1;
    // (the slot of row j inside a 32-lane half is rotated by 8 kb + 4 g: the 16 chunks of one row, written by 16 adjacent lanes,
    // would otherwise all start at multiples of 256 bytes — an 8-way bank conflict on every write, PMC: half of all LDS cycles)
    a_off[u] = static_cast<uint32_t>(((((r >> 5) * KB + kb_) * 3) * 64 + g_ * 32 + (((r & 31) + 8 * kb_ + 4 * g_) & 31)) * 16 + (c4 & 7) * 2);
  }

  float4 pre[NLD];
  float prel[NLD];
  uint32_t pre_ok = 0;
  int32_t idn[NLD];
  bf16x8 pw[kBDirect ? 1 : NWL];
  bf16x8 bq[kBDirect ? 2 : 1][kBDirect ? KB * CPW * 3 : 1];      // [field parity][kb, column tile, plane]
  float4 S[NLD], Q[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) { S[u] = f4_zero(); Q[u] = f4_zero(); prel[u] = 0.f; }

  auto ids_load = [&](int f) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int r = srow + u * RPP;
      idn[u] = (r < nb && f < F) ? idx[(b0 + r) * F + f] : -1;
    }
  };
  auto stage_load = [&](int f) {                  // rows (ids requested one call earlier) + weight planes of field f
    pre_ok = 0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = static_cast<uint32_t>(idn[u]) < Vu;
      const uint32_t id = ok ? static_cast<uint32_t>(idn[u]) : 0u;
      if (ok) pre_ok |= 1u << u;
      pre[u] = ld4(table + static_cast<uint64_t>(id) * KD + c4);
      if (kLin) prel[u] = lin[id];
    }
    ids_load(f + 1);
    if (!kBDirect) {
      const bf16x8* src = Wsb + static_cast<int64_t>(f) * WF + tid;
#pragma unroll
      for (int u = 0; u < NWL; ++u) pw[u] = src[u * NT];
    }
  };
  auto b_req = [&](int f, auto set_c) {           // kBDirect: this wave's weight fragments of field f -> register set
    constexpr int P = decltype(set_c)::value;
    const bf16x8* src = Wsb + static_cast<int64_t>(f) * WF + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) bq[kBDirect ? P : 0][kBDirect ? (kb * CPW + c) * 3 + q : 0] = src[((kb * CT + ct0 + c) * 3 + q) * 64];
  };
  auto stage_write = [&](int f) {                 // registers -> LDS buffers f & 1: rows split into planes; FM sums; linear weights out
    char* da = reinterpret_cast<char*>(al + (f & 1) * AF);
    bf16x8* dw = wl + (f & 1) * WF + tid;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = (pre_ok >> u) & 1u;
      const float4 x = ok ? pre[u] : f4_zero();
      S[u] = f4_add(S[u], x);
      Q[u] = f4_fma(x, x, Q[u]);
      uint2 p1, p2, p3;
      split4(x, p1, p2, p3);
      *reinterpret_cast<uint2*>(da + a_off[u]) = p1;
      *reinterpret_cast<uint2*>(da + a_off[u] + 1024) = p2;
      *reinterpret_cast<uint2*>(da + a_off[u] + 2048) = p3;
      if (kLin && c4 == 0 && srow + u * RPP < nb) lin_out[(b0 + srow + u * RPP) * F + f] = ok ? prel[u] : 0.f;
    }
    if (!kBDirect) {
#pragma unroll
      for (int u = 0; u < NWL; ++u) dw[u * NT] = pw[u];
    }
  };

  f32x16 acc[RPW * CPW];
#pragma unroll
  for (int c = 0; c < RPW * CPW; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (is_stage) {
    ids_load(0);
    stage_load(0);
    stage_write(0);
    if (F > 1) stage_load(1);
  }
  if (kBDirect) b_req(0, I0{});
  __syncthreads();
  auto step = [&](int f, auto p_c) {              // P = f & 1 at compile time (register set of the field's weight fragments)
    constexpr int P = decltype(p_c)::value;
    using PN = std::integral_constant<int, 1 - P>;
    if (kBDirect && f + 1 < F) b_req(f + 1, PN{});
    const bf16x8* ar = al + (f & 1) * AF + rt * (KB * 3 * 64) + g * 32;
    const bf16x8* wr = wl + (f & 1) * WF + lane;
    if (is_comp) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int sl = (j + 8 * kb + 4 * g) & 31;            // the rotated slot of this lane's row (see a_off)
      bf16x8 a1[RPW], a2[RPW], a3[RPW];
#pragma unroll
      for (int t = 0; t < RPW; ++t) {
        const bf16x8* at = ar + t * (KB * 3 * 64);
        a1[t] = at[(kb * 3 + 0) * 64 + sl]; a2[t] = at[(kb * 3 + 1) * 64 + sl]; a3[t] = at[(kb * 3 + 2) * 64 + sl];
      }
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        bf16x8 b1, b2, b3;
        if (kBDirect) {
          b1 = bq[kBDirect ? P : 0][kBDirect ? (kb * CPW + c) * 3 + 0 : 0];
          b2 = bq[kBDirect ? P : 0][kBDirect ? (kb * CPW + c) * 3 + 1 : 0];
          b3 = bq[kBDirect ? P : 0][kBDirect ? (kb * CPW + c) * 3 + 2 : 0];
        } else {
          const bf16x8* wp = wr + ((kb * CT + ct0 + c) * 3) * 64;
          b1 = wp[0]; b2 = wp[64]; b3 = wp[128];
        }
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
          f32x16& A_ = acc[t * CPW + c];
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[t], b1, A_, 0, 0, 0);      // smallest terms first
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b3, A_, 0, 0, 0);
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], b2, A_, 0, 0, 0);
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], b1, A_, 0, 0, 0);
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b2, A_, 0, 0, 0);
          A_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b1, A_, 0, 0, 0);
        }
      }
    }
    }
    if (is_stage && f + 1 < F) {
      stage_write(f + 1);                          // buffers (f + 1) & 1 were last read while field f - 1 was computed: free since the last barrier
      if (f + 2 < F) stage_load(f + 2);
    }
    __syncthreads();
  };
  for (int f = 0; f < F; f += 2) {
    step(f, I0{});
    if (f + 1 < F) step(f + 1, I1{});
  }

  // ---- epilogue: z1 = acc + bias; fsum / pair from the staging threads' running sums -----------------
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (!is_comp) break;
    const int col = (ct0 + c) * 32 + j;
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < RPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int smp = (rt + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (smp < nb) z1[(b0 + smp) * H1 + col] = acc[t * CPW + c][r] + bv;
      }
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int smp = srow + u * RPP;
    if (is_stage && smp < nb) {
      float4 p;
      p.x = 0.5f * (S[u].x * S[u].x - Q[u].x);
      p.y = 0.5f * (S[u].y * S[u].y - Q[u].y);
      p.z = 0.5f * (S[u].z * S[u].z - Q[u].z);
      p.w = 0.5f * (S[u].w * S[u].w - Q[u].w);
      st4(pair + (b0 + smp) * KD + c4, p);
      if (fsum != nullptr) st4(fsum + (b0 + smp) * KD + c4, S[u]);
    }
  }
}

}  // namespace lr

using namespace lr;

extern "C" size_t lr_deepfm_l1_sb_pack_bytes(int F, int K, int H1) {
  if (F < 1 || K < 16 || K % 16 != 0 || H1 < 32 || H1 % 32 != 0) return 0;
  return static_cast<size_t>(F) * (K / 16) * (H1 / 32) * 3 * 64 * 16;
}

extern "C" int lr_deepfm_l1_sb_pack(const float* W, const float* scale, int F, int K, int H1, void* out, lr_stream_t stream) {
  LR_CHECK_ARG(W && out && F >= 1);
  if (lr_deepfm_l1_sb_pack_bytes(F, K, H1) == 0) return LR_ESHAPE;
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0) return LR_EINVAL;
  const int64_t total = static_cast<int64_t>(F) * (K / 16) * (H1 / 32) * 64;
  hipLaunchKernelGGL(l1_sb_pack_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, as_stream(stream), W, scale, F, K, H1,
                     static_cast<bf16x8*>(out));
  return launch_status();
}

extern "C" int lr_deepfm_l1_fwd_sb_supported(int K, int H1) { return (K == 64 && H1 == 128) ? 1 : 0; }

static int g_sb_waves = 8;
// profiling: 4 or 8 waves per workgroup; 16 + that = weight fragments straight into registers (kBDirect)
extern "C" void lr_deepfm_l1_sb_waves_override(int waves) { g_sb_waves = waves; }

extern "C" int lr_deepfm_l1_fwd_sb_f32(const float* table, const float* lin, int64_t V, int K, const int32_t* idx, int64_t B,
                                       int F, const void* Wsb, const float* bias, int H1, float* z1, float* pair, float* fsum,
                                       float* lin_out, lr_stream_t stream) {
  LR_CHECK_ARG(V >= 1 && B >= 0 && F >= 1);
  if (B == 0) return LR_OK;
  LR_CHECK_ARG(table && idx && Wsb && z1 && pair);
  LR_CHECK_ARG((lin == nullptr) == (lin_out == nullptr));
  if (!lr_deepfm_l1_fwd_sb_supported(K, H1)) return LR_ESHAPE;
  for (const void* p : {static_cast<const void*>(table), static_cast<const void*>(Wsb), static_cast<const void*>(z1),
                        static_cast<const void*>(pair), static_cast<const void*>(fsum)})
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) return LR_EINVAL;
  constexpr int KD = 64, HD = 128;
  size_t lds = static_cast<size_t>(2) * (2 * (KD / 16) * 3 * 64 * 16) + static_cast<size_t>(2) * (KD / 16) * (HD / 32) * 3 * 64 * 16;
  hipStream_t s = as_stream(stream);
  int threads = 512;
  bool grid128 = false;
  auto launch = [&](auto kern) -> int {
    const dim3 grid(static_cast<unsigned>(ceil_div(B, grid128 ? 128 : 64)));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, table, lin, V, idx, B, F, static_cast<const bf16x8*>(Wsb), bias, z1, pair,
                       fsum, lin_out);
    return launch_status();
  };
  if (g_sb_waves == 56) {        // 128 samples per workgroup, 8 waves, weight fragments direct
    threads = 512;
    lds = static_cast<size_t>(2) * (4 * (KD / 16) * 3 * 64 * 16);
    grid128 = true;
    return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 8, true, false, 128>) : launch(l1_fwd_sb_kernel<KD, HD, false, 8, true, false, 128>);
  }
  if (g_sb_waves == 40) {        // 8 waves, specialised roles
    threads = 512;
    return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 8, false, true>) : launch(l1_fwd_sb_kernel<KD, HD, false, 8, false, true>);
  }
  const int nw = (g_sb_waves & 15) == 4 ? 4 : 8;
  const bool direct = (g_sb_waves & 16) != 0;
  threads = nw * 64;
  if (direct) lds = static_cast<size_t>(2) * (2 * (KD / 16) * 3 * 64 * 16);
  if (nw == 4 && !direct) return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 4, false>) : launch(l1_fwd_sb_kernel<KD, HD, false, 4, false>);
  if (nw == 4) return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 4, true>) : launch(l1_fwd_sb_kernel<KD, HD, false, 4, true>);
  if (!direct) return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 8, false>) : launch(l1_fwd_sb_kernel<KD, HD, false, 8, false>);
  return lin != nullptr ? launch(l1_fwd_sb_kernel<KD, HD, true, 8, true>) : launch(l1_fwd_sb_kernel<KD, HD, false, 8, true>);
}
