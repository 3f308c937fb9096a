/*
 * libreco_hip.h — C ABI of the MI355X (gfx950) hot path for LibRecommender.
 *
 * The reference (massquantity/LibRecommender v1.5.2) has no FFI for this path: its
 * "kernels" are TensorFlow / PyTorch / numpy calls made from Python.  Each entry point
 * below therefore cites the reference *call site* it replaces (path:line relative to the
 * reference checkout) instead of an existing binding.  INTEGRATION.md shows the ctypes
 * stub a LibRecommender maintainer would add to call them.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - tensors are dense row-major, fp32 unless stated, indices int32 (the reference feeds
 *     tf.int32 placeholders, e.g. algorithms/deepfm.py:176-177), 64-bit counts/sizes;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); kernels are
 *     enqueued asynchronously, nothing synchronises, nothing allocates;
 *   - scratch memory is provided by the caller (`ws`, `ws_bytes`); the matching
 *     `*_ws_bytes` query is a pure host function;
 *   - return value: LR_OK (0) on success, a negative LR_E* code for argument errors,
 *     or a positive hipError_t from the launch.  `lr_strerror` renders either.
 *   - no global state; safe to call from several host threads on distinct streams.
 */
#ifndef LIBRECO_HIP_H_
#define LIBRECO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_OK 0
#define LR_EINVAL (-1)    /* bad argument (null pointer, negative size, ...)            */
#define LR_ESHAPE (-2)    /* shape not supported by the compiled kernels (see function) */
#define LR_EWORKSPACE (-3) /* workspace too small                                        */

typedef void* lr_stream_t;

const char* lr_strerror(int code);
/* ABI version of this header: bumped on any signature change. */
int lr_abi_version(void);
/* Nodes of a captured hipGraph (a hipGraph_t) that are neither kernel nor empty nodes — a captured training step must hold
 * kernel nodes only (memset / memcpy nodes replayed beside eager work faulted on this stack).  *n_nodes_out (nullable):
 * all nodes.  < 0: the negated error.  Host-only call.                                                             */
int lr_graph_foreign_nodes(void* graph, int* n_nodes_out);

/* ------------------------------------------------------------------------------------
 * (a1) Row gather — replaces tf.nn.embedding_lookup at layers/embedding.py:23,
 * tfops/features.py:40,69,102, algorithms/two_tower.py:307,325,370-374 and the batch-row
 * gathers of training/torch_trainer.py:154-156.
 *   out[i, :] = table[idx[i], :]   for i in [0, n);  idx outside [0, V) yields a zero row
 *   (TF-on-GPU semantics; TF-on-CPU raises).  Any K >= 1; K % 4 == 0 takes the 16-byte path.
 * ---------------------------------------------------------------------------------- */
int lr_embed_gather_f32(const float* table, int64_t V, int K, const int32_t* idx,
                        int64_t n, float* out, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a2) Fixed-length bag pooling with OOV -> 0 — replaces multi_sparse_alone
 * (tfops/features.py:90-118) and seq_embeds_pooling (layers/embedding.py:54-85).
 *   idx is [nbags, bag_len]; entries equal to `oov` (or outside [0,V)) contribute a zero
 *   vector; out[b,:] = sum / d, d = 1 (sum), count (mean) or sqrt(count) (sqrtn) of the
 *   non-OOV entries, with x/0 = 0 (tf.div_no_nan).  The reference zeroes the OOV row *in the
 *   variable* on every forward (quirk 8); here the row is simply never read.
 * ---------------------------------------------------------------------------------- */
#define LR_COMBINER_SUM 0
#define LR_COMBINER_MEAN 1
#define LR_COMBINER_SQRTN 2
int lr_embed_bag_pool_f32(const float* table, int64_t V, int K, const int32_t* idx,
                          int64_t nbags, int bag_len, int combiner, int32_t oov,
                          float* out, lr_stream_t stream);
/* backward of the pooling: expands d(out)[nbags,K] to per-entry gradients
 * gentry[nbags*bag_len, K] (zero for OOV entries), ready for lr_embed_scatter_*.      */
int lr_embed_bag_pool_bwd_f32(const float* gout, int K, const int32_t* idx, int64_t V,
                              int64_t nbags, int bag_len, int combiner, int32_t oov,
                              float* gentry, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Segment construction ("CSR by touched row") — the index half of the gradient path that
 * TF performs inside the IndexedSlices gradient of embedding_lookup + AdamOptimizer
 * (training/tf_trainer.py:120-121: duplicates are summed before the update).
 *   Sorts (idx[i], i) by idx (stable), then emits
 *     seg_pos   [n]      : original positions i, grouped by row, ascending i inside a row
 *     seg_rows  [n]      : the distinct rows, ascending          (first *n_seg valid)
 *     seg_start [n + 1]  : start of each row's run in seg_pos    (first *n_seg + 1 valid)
 *     n_seg     [1]      : number of distinct rows (device int32; never read by the host)
 *     pos_to_seg[n]      : (nullable) run number of every position, -1 for dropped entries —
 *                          the inverse map that turns a de-duplicated row cache back into
 *                          per-position slots (multi-GPU row exchange)
 *   Entries with idx outside [0,V) are dropped (their run is not emitted).
 *   Bit-exact integer work: the oracle is np.unique/argsort(kind="stable").
 * ---------------------------------------------------------------------------------- */
size_t lr_segments_ws_bytes(int64_t n, int64_t V);
int lr_segments_build(const int32_t* idx, int64_t n, int64_t V, int32_t* seg_pos,
                      int32_t* seg_rows, int32_t* seg_start, int32_t* n_seg,
                      int32_t* pos_to_seg, void* ws, size_t ws_bytes, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a1 bwd / a11) Gradient scatter — replaces the IndexedSlices gradient of
 * tf.nn.embedding_lookup and nn.Embedding's dense gradient
 * (training/torch_trainer.py:116-121).
 *   lr_embed_segment_sum_f32 : grows[s,:] = sum_{p in run s} grad[seg_pos[p], :]
 *                              (deterministic: ascending position order per row)
 *   lr_embed_scatter_add_f32 : table[seg_rows[s],:] += alpha * (that sum)   (SGD-style / dense-grad build)
 * ---------------------------------------------------------------------------------- */
/* Long runs (a Zipf head row collecting thousands of positions) are cut into 512-position chunks summed by whole
 * workgroups when a workspace of lr_embed_scatter_ws_bytes(n_max, K) bytes is passed (`ws`; NULL: every run is walked by
 * one row group).  Chunk partials are added in chunk order: results stay run-to-run identical.  Vector path only
 * (K in {16, 32, 64, 128}, 16-byte aligned operands). */
size_t lr_embed_scatter_ws_bytes(int64_t n_max, int K);
int lr_embed_segment_sum_f32(const float* grad, int K, const int32_t* seg_pos,
                             const int32_t* seg_start, const int32_t* n_seg,
                             int64_t n_max, float* grows, void* ws, size_t ws_bytes,
                             lr_stream_t stream);
int lr_embed_scatter_add_f32(float* table, int64_t V, int K, const float* grad,
                             const int32_t* seg_pos, const int32_t* seg_rows,
                             const int32_t* seg_start, const int32_t* n_seg,
                             int64_t n_max, float alpha, void* ws, size_t ws_bytes,
                             lr_stream_t stream);

/* Adam hyper-parameters.  TF1 form (tf.train.AdamOptimizer, training/tf_trainer.py:120):
 *   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  w -= lr_t * m / (sqrt(v) + eps)
 * torch form (torch.optim.Adam, training/torch_trainer.py:63-69; `tf_style` = 0):
 *   w -= lr / (1 - beta1^t) * m / (sqrt(v / (1 - beta2^t)) + eps), g += weight_decay * w */
typedef struct lr_adam_hp {
  double lr;           /* doubles: the reference passes Python floats; (1 - beta) is formed  */
  double beta1;        /* in fp32 by TF (1.f - 0.999f) but in double by torch (1 - 0.999),   */
  double beta2;        /* a 1.3e-5 relative difference that both styles reproduce exactly    */
  double eps;
  double weight_decay; /* torch-style L2 added to the gradient of touched rows (0 = off)     */
  int32_t step;        /* t >= 1, the step being applied                                     */
  int32_t tf_style;    /* 1 = TF1 (sparse-apply form), 0 = torch.optim.Adam                  */
} lr_adam_hp;

/* Fused segment-sum + row-wise Adam on touched rows only ("lazy" Adam): one pass that
 * reads each gradient row once and read-modify-writes w, m, v of each distinct row once. */
int lr_embed_scatter_adam_f32(float* table, float* m, float* v, int64_t V, int K,
                              const float* grad, const int32_t* seg_pos,
                              const int32_t* seg_rows, const int32_t* seg_start,
                              const int32_t* n_seg, int64_t n_max, lr_adam_hp hp,
                              void* ws, size_t ws_bytes, lr_stream_t stream);
/* The same update on a table AND its per-row linear weight ([V,1] arrays, scalar gradients glin [n_max])
 * from one pass over the segments — the owner-side update of the row-sharded tables. */
int lr_embed_scatter_adam_lin_f32(float* table, float* m, float* v, int64_t V, int K, const float* grad,
                                  float* lin, float* lin_m, float* lin_v, const float* glin,
                                  const int32_t* seg_pos, const int32_t* seg_rows,
                                  const int32_t* seg_start, const int32_t* n_seg, int64_t n_max,
                                  lr_adam_hp hp, lr_stream_t stream);
/* (e) The owner-side update of the row-sharded tables straight from the peers' lists: after the gradient all-to-all
 * the owner holds W lists back to back (peer_counts[p] entries each, a HOST array), ids[i] = local row, grad [n, K] /
 * glin [n] its gradients; a row appears at most once per list.  Rows are grouped through peer_tab (int32 [V * W], all
 * zero on entry and on return; unused and may be NULL for W == 1) instead of a sort; gradients of a row are added in
 * ascending peer order — the order (and the bits) of lr_segments_build + lr_embed_scatter_adam_lin_f32 on the same
 * lists.  lin / lin_m / lin_v / glin may all be NULL (no per-row linear weight).  K in {16, 32, 64, 128}, W <= 64,
 * 16-byte aligned arrays; otherwise LR_ESHAPE (use the segment path).                                              */
int lr_embed_peer_adam_f32(float* table, float* m, float* v, int64_t V, int K, const float* grad, float* lin,
                           float* lin_m, float* lin_v, const float* glin, const int32_t* ids,
                           const int64_t* peer_counts, int W, int32_t* peer_tab, lr_adam_hp hp, lr_stream_t stream);
/* Dense Adam over the whole table with TF1 semantics (every row decays m, v and moves every
 * step, training/tf_trainer.py:120; SURVEY §7 "TF1 Adam is dense").  `grows`/`seg_rows`
 * are the segment sums of the touched rows (may be empty); `l2` adds 2*l2*w to every row's
 * gradient (tf.keras.regularizers.l2, tfops/configs.py:20-26).  `row_slot` is an int32[V]
 * scratch array owned by the caller; it must be all -1 on entry and is all -1 on return.
 * With `seg_rows == NULL` and `grows != NULL`, `grows` is a full dense gradient [V,K] (the
 * dense-layer parameters: MLP kernels/biases, BatchNorm gamma/beta) and no scratch is used.
 * `vmax` (nullable, [V,K]) switches on AMSGrad (torch.optim.Adam(amsgrad=True),
 * training/torch_trainer.py:63-69): running element-wise maximum of v used in the denominator. */
int lr_adam_dense_f32(float* table, float* m, float* v, float* vmax, int64_t V, int K,
                      const float* grows, const int32_t* seg_rows, const int32_t* n_seg,
                      int64_t n_max, int32_t* row_slot, float l2, lr_adam_hp hp,
                      lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a4) FM pairwise interaction — replaces algorithms/fm.py:158-161, deepfm.py:160-163:
 *   pair[b,k] = 0.5 * ((sum_f e[b,f,k])^2 - sum_f e[b,f,k]^2)
 * Stand-alone forward/backward on a materialised e[B,F,K] ...
 * ---------------------------------------------------------------------------------- */
int lr_fm_pairwise_fwd_f32(const float* e, int64_t B, int F, int K, float* pair,
                           float* fsum /* [B,K] or NULL */, lr_stream_t stream);
/* ge[b,f,k] (+)= gpair[b,k] * (fsum[b,k] - e[b,f,k]);  accumulate != 0 adds to ge. */
int lr_fm_pairwise_bwd_f32(const float* e, const float* fsum, const float* gpair,
                           int64_t B, int F, int K, float* ge, int accumulate,
                           lr_stream_t stream);
/* ... and the fused form the models use: gather F rows per sample from one table addressed
 * by global row ids (user / item / sparse fields concatenated with offsets, the layout of
 * tfops/features.py:6-44), write them once as the MLP input e[B,F,K] and reduce the
 * pairwise term in the same pass (one wavefront per sample, no second read of e).
 * `e` may be NULL (plain FM has no deep part).  `lin` / `lin_out[B,F]` (both or neither)
 * additionally gather the linear weights (the `*_linear_var` tables, deepfm.py:181-192) in
 * the same pass.                                                                        */
int lr_fm_embed_fwd_f32(const float* table, const float* lin, int64_t V, int K,
                        const int32_t* idx, int64_t B, int F, float* e, float* pair,
                        float* fsum, float* lin_out, lr_stream_t stream);
/* Fused backward + optimiser for the same layout: for every distinct row r touched by the
 * batch, with P(r) its (b,f) positions (segments built over idx[B*F]),
 *   g_r = sum_{(b,f) in P(r)} ( gdeep[b,f,:] - bn_a[f,:]
 *                               + gpair[b,:] * (fsum[b,:] - table[r,:]) - bn_c[f,:] * table[r,:] )
 * then one row-wise Adam update of (table, m, v)[r]; with `lin` != NULL also
 *   glin_r = sum glin[f,b] and one Adam update of (lin, lin_m, lin_v)[r].  `glin` is
 *   FIELD-MAJOR [F,B] (a row's positions share f: its reads stay in one L2-resident strip).
 * `gdeep` may be NULL (plain FM).  bn_a / bn_c [F,K] (both or neither) carry the per-feature
 * affine terms of a batch-statistics BatchNorm folded into the first dense layer
 * (layers/dense.py:30-31: d x = G - a - c*x), so the normalised copy of e is never formed.
 * Valid because every position of row r holds the same value table[r] (no bag pooling).
 * Runs longer than 32 positions (Zipf head) are summed by a whole workgroup; `ws` holds that
 * work list.  Deterministic: no floating-point atomics.                                  */
size_t lr_fm_embed_bwd_ws_bytes(int64_t B, int F);
int lr_fm_embed_bwd_adam_f32(float* table, float* m, float* v, float* lin, float* lin_m,
                             float* lin_v, int64_t V, int K, const float* gdeep,
                             const float* gpair, const float* fsum, const float* glin,
                             const float* bn_a, const float* bn_c, int64_t B, int F,
                             const int32_t* seg_pos, const int32_t* seg_rows,
                             const int32_t* seg_start, const int32_t* n_seg, lr_adam_hp hp,
                             void* ws, size_t ws_bytes, lr_stream_t stream);

/* "Rows" form for row-sharded tables: `row_cache[U,K]` holds the batch's distinct
 * rows in run order (fetched from their owners); the summed per-row gradients are written to
 * grows_out[U,K] / glin_out[U] instead of being applied, to be sent back to the owning ranks. */
int lr_fm_embed_bwd_rows_f32(const float* row_cache, int K, const float* gdeep,
                             const float* gpair, const float* fsum, const float* glin,
                             const float* bn_a, const float* bn_c, int64_t B, int F,
                             const int32_t* seg_pos, const int32_t* seg_start,
                             const int32_t* n_seg, float* grows_out, float* glin_out, void* ws,
                             size_t ws_bytes, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a1 + a4 + a5, fused) DeepFM: embedding lookup fused with the FIRST Dense layer of dense_nn on
 * the f32 MFMA pipe — replaces the concatenation deep_embed [B, F*K] (algorithms/deepfm.py:236-247),
 * its batch_normalization + first tf_dense (layers/dense.py:30-41 via deepfm.py:163-169), the FM
 * terms (deepfm.py:158-162) and, in the backward, the IndexedSlices gradient + AdamOptimizer
 * (training/tf_trainer.py:120-121).  deep_embed and its gradient never exist in memory.
 *
 * Wp [F*K, H1] is the BatchNorm-FOLDED first kernel, Wp = diag(gamma * rsqrt(var + eps)) @ W1, and
 * `bias` = b1 + (beta - mean * gamma * rsqrt(var + eps)) @ W1, both formed by the caller from the
 * batch statistics (lr_fm_field_stats_f32 reads them off the batch's runs); the remaining BatchNorm
 * backward terms enter lr_fm_rows_adam_f32 as bn_a / bn_c (dx = G - a - c * x).
 * Shapes compiled: (K, H1) with lr_deepfm_l1_supported(K, H1) != 0; others return LR_ESHAPE.
 *
 *   lr_deepfm_l1_pack_f32   Wp -> WpA / WpB (F*K*H1 floats each): MFMA fragment order for the
 *                           forward / the row-gradient kernel (one coalesced 16-byte load per lane
 *                           and four MFMAs).
 *   lr_deepfm_l1_fwd_f32    z1[b,:]   = sum_f table[idx[b,f],:] @ Wp[f*K:(f+1)*K,:] + bias
 *                           fsum[b,:] = sum_f table[idx[b,f],:];  pair = 0.5*(fsum^2 - sum_f row^2)
 *                           lin_out[b,f] = lin[idx[b,f]]          (lin / lin_out both NULL or both set)
 *                           ids outside [0,V) contribute a zero row.
 *   lr_deepfm_l1_wgrad_f32  partial[c, f*K+i, n] = sum over the samples of batch chunk c of
 *                           table[idxT[f,b], i] * gz[b, n];  dWp = sum_c partial[c] (caller, fixed
 *                           order).  idxT is idx transposed to [F, B] (lr_idx_transpose_i32);
 *                           n_chunks from lr_deepfm_l1_wgrad_chunks (any value >= 1 is valid).
 *   lr_deepfm_l1_dgrad_f32  ge[slotT[f,b], :] = gz[b,:] @ Wp[f*K:(f+1)*K,:]^T + gl[b]*wp[:]*fsum[b,:]
 *                           i.e. the per-position row gradient WITHOUT the terms that only depend on
 *                           the row itself, written in RUN ORDER (slotT from lr_segments_build_fields;
 *                           positions with slot -1 are written to the spare row B*F: ge holds B*F + 1 rows).  gl = d loss / d logit, wp = the
 *                           output layer's weights of the pairwise term (deepfm.py:171-172: the FM
 *                           term feeds one Dense(1), so d loss / d pair[b,:] = gl[b] * wp[:]).
 *   lr_fm_rows_adam_f32     per distinct row r (run s of n positions, field f):
 *                             g = sum_p ge[p] - n*bn_a[f] - w_r * (n*bn_c[f] + wp * sum_p gl[b(p)])
 *                             Adam(w_r, m_r, v_r, g);  lin rows: g_lin = lin_scale[f] * sum_p gl[b(p)]
 *                           (lin_scale[f] = d logit / d lin_out[b,f] = out_kernel[0] * linear_kernel[f]).
 *                           Same bucketing / determinism as lr_fm_embed_bwd_adam_f32; `ws` sized by
 *                           lr_fm_embed_bwd_ws_bytes.
 * ---------------------------------------------------------------------------------- */
int lr_deepfm_l1_supported(int K, int H1);
/* Round 4: the forward and the row-gradient kernel exist in two tilings — 32 samples per workgroup (two workgroups per CU)
 * and 64 samples per workgroup with one wave per SIMD and two accumulators per wave (half the L2 -> CU weight traffic;
 * chosen when the batch fills the chip with one workgroup per CU: B >= 12,288).  Both run the same k-ordered f32 fma
 * chain per output element: bit-identical results.  `tile` = 32 or 64 pins the family (tests, profiling), 0 restores the
 * automatic choice.  (The weight-gradient kernel has the 32-sample form only: its wide form measured slower.) */
void lr_deepfm_l1_tile_override(int tile);
int lr_deepfm_l1_pack_f32(const float* Wp, int F, int K, int H1, float* WpA, float* WpB,
                          lr_stream_t stream);
int lr_idx_transpose_i32(const int32_t* idx, int64_t B, int F, int32_t* idxT, lr_stream_t stream);
int lr_deepfm_l1_fwd_f32(const float* table, const float* lin, int64_t V, int K,
                         const int32_t* idx, int64_t B, int F, const float* WpA, const float* bias,
                         int H1, float* z1, float* pair, float* fsum, float* lin_out,
                         lr_stream_t stream);
/* Round 5: the three contractions of the layer as SPLIT-bf16 MFMA products with f32 accumulation (csrc/deepfm_l1_sb.hip) — every
 * f32 operand split exactly into three bf16 values, a product taken as the six largest cross terms (v_mfma_f32_32x32x16_bf16).
 * As close to f64 as the f32 fma chain above on this layer's reductions (relative rms error 1.88e-6 vs 2.08e-6 at K = 12,928,
 * profiles/r04_bf16_split_probe.txt), NOT bit-identical to it; the host mirror selects it by default where the shape is
 * compiled (K = 64, H1 = 128: lr_deepfm_l1_sb_supported; LR_ESHAPE otherwise) and keeps the f32 chain selectable.
 * Same contracts as lr_deepfm_l1_fwd / wgrad / dgrad_f32 except the packed operands:
 *   lr_deepfm_l1_sb_pack     W [F*K, H1] row-major (optional row scale = the BatchNorm fold) -> WsbA (forward) and / or WsbB
 *                            (row gradient): three bf16 planes in fragment order, lr_deepfm_l1_sb_pack_bytes bytes each
 *   lr_deepfm_l1_sb_gz_pack  gz [B, H1] -> planes for the weight gradient (lr_deepfm_l1_sb_gz_pack_bytes bytes; samples
 *                            padded with zeros to a multiple of 16)
 *   lr_deepfm_l1_fwd_sb_f32  `ws` (lr_deepfm_l1_fwd_sb_ws_bytes): partial sums when the fields are split across workgroups
 *                            (batches that do not fill the chip with 128-sample tiles); z1 / pair / fsum are then summed in
 *                            field-group order by a second launch
 *   lr_deepfm_l1_wgrad_sb_f32  n_chunks: any value >= 1 (lr_deepfm_l1_wgrad_sb_chunks: the grid that fills the chip)
 * All pointers 16-byte aligned.                                                                                              */
int lr_deepfm_l1_sb_supported(int K, int H1);
int lr_deepfm_l1_fwd_sb_supported(int K, int H1);   /* = lr_deepfm_l1_sb_supported (round-4 name) */
size_t lr_deepfm_l1_sb_pack_bytes(int F, int K, int H1);
/* `red_partial` / `red_nblk` / `red_out` (NULL / 0 / NULL: none): the launch also sums the `red_nblk` slab partials [red_nblk][H1]
 * of the folded bias into red_out [H1] (the arithmetic of lr_reduce_partials_f32, ceil(H1 / 16) extra workgroups). */
int lr_deepfm_l1_sb_pack(const float* W, const float* scale, int F, int K, int H1, void* WsbA, void* WsbB,
                         const float* red_partial, int red_nblk, float* red_out, lr_stream_t stream);
size_t lr_deepfm_l1_sb_gz_pack_bytes(int64_t B, int H1);
int lr_deepfm_l1_sb_gz_pack(const float* gz, int64_t B, int H1, void* gzp, lr_stream_t stream);
/* profiling / tests (same results in every mode up to the order of the field-group sums): samples per workgroup of the forward
 * (64 / 128; 0 = automatic), field groups of the forward / row-gradient grids (0 = automatic), multiplying waves (4 / 8) and
 * fields (2 / 4) per workgroup of the weight gradient (0 = automatic) */
void lr_deepfm_l1_sb_override(int fwd_tile, int ksplit, int wgrad_cw, int wgrad_fg);
size_t lr_deepfm_l1_fwd_sb_ws_bytes(int64_t B, int F);
int lr_deepfm_l1_fwd_sb_f32(const float* table, const float* lin, int64_t V, int K, const int32_t* idx, int64_t B,
                            int F, const void* WsbA, const float* bias, int H1, float* z1, float* pair, float* fsum,
                            float* lin_out, void* ws, size_t ws_bytes, lr_stream_t stream);
int lr_deepfm_l1_wgrad_sb_chunks(int64_t B, int F);
int lr_deepfm_l1_wgrad_sb_f32(const float* table, int64_t V, int K, const int32_t* idxT, int64_t B, int F,
                              const void* gzp, int H1, int n_chunks, float* partial, lr_stream_t stream);
int lr_deepfm_l1_dgrad_sb_f32(const float* gz, int H1, const void* WsbB, int K, int F, int64_t B, const float* gl,
                              const float* wp, const float* fsum, const int32_t* slotT, float* ge, lr_stream_t stream);
int lr_deepfm_l1_wgrad_chunks(int64_t B, int F);
int lr_deepfm_l1_wgrad_f32(const float* table, int64_t V, int K, const int32_t* idxT, int64_t B,
                           int F, const float* gz, int H1, int n_chunks, float* partial,
                           lr_stream_t stream);
int lr_deepfm_l1_dgrad_f32(const float* gz, int H1, const float* WpB, int K, int F, int64_t B,
                           const float* gl, const float* wp, const float* fsum,
                           const int32_t* slotT, float* ge, lr_stream_t stream);
/* Row-sharded tables: the batch's rows live in a per-step row cache [n_cache, K] addressed through the
 * position -> cache-slot map `slots` [B*F]; the segments are the per-field runs of the GLOBAL row ids.
 *   lr_fm_field_stats_slots_f32  lr_fm_field_stats_f32 with the run's row read from cache[slots[first position]]
 *   lr_fm_rows_grad_f32          lr_fm_rows_adam_f32's per-row gradient WITHOUT the update: grows[slot] = g,
 *                                glin_rows[slot] = g_lin (handed to the owners by the gradient all-to-all) */
int lr_fm_field_stats_slots_f32(const float* cache, int K, const int32_t* seg_rows, const int32_t* seg_start,
                                const int32_t* n_seg, const int32_t* field_row_start, int F, int C,
                                float* partial, const int32_t* seg_pos, const int32_t* slots,
                                lr_stream_t stream);
int lr_fm_rows_grad_f32(const float* cache, const float* lin_cache, int64_t n_cache, int K, const float* ge,
                        const float* gl, const float* wp, const float* bn_a, const float* bn_c,
                        const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                        const int32_t* seg_rows, const int32_t* seg_start, const int32_t* n_seg,
                        const int32_t* slots, float* grows, float* glin_rows, void* ws, size_t ws_bytes,
                        lr_stream_t stream);
/* The reference's own optimiser semantics on the fused step (training/tf_trainer.py:120 — tf.train.AdamOptimizer moves EVERY row
 * of a table every step, touched or not):
 *   lr_fm_rows_grad_compact_f32  lr_fm_rows_adam_f32's per-row gradient WITHOUT the update, rows read from the tables
 *                                themselves, run s writing grows[s] / glin_rows[s] (compact: n_seg rows)
 *   lr_adam_dense_rows_f32       ONE streaming pass over (table, m, v) [V,K] and (lin, lin_m, lin_v) [V] (nullable, all or
 *                                none): every row decays its moments and moves; the rows listed in seg_rows[0 .. *n_seg) take
 *                                grows[s] / glin_rows[s].  `row_slot`: int32[V] scratch, all -1 on entry and on return.
 *                                K in {16, 32, 64, 128}.  `_dc_`: coefficients from device memory (hipGraph-captured steps). */
int lr_fm_rows_grad_compact_f32(const float* table, const float* lin, int64_t V, int K, const float* ge,
                                const float* gl, const float* wp, const float* bn_a, const float* bn_c,
                                const float* lin_scale, int64_t B, int F, const int32_t* seg_pos,
                                const int32_t* seg_rows, const int32_t* seg_start, const int32_t* n_seg,
                                float* grows, float* glin_rows, void* ws, size_t ws_bytes, lr_stream_t stream);
/* row_slot[seg_rows[s]] = s for s < *n_seg (set != 0), or = -1 (set == 0): the row -> segment map the dense passes read */
int lr_row_slots_i32(const int32_t* seg_rows, const int32_t* n_seg, int64_t n_max, int32_t* row_slot, int set,
                     lr_stream_t stream);
int lr_adam_dense_rows_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v, int64_t V, int K,
                           const float* grows, const float* glin_rows, const int32_t* seg_rows, const int32_t* n_seg,
                           int64_t n_max, int32_t* row_slot, lr_adam_hp hp, lr_stream_t stream);
int lr_adam_dense_rows_dc_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v, int64_t V, int K,
                              const float* grows, const float* glin_rows, const int32_t* seg_rows, const int32_t* n_seg,
                              int64_t n_max, int32_t* row_slot, const void* coef_dev, lr_stream_t stream);
int lr_fm_rows_adam_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v,
                        int64_t V, int K, const float* ge, const float* gl, const float* wp,
                        const float* bn_a, const float* bn_c, const float* lin_scale, int64_t B,
                        int F, const int32_t* seg_pos, const int32_t* seg_rows,
                        const int32_t* seg_start, const int32_t* n_seg, lr_adam_hp hp, void* ws,
                        size_t ws_bytes, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a5 + a10) DeepFM "tail": the layers of dense_nn after the first Dense (layers/dense.py:33-49:
 * relu -> batch-statistics BatchNorm -> Dense, last layer linear), the output layer over
 * [linear term | pairwise term | deep term] (algorithms/deepfm.py:158, 171-172), the sigmoid
 * cross-entropy loss (tfops/loss.py:14-16) and their backward — as a few small launches cut only
 * where BatchNorm needs a batch-wide reduction.  Widths: multiples of 16, <= 256
 * (lr_mlp_tail_supported).  Every batch reduction is a per-workgroup partial (64 samples) + a
 * fixed-order second pass: no atomics, run-to-run bit identical.
 *   lr_mlp_colstats_f32     partial[blk][{sum, sumsq}][d] of relu(z)
 *   lr_mlp_bn_finalize_f32  mean, rsqrt(var + eps) (biased var) + moving averages (momentum)
 *   lr_mlp_layer_fwd_f32    z_out = BN(relu(z_in)) @ W + b  (+ colstats of relu(z_out)); BN pointers NULL = none
 *   lr_mlp_head_f32         logits, gl = d loss / d logit, per-workgroup partials
 *                           [d wo (1+K+dn) | d bo | d wl (F) | d bl | loss sum]  (dn+K+F+4 floats each).
 *                           Plain form (the output layer of DIN / YouTubeRanking, algorithms/din.py:190-192):
 *                           F == 0 (lin_out, wl, bl NULL) drops the linear term and wo[0]; K == 0 (pair NULL) the
 *                           pairwise term: logit = zn @ wo + bo, partials [d wo (K+dn) | d bo | loss sum].
 *   lr_mlp_layer_bwd_f32    through z_out = h_in @ W + b: upstream gz_out = gl*wd (mode 0: z_out is the
 *                           last layer) or the activation/BatchNorm backward of gh_out (mode 1);
 *                           gh_in = gz_out @ W^T, partials of dW, db and of the input BatchNorm's
 *                           (sum gh_in, sum gh_in * x_hat_in) = (d beta, d gamma)
 *   lr_mlp_first_bwd_f32    gz_1 from gh_1 (+ partial column sums)
 *   lr_reduce_partials_f32  out[c] = sum_k partial[k*stride + c], fixed order
 * ---------------------------------------------------------------------------------- */
/* Round 5: the whole tail of a THREE-layer dense_nn (128 -> 64 -> 32 after the first Dense: the reference's default
 * hidden_units) as ONE persistent launch — the same per-tile arithmetic as the chain of lr_mlp_* launches below (bit-identical
 * results), min(tiles, CUs) resident workgroups meeting at four grid barriers where a batch-statistics BatchNorm needs the whole
 * batch.  Inputs z0 = first Dense output [B, 128], pair [B, K] / lin_out [B, F] / labels as lr_mlp_head_f32; buffers as the
 * chain's (tiles = ceil(B / 64)): z1 [B, 64], z2 [B, 32], gh0 [B, 128], gh1 [B, 64], stat* / bnp* [tiles, 2, d], dW*p
 * [tiles, d_in * d_out], db*p [tiles, d_out], headp [tiles, G + 1], gl [B], gz0 [B, 128], sgzp [tiles, 128]; mean* / inv* [d]:
 * the batch statistics; d gamma / d beta of both BatchNorms are written in full, the weight / bias / head partials are left
 * for lr_reduce_partials_multi_f32.  A BatchNorm's pointers (gamma, beta, dgamma, dbeta, stat, bnp, mean, inv; mm / mv
 * optional) are NULL together.
 * `sync`: 24 device words the caller allocates ZEROED once and then only reads: [0] arrival counter (left at 0 by every launch
 * that completes: the last workgroup to finish clears it, so no zeroing launch precedes the kernel), [1] "a launch gave up",
 * [2..17] phase time stamps of workgroup 0 (a profiling aid), [18] the STICKY error word, never cleared by the library,
 * [19] the poll bound in polls (0: the default, ~2 s), [20] finished-workgroup counter, [21..23] reserved.  The launch is a plain
 * one whose grid barrier needs every workgroup resident at once, so the grid is min(tiles, CUs of the current device x the
 * kernel's own occupancy) — lr_mlp_tail3_resident_blocks() — and the poll is bounded: when a barrier does not complete (other
 * work holds the CUs), the workgroups stop BEFORE the next phase, write NaN into their loss partials (headp) and set sync[1] and
 * sync[18]; every later launch on the same `sync` returns at once with a NaN loss.  The caller reads sync[18] where it reads
 * the loss back and raises (librecommender_amd/layers/tail.py:DeepFMTail.check). */
typedef struct lr_mlp_tail3_args {
  int64_t B;
  int K, F;
  const float* z0; const float* pair; const float* lin_out; const float* labels;
  float eps0, mom0; float* mm0; float* mv0; const float* gamma0; const float* beta0; float* dgamma0; float* dbeta0;
  float eps1, mom1; float* mm1; float* mv1; const float* gamma1; const float* beta1; float* dgamma1; float* dbeta1;
  const float* W1; const float* b1; const float* W2; const float* b2;
  const float* wl; const float* bl; const float* wo; const float* bo;
  float* z1; float* z2; float* gh0; float* gh1;
  float* stat0; float* stat1; float* bnp0; float* bnp1;
  float* mean0; float* inv0; float* mean1; float* inv1;
  float* dW1p; float* db1p; float* dW2p; float* db2p; float* headp;
  float* gl; float* gz0; float* sgzp;
  uint32_t drop_seed; float keep;
  unsigned* sync;
} lr_mlp_tail3_args;
int lr_mlp_tail3_supported(int d0, int d1, int d2, int K, int F);
int lr_mlp_tail3_f32(const lr_mlp_tail3_args* args, lr_stream_t stream);
int lr_mlp_tail3_resident_blocks(void);
int lr_mlp_tail_supported(int d_in, int d_out);
int lr_mlp_colstats_f32(const float* z, int64_t B, int d, float* partial, lr_stream_t stream);
int lr_mlp_bn_finalize_f32(const float* partial, int nblk, int d, int64_t B, float eps, float momentum,
                           float* moving_mean, float* moving_var, float* mean_out, float* inv_out,
                           lr_stream_t stream);
int lr_mlp_layer_fwd_f32(const float* z_in, int64_t B, int d_in, const float* mean, const float* inv,
                         const float* gamma, const float* beta, const float* W, const float* bias,
                         int d_out, float* z_out, float* partial_out, uint32_t drop_seed, float drop_keep,
                         int drop_layer, lr_stream_t stream);
int lr_mlp_head_f32(const float* zn, int dn, const float* pair, int K, const float* lin_out, int F,
                    const float* labels, const float* wl, const float* bl, const float* wo,
                    const float* bo, int64_t B, float* logits, float* gl, float* partial,
                    lr_stream_t stream);
int lr_mlp_layer_bwd_f32(int mode, const float* gl, const float* wd, const float* gh_out,
                         const float* z_out, const float* up_mean, const float* up_inv,
                         const float* up_gamma, const float* up_dgamma, const float* up_dbeta,
                         const float* z_in, const float* in_mean, const float* in_inv,
                         const float* in_gamma, const float* in_beta, const float* W, int d_in,
                         int d_out, int64_t B, float* gh_in, float* dW_partial, float* db_partial,
                         float* bn_partial, uint32_t drop_seed, float drop_keep, int drop_layer, lr_stream_t stream);
/* Dropout (layers/dense.py:44-47: after a hidden layer's BatchNorm; kept entries scaled by 1 / keep): `drop_keep` < 1 applies
 * the mask keep(seed, layer, sample, column) = [u < drop_keep], u = the low 24 bits of splitmix64's finaliser over
 * (sample * 4096 + column) ^ (seed * 0x9E3779B97F4A7C15 + layer * 0xD1B54A32D192ED03), to h_in = BN(relu(z_in)) in the
 * forward kernel and regenerates it in the backward kernel (the Dense's input for dW, the gradient through the dropout);
 * `drop_keep` >= 1: no dropout.  The caller passes the same (seed, layer) to both kernels of a layer and a new seed per step. */
int lr_mlp_first_bwd_f32(const float* gh, const float* z, const float* mean, const float* inv,
                         const float* gamma, const float* dgamma, const float* dbeta, int64_t B, int d,
                         float* gz, float* partial, lr_stream_t stream);
int lr_reduce_partials_f32(const float* partial, int nblk, int64_t n, int64_t stride, float* out,
                           lr_stream_t stream);
/* n_jobs independent lr_reduce_partials_f32 in one launch; `jobs_dev`: DEVICE array of
 * { const float* partial; float* out; int64_t n; int64_t stride; int32_t nblk; float div; }
 * (lr_reduce_job_bytes() = 40), max_n = the largest n among them.  div != 0: out = (float)sum / div (the loss's 1 / B). */
size_t lr_reduce_job_bytes(void);
int lr_reduce_partials_multi_f32(const void* jobs_dev, int n_jobs, int64_t max_n, lr_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Streaming in-batch softmax cross-entropy (csrc/softmax_ce.hip) — replaces, for the two-tower
 * retrieval loss, `softmax_cross_entropy` (libreco/tfops/loss.py:71-75) over `adjust_logits`
 * (libreco/algorithms/two_tower.py:458-479): logits = X Y^T (+ col_bias[j] = -log Q(j)), entries
 * with row_ids[i] == col_ids[j] and j != pos0 + i masked to float32.min (accidental hits), labels
 * = the diagonal (column pos0 + i for row i; pos0 > 0 when the columns are the all-gathered
 * item-tower outputs of every rank).  The B x N logits are never written.
 *   lr_softmax_ce_fwd_f32       lse[i] = logsumexp_j logits[i][:], pos_logit[i] = logits[i][pos0+i]
 *                               (loss_i = lse - pos_logit) and, if W != NULL,
 *                               W[i] = sum_j softmax(logits[i])[j] Y[j]   so that
 *                               d loss_i / d X[i] = W[i] - Y[pos0+i]
 *   lr_softmax_ce_bwd_cols_f32  V[j] = sum_i g[i] softmax(logits[i])[j] X[i]   so that
 *                               d (sum_i g_i loss_i) / d Y[j] = V[j] - [j = pos0+i] g[i] X[i]
 * When the rows are a small share of the columns (a rank's users against the all-gathered items) the forward
 * sweep is cut into column ranges and merged (workspace from lr_softmax_ce_fwd_ws_bytes).
 * Fixed summation order (run-to-run identical).  D <= 128, D % 4 == 0
 * (lr_softmax_ce_supported); col_bias and the id pair are nullable; pointers 16-byte aligned.
 * `arith`: the arithmetic of the two contractions — an explicit argument of the workspace query and of both launches (it
 * decides the launch shape and the workspace size; the library keeps no setting of its own):
 *   1  every f32 product as six bf16 MFMA products (each operand split exactly into three bf16
 *      values), f32 accumulation — error against f64 = that of the f32 fma chain;
 *   0  the f32 MFMA fma chain.
 * ---------------------------------------------------------------------------------- */
int lr_softmax_ce_supported(int64_t B, int64_t N, int D);
size_t lr_softmax_ce_fwd_ws_bytes(int64_t B, int64_t N, int D, int arith);   /* 0 unless B is a small share of N */
int lr_softmax_ce_fwd_f32(const float* X, int64_t B, const float* Y, int64_t N, int D,
                          const float* col_bias, const int32_t* row_ids, const int32_t* col_ids,
                          int64_t pos0, float* lse, float* pos_logit, float* W, void* ws, size_t ws_bytes,
                          int arith, lr_stream_t stream);
int lr_softmax_ce_bwd_cols_f32(const float* X, int64_t B, const float* Y, int64_t N, int D,
                               const float* col_bias, const int32_t* row_ids, const int32_t* col_ids,
                               int64_t pos0, const float* lse, const float* g, float* V,
                               int arith, lr_stream_t stream);

/* BatchNorm-fold algebra of the fused first layer (csrc/deepfm_fold.hip) — `tf.layers.batch_normalization`
 * on the concatenated embeddings (libreco/layers/dense.py:30-31) folded into the first Dense of `dense_nn`:
 *   lr_deepfm_l1_fold_stats_f32  per-field partial sums (lr_fm_field_stats_f32, [F][C][2][K]) -> mean,
 *                                inv = rsqrt(var + eps), s = gamma * inv, t = beta - mean * s; moving
 *                                averages updated in place (momentum)
 *   lr_deepfm_l1_pack_scaled_f32 lr_deepfm_l1_pack_f32 of diag(scale) W without materialising it
 *   lr_deepfm_l1_fold_bias_f32   partial[slab][h] = sum_{r in slab} t[r] W[r][h], last slab = b:
 *                                lr_reduce_partials_f32 over lr_deepfm_l1_fold_bias_slabs(n_rows) slabs
 *                                gives the folded bias b + t^T W
 *   lr_deepfm_l1_fold_bwd_f32    weight-gradient slabs of lr_deepfm_l1_wgrad_f32 (summed in slab order)
 *                                + sgz -> dW, dgamma, dbeta, db, bn_a, bn_c (gamma == NULL: dW, db only)
 * Round 6 — the same algebra in two launches instead of four (bit-identical results):
 *   lr_deepfm_l1_fold_stats_bias_f32  = fold_stats + fold_bias: the workgroup of a 64-row slab finalises the statistics of
 *                                its own rows, then forms the slab's bias partial
 *   the pack launches' `red_*` arguments = the reduction of those partials as extra workgroups of the weight pack
 * ---------------------------------------------------------------------------------- */
int lr_deepfm_l1_fold_stats_f32(const float* partial, int F, int C, int K, int64_t B, float eps,
                                float momentum, const float* gamma, const float* beta,
                                float* moving_mean, float* moving_var, float* mean, float* inv, float* s,
                                float* t, lr_stream_t stream);
int lr_deepfm_l1_fold_stats_bias_f32(const float* partial, int F, int C, int K, int64_t B, float eps,
                                     float momentum, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float* mean, float* inv, float* s,
                                     float* t, const float* W, const float* b, int H1, float* bias_partial,
                                     lr_stream_t stream);
int lr_deepfm_l1_pack_scaled_f32(const float* W, const float* scale, int F, int K, int H1, float* WpA,
                                 float* WpB, const float* red_partial, int red_nblk, float* red_out,
                                 lr_stream_t stream);
int lr_deepfm_l1_fold_bias_slabs(int n_rows);
int lr_deepfm_l1_fold_bias_f32(const float* t, const float* W, const float* b, int n_rows, int H1,
                               float* partial, lr_stream_t stream);
int lr_deepfm_l1_fold_bwd_f32(const float* part, int n_slabs, int n_rows, int H1, int64_t B,
                              const float* sgz, const float* W, const float* gamma, const float* beta,
                              const float* mean, const float* inv, float* dW, float* dgamma,
                              float* dbeta, float* db, float* bn_a, float* bn_c, lr_stream_t stream);

/* Step-dependent Adam coefficients in DEVICE memory — for training steps captured in a hipGraph
 * (one `sess.run` per step in the reference, training/tf_trainer.py:76-101): kernel arguments are
 * frozen at capture, so the bias corrections / decayed learning rate of step t are written into a
 * small device buffer (lr_adam_coef_store: a one-thread kernel enqueued ahead of the replay, stream
 * ordered) that the `_dc` forms read.  Same arithmetic as the by-value forms.                     */
size_t lr_adam_coef_bytes(void);
int lr_adam_coef_store(lr_adam_hp hp, void* coef_dev, lr_stream_t stream);
int lr_adam_dense_dc_f32(float* table, float* m, float* v, int64_t n, const float* grad,
                         const void* coef_dev, lr_stream_t stream);
int lr_fm_rows_adam_dc_f32(float* table, float* m, float* v, float* lin, float* lin_m, float* lin_v,
                           int64_t V, int K, const float* ge, const float* gl, const float* wp,
                           const float* bn_a, const float* bn_c, const float* lin_scale, int64_t B,
                           int F, const int32_t* seg_pos, const int32_t* seg_rows,
                           const int32_t* seg_start, const int32_t* n_seg, const void* coef_dev,
                           void* ws, size_t ws_bytes, lr_stream_t stream);
/* lr_embed_scatter_adam_f32 / lr_embed_scatter_adam_lin_f32 with device-resident coefficients: the table update of
 * the general feature nets' captured step (DIN / YouTube* / Transformer / SIM: algorithms/din.py:241-250 runs as one
 * sess.run, training/tf_trainer.py:76-101). */
int lr_embed_scatter_adam_dc_f32(float* table, float* m, float* v, int64_t V, int K, const float* grad,
                                 const int32_t* seg_pos, const int32_t* seg_rows, const int32_t* seg_start,
                                 const int32_t* n_seg, int64_t n_max, const void* coef_dev,
                                 void* ws, size_t ws_bytes, lr_stream_t stream);
int lr_embed_scatter_adam_lin_dc_f32(float* table, float* m, float* v, int64_t V, int K, const float* grad,
                                     float* lin, float* lin_m, float* lin_v, const float* glin,
                                     const int32_t* seg_pos, const int32_t* seg_rows,
                                     const int32_t* seg_start, const int32_t* n_seg, int64_t n_max,
                                     const void* coef_dev, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a5) First Dense layer of dense_nn over a MATERIALISED block of rows (the general feature nets, e.g. DIN's
 * concat([user, item, sparse..., attention output]), algorithms/din.py:186-192, layers/dense.py:12-49): the block is
 * a [rows, K] array addressed through an id map idx [B, F], so lr_deepfm_l1_fwd/wgrad/dgrad_f32 and the fold kernels
 * run on it as on an embedding table.  The two HBM-bound pieces that are specific to it:
 *   lr_table_colstats_f32  partial[f][c][{sum, sumsq}][K] over the samples of chunk c (contiguous ranges of
 *                          ceil(B / C) samples) of table[idx[b, f], :]  — input of lr_deepfm_l1_fold_stats_f32
 *                          (batch statistics of tf.layers.batch_normalization(training=True), layers/dense.py:30-31);
 *                          ids outside [0, V) contribute zeros.  K in {16, 32, 64, 128}.
 *   lr_bn_remainder_f32    G[r, :] -= a[p(r), :] + c[p(r), :] * x[r, :]: the part of the BatchNorm backward that does not
 *                          pass through the GEMM (dx = G - a - c * x).  period == 0: block stored plane by plane
 *                          ([P][B][Kp], p(r) = r / rows_per_plane); period == P > 0: row-major [B, P * Kp] block viewed
 *                          as B * P rows (p(r) = r % P — a materialised deep_embed [B, F * K], deepfm.py:236-247).
 *                          a, c are [P * Kp].
 * ---------------------------------------------------------------------------------- */
int lr_table_colstats_f32(const float* table, int64_t V, int K, const int32_t* idx, int64_t B, int F, int C,
                          float* partial, lr_stream_t stream);
int lr_bn_remainder_f32(float* G, const float* x, const float* a, const float* c, int64_t rows,
                        int64_t rows_per_plane, int64_t period, int Kp, lr_stream_t stream);

/* Field-partitioned segment build: same outputs as lr_segments_build for idx [B, F] whose column
 * f only holds rows of [field_row_start[f], field_row_start[f+1]) (the feature models' global row
 * layout, algorithms/deepfm.py:181-234 as one table); entries outside their field's range are
 * dropped.  Takes the TRANSPOSED ids idxT [F, B]; every column is sorted on its own in LDS (stable
 * 8-bit LSD passes over the local id), so no device-wide sort runs.  slotT [F, B] (nullable):
 * index of position (b, f) in seg_pos, -1 if dropped.  B <= 16384, otherwise LR_ESHAPE (use
 * lr_segments_build).  Bit-exact integer work.                                               */
size_t lr_segments_fields_ws_bytes(int64_t B, int F);
int lr_segments_build_fields(const int32_t* idxT, int64_t B, int F, const int32_t* field_row_start,
                             int32_t* seg_pos, int32_t* seg_rows, int32_t* seg_start,
                             int32_t* n_seg, int32_t* slotT, void* ws, size_t ws_bytes,
                             lr_stream_t stream);
/* The same build, also writing runT [F, B]: the RUN NUMBER (index into seg_rows / seg_start) of position (b, f), -1 if
 * dropped — what maps a position to its row in a per-step row cache laid out in run order.                          */
int lr_segments_build_fields_runs(const int32_t* idxT, int64_t B, int F, const int32_t* field_row_start,
                                  int32_t* seg_pos, int32_t* seg_rows, int32_t* seg_start,
                                  int32_t* n_seg, int32_t* slotT, int32_t* runT, void* ws, size_t ws_bytes,
                                  lr_stream_t stream);

/* (e) Owner partition of a batch's distinct rows for the row-sharded tables (round-robin rows: owner = row % W, the
 * owner's local row = row / W) — the device half of the id exchange the reference does not have (it trains on one
 * device).  rows[0 .. *n_seg) are distinct (any order; ascending from the builds above).  Writes the
 * STABLE owner-major order: perm[r] = place of rows[r], send_ids[perm[r]] = its local row at the owner,
 * counts[o] (int64) = rows asked from owner o, counts[W] = their total.  No sort.  W <= 64 (LR_ESHAPE).  n_max =
 * capacity of rows / perm / send_ids.  Bit-exact integer work.                                                       */
size_t lr_owner_partition_ws_bytes(int64_t n_max, int W);
int lr_owner_partition_i32(const int32_t* rows, const int32_t* n_seg, int64_t n_max, int W, int32_t* perm,
                           int32_t* send_ids, int64_t* counts, void* ws, size_t ws_bytes, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a7) DIN attention pooling — replaces DIN._build_seq_attention (algorithms/din.py:241-250)
 * + din_attention (layers/attention.py:28-64) + the [N+1,K'] materialisation of
 * combine_seq_features (tfops/features.py:151-202, quirk: rebuilt every step) by gathering
 * only the rows a batch touches.
 *   q[b,:]   = item_table[item[b], :]
 *   key[b,l] = item_table[seq[b,l], :]           l < L (pad id allowed, masked by len)
 *   h  = sigmoid( [q, key, q-key, q*key] @ W1[4K,H] + b1[H] )          H = 16 in the reference
 *   s  = (h @ W2[H] + b2) * rsqrt(K);  s[l >= len[b]] = -(2^32) + 1
 *   a  = softmax_l(s);   out[b,:] = sum_l a[l] * key[b,l,:]
 * fwd saves a[B,L] for the backward.  bwd returns per-position gradients w.r.t. the gathered
 * rows (gq[B,K], gkey[B,L,K]; masked positions are written as zeros) and the MLP parameter
 * gradients gW1[4K,H], gb1[H], gW2[H], gb2[1] (overwritten; reduced block-wise in a fixed
 * order through `ws`, no atomics).  K in {16,32,64,128}, H == 16 (the reference's value),
 * otherwise LR_ESHAPE.
 * The `_dense_` forms take already materialised q[B,K'] / keys[B,L,K'] (items with side
 * features: K' = K*(1+n_item_feats), tfops/features.py:204-218) instead of table + ids.
 * ---------------------------------------------------------------------------------- */
size_t lr_din_attn_ws_bytes(int64_t B, int L, int K, int H);
/* The id stream of the fused DIN step (nets/din_fused.py) in one launch: ids[(2 + n_plain + 2 + L) * B], plane-major
 * [users + user_off | items + item_off | sparse[b][cols[p]] + sparse_off (cols NULL: column p) | -1 (the attention-output plane) |
 *  items + item_off (the query rows) | seqs[b][l] + item_off for l < lens[b], else -1] — the global row ids of
 * libreco/tfops/features.py:6-44 and the pad rule of libreco/batch/sequence.py:56-58. */
int lr_din_build_ids_i32(const int32_t* users, const int32_t* items, const int32_t* sparse, int sparse_ld,
                         const int32_t* cols, int n_plain, const int32_t* seqs, const int32_t* lens, int64_t B, int L,
                         int32_t user_off, int32_t item_off, int32_t sparse_off, int32_t* ids, lr_stream_t stream);
int lr_din_attn_pool_fwd_f32(const float* item_table, int64_t V, int K,
                             const int32_t* item, const int32_t* seq, const int32_t* len,
                             int64_t B, int L, const float* W1, const float* b1,
                             const float* W2, const float* b2, int H, float* out,
                             float* attn, float* hid /* [B*L, 16] or NULL */, int32_t* order_out /* [B] or NULL */,
                             lr_stream_t stream);
int lr_din_attn_pool_bwd_f32(const float* item_table, int64_t V, int K,
                             const int32_t* item, const int32_t* seq, const int32_t* len,
                             int64_t B, int L, const float* W1, const float* b1,
                             const float* W2, const float* b2, int H, const float* attn,
                             const float* gout, float* gq, float* gkey, float* gW1,
                             float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                             lr_stream_t stream);
/* The same backward cut in two for callers that run the halves on different streams (nets/din_fused.py: the
 * parameter half beside the table update): parts = 1: data half (gq, gkey and the dz spill in `ws`), 2: parameter
 * half (gW1, gb1, gW2, gb2 from the spill: must follow the data half of the same ws in stream / event order, and —
 * it re-reads the key rows — must not run beside an update of item_table), 3: both.  keep_pad_rows != 0: rows gkey[b, l >= len[b]] are left untouched instead of zeroed (for callers that
 * drop those positions from the table update: 103 MB of zeros per launch at BASELINE cfg 3).                      */
int lr_din_attn_pool_bwd_parts_f32(const float* item_table, int64_t V, int K,
                                   const int32_t* item, const int32_t* seq, const int32_t* len,
                                   int64_t B, int L, const float* W1, const float* b1,
                                   const float* W2, const float* b2, int H, const float* attn,
                                   const float* gout, float* gq, float* gkey, float* gW1,
                                   float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                                   int parts, int keep_pad_rows, const float* hid /* or NULL */,
                                   const int32_t* order /* [B] or NULL */, lr_stream_t stream);
/* When BOTH `hid` and the order buffer are given, `hid` must hold B * L * 16 + 3 * (K / 16) * 256 floats: the forward's extra
 * workgroup also leaves the data kernel's transposed weight images behind the hidden activations, and a backward given `hid` and
 * `order` reads them from there (the pair belongs to ONE forward / backward of the same W1). */
/* `order` (round 6, MFMA widths only): a permutation of the samples — slot s of the backward kernels' wave-strided walk takes
 * sample order[s].  The forward writes it on request (`order_out`, one extra workgroup of its launch): the stable partition by
 * descending key-tile count (class = min(ceil(len / 16), min(ceil(L / 16), 16))), which hands every wave one sample of each
 * length class (fixed strides left a third of the kernels' run time to a fraction of the waves).  Results per sample do not
 * depend on it; the parameter gradients are summed per wave, so a DIFFERENT order changes their last bits (a fixed order keeps
 * them run-to-run identical). */
/* `hid` (round 6; K in 16 / 32 / 64 / 128, otherwise LR_ESHAPE): the forward keeps the attention MLP's hidden activations
 * h = sigmoid(z) of every live (sample, key) pair ([B*L, 16] floats, rows past a sample's length untouched); the data half of
 * the backward given the same buffer reads them instead of recomputing the first layer (half of its MFMAs, no forward weight
 * images) — the gradients are those of the forward's own h.  NULL on both sides: the recomputing form. */
int lr_din_attn_dense_fwd_f32(const float* q, const float* keys, int K, const int32_t* len,
                              int64_t B, int L, const float* W1, const float* b1,
                              const float* W2, const float* b2, int H, float* out,
                              float* attn, lr_stream_t stream);
int lr_din_attn_dense_bwd_f32(const float* q, const float* keys, int K, const int32_t* len,
                              int64_t B, int L, const float* W1, const float* b1,
                              const float* W2, const float* b2, int H, const float* attn,
                              const float* gout, float* gq, float* gkey, float* gW1,
                              float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                              lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a16 + a17) Full-catalog scoring with fused top-k — replaces
 * `preds = user_embed @ item_embeds.T` (recommendation/recommend.py:66-68,
 * bases/dyn_embed_base.py:146) + rank_recommendations (recommendation/ranking.py:10-56,
 * filter_items :59-61, partition_select :76-78) + tf.math.top_k (bases/dyn_embed_base.py:303).
 *   scores[u,i] = <users[u,:], items[i,:]>   (exact f32 fma chain on the f32 MFMA pipe)
 *   per user: drop consumed items (CSR consumed_ptr/consumed_idx: ascending GLOBAL item ids,
 *   only for users with filter_flag[u] != 0 — the host sets it per ranking.py:38), keep the
 *   k largest, return them sorted by (score desc, id asc).
 *   out_ids are GLOBAL ids (= local row + item_base) so item-sharded callers can merge.
 *   Rows past the number of valid candidates are filled with id -1 / score -inf.
 * The B x N score matrix is never materialised.
 * ---------------------------------------------------------------------------------- */
/* Test hook of the loose lockstep between the workgroups of an item range (csrc/score_topk.hip): user-tile workgroup `ut` never
 * publishes its progress word, as if it were not resident — its partners run into the bound of the window-edge wait and drop the
 * lockstep; results must not change.  -1 (the default) mutes nobody. */
void lr_score_topk_test_mute(int ut);
size_t lr_score_topk_ws_bytes(int64_t B, int64_t N, int D, int k);
int lr_score_topk_f32(const float* users, int64_t B, const float* items, int64_t N, int D,
                      const int64_t* consumed_ptr /* [B+1] or NULL */,
                      const int32_t* consumed_idx, const uint8_t* filter_flag /* [B] or NULL */,
                      int k, int64_t item_base, float* out_scores /* [B,k] */,
                      int64_t* out_ids /* [B,k] */, void* ws, size_t ws_bytes,
                      lr_stream_t stream);
/* The same contract (same arguments, same workspace) with the scores taken as SPLIT-bf16 products: users and items are each
 * split exactly into three bf16 planes (the items on the fly, when a stage is written to LDS — no second copy of the catalogue),
 * the six largest cross products go through v_mfma_f32_32x32x16_bf16 with f32 accumulation.  Scores equal the f32 chain's to
 * f32 rounding (both within 1e-6 relative of fp64 at D = 128), NOT bit for bit, so the order of items whose fp64 scores differ by
 * less than that can differ.  Reduction widths above 128 run the f32 chain. */
int lr_score_topk_sb_f32(const float* users, int64_t B, const float* items, int64_t N, int D,
                         const int64_t* consumed_ptr, const int32_t* consumed_idx, const uint8_t* filter_flag,
                         int k, int64_t item_base, float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes,
                         lr_stream_t stream);
/* Filtered form (csrc/score_topk.hip, "Filtered scoring"): a one-product bf16 MFMA pass keeps, per user, the
 * k' = lr_score_topk_filter_kp(k) items of largest UPPER BOUND approx + 0.004 |u| |i| >= exact (the term rides in the MFMA chain);
 * their scores are recomputed in f32 and the k best kept; the result of a user is accepted only when its k-th exact score exceeds
 * the k'-th bound, which PROVES that no item outside the k' can belong to the top k — every other user is re-run by the exact
 * kernel (`exact_arith`: 0 the f32 chain, 1 split-bf16; the users concerned are gathered to the front on the device, few of them run
 * under a plan for 128 users) and its rows replaced.  Output contract = lr_score_topk_f32's, for any data (scores are f32 dot products; ids those of the exact ranking
 * up to the order of scores closer than f32 rounding).  Taken at N >= 2^20 items (flags bit 0: at any N), reduction widths 33..128,
 * k <= 100 (kp == 0 above) and k' < N; every other shape runs the exact kernel directly.  `failed_out` (nullable, [B] bytes):
 * 1 where a user went to the exact pass.  Workspace: lr_score_topk_filter_ws_bytes, 256-byte aligned. */
int lr_score_topk_filter_kp(int k);
size_t lr_score_topk_filter_ws_bytes(int64_t B, int64_t N, int D, int k);
int lr_score_topk_filter_f32(const float* users, int64_t B, const float* items, int64_t N, int D,
                             const int64_t* consumed_ptr, const int32_t* consumed_idx, const uint8_t* filter_flag,
                             int k, int64_t item_base, float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes,
                             int exact_arith, int flags, uint8_t* failed_out /* [B] or NULL */, lr_stream_t stream);
/* k-way merge of per-shard results (multi-GPU: after all-gather of [S,B,k] candidates). */
int lr_topk_merge_f32(const float* scores /* [S,B,k] */, const int64_t* ids /* [S,B,k] */,
                      int S, int64_t B, int k, float* out_scores, int64_t* out_ids,
                      lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a12) CSR SpMM — replaces torch.sparse.mm(laplacian_norm, all_embeddings[-1])
 * (algorithms/torch_modules/lightgcn_module.py:80).  Y[r,:] = sum_j val[j] * X[col[j],:].
 * `beta_acc` != NULL fuses the layer mean of lightgcn_module.py:83-84:
 *   acc[r,:] += Y[r,:] (the running sum of E^0..E^L, divided by L+1 by the caller).
 * Y == NULL with acc != NULL: accumulate only (acc += A X, the product's rows are not stored) — the
 * row-sharded net multiplies by one column block of its slice per arrived chunk of X and adds
 * block after block into one table.  Holds for the bucketed / masked forms below as well.
 * Â is symmetric, so the backward is the same operator.
 * ---------------------------------------------------------------------------------- */
int lr_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                    int64_t rows, const float* X, int K, float* Y, float* acc,
                    lr_stream_t stream);
/* Degree-bucketed form of the same product (Zipf-degree graphs): rows of more than 128 nonzeros are cut
 * into 2,048-nonzero chunks listed by a device pre-pass and summed by whole workgroups (every workgroup takes
 * its share of the chunks, then of the short rows); multi-chunk rows are finished in chunk order (fixed summation order per row, no atomics
 * on the data).  `nnz` = rowptr[rows]; workspace from lr_spmm_csr_ws_bytes(rows, nnz, K), 16-byte
 * aligned.  Shapes the vector kernels do not take (K not in {16,32,64,128} or unaligned) run
 * lr_spmm_csr_f32. */
size_t lr_spmm_csr_ws_bytes(int64_t rows, int64_t nnz, int K);
int lr_spmm_csr_bucketed_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows,
                             int64_t nnz, const float* X, int K, float* Y, float* acc, void* ws,
                             size_t ws_bytes, int lists_ready, lr_stream_t stream);
/* `lists_ready` != 0: `ws` still holds the chunk lists of an earlier call with the SAME rowptr (they depend on the
 * graph's row lengths alone) — the classification pre-pass is skipped (LightGCN multiplies by one static graph six times
 * per step).  The chunk-sum scratch inside `ws` is rewritten by every call. */
/* The same product restricted by row bitmaps (uint32 words, bit r of word r / 32; each nullable):
 *   xmask  rows of X whose bit is clear are known to be zero and are not read — the first backward product of a LightGCN
 *          training step multiplies by d loss / d (layer sum), nonzero on the batch's rows only (lightgcn_module.py:66-88
 *          under autograd; torchops/loss.py:bpr_loss reads the batch's rows).  Same bits as the plain product (y + a * 0 == y).
 *   ymask  only the rows of Y whose bit is set are computed and written (the others keep their contents) — the last forward
 *          product of a training step is read at the batch's rows only.
 * K in {16, 32, 64, 128} and 16-byte aligned operands (LR_ESHAPE otherwise).  lr_bitmap_ids_i32 sets (set != 0) the bits of the
 * listed ids or clears their words (set == 0: use it with the list that set them). */
int lr_spmm_csr_masked_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows, int64_t nnz,
                           const float* X, int K, float* Y, float* acc, const uint32_t* xmask, const uint32_t* ymask,
                           void* ws, size_t ws_bytes, int lists_ready, lr_stream_t stream);
int lr_bitmap_ids_i32(const int32_t* ids, int64_t n, int64_t n_bits, uint32_t* bitmap, int set, lr_stream_t stream);
/* The product whose rows ARE the gradient of a parameter table, with the optimiser step as its epilogue (the last backward
 * product of a LightGCN step: libreco/algorithms/torch_modules/lightgcn_module.py:66-88 under autograd, then
 * libreco/training/torch_trainer.py:63-69): for every row r, g = sum_j val[j] X[col[j]] (+ alpha * gsum[row_slot[r]] where
 * row_slot[r] >= 0: the loss's own gradient rows of the batch, summed per distinct row) and one Adam step of (w, m, v[, vmax])
 * row r with g — the arithmetic of lr_embed_scatter_add_f32 + lr_adam_dense_f32 without the gradient table in between.
 * `row_slot` int32[rows] / `gsum` [n, K]: both or neither.  K in {16, 32, 64, 128}, 16-byte aligned (LR_ESHAPE otherwise). */
int lr_spmm_csr_adam_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t rows, int64_t nnz,
                         const float* X, int K, float* w, float* m, float* v, float* vmax, const int32_t* row_slot,
                         const float* gsum, float alpha, lr_adam_hp hp, void* ws, size_t ws_bytes, int lists_ready,
                         lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a19) Pointwise scoring — replaces predict_from_embedding (prediction/predict.py:36-40):
 *   out[i] = <U[user[i],:], I[item[i],:]>
 * ---------------------------------------------------------------------------------- */
int lr_pair_dot_f32(const float* U, int64_t nU, const float* I, int64_t nI, int D,
                    const int32_t* user, const int32_t* item, int64_t n, float* out,
                    lr_stream_t stream);

/* Batch statistics of the gathered block e[B,F,K] (the input of the first BatchNorm,
 * layers/dense.py:30-31) computed from the batch's segments instead of from e:
 *   sum_b e[b,f,k] = sum over field f's runs of len(run) * table[row,k]   (same for squares).
 * `field_row_start[F+1]` (device) gives each field's global row range; the runs must come from
 * lr_segments_build over the same idx.  Writes C partial sums per field:
 * partial[F][C][2][K] = {sum, sum of squares}; the caller adds the C chunks (fixed order).     */
int lr_fm_field_stats_f32(const float* table, int K, const int32_t* seg_rows,
                          const int32_t* seg_start, const int32_t* n_seg,
                          const int32_t* field_row_start, int F, int C, float* partial,
                          lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a12) Normalised bipartite Laplacian on the device — replaces LightGCNModel._build_laplacian_matrix
 * (algorithms/torch_modules/lightgcn_module.py:36-61).  Input: the interaction list (users[e], items[e]), any
 * order, repeats allowed (they collapse, like the reference's dok assignment); pairs outside the id ranges are
 * dropped.  Output: CSR over nodes [users | items] of  A^ = D^-1/2 [[0,R],[R^T,0]] D^-1/2  with ascending columns
 * per row: rowptr [n_users + n_items + 1], col / val [2 * n_pairs] (caller sizes them for 2 * E), the number of
 * distinct pairs in n_pairs[0] (device), and — if tperm != NULL — the transpose map  A^T.val = A.val[tperm]
 * (needed by edge dropout, lightgcn_module.py:90-96).  Integer work bit-exact; val = fp32 product of
 * deg^-1/2 factors rounded from double.  2 * E < 2^31.  Workspace from lr_csr_laplacian_ws_bytes(E).
 * ---------------------------------------------------------------------------------- */
size_t lr_csr_laplacian_ws_bytes(int64_t E);
int lr_csr_laplacian_build(const int32_t* users, const int32_t* items, int64_t E, int64_t n_users,
                           int64_t n_items, int64_t* rowptr, int32_t* col, float* val, int32_t* tperm,
                           int64_t* n_pairs, void* ws, size_t ws_bytes, lr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (a15, device form — SURVEY row f1) Negative sampling on the device with the acceptance rules of
 * sampling/negatives.py:17-31 (random: != positive) and :55-82 (unconsumed: additionally not among
 * the negatives already drawn for that positive and, for the first 10 of 20 tries, not in the
 * user's consumed list), driven by a counter-based generator: out[p*num_neg + j] is a pure
 * function of (seed, p, j).  consumed_ptr/consumed_idx (both or neither): CSR of ascending item
 * ids per user.  The host samplers of the reference use numpy / Python RNG streams; this sampler
 * is opt-in and has its own bit-exact oracle.
 * ---------------------------------------------------------------------------------- */
int lr_sample_negatives_i32(const int32_t* users, const int32_t* items_pos, int64_t n,
                            int num_neg, int32_t n_items, const int64_t* consumed_ptr,
                            const int32_t* consumed_idx, uint64_t seed, int32_t* out,
                            lr_stream_t stream);

/* (f2) MLP tail of a DeepFM (user, item) pair for full-catalog ranking without the feature cross product
 * (libreco/recommendation/recommend.py:81-105, preprocess.py:110-172 materialise one feature row per pair):
 *   out[u][i] (+)= sum_c relu( sum_k relu(P[u][k] + Q[i][k]) * W2[k][c] + b2[c] ) * v3[c] + c3
 * P [B,H1] / Q [N,H1]: user / item part of the first Dense layer; W2 [H1,H2], b2, v3 [H2], c3: the rest of a
 * three-layer `dense_nn` + output weights with the inference BatchNorms folded (recommendation/catalog.py).
 * (H1, H2) in {64,128} x {32,64} (lr_pair_mlp_supported); out row stride ld_out >= N. */
int lr_pair_mlp_supported(int H1, int H2);
int lr_pair_mlp_f32(const float* P, int64_t B, const float* Q, int64_t N, int H1, const float* W2,
                    const float* b2, int H2, const float* v3, float c3, float* out, int64_t ld_out,
                    int accumulate, lr_stream_t stream);
/* The same contract with the H1 x H2 product as six-term split-bf16 products (relu(p + q) split exactly into three bf16 planes
 * per pair on the fly, W2's planes in LDS, f32 accumulation): equal to lr_pair_mlp_f32 to f32 rounding, not bit for bit. */
int lr_pair_mlp_sb_f32(const float* P, int64_t B, const float* Q, int64_t N, int H1, const float* W2,
                       const float* b2, int H2, const float* v3, float c3, float* out, int64_t ld_out,
                       int accumulate, lr_stream_t stream);

/* Measurement probe (scripts/mfma_peak.py): iters x 8 back-to-back v_mfma_f32_32x32x2_f32 per wave on
 * 256 x waves_per_simd workgroups — the f32 MFMA rate the chip sustains at the clock it holds under that load. */
int lr_mfma_f32_probe(int iters, int waves_per_simd, float* out, lr_stream_t stream);
/* Test aid (tests/test_tail_fused_gpu.py): `grid` workgroups that each claim `lds_bytes` of LDS (160 KB: one per CU) and
 * idle for `usec` microseconds — takes residency away from whatever runs beside it on another stream. */
int lr_probe_occupy(int grid, size_t lds_bytes, int64_t usec, lr_stream_t stream);
/* Measurement aid (bench.py): out[2 s] = s_memtime (shader-clock ticks), out[2 s + 1] = s_memrealtime (constant rate) read by one
 * lane on the compute unit of slot s = (XCC id, shader engine, shader array, CU); `out` holds 2 * lr_clock_probe_slots() words
 * the caller zeroes once.  Only differences of the SAME slot between two calls mean anything (the counters of different CUs carry
 * different offsets): two calls around a timed region give the mean shader clock the part sustained over it. */
int lr_clock_probe_slots(void);
int lr_clock_probe(uint64_t* out, lr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LIBRECO_HIP_H_ */
