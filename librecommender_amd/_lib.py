"""ctypes binding of the C-ABI declared in ``include/libreco_hip.h``.

The shared library is plain HIP (no torch types in any signature).  PyTorch is imported first
only so that this process ends up with ONE HIP runtime (torch bundles ``libamdhip64.so.7``; our
library's NEEDED entry resolves to the already-loaded soname).  There is no CPU fallback: if the
library is missing, every op raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "liblibreco_hip.so"

LR_OK, LR_EINVAL, LR_ESHAPE, LR_EWORKSPACE = 0, -1, -2, -3
ABI_VERSION = 25        # == lr_abi_version() of the library these signatures were written for

COMBINERS = {"sum": 0, "mean": 1, "sqrtn": 2}


class AdamHP(C.Structure):
    """Mirror of ``lr_adam_hp``."""

    _fields_ = [
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("step", C.c_int32),
        ("tf_style", C.c_int32),
    ]


class MlpTail3Args(C.Structure):
    """Mirror of ``lr_mlp_tail3_args`` (include/libreco_hip.h): the fused three-layer tail's pointers and shapes."""

    _fields_ = ([("B", C.c_int64), ("K", C.c_int), ("F", C.c_int)]
                + [(n, C.c_void_p) for n in ("z0", "pair", "lin_out", "labels")]
                + [("eps0", C.c_float), ("mom0", C.c_float)]
                + [(n, C.c_void_p) for n in ("mm0", "mv0", "gamma0", "beta0", "dgamma0", "dbeta0")]
                + [("eps1", C.c_float), ("mom1", C.c_float)]
                + [(n, C.c_void_p) for n in ("mm1", "mv1", "gamma1", "beta1", "dgamma1", "dbeta1")]
                + [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "wl", "bl", "wo", "bo", "z1", "z2", "gh0", "gh1",
                                             "stat0", "stat1", "bnp0", "bnp1", "mean0", "inv0", "mean1", "inv1",
                                             "dW1p", "db1p", "dW2p", "db2p", "headp", "gl", "gz0", "sgzp")]
                + [("drop_seed", C.c_uint32), ("keep", C.c_float), ("sync", C.c_void_p)])


_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_int = C.c_int
_f32 = C.c_float
_u32 = C.c_uint32
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/libreco_hip.h
SIGNATURES = {
    "lr_strerror": (C.c_char_p, [_int]),
    "lr_abi_version": (_int, []),
    "lr_graph_foreign_nodes": (_int, [_p, _p]),
    "lr_fm_field_stats_f32": (_int, [_p, _int, _p, _p, _p, _p, _int, _int, _p, _p]),
    "lr_sample_negatives_i32": (_int, [_p, _p, _i64, _int, _i32, _p, _p, C.c_uint64, _p, _p]),
    "lr_embed_gather_f32": (_int, [_p, _i64, _int, _p, _i64, _p, _p]),
    "lr_embed_bag_pool_f32": (_int, [_p, _i64, _int, _p, _i64, _int, _int, _i32, _p, _p]),
    "lr_embed_bag_pool_bwd_f32": (_int, [_p, _int, _p, _i64, _i64, _int, _int, _i32, _p, _p]),
    "lr_segments_ws_bytes": (_sz, [_i64, _i64]),
    "lr_segments_build": (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_embed_scatter_ws_bytes": (_sz, [_i64, _int]),
    "lr_embed_segment_sum_f32": (_int, [_p, _int, _p, _p, _p, _i64, _p, _p, _sz, _p]),
    "lr_embed_scatter_add_f32": (_int, [_p, _i64, _int, _p, _p, _p, _p, _p, _i64, _f32, _p, _sz, _p]),
    "lr_embed_scatter_adam_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _i64,
                                         AdamHP, _p, _sz, _p]),
    "lr_embed_peer_adam_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _int, _p, AdamHP, _p]),
    "lr_embed_scatter_adam_lin_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64,
                                             AdamHP, _p]),
    "lr_adam_dense_f32": (_int, [_p, _p, _p, _p, _i64, _int, _p, _p, _p, _i64, _p, _f32, AdamHP, _p]),
    "lr_fm_pairwise_fwd_f32": (_int, [_p, _i64, _int, _int, _p, _p, _p]),
    "lr_fm_pairwise_bwd_f32": (_int, [_p, _p, _p, _i64, _int, _int, _p, _int, _p]),
    "lr_fm_embed_fwd_f32": (_int, [_p, _p, _i64, _int, _p, _i64, _int, _p, _p, _p, _p, _p]),
    "lr_fm_embed_bwd_ws_bytes": (_sz, [_i64, _int]),
    "lr_fm_embed_bwd_adam_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _p,
                                        _p, _i64, _int, _p, _p, _p, _p, AdamHP, _p, _sz, _p]),
    "lr_fm_embed_bwd_rows_f32": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p,
                                        _p, _p, _p, _sz, _p]),
    "lr_deepfm_l1_supported": (_int, [_int, _int]),
    "lr_deepfm_l1_tile_override": (None, [_int]),
    "lr_deepfm_l1_pack_f32": (_int, [_p, _int, _int, _int, _p, _p, _p]),
    "lr_idx_transpose_i32": (_int, [_p, _i64, _int, _p, _p]),
    "lr_deepfm_l1_fwd_f32": (_int, [_p, _p, _i64, _int, _p, _i64, _int, _p, _p, _int, _p, _p, _p, _p, _p]),
    "lr_deepfm_l1_sb_supported": (_int, [_int, _int]),
    "lr_deepfm_l1_fwd_sb_supported": (_int, [_int, _int]),
    "lr_deepfm_l1_sb_pack_bytes": (_sz, [_int, _int, _int]),
    "lr_deepfm_l1_sb_pack": (_int, [_p, _p, _int, _int, _int, _p, _p, _p, _int, _p, _p]),
    "lr_deepfm_l1_sb_gz_pack_bytes": (_sz, [_i64, _int]),
    "lr_deepfm_l1_sb_gz_pack": (_int, [_p, _i64, _int, _p, _p]),
    "lr_deepfm_l1_sb_override": (None, [_int, _int, _int, _int]),
    "lr_deepfm_l1_fwd_sb_ws_bytes": (_sz, [_i64, _int]),
    "lr_deepfm_l1_fwd_sb_f32": (_int, [_p, _p, _i64, _int, _p, _i64, _int, _p, _p, _int, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_deepfm_l1_wgrad_sb_chunks": (_int, [_i64, _int]),
    "lr_deepfm_l1_wgrad_sb_f32": (_int, [_p, _i64, _int, _p, _i64, _int, _p, _int, _int, _p, _p]),
    "lr_deepfm_l1_dgrad_sb_f32": (_int, [_p, _int, _p, _int, _int, _i64, _p, _p, _p, _p, _p, _p]),
    "lr_deepfm_l1_wgrad_chunks": (_int, [_i64, _int]),
    "lr_deepfm_l1_wgrad_f32": (_int, [_p, _i64, _int, _p, _i64, _int, _p, _int, _int, _p, _p]),
    "lr_deepfm_l1_dgrad_f32": (_int, [_p, _int, _p, _int, _int, _i64, _p, _p, _p, _p, _p, _p]),
    "lr_fm_rows_adam_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _i64,
                                   _int, _p, _p, _p, _p, AdamHP, _p, _sz, _p]),
    "lr_mlp_tail_supported": (_int, [_int, _int]),
    "lr_mlp_tail3_supported": (_int, [_int, _int, _int, _int, _int]),
    "lr_mlp_tail3_f32": (_int, [C.POINTER(MlpTail3Args), _p]),
    "lr_mlp_colstats_f32": (_int, [_p, _i64, _int, _p, _p]),
    "lr_mlp_bn_finalize_f32": (_int, [_p, _int, _int, _i64, _f32, _f32, _p, _p, _p, _p, _p]),
    "lr_mlp_layer_fwd_f32": (_int, [_p, _i64, _int, _p, _p, _p, _p, _p, _p, _int, _p, _p, _u32, _f32, _int, _p]),
    "lr_mlp_head_f32": (_int, [_p, _int, _p, _int, _p, _int, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "lr_mlp_layer_bwd_f32": (_int, [_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _int,
                                    _i64, _p, _p, _p, _p, _u32, _f32, _int, _p]),
    "lr_mlp_first_bwd_f32": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p]),
    "lr_reduce_partials_f32": (_int, [_p, _int, _i64, _i64, _p, _p]),
    "lr_deepfm_l1_fold_stats_f32": (_int, [_p, _int, _int, _int, _i64, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lr_deepfm_l1_pack_scaled_f32": (_int, [_p, _p, _int, _int, _int, _p, _p, _p, _int, _p, _p]),
    "lr_deepfm_l1_fold_stats_bias_f32": (_int, [_p, _int, _int, _int, _i64, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int,
                                                _p, _p]),
    "lr_deepfm_l1_fold_bias_slabs": (_int, [_int]),
    "lr_deepfm_l1_fold_bias_f32": (_int, [_p, _p, _p, _int, _int, _p, _p]),
    "lr_deepfm_l1_fold_bwd_f32": (_int, [_p, _int, _int, _int, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lr_fm_field_stats_slots_f32": (_int, [_p, _int, _p, _p, _p, _p, _int, _int, _p, _p, _p, _p]),
    "lr_fm_rows_grad_f32": (_int, [_p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _p,
                                   _sz, _p]),
    "lr_fm_rows_grad_compact_f32": (_int, [_p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _p,
                                           _sz, _p]),
    "lr_adam_dense_rows_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _i64, _p, AdamHP, _p]),
    "lr_adam_dense_rows_dc_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "lr_spmm_csr_ws_bytes": (_sz, [_i64, _i64, _int]),
    "lr_spmm_csr_bucketed_f32": (_int, [_p, _p, _p, _i64, _i64, _p, _int, _p, _p, _p, _sz, _int, _p]),
    "lr_spmm_csr_masked_f32": (_int, [_p, _p, _p, _i64, _i64, _p, _int, _p, _p, _p, _p, _p, _sz, _int, _p]),
    "lr_bitmap_ids_i32": (_int, [_p, _i64, _i64, _p, _int, _p]),
    "lr_row_slots_i32": (_int, [_p, _p, _i64, _p, _int, _p]),
    "lr_spmm_csr_adam_f32": (_int, [_p, _p, _p, _i64, _i64, _p, _int, _p, _p, _p, _p, _p, _p, _f32, AdamHP, _p, _sz, _int, _p]),
    "lr_din_build_ids_i32": (_int, [_p, _p, _p, _int, _p, _int, _p, _p, _i64, _int, _int, _int, _int, _p, _p]),
    "lr_softmax_ce_supported": (_int, [_i64, _i64, _int]),
    "lr_softmax_ce_fwd_ws_bytes": (_sz, [_i64, _i64, _int, _int]),
    "lr_softmax_ce_fwd_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _p, _i64, _p, _p, _p, _p, _sz, _int, _p]),
    "lr_softmax_ce_bwd_cols_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _p, _i64, _p, _p, _p, _int, _p]),
    "lr_reduce_job_bytes": (_sz, []),
    "lr_reduce_partials_multi_f32": (_int, [_p, _int, _i64, _p]),
    "lr_pair_mlp_supported": (_int, [_int, _int]),
    "lr_pair_mlp_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _int, _p, _f32, _p, _i64, _int, _p]),
    "lr_pair_mlp_sb_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _int, _p, _f32, _p, _i64, _int, _p]),
    "lr_score_topk_test_mute": (None, [_int]),
    "lr_mfma_f32_probe": (_int, [_int, _int, _p, _p]),
    "lr_probe_occupy": (_int, [_int, _sz, _i64, _p]),
    "lr_clock_probe": (_int, [_p, _p]),
    "lr_clock_probe_slots": (_int, []),
    "lr_mlp_tail3_resident_blocks": (_int, []),
    "lr_adam_coef_bytes": (_sz, []),
    "lr_adam_coef_store": (_int, [AdamHP, _p, _p]),
    "lr_adam_dense_dc_f32": (_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "lr_fm_rows_adam_dc_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _i64,
                                      _int, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_embed_scatter_adam_dc_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _i64, _p, _p, _sz, _p]),
    "lr_embed_scatter_adam_lin_dc_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "lr_table_colstats_f32": (_int, [_p, _i64, _int, _p, _i64, _int, _int, _p, _p]),
    "lr_bn_remainder_f32": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "lr_csr_laplacian_ws_bytes": (_sz, [_i64]),
    "lr_csr_laplacian_build": (_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_segments_fields_ws_bytes": (_sz, [_i64, _int]),
    "lr_segments_build_fields": (_int, [_p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_segments_build_fields_runs": (_int, [_p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_owner_partition_ws_bytes": (_sz, [_i64, _int]),
    "lr_owner_partition_i32": (_int, [_p, _p, _i64, _int, _p, _p, _p, _p, _sz, _p]),
    "lr_din_attn_ws_bytes": (_sz, [_i64, _int, _int, _int]),
    "lr_din_attn_pool_fwd_f32": (_int, [_p, _i64, _int, _p, _p, _p, _i64, _int, _p, _p, _p, _p,
                                        _int, _p, _p, _p, _p, _p]),
    "lr_din_attn_pool_bwd_f32": (_int, [_p, _i64, _int, _p, _p, _p, _i64, _int, _p, _p, _p, _p,
                                        _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_din_attn_pool_bwd_parts_f32": (_int, [_p, _i64, _int, _p, _p, _p, _i64, _int, _p, _p, _p, _p,
                                              _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _int, _int, _p, _p, _p]),
    "lr_din_attn_dense_fwd_f32": (_int, [_p, _p, _int, _p, _i64, _int, _p, _p, _p, _p, _int, _p,
                                         _p, _p]),
    "lr_din_attn_dense_bwd_f32": (_int, [_p, _p, _int, _p, _i64, _int, _p, _p, _p, _p, _int, _p,
                                         _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lr_score_topk_ws_bytes": (_sz, [_i64, _i64, _int, _int]),
    "lr_score_topk_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _p, _int, _i64, _p, _p, _p,
                                 _sz, _p]),
    "lr_score_topk_sb_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _p, _int, _i64, _p, _p, _p,
                                 _sz, _p]),
    "lr_score_topk_filter_kp": (_int, [_int]),
    "lr_score_topk_filter_ws_bytes": (_sz, [_i64, _i64, _int, _int]),
    "lr_score_topk_filter_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _p, _int, _i64, _p, _p, _p,
                                        _sz, _int, _int, _p, _p]),
    "lr_topk_merge_f32": (_int, [_p, _p, _int, _i64, _int, _p, _p, _p]),
    "lr_spmm_csr_f32": (_int, [_p, _p, _p, _i64, _p, _int, _p, _p, _p]),
    "lr_pair_dot_f32": (_int, [_p, _i64, _p, _i64, _int, _p, _p, _i64, _p, _p]),
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load(path: os.PathLike | None = None) -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get("LIBRECO_HIP_LIB"):   # profiling builds (ablations)
        path = os.environ["LIBRECO_HIP_LIB"]
    p = Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise HipExtensionMissing(
            f"{p} not found: build it with `python -m librecommender_amd.csrc.build` "
            "(there is no CPU fallback for the MI355X hot path)"
        )
    try:  # make sure torch's bundled HIP runtime (same soname) is the one in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less C users load the lib themselves
        pass
    lib = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.lr_abi_version() != ABI_VERSION:
        raise HipExtensionMissing(
            f"{p} has ABI version {lib.lr_abi_version()}, this package needs {ABI_VERSION}: "
            "rebuild with `python -m librecommender_amd.csrc.build`")
    if path is None:
        _lib = lib
    return lib


def strerror(code: int) -> str:
    return load().lr_strerror(int(code)).decode()


def check(code: int, what: str = "") -> None:
    """Translate a C-ABI status into the exception type the reference's callers expect."""
    if code == LR_OK:
        return
    msg = f"{what}: {strerror(code)} (status {code})" if what else f"{strerror(code)} ({code})"
    if code in (LR_EINVAL, LR_ESHAPE):
        raise ValueError(msg)
    raise RuntimeError(msg)
