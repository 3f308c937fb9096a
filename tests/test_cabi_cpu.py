"""The C-ABI library loads without a GPU and exports exactly what include/libreco_hip.h
declares; host-only entry points behave (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from librecommender_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "libreco_hip.h").read_text()


def header_symbols():
    code = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", code)))


def test_header_declares_the_whole_binding_table():
    syms = header_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_host_only_entry_points():
    lib = _lib.load()
    from librecommender_amd import _lib as L
    assert lib.lr_abi_version() == L.ABI_VERSION == 25
    assert lib.lr_csr_laplacian_ws_bytes(1000) >= 1000 * 48
    assert lib.lr_strerror(0) == b"ok"
    assert b"invalid" in lib.lr_strerror(_lib.LR_EINVAL)
    assert b"workspace" in lib.lr_strerror(_lib.LR_EWORKSPACE)
    assert lib.lr_segments_ws_bytes(1000, 50) >= 3 * 4 * 1000
    assert lib.lr_score_topk_ws_bytes(1024, 1_000_000, 128, 100) > 0
    assert lib.lr_score_topk_ws_bytes(4, 100, 6, 10) == 0       # D % 4 != 0: the host must pad
    assert lib.lr_score_topk_ws_bytes(4, 100, 16, 5000) == 0    # k > 4096 unsupported
    # the filtered form: k' = 2k + 56 (>= 64, multiple of 8) up to k = 100; its workspace holds the exact pass's too
    assert [lib.lr_score_topk_filter_kp(k) for k in (1, 10, 100, 101)] == [64, 80, 256, 0]
    assert lib.lr_score_topk_filter_ws_bytes(1024, 100_000_000, 128, 100) > lib.lr_score_topk_ws_bytes(1024, 100_000_000, 128, 100)
    assert lib.lr_score_topk_filter_ws_bytes(1024, 1_000_000, 16, 100) >= lib.lr_score_topk_ws_bytes(1024, 1_000_000, 16, 100) > 0
    assert lib.lr_din_attn_ws_bytes(8192, 50, 128, 16) > 0
    assert lib.lr_din_attn_ws_bytes(8192, 50, 128, 8) == 0      # H is 16 in the reference
    assert lib.lr_deepfm_l1_supported(64, 128) == 1 and lib.lr_deepfm_l1_supported(48, 128) == 0
    assert 1 <= lib.lr_deepfm_l1_wgrad_chunks(16384, 202) <= 16
    # round 5: the split-bf16 first-layer kernels (host-side size / shape queries)
    assert lib.lr_deepfm_l1_sb_supported(64, 128) == 1 and lib.lr_deepfm_l1_sb_supported(32, 128) == 0
    assert lib.lr_deepfm_l1_sb_pack_bytes(202, 64, 128) == 202 * 64 * 128 * 6
    assert lib.lr_deepfm_l1_sb_gz_pack_bytes(16384, 128) == 16384 * 128 * 6 and lib.lr_deepfm_l1_sb_gz_pack_bytes(17, 128) == 32 * 128 * 6
    assert lib.lr_deepfm_l1_wgrad_sb_chunks(16384, 202) == 5          # 51 groups of four fields x 5 chunks: one workgroup per CU
    assert lib.lr_deepfm_l1_fwd_sb_ws_bytes(16384, 202) == 2 * 16384 * 256 * 4   # 128 tiles x 2 field groups fill the chip
    assert lib.lr_segments_fields_ws_bytes(16384, 202) >= 16384 * 202 * 8


def test_argument_errors_map_to_reference_exception_types():
    with pytest.raises(ValueError):
        _lib.check(_lib.LR_EINVAL, "x")
    with pytest.raises(ValueError):
        _lib.check(_lib.LR_ESHAPE, "x")
    with pytest.raises(RuntimeError):
        _lib.check(_lib.LR_EWORKSPACE, "x")
    with pytest.raises(RuntimeError):
        _lib.check(700, "x")  # a hipError_t


def test_missing_extension_fails_loudly(tmp_path):
    with pytest.raises(_lib.HipExtensionMissing, match="no CPU fallback"):
        _lib.load(tmp_path / "nope.so")


def lib_sizes_agree():
    lib = _lib.load()
    return lib.lr_mlp_tail3_supported(128, 64, 32, 64, 202) == 1 and lib.lr_mlp_tail3_supported(128, 64, 16, 64, 202) == 0


def test_adam_struct_layout_matches_header():
    assert C.sizeof(_lib.AdamHP) == 5 * 8 + 2 * 4
    # lr_mlp_tail3_args: int64 + 2 ints, 44 pointers, 6 floats / uint32 in aligned pairs (the library asserts the same size)
    assert C.sizeof(_lib.MlpTail3Args) == 8 + 8 + 4 * 8 + 2 * (8 + 6 * 8) + 28 * 8 + 8 + 8
    assert lib_sizes_agree()


def test_scoring_seam_functions_are_exported_and_refuse_to_run_without_a_device():
    """`recommendation.rank_recommendations` / `recommend_from_embedding` are part of the seam; the argument
    check comes first (reference KAT), and without a HIP device the call fails loudly instead of ranking on the host."""
    import numpy as np
    import pytest
    import torch

    from librecommender_amd import recommendation as rec

    for name in ("rank_recommendations", "recommend_from_embedding", "cold_start_rec", "popular_recommendations",
                 "construct_rec", "check_dynamic_rec_feats"):
        assert callable(getattr(rec, name))
    with pytest.raises(ValueError):
        rec.rank_recommendations("ranking", [1, 2], np.zeros(10), 12, 5, {})
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            rec.rank_recommendations("ranking", [1, 2], np.zeros(10), 2, 5, {})
