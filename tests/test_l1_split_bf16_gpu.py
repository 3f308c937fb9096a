"""EXPERIMENTAL opt-in kernel `lr_deepfm_l1_fwd_sb_f32` (the first layer's forward as split-bf16 MFMA products): against the f32
kernel it stands beside and against an f64 reference.  Tolerances (written here, pinned against f64 rather than against the f32
chain): z1 within 2e-5 of the f64 result relative to the result's rms, and not worse than 1.5x the f32 kernel's own error; the
FM sums / pairwise term / gathered linear weights are plain f32 work shared with the f32 kernel: bit-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,F", [(64, 1), (100, 3), (1000, 37), (4096, 202)])
def test_split_bf16_forward_matches_f64_as_well_as_the_f32_kernel(dev, B, F):
    from librecommender_amd import ops

    K, H1, V = 64, 128, 5000
    assert ops.deepfm_l1_sb_supported(K, H1)
    g = torch.Generator().manual_seed(B + F)
    table = (torch.randn((V, K), generator=g) * 0.1).to(dev)
    lin = (torch.randn((V, 1), generator=g) * 0.1).to(dev)
    W = (torch.randn((F * K, H1), generator=g) * 0.05).to(dev)
    scale = (0.5 + torch.rand(F * K, generator=g)).to(dev)
    bias = torch.randn(H1, generator=g).to(dev)
    idx = torch.randint(0, V, (B, F), generator=g).to(torch.int32)
    idx[0, 0] = -1                                   # an id outside the table: a zero row, as in the f32 kernel
    if B > 10 and F > 1:
        idx[7, F - 1] = V
    idx = idx.to(dev)
    Ws = W * scale[:, None]
    WpA, _ = ops.deepfm_l1_pack(Ws, F, K)
    z_f32, pair_a, fsum_a, lin_a = ops.deepfm_l1_fwd(table, idx, WpA, bias, H1, lin=lin)
    Wsb = ops.deepfm_l1_sb_pack(W, F, K, scale=scale)
    z_sb, pair_b, fsum_b, lin_b = ops.deepfm_l1_fwd_sb(table, idx, Wsb, bias, H1, lin=lin)
    assert torch.equal(pair_a, pair_b) and torch.equal(fsum_a, fsum_b) and torch.equal(lin_a, lin_b)
    ok = ((idx >= 0) & (idx < V)).to(torch.float64)
    rows = table.double()[idx.clamp(0, V - 1).long()] * ok[:, :, None]             # [B, F, K]
    ref = rows.reshape(B, F * K) @ Ws.double() + bias.double()
    rms = float(ref.pow(2).mean().sqrt())
    e_f32 = float((z_f32.double() - ref).pow(2).mean().sqrt()) / rms
    e_sb = float((z_sb.double() - ref).pow(2).mean().sqrt()) / rms
    assert e_sb < 2e-5, (e_sb, e_f32)
    assert e_sb <= 1.5 * e_f32 + 1e-7, (e_sb, e_f32)
    # (the scale is folded before the split: packing the pre-scaled kernel gives the same planes)
    z_sb2, *_ = ops.deepfm_l1_fwd_sb(table, idx, ops.deepfm_l1_sb_pack(Ws, F, K), bias, H1, lin=lin)
    assert torch.equal(z_sb, z_sb2)


def test_split_bf16_refuses_other_shapes(dev):
    from librecommender_amd import ops

    assert not ops.deepfm_l1_sb_supported(32, 128) and not ops.deepfm_l1_sb_supported(64, 256)
    table = torch.zeros((10, 32), device=dev)
    idx = torch.zeros((4, 2), dtype=torch.int32, device=dev)
    Wsb = ops.deepfm_l1_sb_pack(torch.zeros((2 * 32, 128), device=dev), 2, 32)
    with pytest.raises(Exception):
        ops.deepfm_l1_fwd_sb(table, idx, Wsb, None, 128)
