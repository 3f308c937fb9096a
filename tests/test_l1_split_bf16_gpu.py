"""The fused first layer's three contractions as split-bf16 MFMA products (csrc/deepfm_l1_sb.hip; round 5: the default
arithmetic where the shape is compiled, K = 64 / H1 = 128) — reference: algorithms/deepfm.py:155-170, layers/dense.py:12-49.

Every result is pinned against the fp64 restatements of `oracle/ops_np.py`, at the SAME tolerances the f32-chain kernels are
held to in tests/test_deepfm_fused_gpu.py (rtol 1e-5, atol 1e-5 x the accumulated magnitude), and against the f32-chain
kernels' own error on the same inputs (the split products must not be further from fp64 than 1.5 x the f32 fma chain, or
3e-7 relative rms — two f32 roundings — where both are that small).
Index work (linear weights looked up, dropped positions, run order) is bit-exact.  Every grid / staging variant the
library can choose (`lr_deepfm_l1_sb_override`) is exercised: 64- and 128-sample tiles, 1-8 field groups, 4 and 8 multiplying
waves and 2 and 4 fields per weight-gradient workgroup."""
import numpy as np
import pytest
import torch

from librecommender_amd import _lib, ops
from oracle import ops_np
from tests.test_deepfm_fused_gpu import make_case, t

pytestmark = pytest.mark.gpu

K, H1 = 64, 128


@pytest.fixture
def override():
    lib = _lib.load()

    def pin(fwd_tile=0, ksplit=0, wgrad_cw=0, wgrad_fg=0):
        lib.lr_deepfm_l1_sb_override(int(fwd_tile), int(ksplit), int(wgrad_cw), int(wgrad_fg))
    yield pin
    lib.lr_deepfm_l1_sb_override(0, 0, 0, 0)


def planes_to_f64(buf, n_slots):
    """[n_slots, 3 planes, 64 lanes, 8 bf16] -> fp64 sums of the three planes [n_slots, 64, 8]."""
    p = buf[: n_slots * 3 * 64 * 16].view(torch.bfloat16).view(n_slots, 3, 64, 8).double()
    return p.sum(1)


def test_pack_layouts_and_exact_split(dev):
    """Documented fragment orders (include/libreco_hip.h); the three planes of every element add up to the f32 value
    EXACTLY (the split loses nothing), with and without the BatchNorm row scale."""
    assert ops.deepfm_l1_sb_supported(K, H1) and not ops.deepfm_l1_sb_supported(32, 128) and not ops.deepfm_l1_sb_supported(64, 256)
    rng = np.random.default_rng(0)
    F = 3
    W = t((rng.standard_normal((F * K, H1)) * rng.choice([1e-3, 1.0, 37.0], (F * K, H1))).astype(np.float32), dev)
    scale = t((rng.standard_normal(F * K) * 0.7 + 1.5).astype(np.float32), dev)
    for sc in (None, scale):
        A, Bp = ops.deepfm_l1_pack(W, F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="split_bf16"), scale=sc)
        assert A.dtype == torch.uint8 and Bp.dtype == torch.uint8
        Wp = (W * sc[:, None]) if sc is not None else W           # formed in f32, like lr_deepfm_l1_pack_scaled_f32
        KB, CT = K // 16, H1 // 32
        a = planes_to_f64(A, F * KB * CT).view(F, KB, CT, 2, 32, 8)             # f, kb, ct, g, j, e
        want = Wp.double().view(F, KB, 2, 8, CT, 32).permute(0, 1, 4, 2, 5, 3)  # [f, kb, g, e, ct, j] -> f, kb, ct, g, j, e
        assert torch.equal(a, want.contiguous())
        KBH, NT = H1 // 16, K // 32
        b = planes_to_f64(Bp, F * KBH * NT).view(F, KBH, NT, 2, 32, 8)          # f, kb, nt, g, j, e
        wantb = Wp.double().view(F, NT, 32, KBH, 2, 8).permute(0, 3, 1, 4, 2, 5)  # [f, nt, j, kb, g, e] -> f, kb, nt, g, j, e
        assert torch.equal(b, wantb.contiguous())
    # gz planes: [slab][nt][plane][lane][8], samples padded with zeros to a multiple of 16
    Bn = 37
    gz = t(rng.standard_normal((Bn, H1)).astype(np.float32), dev)
    n = _lib.load().lr_deepfm_l1_sb_gz_pack_bytes(Bn, H1)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    ops._call("lr_deepfm_l1_sb_gz_pack", ops._ptr(gz), Bn, H1, ops._ptr(out), ops._stream())
    slabs = (Bn + 15) // 16
    g = planes_to_f64(out, slabs * 4).view(slabs, 4, 2, 32, 8)                  # slab, nt, g, j, e
    pad = torch.zeros((slabs * 16, H1), dtype=torch.float64, device=dev)
    pad[:Bn] = gz.double()
    wantg = pad.view(slabs, 2, 8, 4, 32).permute(0, 3, 1, 4, 2)                 # [slab, g, e, nt, j] -> slab, nt, g, j, e
    assert torch.equal(g, wantg.contiguous())


def rel_rms(a, ref):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - ref) ** 2)) / (np.sqrt(np.mean(ref ** 2)) + 1e-300))


FWD_MODES = [dict(), dict(fwd_tile=64), dict(fwd_tile=128, ksplit=1), dict(fwd_tile=128, ksplit=2), dict(fwd_tile=128, ksplit=3),
             dict(fwd_tile=64, ksplit=8)]


@pytest.mark.parametrize("B,F", [(64, 1), (100, 2), (257, 3), (129, 5), (1000, 23), (320, 70), (2048, 202)])
def test_fwd_against_fp64_and_the_f32_chain(dev, override, B, F):
    rng = np.random.default_rng(B * 7 + F)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    o_z1, o_pair, o_fsum, o_lin = ops_np.deepfm_l1_fwd(table, lin, idx, Wp, bias)
    scale = float(np.abs(o_z1).max()) + 1.0
    Wf = ops.deepfm_l1_pack(t(Wp, dev), F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="f32_chain"))[0]
    z_f32 = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), Wf, t(bias, dev), H1, lin=t(lin, dev))[0]
    err_f32 = rel_rms(z_f32.cpu().numpy(), o_z1)
    Ws = ops.deepfm_l1_pack(t(Wp, dev), F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="split_bf16"))[0]
    for mode in FWD_MODES:
        if mode.get("ksplit", 1) > F:
            continue
        override(**mode)
        z1, pair, fsum, lin_out = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), Ws, t(bias, dev), H1, lin=t(lin, dev))
        np.testing.assert_allclose(z1.cpu().numpy(), o_z1, rtol=1e-5, atol=1e-5 * scale, err_msg=str(mode))
        np.testing.assert_allclose(fsum.cpu().numpy(), o_fsum, rtol=1e-5, atol=1e-5, err_msg=str(mode))
        np.testing.assert_allclose(pair.cpu().numpy(), o_pair, rtol=1e-4, atol=1e-4, err_msg=str(mode))
        np.testing.assert_array_equal(lin_out.cpu().numpy(), o_lin.astype(np.float32), err_msg=str(mode))
        err = rel_rms(z1.cpu().numpy(), o_z1)
        assert err <= max(1.5 * err_f32, 3e-7), (mode, err, err_f32)
        # without the linear table and without a bias
        z1b, _, _, lb = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), Ws, None, H1)
        assert lb is None
        np.testing.assert_allclose(z1b.cpu().numpy(), o_z1 - bias, rtol=1e-5, atol=1e-5 * scale, err_msg=str(mode))
        # run-to-run identical (no atomics; the field-group sums are added in a fixed order)
        z1c = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), Ws, t(bias, dev), H1, lin=t(lin, dev))[0]
        assert torch.equal(z1, z1c)


@pytest.mark.parametrize("B,F,nch", [(64, 2, 1), (300, 3, 2), (1000, 5, 5), (1000, 5, None), (33, 1, 3), (4100, 7, None), (2048, 202, None)])
def test_wgrad_against_fp64_and_the_f32_chain(dev, override, B, F, nch):
    rng = np.random.default_rng(B + F)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    gz = rng.standard_normal((B, H1)).astype(np.float32)
    idxT = ops.idx_transpose(t(idx, dev))
    want = ops_np.deepfm_l1_wgrad(table, idx, gz)
    tol = dict(rtol=1e-5, atol=1e-5 * (float(np.abs(want).max()) + 1.0))
    ref = ops.deepfm_l1_wgrad(t(table, dev), idxT, t(gz, dev), n_chunks=nch, arith="f32_chain").double().sum(0).cpu().numpy()
    err_f32 = rel_rms(ref, want)
    for fg, cw in ((2, 8), (2, 4), (4, 4)):
        override(wgrad_fg=fg, wgrad_cw=cw)
        part = ops.deepfm_l1_wgrad(t(table, dev), idxT, t(gz, dev), n_chunks=nch, arith="split_bf16")
        got = part.double().sum(0).cpu().numpy()
        np.testing.assert_allclose(got, want, err_msg=f"FG={fg} CW={cw}", **tol)
        assert rel_rms(got, want) <= max(1.5 * err_f32, 3e-7), (fg, rel_rms(got, want), err_f32)
        part2 = ops.deepfm_l1_wgrad(t(table, dev), idxT, t(gz, dev), n_chunks=nch, arith="split_bf16")
        assert torch.equal(part, part2)


@pytest.mark.parametrize("B,F", [(64, 1), (130, 3), (1000, 9), (257, 70), (2048, 202)])
def test_dgrad_against_fp64_and_the_f32_chain(dev, override, B, F):
    rng = np.random.default_rng(B + 3 * F)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    gz = rng.standard_normal((B, H1)).astype(np.float32)
    gl = rng.standard_normal(B).astype(np.float32)
    wp = rng.standard_normal(K).astype(np.float32)
    fsum = rng.standard_normal((B, K)).astype(np.float32)
    _, _, _, slotT = ops_np.segments_fields(idx, frs)
    want = ops_np.deepfm_l1_dgrad(gz, Wp, K, gl, wp, fsum, slotT)
    want2 = ops_np.deepfm_l1_dgrad(gz, Wp, K, None, None, None, slotT)
    tol = dict(rtol=1e-5, atol=1e-5 * (float(np.abs(want).max()) + 1.0))
    args = (t(gz, dev),)
    kw = dict(gl=t(gl, dev), wp=t(wp, dev), fsum=t(fsum, dev))
    WBf = ops.deepfm_l1_pack(t(Wp, dev), F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="f32_chain"))[1]
    out = torch.zeros((B * F + 1, K), device=dev)
    ref = ops.deepfm_l1_dgrad(*args, WBf, K, F, t(slotT, dev), out=out, **kw)[:B * F].cpu().numpy()
    err_f32 = rel_rms(ref, want)
    WBs = ops.deepfm_l1_pack(t(Wp, dev), F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="split_bf16"))[1]
    for mode in (dict(), dict(ksplit=1), dict(ksplit=3), dict(ksplit=8)):
        if mode.get("ksplit", 1) > F:
            continue
        override(**mode)
        out.zero_()
        ge = ops.deepfm_l1_dgrad(*args, WBs, K, F, t(slotT, dev), out=out, **kw)[:B * F]
        np.testing.assert_allclose(ge.cpu().numpy(), want, err_msg=str(mode), **tol)
        assert rel_rms(ge.cpu().numpy(), want) <= max(1.5 * err_f32, 3e-7), (mode, rel_rms(ge.cpu().numpy(), want), err_f32)
        keep = ge.clone()
        out.zero_()
        ge2 = ops.deepfm_l1_dgrad(*args, WBs, K, F, t(slotT, dev), out=out)[:B * F]          # no FM term
        np.testing.assert_allclose(ge2.cpu().numpy(), want2, err_msg=str(mode), **tol)
        out.zero_()
        assert torch.equal(ops.deepfm_l1_dgrad(*args, WBs, K, F, t(slotT, dev), out=out, **kw)[:B * F], keep)


def test_split_bf16_is_the_default_where_compiled_and_the_f32_chain_stays_selectable(dev):
    from librecommender_amd.nets import DeepFMNet

    assert ops.L1_ARITH == "split_bf16"
    kw = dict(hidden_units=(128, 64, 32), device=dev, sparse_offsets=np.arange(4) * 11)
    a = DeepFMNet(50, 40, 44, 4, embed_size=64, **kw)
    assert a.fused_l1 and a.l1_arith == "split_bf16"
    b = DeepFMNet(50, 40, 44, 4, embed_size=32, **kw)                 # a shape only the f32 chain is compiled for
    assert b.fused_l1 and b.l1_arith == "f32_chain"
    prev = ops.set_l1_arith("f32_chain")
    try:
        c = DeepFMNet(50, 40, 44, 4, embed_size=64, **kw)
        assert c.l1_arith == "f32_chain"
    finally:
        ops.set_l1_arith(prev)
    assert a.l1_arith == "split_bf16"                                 # a net keeps the arithmetic of its buffers
    with pytest.raises(ValueError):
        ops.set_l1_arith("bf16")


def test_unsupported_shapes_are_refused(dev):
    table = torch.zeros((10, 32), device=dev)
    idx = torch.zeros((4, 2), dtype=torch.int32, device=dev)
    planes = torch.zeros(2 * 32 * 128 * 6, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.deepfm_l1_fwd(table, idx, planes, None, 128)
    with pytest.raises(ValueError):
        ops.deepfm_l1_pack(torch.zeros((64, 128), device=dev), 2, 32, out=(planes, planes))
