"""The reference's optimiser semantics on the fused DeepFM step (`dense_adam=True`; training/tf_trainer.py:120 —
tf.train.AdamOptimizer moves EVERY row of a table every step): the per-row gradient kernel in compact form
(`lr_fm_rows_grad_compact_f32`) + ONE streaming pass over both tables (`lr_adam_dense_rows_f32`).

Kernel level against the numpy restatements (oracle/ops_np.py); model level against the fp64 oracle over several steps (where
untouched rows keep moving on their moments) and against the same model replayed as a hipGraph."""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from librecommender_amd.nets import DeepFMNet
from oracle import ops_np
from oracle.models_torch import DeepFMOracle, export_fieldnet_weights
from tests.test_deepfm_fused_gpu import field_ids, t
from tests.test_fm_models_gpu import cpu_batch, make_batch, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [16, 64, 128])
@pytest.mark.parametrize("with_bn", [True, False])
def test_rows_grad_compact_and_dense_table_pass(dev, K, with_bn):
    rng = np.random.default_rng(K + with_bn)
    B, F = 700, 4
    frs = np.array([0, 3, 40, 41, 300])              # a 3-row field (runs of > 200 positions), a 1-row field
    V = int(frs[-1])
    idx = field_ids(rng, frs, B)
    idx[9, 2] = -1
    pos, rows, start, _ = ops_np.segments_fields(idx, frs)
    ge = np.zeros((B * F, K), np.float32)
    ge[:len(pos)] = rng.standard_normal((len(pos), K)).astype(np.float32)
    gl = rng.standard_normal(B).astype(np.float32) * 0.1
    wp = rng.standard_normal(K).astype(np.float32)
    a = rng.standard_normal(F * K).astype(np.float32) * 0.1 if with_bn else None
    c = rng.standard_normal(F * K).astype(np.float32) * 0.1 if with_bn else None
    lin_scale = rng.standard_normal(F).astype(np.float32)
    w0 = rng.standard_normal((V, K)).astype(np.float32)
    m0 = (rng.standard_normal((V, K)) * 0.01).astype(np.float32)
    v0 = (rng.random((V, K)) * 0.01).astype(np.float32)
    l0 = rng.standard_normal((V, 1)).astype(np.float32)
    lm0, lv0 = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32), (rng.random((V, 1)) * 0.01).astype(np.float32)
    sb = ops.FieldSegmentBuilder(B, F, V, dev)
    seg = sb.build(ops.idx_transpose(t(idx, dev)), t(frs.astype(np.int32), dev))
    w, m, v, l, lm, lv = (t(x.copy(), dev) for x in (w0, m0, v0, l0, lm0, lv0))
    kw = dict(bn_a=None if a is None else t(a, dev), bn_c=None if c is None else t(c, dev), lin_scale=t(lin_scale, dev))
    grows, glin = ops.fm_rows_grad_compact(w, l, t(ge, dev), seg, B, F, t(gl, dev), t(wp, dev), **kw)
    ns = seg.count()
    g, sgl = ops_np.fm_rows_gradient(w0, ge, pos, rows, start, F, gl, wp, a, c)
    assert ns == len(rows) and (np.diff(start) > 32).sum() >= 3          # the whole-workgroup path ran
    np.testing.assert_allclose(grows[:ns].cpu().numpy(), g, rtol=2e-5, atol=2e-5)
    fld = pos[start[:-1]] % F
    gl_rows = (sgl * lin_scale[fld]).astype(np.float32)
    np.testing.assert_allclose(glin[:ns].cpu().numpy(), gl_rows, rtol=2e-5, atol=1e-6)
    # the same numbers as the slot form used by the row-sharded step, read at the rows' own slots
    slots = t(np.where((idx >= 0), idx, 0).astype(np.int32).reshape(-1), dev)
    full, full_lin = ops.fm_rows_grad(w, l, t(ge, dev), seg, B, F, slots, t(gl, dev), t(wp, dev), **kw)
    assert torch.equal(full[seg.rows[:ns].long()], grows[:ns]) and torch.equal(full_lin[seg.rows[:ns].long()], glin[:ns])
    # the table pass: every row moves (TF1), the touched ones with their gradient
    row_slot = torch.full((V,), -1, dtype=torch.int32, device=dev)
    hp = ops.adam_hp(1e-2, 3)
    ops.adam_dense_rows(w, m, v, hp, grows, seg, row_slot, lin=l, lin_m=lm, lin_v=lv, glin_rows=glin)
    assert int((row_slot != -1).sum()) == 0
    gd = np.zeros((V, K), np.float32)
    gd[rows] = grows[:ns].cpu().numpy()
    gld = np.zeros((V, 1), np.float32)
    gld[rows, 0] = glin[:ns].cpu().numpy()
    ew, em, ev = ops_np.adam_step(w0, m0, v0, gd, 1e-2, 3)
    el, elm, elv = ops_np.adam_step(l0, lm0, lv0, gld, 1e-2, 3)
    np.testing.assert_allclose(w.cpu().numpy(), ew, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m.cpu().numpy(), em, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(v.cpu().numpy(), ev, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(l.cpu().numpy(), el, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lm.cpu().numpy(), elm, rtol=1e-5, atol=1e-7)
    untouched = np.setdiff1d(np.arange(V), rows)
    assert np.abs(w.cpu().numpy()[untouched] - w0[untouched]).max() > 0          # TF1: they move on their moments
    with pytest.raises(ValueError):
        ops.adam_dense_rows(w, m, v, hp, grows, seg, row_slot[:-1], lin=l, lin_m=lm, lin_v=lv, glin_rows=glin)


def test_dense_table_pass_refuses_other_widths(dev):
    V, K = 10, 24
    w = torch.zeros((V, K), device=dev)
    seg = ops.build_segments(torch.zeros(4, dtype=torch.int32, device=dev), V)
    with pytest.raises(ValueError):
        ops.adam_dense_rows(w, w.clone(), w.clone(), ops.adam_hp(1e-3, 1), torch.zeros((4, K), device=dev), seg,
                            torch.full((V,), -1, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("graph", [False, True])
def test_fused_dense_adam_trajectory_vs_fp64_oracle(dev, l1_arith, graph):
    """Five steps of the fused step with TF1's dense Adam (rows touched once keep moving afterwards) against the fp64 oracle
    from the same weights; with `graph` the steps after the second are hipGraph replays."""
    rng = np.random.default_rng(17)
    K, Fs, hidden = 64, 12, (128, 64, 32)
    nu, ni, vocab, B = 50, 70, 11, 96
    net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=hidden, lr=1e-2, device=dev, dense_adam=True,
                    sparse_offsets=np.arange(Fs) * (vocab + 1))
    assert net.fused_l1 and net.hip_tail and net.l1_arith == l1_arith
    W = export_fieldnet_weights(net)
    o64 = DeepFMOracle(W, hidden, lr=1e-2, dtype=torch.float64)
    if graph:
        net.enable_graph(True)
    batches = [make_batch(rng, B, nu, ni, vocab, Fs) for _ in range(5)]
    for b in batches:
        idx, lab = to_dev(net, *b, dev)
        l_hip = float(net.train_step(idx, lab))
        l_64 = float(o64.train_step(*cpu_batch(*b)))
        assert abs(l_hip - l_64) < 2e-5, (l_hip, l_64)
    if graph:
        assert net._graphs, "no step was captured"
    W2 = export_fieldnet_weights(net)
    for name, ref in o64.V.v.items():
        got = W2[name].numpy().reshape(ref.shape)
        # Adam normalises: a weight whose gradient is rounding noise (biases in front of a BatchNorm) may move by
        # +-lr per step in either direction; everything else follows the fp64 trajectory
        noise = name.endswith("/bias") and "_layer" in name
        np.testing.assert_allclose(got, ref.detach().numpy(), rtol=1e-4, atol=6e-2 if noise else 5e-6, err_msg=name)
