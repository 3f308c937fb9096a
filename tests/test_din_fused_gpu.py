"""The fused DIN training step (nets/din_fused.py: hand-written kernels end to end, one hipGraph per batch shape)
against (i) the autograd path of the same net, (ii) the fp64 oracle restatement of the reference graph
(algorithms/din.py:165-250) and (iii) itself under graph replay (`-m gpu`).

Tolerances: losses 1e-5; weights after 3 row-wise-Adam steps rtol 1e-4 / atol 3e-6 between the two HIP paths
(different summation orders); first step against the fp64 oracle as in test_din_lazy_adam_first_step; replays are
bit-identical to eager launches of the same kernels."""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from librecommender_amd.nets import FeatDINNet, FeatSpec
from oracle.models_torch import DINOracle, export_net_weights

from tests.test_din_tower_models_gpu import T, din_batch, din_call, din_oracle_args

pytestmark = pytest.mark.gpu


def make(dev, K, hidden, L, n_sp, vocab=9, use_bn=True, lr=1e-3, **kw):
    nu, ni = 70, 90
    spec = FeatSpec(nu, ni, n_sp, n_sp * (vocab + 1), 0)
    net = FeatDINNet(spec, K, hidden, use_bn=use_bn, max_seq_len=L, lr=lr, device=dev, **kw)
    return net, (nu, ni, L, n_sp, vocab, 0)


def state(net):
    return {**{k: v.clone() for k, v in export_net_weights(net).items()},
            "opt/m": net.tables.m.cpu().clone(), "opt/v": net.tables.v.cpu().clone(),
            "opt/Pm": net.P.m.cpu().clone(), "opt/Pv": net.P.v.cpu().clone()}


@pytest.mark.parametrize("K,hidden,n_sp,use_bn", [(64, (128, 64, 32), 2, True), (128, (128, 32), 0, True),
                                                  (32, (128, 64), 1, False), (64, (64, 32, 16), 0, True)])
def test_fused_step_matches_autograd_path(dev, K, hidden, n_sp, use_bn):
    a, shp = make(dev, K, hidden, 7, n_sp, use_bn=use_bn, graph_step=False)
    b, _ = make(dev, K, hidden, 7, n_sp, use_bn=use_bn, fused_step=False)
    assert a._fstep is not None and b._fstep is None
    rng = np.random.default_rng(5)
    for B in (96, 96, 41):                         # 41: partial MFMA tiles / partial tail workgroups
        bt = din_batch(rng, B, *shp)
        la = float(a.train_step(labels=bt[-1], **din_call(bt)))
        lb = float(b.train_step(labels=bt[-1], **din_call(bt)))
        assert abs(la - lb) < 1e-5, (la, lb)
    sa, sb = state(a), state(b)
    for name in sa:
        # Adam normalises: a weight whose gradient is rounding noise (biases in front of a BatchNorm) may move by
        # +-lr in either direction -> compare with an absolute slack of a few lr for those, tight otherwise
        slack = 3.5e-3 if (name.endswith("/bias") and use_bn) else 3e-6
        np.testing.assert_allclose(sa[name].numpy(), sb[name].numpy(), rtol=2e-4, atol=slack, err_msg=name)


def test_fused_first_step_vs_fp64_oracle(dev):
    net, shp = make(dev, 64, (128, 64, 32), 6, 2, graph_step=False)
    W = export_net_weights(net)
    o = DINOracle(W, (128, 64, 32), True, 6, lr=1e-3, dtype=torch.float64)
    rng = np.random.default_rng(11)
    bt = din_batch(rng, 80, *shp)
    np.testing.assert_allclose(net.forward(**din_call(bt)).cpu().numpy(), o.forward(*din_oracle_args(bt)).detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    l_hip = float(net.train_step(labels=bt[-1], **din_call(bt)))
    l_ref = float(o.train_step(*din_oracle_args(bt), T(bt[-1])))
    assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    users, items, sparse, _, seqs, lens, _ = bt
    valid = np.arange(seqs.shape[1])[None, :] < lens[:, None]
    touched = {"user_embeds_var": np.unique(users), "item_embeds_var": np.unique(np.concatenate([items, seqs[valid]])),
               "sparse_embeds_var": np.unique(sparse)}
    for name, rows in touched.items():
        # gradients through Adam's first moment (linear in g) ...
        lo = {"user_embeds_var": net.tables.user_off, "item_embeds_var": net.tables.item_off,
              "sparse_embeds_var": net.tables.sparse_off}[name]
        m_hip = net.tables.m[lo: lo + W[name].shape[0]].cpu().numpy()[rows]
        m_ref = o.opt.state[id(o.V.v[name])][0].numpy()[rows]
        np.testing.assert_allclose(m_hip, m_ref, rtol=1e-3, atol=1e-4 * np.abs(m_ref).max(), err_msg=name + " (m)")
        # ... and the updated rows; untouched rows frozen
        np.testing.assert_allclose(W2[name].numpy()[rows], o.V.v[name].detach().numpy()[rows], rtol=1e-4, atol=2e-5, err_msg=name)
        rest = np.setdiff1d(np.arange(W[name].shape[0]), rows)
        np.testing.assert_array_equal(W2[name].numpy()[rest], W[name].numpy()[rest])
    for name in ("attention/attention_layer1/kernel", "attention/attention_layer1/bias", "attention/attention_layer2/kernel",
                 "mlp/mlp_layer1/kernel", "mlp/mlp_layer2/kernel", "mlp/mlp_layer3/kernel", "mlp/bn_in/gamma", "mlp/bn_in/beta",
                 "mlp/bn1/gamma", "out/kernel", "out/bias"):
        ref = o.V.v[name]
        p = net.P[name]
        off = (p.data_ptr() - net.P.flat.data_ptr()) // 4
        m_got = net.P.m[off: off + p.numel()].cpu().numpy().reshape(ref.shape)
        m_ref = o.opt.state[id(ref)][0].numpy()
        np.testing.assert_allclose(m_got, m_ref, rtol=1e-3, atol=2e-4 * (np.abs(m_ref).max() + 1e-12), err_msg=name + " (gradient)")
    for k in ("mlp/bn_in/moving_mean", "mlp/bn_in/moving_var", "mlp/bn1/moving_mean", "mlp/bn1/moving_var"):
        np.testing.assert_allclose(W2[k].numpy(), o.V.buffers[k].numpy(), rtol=1e-4, atol=1e-7, err_msg=k)


def test_graph_replays_bit_identical_to_eager_and_alternating_shapes(dev):
    g, shp = make(dev, 64, (128, 64, 32), 7, 2, graph_step=True)
    e, _ = make(dev, 64, (128, 64, 32), 7, 2, graph_step=False)
    rng = np.random.default_rng(2)
    big = [din_batch(rng, 128, *shp) for _ in range(3)]
    small = [din_batch(rng, 37, *shp) for _ in range(2)]
    order = [big[0], big[1], small[0], big[2], big[0], small[1], small[0], big[1], big[2]]   # epochs with a short last batch
    for bt in order:
        lg = g.train_step(labels=bt[-1], **din_call(bt)).clone()
        le = e.train_step(labels=bt[-1], **din_call(bt))
        assert torch.equal(lg, le), (float(lg), float(le))
    assert len(g._fstep.runner.graphs) == 2 and all("graph" in st for st in g._fstep.runner.graphs.values())
    sg, se = state(g), state(e)
    for name in sg:
        assert torch.equal(sg[name], se[name]), name


def test_block_kernels_vs_numpy(dev):
    """lr_table_colstats_f32 / lr_bn_remainder_f32 against their definitions."""
    rng = np.random.default_rng(0)
    P_, B, K = 3, 157, 64
    x = rng.standard_normal((P_ * B, K)).astype(np.float32)
    idxT = np.arange(P_ * B, dtype=np.int32).reshape(P_, B)
    idx = np.ascontiguousarray(idxT.T)
    idx[5, 1] = -1                                                   # dropped id: contributes zeros
    part = ops.table_colstats(torch.from_numpy(x).to(dev), torch.from_numpy(idx).to(dev), 6).cpu().numpy().astype(np.float64)
    for f in range(P_):
        rows = x[idx[:, f][idx[:, f] >= 0]].astype(np.float64)
        np.testing.assert_allclose(part[f, :, 0].sum(0), rows.sum(0), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(part[f, :, 1].sum(0), (rows ** 2).sum(0), rtol=1e-5, atol=1e-4)
    G = rng.standard_normal((P_ * B, K)).astype(np.float32)
    a = rng.standard_normal(P_ * K).astype(np.float32)
    c = rng.standard_normal(P_ * K).astype(np.float32)
    got = ops.bn_remainder_(torch.from_numpy(G.copy()).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev),
                            torch.from_numpy(c).to(dev), B).cpu().numpy()
    ref = G - np.repeat(a.reshape(P_, 1, K), B, 1).reshape(-1, K) - np.repeat(c.reshape(P_, 1, K), B, 1).reshape(-1, K) * x
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("K", [16, 64, 128])
def test_scatter_long_runs_vs_fp64(dev, K):
    """Zipf-head rows (runs of thousands of positions: the chunked whole-workgroup path of csrc/embed_scatter.hip)
    in segment-sum / scatter-add / scatter-Adam against fp64, and run-to-run bit identity."""
    rng = np.random.default_rng(K)
    V, n = 5000, 60_000
    idx = rng.integers(0, V, n).astype(np.int32)
    idx[rng.random(n) < 0.35] = 7                       # ~21,000 positions on one row
    idx[rng.random(n) < 0.02] = 4999                    # ~1,200 on another (2 chunks)
    idx[rng.random(n) < 0.006] = 123                    # ~360: just above the long-run threshold
    idx[:50] = -1                                       # dropped
    g = (rng.standard_normal((n, K)) * 0.1).astype(np.float32)
    table = (rng.standard_normal((V, K)) * 0.1).astype(np.float32)
    b = ops.SegmentBuilder(n, V, dev)
    seg = b.build(torch.from_numpy(idx).to(dev))
    gd = torch.from_numpy(g).to(dev)
    ref = np.zeros((V, K))
    np.add.at(ref, idx[idx >= 0], g[idx >= 0].astype(np.float64))
    rows = seg.rows[: seg.count()].cpu().numpy()
    np.testing.assert_array_equal(rows, np.unique(idx[idx >= 0]))
    gs = ops.embed_segment_sum(gd, seg)[: len(rows)].cpu().numpy()
    np.testing.assert_allclose(gs, ref[rows], rtol=1e-5, atol=2e-5)
    t1 = torch.from_numpy(table).to(dev)
    ops.embed_scatter_add(t1, gd, seg, alpha=0.5)
    np.testing.assert_allclose(t1.cpu().numpy(), table + 0.5 * ref, rtol=1e-5, atol=2e-5)
    outs = []
    for _ in range(2):
        t2, m, v = torch.from_numpy(table).to(dev), torch.zeros((V, K), device=dev), torch.zeros((V, K), device=dev)
        ops.embed_scatter_adam(t2, m, v, gd, seg, ops.adam_hp(1e-3, 1))
        outs.append((t2.clone(), m.clone(), v.clone()))
    assert all(torch.equal(a, c) for a, c in zip(*outs))
    from oracle import ops_np
    w_ref, m_ref, _ = ops_np.adam_step(table[rows].astype(np.float64), np.zeros((len(rows), K)), np.zeros((len(rows), K)),
                                       ref[rows], 1e-3, 1)
    np.testing.assert_allclose(outs[0][1].cpu().numpy()[rows], m_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(outs[0][0].cpu().numpy()[rows], w_ref, rtol=1e-4, atol=2e-6)
    untouched = np.setdiff1d(np.arange(V), rows)
    np.testing.assert_array_equal(outs[0][0].cpu().numpy()[untouched], table[untouched])


@pytest.mark.parametrize("n_sp,cols", [(0, None), (3, None), (4, [2, 0])])
def test_din_build_ids_matches_the_elementwise_formula(dev, n_sp, cols):
    """`lr_din_build_ids_i32` (one launch) against the torch expressions it replaced in the captured step."""
    from librecommender_amd import ops

    rng = np.random.default_rng(3 + n_sp)
    B, L, u_off, i_off, s_off = 37, 6, 5, 1000, 5000
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)  # noqa: E731
    users, items = t32(rng.integers(0, 900, B)), t32(rng.integers(0, 3000, B))
    seqs, lens = t32(rng.integers(0, 3000, (B, L))), t32(rng.integers(0, L + 1, B))
    sparse = t32(rng.integers(0, 50, (B, n_sp))) if n_sp else None
    cols_t = None if cols is None else t32(np.asarray(cols))
    n_plain = 0 if sparse is None else (len(cols) if cols is not None else n_sp)
    out = torch.empty((2 + n_plain + 2 + L) * B, dtype=torch.int32, device=dev)
    ops.din_build_ids(users, items, sparse, cols_t, seqs, lens, u_off, i_off, s_off, out)
    parts = [users + u_off, items + i_off]
    if sparse is not None:
        sp = sparse if cols is None else sparse[:, torch.as_tensor(cols, device=dev)]
        parts.append((sp.t() + s_off).reshape(-1))
    parts.append(torch.full((B,), -1, dtype=torch.int32, device=dev))
    parts.append(items + i_off)
    valid = torch.arange(L, device=dev)[None, :] < lens[:, None]
    parts.append(torch.where(valid, seqs + i_off, torch.full_like(seqs, -1)).reshape(-1))
    assert torch.equal(out, torch.cat(parts).to(torch.int32))
