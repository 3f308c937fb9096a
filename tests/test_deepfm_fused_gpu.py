"""DeepFM lookup fused with the first Dense layer (csrc/deepfm_l1.hip), the run-ordered Adam kernel
and the per-field LDS segment sort — each C-ABI entry against its fp64 / integer restatement in
`oracle/ops_np.py`, then the whole training step against the unfused HIP path and the oracle graph.

Tolerances: index work bit-exact; f32 MFMA products are exact-f32 fma chains over F*K (<= 13k) terms of
|x| <= 1 values: 1e-5 relative to the accumulated magnitude (vs the fp64 restatement)."""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from librecommender_amd.nets import DeepFMNet
from oracle import ops_np
from oracle.models_torch import DeepFMOracle, export_fieldnet_weights

pytestmark = pytest.mark.gpu

SHAPES = [(64, 128), (32, 128), (128, 128), (64, 64), (32, 64), (64, 256)]


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def field_layout(rng, F, lo=3, hi=40, big=None):
    sizes = rng.integers(lo, hi, F)
    if big is not None:
        sizes[0] = big
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def field_ids(rng, frs, B, zipf=False):
    F = len(frs) - 1
    cols = []
    for f in range(F):
        n = int(frs[f + 1] - frs[f])
        loc = (rng.zipf(1.2, B) - 1) % n if zipf else rng.integers(0, n, B)
        cols.append(loc + frs[f])
    return np.stack(cols, axis=1).astype(np.int32)


def test_idx_transpose(dev):
    rng = np.random.default_rng(0)
    for B, F in ((1, 1), (37, 5), (64, 202), (1000, 33)):
        idx = rng.integers(-5, 1000, (B, F)).astype(np.int32)
        np.testing.assert_array_equal(ops.idx_transpose(t(idx, dev)).cpu().numpy(), idx.T)


@pytest.mark.parametrize("B,F,big,zipf", [(1, 1, None, False), (50, 3, None, False), (64, 7, None, True),
                                          (1000, 12, 300_000, True), (4096, 5, None, True),
                                          (16384, 6, 1_000_001, True), (16383, 3, 70_000, False)])
def test_segments_fields_bit_exact(dev, B, F, big, zipf):
    rng = np.random.default_rng(B + F)
    frs = field_layout(rng, F, big=big)
    idx = field_ids(rng, frs, B, zipf)
    if B >= 50:      # dropped entries: negative, beyond the table, and inside ANOTHER field's range
        idx[3, 0] = -1
        idx[7, F - 1] = int(frs[-1]) + 5
        if F > 1:
            idx[11, 1] = int(frs[0])
    V = int(frs[-1])
    sb = ops.FieldSegmentBuilder(B, F, V, dev)
    frs_d = t(frs.astype(np.int32), dev)
    for _ in range(2):      # buffers are reused between steps
        seg = sb.build(ops.idx_transpose(t(idx, dev)), frs_d)
        ns = seg.count()
        o_pos, o_rows, o_start, o_slot = ops_np.segments_fields(idx, frs)
        assert ns == len(o_rows)
        np.testing.assert_array_equal(seg.rows[:ns].cpu().numpy(), o_rows)
        np.testing.assert_array_equal(seg.start[:ns + 1].cpu().numpy(), o_start)
        np.testing.assert_array_equal(seg.pos[:len(o_pos)].cpu().numpy(), o_pos)
        np.testing.assert_array_equal(seg.slotT.cpu().numpy(), o_slot)
    if B < 50:              # nothing dropped: identical to the general device-wide build
        ref = ops.build_segments(t(idx.reshape(-1), dev), V)
        assert ref.count() == ns
        assert torch.equal(ref.pos[:B * F], seg.pos[:B * F]) and torch.equal(ref.rows[:ns], seg.rows[:ns])
        assert torch.equal(ref.start[:ns + 1], seg.start[:ns + 1])


def unpack_A(WpA, F, K, H1):
    """Inverse of the documented WpA layout (include/libreco_hip.h / csrc/deepfm_l1.hip)."""
    a = WpA.reshape(F, H1 // 32, K // 8, 2, 32, 4)            # f, ct, s4, h, j, c
    out = np.zeros((F * K, H1), np.float32)
    for f in range(F):
        for ct in range(H1 // 32):
            blk = a[f, ct]                                     # [s4, h, j, c]
            rows = (np.arange(2)[None, :, None] * (K // 2) + np.arange(K // 8)[:, None, None] * 4 + np.arange(4)[None, None, :])
            for s4 in range(K // 8):
                for h in range(2):
                    for c in range(4):
                        out[f * K + rows[s4, h, c], ct * 32:(ct + 1) * 32] = blk[s4, h, :, c]
    return out


def unpack_B(WpB, F, K, H1):
    b = WpB.reshape(F, K // 32, H1 // 8, 2, 32, 4)            # f, ni, s4, h, j, c
    out = np.zeros((F * K, H1), np.float32)
    for f in range(F):
        for ni in range(K // 32):
            for s4 in range(H1 // 8):
                for h in range(2):
                    cols = h * (H1 // 2) + s4 * 4
                    out[f * K + ni * 32: f * K + (ni + 1) * 32, cols:cols + 4] = b[f, ni, s4, h]
    return out


@pytest.mark.parametrize("K,H1", SHAPES)
def test_l1_pack_layouts(dev, f32_chain, K, H1):
    assert ops.deepfm_l1_supported(K, H1) and not ops.deepfm_l1_supported(48, 128)
    rng = np.random.default_rng(K + H1)
    F = 3
    Wp = rng.standard_normal((F * K, H1)).astype(np.float32)
    WpA, WpB = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    np.testing.assert_array_equal(unpack_A(WpA.cpu().numpy(), F, K, H1), Wp)
    np.testing.assert_array_equal(unpack_B(WpB.cpu().numpy(), F, K, H1), Wp)


def make_case(rng, B, F, K, H1, dev, zipf=True, bad=True):
    frs = field_layout(rng, F, 5, 60)
    V = int(frs[-1])
    idx = field_ids(rng, frs, B, zipf)
    if bad and B > 10:
        idx[2, 0] = -1
        idx[5, F - 1] = V + 3
    table = (rng.standard_normal((V, K)) * 0.5).astype(np.float32)
    lin = rng.standard_normal(V).astype(np.float32)
    Wp = (rng.standard_normal((F * K, H1)) / np.sqrt(F * K)).astype(np.float32)
    bias = rng.standard_normal(H1).astype(np.float32)
    return frs, V, idx, table, lin, Wp, bias


@pytest.mark.parametrize("K,H1", SHAPES)
@pytest.mark.parametrize("B,F", [(64, 1), (100, 2), (257, 7), (1000, 23)])
def test_l1_fwd_matches_fp64(dev, f32_chain, K, H1, B, F):
    rng = np.random.default_rng(B * 7 + F + K + H1)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    WpA, _ = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    z1, pair, fsum, lin_out = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), WpA, t(bias, dev), H1, lin=t(lin, dev))
    o_z1, o_pair, o_fsum, o_lin = ops_np.deepfm_l1_fwd(table, lin, idx, Wp, bias)
    scale = float(np.abs(o_z1).max()) + 1.0
    np.testing.assert_allclose(z1.cpu().numpy(), o_z1, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(fsum.cpu().numpy(), o_fsum, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pair.cpu().numpy(), o_pair, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(lin_out.cpu().numpy(), o_lin.astype(np.float32))
    # same FM outputs as the unfused gather kernel, bit for bit on fsum's inputs (exact rows)
    e, pair2, fsum2, lin2 = ops.fm_embed_fwd(t(table, dev), t(idx, dev), lin=t(lin, dev))
    assert torch.equal(lin2, lin_out)
    torch.testing.assert_close(fsum2, fsum, rtol=1e-5, atol=1e-5)
    # without the linear table
    z1b, *_ = ops.deepfm_l1_fwd(t(table, dev), t(idx, dev), WpA, None, H1)
    np.testing.assert_allclose(z1b.cpu().numpy(), o_z1 - bias, rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("K,H1", SHAPES)
@pytest.mark.parametrize("B,F,nch", [(64, 2, 1), (300, 3, 2), (1000, 5, 5), (1000, 5, None)])
def test_l1_wgrad_matches_fp64(dev, f32_chain, K, H1, B, F, nch):
    if K == 16:
        pytest.skip("wgrad tiles need K % 32 == 0")
    rng = np.random.default_rng(B + F + K + H1)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    gz = rng.standard_normal((B, H1)).astype(np.float32)
    idxT = ops.idx_transpose(t(idx, dev))
    part = ops.deepfm_l1_wgrad(t(table, dev), idxT, t(gz, dev), n_chunks=nch)
    got = part.double().sum(0).cpu().numpy()
    want = ops_np.deepfm_l1_wgrad(table, idx, gz)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * (float(np.abs(want).max()) + 1.0))


@pytest.mark.parametrize("K,H1", SHAPES)
@pytest.mark.parametrize("B,F", [(64, 1), (130, 3), (1000, 9)])
def test_l1_dgrad_matches_fp64(dev, f32_chain, K, H1, B, F):
    rng = np.random.default_rng(B + 3 * F + K + H1)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    gz = rng.standard_normal((B, H1)).astype(np.float32)
    gl = rng.standard_normal(B).astype(np.float32)
    wp = rng.standard_normal(K).astype(np.float32)
    fsum = rng.standard_normal((B, K)).astype(np.float32)
    _, _, _, slotT = ops_np.segments_fields(idx, frs)
    _, WpB = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    out = torch.zeros((B * F + 1, K), device=dev)
    ge = ops.deepfm_l1_dgrad(t(gz, dev), WpB, K, F, t(slotT, dev), gl=t(gl, dev), wp=t(wp, dev), fsum=t(fsum, dev), out=out)[:B * F]
    want = ops_np.deepfm_l1_dgrad(gz, Wp, K, gl, wp, fsum, slotT)
    np.testing.assert_allclose(ge.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * (float(np.abs(want).max()) + 1.0))
    out.zero_()
    ge2 = ops.deepfm_l1_dgrad(t(gz, dev), WpB, K, F, t(slotT, dev), out=out)[:B * F]     # no FM term
    want2 = ops_np.deepfm_l1_dgrad(gz, Wp, K, None, None, None, slotT)
    np.testing.assert_allclose(ge2.cpu().numpy(), want2, rtol=1e-5, atol=1e-5 * (float(np.abs(want2).max()) + 1.0))


@pytest.mark.parametrize("K", [16, 32, 64, 128])
@pytest.mark.parametrize("with_bn", [True, False])
def test_fm_rows_adam_matches_oracle(dev, K, with_bn):
    rng = np.random.default_rng(K)
    B, F = 700, 4
    frs = np.array([0, 3, 40, 41, 300])              # a 3-row field (runs of > 200 positions), a 1-row field
    V = int(frs[-1])
    idx = field_ids(rng, frs, B)
    idx[9, 2] = -1
    pos, rows, start, slotT = ops_np.segments_fields(idx, frs)
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
    l0, lm0, lv0 = rng.standard_normal((V, 1)).astype(np.float32), np.zeros((V, 1), np.float32), np.zeros((V, 1), np.float32)
    sb = ops.FieldSegmentBuilder(B, F, V, dev)
    seg = sb.build(ops.idx_transpose(t(idx, dev)), t(frs.astype(np.int32), dev))
    w, m, v, l, lm, lv = (t(x.copy(), dev) for x in (w0, m0, v0, l0, lm0, lv0))
    hp = ops.adam_hp(1e-2, 3)
    ops.fm_rows_adam(w, m, v, t(ge, dev), seg, hp, B, F, gl=t(gl, dev), wp=t(wp, dev), lin=l, lin_m=lm, lin_v=lv,
                     bn_a=None if a is None else t(a, dev), bn_c=None if c is None else t(c, dev), lin_scale=t(lin_scale, dev))
    g, sgl = ops_np.fm_rows_gradient(w0, ge, pos, rows, start, F, gl, wp, a, c)
    assert (np.diff(start) > 32).sum() >= 3          # the whole-workgroup path ran
    ew, em, ev = w0.copy(), m0.copy(), v0.copy()
    ew[rows], em[rows], ev[rows] = ops_np.adam_step(w0[rows], m0[rows], v0[rows], g.astype(np.float32), 1e-2, 3)
    np.testing.assert_allclose(w.cpu().numpy(), ew, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(m.cpu().numpy(), em, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v.cpu().numpy(), ev, rtol=1e-4, atol=1e-6)
    fld = pos[start[:-1]] % F
    gl_rows = (sgl * lin_scale[fld]).astype(np.float32)[:, None]
    el, elm, elv = l0.copy(), lm0.copy(), lv0.copy()
    el[rows], elm[rows], elv[rows] = ops_np.adam_step(l0[rows], lm0[rows], lv0[rows], gl_rows, 1e-2, 3)
    np.testing.assert_allclose(l.cpu().numpy(), el, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(lm.cpu().numpy(), elm, rtol=1e-4, atol=1e-6)
    untouched = np.setdiff1d(np.arange(V), rows)
    np.testing.assert_array_equal(w.cpu().numpy()[untouched], w0[untouched])


def batch(rng, B, nu, ni, vocab, Fs, dev, net):
    users, items = rng.integers(0, nu, B), rng.integers(0, ni, B)
    sparse = (rng.zipf(1.3, (B, Fs)) - 1) % vocab + np.arange(Fs) * (vocab + 1)
    labels = rng.integers(0, 2, B).astype(np.float32)
    idx = net.tables.global_idx(t(users, dev), t(items, dev), t(sparse, dev))
    return (users, items, sparse, labels), idx, t(labels, dev)


@pytest.mark.parametrize("K,hidden,use_bn", [(64, (128, 64, 32), True), (32, (64, 16), True), (64, (128,), False),
                                             (128, (128, 32), True)])
def test_fused_step_equals_unfused_step_and_oracle(dev, l1_arith, K, hidden, use_bn):
    """Same seed -> same initial state; three steps of the fused path vs (a) the unfused HIP path
    (materialised deep_embed, hipBLASLt GEMMs, lr_fm_embed_bwd_adam_f32) and (b) the first step of the
    reference-graph oracle (TF1 dense Adam == row-wise Adam at step 1)."""
    nu, ni, vocab, Fs, B = 300, 200, 37, 9, 777
    kw = dict(embed_size=K, hidden_units=hidden, use_bn=use_bn, lr=1e-2, device=dev,
              sparse_offsets=np.arange(Fs) * (vocab + 1))
    fused = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    plain = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, fused_l1=False, **kw)
    assert fused.fused_l1 and not plain.fused_l1
    assert torch.equal(fused.tables.embed, plain.tables.embed) and torch.equal(fused.P.flat, plain.P.flat)
    W0 = export_fieldnet_weights(fused)
    oracle = DeepFMOracle(W0, hidden, use_bn=use_bn, lr=1e-2, dtype=torch.float64)
    rng = np.random.default_rng(K)
    for step in range(3):
        host, idx, lab = batch(rng, B, nu, ni, vocab, Fs, dev, fused)
        lf, lp = float(fused.train_step(idx, lab)), float(plain.train_step(idx, lab))
        assert abs(lf - lp) < 2e-5
        if step == 0:
            cpu = tuple(torch.from_numpy(np.asarray(x)).long() for x in host[:3]) + (torch.from_numpy(host[3]),)
            lo = float(oracle.train_step(*cpu))
            assert abs(lf - lo) < 1e-5
            W1 = export_fieldnet_weights(fused)
            for name, ref in oracle.V.v.items():
                got = W1[name].numpy().reshape(ref.shape)
                np.testing.assert_allclose(got, ref.detach().numpy(), rtol=1e-4, atol=5e-5, err_msg=name)
        if step == 0:
            # first moments after step 1 are (1 - beta1) * gradient: a LINEAR image of the row gradients (the
            # weights themselves saturate at +-lr once |g| >> eps and would hide gradient errors)
            def rows_agree(a_, b_, budget):
                """All rows but those of at most `budget` samples agree (a ReLU pre-activation within rounding of
                zero may take the other branch: that sample's rows then differ by percents)."""
                a_, b_ = a_.double().reshape(a_.shape[0], -1), b_.double().reshape(b_.shape[0], -1)
                scale = float(b_.pow(2).mean().sqrt())
                bad = ((a_ - b_).abs() > 1e-3 * b_.abs() + 1e-3 * scale).any(dim=1)
                assert int(bad.sum()) <= budget, (int(bad.sum()), budget)

            rows_agree(fused.tables.m, plain.tables.m, 2 * (Fs + 2))
            rows_agree(fused.tables.lin_m, plain.tables.lin_m, 2 * (Fs + 2))
            torch.testing.assert_close(fused.P.m, plain.P.m, rtol=5e-3, atol=5e-3 * float(plain.P.m.abs().max()))
            st = oracle.opt.state
            for kind in ("user", "item", "sparse"):
                om = torch.from_numpy(st[id(oracle.V.v[f"{kind}_embeds_var"])][0].numpy())
                got = fused.tables.variable(f"{kind}_embeds_var")
                lo_ = got.storage_offset() // K
                # TF forms (1 - beta1) in the variable dtype: the fp64 oracle's moments carry 0.1, the fp32 path 0.100000024
                rows_agree(fused.tables.m[lo_: lo_ + got.shape[0]].cpu(), om * (float(np.float32(1) - np.float32(0.9)) / 0.1), 2 * (Fs + 2))
        torch.testing.assert_close(fused.tables.embed, plain.tables.embed, rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(fused.tables.lin, plain.tables.lin, rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(fused.P.flat, plain.P.flat, rtol=1e-4, atol=5e-5)
    if use_bn:
        torch.testing.assert_close(fused.mlp.bn_in.moving_mean, plain.mlp.bn_in.moving_mean, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(fused.mlp.bn_in.moving_var, plain.mlp.bn_in.moving_var, rtol=1e-4, atol=1e-6)
    host, idx, lab = batch(rng, 333, nu, ni, vocab, Fs, dev, fused)
    torch.testing.assert_close(fused.forward(idx), plain.forward(idx), rtol=1e-4, atol=1e-4)
    # run-to-run bit identity of the fused step (no floating-point atomics anywhere)
    a = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    b = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    a.train_step(idx, lab)
    b.train_step(idx, lab)
    assert torch.equal(a.tables.embed, b.tables.embed) and torch.equal(a.P.flat, b.P.flat)


def test_graph_replayed_steps_equal_eager_steps(dev):
    """One hipGraph replay per step (Adam coefficients read from device memory) == eager launches, bit
    for bit, over steps that span the eager warm-up, the capture and several replays."""
    nu, ni, vocab, Fs, B, K = 300, 200, 37, 6, 512, 64
    kw = dict(embed_size=K, hidden_units=(128, 32), lr=1e-2, device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    eager = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    graph = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    graph.enable_graph(True, warm_steps=2)
    rng = np.random.default_rng(3)
    for step in range(7):
        _, idx, lab = batch(rng, B, nu, ni, vocab, Fs, dev, eager)
        if step == 5:
            eager.lr = graph.lr = 5e-3              # lr schedule between replays
        le, lg = float(eager.train_step(idx, lab)), float(graph.train_step(idx, lab))
        assert le == lg, step
    assert "graph" in graph._graphs[((B, Fs + 2), "cross_entropy")]
    assert torch.equal(eager.tables.embed, graph.tables.embed) and torch.equal(eager.tables.m, graph.tables.m)
    assert torch.equal(eager.tables.lin, graph.tables.lin) and torch.equal(eager.P.flat, graph.P.flat)
    assert torch.equal(eager.mlp.bn_in.moving_var, graph.mlp.bn_in.moving_var)


def test_graph_replays_with_alternating_batch_shapes(dev):
    """The last batch of an epoch is shorter: graphs of two batch shapes are captured and replayed in turn, with eager
    steps of the other shape in between.  Every shape keeps its own tail buffers (activations, partial sums, the
    device-resident reduction job table) — replays must not see buffers re-used by the other shape.  Bit-identical to
    eager launches throughout (regression: the tail used to hold ONE buffer set, so the first short batch freed what
    the full-batch graph still addressed)."""
    nu, ni, vocab, Fs, K = 300, 200, 37, 6, 64
    kw = dict(embed_size=K, hidden_units=(128, 64, 32), lr=1e-2, device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    eager = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    graph = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    graph.enable_graph(True, warm_steps=2)
    rng = np.random.default_rng(5)
    junk = []
    for step, B in enumerate([512, 512, 512, 200, 512, 512, 200, 512, 200, 200, 512, 200, 512, 136, 512]):
        _, idx, lab = batch(rng, B, nu, ni, vocab, Fs, dev, eager)
        le, lg = float(eager.train_step(idx, lab)), float(graph.train_step(idx, lab))
        assert le == lg, (step, B)
        junk.append(torch.full((1 << 18,), float(step), device=dev))      # allocator churn: freed blocks get re-used
        if len(junk) > 3:
            junk.pop(0)
    assert "graph" in graph._graphs[((512, Fs + 2), "cross_entropy")] and "graph" in graph._graphs[((200, Fs + 2), "cross_entropy")]
    assert torch.equal(eager.tables.embed, graph.tables.embed) and torch.equal(eager.tables.m, graph.tables.m)
    assert torch.equal(eager.tables.lin, graph.tables.lin) and torch.equal(eager.P.flat, graph.P.flat)
    # a LARGER batch re-allocates the shared segment / gradient workspaces: the captured graphs are dropped, not replayed
    _, idx, lab = batch(rng, 640, nu, ni, vocab, Fs, dev, eager)
    assert float(eager.train_step(idx, lab)) == float(graph.train_step(idx, lab))
    assert ((512, Fs + 2), "cross_entropy") not in graph._graphs or "graph" not in graph._graphs[((512, Fs + 2), "cross_entropy")]
    _, idx, lab = batch(rng, 512, nu, ni, vocab, Fs, dev, eager)
    assert float(eager.train_step(idx, lab)) == float(graph.train_step(idx, lab))
    assert torch.equal(eager.tables.embed, graph.tables.embed) and torch.equal(eager.P.flat, graph.P.flat)


@pytest.mark.parametrize("hidden,use_bn,B", [((128, 64, 32), True, 1000), ((64, 32), True, 777), ((128, 64, 32), False, 640),
                                             ((128,), True, 300), ((64, 64, 32), True, 4100), ((32, 16), True, 64)])
def test_hip_tail_matches_torch_autograd(dev, hidden, use_bn, B):
    """csrc/deepfm_tail.hip (layers after the first Dense, output layer, BCE loss, backward) against torch
    autograd over the same parameters: loss, d loss / d logit, d loss / d z1 and every parameter gradient."""
    import torch.nn.functional as Fn

    from librecommender_amd.layers.tail import DeepFMTail

    Fs, K = 7, 64
    net = DeepFMNet(50, 60, Fs * 10, Fs, embed_size=K, hidden_units=hidden, use_bn=use_bn, device=dev,
                    sparse_offsets=np.arange(Fs) * 10)
    assert DeepFMTail.supported(net.mlp) and net.hip_tail == net.fused_l1
    wide = DeepFMNet(50, 60, Fs * 10, Fs, embed_size=K, hidden_units=(128, 256, 128), device=dev, sparse_offsets=np.arange(Fs) * 10)
    assert not wide.hip_tail and wide.fused_l1          # 256 x 128 tiles exceed the LDS: torch tail behind the fused first layer
    g = torch.Generator(device=dev).manual_seed(B)
    with torch.no_grad():                       # non-trivial BatchNorm parameters / output weights
        for name, p in net.P.params.items():
            if name.endswith("gamma"):
                p.add_(torch.randn(p.shape, device=dev, generator=g) * 0.2)
            elif name.endswith("beta") or name.endswith("bias"):
                p.add_(torch.randn(p.shape, device=dev, generator=g) * 0.1)
    z1 = (torch.randn((B, hidden[0]), device=dev, generator=g) * 0.7).requires_grad_(True)
    pair = torch.randn((B, K), device=dev, generator=g)
    lin_out = torch.randn((B, Fs + 2), device=dev, generator=g)
    labels = (torch.rand(B, device=dev, generator=g) > 0.5).float()
    mm0 = [None if bn is None else (bn.moving_mean.clone(), bn.moving_var.clone()) for bn in net.mlp.bns]
    net.P.zero_grad()
    deep = net.mlp.tail(z1, True)
    logits = net.out(torch.cat([net.linear(lin_out), pair, deep], dim=1)).squeeze(1)
    logits.retain_grad()
    loss = Fn.binary_cross_entropy_with_logits(logits, labels)
    loss.backward()
    ref = dict(loss=float(loss), gl=logits.grad.clone(), gz1=z1.grad.clone(), grad=net.P.grad.clone(),
               mm=[None if bn is None else (bn.moving_mean.clone(), bn.moving_var.clone()) for bn in net.mlp.bns])
    for bn, st in zip(net.mlp.bns, mm0):       # rewind the moving averages
        if bn is not None:
            bn.moving_mean.copy_(st[0])
            bn.moving_var.copy_(st[1])
    net.P.zero_grad()
    tail = DeepFMTail(net.P, net.mlp, net.linear, net.out, Fs + 2, K, dev)
    loss2, gl, gz1, sgz1 = tail.run(z1.detach().contiguous(), pair, lin_out, labels)
    assert abs(float(loss2) - ref["loss"]) < 2e-6
    torch.testing.assert_close(gl, ref["gl"], rtol=1e-5, atol=1e-9)
    scale = float(ref["gz1"].abs().max())
    torch.testing.assert_close(gz1, ref["gz1"], rtol=1e-4, atol=2e-6 * scale)
    torch.testing.assert_close(sgz1, ref["gz1"].sum(0), rtol=1e-4, atol=1e-5 * scale * B ** 0.5)
    for name, p in net.P.params.items():
        if name.startswith("mlp/bn_in") or name.startswith("mlp/mlp_layer1/"):
            continue                            # first layer: not the tail's
        gs = float(ref["grad"].abs().max())
        off = p.storage_offset()
        torch.testing.assert_close(p.grad.reshape(-1), ref["grad"][off:off + p.numel()], rtol=1e-4,
                                   atol=1e-6 * max(gs, 1e-3), msg=lambda m, n=name: f"{n}: {m}")
    for bn, st in zip(net.mlp.bns, ref["mm"]):
        if bn is not None:
            torch.testing.assert_close(bn.moving_mean, st[0], rtol=1e-5, atol=1e-7)
            torch.testing.assert_close(bn.moving_var, st[1], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("hidden,use_bn", [((128, 64, 32), True), ((128,), False)])
def test_hip_tail_step_equals_torch_tail_step(dev, l1_arith, hidden, use_bn):
    nu, ni, vocab, Fs, B, K = 300, 200, 37, 9, 700, 64
    kw = dict(embed_size=K, hidden_units=hidden, use_bn=use_bn, lr=1e-2, device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    a = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    b = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, hip_tail=False, **kw)
    assert a.hip_tail and not b.hip_tail and b.fused_l1
    rng = np.random.default_rng(11)
    for step in range(3):
        _, idx, lab = batch(rng, B, nu, ni, vocab, Fs, dev, a)
        la, lb = float(a.train_step(idx, lab)), float(b.train_step(idx, lab))
        assert abs(la - lb) < 2e-6
        torch.testing.assert_close(a.P.flat, b.P.flat, rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(a.tables.embed, b.tables.embed, rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(a.mlp.bn_in.moving_var, b.mlp.bn_in.moving_var, rtol=1e-5, atol=1e-7) if use_bn else None


@pytest.mark.parametrize("hidden,use_bn,Fs", [((128, 64, 32), True, 8), ((64, 32), True, 4), ((128, 64), False, 10)])
def test_block_first_layer_k16_equals_unfused_step_and_oracle(dev, hidden, use_bn, Fs):
    """The reference's default embed_size = 16 (algorithms/deepfm.py:90-111): the fused lookup + first-layer kernels
    are not compiled for K = 16, so deep_embed is materialised and re-cut into 32-wide blocks for the same MFMA
    first-layer / fold / tail kernels (`DeepFMNet.block_l1`, layers/dense.py:BlockFirstLayer row-major).  Three steps
    against the unfused HIP path (autograd + library GEMMs) and the first step against the reference-graph oracle."""
    nu, ni, vocab, B, K = 300, 200, 37, 777, 16
    kw = dict(embed_size=K, hidden_units=hidden, use_bn=use_bn, lr=1e-2, device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    blk = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    plain = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, fused_l1=False, **kw)
    assert blk.block_l1 and not blk.fused_l1 and not plain.block_l1
    oracle = DeepFMOracle(export_fieldnet_weights(blk), hidden, use_bn=use_bn, lr=1e-2, dtype=torch.float64)
    rng = np.random.default_rng(16)
    for step in range(3):
        host, idx, lab = batch(rng, B if step < 2 else 333, nu, ni, vocab, Fs, dev, blk)
        lf, lp = float(blk.train_step(idx, lab)), float(plain.train_step(idx, lab))
        assert abs(lf - lp) < 2e-5
        if step == 0:
            cpu = tuple(torch.from_numpy(np.asarray(x)).long() for x in host[:3]) + (torch.from_numpy(host[3]),)
            assert abs(lf - float(oracle.train_step(*cpu))) < 1e-5
            W1 = export_fieldnet_weights(blk)
            for name, ref in oracle.V.v.items():
                np.testing.assert_allclose(W1[name].numpy().reshape(ref.shape), ref.detach().numpy(), rtol=1e-4, atol=5e-5,
                                           err_msg=name)
            torch.testing.assert_close(blk.tables.m, plain.tables.m, rtol=2e-3, atol=2e-3 * float(plain.tables.m.abs().max()))
        torch.testing.assert_close(blk.tables.embed, plain.tables.embed, rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(blk.tables.lin, plain.tables.lin, rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(blk.P.flat, plain.P.flat, rtol=1e-4, atol=5e-5)
    if use_bn:
        torch.testing.assert_close(blk.mlp.bn_in.moving_mean, plain.mlp.bn_in.moving_mean, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(blk.mlp.bn_in.moving_var, plain.mlp.bn_in.moving_var, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("arith", ["split_bf16", "f32_chain"])
def test_merged_fold_chain_equals_the_four_launch_chain(dev, monkeypatch, arith):
    """Round 6: the BatchNorm-fold algebra in front of the first layer as two launches (`lr_deepfm_l1_fold_stats_bias_f32`:
    statistics + bias partials per 64-row slab; the weight pack with the partials' reduction as extra workgroups) against the
    four-launch chain (fold_stats, pack, fold_bias, reduce_partials; layers/dense.py:30-41 of the reference): the same
    arithmetic per row / slab / column, so one training step gives the same bits everywhere — loss, every table, every
    parameter, the moving averages."""
    from librecommender_amd import ops as _ops

    monkeypatch.setattr(_ops, "L1_ARITH", arith, raising=False)
    rng = np.random.default_rng(11)
    nu, ni, vocab, Fs, Bs = 300, 200, 37, 9, 777
    users, items = rng.integers(0, nu + 1, Bs), rng.integers(0, ni + 1, Bs)
    sparse = rng.integers(0, vocab + 1, (Bs, Fs)) + np.arange(Fs) * (vocab + 1)
    labels = rng.integers(0, 2, Bs).astype(np.float32)

    def one(mode):
        monkeypatch.setenv("LIBRECO_FOLD_CHAIN", mode)
        monkeypatch.setenv("LIBRECO_L1_ARITH", arith)
        net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=64, hidden_units=(128, 64, 32), lr=1e-2, device=dev,
                        sparse_offsets=np.arange(Fs) * (vocab + 1), seed=5)
        gidx = net.tables.global_idx(torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev),
                                     torch.from_numpy(sparse).to(dev)).contiguous()
        losses = [float(net.train_step(gidx, torch.from_numpy(labels).to(dev))) for _ in range(2)]
        bn = net.mlp.bn_in
        return (losses, net.tables.embed.clone(), net.tables.lin.clone(), net.P.flat.clone(), bn.moving_mean.clone(),
                bn.moving_var.clone(), getattr(net, "l1_arith", None))

    a, b = one("merged"), one("chain")
    assert a[6] == b[6]
    assert a[0] == b[0]
    for x, y in zip(a[1:6], b[1:6]):
        assert torch.equal(x, y)
