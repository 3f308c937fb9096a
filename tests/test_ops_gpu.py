"""Parity of every C-ABI kernel against the CPU oracle on the same seeded inputs (`-m gpu`).

Bars: bit-exact for index work (gather output, segments, top-k ids when scores are separated);
fp32 tolerances stated per test for reductions whose summation order differs from numpy's.
"""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from oracle import ops_np

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6  # fp32 forward reductions


def t(x, dev, dtype=None):
    y = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        y = y.to(dtype)
    return y.to(dev).contiguous()


def zipf_ids(rng, V, size, a=1.05):
    r = rng.zipf(a, size=size).astype(np.int64)
    return ((r - 1) % V).astype(np.int32)


# ---------------------------------------------------------------------------------------
# gather / pooling / dot
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [1, 3, 16, 32, 64, 128, 256, 20])
@pytest.mark.parametrize("n", [0, 1, 77, 5000])
def test_embed_gather_bit_exact(dev, K, n):
    rng = np.random.default_rng(42 + K + n)
    V = 997
    table = rng.standard_normal((V, K)).astype(np.float32)
    idx = rng.integers(0, V, size=n).astype(np.int32)
    if n > 10:
        idx[3], idx[7] = -1, V + 5  # out-of-range -> zero rows (TF-GPU semantics)
    out = ops.embed_gather(t(table, dev), t(idx, dev)).cpu().numpy()
    np.testing.assert_array_equal(out, ops_np.embedding_lookup(table, idx))


def test_embed_gather_2d_indices_and_large(dev):
    rng = np.random.default_rng(1)
    V, K, B, F = 200_000, 64, 4096, 39
    table = rng.standard_normal((V, K)).astype(np.float32)
    idx = zipf_ids(rng, V, (B, F))
    out = ops.embed_gather(t(table, dev), t(idx, dev))
    assert out.shape == (B, F, K)
    np.testing.assert_array_equal(out.cpu().numpy(), table[idx])


def test_embed_gather_rejects_cpu_tensors():
    table = torch.zeros(4, 4)
    idx = torch.zeros(2, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.embed_gather(table, idx)


@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("K", [1, 16, 64, 24])
def test_bag_pool_fwd_bwd(dev, combiner, K):
    rng = np.random.default_rng(7)
    V, nb, L = 301, 513, 3
    oov = V - 1
    table = rng.standard_normal((V, K)).astype(np.float32)
    idx = rng.integers(0, V, size=(nb, L)).astype(np.int32)
    idx[rng.random((nb, L)) < 0.3] = oov
    idx[5] = oov  # an all-OOV bag -> zeros (div_no_nan)
    out = ops.embed_bag_pool(t(table, dev), t(idx, dev), combiner, oov).cpu().numpy()
    np.testing.assert_allclose(out, ops_np.bag_pool(table, idx, combiner, oov), rtol=RTOL, atol=ATOL)
    assert np.all(out[5] == 0)
    gout = rng.standard_normal((nb, K)).astype(np.float32)
    ge = ops.embed_bag_pool_bwd(t(gout, dev), t(idx, dev), V, combiner, oov).cpu().numpy()
    np.testing.assert_allclose(ge, ops_np.bag_pool_bwd(gout, idx, V, combiner, oov), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("D", [16, 33, 128])
def test_pair_dot(dev, D):
    rng = np.random.default_rng(3)
    U = rng.standard_normal((50, D)).astype(np.float32)
    I = rng.standard_normal((70, D)).astype(np.float32)
    u = rng.integers(0, 50, 1000).astype(np.int32)
    i = rng.integers(0, 70, 1000).astype(np.int32)
    out = ops.pair_dot(t(U, dev), t(I, dev), t(u, dev), t(i, dev)).cpu().numpy()
    np.testing.assert_allclose(out, ops_np.pair_dot(U, I, u, i), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------
# segments + scatter
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,V", [(0, 10), (1, 1), (1000, 7), (5000, 100_000), (200_000, 50_001),
                                 # the library's own LSD radix sort (up to 2^22 ids): 1 / 2 / 3 / 4 digit passes, chunk edges
                                 (2047, 200), (2048, 65_535), (2049, 65_537), (442_368, 11_000_002), (300_000, 100_000_003),
                                 (4_194_304, 1_000_000), (4_194_305, 1_000_000)])        # ... and the rocPRIM sort above it
def test_segments_bit_exact(dev, n, V):
    rng = np.random.default_rng(n + V)
    idx = zipf_ids(rng, V, n) if n else np.zeros(0, np.int32)
    if n >= 1000:
        idx[11], idx[500] = -3, V  # dropped entries
    seg = ops.build_segments(t(idx, dev), V)
    pos, rows, start = ops_np.segments(idx, V)
    ns = seg.count()
    assert ns == len(rows)
    np.testing.assert_array_equal(seg.rows[:ns].cpu().numpy(), rows)
    np.testing.assert_array_equal(seg.start[: ns + 1].cpu().numpy(), start)
    np.testing.assert_array_equal(seg.pos[: len(pos)].cpu().numpy(), pos)


def test_segments_all_invalid(dev):
    idx = np.full(100, -1, np.int32)
    seg = ops.build_segments(t(idx, dev), 10)
    assert seg.count() == 0 and int(seg.start[0]) == 0


@pytest.mark.parametrize("K", [1, 16, 64, 128, 12])
def test_segment_sum_and_scatter_add(dev, K):
    rng = np.random.default_rng(K)
    V, n = 1000, 20_000
    idx = zipf_ids(rng, V, n)
    grad = rng.standard_normal((n, K)).astype(np.float32)
    seg = ops.build_segments(t(idx, dev), V)
    ns = seg.count()
    grows = ops.embed_segment_sum(t(grad, dev), seg)[:ns].cpu().numpy()
    pos, rows, start = ops_np.segments(idx, V)
    # same ascending-position order as the oracle loop -> bit-exact; runs of more than 32 positions (Zipf head) are
    # summed chunk-wise by whole workgroups (csrc/embed_scatter.hip): another fixed order, fp32 rounding apart
    ref_rows = ops_np.segment_sum(grad, pos, start)
    short = np.diff(start) <= 32
    assert short.sum() > 0 and (~short).sum() > 0
    np.testing.assert_array_equal(grows[short], ref_rows[short])
    np.testing.assert_allclose(grows[~short], ref_rows[~short], rtol=1e-4, atol=2e-4)
    again = ops.embed_segment_sum(t(grad, dev), seg)[:ns].cpu().numpy()
    np.testing.assert_array_equal(grows, again)                      # run-to-run identical
    table = rng.standard_normal((V, K)).astype(np.float32)
    td = t(table, dev)
    ops.embed_scatter_add(td, t(grad, dev), seg, alpha=-0.5)
    ref = table.astype(np.float64) - 0.5 * ops_np.scatter_add_dense(V, idx, grad)
    np.testing.assert_allclose(td.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)  # grads 1e-4 (atomic-free order differs from add.at)


@pytest.mark.parametrize("tf_style", [True, False])
@pytest.mark.parametrize("K", [1, 16, 64, 128])
def test_scatter_adam_matches_oracle_on_touched_rows(dev, tf_style, K):
    rng = np.random.default_rng(5 + K)
    V, n = 500, 6000
    idx = zipf_ids(rng, V, n)
    w = rng.standard_normal((V, K)).astype(np.float32)
    m = (rng.standard_normal((V, K)) * 0.01).astype(np.float32)
    v = (rng.random((V, K)) * 0.01).astype(np.float32)
    grad = rng.standard_normal((n, K)).astype(np.float32)
    wd, md, vd = t(w, dev), t(m, dev), t(v, dev)
    seg = ops.build_segments(t(idx, dev), V)
    hp = ops.adam_hp(lr=1e-2, step=3, eps=1e-5 if tf_style else 1e-8, tf_style=tf_style,
                     weight_decay=0.0 if tf_style else 0.01)
    ops.embed_scatter_adam(wd, md, vd, t(grad, dev), seg, hp)
    pos, rows, start = ops_np.segments(idx, V)
    g = ops_np.segment_sum(grad, pos, start)
    w2, m2, v2 = w.copy(), m.copy(), v.copy()
    w2[rows], m2[rows], v2[rows] = ops_np.adam_step(
        w[rows], m[rows], v[rows], g, 1e-2, 3, eps=hp.eps, weight_decay=hp.weight_decay, tf_style=tf_style)
    # (runs of more than 32 positions are summed chunk-wise: a fixed order, but not the oracle loop's — fp32 rounding
    # of a sum of up to ~1,000 unit normals apart)
    np.testing.assert_allclose(wd.cpu().numpy(), w2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(md.cpu().numpy(), m2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vd.cpu().numpy(), v2, rtol=1e-5, atol=1e-6)
    untouched = np.setdiff1d(np.arange(V), rows)
    np.testing.assert_array_equal(wd.cpu().numpy()[untouched], w[untouched])  # lazy: untouched rows frozen


def test_adam_dense_tf_semantics(dev):
    rng = np.random.default_rng(9)
    V, K, n = 300, 16, 1000
    idx = zipf_ids(rng, V, n)
    w = rng.standard_normal((V, K)).astype(np.float32)
    m = (rng.standard_normal((V, K)) * 0.01).astype(np.float32)
    v = (rng.random((V, K)) * 0.01).astype(np.float32)
    grad = rng.standard_normal((n, K)).astype(np.float32)
    wd, md, vd = t(w, dev), t(m, dev), t(v, dev)
    seg = ops.build_segments(t(idx, dev), V)
    grows = ops.embed_segment_sum(t(grad, dev), seg)
    slot = torch.full((V,), -1, dtype=torch.int32, device=dev)
    hp = ops.adam_hp(lr=1e-2, step=2, eps=1e-5)
    ops.adam_dense(wd, md, vd, hp, grows=grows, seg=seg, row_slot=slot, l2=1e-3)
    assert bool((slot == -1).all())
    gd = ops_np.scatter_add_dense(V, idx, grad).astype(np.float32) + np.float32(2e-3) * w
    w2, m2, v2 = ops_np.adam_step(w, m, v, gd, 1e-2, 2, eps=1e-5)
    np.testing.assert_allclose(wd.cpu().numpy(), w2, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(md.cpu().numpy(), m2, rtol=1e-4, atol=1e-6)
    # no sparse part: every row still decays (TF1 dense semantics)
    wd2, md2, vd2 = t(w, dev), t(m, dev), t(v, dev)
    ops.adam_dense(wd2, md2, vd2, hp)
    w3, m3, v3 = ops_np.adam_step(w, m, v, np.zeros_like(w), 1e-2, 2, eps=1e-5)
    np.testing.assert_allclose(wd2.cpu().numpy(), w3, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------
# FM
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [16, 32, 64, 128, 256, 8])
@pytest.mark.parametrize("B,F", [(1, 2), (33, 7), (300, 202)])
def test_fm_pairwise_fwd_bwd(dev, K, B, F):
    rng = np.random.default_rng(B * F + K)
    e = (rng.standard_normal((B, F, K)) * 0.3).astype(np.float32)
    pair, fsum = ops.fm_pairwise_fwd(t(e, dev))
    rp, rs = ops_np.fm_pairwise(e.astype(np.float64))
    np.testing.assert_allclose(pair.cpu().numpy(), rp, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(fsum.cpu().numpy(), rs, rtol=1e-5, atol=1e-5)
    gp = rng.standard_normal((B, K)).astype(np.float32)
    ge = ops.fm_pairwise_bwd(t(e, dev), fsum, t(gp, dev)).cpu().numpy()
    np.testing.assert_allclose(ge, ops_np.fm_pairwise_bwd(e.astype(np.float64), gp), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("K,F", [(16, 5), (64, 202), (128, 39)])
def test_fm_embed_fused_forward(dev, K, F):
    rng = np.random.default_rng(K + F)
    V, B = 5000, 257
    table = (rng.standard_normal((V, K)) * 0.1).astype(np.float32)
    idx = zipf_ids(rng, V, (B, F))
    e, pair, fsum = ops.fm_embed_fwd(t(table, dev), t(idx, dev))
    np.testing.assert_array_equal(e.cpu().numpy(), table[idx])  # gather part is a pure copy
    rp, rs = ops_np.fm_pairwise(table[idx].astype(np.float64))
    np.testing.assert_allclose(pair.cpu().numpy(), rp, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fsum.cpu().numpy(), rs, rtol=1e-5, atol=1e-5)
    e2, pair2, _ = ops.fm_embed_fwd(t(table, dev), t(idx, dev), want_e=False)
    assert e2 is None and torch.equal(pair2, pair)


@pytest.mark.parametrize("with_deep,with_lin,with_bn", [(True, True, True), (False, False, False), (True, False, True), (False, True, False)])
@pytest.mark.parametrize("K", [16, 64, 128])
def test_fm_embed_fused_backward_adam(dev, with_deep, with_lin, with_bn, K):
    """Fused bwd == (oracle FM backward (+BN-fold terms) -> segment sum -> Adam) on every touched
    row, including rows whose run is long enough (> 32 positions) for the workgroup path."""
    rng = np.random.default_rng(11 + K)
    V, B, F = 800, 300, 9
    table = (rng.standard_normal((V, K)) * 0.1).astype(np.float32)
    lin = (rng.standard_normal((V, 1)) * 0.1).astype(np.float32)
    m = np.zeros((V, K), np.float32)
    v = np.zeros((V, K), np.float32)
    idx = zipf_ids(rng, V, (B, F))
    assert np.bincount(idx.reshape(-1)).max() > 100  # a hot row (> kLongSeg positions)
    e = table[idx]
    gdeep = (rng.standard_normal((B, F, K)) * 0.1).astype(np.float32) if with_deep else None
    gpair = rng.standard_normal((B, K)).astype(np.float32)
    glin = rng.standard_normal((B, F)).astype(np.float32)
    bn_a = (rng.standard_normal((F, K)) * 0.05).astype(np.float32)
    bn_c = (rng.standard_normal((F, K)) * 0.05).astype(np.float32)
    td, md, vd = t(table, dev), t(m, dev), t(v, dev)
    ld, lmd, lvd = t(lin, dev), torch.zeros((V, 1), device=dev), torch.zeros((V, 1), device=dev)
    out = ops.fm_embed_fwd(td, t(idx, dev), want_e=False, lin=ld if with_lin else None)
    fsum = out[2]
    if with_lin:
        np.testing.assert_array_equal(out[3].cpu().numpy(), lin[idx][..., 0])
    seg = ops.build_segments(t(idx.reshape(-1), dev), V)
    hp = ops.adam_hp(lr=1e-3, step=1, eps=1e-5)
    kw = {}
    if with_lin:
        kw.update(lin=ld, lin_m=lmd, lin_v=lvd, glin=t(glin, dev))
    if with_bn:
        kw.update(bn_a=t(bn_a, dev), bn_c=t(bn_c, dev))
    ops.fm_embed_bwd_adam(td, md, vd, t(gdeep, dev) if with_deep else None, t(gpair, dev), fsum, B, F, seg, hp, **kw)
    ge = ops_np.fm_pairwise_bwd(e.astype(np.float64), gpair.astype(np.float64))
    if with_deep:
        ge = ge + gdeep
    if with_bn:
        ge = ge - bn_a[None] - bn_c[None] * e
    gd = ops_np.scatter_add_dense(V, idx, ge)
    rows = np.unique(idx)
    w2, m2, v2 = ops_np.adam_step(table[rows].astype(np.float64), m[rows].astype(np.float64),
                                  v[rows].astype(np.float64), gd[rows], 1e-3, 1, eps=1e-5)
    np.testing.assert_allclose(md.cpu().numpy()[rows], m2, rtol=1e-4, atol=2e-6)   # gradients 1e-4
    np.testing.assert_allclose(vd.cpu().numpy()[rows], v2, rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(td.cpu().numpy()[rows], w2, rtol=1e-5, atol=1e-5)
    if with_lin:
        gl = ops_np.scatter_add_dense(V, idx, glin[..., None])
        l2, lm2, _ = ops_np.adam_step(lin[rows].astype(np.float64), np.zeros((len(rows), 1)),
                                      np.zeros((len(rows), 1)), gl[rows], 1e-3, 1, eps=1e-5)
        np.testing.assert_allclose(lmd.cpu().numpy()[rows], lm2, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ld.cpu().numpy()[rows], l2, rtol=1e-5, atol=1e-5)
    # run-to-run determinism (no float atomics, list order irrelevant)
    td2, md2, vd2 = t(table, dev), t(m, dev), t(v, dev)
    kw2 = dict(kw)
    if with_lin:
        kw2.update(lin=t(lin, dev), lin_m=torch.zeros((V, 1), device=dev), lin_v=torch.zeros((V, 1), device=dev))
    ops.fm_embed_bwd_adam(td2, md2, vd2, t(gdeep, dev) if with_deep else None, t(gpair, dev), fsum, B, F, seg, hp, **kw2)
    assert torch.equal(td2, td) and torch.equal(md2, md)


def test_scatter_adam_lin_equals_two_scatter_adams(dev):
    """`lr_embed_scatter_adam_lin_f32` (owner-side update of the row-sharded tables: embedding rows and their linear
    weights from one pass over the segments) == `lr_embed_scatter_adam_f32` on each table, bit for bit."""
    rng = np.random.default_rng(21)
    V, K, n = 5000, 64, 20000
    ids = rng.integers(0, V, n).astype(np.int32)
    ids[:3000] = rng.integers(0, 7, 3000)                      # long runs
    table = rng.standard_normal((V, K)).astype(np.float32) * 0.1
    lin = rng.standard_normal((V, 1)).astype(np.float32) * 0.1
    grad = rng.standard_normal((n, K)).astype(np.float32)
    glin = rng.standard_normal(n).astype(np.float32)
    seg = ops.build_segments(t(ids, dev), V)
    hp = ops.adam_hp(1e-2, 3)
    a = [t(table, dev), torch.rand((V, K), device=dev), torch.rand((V, K), device=dev),
         t(lin, dev), torch.rand((V, 1), device=dev), torch.rand((V, 1), device=dev)]
    b = [x.clone() for x in a]
    ops.embed_scatter_adam(a[0], a[1], a[2], t(grad, dev), seg, hp)
    ops.embed_scatter_adam(a[3], a[4], a[5], t(glin, dev).view(-1, 1), seg, hp)
    ops.embed_scatter_adam_lin(b[0], b[1], b[2], t(grad, dev), b[3], b[4], b[5], t(glin, dev), seg, hp)
    # rows with runs of at most 32 positions: the same ascending order in both kernels -> bit for bit; the 7 head rows
    # go through the chunked long-run path in `embed_scatter_adam` only (another fixed summation order)
    head = torch.arange(7, device=dev)
    rest = torch.arange(7, V, device=dev)
    for x, y in zip(a, b):
        assert torch.equal(x[rest], y[rest])
        torch.testing.assert_close(x[head], y[head], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------
# SpMM
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [16, 64, 128, 10])
def test_spmm_csr(dev, K):
    import scipy.sparse as ssp

    rng = np.random.default_rng(K)
    n = 3000
    A = ssp.random(n, n, density=0.004, format="csr", dtype=np.float32, random_state=1)
    A.sort_indices()
    X = rng.standard_normal((n, K)).astype(np.float32)
    rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32)
    acc0 = rng.standard_normal((n, K)).astype(np.float32)
    acc = t(acc0, dev)
    Y = ops.spmm_csr(t(rp, dev), t(ci, dev), t(va, dev), t(X, dev), acc=acc).cpu().numpy()
    ref = ops_np.spmm_csr(rp, ci, va, X.astype(np.float64))
    np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(acc.cpu().numpy(), acc0 + ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("K", [16, 64, 128])
def test_spmm_csr_zipf_degrees_long_rows(dev, K):
    """Degree-bucketed path: empty rows, rows at the bucket edges (128 / 129 nonzeros: row group vs whole
    workgroup), a single-chunk long row (2,048), multi-chunk rows (2,049 and 70,001 nonzeros: chunk sums added in
    chunk order) — against the fp64 restatement; repeated launches are bit-identical."""
    rng = np.random.default_rng(100 + K)
    n_cols = 50_000
    degs = np.concatenate([np.array([0, 1, 3, 4, 5, 127, 128, 129, 300, 2047, 2048, 2049, 4096, 70_001, 0, 9000]),
                           np.minimum((rng.pareto(1.1, 3000) * 3).astype(np.int64), 5000)])
    rng.shuffle(degs)
    rp = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    nnz = int(rp[-1])
    ci = rng.integers(0, n_cols, nnz).astype(np.int32)
    va = rng.uniform(0.0, 1.0, nnz).astype(np.float32) / np.sqrt(np.repeat(np.maximum(degs, 1), degs)).astype(np.float32)
    X = rng.standard_normal((n_cols, K)).astype(np.float32)
    acc0 = rng.standard_normal((len(degs), K)).astype(np.float32)
    ref = ops_np.spmm_csr(rp, ci, va, X.astype(np.float64))
    outs = []
    for _ in range(3):
        acc = t(acc0, dev)
        Y = ops.spmm_csr(t(rp, dev), t(ci, dev), t(va, dev), t(X, dev), acc=acc)
        outs.append((Y.clone(), acc.clone()))
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), acc0 + ref, rtol=2e-5, atol=2e-5)
    for y, a in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(a, outs[0][1])


@pytest.mark.parametrize("K", [16, 64, 128])
def test_spmm_csr_row_bitmaps_are_bit_exact(dev, K):
    """`x_rows`: X is zero outside the bitmap — the masked product (zero rows not read) equals the plain one bit for bit;
    `y_rows`: the rows inside the bitmap equal the plain product, the others keep what the buffer held.  Square graph with
    short, 128 / 129-nonzero, single- and multi-chunk rows, the marked rows drawn among all of them."""
    rng = np.random.default_rng(7 + K)
    degs = np.concatenate([np.array([0, 1, 5, 127, 128, 129, 300, 2048, 2049, 4097, 30_001, 0]),
                           np.minimum((rng.pareto(1.1, 4000) * 3).astype(np.int64), 5000)])
    rng.shuffle(degs)
    n = len(degs)
    rp = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    nnz = int(rp[-1])
    ci = rng.integers(0, n, nnz).astype(np.int32)
    va = rng.uniform(0.1, 1.0, nnz).astype(np.float32)
    long_rows = np.flatnonzero(degs > 128)
    marked = np.unique(np.concatenate([rng.integers(0, n, 300), long_rows[::2], [0, n - 1]])).astype(np.int32)
    ids = t(np.concatenate([marked, marked[:50], [-1]]).astype(np.int32), dev)       # duplicates and a dropped id
    rpd, cid, vad = t(rp, dev), t(ci, dev), t(va, dev)
    plan = ops.SpmmPlan(rpd, nnz, K)
    bm = ops.RowBitmap(n, dev).set(ids)
    X = np.zeros((n, K), np.float32)
    X[marked] = rng.standard_normal((len(marked), K)).astype(np.float32)
    Xd = t(X, dev)
    plain = ops.spmm_csr(rpd, cid, vad, Xd, plan=plan)
    masked = ops.spmm_csr(rpd, cid, vad, Xd, out=torch.empty_like(plain), plan=plan, x_rows=bm)
    assert torch.equal(plain, masked)
    np.testing.assert_allclose(plain.cpu().numpy(), ops_np.spmm_csr(rp, ci, va, X.astype(np.float64)), rtol=2e-5, atol=2e-5)
    Xf = t(rng.standard_normal((n, K)).astype(np.float32), dev)
    full = ops.spmm_csr(rpd, cid, vad, Xf, plan=plan)
    out = torch.full_like(full, 7.0)
    ops.spmm_csr(rpd, cid, vad, Xf, out=out, plan=plan, y_rows=bm)
    sel = torch.zeros(n, dtype=torch.bool, device=dev)
    sel[t(marked, dev).long()] = True
    assert torch.equal(out[sel], full[sel]) and bool((out[~sel] == 7.0).all())
    both = torch.full_like(full, 7.0)
    ops.spmm_csr(rpd, cid, vad, Xd, out=both, plan=plan, x_rows=bm, y_rows=bm)
    assert torch.equal(both[sel], plain[sel]) and bool((both[~sel] == 7.0).all())
    bm.clear(ids)
    assert int(bm.words.abs().sum()) == 0
    with pytest.raises(ValueError):
        ops.spmm_csr(rpd, cid, vad, Xd, plan=plan, y_rows=bm)                     # no buffer for the unwritten rows
    with pytest.raises(ValueError):
        ops.spmm_csr(rpd, cid, vad, Xd, out=both, x_rows=bm)                      # no plan


@pytest.mark.parametrize("amsgrad,wd", [(False, 0.0), (True, 0.0), (True, 0.01)])
def test_adam_dense_torch_style_matches_torch_optim(dev, amsgrad, wd):
    """`lr_adam_dense_f32` (torch style, optional AMSGrad / weight decay) against torch.optim.Adam on
    CPU over several steps — the optimiser of the torch-backend models (torch_trainer.py:63-69)."""
    torch.manual_seed(0)
    w0 = torch.randn(37, 16)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=1e-2, eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    w = w0.clone().to(dev)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    vmax = torch.zeros_like(w) if amsgrad else None
    for step in range(1, 6):
        g = torch.randn(37, 16) * (0.1 if step % 2 else 3.0)       # v goes up and down: max matters
        p.grad = g.clone()
        opt.step()
        ops.adam_dense(w, m, v, ops.adam_hp(1e-2, step, eps=1e-8, weight_decay=wd, tf_style=False),
                       grows=g.to(dev), vmax=vmax)
    torch.testing.assert_close(w.cpu(), p.detach(), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("K", [16, 64])
def test_fm_field_stats_equal_batch_statistics(dev, K):
    """`lr_fm_field_stats_f32`: mean / biased variance of e[B,F,K] from the runs (distinct rows x run
    length) == the statistics of the materialised block (fp64 shadow)."""
    rng = np.random.default_rng(K)
    sizes = [40, 25, 7, 300, 1]                               # rows per field (user, item, 3 sparse columns)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    V, F, B = int(starts[-1]), len(sizes), 512
    table = rng.standard_normal((V, K)).astype(np.float32)
    idx = np.stack([rng.zipf(1.3, B) % sizes[f] + starts[f] for f in range(F)], axis=1).astype(np.int32)
    td, idd = t(table, dev), t(idx, dev)
    seg = ops.build_segments(idd.reshape(-1), V)
    mean, var = ops.fm_field_stats(td, seg, t(starts, dev), B, chunks=3)
    e = table[idx].astype(np.float64).reshape(B, F * K)
    np.testing.assert_allclose(mean.cpu().numpy(), e.mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var.cpu().numpy(), e.var(0), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("arith", ["split_bf16", "f32_chain"])
@pytest.mark.parametrize("B,N,H1,H2", [(1, 1, 128, 64), (3, 200, 128, 64), (70, 1000, 64, 32), (130, 333, 128, 32), (5, 129, 64, 64)])
def test_pair_mlp_matches_torch(dev, B, N, H1, H2, arith):
    """`lr_pair_mlp_f32` / `lr_pair_mlp_sb_f32` (MLP tail of every (user, item) pair of a DeepFM catalogue ranking; the f32 fma
    chain and the six-term split-bf16 form) against the same expression in torch fp64 at the same tolerance; accumulate /
    overwrite, ragged tiles and user blocks, a column slice as output."""
    g = torch.Generator(device=dev).manual_seed(B * 1000 + N)
    P = torch.randn((B, H1), device=dev, generator=g)
    Q = torch.randn((N, H1), device=dev, generator=g)
    W2 = torch.randn((H1, H2), device=dev, generator=g) / H1 ** 0.5
    b2 = torch.randn(H2, device=dev, generator=g)
    v3 = torch.randn(H2, device=dev, generator=g)
    c3 = 0.37
    ref = (torch.relu(torch.relu(P.double()[:, None, :] + Q.double()[None, :, :]) @ W2.double() + b2.double()) @ v3.double()) + c3
    wide = torch.full((B, N + 5), 2.0, device=dev)
    ops.pair_mlp(P, Q, W2, b2, v3, c3, wide[:, 2:2 + N], accumulate=True, arith=arith)
    torch.testing.assert_close(wide[:, 2:2 + N].double(), ref + 2.0, rtol=1e-5, atol=1e-5)
    assert bool((wide[:, :2] == 2.0).all()) and bool((wide[:, 2 + N:] == 2.0).all())
    out = torch.empty((B, N), device=dev)
    ops.pair_mlp(P, Q, W2, b2, v3, c3, out, accumulate=False, arith=arith)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)


def test_pair_mlp_split_bf16_error_is_the_f32_chains(dev):
    """Both arithmetics of the pair MLP against fp64 on operands with a wide dynamic range: the split form's worst and rms
    error within 1.5x the chain's (+ half an ulp of the output scale)."""
    g = torch.Generator(device=dev).manual_seed(5)
    B, N, H1, H2 = 96, 4000, 128, 64
    P = torch.randn((B, H1), device=dev, generator=g) * torch.exp(torch.randn((B, H1), device=dev, generator=g))
    Q = torch.randn((N, H1), device=dev, generator=g) * torch.exp(torch.randn((N, H1), device=dev, generator=g))
    W2 = torch.randn((H1, H2), device=dev, generator=g) / H1 ** 0.5
    b2, v3 = torch.randn(H2, device=dev, generator=g), torch.randn(H2, device=dev, generator=g)
    hid = torch.relu(torch.relu(P.double()[:, None, :] + Q.double()[None, :, :]) @ W2.double() + b2.double())
    ref = hid @ v3.double()
    scale = (torch.relu(P.double()[:, None, :] + Q.double()[None, :, :]) @ W2.double().abs() + b2.double().abs()) @ v3.double().abs()
    err = {}
    for arith in ("f32_chain", "split_bf16"):
        out = torch.empty((B, N), device=dev)
        ops.pair_mlp(P, Q, W2, b2, v3, 0.0, out, accumulate=False, arith=arith)
        e = ((out.double() - ref) / scale).abs()
        err[arith] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    ulp = 2.0 ** -24
    assert err["split_bf16"][0] <= 1.5 * err["f32_chain"][0] + ulp, err
    assert err["split_bf16"][1] <= 1.5 * err["f32_chain"][1] + ulp / 4, err
