"""DIN attention pooling (layers/attention.py:28-64) vs the oracle restatement; backward vs
torch autograd (fp64) of the same restatement (`-m gpu`).  Tolerances: forward 1e-5/1e-6 against
an fp64 shadow, gradients 1e-4."""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from oracle import ops_np

pytestmark = pytest.mark.gpu


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def make_case(K, B, L, V, seed):
    rng = np.random.default_rng(seed)
    table = (rng.standard_normal((V, K)) * 0.5).astype(np.float32)
    item = rng.integers(0, V - 1, B).astype(np.int32)
    lens = rng.integers(1, L + 1, B).astype(np.int32)
    lens[0], lens[1 % B] = L, 1
    seq = np.full((B, L), V - 1, np.int32)  # pad id = last row (n_items)
    for b in range(B):
        seq[b, : lens[b]] = rng.integers(0, V - 1, lens[b])
    W1 = (rng.standard_normal((4 * K, 16)) / np.sqrt(4 * K)).astype(np.float32)
    b1 = (rng.standard_normal(16) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((16, 1)) * 0.5).astype(np.float32)
    b2 = (rng.standard_normal(1) * 0.1).astype(np.float32)
    return table, item, seq, lens, W1, b1, W2, b2


def torch_ref(table, item, seq, lens, W1, b1, W2, b2, gout):
    """fp64 autograd of the oracle's arithmetic, gradients w.r.t. gathered rows and MLP params."""
    d = torch.float64
    tb = torch.from_numpy(table).to(d)
    q = tb[torch.from_numpy(item).long()].clone().requires_grad_(True)
    keys = tb[torch.from_numpy(seq).long()].clone().requires_grad_(True)
    P = [torch.from_numpy(x).to(d).requires_grad_(True) for x in (W1, b1, W2, b2)]
    B, L, K = keys.shape
    qt = q[:, None, :].expand(-1, L, -1)
    cross = torch.cat([qt, keys, qt - keys, qt * keys], dim=2)
    h = torch.sigmoid(cross @ P[0] + P[1])
    s = (h @ P[2]).reshape(B, L) + P[3]
    s = s / np.sqrt(K)
    mask = torch.arange(L)[None, :] < torch.from_numpy(lens).long()[:, None]
    s = torch.where(mask, s, torch.full_like(s, -(2.0 ** 32) + 1))
    a = torch.softmax(s, dim=1)
    out = (a[:, None, :] @ keys).reshape(B, K)
    out.backward(torch.from_numpy(gout).to(d))
    return (out.detach().numpy(), a.detach().numpy(), q.grad.numpy(), keys.grad.numpy(),
            [p.grad.numpy() for p in P])


@pytest.mark.parametrize("K,B,L", [(16, 37, 10), (32, 64, 7), (64, 130, 50), (128, 257, 50), (128, 3, 1)])
def test_din_attention_fused_gather(dev, K, B, L):
    V = 5000
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=K + B)
    rng = np.random.default_rng(1)
    gout = rng.standard_normal((B, K)).astype(np.float32)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    out, attn = ops.din_attn_pool_fwd(*args)
    q, keys = table[item], table[seq]
    o_np, a_np = ops_np.din_attention(q.astype(np.float64), keys.astype(np.float64), lens,
                                      W1.astype(np.float64), b1.astype(np.float64),
                                      W2.astype(np.float64), b2.astype(np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), o_np, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(attn.cpu().numpy(), a_np, rtol=1e-5, atol=1e-6)
    assert np.all(attn.cpu().numpy()[np.arange(L)[None, :] >= lens[:, None]] == 0)

    r_out, r_a, r_gq, r_gk, r_gp = torch_ref(table, item, seq, lens, W1, b1, W2, b2, gout)
    np.testing.assert_allclose(o_np, r_out, rtol=1e-9, atol=1e-12)  # oracle == its torch twin
    gq, gkey, gW1, gb1, gW2, gb2 = ops.din_attn_pool_bwd(*args, attn, t(gout, dev))
    np.testing.assert_allclose(gq.cpu().numpy(), r_gq, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gkey.cpu().numpy(), r_gk, rtol=1e-4, atol=1e-5)
    for got, want, name in ((gW1, r_gp[0], "gW1"), (gb1, r_gp[1], "gb1"), (gW2, r_gp[2], "gW2"), (gb2, r_gp[3], "gb2")):
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * scale, err_msg=name)


def test_din_attention_dense_variant_matches_gather(dev):
    K, B, L, V = 64, 50, 12, 300
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=3)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    out, attn = ops.din_attn_pool_fwd(*args)
    q = args[0][args[1].long()].contiguous()
    keys = args[0][args[2].long()].contiguous()
    out2, attn2 = ops.din_attn_dense_fwd(q, keys, args[3], *args[4:])
    assert torch.equal(out, out2) and torch.equal(attn, attn2)
    gout = torch.randn((B, K), device=dev)
    a = ops.din_attn_pool_bwd(*args, attn, gout)
    b = ops.din_attn_dense_bwd(q, keys, args[3], *args[4:], attn, gout)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_din_attention_deterministic(dev):
    K, B, L, V = 128, 600, 50, 10_000
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=9)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    out, attn = ops.din_attn_pool_fwd(*args)
    gout = torch.randn((B, K), device=dev)
    r1 = ops.din_attn_pool_bwd(*args, attn, gout)
    r2 = ops.din_attn_pool_bwd(*args, attn, gout)
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)  # no atomics anywhere: bitwise run-to-run identical


@pytest.mark.parametrize("K", [32, 128])
def test_din_attention_backward_halves_and_pad_rows(dev, K):
    """`lr_din_attn_pool_bwd_parts_f32`: data half then parameter half == the one-call backward (bit for bit); with
    `keep_pad_rows` the gradient rows past a sample's length keep whatever the buffer held, everything else is equal."""
    B, L, V = 300, 50, 8_000
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=21 + K)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    out, attn = ops.din_attn_pool_fwd(*args)
    gout = torch.randn((B, K), device=dev)
    ref = ops.din_attn_pool_bwd(*args, attn, gout)
    lib = ops._lib.load()
    ws = torch.empty(lib.lr_din_attn_ws_bytes(B, L, K, 16), dtype=torch.uint8, device=dev)
    gq = torch.empty((B, K), device=dev)
    gkey = torch.full((B, L, K), 7.0, device=dev)
    pout = tuple(torch.full_like(a, 3.0) for a in (args[4], args[5], args[6], args[7]))
    ops.din_attn_pool_bwd(*args, attn, gout, gq_out=gq, gkey_out=gkey, param_out=pout, ws=ws, parts=1, keep_pad_rows=True)
    assert all(bool((p == 3.0).all()) for p in pout)                  # the data half leaves the parameter gradients alone
    ops.din_attn_pool_bwd(*args, attn, gout, gq_out=gq, gkey_out=gkey, param_out=pout, ws=ws, parts=2)
    valid = (torch.arange(L, device=dev)[None, :] < args[3][:, None].long())
    assert torch.equal(gq, ref[0])
    assert torch.equal(gkey[valid], ref[1][valid])
    assert bool((gkey[~valid] == 7.0).all()) and bool((ref[1][~valid] == 0).all())
    for a, b_ in zip(pout, ref[2:]):
        assert torch.equal(a.view(-1), b_.view(-1))
    with pytest.raises(ValueError):
        ops.din_attn_pool_bwd(*args, attn, gout, parts=2)              # a half needs the caller's workspace


@pytest.mark.parametrize("K,B,L", [(16, 37, 10), (64, 130, 50), (128, 257, 50), (128, 3, 1)])
def test_din_backward_from_saved_hidden_activations(dev, K, B, L):
    """Round 6: the forward keeps h = sigmoid(z) of every live (sample, key) pair (`hid`), the data half of the backward reads
    it instead of recomputing the attention MLP's first layer (layers/attention.py:50-56).  Same bar against the fp64 torch twin
    as the recomputing form; the two forms agree to f32 rounding (the forward's folded product and the backward's unfolded one
    round differently, so not bit for bit); rows of `hid` past a sample's length are never written; run-to-run identical."""
    V = 5000
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=K + B + 1)
    rng = np.random.default_rng(2)
    gout = rng.standard_normal((B, K)).astype(np.float32)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    hid = torch.full((B * L * 16,), 123.0, device=dev)
    out_h, attn_h = ops.din_attn_pool_fwd(*args, hid=hid)
    out, attn = ops.din_attn_pool_fwd(*args)
    assert torch.equal(out, out_h) and torch.equal(attn, attn_h)          # keeping h changes nothing in the forward
    pad = torch.from_numpy(np.arange(L)[None, :] >= lens[:, None]).to(dev)
    assert bool((hid.view(B, L, 16)[pad] == 123.0).all()) and bool(((hid.view(B, L, 16)[~pad] > 0) & (hid.view(B, L, 16)[~pad] < 1)).all())
    _, _, r_gq, r_gk, r_gp = torch_ref(table, item, seq, lens, W1, b1, W2, b2, gout)
    got = ops.din_attn_pool_bwd(*args, attn, t(gout, dev), hid=hid)
    plain = ops.din_attn_pool_bwd(*args, attn, t(gout, dev))
    np.testing.assert_allclose(got[0].cpu().numpy(), r_gq, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got[1].cpu().numpy(), r_gk, rtol=1e-4, atol=1e-5)
    for g_, want, name in zip(got[2:], r_gp, ("gW1", "gb1", "gW2", "gb2")):
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(g_.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * scale, err_msg=name)
    for x, y in zip(got, plain):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=2e-6)
    again = ops.din_attn_pool_bwd(*args, attn, t(gout, dev), hid=hid)
    for x, y in zip(got, again):
        assert torch.equal(x, y)


def test_din_saved_hidden_needs_a_compiled_width(dev):
    table, item, seq, lens, W1, b1, W2, b2 = make_case(24, 5, 4, 50, seed=1)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    with pytest.raises(ValueError):
        ops.din_attn_pool_fwd(*args, hid=torch.empty(5 * 4 * 16, device=dev))


@pytest.mark.parametrize("K,B,L", [(64, 130, 50), (128, 1000, 50), (32, 257, 300), (128, 3, 1), (16, 70_000, 50), (16, 300_000, 4)])
def test_din_balanced_sample_order(dev, K, B, L):
    """Round 6: the forward's launch also writes the samples as a stable partition by descending key-tile count (`order_out`);
    the backward kernels walk the batch in that order (`order`).  The permutation is checked against numpy's stable argsort of the classes;
    per-sample outputs (out, attn, gq, gkey) are bit-identical to the identity order — each sample is computed by one wave
    from its own rows — and the parameter gradients, which are summed per wave, agree to f32 rounding and are run-to-run
    identical."""
    V = 5000
    table, item, seq, lens, W1, b1, W2, b2 = make_case(K, B, L, V, seed=K + B + 7)
    args = [t(x, dev) for x in (table, item, seq, lens, W1, b1, W2, b2)]
    order = torch.full((B,), -1, dtype=torch.int32, device=dev)
    out0, attn0 = ops.din_attn_pool_fwd(*args)
    out1, attn1 = ops.din_attn_pool_fwd(*args, order_out=order)
    assert torch.equal(out0, out1) and torch.equal(attn0, attn1)          # the forward itself is unchanged by writing the order
    nclass = min((L + 15) // 16, 16)
    nchunks = (B + 63) // 64
    if nclass * nchunks > 4096:                    # very large batches: fewer classes (4,096 counters in LDS); none: identity
        nclass = 4096 // nchunks
    if nclass < 1:
        nclass = 1                                 # (one class = the identity order)
    tiles = np.minimum((np.clip(lens, 0, L) + 15) // 16, nclass)
    tiles[np.clip(lens, 0, L) == 0] = 1
    want = np.argsort(-tiles, kind="stable").astype(np.int32)
    np.testing.assert_array_equal(order.cpu().numpy(), want)
    gout = torch.randn((B, K), device=dev, generator=torch.Generator(device=dev).manual_seed(B + K))
    r0 = ops.din_attn_pool_bwd(*args, attn0, gout)
    r1 = ops.din_attn_pool_bwd(*args, attn0, gout, order=order)
    assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])
    for x, y in zip(r0[2:], r1[2:]):
        # (sums over B x L terms of both signs taken in two orders: the rounding noise grows with sqrt(B), not with |sum|)
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5 * max(1.0, float(x.abs().max())) * max(1.0, (B / 4096) ** 0.5))
    r2 = ops.din_attn_pool_bwd(*args, attn0, gout, order=order)
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)
    # hid + order together: the forward's extra workgroup also leaves the data kernel's weight images behind `hid`
    hid = torch.empty(ops.din_hid_floats(B, L, K), device=dev)
    order2 = torch.empty_like(order)
    ops.din_attn_pool_fwd(*args, hid=hid, order_out=order2)
    assert torch.equal(order2, order)
    hid_only = torch.empty(ops.din_hid_floats(B, L, K, with_order=False), device=dev)
    ops.din_attn_pool_fwd(*args, hid=hid_only)
    r3 = ops.din_attn_pool_bwd(*args, attn0, gout, hid=hid, order=order)
    r4 = ops.din_attn_pool_bwd(*args, attn0, gout, hid=hid_only)
    assert torch.equal(r3[0], r4[0]) and torch.equal(r3[1], r4[1])        # same images, same h: per-sample outputs bit-identical
    with pytest.raises(ValueError):
        ops.din_attn_pool_bwd(*args, attn0, gout, hid=hid_only, order=order)   # too small to hold the images
