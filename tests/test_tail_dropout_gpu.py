"""Dropout on the hand-written steps (round 4; `layers/dense.py:44-47` of the reference: tf.layers.dropout after every hidden
layer's BatchNorm).  The tail kernels draw a counter-based mask keep(seed, layer, sample, column); the test restates that
function in torch, hands it to the fp64 oracle's `dense_nn` and compares one training step — fused DeepFM (plain sparse
columns) and the feature-block DeepFM (multi-sparse + dense columns).  Also: the kept fraction, inference without dropout."""
import numpy as np
import pytest
import torch

from librecommender_amd.nets import DeepFMNet, FeatDeepFMNet
from oracle.models_torch import DeepFMOracle, FeatDeepFMOracle, export_fieldnet_weights
from tests.test_feat_block_gpu import batch as feat_batch, make_spec

pytestmark = pytest.mark.gpu


def keep_mask(seed, layer, B, d, keep):
    """The kernels' mask (csrc/deepfm_tail.hip:drop_scale) in int64 torch arithmetic (wrapping; logical shifts by masking)."""
    def c64(x):
        x &= (1 << 64) - 1
        return x - (1 << 64) if x >= (1 << 63) else x
    b = torch.arange(B, dtype=torch.int64)[:, None]
    c = torch.arange(d, dtype=torch.int64)[None, :]
    x = (b * 4096 + c) ^ c64(seed * 0x9E3779B97F4A7C15 + layer * 0xD1B54A32D192ED03)
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * c64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * c64(0x94D049BB133111EB)
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    u = (x & 0xFFFFFF).to(torch.float64) / 16777216.0
    return (u < keep).to(torch.float64)


def dropout_fn(seed, keep):
    return lambda layer, x: x * keep_mask(seed, layer, x.shape[0], x.shape[1], keep) / keep


@pytest.mark.parametrize("use_bn", [True, False])
def test_fused_deepfm_step_with_dropout_equals_oracle(dev, use_bn):
    nu, ni, vocab, Fs, B, K, hidden, rate = 300, 200, 37, 9, 640, 64, (128, 64, 32), 0.3
    net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=hidden, use_bn=use_bn, dropout_rate=rate, lr=1e-2,
                    device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    assert net.fused_l1 and net.hip_tail
    net.enable_graph(True)                                   # ignored with dropout: the mask seed is passed by value
    oracle = DeepFMOracle(export_fieldnet_weights(net), hidden, use_bn=use_bn, lr=1e-2, dtype=torch.float64)
    oracle.mlp.dropout_fn = dropout_fn(1, 1.0 - rate)         # the tail's first step uses seed 1
    rng = np.random.default_rng(1)
    users, items = rng.integers(0, nu + 1, B), rng.integers(0, ni + 1, B)
    sparse = (rng.zipf(1.3, (B, Fs)) - 1) % vocab + np.arange(Fs) * (vocab + 1)
    labels = rng.integers(0, 2, B).astype(np.float32)
    idx = net.tables.global_idx(*(torch.from_numpy(x).to(dev) for x in (users, items, sparse)))
    l_net = float(net.train_step(idx, torch.from_numpy(labels).to(dev)))
    l_or = float(oracle.train_step(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(sparse), torch.from_numpy(labels)))
    assert abs(l_net - l_or) < 1e-5
    W1 = export_fieldnet_weights(net)
    for name, want in oracle.V.v.items():
        np.testing.assert_allclose(W1[name].numpy().reshape(want.shape), want.detach().numpy(), rtol=1e-4, atol=5e-5, err_msg=name)
    # a different mask on the next step, and no dropout at inference
    assert net._tail.drop_seed == 1
    net.train_step(idx, torch.from_numpy(labels).to(dev))
    assert net._tail.drop_seed == 2 and not net._graphs
    a, b = net.forward(idx), net.forward(idx)
    assert torch.equal(a, b)


def test_feature_block_step_with_dropout_equals_oracle(dev):
    spec, offs = make_spec("sqrtn", 2)
    K, hidden, rate = 16, (128, 64, 32), 0.25
    net = FeatDeepFMNet(spec, embed_size=K, hidden_units=hidden, use_bn=True, dropout_rate=rate, lr=1e-2, device=dev)
    assert net.block_l1
    oracle = FeatDeepFMOracle(export_fieldnet_weights(net), hidden, use_bn=True, lr=1e-2, dtype=torch.float64,
                              plain_cols=spec.plain_cols, fields=[(3, 3, spec.field_oov[0])], combiner="sqrtn")
    oracle.mlp.dropout_fn = dropout_fn(1, 1.0 - rate)
    users, items, sp, dense, labels = feat_batch(np.random.default_rng(4), spec, offs, 512)
    l_net = float(net.train_step(users, items, labels, sparse=sp, dense=dense))
    l_or = float(oracle.train_step(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(sp),
                                   torch.from_numpy(dense).double(), torch.from_numpy(labels)))
    assert abs(l_net - l_or) < 1e-5
    W1 = export_fieldnet_weights(net)
    for name, want in oracle.V.v.items():
        np.testing.assert_allclose(W1[name].numpy().reshape(want.shape), want.detach().numpy(), rtol=1e-4, atol=5e-5, err_msg=name)


def test_mask_law():
    m = keep_mask(7, 1, 4096, 128, 0.7)
    assert abs(float(m.mean()) - 0.7) < 0.01
    assert not torch.equal(m, keep_mask(8, 1, 4096, 128, 0.7)) and not torch.equal(m, keep_mask(7, 0, 4096, 128, 0.7))
