"""FM / DeepFM graphs on the HIP path vs the PyTorch-CPU oracle restatement of the reference's
TF graphs, from identical weights (`-m gpu`).  Parity contract: forward scores,
per-step gradients (through the first update) and — with dense_adam=True — the exact TF1
dense-Adam trajectory over several steps.  fp32 tolerances: logits 1e-5 abs (vs fp64 shadow),
weights after a step 1e-5 rel of lr-scaled updates."""
import numpy as np
import pytest
import torch

from librecommender_amd.nets import DeepFMNet, FMNet
from oracle.models_torch import DeepFMOracle, FMOracle, export_fieldnet_weights

pytestmark = pytest.mark.gpu


def make_batch(rng, B, n_users, n_items, vocab, Fs):
    users = rng.integers(0, n_users, B)
    items = rng.integers(0, n_items, B)
    off = np.arange(Fs) * (vocab + 1)
    sparse = rng.integers(0, vocab, (B, Fs)) + off  # global offsets as in feature/sparse.py:165-211
    labels = rng.integers(0, 2, B).astype(np.float32)
    return users, items, sparse, labels


def to_dev(net, users, items, sparse, labels, dev):
    u = torch.from_numpy(users).to(dev)
    i = torch.from_numpy(items).to(dev)
    s = torch.from_numpy(sparse).to(dev)
    return net.tables.global_idx(u, i, s), torch.from_numpy(labels).to(dev)


def cpu_batch(users, items, sparse, labels):
    return (torch.from_numpy(users).long(), torch.from_numpy(items).long(),
            torch.from_numpy(sparse).long(), torch.from_numpy(labels))


@pytest.mark.parametrize("K,Fs,hidden", [(16, 5, (32, 16)), (64, 12, (128, 64, 32))])
def test_deepfm_forward_and_tf_dense_adam_trajectory(dev, K, Fs, hidden):
    rng = np.random.default_rng(K)
    nu, ni, vocab, B = 50, 70, 11, 96
    net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=hidden, lr=1e-2,
                    device=dev, dense_adam=True)
    W = export_fieldnet_weights(net)
    o32 = DeepFMOracle(W, hidden, lr=1e-2, dtype=torch.float32)
    o64 = DeepFMOracle(W, hidden, lr=1e-2, dtype=torch.float64)
    batches = [make_batch(rng, B, nu, ni, vocab, Fs) for _ in range(3)]
    idx, _ = to_dev(net, *batches[0], dev)
    lg = net.forward(idx).cpu().numpy()
    np.testing.assert_allclose(lg, o64.forward(*cpu_batch(*batches[0])[:3]).detach().numpy(), rtol=1e-5, atol=1e-5)
    for b in batches:
        idx, lab = to_dev(net, *b, dev)
        l_hip = float(net.train_step(idx, lab))
        l_64 = float(o64.train_step(*cpu_batch(*b)))
        o32.train_step(*cpu_batch(*b))
        assert abs(l_hip - l_64) < 1e-5
    W2 = export_fieldnet_weights(net)
    for name, ref in o64.V.v.items():
        got = W2[name].numpy().reshape(ref.shape)
        # three Adam steps of lr=1e-2 move weights by ~3e-2; fp32 vs fp64 differ ~1e-6 of that
        np.testing.assert_allclose(got, ref.detach().numpy(), rtol=1e-4, atol=3e-6, err_msg=name)
    # fp32 oracle vs fp64 oracle drift gives the scale of acceptable error: HIP must be in-family
    for name, ref in o64.V.v.items():
        d_hip = np.abs(W2[name].numpy().reshape(ref.shape) - ref.detach().numpy()).max()
        d_o32 = np.abs(o32.V.v[name].detach().numpy() - ref.detach().numpy()).max()
        assert d_hip <= 10 * d_o32 + 3e-6, name
    for k in ("mlp/bn_in/moving_mean", "mlp/bn1/moving_var"):
        np.testing.assert_allclose(W2[k].numpy(), o64.V.buffers[k].numpy(), rtol=1e-4, atol=1e-6)


def test_deepfm_lazy_adam_first_step_equals_tf_on_touched_rows(dev):
    rng = np.random.default_rng(1)
    nu, ni, vocab, Fs, K, B = 200, 300, 50, 8, 32, 128
    net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=(64, 32), lr=1e-3, device=dev)
    W = export_fieldnet_weights(net)
    o64 = DeepFMOracle(W, (64, 32), lr=1e-3, dtype=torch.float64)
    b = make_batch(rng, B, nu, ni, vocab, Fs)
    idx, lab = to_dev(net, *b, dev)
    net.train_step(idx, lab)
    o64.train_step(*cpu_batch(*b))
    W2 = export_fieldnet_weights(net)
    touched = {"user_embeds_var": np.unique(b[0]), "item_embeds_var": np.unique(b[1]),
               "sparse_embeds_var": np.unique(b[2])}
    for name, rows in touched.items():
        np.testing.assert_allclose(W2[name].numpy()[rows], o64.V.v[name].detach().numpy()[rows],
                                   rtol=1e-4, atol=2e-6, err_msg=name)
        rest = np.setdiff1d(np.arange(W[name].shape[0]), rows)
        np.testing.assert_array_equal(W2[name].numpy()[rest], W[name].numpy()[rest])  # untouched rows frozen
    for name in ("linear/kernel", "out/kernel", "mlp/mlp_layer1/kernel", "mlp/bn_in/gamma"):
        np.testing.assert_allclose(W2[name].numpy(), o64.V.v[name].detach().numpy(), rtol=1e-4, atol=2e-6)


def test_fm_forward_and_training(dev):
    rng = np.random.default_rng(2)
    nu, ni, vocab, Fs, K, B = 40, 60, 9, 4, 16, 64
    net = FMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, lr=1e-2, device=dev, dense_adam=True)
    W = export_fieldnet_weights(net)
    o64 = FMOracle(W, lr=1e-2, dtype=torch.float64)
    for _ in range(2):
        b = make_batch(rng, B, nu, ni, vocab, Fs)
        idx, lab = to_dev(net, *b, dev)
        l_hip = float(net.train_step(idx, lab))
        l_64 = float(o64.train_step(*cpu_batch(*b)))
        assert abs(l_hip - l_64) < 1e-5
    W2 = export_fieldnet_weights(net)
    for name, ref in o64.V.v.items():
        np.testing.assert_allclose(W2[name].numpy().reshape(ref.shape), ref.detach().numpy(),
                                   rtol=1e-4, atol=3e-6, err_msg=name)
    b = make_batch(rng, B, nu, ni, vocab, Fs)
    idx, _ = to_dev(net, *b, dev)
    np.testing.assert_allclose(net.forward(idx).cpu().numpy(),
                               o64.forward(*cpu_batch(*b)[:3]).detach().numpy(), rtol=1e-5, atol=1e-5)


def test_sharded_net_world1_equals_unsharded_on_hip(dev):
    """The row-cache / rows-mode kernels and the RCCL exchange code path (world_size 1) reproduce
    the unsharded fused step bit-for-bit on touched rows (same kernels, same summation order)."""
    import os
    import torch.distributed as dist
    from librecommender_amd.nets import ShardedDeepFMNet

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(4)
        nu, ni, vocab, Fs, K, B = 100, 150, 20, 6, 64, 256
        net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=(64, 32), lr=1e-2, device=dev)
        sh = ShardedDeepFMNet(net.tables.V, Fs, embed_size=K, hidden_units=(64, 32), lr=1e-2, device=dev)
        sh.tables.load_full(net.tables.embed, net.tables.lin)
        sh.P.flat.data.copy_(net.P.flat.data)
        for _ in range(3):
            b = make_batch(rng, B, nu, ni, vocab, Fs)
            idx, lab = to_dev(net, *b, dev)
            l1 = net.train_step(idx, lab)
            l2 = sh.train_step(idx, lab)
            torch.testing.assert_close(l1, l2, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(sh.tables.embed, net.tables.embed, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sh.tables.lin, net.tables.lin, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sh.P.flat, net.P.flat, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sh.forward(idx), net.forward(idx), rtol=1e-5, atol=1e-5)
    finally:
        dist.destroy_process_group()


def test_deepfm_bn_stats_from_segments_equals_welford_path(dev):
    """Opt-in input-BN statistics computed from the batch's runs (`lr_fm_field_stats_f32`) give the
    same training step as the statistics of the materialised block."""
    rng = np.random.default_rng(9)
    nu, ni, vocab, Fs, K, B = 60, 80, 13, 6, 32, 256
    offs = np.arange(Fs) * (vocab + 1)
    kw = dict(embed_size=K, hidden_units=(32, 16), lr=1e-2, device=dev)
    a = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, **kw)
    b = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, sparse_offsets=offs, bn_stats_from_segments=True, **kw)
    assert b._want_stats and not a._want_stats
    for _ in range(3):
        batch = make_batch(rng, B, nu, ni, vocab, Fs)
        ia, la = to_dev(a, *batch, dev)
        l1, l2 = a.train_step(ia, la), b.train_step(ia, la)
        torch.testing.assert_close(l1, l2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.tables.embed, b.tables.embed, rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(a.P.flat, b.P.flat, rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(a.mlp.bn_in.moving_var, b.mlp.bn_in.moving_var, rtol=1e-4, atol=1e-7)
