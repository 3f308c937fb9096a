"""The whole three-layer tail as ONE persistent launch (`lr_mlp_tail3_f32`, csrc/deepfm_tail.hip: mlp_tail3_kernel — the layers of
`dense_nn` after the first Dense, layers/dense.py:33-49; the output layer, algorithms/deepfm.py:158-172 / din.py:190-192; the
sigmoid cross-entropy, tfops/loss.py:14-16; their backward) against the chain of `lr_mlp_*` launches it replaces: the same
per-tile arithmetic and the same fixed-order reductions, so every output, every parameter gradient, the batch statistics and
the moving averages must agree BIT FOR BIT — with and without BatchNorm, with the DeepFM and the plain (DIN) output layer, with
dropout, for batches below one tile, a partial last tile, one tile per CU and several tiles per workgroup.  (Against torch
autograd the tail is checked in tests/test_deepfm_fused_gpu.py::test_hip_tail_matches_torch_autograd, which now runs this form.)"""
import numpy as np
import pytest
import torch

from librecommender_amd.layers.tail import DeepFMTail
from librecommender_amd.nets import DeepFMNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [37, 64, 1000, 16384, 20001])
@pytest.mark.parametrize("use_bn,plain,drop", [(True, False, 0.0), (False, False, 0.0), (True, True, 0.0), (True, False, 0.25)])
def test_fused_tail_equals_the_chain_bit_for_bit(dev, B, use_bn, plain, drop):
    Fs, K = 7, 64
    net = DeepFMNet(50, 60, Fs * 10, Fs, embed_size=K, hidden_units=(128, 64, 32), use_bn=use_bn, dropout_rate=drop, device=dev,
                    sparse_offsets=np.arange(Fs) * 10)
    g = torch.Generator(device=dev).manual_seed(B + 17 * use_bn + 3 * plain)
    with torch.no_grad():
        for name, p in net.P.params.items():
            if name.endswith("gamma"):
                p.add_(torch.randn(p.shape, device=dev, generator=g) * 0.2)
            elif name.endswith("beta") or name.endswith("bias"):
                p.add_(torch.randn(p.shape, device=dev, generator=g) * 0.1)
    F_, K_ = (0, 0) if plain else (Fs + 2, K)
    out = net.out
    # (plain form: the output layer takes the deep term only — the first 32 rows of the DeepFM net's output kernel serve as its weights)
    z1 = torch.randn((B, 128), device=dev, generator=g) * 0.7
    pair = None if plain else torch.randn((B, K), device=dev, generator=g)
    lin_out = None if plain else torch.randn((B, Fs + 2), device=dev, generator=g)
    labels = (torch.rand(B, device=dev, generator=g) > 0.5).float()
    mm0 = [None if bn is None else (bn.moving_mean.clone(), bn.moving_var.clone()) for bn in net.mlp.bns]

    def run(fused):
        for bn, st in zip(net.mlp.bns, mm0):
            if bn is not None:
                bn.moving_mean.copy_(st[0])
                bn.moving_var.copy_(st[1])
        net.P.zero_grad()
        tail = DeepFMTail(net.P, net.mlp, net.linear if not plain else None, out, F_, K_, dev)
        assert tail.fused, "the fused tail is not compiled for (128, 64, 32)"
        tail.fused = fused
        loss, gl, gz1, sgz1 = tail.run(z1, pair, lin_out, labels, drop_seed=1234)
        torch.cuda.synchronize()
        assert int(tail.sync_words[1]) == 0, "the grid barrier timed out"
        stats = [t.clone() for t in tail.mean[:2] + tail.inv[:2]] if use_bn else []
        mm = [None if bn is None else (bn.moving_mean.clone(), bn.moving_var.clone()) for bn in net.mlp.bns]
        return float(loss), gl.clone(), gz1.clone(), sgz1.clone(), net.P.grad.clone(), stats, mm

    a, b = run(False), run(True)
    assert a[0] == b[0]
    for x, y in zip(a[1:5], b[1:5]):
        assert torch.equal(x, y)
    for x, y in zip(a[5], b[5]):
        assert torch.equal(x, y)
    for x, y in zip(a[6], b[6]):
        if x is not None:
            assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
    c = run(True)                  # run-to-run identical
    assert b[0] == c[0] and torch.equal(b[2], c[2]) and torch.equal(b[4], c[4])


def test_fused_tail_is_what_the_models_run(dev):
    net = DeepFMNet(50, 60, 70, 7, embed_size=64, hidden_units=(128, 64, 32), device=dev, sparse_offsets=np.arange(7) * 10)
    idx = torch.randint(0, 50, (256, 9), device=dev, dtype=torch.int32)
    net.train_step(idx, (torch.rand(256, device=dev) > 0.5).float())
    assert net._tail is not None and net._tail.fused and net._tail._jobs_mode == "fused"
    other = DeepFMNet(50, 60, 70, 7, embed_size=64, hidden_units=(128, 64), device=dev, sparse_offsets=np.arange(7) * 10)
    other.train_step(idx, (torch.rand(256, device=dev) > 0.5).float())
    assert not other._tail.fused            # other depths / widths keep the chain of launches


# ---- the grid barrier when the launch's workgroups cannot all be resident ---------------------------------------------------
def _tail_case(dev, B=16384, seed=5):
    Fs, K = 7, 64
    net = DeepFMNet(50, 60, Fs * 10, Fs, embed_size=K, hidden_units=(128, 64, 32), use_bn=True, device=dev,
                    sparse_offsets=np.arange(Fs) * 10)
    g = torch.Generator(device=dev).manual_seed(seed)
    z1 = torch.randn((B, 128), device=dev, generator=g) * 0.7
    pair = torch.randn((B, K), device=dev, generator=g)
    lin_out = torch.randn((B, Fs + 2), device=dev, generator=g)
    labels = (torch.rand(B, device=dev, generator=g) > 0.5).float()
    mm0 = [(bn.moving_mean.clone(), bn.moving_var.clone()) for bn in net.mlp.bns if bn is not None]

    def make(fused=True):
        for bn, st in zip([b for b in net.mlp.bns if b is not None], mm0):
            bn.moving_mean.copy_(st[0])
            bn.moving_var.copy_(st[1])
        net.P.zero_grad()
        tail = DeepFMTail(net.P, net.mlp, net.linear, net.out, Fs + 2, K, dev)
        tail.fused = fused
        return tail

    return net, make, (z1, pair, lin_out, labels)


def test_grid_is_sized_from_the_device_not_a_constant(dev):
    from librecommender_amd import _lib

    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    r = lib.lr_mlp_tail3_resident_blocks()
    assert 1 <= r <= cus and r == cus, (r, cus)         # one workgroup per CU of THIS device (LDS: > 80 KB per workgroup)


def test_barrier_with_cus_taken_away_is_identical_or_raises(dev):
    """A long-running kernel on a second stream owns half of the CUs (one 160 KB-LDS workgroup each) while the one-launch tail
    starts: only part of its workgroups are resident.  (a) the spinner ends within the poll bound: the late workgroups arrive,
    every barrier completes, the results are the undisturbed run's bits.  (b) the spinner outlasts the bound: sync[1] and the
    sticky word are set, the loss is NaN, `check` raises, and the next step on the same buffers is refused too (NaN, raise) —
    never finite numbers computed from statistics of a part of the batch."""
    from librecommender_amd import _lib, ops
    from librecommender_amd.layers.tail import TailBarrierError, check_all

    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    net, make, inp = _tail_case(dev)
    ref_tail = make()
    ref = ref_tail.run(*inp, drop_seed=7)
    torch.cuda.synchronize()
    ref = (float(ref[0]), ref[1].clone(), ref[2].clone(), ref[3].clone(), net.P.grad.clone())
    ref_tail.check()
    side = torch.cuda.Stream(device=dev)

    # (a) half the CUs are held for 0.2 s: the barrier waits, then completes
    tail = make()
    rc = lib.lr_probe_occupy(cus // 2, 160 * 1024, 200_000, side.cuda_stream)
    assert rc == 0
    import time

    time.sleep(0.02)                                   # the spinner is resident before the tail is launched
    out = tail.run(*inp, drop_seed=7)
    torch.cuda.synchronize()
    assert int(tail.sync_words[1]) == 0 and int(tail.sync_words[tail.STICKY]) == 0
    assert float(out[0]) == ref[0]
    for x, y in zip(out[1:], ref[1:4]):
        assert torch.equal(x, y)
    assert torch.equal(net.P.grad, ref[4])
    tail.check()

    # (b) the spinner outlasts a short poll bound: loud failure
    tail = make()
    tail.spin_limit = 20_000                           # ~ 20 - 40 ms of polling
    rc = lib.lr_probe_occupy(cus // 2, 160 * 1024, 1_500_000, side.cuda_stream)
    assert rc == 0
    time.sleep(0.02)
    out = tail.run(*inp, drop_seed=7)
    torch.cuda.synchronize()
    assert int(tail.sync_words[1]) == 1 and int(tail.sync_words[tail.STICKY]) == 1
    assert np.isnan(float(out[0])), "a step whose barrier gave up must not return a finite loss"
    with pytest.raises(TailBarrierError):
        tail.check()
    with pytest.raises(TailBarrierError):
        check_all()
    # the next launch on the same buffers (free device now) is refused at the kernel's entry
    out = tail.run(*inp, drop_seed=8)
    torch.cuda.synchronize()
    assert np.isnan(float(out[0])) and int(tail.sync_words[tail.STICKY]) == 1
    with pytest.raises(TailBarrierError):
        tail.check()
    del tail
    import gc

    gc.collect()
    check_all()                                        # only live tails are polled


def test_chain_form_under_more_than_one_rank(dev, monkeypatch):
    """Under a process group with more than one rank the tail runs as the chain of launches (no grid barrier beside RCCL's
    kernels) unless LIBRECO_TAIL_MULTI_RANK=fused."""
    from librecommender_amd.layers import tail as tail_mod

    net, make, inp = _tail_case(dev, B=1000)
    monkeypatch.setattr(tail_mod, "_multi_rank", lambda: True)
    t = make()
    t.run(*inp, drop_seed=3)
    assert t._jobs_mode == "chain" and not t._ran_fused
    t.fused_multi_rank = True
    t.run(*inp, drop_seed=3)
    assert t._jobs_mode == "fused"
