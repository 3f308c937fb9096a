"""captured training steps must hold kernel nodes only — `GraphRunner.capture` walks the graph's nodes
(`lr_graph_foreign_nodes`: hipGraphGetNodes / hipGraphNodeGetType) and refuses a step with a memset / memcpy node; the nets
then launch that shape eagerly.  `lazy_join(model=...)` switches one model's runners, not a process-global flag."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_capture_refuses_memset_nodes_and_accepts_kernel_steps(dev):
    from librecommender_amd.nets.din_fused import GraphNotCapturable, GraphRunner

    r = GraphRunner(dev)
    x = torch.zeros(1 << 16, device=dev)
    y = torch.ones(1 << 16, device=dev)

    def kernels_only():
        torch.add(x, y, out=x)
        return x

    st = r.capture("ok", kernels_only)
    assert "graph" in st
    before = x.clone()
    r.replay("ok", lambda: None)
    torch.cuda.synchronize()
    assert torch.equal(x, before + 1)

    def with_memcpy():
        x.copy_(y)                      # a device-to-device copy of contiguous tensors is captured as a memcpy node
        torch.add(x, y, out=x)
        return x

    with pytest.raises(GraphNotCapturable, match="memset / memcpy"):
        r.capture("bad", with_memcpy)
    assert "bad" not in r.graphs


def test_fused_steps_hold_kernel_nodes_only(dev):
    """The DeepFM and DIN fused steps pass the check (they would fall back to eager launches with a warning otherwise)."""
    import warnings

    from librecommender_amd.nets import DeepFMNet
    from librecommender_amd.nets.feat_embedding import FeatSpec
    from librecommender_amd.nets.feat_nets import FeatDINNet

    g = torch.Generator().manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = DeepFMNet(500, 400, 6 * 50, 6, embed_size=64, hidden_units=(128, 64, 32), device=dev,
                        sparse_offsets=[50 * j for j in range(6)])
        net.enable_graph(True)
        B = 1024
        for _ in range(4):
            idx = torch.cat([torch.randint(0, 500, (B, 1), generator=g), 501 + torch.randint(0, 400, (B, 1), generator=g),
                             902 + torch.arange(6)[None] * 50 + torch.randint(0, 50, (B, 6), generator=g)], 1).to(torch.int32).to(dev)
            net.train_step(idx, torch.randint(0, 2, (B,), generator=g).float().to(dev))
        assert any("graph" in st for st in net._runner.graphs.values())
        din = FeatDINNet(FeatSpec(300, 200), embed_size=64, max_seq_len=10, device=dev)
        for _ in range(4):
            din.train_step(torch.randint(0, 300, (512,), generator=g), torch.randint(0, 200, (512,), generator=g),
                           torch.randint(0, 2, (512,), generator=g).float(), seqs=torch.randint(0, 200, (512, 10), generator=g),
                           seq_lens=torch.randint(1, 11, (512,), generator=g))
        assert any("graph" in st for st in din._fstep.runner.graphs.values())
    torch.cuda.synchronize()


def test_lazy_join_is_per_model(dev):
    from types import SimpleNamespace

    from librecommender_amd.nets.din_fused import GraphRunner, lazy_join

    a, b = GraphRunner(dev), GraphRunner(dev)
    ma, mb = SimpleNamespace(net=SimpleNamespace(_runner=a)), SimpleNamespace(net=SimpleNamespace(_runner=b))
    with lazy_join(True, model=ma):
        assert a.lazy is True and b.lazy is None and GraphRunner.LAZY is False
        with lazy_join(False, model=mb):
            assert b.lazy is False and a.lazy is True
    assert a.lazy is None and b.lazy is None
