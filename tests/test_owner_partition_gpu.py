"""Exchange plans off the field-wise sort: the run numbers `lr_segments_build_fields_runs` writes and the stable
owner-major order of `lr_owner_partition_i32`, against their restatement in torch integer arithmetic.  Bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fields(B, F, rows_per_field, seed):
    g = torch.Generator().manual_seed(seed)
    frs = torch.arange(F + 1, dtype=torch.int64) * rows_per_field
    # heavy repeats in some fields, nearly distinct ids in others
    span = torch.where(torch.arange(F) % 3 == 0, torch.tensor(7), torch.tensor(rows_per_field))
    loc = (torch.rand((B, F), generator=g) * span).long().clamp_(max=rows_per_field - 1)
    return (loc + frs[:-1]).to(torch.int32), frs.to(torch.int32)


@pytest.mark.parametrize("B,F", [(1, 3), (257, 5), (4096, 13), (16384, 4)])
def test_run_numbers_match_unique(B, F):
    from librecommender_amd import ops

    dev = torch.device("cuda")
    idx, frs = _fields(B, F, 1000, B + F)
    idx, frs = idx.to(dev), frs.to(dev)
    fb = ops.FieldSegmentBuilder(B, F, int(frs[-1]), dev, want_runs=True)
    seg = fb.build(ops.idx_transpose(idx), frs)
    rows, inv = torch.unique(idx.long().view(-1), sorted=True, return_inverse=True)
    n = int(seg.n_seg.item())
    assert n == rows.numel()
    assert torch.equal(seg.rows[:n].long(), rows)
    assert torch.equal(seg.runT.t().contiguous().view(-1).long(), inv)      # position (b, f) -> its row's rank


@pytest.mark.parametrize("W", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("n,cap", [(0, 16), (1, 1), (1023, 1024), (1025, 5000), (70001, 70001)])
def test_owner_partition_is_the_stable_order(W, n, cap):
    from librecommender_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(W * 1000 + n)
    rows = torch.sort(torch.randperm(4 * cap + 8, generator=g)[:cap])[0].to(torch.int32).to(dev)
    n_seg = torch.tensor([n], dtype=torch.int32, device=dev)
    part = ops.OwnerPartition(cap, W, dev)
    part.perm.fill_(-7)
    part.send_ids.fill_(-7)
    perm, send_ids, counts = part.run(rows, n_seg)
    r = rows[:n].long()
    order = torch.argsort(r % W, stable=True)                 # owner-major, ascending rows within an owner
    want_perm = torch.empty(n, dtype=torch.int64, device=dev)
    want_perm[order] = torch.arange(n, device=dev)
    assert torch.equal(perm[:n].long(), want_perm)
    assert torch.equal(send_ids[:n].long(), (r // W)[order])
    assert torch.equal(counts[:W], torch.bincount(r % W, minlength=W))
    assert int(counts[W]) == n
    assert bool((perm[n:] == -7).all()) and bool((send_ids[n:] == -7).all())     # nothing past n_seg is touched


def test_partition_refuses_too_many_owners():
    from librecommender_amd import _lib, ops

    dev = torch.device("cuda")
    with pytest.raises(Exception):
        ops.OwnerPartition(16, 65, dev).run(torch.arange(16, dtype=torch.int32, device=dev),
                                            torch.tensor([16], dtype=torch.int32, device=dev))
    assert _lib.load().lr_owner_partition_ws_bytes(-1, 4) == 0


@pytest.mark.parametrize("W,with_lin", [(1, True), (2, True), (5, False), (8, True)])
@pytest.mark.parametrize("K", [16, 64])
def test_peer_adam_equals_sorted_segment_adam(W, with_lin, K):
    """`lr_embed_peer_adam_f32` (rows grouped through the [V, W] table) against `lr_segments_build` +
    `lr_embed_scatter_adam_lin_f32` / `lr_embed_scatter_adam_f32` on the same lists: the same bits, and the table is
    all zero again afterwards."""
    from librecommender_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(W * 100 + K)
    V = 5000
    lists = [torch.sort(torch.randperm(V, generator=g)[: int(torch.randint(0, 3000, (1,), generator=g))])[0] for _ in range(W)]
    if W >= 5:
        lists[2] = lists[2][:0]                     # a peer that asks for nothing
    ids = torch.cat(lists).to(torch.int32).to(dev)
    counts = [int(l.numel()) for l in lists]
    n = ids.numel()
    grad = torch.randn((n, K), generator=g).to(dev)
    glin = torch.randn(n, generator=g).to(dev)

    def state():
        gg = torch.Generator().manual_seed(7)
        return [torch.randn((V, K), generator=gg).to(dev), torch.rand((V, K), generator=gg).to(dev) * 0.1,
                torch.rand((V, K), generator=gg).to(dev) * 0.01, torch.randn((V, 1), generator=gg).to(dev),
                torch.rand((V, 1), generator=gg).to(dev) * 0.1, torch.rand((V, 1), generator=gg).to(dev) * 0.01]

    hp = ops.adam_hp(1e-2, 3, eps=1e-8, tf_style=True)
    a, b = state(), state()
    tab = torch.zeros(V * W, dtype=torch.int32, device=dev) if W > 1 else None
    ops.embed_peer_adam(a[0], a[1], a[2], grad, ids, counts, hp, *( (a[3], a[4], a[5], glin) if with_lin else (None,) * 4),
                        peer_tab=tab)
    seg = ops.build_segments(ids, V)
    if with_lin:
        ops.embed_scatter_adam_lin(b[0], b[1], b[2], grad, b[3], b[4], b[5], glin, seg, hp)
    else:
        ops.embed_scatter_adam(b[0], b[1], b[2], grad, seg, hp)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    if tab is not None:
        assert int(tab.abs().sum()) == 0
