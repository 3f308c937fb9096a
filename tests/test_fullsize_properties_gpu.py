"""BASELINE cfg 2 at FULL size (12,000,202 rows x 64, B=16,384 x 202 fields, Zipf(1.05) ids): the
oracle cannot run here in seconds, so the HIP path is checked through size-independent properties
— permutation / sortedness of the segment build, exact gather, the pairwise identity, a checksum of
checksums for the fused backward, zero-gradient idempotence and run-to-run bit identity."""
import pytest
import torch

from bench import CFG, make_batches
from librecommender_amd import ops
from librecommender_amd.layers import FieldTables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(dev):
    cfg = dict(CFG)
    Fs, K = cfg["n_sparse_fields"], cfg["embed_size"]
    t = FieldTables(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), K, dev)
    users, items, sparse, _ = make_batches(cfg, 1, 42)[0]
    idx = t.global_idx(torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(sparse).to(dev))
    return cfg, t, idx


def test_segments_permutation_and_order(world):
    cfg, t, idx = world
    seg = t.segments(idx)
    n, ns = idx.numel(), seg.count()
    pos, rows, start = seg.pos[:n].long(), seg.rows[:ns].long(), seg.start[: ns + 1].long()
    assert torch.equal(torch.sort(pos).values, torch.arange(n, device=pos.device))        # a permutation
    assert bool((rows[1:] > rows[:-1]).all())                                              # distinct, ascending
    assert int(start[0]) == 0 and int(start[-1]) == n and bool((start[1:] > start[:-1]).all())
    flat = idx.reshape(-1).long()
    run_of = torch.repeat_interleave(torch.arange(ns, device=pos.device), start[1:] - start[:-1])
    assert torch.equal(flat[pos], rows[run_of])                                            # every position sits in its row's run
    inside = pos[1:] > pos[:-1]
    same_run = run_of[1:] == run_of[:-1]
    assert bool(inside[same_run].all())                                                    # ascending positions inside a run


def test_forward_exact_gather_and_pairwise_identity(world):
    cfg, t, idx = world
    e, pair, fsum, lin = ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
    assert torch.equal(e, t.embed[idx.long()])                                             # bit-exact rows
    assert torch.equal(lin, t.lin[idx.long()].squeeze(-1))
    s64 = e.double().sum(1)
    ref = 0.5 * (s64 * s64 - (e.double() ** 2).sum(1))
    # 202 fp32 additions of |e| <= 0.01: absolute error <= 202 * 0.01 * 2^-24 * few ~ 1e-7
    torch.testing.assert_close(fsum.double(), s64, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(pair.double(), ref, rtol=1e-5, atol=2e-8)
    # homogeneity: pair(2 * table) == 4 * pair(table), exactly (power-of-two scaling commutes with fp32 rounding)
    t2 = t.embed * 2.0
    pair2 = ops.fm_embed_fwd(t2, idx)[1]
    assert torch.equal(pair2, pair * 4.0)


def test_backward_checksum_idempotence_determinism(world):
    cfg, t, idx = world
    B, F = idx.shape
    K = t.embed.shape[1]
    dev = idx.device
    g = torch.Generator(device=dev).manual_seed(7)
    seg = t.segments(idx)
    e, pair, fsum, lin = ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
    gdeep = torch.randn((B, F, K), device=dev, generator=g) * 0.01
    gpair = torch.randn((B, K), device=dev, generator=g) * 0.01
    glin = torch.randn((B, F), device=dev, generator=g) * 0.01
    # checksum of checksums: sum_r g_r == sum_p gdeep_p + (F - 1) * sum_b gpair_b * fsum_b
    cache = ops.embed_gather(t.embed, seg.rows[: seg.count()].contiguous())
    slots = torch.empty(B * F, dtype=torch.int32, device=dev)
    start = seg.start[: seg.count() + 1].long()
    run_of = torch.repeat_interleave(torch.arange(seg.count(), device=dev, dtype=torch.int32), start[1:] - start[:-1])
    slots[seg.pos[: B * F].long()] = run_of
    grows, glin_rows = ops.fm_embed_bwd_rows(cache, gdeep, gpair, fsum, B, F, seg, glin=glin)
    want = gdeep.double().sum((0, 1)) + (F - 1) * (gpair.double() * fsum.double()).sum(0)
    torch.testing.assert_close(grows[: seg.count()].double().sum(0), want, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(glin_rows[: seg.count()].double().sum(), glin.double().sum(), rtol=1e-6, atol=1e-6)
    # zero gradients + zero moments: the fused update is the identity, bit for bit, on every row
    before = t.embed.clone()
    z3, z2, z1 = torch.zeros_like(gdeep), torch.zeros_like(gpair), torch.zeros_like(glin)
    ops.fm_embed_bwd_adam(t.embed, t.m, t.v, z3, z2, fsum, B, F, seg, ops.adam_hp(1e-3, 1), lin=t.lin,
                          lin_m=t.lin_m, lin_v=t.lin_v, glin=z1)
    assert torch.equal(t.embed, before) and float(t.m.abs().max()) == 0.0
    # run-to-run bit identity of the real update; untouched rows stay untouched
    def step():
        w, m, v = before.clone(), torch.zeros_like(before), torch.zeros_like(before)
        l, lm, lv = t.lin.clone(), torch.zeros_like(t.lin), torch.zeros_like(t.lin)
        ops.fm_embed_bwd_adam(w, m, v, gdeep, gpair, fsum, B, F, seg, ops.adam_hp(1e-3, 1), lin=l, lin_m=lm,
                              lin_v=lv, glin=glin)
        return w, l
    w1, l1 = step()
    w2, l2 = step()
    assert torch.equal(w1, w2) and torch.equal(l1, l2)
    touched = torch.zeros(before.shape[0], dtype=torch.bool, device=dev)
    touched[seg.rows[: seg.count()].long()] = True
    assert torch.equal(w1[~touched], before[~touched])
    assert bool((w1[touched] != before[touched]).any(dim=1).float().mean() > 0.99)
