"""Full-catalog scoring + fused top-k vs the oracle restatement of
recommendation/recommend.py:57-78 + recommendation/ranking.py:10-56 (`-m gpu`).

Bar: ids identical to the oracle wherever neighbouring scores are separated by more than the
fp32 tolerance (numpy's sgemm and the MFMA fma-chain round differently); scores within
rtol 1e-5 / atol 1e-5 of an fp64 shadow; consumed items never returned when filtering applies.
"""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from oracle import ops_np

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(autouse=True, params=["f32_chain", "split_bf16", "filter"])
def topk_arith(request, monkeypatch):
    """Every case of this file runs under both arithmetics of the score contraction (lr_score_topk_f32: the exact f32 fma chain;
    lr_score_topk_sb_f32: six bf16 MFMA products per f32 product, f32 accumulation) and under the filtered form
    (lr_score_topk_filter_f32, taken at these small catalogues too) against the same fp64 bar."""
    monkeypatch.setattr(ops, "TOPK_ARITH", request.param)
    monkeypatch.setattr(ops, "TOPK_FILTER_FORCE", True)
    return request.param


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def consumed_csr(user_consumed, user_ids, n_rec, n_items, dev, filter_consumed=True):
    ptr, idx, flag = [0], [], []
    for u in user_ids:
        c = sorted(set(user_consumed.get(u, [])))
        flag.append(1 if ops_np.can_filter(user_consumed.get(u, []), n_rec, n_items, filter_consumed) else 0)
        idx.extend(c)
        ptr.append(len(idx))
    return (t(np.asarray(ptr, np.int64), dev), t(np.asarray(idx if idx else [0], np.int32), dev),
            t(np.asarray(flag, np.uint8), dev))


def check_topk(U, I, user_rows, ids, scores, k, user_consumed=None, n_items=None, filter_consumed=True):
    n_items = I.shape[0] if n_items is None else n_items
    P = U[user_rows].astype(np.float64) @ I.astype(np.float64).T
    for r, u in enumerate(user_rows):
        p = P[r].copy()
        banned = np.zeros(n_items, bool)
        if user_consumed is not None and ops_np.can_filter(user_consumed.get(u, []), k, n_items, filter_consumed):
            banned[np.asarray(sorted(set(user_consumed[u])), dtype=np.int64)] = True
        p[banned] = -np.inf
        order = np.lexsort((np.arange(n_items), -p))[:k]
        got = ids[r]
        assert len(set(got.tolist())) == k and got.min() >= 0 and got.max() < n_items
        assert not banned[got].any(), "a consumed item was recommended"
        np.testing.assert_allclose(scores[r], P[r][got], rtol=1e-5, atol=1e-5)
        assert np.all(np.diff(scores[r]) <= 0), "scores must be sorted descending"
        kth = p[order[-1]]
        assert np.all(P[r][got] >= kth - TOL), "returned an item below the k-th best score"
        must = np.where(p > kth + TOL)[0]
        assert set(must.tolist()) <= set(got.tolist()), "missed an item clearly inside the top-k"
        # positions whose neighbours are separated by > TOL must match exactly
        ps = p[order]
        sep = np.ones(k, bool)
        sep[1:] &= (ps[:-1] - ps[1:]) > TOL
        sep[:-1] &= (ps[:-1] - ps[1:]) > TOL
        if k < n_items - banned.sum():
            nxt = np.partition(p, -(k + 1))[-(k + 1)]
            sep[-1] &= (ps[-1] - nxt) > TOL
        np.testing.assert_array_equal(got[sep], order[sep])


def test_reference_known_answer_vectors(dev):
    """tests/test_rank_reco.py:7-87 of the reference, through the HIP path: the score matrix of
    that test is reproduced as U @ I^T with U = preds rows and I = identity."""
    preds = np.array([[-0.1, -0.01, 0, 0.1, 0.01], [1, -2, 4, 5, 6]], np.float32)
    U = np.zeros((3, 8), np.float32)
    U[1, :5], U[2, :5] = preds[0], preds[1]
    I = np.zeros((5, 8), np.float32)
    I[np.arange(5), np.arange(5)] = 1
    consumed = {1: [3, 4], 2: [4]}
    users = [1, 2]
    for n_rec, want in ((2, [[2, 1], [3, 2]]), (4, [[3, 4, 2, 1], [3, 2, 0, 1]])):
        ptr, cidx, flag = consumed_csr(consumed, users, n_rec, 5, dev)
        s, ids = ops.score_topk(t(U[users], dev), t(I, dev), n_rec, ptr, cidx, flag)
        if n_rec == 4:  # "can't filter consumed" branch (ranking.py:38) applies to user 1 only
            assert flag.cpu().tolist() == [0, 1]
        o_ids, _ = ops_np.rank_recommendations(users, preds, n_rec, 5, consumed)
        np.testing.assert_array_equal(o_ids, np.asarray(want))  # oracle == reference KAT
        np.testing.assert_array_equal(ids.cpu().numpy(), np.asarray(want))
    with pytest.raises(ValueError, match="exceeds num of items"):
        ops.score_topk(t(U[users], dev), t(I, dev), 12)


@pytest.mark.parametrize("B,N,D,k", [
    (1, 100, 16, 10),        # single user, tiny catalog, one stage range
    (7, 3231, 16, 10),       # movielens-sized (BASELINE config 1)
    (64, 5000, 32, 50),
    (65, 20_000, 128, 100),  # WU=4 path, several ranges
    (300, 70_001, 64, 7),    # ragged last tile
    (33, 9000, 20, 5),       # D not a multiple of 8 (zero-padded lanes)
    (5, 4000, 18, 3),        # D % 4 != 0 -> host pads
    (2, 3000, 256, 20),
    (40, 2500, 8, 2000),     # default_recs-sized k (bases/embed_base.py:153-161)
])
def test_score_topk_random(dev, B, N, D, k):
    rng = np.random.default_rng(B * 1000 + D)
    U = rng.standard_normal((B, D)).astype(np.float32)
    I = rng.standard_normal((N, D)).astype(np.float32)
    s, ids = ops.score_topk(t(U, dev), t(I, dev), k)
    check_topk(U, I, list(range(B)), ids.cpu().numpy(), s.cpu().numpy(), k)


def test_score_topk_with_consumed_filter(dev):
    rng = np.random.default_rng(0)
    B, N, D, k = 96, 12_345, 64, 20
    U = rng.standard_normal((B, D)).astype(np.float32)
    I = rng.standard_normal((N, D)).astype(np.float32)
    P = U @ I.T
    consumed = {}
    for u in range(B):
        if u % 5 == 4:
            continue  # users without history
        top = np.argsort(-P[u])[:30]  # consume exactly the items that would be recommended
        consumed[u] = [int(x) for x in rng.permutation(np.concatenate([top, rng.integers(0, N, 20)]))]
    consumed[0] = list(range(N - k + 1))  # too many consumed -> filter silently skipped (ranking.py:38)
    users = list(range(B))
    ptr, cidx, flag = consumed_csr(consumed, users, k, N, dev)
    assert flag[0].item() == 0 and flag[1].item() == 1 and flag[4].item() == 0
    s, ids = ops.score_topk(t(U, dev), t(I, dev), k, ptr, cidx, flag)
    check_topk(U, I, users, ids.cpu().numpy(), s.cpu().numpy(), k, consumed, N)
    # filter_consumed=False: flags all zero -> same as no consumed lists
    ptr, cidx, flag = consumed_csr(consumed, users, k, N, dev, filter_consumed=False)
    s2, ids2 = ops.score_topk(t(U, dev), t(I, dev), k, ptr, cidx, flag)
    check_topk(U, I, users, ids2.cpu().numpy(), s2.cpu().numpy(), k)


def test_score_topk_ties_are_deterministic(dev):
    """All-equal scores: order is (score desc, id asc) -> the k smallest ids, every run."""
    U = np.ones((3, 16), np.float32)
    I = np.ones((1000, 16), np.float32)
    for _ in range(2):
        s, ids = ops.score_topk(t(U, dev), t(I, dev), 10)
        np.testing.assert_array_equal(ids.cpu().numpy(), np.tile(np.arange(10), (3, 1)))
        np.testing.assert_array_equal(s.cpu().numpy(), np.full((3, 10), 16, np.float32))


def test_score_topk_sharded_merge_equals_single(dev):
    """Item-sharded scoring: per-shard top-k + lr_topk_merge == unsharded top-k."""
    rng = np.random.default_rng(5)
    B, N, D, k, S = 50, 40_000, 32, 25, 4
    U = rng.standard_normal((B, D)).astype(np.float32)
    I = rng.standard_normal((N, D)).astype(np.float32)
    Ud, Id = t(U, dev), t(I, dev)
    s_full, i_full = ops.score_topk(Ud, Id, k)
    per = N // S
    ss, ii = [], []
    for sh in range(S):
        s, i = ops.score_topk(Ud, Id[sh * per:(sh + 1) * per].contiguous(), k, item_base=sh * per)
        ss.append(s)
        ii.append(i)
    s_m, i_m = ops.topk_merge(torch.stack(ss), torch.stack(ii))
    assert torch.equal(i_m, i_full) and torch.equal(s_m, s_full)


def test_score_topk_large_catalog_properties(dev):
    """Size-independent properties at a catalog that spans many item ranges (1M x 128)."""
    g = torch.Generator(device=dev).manual_seed(42)
    B, N, D, k = 256, 1_000_000, 128, 100
    U = torch.randn((B, D), device=dev, generator=g)
    I = torch.randn((N, D), device=dev, generator=g)
    s, ids = ops.score_topk(U, I, k)
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    rec = (U[:, None, :] * I[ids]).sum(-1)  # recompute the returned scores
    torch.testing.assert_close(rec, s, rtol=1e-5, atol=1e-4)
    full = U[:8] @ I.T  # exact check on a few users against torch's own GEMM + topk
    ts, ti = torch.topk(full, k, dim=1)
    torch.testing.assert_close(ts, s[:8], rtol=1e-5, atol=1e-4)
    assert (ti == ids[:8]).float().mean().item() > 0.98  # near-ties may swap


def test_score_topk_long_consumed_lists(dev):
    """Thousands of consumed ids per user: per item range the in-register fast path (<= 4 ids)
    does not apply and the kernel falls back to the binary search over the narrowed range."""
    rng = np.random.default_rng(11)
    B, N, D, k = 40, 20_000, 32, 50
    U = rng.standard_normal((B, D)).astype(np.float32)
    I = rng.standard_normal((N, D)).astype(np.float32)
    P = U @ I.T
    consumed = {u: [int(x) for x in np.argsort(-P[u])[: 3000 + 37 * u]] for u in range(B)}
    users = list(range(B))
    ptr, cidx, flag = consumed_csr(consumed, users, k, N, dev)
    assert int(flag.sum()) == B
    s, ids = ops.score_topk(t(U, dev), t(I, dev), k, ptr, cidx, flag)
    check_topk(U, I, users, ids.cpu().numpy(), s.cpu().numpy(), k, consumed, N)


def test_lockstep_give_up_path_changes_nothing(dev):
    """Loose lockstep between the workgroups of an item range (csrc/score_topk.hip): a workgroup that never publishes its
    progress word — `lr_score_topk_test_mute`: as if it were not resident — makes its partners run into the BOUNDED wait
    at a window edge and drop the lockstep for good.  Ids and scores must be those of the undisturbed launch."""
    from librecommender_amd import _lib

    B, N, D, k = 1024, 4_000_000, 128, 50           # 2 GB of items: item ranges far beyond what an L2 holds (lockstep on)
    g = torch.Generator(device=dev).manual_seed(7)
    U = torch.randn((B, D), device=dev, generator=g)
    I = torch.randn((N, D), device=dev, generator=g)
    lib = _lib.load()
    s0, i0 = ops.score_topk(U, I, k)
    try:
        for ut in (0, 3):
            lib.lr_score_topk_test_mute(ut)
            s1, i1 = ops.score_topk(U, I, k)
            assert torch.equal(i0, i1) and torch.equal(s0, s1), ut
    finally:
        lib.lr_score_topk_test_mute(-1)
    s2, i2 = ops.score_topk(U, I, k)
    assert torch.equal(i0, i2) and torch.equal(s0, s2)


def test_split_bf16_scores_are_as_close_to_fp64_as_the_f32_chain(dev):
    """The two arithmetics against fp64 on the same (user, item) pairs: operands with a wide dynamic range (the planes of a
    split must carry mantissa bits 9-24 of every element), all compiled reduction widths.  Bar: the split form's worst and rms
    error within 1.5x the chain's (+ one ulp of the score scale)."""
    for D in (16, 32, 64, 128, 20, 100):
        g = torch.Generator(device=dev).manual_seed(D)
        B, N, k = 128, 50_000, 64
        U = torch.randn((B, D), device=dev, generator=g) * torch.exp(torch.randn((B, D), device=dev, generator=g))
        I = torch.randn((N, D), device=dev, generator=g) * torch.exp(torch.randn((N, D), device=dev, generator=g))
        err = {}
        for arith in ("f32_chain", "split_bf16"):
            s, ids = ops.score_topk(U, I, k, arith=arith)
            ref = (U.double()[:, None, :] * I.double()[ids]).sum(-1)
            scale = (U.double().abs()[:, None, :] * I.double().abs()[ids]).sum(-1)       # sum |u_d i_d|: the rounding scale
            e = ((s.double() - ref) / scale).abs()
            err[arith] = (float(e.max()), float(e.pow(2).mean().sqrt()))
        ulp = 2.0 ** -24
        assert err["split_bf16"][0] <= 1.5 * err["f32_chain"][0] + ulp, (D, err)
        assert err["split_bf16"][1] <= 1.5 * err["f32_chain"][1] + ulp / 4, (D, err)


def test_split_bf16_wide_reduction_runs_the_chain(dev):
    """Reduction widths above 128 are not compiled in the split form: the sb entry point runs the f32 chain (same bits)."""
    g = torch.Generator(device=dev).manual_seed(3)
    U, I = torch.randn((9, 200), device=dev, generator=g), torch.randn((3000, 200), device=dev, generator=g)
    a, b = ops.score_topk(U, I, 10, arith="f32_chain"), ops.score_topk(U, I, 10, arith="split_bf16")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# ---- the filtered form: a cheap pass that must never lose a winner ------------------------------------------------------------
def _exact(U, I, k, **kw):
    return ops.score_topk(U, I, k, arith="split_bf16", **kw)


def _filtered(U, I, k, **kw):
    failed = torch.full((U.shape[0],), 7, dtype=torch.uint8, device=U.device)
    s, i = ops.score_topk(U, I, k, arith="filter", failed_out=failed, **kw)
    assert int(failed.max()) <= 1
    return s, i, failed


@pytest.mark.parametrize("D", [64, 128, 100, 40])
def test_filter_certifies_separated_scores_and_matches_the_exact_ranking(dev, D):
    """Random catalogue: (nearly) every user is certified by the bound, the ranking is the exact kernel's (positions may swap
    only where two f32 scores are within rounding), the scores are f32 dot products of the returned pairs."""
    g = torch.Generator(device=dev).manual_seed(D)
    B, N, k = 300, 150_000, 100
    U, I = torch.randn((B, D), device=dev, generator=g), torch.randn((N, D), device=dev, generator=g)
    s, i, failed = _filtered(U, I, k)
    assert float(failed.float().mean()) < 0.05, "the filter should certify a random catalogue"
    se, ie = _exact(U, I, k)
    assert (i == ie).float().mean().item() > 0.995
    assert bool((torch.sort(i, 1).values == torch.sort(ie, 1).values).float().mean() > 0.999)
    torch.testing.assert_close(s, se, rtol=2e-6, atol=2e-5)
    check_topk(U.cpu().numpy(), I.cpu().numpy(), list(range(16)), i[:16].cpu().numpy(), s[:16].cpu().numpy(), k)


def test_filter_falls_back_where_the_bound_proves_nothing(dev):
    """(a) all scores equal: the k-th exact score cannot beat the k'-th bound — every user goes to the exact kernel (ids 0..k-1);
    (b) items of enormous norm orthogonal to every user: the bound is per item, so they only take a candidate slot each and
    everybody is still certified;  (c) an item holding inf: no bound, nobody is certified;  (d) only SOME users have near-ties:
    only they fall back.  In every case the result is the exact kernel's."""
    D, N, k = 128, 5000, 10
    U, I = torch.ones((3, D), device=dev), torch.ones((N, D), device=dev)
    s, i, failed = _filtered(U, I, k)
    assert int(failed.sum()) == 3
    np.testing.assert_array_equal(i.cpu().numpy(), np.tile(np.arange(k), (3, 1)))
    np.testing.assert_array_equal(s.cpu().numpy(), np.full((3, k), D, np.float32))

    g = torch.Generator(device=dev).manual_seed(1)
    B, N, k = 200, 60_000, 20
    U, I = torch.randn((B, D), device=dev, generator=g), torch.randn((N, D), device=dev, generator=g)
    U[:, 0] = 0
    I2 = I.clone()
    for r, big in ((777, 1e6), (5, 3e4), (59_999, 1e3)):
        I2[r] = 0
        I2[r, 0] = big                                # |i| = big, score 0 with every user: its bound outranks everything
    s, i, failed = _filtered(U, I2, k)
    assert int(failed.sum()) == 0
    se, ie = _exact(U, I2, k)
    assert torch.equal(torch.sort(i, 1).values, torch.sort(ie, 1).values)
    torch.testing.assert_close(s, se, rtol=2e-6, atol=2e-5)

    I3 = I.clone()
    I3[31, 5] = float("inf")
    s, i, failed = _filtered(U, I3, k)
    assert int(failed.sum()) == B
    se, ie = _exact(U, I3, k)
    assert torch.equal(i, ie) and torch.equal(s, se)

    U5 = U.clone()
    U5[3, 7] = float("inf")                           # a user holding inf: never certified (its row is the exact kernel's, whatever that is)
    U5[4, 9] = float("nan")
    s, i, failed = _filtered(U5, I, k)
    assert int(failed[3]) == 1 and int(failed[4]) == 1 and int(failed.sum()) <= 4
    se, ie = _exact(U5, I, k)
    assert torch.equal(i[3:5], ie[3:5]) and torch.equal(s[3:5].nan_to_num(), se[3:5].nan_to_num())

    U4 = U.clone()
    U4[::7] = 0                                       # these users score 0 everywhere: ties, no proof; the others are certified
    s, i, failed = _filtered(U4, I, k)
    f = failed.bool()
    assert bool(f[::7].all()) and float(f.float().mean()) < 0.2
    se, ie = _exact(U4, I, k)
    assert torch.equal(i[f], ie[f]) and torch.equal(s[f], se[f])
    np.testing.assert_array_equal(i[0].cpu().numpy(), np.arange(k))
    assert (i[~f] == ie[~f]).float().mean().item() > 0.995


def test_filter_never_loses_a_winner_hidden_by_bf16_rounding(dev):
    """Adversarial catalogue: 3,000 decoys whose bf16 images round UP (approximate score 1.0078 D, exact 1.0040 D) against k true
    winners whose images round DOWN on balance (approximate 1.0039 D, exact 1.0076 D): the approximate ranking fills its k' > k
    candidates with decoys only.  The bound cannot certify that, so the users go to the exact pass — the returned set is the
    fp64 top k."""
    D, k, N, B, n_dec = 64, 16, 40_000, 64, 3000
    g = torch.Generator(device=dev).manual_seed(9)
    U = torch.ones((B, D), device=dev) * (1 + 0.001 * torch.arange(B, device=dev)[:, None])
    I = torch.randn((N, D), device=dev, generator=g) * 0.01
    I[:n_dec] = (1 + 2.0 ** -8 + 2.0 ** -12) * (1 - 1e-5 * torch.rand((n_dec, 1), device=dev, generator=g))
    win = torch.arange(n_dec, n_dec + k, device=dev)
    I[win, : D // 2] = 1 + 2.0 ** -8 - 2.0 ** -12                            # -> 1.0 in bf16
    I[win, D // 2:] = 1 + 2.0 ** -7 + 2.0 ** -8 - 2.0 ** -12                 # -> 1 + 2^-7
    I[win] *= (1 - 1e-5 * torch.arange(k, device=dev)[:, None])            # (distinct scores; every element stays below its rounding midpoint)
    P = U.double() @ I.double().T
    ref = torch.topk(P, k, dim=1).indices
    assert torch.equal(torch.sort(ref, 1).values, win[None, :].expand(B, k))
    Pb = U.bfloat16().double() @ I.bfloat16().double().T
    assert int(torch.topk(Pb, k, dim=1).indices.max()) < n_dec, "the case must fool a bf16 ranking"
    s, i, failed = _filtered(U, I, k)
    assert torch.equal(torch.sort(i, 1).values, torch.sort(ref, 1).values)
    assert int(failed.sum()) == B


def test_filter_shapes_outside_its_range_run_the_exact_kernel(dev):
    g = torch.Generator(device=dev).manual_seed(3)
    for D, k in ((16, 10), (128, 101), (128, 300), (200, 10)):
        U, I = torch.randn((9, D), device=dev, generator=g), torch.randn((3000, D), device=dev, generator=g)
        s, i, failed = _filtered(U, I, k)
        se, ie = _exact(U, I, k)
        assert int(failed.sum()) == 0 and torch.equal(i, ie) and torch.equal(s, se)


def test_tied_scores_at_the_threshold_keep_the_id_order(dev):
    """Users whose scores tie massively (an all-zero embedding; a catalogue of duplicated rows) beside ordinary ones: ties are
    ordered by ascending id in every list and across the lists of the item ranges, whichever range's threshold is published
    first — the fast reject of a sub-tile only drops tied scores whose ids lie above the threshold's."""
    g = torch.Generator(device=dev).manual_seed(21)
    B, N, D, k = 130, 400_000, 64, 20
    U = torch.randn((B, D), device=dev, generator=g)
    U[::13] = 0
    I = torch.randn((N, D), device=dev, generator=g)
    s, i = ops.score_topk(U, I, k)
    assert torch.equal(i[::13], torch.arange(k, device=dev)[None, :].expand(len(range(0, B, 13)), k))
    assert float(s[::13].abs().max()) == 0.0
    check_topk(U.cpu().numpy(), I.cpu().numpy(), [1, 2, 3, 27], i[[1, 2, 3, 27]].cpu().numpy(), s[[1, 2, 3, 27]].cpu().numpy(), k)
    # duplicated rows: every row appears 4 times (ids r, r + N/4, ...): the winners are the FIRST copies, then the second ...
    Q = N // 4
    I2 = I[:Q].repeat(4, 1).contiguous()
    s2, i2 = ops.score_topk(U[1:9], I2, k)
    P = U[1:9].double() @ I2[:Q].double().T
    top = torch.topk(P, k // 4, dim=1).indices                       # the 5 best distinct rows, each with its 4 copies
    want = (top[:, :, None] + Q * torch.arange(4, device=dev)[None, None, :]).reshape(8, k)
    assert torch.equal(i2, want)


@pytest.mark.parametrize("D", [128, 96, 36])
@pytest.mark.parametrize("kind", ["student_t", "scaled_1e-12", "scaled_1e12", "lognormal_rows", "sparse", "low_rank", "integers"])
def test_filter_on_hostile_distributions_equals_the_exact_kernel(dev, D, kind):
    """The filtered form against the exact split-bf16 kernel on element distributions chosen against a bf16 first pass: heavy
    tails, tiny and huge scales, row norms over orders of magnitude, mostly-zero rows, a rank-4 catalogue (scores cluster),
    small integers (exact ties in bulk).  Whatever is certified or re-run, the ranking must be the exact kernel's: equal id sets
    per user wherever the exact k-th and (k+1)-th scores are separated, equal scores to f32 rounding."""
    import zlib

    g = torch.Generator(device=dev).manual_seed(zlib.crc32(kind.encode()) % 1000 + D)
    B, N, k = 150, 120_000, 30
    U = torch.randn((B, D), device=dev, generator=g)
    I = torch.randn((N, D), device=dev, generator=g)
    if kind == "student_t":
        I = I / torch.sqrt(torch.randn((N, D), device=dev, generator=g).pow(2) + 0.05)
        U = U / torch.sqrt(torch.randn((B, D), device=dev, generator=g).pow(2) + 0.05)
    elif kind == "scaled_1e-12":
        U, I = U * 1e-12, I * 1e-12
    elif kind == "scaled_1e12":
        U, I = U * 1e12, I * 1e12
    elif kind == "lognormal_rows":
        I = I * torch.exp(2.0 * torch.randn((N, 1), device=dev, generator=g))
    elif kind == "sparse":
        I = I * (torch.rand((N, D), device=dev, generator=g) < 0.05)
    elif kind == "low_rank":
        I = torch.randn((N, 4), device=dev, generator=g) @ torch.randn((4, D), device=dev, generator=g)
    elif kind == "integers":
        U, I = torch.round(U * 2), torch.round(I)
    U, I = U.contiguous(), I.contiguous()
    s, i, failed = _filtered(U, I, k)
    se, ie = ops.score_topk(U, I, k + 1, arith="split_bf16")
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    scale = float(se.abs().max())
    torch.testing.assert_close(s, se[:, :k], rtol=1e-5, atol=1e-6 * scale)
    sep = (se[:, k - 1] - se[:, k]) > 1e-5 * se[:, :k].abs().max(dim=1).values          # users whose k-th place is not a near-tie
    assert bool(sep.any()) if kind == "integers" else float(sep.float().mean()) > 0.9, kind
    assert torch.equal(torch.sort(i[sep], 1).values, torch.sort(ie[sep, :k], 1).values)
    # among exact ties the order is by ascending id: positions whose neighbours differ in score must agree exactly
    strict = torch.ones_like(i, dtype=torch.bool)
    d = (se[:, :k - 1] - se[:, 1:k]) > 1e-5 * scale
    strict[:, 1:] &= d
    strict[:, :-1] &= d
    strict &= sep[:, None]
    assert torch.equal(i[strict], ie[:, :k][strict])
    if kind == "integers":                    # bulk exact ties: equal scores, ids ascending inside every run of equal scores
        assert torch.equal(s, se[:, :k])
        assert torch.equal(i[sep], ie[sep, :k])


@pytest.mark.parametrize("B,N,D,k", [
    (3, 400_000, 64, 100),        # 768 item ranges x 2 lists: ten groups of 163 lists
    (1, 1_200_000, 128, 100),     # + the threshold pre-pass through both merge levels
    (40, 1_100_000, 128, 10),
    (130, 600_000, 64, 100),      # WU = 4 (one list per range), 384 ranges
    (2, 500_000, 32, 700),        # gl = 23 lists per block: as many groups as one block takes (the cap of two levels)
    (5, 300_000, 16, 1500),       # gl = 10
])
def test_small_batches_run_many_item_ranges_through_two_merge_levels(dev, B, N, D, k):
    """Plans for one or two user tiles take far more item ranges than one merge block holds lists (a workgroup on every CU):
    the lists are merged in groups and the groups with one another.  Result against torch's own GEMM + topk in f32 (ids where
    neighbouring scores are separated, scores everywhere)."""
    g = torch.Generator(device=dev).manual_seed(B * 7 + k)
    U = torch.randn((B, D), device=dev, generator=g)
    I = torch.randn((N, D), device=dev, generator=g)
    cons = torch.sort(torch.randint(0, N, (B, 20), device=dev, generator=g, dtype=torch.int32), dim=1).values
    full = (U.double() @ I.double().T)
    best = torch.topk(full, 5, dim=1).indices.to(torch.int32)
    cons[:, :5] = best                                   # the five best items of every user are consumed
    cons = torch.sort(cons, dim=1).values
    ptr = torch.arange(B + 1, device=dev, dtype=torch.int64) * 20
    flag = torch.ones(B, dtype=torch.uint8, device=dev)
    s, i = ops.score_topk(U, I, k, ptr, cons.reshape(-1).contiguous(), flag)
    full.scatter_(1, cons.long(), float("-inf"))
    rs, ri = torch.topk(full, k + 1, dim=1)
    assert not bool((i[:, :, None] == cons.long()[:, None, :]).any())
    torch.testing.assert_close(s.double(), rs[:, :k], rtol=1e-5, atol=1e-4)
    tol = 2e-4
    gp = torch.cat([torch.full_like(rs[:, :1], float("inf")), rs[:, :-2] - rs[:, 1:-1]], dim=1)
    gn = rs[:, :-1] - rs[:, 1:]
    sep = (gp > tol) & (gn > tol)
    assert float(sep.float().mean()) > 0.5
    assert torch.equal(i[sep], ri[:, :k][sep])


def test_random_shapes_fuzz(dev, topk_arith):
    """Seeded fuzz over batch sizes around the tile boundaries (1 .. 300), catalogue sizes around the stage boundaries, widths
    4 .. 160 (multiples of 4 and not), k from 1 to a third of the catalogue, with and without consumed lists — the plans, the
    one- and two-level merges, the filter's candidate count and its fallbacks — against the fp64 ranking (check_topk)."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        B = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 257, 300]))
        N = int(rng.choice([37, 64, 65, 257, 1000, 4095, 4096, 4097, 12_345, 40_000]))
        D = int(rng.choice([4, 6, 16, 20, 32, 33, 48, 64, 66, 100, 128, 160]))
        k = int(min(rng.choice([1, 2, 10, 64, 100, 101, 300]), max(1, N // 3)))
        U = rng.standard_normal((B, D)).astype(np.float32)
        I = rng.standard_normal((N, D)).astype(np.float32)
        if case % 5 == 4:
            I[rng.integers(0, N, N // 10)] = I[0]            # a tenth of the rows are copies of row 0: exact ties in bulk
        consumed = None
        args = ()
        users = list(range(B))
        if case % 2 == 0:
            consumed = {u: [int(x) for x in rng.integers(0, N, int(rng.integers(0, min(60, N // 2))))] for u in users if u % 3}
            args = consumed_csr(consumed, users, k, N, dev)
        s, ids = ops.score_topk(t(U, dev), t(I, dev), k, *args)
        sel = users if B <= 8 else [0, B // 2, B - 1]
        try:
            check_topk(U, I, sel, ids[sel].cpu().numpy(), s[sel].cpu().numpy(), k, consumed, N)
        except AssertionError as e:
            raise AssertionError(f"case {case}: B={B} N={N} D={D} k={k} consumed={consumed is not None} arith={topk_arith}: {e}") from e
