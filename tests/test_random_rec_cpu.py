"""`random_rec=True` without the [B, N] score matrix (`recommendation/ranking.py:65-73`): the streaming
Gumbel-top-k draw follows the same distribution as the materialised `torch.multinomial` draw (both: n_rec items
without replacement, weights softmax(score)^0.75 + 1e-8, consumed items excluded), is chunk-size independent in
law, returns distinct un-consumed ids ordered by score."""
import numpy as np
import torch

from librecommender_amd.recommendation.recommend import random_select_device, random_select_streaming


def test_streaming_random_rec_matches_the_materialised_draw_in_distribution():
    torch.manual_seed(0)
    N, D, n_rec, B = 23, 4, 3, 40_000
    I = torch.randn(N, D)
    u = torch.randn(1, D) * 1.5
    U = u.repeat(B, 1)
    consumed = torch.tensor([2, 7], dtype=torch.int32)
    ptr = torch.arange(B + 1, dtype=torch.int64) * 2
    cidx = consumed.repeat(B)
    flag = torch.ones(B, dtype=torch.uint8)
    got = random_select_streaming(U, I, ptr, cidx, flag, n_rec, max_elems=B * 7)      # 4 chunks of 7 items
    scores = U @ I.T
    banned = torch.zeros_like(scores, dtype=torch.bool)
    banned[:, consumed.long()] = True
    ref = random_select_device(scores, banned, n_rec)
    for picks in (got, ref):
        assert picks.shape == (B, n_rec)
        assert not bool(((picks == 2) | (picks == 7)).any())
        srt = torch.sort(picks, dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())                               # distinct
        sc = torch.gather(scores, 1, picks)
        assert bool((sc[:, :-1] >= sc[:, 1:]).all())                                 # ordered by score
    f_got = np.bincount(got.reshape(-1).numpy(), minlength=N) / B
    f_ref = np.bincount(ref.reshape(-1).numpy(), minlength=N) / B
    np.testing.assert_allclose(f_got, f_ref, atol=0.012)                             # inclusion probabilities
    # the FIRST pick in draw order is not observable (results are re-ordered by score); the joint law of pairs is:
    pair = lambda p: np.bincount((torch.sort(p, 1).values[:, 0] * N + torch.sort(p, 1).values[:, 1]).numpy(), minlength=N * N) / B  # noqa: E731
    np.testing.assert_allclose(pair(got), pair(ref), atol=0.012)


def test_streaming_random_rec_without_consumed_lists():
    torch.manual_seed(1)
    U, I = torch.randn(5, 8), torch.randn(100, 8)
    ptr = torch.zeros(6, dtype=torch.int64)
    picks = random_select_streaming(U, I, ptr, torch.zeros(1, dtype=torch.int32), torch.zeros(5, dtype=torch.uint8), 10, max_elems=5 * 16)
    assert picks.shape == (5, 10) and all(len(set(r.tolist())) == 10 for r in picks)
