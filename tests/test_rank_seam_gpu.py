"""`recommendation.rank_recommendations` — the reference's scoring seam for callers that hold a [B, n_items] block of
predictions (`libreco/recommendation/ranking.py:10-56`) — against the reference's own known-answer vectors
(`tests/test_rank_reco.py:7-87`) and the oracle restatement on random blocks (`-m gpu`)."""
import numpy as np
import pytest

from librecommender_amd.recommendation import rank_recommendations
from oracle import ops_np

pytestmark = pytest.mark.gpu


def test_reference_known_answers(dev):
    user_ids, n_items = [1, 2], 5
    preds = np.array([-0.1, -0.01, 0, 0.1, 0.01, 1, -2, 4, 5, 6])
    consumed = {1: [3, 4], 2: [4]}
    with pytest.raises(ValueError):
        rank_recommendations("ranking", user_ids, preds, 12, n_items, consumed)
    rec = rank_recommendations("ranking", user_ids, preds, 2, n_items, consumed)
    assert rec.shape == (2, 2) and rec.tolist() == [[2, 1], [3, 2]]
    rec = rank_recommendations("ranking", user_ids, preds, 4, n_items, consumed)          # can't-filter branch
    assert rec.tolist() == [[3, 4, 2, 1], [3, 2, 0, 1]]
    _, scores = rank_recommendations("ranking", user_ids, preds, 2, n_items, consumed, return_scores=True)
    assert scores.shape == (2, 2) and (np.diff(scores, axis=1) <= 0).all() and ((scores > 0) & (scores < 1)).all()
    rec = rank_recommendations("ranking", user_ids, preds.reshape(2, 5), 2, n_items, consumed)
    assert rec.tolist() == [[2, 1], [3, 2]]
    _, raw = rank_recommendations("rating", user_ids, preds, 2, n_items, consumed, return_scores=True)
    np.testing.assert_allclose(raw, [[0.0, -0.01], [5.0, 4.0]], rtol=0, atol=1e-7)          # no expit for rating tasks


def test_random_blocks_match_the_oracle_and_random_rec_respects_the_filter(dev):
    rng = np.random.default_rng(3)
    B, N, k = 37, 501, 10
    preds = rng.standard_normal((B, N)).astype(np.float32)                                 # distinct scores: no tie order
    users = list(range(100, 100 + B))
    consumed = {u: rng.choice(N, size=int(rng.integers(0, 60)), replace=False).tolist() for u in users[::2]}
    consumed[users[1]] = list(range(N - 5))                                                 # too long to filter: kept whole
    ids = rank_recommendations("ranking", users, preds, k, N, consumed)
    ref, _ = ops_np.rank_recommendations(users, preds, k, N, consumed)
    np.testing.assert_array_equal(ids, ref)
    ids = rank_recommendations("ranking", users, preds, k, N, consumed, filter_consumed=False)
    np.testing.assert_array_equal(ids, np.argsort(-preds, axis=1)[:, :k])
    pick = rank_recommendations("ranking", users, preds, k, N, consumed, random_rec=True)
    assert pick.shape == (B, k)
    for r, u in enumerate(users):
        assert len(set(pick[r].tolist())) == k
        if u in consumed and k + len(consumed[u]) <= N:
            assert not set(pick[r].tolist()) & set(consumed[u])
        s = preds[r][pick[r]]
        assert (np.diff(s) <= 0).all()                                                      # ordered by score
