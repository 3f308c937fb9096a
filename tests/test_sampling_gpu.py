"""Device negative sampler (`lr_sample_negatives_i32`): bit-exact against the numpy
restatement of the same counter-based algorithm, the reference's acceptance rules
(sampling/negatives.py:17-31, 55-82) as properties, determinism and uniformity."""
import numpy as np
import pytest
import torch

from librecommender_amd import ops
from oracle import ops_np

pytestmark = pytest.mark.gpu


def csr(user_consumed, n_users, dev):
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    flat = []
    for u in range(n_users):
        c = sorted(set(user_consumed.get(u, [])))
        flat.extend(c)
        ptr[u + 1] = len(flat)
    return torch.from_numpy(ptr).to(dev), torch.tensor(flat, dtype=torch.int32, device=dev)


@pytest.mark.parametrize("num_neg,n_items", [(1, 50), (3, 50), (4, 7), (2, 100000)])
def test_random_rules_bit_exact(dev, num_neg, n_items):
    rng = np.random.default_rng(num_neg)
    pos = rng.integers(0, n_items, 500).astype(np.int32)
    got = ops.sample_negatives(torch.from_numpy(pos).to(dev), num_neg, n_items, seed=1234).cpu().numpy()
    ref = ops_np.sample_negatives_counter(None, pos, num_neg, n_items, None, seed=1234)
    np.testing.assert_array_equal(got, ref)
    neg = got.reshape(-1, num_neg)
    assert (neg != pos[:, None]).all()                        # negatives.py:24-31
    assert ((neg >= 0) & (neg < n_items)).all()
    if num_neg < n_items - 1:
        assert all(len(set(r)) == num_neg for r in neg.tolist())   # no repeats per positive
    again = ops.sample_negatives(torch.from_numpy(pos).to(dev), num_neg, n_items, seed=1234).cpu().numpy()
    np.testing.assert_array_equal(again, got)                 # pure function of (seed, position)
    other = ops.sample_negatives(torch.from_numpy(pos).to(dev), num_neg, n_items, seed=1235).cpu().numpy()
    assert (other != got).any()


def test_unconsumed_rules_bit_exact(dev):
    rng = np.random.default_rng(7)
    n_users, n_items, n, num_neg = 40, 120, 600, 3
    consumed = {u: rng.choice(n_items, size=rng.integers(1, 60), replace=False).tolist() for u in range(n_users)}
    consumed[3] = list(range(n_items))                         # everything consumed -> rule relaxed after 10 tries
    users = rng.integers(0, n_users, n).astype(np.int32)
    pos = np.array([consumed[u][0] for u in users], dtype=np.int32)
    ptr, idx = csr(consumed, n_users, dev)
    got = ops.sample_negatives(torch.from_numpy(pos).to(dev), num_neg, n_items, seed=99,
                               users=torch.from_numpy(users).to(dev), consumed_ptr=ptr, consumed_idx=idx).cpu().numpy()
    ref = ops_np.sample_negatives_counter(users, pos, num_neg, n_items, consumed, seed=99)
    np.testing.assert_array_equal(got, ref)
    neg = got.reshape(-1, num_neg)
    assert (neg != pos[:, None]).all()
    hits = sum(int(x in set(consumed[u])) for u, row in zip(users.tolist(), neg.tolist()) if u != 3 for x in row)
    # P(10 consecutive consumed draws) <= 0.5^10 per negative: essentially never
    assert hits <= 3


def test_uniformity(dev):
    n_items, n = 1000, 200_000
    pos = torch.zeros(n, dtype=torch.int32, device=dev)
    neg = ops.sample_negatives(pos, 1, n_items, seed=5).cpu().numpy()
    counts = np.bincount(neg, minlength=n_items)
    assert counts[0] == 0
    exp = n / (n_items - 1)
    chi2 = ((counts[1:] - exp) ** 2 / exp).sum()
    assert chi2 < 1200, chi2                                   # 999 dof: mean 999, sd 45
