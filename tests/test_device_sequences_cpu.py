"""Device-side sequences / DIN batches (row f1) as device-agnostic tensor code, run on CPU tensors:
`DeviceSequences` row for row against the host `SequenceBuilder` (itself bit-exact against the
reference fixtures), and the whole `DevicePointwiseLoader` for a sequence model with the numpy
restatement of the counter-based sampler injected for `lr_sample_negatives_i32`."""
import types

import numpy as np
import torch

from librecommender_amd.batch.device_loader import DevicePointwiseLoader, DeviceSequences, device_loader_supported
from librecommender_amd.batch.sequence import SequenceBuilder
from librecommender_amd.data import DatasetFeat, split_by_ratio_chrono
from oracle import ops_np
from oracle.make_golden import FEAT_KW, synthetic_frame


def _data():
    train, _ = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    return DatasetFeat.build_trainset(train, **FEAT_KW)


def test_sequences_match_host_builder():
    ts, info = _data()
    for L in (1, 3, 10):
        host = SequenceBuilder(info.user_consumed, info.n_items, L, "recent")
        devs = DeviceSequences(info.user_consumed, info.n_items, L, torch.device("cpu"))
        rng = np.random.default_rng(L)
        users = np.concatenate([ts.user_indices, rng.integers(0, info.n_users, 300)])
        items = np.concatenate([ts.item_indices, rng.integers(0, info.n_items, 300)])
        pos = host.positions(users, items)
        np.testing.assert_array_equal(devs.positions(torch.from_numpy(users), torch.from_numpy(items)).numpy(), pos)
        gen = torch.Generator().manual_seed(7)
        seqs, lens = devs.build(torch.from_numpy(users).int(), torch.from_numpy(items).int(), gen)
        seqs, lens = seqs.numpy(), lens.numpy()
        assert seqs.dtype == np.int32 and lens.dtype == np.int32 and seqs.shape == (len(users), L)
        import random
        random.seed(0)
        h_seqs, h_lens = host.training_seqs(users, items)
        known = pos >= 0                                     # the item is in the history: deterministic
        np.testing.assert_array_equal(seqs[known], h_seqs[known])
        np.testing.assert_array_equal(lens[known], h_lens[known])
        drawn = set()
        for j in np.flatnonzero(~known):                     # random position p: window hist[max(p-L,0):p]
            hist = info.user_consumed[int(users[j])]
            n = int(lens[j])
            window = seqs[j, :n].tolist() if not (n == 1 and seqs[j, 0] == info.n_items) else []
            assert (seqs[j, n:] == info.n_items).all() or n == L
            ok = [p for p in range(len(hist)) if hist[max(p - L, 0):p] == window and max(min(p, L), 1) == n]
            assert ok, (hist, window, n)
            drawn.add(ok[0] if len(ok) == 1 else -1)
        assert len(drawn) > 3                                # positions really vary


def _sampler(user_consumed):
    def fn(pos, num_neg, n_items, seed, users=None, consumed_ptr=None, consumed_idx=None):
        out = ops_np.sample_negatives_counter(users.numpy(), pos.numpy(), num_neg, n_items,
                                              user_consumed if consumed_ptr is not None else None, seed)
        return torch.from_numpy(out)
    return fn


def test_din_device_loader_batches():
    ts, info = _data()
    model = types.SimpleNamespace(model_name="DIN", data_info=info, device=torch.device("cpu"), task="ranking",
                                  loss_type="cross_entropy", sampler="unconsumed", num_neg=2, n_users=info.n_users,
                                  n_items=info.n_items, uses_features=True, uses_sequence=True, seq_mode="recent",
                                  max_seq_len=4, device_sampling=True)
    assert device_loader_supported(model, True)
    model.seq_mode = "random"
    assert not device_loader_supported(model, True)           # random windows stay on the host loader
    model.seq_mode = "recent"
    loader = DevicePointwiseLoader(model, ts, 32, shuffle=True, seed=5, sample_negatives=_sampler(info.user_consumed))
    host = SequenceBuilder(info.user_consumed, info.n_items, 4, "recent")
    seen, k = 0, 3
    for b in loader:
        users, items, labels = b.users.numpy(), b.items.numpy(), b.labels.numpy()
        np.testing.assert_array_equal(labels.reshape(-1, k), np.tile([1.0, 0.0, 0.0], (len(users) // k, 1)))
        it = items.reshape(-1, k)
        for u, row in zip(users.reshape(-1, k)[:, 0], it):
            assert row[0] in info.user_consumed[u] and row[1] not in info.user_consumed[u] and row[1] != row[2]
        np.testing.assert_array_equal(b.sparse_indices.numpy()[:, info.item_sparse_col.index], info.item_sparse_unique[items])
        seqs, lens = b.seqs.interacted_seq.numpy(), b.seqs.interacted_len.numpy()
        assert seqs.shape == (len(users), 4) and lens.min() >= 1 and lens.max() <= 4
        pos = host.positions(users, items)
        import random
        random.seed(1)
        h_seqs, h_lens = host.training_seqs(users, items)
        np.testing.assert_array_equal(seqs[pos >= 0], h_seqs[pos >= 0])          # positives: the real history window
        np.testing.assert_array_equal(lens[pos >= 0], h_lens[pos >= 0])
        assert (pos.reshape(-1, k)[:, 0] >= 0).all()
        seen += len(users) // k
    assert seen == len(ts)
    mk = lambda: DevicePointwiseLoader(model, ts, 32, shuffle=True, seed=5,                     # noqa: E731
                                       sample_negatives=_sampler(info.user_consumed))
    a, b = next(iter(mk())), next(iter(mk()))                                               # seeded -> reproducible
    torch.testing.assert_close(a.seqs.interacted_seq, b.seqs.interacted_seq, rtol=0, atol=0)
    torch.testing.assert_close(a.items, b.items, rtol=0, atol=0)
