"""lib/liblibreco_host.so (hostsrc/host_loops.c) against the Python loops it accelerates: same
values, same final state of Python's `random` generator."""
import random

import numpy as np
import pytest

from librecommender_amd import _hostlib
from librecommender_amd.sampling.negatives import _negatives_from_unconsumed_py, negatives_from_unconsumed


@pytest.fixture(scope="module", autouse=True)
def lib():
    from librecommender_amd.csrc.build import build_host
    build_host(verbose=False)
    _hostlib._tried = False
    assert _hostlib.load() is not None
    import re
    from pathlib import Path
    header = (Path(__file__).resolve().parent.parent / "include" / "libreco_host.h").read_text()
    declared = sorted(set(re.findall(r"\b(lrh_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S))))
    assert declared == ["lrh_abi_version", "lrh_gather_rows_u32", "lrh_merge_pointwise_u32", "lrh_negatives_unconsumed", "lrh_pair_positions",
                        "lrh_randrange_stream", "lrh_seq_windows_i32"]
    for name in declared:                                  # everything include/libreco_host.h declares is exported
        assert hasattr(_hostlib.load(), name)
    assert _hostlib.load().lrh_abi_version() == _hostlib.ABI_VERSION


def test_randrange_stream_is_pythons_randrange():
    rng = np.random.default_rng(0)
    for trial in range(40):
        c = int(rng.integers(0, 2500))
        widths = rng.integers(1, int(rng.choice([2, 3, 5, 17, 100, 1000, 70000, 2 ** 31, 2 ** 32 - 1])) + 1, c)
        if trial % 4 == 0 and c:
            widths[rng.integers(0, c, max(1, c // 8))] = rng.choice([1, 2, 4, 8, 1024, 2 ** 20, 2 ** 31, 2 ** 32 - 1])
        random.seed(trial)
        for _ in range(trial * 37 % 700):            # start from arbitrary positions inside the state block
            random.random()
        state = random.getstate()
        want = [random.randrange(0, int(n)) for n in widths]
        after = random.getstate()
        random.setstate(state)
        got = _hostlib.randrange_stream(widths)
        assert got.tolist() == want and random.getstate() == after
    with pytest.raises(ValueError):
        _hostlib.randrange_stream([3, 0, 2])


@pytest.mark.parametrize("as_dict", [False, True])
def test_unconsumed_sampler_matches_python_loop(as_dict):
    rng = np.random.default_rng(1)
    for trial, (n_items, num_neg, hist) in enumerate([(50, 1, 5), (50, 3, 30), (12, 4, 11), (1000, 2, 40), (7, 3, 7)]):
        n_users = 30
        sets = [set(rng.integers(0, n_items, hist).tolist()) for _ in range(n_users)]
        consumed = {u: s for u, s in enumerate(sets)} if as_dict else sets
        users = rng.integers(0, n_users, 400)
        items = np.asarray([rng.choice(sorted(sets[u])) for u in users])
        random.seed(100 + trial)
        want = _negatives_from_unconsumed_py(consumed, users, items, n_items, num_neg)
        after = random.getstate()
        random.seed(100 + trial)
        got = negatives_from_unconsumed(consumed, users, items, n_items, num_neg)
        np.testing.assert_array_equal(got, want)
        assert random.getstate() == after


def test_python_fallback_without_library(monkeypatch):
    monkeypatch.setattr(_hostlib, "_lib", None)
    monkeypatch.setattr(_hostlib, "_tried", True)
    assert _hostlib.randrange_stream([3, 4]) is None
    random.seed(5)
    a = negatives_from_unconsumed([{1, 2}, {3}], [0, 1, 0], [1, 3, 2], 10, 2)
    random.seed(5)
    b = _negatives_from_unconsumed_py([{1, 2}, {3}], [0, 1, 0], [1, 3, 2], 10, 2)
    np.testing.assert_array_equal(a, b)


def test_merge_pointwise_equals_numpy_collation():
    """The one-pass C merge of a pointwise feature block == repeat / gather / concatenate / column permutation in numpy
    (collators.py:PointwiseCollator._feats), for interleaved and user-first column orders, int32 and float32."""
    from librecommender_amd.batch.collators import restore_column_order

    rng = np.random.default_rng(4)
    for dtype in (np.int32, np.float32):
        for u_cols, i_cols in (([0, 1, 2], [3, 4]), ([1, 4], [0, 2, 3]), ([3], [0, 1, 2, 4])):
            n_pos, k, n_items = 37, 3, 11
            batch = (rng.integers(0, 1000, (n_pos, 5)) if dtype == np.int32 else rng.standard_normal((n_pos, 5))).astype(dtype)
            item_rows = (rng.integers(0, 1000, (n_items + 1, len(i_cols))) if dtype == np.int32
                         else rng.standard_normal((n_items + 1, len(i_cols)))).astype(dtype)
            items = rng.integers(0, n_items + 1, n_pos * k)
            got = _hostlib.merge_pointwise(batch, item_rows, i_cols, items, k)
            want = restore_column_order(np.repeat(batch[:, u_cols], k, axis=0), item_rows[items], u_cols, i_cols)
            np.testing.assert_array_equal(got, want)
            assert got.dtype == dtype
    with pytest.raises(IndexError):
        _hostlib.merge_pointwise(batch, item_rows, i_cols, np.full(n_pos * k, n_items + 5), k)


def test_gather_rows_equals_numpy_indexing():
    rng = np.random.default_rng(8)
    for dtype in (np.int32, np.float32):
        base = (rng.standard_normal((500, 7)) * 100).astype(dtype)
        idx = rng.integers(0, 500, 123)
        got = _hostlib.gather_rows(base, idx)
        np.testing.assert_array_equal(got, base[idx])
        assert got.dtype == dtype and got.flags.c_contiguous
    np.testing.assert_array_equal(_hostlib.gather_rows(base.astype(np.float64), idx), base.astype(np.float64)[idx])   # falls back
    np.testing.assert_array_equal(_hostlib.gather_rows(base, idx.astype(np.int32)), base[idx])
    with pytest.raises(IndexError):
        _hostlib.gather_rows(base, np.array([0, 500]))


def test_seq_windows_equal_their_numpy_definition_and_the_builders_use_them():
    """`lrh_seq_windows_i32` == the padded-gather expression of `SequenceBuilder._windows`; the DIN / SIM training windows
    are the same with and without the library (ragged histories: empty, shorter and longer than the window)."""
    from librecommender_amd.batch.sequence import SequenceBuilder

    rng = np.random.default_rng(7)
    hist = rng.integers(0, 1 << 31, 5000).astype(np.int64)
    for width in (1, 3, 50, 128):
        n = 700
        count = rng.integers(0, width + 1, n)
        count[:5] = [0, width, 0, width, min(1, width)]
        start = rng.integers(0, len(hist) - width, n)
        got = _hostlib.seq_windows(hist, start, count, width, -7)
        t = np.arange(width)[None, :]
        valid = t < count[:, None]
        want = np.where(valid, hist[np.where(valid, start[:, None] + t, 0)], -7).astype(np.int32)
        assert got.dtype == np.int32 and np.array_equal(got, want)
    assert _hostlib.seq_windows(hist, np.zeros(0, np.int64), np.zeros(0, np.int64), 4, 0).shape == (0, 4)
    for bad_start, bad_count in (([len(hist) - 1], [2]), ([-1], [1]), ([0], [5]), ([0], [-1])):
        with pytest.raises(IndexError):
            _hostlib.seq_windows(hist, np.array(bad_start), np.array(bad_count), 4, 0)

    n_users, n_items = 60, 400
    consumed = {u: rng.integers(0, n_items, int(rng.choice([0, 1, 2, 7, 30, 90]))).tolist() for u in range(n_users)}
    consumed[0], consumed[1] = [], [5]
    users = rng.integers(0, n_users, 900)
    users = users[np.array([len(consumed[u]) > 0 for u in users])]        # a negative item needs a non-empty history
    items = np.array([consumed[u][rng.integers(0, len(consumed[u]))] if rng.random() < 0.6 else int(rng.integers(0, n_items))
                      for u in users])
    outs = []
    for use_lib in (True, False):
        saved = _hostlib._lib
        if not use_lib:
            _hostlib._lib = None
        try:
            random.seed(11)
            a = SequenceBuilder(consumed, n_items, 10).training_seqs(users, items)
            random.seed(11)
            b = SequenceBuilder(consumed, n_items, 1).training_dual_seqs(users, items, 20, 4)
        finally:
            _hostlib._lib = saved
        outs.append((*a, *b))
    for x, y in zip(*outs):
        assert x.dtype == y.dtype and np.array_equal(x, y)


def test_pair_positions_equal_the_searchsorted_definition():
    """`lrh_pair_positions` == `SequenceBuilder.positions` without the library: first occurrence of the item in the user's
    history, -1 for absent items, users with empty histories, the largest ids of the table."""
    from librecommender_amd.batch.sequence import SequenceBuilder

    rng = np.random.default_rng(5)
    n_users, n_items = 80, 300
    consumed = {u: rng.integers(0, n_items, int(rng.choice([0, 0, 1, 3, 40, 200]))).tolist() for u in range(n_users)}
    consumed[n_users - 1] = [n_items - 1, 0, n_items - 1]                   # extremes of the key space, with a repeat
    b = SequenceBuilder(consumed, n_items, 5)
    users = rng.integers(0, n_users, 4000)
    items = rng.integers(0, n_items, 4000)
    hit = rng.random(4000) < 0.5
    for q in np.flatnonzero(hit):
        h = consumed[int(users[q])]
        if h:
            items[q] = h[int(rng.integers(0, len(h)))]
    users[:2], items[:2] = n_users - 1, [n_items - 1, 0]
    with_lib = b.positions(users, items)
    saved, _hostlib._lib = _hostlib._lib, None
    try:
        without = b.positions(users, items)
    finally:
        _hostlib._lib = saved
    assert np.array_equal(with_lib, without) and with_lib[0] == 0 and with_lib[1] == 1
    want = np.array([consumed[int(u)].index(int(i)) if int(i) in consumed[int(u)] else -1 for u, i in zip(users, items)])
    assert np.array_equal(with_lib, want)                                   # the reference's `list.index`
    empty = SequenceBuilder({0: [], 1: []}, 10, 3)
    assert np.array_equal(empty.positions(np.array([0, 1]), np.array([3, 4])), [-1, -1])
