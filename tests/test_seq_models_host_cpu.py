"""Host-side sequence windows of the row-f4 models (no GPU): the dynamic-inference windows of YouTubeRetrieval
(`recommendation/preprocess.py:7-23,79-85`: last L entries of the given sequence or of the consumed list, unknown items
pruned, the OOV user has no history) and SIM (`preprocess.py:49-76` `build_dual_seq`: long / short split of an explicit
sequence), and the packed [long | short] layout SIM hands to the net."""
import numpy as np
import pandas as pd

from librecommender_amd.algorithms import SIM, YouTubeRetrieval
from librecommender_amd.data import DatasetPure


def info_of():
    df = pd.DataFrame({"user": [1, 1, 1, 1, 1, 1, 1, 2, 2, 3], "item": [10, 11, 12, 13, 14, 15, 16, 10, 12, 11],
                       "label": [1] * 10, "time": np.arange(10)})
    _, info = DatasetPure.build_trainset(df)
    return info


def test_youtube_retrieval_windows():
    info = info_of()
    m = YouTubeRetrieval("ranking", info, recent_num=3)
    N = info.n_items
    i = info.item2id
    u1 = info.user2id[1]
    # known user, no explicit sequence: the tail of the consumed list
    np.testing.assert_array_equal(m._window(u1, None, False), [[i[14], i[15], i[16]]])
    # a short history is left-aligned and padded with the OOV id
    np.testing.assert_array_equal(m._window(info.user2id[3], None, False), [[i[11], N, N]])
    # explicit raw-id sequence: last 3 entries, the unknown item keeps its slot as a pruned (pad) entry
    np.testing.assert_array_equal(m._window(u1, [10, 11, 999, 12], False), [[i[11], N, i[12]]])
    # inner ids, shorter than the window
    np.testing.assert_array_equal(m._window(u1, [2, 5], True), [[2, 5, N]])
    # the OOV user has no history
    np.testing.assert_array_equal(m._window(info.n_users, None, False), [[N, N, N]])
    assert m.recent_seqs.shape == (info.n_users + 1, 3) and m.hidden_units[-1] == m.embed_size


def test_sim_dual_windows_and_packing():
    info = info_of()
    m = SIM("ranking", info, long_max_len=4, short_max_len=2, search_topk=2)
    N, i = info.n_items, info.item2id
    u1 = info.user2id[1]                                       # 7 consumed items >= long + short
    seqs, lens = m._seq_for(u1, None)
    np.testing.assert_array_equal(seqs, [[i[11], i[12], i[13], i[14], i[15], i[16]]])       # [long(4) | short(2)]
    np.testing.assert_array_equal(lens, [[4, 2]])
    seqs, lens = m._seq_for(info.user2id[2], None)             # 2 items: all short, long is one pad
    np.testing.assert_array_equal(seqs, [[N, N, N, N, i[10], i[12]]])
    np.testing.assert_array_equal(lens, [[1, 2]])
    # explicit sequences (build_dual_seq): longer than both windows / between / shorter than the short window
    raw = [10, 11, 12, 13, 14, 15, 16]
    seqs, lens = m._seq_for(u1, raw)
    np.testing.assert_array_equal(seqs, [[i[11], i[12], i[13], i[14], i[15], i[16]]])
    np.testing.assert_array_equal(lens, [[4, 2]])
    seqs, lens = m._seq_for(u1, [10, 11, 12])
    np.testing.assert_array_equal(seqs, [[i[10], N, N, N, i[11], i[12]]])
    np.testing.assert_array_equal(lens, [[1, 2]])
    seqs, lens = m._seq_for(u1, [13])
    np.testing.assert_array_equal(seqs, [[N, N, N, N, i[13], N]])
    np.testing.assert_array_equal(lens, [[1, 1]])
    assert m.recent_seqs.shape == (info.n_users + 1, 6) and m.recent_seq_lens.shape == (info.n_users + 1, 2)
    assert m.max_seq_len == 6
