"""`SIM` (`libreco/algorithms/sim.py`): same constructor and errors; embedding rows and their Adam update on the HIP
kernels, the search / attention blocks over the (long, short) windows as device torch ops (`nets/seq_nets.py`)."""
from __future__ import annotations

import numpy as np

from ..bases import FeatBase
from ..bases.base import hip_device
from ..batch.sequence import get_recent_dual_seqs
from ..nets import FeatSpec
from ..nets.seq_nets import FeatSIMNet
from ..utils.validate import check_multi_sparse, dropout_config, hidden_units_config, reg_config


class SIM(FeatBase):
    uses_sequence = True
    seq_mode = "recent"

    def __init__(self, task, data_info=None, loss_type="cross_entropy", embed_size=16, n_epochs=20, lr=0.001,
                 lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random", num_neg=1, use_bn=True,
                 dropout_rate=None, hidden_units=(200, 80), alpha=1.0, beta=1.0, search_topk=10, long_max_len=100,
                 short_max_len=10, num_heads=2, multi_sparse_combiner="sqrtn", seed=42, lower_upper_bound=None,
                 tf_sess_config=None, device="cuda", dense_adam=False):
        super().__init__(task, data_info, lower_upper_bound)
        self.all_args = locals()
        self.loss_type, self.embed_size, self.n_epochs = loss_type, embed_size, n_epochs
        self.lr, self.lr_decay, self.epsilon, self.reg = lr, lr_decay, epsilon, reg_config(reg)
        self.batch_size, self.sampler, self.num_neg, self.use_bn = batch_size, sampler, num_neg, use_bn
        self.dropout_rate = dropout_config(dropout_rate)
        self.hidden_units = hidden_units_config(hidden_units)
        self.alpha, self.beta, self.search_topk = alpha, beta, search_topk
        self.long_max_len, self.short_max_len, self.num_heads = long_max_len, short_max_len, num_heads
        self.max_seq_len = long_max_len + short_max_len            # width of the packed [long | short] sequence
        self.seed = seed
        self.sparse = bool(data_info.sparse_col.name)
        self.dense = bool(data_info.dense_col.name)
        self.multi_sparse_combiner = check_multi_sparse(data_info, multi_sparse_combiner) if self.sparse else "normal"
        # `reg` is the reference's L2 regulariser on the embedding VARIABLES (deepfm.py:181-259): its gradient reaches every row of
        # every table each step, i.e. it exists only under TF1's dense update — asking for it selects `dense_adam` (row-wise
        # Adam on the touched rows would drop the term silently)
        self._device_arg, self.dense_adam = device, bool(dense_adam or self.reg)
        assert 0.0 <= alpha <= 1.0
        assert 0.0 <= beta <= 1.0
        assert short_max_len > 0
        assert long_max_len >= search_topk > 0
        if self.task == "ranking" and self.loss_type not in ("cross_entropy", "focal"):
            raise ValueError(f"unsupported `loss_type`: {self.loss_type}")
        lg, ln, sh, sn = get_recent_dual_seqs(self.n_users, self.user_consumed, self.n_items, long_max_len, short_max_len)
        self.cached_long_seqs, self.cached_long_lens, self.cached_short_seqs, self.cached_short_lens = lg, ln, sh, sn
        self.recent_seqs = np.concatenate([lg, sh], axis=1)        # packed like the training batches
        self.recent_seq_lens = np.stack([ln, sn], axis=1)

    def build_model(self):
        self.device = hip_device(self._device_arg)
        d = self.data_info
        self.net = FeatSIMNet(FeatSpec.from_data_info(d, self.multi_sparse_combiner), self.embed_size,
                              self.hidden_units, self.use_bn, self.dropout_rate, self.alpha, self.beta,
                              self.search_topk, self.long_max_len, self.short_max_len, self.num_heads,
                              d.item_sparse_unique, d.item_dense_unique, d.item_dense_col.index, self.lr,
                              self.epsilon, self.seed, self.device, self.dense_adam, self.reg)

    def _seq_args(self, b):
        return {"seqs": b.seqs.interacted_seq, "seq_lens": b.seqs.interacted_len}

    def _cached_seq(self, users):
        return self.recent_seqs[users], self.recent_seq_lens[users]

    def _seq_for(self, uid, seq):
        """`build_dual_seq` (recommendation/preprocess.py:49-76) for an explicit `seq`, the cached windows otherwise."""
        if seq is None or len(seq) == 0:
            return self.recent_seqs[[uid]], self.recent_seq_lens[[uid]]
        N, Lg, S = self.n_items, self.long_max_len, self.short_max_len
        ids = list(seq) if self._inner_seq else [self.data_info.item2id.get(i, N) for i in seq]
        long = np.full((1, Lg), N, dtype=np.int32)
        if len(ids) >= Lg + S:
            long_len = Lg
            long[0] = ids[len(ids) - Lg - S: len(ids) - S]
        elif len(ids) > S:
            long_len = len(ids) - S
            long[0, :long_len] = ids[:long_len]
        else:
            long_len = 1
        short = np.full((1, S), N, dtype=np.int32)
        short_len = min(S, len(ids))
        short[0, :short_len] = ids[-short_len:]
        return np.concatenate([long, short], axis=1), np.array([[long_len, short_len]], dtype=np.int32)

    _inner_seq = False

    def recommend_user(self, user, n_rec, user_feats=None, seq=None, cold_start="average", inner_id=False,
                       filter_consumed=True, random_rec=False):
        self._inner_seq = inner_id
        return super().recommend_user(user, n_rec, user_feats, seq, cold_start, inner_id, filter_consumed, random_rec)
