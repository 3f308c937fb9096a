"""`DIN` (`libreco/algorithms/din.py`): same constructor, HIP path; the attention over a user's
recent items runs in `lr_din_attn_pool_fwd/bwd_f32` when items carry no side features."""
from __future__ import annotations

import numpy as np
import torch

from ..bases import FeatBase
from ..bases.base import hip_device
from ..batch.sequence import get_recent_seqs
from ..nets import FeatDINNet, FeatSpec
from ..utils.validate import (check_multi_sparse, check_seq_mode, dropout_config, hidden_units_config,
                              reg_config)


class DIN(FeatBase):
    uses_sequence = True

    def __init__(self, task, data_info=None, loss_type="cross_entropy", embed_size=16, n_epochs=20,
                 lr=0.001, lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random",
                 num_neg=1, use_bn=True, dropout_rate=None, hidden_units=(128, 64, 32), recent_num=10,
                 random_num=None, use_tf_attention=False, multi_sparse_combiner="sqrtn", seed=42,
                 lower_upper_bound=None, tf_sess_config=None, device="cuda", dense_adam=False,
                 device_sampling=False, graph_step=True):
        super().__init__(task, data_info, lower_upper_bound)
        self.all_args = locals()
        self.graph_step = graph_step                 # fused step replayed as one hipGraph where the net supports it
        self.loss_type, self.embed_size, self.n_epochs = loss_type, embed_size, n_epochs
        self.lr, self.lr_decay, self.epsilon, self.reg = lr, lr_decay, epsilon, reg_config(reg)
        self.batch_size, self.sampler, self.num_neg, self.use_bn = batch_size, sampler, num_neg, use_bn
        self.dropout_rate = dropout_config(dropout_rate)
        self.hidden_units = hidden_units_config(hidden_units)
        self.use_tf_attention = use_tf_attention
        self.seq_mode, self.max_seq_len = check_seq_mode(recent_num, random_num)
        self.recent_seqs, self.recent_seq_lens = get_recent_seqs(self.n_users, self.user_consumed,
                                                                 self.n_items, self.max_seq_len)
        self.seed = seed
        self.sparse = bool(data_info.sparse_col.name)
        self.dense = bool(data_info.dense_col.name)
        self.multi_sparse_combiner = check_multi_sparse(data_info, multi_sparse_combiner) if self.sparse else "normal"
        # `reg` is the reference's L2 regulariser on the embedding VARIABLES (deepfm.py:181-259): its gradient reaches every row of
        # every table each step, i.e. it exists only under TF1's dense update — asking for it selects `dense_adam` (row-wise
        # Adam on the touched rows would drop the term silently)
        self._device_arg, self.dense_adam = device, bool(dense_adam or self.reg)
        self.device_sampling = device_sampling       # row f1: negatives, collation, sequences on the device

    def build_model(self):
        from .. import distributed as D

        self._dist = D.active()
        if self._dist is not None:
            # one process per GPU: the [user | item] table row-sharded over the ranks, the batch data-parallel, the rows of
            # [user, item, window] fetched through the tables' lookup collective (nets/feat_nets.py:ShardedDINNet)
            if self.task != "ranking" or self.loss_type != "cross_entropy":
                raise ValueError("the row-sharded DIN trains with the cross-entropy loss on a ranking task")
            from ..nets.feat_nets import ShardedDINNet

            self.device = D.device_for(self._device_arg)
            if self.sparse or self.dense or self.dropout_rate or self.use_tf_attention:
                # feature columns (item side features join the attention keys, algorithms/din.py:165-250 of the reference),
                # dropout, the plain dot-product attention: the general feature layer on the step's row cache, the rows
                # of the target and of the behaviour window riding in the same exchange (nets/feat_nets.py:FeatDINNet)
                d = self.data_info
                self.net = FeatDINNet(FeatSpec.from_data_info(d, self.multi_sparse_combiner), self.embed_size,
                                      self.hidden_units, self.use_bn, self.dropout_rate, self.max_seq_len,
                                      d.item_sparse_unique, d.item_dense_unique, d.item_dense_col.index, self.lr,
                                      self.epsilon, self.seed, self.device, use_tf_attention=self.use_tf_attention,
                                      sharded=True, kern=D.kernels())
                self.net.tables.dense_adam, self.net.tables.l2 = bool(self.dense_adam), float(self.reg or 0.0)
                return
            self.net = ShardedDINNet(self.n_users + 1 + self.n_items + 1, self.embed_size, self.hidden_units, self.use_bn,
                                     self.max_seq_len, self.lr, self.epsilon, self.seed, self.device, kern=D.kernels())
            self.net.tables.set_layout(self.n_users, self.n_items)
            self.net.tables.dense_adam, self.net.tables.l2 = bool(self.dense_adam), float(self.reg or 0.0)
            return
        self.device = hip_device(self._device_arg)
        d = self.data_info
        self.net = FeatDINNet(FeatSpec.from_data_info(d, self.multi_sparse_combiner), self.embed_size,
                              self.hidden_units, self.use_bn, self.dropout_rate, self.max_seq_len,
                              d.item_sparse_unique, d.item_dense_unique, d.item_dense_col.index, self.lr,
                              self.epsilon, self.seed, self.device, self.dense_adam, self.reg,
                              use_tf_attention=self.use_tf_attention, graph_step=self.graph_step)

    def _seq_args(self, b):
        return {"seqs": b.seqs.interacted_seq, "seq_lens": b.seqs.interacted_len}

    def takes_next_batch(self) -> bool:
        """See `FM.takes_next_batch`: the trainer announces the next batch, whose exchange plan is built a step ahead."""
        return getattr(self, "_dist", None) is not None and not hasattr(self.net, "emb")

    def _rank_inputs(self, b):
        """This rank's slice of the batch as (idx, seq_lens, labels), or None when the batch has fewer samples than ranks; the
        tensors of the batch announced one step earlier are re-used (the prefetched plan is recognised by its id tensor)."""
        from .. import distributed as D

        held = getattr(self, "_held_inputs", None)
        if held is not None and held[0] is b:
            return held[1]
        rank, world = self._dist
        sl = D.batch_slice(len(b.users), rank, world)
        if sl.stop == sl.start:
            return None
        idx = self.net._idx(D.take(b.users, sl), D.take(b.items, sl), D.take(b.seqs.interacted_seq, sl))
        return idx, D.take(b.seqs.interacted_len, sl), D.take(b.labels, sl)

    def train_on_batch(self, b, next_batch=None):
        if getattr(self, "_dist", None) is None:
            return super().train_on_batch(b)
        from .. import distributed as D          # this rank's contiguous slice of the (identical on every rank) batch

        self.apply_lr_schedule()
        if hasattr(self.net, "emb"):
            rank, world = self._dist
            sl = D.batch_slice(len(b.users), rank, world)
            if sl.stop == sl.start:        # fewer samples than ranks (a tiny last batch): every rank skips the step
                return torch.zeros((), device=self.device)
            return self.net.train_step(D.take(b.users, sl), D.take(b.items, sl), D.take(b.labels, sl),
                                       sparse=D.take(b.sparse_indices, sl), dense=D.take(b.dense_values, sl),
                                       seqs=D.take(b.seqs.interacted_seq, sl), seq_lens=D.take(b.seqs.interacted_len, sl),
                                       loss_type=self._loss_name())
        cur = self._rank_inputs(b)
        self._held_inputs = None
        nxt = None
        if next_batch is not None:
            nxt = self._rank_inputs(next_batch)
            if nxt is not None:
                self._held_inputs = (next_batch, nxt)
        if cur is None:                # fewer samples than ranks (a tiny last batch): every rank skips the step
            return torch.zeros((), device=self.device)
        return self.net.train_step(cur[0], cur[1], cur[2], next_idx=None if nxt is None else nxt[0])

    def _cached_seq(self, users):
        return self.recent_seqs[users], self.recent_seq_lens[users]

    def _seq_for(self, uid, seq):
        if seq is not None and len(seq) > 0:          # `build_rec_seq`, recommendation/preprocess.py:47-56
            ids = [self.data_info.item2id.get(i, self.n_items) for i in seq] if not self._inner_seq else list(seq)
            n = min(self.max_seq_len, len(ids))
            out = np.full((1, self.max_seq_len), self.n_items, dtype=np.int32)
            out[0, :n] = ids[-n:]
            return out, np.array([n], dtype=np.int32)
        return self.recent_seqs[[uid]], self.recent_seq_lens[[uid]]

    _inner_seq = False

    def recommend_user(self, user, n_rec, user_feats=None, seq=None, cold_start="average", inner_id=False,
                       filter_consumed=True, random_rec=False):
        self._inner_seq = inner_id
        return super().recommend_user(user, n_rec, user_feats, seq, cold_start, inner_id, filter_consumed,
                                      random_rec)
