"""`FM` and `DeepFM` (`libreco/algorithms/fm.py`, `deepfm.py`): same constructors, on the HIP path.
Data whose sparse columns are all plain and which has no dense columns runs on the fully fused
nets (`nets/fm_nets.py`); anything else on the general feature nets (`nets/feat_nets.py`)."""
from __future__ import annotations

import torch

from ..bases import FeatBase
from ..bases.base import hip_device
from ..nets import DeepFMNet, FeatDeepFMNet, FeatFMNet, FeatSpec, FMNet
from ..utils.validate import check_multi_sparse, dropout_config, hidden_units_config, reg_config


class _FMCommon(FeatBase):
    def _common(self, data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, reg, batch_size,
                sampler, num_neg, use_bn, dropout_rate, multi_sparse_combiner, seed, device, dense_adam):
        self.loss_type, self.embed_size, self.n_epochs = loss_type, embed_size, n_epochs
        self.lr, self.lr_decay, self.epsilon, self.reg = lr, lr_decay, epsilon, reg_config(reg)
        self.batch_size, self.sampler, self.num_neg, self.use_bn = batch_size, sampler, num_neg, use_bn
        self.dropout_rate = dropout_config(dropout_rate)
        self.seed = seed
        self.sparse = bool(data_info.sparse_col.name)
        self.dense = bool(data_info.dense_col.name)
        self.multi_sparse_combiner = check_multi_sparse(data_info, multi_sparse_combiner) if self.sparse else "normal"
        # `reg` is the reference's L2 regulariser on the embedding VARIABLES (deepfm.py:181-259): its gradient reaches every row of
        # every table each step, i.e. it exists only under TF1's dense update — asking for it selects `dense_adam` (row-wise
        # Adam on the touched rows would drop the term silently)
        self._device_arg, self.dense_adam = device, bool(dense_adam or self.reg)
        if self.task == "ranking" and loss_type not in ("cross_entropy", "focal"):
            raise ValueError(f"unsupported `loss_type`: {loss_type}")

    def _spec(self):
        return FeatSpec.from_data_info(self.data_info, self.multi_sparse_combiner)

    def _shard_optimizer(self):
        """`dense_adam` / `reg` under a process group: every owner runs TF1's dense update (all rows decay and move every step,
        2 * reg * w joins every row's gradient) over its own rows (`ShardedFieldTables._apply_gradients`)."""
        self.net.tables.dense_adam, self.net.tables.l2 = bool(self.dense_adam), float(self.reg or 0.0)

    def takes_next_batch(self) -> bool:
        """The trainer hands `train_on_batch` the batch after the current one when the tables are row-sharded and the net
        works on the packed id matrix: its exchange plan is then built a step ahead (`ShardedFieldTables.prefetch`)."""
        return getattr(self, "_dist", None) is not None and not hasattr(self.net, "emb")

    def _rank_inputs(self, b):
        """This rank's contiguous slice of the (identical on every rank) batch as (idx, labels) on the device, or None when
        the batch has fewer samples than ranks.  The tensors of the batch announced as `next_batch` one step earlier are
        re-used (the prefetched plan is recognised by the identity of its id tensor)."""
        from .. import distributed as D

        held = getattr(self, "_held_inputs", None)
        if held is not None and held[0] is b:
            return held[1]
        rank, world = self._dist
        sl = D.batch_slice(len(b.users), rank, world)
        if sl.stop == sl.start:
            return None
        idx = self.net._idx(D.take(b.users, sl), D.take(b.items, sl), D.take(b.sparse_indices, sl))
        labels = torch.as_tensor(D.take(b.labels, sl), device=self.device, dtype=torch.float32)
        return idx, labels

    def train_on_batch(self, b, next_batch=None):
        if getattr(self, "_dist", None) is None:
            return super().train_on_batch(b)
        from .. import distributed as D

        self.apply_lr_schedule()
        if hasattr(self.net, "emb"):     # the general feature layer over row-sharded tables (pooled / dense columns, dropout)
            rank, world = self._dist
            sl = D.batch_slice(len(b.users), rank, world)
            if sl.stop == sl.start:    # fewer samples than ranks (a tiny last batch): every rank skips the step
                return torch.zeros((), device=self.device)
            return self.net.train_step(D.take(b.users, sl), D.take(b.items, sl), D.take(b.labels, sl),
                                       sparse=D.take(b.sparse_indices, sl), dense=D.take(b.dense_values, sl),
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
        return self.net.train_step(cur[0], cur[1], loss_type=self._loss_name(), next_idx=None if nxt is None else nxt[0])


class FM(_FMCommon):
    def __init__(self, task, data_info, loss_type="cross_entropy", embed_size=16, n_epochs=20, lr=0.001,
                 lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random", num_neg=1,
                 use_bn=True, dropout_rate=None, multi_sparse_combiner="sqrtn", seed=42,
                 lower_upper_bound=None, tf_sess_config=None, device="cuda", dense_adam=False,
                 device_sampling=False):
        super().__init__(task, data_info, lower_upper_bound)
        self.all_args = locals()
        self._common(data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, reg, batch_size,
                     sampler, num_neg, use_bn, dropout_rate, multi_sparse_combiner, seed, device, dense_adam)
        self.device_sampling = device_sampling

    def build_model(self):
        from .. import distributed as D

        spec = self._spec()
        self._dist = D.active()
        if self._dist is not None:
            # one process per GPU (round 4): tables row-sharded over the ranks, the batch data-parallel (nets/fm_nets.py:ShardedFMNet)
            from ..nets import ShardedFMNet

            self.device = D.device_for(self._device_arg)
            if spec.pooled or spec.n_dense_cols:      # the general feature layer on the step's row cache (nets/feat_embedding.py)
                self.net = FeatFMNet(spec, self.embed_size, self.use_bn, self.lr, self.epsilon, self.seed, self.device,
                                     sharded=True, kern=D.kernels())
                self._shard_optimizer()
                return
            self.net = ShardedFMNet(self.n_users + 1 + self.n_items + 1 + spec.sparse_rows, spec.n_sparse_cols, self.embed_size,
                                    self.use_bn, self.lr, self.epsilon, self.seed, self.device, kern=D.kernels())
            self.net.tables.set_layout(self.n_users, self.n_items)
            self._shard_optimizer()
            return
        self.device = hip_device(self._device_arg)
        if spec.pooled or spec.n_dense_cols or self.embed_size not in (16, 32, 64, 128):
            self.net = FeatFMNet(spec, self.embed_size, self.use_bn, self.lr, self.epsilon, self.seed,
                                 self.device, self.dense_adam, self.reg)
        else:
            self.net = FMNet(self.n_users, self.n_items, spec.sparse_rows, spec.n_sparse_cols,
                             self.embed_size, self.use_bn, self.lr, self.epsilon, self.seed, self.device,
                             self.dense_adam, self.reg)


class DeepFM(_FMCommon):
    def __init__(self, task, data_info, loss_type="cross_entropy", embed_size=16, n_epochs=20, lr=0.001,
                 lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random", num_neg=1,
                 use_bn=True, dropout_rate=None, hidden_units=(128, 64, 32), multi_sparse_combiner="sqrtn",
                 seed=42, lower_upper_bound=None, tf_sess_config=None, device="cuda", dense_adam=False,
                 device_sampling=False, graph_step=True):
        super().__init__(task, data_info, lower_upper_bound)
        self.all_args = locals()
        self.graph_step = graph_step
        self._common(data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, reg, batch_size,
                     sampler, num_neg, use_bn, dropout_rate, multi_sparse_combiner, seed, device, dense_adam)
        self.hidden_units = hidden_units_config(hidden_units)
        self.device_sampling = device_sampling

    def build_model(self):
        from .. import distributed as D

        spec = self._spec()
        self._dist = D.active()
        if self._dist is not None:
            # one process per GPU: the [user | item | sparse] tables row-sharded over the ranks (round-robin rows, RCCL
            # all-to-all of the de-duplicated ids / rows / row gradients), the batch data-parallel, dense parameters
            # replicated with one all-reduce per step (nets/fm_nets.py:ShardedDeepFMNet)
            from ..nets.fm_nets import ShardedDeepFMNet

            self.device = D.device_for(self._device_arg)
            if spec.pooled or spec.n_dense_cols or self.dropout_rate:
                # multi-sparse pooling, dense columns, dropout: the general feature layer on the step's row cache
                # (nets/feat_embedding.py:ShardedFeatEmbedding) under the autograd dense layers
                self.net = FeatDeepFMNet(spec, self.embed_size, self.hidden_units, self.use_bn, self.dropout_rate, self.lr,
                                         self.epsilon, self.seed, self.device, sharded=True, kern=D.kernels())
                self._shard_optimizer()
                return
            u_rows, i_rows = self.n_users + 1, self.n_items + 1
            offs = [int(o) for o in self.data_info.sparse_offset] if spec.n_sparse_cols else []
            starts = [0, u_rows] + [u_rows + i_rows + o for o in offs] + [u_rows + i_rows + spec.sparse_rows]
            frs = starts if all(b > a for a, b in zip(starts[:-1], starts[1:])) else None
            self.net = ShardedDeepFMNet(u_rows + i_rows + spec.sparse_rows, spec.n_sparse_cols, self.embed_size,
                                        self.hidden_units, self.use_bn, self.lr, self.epsilon, self.seed, self.device,
                                        kern=D.kernels(), field_row_start=frs)
            self.net.tables.set_layout(self.n_users, self.n_items)
            self._shard_optimizer()
            return
        self.device = hip_device(self._device_arg)
        if spec.pooled or spec.n_dense_cols or self.embed_size not in (16, 32, 64, 128):
            self.net = FeatDeepFMNet(spec, self.embed_size, self.hidden_units, self.use_bn, self.dropout_rate,
                                     self.lr, self.epsilon, self.seed, self.device, self.dense_adam, self.reg)
        else:
            self.net = DeepFMNet(self.n_users, self.n_items, spec.sparse_rows, spec.n_sparse_cols,
                                 self.embed_size, self.hidden_units, self.use_bn, self.dropout_rate or 0.0, self.lr, self.epsilon,
                                 self.seed, self.device, self.dense_adam, self.reg,
                                 sparse_offsets=self.data_info.sparse_offset if spec.n_sparse_cols else None)
            if getattr(self.net, "hip_tail", False) and self.graph_step:
                # the fused step is one hipGraph replay per batch shape (one `sess.run` per step in the reference,
                # training/tf_trainer.py:76-101), bit-identical to the eager launches; replays run on a dedicated
                # stream that is event-ordered against the loader's stream (nets/din_fused.py:GraphRunner), with the
                # host loader and with the device loader (`device_sampling=True`) alike
                self.net.enable_graph(True)
