"""`LightGCN` (`libreco/algorithms/lightgcn.py`): same constructor and checks, HIP hot path."""
from __future__ import annotations

import math

import numpy as np
import torch

from ..bases import EmbedBase
from ..bases.base import hip_device
from ..batch.batch_unit import PairwiseBatch
from ..nets.graph_nets import LightGCNNet, cosine_warm_restart_lr


class LightGCN(EmbedBase):
    graph_backend = "torch"     # positives are not repeated per negative (batch_data.py:87-88)

    def __init__(self, task, data_info, loss_type="bpr", embed_size=16, n_epochs=20, lr=0.001,
                 lr_decay=False, epsilon=1e-8, amsgrad=False, reg=None, batch_size=256, num_neg=1,
                 dropout_rate=0.0, n_layers=3, margin=1.0, sampler="random", seed=42, device="cuda",
                 lower_upper_bound=None, with_training=True, device_sampling=False):
        super().__init__(task, data_info, embed_size, lower_upper_bound)
        self.all_args = locals()
        self.device_sampling = device_sampling      # row f1: permutation, negatives, triples on the device
        self.loss_type, self.n_epochs, self.lr, self.lr_decay = loss_type, n_epochs, lr, lr_decay
        self.epsilon, self.amsgrad, self.reg = epsilon, amsgrad, reg
        self.batch_size, self.num_neg, self.dropout_rate = batch_size, num_neg, dropout_rate
        self.n_layers, self.margin, self.sampler, self.seed = n_layers, margin, sampler, seed
        self._device_arg = device
        if self.task != "ranking":
            raise ValueError("LightGCN is only suitable for ranking")
        if self.loss_type not in ("cross_entropy", "focal", "bpr", "max_margin"):
            raise ValueError(f"unsupported `loss_type` for LightGCN: {self.loss_type}")
        self._epoch, self._batch_in_epoch, self._n_batches = 1, 0, 1

    def build_model(self):
        from .. import distributed as D

        self._dist = D.active()
        if self._dist is not None:
            # one process per GPU: 1-D row partition of the node table / Laplacian (nets/graph_nets.py:ShardedLightGCNNet)
            from ..nets.graph_nets import ShardedLightGCNNet

            self.device = D.device_for(self._device_arg)
            self.net = ShardedLightGCNNet(self.n_users, self.n_items, self.embed_size, self.n_layers, self.user_consumed,
                                          self.device, kern=D.kernels(), seed=self.seed, lr=self.lr, epsilon=self.epsilon,
                                          reg=self.reg, margin=self.margin, dropout=self.dropout_rate, amsgrad=self.amsgrad)
            return
        self.device = hip_device(self._device_arg)
        self.net = LightGCNNet(self.n_users, self.n_items, self.embed_size, self.n_layers,
                               self.dropout_rate, self.user_consumed, self.device, self.seed, self.lr,
                               self.epsilon, self.reg, self.margin, amsgrad=self.amsgrad)

    def fit(self, train_data, neg_sampling, verbose=1, shuffle=True, eval_data=None, metrics=None, k=10,
            eval_batch_size=8192, eval_user_num=None, num_workers=0):
        from ..batch import adjust_batch_size
        if getattr(self, "loaded", False):           # `check_fitting` (utils/validate.py:156-161)
            raise RuntimeError("Loaded model doesn't support retraining, use `rebuild_model` instead.")
        self._n_batches = max(1, math.ceil(len(train_data) / adjust_batch_size(self, self.batch_size)))
        super().fit(train_data, neg_sampling, verbose, shuffle, eval_data, metrics, k, eval_batch_size, eval_user_num,
                    num_workers)

    def current_lr(self):
        if not self.lr_decay:
            return self.lr
        return cosine_warm_restart_lr(self.lr, (self._epoch - 1) + self._batch_in_epoch / self._n_batches)

    def on_epoch_end(self, epoch):
        self._epoch, self._batch_in_epoch = epoch + 1, 0

    def train_on_batch(self, b):
        lr = self.current_lr()
        self._batch_in_epoch += 1
        if getattr(self, "_dist", None) is not None:         # this rank's slice of the (identical on every rank) batch
            from .. import distributed as D

            rank, world = self._dist
            if isinstance(b, PairwiseBatch):
                sl = D.batch_slice(len(b.queries), rank, world)
                if sl.stop == sl.start:        # fewer samples than ranks (a tiny last batch): every rank skips the step
                    return torch.zeros((), device=self.device)
                f = len(b.item_pairs[1]) // max(len(b.item_pairs[0]), 1)       # negatives per positive
                nsl = slice(sl.start * f, sl.stop * f)
                return self.net.train_step(self.loss_type, b.queries[sl], b.item_pairs[0][sl], items_neg=b.item_pairs[1][nsl],
                                           lr=lr)[0]
            sl = D.batch_slice(len(b.users), rank, world)
            if sl.stop == sl.start:        # fewer samples than ranks (a tiny last batch): every rank skips the step
                return torch.zeros((), device=self.device)
            return self.net.train_step(self.loss_type, b.users[sl], b.items[sl], labels=b.labels[sl], lr=lr)[0]
        if isinstance(b, PairwiseBatch):
            loss, _ = self.net.train_step(self.loss_type, b.queries, b.item_pairs[0], items_neg=b.item_pairs[1], lr=lr)
        else:
            loss, _ = self.net.train_step(self.loss_type, b.users, b.items, labels=b.labels, lr=lr)
        return loss

    def set_embeddings(self):
        if getattr(self, "_dist", None) is not None:
            # sharded export: user rows all-gathered, the item rows of this rank's node range stay local
            from .. import distributed as D

            self.user_embeds, loc, base, n_local = self.net.embeddings_sharded()
            self.item_embeds = D.ShardedItemEmbeds(loc, self.n_items, base, n_local, self.net.group, self.net.kern)
            return
        self.user_embeds, self.item_embeds = self.net.embeddings()

    def variables_np(self):
        return {"init_embeds": self.net.E.cpu().numpy()}

    def optimizer_arrays(self):
        n = self.net
        out = {"opt::m": n.m.cpu().numpy(), "opt::v": n.v.cpu().numpy(), "opt::step": np.asarray(n.step, dtype=np.int64)}
        if n.vmax is not None:
            out["opt::vmax"] = n.vmax.cpu().numpy()
        return out

    def rebuild_model(self, path, model_name):
        """`torchops/rebuild.py:13-105`: saved user / item embedding rows and their Adam states are
        copied into the (larger) new tables; new ids keep the fresh initialisation / zero moments."""
        old = self.data_info.old_info
        if old is None:
            raise ValueError("`rebuild_model` needs a `data_info` produced by `merge_trainset`")
        self.build_model()
        self.model_built = True
        arrays = self._saved_arrays(path, model_name)
        n = self.net
        src = np.concatenate([np.arange(old.n_users), old.n_users + np.arange(old.n_items)])
        dst = np.concatenate([np.arange(old.n_users), self.n_users + np.arange(old.n_items)])
        dst_t = torch.from_numpy(dst).to(self.device)
        with torch.no_grad():
            for key, new in (("init_embeds", n.E), ("opt::m", n.m), ("opt::v", n.v), ("opt::vmax", n.vmax)):
                if new is not None and key in arrays:
                    new[dst_t] = torch.from_numpy(arrays[key][src]).to(self.device)
            if "opt::step" in arrays:
                n.step = int(arrays["opt::step"])

    def load_variables_np(self, arrays):
        if "init_embeds" in arrays:
            self.net.E.copy_(torch.from_numpy(arrays["init_embeds"]))
