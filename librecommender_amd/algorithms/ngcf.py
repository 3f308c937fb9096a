"""`NGCF` (`libreco/algorithms/ngcf.py:12-143`): same constructor and checks; propagation, batch-row
gather / scatter and Adam on the HIP kernels LightGCN uses."""
from __future__ import annotations

import numpy as np
import torch

from ..bases.base import hip_device
from ..nets.ngcf_net import NGCFNet
from ..utils.validate import hidden_units_config
from .lightgcn import LightGCN


class NGCF(LightGCN):
    def __init__(self, task, data_info, loss_type="cross_entropy", embed_size=16, n_epochs=20, lr=0.001,
                 lr_decay=False, epsilon=1e-8, amsgrad=False, reg=None, batch_size=256, num_neg=1,
                 node_dropout=0.0, message_dropout=0.0, hidden_units=(64, 64, 64), margin=1.0,
                 sampler="random", seed=42, device="cuda", lower_upper_bound=None):
        all_args = dict(locals())
        if task != "ranking":
            raise ValueError("NGCF is only suitable for ranking")
        if loss_type not in ("cross_entropy", "focal", "bpr", "max_margin"):
            raise ValueError(f"unsupported `loss_type` for NGCF: {loss_type}")
        super().__init__(task, data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, amsgrad, reg,
                         batch_size, num_neg, 0.0, len(hidden_units_config(hidden_units)), margin, sampler, seed,
                         device, lower_upper_bound)
        self.all_args = all_args
        self.node_dropout, self.message_dropout = node_dropout, message_dropout
        self.hidden_units = hidden_units_config(hidden_units)

    def build_model(self):
        self.device = hip_device(self._device_arg)
        self.net = NGCFNet(self.n_users, self.n_items, self.embed_size, self.hidden_units, self.node_dropout,
                           self.message_dropout, self.user_consumed, self.device, self.seed, self.lr,
                           self.epsilon, self.reg, self.margin, amsgrad=self.amsgrad)

    def variables_np(self):
        return {f"var::{k}": p.cpu().numpy() for k, p in self.net.params.items()}

    def load_variables_np(self, arrays):
        for k, p in self.net.params.items():
            if f"var::{k}" in arrays:
                p.copy_(torch.from_numpy(arrays[f"var::{k}"]))

    def optimizer_arrays(self):
        n = self.net
        out = {"opt::step": np.asarray(n.step, dtype=np.int64)}
        for k in n.params:
            out[f"opt::m::{k}"], out[f"opt::v::{k}"] = n.m[k].cpu().numpy(), n.v[k].cpu().numpy()
            if n.vmax is not None:
                out[f"opt::vmax::{k}"] = n.vmax[k].cpu().numpy()
        return out

    def rebuild_model(self, path, model_name):
        """`torchops/rebuild.py:13-105`: node-table rows (and their Adam states) of known users / items
        move to their new positions, the layer weights and their states are taken over whole."""
        old = self.data_info.old_info
        if old is None:
            raise ValueError("`rebuild_model` needs a `data_info` produced by `merge_trainset`")
        self.build_model()
        self.model_built = True
        arrays = self._saved_arrays(path, model_name)
        n = self.net
        src = np.concatenate([np.arange(old.n_users), old.n_users + np.arange(old.n_items)])
        dst = torch.from_numpy(np.concatenate([np.arange(old.n_users), self.n_users + np.arange(old.n_items)])).to(self.device)
        states = [("var::", n.params), ("opt::m::", n.m), ("opt::v::", n.v)] + ([("opt::vmax::", n.vmax)] if n.vmax else [])
        with torch.no_grad():
            for prefix, group in states:
                for k, new in group.items():
                    saved = arrays.get(prefix + k)
                    if saved is None:
                        continue
                    if k == "embed":
                        new[dst] = torch.from_numpy(saved[src]).to(self.device)
                    else:
                        new.copy_(torch.from_numpy(saved))
            if "opt::step" in arrays:
                n.step = int(arrays["opt::step"])
