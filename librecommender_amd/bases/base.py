"""Common model surface (`libreco/bases/base.py:8-135`): `fit / predict / recommend_user /
save / load`.  Subclasses hold device-resident state and run the HIP hot path."""
from __future__ import annotations

import abc
import inspect
import json
import os
import time

import numpy as np
import torch

from ..training import get_trainer
from ..utils.validate import check_fitting


def hip_device(device="cuda") -> torch.device:
    """The compute device.  There is no CPU execution path: models need an MI355X."""
    dev = torch.device(device if isinstance(device, str) else device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("librecommender_amd models run on a HIP device (MI355X); no device is "
                           "visible and there is no CPU fallback")
    return dev


class Base(abc.ABC):
    uses_features = False      # collators slice sparse/dense feature columns
    uses_sequence = False      # collators build behaviour sequences
    graph_backend = "tf"       # which batch convention of the reference the model follows
    eval_user_batch = 1024     # users per recommend_user call during evaluation

    def __init__(self, task, data_info, lower_upper_bound=None):
        if task not in ("rating", "ranking"):
            raise ValueError("task must either be rating or ranking")
        self.task = task
        self.data_info = data_info
        self.n_users, self.n_items = data_info.n_users, data_info.n_items
        self.user_consumed = data_info.user_consumed
        if task == "rating":
            self.global_mean = data_info.global_mean
            if lower_upper_bound is not None:
                assert isinstance(lower_upper_bound, (list, tuple)), \
                    "must contain both lower and upper bound if provided"
                self.lower_bound, self.upper_bound = lower_upper_bound
            else:
                self.lower_bound, self.upper_bound = data_info.min_max_rating
        self.default_pred = data_info.global_mean if task == "rating" else 0.0
        self.default_recs = None
        self.model_built = False
        self.trainer = None
        self.loaded = False
        self._consumed_index = None

    @property
    def model_name(self):
        return type(self).__name__

    @property
    def consumed_index(self):
        if self._consumed_index is None:
            from ..recommendation import ConsumedIndex
            self._consumed_index = ConsumedIndex(self.user_consumed, self.n_users)
        return self._consumed_index

    # ---- template ---------------------------------------------------------------------------
    @abc.abstractmethod
    def build_model(self):
        ...

    @abc.abstractmethod
    def train_on_batch(self, batch) -> torch.Tensor:
        """One optimisation step on a collated batch; returns the (device) loss."""

    def after_fit(self):
        """Hook: export embeddings, OOV rows, default recommendations."""

    # ---- learning-rate decay of the TF trainer (`training/tf_trainer.py:111-121`,
    # `tfops/configs.py:38-45`): lr0 * 0.96 ** floor(global_step / decay_steps), staircase, with
    # decay_steps = int(data_size / batch_size) and one global step per optimiser step -------------
    def current_lr(self):
        lr0 = self.lr
        net = getattr(self, "net", None)
        if not getattr(self, "lr_decay", False) or net is None:
            return lr0
        decay_steps = max(1, int(self.data_info.data_size / self.batch_size))
        return lr0 * 0.96 ** (int(net.step) // decay_steps)

    def apply_lr_schedule(self):
        """Called before every optimiser step by the TF-graph models."""
        if getattr(self, "lr_decay", False):
            self.net.lr = self.current_lr()

    @staticmethod
    def show_start_time():
        print(f"Training start time: \x1b[35m{time.strftime('%Y-%m-%d %H:%M:%S')}\x1b[0m")

    def fit(self, train_data, neg_sampling, verbose=1, shuffle=True, eval_data=None, metrics=None,
            k=10, eval_batch_size=8192, eval_user_num=None, num_workers=0):
        check_fitting(self, train_data, eval_data, neg_sampling, k)
        if verbose > 0:
            self.show_start_time()
        if not self.model_built:
            self.build_model()
            self.model_built = True
        from .. import distributed as D

        if D.active() is not None and getattr(self, "_dist", None) is None:
            # a model without a sharded net would silently train one independent replica per rank
            raise RuntimeError(f"{self.model_name}: multi-GPU `fit` (torch.distributed is initialised with more than one rank) "
                               "is implemented for TwoTower, LightGCN, FM / DeepFM with plain sparse columns and DIN on pure "
                               "ids; run this model in a single process")
        if self.trainer is None:
            self.trainer = get_trainer(self)
        self.trainer.run(train_data, neg_sampling, verbose, shuffle, eval_data, metrics, k,
                         eval_batch_size, eval_user_num, num_workers)
        self.after_fit()

    @abc.abstractmethod
    def predict(self, user, item, cold_start="average", inner_id=False):
        ...

    @abc.abstractmethod
    def recommend_user(self, user, n_rec, **kwargs):
        ...

    # ---- persistence (`utils/save_load.py:11-23,70-80`): hyper-parameters as json, variables as npz
    def _hparams(self):
        sig = inspect.signature(type(self).__init__)
        out = {}
        for name in sig.parameters:
            if name in ("self", "data_info") or name not in self.all_args:
                continue
            v = self.all_args[name]
            if isinstance(v, (np.integer, np.floating)):
                v = v.item()
            if isinstance(v, (int, float, str, bool, list, tuple, type(None))):
                out[name] = v
        return out

    @abc.abstractmethod
    def state_arrays(self) -> dict:
        """name -> numpy array of every variable needed for inference."""

    @abc.abstractmethod
    def load_state_arrays(self, arrays: dict):
        ...

    def optimizer_arrays(self) -> dict:
        """name -> numpy array of the optimiser state (keys start with ``opt::``)."""
        return {}

    def _saved_arrays(self, path, model_name) -> dict:
        return dict(np.load(os.path.join(path, f"{model_name}_variables.npz")))

    def save(self, path, model_name, inference_only=False, **_):
        if getattr(self, "_dist", None) is not None:
            # one process per GPU: tables per shard, replicated parameters once (distributed.save_sharded); every rank calls
            from .. import distributed as D

            return D.save_sharded(self, path, model_name)
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, f"{model_name}_hyper_parameters.json"), "w") as f:
            json.dump(self._hparams(), f, separators=(",", ":"), indent=4)
        arrays = self.state_arrays()
        if not inference_only:           # optimiser state for `rebuild_model(full_assign=True)`
            arrays.update(self.optimizer_arrays())
        if self.default_recs is not None:
            arrays["default_recs"] = np.asarray(self.default_recs)
        np.savez_compressed(os.path.join(path, f"{model_name}_variables.npz"), **arrays)

    @classmethod
    def load(cls, path, model_name, data_info, **_):
        with open(os.path.join(path, f"{model_name}_hyper_parameters.json")) as f:
            hp = json.load(f)
        model = cls(data_info=data_info, **hp)
        model.build_model()
        model.model_built = True
        if getattr(model, "_dist", None) is not None:      # a sharded checkpoint, re-sharded if the world size changed
            from .. import distributed as D

            D.load_sharded(model, path, model_name)
            if hasattr(model, "set_embeddings"):
                model.set_embeddings()
            model.loaded = True
            return model
        if not os.path.exists(os.path.join(path, f"{model_name}_variables.npz")):
            from .. import distributed as D

            if D.has_sharded_checkpoint(path, model_name):   # written under a process group, read by one process
                D.load_sharded_single(model, path, model_name)
                if hasattr(model, "set_embeddings"):
                    model.set_embeddings()
                model.loaded = True
                return model
        arrays = dict(np.load(os.path.join(path, f"{model_name}_variables.npz")))
        if "default_recs" in arrays:
            model.default_recs = arrays.pop("default_recs")
        arrays = {k: v for k, v in arrays.items() if not k.startswith("opt::")}
        model.load_state_arrays(arrays)
        model.loaded = True
        return model
