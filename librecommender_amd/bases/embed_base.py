"""Models that export user / item embeddings and serve them by dot product
(`libreco/bases/embed_base.py:24-265`).  Embeddings stay on the device: `predict` is
`lr_pair_dot_f32`, `recommend_user` is `lr_score_topk_f32`."""
from __future__ import annotations

import numpy as np
import torch

from ..prediction import predict_from_embedding
from ..recommendation import cold_start_rec, construct_rec, recommend_from_embedding
from ..utils.validate import check_unknown_user
from .base import Base


class EmbedBase(Base):
    def __init__(self, task, data_info, embed_size, lower_upper_bound=None):
        super().__init__(task, data_info, lower_upper_bound)
        self.embed_size = embed_size
        self.user_embeds: torch.Tensor = None   # [n_users + 1, D] on device (last row = OOV)
        self.item_embeds: torch.Tensor = None   # [n_items + 1, D]

    # numpy views with the reference's attribute names
    @property
    def user_embeds_np(self):
        return None if self.user_embeds is None else self.user_embeds.cpu().numpy()

    @property
    def item_embeds_np(self):
        if self.item_embeds is None:
            return None
        e = self.item_embeds if isinstance(self.item_embeds, torch.Tensor) else self.item_embeds.gather()
        return e.cpu().numpy()

    def on_epoch_end(self, epoch):
        pass

    def prepare_for_eval(self):
        self.set_embeddings()
        self.assign_embedding_oov()

    def set_embeddings(self):
        raise NotImplementedError

    def assign_embedding_oov(self):
        """Append the mean row as the OOV embedding (`embed_base.py:257-265`)."""
        for name, n in (("user_embeds", self.n_users), ("item_embeds", self.n_items)):
            e = getattr(self, name)
            if not isinstance(e, torch.Tensor):      # sharded item embeddings carry their (replicated) OOV row
                continue
            if e.shape[0] == n:
                setattr(self, name, torch.cat([e, e.mean(dim=0, keepdim=True)], dim=0).contiguous())

    def after_fit(self):
        self.set_embeddings()
        self.assign_embedding_oov()
        # recommendations of the OOV user, unfiltered (`embed_base.py:153-161`)
        self.default_recs = recommend_from_embedding(
            self, [self.n_users], min(2000, self.n_items), self.user_embeds, self.item_embeds,
            filter_consumed=False, random_rec=False).flatten()

    def predict(self, user, item, cold_start="average", inner_id=False):
        return predict_from_embedding(self, user, item, cold_start, inner_id)

    def recommend_user(self, user, n_rec, cold_start="average", inner_id=False,
                       filter_consumed=True, random_rec=False):
        out = {}
        known, unknown = check_unknown_user(self.data_info, user, inner_id)
        if unknown:
            out.update(cold_start_rec(self.data_info, self.default_recs, cold_start, unknown, n_rec, inner_id))
        if known:
            recs = recommend_from_embedding(self, known, n_rec, self.user_embeds, self.item_embeds,
                                            filter_consumed, random_rec)
            out.update(construct_rec(self.data_info, known, recs, inner_id))
        return out

    def get_user_id(self, user):
        if user not in self.data_info.user2id:
            raise ValueError(f"unknown user: {user}")
        return self.data_info.user2id[user]

    def get_item_id(self, item):
        if item not in self.data_info.item2id:
            raise ValueError(f"unknown item: {item}")
        return self.data_info.item2id[item]

    def _known_rows(self, side: str, include_bias: bool) -> torch.Tensor:
        """Rows of the known users / items (OOV row dropped); only the first `embed_size` columns
        unless `include_bias` (`embed_base.py:368-372,404-408`)."""
        e = self.user_embeds if side == "user" else self.item_embeds
        if e is not None and not isinstance(e, torch.Tensor):
            e = e.gather()                        # sharded export: assembled on request
        assert e is not None, f"call `model.fit()` before getting {side} embeddings"
        e = e[: self.n_users if side == "user" else self.n_items]
        return e if include_bias else e[:, : self.embed_size]

    def get_user_embedding(self, user=None, include_bias=False):
        e = self._known_rows("user", include_bias).cpu().numpy()
        return e if user is None else e[self.get_user_id(user)]

    def get_item_embedding(self, item=None, include_bias=False):
        e = self._known_rows("item", include_bias).cpu().numpy()
        return e if item is None else e[self.get_item_id(item)]

    # ---- nearest neighbours in embedding space (`embed_base.py:415-551`) ------------------------
    def init_knn(self, approximate, sim_type, M=100, ef_construction=200, ef_search=200):
        """Prepare `search_knn_users / search_knn_items`.  The search is always the exact
        full scan on the MFMA pipe (`lr_score_topk_f32` over unit-normalised rows for "cosine"):
        there is no approximate index here, `approximate=True` (HNSW via nmslib in the reference)
        is accepted and served exactly; `M / ef_construction / ef_search` are ignored."""
        if sim_type == "cosine":
            self.include_bias = False
        elif sim_type == "inner-product":
            self.include_bias = True
        else:
            raise ValueError(f"unknown sim_type: {sim_type}, only `cosine` and `inner-product` are supported")
        self._knn_rows = {}
        for side in ("user", "item"):
            e = self._known_rows(side, self.include_bias)
            if sim_type == "cosine":
                norm = torch.linalg.vector_norm(e, dim=1, keepdim=True)
                e = e / torch.where(norm == 0, torch.ones_like(norm), norm)
            self._knn_rows[side] = e.contiguous()
        self.approximate, self.sim_type = approximate, sim_type

    def _knn_topk(self, queries, rows, k):
        from .. import ops
        return ops.score_topk(queries, rows, k)

    def _search_knn(self, side, inner, k):
        rows = self._knn_rows[side]
        _, ids = self._knn_topk(rows[inner:inner + 1].contiguous(), rows, k)
        return [int(i) for i in ids[0].tolist()]

    def search_knn_users(self, user, k):
        """The k users most similar to `user` (itself included), most similar first."""
        return [self.data_info.id2user[i] for i in self._search_knn("user", self.get_user_id(user), k)]

    def search_knn_items(self, item, k):
        return [self.data_info.id2item[i] for i in self._search_knn("item", self.get_item_id(item), k)]

    def state_arrays(self):
        return {"user_embed": self.user_embeds_np, "item_embed": self.item_embeds_np, **self.variables_np()}

    def variables_np(self) -> dict:
        return {}

    def load_state_arrays(self, arrays):
        dev = self.device
        self.user_embeds = torch.from_numpy(arrays.pop("user_embed")).to(dev)
        self.item_embeds = torch.from_numpy(arrays.pop("item_embed")).to(dev)
        self.load_variables_np(arrays)

    def load_variables_np(self, arrays):
        pass

    # ---- persistence: the reference's inference layout (`bases/embed_base.py:267-331`) ----------
    def save(self, path, model_name, inference_only=False, **kw):
        """Always writes `{model_name}.npz` (user_embed / item_embed), `{model_name}_default_recs.npz`
        and `{model_name}_hyper_parameters.json` exactly like the reference, so an inference
        checkpoint is interchangeable in both directions; the full state for `rebuild_model` /
        dynamic features goes to `{model_name}_variables.npz` unless `inference_only`."""
        import json
        import os

        if getattr(self, "_dist", None) is not None:          # one process per GPU: per-shard checkpoint (every rank calls)
            from .. import distributed as D

            return D.save_sharded(self, path, model_name)
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, f"{model_name}_hyper_parameters.json"), "w") as f:
            json.dump(self._hparams(), f, separators=(",", ":"), indent=4)
        if self.default_recs is not None:
            np.savez_compressed(os.path.join(path, f"{model_name}_default_recs"), default_recs=np.asarray(self.default_recs))
        np.savez_compressed(os.path.join(path, model_name), user_embed=self.user_embeds_np, item_embed=self.item_embeds_np)
        if not inference_only:
            super().save(path, model_name, inference_only=False, **kw)
        else:   # `load` prefers the full checkpoint: an older one would shadow these embeddings
            stale = os.path.join(path, f"{model_name}_variables.npz")
            if os.path.exists(stale):
                os.remove(stale)

    @classmethod
    def load(cls, path, model_name, data_info, **kw):
        import inspect
        import json
        import os

        from .. import distributed as D

        if D.active() is not None:                             # a sharded checkpoint (distributed.load_sharded)
            return super().load(path, model_name, data_info, **kw)
        full = os.path.join(path, f"{model_name}_variables.npz")
        if os.path.exists(full) or D.has_sharded_checkpoint(path, model_name):
            return super().load(path, model_name, data_info, **kw)
        # inference checkpoint (ours or one written by the reference): embeddings only
        with open(os.path.join(path, f"{model_name}_hyper_parameters.json")) as f:
            hp = json.load(f)
        accepted = set(inspect.signature(cls.__init__).parameters)
        model = cls(data_info=data_info, **{k: v for k, v in hp.items() if k in accepted})
        from .base import hip_device
        model.device = hip_device(getattr(model, "_device_arg", None) or "cuda")
        arrays = np.load(os.path.join(path, f"{model_name}.npz"))
        model.user_embeds = torch.from_numpy(arrays["user_embed"]).to(model.device).contiguous()
        model.item_embeds = torch.from_numpy(arrays["item_embed"]).to(model.device).contiguous()
        rec = os.path.join(path, f"{model_name}_default_recs.npz")
        if os.path.exists(rec):
            model.default_recs = np.load(rec)["default_recs"]
        model.loaded = True
        return model
