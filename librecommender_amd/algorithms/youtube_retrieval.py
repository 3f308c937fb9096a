"""`YouTubeRetrieval` (`libreco/algorithms/youtube_retrieval.py`): same constructor, same errors, same
`fit / predict / recommend_user / dyn_user_embedding`, on the MI355X path: history pooling through
`lr_embed_bag_pool_f32`, the exported embeddings served by `lr_score_topk_f32` like every `EmbedBase` model."""
from __future__ import annotations

import numpy as np
import torch

from ..bases import EmbedBase
from ..bases.base import hip_device
from ..batch.sequence import get_recent_seqs
from ..nets import FeatSpec
from ..nets.youtube_nets import YouTubeRetrievalNet
from ..utils.validate import (check_multi_sparse, check_seq_mode, dropout_config, hidden_units_config,
                              reg_config)


class YouTubeRetrieval(EmbedBase):
    uses_features = True
    uses_sequence = True

    def __init__(self, task="ranking", data_info=None, loss_type="sampled_softmax", embed_size=16, norm_embed=False,
                 n_epochs=20, lr=0.001, lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, use_bn=True,
                 dropout_rate=None, hidden_units=(128, 64), num_sampled_per_batch=None, sampler="uniform",
                 recent_num=10, random_num=None, multi_sparse_combiner="sqrtn", seed=42, lower_upper_bound=None,
                 tf_sess_config=None, device="cuda", dense_adam=False):
        super().__init__(task, data_info, embed_size, lower_upper_bound)
        assert task == "ranking", "YouTube-type models is only suitable for ranking"
        if len(data_info.item_col) > 0:
            raise ValueError("The `YouTuBeRetrieval` model assumes no item features.")
        if loss_type not in ("sampled_softmax", "nce"):
            raise ValueError("Loss type must either be `nce` or `sampled_softmax`")
        if sampler != "uniform":
            # TF's `log_uniform_candidate_sampler` fallback (tf_trainer.py:168-183) assumes ids sorted by frequency,
            # which the reference's vocabulary does not guarantee either; only the uniform sampler is provided
            raise ValueError("only the `uniform` candidate sampler is available on this backend")
        self.all_args = locals()
        self.loss_type, self.norm_embed = loss_type, norm_embed
        self.n_epochs, self.lr, self.lr_decay, self.epsilon = n_epochs, lr, lr_decay, epsilon
        self.hidden_units = [*hidden_units_config(hidden_units), embed_size]      # youtube_retrieval.py:145
        self.reg = reg_config(reg)
        self.batch_size, self.use_bn = batch_size, use_bn
        self.dropout_rate = dropout_config(dropout_rate)
        self.num_sampled_per_batch, self.sampler, self.seed = num_sampled_per_batch, sampler, seed
        self.num_neg = None
        self.seq_mode, self.max_seq_len = check_seq_mode(recent_num, random_num)
        self.recent_seqs, self.recent_seq_lens = get_recent_seqs(self.n_users, self.user_consumed, self.n_items,
                                                                 self.max_seq_len)
        self.user_sparse = bool(data_info.user_sparse_col.name)
        self.user_dense = bool(data_info.user_dense_col.name)
        self.multi_sparse_combiner = check_multi_sparse(data_info, multi_sparse_combiner) if self.user_sparse else "normal"
        self._device_arg, self.dense_adam = device, dense_adam

    def build_model(self):
        self.device = hip_device(self._device_arg)
        spec = FeatSpec.from_data_info(self.data_info, self.multi_sparse_combiner)
        self.net = YouTubeRetrievalNet(self.n_items, spec, self.embed_size, self.hidden_units, self.use_bn,
                                       self.dropout_rate, self.norm_embed, self.max_seq_len, self.lr, self.epsilon,
                                       self.seed, self.device, self.dense_adam, self.loss_type,
                                       self.num_sampled_per_batch, reg=self.reg, batch_size=self.batch_size)

    def train_on_batch(self, b):
        self.apply_lr_schedule()
        return self.net.train_step(b.items, b.seqs.interacted_seq, b.sparse_indices, b.dense_values)

    # ---- embeddings (`bases/dyn_embed_base.py:240-269`) ------------------------------------------
    def set_embeddings(self):
        """Users: MLP output on the cached recent windows + stored user features, a column of ones appended
        (the item side carries `item_bias_var` there); items: `item_embeds_var` rows (+ bias column)."""
        d = self.data_info
        us = d.user_sparse_unique[:-1] if d.user_sparse_unique is not None else None
        ud = d.user_dense_unique[:-1] if d.user_dense_unique is not None else None
        ue = self.net.embed_users(self.recent_seqs[: self.n_users], us, ud)
        w, bias = self.net.item_matrix()
        self.user_embeds = torch.cat([ue, torch.ones_like(ue[:, :1])], dim=1).contiguous()
        self.item_embeds = torch.cat([w, bias.view(-1, 1)], dim=1).contiguous()

    def convert_array_id(self, user, inner_id):
        assert np.isscalar(user), f"User to convert must be scalar, got: {user}"
        if inner_id:
            if not isinstance(user, (int, np.integer)):
                raise ValueError(f"`inner id` user must be int, got {user}")
            return np.array([user if 0 <= user < self.n_users else self.n_users])
        return np.array([self.data_info.user2id.get(user, self.n_users)])

    def _window(self, uid, seq, inner_id):
        """[1, L] history window (`recommendation/preprocess.py:7-23,79-85`): the last L entries of `seq` (unknown
        items pruned) or of the user's consumed list; the OOV user has no history."""
        L, N = self.max_seq_len, self.n_items
        if seq is not None and len(seq) > 0:
            ids = list(seq) if inner_id else [self.data_info.item2id.get(i, N) for i in seq]
        elif uid != self.n_users:
            ids = list(self.user_consumed[uid])
        else:
            ids = []
        ids = [i if 0 <= i < N else N for i in ids[-min(L, len(ids)):]] if ids else []
        out = np.full((1, L), N, dtype=np.int32)
        out[0, : len(ids)] = ids
        return out

    def dyn_user_embedding(self, user, user_feats=None, seq=None, include_bias=False, inner_id=False):
        from ..feature_override import override_dense, override_sparse
        from ..recommendation import check_dynamic_rec_feats

        check_dynamic_rec_feats(self.model_name, user, user_feats, seq)
        d = self.data_info
        uid = int(self.convert_array_id(user, inner_id)[0])
        sp = de = None
        if d.user_sparse_unique is not None:
            sp = d.user_sparse_unique[[uid]]
            if user_feats:
                sp = override_sparse(d, sp, user_feats, d.user_sparse_col.name)
        if d.user_dense_unique is not None:
            de = d.user_dense_unique[[uid]]
            if user_feats:
                de = override_dense(d, de, user_feats, d.user_dense_col.name)
        vec = self.net.embed_users(self._window(uid, seq, inner_id), sp, de)[0].cpu().numpy()
        return np.append(vec, np.float32(1.0)) if include_bias else vec

    def recommend_user(self, user, n_rec, user_feats=None, seq=None, cold_start="average", inner_id=False,
                       filter_consumed=True, random_rec=False):
        if user_feats is None and seq is None:
            return super().recommend_user(user, n_rec, cold_start, inner_id, filter_consumed, random_rec)
        from ..recommendation import check_dynamic_rec_feats, recommend_from_embedding

        check_dynamic_rec_feats(self.model_name, user, user_feats, seq)
        vec = torch.from_numpy(self.dyn_user_embedding(user, user_feats, seq, include_bias=True,
                                                       inner_id=inner_id)).view(1, -1)
        uid = int(self.convert_array_id(user, inner_id)[0])
        recs = recommend_from_embedding(self, [uid], n_rec, None, self.item_embeds, filter_consumed, random_rec,
                                        user_vectors=vec)[0]
        return {user: recs if inner_id else np.array([self.data_info.id2item[i] for i in recs.tolist()])}

    # ---- persistence ----------------------------------------------------------------------------
    def _bn_layers(self):
        out, st = {}, self.net.mlp
        if st.bn_in is not None:
            out["mlp/bn_in"] = st.bn_in
        for i, bn in enumerate(st.bns, start=1):
            if bn is not None:
                out[f"mlp/bn{i}"] = bn
        return out

    def variables_np(self):
        t = self.net.tables
        out = {f"embedding/{k}": t.variable(k).cpu().numpy() for k in ("seq_embeds_var", "item_embeds_var")}
        if t.sparse_size:
            out["embedding/sparse_embeds_var"] = t.variable("sparse_embeds_var").cpu().numpy()
        out.update({k: p.detach().cpu().numpy() for k, p in self.net.P.params.items()})
        for k, bn in self._bn_layers().items():
            out[f"bn::{k}::mean"], out[f"bn::{k}::var"] = bn.moving_mean.cpu().numpy(), bn.moving_var.cpu().numpy()
        return out

    def optimizer_arrays(self):
        t, P = self.net.tables, self.net.P
        return {"opt::m": t.m.cpu().numpy(), "opt::v": t.v.cpu().numpy(), "opt::dense_m": P.m.cpu().numpy(),
                "opt::dense_v": P.v.cpu().numpy(), "opt::step": np.asarray(self.net.step, dtype=np.int64)}

    def rebuild_model(self, path, model_name, full_assign=True):
        """Retraining on merged data (`bases/meta.py:7-14` gives every TF model `rebuild_tf_model`; `tfops/rebuild.py:12-139`):
        a freshly built, larger model takes over the saved one's variables.  Items keep their inner ids (new ones are
        appended), so the rows of `seq_embeds_var` / `item_embeds_var` and the entries of `item_bias_var` — all three
        item-indexed, no OOV rows (youtube_retrieval.py:193-205, 243-257) — are copied 1:1; the user sparse table is re-based
        column by column (`training/rebuild.py:sparse_growth_index` == rebuild.py:63-73); rows of new items / categories keep
        the new model's initialisation; with `full_assign` the Adam moments follow the same map and the step counter is
        restored."""
        from ..training.rebuild import sparse_growth_index

        old = self.data_info.old_info
        if old is None:
            raise ValueError("`rebuild_model` needs a `data_info` produced by `merge_trainset`")
        self.build_model()
        self.model_built = True
        arrays = self._saved_arrays(path, model_name)
        t, P, dev = self.net.tables, self.net.P, self.device
        n_old = int(old.n_items)
        off_old = {"seq_embeds_var": 0, "item_embeds_var": n_old + 1, "sparse_embeds_var": 2 * n_old + 1}   # RetrievalTables layout
        off_new = {"seq_embeds_var": t.seq_off, "item_embeds_var": t.item_off, "sparse_embeds_var": t.sparse_off}
        with torch.no_grad():
            for name in ("seq_embeds_var", "item_embeds_var", "sparse_embeds_var"):
                key = f"embedding/{name}"
                if key not in arrays or arrays[key].shape[0] == 0:
                    continue
                a = arrays[key]
                if name == "sparse_embeds_var":
                    src, dst = sparse_growth_index(a.shape[0], old, self.data_info.sparse_offset)
                else:
                    src = dst = np.arange(n_old)
                t.variable(name)[torch.from_numpy(dst).to(dev)] = torch.from_numpy(a[src]).to(dev)
                if full_assign and "opt::m" in arrays:
                    for k, mom in (("opt::m", t.m), ("opt::v", t.v)):
                        mom[torch.from_numpy(off_new[name] + dst).to(dev)] = torch.from_numpy(arrays[k][off_old[name] + src]).to(dev)
            # dense parameters: equal shapes are copied; `item_bias_var [n_items]` grows with the catalogue — its first
            # n_old entries (and their moments) are the saved ones.  The flat moment buffers are addressed through each
            # parameter's offset in the flat storage (the offsets shift when item_bias_var grows).
            saved_off, off = {}, 0
            for k in P.params:                       # the saved flat layout: same parameter order, saved shapes
                if k in arrays:
                    saved_off[k] = off
                    off += -(-int(np.prod(arrays[k].shape)) // 4) * 4
            have_moments = full_assign and "opt::dense_m" in arrays and off == arrays["opt::dense_m"].shape[0]
            for k, p in P.params.items():
                if k not in arrays:
                    print(f'variable "{k}" is not in the saved model, will be skipped.')
                    continue
                a = arrays[k]
                if tuple(a.shape) == tuple(p.shape):
                    n_copy = a.size
                elif k == "embedding/item_bias_var" and a.ndim == 1 and a.shape[0] == n_old:
                    n_copy = n_old
                else:
                    print(f'old and new shape of variable "{k}" doesn\'t match, will be skipped.')
                    continue
                p.view(-1)[:n_copy] = torch.from_numpy(a.reshape(-1)[:n_copy]).to(dev)
                if have_moments:
                    o_new, o_old = p.storage_offset(), saved_off[k]
                    P.m[o_new:o_new + n_copy] = torch.from_numpy(arrays["opt::dense_m"][o_old:o_old + n_copy]).to(dev)
                    P.v[o_new:o_new + n_copy] = torch.from_numpy(arrays["opt::dense_v"][o_old:o_old + n_copy]).to(dev)
            for k, bn in self._bn_layers().items():
                if f"bn::{k}::mean" in arrays:
                    bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
            if full_assign and "opt::step" in arrays:
                self.net.step = int(arrays["opt::step"])

    def load_variables_np(self, arrays):
        t = self.net.tables
        with torch.no_grad():
            for k in ("seq_embeds_var", "item_embeds_var", "sparse_embeds_var"):
                if f"embedding/{k}" in arrays:
                    t.variable(k).copy_(torch.from_numpy(arrays[f"embedding/{k}"]))
            for k, p in self.net.P.params.items():
                if k in arrays:
                    p.copy_(torch.from_numpy(arrays[k]))
            for k, bn in self._bn_layers().items():
                if f"bn::{k}::mean" in arrays:
                    bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
