"""`TwoTower` (`libreco/algorithms/two_tower.py`): same constructor, same errors, same
`fit / predict / recommend_user`, on the MI355X hot path."""
from __future__ import annotations

import numpy as np
import torch

from ..bases import EmbedBase
from ..bases.base import hip_device
from ..batch.batch_unit import PairwiseBatch
from ..nets import TwoTowerNet
from ..utils.validate import dropout_config, hidden_units_config, reg_config, sparse_feat_size


class TwoTower(EmbedBase):
    uses_features = True

    def __init__(self, task, data_info=None, loss_type="softmax", embed_size=16, norm_embed=False,
                 n_epochs=20, lr=0.001, lr_decay=False, epsilon=1e-5, reg=None, batch_size=256,
                 sampler="random", num_neg=1, use_bn=True, dropout_rate=None,
                 hidden_units=(128, 64, 32), margin=1.0, use_correction=True, temperature=1.0,
                 remove_accidental_hits=False, ssl_pattern=None, alpha=0.2, seed=42,
                 tf_sess_config=None, device="cuda", device_sampling=False):
        super().__init__(task, data_info, embed_size)
        self.all_args = locals()
        self.device_sampling = device_sampling      # row f1: permutation, negatives, collation on the device
        self.loss_type, self.norm_embed = loss_type, norm_embed
        self.n_epochs, self.lr, self.lr_decay, self.epsilon = n_epochs, lr, lr_decay, epsilon
        self.reg = reg_config(reg)
        self.batch_size, self.sampler, self.num_neg = batch_size, sampler, num_neg
        self.use_bn = use_bn
        self.dropout_rate = dropout_config(dropout_rate)
        self.hidden_units = hidden_units_config(hidden_units)
        self.margin, self.use_correction, self.temperature = margin, use_correction, temperature
        self.remove_accidental_hits = remove_accidental_hits
        self.ssl_pattern, self.alpha, self.seed = ssl_pattern, alpha, seed
        self.user_sparse = bool(data_info.user_sparse_col.name)
        self.item_sparse = bool(data_info.item_sparse_col.name)
        self.user_dense = bool(data_info.user_dense_col.name)
        self.item_dense = bool(data_info.item_dense_col.name)
        self._device_arg = device
        self.item_corrections = None
        self._check_params()

    def _check_params(self):
        if self.task != "ranking":
            raise ValueError("`TwoTower` is only suitable for ranking")
        if self.loss_type not in ("cross_entropy", "max_margin", "softmax"):
            raise ValueError(f"Unsupported `loss_type`: `{self.loss_type}`")
        if self.ssl_pattern is not None:
            if self.ssl_pattern not in ("rfm", "rfm-complementary", "cfm"):
                raise ValueError("`ssl` pattern supports `rfm`, `rfm-complementary` and `cfm`, "
                                 f"got `{self.ssl_pattern}`")
            if not self.item_sparse:
                raise ValueError("`ssl`(self-supervised learning) relies on item sparse features, "
                                 "which are not available in training data.")
            if self.loss_type != "softmax":
                raise ValueError("`ssl`(self-supervised learning) can only be used in `softmax` loss.")

    def build_model(self):
        from .. import distributed as D

        d = self.data_info
        n_sparse_rows = sparse_feat_size(d) if (self.user_sparse or self.item_sparse) else 0
        self._dist = D.active()
        if self._dist is not None:
            # one process per GPU: tables row-sharded over the ranks, global in-batch softmax, sharded export
            from ..nets import ShardedTwoTowerNet

            self.device = D.device_for(self._device_arg)
            n_dense = len(d.dense_col.name) if (self.user_dense or self.item_dense) else 0
            # one table: [users + OOV | items | sparse feature rows | one row per dense column | (ssl) one pad row]
            self._row_off = {"user": 0, "item": self.n_users + 1, "sparse": self.n_users + 1 + self.n_items,
                             "dense": self.n_users + 1 + self.n_items + n_sparse_rows}
            n_pad = 1 if self.ssl_pattern is not None else 0      # stands in for the zero row of the reference's ssl table
            self._row_off["pad"] = self._row_off["dense"] + n_dense
            self.net = ShardedTwoTowerNet(
                self.n_users + 1 + self.n_items + n_sparse_rows + n_dense + n_pad, 1 + len(d.user_sparse_col.name),
                1 + len(d.item_sparse_col.name), self.embed_size, self.hidden_units, self.use_bn, self.norm_embed, self.lr,
                self.epsilon, self.seed, self.device, self.margin, self.temperature, self.use_correction,
                self.remove_accidental_hits, kern=D.kernels(),
                user_dense_cols=d.user_dense_col.index if self.user_dense else (),
                item_dense_cols=d.item_dense_col.index if self.item_dense else (),
                dense_row0=self._row_off["dense"], dropout_rate=self.dropout_rate or 0.0,
                pad_row=self._row_off["pad"] if n_pad else None)
            return
        self.device = hip_device(self._device_arg)
        self.net = TwoTowerNet(
            self.n_users, self.n_items, n_sparse_rows, len(d.user_sparse_col.name),
            len(d.item_sparse_col.name), d.user_dense_col.index, d.item_dense_col.index,
            len(d.dense_col.name), self.embed_size, self.hidden_units, self.use_bn, self.dropout_rate,
            self.norm_embed, self.lr, self.epsilon, self.seed, self.device, self.margin,
            self.temperature, self.use_correction, self.remove_accidental_hits)

    def fit(self, train_data, neg_sampling, verbose=1, shuffle=True, eval_data=None, metrics=None,
            k=10, eval_batch_size=8192, eval_user_num=None, num_workers=0):
        if self.loss_type == "softmax" and self.use_correction:
            # sampling-bias correction Q(item) = count / len(train) (two_tower.py:425-435)
            if not self.data_info.old_info:
                _, counts = np.unique(train_data.item_indices, return_counts=True)
                assert len(counts) == self.n_items
                self.item_corrections = counts / len(train_data)
            else:   # retrain: only the new interactions are in `train_data`; unseen items keep Q = 1
                self.item_corrections = np.ones(self.n_items, dtype=np.float32)
                seen, counts = np.unique(train_data.item_indices, return_counts=True)
                self.item_corrections[seen] = counts / len(train_data)
        if self.ssl_pattern == "cfm":           # two_tower.py:437-438
            from ..feature_ssl import get_mutual_info
            self.sparse_feat_mutual_info = get_mutual_info(train_data, self.data_info)
        # `num_workers` is dropped like in the reference (quirk 2 of SURVEY §8)
        super().fit(train_data, neg_sampling, verbose, shuffle, eval_data, metrics, k, eval_batch_size,
                    eval_user_num)

    def _global_rows(self, ids, sparse, side):
        """[n, 1 + n_sparse] global table rows of one side's id + sparse feature columns."""
        dev = self.device
        cols = [torch.as_tensor(np.asarray(ids) if not isinstance(ids, torch.Tensor) else ids, device=dev).to(torch.int32).view(-1, 1)
                + self._row_off[side]]
        if sparse is not None:
            sp = torch.as_tensor(np.asarray(sparse) if not isinstance(sparse, torch.Tensor) else sparse, device=dev).to(torch.int32)
            cols.append(sp + self._row_off["sparse"])
        return torch.cat(cols, dim=1).contiguous()

    def takes_next_batch(self) -> bool:
        """Under a process group the trainer announces the next batch: its exchange plan is built a step ahead
        (`ShardedFieldTables.prefetch`), except with self-supervised views or dense columns (the net assembles the id block
        itself then)."""
        net = getattr(self, "net", None)
        return (getattr(self, "_dist", None) is not None and self.ssl_pattern is None
                and not (getattr(net, "ud_cols", None) or getattr(net, "id_cols", None)))     # (dense columns: the same)

    def _rank_inputs(self, b):
        """This rank's contiguous slice of the (identical on every rank) batch as (positional args, keyword args) of
        `ShardedTwoTowerNet.train_step`, `idx` = the packed id block [user rows | item rows (| negative rows)] the exchange
        plan is built from — or None when the batch has fewer samples than ranks.  The inputs of the batch announced one step
        earlier are re-used (the prefetched plan is recognised by the identity of its id tensor)."""
        from .. import distributed as D

        held = getattr(self, "_held_inputs", None)
        if held is not None and held[0] is b:
            return held[1]
        rank, world = self._dist
        sp, de = b.sparse_indices, b.dense_values
        def dv(name, sl):          # this rank's slice of one side's dense feature values
            x = getattr(de, name, None)
            if x is None:
                return None
            x = D.take(x, sl)
            return (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(device=self.device, dtype=torch.float32)

        if isinstance(b, PairwiseBatch):
            sl = D.batch_slice(len(b.queries), rank, world)
            if sl.stop == sl.start:
                return None
            u = self._global_rows(D.take(b.queries, sl), D.take(getattr(sp, "query_feats", None), sl), "user")
            i = self._global_rows(D.take(b.item_pairs[0], sl), D.take(getattr(sp, "item_pos_feats", None), sl), "item")
            n = self._global_rows(D.take(b.item_pairs[1], sl), D.take(getattr(sp, "item_neg_feats", None), sl), "item")
            kw = dict(item_neg_idx=n, user_dense=dv("query_feats", sl), item_dense=dv("item_pos_feats", sl),
                      item_dense_neg=dv("item_neg_feats", sl), idx=torch.cat([u, i, n], dim=1).contiguous())
            return ("max_margin", u, i), kw
        sl = D.batch_slice(len(b.users), rank, world)
        if sl.stop == sl.start:
            return None
        items = D.take(b.items, sl)
        corr = None
        if self.loss_type == "softmax" and self.use_correction:
            it_np = items.cpu().numpy() if isinstance(items, torch.Tensor) else np.asarray(items)
            corr = torch.as_tensor(np.asarray(self.item_corrections)[it_np], dtype=torch.float32, device=self.device)
        u = self._global_rows(D.take(b.users, sl), D.take(getattr(sp, "user_feats", None), sl), "user")
        i = self._global_rows(items, D.take(getattr(sp, "item_feats", None), sl), "item")
        kw = dict(labels=D.take(b.labels, sl), items=torch.as_tensor(np.asarray(items) if not isinstance(items, torch.Tensor) else items),
                  corrections=corr, user_dense=dv("user_feats", sl), item_dense=dv("item_feats", sl),
                  idx=torch.cat([u, i], dim=1).contiguous())
        return (self.loss_type, u, i), kw

    def _train_on_batch_sharded(self, b, next_batch=None):
        """This rank's contiguous slice of the (identical on every rank) batch through `ShardedTwoTowerNet`."""
        from .. import distributed as D

        if self.ssl_pattern is not None:        # batch/tf_feed_dicts.py:131-133, feature/ssl.py:6-40
            # every rank holds the same generator state and draws the views of the WHOLE batch (the draw one rank would
            # make), then keeps its slice: index j of the ssl table [zero row | items | sparse rows] is global row
            # item offset + j - 1 (the two blocks are adjacent in the sharded table too), j == 0 the masked column
            from ..feature_ssl import get_ssl_features
            rank, world = self._dist
            sl = D.batch_slice(len(b.users), rank, world)
            left, right, dense = get_ssl_features(self, len(b.items))
            def view(x):
                j = torch.as_tensor(np.ascontiguousarray(x[sl]), device=self.device).to(torch.int32)
                return torch.where(j > 0, j - 1 + self._row_off["item"], torch.full_like(j, -1))
            ssl = dict(ssl_left=view(left), ssl_right=view(right), alpha=self.alpha,
                       ssl_dense=None if dense is None else torch.as_tensor(np.asarray(dense[sl]), dtype=torch.float32, device=self.device))
            cur = self._rank_inputs(b)
            if cur is None:            # fewer samples than ranks (a tiny last batch): every rank skips the step
                return torch.zeros((), device=self.device)
            kw = dict(cur[1])
            kw.pop("idx")              # (with views the net assembles the id block itself)
            return self.net.train_step(*cur[0], **kw, **ssl)
        cur = self._rank_inputs(b)
        self._held_inputs = None
        if not self.takes_next_batch():          # dense columns: no packed id block from here, plans are built in line
            if cur is None:
                return torch.zeros((), device=self.device)
            kw = dict(cur[1])
            kw.pop("idx")
            return self.net.train_step(*cur[0], **kw)
        nxt = None
        if next_batch is not None:
            nxt = self._rank_inputs(next_batch)
            if nxt is not None:
                self._held_inputs = (next_batch, nxt)
        if cur is None:                # fewer samples than ranks (a tiny last batch): every rank skips the step
            return torch.zeros((), device=self.device)
        return self.net.train_step(*cur[0], **cur[1], next_idx=None if nxt is None else nxt[1]["idx"])

    def train_on_batch(self, b, next_batch=None):
        self.apply_lr_schedule()
        if getattr(self, "_dist", None) is not None:
            return self._train_on_batch_sharded(b, next_batch)
        if isinstance(b, PairwiseBatch):
            sp, de = b.sparse_indices, b.dense_values
            return self.net.train_step(
                "max_margin", b.queries, b.item_pairs[0], items_neg=b.item_pairs[1],
                user_sparse=getattr(sp, "query_feats", None), item_sparse=getattr(sp, "item_pos_feats", None),
                item_sparse_neg=getattr(sp, "item_neg_feats", None),
                user_dense=getattr(de, "query_feats", None), item_dense=getattr(de, "item_pos_feats", None),
                item_dense_neg=getattr(de, "item_neg_feats", None))
        sp, de = b.sparse_indices, b.dense_values
        corr = None
        if self.loss_type == "softmax" and self.use_correction:
            if isinstance(b.items, torch.Tensor):          # device loader: the correction table lives on the device too
                if getattr(self, "_corr_dev", None) is None or self._corr_src is not self.item_corrections:
                    self._corr_dev = torch.as_tensor(np.asarray(self.item_corrections), dtype=torch.float32, device=b.items.device)
                    self._corr_src = self.item_corrections
                corr = self._corr_dev[b.items.long()]
            else:
                corr = self.item_corrections[b.items]
        ssl = {}
        if self.ssl_pattern is not None:        # batch/tf_feed_dicts.py:131-133, feature/ssl.py:6-40
            from ..feature_ssl import get_ssl_features
            left, right, dense = get_ssl_features(self, len(b.items))
            ssl = dict(ssl_left=left, ssl_right=right, ssl_dense=dense, alpha=self.alpha)
        return self.net.train_step(
            self.loss_type, b.users, b.items, labels=b.labels,
            user_sparse=getattr(sp, "user_feats", None), item_sparse=getattr(sp, "item_feats", None),
            user_dense=getattr(de, "user_feats", None), item_dense=getattr(de, "item_feats", None),
            corrections=corr, **ssl)

    def set_embeddings(self):
        """User / item tower outputs for every known id (`dyn_embed_base.py:240-269`); the user
        table's OOV row first becomes the mean user row (`_assign_user_oov`)."""
        if getattr(self, "_dist", None) is not None:
            return self._set_embeddings_sharded()
        d, t = self.data_info, self.net.tables
        with torch.no_grad():
            uv = t.variable("user_embeds_var")
            uv[self.n_users] = uv[: self.n_users].mean(dim=0)
        us = d.user_sparse_unique[:-1] if d.user_sparse_unique is not None else None
        ud = d.user_dense_unique[:-1] if d.user_dense_unique is not None else None
        its = d.item_sparse_unique[:-1] if d.item_sparse_unique is not None else None
        itd = d.item_dense_unique[:-1] if d.item_dense_unique is not None else None
        self.user_embeds = self.net.embed_users(np.arange(self.n_users), us, ud).contiguous()
        self.item_embeds = self.net.embed_items(np.arange(self.n_items), its, itd).contiguous()

    def _set_embeddings_sharded(self):
        """Sharded export: every rank computes the tower outputs of ITS block of users / items through
        the tables' lookup collective; user embeddings are all-gathered (n_users x D is small), item embeddings stay
        sharded (`distributed.ShardedItemEmbeds`) and are served by `parallel.sharded_score_topk`."""
        from .. import distributed as D

        rank, world = self._dist
        d, net = self.data_info, self.net
        # OOV user row := mean user row (`_assign_user_oov`): owners contribute their rows, one all-reduce
        t = net.tables
        from ..parallel import allreduce_sum_

        with torch.no_grad():
            loc_rows = torch.arange(t.rank, t.V, t.world, device=self.device)[: t.embed.shape[0]]
            is_user = loc_rows < self.n_users
            s = t.embed[is_user].double().sum(dim=0)
            allreduce_sum_(s, net.group)
            if self.n_users % world == rank:
                t.embed[self.n_users // world] = (s / max(self.n_users, 1)).float()
        # side features (round 4): the towers' inputs are [id row, stored sparse feature rows of the id]
        ue_loc, _ = D.blockwise_tower(net, "user", self.n_users, self._row_off["user"], rank, world,
                                      sparse_unique=d.user_sparse_unique, sparse_offset=self._row_off["sparse"],
                                      dense_unique=d.user_dense_unique if self.user_dense else None)
        self.user_embeds = D.all_gather_rows(ue_loc, self.n_users, net.group)
        ie_loc, per = D.blockwise_tower(net, "item", self.n_items, self._row_off["item"], rank, world,
                                        sparse_unique=d.item_sparse_unique, sparse_offset=self._row_off["sparse"],
                                        dense_unique=d.item_dense_unique if self.item_dense else None)
        base = rank * per
        self.item_embeds = D.ShardedItemEmbeds(ie_loc, self.n_items, base, max(0, min(per, self.n_items - base)), net.group,
                                               net.kern)

    # ---- dynamic inference (`bases/dyn_embed_base.py:74-238`) -------------------------------------
    def convert_array_id(self, user, inner_id):
        """One raw (or inner) user id -> `[inner id]`; unknown users map to the OOV id
        (`bases/dyn_embed_base.py:60-72`)."""
        assert np.isscalar(user), f"User to convert must be scalar, got: {user}"
        if inner_id:
            if not isinstance(user, (int, np.integer)):
                raise ValueError(f"`inner id` user must be int, got {user}")
            return np.array([user if 0 <= user < self.n_users else self.n_users])
        return np.array([self.data_info.user2id.get(user, self.n_users)])

    def dyn_user_embedding(self, user, user_feats=None, seq=None, include_bias=False, inner_id=False):
        """User-tower output for ONE user with optional feature overrides (numpy, like the reference)."""
        from ..feature_override import override_dense, override_sparse
        from ..recommendation import check_dynamic_rec_feats

        check_dynamic_rec_feats(self.model_name, user, user_feats, seq)
        d = self.data_info
        uid = int(self.convert_array_id(user, inner_id)[0])
        if not 0 <= uid <= self.n_users:
            uid = self.n_users
        sp = de = None
        if d.user_sparse_unique is not None:
            sp = d.user_sparse_unique[[uid]]
            if user_feats:
                sp = override_sparse(d, sp, user_feats, d.user_sparse_col.name)
        if d.user_dense_unique is not None:
            de = d.user_dense_unique[[uid]]
            if user_feats:
                de = override_dense(d, de, user_feats, d.user_dense_col.name)
        return self.net.embed_users(np.asarray([uid]), sp, de)[0].cpu().numpy()

    def recommend_user(self, user, n_rec, user_feats=None, seq=None, cold_start="average",
                       inner_id=False, filter_consumed=True, random_rec=False):
        if user_feats is None and seq is None:
            return super().recommend_user(user, n_rec, cold_start, inner_id, filter_consumed, random_rec)
        from ..recommendation import check_dynamic_rec_feats, recommend_from_embedding

        check_dynamic_rec_feats(self.model_name, user, user_feats, seq)
        vec = torch.from_numpy(self.dyn_user_embedding(user, user_feats, seq, inner_id=inner_id)).view(1, -1)
        uid = int(self.convert_array_id(user, inner_id)[0])
        recs = recommend_from_embedding(self, [uid], n_rec, None, self.item_embeds, filter_consumed,
                                        random_rec, user_vectors=vec)[0]
        return {user: recs if inner_id else np.array([self.data_info.id2item[i] for i in recs.tolist()])}

    def variables_np(self):
        t = self.net.tables
        out = {f"embedding/{k}": t.variable(k).cpu().numpy() for k in ("user_embeds_var", "item_embeds_var")}
        if t.sparse_size:
            out["embedding/sparse_embeds_var"] = t.variable("sparse_embeds_var").cpu().numpy()
        out.update({k: p.detach().cpu().numpy() for k, p in self.net.P.params.items()})
        for k, bn in self._bn_layers().items():     # moving statistics (non-trainable TF variables)
            out[f"bn::{k}::mean"], out[f"bn::{k}::var"] = bn.moving_mean.cpu().numpy(), bn.moving_var.cpu().numpy()
        return out

    def _bn_layers(self):
        out = {}
        for tower in ("user_tower", "item_tower"):
            st = getattr(self.net, tower)
            if st.bn_in is not None:
                out[f"{tower}/bn_in"] = st.bn_in
            for i, bn in enumerate(st.bns, start=1):
                if bn is not None:
                    out[f"{tower}/bn{i}"] = bn
        return out

    def optimizer_arrays(self):
        t, P = self.net.tables, self.net.P
        return {"opt::m": t.m.cpu().numpy(), "opt::v": t.v.cpu().numpy(), "opt::dense_m": P.m.cpu().numpy(),
                "opt::dense_v": P.v.cpu().numpy(), "opt::step": np.asarray(self.net.step, dtype=np.int64)}

    def rebuild_model(self, path, model_name, full_assign=True):
        """`tfops/rebuild.py:12-139` for the two-tower variables (`item_embeds_var` has no OOV row,
        two_tower.py:266-271)."""
        from ..training.rebuild import sparse_growth_index
        old = self.data_info.old_info
        if old is None:
            raise ValueError("`rebuild_model` needs a `data_info` produced by `merge_trainset`")
        self.build_model()
        self.model_built = True
        arrays = self._saved_arrays(path, model_name)
        t, P, dev = self.net.tables, self.net.P, self.device
        with torch.no_grad():
            for kind, n_old in (("user", old.n_users), ("item", old.n_items), ("sparse", None)):
                key = f"embedding/{kind}_embeds_var"
                if key not in arrays:
                    continue
                a = arrays[key]
                if kind == "sparse":
                    src, dst = sparse_growth_index(a.shape[0], old, self.data_info.sparse_offset)
                else:
                    src = dst = np.arange(n_old)
                view = t.variable(f"{kind}_embeds_var")
                view[torch.from_numpy(dst).to(dev)] = torch.from_numpy(a[src]).to(dev)
                if full_assign and "opt::m" in arrays:          # saved moments use the old concatenated layout
                    off_old = {"user": 0, "item": old.n_users + 1, "sparse": old.n_users + 1 + old.n_items}[kind]
                    off_new = {"user": t.user_off, "item": t.item_off, "sparse": t.sparse_off}[kind]
                    for name, new in (("opt::m", t.m), ("opt::v", t.v)):
                        new[torch.from_numpy(off_new + dst).to(dev)] = torch.from_numpy(arrays[name][off_old + src]).to(dev)
            all_match = True
            for k, p in P.params.items():
                if k in arrays and tuple(arrays[k].shape) == tuple(p.shape):
                    p.copy_(torch.from_numpy(arrays[k]))
                else:
                    all_match = False
                    print(f'old and new shape of variable "{k}" doesn\'t match, will be skipped.')
            for k, bn in self._bn_layers().items():
                if f"bn::{k}::mean" in arrays:
                    bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
            if full_assign and "opt::step" in arrays:
                if all_match and arrays["opt::dense_m"].shape == tuple(P.m.shape):
                    P.m.copy_(torch.from_numpy(arrays["opt::dense_m"]))
                    P.v.copy_(torch.from_numpy(arrays["opt::dense_v"]))
                self.net.step = int(arrays["opt::step"])

    def load_variables_np(self, arrays):
        t = self.net.tables
        with torch.no_grad():
            for k in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var"):
                if f"embedding/{k}" in arrays:
                    t.variable(k).copy_(torch.from_numpy(arrays[f"embedding/{k}"]))
            for k, p in self.net.P.params.items():
                if k in arrays:
                    p.copy_(torch.from_numpy(arrays[k]))
            for k, bn in self._bn_layers().items():
                if f"bn::{k}::mean" in arrays:
                    bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
