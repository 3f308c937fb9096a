"""Base of the feature models FM / DeepFM / DIN (the reference's `TfBase`,
`libreco/bases/tf_base.py:28-416`): scores come from a full model forward, so `recommend_user`
scores the whole catalog through the net in device-side chunks (`recommend_tf_feat`,
`recommendation/recommend.py:81-105`) and ranks on the device."""
from __future__ import annotations

import os

import numpy as np
import torch

from ..feature_override import override_dense, override_sparse
from ..prediction.predict import convert_id, normalize_prediction
from ..recommendation import check_dynamic_rec_feats, cold_start_rec, construct_rec
from ..recommendation.recommend import random_select_device
from ..utils.validate import check_unknown, check_unknown_user
from .base import Base


def merge_user_item_feats(data_info, users, items):
    """[n, Fs] sparse rows / [n, Fd] dense values of (user, item) pairs in original column order
    (`prediction/preprocess.py:15-61`, `_extract_feats` :171-190)."""
    d = data_info
    users, items = np.asarray(users), np.asarray(items)
    out = []
    for kind, total, uc, ic, um, im, dt in (
            ("sparse", len(d.sparse_col.name), d.user_sparse_col.index, d.item_sparse_col.index,
             d.user_sparse_unique, d.item_sparse_unique, np.int32),
            ("dense", len(d.dense_col.name), d.user_dense_col.index, d.item_dense_col.index,
             d.user_dense_unique, d.item_dense_unique, np.float32)):
        if total == 0:
            out.append(None)
            continue
        m = np.zeros((len(users), total), dtype=dt)
        if uc:
            m[:, uc] = um[users]
        if ic:
            m[:, ic] = im[items]
        out.append(m)
    return out[0], out[1]


class FeatBase(Base):
    uses_features = True
    score_chunk = 1 << 18       # (user, item) pairs per forward during full-catalog scoring

    def __init__(self, task, data_info, lower_upper_bound=None):
        super().__init__(task, data_info, lower_upper_bound)
        self.net = None

    # ---- hooks used by the trainer ------------------------------------------------------------
    def on_epoch_end(self, epoch):
        pass

    def prepare_for_eval(self):
        pass

    def _seq_args(self, b):
        return {}

    def train_on_batch(self, b):
        self.apply_lr_schedule()
        loss = self.net.train_step(b.users, b.items, b.labels, sparse=b.sparse_indices,
                                   dense=b.dense_values, loss_type=self._loss_name(), **self._seq_args(b))
        # a replayed hipGraph returns its static output tensor (overwritten by the next step): keep this step's value
        graphed = getattr(self.net, "_use_graph", False) or (getattr(self.net, "_fstep", None) is not None
                                                             and getattr(self.net, "graph_step", False))
        if getattr(loss, "_lr_own", False):          # already this step's own copy (made on the replay stream)
            return loss
        return loss.clone() if graphed else loss

    def _loss_name(self):
        return "mse" if self.task == "rating" else self.loss_type

    def after_fit(self):
        self.assign_oov()
        self.default_recs = self._recommend_inner([self.n_users], min(2000, self.n_items), None, None,
                                                  filter_consumed=False, random_rec=False)[0]

    def assign_oov(self):
        self.net.emb.assign_oov(self.data_info.sparse_oov) if hasattr(self.net, "emb") else \
            self.net.assign_oov(self.data_info.sparse_oov)

    # ---- scoring ------------------------------------------------------------------------------
    def _cached_seq(self, users):
        return None, None

    def _forward(self, users, items, sparse, dense, seqs=None, seq_lens=None):
        return self.net.forward(users, items, sparse=sparse, dense=dense, seqs=seqs, seq_lens=seq_lens)

    def predict(self, user, item, feats=None, cold_start="average", inner_id=False):
        user, item = convert_id(self, user, item, inner_id)
        unknown_num, unknown_index, user, item = check_unknown(self, user, item)
        sparse, dense = merge_user_item_feats(self.data_info, user, item)
        if feats is not None:
            assert isinstance(feats, dict), "`feats` must be `dict`."
            assert len(user) == 1, "Predict with feats only supports single user."
            sparse = override_sparse(self.data_info, sparse, feats) if sparse is not None else None
            dense = override_dense(self.data_info, dense, feats) if dense is not None else None
        seqs, lens = self._cached_seq(user)
        preds = self._forward(user, item, sparse, dense, seqs, lens).cpu().numpy()
        return normalize_prediction(preds, self, cold_start, unknown_num, unknown_index)

    def _item_side_device(self):
        """Per-item feature rows (`item_sparse_unique` / `item_dense_unique`, one row per catalogue item) resident on
        the device; rebuilt when `data_info` swaps them (retrain / rebuild)."""
        d = self.data_info
        key = (id(d.item_sparse_unique), id(d.item_dense_unique), getattr(d, "feat_version", 0))
        c = getattr(self, "_item_side", None)
        if c is None or c[0] != key:
            sp = None if d.item_sparse_unique is None or not d.item_sparse_col.index else \
                torch.as_tensor(np.ascontiguousarray(d.item_sparse_unique), device=self.device).to(torch.int32)
            dn = None if d.item_dense_unique is None or not d.item_dense_col.index else \
                torch.as_tensor(np.ascontiguousarray(d.item_dense_unique), device=self.device, dtype=torch.float32)
            c = self._item_side = (key, sp, dn)
        return c[1], c[2]

    def _scores_all_items(self, uid, user_feats, seq):
        """[n_items] scores of one user against the whole catalog (`recommend_tf_feat`, recommendation/recommend.py:81-105
        + preprocess.py:110-172).  The (user, item) feature rows are assembled ON THE DEVICE per chunk — the user's side
        (one host row, temporary `user_feats` applied) broadcast over the chunk, the item side sliced from the resident
        per-item rows, the user's sequence as a stride-0 view — so a chunk costs its kernels, not host concatenations
        and a 50 MB upload."""
        d, dev, N = self.data_info, self.device, self.n_items
        sp1, dn1 = merge_user_item_feats(d, [uid], [0])
        item_sp, item_dn = self._item_side_device()
        forced = []                       # (kind, column positions, values) a `user_feats` override pins for every row
        if user_feats is not None:
            for kind, row, fn in (("sparse", sp1, override_sparse), ("dense", dn1, override_dense)):
                if row is None:
                    continue
                new = fn(d, row, user_feats)
                cols = np.nonzero((new != row)[0])[0]
                if len(cols):
                    forced.append((kind, torch.as_tensor(cols, device=dev), torch.as_tensor(new[0, cols], device=dev)))
                if kind == "sparse":
                    sp1 = new
                else:
                    dn1 = new
        sp_row = None if sp1 is None else torch.as_tensor(sp1, device=dev).to(torch.int32)
        dn_row = None if dn1 is None else torch.as_tensor(dn1, device=dev, dtype=torch.float32)
        ic_sp = torch.as_tensor(list(d.item_sparse_col.index), device=dev, dtype=torch.long) if item_sp is not None else None
        ic_dn = torch.as_tensor(list(d.item_dense_col.index), device=dev, dtype=torch.long) if item_dn is not None else None
        seqs1, lens1 = self._seq_for(uid, seq)
        seq_t = None if seqs1 is None else torch.as_tensor(np.ascontiguousarray(seqs1), device=dev).to(torch.int32)
        len_t = None if lens1 is None else torch.as_tensor(np.ascontiguousarray(lens1), device=dev).to(torch.int32)
        out = torch.empty(N, dtype=torch.float32, device=dev)
        for s in range(0, N, self.score_chunk):
            n = min(N, s + self.score_chunk) - s
            items = torch.arange(s, s + n, device=dev, dtype=torch.int32)
            users = torch.full((n,), int(uid), device=dev, dtype=torch.int32)
            sparse = dense = None
            if sp_row is not None:
                sparse = sp_row.expand(n, -1).clone()
                if ic_sp is not None:
                    sparse[:, ic_sp] = item_sp[s:s + n]
            if dn_row is not None:
                dense = dn_row.expand(n, -1).clone()
                if ic_dn is not None:
                    dense[:, ic_dn] = item_dn[s:s + n]
            for kind, cols, vals in forced:
                tgt = sparse if kind == "sparse" else dense
                tgt[:, cols] = vals.to(tgt.dtype)
            seqs = None if seq_t is None else seq_t.expand(n, -1)
            lens = None if len_t is None else len_t.expand(n, *len_t.shape[1:])     # [n] (or [n, 2]: SIM)
            out[s:s + n] = self._forward(users, items, sparse, dense, seqs, lens)
        return out

    def _seq_for(self, uid, seq):
        return None, None

    def _catalog_scorer(self):
        """Factorised full-catalog scorer for the models whose embedding-facing layers
        are (bi)linear in user-side + item-side fields: FM and DeepFM.  DIN (attention over
        (sequence, item) pairs) keeps the chunked forward."""
        net = self.net
        if getattr(self, "_dist", None) is not None:          # row-sharded tables: the chunked forward (a collective)
            return None
        if not (hasattr(net, "linear") and (hasattr(net, "pair_dense") or hasattr(net, "out"))):
            return None
        if getattr(net, "mlp_dtype", torch.float32) != torch.float32:
            return None
        sc = getattr(self, "_scorer", None)
        if sc is None or sc.net is not net:
            from ..recommendation.catalog import CatalogScorer
            sc = self._scorer = CatalogScorer(self)
        return sc

    def _recommend_inner(self, user_ids, n_rec, user_feats, seq, filter_consumed, random_rec):
        if n_rec > self.n_items:
            raise ValueError(f"`n_rec` {n_rec} exceeds num of items {self.n_items}")
        recs = []
        scorer = self._catalog_scorer() if seq is None else None
        ub = max(1, (1 << 29) // max(1, self.n_items * 4))          # users per [B, N] score block
        block, block_start = None, 0
        if scorer is not None and not random_rec:
            # whole blocks of users at once: one scatter of the consumed ids (CSR of the block) and one top-k
            for s0 in range(0, len(user_ids), ub):
                uids = list(user_ids[s0:s0 + ub])
                block = scorer.scores(uids, user_feats)
                ptr, cidx, _ = self.consumed_index.batch_csr(uids, n_rec, self.n_items, filter_consumed, self.device)
                lens = ptr[1:] - ptr[:-1]
                if int(ptr[-1]) > 0:
                    rows = torch.repeat_interleave(torch.arange(len(uids), device=self.device), lens)
                    block[rows, cidx[: rows.numel()].long()] = float("-inf")
                recs.append(torch.topk(block, n_rec, dim=1, sorted=True).indices.cpu().numpy())
            return np.concatenate(recs, axis=0)
        for pos, uid in enumerate(user_ids):
            if scorer is not None:
                if block is None or pos >= block_start + block.shape[0]:
                    block_start = pos
                    block = scorer.scores(list(user_ids[pos:pos + ub]), user_feats)
                scores = block[pos - block_start]
            else:
                scores = self._scores_all_items(uid, user_feats, seq)
            consumed = self.consumed_index.consumed(uid)
            n_hist = self.consumed_index.hist_len[int(uid)] if 0 <= int(uid) < len(self.consumed_index.hist_len) else 0
            banned = None
            if filter_consumed and consumed is not None and n_hist > 0 and n_rec + n_hist <= self.n_items:
                banned = torch.zeros(self.n_items, dtype=torch.bool, device=self.device)
                banned[torch.from_numpy(consumed.astype(np.int64)).to(self.device)] = True
            if random_rec:
                ids = random_select_device(scores.view(1, -1), None if banned is None else banned.view(1, -1), n_rec)[0]
            else:
                if banned is not None:
                    scores = scores.masked_fill(banned, float("-inf"))
                ids = torch.topk(scores, n_rec, sorted=True).indices
            recs.append(ids.cpu().numpy())
        return np.stack(recs)

    def recommend_user(self, user, n_rec, user_feats=None, seq=None, cold_start="average",
                       inner_id=False, filter_consumed=True, random_rec=False):
        check_dynamic_rec_feats(self.model_name, user, user_feats, seq)
        if user_feats is None and seq is None:
            out = {}
            known, unknown = check_unknown_user(self.data_info, user, inner_id)
            if unknown:
                out.update(cold_start_rec(self.data_info, self.default_recs, cold_start, unknown, n_rec, inner_id))
            if known:
                recs = self._recommend_inner(known, n_rec, None, None, filter_consumed, random_rec)
                out.update(construct_rec(self.data_info, known, recs, inner_id))
            return out
        # dynamic features / sequence: single user; unknown users map to the padding id
        if inner_id:
            if not isinstance(user, (int, np.integer)):
                raise ValueError(f"`inner id` user must be int, got {user}")
            uid = user if 0 <= user < self.n_users else self.n_users
        else:
            uid = self.data_info.user2id.get(user, self.n_users)
        recs = self._recommend_inner([uid], n_rec, user_feats, seq, filter_consumed, random_rec)[0]
        return {user: recs if inner_id else np.array([self.data_info.id2item[i] for i in recs.tolist()])}

    # ---- persistence ----------------------------------------------------------------------------
    def state_arrays(self):
        if getattr(self, "_dist", None) is not None:
            raise NotImplementedError("row-sharded tables are checkpointed per shard: `model.net.tables.save_shard(path)` "
                                      "(parallel.py), dense parameters from `model.net.P`")
        t = self.net.tables
        out = {"embed": t.embed.cpu().numpy()}
        if t.lin is not None:
            out["lin"] = t.lin.cpu().numpy()
        out.update({f"dense::{k}": p.detach().cpu().numpy() for k, p in self.net.P.params.items()})
        for k, bn in self._batch_norms().items():
            out[f"bn::{k}::mean"], out[f"bn::{k}::var"] = bn.moving_mean.cpu().numpy(), bn.moving_var.cpu().numpy()
        return out

    def optimizer_arrays(self):
        t, P = self.net.tables, self.net.P
        out = {"opt::m": t.m.cpu().numpy(), "opt::v": t.v.cpu().numpy(),
               "opt::dense_m": P.m.cpu().numpy(), "opt::dense_v": P.v.cpu().numpy(),
               "opt::step": np.asarray(self.net.step, dtype=np.int64)}
        if t.lin is not None:
            out["opt::lin_m"], out["opt::lin_v"] = t.lin_m.cpu().numpy(), t.lin_v.cpu().numpy()
        return out

    def rebuild_model(self, path, model_name, full_assign=True):
        """Take over the variables of a saved (smaller) model before retraining on merged data
        (`bases/tf_base.py` -> `tfops/rebuild.py:12-139`): `self.data_info` must come from
        `DatasetFeat/DatasetPure.merge_trainset` (it carries `old_info`).  `full_assign` also
        restores the Adam moments and step count."""
        from ..training.rebuild import table_growth_index
        old = self.data_info.old_info
        if old is None:
            raise ValueError("`rebuild_model` needs a `data_info` produced by `merge_trainset`")
        self.build_model()
        self.model_built = True
        if getattr(self, "_dist", None) is not None:
            return self._rebuild_sharded(path, model_name, full_assign)
        arrays = self._saved_arrays(path, model_name)
        t, P, dev = self.net.tables, self.net.P, self.device
        src, dst = table_growth_index(arrays["embed"].shape[0], old, self.data_info)
        src_t, dst_t = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
        pairs = [("embed", t.embed), ("lin", t.lin)]
        if full_assign:
            pairs += [("opt::m", t.m), ("opt::v", t.v), ("opt::lin_m", getattr(t, "lin_m", None)),
                      ("opt::lin_v", getattr(t, "lin_v", None))]
        with torch.no_grad():
            for key, new in pairs:
                if new is None or key not in arrays:
                    continue
                new[dst_t] = torch.from_numpy(arrays[key]).to(dev)[src_t]
            all_match = True
            for k, p in P.params.items():
                a = arrays.get(f"dense::{k}")
                if a is None or tuple(a.shape) != tuple(p.shape):
                    all_match = False
                    print(f'old and new shape of variable "{k}" doesn\'t match, will be skipped.')
                    continue
                p.copy_(torch.from_numpy(a))
            for k, bn in self._batch_norms().items():
                if f"bn::{k}::mean" in arrays and arrays[f"bn::{k}::mean"].shape == tuple(bn.moving_mean.shape):
                    bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
            if full_assign and "opt::step" in arrays:
                if all_match and arrays["opt::dense_m"].shape == tuple(P.m.shape):
                    P.m.copy_(torch.from_numpy(arrays["opt::dense_m"]))
                    P.v.copy_(torch.from_numpy(arrays["opt::dense_v"]))
                self.net.step = int(arrays["opt::step"])

    def _rebuild_sharded(self, path, model_name, full_assign):
        """`rebuild_model` under a process group (round 4): the saved model is a per-shard checkpoint (`distributed.save_sharded`,
        any world size); its rows go through the same growth map as in one process, every rank copying only the rows it owns
        out of the memory-mapped old shards; the replicated dense parameters are taken over when their shapes still match."""
        import os

        from ..training.rebuild import table_growth_index

        old, t, P = self.data_info.old_info, self.net.tables, self.net.P
        # the old table's row count is read from the checkpoint itself
        import glob

        f0 = sorted(glob.glob(os.path.join(path, f"{model_name}_tables_shard0of*.npz")))
        if not f0:
            raise FileNotFoundError(f"no {model_name}_tables_shard*of*.npz under {path} (a checkpoint written under a process group)")
        with np.load(f0[0]) as z:
            V_old = int(z["V"])
        src, dst = table_growth_index(V_old, old, self.data_info)
        keys = ["embed"] + (["lin"] if t.lin is not None else [])
        if full_assign:
            keys += ["m", "v"] + (["lin_m", "lin_v"] if t.lin is not None else [])
        t.load_shards_mapped(path, f"{model_name}_tables", src, dst, V_old, keys)
        with np.load(os.path.join(path, f"{model_name}_replicated.npz")) as z:
            if "flat" in z and z["flat"].shape[0] == P.flat.numel() and ("names" not in z or list(z["names"]) == list(P.params)):
                with torch.no_grad():
                    P.flat.copy_(torch.from_numpy(z["flat"]))
                    if full_assign:
                        P.m.copy_(torch.from_numpy(z["m"]))
                        P.v.copy_(torch.from_numpy(z["v"]))
                from .. import distributed as D

                for k, bn in D._batch_norms(self.net).items():
                    if f"bn::{k}::mean" in z:
                        bn.moving_mean.copy_(torch.from_numpy(z[f"bn::{k}::mean"]))
                        bn.moving_var.copy_(torch.from_numpy(z[f"bn::{k}::var"]))
            else:
                print("old and new dense parameters do not match, they keep their initialisation.")
            if full_assign:
                self.net.step = int(z["step"])

    def _batch_norms(self):
        out = {}
        for name in ("mlp", "first_stage_mlp", "second_stage_mlp"):      # the `dense_nn` stacks of the feature models
            mlp = getattr(self.net, name, None)
            if mlp is not None:
                if mlp.bn_in is not None:
                    out[f"{name}/bn_in"] = mlp.bn_in
                for i, bn in enumerate(mlp.bns, start=1):
                    if bn is not None:
                        out[f"{name}/bn{i}"] = bn
        if getattr(self.net, "bn", None) is not None:
            out["bn"] = self.net.bn
        return out

    def _tf_layout(self):
        from collections import OrderedDict
        t = self.net.tables
        shapes = OrderedDict((k, tuple(p.shape)) for k, p in self.net.P.params.items())
        bns = [(k, bn.gamma, bn.beta, bn.moving_mean.numel()) for k, bn in self._batch_norms().items()]
        rows = {"user": t.n_users + 1, "item": t.n_items + 1, "sparse": t.V - t.n_users - t.n_items - 2}
        return shapes, bns, rows, t.lin is not None

    def save(self, path, model_name, manual=True, inference_only=False):
        """`bases/tf_base.py:360-386`, same positional order.  `manual` chooses between numpy arrays and a TF checkpoint in
        the reference; variables are always written as arrays here (`<model_name>_variables.npz`), so it has no effect."""
        return super().save(path, model_name, inference_only=inference_only)

    @classmethod
    def load(cls, path, model_name, data_info, manual=True):
        """`bases/tf_base.py:388-424`."""
        return super().load(path, model_name, data_info)

    def save_tf_variables(self, path, model_name):
        """Write `<model_name>_tf_variables.npz` under the reference's graph-variable names (inference
        variables: trainable + BatchNorm moving statistics), next to the hyper-parameter json `save` writes."""
        from ..utils.tf_checkpoint import to_tf_variables
        os.makedirs(path, exist_ok=True)
        shapes, bns, rows, with_lin = self._tf_layout()
        np.savez_compressed(os.path.join(path, f"{model_name}_tf_variables"),
                            **to_tf_variables(self.state_arrays(), shapes, bns, rows, with_lin))

    def load_tf_variables(self, path, model_name):
        """Take over the variables of a reference TF model saved with `manual=True`
        (`<model_name>_tf_variables.npz`, `utils/save_load.py:70-80`); the model must have been
        constructed with the same hyper-parameters and `DataInfo`.  See `utils/tf_checkpoint.py` for
        what the name mapping rests on."""
        from ..utils.tf_checkpoint import map_tf_variables, read_tf_variables
        if not self.model_built:
            self.build_model()
            self.model_built = True
        shapes, bns, rows, with_lin = self._tf_layout()
        self.load_state_arrays(map_tf_variables(read_tf_variables(path, model_name), shapes, bns, rows, with_linear=with_lin))
        default = os.path.join(path, f"{model_name}_default_recs.npz")
        if os.path.exists(default):
            self.default_recs = np.load(default)["default_recs"]

    def load_state_arrays(self, arrays):
        t = self.net.tables
        with torch.no_grad():
            t.embed.copy_(torch.from_numpy(arrays["embed"]))
            if t.lin is not None and "lin" in arrays:
                t.lin.copy_(torch.from_numpy(arrays["lin"]))
            for k, p in self.net.P.params.items():
                if f"dense::{k}" in arrays:
                    p.copy_(torch.from_numpy(arrays[f"dense::{k}"]))
            for k, bn in self._batch_norms().items():
                bn.moving_mean.copy_(torch.from_numpy(arrays[f"bn::{k}::mean"]))
                bn.moving_var.copy_(torch.from_numpy(arrays[f"bn::{k}::var"]))
