"""`DataInfo`: everything the models read about the training data
(`libreco/data/data_info.py:107-433`).  Attribute names follow the reference so that model code
and user code written against it keep working."""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List

import numpy as np

Feature = namedtuple("Feature", ["name", "index"])
EmptyFeature = Feature(name=[], index=[])


@dataclass
class MultiSparseInfo:
    """Layout of multi-sparse fields (`data/data_info.py:24-52`)."""

    field_offset: Iterable[int]   # position of each field's first column among all sparse columns
    field_len: Iterable[int]      # number of columns per field
    feat_oov: np.ndarray          # OOV row of each field in the shared sparse table
    pad_val: Dict[str, Any]


class DataInfo:
    def __init__(self, col_name_mapping=None, interaction_data=None, user_sparse_unique=None,
                 user_dense_unique=None, item_sparse_unique=None, item_dense_unique=None,
                 user_consumed=None, item_consumed=None, user_unique_vals=None,
                 item_unique_vals=None, sparse_unique_vals=None, sparse_offset=None,
                 sparse_oov=None, multi_sparse_unique_vals=None, multi_sparse_combine_info=None,
                 seed=42):
        self.all_args = dict(locals())            # as passed in (no OOV rows): what `save` writes
        self.all_args.pop("self", None)
        self.col_name_mapping = col_name_mapping
        self.interaction_data = interaction_data
        self.user_consumed = user_consumed
        self.item_consumed = item_consumed
        self.user_unique_vals = user_unique_vals
        self.item_unique_vals = item_unique_vals
        self.sparse_unique_vals = sparse_unique_vals
        self.sparse_offset = sparse_offset
        self.sparse_oov = sparse_oov
        self.multi_sparse_unique_vals = multi_sparse_unique_vals
        self.multi_sparse_combine_info = multi_sparse_combine_info
        self.sparse_idx_mapping = self._value_index_maps()
        self.np_rng = np.random.default_rng(seed)
        self.seed = seed
        self.old_info = None
        self._cache: Dict[str, Any] = {}
        # unique feature matrices get one extra OOV row (`add_oovs`, data_info.py:399-413)
        self.user_sparse_unique = self._with_oov_row(user_sparse_unique, self.user_sparse_col.index, True)
        self.item_sparse_unique = self._with_oov_row(item_sparse_unique, self.item_sparse_col.index, True)
        self.user_dense_unique = self._with_oov_row(user_dense_unique, None, False)
        self.item_dense_unique = self._with_oov_row(item_dense_unique, None, False)

    # ---- construction helpers ---------------------------------------------------------------
    def _value_index_maps(self):
        if self.sparse_unique_vals is None and self.multi_sparse_unique_vals is None:
            return None
        out = {}
        for src in (self.sparse_unique_vals, self.multi_sparse_unique_vals):
            for col, vals in (src or {}).items():
                out[col] = {v: j for j, v in enumerate(vals.tolist() if hasattr(vals, "tolist") else vals)}
        return out

    def _with_oov_row(self, mat, cols, sparse: bool):
        if mat is None:
            return None
        if sparse:   # sparse features: each column's OOV row; dense: the column mean
            oov = np.asarray(self.sparse_oov)[list(cols)]
        else:
            oov = np.mean(mat, axis=0)
        return np.vstack([mat, oov])

    def _family(self, key: str) -> Feature:
        m = self.col_name_mapping
        if not m or key not in m:
            return EmptyFeature
        return Feature(name=list(m[key].keys()), index=list(m[key].values()))

    # ---- feature families ----------------------------------------------------------------------
    sparse_col = property(lambda self: self._family("sparse_col"))
    dense_col = property(lambda self: self._family("dense_col"))
    user_sparse_col = property(lambda self: self._family("user_sparse_col"))
    user_dense_col = property(lambda self: self._family("user_dense_col"))
    item_sparse_col = property(lambda self: self._family("item_sparse_col"))
    item_dense_col = property(lambda self: self._family("item_dense_col"))

    @property
    def user_col(self) -> List[str]:
        return self.user_sparse_col.name + self.user_dense_col.name

    @property
    def item_col(self) -> List[str]:
        return self.item_sparse_col.name + self.item_dense_col.name

    # ---- sizes / id maps -------------------------------------------------------------------------
    @property
    def n_users(self) -> int:
        return len(self.user_unique_vals)

    @property
    def n_items(self) -> int:
        return len(self.item_unique_vals)

    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def user2id(self):
        return self._cached("user2id", lambda: {u: j for j, u in enumerate(self.user_unique_vals.tolist())})

    @property
    def item2id(self):
        return self._cached("item2id", lambda: {i: j for j, i in enumerate(self.item_unique_vals.tolist())})

    @property
    def id2user(self):
        return self._cached("id2user", lambda: dict(enumerate(self.user_unique_vals.tolist())))

    @property
    def id2item(self):
        return self._cached("id2item", lambda: dict(enumerate(self.item_unique_vals.tolist())))

    @property
    def data_size(self) -> int:
        return len(self.interaction_data)

    @property
    def global_mean(self):
        return self.interaction_data.label.mean()

    @property
    def min_max_rating(self):
        return self.interaction_data.label.min(), self.interaction_data.label.max()

    @property
    def popular_items(self):
        """Up to 100 most-interacted raw item ids, distinct users counted (data_info.py:415-433)."""
        def compute(num=100):
            pairs = self.interaction_data.drop_duplicates(subset=["user", "item"])
            counts = pairs.groupby("item")["user"].count().sort_values(ascending=False)
            chosen = counts.index.tolist()[:num]
            if len(chosen) < num and self.old_info is not None:
                chosen.extend(self.old_info.popular_items[: num - len(chosen)])
            return chosen
        if getattr(self, "_popular_items", None) is None:      # same private cache slot as the reference (reset = None)
            self._popular_items = compute()
        return self._popular_items

    def __repr__(self):
        n_u, n_i, n = self.n_users, self.n_items, len(self.interaction_data)
        return "n_users: %d, n_items: %d, data density: %.4f %%" % (n_u, n_i, 100 * n / (n_u * n_i))

    # ---- in-place feature refresh (data_info.py:330-397, feature/update.py:179-228) -------------
    def _assign_features(self, data, key, is_user):
        from .vocab import encode
        self.feat_version = getattr(self, "feat_version", 0) + 1   # device-side copies of the rows are keyed on it
        assert key in data.columns, f"Data must contain `{key}` column."
        data = data.drop_duplicates(subset=[key], keep="last")
        ids = self.user_unique_vals if is_user else self.item_unique_vals
        rows_all = encode(data[key].to_numpy(), ids, allow_unknown=True)
        known_id = rows_all < len(ids)                      # unknown users / items are skipped
        sp_info = self.user_sparse_col if is_user else self.item_sparse_col
        ds_info = self.user_dense_col if is_user else self.item_dense_col
        sp = self.user_sparse_unique if is_user else self.item_sparse_unique
        ds = self.user_dense_unique if is_user else self.item_dense_unique
        multi_map = (self.col_name_mapping or {}).get("multi_sparse", {})
        if sp is not None:
            for j, (col, ci) in enumerate(zip(sp_info.name, sp_info.index)):
                if col not in data.columns:                 # features absent from the new data stay
                    continue
                if col in multi_map:
                    vocab = self.multi_sparse_unique_vals[multi_map[col]]
                elif self.multi_sparse_unique_vals and col in self.multi_sparse_unique_vals:
                    vocab = self.multi_sparse_unique_vals[col]
                else:
                    vocab = self.sparse_unique_vals[col]
                idx = encode(data[col].to_numpy(), vocab, allow_unknown=True)
                ok = known_id & (idx < len(vocab))           # unknown category values are skipped
                sp[rows_all[ok], j] = self.sparse_offset[ci] + idx[ok]
        if ds is not None:
            for j, col in enumerate(ds_info.name):
                if col in data.columns:
                    ds[rows_all[known_id], j] = data[col].to_numpy(np.float32)[known_id]

    def assign_user_features(self, user_data):
        """Refresh the stored feature rows of known users from `user_data` (last occurrence wins;
        unknown users, unknown categories and missing columns are ignored)."""
        self._assign_features(user_data, "user", True)

    def assign_item_features(self, item_data):
        self._assign_features(item_data, "item", False)

    # ---- persistence: the reference's on-disk layout (data_info.py:435-541) --------------------
    def save(self, path, model_name):
        """`{model_name}_data_info.npz` + `_data_info_name_mapping.json` + `_user/_item_consumed.pkl`,
        key for key what the reference writes, so files written by either side load on the other
        (`multi_sparse_combine_info` is a pickled dataclass: see `load`)."""
        import json
        import pickle
        from pathlib import Path

        path = Path(path)
        if not path.is_dir():
            print(f"file folder {path} doesn't exists, creating a new one...")
            path.mkdir(parents=True)
        if self.col_name_mapping is not None:
            with open(path / f"{model_name}_data_info_name_mapping.json", "w") as f:
                json.dump(self.all_args["col_name_mapping"], f, separators=(",", ":"), indent=4)
        for kind in ("user_consumed", "item_consumed"):
            if getattr(self, kind) is not None:
                with open(path / f"{model_name}_{kind}.pkl", "wb") as f:
                    pickle.dump(getattr(self, kind), f, protocol=pickle.HIGHEST_PROTOCOL)
        hp = {}
        for arg, val in self.all_args.items():
            if arg in ("col_name_mapping", "user_consumed", "item_consumed") or val is None:
                continue
            if arg == "interaction_data":
                hp[arg] = val.to_numpy()
            elif arg == "sparse_unique_vals":
                hp.update({"unique_" + str(c): np.asarray(v) for c, v in val.items()})
            elif arg == "multi_sparse_unique_vals":
                hp.update({"munique_" + str(c): np.asarray(v) for c, v in val.items()})
            else:
                hp[arg] = val
        np.savez_compressed(path / f"{model_name}_data_info", **hp)

    @classmethod
    def load(cls, path, model_name):
        import json
        import pickle
        import sys
        import types
        from pathlib import Path

        import pandas as pd

        path = Path(path)
        if not path.exists():
            raise OSError(f"file folder {path} doesn't exists...")
        hp = {}
        nm = path / f"{model_name}_data_info_name_mapping.json"
        if nm.exists():
            with open(nm) as f:
                hp["col_name_mapping"] = json.load(f)
        for kind in ("user_consumed", "item_consumed"):
            fp = path / f"{model_name}_{kind}.pkl"
            if fp.exists():
                with open(fp, "rb") as f:
                    hp[kind] = pickle.load(f)
        # a file written by the reference pickles `libreco.data.data_info.MultiSparseInfo`; without
        # that package installed the name is aliased to the dataclass of the same fields here
        alias = None
        if "libreco.data.data_info" not in sys.modules:
            alias = types.ModuleType("libreco.data.data_info")
            alias.MultiSparseInfo = MultiSparseInfo
            for name in ("libreco", "libreco.data"):
                sys.modules.setdefault(name, types.ModuleType(name))
            sys.modules["libreco.data.data_info"] = alias
        try:
            info = dict(np.load(path / f"{model_name}_data_info.npz", allow_pickle=True).items())
        finally:
            if alias is not None:
                for name in ("libreco.data.data_info", "libreco.data", "libreco"):
                    if isinstance(sys.modules.get(name), types.ModuleType) and not hasattr(sys.modules[name], "__file__"):
                        sys.modules.pop(name, None)
        for arg, val in info.items():
            if arg == "interaction_data":
                hp[arg] = pd.DataFrame(val, columns=["user", "item", "label"])
            elif arg in ("multi_sparse_combine_info", "seed"):
                v = val.item()
                if arg == "multi_sparse_combine_info" and not isinstance(v, MultiSparseInfo):
                    v = MultiSparseInfo(v.field_offset, v.field_len, v.feat_oov, v.pad_val)
                hp[arg] = v
            elif arg.startswith("unique_"):
                hp.setdefault("sparse_unique_vals", {})[arg[7:]] = val
            elif arg.startswith("munique_"):
                hp.setdefault("multi_sparse_unique_vals", {})[arg[8:]] = val
            else:
                hp[arg] = val
        return cls(**hp)


def __getattr__(name):
    """`OldInfo` / `store_old_info` live in `data/retrain.py` here; the reference keeps them in this module
    (`data/data_info.py:540-578`) and code written against it imports them from here."""
    if name in ("OldInfo", "store_old_info"):
        from . import retrain

        return getattr(retrain, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
