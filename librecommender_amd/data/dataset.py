"""`DatasetPure` / `DatasetFeat`: raw interaction frames -> index-encoded sets + `DataInfo`
(`libreco/data/dataset.py:207-259,386-545`).  Class-level state (vocabularies of the last
`build_trainset`) is kept like the reference: build the train set before eval/test sets."""
from __future__ import annotations

import numpy as np

from .consumed import interaction_consumed
from .data_info import DataInfo, MultiSparseInfo
from .transformed import TransformedEvalSet, TransformedSet
from .vocab import SparseSchema, encode, last_seen_rows


def _ordered_common(cols, members):
    """Columns of `cols` that are in `members`, in `cols` order (feature/column_mapping.py:45-52)."""
    return [c for c in cols if c in set(members)]


def column_families(user_col, item_col, sparse_col, dense_col):
    """{family: {column: position}} — `col_name2index` of the reference."""
    m = {}
    if sparse_col:
        m["sparse_col"] = {c: j for j, c in enumerate(sparse_col)}
    if dense_col:
        m["dense_col"] = {c: j for j, c in enumerate(dense_col)}
    for who, cols in (("user", user_col), ("item", item_col)):
        if cols and sparse_col:
            sub = _ordered_common(sparse_col, cols)
            if sub:
                m[f"{who}_sparse_col"] = {c: m["sparse_col"][c] for c in sub}
        if cols and dense_col:
            sub = _ordered_common(dense_col, cols)
            if sub:
                m[f"{who}_dense_col"] = {c: m["dense_col"][c] for c in sub}
    return m


class _Dataset:
    user_unique_vals = None
    item_unique_vals = None
    train_called = False

    @staticmethod
    def _check_col_names(data, is_train):
        cols = list(data.columns)
        if len(cols) < 2 or cols[0] != "user" or cols[1] != "item":
            raise ValueError("'user', 'item' must be the first two columns of the data")
        if is_train:
            assert "label" in data.columns, "train data should contain label column"

    @classmethod
    def _check_subclass(cls):
        if cls is _Dataset:
            raise NameError("Please use 'DatasetPure' or 'DatasetFeat' to call method")

    @staticmethod
    def shuffle_data(data, seed):
        return data.sample(frac=1, random_state=seed).reset_index(drop=True)

    @staticmethod
    def _labels(data):
        if "label" in data.columns:
            return data["label"].to_numpy(dtype=np.float32)
        return np.zeros(len(data), dtype=np.float32)   # dummy labels for label-less test data

    @classmethod
    def _encode_ids(cls, data, is_train):
        u = encode(data["user"].to_numpy(), cls.user_unique_vals, allow_unknown=not is_train)
        i = encode(data["item"].to_numpy(), cls.item_unique_vals, allow_unknown=not is_train)
        return u, i

    @classmethod
    def _build_eval(cls, data, shuffle, seed):
        if not cls.train_called:
            raise RuntimeError("Must first build trainset before building evalset or testset")
        cls._check_subclass()
        cls._check_col_names(data, is_train=False)
        if shuffle:
            data = cls.shuffle_data(data, seed)
        u, i = cls._encode_ids(data, is_train=False)
        return TransformedEvalSet(u, i, cls._labels(data))

    @classmethod
    def build_evalset(cls, eval_data, shuffle=False, seed=42):
        return cls._build_eval(eval_data, shuffle, seed)

    # ---- retrain (data/dataset.py:148-196): eval / test sets encoded against a merged DataInfo ----
    @classmethod
    def _merge_eval(cls, data, data_info, shuffle, seed):
        from .data_info import DataInfo
        assert isinstance(data_info, DataInfo), "Invalid passed `data_info`."
        cls._check_subclass()
        cls._check_col_names(data, is_train=False)
        if shuffle:
            data = cls.shuffle_data(data, seed)
        u = encode(data["user"].to_numpy(), data_info.user_unique_vals, allow_unknown=True)
        i = encode(data["item"].to_numpy(), data_info.item_unique_vals, allow_unknown=True)
        return TransformedEvalSet(u, i, cls._labels(data))

    @classmethod
    def merge_evalset(cls, eval_data, data_info, shuffle=False, seed=42):
        return cls._merge_eval(eval_data, data_info, shuffle, seed)

    @classmethod
    def merge_testset(cls, test_data, data_info, shuffle=False, seed=42):
        return cls._merge_eval(test_data, data_info, shuffle, seed)

    @classmethod
    def build_testset(cls, test_data, shuffle=False, seed=42):
        return cls._build_eval(test_data, shuffle, seed)


class DatasetPure(_Dataset):
    """Collaborative-filtering data: user, item, label (`data/dataset.py:197-259`)."""

    @classmethod
    def build_trainset(cls, train_data, shuffle=False, seed=42):
        cls._check_subclass()
        cls._check_col_names(train_data, is_train=True)
        cls.user_unique_vals = np.sort(train_data["user"].unique())
        cls.item_unique_vals = np.sort(train_data["item"].unique())
        if shuffle:
            train_data = cls.shuffle_data(train_data, seed)
        u, i = cls._encode_ids(train_data, is_train=True)
        user_consumed, item_consumed = interaction_consumed(u, i)
        info = DataInfo(interaction_data=train_data[["user", "item", "label"]],
                        user_consumed=user_consumed, item_consumed=item_consumed,
                        user_unique_vals=cls.user_unique_vals, item_unique_vals=cls.item_unique_vals,
                        seed=seed)
        cls.train_called = True
        return TransformedSet(u, i, cls._labels(train_data)), info


    @classmethod
    def merge_trainset(cls, train_data, data_info, merge_behavior=True, shuffle=False, seed=42):
        """New data on top of a previous `DataInfo` (`data/dataset.py:262-331`): returns the
        transformed new data and a NEW DataInfo (with `old_info` for `rebuild_model`)."""
        from .retrain import store_old_info, update_consumed, update_unique_vals
        assert isinstance(data_info, DataInfo), "Invalid passed `data_info`."
        cls._check_col_names(train_data, is_train=True)
        cls.user_unique_vals = update_unique_vals(np.unique(train_data["user"]), data_info.user_unique_vals)
        cls.item_unique_vals = update_unique_vals(np.unique(train_data["item"]), data_info.item_unique_vals)
        if shuffle:
            train_data = cls.shuffle_data(train_data, seed)
        u, i = cls._encode_ids(train_data, is_train=True)
        user_consumed, item_consumed = update_consumed(u, i, len(cls.user_unique_vals), len(cls.item_unique_vals),
                                                       data_info, merge_behavior)
        info = DataInfo(interaction_data=train_data[["user", "item", "label"]], user_consumed=user_consumed,
                        item_consumed=item_consumed, user_unique_vals=cls.user_unique_vals,
                        item_unique_vals=cls.item_unique_vals, seed=seed)
        info.old_info = store_old_info(data_info)
        cls.train_called = True
        return TransformedSet(u, i, cls._labels(train_data)), info


class DatasetFeat(_Dataset):
    """Data with sparse / dense / multi-sparse feature columns (`data/dataset.py:345-545`)."""

    schema: SparseSchema = None
    sparse_unique_vals = None
    multi_sparse_unique_vals = None
    sparse_col = None
    multi_sparse_col = None
    dense_col = None

    @classmethod
    def _check_feature_cols(cls, user_col, item_col):
        sparse = cls.schema.all_cols
        dense = cls.dense_col or []
        users, items = user_col or [], item_col or []
        if len(sparse) + len(dense) != len(users) + len(items):
            raise ValueError(
                "Please make sure length of columns match, i.e. `len(sparse_cols) + len(dense_cols) "
                f"== len(user_cols) + len(item_cols)`, got sparse columns: {sparse}, dense columns: "
                f"{dense}, user columns: {users}, item columns: {items}")
        odd = set(sparse + dense) ^ set(users + items)
        if odd:
            raise ValueError(f"Got inconsistent columns: {sorted(odd)}, please check the column names")

    @classmethod
    def build_trainset(cls, train_data, user_col=None, item_col=None, sparse_col=None,
                       dense_col=None, multi_sparse_col=None, unique_feat=False,
                       pad_val="missing", shuffle=False, seed=42):
        cls._check_subclass()
        cls._check_col_names(train_data, is_train=True)
        cls.sparse_col = list(sparse_col) if sparse_col else None
        cls.dense_col = list(dense_col) if dense_col else None
        if multi_sparse_col:
            nested = all(isinstance(f, list) for f in multi_sparse_col)
            cls.multi_sparse_col = list(multi_sparse_col) if nested else [list(multi_sparse_col)]
        else:
            cls.multi_sparse_col = None

        schema = SparseSchema(sparse_cols=cls.sparse_col or [], multi_fields=cls.multi_sparse_col or [])
        for c in schema.sparse_cols:
            schema.vocab[c] = np.sort(train_data[c].unique())
        if cls.multi_sparse_col:
            pads = pad_val if isinstance(pad_val, (list, tuple)) else [pad_val] * len(cls.multi_sparse_col)
            if len(pads) != len(cls.multi_sparse_col):
                raise ValueError("Length of `multi_sparse_col` and `pad_val` doesn't match")
            for f, pad in zip(cls.multi_sparse_col, pads):
                vals = set(train_data[f].to_numpy().ravel().tolist())
                vals.discard(pad)
                schema.multi_vocab[f[0]] = np.sort(list(vals))
                schema.pad_val[f[0]] = pad
        cls.schema = schema
        cls._check_feature_cols(user_col, item_col)
        cls.sparse_unique_vals = dict(schema.vocab) or None
        cls.multi_sparse_unique_vals = dict(schema.multi_vocab) or None

        cls.user_unique_vals = np.sort(train_data["user"].unique())
        cls.item_unique_vals = np.sort(train_data["item"].unique())
        if shuffle:
            train_data = cls.shuffle_data(train_data, seed)
        u, i = cls._encode_ids(train_data, is_train=True)
        sparse_indices = schema.encode_frame(train_data, is_train=True)
        dense_values = train_data[cls.dense_col].to_numpy(dtype=np.float32) if cls.dense_col else None

        families = column_families(user_col, item_col, schema.all_cols or None, cls.dense_col)
        n_u, n_i = len(cls.user_unique_vals), len(cls.item_unique_vals)
        uniq = {}
        for who, ids, n in (("user", u, n_u), ("item", i, n_i)):
            for kind, mat in (("sparse", sparse_indices), ("dense", dense_values)):
                fam = families.get(f"{who}_{kind}_col")
                uniq[f"{who}_{kind}"] = last_seen_rows(ids, mat, list(fam.values()), n) if fam else None

        multi_info = None
        if cls.multi_sparse_col:
            all_cols = schema.all_cols
            multi_info = MultiSparseInfo(
                field_offset=[all_cols.index(f[0]) for f in cls.multi_sparse_col],
                field_len=[len(f) for f in cls.multi_sparse_col],
                feat_oov=schema.field_oov_rows.copy(), pad_val=dict(schema.pad_val))
            families["multi_sparse"] = {c: f[0] for f in cls.multi_sparse_col for c in f[1:]}

        user_consumed, item_consumed = interaction_consumed(u, i)
        info = DataInfo(families, train_data[["user", "item", "label"]], uniq["user_sparse"],
                        uniq["user_dense"], uniq["item_sparse"], uniq["item_dense"], user_consumed,
                        item_consumed, cls.user_unique_vals, cls.item_unique_vals,
                        cls.sparse_unique_vals, schema.offsets, schema.oov_rows,
                        cls.multi_sparse_unique_vals, multi_info, seed)
        cls.train_called = True
        return TransformedSet(u, i, cls._labels(train_data), sparse_indices, dense_values), info

    @classmethod
    def merge_trainset(cls, train_data, data_info, merge_behavior=True, shuffle=False, seed=42):
        """New data on top of a previous `DataInfo` (`data/dataset.py:548-700`): ids, category
        vocabularies, sparse offsets / OOV rows and the unique feature matrices grow; known
        indices keep their position relative to their column's offset."""
        from .retrain import (merged_schema, store_old_info, update_consumed, update_unique_feats,
                              update_unique_vals)
        assert isinstance(data_info, DataInfo), "Invalid passed `data_info`."
        cls._check_col_names(train_data, is_train=True)
        cls.user_unique_vals = update_unique_vals(np.unique(train_data["user"]), data_info.user_unique_vals)
        cls.item_unique_vals = update_unique_vals(np.unique(train_data["item"]), data_info.item_unique_vals)
        schema = merged_schema(train_data, data_info)
        cls.schema = schema
        cls.sparse_col = list(schema.sparse_cols) or None
        cls.multi_sparse_col = [list(f) for f in schema.multi_fields] or None
        cls.dense_col = list(data_info.dense_col.name) or None
        cls.sparse_unique_vals = dict(schema.vocab) or None
        cls.multi_sparse_unique_vals = dict(schema.multi_vocab) or None
        if shuffle:
            train_data = cls.shuffle_data(train_data, seed)
        u, i = cls._encode_ids(train_data, is_train=True)
        sparse_indices = schema.encode_frame(train_data, is_train=True)
        dense_values = train_data[cls.dense_col].to_numpy(dtype=np.float32) if cls.dense_col else None
        usp, uds = update_unique_feats(train_data, data_info, cls.user_unique_vals, schema, is_user=True)
        isp, ids_ = update_unique_feats(train_data, data_info, cls.item_unique_vals, schema, is_user=False)
        multi_info = None
        if cls.multi_sparse_col:
            all_cols = schema.all_cols
            multi_info = MultiSparseInfo(
                field_offset=[all_cols.index(f[0]) for f in cls.multi_sparse_col],
                field_len=[len(f) for f in cls.multi_sparse_col],
                feat_oov=schema.field_oov_rows.copy(), pad_val=dict(schema.pad_val))
        user_consumed, item_consumed = update_consumed(u, i, len(cls.user_unique_vals), len(cls.item_unique_vals),
                                                       data_info, merge_behavior)
        info = DataInfo(data_info.col_name_mapping, train_data[["user", "item", "label"]], usp, uds, isp, ids_,
                        user_consumed, item_consumed, cls.user_unique_vals, cls.item_unique_vals,
                        cls.sparse_unique_vals, schema.offsets, schema.oov_rows,
                        cls.multi_sparse_unique_vals, multi_info, seed)
        info.old_info = store_old_info(data_info)
        cls.train_called = True
        return TransformedSet(u, i, cls._labels(train_data), sparse_indices, dense_values), info
