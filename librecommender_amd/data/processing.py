"""Frame-level preprocessing in front of `DatasetFeat` (`libreco/data/processing.py:10-181`): dense
column scaling with derived columns, and splitting delimiter-joined multi-value columns into the
padded sub-columns `multi_sparse_col` expects."""
from __future__ import annotations

import numpy as np

_DERIVED = (("log", np.log1p), ("sqrt", np.sqrt), ("square", np.square))


def _scaler(kind):
    from sklearn import preprocessing as sk
    table = {"min_max": sk.MinMaxScaler, "standard": sk.StandardScaler, "robust": sk.RobustScaler,
             "power": sk.PowerTransformer}
    try:
        return table[kind.lower()]()
    except KeyError:
        raise ValueError("unknown normalize type...") from None


def process_data(data, dense_col=None, normalizer="min_max", transformer=("log", "sqrt", "square")):
    """Scale `dense_col` in place (fitted on the first frame when a list/tuple is given, applied to the
    rest) and append `<col>_log / _sqrt / _square` for columns without negative values.  Returns
    `(data, dense column names including the derived ones)`."""
    if not isinstance(dense_col, list):
        raise ValueError("dense_col must be a list...")
    scaler = _scaler(normalizer)
    many = isinstance(data, (list, tuple))
    names = list(dense_col)
    for i, frame in enumerate(data if many else [data]):
        scaled = scaler.fit_transform(frame[dense_col]) if i == 0 else scaler.transform(frame[dense_col])
        frame[dense_col] = scaled.astype(np.float32) if many else scaled
        for col in dense_col:
            if frame[col].min() < 0.0:
                print("can't transform negative values...")
                continue
            for tag, fn in _DERIVED:
                if transformer is not None and tag in transformer:
                    frame[f"{col}_{tag}"] = fn(frame[col])
                    if i == 0:
                        names.append(f"{col}_{tag}")
    return data, names


def split_multi_value(data, multi_value_col, sep, max_len=None, pad_val="missing", user_col=None,
                      item_col=None):
    """`"a|b|c"`-style columns -> `col_1 .. col_n` sub-columns padded with `pad_val`.  Returns
    `(data, multi_sparse_col, user_sparse_col, item_sparse_col)`."""
    if max_len is not None:
        assert isinstance(max_len, (list, tuple)), "`max_len` must be list or tuple"
        assert len(max_len) == len(multi_value_col), "`max_len` must have same length as `multi_value_col`"
    pads = list(pad_val) if isinstance(pad_val, (list, tuple)) else [pad_val] * len(multi_value_col)
    assert len(multi_value_col) == len(pads), "length of `multi_sparse_col` and `pad_val` doesn't match"
    groups, user_side, item_side = [], [], []
    for j, col in enumerate(multi_value_col):
        cleaned = data[col].str.strip(sep + " ").str.replace("\\s+", "", regex=True).str.lower()
        data[col] = cleaned.mask(cleaned == "", pads[j])
        parts = data[col].str.split(sep)
        width = int(parts.str.len().max()) if max_len is None else max_len[j]
        subs = [f"{col}_{i + 1}" for i in range(width)]
        for i, name in enumerate(subs):
            data[name] = parts.str.get(i).fillna(pads[j])
        groups.append(subs)
        if user_col is not None and col in user_col:
            user_side.extend(subs)
        elif item_col is not None and col in item_col:
            item_side.extend(subs)
    return data.fillna(pads[0]).drop(multi_value_col, axis=1), groups, user_side, item_side
