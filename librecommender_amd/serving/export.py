"""Export of trained models in the file layout the reference's serving tier reads
(`libserving/serialization/{common,embed,online}.py`): JSON id maps, consumed lists, embedding
vectors and feature tables.  The reference's online export additionally writes a TF SavedModel
(`online.py:84-110`); here the network variables go to `<model_name>_variables.npz` instead (the
same arrays `model.save` writes), to be served by this library's own `recommend_user`.

Only the file formats live here; the HTTP/Redis serving processes of the reference are out of
scope."""
from __future__ import annotations

import json
import os

import numpy as np


def _py(v):
    return v.item() if isinstance(v, np.generic) else v


def _dump(path, name, obj):
    with open(os.path.join(path, name), "w") as f:
        json.dump(obj, f, ensure_ascii=False)


def _ensure_dir(path):
    assert isinstance(path, str) and path, f"invalid saving path: `{path}`"
    os.makedirs(path, exist_ok=True)


def _save_common(path, model):
    """`common.py:13-36`: model name, raw<->inner id maps (JSON keys are strings), consumed lists."""
    info = model.data_info
    _dump(path, "model_name.json", {"model_name": model.model_name})
    _dump(path, "user2id.json", {str(int(k)): int(v) for k, v in info.user2id.items()})
    _dump(path, "id2item.json", {str(int(k)): int(v) for k, v in info.id2item.items()})
    _dump(path, "item2id.json", {str(int(k)): int(v) for k, v in info.item2id.items()})
    _dump(path, "user_consumed.json", {str(int(u)): [int(i) for i in items] for u, items in info.user_consumed.items()})


def _vectors(embeds, num):
    embeds = np.asarray(embeds)
    return {str(i): embeds[i].tolist() for i in range(num)}


def save_embed(path: str, model):
    """Embedding models (`embed.py:16-41`): `user_embed.json` / `item_embed.json` hold one vector per
    known id (the OOV row is not exported)."""
    _ensure_dir(path)
    _save_common(path, model)
    _dump(path, "user_embed.json", _vectors(model.user_embeds_np, model.n_users))
    _dump(path, "item_embed.json", _vectors(model.item_embeds_np, model.n_items))


def _feature_tables(info, model):
    """`common.py:45-71`: per-id feature rows incl. the OOV row."""
    feats = {"n_users": int(info.n_users), "n_items": int(info.n_items)}
    if info.col_name_mapping:
        for side, n in (("user", info.n_users), ("item", info.n_items)):
            for kind in ("sparse", "dense"):
                col = getattr(info, f"{side}_{kind}_col")
                if not col.name:
                    continue
                table = getattr(info, f"{side}_{kind}_unique")
                assert len(table) == n + 1, f"feature sizes don't match, got {len(table)} and {n + 1}"
                feats[f"{side}_{kind}_col_index"] = [int(i) for i in col.index]
                feats[f"{side}_{kind}_values"] = np.asarray(table).tolist()
    if hasattr(model, "max_seq_len"):
        feats["max_seq_len"] = int(model.max_seq_len)
    return feats


def _user_sparse_mapping(info):
    """`online.py:41-70`: user sparse column -> position among user sparse fields, and per column
    {raw value: global table row}."""
    fields, rows = {}, {}
    mapping = info.col_name_mapping
    for pos, col in enumerate(info.user_sparse_col.name):
        fields[col] = pos
        main = mapping.get("multi_sparse", {}).get(col, col)
        offset = int(info.sparse_offset[mapping["sparse_col"][col]])
        rows[col] = {str(_py(val)): int(idx) + offset for val, idx in info.sparse_idx_mapping[main].items()}
    return fields, rows


def save_online(path: str, model, version: int = 1):
    """Feature models scored online (`online.py:23-82`).  Returns the directory holding the network
    variables (`<path>/<model_name lower>/<version>/`)."""
    _ensure_dir(path)
    info = model.data_info
    _save_common(path, model)
    _dump(path, "features.json", _feature_tables(info, model))
    if info.col_name_mapping and info.user_sparse_col.name:
        fields, rows = _user_sparse_mapping(info)
        _dump(path, "user_sparse_fields.json", fields)
        _dump(path, "user_sparse_idx_mapping.json", rows)
    if info.col_name_mapping and info.user_dense_col.name:
        _dump(path, "user_dense_fields.json", {c: i for i, c in enumerate(info.user_dense_col.name)})
    export_dir = os.path.join(path, model.model_name.lower(), str(version))
    if os.path.isdir(export_dir):
        raise FileExistsError(f"Could not export model because '{export_dir}' already exists")
    os.makedirs(export_dir)
    model.save(export_dir, model.model_name.lower(), inference_only=True)
    return export_dir
