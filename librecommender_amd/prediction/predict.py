"""Pointwise prediction (`libreco/prediction/predict.py:18-40`, `preprocess.py:6-12`)."""
import numpy as np
import torch

from .. import ops
from ..utils.validate import check_unknown


def convert_id(model, user, item, inner_id=False):
    user = [user] if np.isscalar(user) else user
    item = [item] if np.isscalar(item) else item
    if not inner_id:
        info = model.data_info
        user = [info.user2id.get(u, model.n_users) for u in user]
        item = [info.item2id.get(i, model.n_items) for i in item]
    return np.asarray(user), np.asarray(item)


def normalize_prediction(preds, model, cold_start, unknown_num, unknown_index):
    if model.task == "rating":
        preds = np.clip(preds, model.lower_bound, model.upper_bound)
    elif model.task == "ranking":
        from scipy.special import expit

        preds = expit(preds)
    if unknown_num > 0 and cold_start == "popular":
        if isinstance(preds, np.ndarray):
            preds[unknown_index] = model.default_pred
        else:
            preds = model.default_pred
    return preds


def predict_from_embedding(model, user, item, cold_start, inner_id):
    user, item = convert_id(model, user, item, inner_id)
    unknown_num, unknown_index, user, item = check_unknown(model, user, item)
    dev = model.user_embeds.device
    u = torch.as_tensor(user.astype(np.int32), device=dev)
    i = torch.as_tensor(item.astype(np.int32), device=dev)
    if not isinstance(model.item_embeds, torch.Tensor):       # sharded item embeddings: rows fetched from their owners
        rows = model.item_embeds.rows(i)
        preds = (model.user_embeds.index_select(0, u.long()) * rows).sum(dim=1).cpu().numpy()
    else:
        preds = ops.pair_dot(model.user_embeds, model.item_embeds, u, i).cpu().numpy()
    return normalize_prediction(preds, model, cold_start, unknown_num, unknown_index)


def predict_data_with_feats(model, data, batch_size=None, cold_start="average", inner_id=False):
    """Predict every row of a frame holding `user`, `item` and ALL feature columns, with the features
    taken from the frame instead of the stored per-id rows (`prediction/predict.py:95-150`)."""
    import pandas as pd

    from .preprocess import features_from_batch
    assert isinstance(data, pd.DataFrame), "Data must be pandas DataFrame"
    user, item = convert_id(model, data.user.tolist(), data.item.tolist(), inner_id)
    unknown_num, unknown_index, user, item = check_unknown(model, user, item)
    batch_size = batch_size or len(data)
    preds = np.zeros(len(data), dtype=np.float32)
    info = model.data_info
    for s in range(0, len(data), batch_size):
        sl = slice(s, s + batch_size)
        sparse, dense = features_from_batch(info, bool(info.sparse_col.name), bool(info.dense_col.name),
                                            data.iloc[sl])
        seqs, lens = model._cached_seq(user[sl])
        preds[sl] = model._forward(user[sl], item[sl], sparse, dense, seqs, lens).cpu().numpy()
    return normalize_prediction(preds, model, cold_start, unknown_num, unknown_index)
