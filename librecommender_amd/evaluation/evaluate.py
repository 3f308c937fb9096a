"""`evaluate` / `print_metrics` (`libreco/evaluation/evaluate.py:62-193`): pointwise metrics call
`model.predict`, listwise metrics call `model.recommend_user` — i.e. the device hot path."""
import math
import numbers

import numpy as np
import pandas as pd

from ..data import TransformedEvalSet
from ..utils.validate import check_labels
from . import metrics as M


def _check_metrics(task, metrics, k):
    if not isinstance(metrics, (list, tuple)):
        metrics = [metrics]
    allowed = M.RATING_METRICS if task == "rating" else (M.POINTWISE_METRICS | M.LISTWISE_METRICS)
    for m in metrics:
        if m not in allowed:
            raise ValueError(f"Metrics `{m}` is not suitable for {task} task...")
    if not isinstance(k, numbers.Integral):
        raise TypeError("`k` must be integer")
    return list(metrics)


def _prepare(model, data, neg_sampling, seed):
    if isinstance(data, pd.DataFrame):
        assert "user" in data and "item" in data and "label" in data
        info = model.data_info
        u = np.array([info.user2id.get(x, model.n_users) for x in data["user"].tolist()])
        i = np.array([info.item2id.get(x, model.n_items) for x in data["item"].tolist()])
        data = TransformedEvalSet(u, i, data["label"].to_numpy(dtype=np.float32))
    if neg_sampling and not data.has_sampled:
        data.build_negatives(model.n_items, (getattr(model, "num_neg", None) or 1), seed=seed)
    else:
        check_labels(model, data.labels, neg_sampling)
    return data


def _predict_all(model, data, batch_size):
    preds = []
    for s in range(0, len(data), batch_size):
        u, i, _ = data[s:s + batch_size]
        preds.append(np.atleast_1d(model.predict(u, i, inner_id=True)))
    return np.concatenate(preds), np.asarray(data.labels)


def sample_users(data, seed, num):
    users = list(data.positive_consumed)
    if isinstance(num, int) and 0 < num < len(users):
        users = np.random.default_rng(seed).choice(users, num, replace=False).tolist()
    return users


def evaluate(model, data, neg_sampling, eval_batch_size=8192, metrics=None, k=10,
             sample_user_num=None, seed=42):
    if not isinstance(data, (pd.DataFrame, TransformedEvalSet)):
        raise ValueError("`data` must be `pandas.DataFrame` or `TransformedEvalSet`")
    data = _prepare(model, data, neg_sampling, seed)
    metrics = _check_metrics(model.task, metrics or ["loss"], k)
    out = {}
    if model.task == "rating":
        y_pred, y_true = _predict_all(model, data, eval_batch_size)
        for m in metrics:
            if m in ("rmse", "loss"):
                out[m] = M.rmse(y_true, y_pred)
            elif m == "mae":
                out[m] = float(np.mean(np.abs(y_true - y_pred)))
            elif m == "r2":
                from sklearn.metrics import r2_score
                out[m] = r2_score(y_true, y_pred)
        return out
    if M.POINTWISE_METRICS & set(metrics):
        from sklearn.metrics import auc, balanced_accuracy_score, log_loss, precision_recall_curve, roc_auc_score

        y_prob, y_true = _predict_all(model, data, eval_batch_size)
        for m in metrics:
            if m in ("log_loss", "loss"):
                out[m] = log_loss(y_true, y_prob)
            elif m == "balanced_accuracy":
                out[m] = balanced_accuracy_score(y_true, np.round(y_prob))
            elif m == "roc_auc":
                out[m] = roc_auc_score(y_true, y_prob)
            elif m == "roc_gauc":
                out[m] = M.roc_gauc(y_true, y_prob, data.user_indices)
            elif m == "pr_auc":
                p, r, _ = precision_recall_curve(y_true, y_prob)
                out[m] = auc(r, p)
    if M.LISTWISE_METRICS & set(metrics):
        users = sample_users(data, seed, sample_user_num)
        # the reference batches max(1, eval_batch_size // n_items) users because it materialises
        # B x N scores (evaluate.py:135); the fused top-k has no such limit
        step = max(1, getattr(model, "eval_user_batch", None) or math.floor(eval_batch_size / model.n_items) or 1)
        recos = {}
        for s in range(0, len(users), step):
            recos.update(model.recommend_user(user=users[s:s + step], n_rec=k, inner_id=True,
                                              filter_consumed=True, random_rec=False))
        fns = {"precision": M.precision_at_k, "recall": M.recall_at_k, "map": M.average_precision_at_k,
               "ndcg": M.ndcg_at_k}
        for m in metrics:
            if m == "coverage":
                out[m] = M.coverage(recos, users, model.n_items)
            elif m in fns:
                out[m] = M.listwise_mean(fns[m], data.positive_consumed, recos, users, k)
    return out


def print_metrics(model, neg_sampling, eval_data=None, metrics=None, eval_batch_size=8192, k=10,
                  sample_user_num=2048, seed=42):
    if not eval_data:
        return
    loss_name = "rmse" if model.task == "rating" else "log_loss"
    res = evaluate(model, eval_data, neg_sampling, eval_batch_size, metrics, k, sample_user_num, seed)
    for m, val in res.items():
        name = loss_name if m == "loss" else (f"{m}@{k}" if m in M.LISTWISE_METRICS else m)
        print(f"\t eval {name}: " + (f"{round(val, 2)}%" if m == "coverage" else f"{val:.4f}"))
