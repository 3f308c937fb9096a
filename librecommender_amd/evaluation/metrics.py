"""Evaluation metrics (`libreco/evaluation/metrics.py`).  Host-side; sklearn for the AUCs."""
import numpy as np

RATING_METRICS = {"loss", "rmse", "mae", "r2"}
POINTWISE_METRICS = {"loss", "log_loss", "balanced_accuracy", "roc_auc", "pr_auc", "roc_gauc"}
LISTWISE_METRICS = {"precision", "recall", "map", "ndcg", "coverage"}


def rmse(y_true, y_pred):
    d = np.asarray(y_true, dtype=np.float64) - np.asarray(y_pred, dtype=np.float64)
    return float(np.sqrt(np.mean(d * d)))


def roc_gauc(y_true, y_prob, users):
    """Per-user AUC weighted by the user's row count; single-class users contribute 0."""
    from sklearn.metrics import roc_auc_score

    y_true, y_prob, users = map(np.asarray, (y_true, y_prob, users))
    order = np.argsort(users, kind="stable")
    bounds = np.flatnonzero(np.r_[True, users[order][1:] != users[order][:-1], True])
    total = 0.0
    for a, b in zip(bounds[:-1], bounds[1:]):
        rows = order[a:b]
        t = y_true[rows]
        if t.min() != t.max():
            total += roc_auc_score(t, y_prob[rows]) * len(rows)
    return total / len(users)


def _hits(y_true, y_reco, k):
    marks = np.zeros(k, dtype=np.float32)
    truth = set(y_true)
    for pos, it in enumerate(list(y_reco)[:k]):
        if it in truth:
            marks[pos] = 1
    return marks


def precision_at_k(y_true, y_reco, k):
    return len(set(y_reco) & set(y_true)) / k


def recall_at_k(y_true, y_reco, _k):
    return len(set(y_reco) & set(y_true)) / len(y_true)


def average_precision_at_k(y_true, y_reco, k):
    marks = _hits(y_true, y_reco, k)
    if marks.sum() == 0:
        return 0
    prec = np.cumsum(marks) / np.arange(1, k + 1)
    return float(np.mean(prec[marks > 0]))


def ndcg_at_k(y_true, y_reco, k):
    marks = _hits(y_true, y_reco, k)
    if marks.sum() == 0:
        return 0
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    return float(np.sum(marks * disc) / np.sum(np.sort(marks)[::-1] * disc))


def listwise_mean(fn, y_trues, y_recos, users, k):
    return float(np.mean([fn(y_trues[u], y_recos[u], k) for u in users]))


def coverage(y_recos, users, n_items):
    seen = set()
    for u in users:
        seen.update(np.asarray(y_recos[u]).tolist())
    return len(seen) / n_items * 100
