"""Argument checks shared by the model classes (`libreco/utils/validate.py`): same conditions,
same exception types — the reference's model tests assert on them."""
import numpy as np

from .misc import colorize


def check_unknown(model, user, item):
    """Positions whose user or item is the OOV index (prints the reference's red warning)."""
    bad = np.flatnonzero((user == model.n_users) | (item == model.n_items)).tolist()
    if bad:
        print(colorize(f"Detect {len(bad)} unknown interaction(s), position: {bad}", "red"))
    return len(bad), bad, user, item


def check_unknown_user(data_info, user, inner_id=False):
    known, unknown = [], []
    for u in ([user] if np.isscalar(user) else user):
        if inner_id:
            (known if 0 <= u < data_info.n_users else unknown).append(u)
        elif u in data_info.user2id:
            known.append(data_info.user2id[u])
        else:
            print(colorize(f"Detect unknown user: {u}", "red"))
            unknown.append(u)
    return known, unknown


def check_seq_mode(recent_num, random_num):
    if recent_num is not None:
        assert isinstance(recent_num, int), "recent_num must be integer"
        return "recent", recent_num
    if random_num is not None:
        assert isinstance(random_num, int), "random_num must be integer"
        return "random", random_num
    return "recent", 10


def sparse_feat_size(data_info):
    mats = [m for m in (data_info.user_sparse_unique, data_info.item_sparse_unique) if m is not None]
    return int(max(np.max(m) for m in mats)) + 1 if mats else None


def check_multi_sparse(data_info, combiner):
    if data_info.multi_sparse_combine_info and combiner is not None:
        if combiner not in ("normal", "sum", "mean", "sqrtn"):
            raise ValueError(f"unsupported multi_sparse_combiner type: {combiner}")
        return combiner
    return "normal"


def check_fitting(model, train_data, eval_data, neg_sampling, k):
    assert isinstance(neg_sampling, bool), (
        f"`neg_sampling` in `fit()` must be bool, got `{neg_sampling}`. Set `model.fit(..., "
        f"neg_sampling=True)` if your data is implicit(i.e., `task` is ranking) and ONLY contains "
        f"positive labels. Otherwise, negative sampling is not needed.")
    if model.task == "rating" and neg_sampling:
        raise ValueError("`rating` task should not use negative sampling")
    if getattr(model, "loss_type", None) in ("bpr", "max_margin") and not neg_sampling:
        raise ValueError(f"`{model.loss_type}` loss must use negative sampling.")
    check_labels(model, train_data.labels, neg_sampling)
    if getattr(model, "loaded", False):
        raise RuntimeError("Loaded model doesn't support retraining, use `rebuild_model` instead. "
                           "Or constructing a new model from scratch.")
    if eval_data is not None and k > model.n_items:
        raise ValueError(f"eval `k` {k} exceeds num of items {model.n_items}")


def check_labels(model, labels, neg_sampling):
    if model.task == "ranking" and not neg_sampling:
        uniq = np.unique(labels)
        if len(uniq) != 2 or uniq.min() != 0.0 or uniq.max() != 1.0:
            raise ValueError("For `ranking` task without negative sampling, labels in data must be 0 "
                             f"and 1, got unique labels: {uniq}")


def hidden_units_config(hidden_units):
    if isinstance(hidden_units, int):
        return [hidden_units]
    if not isinstance(hidden_units, (list, tuple)) or not hidden_units:
        raise ValueError(f"`hidden_units` must be one of (int, list of int, tuple of int), got: {hidden_units}")
    for u in hidden_units:
        if not isinstance(u, int):
            raise ValueError(f"`hidden_units` must be one of (int, list of int, tuple of int), got: {hidden_units}")
    return list(hidden_units)


def dropout_config(rate):
    if not rate:
        return 0.0
    if rate <= 0.0 or rate >= 1.0:
        raise ValueError("dropout_rate must be in (0.0, 1.0)")
    return rate


def reg_config(reg):
    if not reg:
        return None
    if isinstance(reg, float) and reg > 0.0:
        return reg
    raise ValueError("reg must be float and positive...")
