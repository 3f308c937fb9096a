"""Cold-start recommendations (`libreco/recommendation/cold_start.py`): unknown users get
`n_rec` draws (with replacement, `data_info.np_rng`) from the OOV user's default list
("average") or from the popular items ("popular")."""
import numpy as np


def popular_recommendations(data_info, inner_id, n_rec):
    """`cold_start.py:4-9`: `n_rec` draws with replacement from the popular (raw-id) items."""
    picks = data_info.np_rng.choice(data_info.popular_items, n_rec)
    return np.array([data_info.item2id[i] for i in picks]) if inner_id else picks


def cold_start_rec(data_info, default_recs, cold_start, users, n_rec, inner_id):
    if cold_start not in ("average", "popular"):
        raise ValueError(f"Unknown cold start strategy: {cold_start}")
    out = {}
    for u in users:
        if cold_start == "average":
            picks = data_info.np_rng.choice(default_recs, n_rec)
            out[u] = picks if inner_id else np.array([data_info.id2item[i] for i in picks])
        else:
            out[u] = popular_recommendations(data_info, inner_id, n_rec)
    return out
