"""Generate tests/golden/tf_<model>.npz by running the REFERENCE's TensorFlow graphs for one
deterministic train step.  TEST INFRASTRUCTURE ONLY.

This needs a real TensorFlow (`tensorflow>=1.15,<2.16`, reference requirements.txt:5) next to the
reference checkout — neither the build container nor the GPU box has one, so the fixtures cannot be
made there.  On any box that has both:

    LIBRECO_REFERENCE=/path/to/LibRecommender python -m oracle.make_tf_golden            # all models
    LIBRECO_REFERENCE=/path/to/LibRecommender python -m oracle.make_tf_golden DeepFM DIN  # a subset

and commit `tests/golden/tf_*.npz` (+ `tests/golden/tf_*_tf_variables.npz`).  With them present
`tests/test_tf_golden_cpu.py` pins `oracle/models_torch.py` against TensorFlow itself and the
"PARITY UNPINNED" header of that file can go.

`--dry-run` (works without TensorFlow, under the import stubs of `oracle/_stubs`) builds the data,
the reference model object and one batch through the reference's own loader and feed-dict code and
prints the feed layout; it writes nothing.

Per model the fixture holds (flat npz keys):
    meta                 JSON: model, hyper-parameters, TF version, variable names in creation order
    feed/<attribute>     the batch exactly as `batch/tf_feed_dicts.py:12-22` feeds it, keyed by the
                         model attribute that holds the placeholder (`user_indices`, `labels`, ...)
    extra/<name>         data_info arrays a restatement needs (item_sparse_unique, ...)
    var0/<name>          every model variable after `global_variables_initializer`
    grad/<name>          d total_loss / d variable on that batch (IndexedSlices densified)
    var1/<name>          every model variable after ONE `training_op` (Adam + BatchNorm update ops,
                         `training/tf_trainer.py:104-124`)
    loss, logits         `trainer.loss` and `model.output` (training mode, same batch)
where <name> is the oracle's name for the variable (`oracle/tf_names.py`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))

from oracle.tf_names import canonical, is_slot  # noqa: E402

SEED = 2026
BATCH = 64

# model -> (module, class, constructor keywords).  Small hidden sizes keep the fixtures a few 100 KB.
MODELS = {
    "FM": ("fm", "FM", dict(embed_size=8, lr=1e-3, use_bn=True, batch_size=BATCH, num_neg=1, seed=42)),
    "DeepFM": ("deepfm", "DeepFM", dict(embed_size=8, lr=1e-3, use_bn=True, hidden_units=(32, 16, 8),
                                        batch_size=BATCH, num_neg=1, seed=42)),
    "DIN": ("din", "DIN", dict(embed_size=8, lr=1e-3, use_bn=True, hidden_units=(32, 16, 8), recent_num=6,
                               batch_size=BATCH, num_neg=1, seed=42)),
    "TwoTower": ("two_tower", "TwoTower", dict(embed_size=8, lr=1e-3, use_bn=True, hidden_units=(32, 16),
                                                loss_type="cross_entropy", batch_size=BATCH, num_neg=1, seed=42)),
}


def import_reference(dry_run):
    """`libreco` from the checkout named by LIBRECO_REFERENCE (default /root/reference).  The package's
    `algorithms/__init__` imports Cython modules none of these four models need -> bypassed the same
    way `oracle/ref_loader.py` does.  Without --dry-run the real `tensorflow` must import."""
    from oracle import ref_loader

    if dry_run:
        return ref_loader.load()
    try:
        import tensorflow  # noqa: F401
    except ImportError as e:
        raise SystemExit(f"TensorFlow is required to generate the fixtures ({e}); use --dry-run to check the "
                         "data path only") from e
    if "_stubs" in (getattr(sys.modules["tensorflow"], "__file__", "") or ""):
        raise SystemExit("the import stub shadows TensorFlow; run without oracle/_stubs on sys.path")
    import types

    if not ref_loader.available():
        raise SystemExit(f"reference checkout not found at {ref_loader.REFERENCE} (set LIBRECO_REFERENCE)")
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(ref_loader.REFERENCE))
    import libreco

    if "libreco.algorithms" not in sys.modules:
        pkg = types.ModuleType("libreco.algorithms")
        pkg.__path__ = [str(ref_loader.REFERENCE / "libreco" / "algorithms")]
        sys.modules["libreco.algorithms"] = pkg
    return libreco


def make_data():
    """600 implicit interactions of 40 users x 30 items with two user and one item categorical column
    (`data/dataset.py` DatasetFeat.build_trainset)."""
    import pandas as pd
    from libreco.data import DatasetFeat

    rng = np.random.default_rng(SEED)
    n, n_users, n_items = 600, 40, 30
    users = np.concatenate([np.arange(n_users), rng.integers(0, n_users, n - n_users)])
    items = np.concatenate([np.arange(n_items), rng.integers(0, n_items, n - n_items)])
    sex_of, occ_of, genre_of = rng.integers(0, 2, n_users), rng.integers(0, 5, n_users), rng.integers(0, 4, n_items)
    df = pd.DataFrame({"user": users, "item": items, "label": 1, "time": np.arange(n),
                       "sex": sex_of[users], "occupation": occ_of[users], "genre": genre_of[items]})
    return DatasetFeat.build_trainset(df, user_col=["sex", "occupation"], item_col=["genre"],
                                      sparse_col=["sex", "occupation", "genre"], dense_col=[])


def first_batch(model, train_data):
    """One batch through the reference's loader + collator (`batch/batch_data.py:46-64`) and feed-dict
    code (`batch/tf_feed_dicts.py:12-22`), keyed by the model attribute of each placeholder."""
    from libreco.batch import get_batch_loader, get_tf_feeds
    from libreco.batch.batch_data import adjust_batch_size

    bs = adjust_batch_size(model, model.batch_size)
    loader = get_batch_loader(model, train_data, True, bs, False, 0, model.seed)
    feed = get_tf_feeds(model, next(iter(loader)), is_training=True)
    attr_of = {id(v): k for k, v in vars(model).items()}
    named = {}
    for ph, value in feed.items():
        if id(ph) not in attr_of:
            raise RuntimeError(f"fed tensor {ph!r} is not an attribute of the model")
        named[attr_of[id(ph)]] = (ph, np.asarray(value))
    return named


def extras(model):
    out = {}
    info = model.data_info
    for k in ("item_sparse_unique", "item_dense_unique", "user_sparse_unique", "user_dense_unique"):
        v = getattr(info, k, None)
        if v is not None:
            out[k] = np.asarray(v)
    for k in ("user_sparse_col", "item_sparse_col", "user_dense_col", "item_dense_col"):
        col = getattr(info, k, None)
        if col is not None and getattr(col, "index", None):
            out[f"{k}_index"] = np.asarray(col.index, dtype=np.int64)
    if hasattr(model, "item_corrections"):
        out["item_corrections"] = np.asarray(model.item_corrections)
    return out


def densify(g, shape):
    """sess.run of an IndexedSlices gradient -> dense array (duplicate indices add, as Adam's
    `_apply_sparse_shared` sums them)."""
    if hasattr(g, "indices"):
        d = np.zeros(shape, dtype=np.asarray(g.values).dtype)
        np.add.at(d, np.asarray(g.indices), np.asarray(g.values))
        return d
    return np.asarray(g)


def generate(name, dry_run):
    import importlib

    module, cls_name, hyper = MODELS[name]
    train_data, data_info = make_data()
    mod = importlib.import_module(f"libreco.algorithms.{module}")
    cls = getattr(mod, cls_name)
    if dry_run:
        mod.count_params = lambda: None       # sums variable shapes; the stub has none
    model = cls("ranking", data_info, **hyper)
    if name == "TwoTower" and getattr(model, "use_correction", False) and model.loss_type == "softmax":
        _, counts = np.unique(train_data.item_indices, return_counts=True)      # two_tower.py:426-429
        model.item_corrections = counts / len(train_data)
    model.build_model()
    model.model_built = True
    feed = first_batch(model, train_data)
    if dry_run:
        print(f"[{name}] feed:")
        for k, (_, v) in feed.items():
            print(f"    {k:28s} {str(v.dtype):8s} {v.shape}")
        for k, v in extras(model).items():
            print(f"    extra/{k:22s} {str(v.dtype):8s} {v.shape}")
        return None

    from libreco.tfops import tf
    from libreco.training.dispatch import get_trainer

    trainer = get_trainer(model)          # loss, Adam, update ops, global_variables_initializer
    model.trainer = trainer
    sess = model.sess
    total_loss = trainer.loss
    if trainer.use_reg:
        total_loss = total_loss + tf.add_n(tf.get_collection(tf.GraphKeys.REGULARIZATION_LOSSES))
    all_vars = [v for v in tf.global_variables() if not is_slot(v.name)]
    names = canonical(name, [v.name for v in all_vars])
    trainable = [v for v in all_vars if v in tf.trainable_variables()]
    feed_dict = {ph: v for ph, v in feed.values()}

    out = {}
    for v, a in zip(all_vars, sess.run(all_vars)):
        out[f"var0/{names[v.name]}"] = a
    fetches = [trainer.loss, tf.gradients(total_loss, trainable)]
    if hasattr(model, "output"):
        fetches.append(model.output)
    res = sess.run(fetches, feed_dict)        # no update ops fetched: moving statistics stay put
    out["loss"] = np.asarray(res[0])
    for v, g in zip(trainable, res[1]):
        out[f"grad/{names[v.name]}"] = densify(g, tuple(v.shape.as_list()))
    if hasattr(model, "output"):
        out["logits"] = np.asarray(res[2])
    step_loss, _ = sess.run((trainer.loss, trainer.training_op), feed_dict)
    out["step_loss"] = np.asarray(step_loss)
    for v, a in zip(all_vars, sess.run(all_vars)):
        out[f"var1/{names[v.name]}"] = a
    for k, (_, v) in feed.items():
        out[f"feed/{k}"] = v
    for k, v in extras(model).items():
        out[f"extra/{k}"] = v
    hyper_json = {k: (list(v) if isinstance(v, tuple) else v) for k, v in hyper.items()}
    hyper_json.update(epsilon=model.epsilon, n_users=int(model.n_users), n_items=int(model.n_items))
    if hasattr(model, "max_seq_len"):
        hyper_json["max_seq_len"] = int(model.max_seq_len)
    out["meta"] = np.asarray(json.dumps({
        "model": name, "hyper": hyper_json, "tf_version": tf.__version__, "source": "tensorflow",
        "variables": [v.name for v in all_vars], "names": {v.name: names[v.name] for v in all_vars},
        "trainable": [names[v.name] for v in trainable]}))
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"tf_{name.lower()}.npz", **out)
    # the reference's own variable file (`utils/save_load.py:70-80`), for the product's
    # `load_tf_variables` / `utils/tf_checkpoint.map_tf_variables`
    from libreco.utils.save_load import save_tf_variables
    save_tf_variables(sess, str(OUT), f"tf_{name.lower()}", inference_only=False)
    print(f"[{name}] loss {float(out['loss']):.6f}  {len(all_vars)} variables -> {OUT / f'tf_{name.lower()}.npz'}")
    tf.reset_default_graph()
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("models", nargs="*", default=list(MODELS), help=f"subset of {list(MODELS)}")
    ap.add_argument("--dry-run", action="store_true", help="no TensorFlow: build data, model object and one batch, print the feed")
    args = ap.parse_args()
    os.environ.setdefault("TF_CPP_MIN_LOG_LEVEL", "2")
    os.environ.setdefault("TF_DETERMINISTIC_OPS", "1")
    import_reference(args.dry_run)
    for name in args.models:
        if name not in MODELS:
            raise SystemExit(f"unknown model {name}; choose from {list(MODELS)}")
        generate(name, args.dry_run)


if __name__ == "__main__":
    main()
