"""Reading `<model_name>_tf_variables.npz` written by the reference's TF models
(`utils/save_load.py:70-80`: every `tf.global_variables()` entry under its graph name).

What pins the names (TensorFlow itself is not available in this build, so none of this has been
checked against a file written by TF — the mapping refuses anything it cannot account for):
  * embedding variables: literal in the reference, `algorithms/fm.py:84-87`, `deepfm.py:85-88`,
    `din.py:99-102`, `two_tower.py:108-111` (`embedding/user_embeds_var`, ...);
  * layers built by `dense_nn(..., name=S)` (`layers/dense.py:30-36`): `S/S_layer<i>/{kernel,bias}` —
    the names this library's parameters already carry;
  * Adam slots `<var>/Adam`, `<var>/Adam_1` (`tfops/variables.py:44-76`) — skipped;
  * layers without a name (`tf_dense(units=1)` of the linear / output terms, every
    `tf.layers.batch_normalization`): TF numbers them `dense, dense_1, ...` and
    `batch_normalization, batch_normalization_1, ...` in construction order.  They are matched by
    that order inside their variable scope to this library's layers in ITS construction order
    (which follows the reference's `build_model`), and every assignment is shape-checked.
"""
from __future__ import annotations

import os
import re
from collections import OrderedDict

import numpy as np

_SLOT = re.compile(r"(/Adam(_\d+)?|/Ftrl(_\d+)?)$|^(beta\d_power|global_step)")
_AUTO = re.compile(r"^(dense|batch_normalization)(?:_(\d+))?$")
_BN_KINDS = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}


def read_tf_variables(path, model_name):
    with np.load(os.path.join(path, f"{model_name}_tf_variables.npz")) as f:
        return {k[:-2] if k.endswith(":0") else k: f[k] for k in f.files}


def _auto_layers(tf_vars, base):
    """{scope: [layer paths of TF-numbered `base` layers in construction order]}."""
    found = {}
    for name in tf_vars:
        layer = name.rsplit("/", 1)[0]
        scope, _, leaf = layer.rpartition("/")
        m = _AUTO.match(leaf)
        if m and m.group(1) == base:
            found.setdefault(scope, {})[layer] = int(m.group(2) or 0)
    return {s: [k for k, _ in sorted(d.items(), key=lambda kv: kv[1])] for s, d in found.items()}


def map_tf_variables(tf_vars, param_shapes, bn_layers, table_rows, with_linear=True):
    """TF graph variables -> the arrays `load_state_arrays` of the feature models takes.

    param_shapes: OrderedDict name -> shape of this model's dense parameters in construction order
    bn_layers:    ordered [(key, gamma_name, beta_name, size)] of its BatchNorm layers
    table_rows:   {"user": U+1, "item": N+1, "sparse": S}  (0 = the model has no such rows)"""
    tf_vars = {k: v for k, v in tf_vars.items() if not _SLOT.search(k)}
    used, out = set(), {}

    def take(name, shape=None):
        if name not in tf_vars:
            raise KeyError(f"`{name}` is not in the TF checkpoint (has: {sorted(tf_vars)[:8]} ...)")
        v = np.asarray(tf_vars[name], dtype=np.float32)
        if shape is not None and tuple(v.shape) != tuple(shape):
            raise ValueError(f"`{name}` has shape {v.shape}, this model expects {tuple(shape)}")
        used.add(name)
        return v

    for kind, key in (("embeds", "embed"),) + ((("linear", "lin"),) if with_linear else ()):
        parts = []
        for side in ("user", "item", "sparse"):
            if table_rows.get(side):
                v = take(f"embedding/{side}_{kind}_var")
                v = v.reshape(v.shape[0], -1)
                if v.shape[0] != table_rows[side]:
                    raise ValueError(f"embedding/{side}_{kind}_var has {v.shape[0]} rows, this model expects {table_rows[side]}")
                parts.append(v)
        out[key] = np.concatenate(parts, axis=0)
    # dense parameters that carry their TF name
    pending = OrderedDict()
    bn_params = {n for _, g, b, _ in bn_layers for n in (g, b)}
    for name, shape in param_shapes.items():
        if name in bn_params:
            continue
        if name in tf_vars or "/" not in name:       # carries its TF name (a bare name has no unnamed form)
            out[f"dense::{name}"] = take(name, shape)
        else:
            pending[name] = shape
    # unnamed dense layers: construction order inside their scope
    mine = OrderedDict()
    for name in pending:
        layer, kind = name.rsplit("/", 1)
        mine.setdefault(layer.rpartition("/")[0], OrderedDict()).setdefault(layer, {})[kind] = name
    theirs = _auto_layers(tf_vars, "dense")
    for scope, layers in mine.items():
        cand = theirs.get(scope, [])
        if len(cand) != len(layers):
            raise ValueError(f"scope `{scope or '<root>'}`: {len(layers)} unnamed dense layers here "
                             f"({list(layers)}), {len(cand)} in the TF checkpoint ({cand})")
        for (layer, kinds), tf_layer in zip(layers.items(), cand):
            for kind, name in kinds.items():
                out[f"dense::{name}"] = take(f"{tf_layer}/{kind}", pending[name])
    # BatchNorm layers: construction order inside their scope
    mine_bn = OrderedDict()
    for key, g, b, size in bn_layers:
        mine_bn.setdefault(key.rpartition("/")[0], []).append((key, g, b, size))
    theirs_bn = _auto_layers(tf_vars, "batch_normalization")
    for scope, layers in mine_bn.items():
        cand = theirs_bn.get(scope, [])
        if len(cand) != len(layers):
            raise ValueError(f"scope `{scope or '<root>'}`: {len(layers)} BatchNorm layers here, "
                             f"{len(cand)} in the TF checkpoint ({cand})")
        for (key, g, b, size), tf_layer in zip(layers, cand):
            out[f"dense::{g}"] = take(f"{tf_layer}/gamma", (size,))
            out[f"dense::{b}"] = take(f"{tf_layer}/beta", (size,))
            out[f"bn::{key}::mean"] = take(f"{tf_layer}/moving_mean", (size,))
            out[f"bn::{key}::var"] = take(f"{tf_layer}/moving_variance", (size,))
    left = sorted(set(tf_vars) - used)
    if left:
        raise ValueError(f"TF checkpoint variables this model has no place for: {left}")
    return out


def to_tf_variables(arrays, param_shapes, bn_layers, table_rows, with_linear=True):
    """Inverse of `map_tf_variables`: this model's state arrays under the reference's graph names
    (with the `:0` suffix `utils/save_load.py:70-80` writes), for `Model.load(..., manual=True)` there.
    Unnamed layers are numbered per scope in construction order — unambiguous for FM / DeepFM / DIN,
    whose BatchNorm layers all live in one scope and whose unnamed dense layers are all at the root."""
    out = {}
    for key, kind in (("embed", "embeds"),) + ((("lin", "linear"),) if with_linear else ()):
        start = 0
        for side in ("user", "item", "sparse"):
            n = table_rows.get(side, 0)
            if n:
                block = np.asarray(arrays[key][start:start + n], dtype=np.float32)
                if kind == "linear" and side == "sparse":
                    block = block.reshape(-1)                     # `sparse_linear_var` is 1-D (deepfm.py:215)
                out[f"embedding/{side}_{kind}_var:0"] = block
                start += n
    bn_params = {n for _, g, b, _ in bn_layers for n in (g, b)}
    counters = {}

    def numbered(scope, base):
        k = counters.get((scope, base), 0)
        counters[(scope, base)] = k + 1
        return (scope + "/" if scope else "") + base + (f"_{k}" if k else "")

    unnamed = {}
    for name in param_shapes:
        if name in bn_params:
            continue
        layer, _, kind = name.rpartition("/")
        scope = layer.rpartition("/")[0]
        tf_name = name
        if layer and not name.startswith("embedding/") and not scope:         # root-level named here, unnamed in TF
            if layer not in unnamed:
                unnamed[layer] = numbered("", "dense")
            tf_name = f"{unnamed[layer]}/{kind}"
        out[f"{tf_name}:0"] = np.asarray(arrays[f"dense::{name}"], dtype=np.float32)
    for key, g, b, _ in bn_layers:
        tf_layer = numbered(key.rpartition("/")[0], "batch_normalization")
        out[f"{tf_layer}/gamma:0"] = np.asarray(arrays[f"dense::{g}"], dtype=np.float32)
        out[f"{tf_layer}/beta:0"] = np.asarray(arrays[f"dense::{b}"], dtype=np.float32)
        out[f"{tf_layer}/moving_mean:0"] = np.asarray(arrays[f"bn::{key}::mean"], dtype=np.float32)
        out[f"{tf_layer}/moving_variance:0"] = np.asarray(arrays[f"bn::{key}::var"], dtype=np.float32)
    return out
