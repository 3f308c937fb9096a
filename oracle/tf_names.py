"""TensorFlow graph-variable names -> the names `oracle/models_torch.py` keys its weights by.
TEST INFRASTRUCTURE ONLY (used by oracle/make_tf_golden.py and tests/test_tf_golden_cpu.py).

The reference names its embedding variables literally (`algorithms/fm.py:84-87`, `deepfm.py:85-88`,
`din.py:99-102`, `two_tower.py:259-300`) and the layers of `dense_nn(name=S)` as `S/S_layer<i>`
(`layers/dense.py:29-34`).  Everything else is numbered by TensorFlow in construction order:
`dense, dense_1, ...` for `tf_dense(units=1)` without a name and `batch_normalization,
batch_normalization_1, ...` for every `tf.layers.batch_normalization` inside its variable scope.
`names` must therefore be given in CREATION order (`tf.global_variables()` is; so is the key order
of the `.npz` that `utils/save_load.py:70-80` writes).
"""
from __future__ import annotations

import re
from typing import Dict, Iterable

# unnamed `tf_dense(units=1)` layers at the graph root, in the order `build_model` creates them
UNNAMED_DENSE = {
    "FM": ("linear", "pair"),          # fm.py:153,168
    "DeepFM": ("linear", "out"),       # deepfm.py:158,172
    "DIN": ("out",),                   # din.py:213
    "TwoTower": (),
}

_SLOT = re.compile(r"(/Adam(_\d+)?)$|^(beta\d_power|global_step)")
_DENSE = re.compile(r"^dense(_\d+)?$")
_BN = re.compile(r"^batch_normalization(_\d+)?$")
_LAYER = re.compile(r"^(\w+)_layer\d+$")
_EMBED = re.compile(r"^embedding/((user|item|sparse)_(embeds|linear)_var)$")
_BN_LEAF = {"moving_variance": "moving_var"}


def is_slot(name: str) -> bool:
    """Optimizer state (`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, ...), not a model variable."""
    return bool(_SLOT.search(strip(name)))


def strip(name: str) -> str:
    return name[:-2] if name.endswith(":0") else name


def canonical(model: str, names: Iterable[str]) -> Dict[str, str]:
    """{tf name (as given): oracle name} for every non-slot variable of `model`'s graph."""
    if model not in UNNAMED_DENSE:
        raise KeyError(f"no naming rule for model `{model}`")
    root_dense, bn_seen, dense_seen, out = UNNAMED_DENSE[model], {}, [], {}
    for raw in names:
        n = strip(raw)
        if _SLOT.search(n):
            continue
        m = _EMBED.match(n)
        if m:
            out[raw] = m.group(1)
            continue
        parts = n.split("/")
        # a keras Dense created under tf.variable_scope may or may not carry the scope in its name
        if len(parts) == 2 and _LAYER.match(parts[0]):
            parts = [_LAYER.match(parts[0]).group(1)] + parts
        for i, p in enumerate(parts[:-1]):
            scope = "/".join(parts[:i])
            if _DENSE.match(p):
                if scope:
                    raise ValueError(f"unnamed dense layer inside scope `{scope}`: {n}")
                if p not in dense_seen:
                    dense_seen.append(p)
                k = dense_seen.index(p)
                if k >= len(root_dense):
                    raise ValueError(f"{model} has {len(root_dense)} unnamed dense layers, found one more: {n}")
                parts[i] = root_dense[k]
            elif _BN.match(p):
                seen = bn_seen.setdefault(scope, [])
                if p not in seen:
                    seen.append(p)
                k = seen.index(p)
                parts[i] = "bn" if not scope else ("bn_in" if k == 0 else f"bn{k}")
        parts[-1] = _BN_LEAF.get(parts[-1], parts[-1])
        out[raw] = "/".join(parts)
    return out
