"""`utils/tf_checkpoint.py`: graph-variable names of the reference's TF models -> this library's state
arrays.  The embedding names are literal in the reference (fm.py:84-87, ...), scoped layers follow
`dense_nn` (layers/dense.py:30-36); TF's auto-numbering of the unnamed layers is reproduced here
by hand (TensorFlow is not available), so this checks the mapping logic and its refusals."""
from collections import OrderedDict

import numpy as np
import pytest

from librecommender_amd.utils.tf_checkpoint import map_tf_variables, read_tf_variables

U, N, S, K, F = 6, 8, 11, 4, 5          # rows incl. OOV; F fields


def _rng():
    return np.random.default_rng(0)


def deepfm_checkpoint(hidden=(7, 7, 3), with_slots=True):
    r = _rng()
    v = {"embedding/user_embeds_var:0": r.random((U, K)), "embedding/item_embeds_var:0": r.random((N, K)),
         "embedding/sparse_embeds_var:0": r.random((S, K)), "embedding/user_linear_var:0": r.random((U, 1)),
         "embedding/item_linear_var:0": r.random((N, 1)), "embedding/sparse_linear_var:0": r.random(S),
         "embedding/dense_embeds_var:0": r.random((2, K)), "embedding/dense_linear_var:0": r.random(2),
         "dense/kernel:0": r.random((F, 1)), "dense/bias:0": r.random(1),
         "dense_1/kernel:0": r.random((1 + K + hidden[-1], 1)), "dense_1/bias:0": r.random(1)}
    d = F * K
    for n, size in enumerate([d, *hidden[:-1]]):
        tag = "mlp/batch_normalization" + (f"_{n}" if n else "")
        v.update({f"{tag}/gamma:0": r.random(size), f"{tag}/beta:0": r.random(size),
                  f"{tag}/moving_mean:0": r.random(size), f"{tag}/moving_variance:0": r.random(size)})
    for i, h in enumerate(hidden, start=1):
        v.update({f"mlp/mlp_layer{i}/kernel:0": r.random((d, h)), f"mlp/mlp_layer{i}/bias:0": r.random(h)})
        d = h
    if with_slots:
        v.update({"embedding/user_embeds_var/Adam:0": r.random((U, K)), "embedding/user_embeds_var/Adam_1:0": r.random((U, K)),
                  "dense/kernel/Adam:0": r.random((F, 1)), "beta1_power:0": np.float32(0.5), "beta2_power:0": np.float32(0.9)})
    return v


def deepfm_model_side(hidden=(7, 7, 3)):
    shapes = OrderedDict([("embedding/dense_embeds_var", (2, K)), ("embedding/dense_linear_var", (2,)),
                          ("linear/kernel", (F, 1)), ("linear/bias", (1,)),
                          ("mlp/bn_in/gamma", (F * K,)), ("mlp/bn_in/beta", (F * K,))])
    bns = [("mlp/bn_in", "mlp/bn_in/gamma", "mlp/bn_in/beta", F * K)]
    d = F * K
    for i, h in enumerate(hidden, start=1):
        shapes[f"mlp/mlp_layer{i}/kernel"], shapes[f"mlp/mlp_layer{i}/bias"] = (d, h), (h,)
        if i != len(hidden):
            shapes[f"mlp/bn{i}/gamma"], shapes[f"mlp/bn{i}/beta"] = (h,), (h,)
            bns.append((f"mlp/bn{i}", f"mlp/bn{i}/gamma", f"mlp/bn{i}/beta", h))
        d = h
    shapes["out/kernel"], shapes["out/bias"] = (1 + K + hidden[-1], 1), (1,)
    return shapes, bns, {"user": U, "item": N, "sparse": S}


def strip(v):
    return {k[:-2]: np.asarray(a, dtype=np.float32) for k, a in v.items()}


def test_deepfm_mapping(tmp_path):
    ckpt = deepfm_checkpoint()
    np.savez_compressed(tmp_path / "m_tf_variables.npz", **ckpt)
    tf_vars = read_tf_variables(str(tmp_path), "m")
    assert "embedding/user_embeds_var" in tf_vars and not any(k.endswith(":0") for k in tf_vars)
    shapes, bns, rows = deepfm_model_side()
    out = map_tf_variables(tf_vars, shapes, bns, rows)
    t = strip(ckpt)
    np.testing.assert_array_equal(out["embed"], np.concatenate([t["embedding/user_embeds_var"], t["embedding/item_embeds_var"],
                                                                t["embedding/sparse_embeds_var"]]))
    np.testing.assert_array_equal(out["lin"][-S:, 0], t["embedding/sparse_linear_var"])
    assert out["lin"].shape == (U + N + S, 1)
    np.testing.assert_array_equal(out["dense::linear/kernel"], t["dense/kernel"])           # first unnamed layer
    np.testing.assert_array_equal(out["dense::out/kernel"], t["dense_1/kernel"])            # second one
    np.testing.assert_array_equal(out["dense::mlp/mlp_layer2/kernel"], t["mlp/mlp_layer2/kernel"])
    np.testing.assert_array_equal(out["dense::mlp/bn_in/gamma"], t["mlp/batch_normalization/gamma"])
    np.testing.assert_array_equal(out["dense::mlp/bn2/beta"], t["mlp/batch_normalization_2/beta"])   # equal sizes: order decides
    np.testing.assert_array_equal(out["bn::mlp/bn1::var"], t["mlp/batch_normalization_1/moving_variance"])
    np.testing.assert_array_equal(out["dense::embedding/dense_embeds_var"], t["embedding/dense_embeds_var"])
    assert set(out) == {"embed", "lin"} | {f"dense::{k}" for k in shapes} | {f"bn::{b[0]}::{s}" for b in bns for s in ("mean", "var")}


def test_fm_and_din_shapes_of_graph():
    r = _rng()
    fm = {"embedding/user_embeds_var": r.random((U, K)), "embedding/item_embeds_var": r.random((N, K)),
          "embedding/user_linear_var": r.random((U, 1)), "embedding/item_linear_var": r.random((N, 1)),
          "dense/kernel": r.random((2, 1)), "dense/bias": r.random(1), "dense_1/kernel": r.random((K, 1)), "dense_1/bias": r.random(1),
          "batch_normalization/gamma": r.random(K), "batch_normalization/beta": r.random(K),
          "batch_normalization/moving_mean": r.random(K), "batch_normalization/moving_variance": r.random(K)}
    shapes = OrderedDict([("linear/kernel", (2, 1)), ("linear/bias", (1,)), ("bn/gamma", (K,)), ("bn/beta", (K,)),
                          ("pair/kernel", (K, 1)), ("pair/bias", (1,))])
    out = map_tf_variables(fm, shapes, [("bn", "bn/gamma", "bn/beta", K)], {"user": U, "item": N, "sparse": 0})
    np.testing.assert_array_equal(out["dense::pair/kernel"], fm["dense_1/kernel"].astype(np.float32))
    np.testing.assert_array_equal(out["bn::bn::mean"], fm["batch_normalization/moving_mean"].astype(np.float32))
    assert out["embed"].shape == (U + N, K)
    din = {"embedding/user_embeds_var": r.random((U, K)), "embedding/item_embeds_var": r.random((N, K)),
           "attention/attention_layer1/kernel": r.random((4 * K, 16)), "attention/attention_layer1/bias": r.random(16),
           "attention/attention_layer2/kernel": r.random((16, 1)), "attention/attention_layer2/bias": r.random(1),
           "mlp/mlp_layer1/kernel": r.random((3 * K, 5)), "mlp/mlp_layer1/bias": r.random(5),
           "dense/kernel": r.random((5, 1)), "dense/bias": r.random(1)}
    shapes = OrderedDict([(k, v.shape) for k, v in din.items() if "/" in k and not k.startswith(("embedding", "dense"))]
                         + [("out/kernel", (5, 1)), ("out/bias", (1,))])
    out = map_tf_variables(din, shapes, [], {"user": U, "item": N, "sparse": 0}, with_linear=False)
    assert "lin" not in out
    np.testing.assert_array_equal(out["dense::out/kernel"], din["dense/kernel"].astype(np.float32))
    np.testing.assert_array_equal(out["dense::attention/attention_layer1/kernel"], din["attention/attention_layer1/kernel"].astype(np.float32))


def test_refuses_what_it_cannot_account_for():
    shapes, bns, rows = deepfm_model_side()
    good = strip(deepfm_checkpoint(with_slots=False))
    with pytest.raises(ValueError, match="no place for"):
        map_tf_variables({**good, "extra/kernel": np.zeros((2, 2))}, shapes, bns, rows)
    with pytest.raises(KeyError):
        map_tf_variables({k: v for k, v in good.items() if k != "embedding/item_embeds_var"}, shapes, bns, rows)
    with pytest.raises(ValueError, match="rows"):
        map_tf_variables(good, shapes, bns, {**rows, "user": U + 1})
    bad = dict(good)
    bad["dense_1/kernel"] = np.zeros((3, 1), np.float32)
    with pytest.raises(ValueError, match="shape"):
        map_tf_variables(bad, shapes, bns, rows)
    fewer = {k: v for k, v in good.items() if not k.startswith("mlp/batch_normalization_2")}
    with pytest.raises(ValueError, match="BatchNorm"):
        map_tf_variables(fewer, shapes, bns, rows)
    no_out = {k: v for k, v in good.items() if not k.startswith("dense_1")}
    with pytest.raises(ValueError, match="unnamed dense"):
        map_tf_variables(no_out, shapes, bns, rows)


def test_export_is_the_inverse_of_the_reader():
    from librecommender_amd.utils.tf_checkpoint import to_tf_variables
    shapes, bns, rows = deepfm_model_side()
    ckpt = deepfm_checkpoint(with_slots=False)
    arrays = map_tf_variables(strip(ckpt), shapes, bns, rows)
    back = to_tf_variables(arrays, shapes, bns, rows)
    assert sorted(back) == sorted(ckpt)
    for k in ckpt:
        assert back[k].shape == np.asarray(ckpt[k]).shape, k
        np.testing.assert_array_equal(back[k], np.asarray(ckpt[k], dtype=np.float32), err_msg=k)
    again = map_tf_variables({k[:-2]: v for k, v in back.items()}, shapes, bns, rows)
    for k in arrays:
        np.testing.assert_array_equal(again[k], arrays[k])
