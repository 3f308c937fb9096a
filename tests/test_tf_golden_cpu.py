"""The oracle (`oracle/models_torch.py`) against TensorFlow itself: fixtures `tests/golden/tf_<model>.npz`
written by `oracle/make_tf_golden.py` on a box that has TensorFlow next to the reference checkout.

Neither the build container nor the GPU box has TensorFlow, so the fixtures may be absent: the pinning
tests then SKIP with the command that makes them.  What always runs here: the naming rules for
TensorFlow's graph variables, and the checker itself on fixtures of the same format made from the
oracle (so that the day the real files arrive, a failure is about numerics, not plumbing)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models_torch as MT
from oracle.tf_names import canonical, is_slot

GOLDEN = Path(__file__).parent / "golden"
MODELS = ("FM", "DeepFM", "DIN", "TwoTower")
HOWTO = ("run `bash scripts/pin_tf_half.sh /path/to/LibRecommender` on a box with network access (it makes a venv with "
         "tensorflow==2.12.*, runs `python -m oracle.make_tf_golden`, writes tests/golden/TF_MANIFEST.json) and commit "
         "tests/golden/tf_*.npz + TF_MANIFEST.json")


# ---------------------------------------------------------------------------------------------------------
# naming rules
# ---------------------------------------------------------------------------------------------------------
def _adam(names):
    out = list(names)
    for n in names:
        if "moving" not in n:
            out += [n.replace(":0", "/Adam:0"), n.replace(":0", "/Adam_1:0")]
    return out + ["beta1_power:0", "beta2_power:0"]


def test_names_deepfm():
    tfn = ["embedding/user_linear_var:0", "embedding/item_linear_var:0", "embedding/user_embeds_var:0",
           "embedding/item_embeds_var:0", "embedding/sparse_linear_var:0", "embedding/sparse_embeds_var:0",
           "dense/kernel:0", "dense/bias:0",
           "mlp/batch_normalization/gamma:0", "mlp/batch_normalization/beta:0",
           "mlp/batch_normalization/moving_mean:0", "mlp/batch_normalization/moving_variance:0",
           "mlp/mlp_layer1/kernel:0", "mlp/mlp_layer1/bias:0",
           "mlp/batch_normalization_1/gamma:0", "mlp/batch_normalization_1/beta:0",
           "mlp/batch_normalization_1/moving_mean:0", "mlp/batch_normalization_1/moving_variance:0",
           "mlp/mlp_layer2/kernel:0", "mlp/mlp_layer2/bias:0",
           "dense_1/kernel:0", "dense_1/bias:0"]
    got = canonical("DeepFM", _adam(tfn))
    assert len(got) == len(tfn)
    assert got["embedding/user_embeds_var:0"] == "user_embeds_var"
    assert got["embedding/sparse_linear_var:0"] == "sparse_linear_var"
    assert got["dense/kernel:0"] == "linear/kernel" and got["dense_1/bias:0"] == "out/bias"
    assert got["mlp/batch_normalization/moving_variance:0"] == "mlp/bn_in/moving_var"
    assert got["mlp/batch_normalization_1/gamma:0"] == "mlp/bn1/gamma"
    assert got["mlp/mlp_layer2/kernel:0"] == "mlp/mlp_layer2/kernel"
    assert all(is_slot(n) for n in _adam(tfn) if n not in got)


def test_names_fm_din_twotower():
    # keras numbers unnamed layers with a process-wide counter: the suffix is not 0-based in a 2nd graph
    fm = canonical("FM", ["embedding/user_linear_var:0", "dense_7/kernel:0", "dense_7/bias:0",
                          "batch_normalization_3/gamma:0", "batch_normalization_3/moving_variance:0",
                          "dense_8/kernel:0"])
    assert fm["dense_7/kernel:0"] == "linear/kernel" and fm["dense_8/kernel:0"] == "pair/kernel"
    assert fm["batch_normalization_3/gamma:0"] == "bn/gamma"
    assert fm["batch_normalization_3/moving_variance:0"] == "bn/moving_var"
    din = canonical("DIN", ["embedding/user_embeds_var:0", "embedding/dense_embeds_var:0",
                            "attention/attention_layer1/kernel:0", "attention_layer2/bias:0",
                            "mlp/batch_normalization/beta:0", "mlp/mlp_layer1/kernel:0", "dense/kernel:0"])
    assert din["embedding/dense_embeds_var:0"] == "embedding/dense_embeds_var"
    assert din["attention_layer2/bias:0"] == "attention/attention_layer2/bias"
    assert din["dense/kernel:0"] == "out/kernel" and din["mlp/batch_normalization/beta:0"] == "mlp/bn_in/beta"
    tt = canonical("TwoTower", ["embedding/item_embeds_var:0", "temperature_var:0",
                                "user_tower/batch_normalization/gamma:0", "user_tower/user_tower_layer1/kernel:0",
                                "user_tower/batch_normalization_1/gamma:0",
                                "item_tower/batch_normalization/gamma:0", "item_tower/item_tower_layer1/bias:0"])
    assert tt["user_tower/batch_normalization_1/gamma:0"] == "user_tower/bn1/gamma"
    assert tt["item_tower/batch_normalization/gamma:0"] == "item_tower/bn_in/gamma"
    assert tt["temperature_var:0"] == "temperature_var"
    with pytest.raises(ValueError):
        canonical("TwoTower", ["dense/kernel:0"])
    with pytest.raises(ValueError):
        canonical("DIN", ["dense/kernel:0", "dense_1/kernel:0"])


# ---------------------------------------------------------------------------------------------------------
# the checker
# ---------------------------------------------------------------------------------------------------------
def _t(a, dtype=torch.long):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _run_oracle(meta, W0, feed, extra):
    """One training-mode forward, its loss, gradients and the post-step variables from the oracle started
    at `W0` (oracle names).  Returns (logits, loss, grads, variables_after)."""
    name, hp = meta["model"], meta["hyper"]
    W = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in W0.items()}
    users, items = _t(feed["user_indices"]), _t(feed["item_indices"])
    labels = _t(feed["labels"], torch.float32)
    kw = dict(use_bn=hp["use_bn"], lr=hp["lr"], epsilon=hp["epsilon"])
    if name == "FM":
        o = MT.FMOracle(W, **kw)
        logits = o.forward(users, items, _t(feed["sparse_indices"]), True, False)
    elif name == "DeepFM":
        o = MT.DeepFMOracle(W, tuple(hp["hidden_units"]), **kw)
        logits = o.forward(users, items, _t(feed["sparse_indices"]), training=True)
    elif name == "DIN":
        o = MT.DINOracle(W, tuple(hp["hidden_units"]), max_seq_len=hp["max_seq_len"],
                         item_sparse_unique=extra.get("item_sparse_unique"), **kw)
        sparse = _t(feed["sparse_indices"]) if "sparse_indices" in feed else None
        logits = o.forward(users, items, sparse, None, _t(feed["user_interacted_seq"]),
                           _t(feed["user_interacted_len"]), True)
    elif name == "TwoTower":
        assert hp["loss_type"] == "cross_entropy"
        o = MT.TwoTowerOracle(W, tuple(hp["hidden_units"]), **kw)
        us = _t(feed["user_sparse_indices"]) if "user_sparse_indices" in feed else None
        isp = _t(feed["item_sparse_indices"]) if "item_sparse_indices" in feed else None
        logits = (o.user_embeds(users, us, None, True) * o.item_embeds(items, isp, None, True)).sum(1)
    else:
        raise KeyError(name)
    loss = F.binary_cross_entropy_with_logits(logits, labels)           # tfops/loss.py:14-16
    loss.backward()
    grads = {k: (p.grad.to_dense() if p.grad.is_sparse else p.grad).clone()
             for k, p in o.V.v.items() if p.grad is not None}
    o.opt.step(o.V.trainable())
    after = {k: v.detach() for k, v in {**o.V.v, **o.V.buffers}.items()}
    return logits.detach(), loss.detach(), grads, after


def check_fixture(path):
    with np.load(path, allow_pickle=False) as z:
        data = {k: z[k] for k in z.files}
    meta = json.loads(str(data["meta"]))
    sect = lambda p: {k[len(p):]: v for k, v in data.items() if k.startswith(p)}  # noqa: E731
    var0, var1, grad, feed, extra = sect("var0/"), sect("var1/"), sect("grad/"), sect("feed/"), sect("extra/")
    assert set(var0) == set(var1) and set(grad) <= set(var0)
    logits, loss, g, after = _run_oracle(meta, var0, feed, extra)
    assert set(after) == set(var0), (sorted(set(after) ^ set(var0)))
    if "logits" in data:
        np.testing.assert_allclose(logits.numpy(), data["logits"].reshape(-1), rtol=2e-5, atol=2e-6)
    assert abs(float(loss) - float(data["loss"])) < 2e-6 * max(1.0, abs(float(data["loss"])))
    assert set(g) == set(grad), sorted(set(g) ^ set(grad))
    for k, ref in grad.items():
        mine = g[k].numpy().reshape(ref.shape)
        np.testing.assert_allclose(mine, ref, rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()) + 1e-9, err_msg=f"grad {k}")
    lr, use_bn = meta["hyper"]["lr"], meta["hyper"]["use_bn"]
    for k, ref in var1.items():
        # a bias in front of a BatchNorm has a zero gradient up to rounding; Adam turns rounding noise into a
        # step of up to lr in either direction -> those are bounded by the step size, everything else is tight
        noise = use_bn and k.endswith("/bias") and "_layer" in k
        atol = 2.2 * lr if noise else 2e-6
        np.testing.assert_allclose(after[k].numpy().reshape(ref.shape), ref, rtol=1e-5, atol=atol, err_msg=f"var1 {k}")
    return meta


# ---------------------------------------------------------------------------------------------------------
# fixtures of the same format, made from the oracle (proves the checker and the file format)
# ---------------------------------------------------------------------------------------------------------
def _glorot(rng, *shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def _bn(rng, W, prefix, n):
    W[f"{prefix}/gamma"] = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    W[f"{prefix}/beta"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    W[f"{prefix}/moving_mean"] = np.zeros(n, np.float32)
    W[f"{prefix}/moving_var"] = np.ones(n, np.float32)


def _mlp(rng, W, scope, d_in, hidden, use_bn):
    if use_bn:
        _bn(rng, W, f"{scope}/bn_in", d_in)
    for i, h in enumerate(hidden, start=1):
        W[f"{scope}/{scope}_layer{i}/kernel"] = _glorot(rng, d_in, h)
        W[f"{scope}/{scope}_layer{i}/bias"] = (0.05 * rng.standard_normal(h)).astype(np.float32)
        if use_bn and i != len(hidden):
            _bn(rng, W, f"{scope}/bn{i}", h)
        d_in = h


def _self_fixture(path, model, seed=0):
    rng = np.random.default_rng(seed)
    U, N, K, B, L = 40, 30, 8, 64, 6
    vocab = (2, 5, 4)                                                # sex, occupation | genre (+1 OOV slot each)
    offs = np.cumsum([0] + [v + 1 for v in vocab])
    S = int(offs[-1])
    hp = dict(embed_size=K, lr=1e-3, epsilon=1e-5, use_bn=True, n_users=U, n_items=N)
    W = {"user_embeds_var": _glorot(rng, U + 1, K), "sparse_embeds_var": _glorot(rng, S, K)}
    users, items = rng.integers(0, U, B), rng.integers(0, N, B)
    sparse = np.stack([offs[c] + rng.integers(0, vocab[c], B) for c in range(3)], axis=1)
    feed = {"user_indices": users, "item_indices": items, "is_training": np.asarray(True),
            "labels": (rng.random(B) < 0.5).astype(np.float32)}
    extra = {}
    if model in ("FM", "DeepFM"):
        W["item_embeds_var"] = _glorot(rng, N + 1, K)
        W["user_linear_var"], W["item_linear_var"] = _glorot(rng, U + 1, 1), _glorot(rng, N + 1, 1)
        W["sparse_linear_var"] = _glorot(rng, S, 1).reshape(-1)
        W["linear/kernel"], W["linear/bias"] = _glorot(rng, 5, 1), np.zeros(1, np.float32)
        feed["sparse_indices"] = sparse
        if model == "FM":
            _bn(rng, W, "bn", K)
            W["pair/kernel"], W["pair/bias"] = _glorot(rng, K, 1), np.zeros(1, np.float32)
        else:
            hp["hidden_units"] = [32, 16, 8]
            _mlp(rng, W, "mlp", 5 * K, hp["hidden_units"], True)
            W["out/kernel"], W["out/bias"] = _glorot(rng, 1 + K + 8, 1), np.zeros(1, np.float32)
    elif model == "DIN":
        hp.update(hidden_units=[32, 16, 8], max_seq_len=L)
        W["item_embeds_var"] = _glorot(rng, N + 1, K)
        extra["item_sparse_unique"] = (offs[2] + rng.integers(0, vocab[2], (N + 1, 1))).astype(np.int64)
        feed["sparse_indices"] = sparse
        lens = rng.integers(1, L + 1, B)
        seqs = rng.integers(0, N, (B, L))
        seqs[np.arange(L)[None, :] >= lens[:, None]] = N
        feed["user_interacted_seq"], feed["user_interacted_len"] = seqs.astype(np.int32), lens.astype(np.int32)
        _mlp(rng, W, "attention", 4 * 2 * K, [16, 1], False)
        _mlp(rng, W, "mlp", 5 * K + 2 * K, hp["hidden_units"], True)
        W["out/kernel"], W["out/bias"] = _glorot(rng, 8, 1), np.zeros(1, np.float32)
    else:
        hp.update(hidden_units=[32, 16], loss_type="cross_entropy")
        W["item_embeds_var"] = _glorot(rng, N, K)
        feed["user_sparse_indices"], feed["item_sparse_indices"] = sparse[:, :2], sparse[:, 2:]
        _mlp(rng, W, "user_tower", 3 * K, hp["hidden_units"], True)
        _mlp(rng, W, "item_tower", 2 * K, hp["hidden_units"], True)
    meta = {"model": model, "hyper": hp, "tf_version": None, "source": "oracle-self-test"}
    logits, loss, g, after = _run_oracle(meta, W, feed, extra)
    out = {"meta": np.asarray(json.dumps(meta)), "logits": logits.numpy(), "loss": loss.numpy()}
    for k, v in W.items():
        out[f"var0/{k}"] = v
        out[f"var1/{k}"] = after[k].numpy().reshape(v.shape)
        if k in g:
            out[f"grad/{k}"] = g[k].numpy().reshape(v.shape)
    out.update({f"feed/{k}": v for k, v in feed.items()})
    out.update({f"extra/{k}": v for k, v in extra.items()})
    np.savez_compressed(path, **out)


@pytest.mark.parametrize("model", MODELS)
def test_checker_on_oracle_made_fixture(tmp_path, model):
    p = tmp_path / f"tf_{model.lower()}.npz"
    _self_fixture(p, model)
    assert check_fixture(p)["source"] == "oracle-self-test"
    # and it is a checker: a perturbed gradient / logit is caught
    with np.load(p) as z:
        data = {k: z[k] for k in z.files}
    k = next(k for k in data if k.startswith("grad/") and k.endswith("kernel"))
    data[k] = data[k] * (1 + 1e-3)
    np.savez_compressed(p, **data)
    with pytest.raises(AssertionError):
        check_fixture(p)


# ---------------------------------------------------------------------------------------------------------
# the pin itself
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", MODELS)
def test_oracle_vs_tensorflow(model):
    p = GOLDEN / f"tf_{model.lower()}.npz"
    if not p.exists():
        pytest.skip(f"PARITY UNPINNED for {model}: {p.name} is absent — {HOWTO}")
    meta = check_fixture(p)
    assert meta["source"] == "tensorflow" and meta["tf_version"]


@pytest.mark.parametrize("model", MODELS)
def test_reference_variable_file_is_fully_named(model):
    """`<name>_tf_variables.npz` as the reference's `save_tf_variables` writes it (utils/save_load.py:70-80):
    every non-slot variable gets an oracle name, and they are the names the fixture carries."""
    p, fx = GOLDEN / f"tf_{model.lower()}_tf_variables.npz", GOLDEN / f"tf_{model.lower()}.npz"
    if not (p.exists() and fx.exists()):
        pytest.skip(f"PARITY UNPINNED for {model}: {p.name} is absent — {HOWTO}")
    with np.load(p) as z:
        names = canonical(model, list(z.files))
        after = {names[k]: z[k] for k in z.files if k in names}
    with np.load(fx) as z:
        want = {k[5:]: z[k] for k in z.files if k.startswith("var1/")}
    assert set(after) == set(want)
    for k in want:
        np.testing.assert_array_equal(after[k], want[k])


def test_generator_dry_run_builds_data_model_and_feeds():
    """`python -m oracle.make_tf_golden --dry-run`: everything of the generator up to `sess.run` — the synthetic data set through
    the reference's `DatasetFeat`, its model classes' `build_model` (under the import stub of TensorFlow), its batch loader and
    feed-dict code — for all four models.  Needs the reference checkout (build container only)."""
    import subprocess
    import sys

    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("the reference checkout is not on this box (the dry run imports it)")
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, "-m", "oracle.make_tf_golden", "--dry-run"], cwd=root, capture_output=True, text=True,
                       timeout=600, env={**__import__("os").environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    for model, keys in (("FM", ("sparse_indices", "labels")), ("DeepFM", ("sparse_indices",)),
                        ("DIN", ("user_interacted_seq", "user_interacted_len")),
                        ("TwoTower", ("user_sparse_indices", "item_sparse_indices"))):
        block = out.split(f"[{model}] feed:")[1]
        for k in keys:
            assert k in block.split("] feed:")[0], (model, k)


# ---------------------------------------------------------------------------------------------------------
# the checksum manifest (tests/golden/TF_MANIFEST.json, written by `python -m oracle.tf_manifest --write`)
# ---------------------------------------------------------------------------------------------------------
def test_manifest_matches_the_committed_fixtures():
    """Every TensorFlow-made fixture in tests/golden/ is listed in the manifest with its sha256 (and vice versa): a fixture cannot
    arrive, change or disappear without a reviewable manifest diff.  With no fixture the manifest must say "unpinned"."""
    from oracle import tf_manifest as TM

    assert TM.MANIFEST.exists(), "tests/golden/TF_MANIFEST.json is missing: python -m oracle.tf_manifest --write"
    committed = json.loads(TM.MANIFEST.read_text())
    now = TM.scan()
    assert committed["expected_files"] == TM.EXPECTED
    assert sorted(committed["files"]) == sorted(now["files"]), "fixtures and manifest list different files: " + HOWTO
    for name, ent in now["files"].items():
        assert committed["files"][name]["sha256"] == ent["sha256"], f"{name} does not match its manifest checksum"
        assert name in TM.EXPECTED, f"{name}: not a file oracle/make_tf_golden.py writes"
    assert committed["status"] == now["status"]
    if now["status"] == "unpinned":
        assert not now["files"]


def test_pin_script_names_the_generator_and_the_reference_pin():
    """scripts/pin_tf_half.sh is the ONE command: it must call the generator, the manifest writer and this test, with the
    TensorFlow range the reference pins (requirements.txt:5)."""
    sh = (Path(__file__).resolve().parent.parent / "scripts" / "pin_tf_half.sh").read_text()
    for needle in ("oracle.make_tf_golden", "oracle.tf_manifest --write", "tests/test_tf_golden_cpu.py", "tensorflow==2.12"):
        assert needle in sh, needle
