"""Serving export files against the reference's own serializers
(`libserving/serialization/{common,embed,online}.py`; fixture: `oracle.make_golden.gen_serving`)."""
import json

import pytest

from librecommender_amd.data import DatasetFeat, DatasetPure
from librecommender_amd.serving import save_embed, save_online
from oracle.make_golden import FEAT_KW, MULTI_KW, serving_stub_model, synthetic_frame


def _read_dir(d):
    return {f.name: json.loads(f.read_text()) for f in sorted(d.iterdir()) if f.suffix == ".json"}


def _expected(golden_dir, tag):
    g = json.loads((golden_dir / "serving.json").read_text())
    return {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(tag + "/")}


def test_save_embed_files_match_reference(golden_dir, tmp_path):
    _, info = DatasetPure.build_trainset(synthetic_frame()[["user", "item", "label"]])
    save_embed(str(tmp_path), serving_stub_model(info, "LightGCN"))
    got, want = _read_dir(tmp_path), _expected(golden_dir, "embed")
    assert sorted(got) == sorted(want)
    assert len(got["user_embed.json"]) == info.n_users and len(got["item_embed.json"]) == info.n_items
    assert got == want


@pytest.mark.parametrize("tag,kw", [("feat", FEAT_KW), ("multi", MULTI_KW)])
def test_save_online_files_match_reference(golden_dir, tmp_path, tag, kw):
    _, info = DatasetFeat.build_trainset(synthetic_frame(), **kw)
    model = serving_stub_model(info, "DIN", with_seq=True)
    saved = []
    model.save = lambda path, name, inference_only=False: saved.append((path, name, inference_only))
    export_dir = save_online(str(tmp_path), model, version=3)
    assert saved == [(export_dir, "din", True)] and export_dir.endswith("din/3")
    got, want = _read_dir(tmp_path), _expected(golden_dir, tag)
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name
    with pytest.raises(FileExistsError):
        save_online(str(tmp_path), model, version=3)


def test_export_rejects_bad_path():
    with pytest.raises(AssertionError):
        save_embed("", None)
