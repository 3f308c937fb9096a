"""In-process embedding server (`serving/embed_server.py`) over the files `serving.save_embed` writes
(`-m gpu`): answers equal the model's own `recommend_user`, i.e. the exported vectors ranked by `lr_score_topk_f32`
with the consumed filter; unknown users are rejected like `libserving/sanic_serving/embed_deploy.py:27-28`."""
import numpy as np
import pytest

from librecommender_amd.algorithms import LightGCN
from librecommender_amd.data import DatasetPure
from librecommender_amd.serving import EmbedServer, InvalidUser, save_embed
from tests.test_api_gpu import movielens_like

pytestmark = pytest.mark.gpu


def test_embed_server_round_trip(dev, tmp_path):
    df = movielens_like(4000, 150, 120)
    train_data, info = DatasetPure.build_trainset(df)
    model = LightGCN("ranking", info, embed_size=16, n_epochs=1, lr=1e-2, batch_size=512, device=str(dev))
    model.fit(train_data, neg_sampling=True, verbose=0)
    save_embed(str(tmp_path), model)
    server = EmbedServer(str(tmp_path), device=str(dev))
    assert server.model_name == "LightGCN" and server.n_users == info.n_users and server.n_items == info.n_items
    # the library applies the reference's `ranking.py:38` rule (no filtering when n_rec + len(history) > n_items); the
    # serving loop always filters (embed_deploy.py:40-56): compare the two on users where the rule filters too
    light = [u for u in range(info.n_users) if 10 + len(info.user_consumed[u]) <= info.n_items][:6]
    heavy0 = [u for u in range(info.n_users) if 10 + len(info.user_consumed[u]) > info.n_items][:1]
    users = [info.id2user[u] for u in light + heavy0]
    got = server.recommend(users, 10)
    want = model.recommend_user(users, 10)
    assert len(light) == 6
    for u in users:
        if info.user2id[u] in light:
            assert got[u] == [int(i) for i in want[u]]
        assert len(got[u]) == min(10, info.n_items - len(set(info.user_consumed[info.user2id[u]])))
        consumed = {info.id2item[i] for i in info.user_consumed[info.user2id[u]]}
        assert not (set(got[u]) & consumed)
    with pytest.raises(InvalidUser):
        server.recommend(["nobody"], 5)
    # a request for more items than a heavy user has left returns what remains, none of it consumed
    heavy = max(range(info.n_users), key=lambda u: len(info.user_consumed[u]))
    left = info.n_items - len(set(info.user_consumed[heavy]))
    rec = server.recommend([info.id2user[heavy]], info.n_items)[info.id2user[heavy]]
    assert len(rec) == left and len(set(rec)) == left
    assert not (set(rec) & {info.id2item[i] for i in info.user_consumed[heavy]})
