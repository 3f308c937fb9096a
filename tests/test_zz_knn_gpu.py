"""Embedding kNN served by `lr_score_topk_f32` (`-m gpu`) vs the reference's neighbours
(tests/golden/knn.npz); host logic is covered on CPU in tests/test_knn_cpu.py.  (Added after this
round's GPU budget was spent: first run is the driver's.)"""
import numpy as np
import pytest

from librecommender_amd.algorithms import LightGCN
from librecommender_amd.data import DataInfo
from tests.test_knn_cpu import check_against_reference

pytestmark = pytest.mark.gpu


def test_knn_on_device_matches_reference(dev, golden_dir):
    d = golden_dir / "refckpt"
    model = LightGCN.load(str(d), "lgcn", DataInfo.load(str(d), "lgcn"))
    check_against_reference(model, np.load(golden_dir / "knn.npz"))
