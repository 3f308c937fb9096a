"""Round-4 tilings of the fused lookup + first-layer kernels (csrc/deepfm_l1.hip: 64 samples per workgroup, one wave per
SIMD, two accumulators per wave) against the fp64 restatements of `oracle/ops_np.py` AND bit for bit against the
32-sample kernels (per output element both run the same k-ordered f32 fma chain), pinned through
`lr_deepfm_l1_tile_override`.  Shapes cover partial last tiles, fewer fields than the pipeline is deep, the id-chunk
rotation (F > 64), ids outside the table and the no-linear-table form."""
import numpy as np
import pytest
import torch

from librecommender_amd import _lib, ops
from oracle import ops_np
from tests.test_deepfm_fused_gpu import make_case, t

pytestmark = pytest.mark.gpu

WIDE = [(64, 128), (32, 128), (64, 256)]


@pytest.fixture
def tile(f32_chain):
    lib = _lib.load()

    def pin(ts):
        lib.lr_deepfm_l1_tile_override(int(ts))
    yield pin
    lib.lr_deepfm_l1_tile_override(0)


@pytest.mark.parametrize("K,H1", WIDE)
@pytest.mark.parametrize("B,F", [(64, 1), (100, 2), (257, 3), (200, 4), (129, 5), (1000, 23), (320, 70), (192, 131)])
def test_wide_fwd_equals_fp64_and_the_32_sample_kernel(dev, tile, K, H1, B, F):
    rng = np.random.default_rng(B * 11 + F + K + H1)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    WpA, _ = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    args = (t(table, dev), t(idx, dev), WpA, t(bias, dev), H1)
    tile(32)
    z32, p32, s32, l32 = ops.deepfm_l1_fwd(*args, lin=t(lin, dev))
    zb32, pb32, sb32, _ = ops.deepfm_l1_fwd(args[0], args[1], WpA, None, H1)
    tile(64)
    z64, p64, s64, l64 = ops.deepfm_l1_fwd(*args, lin=t(lin, dev))
    zb64, pb64, sb64, lb64 = ops.deepfm_l1_fwd(args[0], args[1], WpA, None, H1)
    assert lb64 is None
    for a, b in ((z32, z64), (p32, p64), (s32, s64), (l32, l64), (zb32, zb64), (pb32, pb64), (sb32, sb64)):
        assert torch.equal(a, b)
    o_z1, o_pair, o_fsum, o_lin = ops_np.deepfm_l1_fwd(table, lin, idx, Wp, bias)
    scale = float(np.abs(o_z1).max()) + 1.0
    np.testing.assert_allclose(z64.cpu().numpy(), o_z1, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(s64.cpu().numpy(), o_fsum, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(p64.cpu().numpy(), o_pair, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(l64.cpu().numpy(), o_lin.astype(np.float32))


def test_wide_fwd_is_the_automatic_choice_from_one_workgroup_per_cu(dev, tile):
    """B >= 0.75 * 256 * 64 selects the wide kernels; results do not depend on the choice."""
    rng = np.random.default_rng(5)
    B, F, K, H1 = 12_288, 6, 64, 128
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    WpA, _ = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    args = (t(table, dev), t(idx, dev), WpA, t(bias, dev), H1)
    tile(0)
    auto = ops.deepfm_l1_fwd(*args, lin=t(lin, dev))
    tile(32)
    ref = ops.deepfm_l1_fwd(*args, lin=t(lin, dev))
    for a, b in zip(auto, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("K,H1", [(64, 128), (32, 128)])
@pytest.mark.parametrize("B,F", [(64, 1), (130, 2), (100, 3), (257, 4), (200, 5), (1000, 9), (320, 70)])
def test_wide_dgrad_equals_fp64_and_the_32_sample_kernel(dev, tile, K, H1, B, F):
    rng = np.random.default_rng(B + 5 * F + K + H1)
    frs, V, idx, table, lin, Wp, bias = make_case(rng, B, F, K, H1, dev)
    gz = rng.standard_normal((B, H1)).astype(np.float32)
    gl = rng.standard_normal(B).astype(np.float32)
    wp = rng.standard_normal(K).astype(np.float32)
    fsum = rng.standard_normal((B, K)).astype(np.float32)
    _, _, _, slotT = ops_np.segments_fields(idx, frs)
    _, WpB = ops.deepfm_l1_pack(t(Wp, dev), F, K)
    res = {}
    for ts in (32, 64):
        tile(ts)
        out = torch.zeros((B * F + 1, K), device=dev)
        a = ops.deepfm_l1_dgrad(t(gz, dev), WpB, K, F, t(slotT, dev), gl=t(gl, dev), wp=t(wp, dev), fsum=t(fsum, dev), out=out)[:B * F].clone()
        out2 = torch.zeros((B * F + 1, K), device=dev)
        b = ops.deepfm_l1_dgrad(t(gz, dev), WpB, K, F, t(slotT, dev), out=out2)[:B * F].clone()     # no FM term
        res[ts] = (a, b)
    assert torch.equal(res[32][0], res[64][0]) and torch.equal(res[32][1], res[64][1])
    want = ops_np.deepfm_l1_dgrad(gz, Wp, K, gl, wp, fsum, slotT)
    np.testing.assert_allclose(res[64][0].cpu().numpy(), want, rtol=1e-5, atol=1e-5 * (float(np.abs(want).max()) + 1.0))
