"""bench.py's synthetic id layout: every field's global rows stay inside the row range that the
field-parallel net assigns to the field's owner (an off-by-one here would be an out-of-bounds table
access on the device, not an exception)."""
import numpy as np

import bench


def test_global_rows_fall_into_their_field_ranges():
    for cfg in (dict(bench.CFG, batch=4096), dict(bench.CFG, n_users=50_000, n_items=50_000, n_sparse_fields=20, vocab=2_000, batch=2_048)):
        frs = bench.field_row_start(cfg)
        F = 2 + cfg["n_sparse_fields"]
        assert len(frs) == F + 1 and frs[0] == 0
        assert frs[-1] == cfg["n_users"] + 1 + cfg["n_items"] + 1 + cfg["n_sparse_fields"] * (cfg["vocab"] + 1)
        for users, items, sparse, _ in bench.make_batches(cfg, 2, seed=1):
            rows = bench.global_rows(cfg, users, items, sparse).astype(np.int64)
            assert rows.shape == (cfg["batch"], F)
            assert (rows >= frs[:-1][None, :]).all() and (rows < frs[1:][None, :] - 1 + 1).all()
            assert (rows < frs[1:][None, :] - 1).all()          # the last row of a field is its OOV row: never sampled
        for world in (1, 2, 4, 8):
            bounds = [(F * r) // world for r in range(world + 1)]
            assert bounds[0] == 0 and bounds[-1] == F and all(b > a for a, b in zip(bounds, bounds[1:]))


def test_weight_grad_slab_rule_keeps_single_gpu_decisions():
    """layers/dense.py: the measured single-GPU choices stay as they were; only global batches of
    >= 32768 rows (field-parallel first layer) get the extrapolated slab counts."""
    import torch

    from librecommender_amd.layers.dense import weight_grad, weight_grad_slabs as slabs
    assert slabs(16384, 128, 64) == 64 and slabs(16384, 64, 32) == 64 and slabs(16384, 97, 1) == 64
    assert slabs(16384, 12928, 128) == 1 and slabs(2048, 128, 64) == 1 and slabs(16384, 400, 128) == 1
    assert {w: slabs(16384 * w, -(-202 // w) * 64, 128) for w in (2, 4, 8)} == {2: 8, 4: 16, 8: 32}
    x, g = torch.randn(32768, 200), torch.randn(32768, 128)
    torch.testing.assert_close(weight_grad(x, g), x.t() @ g, rtol=1e-4, atol=1e-2)


def test_lightgcn_bench_graph_holds_exactly_the_asked_number_of_distinct_pairs():
    """bench_workloads.distinct_interactions: cfg 5's graph is E DISTINCT (user, item) pairs (nnz = 2 E after the
    reference's de-duplicating Laplacian build), reproducible from the seed, ids inside their ranges."""
    import torch

    import bench_workloads as bw

    dev = torch.device("cpu")
    for E, nu, ni in ((50_000, 3_000, 2_000), (1_000, 40, 30)):          # the second asks for most of the 1,200 possible pairs
        g = torch.Generator(device=dev).manual_seed(42)
        eu, ei = bw.distinct_interactions(E, nu, ni, g, dev)
        assert eu.dtype == ei.dtype == torch.int32 and eu.numel() == ei.numel() == E
        assert 0 <= int(eu.min()) and int(eu.max()) < nu and 0 <= int(ei.min()) and int(ei.max()) < ni
        assert torch.unique(eu.long() * ni + ei.long()).numel() == E
        g2 = torch.Generator(device=dev).manual_seed(42)
        eu2, ei2 = bw.distinct_interactions(E, nu, ni, g2, dev)
        assert torch.equal(eu, eu2) and torch.equal(ei, ei2)


def test_bench_json_line_is_alone_on_stdout_when_a_library_prints_through_c_stdio():
    """bench._emit: output written through C stdio while fd 1 points at stderr (RCCL's banner) must not reach stdout."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent('''
        import ctypes, os, sys
        sys.path.insert(0, %r)
        import bench
        sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)
        ctypes.CDLL(None).printf(b"banner line\\n")
        bench._emit({"metric": "m"}, 0, fd)
    ''') % (bench.__file__.rsplit("/", 1)[0],)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout == '{"metric": "m"}\n' and "banner line" in r.stderr
