"""Checkpoint hygiene of the node-partitioned graph nets (`distributed._node_shard_files`): a directory with a stale or
incomplete shard set is refused instead of being overlaid silently (round-3 advisor finding)."""
import os

import numpy as np
import pytest

from librecommender_amd.distributed import _node_shard_files


def _touch(d, name):
    np.savez(os.path.join(d, name), x=np.zeros(1))


def test_exact_set_of_the_current_world_size_is_preferred(tmp_path):
    d = str(tmp_path)
    for r in range(2):
        _touch(d, f"m_nodes_shard{r}of2.npz")
    for r in range(3):
        _touch(d, f"m_nodes_shard{r}of3.npz")
    got = _node_shard_files(d, "m", 2)
    assert [(r, w) for r, w, _ in got] == [(0, 2), (1, 2)]
    got = _node_shard_files(d, "m", 3)
    assert [(r, w) for r, w, _ in got] == [(0, 3), (1, 3), (2, 3)]


def test_other_world_size_is_taken_only_when_it_is_the_only_one(tmp_path):
    d = str(tmp_path)
    for r in range(2):
        _touch(d, f"m_nodes_shard{r}of2.npz")
    assert [w for _, w, _ in _node_shard_files(d, "m", 4)] == [2, 2]          # re-sharded 2 -> 4
    for r in range(3):
        _touch(d, f"m_nodes_shard{r}of3.npz")
    with pytest.raises(ValueError, match="world sizes"):
        _node_shard_files(d, "m", 4)


def test_missing_shard_and_missing_checkpoint(tmp_path):
    d = str(tmp_path)
    with pytest.raises(FileNotFoundError):
        _node_shard_files(d, "m", 2)
    _touch(d, "m_nodes_shard0of3.npz")
    _touch(d, "m_nodes_shard2of3.npz")
    with pytest.raises(ValueError, match="incomplete"):
        _node_shard_files(d, "m", 3)
    # another model's shards in the same directory are not picked up
    _touch(d, "other_nodes_shard0of1.npz")
    with pytest.raises(ValueError, match="incomplete"):
        _node_shard_files(d, "m", 2)


def test_shard_arrays_are_read_by_the_saved_key_list_only(tmp_path):
    """Round-4 advisor finding: `shard_array` preferred ANY side file `<stem>.<key>.npy` — a re-save into the same directory
    with another key set (a table saved first with linear weights, later without) left a stale file that loaded silently.
    The meta file's `keys` list is what a checkpoint holds; anything else is refused."""
    import numpy as np
    import pytest

    from librecommender_amd.parallel import ShardedFieldTables

    stem = tmp_path / "tables_shard0of1"
    np.save(f"{stem}.embed.npy", np.ones((3, 2), np.float32))
    np.save(f"{stem}.lin.npy", np.full((3, 1), 7.0, np.float32))          # stale: not part of the save below
    np.savez(f"{stem}.npz", V=np.int64(3), K=np.int64(2), rank=np.int64(0), world=np.int64(1), keys=np.asarray(["embed"]))
    with np.load(f"{stem}.npz") as z:
        assert ShardedFieldTables.shard_array(str(tmp_path), "tables", 0, 1, "embed", z).shape == (3, 2)
        with pytest.raises(FileNotFoundError, match="lists"):
            ShardedFieldTables.shard_array(str(tmp_path), "tables", 0, 1, "lin", z)


def test_interrupted_save_leaves_the_previous_checkpoint_whole(tmp_path, monkeypatch):
    """Round-5 advisor finding: `save_shard` replaced the (untagged) side files in place before it renamed the meta file — a
    crash in between left NEW `embed` beside OLD moments under the OLD meta, and `load_shard` took the mix.  Arrays now carry
    the tag of their save, the meta names the tag and is renamed last: a save that dies after some arrays were written changes
    nothing a loader sees; the next complete save removes the orphans."""
    import numpy as np
    import torch

    from librecommender_amd.parallel import ShardedFieldTables
    from tests.oracle_kernels import OracleKernels

    class _NoGroup(ShardedFieldTables):          # world size 1 without a process group
        def __init__(self, V, K):
            self.V, self.K, self.rank, self.world, self.device = V, K, 0, 1, torch.device("cpu")
            self.embed = torch.zeros((V, K))
            self.m, self.v = torch.zeros((V, K)), torch.zeros((V, K))
            self.lin = self.lin_m = self.lin_v = None

    t = _NoGroup(5, 2)
    t.embed.fill_(1.0); t.m.fill_(10.0); t.v.fill_(100.0)
    t.save_shard(str(tmp_path))
    first = sorted(p.name for p in tmp_path.iterdir())
    assert first == ["tables_shard0of1.npz", "tables_shard0of1.s1.embed.npy", "tables_shard0of1.s1.m.npy", "tables_shard0of1.s1.v.npy"]

    # second save dies after `embed` and `m` were written (before the meta)
    t.embed.fill_(2.0); t.m.fill_(20.0); t.v.fill_(200.0)
    real_save, n = np.save, {"calls": 0}

    def dying_save(f, a, *args, **kw):
        n["calls"] += 1
        if n["calls"] == 3:
            raise OSError("disk full")
        return real_save(f, a, *args, **kw)

    monkeypatch.setattr(np, "save", dying_save)
    try:
        t.save_shard(str(tmp_path))
        raise AssertionError("the save should have failed")
    except OSError:
        pass
    monkeypatch.setattr(np, "save", real_save)
    r = _NoGroup(5, 2)
    r.load_shard(str(tmp_path))
    assert float(r.embed[0, 0]) == 1.0 and float(r.m[0, 0]) == 10.0 and float(r.v[0, 0]) == 100.0     # the FIRST save, whole

    # a complete save takes over and removes the orphans of the failed one and the files of the first
    t.save_shard(str(tmp_path))
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["tables_shard0of1.npz", "tables_shard0of1.s2.embed.npy", "tables_shard0of1.s2.m.npy", "tables_shard0of1.s2.v.npy"]
    r.load_shard(str(tmp_path))
    assert float(r.embed[0, 0]) == 2.0 and float(r.v[0, 0]) == 200.0
    # a meta whose tagged array is gone is an error, not a silent fall-back
    (tmp_path / "tables_shard0of1.s2.m.npy").unlink()
    import pytest

    with pytest.raises(FileNotFoundError, match="missing"):
        r.load_shard(str(tmp_path))
