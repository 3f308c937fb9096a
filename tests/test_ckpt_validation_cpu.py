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
