"""Buffer-set bookkeeping of the hand-written DeepFM tail (`layers/tail.py`) on CPU tensors: one set of activation /
partial-sum buffers per batch size, parked (kept alive at the same addresses) while another size is in use — a
hipGraph captured for one batch shape holds raw addresses of its set — and the owner is told when a set is dropped."""
import torch

from librecommender_amd.layers import DenseParams, DenseStack, TFDense
from librecommender_amd.layers.tail import DeepFMTail


def make_tail():
    dev = torch.device("cpu")
    P = DenseParams(dev, 1)
    mlp = DenseStack(P, "mlp", 4 * 16, (32, 16), True, 0.0)
    linear = TFDense(P, "linear", 4, 1)
    out = TFDense(P, "out", 1 + 16 + 16, 1)
    P.finalize()
    return DeepFMTail(P, mlp, linear, out, 4, 16, dev)


def test_sets_are_parked_and_restored_at_the_same_addresses():
    tail = make_tail()
    tail._alloc(128)
    a_ptrs = (tail.gz1.data_ptr(), tail.head_partial.data_ptr(), tail.z[1].data_ptr())
    tail._jobs_dev = torch.zeros(8)                       # stands for the device-resident job table of this set
    a_jobs = tail._jobs_dev
    tail._alloc(40)                                       # the shorter last batch of an epoch
    assert tail._B == 40 and tail.gz1.shape[0] == 40 and tail._jobs_dev is None and tail._jobs == []
    assert 128 in tail._sets and tail._sets[128]["_jobs_dev"] is a_jobs      # alive, not re-used
    b_ptr = tail.gz1.data_ptr()
    tail._alloc(128)                                      # next epoch: the full-batch set comes back unchanged
    assert (tail.gz1.data_ptr(), tail.head_partial.data_ptr(), tail.z[1].data_ptr()) == a_ptrs
    assert tail._jobs_dev is a_jobs and tail.nblk == 2 and 40 in tail._sets and 128 not in tail._sets
    tail._alloc(40)
    assert tail.gz1.data_ptr() == b_ptr


def test_owner_is_notified_when_a_set_is_dropped():
    tail = make_tail()
    dropped = []
    tail.on_release = lambda: dropped.append(True)
    for B in range(64, 64 * (DeepFMTail.MAX_SETS + 3), 64):
        tail._alloc(B)
    assert dropped and len(tail._sets) <= DeepFMTail.MAX_SETS
