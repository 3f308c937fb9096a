"""ctypes loader of `lib/liblibreco_host.so` (hostsrc/host_loops.c): C versions of the host loops that
must consume Python's `random` generator draw for draw.  An ACCELERATOR of host code only — the
Python loops in batch/sequence.py and sampling/negatives.py define the behaviour and are used when
the library has not been built; nothing of the HIP hot path lives here."""
from __future__ import annotations

import ctypes as C
import random
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "liblibreco_host.so"
ABI_VERSION = 3
_lib = None
_tried = False

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def load():
    """The library, or None when it is not built / of another ABI."""
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if LIB_PATH.exists():
        lib = C.CDLL(str(LIB_PATH))
        lib.lrh_abi_version.restype = C.c_int
        if lib.lrh_abi_version() == ABI_VERSION:
            lib.lrh_randrange_stream.restype = C.c_int
            lib.lrh_randrange_stream.argtypes = [_u32p, C.POINTER(C.c_int32), _i64p, C.c_int64, _i64p]
            lib.lrh_negatives_unconsumed.restype = C.c_int
            lib.lrh_negatives_unconsumed.argtypes = [_u32p, C.POINTER(C.c_int32), _i64p, _i64p, _i64p, _i64p,
                                                     C.c_int64, C.c_int64, C.c_int32, C.c_int32, _i64p]
            lib.lrh_merge_pointwise_u32.restype = C.c_int
            lib.lrh_merge_pointwise_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                                    C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), _i64p]
            lib.lrh_gather_rows_u32.restype = C.c_int
            lib.lrh_gather_rows_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_int]
            lib.lrh_seq_windows_i32.restype = C.c_int
            lib.lrh_seq_windows_i32.argtypes = [np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), _i64p, C.c_int64,
                                                _i64p, _i64p, C.c_int64, C.c_int, C.c_int32]
            lib.lrh_pair_positions.restype = C.c_int
            lib.lrh_pair_positions.argtypes = [_i64p, _i64p, _i64p, C.c_int64, C.c_int64, _i64p, _i64p, C.c_int64, _i64p]
            _lib = lib
    return _lib


class _PyRandomState:
    """`random.getstate()` as (uint32[624], position) and back."""

    def __enter__(self):
        version, internal, self.gauss = random.getstate()
        assert version == 3 and len(internal) == 625
        self.mt = np.array(internal[:624], dtype=np.uint32)
        self.pos = C.c_int32(internal[624])
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            random.setstate((3, tuple(self.mt.tolist()) + (int(self.pos.value),), self.gauss))
        return False


def randrange_stream(widths):
    """`[random.randrange(0, n) for n in widths]` as an int64 array (None: library unavailable)."""
    lib = load()
    if lib is None:
        return None
    widths = np.ascontiguousarray(widths, dtype=np.int64)
    out = np.empty(len(widths), dtype=np.int64)
    with _PyRandomState() as st:
        if lib.lrh_randrange_stream(st.mt, C.byref(st.pos), widths, len(widths), out) != 0:
            raise ValueError("empty range for randrange()")
    return out


def negatives_unconsumed(cons_ptr, cons_items, users, items, n_items, num_neg, tolerance):
    lib = load()
    if lib is None:
        return None
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    out = np.empty(len(users) * num_neg, dtype=np.int64)
    with _PyRandomState() as st:
        lib.lrh_negatives_unconsumed(st.mt, C.byref(st.pos), cons_ptr, cons_items, users, items, len(users),
                                     int(n_items), int(num_neg), int(tolerance), out)
    return out


def merge_pointwise(batch_feats, item_rows, i_cols, items, k):
    """`[n_pos * k, n_cols]` feature block of a pointwise batch in one pass (see lrh_merge_pointwise_u32); None when
    the library is unavailable or the arrays are not 4-byte C-contiguous matrices of one dtype."""
    lib = load()
    if lib is None or batch_feats.dtype != item_rows.dtype or batch_feats.dtype.itemsize != 4:
        return None
    if not (batch_feats.flags.c_contiguous and item_rows.flags.c_contiguous) or batch_feats.ndim != 2 or item_rows.ndim != 2:
        return None
    n_pos, n_cols = batch_feats.shape
    items = np.ascontiguousarray(items, dtype=np.int64)
    cols = np.ascontiguousarray(i_cols, dtype=np.int32)
    if len(items) != n_pos * k or item_rows.shape[1] != len(cols):
        return None
    out = np.empty((n_pos * k, n_cols), dtype=batch_feats.dtype)
    rc = lib.lrh_merge_pointwise_u32(out.ctypes.data, batch_feats.ctypes.data, n_pos, int(k), n_cols, item_rows.ctypes.data,
                                     item_rows.shape[0], len(cols), cols, items)
    if rc != 0:
        raise IndexError("item id / column index out of range in the pointwise feature merge")
    return out


def gather_rows(base, idx):
    """`base[idx]` for a C-contiguous 2-D matrix of 4-byte elements and an integer index array (numpy otherwise)."""
    lib = load()
    if lib is None or base.ndim != 2 or base.dtype.itemsize != 4 or not base.flags.c_contiguous or \
            not isinstance(idx, np.ndarray) or idx.ndim != 1 or idx.dtype.kind not in "iu":
        return base[idx]
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if len(idx) and idx.min() < 0:          # numpy semantics for negative indices (wrap): same result with or without the library
        return base[idx]
    out = np.empty((len(idx), base.shape[1]), dtype=base.dtype)
    if lib.lrh_gather_rows_u32(out.ctypes.data, base.ctypes.data, base.shape[0], idx, len(idx), base.shape[1]) != 0:
        raise IndexError("row index out of range")
    return out


def seq_windows(hist, start, count, width, pad):
    """int32 [n, width]: row r = hist[start[r] : start[r] + count[r]] left-aligned, `pad` behind it (None: library
    unavailable — the caller's numpy expression is the definition)."""
    lib = load()
    if lib is None or hist.dtype != np.int64 or not hist.flags.c_contiguous:
        return None
    start = np.ascontiguousarray(start, dtype=np.int64)
    count = np.ascontiguousarray(count, dtype=np.int64)
    out = np.empty((len(start), int(width)), dtype=np.int32)
    if lib.lrh_seq_windows_i32(out, hist, len(hist), start, count, len(start), int(width), int(pad)) != 0:
        raise IndexError("history window out of range")
    return out


def pair_positions(keys, first_pos, kptr, stride, users, items):
    """int64 [n]: first_pos of the (user, item) pairs found in the per-user sorted key table, -1 otherwise (None: library
    unavailable or a user outside the table — the caller's searchsorted expression is the definition)."""
    lib = load()
    if lib is None:
        return None
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    out = np.empty(len(users), dtype=np.int64)
    if lib.lrh_pair_positions(keys, first_pos, kptr, len(kptr) - 1, int(stride), users, items, len(users), out) != 0:
        return None
    return out
