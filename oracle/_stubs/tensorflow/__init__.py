"""Import stub so the reference's numpy/torch path can be imported without TensorFlow
(TEST INFRASTRUCTURE).  Nothing here computes anything."""
import sys
import types


class _Any:
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return iter((_Any(), _Any()))

    def _arith(self, *a):
        return _Any()

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = _arith
    __neg__ = __getitem__ = __matmul__ = __rmatmul__ = __pow__ = _arith


compat = types.ModuleType("tensorflow.compat")
v1 = _Any()
v1.__version__ = "2.12.0"
compat.v1 = v1
sys.modules["tensorflow.compat"] = compat
__version__ = "2.12.0"
