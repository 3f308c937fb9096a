"""Import stub so the reference's numpy/torch path can be imported without TensorFlow
(TEST INFRASTRUCTURE).  Nothing here computes anything."""
import sys
import types


class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


compat = types.ModuleType("tensorflow.compat")
v1 = _Any()
v1.__version__ = "2.12.0"
compat.v1 = v1
sys.modules["tensorflow.compat"] = compat
__version__ = "2.12.0"
