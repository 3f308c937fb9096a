"""Import the reference's own numpy/torch code (read-only at /root/reference) — TEST
INFRASTRUCTURE.  Only usable in the build container; never imported at GPU-test time.
Recipe: SURVEY.md Appendix A."""
import os
import sys
import types
from pathlib import Path

REFERENCE = Path(os.environ.get("LIBRECO_REFERENCE", "/root/reference"))
STUBS = Path(__file__).resolve().parent / "_stubs"


def available() -> bool:
    return (REFERENCE / "libreco" / "__init__.py").exists()


def load():
    """Make `import libreco...` resolve to the reference checkout (torch/numpy path only)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE}")
    sys.dont_write_bytecode = True
    for p in (str(STUBS), str(REFERENCE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import libreco  # noqa: F401

    if "libreco.algorithms" not in sys.modules:
        pkg = types.ModuleType("libreco.algorithms")
        pkg.__path__ = [str(REFERENCE / "libreco" / "algorithms")]  # skip __init__ (needs Cython .so)
        sys.modules["libreco.algorithms"] = pkg
    return libreco
