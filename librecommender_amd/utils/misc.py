"""Small console helpers (`libreco/utils/misc.py:46-107`)."""
import functools
import time
from contextlib import contextmanager

_COLORS = {"red": 31, "green": 32, "yellow": 33, "blue": 34, "magenta": 35, "cyan": 36}


def colorize(text, color, bold=False, highlight=False):
    num = _COLORS.get(color, 37) + (10 if highlight else 0)
    attrs = [str(num)] + (["1"] if bold else [])
    return f"\x1b[{';'.join(attrs)}m{text}\x1b[0m"


@contextmanager
def time_block(name="block", verbose=1):
    """Prints the wall time of the block when it completes (nothing if it raises, as in the reference)."""
    t0 = time.perf_counter()
    yield
    if verbose > 0:
        print(f"{name} elapsed: {time.perf_counter() - t0:.3f}s")


def time_func(fn):
    """Decorator form of `time_block` (`misc.py:46-56`)."""
    @functools.wraps(fn)
    def timed(*args, **kwargs):
        t0 = time.perf_counter()
        out = fn(*args, **kwargs)
        print(f"{fn.__name__} elapsed: {time.perf_counter() - t0:.3f}s")
        return out

    return timed
