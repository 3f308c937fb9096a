import numpy as np


def unflatten(flat):
    """[key, n, v0..v{n-1}, key, n, ...] -> dict(key -> list)."""
    out, i = {}, 0
    flat = np.asarray(flat).tolist()
    while i < len(flat):
        k, n = flat[i], flat[i + 1]
        out[int(k)] = [int(x) for x in flat[i + 2: i + 2 + n]]
        i += 2 + n
    return out
