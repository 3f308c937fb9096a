"""Host -> device hand-over used by the nets: batches may arrive as numpy arrays (host collators,
the reference's wire format) or as device tensors (device-side sampling / collation)."""
from __future__ import annotations

import numpy as np
import torch


def to_device(x, device, dtype=None) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x.to(device)
    else:
        t = torch.as_tensor(np.ascontiguousarray(x), device=device)
    return t if dtype is None else t.to(dtype)
