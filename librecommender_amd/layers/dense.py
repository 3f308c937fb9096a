"""Dense part of the feature models: ``dense_nn`` / ``tf_dense`` / ``tf.layers.batch_normalization``
(layers/dense.py:12-80 of the reference) on device.

These are plain dense contractions: they run through torch (hipBLASLt) with autograd — they
are not part of the hand-written hot path.  All dense parameters live in ONE flat fp32 buffer
(views per tensor) with one flat gradient buffer, so the TF-style Adam update of every dense
parameter is a single ``lr_adam_dense_f32`` launch.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from .. import ops


class DenseParams:
    """Flat storage for dense-layer parameters (+ flat grads, Adam moments)."""

    def __init__(self, device: torch.device, seed: int = 42):
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + 1)
        self._specs = []  # (name, shape, init)
        self.params = {}
        self.flat = self.grad = self.m = self.v = None

    def add(self, name: str, shape: Sequence[int], init: str) -> str:
        assert self.flat is None, "finalize() already called"
        self._specs.append((name, tuple(shape), init))
        return name

    def finalize(self) -> None:
        total = sum(math.prod(s) for _, s, _ in self._specs)
        pad = (-total) % 4
        self.flat = torch.zeros(total + pad, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        off = 0
        for name, shape, init in self._specs:
            n = math.prod(shape)
            p = self.flat[off:off + n].view(shape)
            if init == "glorot_uniform":
                fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
                limit = math.sqrt(6.0 / (fan_in + fan_out))
                p.uniform_(-limit, limit, generator=self.gen)
            elif init == "ones":
                p.fill_(1.0)
            p.requires_grad_(True)
            p.grad = self.grad[off:off + n].view(shape)  # autograd accumulates in place here
            self.params[name] = p
            off += n

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.params[name]

    def zero_grad(self) -> None:
        self.grad.zero_()

    def adam_step(self, hp) -> None:
        with torch.no_grad():
            ops.adam_dense(self.flat.view(-1, 1), self.m.view(-1, 1), self.v.view(-1, 1), hp,
                           grows=self.grad)


class TFDense:
    """tf.layers.dense / tf.keras.layers.Dense: glorot_uniform kernel, zero bias (dense.py:52-80)."""

    def __init__(self, P: DenseParams, name: str, n_in: int, units: int):
        self.P = P
        self.w = P.add(f"{name}/kernel", (n_in, units), "glorot_uniform")
        self.b = P.add(f"{name}/bias", (units,), "zeros")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return torch.addmm(self.P[self.b], x, self.P[self.w])


class TFBatchNorm:
    """tf.layers.batch_normalization(training=is_training): momentum 0.99, epsilon 1e-3, batch
    statistics (biased variance) in training, moving averages otherwise (dense.py:31-41)."""

    def __init__(self, P: DenseParams, name: str, n: int, momentum: float = 0.99, eps: float = 1e-3):
        self.P = P
        self.gamma = P.add(f"{name}/gamma", (n,), "ones")
        self.beta = P.add(f"{name}/beta", (n,), "zeros")
        self.moving_mean = torch.zeros(n, dtype=torch.float32, device=P.device)
        self.moving_var = torch.ones(n, dtype=torch.float32, device=P.device)
        self.momentum, self.eps = momentum, eps

    def __call__(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        g, b = self.P[self.gamma], self.P[self.beta]
        if training:
            var, mean = torch.var_mean(x, dim=0, unbiased=False)
            with torch.no_grad():  # UPDATE_OPS (training/tf_trainer.py:122-123)
                self.moving_mean.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
                self.moving_var.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
        else:
            mean, var = self.moving_mean, self.moving_var
        return (x - mean) * (g * torch.rsqrt(var + self.eps)) + b


class DenseStack:
    """dense_nn(net, hidden_units, use_bn, bn_after_activation=True, dropout) — dense.py:12-49:
    optional input BN; Dense -> act -> BN -> dropout per layer; the LAST layer has no
    activation / BN / dropout."""

    def __init__(self, P: DenseParams, name: str, n_in: int, hidden_units: Sequence[int],
                 use_bn: bool = True, dropout_rate: float = 0.0, activation=F.relu):
        self.use_bn, self.dropout_rate, self.act = use_bn, dropout_rate or 0.0, activation
        self.bn_in = TFBatchNorm(P, f"{name}/bn_in", n_in) if use_bn else None
        self.layers: List[TFDense] = []
        self.bns: List[Optional[TFBatchNorm]] = []
        d = n_in
        for i, units in enumerate(hidden_units, start=1):
            self.layers.append(TFDense(P, f"{name}/{name}_layer{i}", d, units))
            last = i == len(hidden_units)
            self.bns.append(TFBatchNorm(P, f"{name}/bn{i}", units) if (use_bn and not last) else None)
            d = units
        self.n_out = d

    def __call__(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        if self.bn_in is not None:
            x = self.bn_in(x, training)
        for i, (layer, bn) in enumerate(zip(self.layers, self.bns)):
            x = layer(x)
            if i != len(self.layers) - 1:
                x = self.act(x)
                if bn is not None:
                    x = bn(x, training)
                if self.dropout_rate and training:
                    x = F.dropout(x, self.dropout_rate, training=True)
        return x
