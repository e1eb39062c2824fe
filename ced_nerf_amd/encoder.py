"""Time ("temporal") encoders with the interface of cednerf/encoder.py.

Inside the renderer these 9-wide encodings are evaluated by the fused HIP field kernel
(`time_feature` in csrc/field.hip); the modules below give callers of the reference API the same
objects (`latent_dim`, `forward`) for use outside the hot path.  Output layouts (x_dim = 1):
  SinusoidalEncoder        [x, sin(2^k x) for k, sin(2^k x + pi/2) for k]         (encoder.py:36-44)
  SinusoidalEncoderWithExp [x, (sin(2^k x), sin(2^k x + pi/2)) * exp(-k 2^k v) per k] (encoder.py:75-90)
"""
from __future__ import annotations

import math

import torch
from torch import nn


class _SinusoidalBase(nn.Module):
    def __init__(self, x_dim: int, min_deg: int, max_deg: int, use_identity: bool = True):
        super().__init__()
        self.x_dim, self.min_deg, self.max_deg, self.use_identity = x_dim, min_deg, max_deg, use_identity
        degs = list(range(min_deg, max_deg))
        self.register_buffer("scales", torch.tensor([2 ** d for d in degs]))

    @property
    def latent_dim(self) -> int:
        n_deg = self.max_deg - self.min_deg
        return (2 * n_deg + (1 if self.use_identity else 0)) * self.x_dim


class SinusoidalEncoder(_SinusoidalBase):
    """Frequency-major sin block followed by the phase-shifted block."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.max_deg == self.min_deg:
            return x
        scaled = [x * s for s in self.scales]                    # each [..., x_dim]
        phases = torch.cat(scaled, dim=-1)
        feats = torch.sin(torch.cat([phases, phases + 0.5 * math.pi], dim=-1))
        return torch.cat([x, feats], dim=-1) if self.use_identity else feats


class SinusoidalEncoderWithExp(_SinusoidalBase):
    """Per-frequency (sin, shifted-sin) pairs damped by exp(-k 2^k x_var)."""

    def __init__(self, x_dim: int, min_deg: int, max_deg: int, use_identity: bool = True):
        super().__init__(x_dim, min_deg, max_deg, use_identity)
        self.register_buffer("scales_move", torch.tensor([d * 2 ** d for d in range(min_deg, max_deg)]))

    def forward(self, x: torch.Tensor, x_var: torch.Tensor) -> torch.Tensor:
        if self.max_deg == self.min_deg:
            return x
        pieces = []
        for s, sm in zip(self.scales, self.scales_move):
            ph = x * s
            damp = torch.exp(-1 * (x_var * sm))
            pieces.append(torch.sin(torch.cat([ph, ph + 0.5 * math.pi], dim=-1)) * damp)
        feats = torch.cat(pieces, dim=-1)
        return torch.cat([x, feats], dim=-1) if self.use_identity else feats
