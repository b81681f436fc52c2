"""RMSNorm (reference: nanovllm/layers/layernorm.py) on mi_rmsnorm / mi_add_rmsnorm."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import ops


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def rms_forward(self, x: torch.Tensor) -> torch.Tensor:
        """layernorm.py:16-25; also the per-head q/k norm on [T, H, D] views (qwen3.py:83-85)."""
        return ops.rmsnorm(x, self.weight, self.eps)

    def add_rms_forward(self, x: torch.Tensor, residual: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """layernorm.py:27-38: returns (normalised, new residual)."""
        return ops.add_rmsnorm(x, residual, self.weight, self.eps)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is None:
            return self.rms_forward(x)
        return self.add_rms_forward(x, residual)
