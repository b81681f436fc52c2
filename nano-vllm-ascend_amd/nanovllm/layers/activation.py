"""SwiGLU gate (reference: nanovllm/layers/activation.py:10-12) on mi_silu_mul."""
import torch
from torch import nn

from nanovllm import ops


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_mul(x)
