"""Token sampling (reference: nanovllm/layers/sampler.py:9-17) on mi_sample.

The reference draws torch.multinomial(softmax(logits.float() / T)).  The kernel draws
from the same distribution with the Gumbel-max identity and a counter-based generator
keyed by (seed, step, row, column); rows with temperature <= 0 take the argmax (the
deterministic path parity runs need).  Logits padded to the graph batch are sliced to
the real batch by the number of temperatures, as sampler.py:10-12.
"""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import ops


class Sampler(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        self.seed = seed
        self.step = 0

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor, out: torch.Tensor | None = None):
        self.step += 1
        return ops.sample(logits, temperatures, self.seed, self.step, out=out)
