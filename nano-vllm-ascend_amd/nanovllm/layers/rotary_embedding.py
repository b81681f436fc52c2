"""NeoX rotary embedding (reference: nanovllm/layers/rotary_embedding.py) on mi_rope.

The fp32 cos/sin table is built with the same torch expressions as the reference
(:26-35) so its bits are identical; the rotation runs in the HIP kernel.
"""
from __future__ import annotations

from functools import lru_cache

import torch
from torch import nn

from nanovllm import ops


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float) -> None:
        super().__init__()
        self.head_size = head_size
        assert rotary_dim == head_size
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float, device="cpu") / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float, device="cpu")
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous()  # [max_pos, head_size] fp32
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """query [T, Hq, D], key [T, Hkv, D] (any token stride) -> rotated contiguous copies."""
        if self.cos_sin_cache.device != query.device:
            self.cos_sin_cache = self.cos_sin_cache.to(query.device)
        return ops.rope(positions, query, key, self.cos_sin_cache, query.shape[-2], key.shape[-2])


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)


def get_rope_llama(head_size: int, rotary_dim: int, max_position: int, base: float):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
