"""Paged GQA attention (reference: nanovllm/layers/attention.py) on the HIP kernels
mi_reshape_and_cache / mi_scatter_update_kv / mi_paged_attn_prefill / mi_paged_attn_decode.

Same constructor and forward signature as the reference class; the runner injects
`k_cache` / `v_cache` into every module that has both attributes
(model_runner.py:222-229) and the per-step metadata arrives through get_context().
The caches use the fragment-native layout of include/mi355_nanovllm.h (head_dim 128 or 64; 1, 2, 4, 7, 8 or 16 query
heads per kv head) - or, for the head geometries those kernels are not built for (`self.plain`), the plain
[blocks, kv heads, block, head_dim] layout and the mi_*_plain kernels (csrc/attn_plain.hip).
"""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import ops
from nanovllm.utils.context import get_context


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scaling: float | None, num_kv_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.block_size = 0
        self.scale = scaling if scaling is not None else 1.0 / (head_dim ** 0.5)
        self.k_cache = torch.tensor([])
        self.v_cache = torch.tensor([])
        self.plain = ops.attention_is_plain(num_heads, num_kv_heads, head_dim)
        # the fused launches (models/qwen3.py) are written for 128-wide heads and power-of-two groups; head_dim 64 and
        # groups of 7 take the module-by-module sequence RoPE -> store -> attention on the same fragment-native kernels
        self.fusable = ops.attention_is_fusable(num_heads, num_kv_heads, head_dim)

    def _store_kvcache(self, k: torch.Tensor, v: torch.Tensor, context) -> None:
        """attention.py:22-35: flat slots in prefill, [block, offset] pairs in decode."""
        if self.plain:
            ops.kv_store_plain(k, v, self.k_cache, self.v_cache, context.slot_mapping, self.num_kv_heads, self.block_size)
        elif context.slot_mapping.dim() == 2:
            ops.scatter_update_kv(k, v, self.k_cache, self.v_cache, context.slot_mapping, self.num_kv_heads,
                                  self.block_size)
        else:
            ops.reshape_and_cache(k, v, self.k_cache, self.v_cache, context.slot_mapping, self.num_kv_heads,
                                  self.block_size)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        context = get_context()
        self.block_size = context.block_size
        if k.dim() == 2:
            k = k.view(-1, self.num_kv_heads, self.head_dim)
            v = v.view(-1, self.num_kv_heads, self.head_dim)
            q = q.view(-1, self.num_heads, self.head_dim)
        self._store_kvcache(k, v, context)
        if context.is_prefill:
            kv_lens = context.kv_lens
            if kv_lens is None:
                kv_lens = (context.cu_seqlens_k[1:] - context.cu_seqlens_k[:-1]).contiguous()
            if self.plain:
                return ops.paged_attn_prefill_plain(q, self.k_cache, self.v_cache, context.block_tables,
                                                    context.cu_seqlens_q, kv_lens, context.max_seqlen_q, self.num_heads,
                                                    self.num_kv_heads, self.block_size, self.scale)
            return ops.paged_attn_prefill(q, self.k_cache, self.v_cache, context.block_tables,
                                          context.cu_seqlens_q, kv_lens, context.max_seqlen_q, self.num_heads,
                                          self.num_kv_heads, self.block_size, self.scale)
        if self.plain:
            return ops.paged_attn_decode_plain(q, self.k_cache, self.v_cache, context.block_tables, context.context_lens,
                                               self.num_heads, self.num_kv_heads, self.block_size, self.scale)
        return ops.paged_attn_decode(q, self.k_cache, self.v_cache, context.block_tables, context.context_lens,
                                     self.num_heads, self.num_kv_heads, self.block_size, self.scale)
