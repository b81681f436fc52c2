"""Per-step metadata side channel between the model runner and the layers
(reference: nanovllm/utils/context.py:5-37 — same field names, same three
functions, so layers written against the reference read the same attributes).

Added fields (None / 0 in code written against the reference):
  kv_lens     [n_seqs] int32 — tokens of each sequence present in the KV cache during
              prefill (== query lengths in the reference, which recomputes cached
              prefixes: model_runner.py:248-249).
  slot_is_2d  decode slot mapping is [B,2] = [block, offset] (model_runner.py:301,353),
              prefill's is flat (:263-270).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class Context:
    is_prefill: bool = False
    cu_seqlens_q: torch.Tensor | None = None
    cu_seqlens_k: torch.Tensor | None = None
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    slot_mapping: torch.Tensor | None = None
    context_lens: torch.Tensor | None = None
    block_tables: torch.Tensor | None = None
    is_enforce_eager: bool = True
    real_bs: int = -1
    block_size: int = 256
    kv_lens: torch.Tensor | None = None


_CONTEXT = Context()


def get_context() -> Context:
    return _CONTEXT


def set_context(is_prefill, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=0, max_seqlen_k=0,
                slot_mapping=None, context_lens=None, block_tables=None, is_enforce_eager=None, real_bs=None,
                block_size=None, kv_lens=None) -> None:
    global _CONTEXT
    _CONTEXT = Context(is_prefill, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, slot_mapping,
                       context_lens, block_tables, is_enforce_eager, real_bs, block_size, kv_lens)


def reset_context() -> None:
    global _CONTEXT
    _CONTEXT = Context()
