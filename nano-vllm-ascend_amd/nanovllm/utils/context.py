"""Per-step metadata side channel between the model runner and the layers.

The reference keeps one module-level record that `Attention.forward` and `ParallelLMHead.forward`
read (nanovllm/utils/context.py:5-37); layers written against it expect the attribute names below
and the three accessor functions, so those are the contract kept here.  Added fields:

  kv_lens  [n_seqs] int32 - tokens of each sequence present in the KV cache during prefill.  Equal to
           the query lengths in the reference (it recomputes cached prefixes, model_runner.py:248-249);
           larger when prefix-aware prefill skips cache-hit blocks.

  token_src / prev_tokens - decode steps queued behind a step whose tokens have not reached the host yet
           read their input ids on the device (engine lookahead; see ModelRunner.launch_decode).

The slot mapping is flat int32 [T] in prefill (model_runner.py:263-270) and [B, 2] = [block, offset]
in decode (:301,353); consumers tell them apart by `slot_mapping.dim()`.
"""
from __future__ import annotations

# attribute -> value when a step does not set it
_DEFAULTS = {
    "is_prefill": False,
    "cu_seqlens_q": None,      # int32 [n_seqs + 1]
    "cu_seqlens_k": None,
    "max_seqlen_q": 0,
    "max_seqlen_k": 0,
    "slot_mapping": None,
    "context_lens": None,      # int32 [B], decode
    "block_tables": None,      # int32 [rows, width], -1 padded
    "is_enforce_eager": True,
    "real_bs": -1,
    "block_size": 256,
    "kv_lens": None,
    "token_src": None,         # int32 [B]: row of prev_tokens a decode row takes its input id from, or -1
    "prev_tokens": None,       # int64 [>= B]: the previous step's sampled tokens, device-resident
}


class Context:
    __slots__ = tuple(_DEFAULTS)

    def __init__(self, **fields):
        unknown = set(fields) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"unknown context field(s): {sorted(unknown)}")
        for name, default in _DEFAULTS.items():
            setattr(self, name, fields.get(name, default))

    def __repr__(self):
        shown = ", ".join(f"{k}={getattr(self, k)!r}" for k in _DEFAULTS if getattr(self, k) is not _DEFAULTS[k])
        return f"Context({shown})"


_current = Context()


def get_context() -> Context:
    return _current


def set_context(is_prefill, **fields) -> None:
    """Install the metadata of the step about to run.  Fields passed as None keep their defaults."""
    global _current
    _current = Context(is_prefill=is_prefill, **{k: v for k, v in fields.items() if v is not None})


def reset_context() -> None:
    global _current
    _current = Context()
