"""Llama decoder (reference: nanovllm/models/llama.py) on the same HIP layers as Qwen3.

The reference's Llama differs from its Qwen3 in wiring only: no per-head q/k RMSNorm and no qkv bias
unless the config asks for one (llama.py:138-144: `attention_bias` / `bias` / `qkv_bias`; the o_proj bias
follows `attention_bias`, the MLP bias `mlp_bias`), rope_theta defaulting to 10000 (:135), the same NeoX
rotary table (get_rope_llama, rotary_embedding.py:63-69), the same residual / RMSNorm structure
(:172-185) and the same packed parameter names (:211-217).  So it is Qwen3DecoderLayer with those
switches - and with them every decode-path kernel (packed GEMMs, the fused attention launch with null norm
weights, split-K add+RMSNorm) is shared.  head_dim must be 128 (Llama-2/3 7B..70B; not the 64 of 1B/3B).
"""
from __future__ import annotations

from nanovllm.models.qwen3 import Qwen3ForCausalLM


def _rope_theta(config) -> float:
    theta = getattr(config, "rope_theta", None)
    if theta is None:
        theta = (getattr(config, "rope_parameters", None) or {}).get("rope_theta", 10000)
    return float(theta)


class LlamaForCausalLM(Qwen3ForCausalLM):
    def __init__(self, config, fused: bool = True) -> None:
        attention_bias = bool(getattr(config, "attention_bias", False) or getattr(config, "bias", False))
        qkv_bias = bool(config.qkv_bias) if hasattr(config, "qkv_bias") else attention_bias
        super().__init__(config, fused, qkv_bias=qkv_bias, qk_norm=False, o_bias=attention_bias,
                         mlp_bias=bool(getattr(config, "mlp_bias", False)), rope_theta=_rope_theta(config),
                         num_kv_heads=getattr(config, "num_key_value_heads", config.num_attention_heads))
