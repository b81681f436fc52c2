"""Whole-model CPU oracle: Qwen3 decoder over a paged KV cache.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Wiring follows
nanovllm/models/qwen3.py:74-90 (attention), :118-122 (MLP), :148-161 (layer),
:175-185 (model), :207-218 (lm head) of the reference; the rounding points are
those of oracle/layers.py.  Weights use the reference's packed parameter names
(`qkv_proj`, `gate_up_proj`, qwen3.py:189-195).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from oracle import layers as L


@dataclass
class OracleConfig:
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    max_position_embeddings: int = 40960
    tie_word_embeddings: bool = True
    attention_bias: bool = False
    qk_norm: bool | None = None  # None: Qwen3 rule, norm iff no qkv bias (qwen3.py:70-72); Llama: False (llama.py:80-93)
    # mixture of experts (models/qwen3_moe.py); num_experts == 0: dense model
    num_experts: int = 0
    num_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    decoder_sparse_step: int = 1
    mlp_only_layers: tuple = ()

    def is_sparse_layer(self, layer_idx: int) -> bool:
        """qwen3_moe.py:208-212"""
        return (layer_idx not in self.mlp_only_layers and self.num_experts > 0
                and (layer_idx + 1) % self.decoder_sparse_step == 0)

    @classmethod
    def from_hf(cls, hf) -> "OracleConfig":
        rope = getattr(hf, "rope_theta", None)
        if rope is None:
            rp = getattr(hf, "rope_parameters", None) or {}
            rope = rp.get("rope_theta", 1000000.0)
        return cls(
            hidden_size=hf.hidden_size,
            num_hidden_layers=hf.num_hidden_layers,
            num_attention_heads=hf.num_attention_heads,
            num_key_value_heads=hf.num_key_value_heads,
            head_dim=getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads,
            intermediate_size=hf.intermediate_size,
            vocab_size=hf.vocab_size,
            rms_norm_eps=hf.rms_norm_eps,
            rope_theta=float(rope),
            max_position_embeddings=hf.max_position_embeddings,
            tie_word_embeddings=bool(getattr(hf, "tie_word_embeddings", False)),
            attention_bias=bool(getattr(hf, "attention_bias", getattr(hf, "model_type", "") != "llama")),
            qk_norm=False if getattr(hf, "model_type", "") == "llama"
            else (True if getattr(hf, "num_experts", 0) else None),  # the MoE attention always norms (qwen3_moe.py:76-77)
            num_experts=int(getattr(hf, "num_experts", 0) or 0),
            num_experts_per_tok=int(getattr(hf, "num_experts_per_tok", 0) or 0),
            moe_intermediate_size=int(getattr(hf, "moe_intermediate_size", 0) or 0),
            decoder_sparse_step=int(getattr(hf, "decoder_sparse_step", 1) or 1),
            mlp_only_layers=tuple(getattr(hf, "mlp_only_layers", None) or ()),
        )

    @property
    def has_qk_norm(self) -> bool:
        return (not self.attention_bias) if self.qk_norm is None else self.qk_norm


def random_weights(cfg: OracleConfig, seed: int = 0, std: float = 0.02, dtype=torch.bfloat16) -> dict:
    """Synthetic checkpoint: N(0, std^2) matrices, unit norm weights.

    One CPU generator, fixed draw order => identical on every host with this torch.
    """
    g = torch.Generator().manual_seed(seed)
    hq, hkv, d, h, i = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size,
                        cfg.intermediate_size)

    def mat(*shape):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * std).to(dtype)

    w = {"model.embed_tokens.weight": mat(cfg.vocab_size, h)}
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}."
        w[p + "self_attn.qkv_proj.weight"] = mat((hq + 2 * hkv) * d, h)
        if cfg.attention_bias:
            w[p + "self_attn.qkv_proj.bias"] = mat((hq + 2 * hkv) * d)
        if cfg.has_qk_norm:
            w[p + "self_attn.q_norm.weight"] = torch.ones(d, dtype=dtype)
            w[p + "self_attn.k_norm.weight"] = torch.ones(d, dtype=dtype)
        w[p + "self_attn.o_proj.weight"] = mat(h, hq * d)
        if cfg.is_sparse_layer(li):  # router + stacked expert weights [E, ...]
            mi = cfg.moe_intermediate_size
            w[p + "mlp.gate.weight"] = mat(cfg.num_experts, h)
            w[p + "mlp.experts.gate_up_proj.weight"] = mat(cfg.num_experts, 2 * mi, h)
            w[p + "mlp.experts.down_proj.weight"] = mat(cfg.num_experts, h, mi)
        else:
            w[p + "mlp.gate_up_proj.weight"] = mat(2 * i, h)
            w[p + "mlp.down_proj.weight"] = mat(h, i)
        w[p + "input_layernorm.weight"] = torch.ones(h, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = torch.ones(h, dtype=dtype)
    w["model.norm.weight"] = torch.ones(h, dtype=dtype)
    if not cfg.tie_word_embeddings:
        w["lm_head.weight"] = mat(cfg.vocab_size, h)
    return w


class OracleQwen3:
    def __init__(self, cfg: OracleConfig, weights: dict, num_blocks: int, block_size: int):
        self.cfg, self.w, self.block_size = cfg, weights, block_size
        dt = weights["model.embed_tokens.weight"].dtype
        shape = (cfg.num_hidden_layers, num_blocks, block_size, cfg.num_key_value_heads, cfg.head_dim)
        self.k_cache = torch.zeros(shape, dtype=dt)
        self.v_cache = torch.zeros(shape, dtype=dt)
        self.cos_sin = L.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        self.scale = 1.0 / math.sqrt(cfg.head_dim)

    # -- one decoder stack pass over T tokens -------------------------------------------------
    def _forward(self, input_ids, positions, slot_flat, attend):
        """self.trace (a list, optional): gets one dict per layer with every intermediate tensor of the layer
        (x, qkv, o, o_proj, x2, residual, act, mlp_out) and finally the normalised hidden states, for
        teacher-forced per-stage and per-layer comparisons."""
        c, w = self.cfg, self.w
        trace = getattr(self, "trace", None)
        hq, hkv, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        h = L.embedding(input_ids, w["model.embed_tokens.weight"])
        residual = None
        for li in range(c.num_hidden_layers):
            p = f"model.layers.{li}."
            if residual is None:
                residual, x = h, L.rms_norm(h, w[p + "input_layernorm.weight"], c.rms_norm_eps)
            else:
                x, residual = L.add_rms_norm(h, residual, w[p + "input_layernorm.weight"], c.rms_norm_eps)
            qkv = L.linear(x, w[p + "self_attn.qkv_proj.weight"], w.get(p + "self_attn.qkv_proj.bias"))
            q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
            q, k, v = q.reshape(-1, hq, d), k.reshape(-1, hkv, d), v.reshape(-1, hkv, d)
            if c.has_qk_norm:
                q = L.rms_norm(q, w[p + "self_attn.q_norm.weight"], c.rms_norm_eps)
                k = L.rms_norm(k, w[p + "self_attn.k_norm.weight"], c.rms_norm_eps)
            q = L.apply_rope(positions, q, self.cos_sin)
            k = L.apply_rope(positions, k, self.cos_sin)
            L.kv_scatter(k, v, self.k_cache[li], self.v_cache[li], slot_flat)
            o = attend(li, q)
            stage = {"x": x, "qkv": qkv, "o": o.reshape(o.shape[0], -1)} if trace is not None else None
            h = L.linear(o, w[p + "self_attn.o_proj.weight"])
            x, residual = L.add_rms_norm(h, residual, w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            if c.is_sparse_layer(li):
                act = x  # no single activation tensor: the block's input stands in for the trace
                if trace is not None:
                    stage.update(o_proj=h, x2=x, residual=residual, act=act)
                h = L.moe_block(x, w[p + "mlp.gate.weight"], w[p + "mlp.experts.gate_up_proj.weight"],
                                w[p + "mlp.experts.down_proj.weight"], c.num_experts_per_tok)
            else:
                gu = L.linear(x, w[p + "mlp.gate_up_proj.weight"])
                act = L.silu_and_mul(gu)
                if trace is not None:
                    stage.update(o_proj=h, x2=x, residual=residual, act=act)
                h = L.linear(act, w[p + "mlp.down_proj.weight"])
            if trace is not None:
                stage["mlp_out"] = h
                trace.append({k: v.clone() for k, v in stage.items()})
        x, _ = L.add_rms_norm(h, residual, w["model.norm.weight"], c.rms_norm_eps)
        if trace is not None:
            trace.append(x.clone())
        return x

    def _logits(self, hidden, fp32: bool):
        head = self.w.get("lm_head.weight", self.w["model.embed_tokens.weight"])
        return L.linear(hidden, head, keep_fp32=fp32)

    def prefill(self, input_ids, positions, cu_seqlens_q, slot_flat, block_tables, kv_lens=None, fp32_logits=False):
        """Mirrors ModelRunner.prepare_prefill + run_model (model_runner.py:238-290, 376-396)."""
        if kv_lens is None:
            kv_lens = cu_seqlens_q[1:] - cu_seqlens_q[:-1]

        def attend(li, q):
            return L.paged_attention_prefill(q, self.k_cache[li], self.v_cache[li], block_tables, cu_seqlens_q,
                                             kv_lens, self.scale)

        hidden = self._forward(input_ids, positions, slot_flat, attend)
        last = (cu_seqlens_q[1:] - 1).long()  # embed_head.py:58-60
        return self._logits(hidden[last], fp32_logits)

    def decode(self, input_ids, positions, slot_2d, context_lens, block_tables, fp32_logits=False):
        """Mirrors prepare_decode[_padding] + run_model (model_runner.py:292-366, 376-396)."""
        slot_flat = torch.where((slot_2d[:, 0] >= 0), slot_2d[:, 0] * self.block_size + slot_2d[:, 1],
                                torch.full_like(slot_2d[:, 0], -1))

        def attend(li, q):
            return L.paged_attention_decode(q, self.k_cache[li], self.v_cache[li], block_tables, context_lens,
                                            self.scale)

        hidden = self._forward(input_ids, positions, slot_flat, attend)
        return self._logits(hidden, fp32_logits)
