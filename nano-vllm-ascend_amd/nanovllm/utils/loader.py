"""Weight loading (reference: nanovllm/utils/loader.py:12-59): safetensors files ->
parameters, routing packed projections (q/k/v -> qkv_proj, gate/up -> gate_up_proj)
through each parameter's `weight_loader` so TP shards land in place.

`init_synthetic_weights` is the documented no-checkpoint mode: every
rank draws the FULL tensors from one seeded CPU generator and passes them through the
same weight_loader path, so TP ranks hold consistent shards of one virtual checkpoint.
"""
from __future__ import annotations

import os
from glob import glob

import torch
from torch import nn


def default_weight_loader(param: nn.Parameter, loaded_weight: torch.Tensor):
    param.data.copy_(loaded_weight)


def has_checkpoint(path: str) -> bool:
    return bool(glob(os.path.join(path, "*.safetensors")))


def _route(model: nn.Module, name: str, tensor: torch.Tensor) -> None:
    mapping = getattr(model, "packed_modules_mapping", {})
    for src, (dst, shard_id) in mapping.items():
        if src in name:
            param = model.get_parameter(name.replace(src, dst))
            if tensor.dtype != param.dtype:
                tensor = tensor.to(param.dtype)
            param.weight_loader(param, tensor, shard_id)
            return
    try:
        param = model.get_parameter(name)
    except AttributeError as e:
        raise AttributeError(f"checkpoint tensor '{name}' has no matching parameter") from e
    if tensor.dtype != param.dtype:
        tensor = tensor.to(param.dtype)
    getattr(param, "weight_loader", default_weight_loader)(param, tensor)


def pack_model_weights(model: nn.Module) -> None:
    """(Re)build the fragment-native weight copies of every linear layer that lives on a GPU."""
    for m in model.modules():
        if hasattr(m, "pack") and m.weight.is_cuda:
            m.pack()


def load_model(model: nn.Module, path: str, name_mapping=None) -> None:
    from safetensors import safe_open

    tied = getattr(getattr(model, "lm_head", None), "weight", None)
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                target = name if name_mapping is None else name_mapping(name)
                if target is None:
                    continue
                if target == "lm_head.weight" and tied is not None and \
                        tied.data_ptr() == model.model.embed_tokens.weight.data_ptr():
                    continue  # tied head: the embedding row is the weight
                _route(model, target, f.get_tensor(name))
    pack_model_weights(model)


def synthetic_state(hf_config, seed: int = 0, std: float = 0.02, dtype=torch.bfloat16):
    """HF-named full tensors of a random Qwen3-shaped checkpoint, generated lazily in a
    fixed order from one CPU generator (identical on every rank and host)."""
    g = torch.Generator().manual_seed(seed)
    c = hf_config
    d = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    hq, hkv, h, inter = c.num_attention_heads, c.num_key_value_heads, c.hidden_size, c.intermediate_size
    llama = getattr(c, "model_type", "") == "llama"
    # Qwen3DecoderLayer's default (and the reference's, qwen3.py:126) is True; Llama's is False (llama.py:138)
    bias = bool(getattr(c, "attention_bias", not llama))
    qk_norm = (not bias and not llama) or bool(getattr(c, "num_experts", 0))  # the MoE attention always norms

    def mat(*shape):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * std).to(dtype)

    yield "model.embed_tokens.weight", mat(c.vocab_size, h)
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        # same draw order as oracle.model.random_weights (packed qkv, then the rest)
        qkv = mat((hq + 2 * hkv) * d, h)
        yield p + "self_attn.q_proj.weight", qkv[: hq * d]
        yield p + "self_attn.k_proj.weight", qkv[hq * d: (hq + hkv) * d]
        yield p + "self_attn.v_proj.weight", qkv[(hq + hkv) * d:]
        if bias:
            b = mat((hq + 2 * hkv) * d)
            yield p + "self_attn.q_proj.bias", b[: hq * d]
            yield p + "self_attn.k_proj.bias", b[hq * d: (hq + hkv) * d]
            yield p + "self_attn.v_proj.bias", b[(hq + hkv) * d:]
        if qk_norm:
            yield p + "self_attn.q_norm.weight", torch.ones(d, dtype=dtype)
            yield p + "self_attn.k_norm.weight", torch.ones(d, dtype=dtype)
        yield p + "self_attn.o_proj.weight", mat(h, hq * d)
        n_exp = int(getattr(c, "num_experts", 0) or 0)
        sparse = (n_exp > 0 and i not in (getattr(c, "mlp_only_layers", None) or [])
                  and (i + 1) % int(getattr(c, "decoder_sparse_step", 1) or 1) == 0)
        if sparse:  # router, then the stacked expert weights (oracle.model.random_weights draws them the same way)
            mi = c.moe_intermediate_size
            yield p + "mlp.gate.weight", mat(n_exp, h)
            gu, dn = mat(n_exp, 2 * mi, h), mat(n_exp, h, mi)
            for e in range(n_exp):
                yield p + f"mlp.experts.{e}.gate_proj.weight", gu[e, :mi]
                yield p + f"mlp.experts.{e}.up_proj.weight", gu[e, mi:]
                yield p + f"mlp.experts.{e}.down_proj.weight", dn[e]
        else:
            gu = mat(2 * inter, h)
            yield p + "mlp.gate_proj.weight", gu[:inter]
            yield p + "mlp.up_proj.weight", gu[inter:]
            yield p + "mlp.down_proj.weight", mat(h, inter)
        yield p + "input_layernorm.weight", torch.ones(h, dtype=dtype)
        yield p + "post_attention_layernorm.weight", torch.ones(h, dtype=dtype)
    yield "model.norm.weight", torch.ones(h, dtype=dtype)
    if not getattr(c, "tie_word_embeddings", False):
        yield "lm_head.weight", mat(c.vocab_size, h)


def init_synthetic_weights(model: nn.Module, hf_config, seed: int = 0, std: float = 0.02) -> None:
    for name, tensor in synthetic_state(hf_config, seed, std):
        _route(model, name, tensor)
    pack_model_weights(model)


def load_state_dict_packed(model: nn.Module, weights: dict) -> None:
    """Load a dict that already uses the packed parameter names (oracle.model.random_weights);
    full (unsharded) tensors are split per rank through the HF-style q/k/v and gate/up routes."""
    cfg_attn = model.model.layers[0].self_attn
    hq, hkv, d = cfg_attn.total_num_heads, cfg_attn.total_num_kv_heads, cfg_attn.head_dim
    for name, t in weights.items():
        if ".mlp.experts." in name and t.dim() == 3:  # stacked [E, ...] expert weights (oracle naming)
            for e in range(t.shape[0]):
                pe = name.replace(".experts.", f".experts.{e}.")
                if "gate_up_proj" in pe:
                    gate, up = t[e].chunk(2, dim=0)
                    _route(model, pe.replace("gate_up_proj", "gate_proj"), gate)
                    _route(model, pe.replace("gate_up_proj", "up_proj"), up)
                else:
                    _route(model, pe, t[e])
        elif "qkv_proj" in name:
            parts = t.split([hq * d, hkv * d, hkv * d], dim=0)
            for tag, part in zip(("q_proj", "k_proj", "v_proj"), parts):
                _route(model, name.replace("qkv_proj", tag), part)
        elif "gate_up_proj" in name:
            gate, up = t.chunk(2, dim=0)
            _route(model, name.replace("gate_up_proj", "gate_proj"), gate)
            _route(model, name.replace("gate_up_proj", "up_proj"), up)
        else:
            _route(model, name, t)
    pack_model_weights(model)
