"""Qwen3-MoE decoder (reference: nanovllm/models/qwen3_moe.py) on the HIP layers.

Attention, norms, embedding and head are the Qwen3 ones (the MoE attention always applies the per-head q/k
RMSNorm, qwen3_moe.py:76-77).  The sparse block (qwen3_moe.py:125-185) keeps the reference's parameter names
- `mlp.gate.weight`, `mlp.experts.<e>.{gate_up_proj,down_proj}.weight`, HF checkpoints reach them through
`packed_modules_mapping` - but stores the experts STACKED ([E, 2I/tp, H] and [E, H, I/tp]; each expert's
parameter is a view), because the device side is not a Python loop over experts with three library GEMMs each
(:171-184) but five launches over expert-sorted (token, expert) pairs (csrc/moe.hip, ops.moe_forward).
Tensor parallelism shards every expert along its intermediate dimension exactly as the reference's
MergedColumnParallelLinear / RowParallelLinear experts do (:104-115); the router is replicated (:138).
"""
from __future__ import annotations

import warnings

import torch
from torch import nn

from nanovllm import ops
from nanovllm.layers.linear import linear_forward
from nanovllm.layers.parallel import all_reduce_sum, collectives_on, divide, tp_rank, tp_size
from nanovllm.models.qwen3 import Qwen3DecoderLayer, Qwen3ForCausalLM


class _ExpertProj(nn.Module):
    """One expert's view into a stacked weight, carrying the weight_loader of the linear class it stands for."""

    def __init__(self, view: torch.Tensor, loader):
        super().__init__()
        self.weight = nn.Parameter(view, requires_grad=False)
        self.weight.weight_loader = loader


class _ExpertMLP(nn.Module):
    def __init__(self, gate_up_view, down_view, inter_local: int):
        super().__init__()
        rank, world = tp_rank(), tp_size()

        def load_gate_up(param, loaded, shard_id):  # MergedColumnParallelLinear.weight_loader, linear.py:87-93
            param.data.narrow(0, shard_id * inter_local, inter_local).copy_(loaded.chunk(world, 0)[rank])

        def load_down(param, loaded):  # RowParallelLinear.weight_loader, linear.py:142-147
            param.data.copy_(loaded.narrow(1, rank * inter_local, inter_local))

        self.gate_up_proj = _ExpertProj(gate_up_view, load_gate_up)
        self.down_proj = _ExpertProj(down_view, load_down)


class Qwen3MoeSparseMoeBlock(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.num_experts, self.top_k = config.num_experts, config.num_experts_per_tok
        hidden, inter = config.hidden_size, divide(config.moe_intermediate_size, tp_size())
        self.gate = nn.Linear(hidden, self.num_experts, bias=False)  # replicated router (:138)
        self.register_buffer("gate_up_stacked", torch.empty(self.num_experts, 2 * inter, hidden), persistent=False)
        self.register_buffer("down_stacked", torch.empty(self.num_experts, hidden, inter), persistent=False)
        self.experts = nn.ModuleList([_ExpertMLP(self.gate_up_stacked[e], self.down_stacked[e], inter)
                                      for e in range(self.num_experts)])
        self.gate_up_packed: torch.Tensor | None = None
        self.down_packed: torch.Tensor | None = None
        self.gate_packed: torch.Tensor | None = None

    @property
    def weight(self) -> torch.Tensor:  # what utils.loader.pack_model_weights looks at
        return self.gate_up_stacked

    def pack(self) -> None:
        # nn.Module.to() may have re-created the buffers: the experts' parameters must alias them again
        for e, ex in enumerate(self.experts):
            if ex.gate_up_proj.weight.data_ptr() != self.gate_up_stacked[e].data_ptr():
                self.gate_up_stacked[e].copy_(ex.gate_up_proj.weight.data)
                self.down_stacked[e].copy_(ex.down_proj.weight.data)
        self.gate_up_packed = ops.pack_expert_weights(self.gate_up_stacked.contiguous(), self.gate_up_packed)
        self.down_packed = ops.pack_expert_weights(self.down_stacked.contiguous(), self.down_packed)
        w = self.gate.weight.data
        self.gate_packed = ops.pack_weight(w, self.gate_packed) if w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0 else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        x = x.reshape(-1, shape[-1])
        logits = linear_forward(x, self.gate.weight, None, self.gate_packed)  # bf16 router logits (:151)
        out, _, _ = ops.moe_forward(x.contiguous(), logits, self.gate_up_packed, self.down_packed, self.top_k,
                                    all_reduce=all_reduce_sum if collectives_on() else None)
        return out.view(shape)


class Qwen3MoeDecoderLayer(Qwen3DecoderLayer):
    def __init__(self, config, layer_idx: int, fused: bool = True) -> None:
        sparse = (layer_idx not in (getattr(config, "mlp_only_layers", None) or [])
                  and config.num_experts > 0 and (layer_idx + 1) % config.decoder_sparse_step == 0)  # :208-212
        # a sparse layer never allocates the dense MLP of config.intermediate_size
        super().__init__(config, fused, qk_norm=True, qkv_bias=getattr(config, "attention_bias", False),
                         build_mlp=not sparse)
        if sparse:
            self.mlp = Qwen3MoeSparseMoeBlock(config)


class Qwen3MoeForCausalLM(Qwen3ForCausalLM):
    def __init__(self, config, fused: bool = True) -> None:
        if getattr(config, "norm_topk_prob", True) is False:
            # the reference's block renormalises the top-k weights unconditionally (qwen3_moe.py:156-158) and so does
            # mi_moe_route: a checkpoint trained with norm_topk_prob = false gets logits that differ from HF's
            warnings.warn("norm_topk_prob=False is ignored (as in the reference): the top-k router weights are "
                          "renormalised to sum to one", stacklevel=2)
        # only the MoE layers are built (no dense Qwen3 stack first)
        super().__init__(config, fused, layer_factory=lambda i: Qwen3MoeDecoderLayer(config, i, fused))
