"""Qwen3 / Qwen2 decoder (reference: nanovllm/models/qwen3.py) wired onto the HIP layers.

Module and parameter names equal the reference's (and therefore HF checkpoints through
`packed_modules_mapping`), so the same loader and the same call sites work.

`fused=True` (default) replaces the four-op sequence q_norm -> k_norm -> rotary ->
KV scatter (qwen3.py:83-88, attention.py:22-35) by the single kernel
mi_qknorm_rope_store, whose rounding points are identical (tests require equal bits).
"""
from __future__ import annotations

import os
import warnings

import torch
from torch import nn

from nanovllm import ops
from nanovllm.layers.activation import SiluAndMul
from nanovllm.layers.attention import Attention
from nanovllm.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
from nanovllm.layers.layernorm import RMSNorm
from nanovllm.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from nanovllm.layers.parallel import all_reduce_sum, collectives_on, get_xgmi_comm, tp_size
from nanovllm.layers.rotary_embedding import get_rope
from nanovllm.utils.context import get_context


def rope_theta_of(config) -> float:
    """transformers >= 5 moved rope_theta into config.rope_parameters; the reference's
    getattr(config, "rope_theta", 1000000) (qwen3.py:137) then silently takes the 1e6
    fallback.  Read the real value, fall back identically."""
    theta = getattr(config, "rope_theta", None)
    if theta is None:
        params = getattr(config, "rope_parameters", None) or {}
        theta = params.get("rope_theta", 1000000)
    return float(theta)


class Qwen3Attention(nn.Module):
    _warned_rope_scaling = False

    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, max_position: int = 4096 * 32,
                 head_dim: int | None = None, rms_norm_eps: float = 1e-06, qkv_bias: bool = False,
                 rope_theta: float = 10000, rope_scaling: tuple | None = None, fused: bool = True,
                 qk_norm: bool | None = None, o_bias: bool = False) -> None:
        super().__init__()
        world = tp_size()
        self.total_num_heads = num_heads
        assert num_heads % world == 0 and num_kv_heads % world == 0
        self.num_heads = num_heads // world
        self.total_num_kv_heads = num_kv_heads
        self.num_kv_heads = num_kv_heads // world
        self.head_dim = head_dim or hidden_size // num_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_bias = qkv_bias
        # Qwen3: per-head q/k RMSNorm instead of qkv biases (qwen3.py:70-72); Llama: neither (llama.py:80-93)
        self.qk_norm = (not qkv_bias) if qk_norm is None else qk_norm
        self.fused = fused
        self.rms_norm_eps = rms_norm_eps

        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, num_heads, num_kv_heads, bias=qkv_bias)
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=o_bias)
        rope_type = (rope_scaling.get("rope_type") or rope_scaling.get("type")) if isinstance(rope_scaling, dict) \
            else rope_scaling
        if rope_type not in (None, "default") and not Qwen3Attention._warned_rope_scaling:
            # the reference hands rope_scaling to get_rope, which asserts it away or ignores it
            # (rotary_embedding.py:52-69); a checkpoint with scaled RoPE (Llama-3.1 "llama3" type) runs with the
            # UNscaled table there and here - logits differ from HF's beyond the original context length
            Qwen3Attention._warned_rope_scaling = True
            warnings.warn(f"rope_scaling={rope_scaling!r} is ignored (as in the reference): positions use the plain "
                          f"RoPE table with base {rope_theta}", stacklevel=2)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=max_position,
                                   base=rope_theta)
        self.attn = Attention(self.num_heads, self.head_dim, None, self.num_kv_heads)
        if self.qk_norm:
            self.q_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)
            self.k_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)

    def _kv_store_in_projection(self, x: torch.Tensor) -> bool:
        """A prefill step whose qkv projection can carry k-norm + RoPE + the K / V cache store in its epilogue
        (mi_gemm_bf16_qkv_store): the large-M tile kernel's shapes, head_dim 128, plain bf16 weights.  OPT-IN
        (MI355_QKV_STORE=1): bit-identical caches, but measured slower than the two launches it replaces (DESIGN.md
        section 4.4: 162 vs 147 us per layer at 16 x 1024 tokens) - with one wave per SIMD the epilogue's dependent
        table loads have nothing to hide behind."""
        ctx, proj = get_context(), self.qkv_proj
        return (self.fused and ctx.is_prefill and self.attn.k_cache.numel() > 0 and self.attn.fusable and x.is_cuda
                and x.dim() == 2 and ctx.slot_mapping is not None and ctx.slot_mapping.dim() == 1
                and not isinstance(getattr(proj, "weight_packed", None), ops.Fp8Weight)
                and proj.weight.dtype == torch.bfloat16 and proj.weight.shape[1] % 64 == 0
                and ops.qkv_store_takes(x.shape[0], proj.weight.shape[0], self.head_dim, ctx.block_size)
                and os.environ.get("MI355_PREFILL_FUSED_Q", "1") != "0" and os.environ.get("MI355_QKV_STORE", "0") == "1")

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        if self._kv_store_in_projection(hidden_states):
            return self.o_proj(self._attend_fused(positions, None, hidden_states).flatten(1, -1))
        qkv = self.qkv_proj(hidden_states)
        if self.fused and self.attn.k_cache.numel() > 0 and self.attn.fusable:
            o = self._attend_fused(positions, qkv)
        else:
            o = self._attend_unfused(positions, qkv)
        return self.o_proj(o.flatten(1, -1))

    def _attend_unfused(self, positions: torch.Tensor, qkv: torch.Tensor) -> torch.Tensor:
        """The reference's operator sequence (qwen3.py:79-90): split, per-head q/k norm, RoPE, store + attention - one
        launch each.  The path of the plain-layout head geometries (head_dim 64, GQA 7:1)."""
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q = q.view(-1, self.num_heads, self.head_dim)
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        if self.qk_norm:
            q = self.q_norm(q)
            k = self.k_norm(k)
        q, k = self.rotary_emb(positions, q, k)
        return self.attn(q, k, v)

    def _attend_fused(self, positions: torch.Tensor, qkv: torch.Tensor | None, x: torch.Tensor | None = None) -> torch.Tensor:
        """qkv None (with the layer's input x, see _kv_store_in_projection): the projection itself stores K / V."""
        ctx, attn = get_context(), self.attn
        attn.block_size = ctx.block_size
        rope = self.rotary_emb
        dev = x.device if qkv is None else qkv.device
        if rope.cos_sin_cache.device != dev:
            rope.cos_sin_cache = rope.cos_sin_cache.to(dev)
        qw = self.q_norm.weight if self.qk_norm else None
        kw = self.k_norm.weight if self.qk_norm else None
        if qkv is None or (ctx.is_prefill and qkv.shape[0] >= 64 and os.environ.get("MI355_PREFILL_FUSED_Q", "1") != "0"):
            # prefill-sized: K / V go to the cache (whole tiles), the queries are normed and rotated inside the
            # attention kernel's Q-operand load - q is never written to and read back from HBM
            if qkv is None:  # ... from the projection's epilogue: the k / v columns of the qkv rows are never written
                qkv = ops.gemm_qkv_store(x, self.qkv_proj.weight, self.qkv_proj.bias, kw, self.rms_norm_eps, positions,
                                         rope.cos_sin_cache, attn.k_cache, attn.v_cache, ctx.slot_mapping,
                                         self.num_heads, self.num_kv_heads, ctx.block_size)
            else:
                ops.qknorm_rope_store(qkv, qw, kw, self.rms_norm_eps, positions, rope.cos_sin_cache, attn.k_cache,
                                      attn.v_cache, ctx.slot_mapping, self.num_heads, self.num_kv_heads,
                                      ctx.block_size, store_q=False)
            kv_lens = ctx.kv_lens
            if kv_lens is None:
                kv_lens = (ctx.cu_seqlens_k[1:] - ctx.cu_seqlens_k[:-1]).contiguous()
            return ops.paged_attn_prefill_fused(qkv, qw, self.rms_norm_eps, positions, rope.cos_sin_cache,
                                                attn.k_cache, attn.v_cache, ctx.block_tables, ctx.cu_seqlens_q,
                                                kv_lens, ctx.max_seqlen_q, self.num_heads, self.num_kv_heads,
                                                ctx.block_size, attn.scale)
        q = ops.qknorm_rope_store(qkv, qw, kw, self.rms_norm_eps, positions, rope.cos_sin_cache, attn.k_cache,
                                  attn.v_cache, ctx.slot_mapping, self.num_heads, self.num_kv_heads,
                                  ctx.block_size)
        if ctx.is_prefill:
            kv_lens = ctx.kv_lens
            if kv_lens is None:
                kv_lens = (ctx.cu_seqlens_k[1:] - ctx.cu_seqlens_k[:-1]).contiguous()
            return ops.paged_attn_prefill(q, attn.k_cache, attn.v_cache, ctx.block_tables, ctx.cu_seqlens_q,
                                          kv_lens, ctx.max_seqlen_q, self.num_heads, self.num_kv_heads,
                                          ctx.block_size, attn.scale)
        return ops.paged_attn_decode(q, attn.k_cache, attn.v_cache, ctx.block_tables, ctx.context_lens,
                                     self.num_heads, self.num_kv_heads, ctx.block_size, attn.scale)


class Qwen3MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str, bias: bool = False) -> None:
        super().__init__()
        assert hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=bias)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=bias)
        self.act_fn = SiluAndMul()

    def forward(self, x):
        gu = self.gate_up_proj
        rows = x.numel() // x.shape[-1]
        if (ops.prefers_tile(rows, gu.weight.shape[0]) and gu.bias is None and x.is_cuda and gu.weight.shape[1] % 64 == 0
                and (gu.weight.shape[0] // 2) % 128 == 0 and not isinstance(gu.weight_packed, ops.Fp8Weight)):
            # prefill-sized: SiluAndMul is the tile GEMM's epilogue (the reference's three roundings; the
            # 2 x intermediate wide gate_up output never exists in memory)
            shape = x.shape
            act = ops.gemm_tile(x.reshape(-1, shape[-1]), gu.weight, silu_mul=True)
            return self.down_proj(act.view(*shape[:-1], -1))
        return self.down_proj(self.act_fn(gu(x)))


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, config, fused: bool = True, **attn_overrides) -> None:
        """attn_overrides / mlp_bias: the Llama wiring (models/llama.py) of the same layer."""
        super().__init__()
        mlp_bias = attn_overrides.pop("mlp_bias", False)
        build_mlp = attn_overrides.pop("build_mlp", True)  # False: the subclass installs its own block (sparse MoE)
        attn_kw = dict(
            hidden_size=config.hidden_size,
            num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads,
            max_position=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps,
            qkv_bias=getattr(config, "attention_bias", True),
            head_dim=getattr(config, "head_dim", None),
            rope_theta=rope_theta_of(config),
            rope_scaling=getattr(config, "rope_scaling", None),
            fused=fused,
        )
        attn_kw.update(attn_overrides)
        self.self_attn = Qwen3Attention(**attn_kw)
        if build_mlp:
            self.mlp = Qwen3MLP(config.hidden_size, config.intermediate_size, config.hidden_act, bias=mlp_bias)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            hidden_states, residual = self.input_layernorm(hidden_states), hidden_states
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class Qwen3Model(nn.Module):
    def __init__(self, config, fused: bool = True, layer_factory=None, **layer_overrides) -> None:
        """layer_factory(layer_idx) -> decoder layer: models whose layers differ (Qwen3-MoE) build ONLY their own
        layers - no dense stack is allocated first and thrown away."""
        super().__init__()
        self.fused = fused
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        make = layer_factory or (lambda i: Qwen3DecoderLayer(config, fused, **layer_overrides))
        self.layers = nn.ModuleList([make(i) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    # prefill steps up to this many tokens take the seven-launch streaming path of the decode step, longer ones the
    # module-by-module path on the tile GEMMs.  Measured on the captured prefill steps of a Qwen3-0.6B-shaped engine
    # (profiles/r05_prefill_bucket_times.txt; ms per step with the switch at 512 / 128 tokens): 192 tokens 2.11 / 1.98,
    # 256: 2.32 / 2.05, 384: 2.77 / 2.57, 512: 3.19 / 2.63; at 128 tokens the streaming path wins (1.73 vs 1.93).
    PREFILL_STREAM_MAX = int(os.environ.get("MI355_PREFILL_STREAM_MAX", "128"))

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        limit = min(ops.SKINNY_MAX_M, self.PREFILL_STREAM_MAX) if get_context().is_prefill else ops.SKINNY_MAX_M
        if self.fused and input_ids.numel() <= limit and self._can_stream():
            return self._forward_streaming(input_ids, positions)
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states

    # -- decode-regime fast path ------------------------------------------------------------------
    def _can_stream(self) -> bool:
        l0 = self.layers[0]
        ok = (l0.self_attn.attn.k_cache.numel() > 0 and l0.self_attn.qkv_proj.weight_packed is not None
              and l0.self_attn.o_proj.weight_packed is not None and l0.self_attn.qkv_proj.bias is None
              and l0.self_attn.o_proj.bias is None)
        for layer in self.layers:  # dense MLPs need their packed weights and no biases; sparse blocks their packed experts
            mlp = layer.mlp
            if hasattr(mlp, "experts"):
                ok = ok and mlp.gate_up_packed is not None
            else:
                ok = ok and (mlp.down_proj.weight_packed is not None and mlp.gate_up_proj.weight_packed is not None
                             and mlp.gate_up_proj.bias is None and mlp.down_proj.bias is None)
        return ok

    @staticmethod
    def _ksplit(weight: torch.Tensor) -> int:
        """Split K over enough workgroups that a small-N projection still covers the 256 CUs."""
        n, k = weight.shape
        want = max(1, 256 // max(1, n // 16))  # 512 workgroups measured slower (1.70 vs 1.68 ms per step)
        if want == 1 and k >= 8192:
            # plenty of row tiles but a long K (hidden 5120 models on one GPU: o_proj 5120 x 8192, down 5120 x 25600): four K
            # slices measured 21.8 vs 28.5 us and 71 vs 87 us (profiles/r05_kbench_32b_pipe.txt), the norm sums them
            want = 4
        ks = 1
        while ks * 2 <= min(want, 16) and k % (ks * 2 * 128) == 0:
            ks *= 2
        return ks

    # the decode chain in five launches per layer (csrc/gemm_chain5_kernel.hpp); MI355_CHAIN5=0: the seven-launch chain
    CHAIN5 = os.environ.get("MI355_CHAIN5", "0") != "0"

    def _chain5(self, rows: int, hidden: int, tp: int, exchange: bool) -> bool:
        """One GPU, dense bf16 layers, a batch of at most 32 rows: the shapes the five-launch chain is built for."""
        if not self.CHAIN5 or tp != 1 or exchange:
            return False
        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            if hasattr(mlp, "experts"):
                return False
            lins = (attn.qkv_proj, attn.o_proj, mlp.gate_up_proj, mlp.down_proj)
            if any(isinstance(l.weight_packed, ops.Fp8Weight) for l in lins):
                return False
        l0 = self.layers[0]
        return ops.chain5_takes(rows, hidden, (l0.self_attn.o_proj.weight.shape[1], l0.mlp.down_proj.weight.shape[1]))

    def _forward_streaming(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """<= ops.SKINNY_MAX_M (512) tokens - every decode step, and short prefills - in chunks of 64 rows inside the
        kernels: SEVEN launches per layer (the reference's eager decode layer
        is ~13):
          add+RMSNorm (summing the previous projection's split-K partials)      (mi_add_rmsnorm_splitk)
          -> packed qkv GEMM                                                     (mi_gemm_bf16_packed)
          -> q/k-norm + RoPE + KV store + paged attention in one launch          (mi_paged_attn_decode_fused)
          -> split-K o_proj -> add+RMSNorm -> gate_up GEMM + SwiGLU epilogue -> split-K down_proj.
        MI355_ATTN_FUSED=0 keeps q/k-norm+RoPE+store as its own launch (mi_qknorm_rope_store).
        With tensor parallelism the two row-parallel projections produce bf16 partial sums as complete
        rows (mi_gemm_bf16_rows4); the all-reduce over the ranks (linear.py:149-153) and the following
        add+RMSNorm are ONE launch over xGMI (mi_allreduce_add_rmsnorm), or RCCL all-reduce + add+RMSNorm
        when that path is off.  Rounding points are those of the module-by-module path in every variant
        (tests require equal results up to fp32 summation order)."""
        tp = tp_size()
        exchange = collectives_on()  # several ranks (or the one-rank bring-up hook, layers/parallel.py)
        ctx = get_context()
        h = self.embed_tokens(input_ids)
        rows = h.shape[0]
        l0 = self.layers[0]
        rows4 = tp > 1 and l0.self_attn.o_proj.weight_rows4 is not None
        fuse_attn = not ctx.is_prefill and os.environ.get("MI355_ATTN_FUSED", "1") != "0"

        xgmi = get_xgmi_comm() if tp > 1 else None
        fused_seam = (xgmi is not None and xgmi.fits_rows(rows, h.shape[1])
                      and os.environ.get("MI355_XGMI_FUSED", "1") != "0")

        def attend(attn, qkv):
            if not attn.attn.fusable:  # head_dim 64 / groups of 7 / the plain-layout family: one launch per operator
                return attn._attend_unfused(positions, qkv).flatten(1, -1)
            if not fuse_attn:
                return attn._attend_fused(positions, qkv)
            a, rope = attn.attn, attn.rotary_emb
            a.block_size = ctx.block_size
            if rope.cos_sin_cache.device != qkv.device:
                rope.cos_sin_cache = rope.cos_sin_cache.to(qkv.device)
            qw = attn.q_norm.weight if attn.qk_norm else None
            kw = attn.k_norm.weight if attn.qk_norm else None
            return ops.paged_attn_decode_fused(qkv, qw, kw, attn.rms_norm_eps, positions, rope.cos_sin_cache,
                                               ctx.slot_mapping, a.k_cache, a.v_cache, ctx.block_tables,
                                               ctx.context_lens, attn.num_heads, attn.num_kv_heads, ctx.block_size,
                                               a.scale)

        def row_parallel(x, lin):
            """-> (tensor, is_partials): bf16 rows, or fp32 split-K partials for add_rmsnorm_splitk"""
            if rows4 and lin.weight_rows4 is not None:
                y = ops.gemm_rows4(x, lin.weight_rows4)
            elif not exchange:
                return ops.gemm_packed_splitk(x, lin.weight_packed, self._ksplit(lin.weight)), True
            else:
                y = ops.gemm_packed(x, lin.weight_packed)  # this rank's bf16 partial sums
            if exchange and not fused_seam:
                y = all_reduce_sum(y)
            return y, ("ranks" if fused_seam else False)

        def add_norm(y, is_partials, res, ln):
            """is_partials: True = fp32 split-K partials of this rank (TP 1); "ranks" = bf16 partial sums that
            still have to be summed over the TP ranks (fused seam); False = a finished bf16 tensor."""
            if is_partials is True:
                return ops.add_rmsnorm_splitk(y, res, ln.weight, ln.eps)
            if is_partials == "ranks":  # all-reduce over xGMI + add + RMSNorm in one launch
                return xgmi.allreduce_add_rmsnorm(y, res, ln.weight, ln.eps)
            return ops.add_rmsnorm(y, res, ln.weight, ln.eps)

        def column_parallel(x, lin, silu_mul=False):
            """qkv / gate_up (+ SwiGLU): the weight-streaming kernel, or - large batches, wide projections - the
            128-tile MFMA kernel on the weight as stored (ops.prefers_tile)"""
            w = lin.weight
            if (ops.prefers_tile(rows, w.shape[0]) and w.shape[1] % 64 == 0
                    and not isinstance(lin.weight_packed, ops.Fp8Weight)
                    and (not silu_mul or (w.shape[0] // 2) % 128 == 0)):
                return ops.gemm_tile(x, w, silu_mul=silu_mul)
            return ops.gemm_packed(x, lin.weight_packed, silu_mul=silu_mul)

        def norm_linear(y, is_partials, res, ln, lin, silu_mul=False):
            """linear(rmsnorm(y + res)) -> (out, new residual)"""
            x, res = add_norm(y, is_partials, res, ln)
            return column_parallel(x, lin, silu_mul), res

        if self._chain5(rows, h.shape[1], tp, exchange):
            # FIVE launches per layer: the row-parallel projections add the residual and emit the norm statistic, the
            # column-parallel ones normalise on load (csrc/gemm_chain5_kernel.hpp) - the bits of the seven-launch chain
            residual = s = stat = None
            for layer in self.layers:
                attn, mlp = layer.self_attn, layer.mlp
                ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
                if residual is None:
                    residual = h
                    qkv = ops.gemm_packed(ops.rmsnorm(h, ln1.weight, ln1.eps), attn.qkv_proj.weight_packed)
                else:
                    qkv = ops.gemm_normed(s, stat, ln1.weight, ln1.eps, attn.qkv_proj.weight_packed)
                o = attend(attn, qkv)
                s, residual, stat = ops.gemm_rowstat(o, attn.o_proj.weight_packed, residual, self._ksplit(attn.o_proj.weight))
                act = ops.gemm_normed(s, stat, ln2.weight, ln2.eps, mlp.gate_up_proj.weight_packed, silu_mul=True)
                s, residual, stat = ops.gemm_rowstat(act, mlp.down_proj.weight_packed, residual,
                                                     self._ksplit(mlp.down_proj.weight))
            return ops.norm_from_stat(s, stat, self.norm.weight, self.norm.eps)

        residual, parts, is_partials = None, None, False
        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
            if residual is None:  # first layer: the residual stream starts as the embedding (qwen3.py:137-138)
                residual = h
                qkv = column_parallel(ops.rmsnorm(h, ln1.weight, ln1.eps), attn.qkv_proj)
            else:
                qkv, residual = norm_linear(parts, is_partials, residual, ln1, attn.qkv_proj)
            o = attend(attn, qkv)
            parts, is_partials = row_parallel(o, attn.o_proj)
            if hasattr(mlp, "experts"):  # sparse block (models/qwen3_moe.py): five launches over expert-sorted pairs
                x, residual = add_norm(parts, is_partials, residual, ln2)
                parts, is_partials = mlp(x), False  # summed over the ranks inside the block (before the combine)
            else:
                act, residual = norm_linear(parts, is_partials, residual, ln2, mlp.gate_up_proj, silu_mul=True)
                parts, is_partials = row_parallel(act, mlp.down_proj)
        x, _ = add_norm(parts, is_partials, residual, self.norm)
        return x


class Qwen3ForCausalLM(nn.Module):
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config, fused: bool = True, layer_factory=None, **layer_overrides) -> None:
        super().__init__()
        self.model = Qwen3Model(config, fused, layer_factory, **layer_overrides)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.lm_head(hidden_states)
