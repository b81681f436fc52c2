"""Layer-level CPU oracle (torch on CPU, fp32 internals, explicit bf16 roundings).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Paths cited are relative to the
reference checkout (linzm1007/nano-vllm-ascend).
"""
from __future__ import annotations

import math

import torch

BF16 = torch.bfloat16


# --------------------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm.rms_forward — nanovllm/layers/layernorm.py:16-25.

    y = bf16( bf16(x32 * rsqrt(mean(x32^2) + eps)) * w ): the product with the
    weight is a bf16*bf16 multiply, i.e. a second rounding.
    """
    dt = x.dtype
    x32 = x.float()
    var = x32.pow(2).mean(dim=-1, keepdim=True)
    x32 = x32 * torch.rsqrt(var + eps)
    return x32.to(dt) * w


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float):
    """RMSNorm.add_rms_forward — nanovllm/layers/layernorm.py:27-38.

    s = x32 + r32; residual_out = bf16(s); variance from the UN-rounded s.
    """
    dt = x.dtype
    s = x.float() + residual.float()
    residual_out = s.to(dt)
    var = s.pow(2).mean(dim=-1, keepdim=True)
    s = s * torch.rsqrt(var + eps)
    return s.to(dt) * w, residual_out


# --------------------------------------------------------------------------- rope
def build_cos_sin_cache(head_dim: int, max_position: int, base: float) -> torch.Tensor:
    """RotaryEmbedding.__init__ — nanovllm/layers/rotary_embedding.py:26-35.

    fp32 table [max_position, head_dim] = cat(cos, sin) of outer(t, inv_freq).
    (The reference keeps an extra singleton dim for broadcasting over heads.)
    """
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous()


def apply_rope(positions: torch.Tensor, x: torch.Tensor, cos_sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb / RotaryEmbedding.forward — rotary_embedding.py:6-14, 37-47.

    x: [T, H, D]; NeoX halves; fp32 mul/mul/sub (no fused multiply-add), cast back.
    """
    cs = cos_sin[positions].unsqueeze(1)  # [T, 1, D]
    cos, sin = cs.chunk(2, dim=-1)
    x1, x2 = torch.chunk(x.float(), 2, dim=-1)
    y1 = x1 * cos - x2 * sin
    y2 = x2 * cos + x1 * sin
    return torch.cat((y1, y2), dim=-1).to(x.dtype)


# --------------------------------------------------------------------------- activation
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """SiluAndMul.forward — nanovllm/layers/activation.py:10-12.

    bf16(silu(x_gate)) * x_up with a bf16 multiply (two roundings).
    """
    g, u = x.chunk(2, -1)
    g32 = g.float()
    s = (g32 / (1.0 + torch.exp(-g32))).to(x.dtype)
    return s * u


# --------------------------------------------------------------------------- linear / embedding
def linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, *, keep_fp32: bool = False):
    """F.linear in bf16 — nanovllm/layers/linear.py:51,73,150; embed_head.py:61.

    fp32 accumulation, one rounding to bf16 (what torch's bf16 GEMM kernels do).
    """
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    return y if keep_fp32 else y.to(x.dtype)


def embedding(ids: torch.Tensor, w: torch.Tensor, vocab_start: int = 0) -> torch.Tensor:
    """VocabParallelEmbedding.forward — nanovllm/layers/embed_head.py:34-42 (one rank's part)."""
    local = ids - vocab_start
    mask = (local >= 0) & (local < w.shape[0])
    out = w[torch.where(mask, local, torch.zeros_like(local))]
    return out * mask.unsqueeze(1).to(w.dtype)


# --------------------------------------------------------------------------- mixture of experts
def moe_route(router_logits: torch.Tensor, top_k: int):
    """Qwen3MoeSparseMoeBlock.forward, routing part — nanovllm/models/qwen3_moe.py:150-161.

    softmax over the experts in fp32 (of the bf16 router logits), top-k probabilities, renormalised by their
    sum in fp32, then cast to the activation dtype.  Returns (weights [T, k] in router_logits.dtype, expert ids
    [T, k] int64).  Equal probabilities: the lower expert id first (torch.topk leaves ties unspecified; the
    restatement fixes the order so that CPU and GPU agree)."""
    p = torch.softmax(router_logits.float(), dim=1)
    order = torch.argsort(p, dim=-1, descending=True, stable=True)[:, :top_k]
    w = torch.gather(p, 1, order)
    w = w / w.sum(dim=-1, keepdim=True)
    return w.to(router_logits.dtype), order


def moe_block(x: torch.Tensor, gate_w: torch.Tensor, gate_up_w: torch.Tensor, down_w: torch.Tensor, top_k: int,
              return_routing: bool = False):
    """Qwen3MoeSparseMoeBlock.forward — nanovllm/models/qwen3_moe.py:150-185.

    gate_w [E, H] (nn.Linear, :138); gate_up_w [E, 2I, H] / down_w [E, H, I]: the experts' Qwen3MoeMLP weights
    (:96-122).  For every expert that was selected by some token, in ascending expert id (:171-172): its MLP on
    those tokens (bf16 output), times the routing weight in bf16 (:178-180), added into a bf16 accumulator
    (`index_add_` on a bf16 tensor, :182-184): one rounding per expert a token selected, in that order."""
    T, H = x.shape
    logits = linear(x, gate_w)
    w, ids = moe_route(logits, top_k)
    out = torch.zeros_like(x)
    for e in torch.unique(ids).tolist():  # ascending, hit experts only
        tok, slot = torch.where(ids == e)
        y = linear(silu_and_mul(linear(x[tok], gate_up_w[e])), down_w[e])
        cur = y * w[tok, slot, None]  # bf16 * bf16 -> bf16
        out.index_add_(0, tok, cur.to(x.dtype))
    return (out, w, ids) if return_routing else out


# --------------------------------------------------------------------------- paged KV
def kv_scatter(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
               slot_flat: torch.Tensor) -> None:
    """Attention._store_kvcache — nanovllm/layers/attention.py:22-35.

    Caches are in the reference's logical layout [num_blocks, block_size, Hkv, D];
    slot = block_id*block_size + offset (model_runner.py:263-270); -1 skips
    (attention_torch_native.py:26-27).  Pure copy => bit-exact.
    """
    nblk, bs, hkv, d = k_cache.shape
    kc = k_cache.view(nblk * bs, hkv, d)
    vc = v_cache.view(nblk * bs, hkv, d)
    keep = slot_flat >= 0
    idx = slot_flat[keep].long()
    kc[idx] = k[keep]
    vc[idx] = v[keep]


def _gather_kv(cache: torch.Tensor, block_row: torch.Tensor, n: int) -> torch.Tensor:
    nblk, bs, hkv, d = cache.shape
    nb = (n + bs - 1) // bs
    blocks = block_row[:nb].long()
    return cache[blocks].reshape(nb * bs, hkv, d)[:n]


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, causal_shift: int | None):
    """q [Lq,Hq,D], k/v [Lk,Hkv,D] -> [Lq,Hq,D]; exact fp32 softmax, GQA by head groups."""
    lq, hq, d = q.shape
    lk, hkv, _ = k.shape
    g = hq // hkv
    q32 = q.float().view(lq, hkv, g, d).permute(1, 2, 0, 3)  # [Hkv, G, Lq, D]
    k32 = k.float().permute(1, 0, 2).unsqueeze(1)  # [Hkv, 1, Lk, D]
    v32 = v.float().permute(1, 0, 2).unsqueeze(1)
    s = torch.matmul(q32, k32.transpose(-1, -2)) * scale  # [Hkv, G, Lq, Lk]
    if causal_shift is not None:
        qi = torch.arange(lq).unsqueeze(1) + causal_shift
        kj = torch.arange(lk).unsqueeze(0)
        s = s.masked_fill(kj > qi, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v32)  # [Hkv, G, Lq, D]
    return o.permute(2, 0, 1, 3).reshape(lq, hq, d)


def paged_attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           block_tables: torch.Tensor, context_lens: torch.Tensor, scale: float | None = None,
                           *, keep_fp32: bool = False) -> torch.Tensor:
    """Decode branch of Attention.forward — nanovllm/layers/attention.py:60-93
    (contract restated on CPU by attention_torch_native.py:147-197).

    q [B, Hq, D]; one query token per sequence attends context_lens[b] cached
    tokens reached through block_tables[b]; context_len 0 (graph padding,
    model_runner.py:303-311) yields zeros.  Returns [B, Hq*D].
    """
    b, hq, d = q.shape
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    out = torch.zeros((b, hq, d), dtype=torch.float32)
    for i in range(b):
        n = int(context_lens[i])
        if n <= 0:
            continue
        k = _gather_kv(k_cache, block_tables[i], n)
        v = _gather_kv(v_cache, block_tables[i], n)
        out[i] = _attend(q[i : i + 1], k, v, scale, None)[0]
    out = out.reshape(b, hq * d)
    return out if keep_fp32 else out.to(q.dtype)


def paged_attention_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                            block_tables: torch.Tensor, cu_seqlens_q: torch.Tensor, kv_lens: torch.Tensor,
                            scale: float | None = None, *, keep_fp32: bool = False) -> torch.Tensor:
    """Prefill branch of Attention.forward — nanovllm/layers/attention.py:46-59
    (CPU statement: attention_torch_native.py:101-145): per-sequence causal GQA
    attention.  K/V are read from the cache the step has just filled (same bytes
    as the step's k, v).  Query i of a sequence sits at position kv_len-q_len+i.
    q [T, Hq, D] -> [T, Hq*D].
    """
    t, hq, d = q.shape
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    out = torch.zeros((t, hq, d), dtype=torch.float32)
    for s in range(cu_seqlens_q.numel() - 1):
        a, e = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        n = int(kv_lens[s])
        if e <= a:
            continue
        k = _gather_kv(k_cache, block_tables[s], n)
        v = _gather_kv(v_cache, block_tables[s], n)
        out[a:e] = _attend(q[a:e], k, v, scale, n - (e - a))
    out = out.reshape(t, hq * d)
    return out if keep_fp32 else out.to(q.dtype)


# ---------------------------------------------------------------------------------------------------
# fp8 weights (BASELINE.json configs[4]).  The reference has no fp8 path: this is the published OCP
# 8-bit floating point format E4M3 (1 sign, 4 exponent bits with bias 7, 3 mantissa bits, no infinities,
# maximum 448, subnormals down to 2^-9) restated with integer arithmetic - independent of
# torch.float8_e4m3fn, which the product's quantiser uses - plus the per-row power-of-two scale rule
# of nanovllm.ops.quantize_fp8 (row maximum scaled into [128, 256)).
def e4m3_encode(x: torch.Tensor) -> torch.Tensor:
    """Round fp32 values (|x| <= 448) to E4M3, nearest even; returns the bytes (uint8)."""
    x = x.double()
    sign = (x < 0) | ((x == 0) & (torch.signbit(x)))
    a = x.abs()
    e = torch.floor(torch.log2(torch.where(a > 0, a, torch.ones_like(a)))).clamp(min=-6, max=8)
    quantum = torch.pow(2.0, e - 3)                      # spacing of representable values in this binade
    n = a / quantum                                      # in [8, 16) for normals, [0, 8) for subnormals
    r = torch.floor(n)
    frac = n - r
    r = r + ((frac > 0.5) | ((frac == 0.5) & (r % 2 == 1))).double()   # ties to even
    carry = r >= 16                                      # rounded up into the next binade
    e = torch.where(carry, e + 1, e)
    r = torch.where(carry, torch.full_like(r, 8.0), r)
    is_sub = r < 8                                       # only possible at e == -6
    exp_field = torch.where(is_sub, torch.zeros_like(e), e + 7)
    mant = torch.where(is_sub, r, r - 8)
    byte = (sign.long() << 7) | (exp_field.long() << 3) | mant.long()
    assert int(byte.max()) <= 0xFF and not bool(((byte & 0x7F) == 0x7F).any()), "value outside E4M3's finite range"
    return byte.to(torch.uint8)


def e4m3_decode(b: torch.Tensor) -> torch.Tensor:
    b = b.long()
    sign = torch.where((b >> 7) == 1, -1.0, 1.0)
    exp_field, mant = (b >> 3) & 0xF, (b & 7).double()
    normal = torch.pow(2.0, exp_field.double() - 7) * (1 + mant / 8)
    sub = torch.pow(torch.tensor(2.0, dtype=torch.float64), -6) * (mant / 8)
    return (sign * torch.where(exp_field == 0, sub, normal)).float()


def quantize_fp8_rows(w: torch.Tensor):
    """(E4M3 bytes [N, K], fp32 scale [N]): scale = 2^(exponent of the row maximum - 8), 2^-8 for a zero row."""
    wf = w.float()
    amax = wf.abs().amax(dim=1)
    exp = torch.where(amax > 0, torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)))) + 1,
                      torch.zeros_like(amax))           # amax = m * 2^exp with m in [0.5, 1)
    scale = torch.pow(2.0, exp - 8)
    return e4m3_encode(wf / scale[:, None]), scale.float()


def dequantize_fp8_rows(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return e4m3_decode(q) * scale[:, None]
