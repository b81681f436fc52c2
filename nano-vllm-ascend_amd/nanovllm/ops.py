"""Tensor-level wrappers over the C ABI: shape checks, output allocation, strides.

Nothing here computes anything: every function ends in exactly one (attention
decode: two) kernel launch inside libmi355_nanovllm.so on torch's current stream.
"""
from __future__ import annotations

import torch

from nanovllm import _C
from nanovllm._C import HEAD_DIM, KV_TILE_ELEMS, check, lib, ptr, require_gpu, stream

_BF16 = torch.bfloat16


def _bf16(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and t.dtype != _BF16:
            raise _C.MiError(f"expected bfloat16 tensor, got {t.dtype}")


# --------------------------------------------------------------------------- KV cache
def kv_cache_shape(num_blocks: int, n_kv_heads: int, block_size: int, head_dim: int = HEAD_DIM) -> tuple[int, int, int, int]:
    """Shape of one layer's K (or V) cache in the fragment-native layout: one tile of 16 tokens x head_dim (4 KiB at
    head_dim 128, 2 KiB at 64) per (16 slots, kv head)."""
    assert block_size % _C.KV_TILE_TOKENS == 0 and head_dim in (64, HEAD_DIM)
    return (num_blocks, n_kv_heads, block_size // _C.KV_TILE_TOKENS, _C.KV_TILE_TOKENS * head_dim)


def _head_dim_of(cache: torch.Tensor) -> int:
    """head_dim of a fragment-native cache tensor [..., 16 * head_dim]"""
    return cache.shape[-1] // _C.KV_TILE_TOKENS


def reshape_and_cache(k, v, k_cache, v_cache, slot_flat, n_kv_heads: int, block_size: int) -> None:
    require_gpu(k, v, k_cache, v_cache, slot_flat)
    _bf16(k, v, k_cache, v_cache)
    n = k.shape[0]
    assert k.stride(-1) == 1 and v.stride(-1) == 1
    assert slot_flat.dtype == torch.int32 and slot_flat.is_contiguous() and slot_flat.numel() == n
    check(
        lib.mi_reshape_and_cache(ptr(k), ptr(v), k.stride(0), v.stride(0), ptr(k_cache), ptr(v_cache),
                                 ptr(slot_flat), n, n_kv_heads, _head_dim_of(k_cache), block_size, stream()),
        "mi_reshape_and_cache",
    )


def scatter_update_kv(k, v, k_cache, v_cache, slot_2d, n_kv_heads: int, block_size: int) -> None:
    require_gpu(k, v, k_cache, v_cache, slot_2d)
    _bf16(k, v, k_cache, v_cache)
    n = k.shape[0]
    assert slot_2d.dtype == torch.int32 and slot_2d.is_contiguous() and slot_2d.shape == (n, 2)
    check(
        lib.mi_scatter_update_kv(ptr(k), ptr(v), k.stride(0), v.stride(0), ptr(k_cache), ptr(v_cache),
                                 ptr(slot_2d), n, n_kv_heads, _head_dim_of(k_cache), block_size, stream()),
        "mi_scatter_update_kv",
    )


def kv_cache_gather(cache, is_v: bool, slot_flat, n_kv_heads: int, block_size: int) -> torch.Tensor:
    """Rows of the cache in the reference's logical layout [n, n_kv_heads*128]."""
    require_gpu(cache, slot_flat)
    n, D = slot_flat.numel(), _head_dim_of(cache)
    out = torch.empty((n, n_kv_heads * D), dtype=_BF16, device=cache.device)
    check(
        lib.mi_kv_cache_gather(ptr(cache), int(is_v), ptr(slot_flat), n, ptr(out), n_kv_heads, D,
                               block_size, stream()),
        "mi_kv_cache_gather",
    )
    return out


# --------------------------------------------------------------------------- attention
_WORKSPACES: dict[torch.device, torch.Tensor] = {}


def attn_workspace(device, batch: int, n_q_heads: int) -> torch.Tensor:
    need = lib.mi_paged_attn_decode_workspace(batch, n_q_heads)
    ws = _WORKSPACES.get(device)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=device)  # arrival counters start at zero
        _WORKSPACES[device] = ws
    return ws


def paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, n_q_heads: int, n_kv_heads: int,
                      block_size: int, scale: float, out=None, workspace=None) -> torch.Tensor:
    require_gpu(q, k_cache, v_cache, block_tables, context_lens)
    _bf16(q, k_cache, v_cache)
    batch, D = q.shape[0], _head_dim_of(k_cache)
    q2 = q.view(batch, -1) if q.dim() == 3 else q
    assert q2.stride(1) == 1 and q2.shape[1] == n_q_heads * D
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert context_lens.dtype == torch.int32 and context_lens.is_contiguous()
    if out is None:
        out = torch.empty((batch, n_q_heads * D), dtype=_BF16, device=q.device)
    if workspace is None:
        workspace = attn_workspace(q.device, batch, n_q_heads)
    check(
        lib.mi_paged_attn_decode(ptr(q2), q2.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                 block_tables.stride(0), ptr(context_lens), ptr(out), ptr(workspace),
                                 workspace.numel() * workspace.element_size(), batch, n_q_heads, n_kv_heads,
                                 D, block_size, float(scale), stream()),
        "mi_paged_attn_decode",
    )
    return out


def paged_attn_decode_fused(qkv, q_w, k_w, eps: float, positions, cos_sin, slots_2d, k_cache, v_cache, block_tables,
                            context_lens, n_q_heads: int, n_kv_heads: int, block_size: int, scale: float,
                            out=None, workspace=None) -> torch.Tensor:
    """q/k-norm + RoPE + KV store of the step's token + paged decode attention in ONE launch, straight
    from the packed qkv rows (bit-identical to qknorm_rope_store followed by paged_attn_decode)."""
    require_gpu(qkv, positions, cos_sin, slots_2d, k_cache, v_cache, block_tables, context_lens)
    _bf16(qkv, k_cache, v_cache, q_w, k_w)
    batch = qkv.shape[0]
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.shape[1] == (n_q_heads + 2 * n_kv_heads) * HEAD_DIM
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == batch
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
    assert slots_2d.dtype == torch.int32 and slots_2d.is_contiguous() and slots_2d.shape == (batch, 2)
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert context_lens.dtype == torch.int32 and context_lens.is_contiguous()
    if out is None:
        out = torch.empty((batch, n_q_heads * HEAD_DIM), dtype=_BF16, device=qkv.device)
    if workspace is None:
        workspace = attn_workspace(qkv.device, batch, n_q_heads)
    check(
        lib.mi_paged_attn_decode_fused(ptr(qkv), qkv.stride(0), ptr(q_w), ptr(k_w), float(eps), ptr(positions),
                                       ptr(cos_sin), ptr(slots_2d), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                       block_tables.stride(0), ptr(context_lens), ptr(out), ptr(workspace),
                                       workspace.numel() * workspace.element_size(), batch, n_q_heads, n_kv_heads,
                                       HEAD_DIM, block_size, float(scale), stream()),
        "mi_paged_attn_decode_fused",
    )
    return out


def paged_attn_decode_fused_stamped(qkv, q_w, k_w, eps: float, positions, cos_sin, slots_2d, k_cache, v_cache,
                                    block_tables, context_lens, n_q_heads: int, n_kv_heads: int, block_size: int,
                                    scale: float, stamps, out=None, workspace=None) -> torch.Tensor:
    """paged_attn_decode_fused through the instrumented kernel (mi_paged_attn_decode_fused_ex): `stamps`
    [batch * n_kv_heads * splits, 8 waves, 8] int64 receives every wave's s_memrealtime (100 MHz) at eight points of its life."""
    require_gpu(qkv, positions, cos_sin, slots_2d, k_cache, v_cache, block_tables, context_lens, stamps)
    _bf16(qkv, k_cache, v_cache, q_w, k_w)
    batch = qkv.shape[0]
    assert stamps.dtype == torch.int64 and stamps.is_contiguous() and stamps.numel() >= batch * n_kv_heads * 16 * 64
    if out is None:
        out = torch.empty((batch, n_q_heads * HEAD_DIM), dtype=_BF16, device=qkv.device)
    if workspace is None:
        workspace = attn_workspace(qkv.device, batch, n_q_heads)
    check(
        lib.mi_paged_attn_decode_fused_ex(ptr(qkv), qkv.stride(0), ptr(q_w), ptr(k_w), float(eps), ptr(positions),
                                          ptr(cos_sin), ptr(slots_2d), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                          block_tables.stride(0), ptr(context_lens), ptr(out), ptr(workspace),
                                          workspace.numel() * workspace.element_size(), batch, n_q_heads, n_kv_heads,
                                          HEAD_DIM, block_size, float(scale), ptr(stamps), stream()),
        "mi_paged_attn_decode_fused_ex",
    )
    return out


def paged_attn_prefill(q, k_cache, v_cache, block_tables, cu_seqlens_q, kv_lens, max_seqlen_q: int,
                       n_q_heads: int, n_kv_heads: int, block_size: int, scale: float, out=None) -> torch.Tensor:
    require_gpu(q, k_cache, v_cache, block_tables, cu_seqlens_q, kv_lens)
    _bf16(q, k_cache, v_cache)
    T, D = q.shape[0], _head_dim_of(k_cache)
    q2 = q.view(T, -1) if q.dim() == 3 else q
    assert q2.stride(1) == 1 and q2.shape[1] == n_q_heads * D
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert cu_seqlens_q.dtype == torch.int32 and kv_lens.dtype == torch.int32
    n_seqs = cu_seqlens_q.numel() - 1
    assert kv_lens.numel() == n_seqs and block_tables.shape[0] >= n_seqs
    if out is None:
        out = torch.empty((T, n_q_heads * D), dtype=_BF16, device=q.device)
    check(
        lib.mi_paged_attn_prefill(ptr(q2), q2.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                  block_tables.stride(0), ptr(cu_seqlens_q), ptr(kv_lens), n_seqs,
                                  int(max_seqlen_q), ptr(out), n_q_heads, n_kv_heads, D, block_size,
                                  float(scale), stream()),
        "mi_paged_attn_prefill",
    )
    return out


def paged_attn_prefill_fused(qkv, q_w, eps: float, positions, cos_sin, k_cache, v_cache, block_tables, cu_seqlens_q,
                             kv_lens, max_seqlen_q: int, n_q_heads: int, n_kv_heads: int, block_size: int, scale: float,
                             out=None, variant: int | None = None) -> torch.Tensor:
    """paged_attn_prefill over the RAW q heads of the packed qkv rows: q-norm (q_w may be None) + RoPE happen in
    the kernel's Q-operand load.  K / V must already be cached (qknorm_rope_store(..., store_q=False))."""
    require_gpu(qkv, positions, cos_sin, k_cache, v_cache, block_tables, cu_seqlens_q, kv_lens)
    _bf16(qkv, k_cache, v_cache)
    T = qkv.shape[0]
    assert qkv.stride(1) == 1 and qkv.shape[1] == (n_q_heads + 2 * n_kv_heads) * HEAD_DIM
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == T
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert cu_seqlens_q.dtype == torch.int32 and kv_lens.dtype == torch.int32
    n_seqs = cu_seqlens_q.numel() - 1
    if out is None:
        out = torch.empty((T, n_q_heads * HEAD_DIM), dtype=_BF16, device=qkv.device)
    if variant is not None:
        check(
            lib.mi_paged_attn_prefill_fused_ex(ptr(qkv), qkv.stride(0), ptr(q_w), float(eps), ptr(positions),
                                               ptr(cos_sin), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                               block_tables.stride(0), ptr(cu_seqlens_q), ptr(kv_lens), n_seqs,
                                               int(max_seqlen_q), ptr(out), n_q_heads, n_kv_heads, HEAD_DIM, block_size,
                                               float(scale), int(variant), stream()),
            "mi_paged_attn_prefill_fused_ex",
        )
        return out
    check(
        lib.mi_paged_attn_prefill_fused(ptr(qkv), qkv.stride(0), ptr(q_w), float(eps), ptr(positions), ptr(cos_sin),
                                        ptr(k_cache), ptr(v_cache), ptr(block_tables), block_tables.stride(0),
                                        ptr(cu_seqlens_q), ptr(kv_lens), n_seqs, int(max_seqlen_q), ptr(out),
                                        n_q_heads, n_kv_heads, HEAD_DIM, block_size, float(scale), stream()),
        "mi_paged_attn_prefill_fused",
    )
    return out


# --------------------------------------------------------------------------- norms
def rmsnorm(x, w, eps: float, out=None) -> torch.Tensor:
    """x: [rows, cols] (contiguous rows) or [T, H, cols] with arbitrary token stride."""
    require_gpu(x, w)
    _bf16(x, w)
    cols = x.shape[-1]
    assert x.stride(-1) == 1 and w.numel() == cols
    if x.dim() == 3:
        outer, inner = x.shape[0], x.shape[1]
        assert x.stride(1) == cols
        outer_stride = x.stride(0)
    else:
        x2 = x.reshape(-1, cols) if x.is_contiguous() else x
        assert x2.dim() == 2
        outer, inner, outer_stride = x2.shape[0], 1, x2.stride(0)
    if out is None:
        out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    check(lib.mi_rmsnorm(ptr(x), outer_stride, ptr(w), ptr(out), outer, inner, cols, float(eps), stream()),
          "mi_rmsnorm")
    return out


def add_rmsnorm(x, residual, w, eps: float, out=None, residual_out=None):
    require_gpu(x, residual, w)
    _bf16(x, residual, w)
    assert x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape
    cols = x.shape[-1]
    rows = x.numel() // cols
    if out is None:
        out = torch.empty_like(x)
    if residual_out is None:
        residual_out = torch.empty_like(x)
    check(
        lib.mi_add_rmsnorm(ptr(x), ptr(residual), ptr(w), ptr(out), ptr(residual_out), rows, cols, float(eps),
                           stream()),
        "mi_add_rmsnorm",
    )
    return out, residual_out


# --------------------------------------------------------------------------- rope (+ fused)
# ---- plain-layout attention family (csrc/attn_plain.hip): head_dim 64, GQA groups that are not a power of two ----
def attention_is_plain(n_q_heads: int, n_kv_heads: int, head_dim: int) -> bool:
    """True: this head geometry runs on the mi_*_plain kernels and the [blocks, kv heads, block, head_dim] cache.
    Round 4: head_dim 64 and 7 query heads per kv head (Llama-3.2-1B, Qwen2-0.5B, Qwen2.5-7B - the reference's README
    models) run on the fragment-native MFMA kernels; the plain family remains for what is left (e.g. groups of 3, 5, 6)."""
    if n_q_heads % n_kv_heads != 0:
        return True
    return not (head_dim in (64, HEAD_DIM) and (n_q_heads // n_kv_heads) in (1, 2, 4, 7, 8, 16))


def attention_is_fusable(n_q_heads: int, n_kv_heads: int, head_dim: int) -> bool:
    """True: the fused forms apply (q/k-norm + RoPE + KV store inside the decode attention launch, q prepared in the
    prefill attention's operand load): written for 128-wide heads and power-of-two groups."""
    return head_dim == HEAD_DIM and n_q_heads % n_kv_heads == 0 and (n_q_heads // n_kv_heads) in (1, 2, 4, 8, 16)


def attention_plain_supported(n_q_heads: int, n_kv_heads: int, head_dim: int) -> bool:
    return head_dim in (64, 128) and n_q_heads % n_kv_heads == 0 and n_q_heads // n_kv_heads <= 8


def kv_cache_shape_plain(num_blocks: int, n_kv_heads: int, block_size: int, head_dim: int) -> tuple[int, int, int, int]:
    return (num_blocks, n_kv_heads, block_size, head_dim)


def kv_store_plain(k, v, k_cache, v_cache, slots, n_kv_heads: int, block_size: int) -> None:
    """k, v: [T, Hkv, D] (token stride free); slots: flat int32 [T] or [T, 2] (block, offset) pairs."""
    require_gpu(k, v, k_cache, v_cache, slots)
    _bf16(k, v, k_cache, v_cache)
    T, D = k.shape[0], k_cache.shape[-1]
    k3 = k.view(T, n_kv_heads, D) if k.dim() == 2 else k
    v3 = v.view(T, n_kv_heads, D) if v.dim() == 2 else v
    assert k3.stride(2) == 1 and k3.stride(1) == D and v3.stride(2) == 1 and v3.stride(1) == D
    assert slots.dtype == torch.int32 and slots.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous()
    check(lib.mi_kv_store_plain(ptr(k3), ptr(v3), k3.stride(0), v3.stride(0), ptr(k_cache), ptr(v_cache), ptr(slots),
                                int(slots.dim() == 2), T, n_kv_heads, D, block_size, stream()), "mi_kv_store_plain")


_PLAIN_WS: dict[torch.device, torch.Tensor] = {}


def paged_attn_decode_plain(q, k_cache, v_cache, block_tables, context_lens, n_q_heads: int, n_kv_heads: int,
                            block_size: int, scale: float, out=None) -> torch.Tensor:
    require_gpu(q, k_cache, v_cache, block_tables, context_lens)
    _bf16(q, k_cache, v_cache)
    batch, D = q.shape[0], k_cache.shape[-1]
    q2 = q.view(batch, -1) if q.dim() == 3 else q
    assert q2.stride(1) == 1 and q2.shape[1] == n_q_heads * D and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert context_lens.dtype == torch.int32 and context_lens.is_contiguous()
    if out is None:
        out = torch.empty((batch, n_q_heads * D), dtype=_BF16, device=q.device)
    need = lib.mi_paged_attn_decode_plain_workspace(max(batch, 1), n_q_heads, D)
    ws = _PLAIN_WS.get(q.device)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _GEMM_WS_RETIRED.append(ws)  # a captured graph may hold its address
        ws = _PLAIN_WS[q.device] = torch.empty(need, dtype=torch.uint8, device=q.device)
    check(lib.mi_paged_attn_decode_plain(ptr(q2), q2.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                         block_tables.stride(0), ptr(context_lens), ptr(out), ptr(ws), ws.numel(), batch,
                                         n_q_heads, n_kv_heads, D, block_size, float(scale), stream()),
          "mi_paged_attn_decode_plain")
    return out


def paged_attn_prefill_plain(q, k_cache, v_cache, block_tables, cu_seqlens_q, kv_lens, max_seqlen_q: int, n_q_heads: int,
                             n_kv_heads: int, block_size: int, scale: float, out=None) -> torch.Tensor:
    require_gpu(q, k_cache, v_cache, block_tables, cu_seqlens_q, kv_lens)
    _bf16(q, k_cache, v_cache)
    T, D = q.shape[0], k_cache.shape[-1]
    q2 = q.view(T, -1) if q.dim() == 3 else q
    assert q2.stride(1) == 1 and q2.shape[1] == n_q_heads * D and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    assert cu_seqlens_q.dtype == torch.int32 and kv_lens.dtype == torch.int32
    if out is None:
        out = torch.empty((T, n_q_heads * D), dtype=_BF16, device=q.device)
    check(lib.mi_paged_attn_prefill_plain(ptr(q2), q2.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                          block_tables.stride(0), ptr(cu_seqlens_q), ptr(kv_lens),
                                          cu_seqlens_q.numel() - 1, int(max_seqlen_q), ptr(out), n_q_heads, n_kv_heads, D,
                                          block_size, float(scale), stream()), "mi_paged_attn_prefill_plain")
    return out


def rope(positions, q, k, cos_sin, n_q_heads: int, n_kv_heads: int):
    """q: [T, Hq, D] / k: [T, Hkv, D] views (token stride free) -> contiguous copies (D = 128: mi_rope, else
    mi_rope_plain)."""
    require_gpu(positions, q, k, cos_sin)
    _bf16(q, k)
    T = q.shape[0]
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == T
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
    D = cos_sin.shape[-1]
    if D != HEAD_DIM:
        q3 = q.view(T, n_q_heads, D) if q.dim() == 2 else q
        k3 = k.view(T, n_kv_heads, D) if k.dim() == 2 else k
        assert q3.stride(2) == 1 and q3.stride(1) == D and k3.stride(2) == 1 and k3.stride(1) == D
        q_out = torch.empty((T, n_q_heads, D), dtype=_BF16, device=q.device)
        k_out = torch.empty((T, n_kv_heads, D), dtype=_BF16, device=q.device)
        check(lib.mi_rope_plain(ptr(positions), ptr(q3), q3.stride(0), ptr(k3), k3.stride(0), ptr(cos_sin), ptr(q_out),
                                ptr(k_out), T, n_q_heads, n_kv_heads, D, stream()), "mi_rope_plain")
        return q_out, k_out
    q3 = q.view(T, n_q_heads, HEAD_DIM) if q.dim() == 2 else q
    k3 = k.view(T, n_kv_heads, HEAD_DIM) if k.dim() == 2 else k
    assert q3.stride(2) == 1 and q3.stride(1) == HEAD_DIM and k3.stride(2) == 1 and k3.stride(1) == HEAD_DIM
    q_out = torch.empty((T, n_q_heads, HEAD_DIM), dtype=_BF16, device=q.device)
    k_out = torch.empty((T, n_kv_heads, HEAD_DIM), dtype=_BF16, device=q.device)
    check(
        lib.mi_rope(ptr(positions), ptr(cos_sin), ptr(q3), q3.stride(0), n_q_heads, ptr(k3), k3.stride(0),
                    n_kv_heads, ptr(q_out), ptr(k_out), T, HEAD_DIM, stream()),
        "mi_rope",
    )
    return q_out, k_out


def qknorm_rope_store(qkv, q_w, k_w, eps: float, positions, cos_sin, k_cache, v_cache, slots,
                      n_q_heads: int, n_kv_heads: int, block_size: int, q_out=None, store_q: bool = True):
    """store_q=False (prefill-sized calls only): K and V go to the cache, the queries are left to
    paged_attn_prefill_fused; returns None."""
    require_gpu(qkv, positions, cos_sin, k_cache, v_cache, slots)
    _bf16(qkv, k_cache, v_cache)
    T = qkv.shape[0]
    assert qkv.stride(1) == 1 and qkv.shape[1] == (n_q_heads + 2 * n_kv_heads) * HEAD_DIM
    assert positions.dtype == torch.int64 and positions.is_contiguous()
    assert slots.dtype == torch.int32 and slots.is_contiguous()
    is2d = int(slots.dim() == 2)
    if not store_q:
        q_out = None
    elif q_out is None:
        q_out = torch.empty((T, n_q_heads * HEAD_DIM), dtype=_BF16, device=qkv.device)
    check(
        lib.mi_qknorm_rope_store(ptr(qkv), qkv.stride(0), ptr(q_w), ptr(k_w), float(eps), ptr(positions),
                                 ptr(cos_sin), ptr(q_out), ptr(k_cache), ptr(v_cache), ptr(slots), is2d, T,
                                 n_q_heads, n_kv_heads, HEAD_DIM, block_size, stream()),
        "mi_qknorm_rope_store",
    )
    return q_out


# --------------------------------------------------------------------------- mlp / linear / embed
def silu_mul(x, out=None) -> torch.Tensor:
    require_gpu(x)
    _bf16(x)
    assert x.is_contiguous()
    inter = x.shape[-1] // 2
    rows = x.numel() // x.shape[-1]
    if out is None:
        out = torch.empty((*x.shape[:-1], inter), dtype=_BF16, device=x.device)
    check(lib.mi_silu_mul(ptr(x), ptr(out), rows, inter, stream()), "mi_silu_mul")
    return out


# rows the weight-streaming decode kernels take in one launch (chunks of 64 rows, csrc kSkinnyMaxRows); the decode
# fast path of the models, the in-graph sampler and the engine's lookahead all go up to this many sequences
SKINNY_MAX_M = 512
# ... but above this many rows a WIDE projection (qkv, gate_up: thousands of output features = enough 128 x 128 tiles
# for the chip) is faster on the MFMA tile kernel: 12.7 vs 15.1 us (qkv) and 13.1 vs 21.7 us (gate_up + SwiGLU) at 256
# rows, 13.4 vs 28.0 and 14.3 vs 50.0 at 512; the narrow row-parallel projections (o_proj, down: N = hidden) stay on
# the streaming kernels up to SKINNY_MAX_M rows (7.4 vs 13.2 us at 256) - tools/gemm_bench.py, GEMM_MID=1
TILE_MIN_ROWS = 128
TILE_MIN_FEATURES = 2048


def prefers_tile(rows: int, n_features: int) -> bool:
    return rows > SKINNY_MAX_M or (rows > TILE_MIN_ROWS and n_features >= TILE_MIN_FEATURES)


def gemm_skinny(x, w, bias=None, out=None) -> torch.Tensor:
    """y = x @ w.T (+ bias) for M <= SKINNY_MAX_M rows (decode)."""
    require_gpu(x, w, bias)
    _bf16(x, w, bias)
    assert x.is_contiguous() and w.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((*x.shape[:-1], N), dtype=_BF16, device=x.device)
    check(lib.mi_gemm_bf16_skinny(ptr(x), ptr(w), ptr(bias), ptr(out), M, N, K, stream()), "mi_gemm_bf16_skinny")
    return out


def tile_gemm_max_rows(n: int, k: int, ldx: int | None = None) -> int:
    """mi_gemm_bf16's shape contract, asked of the library itself: the most rows one launch takes on a [n][k] weight
    (activation row stride ldx, default k); 0 = the kernel refuses the weight shape."""
    return int(lib.mi_gemm_bf16_max_rows(n, k, k if ldx is None else ldx))


def gemm_tile(x, w, bias=None, out=None, silu_mul: bool = False, variant: int | None = None) -> torch.Tensor:
    """y = x @ w.T (+ bias) for any number of rows on the MFMA tile kernels (mi_gemm_bf16: 256 x 256 tiles, 128 x 128
    for shapes with few of those); silu_mul:
    w stacks gate | up rows and y = SiluAndMul(x @ w.T).  x may be a row-strided 2-D view."""
    require_gpu(x, w, bias)
    _bf16(x, w, bias)
    assert w.dim() == 2 and w.is_contiguous() and x.stride(-1) == 1
    K = x.shape[-1]
    N = w.shape[0]
    assert w.shape[1] == K
    if x.dim() != 2 or x.stride(0) % 8:
        x = x.reshape(-1, K).contiguous()
    M = x.shape[0]
    n_out = N // 2 if silu_mul else N
    if out is None:
        out = torch.empty((M, n_out), dtype=_BF16, device=x.device)
    assert out.dim() == 2 and out.shape == (M, n_out) and out.stride(1) == 1
    max_rows = tile_gemm_max_rows(N, K, x.stride(0))
    if M > max_rows >= 256 and variant is None:
        # more rows than the kernel's 32-bit operand offsets reach (65 536 tokens x K = 25 600): whole-tile
        # row pieces, each its own launch over the same weight - same bits, every row's K chain is one tile's
        step = max_rows // 256 * 256
        for r0 in range(0, M, step):
            gemm_tile(x[r0:r0 + step], w, bias, out[r0:r0 + step], silu_mul)
        return out
    if variant is not None:
        assert bias is None and not silu_mul
        check(lib.mi_gemm_bf16_ex(ptr(x), x.stride(0), ptr(w), ptr(out), out.stride(0), M, N, K, variant, stream()),
              "mi_gemm_bf16_ex")
        return out
    need = lib.mi_gemm_bf16_workspace(M, N, K, int(silu_mul))
    ws = _gemm_workspace(x.device, need) if need else None
    check(lib.mi_gemm_bf16(ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(out), out.stride(0), M, N, K, int(silu_mul),
                           ptr(ws), need, stream()), "mi_gemm_bf16")
    return out


def qkv_store_takes(rows: int, n_features: int, head_dim: int, block_size: int) -> bool:
    """Shapes mi_gemm_bf16_qkv_store accepts: the large-M tile kernel (at least 256 tiles of 256 x 256), head_dim 128,
    whole feature tiles, block sizes in whole cache tiles."""
    tiles = -(-rows // 256) * -(-n_features // 256)
    return head_dim == HEAD_DIM and n_features % 256 == 0 and block_size % 16 == 0 and tiles >= 256


def gemm_qkv_store(x, w, bias, k_w, eps: float, positions, cos_sin, k_cache, v_cache, slots, n_q_heads: int,
                   n_kv_heads: int, block_size: int, out=None) -> torch.Tensor:
    """The packed qkv projection of a prefill step with k-norm + RoPE + the K / V cache store in the GEMM's epilogue
    (mi_gemm_bf16_qkv_store): returns the [M][N] qkv rows of which ONLY the q columns are written."""
    require_gpu(x, w, bias, positions, cos_sin, k_cache, v_cache, slots)
    _bf16(x, w, bias, k_cache, v_cache)
    assert w.dim() == 2 and w.is_contiguous() and x.stride(-1) == 1
    K, N = x.shape[-1], w.shape[0]
    if x.dim() != 2 or x.stride(0) % 8:
        x = x.reshape(-1, K).contiguous()
    M = x.shape[0]
    assert N == (n_q_heads + 2 * n_kv_heads) * HEAD_DIM and positions.dtype == torch.int64 and positions.is_contiguous()
    assert slots.dtype == torch.int32 and slots.dim() == 1 and slots.is_contiguous() and slots.numel() == M
    if out is None:
        out = torch.empty((M, N), dtype=_BF16, device=x.device)
    check(lib.mi_gemm_bf16_qkv_store(ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(out), out.stride(0), M, N, K, ptr(k_w),
                                     float(eps), ptr(positions), ptr(cos_sin), ptr(k_cache), ptr(v_cache), ptr(slots),
                                     n_q_heads, n_kv_heads, HEAD_DIM, block_size, stream()), "mi_gemm_bf16_qkv_store")
    return out


_GEMM_WS: dict[tuple, torch.Tensor] = {}
_GEMM_WS_RETIRED: list[torch.Tensor] = []


def _gemm_workspace(device, nbytes: int) -> torch.Tensor:
    """split-K partial sums of the tile GEMM (one buffer per device AND stream, grown on demand; launches on one
    stream are ordered, so consecutive GEMMs may share it - the prefill micro-batches of ModelRunner run on two)"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)  # per stream: two streams never share partial sums
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:  # a captured graph may hold the old buffer's address: it stays allocated
            _GEMM_WS_RETIRED.append(ws)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = ws
    return ws


def pack_weight(w, out=None) -> torch.Tensor:
    """Fragment-native copy of a [N, K] weight for the decode GEMMs (same shape/bytes).
    Pass the previous copy as `out` to refresh it in place (captured graphs keep its address)."""
    require_gpu(w)
    _bf16(w)
    assert w.dim() == 2 and w.is_contiguous()
    if out is None or out.shape != w.shape or out.device != w.device:
        out = torch.empty_like(w)
    check(lib.mi_pack_weight(ptr(w), ptr(out), w.shape[0], w.shape[1], stream()), "mi_pack_weight")
    return out


class Fp8Weight:
    """Fragment-native e4m3 copy of a [N, K] weight plus one fp32 scale per row (mi_pack_weight_fp8)."""
    __slots__ = ("data", "scale", "shape")

    def __init__(self, data: torch.Tensor, scale: torch.Tensor, shape):
        self.data, self.scale, self.shape = data, scale, tuple(shape)


FP8_MAX = 448.0  # largest finite OCP e4m3 value


def quantize_fp8(w: torch.Tensor):
    """Per-row symmetric quantisation: (uint8 view of e4m3 [N, K], fp32 scale [N]).  The scale is a
    power of two (row maximum lands in [128, 256) of e4m3's +-448): e4m3 is a floating-point format, so a
    power-of-two scale costs no precision, and scaling by it is exact - CPU and GPU produce the same bytes."""
    wf = w.float()
    _, exp = torch.frexp(wf.abs().amax(dim=1))          # amax = m * 2^exp, m in [0.5, 1)
    exp = torch.where(wf.abs().amax(dim=1) > 0, exp, torch.zeros_like(exp))
    scale = torch.ldexp(torch.ones_like(wf[:, 0]), exp - 8)
    q = torch.ldexp(wf, (8 - exp)[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_fp8(q_u8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q_u8.view(torch.float8_e4m3fn).float() * scale[:, None]


def pack_weight_fp8(w, out: Fp8Weight | None = None) -> Fp8Weight:
    """Quantise a bf16 [N, K] weight and lay it out for mi_gemm_fp8w_packed (refreshes `out` in place)."""
    require_gpu(w)
    assert w.dim() == 2 and w.is_contiguous() and w.shape[0] % 16 == 0 and w.shape[1] % 64 == 0
    q, scale = quantize_fp8(w)
    if out is None or out.shape != tuple(w.shape) or out.data.device != w.device:
        out = Fp8Weight(torch.empty_like(q), torch.empty_like(scale), w.shape)
    out.scale.copy_(scale)
    check(lib.mi_pack_weight_fp8(ptr(q), ptr(out.data), w.shape[0], w.shape[1], stream()), "mi_pack_weight_fp8")
    return out


def gemm_packed(x, w_packed, bias=None, out=None, silu_mul: bool = False) -> torch.Tensor:
    require_gpu(x, bias)
    _bf16(x, bias)
    assert x.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = w_packed.shape[0]
    assert w_packed.shape[1] == K
    n_out = N // 2 if silu_mul else N
    if out is None:
        out = torch.empty((*x.shape[:-1], n_out), dtype=_BF16, device=x.device)
    if isinstance(w_packed, Fp8Weight):
        assert bias is None
        check(lib.mi_gemm_fp8w_packed(ptr(x), ptr(w_packed.data), ptr(w_packed.scale), ptr(out), M, N, K,
                                      int(silu_mul), stream()), "mi_gemm_fp8w_packed")
        return out
    require_gpu(w_packed)
    _bf16(w_packed)
    assert w_packed.is_contiguous()
    check(lib.mi_gemm_bf16_packed(ptr(x), ptr(w_packed), ptr(bias), ptr(out), M, N, K, int(silu_mul), stream()),
          "mi_gemm_bf16_packed")
    return out


def pack_weight_rows4(w, out=None) -> torch.Tensor:
    """[N/4][K/32][4][4][8] copy of a [N, K] weight for gemm_rows4 (same shape/bytes; refreshes `out` in place)."""
    require_gpu(w)
    _bf16(w)
    assert w.dim() == 2 and w.is_contiguous() and w.shape[0] % 4 == 0 and w.shape[1] % 32 == 0
    if out is None or out.shape != w.shape or out.device != w.device:
        out = torch.empty_like(w)
    check(lib.mi_pack_weight_rows4(ptr(w), ptr(out), w.shape[0], w.shape[1], stream()), "mi_pack_weight_rows4")
    return out


def gemm_rows4(x, w_packed4, out=None) -> torch.Tensor:
    """y = bf16(x @ w.T) for the small-N row-parallel projections, complete rows (no split-K partials)."""
    require_gpu(x, w_packed4)
    _bf16(x, w_packed4)
    assert x.is_contiguous() and w_packed4.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = w_packed4.shape[0]
    assert w_packed4.shape[1] == K
    if out is None:
        out = torch.empty((*x.shape[:-1], N), dtype=_BF16, device=x.device)
    check(lib.mi_gemm_bf16_rows4(ptr(x), ptr(w_packed4), ptr(out), M, N, K, stream()), "mi_gemm_bf16_rows4")
    return out


def gemm_packed_splitk(x, w_packed, ksplit: int, out=None) -> torch.Tensor:
    """fp32 partials [ksplit, M, N] of x @ w.T; consume with add_rmsnorm_splitk."""
    require_gpu(x)
    _bf16(x)
    K = x.shape[-1]
    M = x.numel() // K
    N = w_packed.shape[0]
    if out is None:
        out = torch.empty((ksplit, M, N), dtype=torch.float32, device=x.device)
    if isinstance(w_packed, Fp8Weight):
        check(lib.mi_gemm_fp8w_packed_splitk(ptr(x), ptr(w_packed.data), ptr(w_packed.scale), ptr(out), M, N, K,
                                             ksplit, stream()), "mi_gemm_fp8w_packed_splitk")
        return out
    require_gpu(w_packed)
    _bf16(w_packed)
    check(lib.mi_gemm_bf16_packed_splitk(ptr(x), ptr(w_packed), ptr(out), M, N, K, ksplit, stream()),
          "mi_gemm_bf16_packed_splitk")
    return out


def add_rmsnorm_splitk(partials, residual, w, eps: float, out=None, residual_out=None):
    require_gpu(partials, residual, w)
    _bf16(residual, w)
    assert partials.dtype == torch.float32 and partials.is_contiguous() and residual.is_contiguous()
    nsplit, cols = partials.shape[0], residual.shape[-1]
    rows = residual.numel() // cols
    if out is None:
        out = torch.empty_like(residual)
    if residual_out is None:
        residual_out = torch.empty_like(residual)
    check(
        lib.mi_add_rmsnorm_splitk(ptr(partials), nsplit, ptr(residual), ptr(w), ptr(out), ptr(residual_out), rows,
                                  cols, float(eps), stream()),
        "mi_add_rmsnorm_splitk",
    )
    return out, residual_out


# ---- the decode chain in five launches per layer (csrc/gemm_chain5_kernel.hpp) ------------------------------
def chain5_takes(rows: int, hidden: int, ks: tuple) -> bool:
    """Shapes the five-launch chain is built for: a decode-sized batch, consumers (qkv, gate_up; K = hidden) on sixteen
    waves, producers (o_proj, down; K in `ks`) in 64-deep wave slices."""
    return (1 <= rows <= 32 and hidden % 1024 == 0 and hidden // 16 <= 512
            and all(k % 64 == 0 and any((k // 64) % w == 0 and (k // 64) // w in (1, 2, 3, 4, 5, 6, 8) for w in (16, 12, 8, 4))
                    for k in ks))


def gemm_rowstat(x, w_packed, residual, ksplit: int, stamps=None):
    """RowParallelLinear + residual add + RMSNorm statistic in one launch (mi_gemm_bf16_rowstat):
    -> (s fp32 [M, N] = float(bf16(x @ w.T)) + float(residual), residual_out bf16 [M, N], stat fp32 [M, N / 16])."""
    require_gpu(x, w_packed, residual)
    _bf16(x, w_packed, residual)
    assert x.is_contiguous() and residual.is_contiguous() and w_packed.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = w_packed.shape[0]
    assert w_packed.shape[1] == K and residual.numel() == M * N
    s = torch.empty((M, N), dtype=torch.float32, device=x.device)
    res = torch.empty_like(residual)
    stat = torch.empty((M, N // 16), dtype=torch.float32, device=x.device)
    if stamps is None:
        check(lib.mi_gemm_bf16_rowstat(ptr(x), ptr(w_packed), ptr(residual), ptr(res), ptr(s), ptr(stat), M, N, K,
                                       ksplit, stream()), "mi_gemm_bf16_rowstat")
    else:
        check(lib.mi_gemm_bf16_rowstat_ex(ptr(x), ptr(w_packed), ptr(residual), ptr(res), ptr(s), ptr(stat), M, N, K,
                                          ksplit, ptr(stamps), stream()), "mi_gemm_bf16_rowstat_ex")
    return s, res, stat


def gemm_normed(s, stat, norm_w, eps: float, w_packed, silu_mul: bool = False, out=None, stamps=None) -> torch.Tensor:
    """linear(rmsnorm(s)) with the norm applied in the operand load (mi_gemm_bf16_normed); (s, stat) from gemm_rowstat."""
    require_gpu(s, stat, norm_w, w_packed)
    _bf16(norm_w, w_packed)
    assert s.dtype == torch.float32 and stat.dtype == torch.float32 and s.is_contiguous() and stat.is_contiguous()
    M, K = s.shape
    N = w_packed.shape[0]
    assert w_packed.shape[1] == K and stat.shape[0] == M and norm_w.numel() == K
    if out is None:
        out = torch.empty((M, N // 2 if silu_mul else N), dtype=_BF16, device=s.device)
    if stamps is None:
        check(lib.mi_gemm_bf16_normed(ptr(s), ptr(stat), stat.shape[1], ptr(norm_w), float(eps), ptr(w_packed), ptr(out),
                                      M, N, K, int(silu_mul), stream()), "mi_gemm_bf16_normed")
    else:
        check(lib.mi_gemm_bf16_normed_ex(ptr(s), ptr(stat), stat.shape[1], ptr(norm_w), float(eps), ptr(w_packed),
                                         ptr(out), M, N, K, int(silu_mul), ptr(stamps), stream()), "mi_gemm_bf16_normed_ex")
    return out


def norm_from_stat(s, stat, norm_w, eps: float, out=None) -> torch.Tensor:
    """rmsnorm(s) from gemm_rowstat's (s, stat): the model's final norm in the five-launch chain."""
    require_gpu(s, stat, norm_w)
    _bf16(norm_w)
    assert s.dtype == torch.float32 and stat.dtype == torch.float32 and s.is_contiguous() and stat.is_contiguous()
    rows, cols = s.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=_BF16, device=s.device)
    check(lib.mi_norm_from_stat(ptr(s), ptr(stat), stat.shape[1], ptr(norm_w), float(eps), ptr(out), rows, cols,
                                stream()), "mi_norm_from_stat")
    return out


def embedding(ids, w, vocab_start: int = 0, out=None) -> torch.Tensor:
    require_gpu(ids, w)
    _bf16(w)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and w.is_contiguous()
    n, hidden = ids.numel(), w.shape[1]
    if out is None:
        out = torch.empty((n, hidden), dtype=_BF16, device=w.device)
    check(lib.mi_embedding(ptr(ids), ptr(w), ptr(out), n, hidden, vocab_start, w.shape[0], stream()),
          "mi_embedding")
    return out


def embedding_from_prev(ids, src_rows, prev_tokens, w, vocab_start: int = 0, out=None) -> torch.Tensor:
    """embedding() whose row i reads its id from prev_tokens[src_rows[i]] when src_rows[i] >= 0 (the previous
    step's sampled tokens, still on the device), from ids[i] otherwise."""
    require_gpu(ids, src_rows, prev_tokens, w)
    _bf16(w)
    assert ids.dtype == torch.int64 and prev_tokens.dtype == torch.int64 and src_rows.dtype == torch.int32
    assert ids.is_contiguous() and w.is_contiguous() and src_rows.numel() >= ids.numel()
    n, hidden = ids.numel(), w.shape[1]
    if out is None:
        out = torch.empty((n, hidden), dtype=_BF16, device=w.device)
    check(lib.mi_embedding_from_prev(ptr(ids), ptr(src_rows), ptr(prev_tokens), ptr(w), ptr(out), n, hidden,
                                     vocab_start, w.shape[0], stream()), "mi_embedding_from_prev")
    return out


def stage_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst[:] = src[:] between a PINNED host buffer and a device buffer (either direction), as a kernel on the current
    stream (mi_stage_copy): a step's metadata upload / token download without the copy engine between two graphs.
    Both contiguous, the same number of bytes (a multiple of 16)."""
    for t in (dst, src):
        assert t.is_contiguous() and (t.is_cuda or t.is_pinned()), "stage_copy: device or pinned host memory only"
    n = dst.numel() * dst.element_size()
    assert n == src.numel() * src.element_size() and (dst.is_cuda or src.is_cuda)
    check(lib.mi_stage_copy(dst.data_ptr(), src.data_ptr(), n, stream()), "mi_stage_copy")


def gather_last_tokens(x, cu_seqlens_q) -> torch.Tensor:
    require_gpu(x, cu_seqlens_q)
    _bf16(x)
    assert x.is_contiguous() and cu_seqlens_q.dtype == torch.int32
    n_seqs, hidden = cu_seqlens_q.numel() - 1, x.shape[-1]
    out = torch.empty((n_seqs, hidden), dtype=_BF16, device=x.device)
    check(lib.mi_gather_last_tokens(ptr(x), ptr(cu_seqlens_q), ptr(out), n_seqs, hidden, stream()),
          "mi_gather_last_tokens")
    return out


# --------------------------------------------------------------------------- sampling
def argmax(logits, out=None) -> torch.Tensor:
    require_gpu(logits)
    _bf16(logits)
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, vocab = logits.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    check(lib.mi_argmax(ptr(logits), logits.stride(0), ptr(out), rows, vocab, stream()), "mi_argmax")
    return out


def sample(logits, temperatures, seed: int, step: int, out=None) -> torch.Tensor:
    require_gpu(logits, temperatures)
    _bf16(logits)
    assert logits.dim() == 2 and logits.stride(1) == 1
    assert temperatures.dtype == torch.float32 and temperatures.is_contiguous()
    rows, vocab = temperatures.numel(), logits.shape[1]
    assert logits.shape[0] >= rows
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    check(
        lib.mi_sample(ptr(logits), logits.stride(0), ptr(temperatures), ptr(out), rows, vocab,
                      seed & 0xFFFFFFFFFFFFFFFF, step & 0xFFFFFFFFFFFFFFFF, stream()),
        "mi_sample",
    )
    return out


def gemm_packed_pick(x, w_packed, temperatures, rng, out_tokens, logits=None, candidates=None, col_offset: int = 0,
                     pairs_out=None):
    """logits = x @ w.T (as gemm_packed) and, in the same pass, the sampler's token per row
    (as sample(logits, temperatures, seed, step) with {seed, step} = the two uint64 of the device tensor `rng`).
    Returns (logits, tokens).  Tensor parallelism: w_packed is a vocabulary shard starting at `col_offset` and
    `pairs_out` [M, 2] int32 receives this rank's best {key bits, global token} per row instead of tokens (the
    ranks then pick among themselves: XgmiComm.pick_exchange)."""
    require_gpu(x, rng, out_tokens)
    _bf16(x)
    assert x.is_contiguous() and rng.dtype == torch.int64 and rng.numel() == 2 and out_tokens.dtype == torch.int64
    K = x.shape[-1]
    M = x.numel() // K
    N = w_packed.shape[0]
    assert w_packed.shape[1] == K and out_tokens.numel() >= M
    assert temperatures is None or (temperatures.dtype == torch.float32 and temperatures.numel() >= M)
    groups = lib.mi_gemm_pick_groups(M, N, K, int(isinstance(w_packed, Fp8Weight)))
    assert groups > 0
    if logits is None:
        logits = torch.empty((M, N), dtype=_BF16, device=x.device)
    if candidates is None:
        candidates = torch.empty((M, groups, 2), dtype=torch.int32, device=x.device)
    assert candidates.numel() >= groups * M * 2
    if isinstance(w_packed, Fp8Weight):
        assert col_offset == 0, "fp8 head shards are not picked in the graph"
        check(lib.mi_gemm_fp8w_packed_pick(ptr(x), ptr(w_packed.data), ptr(w_packed.scale), ptr(logits), M, N, K,
                                           ptr(temperatures), ptr(rng), ptr(candidates), stream()),
              "mi_gemm_fp8w_packed_pick")
    else:
        require_gpu(w_packed)
        _bf16(w_packed)
        check(lib.mi_gemm_bf16_packed_pick_shard(ptr(x), ptr(w_packed), ptr(logits), M, N, K, ptr(temperatures),
                                                 ptr(rng), ptr(candidates), int(col_offset), stream()),
              "mi_gemm_bf16_packed_pick_shard")
    if pairs_out is not None:
        assert pairs_out.dtype == torch.int32 and pairs_out.is_contiguous() and pairs_out.numel() >= 2 * M
        check(lib.mi_pick_final_pairs(ptr(candidates), groups, M, ptr(pairs_out), stream()), "mi_pick_final_pairs")
        return logits, pairs_out
    check(lib.mi_pick_final(ptr(candidates), groups, M, ptr(out_tokens), stream()), "mi_pick_final")
    return logits, out_tokens


# --------------------------------------------------------------------------- mixture of experts
def pack_expert_weights(w, out=None) -> torch.Tensor:
    """[E, N, K] expert weights -> one fragment-native slab per expert (mi_pack_weight), same shape."""
    require_gpu(w)
    _bf16(w)
    assert w.dim() == 3 and w.is_contiguous() and w.shape[1] % 16 == 0 and w.shape[2] % 32 == 0
    if out is None or out.shape != w.shape or out.device != w.device:
        out = torch.empty_like(w)
    for e in range(w.shape[0]):
        check(lib.mi_pack_weight(ptr(w[e]), ptr(out[e]), w.shape[1], w.shape[2], stream()), "mi_pack_weight")
    return out


def moe_forward(x, router_logits, gate_up_packed, down_packed, top_k: int, all_reduce=None):
    """Qwen3MoeSparseMoeBlock.forward (qwen3_moe.py:150-185) behind the router GEMM: route -> group the pairs by
    expert -> grouped gate_up+SwiGLU -> grouped down (-> all-reduce of the ranks' partial sums) -> combine.
    x [T, H], router_logits [T, E] bf16; returns [T, H] bf16."""
    require_gpu(x, router_logits, gate_up_packed, down_packed)
    _bf16(x, router_logits, gate_up_packed, down_packed)
    assert x.is_contiguous() and router_logits.is_contiguous()
    T, H = x.shape
    E, two_i, _ = gate_up_packed.shape
    inter = two_i // 2
    assert router_logits.shape == (T, E) and down_packed.shape == (E, H, inter)
    dev = x.device
    ids = torch.empty((T, top_k), dtype=torch.int32, device=dev)
    w = torch.empty((T, top_k), dtype=_BF16, device=dev)
    check(lib.mi_moe_route(ptr(router_logits), T, E, top_k, ptr(ids), ptr(w), stream()), "mi_moe_route")
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    pair_token = torch.empty(T * top_k, dtype=torch.int32, device=dev)
    pair_index = torch.empty(T * top_k, dtype=torch.int32, device=dev)
    check(lib.mi_moe_sort(ptr(ids), T, top_k, E, ptr(offsets), ptr(pair_token), ptr(pair_index), stream()),
          "mi_moe_sort")
    act = torch.empty((T * top_k, inter), dtype=_BF16, device=dev)
    check(lib.mi_moe_gate_up(ptr(x), ptr(gate_up_packed), ptr(offsets), ptr(pair_token), ptr(act), E, H, inter,
                             stream()), "mi_moe_gate_up")
    y = torch.empty((T * top_k, H), dtype=_BF16, device=dev)  # row t * top_k + j: the same on every TP rank
    check(lib.mi_moe_down(ptr(act), ptr(down_packed), ptr(offsets), ptr(pair_index), ptr(y), E, H, inter, stream()),
          "mi_moe_down")
    if all_reduce is not None:
        y = all_reduce(y)
    out = torch.empty((T, H), dtype=_BF16, device=dev)
    check(lib.mi_moe_combine(ptr(y), ptr(w), ptr(out), T, top_k, H, stream()), "mi_moe_combine")
    return out, ids, w


# ---- instrumented forms of the decode chain's launches (tools/chain_timeline.py) --------------------------------
def gemm_packed_stamped(x, w_packed, stamps, silu_mul: bool = False, ksplit: int = 0, out=None):
    """mi_gemm_bf16_packed_ex: gemm_packed (ksplit 0) / gemm_packed_splitk (ksplit > 0) with every wave's clock stamps
    in stamps[workgroups][waves][8] (int64).  The decode chain's own configurations only."""
    require_gpu(x, w_packed, stamps)
    _bf16(x, w_packed)
    assert stamps.dtype == torch.int64 and stamps.is_contiguous()
    M, K = x.shape
    N = w_packed.shape[0]
    if ksplit:
        if out is None:
            out = torch.empty((ksplit, M, N), dtype=torch.float32, device=x.device)
        y, part = None, out
    else:
        if out is None:
            out = torch.empty((M, N // 2 if silu_mul else N), dtype=_BF16, device=x.device)
        y, part = out, None
    check(lib.mi_gemm_bf16_packed_ex(ptr(x), ptr(w_packed), ptr(y), ptr(part), M, N, K, int(silu_mul), ksplit,
                                     ptr(stamps), stream()), "mi_gemm_bf16_packed_ex")
    return out


def add_rmsnorm_splitk_stamped(partials, residual, w, eps: float, stamps, out=None, residual_out=None):
    """mi_add_rmsnorm_splitk_ex: add_rmsnorm_splitk with every wave's clock stamps in stamps[rows][4][8] (int64)."""
    require_gpu(partials, residual, w, stamps)
    _bf16(residual, w)
    assert stamps.dtype == torch.int64 and stamps.is_contiguous()
    nsplit, cols = partials.shape[0], residual.shape[-1]
    rows = residual.numel() // cols
    if out is None:
        out = torch.empty_like(residual)
    if residual_out is None:
        residual_out = torch.empty_like(residual)
    check(lib.mi_add_rmsnorm_splitk_ex(ptr(partials), nsplit, ptr(residual), ptr(w), ptr(out), ptr(residual_out), rows,
                                       cols, float(eps), ptr(stamps), stream()), "mi_add_rmsnorm_splitk_ex")
    return out, residual_out
