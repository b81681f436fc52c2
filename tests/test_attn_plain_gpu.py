"""GPU parity of the plain-layout attention family (csrc/attn_plain.hip: head_dim 64, GQA groups that are not a power
of two - Qwen2-0.5B 14/2 x 64, Llama-3.2-1B 32/8 x 64, Qwen2.5-7B 28/4 x 128) against the CPU oracle, through the C
ABI: cache writes and RoPE bit-exact, attention |out - oracle_fp32| <= 2^-8 |out| + 1e-4 (the bound of the
fragment-native kernels), and tiny models of both geometries end to end against the oracle model."""
import math

import pytest
import torch

import oracle
from model_configs import TINY_LLAMA_HD64, TINY_LLAMA_HD64_G3, TINY_QWEN2_HD64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GEOMETRIES = [(14, 2, 64), (32, 8, 64), (28, 4, 128), (4, 4, 64), (8, 1, 64), (6, 2, 128)]


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


def _plain(cache_logical):
    """oracle layout [blocks, block, kv heads, D] -> device layout [blocks, kv heads, block, D]"""
    return cache_logical.permute(0, 2, 1, 3).contiguous()


def _case(gen, hq, hkv, d, block_size, ctx_lens, extra_blocks=3):
    need = sum((n + block_size - 1) // block_size for n in ctx_lens)
    nblk = need + extra_blocks
    kc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    vc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    perm = torch.randperm(nblk, generator=gen).tolist()
    tables = [[perm.pop() for _ in range((n + block_size - 1) // block_size)] for n in ctx_lens]
    width = max(1, max(len(t) for t in tables)) + 2
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    return kc, vc, bt


@pytest.mark.parametrize("hq,hkv,d", [(14, 2, 64), (4, 1, 64), (28, 4, 128)])
def test_rope_and_cache_writes_plain(ops, hq, hkv, d):
    gen = torch.Generator().manual_seed(hq + d)
    T, bs, nblk = 37, 16, 9
    table = oracle.build_cos_sin_cache(d, 512, 1e4)
    pos = torch.randint(0, 512, (T,), generator=gen)
    qkv = torch.randn(T, (hq + 2 * hkv) * d, generator=gen).bfloat16()
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    q3, k3, v3 = q.view(T, hq, d), k.view(T, hkv, d), v.view(T, hkv, d)
    qd, kd, vd = (t.to(DEV) for t in qkv.to(DEV).split([hq * d, hkv * d, hkv * d], dim=-1))  # strided views of one row
    if d != 128:  # head_dim 128 takes mi_rope (tested in test_kernels_gpu.py); here: the plain kernel
        qr, kr = ops.rope(pos.to(DEV), qd.view(T, hq, d), kd.view(T, hkv, d), table.to(DEV), hq, hkv)
        assert torch.equal(qr.cpu().view(torch.int16), oracle.apply_rope(pos, q3, table).view(torch.int16))
        assert torch.equal(kr.cpu().view(torch.int16), oracle.apply_rope(pos, k3, table).view(torch.int16))
    # flat slots (prefill; -1 = skip) and [block, offset] pairs (decode)
    perm = torch.randperm(nblk * bs, generator=gen).to(torch.int32)
    slots, spare = perm[:T].clone(), int(perm[T])
    slots[5] = -1
    kc = torch.zeros(ops.kv_cache_shape_plain(nblk, hkv, bs, d), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.kv_store_plain(kd.view(T, hkv, d), vd.view(T, hkv, d), kc, vc, slots.to(DEV), hkv, bs)
    want_k = torch.zeros(nblk, bs, hkv, d).bfloat16()
    want_v = torch.zeros_like(want_k)
    for t in range(T):
        if slots[t] >= 0:
            want_k[slots[t] // bs, slots[t] % bs] = k3[t]
            want_v[slots[t] // bs, slots[t] % bs] = v3[t]
    assert torch.equal(kc.cpu(), _plain(want_k)) and torch.equal(vc.cpu(), _plain(want_v))
    kc.zero_()
    vc.zero_()
    slots[5] = spare  # the pair form has no skip value: the row goes to a slot of its own
    s2 = torch.stack([slots // bs, slots % bs], 1).to(torch.int32).contiguous()
    ops.kv_store_plain(kd.view(T, hkv, d), vd.view(T, hkv, d), kc, vc, s2.to(DEV), hkv, bs)
    want_k[spare // bs, spare % bs], want_v[spare // bs, spare % bs] = k3[5], v3[5]
    assert torch.equal(kc.cpu(), _plain(want_k)) and torch.equal(vc.cpu(), _plain(want_v))


@pytest.mark.parametrize("hq,hkv,d", GEOMETRIES)
@pytest.mark.parametrize("block_size", [16, 8])
def test_decode_attention_plain_vs_oracle(ops, hq, hkv, d, block_size):
    gen = torch.Generator().manual_seed(hq * 3 + d + block_size)
    ctx_lens = [0, 1, 17, 300, 1025, 64, 31, 5]  # 0: a graph-padded row (zeros)
    kc, vc, bt = _case(gen, hq, hkv, d, block_size, ctx_lens)
    q = torch.randn(len(ctx_lens), hq, d, generator=gen).bfloat16()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc, vc, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode_plain(q.to(DEV), _plain(kc).to(DEV), _plain(vc).to(DEV), bt.to(DEV), ctx.to(DEV), hq, hkv,
                                      block_size, 1.0 / math.sqrt(d)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 1e-4).all()), err.max().item()
    assert float(out[0].float().abs().sum()) == 0.0


def test_decode_attention_plain_large_batch_and_sharp_softmax(ops):
    """enough (sequence, kv head) pairs that the context is not split; queries scaled so that one key dominates"""
    hq, hkv, d, bs = 14, 2, 64, 16
    gen = torch.Generator().manual_seed(5)
    ctx_lens = [int(v) for v in torch.randint(1, 400, (300,), generator=gen)]
    kc, vc, bt = _case(gen, hq, hkv, d, bs, ctx_lens)
    q = (torch.randn(len(ctx_lens), hq, d, generator=gen) * 6).bfloat16()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc, vc, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode_plain(q.to(DEV), _plain(kc).to(DEV), _plain(vc).to(DEV), bt.to(DEV), ctx.to(DEV), hq, hkv,
                                      bs, 1.0 / math.sqrt(d)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 1e-4).all()), err.max().item()


@pytest.mark.parametrize("hq,hkv,d", GEOMETRIES)
def test_prefill_attention_plain_vs_oracle(ops, hq, hkv, d):
    """ragged query lengths, one sequence behind a 64-token cached prefix (queries start at position 64)"""
    gen = torch.Generator().manual_seed(hq + d)
    bs = 16
    q_lens = [1, 7, 16, 33, 129, 260, 64]
    kv_lens = [1, 7, 16, 33, 129 + 64, 260, 64]
    T = sum(q_lens)
    kc, vc, bt = _case(gen, hq, hkv, d, bs, kv_lens)
    q = torch.randn(T, hq, d, generator=gen).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32)
    kvl = torch.tensor(kv_lens, dtype=torch.int32)
    want = oracle.paged_attention_prefill(q, kc, vc, bt, cu, kvl, keep_fp32=True)
    out = ops.paged_attn_prefill_plain(q.to(DEV), _plain(kc).to(DEV), _plain(vc).to(DEV), bt.to(DEV), cu.to(DEV),
                                       kvl.to(DEV), max(q_lens), hq, hkv, bs, 1.0 / math.sqrt(d)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 1e-4).all()), err.max().item()


def test_plain_attention_rejects_what_it_cannot_do(ops):
    from nanovllm._C import MiError

    q = torch.zeros(2, 9 * 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(4, 1, 16, 64, dtype=torch.bfloat16, device=DEV)
    bt = torch.zeros(2, 2, dtype=torch.int32, device=DEV)
    ctx = torch.ones(2, dtype=torch.int32, device=DEV)
    with pytest.raises(MiError):  # nine query heads per kv head
        ops.paged_attn_decode_plain(q, kc, kc, bt, ctx, 9, 1, 16, 0.125)
    # (since round 4 the README models' geometries run on the fragment-native kernels: tests/test_attn_hd64_gpu.py)
    assert ops.attention_is_plain(6, 2, 64) and ops.attention_is_plain(12, 4, 128) and not ops.attention_is_plain(16, 8, 128)
    assert not ops.attention_plain_supported(16, 1, 64) and not ops.attention_plain_supported(8, 1, 96)


@pytest.mark.parametrize("cfg_name", ["qwen2_hd64_gqa7", "llama_hd64_gqa4", "llama_hd64_gqa3"])
@pytest.mark.parametrize("enforce_eager", [True, False])
def test_engine_with_small_head_geometries_matches_oracle(cfg_name, enforce_eager):
    """tiny models with Qwen2-0.5B's and Llama-3.2-1B's head geometry (fragment-native MFMA kernels on 2 KiB tiles since
    round 4: RoPE -> store -> attention, one launch each) and with three query heads per kv head (the plain-layout
    family) through the whole engine (eager prefill, eager and hipGraph decode incl. the in-graph sampler) against the
    oracle model on the same weights"""
    from test_engine_gpu import _engine_vs_oracle

    cfg = {"qwen2_hd64_gqa7": TINY_QWEN2_HD64, "llama_hd64_gqa4": TINY_LLAMA_HD64, "llama_hd64_gqa3": TINY_LLAMA_HD64_G3}[cfg_name]
    from nanovllm import ops

    hq, hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    assert ops.attention_is_plain(hq, hkv, 64) == (cfg_name == "llama_hd64_gqa3")
    _engine_vs_oracle(cfg, lens=[5, 33, 64, 17, 100], enforce_eager=enforce_eager, seed=11, tol=4e-2, max_tokens=6)
