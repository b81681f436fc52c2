"""GPU parity of the fragment-native MFMA attention kernels at head_dim 64 and with 7 query heads per kv head
(round 4: Llama-3.2-1B = 32 / 8 x 64, Qwen2-0.5B = 14 / 2 x 64, Qwen2.5-7B = 28 / 4 x 128 - the reference's README
models, which ran on the plain-layout family until round 3): KV scatter / gather bit-exact, decode and prefill
attention against the fp32-softmax oracle, all through the C ABI."""
import math

import pytest
import torch

import oracle
from kv_layout import to_fragment, to_logical

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (query heads, kv heads, head_dim)
GEOMETRIES = [(32, 8, 64), (14, 2, 64), (8, 8, 64), (16, 1, 64), (28, 4, 128), (7, 1, 128)]


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


def _case(gen, hkv, d, block_size, ctx_lens, extra_blocks=3):
    need = sum((n + block_size - 1) // block_size for n in ctx_lens)
    nblk = need + extra_blocks
    kc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    vc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    perm = torch.randperm(nblk, generator=gen).tolist()
    tables = [[perm.pop() for _ in range((n + block_size - 1) // block_size)] for n in ctx_lens]
    width = max(1, max(len(t) for t in tables)) + 2
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    return kc, vc, bt


@pytest.mark.parametrize("hkv", [1, 8])
@pytest.mark.parametrize("block_size", [16, 48, 256])
def test_kv_scatter_and_gather_head_dim_64(ops, hkv, block_size):
    """mi_reshape_and_cache (flat slots, incl. the whole-tile path of >= 64 tokens and skipped rows),
    mi_scatter_update_kv ([block, offset] pairs) and mi_kv_cache_gather on 2 KiB tiles: exact copies, and the device
    layout is the host model's (tests/kv_layout.py)."""
    gen = torch.Generator().manual_seed(hkv + block_size)
    d, nblk = 64, max(9, (200 + 48) // block_size + 2)
    for T in (5, 200):
        k = torch.randn(T, hkv, d, generator=gen).bfloat16()
        v = torch.randn(T, hkv, d, generator=gen).bfloat16()
        start = 16 * int(torch.randint(0, 3, (1,), generator=gen))
        slots = (torch.arange(T) + start).to(torch.int32)  # consecutive slots from a tile boundary: whole tiles
        if T > 64:
            slots[70] = -1  # a skipped row inside a tile: that tile takes the element path
        kc = torch.zeros(ops.kv_cache_shape(nblk, hkv, block_size, d), dtype=torch.bfloat16, device=DEV)
        vc = torch.zeros_like(kc)
        ops.reshape_and_cache(k.to(DEV), v.to(DEV), kc, vc, slots.to(DEV), hkv, block_size)
        want_k = torch.zeros(nblk, block_size, hkv, d).bfloat16()
        want_v = torch.zeros_like(want_k)
        for t in range(T):
            if slots[t] >= 0:
                want_k[slots[t] // block_size, slots[t] % block_size] = k[t]
                want_v[slots[t] // block_size, slots[t] % block_size] = v[t]
        assert torch.equal(to_logical(kc.cpu(), block_size, False).view(torch.int16), want_k.view(torch.int16))
        assert torch.equal(to_logical(vc.cpu(), block_size, True).view(torch.int16), want_v.view(torch.int16))
        assert torch.equal(kc.cpu().view(torch.int16), to_fragment(want_k, False).view(torch.int16))
        live = slots.clamp(min=0)
        got = ops.kv_cache_gather(kc, False, live.to(DEV), hkv, block_size).cpu().view(T, hkv, d)
        assert torch.equal(got[slots >= 0].view(torch.int16), k[slots >= 0].view(torch.int16))
        got = ops.kv_cache_gather(vc, True, live.to(DEV), hkv, block_size).cpu().view(T, hkv, d)
        assert torch.equal(got[slots >= 0].view(torch.int16), v[slots >= 0].view(torch.int16))
    # decode form
    kc.zero_()
    vc.zero_()
    pairs = torch.tensor([[3, 5], [0, block_size - 1], [8, 0]], dtype=torch.int32)
    k3, v3 = torch.randn(3, hkv, d, generator=gen).bfloat16(), torch.randn(3, hkv, d, generator=gen).bfloat16()
    ops.scatter_update_kv(k3.to(DEV), v3.to(DEV), kc, vc, pairs.to(DEV), hkv, block_size)
    kl, vl = to_logical(kc.cpu(), block_size, False), to_logical(vc.cpu(), block_size, True)
    for i, (b, o) in enumerate(pairs.tolist()):
        assert torch.equal(kl[b, o].view(torch.int16), k3[i].view(torch.int16))
        assert torch.equal(vl[b, o].view(torch.int16), v3[i].view(torch.int16))
    assert int((kl.view(torch.int16) != 0).sum()) <= 3 * hkv * d


@pytest.mark.parametrize("hq,hkv,d", GEOMETRIES)
@pytest.mark.parametrize("block_size", [16, 64])
def test_decode_attention_vs_oracle(ops, hq, hkv, d, block_size):
    gen = torch.Generator().manual_seed(hq * 3 + d + block_size)
    ctx_lens = [0, 1, 15, 16, 17, 33, 300, 1025, 64, 31, 5, 2049]  # 0: a graph-padded row (zeros)
    kc, vc, bt = _case(gen, hkv, d, block_size, ctx_lens)
    q = torch.randn(len(ctx_lens), hq, d, generator=gen).bfloat16()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc, vc, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode(q.to(DEV), to_fragment(kc, False).to(DEV), to_fragment(vc, True).to(DEV), bt.to(DEV),
                                ctx.to(DEV), hq, hkv, block_size, 1.0 / math.sqrt(d)).cpu()
    assert out.shape == (len(ctx_lens), hq * d)
    err = (out.float() - want.reshape(len(ctx_lens), -1)).abs()
    assert bool((err <= want.reshape(len(ctx_lens), -1).abs() * 2 ** -8 + 2e-4).all()), err.max().item()
    assert float(out[0].float().abs().sum()) == 0.0


@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 64), (14, 2, 64), (28, 4, 128)])
def test_decode_attention_large_batch_not_split(ops, hq, hkv, d):
    """enough (sequence, kv head) pairs that no context is split over workgroups; sharp softmax (one key dominates)"""
    gen = torch.Generator().manual_seed(5 + hq)
    ctx_lens = [int(v) for v in torch.randint(1, 400, (160,), generator=gen)]
    kc, vc, bt = _case(gen, hkv, d, 16, ctx_lens)
    q = (torch.randn(len(ctx_lens), hq, d, generator=gen) * 6).bfloat16()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc, vc, bt, ctx, keep_fp32=True).reshape(len(ctx_lens), -1)
    out = ops.paged_attn_decode(q.to(DEV), to_fragment(kc, False).to(DEV), to_fragment(vc, True).to(DEV), bt.to(DEV),
                                ctx.to(DEV), hq, hkv, 16, 1.0 / math.sqrt(d)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 2e-4).all()), err.max().item()


@pytest.mark.parametrize("hq,hkv,d", GEOMETRIES)
@pytest.mark.parametrize("block_size", [16, 48])
def test_prefill_attention_vs_oracle(ops, hq, hkv, d, block_size):
    """ragged query lengths, one sequence behind a 64-token cached prefix (queries start at position 64).  P is one
    bf16 per key (the default): bound = the output's rounding + 2^-8 x the same attention over |V|."""
    gen = torch.Generator().manual_seed(hq + d + block_size)
    q_lens = [1, 7, 16, 33, 129, 260, 64]
    kv_lens = [1, 7, 16, 33, 129 + 64, 260, 64]
    T = sum(q_lens)
    kc, vc, bt = _case(gen, hkv, d, block_size, kv_lens)
    q = torch.randn(T, hq, d, generator=gen).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32)
    kvl = torch.tensor(kv_lens, dtype=torch.int32)
    want = oracle.paged_attention_prefill(q, kc, vc, bt, cu, kvl, keep_fp32=True).reshape(T, -1)
    want_absv = oracle.paged_attention_prefill(q, kc, vc.abs(), bt, cu, kvl, keep_fp32=True).reshape(T, -1)
    out = ops.paged_attn_prefill(q.to(DEV), to_fragment(kc, False).to(DEV), to_fragment(vc, True).to(DEV), bt.to(DEV),
                                 cu.to(DEV), kvl.to(DEV), max(q_lens), hq, hkv, block_size, 1.0 / math.sqrt(d)).cpu()
    assert out.shape == (T, hq * d)
    err = (out.float() - want).abs()
    tol = want.abs() * 2 ** -8 + want_absv * 2 ** -8 + 1e-4
    assert bool((err <= tol).all()), (err - tol).max().item()


def test_geometry_dispatch(ops):
    assert not ops.attention_is_plain(32, 8, 64) and not ops.attention_is_plain(14, 2, 64)
    assert not ops.attention_is_plain(28, 4, 128) and not ops.attention_is_plain(16, 8, 128)
    assert ops.attention_is_plain(12, 4, 64) and ops.attention_is_plain(10, 2, 128) and ops.attention_is_plain(8, 2, 96)
    assert ops.attention_is_fusable(16, 8, 128) and not ops.attention_is_fusable(32, 8, 64)
    assert not ops.attention_is_fusable(28, 4, 128)
    assert ops.kv_cache_shape(10, 8, 32, 64) == (10, 8, 2, 1024) and ops.kv_cache_shape(10, 8, 32) == (10, 8, 2, 2048)
