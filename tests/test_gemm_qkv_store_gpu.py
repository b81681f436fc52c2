"""The packed qkv projection with k-norm + RoPE + the K / V cache store in its epilogue (mi_gemm_bf16_qkv_store,
VERDICT r03 item 1b; reference: qwen3.py:79-90 split / q_norm, k_norm / rotary, attention.py:55-58 store_kvcache)
against the two launches it replaces - mi_gemm_bf16 + mi_qknorm_rope_store(q_out = NULL) - bit for bit: the q columns
of the qkv rows and every byte of both caches (sentinel-filled: nothing else may be touched).  The two-launch path
itself is held to the CPU oracle by test_kernels_gpu.py / test_gemm_tile_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


def _slots(seq_lens, block_size, n_blocks, g, skip_prefix=()):
    """flat slots of packed sequences on scrambled block tables; skip_prefix[i] leading tokens of sequence i get -1
    (cached prefix recomputed only for its logits)"""
    perm = torch.randperm(n_blocks, generator=g).tolist()
    out, nxt = [], 0
    for i, n in enumerate(seq_lens):
        nb = -(-n // block_size)
        table = perm[nxt:nxt + nb]
        nxt += nb
        pos = torch.arange(n)
        s = torch.tensor(table)[pos // block_size] * block_size + pos % block_size
        k = skip_prefix[i] if i < len(skip_prefix) else 0
        s[:k] = -1
        out.append(s)
    return torch.cat(out).to(torch.int32)


@pytest.mark.parametrize("case", [
    # (n_q, n_kv, K, seq_lens, block_size, bias, k_norm)
    dict(n_q=16, n_kv=8, K=1024, seqs=[1024] * 4, bs=16, bias=False, norm=True),        # the bench's shape, regular
    dict(n_q=16, n_kv=8, K=256, seqs=[1000, 37, 519, 1, 1711, 16, 900, 35], bs=16, bias=False, norm=True, skip=(0, 0, 48, 0, 160)),
    dict(n_q=16, n_kv=8, K=192, seqs=[1300, 1301, 1302, 700], bs=48, bias=True, norm=False),  # Qwen2-style, block 48
    dict(n_q=8, n_kv=1, K=320, seqs=[4000, 4000, 4000, 1111], bs=256, bias=False, norm=True),  # a TP-8 shard: K | V share a tile
    dict(n_q=14, n_kv=2, K=128, seqs=[2500, 2501, 2502], bs=32, bias=True, norm=True, skip=(32, 0, 64)),
])
def test_qkv_gemm_store_matches_two_launches(ops, case):
    n_q, n_kv, K, bs = case["n_q"], case["n_kv"], case["K"], case["bs"]
    N = (n_q + 2 * n_kv) * 128
    seqs = case["seqs"]
    M = sum(seqs)
    assert ops.qkv_store_takes(M, N, 128, bs), (M, N)
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV) if case["bias"] else None
    kw = (1.0 + 0.1 * torch.randn(128, generator=g)).bfloat16().to(DEV) if case["norm"] else None
    n_blocks = sum(-(-n // bs) for n in seqs) + 3
    slots = _slots(seqs, bs, n_blocks, g, case.get("skip", ())).to(DEV)
    positions = torch.cat([torch.arange(n) for n in seqs]).to(torch.int64).to(DEV)
    max_pos = max(seqs)
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(max_pos).float()[:, None] * inv[None, :]
    cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous().to(DEV)
    shape = ops.kv_cache_shape(n_blocks, n_kv, bs)
    sentinel = 0x7B7B

    def caches():
        return (torch.full(shape, sentinel, dtype=torch.int16, device=DEV).view(torch.bfloat16),
                torch.full(shape, sentinel, dtype=torch.int16, device=DEV).view(torch.bfloat16))

    # the two launches
    k1, v1 = caches()
    qkv1 = ops.gemm_tile(x, w, b)
    ops.qknorm_rope_store(qkv1, kw, kw, 1e-6, positions, cos_sin, k1, v1, slots, n_q, n_kv, bs, store_q=False)
    # the fused launch
    k2, v2 = caches()
    qkv2 = torch.full((M, N), sentinel, dtype=torch.int16, device=DEV).view(torch.bfloat16)
    ops.gemm_qkv_store(x, w, b, kw, 1e-6, positions, cos_sin, k2, v2, slots, n_q, n_kv, bs, out=qkv2)
    torch.cuda.synchronize()
    qcols = n_q * 128
    assert torch.equal(qkv1[:, :qcols].view(torch.int16), qkv2[:, :qcols].view(torch.int16))
    assert bool((qkv2[:, qcols:].view(torch.int16) == sentinel).all()), "the k / v columns of the qkv rows must stay untouched"
    dk = (k1.view(torch.int16) != k2.view(torch.int16))
    dv = (v1.view(torch.int16) != v2.view(torch.int16))
    assert not bool(dk.any()), f"K cache differs in {int(dk.sum())} elements"
    assert not bool(dv.any()), f"V cache differs in {int(dv.sum())} elements"
    # and the caches were written at all (stored tokens x kv heads x 128 elements each)
    stored = int((slots >= 0).sum())
    assert int((k2.view(torch.int16) != sentinel).sum()) >= stored * n_kv * 120


def test_qkv_gemm_store_rejects_what_it_cannot_do(ops):
    from nanovllm._C import MiError

    assert not ops.qkv_store_takes(1024, 4096, 128, 16)      # too few tiles: the 128-tile kernel's shapes
    assert not ops.qkv_store_takes(16384, 4096, 64, 16)      # head_dim 64
    assert not ops.qkv_store_takes(16384, 2176, 128, 16)     # N not in whole feature tiles
    x = torch.zeros(1024, 256, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(4096, 256, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros(ops.kv_cache_shape(80, 8, 16), dtype=torch.bfloat16, device=DEV)
    pos = torch.zeros(1024, dtype=torch.int64, device=DEV)
    cs = torch.zeros(16, 128, device=DEV)
    sl = torch.zeros(1024, dtype=torch.int32, device=DEV)
    with pytest.raises(MiError):
        ops.gemm_qkv_store(x, w, None, None, 1e-6, pos, cs, k, k.clone(), sl, 16, 8, 16)
