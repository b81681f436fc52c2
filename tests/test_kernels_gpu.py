"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle
and the reference-generated golden vectors.  Integer / copy work is bit-exact;
floating-point tolerances are stated per test."""
import math

import numpy as np
import pytest
import torch

import oracle
from conftest import bf16_from_bits as bf
from kv_layout import to_fragment, to_logical

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


@pytest.fixture(params=[1, 0], ids=["pipe8", "waves16"])
def attn_geometry(request):
    """both geometries of the decode attention kernel: 8 waves with two chunks in flight (default) and the
    16-wave one-chunk form (tuning knob MI_TUNE_ATTN_PIPE; the library never reads the environment)"""
    from nanovllm import _C

    _C.set_tuning(_C.TUNE_ATTN_PIPE, request.param)
    yield request.param
    _C.set_tuning(_C.TUNE_ATTN_PIPE, 1)


@pytest.fixture(params=[0, 1], ids=["p_bf16", "p_hi_lo"])
def prefill_p(request):
    """both precisions of the prefill attention's probabilities (MI_TUNE_PREFILL_P_SPLIT): one bf16 per key (the
    default, the reference's own CPU statement keeps P in bf16: attention_torch_native.py:127,188) and bf16 hi + lo.
    Yields the weight of the P-rounding term of the error bound against the fp32-softmax oracle (see _prefill_bound):
    2^-8 (the unit roundoff of bf16) for P as one bf16, 0 for hi + lo."""
    from nanovllm import _C

    _C.set_tuning(_C.TUNE_PREFILL_P_SPLIT, request.param)
    yield 0.0 if request.param else 2 ** -8
    _C.set_tuning(_C.TUNE_PREFILL_P_SPLIT, 0)


def _prefill_bound(want, want_absv, p_term):
    """|out - oracle| of the prefill attention: the output's one rounding to bf16 (unit roundoff 2^-8: |out| 2^-8) + -
    with P carried as ONE bf16 per key - the first-order effect of rounding every probability by at most 2^-8
    relatively: |sum_i p_i d_i v_i| / L <= 2^-8 sum_i p_i |v_i| / L, i.e. 2^-8 times the same attention over |V|
    (`want_absv`, computed by the oracle).  No slack beyond that but 1e-4 absolute."""
    return want.abs() * 2 ** -8 + want_absv * p_term + 1e-4


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """distance in bf16 ulps between two bf16 tensors (monotone integer mapping)"""
    def key(t):
        i = t.cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a) - key(b)).abs()


def assert_bf16_close(got, want, max_ulp=1, max_frac=1e-3, atol=0.0):
    """Equal up to `max_ulp` bf16 ulps on at most `max_frac` of the elements (fp32
    summation order / rsqrt / exp last-bit effects before a bf16 rounding).  `atol`
    covers outputs that cancel to ~0, where the fp32 accumulation noise of the dot
    product (K * 2^-24 * |x||w|) is many ulps of a tiny result."""
    d = ulp_diff(got, want)
    bad = d > max_ulp
    if atol > 0:
        bad &= (got.cpu().float() - want.float()).abs() > atol
    assert not bool(bad.any()), f"max ulp diff {int(d[bad].max())} on {int(bad.sum())} elements"
    frac = float((d > 0).float().mean())
    assert frac <= max_frac, f"{frac:.2e} of elements differ"


# --------------------------------------------------------------------------- norms / rope / silu vs golden
def test_rmsnorm_golden(ops, golden_layers):
    g = golden_layers
    for tag in ("h1024", "head128", "h256"):
        x, w, r = bf(g[f"rms_{tag}_x"]).to(DEV), bf(g[f"rms_{tag}_w"]).to(DEV), bf(g[f"rms_{tag}_r"]).to(DEV)
        assert_bf16_close(ops.rmsnorm(x, w, 1e-6), bf(g[f"rms_{tag}_y"]))
        y, r2 = ops.add_rmsnorm(x, r, w, 1e-6)
        assert_bf16_close(y, bf(g[f"rms_{tag}_addy"]))
        assert torch.equal(r2.cpu().view(torch.int16), bf(g[f"rms_{tag}_addr"]).view(torch.int16))  # pure add+round


@pytest.mark.parametrize("rows,cols", [(1, 1024), (32, 1024), (33, 5120), (257, 2048), (7, 64), (512, 128), (3, 8192)])
def test_rmsnorm_random(ops, rows, cols):
    g = torch.Generator().manual_seed(rows * 131 + cols)
    x = (torch.randn(rows, cols, generator=g) * 2).bfloat16()
    r = torch.randn(rows, cols, generator=g).bfloat16()
    w = (1 + 0.2 * torch.randn(cols, generator=g)).bfloat16()
    assert_bf16_close(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6), oracle.rms_norm(x, w, 1e-6))
    y, r2 = ops.add_rmsnorm(x.to(DEV), r.to(DEV), w.to(DEV), 1e-6)
    yo, ro = oracle.add_rms_norm(x, r, w, 1e-6)
    assert_bf16_close(y, yo)
    assert torch.equal(r2.cpu().view(torch.int16), ro.view(torch.int16))
    # in-place aliasing (what the runner does)
    xd, rd = x.to(DEV), r.to(DEV)
    ops.add_rmsnorm(xd, rd, w.to(DEV), 1e-6, out=xd, residual_out=rd)
    assert_bf16_close(xd, yo)
    assert torch.equal(rd.cpu().view(torch.int16), ro.view(torch.int16))


@pytest.mark.parametrize("rows,cols", [(32, 4096), (64, 5120), (16, 2048), (1, 8192), (5, 4104), (32, 1032), (64, 8192)])
def test_add_rmsnorm_few_wide_rows(ops, rows, cols):
    """Round 5: up to 64 rows of 1032 ... 8192 columns (decode steps of hidden 2048 ... 8192 models) run four or eight
    waves per row (add_rmsnorm_splitk_rows_kernel<0 | NS, 4 | 8>), for finished bf16 rows and for split-K partials:
    the reference's arithmetic (layernorm.py:27-38: variance of the un-rounded sum), residual bit-exact."""
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) * 2).bfloat16()
    r = torch.randn(rows, cols, generator=g).bfloat16()
    w = (1 + 0.2 * torch.randn(cols, generator=g)).bfloat16()
    yo, ro = oracle.add_rms_norm(x, r, w, 1e-6)
    y, r2 = ops.add_rmsnorm(x.to(DEV), r.to(DEV), w.to(DEV), 1e-6)
    assert_bf16_close(y, yo)
    assert torch.equal(r2.cpu().view(torch.int16), ro.view(torch.int16))
    for ns in (2, 4):  # the same rows as fp32 partials whose sum rounds to x exactly (x + tiny, -tiny, 0 ...)
        parts = torch.zeros(ns, rows, cols)
        parts[0] = x.float() + 2.0 ** -20
        parts[1] = -(2.0 ** -20)
        y2, r3 = ops.add_rmsnorm_splitk(parts.to(DEV), r.to(DEV), w.to(DEV), 1e-6)
        assert_bf16_close(y2, yo)
        assert torch.equal(r3.cpu().view(torch.int16), ro.view(torch.int16))


def test_rmsnorm_strided_heads(ops):
    """per-head q/k norm on strided views of a packed qkv row (qwen3.py:79-85)"""
    g = torch.Generator().manual_seed(5)
    T, hq, hkv = 9, 16, 8
    qkv = torch.randn(T, (hq + 2 * hkv) * 128, generator=g).bfloat16()
    w = (1 + 0.2 * torch.randn(128, generator=g)).bfloat16()
    qd = qkv.to(DEV)
    q_view = qd[:, : hq * 128].view(T, hq, 128)
    k_view = qd[:, hq * 128 : (hq + hkv) * 128].view(T, hkv, 128)
    assert_bf16_close(ops.rmsnorm(q_view, w.to(DEV), 1e-6), oracle.rms_norm(qkv[:, : hq * 128].view(T, hq, 128), w, 1e-6))
    assert_bf16_close(ops.rmsnorm(k_view, w.to(DEV), 1e-6),
                      oracle.rms_norm(qkv[:, hq * 128 : (hq + hkv) * 128].view(T, hkv, 128), w, 1e-6))


def test_rope_golden_bit_exact(ops, golden_layers):
    g = golden_layers
    pos = torch.from_numpy(g["rope_pos"])
    table = oracle.build_cos_sin_cache(128, 40960, 1000000.0)
    q, k = bf(g["rope_q"]), bf(g["rope_k"])
    q2, k2 = ops.rope(pos.to(DEV), q.to(DEV), k.to(DEV), table.to(DEV), 4, 2)
    assert torch.equal(q2.cpu().view(torch.int16), bf(g["rope_q_out"]).view(torch.int16))
    assert torch.equal(k2.cpu().view(torch.int16), bf(g["rope_k_out"]).view(torch.int16))


def test_silu_mul(ops, golden_layers):
    g = golden_layers
    assert_bf16_close(ops.silu_mul(bf(g["silu_x"]).to(DEV)), bf(g["silu_y"]))
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(32, 2 * 3072, generator=gen) * 3).bfloat16()
    assert_bf16_close(ops.silu_mul(x.to(DEV)), oracle.silu_and_mul(x))


# --------------------------------------------------------------------------- KV scatter (bit-exact)
@pytest.mark.parametrize("block_size", [16, 32, 256])
def test_kv_scatter_roundtrip(ops, block_size):
    g = torch.Generator().manual_seed(block_size)
    hkv, nblk, T = 8, 12, 150
    # k, v are strided views of a packed qkv row, as in the model
    qkv = torch.randn(T, (16 + 2 * hkv) * 128, generator=g).bfloat16()
    k = qkv[:, 16 * 128 : (16 + hkv) * 128].view(T, hkv, 128)
    v = qkv[:, (16 + hkv) * 128 :].view(T, hkv, 128)
    slots = torch.randperm(nblk * block_size, generator=g)[:T].to(torch.int32)
    slots[5] = -1  # skipped token
    kc_o = torch.zeros(nblk, block_size, hkv, 128, dtype=torch.bfloat16)
    vc_o = torch.zeros_like(kc_o)
    oracle.kv_scatter(k, v, kc_o, vc_o, slots)
    qd = qkv.to(DEV)
    kd = qd[:, 16 * 128 : (16 + hkv) * 128].view(T, hkv, 128)
    vd = qd[:, (16 + hkv) * 128 :].view(T, hkv, 128)
    kc = torch.zeros(ops.kv_cache_shape(nblk, hkv, block_size), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.reshape_and_cache(kd, vd, kc, vc, slots.to(DEV), hkv, block_size)
    assert torch.equal(to_logical(kc.cpu(), block_size, False).view(torch.int16), kc_o.view(torch.int16))
    assert torch.equal(to_logical(vc.cpu(), block_size, True).view(torch.int16), vc_o.view(torch.int16))
    # device-side gather helper agrees with the host model of the layout
    allslots = torch.arange(nblk * block_size, dtype=torch.int32, device=DEV)
    rows = ops.kv_cache_gather(kc, False, allslots, hkv, block_size).cpu()
    assert torch.equal(rows.view(torch.int16), kc_o.view(nblk * block_size, hkv * 128).view(torch.int16))
    rows = ops.kv_cache_gather(vc, True, allslots, hkv, block_size).cpu()
    assert torch.equal(rows.view(torch.int16), vc_o.view(nblk * block_size, hkv * 128).view(torch.int16))
    # decode-style 2-D slots ([block, offset], model_runner.py:353) on top of it
    B = 7
    k2 = torch.randn(B, hkv, 128, generator=g).bfloat16()
    v2 = torch.randn(B, hkv, 128, generator=g).bfloat16()
    s2 = torch.stack([torch.randperm(nblk, generator=g)[:B], torch.randint(0, block_size, (B,), generator=g)], 1).to(torch.int32)
    oracle.kv_scatter(k2, v2, kc_o, vc_o, (s2[:, 0] * block_size + s2[:, 1]))
    ops.scatter_update_kv(k2.to(DEV), v2.to(DEV), kc, vc, s2.to(DEV), hkv, block_size)
    assert torch.equal(to_logical(kc.cpu(), block_size, False).view(torch.int16), kc_o.view(torch.int16))
    assert torch.equal(to_logical(vc.cpu(), block_size, True).view(torch.int16), vc_o.view(torch.int16))


@pytest.mark.parametrize("block_size", [16, 64])
def test_kv_scatter_many_tokens_tiled_path(ops, block_size):
    """>= 64 tokens with flat slots take the tile-transposing V path: aligned consecutive runs
    (whole cache tiles), a run that starts mid-tile, a ragged tail, a skipped token and fully
    random slots must all land bit-exactly."""
    g = torch.Generator().manual_seed(block_size + 1)
    hkv, nblk = 8, 40
    T = 16 * 9 + 5
    k = torch.randn(T, hkv, 128, generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, generator=g).bfloat16()
    slots = torch.empty(T, dtype=torch.int32)
    slots[0:64] = torch.arange(64) + 3 * block_size          # 4 aligned tiles
    slots[64:96] = torch.arange(32) + 10 * block_size + 8     # consecutive but starting mid-tile
    free = torch.arange(20 * block_size, 30 * block_size)
    slots[96:] = free[torch.randperm(free.numel(), generator=g)[: T - 96]].to(torch.int32)  # random
    slots[100] = -1
    kc_o = torch.zeros(nblk, block_size, hkv, 128, dtype=torch.bfloat16)
    vc_o = torch.zeros_like(kc_o)
    oracle.kv_scatter(k, v, kc_o, vc_o, slots)
    kc = torch.zeros(ops.kv_cache_shape(nblk, hkv, block_size), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.reshape_and_cache(k.to(DEV), v.to(DEV), kc, vc, slots.to(DEV), hkv, block_size)
    assert torch.equal(to_logical(kc.cpu(), block_size, False).view(torch.int16), kc_o.view(torch.int16))
    assert torch.equal(to_logical(vc.cpu(), block_size, True).view(torch.int16), vc_o.view(torch.int16))


def test_kv_scatter_golden(ops, golden_attention):
    g = golden_attention
    hq, hkv, d, bs, nblk = (int(v) for v in g["meta"])
    kc = torch.zeros(ops.kv_cache_shape(nblk, hkv, bs), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    slots = torch.from_numpy(g["pre_slots"]).to(DEV)
    ops.reshape_and_cache(bf(g["pre_k"]).to(DEV), bf(g["pre_v"]).to(DEV), kc, vc, slots, hkv, bs)
    assert torch.equal(ops.kv_cache_gather(kc, False, slots, hkv, bs).cpu().view(torch.int16),
                       bf(g["pre_krows"]).reshape(-1, hkv * d).view(torch.int16))
    assert torch.equal(ops.kv_cache_gather(vc, True, slots, hkv, bs).cpu().view(torch.int16),
                       bf(g["pre_vrows"]).reshape(-1, hkv * d).view(torch.int16))
    assert int((to_logical(kc.cpu(), bs, False).reshape(-1, hkv * d) != 0).any(dim=1).sum()) == int(g["pre_cache_nonzero_rows"])


@pytest.mark.parametrize("T_tokens", [37, 200])
def test_fused_qknorm_rope_store_equals_unfused(ops, T_tokens):
    g = torch.Generator().manual_seed(77)
    T, hq, hkv, bs, nblk = T_tokens, 16, 8, 16, 24
    qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=g) * 2).bfloat16().to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    kw = (1 + 0.2 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    pos = torch.randint(0, 4096, (T,), generator=g).to(DEV)
    table = oracle.build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    slots = torch.randperm(nblk * bs, generator=g)[:T].to(torch.int32).to(DEV)
    q_view = qkv[:, : hq * 128].view(T, hq, 128)
    k_view = qkv[:, hq * 128 : (hq + hkv) * 128].view(T, hkv, 128)
    v_view = qkv[:, (hq + hkv) * 128 :].view(T, hkv, 128)
    qn, kn = ops.rmsnorm(q_view, qw, 1e-6), ops.rmsnorm(k_view, kw, 1e-6)
    qr, kr = ops.rope(pos, qn, kn, table, hq, hkv)
    kc1 = torch.zeros(ops.kv_cache_shape(nblk, hkv, bs), dtype=torch.bfloat16, device=DEV)
    vc1 = torch.zeros_like(kc1)
    ops.reshape_and_cache(kr, v_view, kc1, vc1, slots, hkv, bs)
    kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(kc1)
    q2 = ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc2, vc2, slots, hq, hkv, bs)
    assert torch.equal(q2.view(torch.int16), qr.reshape(T, -1).view(torch.int16))
    assert torch.equal(kc1.view(torch.int16), kc2.view(torch.int16))
    assert torch.equal(vc1.view(torch.int16), vc2.view(torch.int16))
    # and against the oracle (norm is last-bit sensitive -> ulp tolerance)
    qo = oracle.apply_rope(pos.cpu(), oracle.rms_norm(q_view.cpu(), qw.cpu(), 1e-6), table.cpu())
    assert_bf16_close(q2.cpu().view(T, hq, 128), qo)
    # 2-D slots, no norm weights (attention_bias models)
    s2 = torch.stack([slots // bs, slots % bs], 1).contiguous()
    kc3, vc3 = torch.zeros_like(kc1), torch.zeros_like(kc1)
    q3 = ops.qknorm_rope_store(qkv, None, None, 1e-6, pos, table, kc3, vc3, s2, hq, hkv, bs)
    qr3, kr3 = ops.rope(pos, q_view, k_view, table, hq, hkv)
    assert torch.equal(q3.view(torch.int16), qr3.reshape(T, -1).view(torch.int16))
    kc4, vc4 = torch.zeros_like(kc1), torch.zeros_like(kc1)
    ops.reshape_and_cache(kr3, v_view, kc4, vc4, slots, hkv, bs)
    assert torch.equal(kc3.view(torch.int16), kc4.view(torch.int16)) and torch.equal(vc3.view(torch.int16), vc4.view(torch.int16))


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 1), (4, 4)])
def test_fused_qknorm_rope_store_prefill_tiles(ops, hq, hkv):
    """>= 64 tokens with flat slots: one workgroup per (16 tokens, kv head) assembles whole K / V cache
    tiles.  Slot layout as a prefill produces it (consecutive runs from block boundaries) plus a run
    that starts mid-tile, skipped tokens (-1) and a ragged tail; must equal the per-token kernel."""
    g = torch.Generator().manual_seed(5 + hq)
    bs, nblk = 16, 40
    runs = [(3 * bs, 64), (9 * bs + 5, 36), (14 * bs, 32), (20 * bs, 45), (25 * bs + 15, 20)]  # (first slot, tokens)
    slot_list = [s0 + i for s0, n in runs for i in range(n)]
    T = len(slot_list)
    slots = torch.tensor(slot_list, dtype=torch.int32)
    slots[70] = -1
    slots[130:133] = -1
    slots = slots.to(DEV)
    qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=g) * 2).bfloat16().to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    kw = (1 + 0.2 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    pos = torch.randint(0, 4096, (T,), generator=g).to(DEV)
    table = oracle.build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    fill = torch.full(ops.kv_cache_shape(nblk, hkv, bs), 7.0, dtype=torch.bfloat16, device=DEV)
    # reference: the same op through its per-token kernel (2-D slots never take the tile kernel)
    s2 = torch.stack([torch.div(slots, bs, rounding_mode="floor"), slots % bs], 1).to(torch.int32)
    s2[slots < 0] = -1
    kc1, vc1 = fill.clone(), fill.clone()
    q1 = ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc1, vc1, s2.contiguous(), hq, hkv, bs)
    kc2, vc2 = fill.clone(), fill.clone()
    q2 = ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc2, vc2, slots, hq, hkv, bs)
    assert torch.equal(q2.view(torch.int16), q1.view(torch.int16))
    assert torch.equal(kc1.view(torch.int16), kc2.view(torch.int16))
    assert torch.equal(vc1.view(torch.int16), vc2.view(torch.int16))
    assert not torch.equal(kc2.view(torch.int16), fill.view(torch.int16))
    # and the oracle for q (norm is last-bit sensitive -> ulp tolerance)
    q_view = qkv[:, : hq * 128].view(T, hq, 128)
    qo = oracle.apply_rope(pos.cpu(), oracle.rms_norm(q_view.cpu(), qw.cpu(), 1e-6), table.cpu())
    assert_bf16_close(q2.cpu().view(T, hq, 128), qo)


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M", [1, 7, 16, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072), (256, 128), (512, 160), (16, 32)])
def test_gemm_skinny(ops, M, N, K):
    """fp32 accumulate, one rounding: <= 1 bf16 ulp from the oracle on a small fraction
    of outputs (summation order)."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    y = ops.gemm_skinny(x.to(DEV), w.to(DEV))
    atol = K * 2.0 ** -22  # fp32 accumulation noise bound for |x| ~ 3, |w| ~ 0.15
    assert_bf16_close(y, oracle.linear(x, w), max_ulp=1, max_frac=2e-2, atol=atol)
    yb = ops.gemm_skinny(x.to(DEV), w.to(DEV), b.to(DEV))
    assert_bf16_close(yb, oracle.linear(x, w, b), max_ulp=1, max_frac=2e-2, atol=atol)
    # transpose / fragment-layout detector: asymmetric weights, identity-like x
    if M >= 16 and K >= 32:
        xe = torch.zeros(M, K).bfloat16()
        xe[3, 5] = 1.0
        ye = ops.gemm_skinny(xe.to(DEV), w.to(DEV)).cpu()
        assert torch.equal(ye[3].view(torch.int16), w[:, 5].contiguous().view(torch.int16))
        assert float(ye.float().abs().sum() - ye[3].float().abs().sum()) == 0.0


def _pack_ref(w):
    """host model of mi_pack_weight: [N/16][K/32][4 (g)][16 (r)][8]"""
    N, K = w.shape
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


@pytest.mark.parametrize("M", [1, 16, 32, 48, 64])
@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072), (256, 128), (32, 160)])
def test_gemm_packed(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    wp = ops.pack_weight(w.to(DEV))
    assert torch.equal(wp.cpu().view(torch.int16), _pack_ref(w).view(torch.int16))  # pure permutation
    atol = K * 2.0 ** -22
    assert_bf16_close(ops.gemm_packed(x.to(DEV), wp), oracle.linear(x, w), max_frac=2e-2, atol=atol)
    assert_bf16_close(ops.gemm_packed(x.to(DEV), wp, b.to(DEV)), oracle.linear(x, w, b), max_frac=2e-2, atol=atol)
    # fused SiluAndMul epilogue == gate_up GEMM followed by the activation kernel
    want = oracle.silu_and_mul(oracle.linear(x, w))
    got = ops.gemm_packed(x.to(DEV), wp, silu_mul=True)
    # a 1-ulp flip of the gate (GEMM summation order) times |up| <= ~6: widen the near-zero atol
    assert_bf16_close(got, want, max_ulp=2, max_frac=3e-2, atol=32 * atol)
    # split-K partials + add_rmsnorm_splitk == GEMM + add_rmsnorm
    r = torch.randn(M, N, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(N, generator=g)).bfloat16()
    yo, ro = oracle.add_rms_norm(oracle.linear(x, w), r, nw, 1e-6)
    for ks in (1, 2, 4):
        if K % (32 * ks):
            continue
        parts = ops.gemm_packed_splitk(x.to(DEV), wp, ks)
        y, r2 = ops.add_rmsnorm_splitk(parts, r.to(DEV), nw.to(DEV), 1e-6)
        # x + residual may cancel: one bf16 ulp of |x| <= 4 (3.1e-2) is the absolute noise floor
        assert_bf16_close(r2, ro, max_frac=2e-2, atol=4e-2)
        assert_bf16_close(y, yo, max_ulp=2, max_frac=3e-2, atol=4e-2)


@pytest.mark.parametrize("M", [65, 128, 200, 512])
@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (6144, 1024), (256, 128)])
def test_gemm_packed_more_than_64_rows(ops, M, N, K):
    """decode batches above 64 sequences: the weight-streaming kernels walk the rows in chunks of 64 (grid.z) -
    plain, SwiGLU, split-K partials + add_rmsnorm, complete rows (rows4), all against the oracle"""
    g = torch.Generator().manual_seed(M * 5 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    wp = ops.pack_weight(w.to(DEV))
    atol = K * 2.0 ** -22
    assert_bf16_close(ops.gemm_skinny(x.to(DEV), w.to(DEV), b.to(DEV)), oracle.linear(x, w, b), max_frac=2e-2, atol=atol)
    assert_bf16_close(ops.gemm_packed(x.to(DEV), wp), oracle.linear(x, w), max_frac=2e-2, atol=atol)
    assert_bf16_close(ops.gemm_packed(x.to(DEV), wp, silu_mul=True), oracle.silu_and_mul(oracle.linear(x, w)),
                      max_ulp=2, max_frac=3e-2, atol=32 * atol)
    assert_bf16_close(ops.gemm_rows4(x.to(DEV), ops.pack_weight_rows4(w.to(DEV))), oracle.linear(x, w),
                      max_frac=2e-2, atol=atol)
    r = torch.randn(M, N, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(N, generator=g)).bfloat16()
    yo, ro = oracle.add_rms_norm(oracle.linear(x, w), r, nw, 1e-6)
    parts = ops.gemm_packed_splitk(x.to(DEV), wp, 2 if K % 64 == 0 else 1)
    y, r2 = ops.add_rmsnorm_splitk(parts, r.to(DEV), nw.to(DEV), 1e-6)
    assert_bf16_close(r2, ro, max_frac=2e-2, atol=4e-2)
    assert_bf16_close(y, yo, max_ulp=2, max_frac=3e-2, atol=4e-2)
    # each 64-row chunk is computed exactly as a 64-row call on its own
    lo = ops.gemm_packed(x[64:128].contiguous().to(DEV), wp) if M >= 128 else None
    if lo is not None:
        assert torch.equal(lo.view(torch.int16), ops.gemm_packed(x.to(DEV), wp)[64:128].view(torch.int16))


@pytest.mark.parametrize("N,K", [(1280, 5120), (5120, 1024), (6400, 5120), (5120, 3200)])
def test_gemm_packed_qwen3_32b_tp8_shapes(ops, N, K):
    """per-rank projection shapes of BASELINE.json configs[2] (Qwen3-32B, TP=8): hidden 5120,
    64 q / 8 kv heads, intermediate 25600"""
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(32, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16()
    wp = ops.pack_weight(w.to(DEV))
    atol = K * 2.0 ** -22
    assert_bf16_close(ops.gemm_packed(x.to(DEV), wp), oracle.linear(x, w), max_frac=2e-2, atol=atol)
    if N % 32 == 0:
        assert_bf16_close(ops.gemm_packed(x.to(DEV), wp, silu_mul=True), oracle.silu_and_mul(oracle.linear(x, w)),
                          max_ulp=2, max_frac=3e-2, atol=32 * atol)


@pytest.mark.parametrize("M", [5, 32])
@pytest.mark.parametrize("N,K,silu", [(1280, 5120, False), (6400, 5120, True), (5120, 3200, False), (4096, 4096, False),
                                      (256, 8192, False), (1024, 12288, False)])
def test_gemm_packed_double_buffered_k_loop_is_the_same_sum(ops, M, N, K, silu):
    """Round 5: K-slices of two or more blocks run the double-buffered loop (gemm_skinny_kernel PIPE: the next block's
    loads issued before the current one is consumed).  One MFMA chain per wave over ascending k either way, so the
    results are the SAME BITS as with the one-block-at-a-time loop (tuning knob MI_TUNE_GEMM_PIPE = 0), plain, SwiGLU
    and split-K partials, and within the GEMM bound of the oracle."""
    from nanovllm import _C

    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16()
    wp = ops.pack_weight(w.to(DEV))
    got = {}
    try:
        for pipe in (1, 0):
            _C.set_tuning(_C.TUNE_GEMM_PIPE, pipe)
            got[pipe] = (ops.gemm_packed(x, wp, silu_mul=silu), ops.gemm_packed_splitk(x, wp, 2))
    finally:
        _C.set_tuning(_C.TUNE_GEMM_PIPE, 1)
    assert torch.equal(got[1][0].view(torch.int16), got[0][0].view(torch.int16))
    assert torch.equal(got[1][1], got[0][1])
    want = oracle.linear(x.cpu(), w)
    atol = K * 2.0 ** -22
    if silu:
        assert_bf16_close(got[1][0], oracle.silu_and_mul(want), max_ulp=2, max_frac=3e-2, atol=32 * atol)
    else:
        assert_bf16_close(got[1][0], want, max_frac=2e-2, atol=atol)
    assert_bf16_close(got[1][1].sum(0).bfloat16(), want, max_frac=2e-2, atol=atol)


def test_instrumented_chain_launches_compute_the_product(ops):
    """mi_gemm_bf16_packed_ex / mi_add_rmsnorm_splitk_ex (tools/chain_timeline.py) are the product kernels plus clock
    stamps: same bits, and every wave's seven stamps are monotonic."""
    g = torch.Generator().manual_seed(12)
    B, H = 32, 1024
    x = torch.randn(B, H, generator=g).bfloat16().to(DEV)
    w_qkv = ops.pack_weight((torch.randn(4096, H, generator=g) * 0.03).bfloat16().to(DEV))
    w_gu = ops.pack_weight((torch.randn(6144, H, generator=g) * 0.03).bfloat16().to(DEV))
    w_dn = ops.pack_weight((torch.randn(H, 3072, generator=g) * 0.03).bfloat16().to(DEV))
    st = lambda wg, wv: torch.zeros(wg, wv, 8, dtype=torch.int64, device=DEV)  # noqa: E731
    s1, s2, s3, s4 = st(256, 16), st(192, 16), st(256, 12), st(B, 4)
    y = ops.gemm_packed_stamped(x, w_qkv, s1)
    assert torch.equal(y.view(torch.int16), ops.gemm_packed(x, w_qkv).view(torch.int16))
    a = ops.gemm_packed_stamped(x, w_gu, s2, silu_mul=True)
    assert torch.equal(a.view(torch.int16), ops.gemm_packed(x, w_gu, silu_mul=True).view(torch.int16))
    p = ops.gemm_packed_stamped(a, w_dn, s3, ksplit=4)
    assert torch.equal(p, ops.gemm_packed_splitk(a, w_dn, 4))
    nw = torch.ones(H, device=DEV).bfloat16()
    n1 = ops.add_rmsnorm_splitk_stamped(p, x, nw, 1e-6, s4)
    n0 = ops.add_rmsnorm_splitk(p, x, nw, 1e-6)
    assert all(torch.equal(u.view(torch.int16), v.view(torch.int16)) for u, v in zip(n1, n0))
    for s in (s1, s2, s3, s4):
        t = s.cpu()[..., :7]
        assert (t > 0).all() and (t[..., 1:] >= t[..., :-1]).all()
        assert (t.max() - t.min()).item() < 100 * 1000  # one launch: well under a millisecond of the 100 MHz clock


def _pack_rows4_ref(w):
    """host model of mi_pack_weight_rows4: [N/4][K/32][4 (g)][4 (n)][8]"""
    N, K = w.shape
    return w.view(N // 4, 4, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


@pytest.mark.parametrize("M", [1, 7, 16, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(1024, 2048), (1024, 3072), (5120, 1024), (5120, 3200), (2048, 1024), (64, 96),
                                 (4, 32), (1024, 256)])
def test_gemm_rows4(ops, M, N, K):
    """row-parallel projections (o_proj / down_proj; also the per-rank shards of TP 2..8 and Qwen3-32B widths):
    complete rows, fp32 accumulate, one rounding"""
    g = torch.Generator().manual_seed(M * 5 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    w4 = ops.pack_weight_rows4(w.to(DEV))
    assert torch.equal(w4.cpu().view(torch.int16), _pack_rows4_ref(w).view(torch.int16))  # pure permutation
    y = ops.gemm_rows4(x.to(DEV), w4)
    assert_bf16_close(y, oracle.linear(x, w), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    if M >= 4 and K >= 32:  # transpose / layout detector
        xe = torch.zeros(M, K).bfloat16()
        xe[3, 5] = 1.0
        ye = ops.gemm_rows4(xe.to(DEV), w4).cpu()
        assert torch.equal(ye[3].view(torch.int16), w[:, 5].contiguous().view(torch.int16))
        ye[3] = 0
        assert not bool(ye.float().abs().max() > 0)  # every other row is exactly zero


def test_gemm_lm_head_shape(ops):
    g = torch.Generator().manual_seed(1)
    M, N, K = 32, 151936, 1024
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16()
    y = ops.gemm_skinny(x.to(DEV), w.to(DEV)).cpu()
    ref = oracle.linear(x, w)
    assert_bf16_close(y, ref, max_ulp=1, max_frac=2e-2, atol=1024 * 2.0 ** -22)
    y = ops.gemm_packed(x.to(DEV), ops.pack_weight(w.to(DEV))).cpu()  # the persistent head kernel
    assert_bf16_close(y, ref, max_ulp=1, max_frac=2e-2, atol=1024 * 2.0 ** -22)


@pytest.mark.parametrize("M,N,K", [(64, 151936, 1024), (40, 65536, 1024), (48, 151936, 896), (32, 128256, 2048)])
def test_gemm_head_kernel_more_rows_and_other_hidden_sizes(ops, M, N, K):
    """round 4: the persistent head kernel beyond 32 rows x K = 1024 against the oracle, and
    against the short-lived-workgroup kernel it replaces for these shapes (same fp32 chains per wave, another order
    of the wave sums: <= 1 ulp)."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16()
    ref = oracle.linear(x, w)
    y = ops.gemm_packed(x.to(DEV), ops.pack_weight(w.to(DEV))).cpu()
    assert_bf16_close(y, ref, max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    # transpose detector: a one-hot activation row reads a weight column back
    xe = torch.zeros(M, K).bfloat16()
    xe[M - 1, 5] = 1.0
    ye = ops.gemm_packed(xe.to(DEV), ops.pack_weight(w.to(DEV))).cpu()
    assert torch.equal(ye[M - 1].view(torch.int16), w[:, 5].contiguous().view(torch.int16))
    assert int((ye[: M - 1].view(torch.int16) & 0x7FFF).count_nonzero()) == 0


# --------------------------------------------------------------------------- attention
def _random_paged_case(gen, hq, hkv, block_size, ctx_lens, extra_blocks=3, scale_q=1.0):
    d = 128
    need = sum((n + block_size - 1) // block_size for n in ctx_lens)
    nblk = need + extra_blocks
    kc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    vc = torch.randn(nblk, block_size, hkv, d, generator=gen).bfloat16()
    perm = torch.randperm(nblk, generator=gen).tolist()
    tables = []
    for n in ctx_lens:
        nb = (n + block_size - 1) // block_size
        tables.append([perm.pop() for _ in range(nb)])
    width = max(1, max(len(t) for t in tables)) + 2  # trailing -1 padding columns
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    q = (torch.randn(len(ctx_lens), hq, d, generator=gen) * scale_q).bfloat16()
    return q, kc, vc, bt


def test_decode_attention_golden(ops, golden_attention, attn_geometry):
    """vs the reference's own CPU attention (bf16 S/P): bound 3e-2; vs oracle: 2e-3."""
    g = golden_attention
    hq, hkv, d, bs, nblk = (int(v) for v in g["meta"])
    kc_l, vc_l = bf(g["dec_kcache_before"]).clone(), bf(g["dec_vcache_before"]).clone()
    kc, vc = to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV)
    slots = torch.from_numpy(g["dec_slots"])
    s2 = torch.stack([slots // bs, slots % bs], 1).to(torch.int32).contiguous().to(DEV)
    ops.scatter_update_kv(bf(g["dec_k"]).to(DEV), bf(g["dec_v"]).to(DEV), kc, vc, s2, hkv, bs)
    ctx = torch.from_numpy(g["dec_ctx"])
    bt = torch.from_numpy(g["dec_tables"])
    out = ops.paged_attn_decode(bf(g["dec_q"]).to(DEV), kc, vc, bt.to(DEV), ctx.to(DEV), hq, hkv, bs,
                                1.0 / math.sqrt(d)).cpu()
    assert (out.float() - bf(g["dec_out"]).float()).abs().max().item() <= 3e-2
    oracle.kv_scatter(bf(g["dec_k"]), bf(g["dec_v"]), kc_l, vc_l, slots)
    want = oracle.paged_attention_decode(bf(g["dec_q"]), kc_l, vc_l, bt, ctx, keep_fp32=True)
    assert (out.float() - want).abs().max().item() <= 8e-3  # half a bf16 ulp at |o|<=2 + fp32 noise


# --------------------------------------------------------------------------- fp8 weights
# (the lm_head-sized case - many row tiles per launch - is checked at one batch size)
@pytest.mark.parametrize("N,K,M", [(n, k, m) for n, k in ((4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072), (256, 128))
                                   for m in (1, 32, 64)] + [(65536, 1024, 32)])
def test_gemm_fp8_weights(ops, M, N, K):
    """e4m3 weights + per-row fp32 scale, bf16 activations: y = x @ (w_q * scale)^T, one rounding.
    The oracle multiplies the dequantised weights in fp32; the kernel multiplies exact bf16 copies of
    w_q and scales the fp32 sums - equal up to fp32 rounding, i.e. <= 1 bf16 ulp on a few outputs."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16()
    q, scale = oracle.quantize_fp8_rows(w)  # integer-arithmetic E4M3 restatement (oracle/layers.py)
    # the product's quantiser, run on the GPU, produces the oracle's bytes and scales
    fw = ops.pack_weight_fp8(w.to(DEV))
    qg, sg = ops.quantize_fp8(w.to(DEV))
    assert torch.equal(qg.cpu(), q) and torch.equal(sg.cpu(), scale)
    wd = oracle.dequantize_fp8_rows(q, scale)
    assert (wd - w.float()).abs().max() <= w.float().abs().amax(dim=1).max() / 16  # e4m3: 3 mantissa bits
    want32 = x.float() @ wd.T
    want = want32.bfloat16()
    got = ops.gemm_packed(x.to(DEV), fw).cpu()
    assert_bf16_close(got, want, max_ulp=1, max_frac=0.02, atol=K * 2.0 ** -22)
    if N <= 2048:
        for ks in (2, 4):
            if K % (64 * ks):
                continue
            parts = ops.gemm_packed_splitk(x.to(DEV), fw, ks).cpu()
            assert_bf16_close(parts.sum(0).bfloat16(), want, max_ulp=1, max_frac=0.02, atol=K * 2.0 ** -22)
    if N == 6144:
        act = ops.gemm_packed(x.to(DEV), fw, silu_mul=True).cpu()
        gate, up = want[:, : N // 2].float(), want[:, N // 2:].float()
        ref = (torch.nn.functional.silu(gate).bfloat16().float() * up).bfloat16()
        assert_bf16_close(act, ref, max_ulp=2, max_frac=0.05, atol=32 * K * 2.0 ** -22)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (64, 8), (16, 1)])
@pytest.mark.parametrize("block_size", [16, 64, 256])
def test_decode_attention_random(ops, hq, hkv, block_size, attn_geometry):
    """fp32-softmax oracle; kernel keeps P to ~16 bits (hi/lo bf16 split), so the only
    error left is the final bf16 rounding: |out - oracle_fp32| <= half a bf16 ulp."""
    gen = torch.Generator().manual_seed(hq * 1000 + hkv * 10 + block_size)
    ctx_lens = [1, 15, 16, 17, 31, 32, 33, 100, 513, 1024, 0, 2049]
    q, kc_l, vc_l, bt = _random_paged_case(gen, hq, hkv, block_size, ctx_lens)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc_l, vc_l, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode(q.to(DEV), to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV),
                                bt.to(DEV), ctx.to(DEV), hq, hkv, block_size, 1.0 / math.sqrt(128)).cpu()
    err = (out.float() - want).abs()
    tol = want.abs() * 2 ** -8 + 1e-4
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}"
    assert float(out[ctx_lens.index(0)].float().abs().max()) == 0.0  # padded row -> zeros
    # strided q (a view into a packed qkv row) gives the same bits
    packed = torch.zeros(len(ctx_lens), (hq + 2 * hkv) * 128, dtype=torch.bfloat16)
    packed[:, : hq * 128] = q.reshape(len(ctx_lens), -1)
    out2 = ops.paged_attn_decode(packed.to(DEV)[:, : hq * 128], to_fragment(kc_l, False).to(DEV),
                                 to_fragment(vc_l, True).to(DEV), bt.to(DEV), ctx.to(DEV), hq, hkv, block_size,
                                 1.0 / math.sqrt(128)).cpu()
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_decode_attention_sharp_softmax(ops, attn_geometry):
    """large score spread (forces the online-softmax rescale path across chunks/waves/splits)"""
    gen = torch.Generator().manual_seed(9)
    ctx_lens = [700, 1024, 3000]
    q, kc_l, vc_l, bt = _random_paged_case(gen, 16, 8, 16, ctx_lens, scale_q=6.0)
    # spike: one late key aligned with q of sequence 1
    blk, off = int(bt[1][1000 // 16]), 1000 % 16
    kc_l[blk, off, :, :] = (q[1].view(8, 2, 128)[:, 0] * 1.5).bfloat16()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc_l, vc_l, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode(q.to(DEV), to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV),
                                bt.to(DEV), ctx.to(DEV), 16, 8, 16, 1.0 / math.sqrt(128)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 2e-4).all()), err.max().item()


@pytest.mark.parametrize("batch", [1, 32, 256])
def test_decode_attention_batch_sizes(ops, batch, attn_geometry):
    gen = torch.Generator().manual_seed(batch)
    ctx_lens = [int(x) for x in torch.randint(1, 300, (batch,), generator=gen)]
    q, kc_l, vc_l, bt = _random_paged_case(gen, 16, 8, 16, ctx_lens)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = oracle.paged_attention_decode(q, kc_l, vc_l, bt, ctx, keep_fp32=True)
    out = ops.paged_attn_decode(q.to(DEV), to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV),
                                bt.to(DEV), ctx.to(DEV), 16, 8, 16, 1.0 / math.sqrt(128)).cpu()
    err = (out.float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 1e-4).all()), err.max().item()


@pytest.mark.parametrize("hq,hkv,with_norm", [(16, 8, True), (8, 8, True), (32, 8, True), (64, 8, True), (16, 1, False),
                                               (16, 8, False)])
@pytest.mark.parametrize("block_size", [16, 64])
def test_decode_attention_fused_step(ops, hq, hkv, with_norm, block_size):
    """mi_paged_attn_decode_fused == mi_qknorm_rope_store followed by mi_paged_attn_decode, bit for bit:
    attention output and cache contents (context ends at tile / chunk / block boundaries, one-token
    contexts, a padded row, contexts long enough for every wave and - small batch - several splits)."""
    gen = torch.Generator().manual_seed(hq * 31 + hkv * 7 + block_size)
    ctx_lens = [1, 2, 15, 16, 17, 31, 32, 33, 48, 49, 100, 257, 512, 1024, 1025, 1040, 0, 2049]
    B = len(ctx_lens)
    _, kc_l, vc_l, bt = _random_paged_case(gen, hq, hkv, block_size, ctx_lens, extra_blocks=4)
    nblk = kc_l.shape[0]
    used = set(int(b) for b in bt.flatten() if b >= 0)
    dummy = next(b for b in range(nblk) if b not in used)  # padded rows point at a block nobody reads
    qkv = (torch.randn(B, (hq + 2 * hkv) * 128, generator=gen) * 1.5).bfloat16().to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=gen)).bfloat16().to(DEV) if with_norm else None
    kw = (1 + 0.2 * torch.randn(128, generator=gen)).bfloat16().to(DEV) if with_norm else None
    table = oracle.build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    pos = torch.tensor([max(n - 1, 0) for n in ctx_lens], dtype=torch.int64)
    slots = torch.tensor([[int(bt[i][(n - 1) // block_size]), (n - 1) % block_size] if n > 0 else [dummy, 0]
                          for i, n in enumerate(ctx_lens)], dtype=torch.int32)
    kc1, vc1 = to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV)
    kc2, vc2 = kc1.clone(), vc1.clone()
    scale = 1.0 / math.sqrt(128)
    args = (bt.to(DEV), ctx.to(DEV), hq, hkv, block_size, scale)
    q = ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos.to(DEV), table, kc1, vc1, slots.to(DEV), hq, hkv, block_size)
    out1 = ops.paged_attn_decode(q, kc1, vc1, *args)
    out2 = ops.paged_attn_decode_fused(qkv, qw, kw, 1e-6, pos.to(DEV), table, slots.to(DEV), kc2, vc2, *args)
    torch.cuda.synchronize()
    assert torch.equal(out1.view(torch.int16), out2.view(torch.int16))
    keep = [b for b in range(nblk) if b != dummy]  # the fused launch does not write a padded row's dummy slot
    assert torch.equal(kc1[keep].view(torch.int16), kc2[keep].view(torch.int16))
    assert torch.equal(vc1[keep].view(torch.int16), vc2[keep].view(torch.int16))
    # ... and it IS the attention of the oracle over the updated cache
    want = oracle.paged_attention_decode(q.cpu().view(B, hq, 128), to_logical(kc1.cpu(), block_size, False),
                                         to_logical(vc1.cpu(), block_size, True), bt, ctx, keep_fp32=True)
    err = (out2.cpu().float() - want).abs()
    assert bool((err <= want.abs() * 2 ** -8 + 1e-4).all()), err.max().item()
    # few sequences: the context is split over several workgroups (merge kernel), same bits again
    few = [13, 17]  # rows with ctx 1024 and 2049
    sel = torch.tensor(few)
    kc3, vc3 = to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV)
    out3 = ops.paged_attn_decode_fused(qkv[sel.to(DEV)].contiguous(), qw, kw, 1e-6, pos[sel].to(DEV), table,
                                       slots[sel].contiguous().to(DEV), kc3, vc3, bt[sel].contiguous().to(DEV),
                                       ctx[sel].to(DEV), hq, hkv, block_size, scale)
    err3 = (out3.cpu().float() - want[sel]).abs()
    assert bool((err3 <= want[sel].abs() * 2 ** -8 + 1e-4).all()), err3.max().item()
    for i in few:  # the stored rows are the same bits
        b, off = int(slots[i][0]), int(slots[i][1])
        assert torch.equal(to_logical(kc3.cpu(), block_size, False)[b, off].view(torch.int16),
                           to_logical(kc1.cpu(), block_size, False)[b, off].view(torch.int16))
        assert torch.equal(to_logical(vc3.cpu(), block_size, True)[b, off].view(torch.int16),
                           to_logical(vc1.cpu(), block_size, True)[b, off].view(torch.int16))


def test_prefill_attention_golden(ops, golden_attention):
    g = golden_attention
    hq, hkv, d, bs, nblk = (int(v) for v in g["meta"])
    kc = torch.zeros(ops.kv_cache_shape(nblk, hkv, bs), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.reshape_and_cache(bf(g["pre_k"]).to(DEV), bf(g["pre_v"]).to(DEV), kc, vc,
                          torch.from_numpy(g["pre_slots"]).to(DEV), hkv, bs)
    cu = torch.from_numpy(g["pre_cu"])
    lens = (cu[1:] - cu[:-1]).to(torch.int32)
    out = ops.paged_attn_prefill(bf(g["pre_q"]).to(DEV), kc, vc, torch.from_numpy(g["pre_tables"]).to(DEV),
                                 cu.to(DEV), lens.to(DEV), int(lens.max()), hq, hkv, bs, 1.0 / math.sqrt(d)).cpu()
    assert (out.float() - bf(g["pre_out"]).float()).abs().max().item() <= 3e-2


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (64, 8), (8, 1), (16, 1)])
@pytest.mark.parametrize("block_size", [16, 48, 256])
def test_prefill_attention_random(ops, hq, hkv, block_size, prefill_p):
    gen = torch.Generator().manual_seed(hq + hkv + block_size)
    q_lens = [1, 7, 16, 33, 129, 260]
    kv_lens = [1, 7, 16, 33, 129 + 64, 260]  # one sequence with a 64-token cached prefix
    T = sum(q_lens)
    _, kc_l, vc_l, bt = _random_paged_case(gen, hq, hkv, block_size, kv_lens)
    q = torch.randn(T, hq, 128, generator=gen).bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(q_lens)), dtype=torch.int32)
    kvl = torch.tensor(kv_lens, dtype=torch.int32)
    want = oracle.paged_attention_prefill(q, kc_l, vc_l, bt, cu, kvl, keep_fp32=True)
    want_absv = oracle.paged_attention_prefill(q, kc_l, vc_l.abs(), bt, cu, kvl, keep_fp32=True)
    out = ops.paged_attn_prefill(q.to(DEV), to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV),
                                 bt.to(DEV), cu.to(DEV), kvl.to(DEV), max(q_lens), hq, hkv, block_size,
                                 1.0 / math.sqrt(128)).cpu()
    err = (out.float() - want).abs()
    tol = _prefill_bound(want, want_absv, prefill_p)
    assert bool((err <= tol).all()), (err - tol).max().item()


def test_prefill_attention_chunk_pipeline_stress(ops):
    """Kernel-level guard of the prefill attention's LDS-DMA ring (the counted `s_waitcnt vmcnt(4)` of its chunk
    loop): the bench's prefill step - 16 sequences x 1024 tokens, Qwen3-0.6B heads, block 16 - with SCRAMBLED block
    tables, 100 launches of each request order (chunks behind / ahead of the Q preparation) next to a competing
    HBM stream.  Every launch must reproduce the first one bit for bit, both orders must agree, and the result must
    be the unscrambled layout's (paging invariance) and the fp32 oracle's on a sample of rows."""
    gen = torch.Generator().manual_seed(77)
    hq, hkv, bs, n_seqs, L = 16, 8, 16, 16, 1024
    T = n_seqs * L
    nblk = n_seqs * (L // bs)
    qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=gen) * 0.8).bfloat16().to(DEV)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
    table = oracle.build_cos_sin_cache(128, 2048, 1e6).to(DEV)
    pos = torch.arange(L, dtype=torch.int64).repeat(n_seqs).to(DEV)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32).to(DEV)
    kvl = torch.full((n_seqs,), L, dtype=torch.int32, device=DEV)

    def run(perm, variant, reps):
        bt = perm.view(n_seqs, L // bs).to(torch.int32).to(DEV)
        slots = (bt.long().repeat_interleave(bs, dim=1) * bs + torch.arange(bs, device=DEV).repeat(L // bs)).view(-1).to(torch.int32)
        kc = torch.zeros(ops.kv_cache_shape(nblk + 7, hkv, bs), dtype=torch.bfloat16, device=DEV)
        vc = torch.zeros_like(kc)
        ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc, vc, slots, hq, hkv, bs, store_q=False)
        first = ops.paged_attn_prefill_fused(qkv, qw, 1e-6, pos, table, kc, vc, bt, cu, kvl, L, hq, hkv, bs, 128 ** -0.5,
                                             variant=variant)
        side, junk = torch.cuda.Stream(), torch.empty(192 << 20, dtype=torch.uint8, device=DEV)
        for it in range(reps):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)  # uneven memory pressure next to the DMA ring
            again = ops.paged_attn_prefill_fused(qkv, qw, 1e-6, pos, table, kc, vc, bt, cu, kvl, L, hq, hkv, bs,
                                                 128 ** -0.5, variant=variant)
            assert torch.equal(again.view(torch.int16), first.view(torch.int16)), (variant, it)
        torch.cuda.synchronize()
        return first, (kc, vc, bt)

    straight = torch.arange(nblk)
    scrambled = torch.randperm(nblk + 7, generator=gen)[:nblk]
    ref, _ = run(straight, 0, 3)
    late, (kc, vc, bt) = run(scrambled, 0, 100)
    early, _ = run(scrambled, 1, 100)
    paired, _ = run(scrambled, 2, 30)
    assert torch.equal(late.view(torch.int16), ref.view(torch.int16))    # paging invariance
    assert torch.equal(early.view(torch.int16), late.view(torch.int16))  # request order does not matter
    assert torch.equal(paired.view(torch.int16), late.view(torch.int16))  # nor does one barrier per two chunks
    # round 4's schedule changes (hand-issued un-merged V reads, block ids read a chunk ahead with the request side as
    # running state) against round 3's forms of the same arithmetic: 8 = merged V reads, 16 = table read + divisions in
    # front of every request, 24 = both
    for v in (8, 16, 24):
        old_form, _ = run(scrambled, v, 10)
        assert torch.equal(old_form.view(torch.int16), late.view(torch.int16)), v
    # P as bf16 hi + lo: the new schedule (4) against the round-3 kernel as a whole (28)
    split_new, _ = run(scrambled, 4, 30)
    split_r03, _ = run(scrambled, 28, 10)
    assert torch.equal(split_new.view(torch.int16), split_r03.view(torch.int16))
    # the fp32 oracle on the last 40 query rows of two sequences (full causal context)
    qo = oracle.apply_rope(pos.cpu(), oracle.rms_norm(qkv[:, : hq * 128].cpu().view(T, hq, 128), qw.cpu(), 1e-6), table.cpu())
    kc_l, vc_l = to_logical(kc.cpu(), bs, False), to_logical(vc.cpu(), bs, True)
    for s_i in (0, n_seqs - 1):
        rows = slice(s_i * L + L - 40, s_i * L + L)
        args = (bt[s_i:s_i + 1].cpu(), torch.tensor([0, 40], dtype=torch.int32), torch.tensor([L], dtype=torch.int32))
        want = oracle.paged_attention_prefill(qo[rows], kc_l, vc_l, *args, keep_fp32=True)
        want_absv = oracle.paged_attention_prefill(qo[rows], kc_l, vc_l.abs(), *args, keep_fp32=True)
        err = (late[rows].cpu().float() - want).abs()
        assert bool((err <= _prefill_bound(want, want_absv, 2 ** -8)).all()), err.max().item()   # P as one bf16
        err = (split_new[rows].cpu().float() - want).abs()
        assert bool((err <= _prefill_bound(want, want_absv, 0.0)).all()), err.max().item()       # P as hi + lo


@pytest.mark.parametrize("hq,hkv,with_norm", [(16, 8, True), (8, 1, True), (4, 4, False), (16, 1, True)])
def test_prefill_attention_with_q_prepared_in_the_kernel(ops, hq, hkv, with_norm):
    """mi_paged_attn_prefill_fused (q-norm + RoPE inside the Q-operand load, K / V-only store before it) against
    the two-launch sequence mi_qknorm_rope_store -> mi_paged_attn_prefill on the same packed qkv rows: identical
    bits (the in-kernel norm associates its sums like the 8-lane kernel), ragged lengths, one sequence with a
    cached prefix (queries start behind it: positions != row index), caches compared too."""
    gen = torch.Generator().manual_seed(hq * 3 + hkv)
    bs = 16
    q_lens = [64, 7, 129, 33, 260, 80]
    prefix = [0, 0, 64, 0, 0, 16]  # tokens already in the cache (whole blocks), prefix-aware prefill
    kv_lens = [a + b for a, b in zip(q_lens, prefix)]
    T, n_seqs = sum(q_lens), len(q_lens)
    nb = [-(-n // bs) for n in kv_lens]
    perm = torch.randperm(sum(nb) + 3, generator=gen)[: sum(nb)].to(torch.int32)
    bt = torch.full((n_seqs, max(nb)), -1, dtype=torch.int32)
    o = 0
    for i, n in enumerate(nb):
        bt[i, :n] = perm[o:o + n]
        o += n
    pos = torch.cat([torch.arange(p, p + n) for p, n in zip(prefix, q_lens)]).to(torch.int64)
    slots = torch.cat([bt[i, (torch.arange(p, p + n) // bs)].to(torch.int64) * bs + torch.arange(p, p + n) % bs
                       for i, (p, n) in enumerate(zip(prefix, q_lens))]).to(torch.int32)
    qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=gen) * 1.5).bfloat16().to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=gen)).bfloat16().to(DEV) if with_norm else None
    kw = (1 + 0.2 * torch.randn(128, generator=gen)).bfloat16().to(DEV) if with_norm else None
    table = oracle.build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    shape = ops.kv_cache_shape(sum(nb) + 3, hkv, bs)
    base_k = torch.randn(shape, generator=gen).bfloat16().to(DEV)  # the cached prefixes' (random) contents
    base_v = torch.randn(shape, generator=gen).bfloat16().to(DEV)
    cu = torch.tensor([0] + list(np.cumsum(q_lens)), dtype=torch.int32).to(DEV)
    kvl = torch.tensor(kv_lens, dtype=torch.int32).to(DEV)
    posd, slotsd, btd = pos.to(DEV), slots.to(DEV), bt.to(DEV)
    scale = 1.0 / math.sqrt(128)
    kc1, vc1 = base_k.clone(), base_v.clone()
    q1 = ops.qknorm_rope_store(qkv, qw, kw, 1e-6, posd, table, kc1, vc1, slotsd, hq, hkv, bs)
    want = ops.paged_attn_prefill(q1, kc1, vc1, btd, cu, kvl, max(q_lens), hq, hkv, bs, scale)
    kc2, vc2 = base_k.clone(), base_v.clone()
    assert ops.qknorm_rope_store(qkv, qw, kw, 1e-6, posd, table, kc2, vc2, slotsd, hq, hkv, bs, store_q=False) is None
    got = ops.paged_attn_prefill_fused(qkv, qw, 1e-6, posd, table, kc2, vc2, btd, cu, kvl, max(q_lens), hq, hkv, bs,
                                       scale)
    assert torch.equal(kc1.view(torch.int16), kc2.view(torch.int16))
    assert torch.equal(vc1.view(torch.int16), vc2.view(torch.int16))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert float(got.float().abs().max()) > 0.1


def test_stage_copy_between_pinned_host_and_device(ops):
    """mi_stage_copy: a step's metadata upload / token download as a kernel - bytes in, bytes out, both directions,
    sizes from one 16-byte piece to a full-house prefill step's staging buffer; refuses pageable memory and odd sizes."""
    g = torch.Generator().manual_seed(3)
    for n in (16, 256, 33008, 1 << 20):
        host = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8).pin_memory()
        dev = torch.zeros(n, dtype=torch.uint8, device=DEV)
        ops.stage_copy(dev, host)
        assert torch.equal(dev.cpu(), host)
        back = torch.zeros(n, dtype=torch.uint8).pin_memory()
        ops.stage_copy(back, dev)
        torch.cuda.synchronize()
        assert torch.equal(back, host)
    toks = torch.arange(32, dtype=torch.int64, device=DEV) * 7
    landing = torch.zeros(32, dtype=torch.int64).pin_memory()
    ops.stage_copy(landing, toks)
    torch.cuda.synchronize()
    assert torch.equal(landing, toks.cpu())
    with pytest.raises(AssertionError):
        ops.stage_copy(torch.zeros(16, dtype=torch.uint8, device=DEV), torch.zeros(16, dtype=torch.uint8))  # pageable
    with pytest.raises(RuntimeError):
        ops.stage_copy(torch.zeros(24, dtype=torch.uint8, device=DEV), torch.zeros(24, dtype=torch.uint8).pin_memory())


# --------------------------------------------------------------------------- gathers / sampling
def test_embedding_and_last_token(ops):
    g = torch.Generator().manual_seed(2)
    w = torch.randn(1000, 1024, generator=g).bfloat16()
    ids = torch.randint(0, 1000, (57,), generator=g)
    out = ops.embedding(ids.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(out.view(torch.int16), oracle.embedding(ids, w).view(torch.int16))
    # vocab-parallel shard: rows outside [250, 500) are zero (embed_head.py:36-40)
    shard = w[250:500].contiguous()
    out = ops.embedding(ids.to(DEV), shard.to(DEV), vocab_start=250).cpu()
    want = oracle.embedding(ids, shard, 250)
    own = (ids >= 250) & (ids < 500)
    assert torch.equal(out[own].view(torch.int16), want[own].view(torch.int16))
    # rows of other ranks are zero; the reference's mask-multiply yields -0.0 for negative entries
    # (embed_head.py:40), the kernel writes +0.0 - identical after the all-reduce sum that follows
    assert bool((out[~own].float() == 0).all()) and bool((want[~own].float() == 0).all())
    x = torch.randn(57, 1024, generator=g).bfloat16()
    cu = torch.tensor([0, 5, 6, 40, 57], dtype=torch.int32)
    last = ops.gather_last_tokens(x.to(DEV), cu.to(DEV)).cpu()
    assert torch.equal(last.view(torch.int16), x[(cu[1:] - 1).long()].view(torch.int16))


def test_argmax_lowest_index_of_max(ops):
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(32, 151936, generator=g).bfloat16()
    logits[3, 777] = 50.0
    logits[3, 90000] = 50.0  # tie -> lowest index
    logits[4, 151935] = 60.0  # last column
    got = ops.argmax(logits.to(DEV)).cpu()
    want = torch.tensor([int(torch.nonzero(r == r.max())[0]) for r in logits.float()])
    assert torch.equal(got, want)
    # padded logits rows (graph mode): only the first `rows` are sampled (sampler.py:10-12)
    temps = torch.zeros(5)
    got5 = ops.sample(logits.to(DEV), temps.to(DEV), seed=1, step=0).cpu()
    assert torch.equal(got5, want[:5])


def test_sample_distribution(ops, golden_layers):
    """Gumbel-max draws follow softmax(logits/T) (sampler.py:13-16): chi-square on the
    reference's own probabilities for its sampler fixture."""
    g = golden_layers
    logits = bf(g["samp_logits"])[:, :64].contiguous()
    temps = torch.from_numpy(g["samp_temps"])
    probs = torch.softmax(logits.float() / temps.unsqueeze(-1), dim=-1)
    n = 4000
    counts = torch.zeros(3, 64)
    ld, td = logits.to(DEV), temps.to(DEV)
    for step in range(n):
        t = ops.sample(ld, td, seed=123, step=step).cpu()
        counts[torch.arange(3), t] += 1
    exp = probs * n
    mask = exp >= 5
    chi2 = (((counts - exp) ** 2 / exp) * mask).sum(dim=1)
    dof = mask.sum(dim=1) - 1
    assert bool((chi2 < dof + 5 * torch.sqrt(2.0 * dof)).all()), (chi2, dof)
    # determinism: same (seed, step) -> same draw; different step -> different stream
    a = ops.sample(ld, td, seed=9, step=5).cpu()
    b = ops.sample(ld, td, seed=9, step=5).cpu()
    assert torch.equal(a, b)


# (the persistent head kernel takes vocabulary-sized N at K = 1024 up to 64 rows, K = 896 up to 48, K = 2048 up to 32:
#  Qwen3-0.6B at bs 64 = BASELINE.json configs[4], Qwen2-0.5B, Llama-3.2-1B)
@pytest.mark.parametrize("M,N,K,fp8", [(32, 151936, 1024, False), (5, 4096, 5120, False), (17, 256, 128, False),
                                        (64, 32768, 1024, False), (32, 151936, 1024, True), (64, 151936, 1024, False),
                                        (33, 65536, 1024, False), (48, 151936, 896, False), (7, 65536, 896, False),
                                        (32, 128256, 2048, False), (16, 65536, 2048, False)])
def test_head_gemm_pick_equals_sampler_over_logits(ops, M, N, K, fp8):
    """The head GEMM's pick epilogue + mi_pick_final choose exactly the tokens mi_sample / mi_argmax choose from
    the logits the same GEMM writes (greedy rows, sampled rows, ties, the last column), and those logits are
    bit-identical to the plain packed GEMM's."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    w[N - 1] = w[3]  # a tie between column 3 and the last column for every row
    w = w.to(DEV)
    wp = ops.pack_weight_fp8(w) if fp8 else ops.pack_weight(w)
    plain = ops.gemm_packed(x, wp)
    temps = torch.zeros(M)
    temps[1::2] = torch.linspace(0.3, 1.7, len(temps[1::2]))  # odd rows sampled, even rows greedy
    temps = temps.to(DEV)
    for seed, step in ((0, 1), (2**63 + 12345, 77), (9, 2**40)):
        rng = torch.tensor([seed - 2**64 if seed >= 2**63 else seed, step], dtype=torch.int64, device=DEV)
        out = torch.full((M,), -1, dtype=torch.int64, device=DEV)
        logits, tokens = ops.gemm_packed_pick(x, wp, temps, rng, out)
        assert torch.equal(logits, plain)
        want = ops.sample(plain, temps, seed=seed, step=step)
        assert torch.equal(tokens, want), (tokens.cpu(), want.cpu())
    rng = torch.tensor([5, 6], dtype=torch.int64, device=DEV)
    _, greedy = ops.gemm_packed_pick(x, wp, None, rng, torch.empty(M, dtype=torch.int64, device=DEV))
    assert torch.equal(greedy, ops.argmax(plain))
    # rows whose maximum sits on the duplicated columns take the lower index
    boost = x.clone()
    boost[0] = (w[3].float() * 40).bfloat16()
    lg, tk = ops.gemm_packed_pick(boost, wp, None, rng, torch.empty(M, dtype=torch.int64, device=DEV))
    assert int(tk[0]) == int(ops.argmax(lg)[0])
    assert lg[0, 3] == lg[0, N - 1]


# --------------------------------------------------------------------------- hipGraph capture
def test_kernels_capture_into_hipgraph(ops):
    gen = torch.Generator().manual_seed(8)
    ctx_lens = [40, 41, 0, 17]
    q, kc_l, vc_l, bt = _random_paged_case(gen, 16, 8, 16, ctx_lens)
    kc, vc = to_fragment(kc_l, False).to(DEV), to_fragment(vc_l, True).to(DEV)
    qd, btd = q.to(DEV), bt.to(DEV)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32, device=DEV)
    w = (torch.randn(1024, 2048, generator=gen) * 0.05).bfloat16().to(DEV)
    nw = torch.ones(1024, dtype=torch.bfloat16, device=DEV)
    out = torch.empty(4, 2048, dtype=torch.bfloat16, device=DEV)
    ws = ops.attn_workspace(torch.device(DEV), 4, 16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        def body():
            ops.paged_attn_decode(qd, kc, vc, btd, ctx, 16, 8, 16, 128 ** -0.5, out=out, workspace=ws)
            return ops.rmsnorm(ops.gemm_skinny(out, w), nw, 1e-6)
        eager = body().clone()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = body()
    new_ctx = torch.tensor([33, 8, 0, 17], dtype=torch.int32)  # new metadata (within the tables), same graph
    ctx.copy_(new_ctx)
    graph.replay()
    torch.cuda.synchronize()
    want = oracle.paged_attention_decode(q, kc_l, vc_l, bt, new_ctx)
    want = oracle.rms_norm(oracle.linear(want, w.cpu()), nw.cpu(), 1e-6)
    assert_bf16_close(res, want, max_ulp=2, max_frac=5e-2, atol=1e-3)
    assert eager.shape == res.shape


# --------------------------------------------------------------------------- mixture of experts
def _assert_moe_close(got, want):
    """An output is a chain of top_k bf16 additions (`index_add_` on a bf16 tensor): every partial sum is rounded
    at ITS magnitude, so a one-ulp difference in one expert's contribution moves the result by an ulp of the
    partial sums, which for a token can be as large as its largest output.  Bound: 2 bf16 ulps of the element,
    or 2 ulps at the scale of the token's largest output; at most 5 % of the elements off at all."""
    g, w = got.cpu().float(), want.float()
    row_ulp = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=1, keepdim=True).clamp_min(1e-6))) - 7)
    bad = (ulp_diff(got, want) > 2) & ((g - w).abs() > 2 * row_ulp)
    assert not bool(bad.any()), f"{int(bad.sum())} elements off, worst {float(((g - w).abs() / row_ulp)[bad].max()):.1f} row-ulps"
    assert float((ulp_diff(got, want) > 0).float().mean()) <= 5e-2


def _moe_case(ops, x, gate_w, gu, dn, top_k):
    """the five MoE launches behind the router GEMM vs oracle.moe_block (restatement of qwen3_moe.py:150-185)"""
    from oracle import layers as L

    want, w_o, ids_o = L.moe_block(x, gate_w, gu, dn, top_k, return_routing=True)
    logits = oracle.linear(x, gate_w)  # the router GEMM itself is mi_gemm_bf16_* (tested above)
    gup, dnp = ops.pack_expert_weights(gu.to(DEV)), ops.pack_expert_weights(dn.to(DEV))
    got, ids, w = ops.moe_forward(x.to(DEV), logits.to(DEV), gup, dnp, top_k)
    ids, w = ids.cpu().long(), w.cpu()
    # routing: every token's picks ascending by expert id; same experts as the oracle, weights to one bf16 ulp
    assert bool((ids[:, 1:] > ids[:, :-1]).all())
    order = ids_o.argsort(-1)
    assert torch.equal(ids, torch.gather(ids_o, 1, order))
    assert int(ulp_diff(w, torch.gather(w_o, 1, order)).max()) <= 1
    _assert_moe_close(got, want)


def test_moe_block_golden(ops, golden_moe_block):
    """vs the reference's own Qwen3MoeSparseMoeBlock run (tests/golden/moe_block.npz) and vs the oracle"""
    g = golden_moe_block
    for tag in ("small", "wide"):
        T, H, E, K, I = (int(v) for v in g[f"{tag}_meta"])
        x, gw = bf(g[f"{tag}_x"]), bf(g[f"{tag}_gate_w"])
        gu, dn = bf(g[f"{tag}_gate_up_w"]).view(E, 2 * I, H).contiguous(), bf(g[f"{tag}_down_w"]).view(E, H, I).contiguous()
        _moe_case(ops, x, gw, gu, dn, K)
        logits = oracle.linear(x, gw)
        got, ids, _ = ops.moe_forward(x.to(DEV), logits.to(DEV), ops.pack_expert_weights(gu.to(DEV)),
                                      ops.pack_expert_weights(dn.to(DEV)), K)
        assert torch.equal(ids.cpu().long(), torch.from_numpy(g[f"{tag}_topk_ids"]).sort(-1).values)
        _assert_moe_close(got, bf(g[f"{tag}_y"]))


@pytest.mark.parametrize("T", [1, 32, 64, 100])
@pytest.mark.parametrize("inter", [768, 192])
def test_moe_block_qwen3_30b_a3b_shapes(ops, T, inter):
    """BASELINE.json configs[3]: hidden 2048, 128 experts, top-8, moe_intermediate 768 (TP=4 shard: 192);
    decode-sized batches and a prefill-sized one (several 16-pair passes per expert)."""
    gen = torch.Generator().manual_seed(T + inter)
    H, E, K = 2048, 128, 8
    x = torch.randn(T, H, generator=gen).bfloat16()
    gate_w = (torch.randn(E, H, generator=gen) * 0.1).bfloat16()
    gu = (torch.randn(E, 2 * inter, H, generator=gen) * 0.03).bfloat16()
    dn = (torch.randn(E, H, inter, generator=gen) * 0.03).bfloat16()
    _moe_case(ops, x, gate_w, gu, dn, K)


@pytest.mark.parametrize("T,world", [(32, 2), pytest.param(100, 4, marks=pytest.mark.gpu_slow)])
def test_moe_block_expert_shards_sum_to_the_whole(ops, T, world):
    """Tensor-parallel experts (qwen3_moe.py:100-128: every rank holds 1/world of every expert's intermediate
    width): the ranks' mi_moe_down outputs are summed row by row, so they must have the SAME row layout on every
    rank whatever order each rank's mi_moe_sort produced.  Emulated in one process: one moe_forward per shard, the
    all-reduce hook adds the shards' y (bf16 adds in rank order, as the exchange kernel); result vs the oracle
    block on the sharded arithmetic (each rank's partial output rounded to bf16 before the sum)."""
    from oracle import layers as L

    gen = torch.Generator().manual_seed(T)
    H, E, K, inter = 2048, 128, 8, 768
    x = torch.randn(T, H, generator=gen).bfloat16()
    gate_w = (torch.randn(E, H, generator=gen) * 0.1).bfloat16()
    gu = (torch.randn(E, 2 * inter, H, generator=gen) * 0.03).bfloat16()
    dn = (torch.randn(E, H, inter, generator=gen) * 0.03).bfloat16()
    logits = oracle.linear(x, gate_w)
    il = inter // world
    ys = []

    def hook(y):
        ys.append(y.clone())
        total = ys[0].float()
        for part in ys[1:]:
            total = (total + part.float()).bfloat16().float()
        return total.bfloat16()

    out = None
    for r in range(world):
        g_r = torch.cat([gu[:, r * il:(r + 1) * il], gu[:, inter + r * il:inter + (r + 1) * il]], 1).contiguous()
        d_r = dn[:, :, r * il:(r + 1) * il].contiguous()
        out, ids, w = ops.moe_forward(x.to(DEV), logits.to(DEV), ops.pack_expert_weights(g_r.to(DEV)),
                                      ops.pack_expert_weights(d_r.to(DEV)), K, all_reduce=hook)
    # the oracle on the same sharded arithmetic: per expert and rank act_r @ W_down_r^T rounded to bf16, summed
    ids_c, w_c = ids.cpu().long(), w.cpu()
    want = torch.zeros(T, H, dtype=torch.bfloat16)
    for t in range(T):
        for j in range(K):
            e = int(ids_c[t, j])
            y = None
            for r in range(world):
                g_r = torch.cat([gu[e, r * il:(r + 1) * il], gu[e, inter + r * il:inter + (r + 1) * il]], 0)
                act = L.silu_and_mul(oracle.linear(x[t:t + 1], g_r))
                part = oracle.linear(act, dn[e][:, r * il:(r + 1) * il])
                y = part if y is None else (y.float() + part.float()).bfloat16()
            want[t] = (want[t].float() + (y[0].float() * w_c[t, j].float()).bfloat16().float()).bfloat16()
    _assert_moe_close(out, want)


def test_moe_route_ties_and_uniform_logits(ops):
    """equal router logits: the lower expert ids win (the restatement's tie rule), weights are exactly 1/k"""
    T, E, K, H, I = 5, 64, 4, 128, 64
    logits = torch.zeros(T, E).bfloat16()
    logits[1, 40] = 3.0
    x = torch.randn(T, H).bfloat16()
    gu = (torch.randn(E, 2 * I, H) * 0.05).bfloat16()
    dn = (torch.randn(E, H, I) * 0.05).bfloat16()
    _, ids, w = ops.moe_forward(x.to(DEV), logits.to(DEV), ops.pack_expert_weights(gu.to(DEV)),
                                ops.pack_expert_weights(dn.to(DEV)), K)
    assert ids[0].tolist() == [0, 1, 2, 3] and ids[1].tolist() == [0, 1, 2, 40]
    assert torch.equal(w[0].cpu().float(), torch.full((K,), 0.25))


def test_moe_route_non_finite_logits_still_pick_distinct_valid_experts(ops):
    """A NaN / Inf in a token's router logits (a fault upstream) must not leave an expert id unwritten or out of range -
    mi_moe_sort's counters and the grouped GEMMs index by it.  Such a token gets k distinct valid experts and NaN
    weights (its output is NaN, as the reference's would be); the other tokens are untouched."""
    T, E, K, H, I = 70, 8, 2, 128, 64
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(T, E, generator=g).bfloat16()
    clean = logits.clone()
    logits[3] = float("nan")
    logits[17, 5] = float("inf")
    logits[40, 2] = float("nan")
    x = torch.randn(T, H, generator=g).bfloat16()
    gu = (torch.randn(E, 2 * I, H, generator=g) * 0.05).bfloat16()
    dn = (torch.randn(E, H, I, generator=g) * 0.05).bfloat16()
    packed = ops.pack_expert_weights(gu.to(DEV)), ops.pack_expert_weights(dn.to(DEV))
    out, ids, w = ops.moe_forward(x.to(DEV), logits.to(DEV), *packed, K)
    ref, rids, rw = ops.moe_forward(x.to(DEV), clean.to(DEV), *packed, K)
    torch.cuda.synchronize()
    ids = ids.cpu()
    assert int(ids.min()) >= 0 and int(ids.max()) < E
    assert all(len(set(row)) == K for row in ids.tolist())
    good = [t for t in range(T) if t not in (3, 17, 40)]
    assert torch.equal(ids[good], rids.cpu()[good]) and torch.equal(w.cpu()[good], rw.cpu()[good])
    assert torch.equal(out.cpu()[good].view(torch.int16), ref.cpu()[good].view(torch.int16))
    assert bool(torch.isfinite(out.cpu()[good].float()).all())
