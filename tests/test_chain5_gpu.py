"""The decode chain in five launches per layer (csrc/gemm_chain5_kernel.hpp) against the seven-launch
chain it replaces and against the CPU oracle.

The reference's operator sequence is RowParallelLinear (linear.py:149-153) -> RMSNorm.add_rms_forward
(layernorm.py:27-38) -> Column/QKV/MergedColumnParallelLinear (linear.py:72-73).  Five launches: the row-parallel
projection adds the residual and emits per-tile sums of squares (mi_gemm_bf16_rowstat), the next projection builds the
normalised operand on load (mi_gemm_bf16_normed).  At hidden 1024 every fp32 sum is taken in the seven-launch chain's
order, so the two chains must agree BIT FOR BIT; other widths are held to the oracle's bounds."""
import pytest
import torch

import oracle
from model_configs import QWEN3_0_6B, make_model_dir

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-6


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


def _bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t


def _assert_seam_close(res_got, xn_got, x, w, res, nw, max_frac=2e-2):
    """RowParallelLinear -> add_rms_forward against the oracle, with the bound of the arithmetic: y = bf16(x . w^T) may
    round to the other side of a tie than the oracle's (fp32 summation order) - ONE ulp of y, <= 2^-7 |y| - which moves
    the un-rounded sum s = y + residual by as much, however small s itself is (cancellation: many ulps of a tiny s, which
    is why a plain ulp count cannot state this bound); then one rounding of the residual output (<= 2^-8 |s|, a flipped
    tie: 2^-7), and for the normalised row dy . rstd . |w| plus its own two roundings (2 ulps: 2^-6 |xn|)."""
    y = oracle.linear(x, w).float()
    want_x, want_res = oracle.add_rms_norm(y.bfloat16(), res, nw, EPS)
    dy = y.abs() * 2.0 ** -7
    s = y.bfloat16().float() + res.float()
    rstd = torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + EPS)
    d_res = (res_got.cpu().float() - want_res.float()).abs()
    assert bool((d_res <= dy + want_res.float().abs() * 2.0 ** -7 + 1e-30).all()), float((d_res - dy).max())
    assert float((d_res > 0).float().mean()) <= max_frac
    d_x = (xn_got.cpu().float() - want_x.float()).abs()
    bound_x = dy * rstd * nw.float().abs() * 1.01 + want_x.float().abs() * 2.0 ** -6 + 1e-30
    assert bool((d_x <= bound_x).all()), float((d_x - bound_x).max())
    assert float((d_x > 0).float().mean()) <= max_frac


@pytest.mark.parametrize("M", [1, 7, 8, 9, 17, 32])
@pytest.mark.parametrize("K,ks", [(2048, 4), (3072, 4), (1024, 4), (2048, 1), (3072, 2)])
def test_rowstat_and_normed_are_the_seven_launch_chain_bit_for_bit(ops, M, K, ks):
    """o_proj / down_proj of Qwen3-0.6B (hidden 1024) with the split-K geometry of the product (4) and others:
    residual, normalised operand (never materialised: seen through the final-norm kernel and through the consumer
    GEMMs' outputs) and the consumers' outputs equal the seven-launch pieces exactly; the statistic and s are what they
    claim to be; everything within the oracle's bounds."""
    H = 1024
    g = torch.Generator().manual_seed(M * 131 + K + ks)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(H, K, generator=g) * 0.03).bfloat16()
    res = torch.randn(M, H, generator=g).bfloat16()
    nw = (1.0 + 0.1 * torch.randn(H, generator=g)).bfloat16()
    w_qkv = (torch.randn(4096, H, generator=g) * 0.03).bfloat16()
    w_gu = (torch.randn(6144, H, generator=g) * 0.03).bfloat16()
    xd, resd, nwd = x.to(DEV), res.to(DEV), nw.to(DEV)
    wp, wq, wg = (ops.pack_weight(t.to(DEV)) for t in (w, w_qkv, w_gu))

    parts = ops.gemm_packed_splitk(xd, wp, ks)
    xn7, res7 = ops.add_rmsnorm_splitk(parts, resd, nwd, EPS)
    s, res5, stat = ops.gemm_rowstat(xd, wp, resd, ks)
    # the split-K kernel runs wave slices of 64 when a split has at most sixteen of them (the product's geometries);
    # longer splits take 128-deep slices there - another fp32 grouping of the same sum, held to the oracle below
    same_sum = K // ks // 64 <= 16
    if same_sum:
        assert torch.equal(_bits(res5), _bits(res7))
        y = parts[0].clone()
        for i in range(1, ks):
            y += parts[i]
        assert torch.equal(s, y.bfloat16().float() + resd.float())  # the un-rounded sum the reference normalises
    assert torch.equal(res5, s.bfloat16())
    torch.testing.assert_close(stat.sum(-1).cpu(), s.cpu().double().pow(2).sum(-1).float(), rtol=1e-5, atol=0)
    xn5 = ops.norm_from_stat(s, stat, nwd, EPS)
    if same_sum:
        assert torch.equal(_bits(xn5), _bits(xn7))
    # the consumers: the same GEMM on the operand that is never written to memory
    assert torch.equal(_bits(ops.gemm_normed(s, stat, nwd, EPS, wq)), _bits(ops.gemm_packed(xn5, wq)))
    assert torch.equal(_bits(ops.gemm_normed(s, stat, nwd, EPS, wg, silu_mul=True)),
                       _bits(ops.gemm_packed(xn5, wg, silu_mul=True)))
    # ... and the oracle: RowParallelLinear -> add_rms_forward
    _assert_seam_close(res5, xn5, x, w, res, nw)


@pytest.mark.parametrize("M,H,K", [(32, 2048, 8192), (5, 2048, 2048), (32, 4096, 4096), (16, 1024, 768), (32, 3072, 1024)])
def test_chain5_other_widths_against_the_oracle(ops, M, H, K):
    """Widths whose seven-launch norm kernel sums in another order (hidden 2048 ... 5120, K-slices of 12 / 10 waves): the
    five-launch pieces against the oracle, in the bounds of their seven-launch counterparts."""
    g = torch.Generator().manual_seed(M + H + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(H, K, generator=g) * 0.03).bfloat16()
    res = torch.randn(M, H, generator=g).bfloat16()
    nw = (1.0 + 0.1 * torch.randn(H, generator=g)).bfloat16()
    w2 = (torch.randn(512, H, generator=g) * 0.03).bfloat16()
    assert ops.chain5_takes(M, H, (K,))
    # (a Qwen3-32B TP-8 rank's down_proj, K = 3200 = 50 wave slices, is not a five-launch shape: the model takes seven)
    assert not ops.chain5_takes(24, 5120, (3200,)) and not ops.chain5_takes(33, 1024, (2048,))
    s, res5, stat = ops.gemm_rowstat(x.to(DEV), ops.pack_weight(w.to(DEV)), res.to(DEV), 1)
    xn = ops.norm_from_stat(s, stat, nw.to(DEV), EPS)
    _assert_seam_close(res5, xn, x, w, res, nw)
    w2p = ops.pack_weight(w2.to(DEV))
    for silu in (False, True):
        got = ops.gemm_normed(s, stat, nw.to(DEV), EPS, w2p, silu_mul=silu)
        assert torch.equal(_bits(got), _bits(ops.gemm_packed(xn, w2p, silu_mul=silu)))  # the same GEMM on the same operand


def test_instrumented_chain5_launches_compute_the_product(ops):
    """mi_gemm_bf16_rowstat_ex / mi_gemm_bf16_normed_ex (tools/chain_timeline.py): same bits, monotonic stamps."""
    g = torch.Generator().manual_seed(3)
    B, H = 32, 1024
    a = torch.randn(B, 3072, generator=g).bfloat16().to(DEV)
    res = torch.randn(B, H, generator=g).bfloat16().to(DEV)
    nw = torch.ones(H).bfloat16().to(DEV)
    w_dn = ops.pack_weight((torch.randn(H, 3072, generator=g) * 0.03).bfloat16().to(DEV))
    w_gu = ops.pack_weight((torch.randn(6144, H, generator=g) * 0.03).bfloat16().to(DEV))
    st = lambda wg, wv: torch.zeros(wg, wv, 8, dtype=torch.int64, device=DEV)  # noqa: E731
    s1, s2 = st(64 * 4, 16), st(192, 16)
    p0 = ops.gemm_rowstat(a, w_dn, res, 4)
    p1 = ops.gemm_rowstat(a, w_dn, res, 4, stamps=s1)
    assert all(torch.equal(_bits(u), _bits(v)) for u, v in zip(p0, p1))
    y0 = ops.gemm_normed(p0[0], p0[2], nw, EPS, w_gu, silu_mul=True)
    y1 = ops.gemm_normed(p0[0], p0[2], nw, EPS, w_gu, silu_mul=True, stamps=s2)
    assert torch.equal(_bits(y0), _bits(y1))
    for s in (s1, s2):
        t = s.cpu()[..., :7]
        assert (t > 0).all() and (t[..., 1:] >= t[..., :-1]).all()


def test_engine_decode_steps_agree_between_the_two_chains(monkeypatch):
    """Qwen3-0.6B at full shape, eager decode steps over the same cache state: the five-launch chain's logits are the
    seven-launch chain's logits in every bit, for a full bucket and a ragged one."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.models.qwen3 import Qwen3Model

    gen = torch.Generator().manual_seed(6)
    llm = LLM(make_model_dir(QWEN3_0_6B), kvcache_block_size=16, max_num_seqs=32, max_num_batched_tokens=4096,
              max_model_len=512, num_kvcache_blocks=600, warmup=False, synthetic_seed=0, enforce_eager=True)
    try:
        for n_seqs in (32, 5):
            for _ in range(n_seqs):
                n = int(torch.randint(20, 90, (1,), generator=gen))
                llm.add_request(torch.randint(0, 10000, (n,), generator=gen).tolist(),
                                SamplingParams(max_tokens=4, ignore_eos=True, greedy=True))
            steps = 0
            while not llm.is_finished():
                sched, is_prefill = llm.scheduler.schedule()
                if is_prefill:
                    toks = llm.model_runner.call("run", sched, True)
                else:
                    got = {}
                    for chain5 in (True, False):  # the same step twice: it rewrites the same K / V rows
                        monkeypatch.setattr(Qwen3Model, "CHAIN5", chain5)
                        toks = llm.model_runner.call("run", sched, False)
                        got[chain5] = llm.model_runner.last_logits[: len(sched)].clone()
                    assert torch.equal(_bits(got[True]), _bits(got[False]))
                    steps += 1
                llm.scheduler.postprocess(sched, toks)
            assert steps == 3
    finally:
        llm.exit()
