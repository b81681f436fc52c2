"""The measured-and-lost experiments (include/mi355_nanovllm_experiments.h): they are not in the default library
(`make EXPERIMENTS=1` builds them), so these tests skip unless that build is the one loaded."""
import pytest
import torch

from nanovllm import _C

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _C.HAS_EXPERIMENTS, reason="library built without -DMI_EXPERIMENTS")]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def experiments():
    from nanovllm import experiments as _e

    return _e


def test_mlp_half_fused_equals_the_three_launches(ops, experiments, rows):
    """csrc/mlp_half.hip (experiment): one persistent launch with in-launch hand-offs == add+RMSNorm over split-K
    partials -> gate_up GEMM + SwiGLU -> down GEMM split-K, bit for bit, also when replayed (the counters re-arm
    themselves) and from a captured graph"""
    g = torch.Generator().manual_seed(100 + rows)
    H, inter = 1024, 3072
    gu = ops.pack_weight((torch.randn(2 * inter, H, generator=g) * 0.05).bfloat16().to(DEV))
    dn = ops.pack_weight((torch.randn(H, inter, generator=g) * 0.05).bfloat16().to(DEV))
    nw = (1 + 0.1 * torch.randn(H, generator=g)).bfloat16().to(DEV)
    sync = torch.zeros(8, dtype=torch.int32, device=DEV)
    for it in range(3):
        parts = torch.randn(4, rows, H, generator=g).to(DEV)
        r = torch.randn(rows, H, generator=g).bfloat16().to(DEV)
        xn, r_ref = ops.add_rmsnorm_splitk(parts, r, nw, 1e-6)
        want = ops.gemm_packed_splitk(ops.gemm_packed(xn, gu, silu_mul=True), dn, 4)
        got, r_got = experiments.mlp_half_fused(parts, r, nw, 1e-6, gu, dn, sync)
        torch.cuda.synchronize()
        assert int(sync[6]) == 0, "a hand-off timed out"
        assert torch.equal(r_got.view(torch.int16), r_ref.view(torch.int16)), it
        assert torch.equal(got, want), it
    scratch = (torch.empty_like(r), torch.empty(rows, H, dtype=torch.bfloat16, device=DEV),
               torch.empty(rows, inter, dtype=torch.bfloat16, device=DEV), torch.empty(4, rows, H, device=DEV))
    experiments.mlp_half_fused(parts, r, nw, 1e-6, gu, dn, sync, scratch)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(3):
            experiments.mlp_half_fused(parts, r, nw, 1e-6, gu, dn, sync, scratch)
    scratch[3].zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert int(sync[6]) == 0 and torch.equal(scratch[3], want)


@pytest.mark.parametrize("rows", [1, 7, 32, 64])
def test_add_rmsnorm_splitk_warm_is_the_same_norm(ops, experiments, rows):
    """mi_add_rmsnorm_splitk_warm: the idle CUs of the norm launch touch the next GEMMs' weights (tile counts that
    are and are not multiples of 8, one and two regions); the norm's outputs are those of the plain launch, bit for
    bit, and the weights are untouched"""
    g = torch.Generator().manual_seed(rows)
    parts = torch.randn(4, rows, 1024, generator=g).to(DEV)
    r = torch.randn(rows, 1024, generator=g).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(1024, generator=g)).bfloat16().to(DEV)
    wa = torch.randn(6144, 1024, generator=g).bfloat16().to(DEV)
    wb = torch.randn(16 * 13, 3072, generator=g).bfloat16().to(DEV)
    wa0, wb0 = wa.clone(), wb.clone()
    y0, r0 = ops.add_rmsnorm_splitk(parts, r, nw, 1e-6)
    for warm in ([wa], [wa, wb], [wb]):
        y, r2 = experiments.add_rmsnorm_splitk_warm(parts, r, nw, 1e-6, warm)
        assert torch.equal(y.view(torch.int16), y0.view(torch.int16)) and torch.equal(r2.view(torch.int16), r0.view(torch.int16))
    torch.cuda.synchronize()
    assert torch.equal(wa, wa0) and torch.equal(wb, wb0)
