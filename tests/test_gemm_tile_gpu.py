"""GPU parity of the MFMA tile GEMM (mi_gemm_bf16, csrc/gemm_tile.hip) - the F.linear of every prefill-sized
activation (linear.py:51,73,150) - against the CPU oracle.  fp32 accumulation in a different order than the
oracle's: <= 1 bf16 ulp on a small fraction of the outputs, absolute floor K * 2^-22 where a dot product cancels."""
import pytest
import torch

import oracle
from test_kernels_gpu import assert_bf16_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nanovllm import ops as _ops

    return _ops


def _case(M, N, K, seed=0, wscale=0.05):
    g = torch.Generator().manual_seed(seed + M * 7 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * wscale).bfloat16()
    return g, x, w


# prefill shapes of Qwen3-0.6B (qkv, o_proj, gate_up, down), ragged token counts, N not a multiple of the tile,
# one and three K steps, the MoE router (N = 8 / 128)
@pytest.mark.parametrize("M", [65, 256, 300, 1000])
@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072), (512, 128), (100, 64),
                                 (8, 128), (128, 192), (1280, 5120)])
def test_gemm_tile_vs_oracle(ops, M, N, K):
    g, x, w = _case(M, N, K)
    b = torch.randn(N, generator=g).bfloat16()
    atol = K * 2.0 ** -22
    y = ops.gemm_tile(x.to(DEV), w.to(DEV))
    assert_bf16_close(y, oracle.linear(x, w), max_ulp=1, max_frac=2e-2, atol=atol)
    yb = ops.gemm_tile(x.to(DEV), w.to(DEV), b.to(DEV))
    assert_bf16_close(yb, oracle.linear(x, w, b), max_ulp=1, max_frac=2e-2, atol=atol)
    # these row counts go to the 128-tile kernel (possibly in K slices); the 256-tile kernel and the unsliced
    # 128-tile kernel run the same fp32 chain per output element: the same bits
    big = ops.gemm_tile(x.to(DEV), w.to(DEV), variant=0)
    for variant in (65536, 65538, 65539):  # ring of 4 / 2 / 3 stages
        mid = ops.gemm_tile(x.to(DEV), w.to(DEV), variant=variant)
        assert torch.equal(big.view(torch.int16), mid.view(torch.int16)), variant
    assert_bf16_close(big, oracle.linear(x, w), max_ulp=1, max_frac=2e-2, atol=atol)
    # transpose / fragment-layout detector: asymmetric weights, one-hot activations
    xe = torch.zeros(M, K).bfloat16()
    xe[M - 3, 5] = 1.0
    xe[1, K - 1] = 1.0
    ye = ops.gemm_tile(xe.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(ye[M - 3].view(torch.int16), w[:, 5].contiguous().view(torch.int16))
    assert torch.equal(ye[1].view(torch.int16), w[:, K - 1].contiguous().view(torch.int16))
    ye[M - 3] = 0
    ye[1] = 0
    assert int((ye.view(torch.int16) & 0x7FFF).count_nonzero()) == 0  # every other row is exactly (+-) zero


@pytest.mark.parametrize("K", [64, 128, 192, 1024])
def test_gemm_tile_persistent_ragged(ops, K):
    """more tiles than workgroups (18 x 16 = 288 tiles on 256 persistent workgroups: the DMA stream crosses tile
    seams) with ragged last tiles in both dimensions, one / two / three K steps (both slot parities at the seam)"""
    M, N = 4400, 4000
    g, x, w = _case(M, N, K, seed=21)
    b = torch.randn(N, generator=g).bfloat16()
    y = ops.gemm_tile(x.to(DEV), w.to(DEV), b.to(DEV))
    assert_bf16_close(y, oracle.linear(x, w, b), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    one_tile_each = ops.gemm_tile(x.to(DEV), w.to(DEV), variant=16)  # grid = tiles
    assert torch.equal(ops.gemm_tile(x.to(DEV), w.to(DEV)).view(torch.int16), one_tile_each.view(torch.int16))
    gu = ops.gemm_tile(x.to(DEV), w[:3840].contiguous().to(DEV), silu_mul=True)  # 15 x 18 = 270 tiles
    assert_bf16_close(gu, ops.silu_mul(ops.gemm_tile(x.to(DEV), w[:3840].contiguous().to(DEV))).cpu(), max_ulp=2,
                      max_frac=1e-3, atol=32 * K * 2.0 ** -22)


@pytest.mark.parametrize("M", [65, 256, 777])
@pytest.mark.parametrize("N,K", [(6144, 1024), (512, 128), (6400, 5120), (1536, 2048)])
def test_gemm_tile_swiglu_epilogue(ops, M, N, K):
    """fused SiluAndMul == gate_up GEMM followed by the activation (activation.py:10-12, three bf16 roundings).
    Against the tile GEMM's own plain output run through mi_silu_mul (same gate / up bits in): the epilogue's
    silu uses the hardware exp2 / reciprocal instead of libm expf + IEEE division - a few fp32 ulps before the bf16
    rounding, i.e. one bf16 ulp at rounding ties only (< 0.1 % of the elements)."""
    g, x, w = _case(M, N, K, seed=3)
    got = ops.gemm_tile(x.to(DEV), w.to(DEV), silu_mul=True)
    two_pass = ops.silu_mul(ops.gemm_tile(x.to(DEV), w.to(DEV)))
    # (the plain GEMM of few-tile shapes sums K in slices: where a gate cancels to ~0 the two fp32 sums differ by
    # K * 2^-24 * |terms| - the absolute floor of every GEMM comparison here -
    # and flip the bf16 rounding of a gate or up value in ~0.1 % of the elements at K = 5120)
    assert_bf16_close(got, two_pass.cpu(), max_ulp=3, max_frac=5e-3, atol=32 * K * 2.0 ** -22)
    atol = K * 2.0 ** -22
    # a 1-ulp flip of the gate (summation order) times |up| <= ~6: widen the near-zero floor
    assert_bf16_close(got, oracle.silu_and_mul(oracle.linear(x, w)), max_ulp=4, max_frac=3e-2, atol=32 * atol)


@pytest.mark.parametrize("M", [129, 3840, 4096, 4097])
def test_gemm_tile_kernel_choice_boundary(ops, M):
    """either side of the tile count (256 tiles of 256 x 256) at which the 128-tile kernel hands over to the large one"""
    N, K = 4096, 128
    g, x, w = _case(M, N, K, seed=13)
    b = torch.randn(N, generator=g).bfloat16()
    y = ops.gemm_tile(x.to(DEV), w.to(DEV), b.to(DEV))
    assert_bf16_close(y, oracle.linear(x, w, b), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    for variant in (0, 65536, 65538, 65539):  # large kernel; 128-tile kernel with 4 / 2 / 3 ring stages
        yv = ops.gemm_tile(x.to(DEV), w.to(DEV), variant=variant)
        assert_bf16_close(yv, oracle.linear(x, w), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    gu = ops.gemm_tile(x.to(DEV), w.to(DEV), silu_mul=True)
    assert_bf16_close(gu, oracle.silu_and_mul(oracle.linear(x, w)), max_ulp=4, max_frac=3e-2, atol=32 * K * 2.0 ** -22)


def test_gemm_tile_strided_rows(ops):
    """x and y as row-strided views (the q heads of a packed qkv row, a slice of a wider output)"""
    g, xw, w = _case(300, 1024, 2048 + 1024 + 1024, seed=5)
    x = xw[:, :2048]
    w = w[:, :2048].contiguous()
    out = torch.zeros(300, 1536, dtype=torch.bfloat16, device=DEV)
    ops.gemm_tile(xw.to(DEV)[:, :2048], w.to(DEV), out=out[:, 256:1280])
    assert_bf16_close(out[:, 256:1280].cpu(), oracle.linear(x, w), max_frac=2e-2, atol=2048 * 2.0 ** -22)
    assert float(out[:, :256].float().abs().sum()) == 0.0 and float(out[:, 1280:].float().abs().sum()) == 0.0


def test_gemm_tile_full_prefill_shape_and_variants(ops):
    """The bench's prefill step: 16384 tokens.  All schedule variants run the same MFMA chains: bit-identical
    outputs; repeated launches under a competing HBM stream stay bit-identical (race screen of the DMA pipeline)."""
    M, N, K = 16384, 4096, 1024
    g, x, w = _case(M, N, K, seed=9)
    xd, wd = x.to(DEV), w.to(DEV)
    y = ops.gemm_tile(xd, wd)
    want = oracle.linear(x[:2048], w)  # the oracle on the first 2048 rows and on the last tile's rows
    assert_bf16_close(y[:2048], want, max_frac=2e-2, atol=K * 2.0 ** -22)
    assert_bf16_close(y[-300:], oracle.linear(x[-300:], w), max_frac=2e-2, atol=K * 2.0 ** -22)
    E8 = 1 << 20  # the eight-wave kernel of rounds 2-3 (kept as a tuning variant) and its schedule flags
    for variant in (8, 16, 1 << 18, 1 << 19, (1 << 19) + (1 << 18), E8, E8 + 2, E8 + 16, E8 + 8192):  # (1 << 18: direct instead of
        # whole-line stores; 1 << 19: the kernel's 32 x 32 x 16 form)
        yv = ops.gemm_tile(xd, wd, variant=variant)
        assert torch.equal(yv.view(torch.int16), y.view(torch.int16)), variant
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for it in range(40):
        with torch.cuda.stream(side):
            junk.add_(1)  # uneven HBM pressure next to the DMA pipeline
        yv = ops.gemm_tile(xd, wd, variant=(0, 16)[it & 1])
        assert torch.equal(yv.view(torch.int16), y.view(torch.int16)), it
    torch.cuda.synchronize()
    # linearity in the rows: permuting the activation rows permutes the output rows, bit for bit
    perm = torch.randperm(M, generator=g)
    yp = ops.gemm_tile(xd[perm.to(DEV)].contiguous(), wd)
    assert torch.equal(yp.view(torch.int16), y[perm.to(DEV)].view(torch.int16))


def test_gemm_tile_rejects_what_it_cannot_do(ops):
    from nanovllm._C import MiError

    x = torch.zeros(128, 96, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(MiError):  # K % 64
        ops.gemm_tile(x, w)
    with pytest.raises(MiError):
        ops.gemm_tile(torch.zeros(128, 64).bfloat16(), torch.zeros(64, 64).bfloat16())  # CPU tensors


def test_gemm_tile_captures_into_hipgraph(ops):
    g, x, w = _case(512, 1024, 1024, seed=11)
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.empty(512, 1024, dtype=torch.bfloat16, device=DEV)
    ops.gemm_tile(xd, wd, out=out)
    want = out.clone()
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.gemm_tile(xd, wd, out=out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("packed", [False, True])
def test_linear_forward_takes_shapes_the_tile_kernel_cannot(ops, packed):
    """a prefill-sized activation on a projection with K % 64 == 32 (an intermediate size of 11008 over 8
    ranks = 1376: Llama-2-7B / Qwen1.5-7B down_proj shards) used to raise MI_EUNSUPPORTED in the middle of a prefill
    step.  layers/linear.linear_forward now walks the rows through the streaming kernels in pieces of 512; shapes no
    kernel family takes are refused when the model is built (check_linear_shape)."""
    from nanovllm.layers import linear

    M, N, K = 1100, 1024, 1376
    g, x, w = _case(M, N, K, seed=4)
    b = torch.randn(N, generator=g).bfloat16()
    assert not linear.tile_gemm_takes(M, N, K) and linear.streaming_gemm_takes(N, K)
    wd = w.to(DEV)
    y = linear.linear_forward(x.to(DEV), wd, b.to(DEV), ops.pack_weight(wd) if packed else None)
    assert_bf16_close(y, oracle.linear(x, w, b), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
    with pytest.raises(NotImplementedError):
        linear.check_linear_shape("odd", 1022, 1384)
    linear.check_linear_shape("down_proj shard", N, K)


def test_more_rows_than_one_launch_addresses_go_through_in_row_pieces(ops):
    """the tile kernel addresses its operands with 31-bit byte offsets, so (M + 256) * ldx * 2 must
    stay below 2^31 - 65 536 tokens x K = 25 600 does not.  ops.gemm_tile asks the library for the limit
    (mi_gemm_bf16_max_rows) and walks longer activations through the kernel in whole-tile row pieces; the bits are those
    of one launch.  Here the limit is reached with a wide row stride instead of a wide K: 1500 rows of a [.., 2^20]
    buffer (max 768 rows per launch)."""
    from nanovllm.layers import linear

    M, N, K, ld = 1500, 256, 128, 1 << 20
    g, x, w = _case(M, N, K, seed=9)
    big = torch.zeros((M, ld), dtype=torch.bfloat16, device=DEV)
    big[:, :K] = x.to(DEV)
    xs = big[:, :K]
    limit = ops.tile_gemm_max_rows(N, K, ld)
    assert 256 <= limit < M and linear.tile_gemm_takes(M, N, K)
    wd = w.to(DEV)
    y = ops.gemm_tile(xs, wd)
    want = ops.gemm_tile(x.to(DEV), wd)  # contiguous rows: one launch
    assert torch.equal(y.view(torch.int16), want.view(torch.int16))
    assert_bf16_close(y, oracle.linear(x, w, None), max_ulp=1, max_frac=2e-2, atol=K * 2.0 ** -22)
