"""Tensor-parallel linear layers (reference: nanovllm/layers/linear.py) — same class
names, constructor arguments and per-parameter `weight_loader` sharding rules.

forward(): decode-sized activations (ops.prefers_tile: at most 128 rows, up to 512 for the narrow
projections) go through the hand-written weight-streaming MFMA kernels (mi_gemm_bf16_packed / _skinny,
the rows in chunks of 64); everything larger - every prefill projection - through the MFMA tile
kernels of mi_gemm_bf16 (csrc/gemm_tile.hip).  No library GEMM anywhere on the path.
"""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import ops
from nanovllm.layers.parallel import all_reduce_sum, divide, tp_rank, tp_size


def tile_gemm_takes(rows: int, n: int, k: int) -> bool:
    """mi_gemm_bf16's shape contract (csrc/gemm_tile.hip: K steps of 64, 16-byte output pieces, 32-bit DMA offsets),
    asked of the library (mi_gemm_bf16_max_rows) instead of restated here (the restatement had drifted).
    The row count does not matter as long as one whole tile row fits: ops.gemm_tile walks longer activations through the
    kernel in whole-tile row pieces."""
    return ops.tile_gemm_max_rows(n, k) >= min(max(rows, 1), 256)


def streaming_gemm_takes(n: int, k: int) -> bool:
    """the weight-streaming kernels' contract (16-feature row tiles, 32-deep MFMA steps)"""
    return n % 16 == 0 and k % 32 == 0


def check_linear_shape(name: str, n: int, k: int) -> None:
    """Start-up validation (ModelRunner): a projection no kernel family takes must fail when the model is built, not
    in the middle of a prefill step."""
    if not (streaming_gemm_takes(n, k) or tile_gemm_takes(1, n, k)):
        raise NotImplementedError(
            f"{name}: a [{n} x {k}] projection fits neither the streaming GEMMs (N % 16, K % 32) nor the tile GEMM "
            "(N % 4, K % 64); there is no library fallback in this build")


def linear_forward(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None,
                   packed: torch.Tensor | None = None) -> torch.Tensor:
    rows = x.numel() // x.shape[-1]
    n, k = weight.shape
    tile_ok = tile_gemm_takes(rows, n, k)
    tileable = tile_ok and not isinstance(packed, ops.Fp8Weight)

    def streamed(xr: torch.Tensor) -> torch.Tensor:
        if isinstance(packed, ops.Fp8Weight):  # the fp8 GEMM has no bias operand
            y = ops.gemm_packed(xr, packed)
            return y if bias is None else y.add_(bias)
        if packed is not None:
            return ops.gemm_packed(xr, packed, bias)
        return ops.gemm_skinny(xr, weight, bias)

    if (rows <= ops.SKINNY_MAX_M and streaming_gemm_takes(n, k) and not (tileable and ops.prefers_tile(rows, n))):
        return streamed(x)
    shape = x.shape
    if tile_ok:
        y = ops.gemm_tile(x.reshape(-1, shape[-1]), weight, bias)
        return y.view(*shape[:-1], n)
    if streaming_gemm_takes(n, k):
        # more rows than one streaming launch takes AND a shape the tile kernel cannot take (K % 64 == 32: e.g. an
        # intermediate size of 11008 over 8 ranks = 1376; an operand of 4 GiB or more): the streaming kernels over the
        # rows in pieces of SKINNY_MAX_M - every piece re-streams the weight, correct and slow, instead of an error in
        # the middle of a prefill step (the reference's F.linear takes any shape)
        x2 = x.reshape(-1, shape[-1])
        y = torch.empty((rows, n), dtype=x.dtype, device=x.device)
        for r0 in range(0, rows, ops.SKINNY_MAX_M):
            y[r0:r0 + ops.SKINNY_MAX_M] = streamed(x2[r0:r0 + ops.SKINNY_MAX_M].contiguous())
        return y.view(*shape[:-1], n)
    raise ops._C.MiError(f"no GEMM kernel takes a [{n} x {k}] projection (see layers/linear.check_linear_shape)")


def can_pack(weight: torch.Tensor) -> bool:
    return weight.is_cuda and weight.dim() == 2 and weight.shape[0] % 16 == 0 and weight.shape[1] % 32 == 0


_QUANTIZATION: str | None = None  # set by the runner from Config.quantization before weights are packed


def set_weight_quantization(mode: str | None) -> None:
    assert mode in (None, "fp8"), mode
    global _QUANTIZATION
    _QUANTIZATION = mode


def pack_for_decode(weight: torch.Tensor, previous):
    """The decode-GEMM copy of a (sharded) weight: fragment-native bf16, or - Config.quantization == "fp8" -
    fragment-native e4m3 + per-row scale.  In fp8 mode the bf16 parameter itself is replaced by the
    dequantised values, so the prefill (the bf16 tile GEMM) and anything tied to it see the same model."""
    if not can_pack(weight):
        return None
    if _QUANTIZATION == "fp8" and weight.shape[1] % 64 == 0:
        packed = ops.pack_weight_fp8(weight, previous if isinstance(previous, ops.Fp8Weight) else None)
        q, scale = ops.quantize_fp8(weight)
        weight.copy_(ops.dequantize_fp8(q, scale).to(weight.dtype))
        return packed
    return ops.pack_weight(weight, previous if isinstance(previous, torch.Tensor) else None)


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank = tp_rank()
        self.tp_size = tp_size()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)
        self.weight_packed: torch.Tensor | None = None
        self.weight_rows4: torch.Tensor | None = None  # row-parallel layers: mi_gemm_bf16_rows4 layout

    def pack(self) -> None:
        """Build the fragment-native copy of the (already sharded) weight that the decode GEMMs
        stream (mi_pack_weight); call again after the weight changes."""
        self.weight_packed = pack_for_decode(self.weight.data, self.weight_packed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return linear_forward(x, self.weight, self.bias, self.weight_packed)


class ColumnParallelLinear(LinearBase):
    """Output features sharded over ranks (linear.py:54-73)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, divide(output_size, tp_size()), bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        rows = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * rows, rows))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return linear_forward(x, self.weight, self.bias, self.weight_packed)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices stacked along dim 0, e.g. gate|up (linear.py:76-93)."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class QKVParallelLinear(ColumnParallelLinear):
    """Packed q|k|v projection, heads sharded over ranks (linear.py:96-128)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp_size())
        self.num_kv_heads = divide(total_num_kv_heads, tp_size())
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        assert loaded_shard_id in ("q", "k", "v")
        q_rows, kv_rows = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class RowParallelLinear(LinearBase):
    """Input features sharded; partial sums all-reduced (linear.py:131-153)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(divide(input_size, tp_size()), output_size, bias, 1)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        cols = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * cols, cols))

    def pack(self) -> None:
        super().pack()
        # tensor parallelism, bf16 weights: a second decode layout whose GEMM returns this rank's partial
        # sums as complete bf16 rows from N/4 workgroups (mi_gemm_bf16_rows4)
        w = self.weight.data
        # complete rows from four-feature workgroups pay off while N is too small to fill the CUs with
        # 16-feature row tiles (N = 1024: 64 tiles); at N = 5120 the packed kernel is 1.6-2.2x faster
        if self.tp_size > 1 and w.is_cuda and not isinstance(self.weight_packed, ops.Fp8Weight) \
                and w.shape[0] % 4 == 0 and w.shape[1] % 32 == 0 and w.shape[0] // 16 < 256:
            self.weight_rows4 = ops.pack_weight_rows4(w, self.weight_rows4)
        else:
            self.weight_rows4 = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = linear_forward(x, self.weight, self.bias if self.tp_rank == 0 else None, self.weight_packed)
        return all_reduce_sum(y)
