"""ctypes binding of libmi355_nanovllm.so (C ABI in include/mi355_nanovllm.h).

The library is the product: there is no eager / CPU fallback anywhere in this
package.  If the shared object is missing or a call fails, we raise.

Tensors cross the boundary as ``tensor.data_ptr()`` plus sizes; every device
call is enqueued on torch's *current* HIP stream so that it composes with
torch.cuda.graph capture (hipGraph) and with RCCL collectives issued through
torch.distributed.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch  # noqa: F401  (must be imported first: shares one libamdhip64 with the extension)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "MI355_NANOVLLM_LIB",
    os.path.normpath(os.path.join(_HERE, "..", "lib", "libmi355_nanovllm.so")),
)

MI_OK = 0
MI_EUNSUPPORTED = -2
HEAD_DIM = 128
KV_TILE_TOKENS = 16
KV_TILE_ELEMS = 2048


class MiError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libmi355_nanovllm.so not found at {LIB_PATH}; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C nano-vllm-ascend_amd/csrc` (there is no fallback path)."
        )
    return ctypes.CDLL(LIB_PATH)


lib = _load()

_p = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "mi_strerror": (c_char_p, [c_int]),
    "mi_version": (c_char_p, []),
    "mi_set_tuning": (c_int, [c_int, c_int]),
    "mi_get_tuning": (c_int, [c_int]),
    "mi_last_launch_error": (c_char_p, []),
    "mi_kv_elem_offset": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi_reshape_and_cache": (c_int, [_p, _p, c_int64, c_int64, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_scatter_update_kv": (c_int, [_p, _p, c_int64, c_int64, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_kv_cache_gather": (c_int, [_p, c_int, _p, c_int, _p, c_int, c_int, c_int, _p]),
    "mi_paged_attn_decode_workspace": (c_size_t, [c_int, c_int]),
    "mi_paged_attn_decode": (
        c_int,
        [_p, c_int64, _p, _p, _p, c_int, _p, _p, _p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_float, _p],
    ),
    "mi_paged_attn_decode_fused": (
        c_int,
        [_p, c_int64, _p, _p, c_float, _p, _p, _p, _p, _p, _p, c_int, _p, _p, _p, c_size_t, c_int, c_int, c_int,
         c_int, c_int, c_float, _p],
    ),
    "mi_paged_attn_decode_fused_ex": (
        c_int,
        [_p, c_int64, _p, _p, c_float, _p, _p, _p, _p, _p, _p, c_int, _p, _p, _p, c_size_t, c_int, c_int, c_int,
         c_int, c_int, c_float, _p, _p],
    ),
    "mi_paged_attn_decode_ex": (
        c_int,
        [_p, c_int64, _p, _p, _p, c_int, _p, _p, _p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
         c_int64, c_int64, c_int64, _p],
    ),
    "mi_paged_attn_prefill": (
        c_int,
        [_p, c_int64, _p, _p, _p, c_int, _p, _p, c_int, c_int, _p, c_int, c_int, c_int, c_int, c_float, _p],
    ),
    "mi_paged_attn_prefill_fused": (
        c_int,
        [_p, c_int64, _p, c_float, _p, _p, _p, _p, _p, c_int, _p, _p, c_int, c_int, _p, c_int, c_int, c_int, c_int,
         c_float, _p],
    ),
    "mi_paged_attn_prefill_fused_ex": (
        c_int,
        [_p, c_int64, _p, c_float, _p, _p, _p, _p, _p, c_int, _p, _p, c_int, c_int, _p, c_int, c_int, c_int, c_int,
         c_float, c_int, _p],
    ),
    "mi_rmsnorm": (c_int, [_p, c_int64, _p, _p, c_int, c_int, c_int, c_float, _p]),
    "mi_add_rmsnorm": (c_int, [_p, _p, _p, _p, _p, c_int, c_int, c_float, _p]),
    "mi_rope": (c_int, [_p, _p, _p, c_int64, c_int, _p, c_int64, c_int, _p, _p, c_int, c_int, _p]),
    "mi_qknorm_rope_store": (
        c_int,
        [_p, c_int64, _p, _p, c_float, _p, _p, _p, _p, _p, _p, c_int, c_int, c_int, c_int, c_int, c_int, _p],
    ),
    "mi_silu_mul": (c_int, [_p, _p, c_int, c_int, _p]),
    "mi_gemm_bf16_skinny": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "mi_gemm_bf16_max_rows": (c_int64, [c_int, c_int, c_int64]),
    "mi_gemm_bf16_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mi_gemm_bf16": (c_int, [_p, c_int64, _p, _p, _p, c_int64, c_int, c_int, c_int, c_int, _p, c_size_t, _p]),
    "mi_gemm_bf16_ex": (c_int, [_p, c_int64, _p, _p, c_int64, c_int, c_int, c_int, c_int, _p]),
    "mi_gemm_bf16_qkv_store": (c_int, [_p, c_int64, _p, _p, _p, c_int64, c_int, c_int, c_int, _p, c_float, _p, _p, _p, _p, _p,
                                       c_int, c_int, c_int, c_int, _p]),
    "mi_pack_weight": (c_int, [_p, _p, c_int, c_int, _p]),
    "mi_gemm_bf16_packed": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_gemm_bf16_packed_splitk": (c_int, [_p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_pack_weight_rows4": (c_int, [_p, _p, c_int, c_int, _p]),
    "mi_gemm_bf16_rows4": (c_int, [_p, _p, _p, c_int, c_int, c_int, _p]),
    "mi_add_rmsnorm_splitk": (c_int, [_p, c_int, _p, _p, _p, _p, c_int, c_int, c_float, _p]),
    "mi_add_rmsnorm_splitk_ex": (c_int, [_p, c_int, _p, _p, _p, _p, c_int, c_int, c_float, _p, _p]),
    "mi_gemm_bf16_packed_ex": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, c_int, _p, _p]),
    "mi_gemm_bf16_rowstat": (c_int, [_p, _p, _p, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_gemm_bf16_normed": (c_int, [_p, _p, c_int, _p, c_float, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_norm_from_stat": (c_int, [_p, _p, c_int, _p, c_float, _p, c_int, c_int, _p]),
    "mi_gemm_bf16_rowstat_ex": (c_int, [_p, _p, _p, _p, _p, _p, c_int, c_int, c_int, c_int, _p, _p]),
    "mi_gemm_bf16_normed_ex": (c_int, [_p, _p, c_int, _p, c_float, _p, _p, c_int, c_int, c_int, c_int, _p, _p]),
    "mi_moe_shapes_supported": (c_int, [c_int, c_int]),
    "mi_kv_store_plain": (c_int, [_p, _p, c_int64, c_int64, _p, _p, _p, c_int, c_int, c_int, c_int, c_int, _p]),
    "mi_rope_plain": (c_int, [_p, _p, c_int64, _p, c_int64, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_paged_attn_decode_plain_workspace": (c_size_t, [c_int, c_int, c_int]),
    "mi_paged_attn_decode_plain": (c_int, [_p, c_int64, _p, _p, _p, c_int, _p, _p, _p, c_size_t, c_int, c_int, c_int,
                                           c_int, c_int, c_float, _p]),
    "mi_paged_attn_prefill_plain": (c_int, [_p, c_int64, _p, _p, _p, c_int, _p, _p, c_int, c_int, _p, c_int, c_int,
                                            c_int, c_int, c_float, _p]),
    "mi_embedding": (c_int, [_p, _p, _p, c_int, c_int, c_int64, c_int64, _p]),
    "mi_embedding_from_prev": (c_int, [_p, _p, _p, _p, _p, c_int, c_int, c_int64, c_int64, _p]),
    "mi_stage_copy": (c_int, [_p, _p, c_int64, _p]),
    "mi_gather_last_tokens": (c_int, [_p, _p, _p, c_int, c_int, _p]),
    "mi_argmax": (c_int, [_p, c_int64, _p, c_int, c_int, _p]),
    "mi_sample": (c_int, [_p, c_int64, _p, _p, c_int, c_int, c_uint64, c_uint64, _p]),
    "mi_gemm_pick_groups": (c_int, [c_int, c_int, c_int, c_int]),
    "mi_gemm_bf16_packed_pick": (c_int, [_p, _p, _p, c_int, c_int, c_int, _p, _p, _p, _p]),
    "mi_gemm_fp8w_packed_pick": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, _p, _p, _p, _p]),
    "mi_pick_final": (c_int, [_p, c_int, c_int, _p, _p]),
    "mi_gemm_bf16_packed_pick_shard": (c_int, [_p, _p, _p, c_int, c_int, c_int, _p, _p, _p, c_int, _p]),
    "mi_pick_final_pairs": (c_int, [_p, c_int, c_int, _p, _p]),
    "mi_pick_exchange": (c_int, [_p, _p, _p, c_int, _p]),
    "mi_comm_status_async": (c_int, [_p, _p, _p]),
    "mi_xxh64_chain": (c_uint64, [_p, c_size_t, c_int, c_uint64]),
    "mi_xxh64_chain_blocks": (c_int, [_p, c_int, c_int, c_int, c_uint64, _p]),
    "mi_pack_weight_fp8": (c_int, [_p, _p, c_int, c_int, _p]),
    "mi_gemm_fp8w_packed": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_gemm_fp8w_packed_splitk": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, _p]),
    "mi_moe_route": (c_int, [_p, c_int, c_int, c_int, _p, _p, _p]),
    "mi_moe_sort": (c_int, [_p, c_int, c_int, c_int, _p, _p, _p, _p]),
    "mi_moe_gate_up": (c_int, [_p, _p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "mi_moe_down": (c_int, [_p, _p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "mi_moe_combine": (c_int, [_p, _p, _p, c_int, c_int, c_int, _p]),
    "mi_comm_region_bytes": (c_size_t, [c_int, c_size_t]),
    "mi_comm_region_alloc": (c_int, [c_size_t, ctypes.POINTER(_p), _p]),
    "mi_comm_region_open": (c_int, [_p, ctypes.POINTER(_p)]),
    "mi_comm_region_close": (c_int, [_p]),
    "mi_comm_region_free": (c_int, [_p]),
    "mi_comm_create": (c_int, [c_int, c_int, ctypes.POINTER(_p), c_size_t, ctypes.POINTER(_p)]),
    "mi_comm_destroy": (c_int, [_p]),
    "mi_allreduce_sum_bf16": (c_int, [_p, _p, _p, c_int64, _p]),
    "mi_allreduce_add_rmsnorm": (c_int, [_p, _p, _p, _p, _p, _p, c_int, c_int, c_float, _p]),
    "mi_comm_set_spin_limit": (c_int, [_p, ctypes.c_uint32]),
    "mi_comm_status": (c_int, [_p, ctypes.POINTER(c_int)]),
    "mi_comm_timeout_info": (c_int, [_p, _p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

# ---- tuning knobs (include/mi355_nanovllm.h: mi_tuning_knob).  The library never reads the environment; the A/B
# switches documented in tools/README.md are mapped onto mi_set_tuning() here, once, at import.
(TUNE_ATTN_PIPE, TUNE_ATTN_RESOLVE, TUNE_NORM_WPR, TUNE_ROPE_BLOCK64, TUNE_PLAIN_SPLIT_TARGET, TUNE_PREFILL_P_SPLIT,
 TUNE_GEMM_PIPE) = range(7)
_ENV_KNOBS = {
    "MI355_ATTN_PIPE": TUNE_ATTN_PIPE,
    "MI355_ATTN_RESOLVE": TUNE_ATTN_RESOLVE,
    "MI355_NORM_WPR": TUNE_NORM_WPR,
    "MI355_ROPE_BLOCK64": TUNE_ROPE_BLOCK64,
    "MI355_PLAIN_SPLIT_TARGET": TUNE_PLAIN_SPLIT_TARGET,
    "MI355_PREFILL_P_SPLIT": TUNE_PREFILL_P_SPLIT,
    "MI355_GEMM_PIPE": TUNE_GEMM_PIPE,
}


def set_tuning(knob: int, value: int) -> None:
    if lib.mi_set_tuning(int(knob), int(value)) != MI_OK:
        raise ValueError(f"mi_set_tuning({knob}, {value}): unknown knob or value out of range")


def get_tuning(knob: int) -> int:
    return lib.mi_get_tuning(int(knob))


for _env, _knob in _ENV_KNOBS.items():
    if os.environ.get(_env) is not None:
        set_tuning(_knob, int(os.environ[_env]))


if os.environ.get("MI355_TRACE"):  # debugging aid: name each launch, make faults synchronous
    class _Traced:
        def __init__(self, inner):
            self._inner = inner

        def __getattr__(self, name):
            fn = getattr(self._inner, name)

            def call(*args):
                print(f"[mi355] {name}{tuple(a for a in args if isinstance(a, (int, float)) and abs(a) < 1 << 40)}",
                      flush=True)
                rc = fn(*args)
                if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize()
                return rc

            return call

    lib = _Traced(lib)


def version() -> str:
    return lib.mi_version().decode()


def check(code: int, what: str) -> None:
    if code != MI_OK:
        msg = lib.mi_strerror(code).decode()
        if code == -4:
            msg += ": " + lib.mi_last_launch_error().decode()
        raise MiError(f"{what} failed: {msg} (code {code})")


def stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MiError(
                "mi355_nanovllm kernels only run on a HIP device; got a "
                f"{t.device} tensor (there is no CPU fallback)"
            )


def xxh64_chain_blocks(token_ids, n_blocks: int, block_size: int, prefix: int = -1) -> list[int]:
    """Chained hashes of the first n_blocks full blocks of token_ids (one C call)."""
    from array import array

    if n_blocks <= 0:
        return []
    # an array('q') (Sequence.ids_array) is hashed in place, a list is converted first
    toks = token_ids if isinstance(token_ids, array) else array("q", token_ids[: n_blocks * block_size])
    assert len(toks) >= n_blocks * block_size
    out = (c_uint64 * n_blocks)()
    addr, _ = toks.buffer_info()
    check(lib.mi_xxh64_chain_blocks(addr, n_blocks, block_size, int(prefix != -1),
                                    0 if prefix == -1 else prefix & 0xFFFFFFFFFFFFFFFF, out),
          "mi_xxh64_chain_blocks")
    return list(out)


def xxh64_chain(data: bytes, prefix: int = -1) -> int:
    """xxh64(prefix_le64 ++ data) — BlockManager.compute_hash (block_manager.py:38-44)."""
    if prefix == -1:
        return lib.mi_xxh64_chain(data, len(data), 0, 0)
    return lib.mi_xxh64_chain(data, len(data), 1, prefix & 0xFFFFFFFFFFFFFFFF)
