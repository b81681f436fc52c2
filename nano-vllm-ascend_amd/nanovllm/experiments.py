"""Wrappers of the measured-and-lost experiments (include/mi355_nanovllm_experiments.h).

Only usable with a library built by `make -C nano-vllm-ascend_amd/csrc EXPERIMENTS=1`; nothing on the product path
imports this module unless an experiment switch (MI355_WARM_L2, MI355_SEAM_OVERLAP) is set."""
from __future__ import annotations

import torch

from nanovllm import _C
from nanovllm._C import check, lib, ptr, require_gpu, stream
from nanovllm.ops import _BF16, _bf16


def require() -> None:
    if not _C.HAS_EXPERIMENTS:
        raise _C.MiError("this switch needs the experiments build of the library: "
                         "make -C nano-vllm-ascend_amd/csrc EXPERIMENTS=1")


def warm_l2(weights, n_workgroups: int = 64) -> None:
    """queue a launch that pulls up to two packed bf16 weights ([N, K], mi_pack_weight) into L2 (mi_warm_l2)"""
    ws = [t for t in weights if isinstance(t, torch.Tensor) and t.dtype == _BF16 and t.dim() == 2
          and (32 * t.shape[1]) % 4096 == 0 and t.numel() * 2 < (1 << 32)][:2]
    if not ws:
        return
    require_gpu(*ws)
    w0, w1 = ws[0], (ws[1] if len(ws) > 1 else None)
    check(lib.mi_warm_l2(ptr(w0), w0.numel() * 2, 32 * w0.shape[1], ptr(w1), w1.numel() * 2 if w1 is not None else 0,
                         32 * w1.shape[1] if w1 is not None else 0, n_workgroups, stream()), "mi_warm_l2")


def mlp_half_fused(partials, residual, norm_w, eps: float, w_gate_up_packed, w_down_packed, sync_words, scratch=None):
    """EXPERIMENT (csrc/mlp_half.hip): add+RMSNorm -> gate_up+SwiGLU -> down split-K as one persistent launch.
    -> (fp32 partials [4, rows, hidden], new residual); sync_words: 8 zeroed int32 on the device, kept across calls."""
    require_gpu(partials, residual, norm_w, w_gate_up_packed, w_down_packed, sync_words)
    _bf16(residual, norm_w, w_gate_up_packed, w_down_packed)
    assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.shape[0] == 4
    rows, hidden = residual.shape
    inter = w_down_packed.shape[1]
    assert sync_words.dtype == torch.int32 and sync_words.numel() >= 8
    if scratch is None:
        scratch = (torch.empty_like(residual), torch.empty(rows, hidden, dtype=_BF16, device=residual.device),
                   torch.empty(rows, inter, dtype=_BF16, device=residual.device),
                   torch.empty(4, rows, hidden, dtype=torch.float32, device=residual.device))
    res_out, xn, act, out = scratch
    check(lib.mi_mlp_half_fused(ptr(partials), ptr(residual), ptr(norm_w), float(eps), ptr(w_gate_up_packed),
                                ptr(w_down_packed), ptr(res_out), ptr(xn), ptr(act), ptr(out), ptr(sync_words), rows,
                                hidden, inter, stream()), "mi_mlp_half_fused")
    return out, res_out


def add_rmsnorm_splitk_warm(partials, residual, w, eps: float, warm, out=None, residual_out=None):
    """mi_add_rmsnorm_splitk with the launch's idle CUs pulling up to two packed bf16 weights ([N, K], mi_pack_weight)
    of the GEMMs behind this norm into L2.  Decode-sized inputs with nsplit in {1, 2, 3, 4, 6, 8} only; anything else
    takes the plain kernel (nsplit 16 used to return MI_EUNSUPPORTED here)."""
    from nanovllm import ops

    require_gpu(partials, residual, w)
    _bf16(residual, w)
    nsplit, cols = partials.shape[0], residual.shape[-1]
    rows = residual.numel() // cols
    warm = [t for t in warm if isinstance(t, torch.Tensor) and t.dtype == _BF16 and t.dim() == 2
            and (32 * t.shape[1]) % 4096 == 0 and t.numel() * 2 < (1 << 32)]
    if not (warm and rows <= 64 and cols <= 1024 and cols % 4 == 0 and nsplit in (1, 2, 3, 4, 6, 8)):
        return ops.add_rmsnorm_splitk(partials, residual, w, eps, out, residual_out)
    if out is None:
        out = torch.empty_like(residual)
    if residual_out is None:
        residual_out = torch.empty_like(residual)
    w0, w1 = warm[0], (warm[1] if len(warm) > 1 else None)
    check(
        lib.mi_add_rmsnorm_splitk_warm(ptr(partials), nsplit, ptr(residual), ptr(w), ptr(out), ptr(residual_out),
                                       rows, cols, float(eps), ptr(w0), w0.numel() * 2, 32 * w0.shape[1],
                                       ptr(w1), w1.numel() * 2 if w1 is not None else 0,
                                       32 * w1.shape[1] if w1 is not None else 0, stream()),
        "mi_add_rmsnorm_splitk_warm",
    )
    return out, residual_out
