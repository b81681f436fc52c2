#!/usr/bin/env python3
"""Experiment: the MLP half of a prefill layer (add+RMSNorm -> gate_up GEMM -> SwiGLU -> down GEMM) over 16384 tokens
at once vs in token chunks small enough for the intermediates to stay in the 256 MiB Infinity Cache / L2."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = torch.device("cuda:0")
H, I, T, L = 1024, 3072, 16384, 8
x = torch.randn(T, H, device=DEV).bfloat16()
res = torch.randn(T, H, device=DEV).bfloat16()
wn = torch.ones(H, device=DEV).bfloat16()
gu = [(torch.randn(2 * I, H, device=DEV) * 0.02).bfloat16() for _ in range(L)]
dn = [(torch.randn(H, I, device=DEV) * 0.02).bfloat16() for _ in range(L)]
out = torch.empty(T, H, device=DEV, dtype=torch.bfloat16)


def mlp(l, chunk):
    for a in range(0, T, chunk):
        xs, rs = x[a:a + chunk], res[a:a + chunk]
        xn, _ = ops.add_rmsnorm(xs, rs, wn, 1e-6)
        act = ops.silu_mul(F.linear(xn, gu[l]))
        torch.matmul(act, dn[l].t(), out=out[a:a + chunk])


for chunk in (16384, 8192, 4096, 2048, 1024, 16384, 4096):
    for l in range(L):
        mlp(l, chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for l in range(L):
            mlp(l, chunk)
    e1.record()
    torch.cuda.synchronize()
    print(f"chunk {chunk:6d}: {e0.elapsed_time(e1) / (3 * L) * 1e3:8.1f} us per layer (MLP half, {T} tokens)", flush=True)
