"""Run the tile GEMM a few times on one shape (for rocprofv3 passes): python tools/gemm_prof.py M N K [variant] [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).bfloat16().to("cuda:0")
w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to("cuda:0")
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda:0")
for _ in range(iters):
    ops.gemm_tile(x, w, out=y, variant=variant)
torch.cuda.synchronize()
