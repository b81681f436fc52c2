"""Shader clock and cycles per K step INSIDE the four-wave tile GEMM (variant 4096 + 512 of mi_gemm_bf16_ex: the tile
loop without output stores, stamped with the cycle counter and the 100 MHz reference clock per workgroup), for the
product kernel (16 x 16 x 32 MFMAs) and its 32 x 32 x 16 form.  Needs a library built with
`make -C nano-vllm-ascend_amd/csrc EXPERIMENTS=1` (the stamped variants are not in the default build).
usage: python tools/gemm_clock.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
from nanovllm import _C, ops  # noqa: E402

if not _C.HAS_EXPERIMENTS:
    sys.exit("tools/gemm_clock.py needs an EXPERIMENTS=1 build of the library")

DEV = "cuda:0"
for form, base in (("16x16x32", 0), ("32x32x16", 1 << 19)):
  print(form)
  for M, N, K, label in ((16384, 4096, 1024, "qkv"), (16384, 1024, 2048, "o_proj"), (16384, 1024, 3072, "down"), (8192, 8192, 8192, "8k^3")):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        ops.gemm_tile(x, w, out=y, variant=base + 4096 + 512)
    torch.cuda.synchronize()
    v = y.view(-1).view(torch.float32)[:512].float().cpu().view(256, 2)
    tiles = (M // 256) * (N // 256)
    steps = (tiles / 256 if tiles >= 256 else 1) * (K // 64)
    cyc, ref = v[:, 0].mean().item(), v[:, 1].mean().item()
    print(f"  {label:7s} {steps:5.0f} K steps per workgroup: {cyc / steps:7.0f} cycles per step, {ref * 10 / steps / 1000:6.3f} us per step, shader clock {cyc / ref * 100:5.0f} MHz")
