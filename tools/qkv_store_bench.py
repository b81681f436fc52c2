"""The qkv projection of a 16 x 1024-token prefill step of Qwen3-0.6B: mi_gemm_bf16 + mi_qknorm_rope_store(q_out = NULL)
against mi_gemm_bf16_qkv_store (the K / V store in the GEMM's epilogue), us per launch (hipGraph replays).
usage: python tools/qkv_store_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=20, rounds=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main():
    M, K, n_q, n_kv, bs = 16384, 1024, 16, 8, 16
    N = (n_q + 2 * n_kv) * 128
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    kw = torch.ones(128).bfloat16().to(DEV)
    n_blocks = M // bs + 8
    slots = torch.arange(M, dtype=torch.int32, device=DEV)
    positions = (torch.arange(M) % 1024).to(torch.int64).to(DEV)
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(1024).float()[:, None] * inv[None, :]
    cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous().to(DEV)
    kc = torch.zeros(ops.kv_cache_shape(n_blocks, n_kv, bs), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    qkv = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t_gemm = timed(lambda: ops.gemm_tile(x, w, out=qkv))
    t_store = timed(lambda: ops.qknorm_rope_store(qkv, kw, kw, 1e-6, positions, cos_sin, kc, vc, slots, n_q, n_kv, bs, store_q=False))
    t_fused = timed(lambda: ops.gemm_qkv_store(x, w, None, kw, 1e-6, positions, cos_sin, kc, vc, slots, n_q, n_kv, bs, out=qkv))
    print(f"qkv GEMM {t_gemm:.1f} us + K/V store {t_store:.1f} us = {t_gemm + t_store:.1f} us;  one launch with the store in the epilogue: {t_fused:.1f} us")


if __name__ == "__main__":
    main()
