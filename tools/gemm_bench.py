"""Tile GEMM (mi_gemm_bf16) against the library GEMM behind F.linear on the prefill shapes of the bench
(16384 tokens x the four Qwen3-0.6B projections) and a few others; every schedule variant of
mi_gemm_bf16_ex.  Timed as hipGraph replays of REPS back-to-back launches on random data (the guide's rule 25:
zero-filled operands clock higher).  (Round 4's timing-only variants - no stores - went with that round's EXPERIMENTS=1
build flavour; their numbers are in profiles/r04_gemm_ablation.txt.)  Usage: python tools/gemm_bench.py [out.json]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nano-vllm-ascend_amd"))
from nanovllm import _C, ops  # noqa: E402

DEV = "cuda:0"
REPS = 10


def timed(fn, reps=REPS, rounds=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = float("inf")
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best  # us per launch


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    shapes = [  # (M, N, K, label)
        (16384, 4096, 1024, "qkv"), (16384, 1024, 2048, "o_proj"), (16384, 6144, 1024, "gate_up"),
        (16384, 1024, 3072, "down"), (4096, 4096, 1024, "qkv@4k"), (1024, 4096, 1024, "qkv@1k"),
        (1024, 1024, 3072, "down@1k"), (2048, 1024, 3072, "down@2k"), (512, 4096, 1024, "qkv@512"),
        (256, 4096, 1024, "qkv@256"), (128, 4096, 1024, "qkv@128"), (128, 1024, 3072, "down@128"), (8192, 8192, 8192, "8k^3"),
    ]
    E8 = 1 << 20  # the eight-wave ping-pong kernel of rounds 2-3 (kept for comparison)
    variants = {"tile": 0, "xcd_rect": 4, "direct_stores": 1 << 18, "mfma_32x32x16": 1 << 19, "eight_wave_r03": E8}
    if os.environ.get("GEMM_QUICK"):  # the four prefill projections only
        shapes = shapes[:4]
    if os.environ.get("GEMM_MID"):  # the two kernels side by side over the mid-size shapes
        variants = {"tile": 0, "mid128": 65536, "mid128_2stage": 65538, "mid128_3stage": 65539}
        shapes = [(M, N, K, f"{name}@{M}") for M in ((16384,) if os.environ.get("GEMM_MID") == "big" else (128, 256, 512, 1024, 2048, 4096))
                  for N, K, name in ((4096, 1024, "qkv"), (1024, 2048, "o_proj"), (6144, 1024, "gate_up"), (1024, 3072, "down"))]
    rows = []
    for M, N, K, label in shapes:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        flops = 2.0 * M * N * K
        row = {"shape": [M, N, K], "label": label}
        t = timed(lambda: F.linear(x, w))
        row["library_us"] = round(t, 2)
        row["library_tflops"] = round(flops / t / 1e6, 1)
        for name, v in variants.items():
            t = timed(lambda: ops.gemm_tile(x, w, out=y, variant=v))
            row[name + "_us"] = round(t, 2)
            row[name + "_tflops"] = round(flops / t / 1e6, 1)
        t = timed(lambda: ops.gemm_tile(x, w, out=y))  # the product entry point (K slices for few-tile shapes)
        row["product_us"] = round(t, 2)
        row["product_tflops"] = round(flops / t / 1e6, 1)
        if M <= 512:  # decode-sized: what the layers actually call
            wp = ops.pack_weight(w)
            t = timed(lambda: ops.gemm_packed(x, wp))
            row["skinny_chunked_us"] = round(t, 2)
        ref = F.linear(x, w)
        ops.gemm_tile(x, w, out=y, variant=0)
        row["max_abs_diff_vs_library"] = float((y.float() - ref.float()).abs().max())
        if "direct_stores" in variants:
            y.zero_()
            ops.gemm_tile(x, w, out=y, variant=variants["direct_stores"])
            row["direct_stores_max_abs_diff"] = float((y.float() - ref.float()).abs().max())
        if N % 256 == 0 and label in ("gate_up",):
            ya = torch.empty(M, N // 2, dtype=torch.bfloat16, device=DEV)
            t = timed(lambda: ops.gemm_tile(x, w, out=ya, silu_mul=True))
            row["swiglu_fused_us"] = round(t, 2)
            t2 = timed(lambda: ops.silu_mul(F.linear(x, w)))
            row["library_plus_silu_mul_us"] = round(t2, 2)
        print(json.dumps(row), flush=True)
        rows.append(row)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
