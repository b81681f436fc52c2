#!/usr/bin/env python3
"""Where does a decode-sized GEMM launch spend its time?  Runs the variants of tools/ubench/gemm_exp.hip
(traffic of x and/or of the weights removed, x loads first, x staged in full lines, rotated K-slices)
back to back inside a hipGraph over 28 rotating weight copies and prints us per launch."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, "ubench", "libgemm_exp.so"))
for f in (lib.exp_rows4, lib.exp_tile16):
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.exp_empty.restype = ctypes.c_int
lib.exp_empty.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
DEV = torch.device("cuda:0")
L = 28


def timeit(fn, iters=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for l in range(L):
            fn(l)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for l in range(L):
            fn(l)
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (iters * L)


def st():
    return torch.cuda.current_stream().cuda_stream


def run(name, fn_c, N, K, waves, variants, M=32):
    ws = [(torch.randn(N, K, device=DEV) * 0.02).bfloat16() for _ in range(L)]
    x = torch.randn(M, K, device=DEV).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    if fn_c is lib.exp_rows4:
        wp = [ops.pack_weight_rows4(w) for w in ws]
    else:
        wp = [ops.pack_weight(w) for w in ws]
    ref = (x.float() @ ws[0].float().T).bfloat16()
    for v, label in variants:
        def go(l):
            rc = fn_c(x.data_ptr(), wp[l].data_ptr(), y.data_ptr(), M, N, K, v, waves, st())
            assert rc == 0, rc
        go(0)
        torch.cuda.synchronize()
        ok = ""
        if not v & 3:
            ok = "ok" if (y.float() - ref.float()).abs().max().item() < 0.05 else "WRONG"
        print(f"{name:26s} waves={waves:2d} variant={v:2d} {label:34s} {timeit(go):7.2f} us {ok}", flush=True)


def main():
    y = torch.empty(4096, dtype=torch.bfloat16, device=DEV)
    for blocks, threads in ((256, 1024), (256, 512), (64, 1024), (32, 256), (1024, 256)):
        t = timeit(lambda l: lib.exp_empty(y.data_ptr(), blocks, threads, st()))
        print(f"empty kernel {blocks} x {threads}: {t:6.2f} us", flush=True)
    V = [(0, "baseline"), (1, "no x traffic"), (2, "no weight traffic"), (3, "neither"), (4, "rotated K-slices"),
         (8, "x in full lines via LDS"), (16, "x loads first"), (12, "rotated + LDS"), (9, "LDS, no x traffic")]
    run("rows4 o_proj 1024x2048", lib.exp_rows4, 1024, 2048, 16, V)
    run("rows4 o_proj 1024x2048", lib.exp_rows4, 1024, 2048, 8, V[:4])
    run("rows4 down 1024x3072", lib.exp_rows4, 1024, 3072, 16, V)
    run("tile16 qkv 4096x1024", lib.exp_tile16, 4096, 1024, 16, V)
    run("tile16 qkv 4096x1024", lib.exp_tile16, 4096, 1024, 8, V)
    run("tile16 o_proj 1024x2048", lib.exp_tile16, 1024, 2048, 16, V[:4])


if __name__ == "__main__":
    main()
