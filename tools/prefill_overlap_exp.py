#!/usr/bin/env python3
"""Experiment: does a prefill step gain from running two half-batches on two streams, so that one half's
GEMMs (MFMA-bound) overlap the other half's HBM-bound elementwise kernels and attention?  (Round 2 measured -6.5 % with
library GEMMs; round 4 re-runs it on the product's own launches: a persistent 256-workgroup tile GEMM leaves no CU to a
second stream, so the overlap can only come from kernel tails.)
Synthetic layer sequence at Qwen3-0.6B widths: per half 8 x 1024 tokens, 28 layers of
[add+RMSNorm, qkv GEMM, attention, o GEMM, add+RMSNorm, gate_up GEMM, SwiGLU, down GEMM]."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = torch.device("cuda:0")
H, HQ, HKV, I, T, NSEQ, BS, L = 1024, 16, 8, 3072, 1024, int(os.environ.get("NSEQ", 8)), 16, 28


class Half:
    def __init__(self, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        n = NSEQ * T
        nb = T // BS
        self.x = torch.randn(n, H, device=DEV).bfloat16()
        self.res = torch.randn(n, H, device=DEV).bfloat16()
        self.kc = torch.randn(ops.kv_cache_shape(NSEQ * nb, HKV, BS), device=DEV).bfloat16()
        self.vc = torch.randn(ops.kv_cache_shape(NSEQ * nb, HKV, BS), device=DEV).bfloat16()
        self.tables = torch.randperm(NSEQ * nb, generator=g).to(torch.int32).view(NSEQ, nb).to(DEV)
        self.cu = (torch.arange(NSEQ + 1, dtype=torch.int32) * T).to(DEV)
        self.kvl = torch.full((NSEQ,), T, dtype=torch.int32, device=DEV)
        self.attn_out = torch.empty(n, HQ * 128, dtype=torch.bfloat16, device=DEV)
        self.pos = torch.arange(T, dtype=torch.int64).repeat(NSEQ).to(DEV)
        self.slots = (self.tables.long().repeat_interleave(BS, dim=1) * BS
                      + torch.arange(BS, device=DEV).repeat(nb)).view(-1).to(torch.int32)


W = {k: (torch.randn(*s, device=DEV) * 0.02).bfloat16() for k, s in
     dict(qkv=((HQ + 2 * HKV) * 128, H), o=(H, HQ * 128), gu=(2 * I, H), dn=(H, I)).items()}
WN = torch.ones(H, device=DEV).bfloat16()
W128 = torch.ones(128, device=DEV).bfloat16()
COS_SIN = torch.randn(T, 128, device=DEV)


def layer(h: Half):
    """round 4: the product's own launches (hand-written tile GEMMs, SwiGLU epilogue, K/V tile store, fused-Q attention)"""
    xn, r = ops.add_rmsnorm(h.x, h.res, WN, 1e-6)
    qkv = ops.gemm_tile(xn, W["qkv"])
    ops.qknorm_rope_store(qkv, W128, W128, 1e-6, h.pos, COS_SIN, h.kc, h.vc, h.slots, HQ, HKV, BS, store_q=False)
    ops.paged_attn_prefill_fused(qkv, W128, 1e-6, h.pos, COS_SIN, h.kc, h.vc, h.tables, h.cu, h.kvl, T, HQ, HKV, BS,
                                 128 ** -0.5, out=h.attn_out)
    y = ops.gemm_tile(h.attn_out, W["o"])
    xn, r = ops.add_rmsnorm(y, r, WN, 1e-6)
    a = ops.gemm_tile(xn, W["gu"], silu_mul=True)
    ops.gemm_tile(a, W["dn"])


def run(halves, streams):
    for h, s in zip(halves, streams):
        with torch.cuda.stream(s):
            for _ in range(L):
                layer(h)


def main():
    a, b = Half(0), Half(1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for mode, streams in (("one stream", (s0, s0)), ("two streams", (s0, s1)), ("one stream", (s0, s0)),
                          ("two streams", (s0, s1))):
        run((a, b), streams)  # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s0.wait_event(e0)
        s1.wait_event(e0)
        run((a, b), streams)
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.current_stream().wait_stream(s1)
        e1.record()
        torch.cuda.synchronize()
        print(f"{mode:12s}: {e0.elapsed_time(e1):8.2f} ms for 2 x {NSEQ} x {T} tokens x {L} layers", flush=True)


if __name__ == "__main__":
    main()
