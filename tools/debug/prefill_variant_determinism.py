#!/usr/bin/env python3
"""Every schedule variant of the fused-Q prefill attention (mi_paged_attn_prefill_fused_ex), launched repeatedly over a
scrambled block table next to a competing HBM stream: does each launch reproduce the first one bit for bit, and do the
variants that share an arithmetic agree?  (The body of tests/test_kernels_gpu.py::test_prefill_attention_chunk_pipeline_stress
as a report instead of an assertion.)   usage: python tools/debug/prefill_variant_determinism.py [reps] [variants...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "nano-vllm-ascend_amd"), ROOT]
import oracle  # noqa: E402
from nanovllm import ops  # noqa: E402

DEV = "cuda:0"


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    variants = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 4, 8, 16, 24, 28, 32, 64, 96]
    gen = torch.Generator().manual_seed(77)
    hq, hkv, bs, n_seqs, L = 16, 8, 16, 16, 1024
    T, nblk = n_seqs * L, n_seqs * (L // bs)
    qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=gen) * 0.8).bfloat16().to(DEV)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
    table = oracle.build_cos_sin_cache(128, 2048, 1e6).to(DEV)
    pos = torch.arange(L, dtype=torch.int64).repeat(n_seqs).to(DEV)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32).to(DEV)
    kvl = torch.full((n_seqs,), L, dtype=torch.int32, device=DEV)
    perm = torch.randperm(nblk + 7, generator=gen)[:nblk]
    bt = perm.view(n_seqs, L // bs).to(torch.int32).to(DEV)
    slots = (bt.long().repeat_interleave(bs, dim=1) * bs + torch.arange(bs, device=DEV).repeat(L // bs)).view(-1).to(torch.int32)
    kc = torch.zeros(ops.kv_cache_shape(nblk + 7, hkv, bs), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc, vc, slots, hq, hkv, bs, store_q=False)
    side, junk = torch.cuda.Stream(), torch.empty(192 << 20, dtype=torch.uint8, device=DEV)
    firsts = {}
    for v in variants:
        run = lambda: ops.paged_attn_prefill_fused(qkv, qw, 1e-6, pos, table, kc, vc, bt, cu, kvl, L, hq, hkv, bs,  # noqa: E731
                                                   128 ** -0.5, variant=v)
        first = run().clone()
        bad = []
        for it in range(reps):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            again = run()
            d = again.view(torch.int16) != first.view(torch.int16)
            if bool(d.any()):
                rows = d.any(1).nonzero().flatten()
                bad.append((it, int(d.sum()), rows[:4].tolist(), int(rows.numel())))
        torch.cuda.synchronize()
        firsts[v] = first
        print(f"variant {v:3d}: {len(bad)} of {reps} launches differ from the first" + (f"  e.g. (launch, elements, rows, n rows) {bad[:3]}" if bad else ""))
    base = firsts[variants[0]]
    for v in variants[1:]:
        d = firsts[v].view(torch.int16) != base.view(torch.int16)
        print(f"variant {v:3d} vs {variants[0]}: {int(d.sum())} elements differ (max abs {float((firsts[v].float() - base.float()).abs().max()):.3e})")


if __name__ == "__main__":
    main()
