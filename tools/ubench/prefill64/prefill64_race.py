"""Where does the 64-columns-per-wave prefill kernel differ from the 32-column kernel under memory pressure?
(tests/test_kernels_gpu.py::test_prefill_attention_chunk_pipeline_stress, variant 128)"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "nano-vllm-ascend_amd")]
import oracle  # noqa: E402
from nanovllm import ops  # noqa: E402

DEV = "cuda:0"
gen = torch.Generator().manual_seed(77)
hq, hkv, bs, n_seqs, L = 16, 8, 16, 16, 1024
T = n_seqs * L
nblk = n_seqs * (L // bs)
qkv = (torch.randn(T, (hq + 2 * hkv) * 128, generator=gen) * 0.8).bfloat16().to(DEV)
qw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
kw = (1 + 0.1 * torch.randn(128, generator=gen)).bfloat16().to(DEV)
table = oracle.build_cos_sin_cache(128, 2048, 1e6).to(DEV)
pos = torch.arange(L, dtype=torch.int64).repeat(n_seqs).to(DEV)
cu = torch.arange(0, T + 1, L, dtype=torch.int32).to(DEV)
kvl = torch.full((n_seqs,), L, dtype=torch.int32, device=DEV)
for name, perm in (("straight", torch.arange(nblk)), ("scrambled", torch.randperm(nblk + 7, generator=gen)[:nblk])):
    bt = perm.view(n_seqs, L // bs).to(torch.int32).to(DEV)
    slots = (bt.long().repeat_interleave(bs, dim=1) * bs + torch.arange(bs, device=DEV).repeat(L // bs)).view(-1).to(torch.int32)
    kc = torch.zeros(ops.kv_cache_shape(nblk + 7, hkv, bs), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.qknorm_rope_store(qkv, qw, kw, 1e-6, pos, table, kc, vc, slots, hq, hkv, bs, store_q=False)
    run = lambda v: ops.paged_attn_prefill_fused(qkv, qw, 1e-6, pos, table, kc, vc, bt, cu, kvl, L, hq, hkv, bs, 128 ** -0.5, variant=v)  # noqa: E731
    ref = run(0)
    torch.cuda.synchronize()
    side, junk = torch.cuda.Stream(), torch.empty(192 << 20, dtype=torch.uint8, device=DEV)
    for pressure in (False, True):
        for it in range(6):
            if pressure:
                with torch.cuda.stream(side):
                    junk.add_(1)
            got = run(128)
            torch.cuda.synchronize()
            bad = (got.view(torch.int16) != ref.view(torch.int16)).view(n_seqs, L, hq, 128)
            nb = int(bad.sum())
            msg = f"{name} pressure={pressure} it={it}: {nb} elements differ"
            if nb:
                idx = bad.nonzero()
                toks = idx[:, 1]
                d = (got.float() - ref.float()).abs().view(n_seqs, L, hq, 128)
                msg += (f"; seqs {sorted(set(idx[:, 0].tolist()))[:8]} tokens {int(toks.min())}..{int(toks.max())} "
                        f"(token % 256 in {sorted(set((toks % 256 // 32).tolist()))} x32) heads {sorted(set(idx[:, 2].tolist()))} "
                        f"rows {len(set(map(tuple, idx[:, :3].tolist())))} max|d| {float(d.max()):.3e} nan {bool(torch.isnan(got).any())}")
                rows = sorted(set(map(tuple, idx[:, :3].tolist())))[:6]
                msg += f" first rows {rows}"
            print(msg, flush=True)
