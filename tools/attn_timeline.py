#!/usr/bin/env python3
"""Where the microseconds of ONE decode attention launch go ("first measure").

Runs the fused decode-attention step of the bench (Qwen3-0.6B heads, bs 32, block 16, ctx from the environment)
through the instrumented kernel (mi_paged_attn_decode_fused_ex): every wave stamps s_memrealtime (the chip-wide 100 MHz clock) at eight points.  The
launch is replayed over 28 distinct caches (nothing served from the Infinity Cache); the stamps of the LAST launch are
reduced to, per phase, the distribution over the 2048 waves (256 workgroups x 8) of
    time since the EARLIEST wave's entry (the launch's own clock)  and  phase durations.

usage: python tools/attn_timeline.py   [CTX=1100]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = ["entry", "ctx_len known", "2 chunks requested", "q/k/v rows published", "first chunk consumed",
         "run attended", "all waves arrived", "merged + stored"]


def main():
    B, ctx, bs, hq, hkv, L = 32, int(os.environ.get("CTX", 1100)), 16, 16, 8, 28
    nb = (ctx + bs - 1) // bs
    nblk = B * nb + 8
    g = torch.Generator(device="cpu").manual_seed(0)
    kc = [torch.randn(ops.kv_cache_shape(nblk, hkv, bs), device=DEV).bfloat16() for _ in range(L)]
    vc = [torch.randn(ops.kv_cache_shape(nblk, hkv, bs), device=DEV).bfloat16() for _ in range(L)]
    perm = torch.randperm(nblk, generator=g)[: B * nb].to(torch.int32).view(B, nb).to(DEV)
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=DEV)
    slots = torch.stack([perm[:, (ctx - 1) // bs], torch.full((B,), (ctx - 1) % bs, dtype=torch.int32, device=DEV)], 1).contiguous()
    qkv = torch.randn(B, (hq + 2 * hkv) * 128, device=DEV).bfloat16()
    w128 = torch.ones(128, device=DEV).bfloat16()
    rope = torch.randn(4096, 128, device=DEV)
    out = torch.empty(B, hq * 128, dtype=torch.bfloat16, device=DEV)
    ws = ops.attn_workspace(DEV, B, hq)
    stamps = torch.zeros(B * hkv * 16, 8, 8, dtype=torch.int64, device=DEV)

    def launch(l, stamped):
        if stamped:
            ops.paged_attn_decode_fused_stamped(qkv, w128, w128, 1e-6, pos, rope, slots, kc[l], vc[l], perm, ctxl, hq, hkv, bs,
                                                128 ** -0.5, stamps, out=out, workspace=ws)
        else:
            ops.paged_attn_decode_fused(qkv, w128, w128, 1e-6, pos, rope, slots, kc[l], vc[l], perm, ctxl, hq, hkv, bs,
                                        128 ** -0.5, out=out, workspace=ws)

    def time_graph(stamped):
        for l in range(L):
            launch(l, stamped)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for l in range(L):
                launch(l, stamped)
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / (5 * L)

    us_plain = time_graph(False)
    us_stamped = time_graph(True)
    want = out.clone()  # the stamped graph's last launch: layer L - 1
    launch(L - 1, False)
    torch.cuda.synchronize()
    ts = stamps[: B * hkv].cpu().numpy().astype(np.int64)  # [workgroup][wave][8] of the last stamped launch (layer L-1)
    # the stamps are s_memrealtime: the chip-wide 100 MHz reference clock (10 ns per tick) - s_memtime's shader-cycle
    # counters are not aligned between compute units, so only a common clock puts 2048 waves on one time axis
    t0 = ts[:, :, 0].min()
    rel = ts - t0
    span = rel[:, :, 7].max()
    tick_us = 0.01  # 100 MHz
    byt = B * 2 * ctx * hkv * 128 * 2
    print(f"decode attention fused step, bs {B} x ctx {ctx}, {byt / 1e6:.1f} MB of K/V per launch")
    print(f"  product kernel {us_plain:.2f} us per launch ({byt / us_plain / 1e6:.2f} TB/s), instrumented kernel {us_stamped:.2f} us")
    print(f"  first wave's entry -> last wave done: {span * tick_us:.2f} us (stamps: s_memrealtime, 10 ns per tick)")
    print("  time since the earliest wave's entry [us]:   min     p10     p50     p90     max")
    for i, name in enumerate(NAMES):
        v = rel[:, :, i].reshape(-1) * tick_us
        print(f"    {i} {name:24s} {v.min():7.2f} {np.percentile(v, 10):7.2f} {np.percentile(v, 50):7.2f} "
              f"{np.percentile(v, 90):7.2f} {v.max():7.2f}")
    print("  phase durations per wave [us]:                p10     p50     p90     max")
    for i in range(1, 8):
        d = (ts[:, :, i] - ts[:, :, i - 1]).reshape(-1) * tick_us
        print(f"    {NAMES[i - 1]:22s} -> {NAMES[i]:22s} {np.percentile(d, 10):7.2f} {np.percentile(d, 50):7.2f} "
              f"{np.percentile(d, 90):7.2f} {d.max():7.2f}")
    # the prologue wave (wave 7) against the others at the barrier
    pro = (ts[:, 7, 2] - ts[:, 7, 0]) * tick_us
    oth = (ts[:, :7, 2] - ts[:, :7, 0]) * tick_us
    wait = (ts[:, :7, 3] - ts[:, :7, 2]) * tick_us
    print(f"  entry -> requests out: prologue wave p50 {np.percentile(pro, 50):.2f} us (norm + RoPE of the step's rows), "
          f"other waves p50 {np.percentile(oth, 50):.2f} us; they then wait p50 {np.percentile(wait, 50):.2f} us at the barrier")
    steady = (ts[:, :, 5] - ts[:, :, 4]).reshape(-1) * tick_us
    chunks = (ctx + 31) // 32 / 8
    print(f"  steady state: {np.percentile(steady, 50):.2f} us for ~{chunks - 1:.1f} chunks per wave = "
          f"{np.percentile(steady, 50) / max(chunks - 1, 1):.3f} us per 32-token chunk (16 KiB per wave)")
    assert torch.equal(out.view(torch.int16), want.view(torch.int16)), "instrumented kernel != product kernel"


if __name__ == "__main__":
    main()
