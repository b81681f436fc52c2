#!/usr/bin/env python3
"""Where the microseconds of the decode chain go.

The six non-attention launches of a Qwen3-0.6B decode layer at bs 32 (add+RMSNorm over split-K partials, qkv GEMM,
o_proj split-K, add+RMSNorm, gate_up + SwiGLU, down split-K) replayed as ONE hipGraph over 28 layers' distinct weights,
as bench.py's chain_roofline does.  The launches of the LAST layer run as their instrumented instantiations
(mi_add_rmsnorm_splitk_ex, mi_gemm_bf16_packed_ex): every wave stamps s_memrealtime - the chip-wide 100 MHz clock, so
stamps of consecutive launches share one time base - at entry / loads issued / data arrived / sums in LDS / barrier
passed / stores issued / stores acknowledged.

Printed per launch, in microseconds:
  gap        first wave's entry minus the previous launch's last acknowledged store (the launch boundary itself)
  span       first entry -> last acknowledged store (the kernel's own time)
  per phase  when the waves reach it, on the launch's own clock (0 = first entry): min / p50 / max over all waves
and the chain's time per layer with and without the instrumented layer (the stamps must not change what they measure).

usage: python tools/chain_timeline.py [replays]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = torch.device("cuda:0")
PHASES = ["entry", "loads issued", "data arrived", "sums in LDS", "barrier passed", "stores issued", "stores acked"]
H, QKV, OD, INTER, B, L, KS = 1024, 4096, 2048, 3072, 32, 28, 4


def main():
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    mk = lambda n, k: ops.pack_weight((torch.randn(n, k, device=DEV) * 0.02).bfloat16())  # noqa: E731
    w_qkv = [mk(QKV, H) for _ in range(L)]
    w_o = [mk(H, OD) for _ in range(L)]
    w_gu = [mk(2 * INTER, H) for _ in range(L)]
    w_dn = [mk(H, INTER) for _ in range(L)]
    wn = torch.ones(H, device=DEV).bfloat16()
    res = torch.randn(B, H, device=DEV).bfloat16()
    parts0 = torch.randn(KS, B, H, device=DEV) * 0.1
    attn_out = torch.randn(B, OD, device=DEV).bfloat16()
    names = ["add_rmsnorm (input)", "qkv GEMM", "o_proj split-K 4", "add_rmsnorm (post-attn)", "gate_up GEMM + SwiGLU",
             "down split-K 4"]
    shapes = [(B, 4), (QKV // 16, 16), (H // 16 * KS, OD // KS // 64), (B, 4), (2 * INTER // 32, 16),
              (H // 16 * KS, INTER // KS // 64)]
    stamps = [torch.zeros(wg, wv, 8, dtype=torch.int64, device=DEV) for wg, wv in shapes]

    def layer(l, p, r, stamped):
        if stamped:
            x, r = ops.add_rmsnorm_splitk_stamped(p, r, wn, 1e-6, stamps[0])
            ops.gemm_packed_stamped(x, w_qkv[l], stamps[1])
            p2 = ops.gemm_packed_stamped(attn_out, w_o[l], stamps[2], ksplit=KS)
            x, r = ops.add_rmsnorm_splitk_stamped(p2, r, wn, 1e-6, stamps[3])
            act = ops.gemm_packed_stamped(x, w_gu[l], stamps[4], silu_mul=True)
            return ops.gemm_packed_stamped(act, w_dn[l], stamps[5], ksplit=KS), r
        x, r = ops.add_rmsnorm_splitk(p, r, wn, 1e-6)
        ops.gemm_packed(x, w_qkv[l])
        p2 = ops.gemm_packed_splitk(attn_out, w_o[l], KS)
        x, r = ops.add_rmsnorm_splitk(p2, r, wn, 1e-6)
        act = ops.gemm_packed(x, w_gu[l], silu_mul=True)
        return ops.gemm_packed_splitk(act, w_dn[l], KS), r

    def chain(stamp_last):
        p, r = parts0, res
        for l in range(L):
            p, r = layer(l, p, r, stamp_last and l == L - 1)
        return p, r

    def graph_of(stamp_last):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = chain(stamp_last)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = chain(stamp_last)
        return g, out

    def time_graph(g, n):
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n / L  # us per layer

    with torch.inference_mode():
        g_plain, out_plain = graph_of(False)
        g_stamp, out_stamp = graph_of(True)
        t_plain = [time_graph(g_plain, replays) for _ in range(3)]
        t_stamp = [time_graph(g_stamp, replays) for _ in range(3)]
        same = torch.equal(out_plain[0], out_stamp[0]) and torch.equal(out_plain[1].view(torch.int16), out_stamp[1].view(torch.int16))
        # the stamps of several replays: medians over replays of every statistic
        rounds = []
        for _ in range(replays):
            g_stamp.replay()
            torch.cuda.synchronize()
            rounds.append([s.cpu().numpy().astype(np.int64).copy() for s in stamps])

    print(f"decode chain, Qwen3-0.6B widths, bs {B}, {L} layers' weights, hipGraph replays: "
          f"{min(t_plain):.2f} us per layer (six launches); with the last layer instrumented {min(t_stamp):.2f}; "
          f"instrumented results identical: {same}")
    print("clock: s_memrealtime, 100 MHz (10 ns); statistics = median over", replays, "replays of the last layer's launches\n")
    summary = []
    for k, name in enumerate(names):
        gaps, spans, ph = [], [], []
        for r in rounds:
            st = r[k].reshape(-1, 8)[:, :7]
            t0 = st[:, 0].min()
            spans.append((st[:, 6].max() - t0) / 100.0)
            if k > 0:
                gaps.append((t0 - r[k - 1].reshape(-1, 8)[:, 6].max()) / 100.0)
            rel = (st - t0) / 100.0
            ph.append(np.stack([rel.min(0), np.percentile(rel, 50, axis=0), rel.max(0)]))
        ph = np.median(np.stack(ph), axis=0)
        gap = float(np.median(gaps)) if gaps else float("nan")
        span = float(np.median(spans))
        summary.append((name, gap, span))
        wg, wv = shapes[k]
        print(f"{name}: {wg} workgroups x {wv} waves   gap to previous launch {gap:.2f} us   span {span:.2f} us")
        for i, pn in enumerate(PHASES):
            print(f"    {pn:<15} min {ph[0, i]:6.2f}   p50 {ph[1, i]:6.2f}   max {ph[2, i]:6.2f}")
    tot_gap = sum(g for _, g, _ in summary[1:])
    tot_span = sum(s for _, _, s in summary)
    print(f"\nsum of the six spans {tot_span:.2f} us, of the five gaps between them {tot_gap:.2f} us "
          f"(+ one gap to the next layer's first launch): {tot_span + tot_gap * 6 / 5:.2f} us per layer by the stamps")


if __name__ == "__main__":
    main()
