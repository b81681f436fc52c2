#!/usr/bin/env python3
"""The decode chain in seven launches per layer against the same chain in five.

Qwen3-0.6B widths, bs 32, 28 layers' distinct weights, the attention launch left out (as bench.py's chain_roofline):

  seven: add+RMSNorm(split-K partials) -> qkv -> [attention] -> o_proj split-K 4 -> add+RMSNorm -> gate_up+SwiGLU -> down split-K 4
  five:  qkv with the norm on load     ->        [attention] -> o_proj + residual + statistic -> gate_up+SwiGLU with the norm
         on load -> down + residual + statistic                         (csrc/gemm_chain5_kernel.hpp)

Part 1: both chains as hipGraphs, alternating rounds, us per layer.  Part 2: the last layer of each chain instrumented
(s_memrealtime stamps in every wave): gap to the previous launch, span, when the waves reach each phase.

usage: python tools/chain5_ab.py [rounds] [replays]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-ascend_amd"))
from nanovllm import ops  # noqa: E402

DEV = torch.device("cuda:0")
PHASES = ["entry", "loads issued", "data arrived", "sums in LDS", "barrier passed", "stores issued", "stores acked"]
H, QKV, OD, INTER, L, KS = 1024, 4096, 2048, 3072, 28, 4
B = int(os.environ.get("ROWS", 32))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 20
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
    zeros = lambda wg, wv: torch.zeros(wg, wv, 8, dtype=torch.int64, device=DEV)  # noqa: E731
    rowchunks = (B + 7) // 8

    names7 = ["add_rmsnorm (input)", "qkv GEMM", "o_proj split-K 4", "add_rmsnorm (post-attn)", "gate_up GEMM + SwiGLU",
              "down split-K 4"]
    shapes7 = [(B, 4), (QKV // 16, 16), (H // 16 * KS, OD // KS // 64), (B, 4), (2 * INTER // 32, 16),
               (H // 16 * KS, INTER // KS // 64)]
    names5 = ["qkv GEMM, norm on load", "o_proj + residual + statistic", "gate_up GEMM + SwiGLU, norm on load",
              "down + residual + statistic"]
    shapes5 = [(QKV // 16, 16), (H // 16 * rowchunks, 16), (2 * INTER // 32, 16), (H // 16 * rowchunks, 16)]
    stamps7 = [zeros(*s) for s in shapes7]
    stamps5 = [zeros(*s) for s in shapes5]

    def layer7(l, state, stamped):
        p, r = state
        if stamped:
            x, r = ops.add_rmsnorm_splitk_stamped(p, r, wn, 1e-6, stamps7[0])
            ops.gemm_packed_stamped(x, w_qkv[l], stamps7[1])
            p2 = ops.gemm_packed_stamped(attn_out, w_o[l], stamps7[2], ksplit=KS)
            x, r = ops.add_rmsnorm_splitk_stamped(p2, r, wn, 1e-6, stamps7[3])
            act = ops.gemm_packed_stamped(x, w_gu[l], stamps7[4], silu_mul=True)
            return ops.gemm_packed_stamped(act, w_dn[l], stamps7[5], ksplit=KS), r
        x, r = ops.add_rmsnorm_splitk(p, r, wn, 1e-6)
        ops.gemm_packed(x, w_qkv[l])
        p2 = ops.gemm_packed_splitk(attn_out, w_o[l], KS)
        x, r = ops.add_rmsnorm_splitk(p2, r, wn, 1e-6)
        act = ops.gemm_packed(x, w_gu[l], silu_mul=True)
        return ops.gemm_packed_splitk(act, w_dn[l], KS), r

    def layer5(l, state, stamped):
        s, r, stat = state
        st = stamps5 if stamped else [None] * 4
        ops.gemm_normed(s, stat, wn, 1e-6, w_qkv[l], stamps=st[0])
        s, r, stat = ops.gemm_rowstat(attn_out, w_o[l], r, KS, stamps=st[1])
        act = ops.gemm_normed(s, stat, wn, 1e-6, w_gu[l], silu_mul=True, stamps=st[2])
        return ops.gemm_rowstat(act, w_dn[l], r, KS, stamps=st[3])

    def chain(layer, state, stamp_last):
        for l in range(L):
            state = layer(l, state, stamp_last and l == L - 1)
        return state

    def graph_of(layer, state0, stamp_last):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(layer, state0, stamp_last)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = chain(layer, state0, stamp_last)
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

    def timeline(g, stamps, names, shapes):
        reps = []
        for _ in range(replays):
            g.replay()
            torch.cuda.synchronize()
            reps.append([s.cpu().numpy().astype(np.int64).copy() for s in stamps])
        summary = []
        for k, name in enumerate(names):
            gaps, spans, ph = [], [], []
            for r in reps:
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
        n = len(names)
        tot_gap = sum(g for _, g, _ in summary[1:])
        tot_span = sum(s for _, _, s in summary)
        print(f"sum of the {n} spans {tot_span:.2f} us, of the {n - 1} gaps between them {tot_gap:.2f} us (+ one gap to the "
              f"next layer's first launch): {tot_span + tot_gap * n / (n - 1):.2f} us per layer by the stamps\n")

    with torch.inference_mode():
        # the five-launch chain starts from the state the seven-launch one starts from
        xn0, r0 = ops.add_rmsnorm_splitk(parts0, res, wn, 1e-6)
        s0, r5, stat0 = ops.gemm_rowstat(attn_out, w_o[0], res, KS)
        g7, out7 = graph_of(layer7, (parts0, res), False)
        g5, out5 = graph_of(layer5, (s0, r5, stat0), False)
        t7, t5 = [], []
        for _ in range(rounds):
            t7.append(round(time_graph(g7, replays), 2))
            t5.append(round(time_graph(g5, replays), 2))
        print(f"decode chain without its attention launch, Qwen3-0.6B widths, bs {B}, {L} layers' weights, hipGraph replays, "
              f"us per layer, {rounds} alternating rounds of {replays} replays:")
        print(f"  seven launches per layer (six here): min {min(t7):.2f}   median {sorted(t7)[len(t7) // 2]:.2f}   {t7}")
        print(f"  five launches per layer (four here): min {min(t5):.2f}   median {sorted(t5)[len(t5) // 2]:.2f}   {t5}")
        print()
        if B > 16:
            print("clock: s_memrealtime, 100 MHz (10 ns); statistics = median over", replays, "replays of the last layer's launches\n")
            g7s, _ = graph_of(layer7, (parts0, res), True)
            print(f"--- seven-launch chain ({time_graph(g7s, replays):.2f} us per layer with the last layer instrumented)")
            timeline(g7s, stamps7, names7, shapes7)
            g5s, _ = graph_of(layer5, (s0, r5, stat0), True)
            print(f"--- five-launch chain ({time_graph(g5s, replays):.2f} us per layer with the last layer instrumented)")
            timeline(g5s, stamps5, names5, shapes5)


def seam_32b():
    """the same fusion on the unsharded Qwen3-32B layer's attention seam (hidden 5120: o_proj
    5120 x 8192 -> add + RMSNorm -> gate_up 51200 x 5120 + SwiGLU), three launches against two.  (The layer's other seam
    has K = 25600 = 400 wave slices of 64, which the sixteen-wave producer does not divide: not built.)"""
    Hh, Ko, Ngu, Ls, ks = 5120, 8192, 51200, 4, 4
    torch.manual_seed(1)
    mk = lambda n, k: ops.pack_weight((torch.randn(n, k, device=DEV) * 0.02).bfloat16())  # noqa: E731
    w_o = [mk(Hh, Ko) for _ in range(Ls)]
    w_gu = [mk(Ngu, Hh) for _ in range(Ls)]
    wn = torch.ones(Hh, device=DEV).bfloat16()
    res = torch.randn(B, Hh, device=DEV).bfloat16()
    o = torch.randn(B, Ko, device=DEV).bfloat16()

    def three(l):
        x, _ = ops.add_rmsnorm_splitk(ops.gemm_packed_splitk(o, w_o[l], ks), res, wn, 1e-6)
        return ops.gemm_packed(x, w_gu[l], silu_mul=True)

    def two(l):
        s, _, stat = ops.gemm_rowstat(o, w_o[l], res, ks)
        return ops.gemm_normed(s, stat, wn, 1e-6, w_gu[l], silu_mul=True)

    def graph_time(fn, n=10):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for l in range(Ls):
                fn(l)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for l in range(Ls):
                fn(l)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n / Ls

    with torch.inference_mode():
        same = torch.equal(three(0).view(torch.int16), two(0).view(torch.int16))
        t3, t2 = [], []
        for _ in range(4):
            t3.append(round(graph_time(three), 1))
            t2.append(round(graph_time(two), 1))
    print(f"Qwen3-32B widths on one GPU, bs {B}, the attention seam of a layer (o_proj 5120 x 8192 -> norm -> gate_up 51200 x 5120 "
          f"+ SwiGLU), us per seam, alternating rounds; outputs equal bit for bit: {same}")
    print(f"  three launches (split-K 4 o_proj, add+RMSNorm, gate_up): min {min(t3):.1f}   {t3}")
    print(f"  two launches (o_proj + residual + statistic, gate_up with the norm on load): min {min(t2):.1f}   {t2}")


if __name__ == "__main__":
    if os.environ.get("SEAM") == "32b":
        seam_32b()
    else:
        main()
