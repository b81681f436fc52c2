#!/usr/bin/env python3
"""Device time of the captured prefill steps (ModelRunner.capture_prefill_graphs) of a Qwen3-0.6B-shaped engine, per
(token bucket, sequence bucket): 20 replays of each graph between HIP events.  What an open-loop arrival's prefill step
costs the device once its launch sequence is a graph replay.  BUCKETS=512x1,1024x1 restricts the run (under rocprofv3:
the per-kernel averages of just those steps)."""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "nano-vllm-ascend_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
from model_configs import QWEN3_0_6B, make_model_dir  # noqa: E402
from nanovllm import LLM  # noqa: E402


def main():
    only = {tuple(int(v) for v in b.split("x")) for b in os.environ.get("BUCKETS", "").split(",") if b}
    if only:  # capture nothing else: a kernel trace of the run then holds these steps alone
        from nanovllm.engine import model_runner as mrm

        mrm.PREFILL_GRAPH_TOKENS = tuple(sorted({t for t, _ in only}))
        mrm.PREFILL_GRAPH_SEQS = tuple(sorted({n for _, n in only}))
    llm = LLM(make_model_dir(QWEN3_0_6B), kvcache_block_size=16, max_num_seqs=64, max_num_batched_tokens=16384,
              max_model_len=4096, num_kvcache_blocks=4096, warmup=False, synthetic_seed=0)
    mr = llm.model_runner
    out = {}
    for (tb, sb), g in sorted(mr.prefill_graphs.items()):
        if only and (tb, sb) not in only:
            continue
        mr._stage_prefill_static([], tb, sb)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        out[f"{tb}x{sb}"] = round(s.elapsed_time(e) / 20, 4)
    print("captured prefill steps, ms per replay (tokens x sequences):")
    print(json.dumps(out, indent=1))
    llm.exit()


if __name__ == "__main__":
    main()
