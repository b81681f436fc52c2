#!/usr/bin/env python3
"""Experiment: one decode step as (a) one 32-row chain vs (b) two 16-row chains on two streams
inside one hipGraph (micro-batch pipelining: one half's latency-bound kernels overlap the other
half's HBM-bound attention)."""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/nano-vllm-ascend_amd", R + "/tests"]
import torch
from model_configs import QWEN3_0_6B, make_model_dir
from nanovllm import LLM, SamplingParams
from nanovllm.utils.context import set_context, reset_context

llm = LLM(make_model_dir(QWEN3_0_6B), kvcache_block_size=16, max_num_seqs=32, max_model_len=4096,
          max_num_batched_tokens=16384, num_kvcache_blocks=4096, warmup=False)
random.seed(0)
prompts = [[random.randint(0, 10000) for _ in range(1024)] for _ in range(32)]
seqs = [llm.add_request(p, SamplingParams(max_tokens=64, ignore_eos=True, greedy=True)) for p in prompts]
for _ in range(6):
    llm.step()
mr = llm.model_runner
model = mr.model
d = mr.dev
B = 32
mr._fill_decode_stage(list(llm.scheduler.running), B)
torch.cuda.synchronize()

def ctx_for(lo, hi):
    set_context(False, slot_mapping=d["slots"][lo:hi], context_lens=d["ctx"][lo:hi], block_tables=d["tables"][lo:hi],
                is_enforce_eager=False, real_bs=hi - lo, block_size=16)

def single():
    ctx_for(0, B)
    return model.compute_logits(model(d["ids"][:B], d["pos"][:B]))

side = torch.cuda.Stream()
def dual(parts=2):
    cur = torch.cuda.current_stream()
    n = B // parts
    outs = []
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ctx_for(0, n)
        hA = model(d["ids"][:n], d["pos"][:n])
    ctx_for(n, B)
    hB = model(d["ids"][n:B], d["pos"][n:B])
    cur.wait_stream(side)
    h = torch.cat([hA, hB], 0)
    ctx_for(0, B)
    return model.compute_logits(h)

def bench(fn, name):
    with torch.inference_mode():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ref = fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1)/20:.3f} ms per step", flush=True)
        return out.float().clone()

a = bench(single, "single chain, 32 rows")
b = bench(dual, "two chains x 16 rows, two streams")
print("max |logit diff| between the two:", (a - b).abs().max().item())
reset_context()
llm.exit()
