"""Qwen2-0.5B and Llama-3.2-1B at their full shapes (synthetic weights) through the engine: the models the reference's
README benchmarks beside Qwen3-0.6B.  Since round 4 both run on the fragment-native attention kernels (head_dim 64 tiles,
7 and 4 query heads per kv head); `plain_attention` in the output says which family the engine picked.  Prints one JSON
line per model: prefill rate of 16 x 1024-token prompts, decode step time and rate at bs 32 and bs 256 (hipGraph, greedy).
usage: python tools/small_models_run.py"""
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from model_configs import LLAMA_3_2_1B, QWEN2_0_5B  # noqa: E402
from nanovllm import LLM, SamplingParams  # noqa: E402


def run(name, cfg, batch, prompt_len=1024, steps=32):
    llm = LLM.from_config_dict(cfg, kvcache_block_size=16, max_num_seqs=batch, max_model_len=4096,
                               max_num_batched_tokens=16384, num_kvcache_blocks=batch * 80 + 64, synthetic_seed=0,
                               sampling_seed=0)
    try:
        random.seed(0)
        sp = SamplingParams(temperature=1.0, max_tokens=steps + 8, ignore_eos=True, greedy=True)
        seqs = [llm.add_request([random.randint(0, 10000) for _ in range(prompt_len)], sp) for _ in range(batch)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_prefill = 0
        while any(s.num_completion_tokens == 0 for s in seqs):
            llm.step()
            n_prefill += 1
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        for _ in range(4):
            llm.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            llm.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        toks = [s.completion_token_ids[:4] for s in seqs[:2]]
        return {"model": name, "batch": batch, "prompt_len": prompt_len, "prefill_steps": n_prefill,
                "prefill_tok_s": round(batch * prompt_len / t_prefill), "decode_ms_per_step": round(dt * 1e3, 3),
                "decode_tok_s": round(batch / dt), "plain_attention": bool(llm.model_runner.model.model.layers[0].self_attn.attn.plain),
                "first_tokens": toks}
    finally:
        llm.exit()


if __name__ == "__main__":
    for name, cfg in (("Qwen2-0.5B", QWEN2_0_5B), ("Llama-3.2-1B", LLAMA_3_2_1B)):
        for batch in (32, 256):
            print(json.dumps(run(name, cfg, batch)), flush=True)
