#!/usr/bin/env python3
"""End-to-end throughput on the reference's only published workload (reference: bench/bench.py:16-40).

    seed(0); 256 requests; prompt = randint(100, 1024) random token ids in [0, 10000];
    SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=randint(100, 1024)); max_model_len 4096;
    one warm-up generate(); then llm.generate(all prompts) timed with time.time();
    throughput = sum(max_tokens) / seconds            (prefill time included, output tokens only)

The same random draws in the same order as the reference script (Python's `random`, seed 0), so the token counts are
the reference's: 133 966 output tokens (README.md:338: "133 966 tok / 75.89 s").  `--max-num-seqs` is the knob the
reference's table varies (README.md:337-342: 1249 tok/s at 16 ... 3954 tok/s at 256 on an Ascend 910C - other
hardware, listed in BASELINE.md section 1 for context only).  Synthetic mode: a model directory holding only config.json
gets random weights (Config); the prompts are random ids either way, as in the reference.

    python bench/throughput_bench.py --model DIR [--max-num-seqs 256] [--num-seqs 256]
prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from random import randint, seed

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def workload(num_seqs: int = 256, max_input_len: int = 1024, max_output_len: int = 1024):
    """bench/bench.py:16-31, draw for draw."""
    seed(0)
    prompts = [[randint(0, 10000) for _ in range(randint(100, max_input_len))] for _ in range(num_seqs)]
    max_tokens = [randint(100, max_output_len) for _ in range(num_seqs)]
    return prompts, max_tokens


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", required=True, help="HF model directory (config.json only: synthetic weights)")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--max-num-seqs", type=int, default=256)
    ap.add_argument("--block-size", type=int, default=256, help="the reference's default kvcache_block_size")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--enforce-eager", action="store_true")
    args = ap.parse_args()

    from nanovllm import LLM, SamplingParams

    prompts, max_tokens = workload(args.num_seqs)
    llm = LLM(args.model, enforce_eager=args.enforce_eager, max_model_len=4096, max_num_seqs=args.max_num_seqs,
              kvcache_block_size=args.block_size, tensor_parallel_size=args.tensor_parallel_size)
    sps = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m) for m in max_tokens]
    try:
        llm.generate([[1, 2, 3, 4]], SamplingParams(max_tokens=8, ignore_eos=True), use_tqdm=False)  # warm-up (:33)
        t = time.time()
        outs = llm.generate(prompts, sps, use_tqdm=False)
        t = time.time() - t
    finally:
        llm.exit()
    total = sum(max_tokens)
    assert sum(len(o["token_ids"]) for o in outs) == total
    print(json.dumps({
        "metric": "end-to-end output tokens/s, reference bench/bench.py workload (prefill time included)",
        "throughput_tok_s": total / t, "output_tokens": total, "input_tokens": sum(len(p) for p in prompts),
        "seconds": t, "requests": args.num_seqs,
        "config": {"max_num_seqs": args.max_num_seqs, "block_size": args.block_size, "max_model_len": 4096,
                   "tensor_parallel_size": args.tensor_parallel_size, "temperature": 0.6,
                   "enforce_eager": args.enforce_eager, "model": os.path.basename(os.path.normpath(args.model))},
        "reference_context": "Ascend 910C, README.md:337-342: 1765 tok/s at max_num_seqs 32, 3954 tok/s at 256 (other hardware)",
    }), flush=True)


if __name__ == "__main__":
    main()
