#!/usr/bin/env python3
"""Open-loop serving benchmark on top of LLMEngine.step() (reference: bench/serving_bench.py).

Requests arrive by a Poisson process while the engine is stepping, so prefill steps of new arrivals
interleave with decode steps of running sequences (continuous batching, scheduler.py:41-77).  Per
request, with the reference's definitions (serving_bench.py:35-58):
    TTFT    = time of the first token  - submission time
    TPOT    = (completion time - time of the first token) / (output tokens - 1)
    latency = completion time - submission time
and over the run: output tokens / wall time.

Differences from the reference script, on purpose:
  * arrival gaps are exponential with mean 1/rate (a Poisson PROCESS).  The reference draws
    np.random.poisson(1 / rate) - integer gaps with mean 1/rate, almost all zero - and so submits its
    requests in bursts (:84-85); --reference-arrivals reproduces that.
  * the first-token time comes from the engine (LLMEngine.ttft: the end of the prefill step that produced the
    token) instead of "whatever is running after the step" (:112-115);
  * step() returns 4-tuples (llm_engine.py:124); the reference script still unpacks pairs (:118).
  * synthetic mode: with a model directory holding only config.json the weights are random (Config) and the
    prompts are random token ids, as in the reference script (:79).

    python bench/serving_bench.py --model DIR --num-requests 256 --request-rate 8 [--max-num-seqs 64]
prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from dataclasses import dataclass, field

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@dataclass
class RequestMetrics:
    input_len: int
    max_output_len: int
    submission_time: float = -1.0
    first_token_time: float = -1.0
    completion_time: float = -1.0
    output_len: int = -1

    @property
    def ttft(self) -> float:
        return self.first_token_time - self.submission_time

    @property
    def tpot(self) -> float:
        if self.output_len > 1:
            return (self.completion_time - self.first_token_time) / (self.output_len - 1)
        return float("nan")

    @property
    def latency(self) -> float:
        return self.completion_time - self.submission_time


def arrival_times(num_requests: int, rate: float, rng: np.random.Generator, reference_style: bool = False) -> np.ndarray:
    """Seconds after the start at which request i is submitted."""
    if rate <= 0:  # everything at once (closed batch)
        return np.zeros(num_requests)
    if reference_style:
        return np.cumsum(rng.poisson(1.0 / rate, num_requests)).astype(np.float64)
    return np.cumsum(rng.exponential(1.0 / rate, num_requests))


@dataclass
class ServingResult:
    metrics: dict = field(default_factory=dict)
    total_time: float = 0.0
    steps: int = 0

    def summary(self) -> dict:
        done = [m for m in self.metrics.values() if m.completion_time >= 0]
        ttft = np.array([m.ttft for m in self.metrics.values() if m.first_token_time >= 0])
        tpot = np.array([m.tpot for m in done if not np.isnan(m.tpot)])
        out_tokens = sum(m.output_len for m in done)

        def pct(a, q):
            return float(np.percentile(a, q)) if len(a) else float("nan")

        return {"requests": len(self.metrics), "completed": len(done), "total_time_s": self.total_time,
                "engine_steps": self.steps, "input_tokens": sum(m.input_len for m in self.metrics.values()),
                "output_tokens": out_tokens, "throughput_tok_s": out_tokens / self.total_time if self.total_time else 0.0,
                "ttft_ms": {"mean": float(ttft.mean()) * 1e3 if len(ttft) else float("nan"), "p50": pct(ttft, 50) * 1e3,
                            "p99": pct(ttft, 99) * 1e3},
                "tpot_ms": {"mean": float(tpot.mean()) * 1e3 if len(tpot) else float("nan"), "p50": pct(tpot, 50) * 1e3,
                            "p99": pct(tpot, 99) * 1e3},
                "latency_s": {"mean": float(np.mean([m.latency for m in done])) if done else float("nan")}}


def run_serving(engine, prompts, sampling_params, arrivals, clock=time.perf_counter, sleep=time.sleep) -> ServingResult:
    """Drive `engine` (add_request / step / is_finished / ttft / scheduler) through the arrival schedule.
    `clock` / `sleep` are injectable so the loop can be tested against a scripted engine."""
    res = ServingResult()
    sent, n = 0, len(prompts)
    start = clock()
    while sent < n or not engine.is_finished():
        now = clock()
        while sent < n and now - start >= arrivals[sent]:
            seq = engine.add_request(prompts[sent], sampling_params[sent])
            res.metrics[seq.seq_id] = RequestMetrics(len(prompts[sent]), sampling_params[sent].max_tokens,
                                                     submission_time=getattr(seq, "arrival_time", now))
            sent += 1
        if engine.is_finished():  # nothing to run yet: wait for the next arrival
            sleep(min(0.01, max(0.0, arrivals[sent] - (clock() - start))))
            continue
        finished, _ = engine.step()
        res.steps += 1
        t = clock()
        for seq_id, first in list(engine.ttft.items()):  # first tokens produced by this step
            m = res.metrics.get(seq_id)
            if m is not None and m.first_token_time < 0:
                m.first_token_time = m.submission_time + first
        for seq_id, token_ids, *_ in finished:
            m = res.metrics.get(seq_id)
            if m is not None:
                m.completion_time, m.output_len = t, len(token_ids)
    res.total_time = clock() - start
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", required=True, help="HF model directory (config.json only: synthetic weights)")
    ap.add_argument("--num-requests", type=int, default=256)
    ap.add_argument("--request-rate", type=float, default=8.0, help="requests per second; 0: all at once")
    ap.add_argument("--max-input-len", type=int, default=1024)
    ap.add_argument("--max-output-len", type=int, default=1024)
    ap.add_argument("--max-num-seqs", type=int, default=256)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--temperature", type=float, default=0.6)
    ap.add_argument("--reference-arrivals", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    from nanovllm import LLM, SamplingParams

    rng = np.random.default_rng(args.seed)
    llm = LLM(args.model, enforce_eager=False, max_model_len=4096, max_num_seqs=args.max_num_seqs,
              kvcache_block_size=args.block_size, tensor_parallel_size=args.tensor_parallel_size)
    prompts = [rng.integers(0, 10000, int(rng.integers(100, args.max_input_len + 1))).tolist()
               for _ in range(args.num_requests)]
    sps = [SamplingParams(temperature=args.temperature, ignore_eos=True,
                          max_tokens=int(rng.integers(100, args.max_output_len + 1))) for _ in range(args.num_requests)]
    arrivals = arrival_times(args.num_requests, args.request_rate, rng, args.reference_arrivals)
    try:
        res = run_serving(llm, prompts, sps, arrivals)
    finally:
        llm.exit()
    out = res.summary()
    out["config"] = {k: v for k, v in vars(args).items()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
