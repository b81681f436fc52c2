"""The headline workload's prefill phase (32 x 1024-token prompts = two steps of 16 x 1024 tokens, Qwen3-0.6B) with and
without the engine's prefill lookahead (LLMEngine._step_prefill: the second step admitted, uploaded and queued while the
first runs), alternating in ONE process on ONE engine: wall time of the phase, time to first token of the two halves.
usage: python tools/prefill_lookahead_ab.py [rounds]"""
import os
import random
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from model_configs import QWEN3_0_6B  # noqa: E402
from nanovllm import LLM, SamplingParams  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    llm = LLM.from_config_dict(QWEN3_0_6B, kvcache_block_size=16, max_num_seqs=32, max_model_len=4096,
                               max_num_batched_tokens=16384, num_kvcache_blocks=4096, synthetic_seed=0, sampling_seed=0)
    default_min = llm.prefill_lookahead_min_tokens
    rows = {"queued": [], "one_at_a_time": []}
    try:
        for trial in range(2 * rounds + 2):
            mode = "queued" if trial % 2 == 0 else "one_at_a_time"
            llm.prefill_lookahead_min_tokens = default_min if mode == "queued" else 1 << 60
            random.seed(trial)
            sp = SamplingParams(temperature=1.0, max_tokens=2, ignore_eos=True, greedy=True)
            llm.ttft.clear()
            before = llm.prefill_lookahead_launches
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seqs = [llm.add_request([random.randint(0, 10000) for _ in range(1024)], sp) for _ in range(32)]
            t1 = time.perf_counter()
            steps = 0
            while any(s.num_completion_tokens == 0 for s in seqs):
                llm.step()
                steps += 1
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ttft = sorted(llm.ttft[s.seq_id] for s in seqs)
            while not llm.is_finished():
                llm.step()
            assert steps == 2 and llm.prefill_lookahead_launches - before == (mode == "queued")
            if trial >= 2:  # first pair: allocator growth, lazy module loads
                rows[mode].append({"phase_ms": (t2 - t1) * 1e3, "ttft_first_half_ms": ttft[0] * 1e3,
                                   "ttft_second_half_ms": ttft[-1] * 1e3, "ttft_p50_ms": statistics.median(ttft) * 1e3,
                                   "add_requests_ms": (t1 - t0) * 1e3})
        for mode, rs in rows.items():
            print(f"{mode} (median of {len(rs)} rounds):")
            for k in rs[0]:
                print(f"  {k:22s} {statistics.median(r[k] for r in rs):8.3f}   (min {min(r[k] for r in rs):8.3f})")
    finally:
        llm.exit()


if __name__ == "__main__":
    main()
