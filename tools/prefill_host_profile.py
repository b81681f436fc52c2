"""Host-side profile of the bench's two prefill steps (16 x 1024 tokens each): cProfile around LLMEngine.step(),
with the device time of the same step for comparison.  usage: python tools/prefill_host_profile.py"""
import cProfile
import os
import pstats
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from model_configs import QWEN3_0_6B  # noqa: E402
from nanovllm import LLM, SamplingParams  # noqa: E402


def main():
    llm = LLM.from_config_dict(QWEN3_0_6B, kvcache_block_size=16, max_num_seqs=32, max_model_len=4096,
                               max_num_batched_tokens=16384, num_kvcache_blocks=4096, synthetic_seed=0, sampling_seed=0)
    random.seed(0)
    sp = SamplingParams(temperature=1.0, max_tokens=8, ignore_eos=True, greedy=True)
    for rnd in range(2):  # the second round is the one reported (allocator and caches warm)
        prompts = [[random.randint(0, 10000) for _ in range(1024)] for _ in range(32)]
        seqs = [llm.add_request(p, sp) for p in prompts]
        prof = cProfile.Profile()
        walls = []
        while any(s.num_completion_tokens == 0 for s in seqs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prof.enable()
            llm.step()
            prof.disable()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        while not llm.is_finished():
            llm.step()
        if rnd == 1:
            print("prefill step wall times (ms):", [round(w, 2) for w in walls])
            pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
    llm.exit()


if __name__ == "__main__":
    main()
