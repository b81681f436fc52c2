"""Where a prefill step of the headline workload (16 x 1024-token prompts, Qwen3-0.6B) spends its wall-clock time
outside the layer kernels: host scheduling + block allocation, metadata packing + upload, the eager launch sequence
running ahead of the device, the head + sampler, the token download.  Host clock stamps around the engine's own calls
plus HIP events on the compute stream; nothing in the engine is changed.
usage: python tools/prefill_host_timeline.py"""
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
    llm = LLM.from_config_dict(QWEN3_0_6B, kvcache_block_size=16, max_num_seqs=32, max_model_len=4096,
                               max_num_batched_tokens=16384, num_kvcache_blocks=4096, synthetic_seed=0, sampling_seed=0)
    runner, sched = llm.model_runner, llm.scheduler
    llm.prefill_lookahead_min_tokens = 1 << 60  # one step at a time: this is the timeline of a step on its own
    marks = {}

    def stamp(name):
        marks.setdefault(name, []).append(time.perf_counter())

    def wrap(obj, attr, before, after, event_after=None):
        fn = getattr(obj, attr)

        def inner(*a, **k):
            stamp(before)
            out = fn(*a, **k)
            stamp(after)
            if event_after is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.setdefault(event_after, []).append(ev)
            return out
        setattr(obj, attr, inner)

    wrap(sched, "schedule", "sched0", "sched1")
    wrap(runner, "prepare_prefill", "prep0", "prep1", "ev_prep")
    wrap(runner, "run_model", "model0", "model1", "ev_model")
    wrap(runner.sampler, "__call__", "samp0", "samp1", "ev_samp") if False else None
    rows = []
    try:
        for trial in range(4):
            marks.clear()
            random.seed(trial)
            sp = SamplingParams(temperature=1.0, max_tokens=2, ignore_eos=True, greedy=True)
            seqs = [llm.add_request([random.randint(0, 10000) for _ in range(1024)], sp) for _ in range(32)]
            steps = []
            while any(s.num_completion_tokens == 0 for s in seqs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                llm.step()
                t1 = time.perf_counter()
                steps.append((t0, t1))
            while not llm.is_finished():
                llm.step()
            for i, (t0, t1) in enumerate(steps):
                gpu = marks["ev_prep"][i].elapsed_time(marks["ev_model"][i]) * 1e3
                rows.append({"step_us": (t1 - t0) * 1e6,
                             "schedule_us": (marks["sched1"][i] - marks["sched0"][i]) * 1e6,
                             "before_schedule_us": (marks["sched0"][i] - t0) * 1e6,
                             "schedule_to_prepare_us": (marks["prep0"][i] - marks["sched1"][i]) * 1e6,
                             "prepare_us": (marks["prep1"][i] - marks["prep0"][i]) * 1e6,
                             "prepare_to_model_us": (marks["model0"][i] - marks["prep1"][i]) * 1e6,
                             "host_launch_us": (marks["model1"][i] - marks["model0"][i]) * 1e6,
                             "device_upload_to_logits_us": gpu,
                             "after_model_launch_us": (t1 - marks["model1"][i]) * 1e6,
                             "step_minus_device_us": (t1 - t0) * 1e6 - gpu})
        rows = rows[2:]  # first trial: allocator growth, lazy module loads
        print(f"prefill step, 16 x 1024 tokens, median of {len(rows)} steps (us):")
        for k in rows[0]:
            print(f"  {k:32s} {statistics.median(r[k] for r in rows):10.1f}")
    finally:
        llm.exit()


if __name__ == "__main__":
    main()
