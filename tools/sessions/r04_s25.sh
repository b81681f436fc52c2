#!/bin/bash
O=gpurun_out/r04_s25; mkdir -p $O
python tools/host_step_bench.py 2>&1 | tee $O/host_step_bench.txt
python - <<'PY'
import cProfile, pstats, sys, os
sys.argv=["x"]
sys.path.insert(0, "nano-vllm-ascend_amd")
import random
from types import SimpleNamespace
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams
cfg = SimpleNamespace(max_num_seqs=32, max_num_batched_tokens=16384, max_model_len=4096, eos=-1, num_kvcache_blocks=4097, kvcache_block_size=16)
random.seed(0)
prompts = [[random.randint(0, 10000) for _ in range(1024)] for _ in range(32)]
sp = SamplingParams(temperature=1.0, max_tokens=8, ignore_eos=True)
pr = cProfile.Profile()
for _ in range(20):
    s = Scheduler(cfg)
    for p in prompts:
        q = Sequence(p, sp, block_size=16); q.prompt_hashes(16); s.add(q)
    pr.enable(); s.schedule(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
PY
