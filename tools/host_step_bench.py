"""Host-side cost of a prefill step of the headline workload without any device: Scheduler.schedule() (block allocation
with chained hashes) and batch_meta.prefill_meta() for 16 x 1024-token prompts, two steps of 32 requests; plus the
pieces of BlockManager.allocate.  usage: python tools/host_step_bench.py"""
import os
import random
import statistics
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
from nanovllm.engine import batch_meta  # noqa: E402
from nanovllm.engine.block_manager import BlockManager  # noqa: E402
from nanovllm.engine.scheduler import Scheduler  # noqa: E402
from nanovllm.engine.sequence import Sequence  # noqa: E402
from nanovllm.sampling_params import SamplingParams  # noqa: E402

cfg = SimpleNamespace(max_num_seqs=32, max_num_batched_tokens=16384, max_model_len=4096, eos=-1, num_kvcache_blocks=4097,
                      kvcache_block_size=16)
random.seed(0)
prompts = [[random.randint(0, 10000) for _ in range(1024)] for _ in range(32)]
sp = SamplingParams(temperature=1.0, max_tokens=8, ignore_eos=True)


def trial():
    s = Scheduler(cfg)
    for p in prompts:
        q = Sequence(p, sp, block_size=16)
        q.prompt_hashes(16)  # as LLMEngine.add_request does, before the request's clock starts
        s.add(q)
    out = []
    for _ in range(2):
        t0 = time.perf_counter()
        seqs, _ = s.schedule()
        t1 = time.perf_counter()
        batch_meta.prefill_meta(seqs, 16, skip_cached=True)
        t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    return out


r = [trial() for _ in range(30)]
for step in range(2):
    print("step %d: schedule %.0f us, prefill_meta %.0f us" % (step, statistics.median(x[step][0] for x in r),
                                                               statistics.median(x[step][1] for x in r)))
ts = []
for _ in range(50):
    bm = BlockManager(4096, 16)
    seqs = [Sequence(p, sp, block_size=16) for p in prompts[:16]]
    for q in seqs:
        q.prompt_hashes(16)
    t0 = time.perf_counter()
    for q in seqs:
        bm.allocate(q)
    ts.append((time.perf_counter() - t0) * 1e6)
print("16 x BlockManager.allocate (64 blocks each, all misses): %.0f us" % statistics.median(ts))
