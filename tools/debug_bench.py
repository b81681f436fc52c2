import os, sys, random, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/nano-vllm-ascend_amd", R + "/tests"]
import torch
from model_configs import QWEN3_0_6B, make_model_dir
from nanovllm import LLM, SamplingParams
print("init", flush=True)
llm = LLM(make_model_dir(QWEN3_0_6B), kvcache_block_size=16, max_num_seqs=32, max_model_len=4096,
          max_num_batched_tokens=16384, num_kvcache_blocks=4096, warmup=False,
          enforce_eager=os.environ.get("EAGER", "0") == "1")
torch.cuda.synchronize(); print("init done", flush=True)
random.seed(0)
n_p, plen = int(os.environ.get("NP", 32)), int(os.environ.get("PLEN", 1024))
prompts = [[random.randint(0, 10000) for _ in range(plen)] for _ in range(n_p)]
sp = SamplingParams(max_tokens=int(os.environ.get("MAXTOK", 8)), ignore_eos=True, greedy=True)
for p in prompts: llm.add_request(p, sp)
i = 0
while not llm.is_finished():
    out, n = llm.step(); torch.cuda.synchronize(); print("step", i, n, flush=True); i += 1
print("ok")
llm.exit()
