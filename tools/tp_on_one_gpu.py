"""TP=2 engine run on one GPU (ranks over gloo): logits of the fused xGMI seam, the all-reduce kernel +
add_rmsnorm, and the gloo all-reduce against the TP=1 run."""
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from model_configs import MID, make_model_dir  # noqa: E402


def run(tp, fused, xgmi):
    from nanovllm import LLM, SamplingParams

    os.environ["MI355_DIST_BACKEND"] = "gloo"
    os.environ["MI355_XGMI_FUSED"] = "1" if fused else "0"
    os.environ["MI355_XGMI_ALLREDUCE"] = "1" if xgmi else "0"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    gen = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 4096, (n,), generator=gen).tolist() for n in (9, 33, 70)]
    sp = SamplingParams(max_tokens=6, ignore_eos=True, greedy=True)
    llm = LLM(make_model_dir(MID), kvcache_block_size=16, max_num_seqs=8, max_num_batched_tokens=1024,
              max_model_len=512, num_kvcache_blocks=64, enforce_eager=True, warmup=False, synthetic_seed=3,
              tensor_parallel_size=tp, hccl_port=port)
    try:
        outs = llm.generate(prompts, sp, use_tqdm=False)
        to = llm.model_runner.xgmi.timed_out() if llm.model_runner.xgmi is not None else None
        return [o["token_ids"] for o in outs], llm.model_runner.last_logits.float().cpu(), to
    finally:
        llm.exit()


if __name__ == "__main__":
    t1, l1, _ = run(1, False, False)
    for fused, xgmi in ((False, False), (False, True), (True, True), (True, True)):
        t, l, to = run(2, fused, xgmi)
        print(f"fused={fused} xgmi={xgmi}: max|dlogits| vs tp1 = {(l - l1).abs().max().item():.4f} tokens_equal={t == t1} timed_out={to}", flush=True)
