import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + s for s in ("", "/nano-vllm-ascend_amd", "/tests")]
import torch
from model_configs import TINY, make_model_dir
from nanovllm import LLM, SamplingParams
from nanovllm.engine import batch_meta
from oracle.model import OracleConfig, OracleQwen3, random_weights
from transformers import Qwen3Config
eager = os.environ.get("EAGER", "1") == "1"
llm = LLM(make_model_dir(TINY), kvcache_block_size=16, max_num_seqs=4, max_num_batched_tokens=256,
          max_model_len=128, num_kvcache_blocks=32, warmup=False, synthetic_seed=5, enforce_eager=eager)
hf = Qwen3Config(**{k: v for k, v in TINY.items() if k not in ("architectures", "model_type", "torch_dtype")})
ocfg = OracleConfig.from_hf(hf)
oracle = OracleQwen3(ocfg, random_weights(ocfg, seed=5), 32, 16)
if os.environ.get("GOLDEN"):
    import numpy as np
    from nanovllm.utils.loader import load_state_dict_packed
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/tiny_model.npz"))
    weights = {k[3:]: torch.from_numpy(g[k]).view(torch.bfloat16) for k in g.files if k.startswith("w::")}
    load_state_dict_packed(llm.model_runner.model, weights)
    oracle.w = weights
# weights equal?
sd = dict(llm.model_runner.model.named_parameters())
for k, v in oracle.w.items():
    d = (sd[k].detach().cpu().float() - v.float()).abs().max().item()
    if d != 0: print("WEIGHT MISMATCH", k, d)
plist = ([1, 2, 3, 4, 5], list(range(10, 43)), list(range(50, 67)))
if os.environ.get("GOLDEN"):
    flat, lens = g["prompts"].tolist(), g["prompt_lens"].tolist()
    plist, o = [], 0
    for n in lens:
        plist.append(flat[o:o + n]); o += n
for p in plist:
    llm.add_request(p, SamplingParams(max_tokens=4, ignore_eos=True, greedy=True))
mr = llm.model_runner
while not llm.is_finished():
    seqs, is_prefill = llm.scheduler.schedule()
    if is_prefill:
        m = batch_meta.prefill_meta(seqs, 16)
        want = oracle.prefill(*(torch.from_numpy(a) for a in (m.input_ids, m.positions, m.cu_seqlens_q, m.slot_mapping, m.block_tables)), fp32_logits=True)
    else:
        m = batch_meta.decode_meta(seqs)
        want = oracle.decode(*(torch.from_numpy(a) for a in (m.input_ids, m.positions, m.slot_mapping, m.context_lens, m.block_tables)), fp32_logits=True)
        print(" meta", m.input_ids.tolist(), m.positions.tolist(), m.slot_mapping.tolist(), m.context_lens.tolist(), m.block_tables.tolist())

    toks = llm.model_runner.call("run", seqs, is_prefill)
    if not is_prefill:
        print(" dev ", mr.dev["ids"][:3].tolist(), mr.dev["pos"][:3].tolist(), mr.dev["slots"][:3].tolist(), mr.dev["ctx"][:3].tolist(), mr.dev["tables"][:3].tolist())
    got = llm.model_runner.last_logits[: len(seqs)].float().cpu()
    print("prefill" if is_prefill else "decode", "err per seq", (got - want).abs().max(dim=1).values.tolist(), toks, want.argmax(-1).tolist())
    # KV cache check layer 0
    from kv_layout import to_logical
    for li in range(2):
        kc = to_logical(mr.kv_cache[0, li].cpu(), 16, False); vc = to_logical(mr.kv_cache[1, li].cpu(), 16, True)
        print("  layer", li, "kcache err", (kc.float() - oracle.k_cache[li].float()).abs().max().item(), "vcache err", (vc.float() - oracle.v_cache[li].float()).abs().max().item())
    llm.scheduler.postprocess(seqs, want.argmax(-1).tolist())
llm.exit()
