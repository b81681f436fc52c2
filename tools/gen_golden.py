#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference's Python modules are imported in place with two
in-memory stub modules for the Ascend-only packages it imports at module scope
(torch_npu, torchair); nothing from the reference is copied into this repo —
the fixtures hold inputs and the outputs the reference produced for them.

    python tools/gen_golden.py            # rewrites tests/golden/*

Fixtures
  hash_kats.json     xxh64 chained block hashes   (engine/block_manager.py:38-44)
  layers.npz         RMSNorm / add-RMSNorm / RoPE / SiluAndMul / Sampler-softmax I/O
  attention.npz      layers/attention_torch_native.py store + prefill + decode I/O
  moe_block.npz      models/qwen3_moe.py Qwen3MoeSparseMoeBlock I/O (router top-k captured, block output)
  engine_traces.json Scheduler + BlockManager + Sequence + ModelRunner.prepare_* traces
  tiny_model.npz     2-layer random Qwen3 driven through the reference's model,
                     scheduler and prepare_* (greedy = argmax of the reference logits);
                     tiny_model_bias.npz / tiny_model_llama.npz: the same run with attention_bias=True
                     and with the reference's LlamaForCausalLM
"""
from __future__ import annotations

import json
import os
import random
import sys
import types
from collections import deque
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NANOVLLM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


# --------------------------------------------------------------------------- import the reference
def import_reference():
    # import the HF stack first: accelerate probes for a real torch_npu via find_spec
    import accelerate  # noqa: F401
    import transformers  # noqa: F401
    from transformers import AutoConfig, AutoTokenizer, Qwen3Config  # noqa: F401
    import transformers.generation.utils  # noqa: F401
    from accelerate.utils.imports import is_npu_available
    from transformers.utils import is_torch_npu_available

    is_torch_npu_available()  # lru_cached: remember "no NPU" before the stubs exist
    is_npu_available(check_device=False)
    is_npu_available()
    import transformers.integrations.npu_flash_attention  # noqa: F401
    import transformers.models.llama.configuration_llama  # noqa: F401

    for name in ("torch_npu", "torchair", "torchair.configs", "torchair.configs.compiler_config",
                 "torchair.inference"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchair"].configs = sys.modules["torchair.configs"]
    sys.modules["torchair"].inference = sys.modules["torchair.inference"]
    sys.modules["torchair.configs"].compiler_config = sys.modules["torchair.configs.compiler_config"]
    sys.modules["torchair.configs.compiler_config"].CompilerConfig = type("CompilerConfig", (), {})
    sys.modules["torchair"].CompilerConfig = sys.modules["torchair.configs.compiler_config"].CompilerConfig
    sys.path.insert(0, REF)
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    import nanovllm  # noqa: F401  (the reference package)

    assert os.path.realpath(nanovllm.__file__).startswith(os.path.realpath(REF)), nanovllm.__file__


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> int16 numpy (bit pattern); other dtypes unchanged."""
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy()
    return t.contiguous().numpy()


# --------------------------------------------------------------------------- A. hashes
def gen_hashes():
    from nanovllm.engine.block_manager import BlockManager

    rng = random.Random(1)
    kats = []
    for n in (1, 3, 4, 16, 17, 256):
        toks = [rng.randrange(0, 151936) for _ in range(n)]
        h0 = BlockManager.compute_hash(toks)
        h1 = BlockManager.compute_hash(toks, h0)
        kats.append({"tokens": toks, "hash": h0, "chained_with_prefix": h0, "chained": h1})
    kats.append({"tokens": [1, 2, 3, 4], "hash": BlockManager.compute_hash([1, 2, 3, 4]),
                 "chained_with_prefix": BlockManager.compute_hash([1, 2, 3, 4]),
                 "chained": BlockManager.compute_hash([5, 6, 7, 8], BlockManager.compute_hash([1, 2, 3, 4])),
                 "chained_tokens": [5, 6, 7, 8]})
    with open(os.path.join(OUT, "hash_kats.json"), "w") as f:
        json.dump(kats, f)


# --------------------------------------------------------------------------- B. layers
def gen_layers():
    from nanovllm.layers.activation import SiluAndMul
    from nanovllm.layers.layernorm import RMSNorm
    from nanovllm.layers.rotary_embedding import RotaryEmbedding

    g = torch.Generator().manual_seed(1234)
    out = {}

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)

    # RMSNorm on hidden rows and on per-head views, with non-trivial weights
    for tag, shape in (("h1024", (5, 1024)), ("head128", (3, 6, 128)), ("h256", (4, 256))):
        n = RMSNorm(shape[-1], eps=1e-6)
        n.weight.data = (1.0 + 0.1 * torch.randn(shape[-1], generator=g)).to(torch.bfloat16)
        x = rnd(*shape, scale=3.0)
        r = rnd(*shape, scale=2.0)
        out[f"rms_{tag}_x"], out[f"rms_{tag}_w"], out[f"rms_{tag}_r"] = bits(x), bits(n.weight.data), bits(r)
        out[f"rms_{tag}_y"] = bits(n(x.clone()))
        y2, r2 = n(x.clone(), r.clone())
        out[f"rms_{tag}_addy"], out[f"rms_{tag}_addr"] = bits(y2), bits(r2)

    # RoPE, Qwen3 parameters (head 128, theta 1e6, 40960 positions)
    rope = RotaryEmbedding(128, 128, 40960, 1000000.0)
    pos = torch.tensor([0, 1, 2, 17, 1023, 4095, 40959], dtype=torch.int64)
    q = rnd(pos.numel(), 4, 128, scale=2.0)
    k = rnd(pos.numel(), 2, 128, scale=2.0)
    q2, k2 = rope(pos, q, k)
    out["rope_pos"], out["rope_q"], out["rope_k"] = pos.numpy(), bits(q), bits(k)
    out["rope_q_out"], out["rope_k_out"] = bits(q2), bits(k2)
    out["rope_table_rows"] = rope.cos_sin_cache[pos, 0].numpy()  # fp32 rows of the table at `pos`

    # SiluAndMul
    x = rnd(6, 2 * 384, scale=3.0)
    out["silu_x"], out["silu_y"] = bits(x), bits(SiluAndMul()(x))

    # Sampler front half (sampler.py:13-15): probabilities the multinomial draws from
    logits = rnd(3, 1000, scale=4.0)
    temps = torch.tensor([0.6, 1.0, 1.7])
    out["samp_logits"], out["samp_temps"] = bits(logits), temps.numpy()
    out["samp_probs"] = torch.softmax(logits.float() / temps.unsqueeze(-1), dim=-1).numpy()
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **out)


# --------------------------------------------------------------------------- C. native attention
def gen_attention():
    from nanovllm.layers.attention_torch_native import Attention as NativeAttention
    from nanovllm.utils.context import reset_context, set_context

    g = torch.Generator().manual_seed(4321)
    hq, hkv, d, bs, nblk = 4, 2, 128, 16, 48
    out = {"meta": np.array([hq, hkv, d, bs, nblk])}
    attn = NativeAttention(hq, d, hkv, kvcache_block_size=bs)
    attn.k_cache = torch.zeros(nblk, bs, hkv, d, dtype=torch.bfloat16)
    attn.v_cache = torch.zeros_like(attn.k_cache)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)

    # --- prefill: 3 sequences (ragged), shuffled block ids
    lens = [1, 17, 45]
    perm = torch.randperm(nblk, generator=g).tolist()
    tables, slots, cu = [], [], [0]
    for n in lens:
        nb = (n + bs - 1) // bs
        t = [perm.pop() for _ in range(nb)]
        tables.append(t)
        for i in range(n):
            slots.append(t[i // bs] * bs + i % bs)
        cu.append(cu[-1] + n)
    T = cu[-1]
    q, k, v = rnd(T, hq, d), rnd(T, hkv, d), rnd(T, hkv, d)
    set_context(True, cu_seqlens_q=torch.tensor(cu, dtype=torch.int32),
                cu_seqlens_k=torch.tensor(cu, dtype=torch.int32), max_seqlen_q=max(lens), max_seqlen_k=max(lens),
                slot_mapping=torch.tensor(slots, dtype=torch.int32), block_size=bs)
    o = attn(q, k, v)
    width = max(len(t) for t in tables)
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    out.update(pre_q=bits(q), pre_k=bits(k), pre_v=bits(v), pre_cu=np.array(cu, dtype=np.int32),
               pre_slots=np.array(slots, dtype=np.int32), pre_tables=bt.numpy(), pre_out=bits(o),
               pre_krows=bits(attn.k_cache.view(-1, hkv, d)[torch.tensor(slots)]),
               pre_vrows=bits(attn.v_cache.view(-1, hkv, d)[torch.tensor(slots)]),
               pre_cache_nonzero_rows=np.array(int((attn.k_cache.view(-1, hkv * d) != 0).any(dim=1).sum())))

    # --- decode: context lens spanning the block-boundary cases, fresh cache contents
    ctx = [1, 15, 16, 17, 100, 513]
    attn.k_cache = rnd(nblk, bs, hkv, d)
    attn.v_cache = rnd(nblk, bs, hkv, d)
    perm = torch.randperm(nblk, generator=g).tolist()
    tables, slots = [], []
    for n in ctx:
        nb = (n + bs - 1) // bs
        t = [perm.pop() for _ in range(nb)]
        tables.append(t)
        slots.append(t[-1] * bs + (n - 1) % bs)  # flat slot of the token being written
    width = max(len(t) for t in tables)
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    B = len(ctx)
    q, k, v = rnd(B, hq, d), rnd(B, hkv, d), rnd(B, hkv, d)
    kc0, vc0 = attn.k_cache.clone(), attn.v_cache.clone()
    set_context(False, slot_mapping=torch.tensor(slots, dtype=torch.int32),
                context_lens=torch.tensor(ctx, dtype=torch.int32), block_tables=bt, block_size=bs)
    o = attn(q, k, v)
    reset_context()
    out.update(dec_q=bits(q), dec_k=bits(k), dec_v=bits(v), dec_ctx=np.array(ctx, dtype=np.int32),
               dec_slots=np.array(slots, dtype=np.int32), dec_tables=bt.numpy(), dec_out=bits(o),
               dec_kcache_before=bits(kc0), dec_vcache_before=bits(vc0),
               dec_krows_after=bits(attn.k_cache.view(-1, hkv, d)[torch.tensor(slots)]),
               dec_vrows_after=bits(attn.v_cache.view(-1, hkv, d)[torch.tensor(slots)]),
               dec_cache_rows_changed=np.array(int(((attn.k_cache != kc0).view(-1, hkv * d)).any(dim=1).sum())))
    np.savez_compressed(os.path.join(OUT, "attention.npz"), **out)


# --------------------------------------------------------------------------- D. engine traces
def fake_runner(block_size, max_num_seqs, num_kvcache_blocks, max_model_len, enforce_eager):
    import nanovllm.engine.model_runner as mr

    # no accelerator here: drop pin_memory (model_runner.py:234,354-357)
    real_tensor = torch.tensor

    def tensor_nopin(*a, **kw):
        kw.pop("pin_memory", None)
        return real_tensor(*a, **kw)

    mr.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    mr.torch.tensor = tensor_nopin
    r = object.__new__(mr.ModelRunner)
    r.block_size = block_size
    r.device = "cpu"
    r.enforce_eager = enforce_eager
    r.config = SimpleNamespace(kvcache_block_size=block_size, max_num_seqs=max_num_seqs,
                               num_kvcache_blocks=num_kvcache_blocks, max_model_len=max_model_len)
    return r


def ctx_dump():
    from nanovllm.utils.context import get_context

    c = get_context()

    def tl(t):
        return None if t is None else t.tolist()

    return {"is_prefill": c.is_prefill, "cu_seqlens_q": tl(c.cu_seqlens_q), "cu_seqlens_k": tl(c.cu_seqlens_k),
            "max_seqlen_q": c.max_seqlen_q, "max_seqlen_k": c.max_seqlen_k, "slot_mapping": tl(c.slot_mapping),
            "context_lens": tl(c.context_lens), "block_tables": tl(c.block_tables), "real_bs": c.real_bs,
            "block_size": c.block_size}


def fake_token(seq, step):
    return (sum(seq.token_ids[-4:]) * 31 + 7 * step + len(seq)) % 1000 + 1


def run_scenario(name, block_size, num_kvcache_blocks, max_num_seqs, max_num_batched_tokens, max_model_len,
                 arrivals, padded, eos=-1):
    """arrivals: list of (step_index, prompt_tokens, max_tokens, ignore_eos)."""
    from nanovllm.engine.scheduler import Scheduler
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams

    cfg = SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=max_num_batched_tokens, eos=eos,
                          num_kvcache_blocks=num_kvcache_blocks, kvcache_block_size=block_size,
                          max_model_len=max_model_len, is_multimodal=False, hf_config=None)
    sched = Scheduler(cfg)
    runner = fake_runner(block_size, max_num_seqs, num_kvcache_blocks, max_model_len, enforce_eager=not padded)
    pending = deque(sorted(arrivals, key=lambda a: a[0]))
    seqs_by_id, order, steps, step = {}, [], [], 0
    while pending or not sched.is_finished():
        while pending and pending[0][0] <= step:
            _, toks, max_tokens, ignore_eos = pending.popleft()
            s = Sequence(toks, SamplingParams(temperature=1.0, max_tokens=max_tokens, ignore_eos=ignore_eos),
                         block_size=block_size)
            seqs_by_id[s.seq_id] = len(order)
            order.append(s)
            sched.add(s)
        if sched.is_finished():
            step += 1
            continue
        seqs, is_prefill = sched.schedule()
        rec = {"step": step, "is_prefill": is_prefill, "seqs": [seqs_by_id[s.seq_id] for s in seqs],
               "block_tables": [list(s.block_table) for s in seqs],
               "num_cached_tokens": [s.num_cached_tokens for s in seqs], "lens": [len(s) for s in seqs],
               "free_block_ids": list(sched.block_manager.free_block_ids),
               "waiting": [seqs_by_id[s.seq_id] for s in sched.waiting],
               "running": [seqs_by_id[s.seq_id] for s in sched.running]}
        if seqs:
            if is_prefill:
                ids, pos = runner.prepare_prefill(seqs)
            elif padded:
                ids, pos = runner.prepare_decode_padding(seqs)
            else:
                ids, pos = runner.prepare_decode(seqs)
            rec["input_ids"], rec["positions"], rec["context"] = ids.tolist(), pos.tolist(), ctx_dump()
            toks = [fake_token(s, step) for s in seqs]
            rec["sampled"] = toks
            sched.postprocess(seqs, toks)
            rec["finished"] = [seqs_by_id[s.seq_id] for s in seqs if s.is_finished]
        steps.append(rec)
        step += 1
        assert step < 2000
    return {"name": name,
            "config": {"block_size": block_size, "num_kvcache_blocks": num_kvcache_blocks,
                       "max_num_seqs": max_num_seqs, "max_num_batched_tokens": max_num_batched_tokens,
                       "max_model_len": max_model_len, "eos": eos, "padded": padded},
            "arrivals": [[a[0], list(a[1]), a[2], a[3]] for a in arrivals], "steps": steps,
            "final_tokens": [list(s.token_ids) for s in order],
            "final_cached": [s.num_cached_tokens for s in order]}


def gen_engine():
    rng = random.Random(7)
    common = [rng.randrange(1, 900) for _ in range(8)]
    scenarios = [
        # prefix sharing + reuse after free, block 16 is the min the reference Config allows,
        # but Scheduler/BlockManager accept any size: use 4 to keep traces readable
        run_scenario("prefix_share_b4", 4, 7, 4, 64, 64,
                     [(0, common + [11], 3, True), (0, common + [12], 3, True), (9, common + [13], 2, True)], False),
        # forced preemption: 3 sequences, too few blocks for all to finish decoding
        run_scenario("preempt_b4", 4, 9, 4, 64, 64,
                     [(0, [rng.randrange(1, 900) for _ in range(7)], 8, True),
                      (0, [rng.randrange(1, 900) for _ in range(6)], 8, True),
                      (0, [rng.randrange(1, 900) for _ in range(9)], 8, True)], False),
        # block-boundary crossings at the real block size, eager metadata
        run_scenario("boundary_b16_eager", 16, 64, 8, 256, 128,
                     [(0, [rng.randrange(1, 900) for _ in range(n)], 20, True) for n in (38, 43, 48, 15, 16, 17)],
                     False),
        # same through the graph-mode padded metadata (dummy slot in the reserved last block)
        run_scenario("boundary_b16_padded", 16, 64, 8, 256, 128,
                     [(0, [rng.randrange(1, 900) for _ in range(n)], 20, True) for n in (38, 43, 48)], True),
        # max_num_batched_tokens head-of-line blocking, staggered arrivals, EOS and max_model_len finishes
        run_scenario("budget_eos_b16", 16, 40, 3, 64, 40,
                     [(0, [rng.randrange(1, 900) for _ in range(30)], 30, False),
                      (0, [rng.randrange(1, 900) for _ in range(30)], 5, True),
                      (1, [rng.randrange(1, 900) for _ in range(10)], 30, False),
                      (3, [rng.randrange(1, 900) for _ in range(33)], 4, True),
                      (3, [rng.randrange(1, 900) for _ in range(5)], 6, True)], False, eos=500),
        # three prefill steps back to back, each closed by the token budget (what the engine's prefill lookahead
        # queues behind one another); a request that ends with its first token frees its blocks between two of them
        run_scenario("prefill_chain_b16", 16, 64, 4, 128, 128,
                     [(0, [rng.randrange(1, 900) for _ in range(n)], m, True)
                      for n, m in ((40, 6), (44, 1), (40, 5), (50, 4), (60, 1), (30, 3), (20, 7))], False),
        # the same chain with so few blocks that the later admissions depend on what the earlier steps free
        run_scenario("prefill_chain_tight_b16", 16, 12, 4, 128, 128,
                     [(0, [rng.randrange(1, 900) for _ in range(n)], m, True)
                      for n, m in ((40, 6), (44, 1), (40, 5), (50, 4), (60, 1), (30, 3), (20, 7))], True),
    ]
    with open(os.path.join(OUT, "engine_traces.json"), "w") as f:
        json.dump(scenarios, f, separators=(",", ":"))


def gen_engine_fuzz(n_scenarios=14):
    """Seeded random request streams under tight memory (shared prefixes, preemption + re-prefill,
    revival of freed blocks, staggered arrivals, EOS hits, graph-padded or eager metadata)."""
    out = []
    for seed in range(n_scenarios):
        rng = random.Random(1000 + seed)
        bs = rng.choice([4, 4, 16])
        max_num_seqs = rng.randrange(2, 6)
        longest = rng.choice([5, 9, 14]) * bs // 4 + rng.randrange(0, bs)
        prefixes = [[rng.randrange(1, 900) for _ in range(bs * rng.randrange(1, 3))] for _ in range(2)]
        arrivals = []
        for _ in range(rng.randrange(5, 10)):
            body = [rng.randrange(1, 900) for _ in range(rng.randrange(1, max(2, longest)))]
            toks = (rng.choice(prefixes) + body) if rng.random() < 0.55 else body
            toks = toks[: longest + bs]
            arrivals.append((rng.randrange(0, 12), toks, rng.randrange(2, 18), rng.random() < 0.7))
        max_model_len = (max(len(a[1]) for a in arrivals) + 20 + bs - 1) // bs * bs
        need_one = (max(len(a[1]) + a[2] for a in arrivals) + bs - 1) // bs + 1
        nblk = max(need_one + 1, rng.randrange(need_one, 3 * need_one)) + 1  # tight: forces preemption often
        budget = max(max_model_len, rng.choice([32, 64, 128]))
        out.append(run_scenario(f"fuzz{seed}_b{bs}", bs, nblk, max_num_seqs, budget, max_model_len, arrivals,
                                padded=rng.random() < 0.4, eos=rng.choice([-1, 500, 250])))
    with open(os.path.join(OUT, "engine_traces_fuzz.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    pre = sum(1 for sc in out if len([i for r in sc["steps"] if r["is_prefill"] for i in r["seqs"]]) >
              len({i for r in sc["steps"] if r["is_prefill"] for i in r["seqs"]}))
    hits = sum(1 for sc in out if any(sum(r["num_cached_tokens"]) > 0 for r in sc["steps"]))
    print(f"fuzz scenarios: {len(out)}, with preemption: {pre}, with prefix hits: {hits}, "
          f"steps: {sum(len(sc['steps']) for sc in out)}")


# --------------------------------------------------------------------------- D2. mixture-of-experts block
def gen_moe_block():
    """Qwen3MoeSparseMoeBlock (models/qwen3_moe.py:125-185) on random inputs: router logits, the top-k the block
    drew (captured from its torch.topk call), and its output, for two shapes (decode-sized and prefill-sized)."""
    import nanovllm.models.qwen3_moe as ref_moe

    out = {}
    for tag, (T, H, E, K, I) in {"small": (19, 128, 8, 2, 64), "wide": (70, 128, 16, 4, 64)}.items():
        g = torch.Generator().manual_seed({"small": 11, "wide": 12}[tag])
        cfg = SimpleNamespace(hidden_size=H, intermediate_size=4 * H, hidden_act="silu", num_experts=E,
                              num_experts_per_tok=K, moe_intermediate_size=I)
        torch.set_default_dtype(torch.bfloat16)
        blk = ref_moe.Qwen3MoeSparseMoeBlock(cfg)
        torch.set_default_dtype(torch.float32)
        gate_w = (torch.randn(E, H, generator=g) * 0.5).bfloat16()
        gu = (torch.randn(E, 2 * I, H, generator=g) * 0.08).bfloat16()
        dn = (torch.randn(E, H, I, generator=g) * 0.08).bfloat16()
        blk.gate.weight.data.copy_(gate_w)
        for e in range(E):
            blk.experts[e].gate_up_proj.weight.data.copy_(gu[e])
            blk.experts[e].down_proj.weight.data.copy_(dn[e])
        x = (torch.randn(T, H, generator=g) * 1.5).bfloat16()
        captured = {}
        real_topk = torch.topk

        def spy(*a, **kw):
            r = real_topk(*a, **kw)
            captured["w"], captured["ids"] = r[0].clone(), r[1].clone()
            return r

        torch.topk = spy
        try:
            with torch.inference_mode():
                y = blk(x)
        finally:
            torch.topk = real_topk
        out[f"{tag}_meta"] = np.array([T, H, E, K, I])
        out[f"{tag}_x"], out[f"{tag}_gate_w"], out[f"{tag}_gate_up_w"], out[f"{tag}_down_w"] = bits(x), bits(gate_w), bits(gu), bits(dn)
        out[f"{tag}_y"] = bits(y)
        out[f"{tag}_topk_ids"] = captured["ids"].numpy().astype(np.int64)
        out[f"{tag}_topk_prob"] = captured["w"].float().numpy()  # before the renormalisation (qwen3_moe.py:158-159)
    np.savez_compressed(os.path.join(OUT, "moe_block.npz"), **out)


# --------------------------------------------------------------------------- E. tiny model
def gen_tiny_model(variant: str = ""):
    """variant "": Qwen3 wiring (q/k norm, no bias); "bias": attention_bias=True - qkv bias, no q/k norm
    (the Qwen2 wiring of qwen3.py:70-72,135); "llama": the reference's LlamaForCausalLM (models/llama.py:
    neither q/k norm nor bias) on the same tiny shapes; "moe": its Qwen3MoeForCausalLM (models/qwen3_moe.py)
    with 8 experts, top-2, every layer sparse."""
    from transformers import LlamaConfig, Qwen3Config, Qwen3MoeConfig

    import nanovllm.models.llama as ref_llama
    import nanovllm.models.qwen3 as ref_qwen3
    import nanovllm.models.qwen3_moe as ref_moe
    from nanovllm.engine.scheduler import Scheduler
    from nanovllm.engine.sequence import Sequence
    from nanovllm.layers.attention_torch_native import Attention as NativeAttention
    from nanovllm.sampling_params import SamplingParams
    from nanovllm.utils.context import get_context, reset_context
    from oracle.model import OracleConfig, random_weights

    block_size, nblk = 16, 24

    class NativeAdapter(NativeAttention):
        """4-arg ctor of layers/attention.py:9 over the torch-native class; flattens 2-D decode slots."""

        def __init__(self, num_heads, head_dim, scaling, num_kv_heads):
            super().__init__(num_heads, head_dim, num_kv_heads, kvcache_block_size=block_size)

        def forward(self, q, k, v):
            c = get_context()
            if c.slot_mapping is not None and c.slot_mapping.dim() == 2:
                c.slot_mapping = c.slot_mapping[:, 0] * block_size + c.slot_mapping[:, 1]
            return super().forward(q, k, v)

    ref_qwen3.Attention = NativeAdapter
    ref_llama.Attention = NativeAdapter
    ref_moe.Attention = NativeAdapter
    sys.path.insert(0, os.path.join(REPO, "tests"))
    # the same dicts the tests build their model directories from
    from model_configs import TINY, TINY_LLAMA, TINY_LLAMA_HD64, TINY_MOE, TINY_QWEN2_HD64

    # "qwen2_hd64" / "llama_hd64": head_dim 64 with 7 / 4 query heads per kv head (Qwen2-0.5B's and Llama-3.2-1B's head
    # geometry: the plain-layout attention family of csrc/attn_plain.hip)
    tiny = {"bias": dict(TINY, attention_bias=True), "llama": TINY_LLAMA, "moe": TINY_MOE,
            "qwen2_hd64": TINY_QWEN2_HD64, "llama_hd64": TINY_LLAMA_HD64}.get(variant, TINY)
    cfg_cls = {"llama": LlamaConfig, "llama_hd64": LlamaConfig, "moe": Qwen3MoeConfig}.get(variant, Qwen3Config)
    hf = cfg_cls(**{k: v for k, v in tiny.items() if k not in ("architectures", "model_type", "torch_dtype")})
    cfg = OracleConfig.from_hf(hf)
    weights = random_weights(cfg, seed={"": 3, "bias": 13, "llama": 23, "moe": 33, "qwen2_hd64": 43, "llama_hd64": 53}[variant],
                             std=0.08)
    if variant == "moe":  # a wider router so that the top-2 choice is not a coin flip between near-equal logits
        for name in list(weights):
            if name.endswith("mlp.gate.weight"):
                weights[name] = (weights[name].float() * 6).to(torch.bfloat16)
    # non-trivial norm weights so the norm multiplies are exercised
    g = torch.Generator().manual_seed(5)
    for name in list(weights):
        if "norm" in name:
            weights[name] = (1.0 + 0.1 * torch.randn(weights[name].shape, generator=g)).to(torch.bfloat16)
    torch.set_default_dtype(torch.bfloat16)
    model = {"llama": ref_llama.LlamaForCausalLM, "llama_hd64": ref_llama.LlamaForCausalLM,
             "moe": ref_moe.Qwen3MoeForCausalLM}.get(variant, ref_qwen3.Qwen3ForCausalLM)(hf)
    torch.set_default_dtype(torch.float32)
    sd = dict(model.named_parameters())
    for name, w in weights.items():
        if ".mlp.experts." in name:  # stacked [E, ...] in the fixture, one module per expert in the reference
            for e in range(w.shape[0]):
                sd[name.replace(".experts.", f".experts.{e}.")].data.copy_(w[e])
        else:
            sd[name].data.copy_(w)
    # rope table must be fp32 (default dtype was bf16 while constructing)
    from nanovllm.layers.rotary_embedding import RotaryEmbedding
    rope = RotaryEmbedding(cfg.head_dim, cfg.head_dim, 512, cfg.rope_theta)
    li = 0
    for m in model.modules():
        if isinstance(m, NativeAdapter):
            m.k_cache = torch.zeros(nblk, block_size, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16)
            m.v_cache = torch.zeros_like(m.k_cache)
            li += 1
        if hasattr(m, "rotary_emb"):
            m.rotary_emb = rope
    assert li == 2

    sc = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=128, eos=-1, num_kvcache_blocks=nblk,
                         kvcache_block_size=block_size, max_model_len=128, is_multimodal=False, hf_config=None)
    sched = Scheduler(sc)
    runner = fake_runner(block_size, 4, nblk, 128, enforce_eager=True)
    rng = random.Random(11)
    prompts = [[rng.randrange(0, 256) for _ in range(n)] for n in (5, 17, 33)]
    seqs_all = []
    for p in prompts:
        s = Sequence(p, SamplingParams(temperature=1.0, max_tokens=14, ignore_eos=True), block_size=block_size)
        seqs_all.append(s)
        sched.add(s)
    out = {f"w::{k}": bits(v) for k, v in weights.items()}
    out["meta"] = np.array([block_size, nblk])
    out["prompt_lens"] = np.array([len(p) for p in prompts])
    out["prompts"] = np.array(sum(prompts, []), dtype=np.int64)
    step = 0
    with torch.inference_mode():
        while not sched.is_finished():
            seqs, is_prefill = sched.schedule()
            ids, pos = runner.prepare_prefill(seqs) if is_prefill else runner.prepare_decode(seqs)
            logits = model.compute_logits(model(ids, pos))
            toks = logits.float().argmax(dim=-1).tolist()
            reset_context()
            out[f"s{step}_prefill"] = np.array(int(is_prefill))
            out[f"s{step}_seqs"] = np.array([seqs_all.index(s) for s in seqs])
            out[f"s{step}_logits"] = bits(logits)
            out[f"s{step}_tokens"] = np.array(toks, dtype=np.int64)
            sched.postprocess(seqs, toks)
            step += 1
    out["n_steps"] = np.array(step)
    out["final_tokens"] = np.array(sum([s.token_ids for s in seqs_all], []), dtype=np.int64)
    out["attention_bias"] = np.array(int(variant in ("bias", "qwen2_hd64")))
    np.savez_compressed(os.path.join(OUT, f"tiny_model{'_' + variant if variant else ''}.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # fixed reduction order for reproducible fixtures
    import_reference()
    only = sys.argv[1:]  # e.g. `gen_golden.py engine_fuzz` regenerates one fixture
    for name, fn in (("hashes", gen_hashes), ("layers", gen_layers), ("attention", gen_attention),
                     ("engine", gen_engine), ("engine_fuzz", gen_engine_fuzz), ("tiny_model", gen_tiny_model),
                     ("tiny_model_bias", lambda: gen_tiny_model("bias")),
                     ("tiny_model_llama", lambda: gen_tiny_model("llama")), ("moe_block", gen_moe_block),
                     ("tiny_model_moe", lambda: gen_tiny_model("moe")),
                     ("tiny_model_qwen2_hd64", lambda: gen_tiny_model("qwen2_hd64")),
                     ("tiny_model_llama_hd64", lambda: gen_tiny_model("llama_hd64"))):
        if not only or name in only:
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f"{f}: {os.path.getsize(os.path.join(OUT, f))} B")


if __name__ == "__main__":
    main()
