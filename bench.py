#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s (+ p50 TTFT) of Qwen3-0.6B, bs=32, seq=1024, block 16,
bf16, paged decode under hipGraph — BASELINE.json configs[1] — on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W          # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--mode replicas|tp]

`python bench.py --gpus N` without torchrun around it starts the N ranks itself (same command line).
N > 1 measures BOTH ways of using N GPUs, back to back, and reports them in ONE line (--mode picks one):
  replicas            the headline `value`: the decode batch is a set of independent sequences, so the path shards
                      with NO data-path collective: every rank runs the whole 0.6 B model on its own GPU over its
                      own 32 sequences (global batch 32 N, weak scaling); `value` is the sum over ranks of
                      tokens / the slowest rank's time (barrier + synchronize on both sides, max over ranks).
  tp   (`tp_run`)     Megatron tensor parallelism over RCCL/xGMI at the fixed batch of 32 (strong scaling):
                      what a model that does not fit one GPU needs (BASELINE configs[2]); for this 0.6 B
                      model every all-reduce is a 64 KiB latency-bound message, so it does not speed up.
                      Carries its own value, ms_per_step, world_seen, backend, xgmi_selftest and the
                      per-rank kernel rooflines; runs under a watchdog, so a failure there costs the
                      `tp_run` object, never the line.

A "step" is one engine decode step over the batch of 32 sequences (scheduler -> metadata
-> graph replay incl. sampling -> postprocess), i.e. 32 new tokens.  On one GPU the engine keeps one step queued
behind the running one (decode_lookahead): the timed region starts - after synchronize - with one
step finished and not yet consumed, and ends - after synchronize - with one finished and not consumed, so exactly
`--steps` steps of device work and of host work lie inside it.  Inputs are synthetic
(random-init Qwen3-0.6B-shaped weights N(0,0.02^2), random prompt ids in [0,10000],
random.seed(0)); the 32 x 1024-token prompts are prefilled through the
engine first (that is where p50 TTFT comes from), so the KV cache is resident in HBM
when the timed region starts.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  prefill_roofline  MFMA flops of one prefill step (16 x 1024 tokens: projections, causal attention, head) / the wall
                time of the whole prefill phase (host included: first step() call -> the last step's first tokens on
                the host; the engine queues the second step behind the first) divided by its steps, vs the 2.5 PFLOP/s
                dense bf16 peak - the TTFT half of the metric;
  roofline      the dominant kernel (paged_attn_decode, the fused step form the engine runs) timed live
                with HIP events on its launch stream over the engine's real KV cache: algorithmic KV
                bytes per launch / average duration vs the 8 TB/s HBM peak; `traffic` = those bytes x
                the PMC ratio of the newest profiles/r*_attn_traffic.json;
  chain_roofline the six non-attention launches of a layer over all layers' weights;
  step_roofline the whole decode step against BASELINE.md's bytes(B, ctx) model;
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's arithmetic) timed on
                this host on a bounded sample of the same workload;
  prefill_steps_ms  per prefill step: when its launch sequence started, the host's launch time, the device time (HIP
                events around the step on the launch stream), when its first tokens were stamped - the line can say
                where a TTFT went;
  gc            the garbage collector during the run (engine/host_gc.py): collections inside the prefill phase and
                the timed region with their generation and duration - no full pass may run inside a step;
  speedup_vs_1gpu (N > 1)  `value` (and `tp_run.value`) as a multiple of the newest committed one-GPU line.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import statistics
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "nano-vllm-ascend_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
BATCH, PROMPT_LEN, BLOCK = 32, 1024, 16


def step_bytes(batch: int, ctx: int, tp: int = 1) -> float:
    """algorithmic HBM bytes of one decode step (whole job, all ranks)."""
    return 1_192_099_840 + batch * (114_688 * ctx + 418_560)


def cpu_baseline(sample_steps: int = 5):
    """The oracle's decode step at bs=32 / ctx=1024 on the host cores (random KV contents
    instead of a CPU prefill: same arithmetic per step, bounded run time)."""
    import torch
    from transformers import Qwen3Config

    from model_configs import QWEN3_0_6B
    from oracle.model import OracleConfig, OracleQwen3, random_weights

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(cores, 32)))  # the cores this process may use, capped
    hf = Qwen3Config(**{k: v for k, v in QWEN3_0_6B.items() if k not in ("architectures", "model_type", "torch_dtype")})
    cfg = OracleConfig.from_hf(hf)
    blocks_per_seq = PROMPT_LEN // BLOCK + 1
    model = OracleQwen3(cfg, random_weights(cfg, seed=0), BATCH * blocks_per_seq, BLOCK)
    g = torch.Generator().manual_seed(0)
    # random KV contents (one layer's worth, repeated: values do not affect the timing)
    layer = torch.randn(model.k_cache.shape[1:], generator=g).to(model.k_cache.dtype)
    model.k_cache.copy_(layer.unsqueeze(0).expand_as(model.k_cache))
    model.v_cache.copy_(layer.flip(0).unsqueeze(0).expand_as(model.v_cache))
    tables = torch.arange(BATCH * blocks_per_seq, dtype=torch.int32).view(BATCH, blocks_per_seq)
    ids = torch.randint(0, 10000, (BATCH,), generator=g)
    t0 = time.perf_counter()
    for s in range(sample_steps):
        n = PROMPT_LEN + 1 + s
        pos = torch.full((BATCH,), n - 1, dtype=torch.int64)
        ctx = torch.full((BATCH,), n, dtype=torch.int32)
        slot = torch.stack([tables[:, (n - 1) // BLOCK], torch.full((BATCH,), (n - 1) % BLOCK, dtype=torch.int32)], 1)
        logits = model.decode(ids, pos, slot, ctx, tables)
        ids = logits.float().argmax(-1)
    dt = time.perf_counter() - t0
    return {"value": BATCH * sample_steps / dt, "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{sample_steps} decode step(s) of bs={BATCH} ctx={PROMPT_LEN} "
            f"(oracle/ CPU port, random KV contents), {dt:.1f} s; threads capped at 32 of {cores} host cores"}


def attention_roofline(llm, seqs, iters: int = 20):
    import torch

    with torch.inference_mode():
        return _attention_roofline(llm, seqs, iters)


def _attention_roofline(llm, seqs, iters):
    """Time mi_paged_attn_decode (+ its split-merge) alone, on the engine's KV cache and block
    tables, cycling over all layers so successive launches never hit the Infinity Cache."""
    import torch

    from nanovllm import ops

    mr = llm.model_runner
    dev = mr.device
    real = len(seqs)
    # KV rows exist for every token but the one just sampled: attend len-1 tokens per sequence
    ctx_host = [len(x) - 1 for x in seqs]
    width = max(len(x.block_table) for x in seqs)
    tables = torch.tensor([x.block_table + [-1] * (width - len(x.block_table)) for x in seqs],
                          dtype=torch.int32, device=dev)
    ctx = torch.tensor(ctx_host, dtype=torch.int32, device=dev)
    if os.environ.get("BENCH_ATTN_RANDOM_TABLES"):  # experiment: same cache, scattered block ids
        nblk = llm.config.num_kvcache_blocks
        tables = torch.randperm(nblk - 1)[: real * width].to(torch.int32).view(real, width).to(dev)
    attn_mods = [m for m in mr.model.modules() if hasattr(m, "k_cache") and hasattr(m, "v_cache")]
    a0 = attn_mods[0]
    out = torch.empty(real, a0.num_heads * 128, dtype=torch.bfloat16, device=dev)
    ws = ops.attn_workspace(dev, real, a0.num_heads)
    fused = os.environ.get("MI355_ATTN_FUSED", "1") != "0"
    if fused:
        # the launch the engine's decode step uses: q/k-norm + RoPE + KV store + attention straight from a
        # packed qkv row.  Position ctx - 1 of every sequence is re-written with that (random) row - the run is
        # over, and this is exactly the store a decode step performs.
        qkv = torch.randn(real, (a0.num_heads + 2 * a0.num_kv_heads) * 128, device=dev).bfloat16()
        pos = torch.tensor([c - 1 for c in ctx_host], dtype=torch.int64, device=dev)
        slots = torch.tensor([[int(tables[i][(c - 1) // mr.block_size]), (c - 1) % mr.block_size]
                              for i, c in enumerate(ctx_host)], dtype=torch.int32, device=dev)
        layers = [l.self_attn for l in mr.model.model.layers]

        def launch_all():
            for sa in layers:
                m = sa.attn
                ops.paged_attn_decode_fused(qkv, sa.q_norm.weight, sa.k_norm.weight, sa.rms_norm_eps, pos,
                                            sa.rotary_emb.cos_sin_cache, slots, m.k_cache, m.v_cache, tables, ctx,
                                            m.num_heads, m.num_kv_heads, mr.block_size, m.scale, out=out, workspace=ws)
    else:
        q = torch.randn(real, a0.num_heads * 128, device=dev).bfloat16()

        def launch_all():
            for m in attn_mods:
                ops.paged_attn_decode(q, m.k_cache, m.v_cache, tables, ctx, m.num_heads, m.num_kv_heads,
                                      mr.block_size, m.scale, out=out, workspace=ws)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch_all()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):  # graph replay: no host launch gaps between the launches
        launch_all()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    dur = s.elapsed_time(e) * 1e-3 / (iters * len(attn_mods))
    ctx_sum = int(sum(ctx_host))
    algo = ctx_sum * 2 * a0.num_kv_heads * 128 * 2  # K+V rows of every context token, bf16
    # HBM bytes per launch: the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 x2
    # read correction) of this same loop, committed under profiles/, give measured / algorithmic bytes for
    # this kernel; the ratio is a property of the access pattern (every K/V tile of the context read once),
    # so it is applied to this run's algorithmic bytes whatever --steps made the contexts
    traffic, traffic_src, traffic_ctx_sum = None, None, None
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_attn_traffic.json")))[-1]
        with open(newest) as f:
            pmc = json.load(f)
        traffic = pmc["traffic_over_algorithmic"] * algo
        traffic_ctx_sum = pmc["ctx_sum"]
        traffic_src = f"{os.path.basename(newest)}: measured/algorithmic = {pmc['traffic_over_algorithmic']:.4f} at ctx_sum {pmc['ctx_sum']}"
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return {"kernel": "paged_attn_decode_kernel" + (" (fused step: q/k-norm + RoPE + KV store + attention)" if fused else ""),
            "bound": "hbm", "achieved": algo / dur / 1e9,
            "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": algo / dur / HBM_PEAK, "traffic": traffic,
            "traffic_kind": "derived: this run's algorithmic bytes x the PMC-measured traffic ratio of the committed "
                            "profile (not a counter of this run)",
            "traffic_source": traffic_src,
            # the contexts the counters were collected at, next to this run's: `traffic` is the profile's ratio applied to
            # this run's bytes - two different ctx_sums here say so at a glance
            "traffic_ctx_sum": traffic_ctx_sum, "ctx_sum": ctx_sum,
            "bytes_per_launch": algo, "avg_launch_us": dur * 1e6, "launches_timed": iters * len(attn_mods)}


def chain_roofline(llm, batch: int, iters: int = 20):
    """The six launches per layer that are NOT attention (add+RMSNorm x2, qkv / o_proj / gate_up+SwiGLU /
    down_proj GEMMs) replayed as one graph over all layers' real weights, HIP events on the launch stream:
    algorithmic bytes = the layers' weights (activations excluded, as BASELINE.md §2) / time."""
    import torch

    from nanovllm import ops

    with torch.inference_mode():
        model = llm.model_runner.model.model
        dev = llm.model_runner.device
        hidden = model.embed_tokens.weight.shape[1]
        ks = [(model._ksplit(l.self_attn.o_proj.weight), model._ksplit(l.mlp.down_proj.weight)) for l in model.layers]
        res = torch.randn(batch, hidden, device=dev).bfloat16()
        parts = torch.randn(ks[0][1], batch, hidden, device=dev) * 0.1
        o = torch.randn(batch, model.layers[0].self_attn.o_proj.weight.shape[1], device=dev).bfloat16()

        def run():
            p = parts
            for layer, (ko, kd) in zip(model.layers, ks):
                attn, mlp = layer.self_attn, layer.mlp
                x, r = ops.add_rmsnorm_splitk(p, res, layer.input_layernorm.weight, layer.input_layernorm.eps)
                ops.gemm_packed(x, attn.qkv_proj.weight_packed)
                p2 = ops.gemm_packed_splitk(o, attn.o_proj.weight_packed, ko)
                x, r = ops.add_rmsnorm_splitk(p2, r, layer.post_attention_layernorm.weight,
                                              layer.post_attention_layernorm.eps)
                act = ops.gemm_packed(x, mlp.gate_up_proj.weight_packed, silu_mul=True)
                p = ops.gemm_packed_splitk(act, mlp.down_proj.weight_packed, kd)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            run()
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        dur = s.elapsed_time(e) * 1e-3 / iters
        wbytes = sum(w.numel() * 2 for l in model.layers
                     for w in (l.self_attn.qkv_proj.weight, l.self_attn.o_proj.weight, l.mlp.gate_up_proj.weight,
                               l.mlp.down_proj.weight))
        n = len(model.layers)
        return {"kernels": "add_rmsnorm_splitk x2, gemm_skinny qkv / o_proj split-K / gate_up+SwiGLU / down_proj split-K",
                "bound": "hbm", "achieved": wbytes / dur / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": wbytes / dur / HBM_PEAK, "bytes_per_layer": wbytes // n, "us_per_layer": dur / n * 1e6,
                "launches_per_layer": 6, "layers": n}


PREFILL_MFMA_PEAK = 2.5e15  # dense bf16 MFMA flop/s (MI355X_MICROARCH.md)


def prefill_flops(n_seqs: int, seq_len: int, cfg: dict) -> float:
    """MFMA flops of one prefill step of n_seqs x seq_len tokens: the four projections of every layer, causal
    attention (QK^T and PV over the lower triangle incl. the diagonal), the head GEMM on the last tokens."""
    h, inter, layers = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    hq, hkv, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    per_token = 2 * (h * (hq + 2 * hkv) * d + hq * d * h + 3 * h * inter)
    attn = 2 * 2 * hq * d * (seq_len * (seq_len + 1) // 2)  # per sequence and layer
    return layers * (n_seqs * seq_len * per_token + n_seqs * attn) + 2 * n_seqs * h * cfg["vocab_size"]


def one_gpu_reference():
    """The newest committed one-GPU line of this bench (profiles/r*bench*.json, n_gpus == 1, the same workload): what
    north_star's ">= 6x aggregate at 8 GPUs" is a multiple of.  -> (tokens/s, file name) or (None, None)."""
    import glob

    best = (None, None, None)
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*bench*.json"))):
        try:
            with open(path) as f:
                line = json.loads(f.read().strip().splitlines()[-1])
            if line.get("n_gpus") == 1 and line.get("config", {}).get("batch") == BATCH and not line.get("dry_run"):
                # newest round first; within a round the file the round calls final (then by name) - never the file
                # time, which is arbitrary after a checkout
                name = os.path.basename(path)
                key = (name[:3], "final" in name, name)
                if best[0] is None or key > best[2]:
                    best = (float(line["value"]), name, key)
        except (OSError, ValueError, KeyError, IndexError, TypeError):
            continue
    return best[0], best[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: start the N ranks ourselves (the same command
    line the driver uses).  On a box with fewer GPUs than ranks the ranks share devices over gloo - a functional dry
    run of the N-rank flow, not a measurement (the JSON line says so)."""
    import socket
    import subprocess

    import torch

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus:
        env.setdefault("BENCH_DIST_BACKEND", "gloo")
        env.setdefault("MI355_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--mode", args.mode]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    return subprocess.call(cmd, env=env)


def run_phase(args, mode: str, rank: int, world: int, port: int, model_dir: str):
    """One measurement: mode "single" (one GPU), "replicas" (every rank its own engine over its own 32 sequences) or
    "tp" (one engine, rank 0, the model sharded over all ranks; the other ranks serve it).  Returns the result
    object on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    from model_configs import QWEN3_0_6B
    from nanovllm import LLM, SamplingParams
    from nanovllm.engine.llm_engine import run_worker

    tp = world if mode == "tp" else 1
    total_new = args.warmup + args.steps + 2
    blocks_needed = BATCH * ((PROMPT_LEN + total_new) // BLOCK + 2) + 64
    kw = dict(tensor_parallel_size=tp, kvcache_block_size=BLOCK, max_num_seqs=BATCH, max_model_len=4096,
              max_num_batched_tokens=16384,
              num_kvcache_blocks=int(os.environ.get("BENCH_NBLK", max(blocks_needed, 4096))), hccl_port=port + 1,
              synthetic_seed=0, sampling_seed=0, warmup=os.environ.get("BENCH_NO_WARMUP") is None,
              quantization=os.environ.get("BENCH_QUANT") or None,  # "fp8": side experiment, never the default line
              enforce_eager=os.environ.get("BENCH_EAGER") is not None)
    if tp > 1 and rank != 0:
        run_worker(model_dir, **kw)
        return None

    llm = LLM(model_dir, **kw)
    random.seed(rank if tp == 1 else 0)  # replicas: every rank its own prompts
    prompts = [[random.randint(0, 10000) for _ in range(PROMPT_LEN)] for _ in range(BATCH)]
    sp = SamplingParams(temperature=1.0, max_tokens=total_new, ignore_eos=True, greedy=True)
    llm.gc.watch()  # every collection of the run with its generation, duration and whether a step was in progress
    seqs = [llm.add_request(p, sp) for p in prompts]
    # prefill (2 steps of 16 x 1024 tokens; the engine queues the second behind the first): the whole phase on the
    # wall clock - scheduling, metadata, launches, both steps on the device - divided by its steps
    prefill_steps = 0
    torch.cuda.synchronize()
    t_p = time.perf_counter()
    while any(s.num_completion_tokens == 0 for s in seqs):
        llm.step()
        prefill_steps += 1
    torch.cuda.synchronize()
    t_p_end = time.perf_counter()
    prefill_phase_ms = (t_p_end - t_p) * 1e3
    prefill_trace = [dict(r) for r in llm.prefill_trace]
    # the phase ends when the LAST prefill step's tokens are on the host (its first-token stamp): since round 5 the engine
    # queues the first decode step behind that step, and the synchronize above waits for it as well - a decode step is
    # not prefill time
    prefill_phase_to_sync_ms = prefill_phase_ms  # (rounds 1-4's definition: reported beside the other)
    stamps = [r["stamp"] for r in prefill_trace if r["stamp"] is not None]
    if len(stamps) == prefill_steps:
        prefill_phase_ms = (max(stamps) - t_p) * 1e3
    ttft = sorted(llm.ttft[s.seq_id] for s in seqs)
    for _ in range(args.warmup):
        llm.step()
    replicas = mode == "replicas"
    if replicas:
        dist.barrier()
    torch.cuda.synchronize()
    ctx0 = len(seqs[0])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, n = llm.step()
        assert n == -BATCH
    torch.cuda.synchronize()
    if replicas:
        dist.barrier()
    elapsed = elapsed_local = time.perf_counter() - t0
    ctx1 = len(seqs[0])
    n_rep = 1
    if replicas:  # the slowest rank's clock; TTFTs of all ranks
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tt = torch.tensor(ttft, dtype=torch.float64, device=dev)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        ttft = sorted(float(x) for x in torch.cat(allt).cpu())
        n_rep = world
    algo = n_rep * sum(step_bytes(BATCH, c) for c in range(ctx0 + 1, ctx1 + 1))
    value = n_rep * BATCH * args.steps / elapsed
    n_dev = world if mode != "single" else 1
    if mode == "single":
        par, agg = "tp1", "one GPU"
    elif replicas:
        par = f"dp{world} (independent replicas of bs {BATCH}; no data-path collective)"
        agg = f"sum over {world} replicas, global batch {BATCH * world}"
    else:
        par, agg = f"tp{world}", f"one model sharded over {world} GPUs (Megatron TP, RCCL / xGMI) at fixed bs {BATCH}"
    # MFMA roofline of the prefill: flops of the prefill steps / their wall time vs the dense bf16 peak of the GPUs
    # that computed them (replicas: rank 0's own steps on its one GPU)
    per_step = BATCH // max(1, prefill_steps)
    pf = prefill_flops(per_step, PROMPT_LEN, QWEN3_0_6B)
    pf_ms = prefill_phase_ms / max(1, prefill_steps)
    result = {
        "metric": "decode tokens/s, Qwen3-0.6B bs=32 seq=1024 (p50 TTFT in ttft_p50_ms)",
        "value": value, "unit": "tokens/s", "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if tp > 1 else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "mode": mode, "aggregate": agg,
        "config": {"workload": "Qwen3-0.6B bf16 paged decode under hipGraph, bs=32, 1024-token prompts, "
                               "block_size=16 (BASELINE.json configs[1])",
                   "batch": BATCH, "global_batch": BATCH * n_rep, "prompt_len": PROMPT_LEN, "ctx_first": ctx0 + 1,
                   "ctx_last": ctx1, "block_size": BLOCK, "parallelism": par, "greedy": True,
                   **({"weights": os.environ["BENCH_QUANT"]} if os.environ.get("BENCH_QUANT") else {})},
        "ttft_p50_ms": statistics.median(ttft) * 1e3, "ttft_max_ms": ttft[-1] * 1e3,
        "prefill_steps": prefill_steps,
        # where the prefill phase's time went, per step, on the phase's own clock (0 = the first step() call): when the
        # host started / finished queueing the step's launches, how long the step ran on the device (HIP events around it
        # on the launch stream), when its first tokens were stamped on the host
        "prefill_steps_ms": [{"tokens": r["tokens"], "seqs": r["seqs"],
                              "launch_start_ms": (r["launch_start"] - t_p) * 1e3, "host_launch_ms": r["host_launch_ms"],
                              "device_ms": r["device_ms"],
                              "stamp_ms": (r["stamp"] - t_p) * 1e3 if r["stamp"] is not None else None,
                              "queued_behind_previous": r["queued_behind_previous"],
                              "graph_replay": bool(r.get("graph"))} for r in prefill_trace],
        "prefill_lookahead_min_tokens": llm.prefill_lookahead_min_tokens,
        # the garbage collector during the run (engine/host_gc.py): nothing may run a full pass inside a step
        "gc": {"control": llm.gc.enabled, "settle_ms": llm.gc.stats["settle_ms"],
               "frozen_objects": llm.gc.stats["frozen_objects"],
               "in_prefill": llm.gc.summary(t_p, t_p_end), "in_timed_region": llm.gc.summary(t0, t0 + elapsed_local),
               "policy": {k: llm.gc.stats[k] for k in ("young", "mid", "full_idle")}},
        "prefill_roofline": {"bound": "mfma", "achieved": pf / (pf_ms * 1e-3) / 1e12, "peak": PREFILL_MFMA_PEAK * tp / 1e12,
                             "unit": "TFLOP/s", "frac": pf / (pf_ms * 1e-3) / (PREFILL_MFMA_PEAK * tp),
                             "flops_per_step": pf, "ms_per_step": pf_ms, "tokens_per_step": per_step * PROMPT_LEN,
                             # the phase on both clocks: to the last first-token stamp (the definition above, rounds 5+)
                             # and to the device synchronisation behind it (rounds 1-4; since round 5 that also waits
                             # for the first decode step queued behind the last prefill step)
                             "phase_ms_to_last_stamp": prefill_phase_ms, "phase_ms_to_sync": prefill_phase_to_sync_ms,
                             "what": "the engine's prefill steps (projections + causal attention + head): wall time of the "
                                     "whole prefill phase incl. host (first step() call -> the last step's tokens on the "
                                     "host) / its steps"},
        "step_roofline": {"bound": "hbm", "achieved": algo / elapsed / 1e9, "peak": HBM_PEAK / 1e9 * n_dev,
                          "unit": "GB/s", "frac": algo / elapsed / (HBM_PEAK * n_dev),
                          "bytes_per_step": algo / args.steps},
    }
    mr = llm.model_runner
    if tp > 1:  # what the ranks really ran on: visible to the driver
        result["tp"] = {"world_seen": dist.get_world_size(), "backend": dist.get_backend(),
                        "xgmi_exchange": mr.xgmi is not None, "xgmi_selftest": getattr(mr, "xgmi_selftest", None),
                        "decode_graphs": sorted(mr.graphs), "lookahead": bool(getattr(llm, "lookahead", False))}
    # device time of the captured decode step alone (graph replays back to back, HIP events):
    # ms_per_step minus this is the host share of a step (scheduler, metadata, sampling, sync)
    bucket = mr._bucket_for(BATCH) if mr.graphs else None
    if bucket is not None and tp == 1:  # with TP the graph holds exchanges: rank 0 must not replay it alone
        g = mr.graphs[bucket]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        result["step_roofline"]["graph_replay_ms"] = e0.elapsed_time(e1) / 20
    if rank == 0:
        # per-GPU kernel rooflines (rank 0's shard of the heads / weights under TP): local launches, no collective
        try:
            result["roofline"] = attention_roofline(llm, seqs)
            result["chain_roofline"] = chain_roofline(llm, BATCH)
            if tp > 1:
                result["roofline"]["per_rank"] = result["chain_roofline"]["per_rank"] = True
        except Exception as e:  # never lose the line over a diagnostic
            result["roofline"] = {"error": repr(e)}
    llm.exit()
    return result if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--mode", choices=("both", "replicas", "tp"), default="both",
                    help="N > 1: independent replicas (weak scaling, the headline value) AND tensor parallelism at the "
                         "fixed batch (reported under `tp_run` of the same line), or only one of the two")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    from model_configs import QWEN3_0_6B

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # launched with another rank count than --gpus says: the ranks that exist are what runs
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)
    port = int(os.environ.get("MASTER_PORT", "29500"))
    shared_gpu = False
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
        shared_gpu = torch.cuda.device_count() < world
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_DIST_BACKEND=gloo: functional run of this N > 1 flow with the ranks sharing one GPU
        # (not a measurement); the driver's runs use RCCL ("nccl"), one GPU per rank
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    tmp = os.environ.get("TMPDIR", "/tmp")
    model_dir = os.path.join(tmp, f"mi355_qwen3_0p6b_{port}")
    if rank == 0:
        os.makedirs(model_dir, exist_ok=True)
        with open(os.path.join(model_dir, "config.json"), "w") as f:
            json.dump(QWEN3_0_6B, f)
    if world > 1:
        dist.barrier()

    if world == 1:
        result = run_phase(args, "single", rank, world, port, model_dir)
    else:
        # Both aggregates in ONE line.  Replicas first: the tensor-parallel engine tears the process group down
        # when it exits.  The TP phase runs under a watchdog: whatever happens to it, the line is printed.
        result = run_phase(args, "replicas", rank, world, port, model_dir) if args.mode in ("both", "replicas") else None
        tp_run = None
        if args.mode in ("both", "tp"):
            import threading

            def give_up():
                if rank == 0 and result is not None:
                    result["tp_run"] = {"error": "tensor-parallel phase did not finish within its time limit"}
                    print(json.dumps(result), flush=True)
                os._exit(0 if result is not None else 3)

            dog = threading.Timer(float(os.environ.get("BENCH_TP_TIMEOUT", "420")), give_up)
            dog.daemon = True
            dog.start()
            try:
                tp_run = run_phase(args, "tp", rank, world, port + 7, model_dir)
            except Exception as e:
                tp_run = {"error": repr(e)} if rank == 0 else None
            dog.cancel()
        if rank == 0:
            if result is None:
                result = tp_run  # --mode tp: the TP line is the line
            elif tp_run is not None:
                result["tp_run"] = {k: tp_run[k] for k in ("value", "ms_per_step", "scaling", "mode", "aggregate",
                                                            "ttft_p50_ms", "prefill_roofline", "step_roofline", "tp",
                                                            "roofline", "chain_roofline", "error") if k in tp_run}
                result["tp_run"].setdefault("n_gpus", world)
        if rank == 0 and result is not None:
            # north_star: ">= 6x aggregate throughput at TP = 8" - both aggregates as multiples of the one-GPU line
            ref, ref_file = one_gpu_reference()
            if ref:
                result["speedup_vs_1gpu"] = result["value"] / ref
                result["speedup_reference"] = f"{ref:.0f} tokens/s on one GPU ({ref_file})"
                if isinstance(result.get("tp_run"), dict) and "value" in result["tp_run"]:
                    result["tp_run"]["speedup_vs_1gpu"] = result["tp_run"]["value"] / ref
        if rank == 0 and result is not None and shared_gpu:
            result["dry_run"] = (f"{world} ranks on {torch.cuda.device_count()} GPU(s) over "
                                 f"{os.environ.get('BENCH_DIST_BACKEND', 'nccl')}: functional run of the multi-rank flow, "
                                 "not a measurement")
    if rank != 0 or result is None:
        return
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
