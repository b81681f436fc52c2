"""Engine loop (reference: nanovllm/engine/llm_engine.py:28-175, text path).

`LLM(model_dir, **Config kwargs)`, `add_request`, `step() -> (finished, ±num_tokens)`,
`is_finished`, `abort_request`, `generate(prompts | token ids, SamplingParams)` keep the
reference's shapes: step() returns +prefill tokens / -decode sequences so callers derive
throughput (llm_engine.py:126) and generate() returns
[{"text","token_ids","prompt_len","cache_tokens"}] ordered by sequence id (:170-173).

Tensor-parallel ranks: either spawned here as child processes (the reference's mode,
:39-45) or already running (`torchrun`: WORLD_SIZE set) — then only rank 0 constructs an
engine and the other ranks call `run_worker(model, **kwargs)`.

TTFT is recorded per request as (end of the prefill step that produced its first token
- add_request time), the definition of bench/serving_bench.py:35-48,112-121.
"""
from __future__ import annotations

import atexit
import os
from dataclasses import fields
from random import randint
from time import perf_counter

import torch.multiprocessing as mp

from nanovllm.config import Config
from nanovllm.engine.model_runner import ModelRunner
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams


def _make_config(model, kwargs) -> Config:
    names = {f.name for f in fields(Config)}
    return Config(model, **{k: v for k, v in kwargs.items() if k in names})


def _externally_launched() -> bool:
    return int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ


def _worker_entry(config: Config, rank: int):
    ModelRunner(config, rank).loop()


def run_worker(model, **kwargs) -> None:
    """Body of ranks > 0 when the ranks were started by torchrun."""
    config = _make_config(model, kwargs)
    ModelRunner(config, int(os.environ["RANK"])).loop()


class LLMEngine:
    def __init__(self, model, **kwargs):
        config = _make_config(model, kwargs)
        self.config = config
        self.block_size = config.kvcache_block_size
        self.ps = []
        if config.tensor_parallel_size > 1 and not _externally_launched():
            ctx = mp.get_context("spawn")
            for rank in range(1, config.tensor_parallel_size):
                p = ctx.Process(target=_worker_entry, args=(config, rank))
                p.start()
                self.ps.append(p)
        self.model_runner = ModelRunner(config, 0)
        self.tokenizer = self._load_tokenizer(config)
        if self.tokenizer is not None and self.tokenizer.eos_token_id is not None:
            config.eos = self.tokenizer.eos_token_id
        self.scheduler = Scheduler(config)
        self.ttft: dict[int, float] = {}
        # lookahead decode (decode graphs that end in the token choice - one GPU, or tensor parallelism with the
        # exchange region up, where every rank picks the same tokens on its own device): the step after the one
        # running is scheduled and queued on the device(s) before the running one's tokens reach the host; the TP
        # workers queue the same step when its message arrives (ModelRunner.loop)
        self.lookahead = config.decode_lookahead and os.environ.get("MI355_LOOKAHEAD", "1") != "0"
        self._inflight = None  # (handle, sequences, rows dropped after launch) of a queued decode step
        # ... and the same between consecutive PREFILL steps (one GPU): while a long prefill step runs, the next one is
        # admitted (Scheduler.lookahead_prefill: only when that is the decision schedule() would take afterwards) and
        # queued behind it, so the device goes from one to the other without the host's 0.6 ms in between
        self._inflight_prefill = None  # (handle, sequences) of a queued prefill step
        self.prefill_lookahead_min_tokens = 4096  # a step in flight this long hides the next one's launch sequence
        self.prefill_lookahead_launches = 0
        self._exited = False
        if kwargs.get("warmup", True):
            self.warmup_model()
        atexit.register(self.exit)

    @staticmethod
    def _load_tokenizer(config: Config):
        files = ("tokenizer.json", "tokenizer_config.json", "vocab.json", "tokenizer.model")
        if not any(os.path.exists(os.path.join(config.model, f)) for f in files):
            return None  # synthetic mode: token-id prompts only
        from transformers import AutoTokenizer

        return AutoTokenizer.from_pretrained(config.model, use_fast=True)

    # ------------------------------------------------------------------ warm-up (llm_engine.py:53-87)
    def warmup_model(self):
        cfg = self.config
        n = max(1, min(cfg.max_num_batched_tokens // cfg.max_model_len, cfg.max_num_seqs))
        prompts = [[randint(0, 10000) % self._vocab() for _ in range(cfg.max_model_len)] for _ in range(n)]
        self.generate(prompts, SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=1), use_tqdm=False)
        prompts = [[randint(0, 10000) % self._vocab() for _ in range(randint(10, 50))]
                   for _ in range(cfg.max_num_seqs)]
        self.generate(prompts, SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=2), use_tqdm=False)
        self.ttft.clear()

    def _vocab(self) -> int:
        text = getattr(self.config.hf_config, "text_config", self.config.hf_config)
        return text.vocab_size

    def exit(self):
        if self._exited:
            return
        self._exited = True
        self.model_runner.call("exit")
        del self.model_runner
        for p in self.ps:
            p.join()

    # ------------------------------------------------------------------ requests
    def add_request(self, prompt: str | list[int], sampling_params: SamplingParams, request_id: str | None = None):
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("this model directory has no tokenizer: pass token ids")
            prompt = self.tokenizer.encode(prompt)
        if len(prompt) > self.scheduler.max_model_len:
            # max_model_len (Config: already clamped to the model's max_position_embeddings) bounds the RoPE table and the
            # static block-table width: a longer prompt would read past both.  The reference does not check either (its
            # rotary cache is then indexed out of range, rotary_embedding.py:42); fail before anything is scheduled.
            raise ValueError(f"prompt of {len(prompt)} tokens exceeds max_model_len = {self.scheduler.max_model_len} "
                             "(Config.max_model_len, clamped to the model's max_position_embeddings)")
        seq = Sequence(prompt, sampling_params, request_id=request_id, block_size=self.block_size)
        seq.prompt_hashes(self.block_size)  # request preprocessing (with the token array built by Sequence): before the clock
        seq.arrival_time = perf_counter()
        self.scheduler.add(seq)
        return seq

    def abort_request(self, request_id: str) -> None:
        if self._inflight is not None:  # rows of a queued step that belong to the request are dropped
            _, seqs, dropped = self._inflight
            for s in seqs:
                if s.request_id == request_id and not s.is_finished:
                    dropped.add(id(s))
        # (a queued PREFILL step's rows of the request are recognised by their finished state when it is collected)
        self.scheduler.abort_seq_group(request_id)

    def is_finished(self) -> bool:
        return self.scheduler.is_finished() and self._inflight is None and self._inflight_prefill is None

    def step(self):
        if self._inflight is not None:
            return self._step_lookahead(*self._inflight)
        if self._inflight_prefill is not None:
            queued, self._inflight_prefill = self._inflight_prefill, None
            return self._step_prefill(*queued)
        seqs, is_prefill = self.scheduler.schedule()
        if self.lookahead and not is_prefill and seqs and self.model_runner.can_launch_decode(len(seqs)):
            return self._step_lookahead(self.model_runner.call("launch_decode", seqs), seqs, set())
        if self.lookahead and is_prefill and seqs and self.model_runner.can_launch_prefill:
            return self._step_prefill(self.model_runner.call("launch_prefill", seqs), seqs)
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        if is_prefill:
            self._stamp_first_tokens(seqs)
        self.scheduler.postprocess(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids, s.num_prompt_tokens, s.num_cached_tokens)
                   for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def _stamp_first_tokens(self, seqs):
        now = perf_counter()
        for s in seqs:
            if s.num_completion_tokens == 0 and s.seq_id not in self.ttft:
                s.first_token_time = now
                self.ttft[s.seq_id] = now - s.arrival_time

    def _step_prefill(self, handle, seqs):
        """One prefill step whose launch is already queued (`handle`): admit and queue the NEXT prefill step first when
        Scheduler.lookahead_prefill allows it, then wait for this step's first tokens and postprocess them."""
        runner, sched = self.model_runner, self.scheduler
        nxt = sched.lookahead_prefill(seqs, self.prefill_lookahead_min_tokens)
        if nxt:
            self._inflight_prefill = (runner.call("launch_prefill", nxt), nxt)
            self.prefill_lookahead_launches += 1
        tokens = runner.collect_prefill(handle)
        num_tokens = sum(len(s) for s in seqs)
        live = [(s, t) for s, t in zip(seqs, tokens) if not s.is_finished]  # (aborted while the step was queued)
        seqs, tokens = [s for s, _ in live], [t for _, t in live]
        self._stamp_first_tokens(seqs)
        sched.postprocess(seqs, tokens)
        outputs = [(s.seq_id, s.completion_token_ids, s.num_prompt_tokens, s.num_cached_tokens)
                   for s in seqs if s.is_finished]
        return outputs, num_tokens

    def _step_lookahead(self, handle, seqs, dropped):
        """One decode step whose launch is already queued (`handle`): decide and queue the NEXT step first
        (Scheduler.lookahead - everything that depends on lengths only; input ids stay on the device), then
        wait for this step's tokens and finish what depends on their values (Scheduler.resolve).  The device
        goes from one step to the next without waiting for the host's scheduling, metadata upload and launch."""
        runner, sched = self.model_runner, self.scheduler
        live = [s for s in seqs if id(s) not in dropped]
        plan = sched.lookahead(live, runner.max_launch_rows)
        queued = None
        if plan is not None:
            nxt, deferred = plan
            row_of = {id(s): i for i, s in enumerate(seqs)}
            src = [row_of[id(s)] if s.token_pending else -1 for s in nxt]
            queued = (runner.call("launch_decode", nxt, src), nxt, set())
        tokens = runner.collect(handle)
        if len(live) != len(seqs):
            tokens = [t for s, t in zip(seqs, tokens) if id(s) not in dropped]
        if plan is not None:
            for s in sched.resolve(live, tokens, deferred, nxt):
                queued[2].add(id(s))
        else:
            sched.postprocess(live, tokens)
        self._inflight = queued
        outputs = [(s.seq_id, s.completion_token_ids, s.num_prompt_tokens, s.num_cached_tokens)
                   for s in live if s.is_finished]
        return outputs, -len(live)

    def generate(self, prompts, sampling_params, use_tqdm: bool = True):
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm

            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        done = {}
        prefill_tps = decode_tps = 0.0
        while not self.is_finished():
            t = perf_counter()
            finished, num_tokens = self.step()
            dt = perf_counter() - t
            if num_tokens > 0:
                prefill_tps = num_tokens / dt
            elif num_tokens < 0:
                decode_tps = -num_tokens / dt
            if pbar is not None:
                pbar.set_postfix({"Prefill": f"{int(prefill_tps)}tok/s", "Decode": f"{int(decode_tps)}tok/s"})
            for seq_id, token_ids, prompt_len, cache_tokens in finished:
                done[seq_id] = (token_ids, prompt_len, cache_tokens)
                if pbar is not None:
                    pbar.update(1)
        if pbar is not None:
            pbar.close()
        out = []
        for seq_id in sorted(done):
            token_ids, prompt_len, cache_tokens = done[seq_id]
            text = self.tokenizer.decode(token_ids) if self.tokenizer is not None else ""
            out.append({"text": text, "token_ids": token_ids, "prompt_len": prompt_len,
                        "cache_tokens": cache_tokens})
        return out
