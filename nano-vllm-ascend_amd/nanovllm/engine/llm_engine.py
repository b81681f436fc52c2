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
- add_request time), the definition of bench/serving_bench.py:35-48,112-121: a HOST clock reading taken when the step's
tokens are on the host.  Two things keep host hiccups out of it: the garbage collector never runs a full pass inside
step() (engine/host_gc.py), and a queued prefill step whose tokens have already arrived is collected and stamped BEFORE
the next step's launch sequence starts (_step_prefill).  `prefill_trace` keeps, per prefill step, when its launch
sequence started and ended on the host, how long it ran on the device (HIP events) and when its tokens were stamped.
"""
from __future__ import annotations

import atexit
import os
from dataclasses import fields
from random import randint
from time import perf_counter

import torch.multiprocessing as mp

from nanovllm.config import Config
from nanovllm.engine.host_gc import HostGc
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
        # a step in flight this long hides the next one's launch sequence: Config.prefill_lookahead_min_tokens, or (< 0)
        # derived after warm-up from what a prefill step's launch sequence costs this host and what a token costs this
        # device (_derive_prefill_lookahead_min_tokens)
        self.prefill_lookahead_min_tokens = (config.prefill_lookahead_min_tokens
                                             if config.prefill_lookahead_min_tokens >= 0 else 4096)
        self.prefill_lookahead_launches = 0
        self.prefill_trace: list[dict] = []  # one record per prefill step (see _launch_prefill)
        self.gc = HostGc(enabled=config.gc_control and os.environ.get("MI355_GC_CONTROL", "1") != "0")
        self._exited = False
        if kwargs.get("warmup", True):
            self.warmup_model()
        # everything alive now is permanent: one full collection, then frozen (engine/host_gc.py).  The full pass walks
        # every object of the process and leaves the CPU caches cold for the launch path - measured: the first prefill
        # step after it takes the host 1.1 ms longer to launch and, launch-bound in its first layers, the device 0.6 ms
        # longer to run (profiles/r05_ttft_2x2.txt) - so it runs BEFORE the warm-up's last phase, not after it.
        self.gc.settle()
        if kwargs.get("warmup", True):
            self.warmup_full_house()
            if config.prefill_lookahead_min_tokens < 0:
                self._derive_prefill_lookahead_min_tokens()
        self.ttft.clear()
        self.prefill_trace.clear()
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

    def warmup_full_house(self):
        """The shape a full house arrives in (not in the reference, whose steps are never queued behind one another):
        max_num_seqs prompts that fill the token budget TWICE, greedy - two prefill steps of max_num_seqs / 2 sequences,
        the second queued behind the first (_step_prefill), then one decode step of the largest bucket.  Whatever a
        first step of that shape costs once (pinned staging of that size class, the sampler's greedy branch, allocator
        growth, code that has never run) is paid here, before any request's clock runs."""
        cfg = self.config
        per_seq = min(cfg.max_model_len - 2, 2 * cfg.max_num_batched_tokens // max(2, cfg.max_num_seqs))
        blocks = cfg.max_num_seqs * (per_seq // self.block_size + 2)
        if (cfg.max_num_seqs >= 2 and per_seq >= 16 and blocks <= len(self.scheduler.block_manager.free_block_ids)
                and os.environ.get("MI355_WARMUP_FULL_HOUSE", "1") != "0"):  # (the switch exists for A/B runs)
            prompts = [[randint(0, 10000) % self._vocab() for _ in range(per_seq)] for _ in range(cfg.max_num_seqs)]
            k = min(cfg.max_num_seqs, cfg.max_num_batched_tokens // per_seq)
            if self.model_runner.can_launch_prefill and k >= 1:
                # a full house arrives as steps of k sequences of per_seq tokens: captured now, so that both warm-up steps -
                # and every full-house step after them - replay a graph (ModelRunner._prefill_bucket; other large shapes
                # are captured when they come by the second time)
                self.model_runner.ensure_prefill_graph(k * per_seq, k, per_seq)
            self.generate(prompts, SamplingParams(ignore_eos=True, max_tokens=2, greedy=True), use_tqdm=False)
        self.ttft.clear()

    def _derive_prefill_lookahead_min_tokens(self):
        """Queueing step k + 1 behind step k pays when k's device time covers k + 1's launch sequence on the host.  Both
        were just measured by the warm-up's prefill steps (prefill_trace): the host's launch time per step (it does not
        depend on the token count: one launch sequence per layer) and the device's time per token of the longest step."""
        steps = [t for t in self.prefill_trace if t.get("device_ms") and t["tokens"] >= 1024 and not t.get("captured")]
        eager = [t for t in steps if not t.get("graph")]  # (the step queued next may be one no graph exists for)
        if not steps or not eager:
            return
        launch_ms = sorted(t["host_launch_ms"] for t in eager)[len(eager) // 2]
        big = max(steps, key=lambda t: t["tokens"])
        per_token_ms = big["device_ms"] / big["tokens"]
        want = int(1.25 * launch_ms / per_token_ms)
        self.prefill_lookahead_min_tokens = max(512, min(want, self.config.max_num_batched_tokens))

    def _vocab(self) -> int:
        text = getattr(self.config.hf_config, "text_config", self.config.hf_config)
        return text.vocab_size

    def exit(self):
        if self._exited:
            return
        self._exited = True
        self.gc.release()
        try:
            self.model_runner.call("exit")
        except RuntimeError as e:  # a worker that is gone never acknowledges the message: the ranks are ended below
            import warnings

            warnings.warn(f"engine exit: {e}; terminating the worker processes")
            try:
                self.model_runner.exit(abort=True)
            except Exception:  # noqa: BLE001 - teardown goes on
                pass
            for p in self.ps:
                if p.is_alive():
                    p.terminate()
        del self.model_runner
        for p in self.ps:
            p.join(timeout=30)

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
        # request preprocessing (the token array built by Sequence, the chained hashes of the prompt's full blocks) runs
        # BEFORE the clock starts: the reference's serving benchmark stamps a request's submission after add_request has
        # returned (bench/serving_bench.py:100-105), i.e. its TTFT excludes add_request as well
        seq.prompt_hashes(self.block_size)
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
        self.gc.enter_step()  # no automatic collection from here on: HostGc.slack() is where the young ones run
        try:
            out = self._step()
        finally:
            self.gc.leave_step()
        if out[1] and self.is_finished():
            self.gc.idle()  # the last request just left: the one place a full collection may run
        return out

    def _step(self):
        if self._inflight is not None:
            return self._step_lookahead(*self._inflight)
        if self._inflight_prefill is not None:
            queued, self._inflight_prefill = self._inflight_prefill, None
            return self._step_prefill(*queued)
        seqs, is_prefill = self.scheduler.schedule()
        if self.lookahead and not is_prefill and seqs and self.model_runner.can_launch_decode(len(seqs)):
            return self._step_lookahead(self.model_runner.call("launch_decode", seqs), seqs, set())
        if self.lookahead and is_prefill and seqs and self._prefill_launchable(seqs):
            return self._step_prefill(self._launch_prefill(seqs), seqs)
        t0 = perf_counter()
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        if is_prefill:
            now = self._stamp_first_tokens(seqs)
            self._trace({"tokens": sum(len(s) - s.num_prefix_tokens for s in seqs), "seqs": len(seqs),
                         "launch_start": t0, "launch_end": now, "host_launch_ms": (now - t0) * 1e3,
                         "device_ms": None, "stamp": now, "queued_behind_previous": False})
        self.scheduler.postprocess(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids, s.num_prompt_tokens, s.num_cached_tokens)
                   for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def _prefill_launchable(self, seqs) -> bool:
        runner = self.model_runner
        fn = getattr(runner, "prefill_launchable", None)  # (scripted runners of the host tests: the property only)
        return fn(seqs) if fn is not None else bool(runner.can_launch_prefill)

    def _launch_prefill(self, seqs, behind_previous: bool = False):
        """Queue a prefill step; the handle carries the step's trace record."""
        runner = self.model_runner
        counters = lambda: (getattr(runner, "prefill_graph_replays", 0), getattr(runner, "prefill_graph_lazy_captures", 0))  # noqa: E731
        replays, captures = counters()
        t0 = perf_counter()
        handle = runner.call("launch_prefill", seqs)
        t1 = perf_counter()
        rec = {"tokens": sum(len(s) - s.num_prefix_tokens for s in seqs), "seqs": len(seqs), "launch_start": t0,
               "launch_end": t1, "host_launch_ms": (t1 - t0) * 1e3, "device_ms": None, "stamp": None,
               "queued_behind_previous": behind_previous, "graph": counters()[0] > replays,
               "captured": counters()[1] > captures}
        self._trace(rec)
        return (handle, rec)

    def _trace(self, rec: dict) -> None:
        """One record per prefill step, on every path (queued, synchronous, tensor-parallel); a serving engine runs for
        days: keep the tail."""
        if len(self.prefill_trace) >= 4096:
            del self.prefill_trace[:2048]
        self.prefill_trace.append(rec)

    def _stamp_first_tokens(self, seqs) -> float:
        now = perf_counter()
        if len(self.ttft) >= 1 << 16:  # a serving engine runs for days: keep the newer half (dicts keep insertion order)
            for key in list(self.ttft)[: 1 << 15]:
                del self.ttft[key]
        for s in seqs:
            # (a first token is the sequence's only completion token; it may already be counted as pending when the
            # first decode step was queued behind the prefill step, _queue_decode_behind_prefill)
            if s.num_completion_tokens == int(s.token_pending) and s.seq_id not in self.ttft:
                s.first_token_time = now
                self.ttft[s.seq_id] = now - s.arrival_time
        return now

    def _step_prefill(self, launched, seqs):
        """One prefill step whose launch is already queued (`launched` = (runner handle, trace record)): admit and queue
        the NEXT prefill step first when Scheduler.lookahead_prefill allows it, then wait for this step's first tokens
        and postprocess them.  A step whose tokens are already on the host is NOT made to wait for the next one's launch
        sequence (3-4 ms of host time): it is collected and stamped at once, the next step is then scheduled by the
        following step() call."""
        handle, rec = launched
        runner, sched = self.model_runner, self.scheduler
        num_tokens = sum(len(s) for s in seqs)
        decode_plan = None
        if not runner.prefill_done(handle):
            # (the NEXT prefill step is admitted ahead only where any step can be queued: a tensor-parallel engine queues
            # the steps that replay a graph, and what the next admission will be is not known before it is made)
            nxt = (sched.lookahead_prefill(seqs, self.prefill_lookahead_min_tokens) if runner.can_launch_prefill
                   else None)
            if nxt:
                self._inflight_prefill = (self._launch_prefill(nxt, behind_previous=True), nxt)
                self.prefill_lookahead_launches += 1
            elif self.lookahead and self._inflight is None and not sched.waiting:
                decode_plan = self._queue_decode_behind_prefill(seqs)
            self.gc.slack()
        tokens = runner.collect_prefill(handle)
        rec["device_ms"] = runner.prefill_device_ms(handle)
        if decode_plan is not None:  # (no row of this step was aborted: checked when the decode step was queued)
            queued, deferred = decode_plan
            rec["stamp"] = self._stamp_first_tokens(seqs)
            for s in sched.resolve(seqs, tokens, deferred, queued[1]):
                queued[2].add(id(s))  # ended on EOS with its first token: its row of the queued step is dropped
            self._inflight = queued
        else:
            live = [(s, t) for s, t in zip(seqs, tokens) if not s.is_finished]  # (aborted while the step was queued)
            seqs, tokens = [s for s, _ in live], [t for _, t in live]
            rec["stamp"] = self._stamp_first_tokens(seqs)
            sched.postprocess(seqs, tokens)
        outputs = [(s.seq_id, s.completion_token_ids, s.num_prompt_tokens, s.num_cached_tokens)
                   for s in seqs if s.is_finished]
        return outputs, num_tokens

    def _queue_decode_behind_prefill(self, seqs):
        """The LAST prefill step of a burst is running and nothing is waiting: the step after it is a decode step over
        every running sequence, and which one follows from lengths alone (Scheduler.lookahead - the same decisions
        postprocess() + schedule() would take when the first tokens are in).  Queue it now: the rows of this prefill
        step take their input ids on the device from the step's token buffer (the sampler of launch_prefill writes the
        buffer the decode graphs read), the device goes from the prefill step into the first decode step without the
        0.4 ms of host work in between (profiles/r04_final2_prefill_step_edges.txt).
        -> ((runner handle, sequences, dropped rows), sequences with a seal outstanding) or None: decide synchronously."""
        runner, sched = self.model_runner, self.scheduler
        if any(s.is_finished for s in seqs):  # a request of this step was aborted while it was queued
            return None
        plan = sched.lookahead(seqs, runner.max_launch_rows)
        if plan is None:
            return None
        nxt, deferred = plan
        row_of = {id(s): i for i, s in enumerate(seqs)}
        src = [row_of[id(s)] if s.token_pending else -1 for s in nxt]
        self.decode_behind_prefill_launches = getattr(self, "decode_behind_prefill_launches", 0) + 1
        return (runner.call("launch_decode", nxt, src), nxt, set()), deferred

    def _step_lookahead(self, handle, seqs, dropped):
        """One decode step whose launch is already queued (`handle`): decide and queue the NEXT step first
        (Scheduler.lookahead - everything that depends on lengths only; input ids stay on the device), then
        wait for this step's tokens and finish what depends on their values (Scheduler.resolve).  The device
        goes from one step to the next without waiting for the host's scheduling, metadata upload and launch."""
        runner, sched = self.model_runner, self.scheduler
        live = [s for s in seqs if id(s) not in dropped]
        # (a prefill step already queued behind this decode step: the step after it is decided when it is collected)
        plan = sched.lookahead(live, runner.max_launch_rows) if self._inflight_prefill is None else None
        queued = None
        if plan is not None:
            nxt, deferred = plan
            row_of = {id(s): i for i, s in enumerate(seqs)}
            src = [row_of[id(s)] if s.token_pending else -1 for s in nxt]
            queued = (runner.call("launch_decode", nxt, src), nxt, set())
        elif self._inflight_prefill is None and sched.waiting and getattr(runner, "prefill_graph_takes", None) is not None:
            # requests arrived while this decode step was queued: their prefill step - a graph replay, 0.1 ms of host
            # time - goes behind it NOW, not after its tokens have come back (Scheduler.lookahead_prefill_behind_decode)
            arrivals = sched.lookahead_prefill_behind_decode(live, runner.prefill_graph_takes)
            if arrivals:
                self._inflight_prefill = (self._launch_prefill(arrivals, behind_previous=True), arrivals)
                self.prefill_behind_decode_launches = getattr(self, "prefill_behind_decode_launches", 0) + 1
        self.gc.slack()  # the device has the next step: the host's only idle time in a step
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
        # every prompt is checked before any is queued: one over-long prompt must not leave the others behind in the
        # scheduler as orphans of a call that raised
        if any(isinstance(p, str) for p in prompts):
            if self.tokenizer is None:
                raise ValueError("this model directory has no tokenizer: pass token ids")
            prompts = [self.tokenizer.encode(p) if isinstance(p, str) else p for p in prompts]
        for prompt in prompts:
            if len(prompt) > self.scheduler.max_model_len:
                raise ValueError(f"prompt of {len(prompt)} tokens exceeds max_model_len = {self.scheduler.max_model_len} "
                                 "(Config.max_model_len, clamped to the model's max_position_embeddings)")
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
