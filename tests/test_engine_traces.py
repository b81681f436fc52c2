"""Scheduler / BlockManager / Sequence / step-metadata parity: replay the request
streams of tests/golden/engine_traces.json and engine_traces_fuzz.json (recorded from the
reference's own classes by tools/gen_golden.py) through this package's host code and require
every index it produces to be identical.  CPU only."""
import json
import os
from collections import deque
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN
from nanovllm.engine import batch_meta
from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams

with open(os.path.join(GOLDEN, "engine_traces.json")) as f:
    SCENARIOS = json.load(f)
with open(os.path.join(GOLDEN, "engine_traces_fuzz.json")) as f:  # 14 seeded random streams, tight memory
    FUZZ = json.load(f)


def fake_token(seq, step):  # the deterministic stand-in sampler of tools/gen_golden.py
    return (sum(seq.token_ids[-4:]) * 31 + 7 * step + len(seq)) % 1000 + 1


@pytest.mark.parametrize("sc", SCENARIOS + FUZZ, ids=[s["name"] for s in SCENARIOS + FUZZ])
def test_trace_replay(sc):
    c = sc["config"]
    cfg = SimpleNamespace(max_num_seqs=c["max_num_seqs"], max_num_batched_tokens=c["max_num_batched_tokens"],
                          eos=c["eos"], num_kvcache_blocks=c["num_kvcache_blocks"],
                          kvcache_block_size=c["block_size"], max_model_len=c["max_model_len"])
    sched = Scheduler(cfg)
    bs = c["block_size"]
    pending = deque(sorted(sc["arrivals"], key=lambda a: a[0]))
    order, index, step = [], {}, 0
    golden = {r["step"]: r for r in sc["steps"]}
    while pending or not sched.is_finished():
        while pending and pending[0][0] <= step:
            _, toks, max_tokens, ignore_eos = pending.popleft()
            s = Sequence(toks, SamplingParams(temperature=1.0, max_tokens=max_tokens, ignore_eos=ignore_eos),
                         block_size=bs)
            index[s.seq_id] = len(order)
            order.append(s)
            sched.add(s)
        if sched.is_finished():
            step += 1
            continue
        seqs, is_prefill = sched.schedule()
        g = golden[step]
        assert is_prefill == g["is_prefill"], step
        assert [index[s.seq_id] for s in seqs] == g["seqs"], step
        assert [list(s.block_table) for s in seqs] == g["block_tables"], step
        assert [s.num_cached_tokens for s in seqs] == g["num_cached_tokens"], step
        assert [len(s) for s in seqs] == g["lens"], step
        assert list(sched.block_manager.free_block_ids) == g["free_block_ids"], step
        assert [index[s.seq_id] for s in sched.waiting] == g["waiting"], step
        assert [index[s.seq_id] for s in sched.running] == g["running"], step
        if seqs:
            ctx = g["context"]
            if is_prefill:
                m = batch_meta.prefill_meta(seqs, bs)
                assert m.input_ids.tolist() == g["input_ids"] and m.positions.tolist() == g["positions"]
                assert m.cu_seqlens_q.tolist() == ctx["cu_seqlens_q"]
                assert m.cu_seqlens_k.tolist() == ctx["cu_seqlens_k"]
                assert (m.max_seqlen_q, m.max_seqlen_k) == (ctx["max_seqlen_q"], ctx["max_seqlen_k"])
                assert m.slot_mapping.tolist() == ctx["slot_mapping"]
                assert m.block_tables.tolist() == ctx["block_tables"]
                assert m.slot_mapping.dtype == np.int32 and m.block_tables.dtype == np.int32
            else:
                if c["padded"]:
                    m = batch_meta.decode_meta(seqs, pad_to=c["max_num_seqs"],
                                               dummy_block=c["num_kvcache_blocks"] - 1,
                                               table_cols=c["max_model_len"] // bs)
                else:
                    m = batch_meta.decode_meta(seqs)
                assert m.input_ids.tolist() == g["input_ids"] and m.positions.tolist() == g["positions"]
                assert m.context_lens.tolist() == ctx["context_lens"]
                assert m.slot_mapping.tolist() == ctx["slot_mapping"]  # 2-D [block, offset]
                assert m.block_tables.tolist() == ctx["block_tables"]
                assert m.real_bs == ctx["real_bs"]
            toks = [fake_token(s, step) for s in seqs]
            assert toks == g["sampled"]
            sched.postprocess(seqs, toks)
            assert [index[s.seq_id] for s in seqs if s.is_finished] == g["finished"], step
        step += 1
        assert step < 2000
    assert [list(s.token_ids) for s in order] == sc["final_tokens"]
    assert [s.num_cached_tokens for s in order] == sc["final_cached"]
    assert len(sched.block_manager.free_block_ids) == c["num_kvcache_blocks"] - 1  # everything returned


def test_scenarios_cover_the_interesting_paths():
    by = {s["name"]: s for s in SCENARIOS}
    assert any(sum(r["num_cached_tokens"]) > 0 for r in by["prefix_share_b4"]["steps"])  # prefix hits
    # preemption happened: a sequence was prefilled twice
    pre = by["preempt_b4"]["steps"]
    prefilled = [i for r in pre if r["is_prefill"] for i in r["seqs"]]
    assert len(prefilled) > len(set(prefilled))
    assert any(r["context"]["slot_mapping"][-1] == [63, 0] for r in by["boundary_b16_padded"]["steps"]
               if not r["is_prefill"])  # dummy slot of padded rows


def test_hash_kats():
    with open(os.path.join(GOLDEN, "hash_kats.json")) as f:
        kats = json.load(f)
    for k in kats:
        assert BlockManager.compute_hash(k["tokens"]) == k["hash"]
        nxt = k.get("chained_tokens", k["tokens"])
        assert BlockManager.compute_hash(nxt, k["chained_with_prefix"]) == k["chained"]
    assert BlockManager.compute_hash([1, 2, 3, 4]) == 0x73F859A04F669E6D  # SURVEY.md §8c probe value


def test_sequence_wire_roundtrip():
    s = Sequence(list(range(100, 140)), SamplingParams(temperature=0.7, max_tokens=5), block_size=16)
    s.block_table = [4, 9, 2]
    s.num_cached_tokens = 16
    s.num_prefix_tokens, s.table_gen = 16, 3  # prefix-aware prefill / staging-row invalidation travel too
    for is_prefill in (True, False):
        buf = np.array(s.to_wire(is_prefill) + [77], dtype=np.int64)
        r, pos = Sequence.from_wire(buf, 0)
        assert pos == len(buf) - 1
        assert (r.seq_id, len(r), r.num_prompt_tokens, r.num_cached_tokens, r.block_size) == (
            s.seq_id, 40, 40, 16, 16)
        assert r.block_table == [4, 9, 2] and r.last_token == 139 and r.temperature == 0.7
        assert (r.num_prefix_tokens, r.table_gen, r.greedy) == (16, 3, False)
        assert r.num_blocks == 3 and r.last_block_num_tokens == 8  # works on the receiver
        if is_prefill:
            assert r.token_ids == s.token_ids


_PREFILL_AHEAD: dict = {}  # scenario -> prefill steps the engine queued behind a running prefill step


class _TraceRunner:
    """ModelRunner stand-in that checks every launched step against the REFERENCE's recorded trace and samples
    with the recording's stand-in sampler.  Under lookahead a step is launched before the previous step's tokens
    are on the host: like the device, it takes them from the previous launch's token buffer (src rows)."""

    def __init__(self, sc, index):
        self.sc, self.c, self.index = sc, sc["config"], index
        self.golden = {r["step"]: r for r in sc["steps"]}
        self.step = 0
        self.device_tokens: list[int] = []
        self.max_launch_rows = self.c["max_num_seqs"]
        self.lookahead_launches = 0

    def can_launch_decode(self, n):
        return 0 < n <= self.max_launch_rows

    def _check_and_sample(self, seqs, is_prefill, src):
        c, g, bs = self.c, self.golden[self.step], self.c["block_size"]
        idx = self.index
        assert is_prefill == g["is_prefill"], self.step
        assert [idx[s.seq_id] for s in seqs] == g["seqs"], self.step
        assert [list(s.block_table) for s in seqs] == g["block_tables"], self.step
        assert [len(s) for s in seqs] == g["lens"], self.step
        views = []  # each sequence's tokens as the DEVICE knows them at this launch
        for s, r in zip(seqs, src):
            toks = list(s.token_ids)
            if r >= 0:
                assert s.token_pending
                toks[-1] = self.device_tokens[r]
            views.append(toks)
        ctx = g["context"]
        if is_prefill:
            m = batch_meta.prefill_meta(seqs, bs)
            assert m.input_ids.tolist() == g["input_ids"] and m.positions.tolist() == g["positions"]
            assert m.slot_mapping.tolist() == ctx["slot_mapping"] and m.block_tables.tolist() == ctx["block_tables"]
        else:
            if c["padded"]:
                m = batch_meta.decode_meta(seqs, pad_to=c["max_num_seqs"], dummy_block=c["num_kvcache_blocks"] - 1,
                                           table_cols=c["max_model_len"] // bs)
            else:
                m = batch_meta.decode_meta(seqs)
            ids = m.input_ids.tolist()
            ids[:len(seqs)] = [v[-1] for v in views]  # what the embedding kernel reads (mi_embedding_from_prev)
            assert ids == g["input_ids"] and m.positions.tolist() == g["positions"], self.step
            assert m.context_lens.tolist() == ctx["context_lens"] and m.slot_mapping.tolist() == ctx["slot_mapping"]
            assert m.block_tables.tolist() == ctx["block_tables"]
        toks = [(sum(v[-4:]) * 31 + 7 * self.step + len(v)) % 1000 + 1 for v in views]
        assert toks == g["sampled"], self.step
        self.step += 1
        self.device_tokens = toks
        return list(toks)

    def launch_decode(self, seqs, src_rows=None):
        self.lookahead_launches += src_rows is not None
        return self._check_and_sample(seqs, False, src_rows if src_rows is not None else [-1] * len(seqs))

    def collect(self, handle):
        return handle

    can_launch_prefill = True

    def launch_prefill(self, seqs):
        return self._check_and_sample(seqs, True, [-1] * len(seqs))

    def collect_prefill(self, handle):
        return handle

    def prefill_done(self, handle):
        return False  # (the device is never ahead of the host here: every lookahead opportunity is taken)

    def prefill_device_ms(self, handle):
        return 0.0

    def call(self, name, seqs, *args):
        if name == "launch_decode":  # the engine's RPC entry: a step queued behind the running one
            return self.launch_decode(seqs, *args)
        if name == "launch_prefill":
            return self.launch_prefill(seqs)
        assert name == "run"
        return self._check_and_sample(seqs, args[0], [-1] * len(seqs)) if seqs else []


@pytest.mark.parametrize("sc", [s for s in SCENARIOS + FUZZ if {a[0] for a in s["arrivals"]} == {0}
                                and all(a[3] for a in s["arrivals"])], ids=lambda s: s["name"])
def test_lookahead_engine_reproduces_the_reference_trace(sc):
    """The reference's recorded index streams (every request present from step 0, no EOS endings: preemption
    under tight memory, block-boundary crossings eager and graph-padded) replayed through LLMEngine with
    decode_lookahead: every step it launches - scheduled one step ahead, from lengths alone - carries exactly the
    sequences, block tables, positions, slots and (device-side) input ids of the reference's step."""
    from nanovllm.engine.llm_engine import LLMEngine

    c = sc["config"]
    cfg = SimpleNamespace(max_num_seqs=c["max_num_seqs"], max_num_batched_tokens=c["max_num_batched_tokens"],
                          eos=c["eos"], num_kvcache_blocks=c["num_kvcache_blocks"],
                          kvcache_block_size=c["block_size"], max_model_len=c["max_model_len"])
    eng = object.__new__(LLMEngine)
    eng.scheduler = Scheduler(cfg)
    eng.block_size, eng.tokenizer, eng.ttft, eng.lookahead, eng._inflight = c["block_size"], None, {}, True, None
    # prefill steps are queued behind one another too, whenever Scheduler.lookahead_prefill can prove the admission
    # (no minimum length of the step in flight here: every opportunity is taken)
    eng._inflight_prefill, eng.prefill_lookahead_min_tokens, eng.prefill_lookahead_launches = None, 0, 0
    from nanovllm.engine.host_gc import HostGc

    eng.gc, eng.prefill_trace = HostGc(enabled=False), []
    order, index = [], {}
    eng.model_runner = _TraceRunner(sc, index)
    for _, toks, max_tokens, ignore_eos in sorted(sc["arrivals"], key=lambda a: a[0]):
        s = eng.add_request(toks, SamplingParams(temperature=1.0, max_tokens=max_tokens, ignore_eos=ignore_eos))
        index[s.seq_id] = len(order)
        order.append(s)
    guard = 0
    while not eng.is_finished():
        eng.step()
        guard += 1
        assert guard < 2000
    assert eng.model_runner.step == len(sc["steps"])  # every recorded step was launched, none extra
    assert eng.model_runner.lookahead_launches > 0    # and some of them one step ahead
    _PREFILL_AHEAD[sc["name"]] = eng.prefill_lookahead_launches
    assert [list(s.token_ids) for s in order] == sc["final_tokens"]
    assert [s.num_cached_tokens for s in order] == sc["final_cached"]
    assert len(eng.scheduler.block_manager.free_block_ids) == c["num_kvcache_blocks"] - 1


def test_some_recorded_prefill_steps_were_queued_ahead():
    """(runs after the parametrised replay above) the replay exercised the prefill lookahead, not only its refusals."""
    assert _PREFILL_AHEAD and sum(_PREFILL_AHEAD.values()) > 0, _PREFILL_AHEAD
