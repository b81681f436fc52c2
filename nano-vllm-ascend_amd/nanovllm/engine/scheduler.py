"""Prefill-first continuous batching with preemption
(reference: nanovllm/engine/scheduler.py:41-110; semantics in SURVEY.md §9).

The schedule determines which block tables / slot mappings reach the device, so it
is reproduced decision for decision: prefill has absolute priority and admits from
the head of `waiting` until the first request that does not fit (no skipping);
decode walks `running` from the left and preempts from the right; a preempted
sequence is fully deallocated and re-queued at the FRONT of `waiting`.
"""
from __future__ import annotations

from collections import deque

from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.sequence import FinishReason, Sequence, SequenceStatus


class Scheduler:
    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.max_model_len = config.max_model_len
        self.eos = config.eos
        # the last KV block is reserved as the dummy slot of graph-padded rows
        # (scheduler.py:26-30, model_runner.py:309)
        self.block_manager = BlockManager(config.num_kvcache_blocks - 1, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    # -- one scheduling decision -----------------------------------------------------------------
    def schedule(self) -> tuple[list[Sequence], bool]:
        picked = self._admit_prefill()
        if picked:
            return picked, True
        return self._pick_decode(), False

    def _admit_prefill(self) -> list[Sequence]:
        bm, picked, budget_used = self.block_manager, [], 0
        while self.waiting and len(picked) < self.max_num_seqs:
            seq = self.waiting[0]
            if budget_used + len(seq) > self.max_num_batched_tokens or not bm.can_allocate(seq):
                break
            self.waiting.popleft()
            bm.allocate(seq)
            seq.status = SequenceStatus.RUNNING
            self.running.append(seq)
            picked.append(seq)
            budget_used += len(seq) - seq.num_cached_tokens
        return picked

    def _pick_decode(self, deferred: list | None = None) -> list[Sequence]:
        """deferred (lookahead): sequences whose filled block could not be sealed because its last token is
        still on the device are appended to it instead."""
        bm, picked = self.block_manager, []
        bs = bm.block_size
        while self.running and len(picked) < self.max_num_seqs:
            seq = self.running.popleft()
            if seq.num_tokens % bs > 1:
                # the new token lands inside an open block: no block to take (can_append holds whatever
                # the free list says, block_manager.py:99-100) and nothing to seal (may_append's last
                # branch only asserts) - 14 of 16 steps at block size 16
                picked.append(seq)
                continue
            evicted_self = False
            while not bm.can_append(seq):
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    evicted_self = True
                    break
            if not evicted_self:
                if bm.may_append(seq, defer_seal=deferred is not None and seq.token_pending):
                    deferred.append(seq)
                picked.append(seq)
        if picked:
            self.running.extendleft(reversed(picked))
        return picked

    def preempt(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        seq.finish_reason = FinishReason.PREEMPTED
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    # -- request lifecycle -------------------------------------------------------------------------
    def abort_seq_group(self, request_id: str) -> None:
        for queue in (self.waiting, self.running):
            for seq in [s for s in queue if s.request_id == request_id]:
                queue.remove(seq)
                self.free_seq(seq, FinishReason.ABORTED)

    def free_seq(self, seq: Sequence, reason: FinishReason) -> None:
        seq.status = SequenceStatus.FINISHED
        seq.finish_reason = reason
        self.block_manager.deallocate(seq)

    # -- lookahead: schedule step k+1 while step k's tokens are still on the device -------------------
    def _ends_by_length(self, seq: Sequence) -> bool:
        """Will the token now being sampled for `seq` be its last one whatever its value?"""
        return (seq.num_completion_tokens + 1 == seq.max_tokens) or (seq.num_tokens + 1 >= self.max_model_len)

    def lookahead(self, inflight: list[Sequence], row_limit: int) -> tuple[list[Sequence], list[Sequence]] | None:
        """The decode step that follows the one in flight over `inflight` (its live sequences), decided
        before that step's tokens are known: every decision of postprocess() + schedule() that depends only
        on LENGTHS is taken now (sequences ending by length leave and free their blocks, blocks are opened),
        what depends on token VALUES is left for resolve() (EOS, sealing a filled block).  Returns
        (sequences of the next step, sequences with a seal outstanding), or None when the next step must be
        decided synchronously: a prompt is waiting (prefill has priority), a preemption would be needed, or
        the step would have more than row_limit rows (what the runner can queue)."""
        if self.waiting or not inflight:
            return None
        bs, flying = self.block_manager.block_size, {id(s) for s in inflight}
        staying, need = 0, 0
        for seq in self.running:
            if staying == self.max_num_seqs:
                break
            n = seq.num_tokens
            if id(seq) in flying:
                if self._ends_by_length(seq):
                    continue
                n += 1
            staying += 1
            need += n % bs == 1
        if staying == 0 or staying > row_limit or need > len(self.block_manager.free_block_ids):
            return None
        for seq in inflight:  # the length half of postprocess(), in its order
            ends = self._ends_by_length(seq)
            seq.append_pending()
            if ends:
                self.free_seq(seq, FinishReason.LENGTH)
                self.running.remove(seq)
        deferred: list[Sequence] = []
        picked = self._pick_decode(deferred)
        return picked, deferred

    def lookahead_prefill(self, inflight: list[Sequence], min_inflight_tokens: int) -> list[Sequence] | None:
        """The prefill step that follows the prefill step in flight over `inflight`, admitted BEFORE that step's first
        tokens are known - only when the admission is provably the one schedule() would make after postprocess():

        * no can_allocate() of this admission can fail (the free list covers every candidate of the window), so the
          blocks the step in flight may free - appended behind the ones taken here - change neither who is admitted
          nor any block id;
        * the step would be closed by the token budget or the sequence count, not by the end of the queue: a request
          arriving while the step in flight runs could not have joined it either (admission never skips).  (Judged on
          the prompts' full lengths; the real admission charges only the tokens behind cache hits and may then reach
          the end of the queue after all - such a step is the same for the requests in it, but a request arriving
          meanwhile joins the NEXT step instead of this one: batching under arrivals, not block ids or tokens, may
          differ from the synchronous engine there - ADVICE r04);
        * the step in flight is long enough (min_inflight_tokens) for the next one's launch sequence to hide behind it -
          queueing a step behind a short one would only delay the short one's tokens.

        Returns the admitted sequences, or None: decide synchronously."""
        if not self.waiting or sum(len(s) - s.num_cached_tokens for s in inflight) < min_inflight_tokens:
            return None
        blocks = tokens = count = 0
        closed = False
        for seq in self.waiting:
            if count == self.max_num_seqs:
                closed = True
                break
            blocks += seq.num_blocks
            if not closed and tokens + len(seq) > self.max_num_batched_tokens:
                closed = True  # (keep summing the blocks: cache hits may let the real admission go further)
            tokens += len(seq)
            count += 1
        if blocks > len(self.block_manager.free_block_ids):
            return None
        if not (closed or count == self.max_num_seqs or tokens == self.max_num_batched_tokens):
            return None
        return self._admit_prefill() or None

    def lookahead_prefill_behind_decode(self, inflight: list[Sequence], takes) -> list[Sequence] | None:
        """Requests are waiting while a DECODE step is in flight: admit them now and let their prefill step be queued
        behind that step, instead of after its tokens have come back - an open-loop arrival then waits for the running
        step on the device only, not for the host's round trip around it.  `takes(n_seqs, n_tokens)`: the runner can
        launch such a step cheaply (a captured graph: an eager launch sequence would hold back the running step's tokens).
        Taken only when the admission is the one schedule() would make after this step's postprocess():

        * everything waiting fits ONE step that `takes` (count, token budget) - nothing is left for a second decision;
        * the free list covers every candidate, so blocks freed by this step - appended behind the ones taken now - change
          neither who is admitted nor any block id;
        * no sequence of the step in flight ends by length or completes a block with this step's token (a block sealed or
          freed by it could be hit / revived by the new prompts' prefix lookup: allocate() must see the same table).
        (An EOS ending is not knowable here: as for decode lookahead, token streams are unaffected, the order of the free
        list behind such an ending may differ from the synchronous engine's.)  Returns the admitted sequences or None."""
        if not self.waiting or not inflight:
            return None
        bs = self.block_manager.block_size
        for s in inflight:
            if s.is_finished:
                continue  # (left by length when this step was planned: its blocks are already free)
            n = s.num_tokens if s.token_pending else s.num_tokens + 1  # its length once this step's token is in
            if n % bs == 0 or (not s.token_pending and self._ends_by_length(s)):
                return None
        blocks = tokens = 0
        for count, seq in enumerate(self.waiting, 1):
            tokens += len(seq)
            blocks += seq.num_blocks
            if count > self.max_num_seqs or tokens > self.max_num_batched_tokens or not takes(count, tokens):
                return None
        if blocks > len(self.block_manager.free_block_ids):
            return None
        picked = self._admit_prefill()
        assert picked and not self.waiting
        return picked

    def resolve(self, seqs: list[Sequence], token_ids: list[int], deferred: list[Sequence],
                queued: list[Sequence] | None) -> list[Sequence]:
        """The value half of postprocess() for a step whose successor `queued` is already in flight: fill in
        the tokens, end sequences on EOS, seal the blocks lookahead() left open.  Returns the sequences that
        ended on EOS although the queued step still computes a row for them (the caller drops that row)."""
        eos, dropped = self.eos, []
        in_queue = {id(s) for s in queued} if queued else ()
        for seq, tok in zip(seqs, token_ids):
            seq.resolve_pending(tok)
            if tok == eos and not seq.ignore_eos:
                if seq.is_finished:  # had already left by length: only the reason changes
                    seq.finish_reason = FinishReason.EOS
                    continue
                self.free_seq(seq, FinishReason.EOS)
                self.running.remove(seq)
                if id(seq) in in_queue:
                    dropped.append(seq)
        for seq in deferred:
            if not seq.is_finished:
                self.block_manager.seal_tail(seq)
        return dropped

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]) -> None:
        eos, max_model_len = self.eos, self.max_model_len
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            n_total = seq.num_tokens
            if tok == eos and not seq.ignore_eos:
                reason = FinishReason.EOS
            # the reference tests `== max_model_len` (scheduler.py:103), which a prompt of exactly
            # max_model_len tokens steps over; `>=` is identical everywhere else and keeps every
            # sequence inside the static block-table width
            elif n_total - seq.num_prompt_tokens == seq.max_tokens or n_total >= max_model_len:
                reason = FinishReason.LENGTH
            else:
                continue
            self.free_seq(seq, reason)
            self.running.remove(seq)
