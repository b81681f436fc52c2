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

    def _pick_decode(self) -> list[Sequence]:
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
                bm.may_append(seq)
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
