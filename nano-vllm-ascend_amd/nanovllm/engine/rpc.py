"""Rank-0 -> TP-worker control channel: a shared-memory seqlock carrying int64 arrays.

The reference publishes each step as pickle.dumps([method, *args]) into a 1 MiB
SharedMemory block guarded by multiprocessing Events (model_runner.py:172-187), which
requires the workers to be children of rank 0 and ships every token id every step.
Here the block is addressed by name (derived from the rendezvous port), so it also
works when the ranks are started independently (torchrun), and a step is a flat int64
message:

    [generation, method, is_prefill, n_seqs, payload_len, n_extra, payload..., extra...]

`extra` carries a method's small integer arguments: for "launch_decode" (a decode step queued behind the running
one, engine lookahead) the row of the previous step's token buffer each sequence's input id comes from.

Messages live in a ring of four slots; a worker consumes them in order.  No acknowledgement is needed: a published
step cannot finish on rank 0 before every worker has joined its collectives, i.e. has read it, and rank 0 never
publishes more than two messages past a step whose results it has waited for (the engine's lookahead queues ONE step
behind the running one) - so at most two messages are unread when a third is written, and a slot is re-used only four
messages later.  A reader checks the slot's own generation and fails loudly if it was lapped anyway.  Steps without
collectives are never published (ModelRunner.call drops empty `run` calls on every rank).
"""
from __future__ import annotations

import time
from multiprocessing import shared_memory

import numpy as np

from nanovllm.engine.sequence import Sequence

_METHODS = ("run", "exit", "launch_decode", "abort")
_HEADER = 6
_DEFAULT_CAPACITY = 1 << 18  # int64 words per slot (2 MiB) when the caller does not size the channel
_SLOTS = 4
_BASE = 8            # word 0: newest generation published


def slot_words(max_num_batched_tokens: int, max_num_seqs: int, max_model_len: int, block_size: int) -> int:
    """int64 words of the largest step message: a prefill step carries every scheduled token id once, every step up to
    max_num_seqs records of 11 header words + a block table (Sequence.to_wire) - sized from the configuration
    (four fixed 16 MiB slots were just over Docker's default 64 MB /dev/shm, which RCCL also uses)."""
    table = -(-(max_model_len + 1) // block_size)
    return _HEADER + max_num_batched_tokens + max_num_seqs * (12 + table + 1) + 1024


def _name(port: int) -> str:
    return f"mi355_nanovllm_{port}"


class StepChannel:
    def __init__(self, port: int, world_size: int, rank: int, capacity_words: int = _DEFAULT_CAPACITY):
        import torch.distributed as dist

        self.rank = rank
        _CAPACITY = self.capacity = int(capacity_words)  # every rank derives it from the same configuration
        nbytes = (_BASE + _SLOTS * _CAPACITY) * 8
        if rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(name=_name(port), create=True, size=nbytes)
            except FileExistsError:  # stale segment from a crashed run
                old = shared_memory.SharedMemory(name=_name(port))
                old.close()
                old.unlink()
                self.shm = shared_memory.SharedMemory(name=_name(port), create=True, size=nbytes)
            self.buf = np.ndarray((_BASE + _SLOTS * _CAPACITY,), dtype=np.int64, buffer=self.shm.buf)
            self.buf[:_BASE] = 0
            for k in range(_SLOTS):
                self.buf[_BASE + k * _CAPACITY] = 0
            dist.barrier()
        else:
            dist.barrier()
            self.shm = shared_memory.SharedMemory(name=_name(port))
            self.buf = np.ndarray((_BASE + _SLOTS * _CAPACITY,), dtype=np.int64, buffer=self.shm.buf)
        self.generation = 0

    def send(self, method: str, seqs: list[Sequence] | None = None, is_prefill: bool = False,
             extra: list[int] | None = None) -> None:
        payload: list[int] = []
        for s in seqs or ():
            payload.extend(s.to_wire(is_prefill))
        n, extra = len(payload), list(extra or ())
        _CAPACITY = self.capacity
        assert _HEADER + n + len(extra) <= _CAPACITY, "step message exceeds the control channel"
        gen = self.generation + 1
        b = self.buf[_BASE + (gen % _SLOTS) * _CAPACITY:]
        b[0] = 0  # the slot is being rewritten
        if n:
            b[_HEADER:_HEADER + n] = payload
        if extra:
            b[_HEADER + n:_HEADER + n + len(extra)] = extra
        b[1] = _METHODS.index(method)
        b[2] = int(is_prefill)
        b[3] = len(seqs or ())
        b[4] = n
        b[5] = len(extra)
        b[0] = gen             # the slot is complete ...
        self.generation = gen
        self.buf[0] = gen      # ... and published

    def recv(self):
        spins, want = 0, self.generation + 1
        while int(self.buf[0]) < want:
            spins += 1
            if spins > 2000:
                time.sleep(0)  # yield, keep latency in the microsecond range
        b = self.buf[_BASE + (want % _SLOTS) * self.capacity:]
        method, is_prefill, n_seqs, n, n_extra = _METHODS[int(b[1])], bool(b[2]), int(b[3]), int(b[4]), int(b[5])
        data = b[_HEADER:_HEADER + n].copy()
        extra = [int(v) for v in b[_HEADER + n:_HEADER + n + n_extra]]
        if int(b[0]) != want:  # rank 0 ran more than _SLOTS - 1 messages ahead of this worker: a protocol error
            raise RuntimeError(f"control channel: message {want} was overwritten before rank {self.rank} read it")
        self.generation = want
        seqs, pos = [], 0
        for _ in range(n_seqs):
            s, pos = Sequence.from_wire(data, pos)
            seqs.append(s)
        return method, seqs, is_prefill, extra

    def close(self) -> None:
        self.buf = None
        self.shm.close()
        if self.rank == 0:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
