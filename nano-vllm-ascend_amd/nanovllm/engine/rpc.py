"""Rank-0 -> TP-worker control channel: a shared-memory seqlock carrying int64 arrays.

The reference publishes each step as pickle.dumps([method, *args]) into a 1 MiB
SharedMemory block guarded by multiprocessing Events (model_runner.py:172-187), which
requires the workers to be children of rank 0 and ships every token id every step.
Here the block is addressed by name (derived from the rendezvous port), so it also
works when the ranks are started independently (torchrun), and a step is a flat int64
message:

    [generation, method, is_prefill, n_seqs, payload_len, payload...]

Workers spin on `generation`; no acknowledgement is needed because a published step
cannot finish on rank 0 before every worker has joined its collectives, i.e. has read
it.  Steps without collectives are therefore never published (ModelRunner.call drops
empty `run` calls on every rank), and a reader that sees the generation move while it
copies the payload reads again (seqlock).
"""
from __future__ import annotations

import time
from multiprocessing import shared_memory

import numpy as np

from nanovllm.engine.sequence import Sequence

_METHODS = ("run", "exit")
_HEADER = 5
_CAPACITY = 1 << 21  # int64 words (16 MiB): > max_num_batched_tokens ids + tables


def _name(port: int) -> str:
    return f"mi355_nanovllm_{port}"


class StepChannel:
    def __init__(self, port: int, world_size: int, rank: int):
        import torch.distributed as dist

        self.rank = rank
        nbytes = _CAPACITY * 8
        if rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(name=_name(port), create=True, size=nbytes)
            except FileExistsError:  # stale segment from a crashed run
                old = shared_memory.SharedMemory(name=_name(port))
                old.close()
                old.unlink()
                self.shm = shared_memory.SharedMemory(name=_name(port), create=True, size=nbytes)
            self.buf = np.ndarray((_CAPACITY,), dtype=np.int64, buffer=self.shm.buf)
            self.buf[:_HEADER] = 0
            dist.barrier()
        else:
            dist.barrier()
            self.shm = shared_memory.SharedMemory(name=_name(port))
            self.buf = np.ndarray((_CAPACITY,), dtype=np.int64, buffer=self.shm.buf)
        self.generation = 0

    def send(self, method: str, seqs: list[Sequence] | None = None, is_prefill: bool = False) -> None:
        payload: list[int] = []
        for s in seqs or ():
            payload.extend(s.to_wire(is_prefill))
        n = len(payload)
        assert _HEADER + n <= _CAPACITY, "step message exceeds the control channel"
        b = self.buf
        if n:
            b[_HEADER:_HEADER + n] = payload
        b[1] = _METHODS.index(method)
        b[2] = int(is_prefill)
        b[3] = len(seqs or ())
        b[4] = n
        self.generation += 1
        b[0] = self.generation  # publish last

    def recv(self):
        b, spins = self.buf, 0
        while int(b[0]) == self.generation:
            spins += 1
            if spins > 2000:
                time.sleep(0)  # yield, keep latency in the microsecond range
        while True:
            gen = int(b[0])
            method, is_prefill, n_seqs, n = _METHODS[int(b[1])], bool(b[2]), int(b[3]), int(b[4])
            data = b[_HEADER:_HEADER + n].copy()
            if int(b[0]) == gen:  # nothing was published while the message was copied
                break
        self.generation = gen
        seqs, pos = [], 0
        for _ in range(n_seqs):
            s, pos = Sequence.from_wire(data, pos)
            seqs.append(s)
        return method, seqs, is_prefill

    def close(self) -> None:
        self.buf = None
        self.shm.close()
        if self.rank == 0:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
