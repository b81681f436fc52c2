"""Rank-0 -> TP-worker control channel: a shared-memory seqlock carrying int64 arrays.

The reference publishes each step as pickle.dumps([method, *args]) into a 1 MiB
SharedMemory block guarded by multiprocessing Events (model_runner.py:172-187), which
requires the workers to be children of rank 0 and ships every token id every step.
Here the block is addressed by name (derived from the rendezvous port), so it also
works when the ranks are started independently (torchrun), and a step is a flat int64
message:

    [generation, method, is_prefill, n_seqs, payload_len, n_extra, part_len, more, body part ...]

`extra` carries a method's small integer arguments: for "launch_decode" (a decode step queued behind the running
one, engine lookahead) the row of the previous step's token buffer each sequence's input id comes from.  The body
(payload then extra) of a message that does not fit one slot continues in the following slots (`more` = 1 on every
part but the last): a prefill step ships the token ids behind each sequence's cached prefix, and with prefix-cache
hits the scheduler's token budget does not bound that sum by one slot (57 sequences sharing a prefix made a
117 819-word step against a 25 094-word slot) - a slot is the unit of transfer, not a limit on a step.

Messages live in a ring of four slots; a worker consumes them in order and acknowledges each part it has copied out
(word 1 + rank of the segment).  Rank 0 re-uses a slot only when every worker has acknowledged the part that lived there
four parts earlier - with one-part messages that never waits (a published step cannot finish on rank 0 before every
worker has joined its collectives, and the engine's lookahead queues ONE step behind the running one), with a
many-part message it is the flow control.  A worker that does not acknowledge within `ACK_TIMEOUT_S` is an error on
rank 0 (never an overwrite, also under `python -O`); a reader checks the slot's own generation and fails loudly if it
was lapped anyway.  Steps without collectives are never published (ModelRunner.call drops empty `run` calls on every
rank).
"""
from __future__ import annotations

import time
from multiprocessing import shared_memory

import numpy as np

from nanovllm.engine.sequence import Sequence

_METHODS = ("run", "exit", "launch_decode", "abort", "launch_prefill")
_HEADER = 8
_DEFAULT_CAPACITY = 1 << 18  # int64 words per slot (2 MiB) when the caller does not size the channel
_SLOTS = 4
_BASE = 8            # word 0: newest generation published; word r (1 <= r < world <= 8): the newest part rank r has consumed
ACK_TIMEOUT_S = 120.0
TEARDOWN_ACK_TIMEOUT_S = 2.0  # "exit" / "abort": a worker that died must not hold rank 0's teardown for two minutes


class ChannelError(RuntimeError):
    """The control channel's framing was violated (interleaved parts, a length that contradicts its header, a slot
    overwritten before it was read).  Raised - never asserted: `python -O` strips asserts, and a worker that went on with
    a torn message would replay a step the other ranks are not running."""


def slot_words(max_num_batched_tokens: int, max_num_seqs: int, max_model_len: int, block_size: int) -> int:
    """int64 words of a slot: what a step message normally needs in ONE part - a prefill step carries the scheduled
    token ids once, every step up to max_num_seqs records of 11 header words + a block table (Sequence.to_wire) - sized
    from the configuration (four fixed 16 MiB slots were just over Docker's default 64 MB /dev/shm, which
    RCCL also uses).  Longer messages (prefix-cache hits, see the module text) travel in several parts."""
    table = -(-(max_model_len + 1) // block_size)
    return _HEADER + max_num_batched_tokens + max_num_seqs * (12 + table + 1) + 1024


def _name(port: int) -> str:
    return f"mi355_nanovllm_{port}"


class StepChannel:
    def __init__(self, port: int, world_size: int, rank: int, capacity_words: int = _DEFAULT_CAPACITY):
        import torch.distributed as dist

        self.rank, self.world_size = rank, world_size
        self.skip_cached_prefix = False  # prefill steps ship only the tokens behind the cached prefix (set by ModelRunner)
        self.parts_sent = 0              # (statistics: parts published, messages that needed more than one)
        self.multipart_messages = 0
        if not 1 <= world_size <= _BASE:
            raise ValueError(f"control channel: world size {world_size} (1 .. {_BASE})")
        _CAPACITY = self.capacity = max(int(capacity_words), _HEADER + 64)  # every rank derives it from the same configuration
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

    def _wait_for_slot(self, gen: int, timeout_s: float = ACK_TIMEOUT_S) -> None:
        """Part `gen` goes where part gen - _SLOTS lived: every worker must have copied that one out."""
        need = gen - _SLOTS
        if need <= 0 or self.world_size == 1:
            return
        acks = self.buf[1:self.world_size]
        spins, deadline = 0, None
        while int(acks.min()) < need:
            spins += 1
            if spins > 2000:
                time.sleep(0)
                if deadline is None:
                    deadline = time.monotonic() + timeout_s
                elif time.monotonic() > deadline:
                    raise RuntimeError(f"control channel: a worker has not read part {need} after {timeout_s:.0f} s "
                                       f"(acknowledged: {[int(a) for a in acks]})")

    def send(self, method: str, seqs: list[Sequence] | None = None, is_prefill: bool = False,
             extra: list[int] | None = None) -> None:
        payload: list[int] = []
        skip = is_prefill and self.skip_cached_prefix
        for s in seqs or ():
            payload.extend(s.to_wire(is_prefill, skip))
        n, extra = len(payload), list(extra or ())
        body = np.asarray(payload + extra, dtype=np.int64) if (n or extra) else np.empty(0, dtype=np.int64)
        room = self.capacity - _HEADER
        parts = max(1, -(-len(body) // room))
        self.multipart_messages += parts > 1
        for k in range(parts):
            piece = body[k * room:(k + 1) * room]
            gen = self.generation + 1
            # (teardown messages: a worker that is gone never acknowledges - give up on it quickly, the caller goes on
            # to terminate the ranks)
            self._wait_for_slot(gen, TEARDOWN_ACK_TIMEOUT_S if method in ("exit", "abort") else ACK_TIMEOUT_S)
            b = self.buf[_BASE + (gen % _SLOTS) * self.capacity:]
            b[0] = 0  # the slot is being rewritten
            if len(piece):
                b[_HEADER:_HEADER + len(piece)] = piece
            b[1] = _METHODS.index(method)
            b[2] = int(is_prefill)
            b[3] = len(seqs or ())
            b[4] = n
            b[5] = len(extra)
            b[6] = len(piece)
            b[7] = int(k + 1 < parts)
            b[0] = gen             # the slot is complete ...
            self.generation = gen
            self.buf[0] = gen      # ... and published
            self.parts_sent += 1

    def _recv_part(self):
        spins, want = 0, self.generation + 1
        while int(self.buf[0]) < want:
            spins += 1
            if spins > 2000:
                time.sleep(0)  # yield, keep latency in the microsecond range
        b = self.buf[_BASE + (want % _SLOTS) * self.capacity:]
        head = [int(v) for v in b[1:_HEADER]]
        data = b[_HEADER:_HEADER + head[5]].copy()
        if int(b[0]) != want:  # rank 0 ran more than _SLOTS - 1 parts ahead of this worker: a protocol error
            raise ChannelError(f"control channel: message {want} was overwritten before rank {self.rank} read it")
        self.generation = want
        self.buf[self.rank] = want  # acknowledged: the slot may be re-used
        return head, data

    def recv(self):
        head, data = self._recv_part()
        chunks = [data]
        while head[6]:
            more, data = self._recv_part()
            if more[:5] != head[:5]:
                raise ChannelError(f"control channel: parts of two messages interleaved on rank {self.rank} "
                                   f"({head[:5]} then {more[:5]})")
            head = more
            chunks.append(data)
        method, is_prefill, n_seqs, n, n_extra = _METHODS[head[0]], bool(head[1]), head[2], head[3], head[4]
        body = chunks[0] if len(chunks) == 1 else np.concatenate(chunks)
        if len(body) != n + n_extra:
            raise ChannelError(f"control channel: {len(body)} body words on rank {self.rank}, the header announces "
                               f"{n} + {n_extra}")
        extra = [int(v) for v in body[n:n + n_extra]]
        seqs, pos = [], 0
        for _ in range(n_seqs):
            s, pos = Sequence.from_wire(body, pos)
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
