"""Per-request state and block arithmetic (reference: nanovllm/engine/sequence.py:23-149).

Text-only (the multimodal fields of the reference are outside the decode path).
The rank-RPC wire format is a flat int64 array (``to_wire`` / ``from_wire``) rather
than the reference's pickled dict (sequence.py:116-146), which ships every
token id each step and drops ``block_size``.
"""
from __future__ import annotations

from enum import Enum, auto
from itertools import count

from nanovllm.sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class FinishReason(Enum):
    EOS = auto()
    LENGTH = auto()
    ABORTED = auto()
    PREEMPTED = auto()


class Sequence:
    counter = count()  # class-global ids, as the reference (warm-up consumes some)

    __slots__ = ("block_size", "seq_id", "request_id", "status", "token_ids", "last_token", "num_tokens",
                 "num_prompt_tokens", "num_cached_tokens", "block_table", "temperature", "max_tokens",
                 "ignore_eos", "greedy", "finish_reason", "arrival_time", "first_token_time", "table_gen", "num_prefix_tokens",
                 "token_pending", "_prompt_q", "_prompt_hashes")

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams | None = None,
                 request_id: str | None = None, block_size: int = 256, **_ignored_multimodal):
        sp = sampling_params if sampling_params is not None else SamplingParams()
        self.block_size = block_size
        self.seq_id = next(Sequence.counter)
        self.request_id = request_id
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1]
        self.num_tokens = self.num_prompt_tokens = len(self.token_ids)
        self.num_cached_tokens = 0
        self.block_table: list[int] = []
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos
        self.greedy = getattr(sp, "greedy", False)
        self.finish_reason = None
        self.arrival_time = 0.0
        self.first_token_time = 0.0
        self.num_prefix_tokens = 0  # leading tokens whose KV rows the current block table already holds (cache hits)
        self.token_pending = False  # the last entry of token_ids stands for a token still on the device (lookahead)
        self.table_gen = 0  # bumped every time the block table is rebuilt (allocate after a preemption)
        # Request preprocessing, done when the request is CREATED (like tokenisation; the reference's serving bench starts
        # a request's clock after add_request has returned, bench/serving_bench.py:100-105): the prompt as array('q')
        # (block hashing and the prefill staging read it: ~20 us per 1024 tokens) and, on first allocation, the chained
        # hashes of its full blocks (prompt_hashes).
        from array import array
        self._prompt_q = array("q", self.token_ids)
        self._prompt_hashes = None  # (block_size, hashes of the prompt's full blocks)

    # -- container protocol ------------------------------------------------------------------
    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    def __repr__(self) -> str:
        why = self.finish_reason.name if self.finish_reason else "None"
        return f"Seq(id={self.seq_id}, status={self.status.name}, reason={why})"

    # -- derived quantities ------------------------------------------------------------------
    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[: self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def ids_array(self):
        """All token ids as a contiguous int64 array (array('q')).  The prompt part is converted once per sequence
        (a 1024-token list costs ~20 us per conversion, and a prefill step needs it twice: block hashes and the
        staged input ids); the completion part - non-empty only when a preempted sequence is prefilled again - is
        converted on every call."""
        from array import array

        pq = getattr(self, "_prompt_q", None)
        if pq is None:
            pq = self._prompt_q = array("q", self.token_ids[: self.num_prompt_tokens])
        if self.num_tokens == self.num_prompt_tokens:
            return pq
        return pq + array("q", self.token_ids[self.num_prompt_tokens:])

    def prompt_hashes(self, block_size: int):
        """Chained xxh64 of the PROMPT's full blocks (a function of the prompt alone: computed once, at creation by
        LLMEngine.add_request or on first use)."""
        ph = self._prompt_hashes
        if ph is None or ph[0] != block_size:
            from nanovllm._C import xxh64_chain_blocks

            ph = self._prompt_hashes = (block_size, xxh64_chain_blocks(self._prompt_q, self.num_prompt_tokens // block_size, block_size))
        return ph[1]

    def block(self, i: int) -> list[int]:
        assert 0 <= i < self.num_blocks
        lo = i * self.block_size
        return self.token_ids[lo: lo + self.block_size]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    def append_pending(self) -> None:
        """Count a token that has been sampled on the device but has not reached the host (engine lookahead)."""
        assert not self.token_pending
        self.append_token(0)
        self.token_pending = True

    def resolve_pending(self, token_id: int) -> None:
        assert self.token_pending
        self.token_ids[-1] = token_id
        self.last_token = token_id
        self.token_pending = False

    # -- rank-RPC wire format -------------------------------------------------------------------
    # header: [seq_id, num_tokens, num_prompt_tokens, num_cached_tokens, block_size, n_blocks,
    #          n_tokens_sent, temperature_bits, greedy, table_gen, num_prefix_tokens]; then block ids; then the token ids the
    # receiver needs (for a prefill step all of them - or, skip_cached_prefix, the ones behind the cached prefix that
    # batch_meta.prefill_meta(skip_cached=True) reads; only the last one for a decode step).
    def to_wire(self, is_prefill: bool, skip_cached_prefix: bool = False) -> list[int]:
        import struct

        if not is_prefill:
            toks = self.token_ids[-1:]
        elif skip_cached_prefix and self.num_prefix_tokens:
            toks = self.token_ids[min(self.num_prefix_tokens, self.num_tokens - 1):]
        else:
            toks = self.token_ids
        tbits = struct.unpack("<q", struct.pack("<d", float(self.temperature)))[0]
        return [self.seq_id, self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.block_size,
                len(self.block_table), len(toks), tbits, int(self.greedy), self.table_gen,
                self.num_prefix_tokens, *self.block_table, *toks]

    @classmethod
    def from_wire(cls, buf, pos: int = 0) -> tuple["Sequence", int]:
        import struct

        (seq_id, num_tokens, num_prompt, num_cached, block_size, n_blocks, n_toks, tbits, greedy, table_gen,
         num_prefix) = (int(v) for v in buf[pos: pos + 11])
        pos += 11
        s = object.__new__(cls)
        s.block_size, s.seq_id, s.request_id = block_size, seq_id, None
        s.status = SequenceStatus.RUNNING
        s.block_table = [int(v) for v in buf[pos: pos + n_blocks]]
        pos += n_blocks
        toks = [int(v) for v in buf[pos: pos + n_toks]]
        pos += n_toks
        # a decode step only carries the last token, a prefix-aware prefill step the tokens behind the cached prefix:
        # pad the front so that indices/len stay right
        s.token_ids = toks if n_toks == num_tokens else [0] * (num_tokens - n_toks) + toks
        s.last_token = toks[-1]
        s.num_tokens, s.num_prompt_tokens, s.num_cached_tokens = num_tokens, num_prompt, num_cached
        s.temperature = struct.unpack("<d", struct.pack("<q", tbits))[0]
        s.greedy = bool(greedy)
        s.table_gen, s.num_prefix_tokens = table_gen, num_prefix
        s.max_tokens, s.ignore_eos, s.finish_reason = 0, True, None
        s.arrival_time = s.first_token_time = 0.0
        s.token_pending = False
        return s, pos
