"""Paged-KV block allocator with chained-hash prefix caching
(reference: nanovllm/engine/block_manager.py:26-118; semantics in SURVEY.md §9).

Block tables are part of the parity contract: for the same request stream this
allocator must hand out the same block ids, in the same order, as the reference's.
The rules it reproduces:
  * free list is FIFO (initially 0..N-1); a miss takes its head; freed ids go to the tail,
    and a sequence's blocks are released last-block-first;
  * only full blocks are hashed: xxh64(prev_hash as 8 LE bytes, then the tokens as int64 LE);
    a hit needs the hash AND equal token ids; after the first miss every later block misses;
  * a freed block keeps its hash/tokens until it is handed out again, so it can be revived;
  * decode: a new block is taken when len % block_size == 1, and a block is sealed (hashed,
    registered) when len % block_size == 0.

The hash itself is computed by the C ABI's mi_xxh64_chain (the reference calls the
`xxhash` pip package; tests pin both against the reference's known answers).
"""
from __future__ import annotations

from array import array
from collections import deque

from nanovllm._C import xxh64_chain, xxh64_chain_blocks
from nanovllm.engine.sequence import Sequence

_NO_HASH = -1


class _Tokens:
    """The token ids of a sealed block as a VIEW of its sequence's token list (full blocks are never modified once
    sealed): allocate() seals 64 blocks per 1024-token prompt, and slicing touches every one of the 16 int objects of a
    block (reference counts: memory writes to cold objects) for a list that is read only if the block's hash is ever
    hit.  Compares, iterates and indexes like the list it stands for."""
    __slots__ = ("src", "lo", "n")

    def __init__(self, src: list[int], lo: int, n: int):
        self.src, self.lo, self.n = src, lo, n

    def list(self) -> list[int]:
        return self.src[self.lo:self.lo + self.n]

    def __eq__(self, other):
        return self.list() == (other.list() if isinstance(other, _Tokens) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter(self.list())

    def __getitem__(self, i):
        return self.list()[i]

    def __repr__(self):
        return repr(self.list())

    __hash__ = None


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = _NO_HASH
        self.token_ids: list[int] = []

    def update(self, hash: int, token_ids: list[int]) -> None:
        self.hash, self.token_ids = hash, token_ids

    def reset(self) -> None:
        self.ref_count, self.hash, self.token_ids = 1, _NO_HASH, []


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int, non_cache_token_ids: list[int] | None = None):
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self.used_block_ids: set[int] = set()
        self.non_cache_token_ids = set(non_cache_token_ids or ())

    # -- hashing ---------------------------------------------------------------------------------
    @classmethod
    def compute_hash(cls, token_ids: list[int], prefix: int = _NO_HASH) -> int:
        return xxh64_chain(array("q", token_ids).tobytes(), prefix)

    # -- block bookkeeping -----------------------------------------------------------------------
    def _take(self, block_id: int) -> Block:
        blk = self.blocks[block_id]
        assert blk.ref_count == 0
        blk.reset()
        self.free_block_ids.remove(block_id)
        self.used_block_ids.add(block_id)
        return blk

    def _release(self, block_id: int) -> None:
        assert self.blocks[block_id].ref_count == 0
        self.used_block_ids.discard(block_id)
        self.free_block_ids.append(block_id)

    # reference names kept as aliases (ut/ tests of the reference poke at them)
    _allocate_block = _take
    _deallocate_block = _release

    # -- prefill -----------------------------------------------------------------------------------
    def can_allocate(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table
        seq.table_gen = getattr(seq, "table_gen", 0) + 1  # a fresh table: cached device rows of it are stale
        bs, lookup = self.block_size, self.hash_to_block_id
        n_blocks, n_full = seq.num_blocks, len(seq) // bs
        if n_full * bs <= seq.num_prompt_tokens:  # the usual case: computed when the request was created
            hashes = seq.prompt_hashes(bs)
        else:  # a preempted sequence comes back with its completion tokens: the whole chain in one C call
            hashes = xxh64_chain_blocks(seq.ids_array(), n_full, bs)
        blocks, free, used, table = self.blocks, self.free_block_ids, self.used_block_ids, seq.block_table
        tokens, guard = seq.token_ids, self.non_cache_token_ids
        # leading run of cache hits (block_manager.py:65-88: after the first miss every later block misses)
        hits = 0
        while hits < n_full:
            chain = hashes[hits]
            hit_id = lookup.get(chain, -1)
            if hit_id == -1:
                break
            toks = tokens[hits * bs:(hits + 1) * bs]
            if blocks[hit_id].token_ids != toks or (guard and not guard.isdisjoint(toks)):
                break
            seq.num_cached_tokens += bs  # reporting counter: only ever grows (block_manager.py:79)
            if hit_id in used:
                blk = blocks[hit_id]
                blk.ref_count += 1
            else:  # freed but not yet recycled: revive it
                blk = self._take(hit_id)
            blk.hash, blk.token_ids = chain, toks
            lookup[chain] = hit_id
            table.append(hit_id)
            hits += 1
        # everything behind it takes the head of the free list, in block order (same as _take(free[0]) per block,
        # without the search); full blocks are registered under their chained hash
        ids = [free.popleft() for _ in range(n_blocks - hits)]
        used.update(ids)
        table.extend(ids)
        n_sealed = n_full - hits  # the first n_sealed of them are full blocks: sealed now
        chains = hashes[hits:n_full]
        for bid, chain, lo in zip(ids, chains, range(hits * bs, n_full * bs, bs)):
            blk = blocks[bid]
            assert blk.ref_count == 0
            blk.ref_count, blk.hash, blk.token_ids = 1, chain, _Tokens(tokens, lo, bs)
        lookup.update(zip(chains, ids))  # in block order: a later duplicate hash replaces an earlier one, as assigning one by one
        for bid in ids[n_sealed:]:  # the open tail block, if any
            blk = blocks[bid]
            assert blk.ref_count == 0
            blk.ref_count, blk.hash, blk.token_ids = 1, _NO_HASH, []
        # hits are always a leading run, and a hit block holds valid KV rows by the time this prefill's attention
        # reads it: it was written by an earlier step, or is written earlier in the same forward pass by the
        # sequence that owns it
        seq.num_prefix_tokens = hits * bs

    def deallocate(self, seq: Sequence) -> None:
        for block_id in reversed(seq.block_table):
            blk = self.blocks[block_id]
            blk.ref_count -= 1
            if blk.ref_count == 0:
                self._release(block_id)
        seq.block_table.clear()
        seq.num_prefix_tokens = 0

    # -- decode ------------------------------------------------------------------------------------
    def can_append(self, seq: Sequence) -> bool:
        needs_block = len(seq) % self.block_size == 1
        return len(self.free_block_ids) >= int(needs_block)

    def may_append(self, seq: Sequence, defer_seal: bool = False) -> bool:
        """Called after the previous step's append_token: len(seq) already counts the token
        whose KV row this step writes.  defer_seal: the last token of the block being filled is not on the
        host yet (engine lookahead) - leave the block open and return True; the caller seals it with
        seal_tail() once the token has arrived."""
        table = seq.block_table
        tail = self.blocks[table[-1]]
        rem = len(seq) % self.block_size
        if rem == 1:  # the new token opens a block
            assert tail.hash != _NO_HASH
            new_id = self.free_block_ids[0]
            self._take(new_id)
            table.append(new_id)
        elif rem == 0:  # the new token fills the tail block: seal it
            if defer_seal:
                return True
            self.seal_tail(seq)
        else:
            assert tail.hash == _NO_HASH
        return False

    def seal_tail(self, seq: Sequence) -> None:
        """Hash and register the (just filled) last block of the sequence."""
        table = seq.block_table
        tail = self.blocks[table[-1]]
        assert len(seq) % self.block_size == 0 and tail.hash == _NO_HASH
        toks = seq.block(seq.num_blocks - 1)
        prev = self.blocks[table[-2]].hash if len(table) > 1 else _NO_HASH
        h = self.compute_hash(toks, prev)
        tail.update(h, toks)
        self.hash_to_block_id[h] = tail.block_id
