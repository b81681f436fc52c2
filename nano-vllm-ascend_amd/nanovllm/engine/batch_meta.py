"""Host-side step metadata: Python sequences -> flat integer arrays.

These arrays are what reaches the device (input ids, positions, slot mappings,
context lengths, block tables), so they are part of the bit-exact parity contract
with the reference's ModelRunner.prepare_prefill (model_runner.py:238-290),
prepare_decode (:344-366), prepare_decode_padding (:292-342) and
prepare_block_tables (:231-236).  Pure numpy: testable without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from nanovllm.engine.sequence import Sequence


@dataclass
class PrefillMeta:
    input_ids: np.ndarray      # int64 [T]
    positions: np.ndarray      # int64 [T]
    cu_seqlens_q: np.ndarray   # int32 [n+1]
    cu_seqlens_k: np.ndarray   # int32 [n+1]
    max_seqlen_q: int
    max_seqlen_k: int
    slot_mapping: np.ndarray   # int32 [T]   flat: block_id*block_size + offset
    block_tables: np.ndarray   # int32 [n, W]  -1 padded
    kv_lens: np.ndarray        # int32 [n]


@dataclass
class DecodeMeta:
    input_ids: np.ndarray      # int64 [B]
    positions: np.ndarray      # int64 [B]
    context_lens: np.ndarray   # int32 [B]
    slot_mapping: np.ndarray   # int32 [B, 2]  [block_id, offset]
    block_tables: np.ndarray   # int32 [B, W]
    real_bs: int


def block_table_matrix(seqs: list[Sequence], rows: int | None = None, cols: int | None = None) -> np.ndarray:
    """-1 padded [rows, cols] table; defaults: one row per sequence, widest table."""
    width = max((len(s.block_table) for s in seqs), default=0)
    cols = width if cols is None else cols
    rows = len(seqs) if rows is None else rows
    out = np.full((rows, max(cols, 1) if cols == 0 else cols), -1, dtype=np.int32)
    for i, s in enumerate(seqs):
        n = min(len(s.block_table), cols)
        if n:
            out[i, :n] = s.block_table[:n]
    return out


def prefill_meta(seqs: list[Sequence], block_size: int, skip_cached: bool = False) -> PrefillMeta:
    """skip_cached=False is the reference: every token of every scheduled sequence is (re)computed -
    cached prefix blocks are not skipped (model_runner.py:248-249) - positions restart at 0.

    skip_cached=True feeds only the tokens behind the leading cache hits
    (`num_prefix_tokens`, whole blocks): queries start at that position, their K/V rows go to the
    slots of the non-shared blocks, and attention reaches the shared prefix through the block
    table (kv_lens stays the full length).  A fully cached prompt still computes its last token
    (the logits row), without rewriting that token's shared KV row (slot -1)."""
    full = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    if skip_cached:
        skip = np.fromiter((min(s.num_prefix_tokens, len(s) - 1) for s in seqs), dtype=np.int64, count=len(seqs))
    else:
        skip = np.zeros(len(seqs), dtype=np.int64)
    lens = full - skip
    cu = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    cu_k = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(full, out=cu_k[1:])
    total = int(cu[-1])
    ids = np.empty(total, dtype=np.int64)
    pos = np.empty(total, dtype=np.int64)
    slots = np.empty(total, dtype=np.int32)
    within = np.arange(int(full.max()) if len(seqs) else 0, dtype=np.int64)
    in_block = np.arange(block_size, dtype=np.int32)
    for s, a, n, k in zip(seqs, cu[:-1].tolist(), lens.tolist(), skip.tolist()):
        ids[a:a + n] = np.frombuffer(s.ids_array(), dtype=np.int64)[k:k + n]
        pos[a:a + n] = within[k:k + n]
        if s.block_table:
            # slot of position p = table[p // block_size] * block_size + p % block_size: one broadcast over whole blocks
            table = np.asarray(s.block_table[: s.num_blocks], dtype=np.int32)
            slots[a:a + n] = (table[:, None] * block_size + in_block).reshape(-1)[k:k + n]
            if skip_cached and k < s.num_prefix_tokens:  # recomputed only for its logits
                slots[a:a + s.num_prefix_tokens - k] = -1
        else:
            slots[a:a + n] = -1
    mq = int(lens.max()) if len(seqs) else 0
    mk = int(full.max()) if len(seqs) else 0
    return PrefillMeta(ids, pos, cu, cu_k, mq, mk, slots, block_table_matrix(seqs), full.astype(np.int32))


def decode_meta(seqs: list[Sequence], pad_to: int | None = None, dummy_block: int = 0,
                table_cols: int | None = None) -> DecodeMeta:
    """One new token per sequence.  With `pad_to` (graph mode) rows beyond the real batch
    are token 0 / position 0 / context_len 0 / slot [dummy_block, 0] and the table is a
    static [pad_to, table_cols] matrix of -1 (model_runner.py:303-331)."""
    real = len(seqs)
    rows = real if pad_to is None else pad_to
    ids = np.zeros(rows, dtype=np.int64)
    pos = np.zeros(rows, dtype=np.int64)
    ctx = np.zeros(rows, dtype=np.int32)
    slot = np.empty((rows, 2), dtype=np.int32)
    slot[:, 0], slot[:, 1] = dummy_block, 0
    for i, s in enumerate(seqs):
        n = s.num_tokens
        ids[i] = s.last_token
        pos[i] = n - 1
        ctx[i] = n
        slot[i, 0] = s.block_table[-1]
        slot[i, 1] = s.last_block_num_tokens - 1
    tables = block_table_matrix(seqs, rows=rows, cols=table_cols)
    return DecodeMeta(ids, pos, ctx, slot, tables, real)


class DecodeStager:
    """decode_meta() written in place into preallocated (pinned) arrays, with the block-table rows
    updated incrementally: a row is rewritten only when another sequence - or the same sequence with
    a rebuilt table (preemption + re-prefill bumps `table_gen`) - moves into it; otherwise only the
    newly appended block ids are written.  Produces exactly decode_meta(seqs, pad_to=bucket, ...)."""

    def __init__(self, ids, pos, ctx, slots, tables, temps=None):
        self.ids, self.pos, self.ctx, self.slots, self.tables, self.temps = ids, pos, ctx, slots, tables, temps
        self.tables[:] = -1
        self._owner = [(-1, -1, 0)] * tables.shape[0]  # (seq_id, table_gen, entries written) per row

    def fill(self, seqs: list[Sequence], bucket: int, dummy_block: int) -> None:
        real = len(seqs)
        tables, owner = self.tables, self._owner
        if real:
            # column-wise: one list per field, one numpy assignment each (per-element numpy stores
            # cost ~0.1 us apiece; a decode step runs this for every sequence)
            lens = [s.num_tokens for s in seqs]
            self.ids[:real] = [s.last_token for s in seqs]
            self.ctx[:real] = lens
            self.pos[:real] = lens
            self.pos[:real] -= 1
            bs = seqs[0].block_size
            self.slots[:real, 0] = [s.block_table[-1] for s in seqs]
            self.slots[:real, 1] = [(n - 1) % bs for n in lens]
            if self.temps is not None:
                self.temps[:real] = [0.0 if s.greedy else s.temperature for s in seqs]
        for i, s in enumerate(seqs):
            table = s.block_table
            nb = len(table)
            sid, gen, written = owner[i]
            if sid == s.seq_id and gen == s.table_gen:
                if written == nb:
                    continue  # the usual case: nothing new in this row
                if written < nb:
                    tables[i, written:nb] = table[written:nb]
                    owner[i] = (sid, gen, nb)
                    continue
            row = tables[i]
            row[:nb] = table
            row[nb:] = -1
            owner[i] = (s.seq_id, s.table_gen, nb)
        if bucket > real:
            self.ids[real:bucket] = 0
            self.pos[real:bucket] = 0
            self.ctx[real:bucket] = 0
            self.slots[real:bucket, 0] = dummy_block
            self.slots[real:bucket, 1] = 0
            for i in range(real, bucket):
                if owner[i][0] != -1:
                    tables[i] = -1
                    owner[i] = (-1, -1, 0)
