"""Tensor-parallel host logic on CPU: world_size 2 / 4 / 8 `gloo` process groups.
Covers the N > 1 path without GPUs: sharded weight loaders, vocab-parallel masks, the
row-parallel all-reduce identity (against the CPU oracle), and the rank-RPC channel."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, fn_name: str, out_q):
    for p in (REPO, os.path.join(REPO, "nano-vllm-ascend_amd"), os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    try:
        globals()[fn_name](rank, world, port)
        out_q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(fn_name: str, world: int = 2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(q.get(timeout=240))
            if results[-1][1] != "ok":
                break  # the other ranks may wait for this one forever: report now
    except Exception:  # queue.Empty
        results.append((-1, "a rank did not report within 240 s"))
    ok = len(results) == len(procs) and all(msg == "ok" for _, msg in results)
    for p in procs:
        p.join(timeout=60 if ok else 2)
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"
    assert ok


# ----------------------------------------------------------------------------- bodies (run in the ranks)
def _body_sharded_layers(rank, world, port):
    import oracle
    from nanovllm.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
    from nanovllm.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear

    g = torch.Generator().manual_seed(0)  # same full tensors on every rank
    # two q heads and ONE kv head per rank: at world 8 these are Qwen3-0.6B's 16 / 8 heads
    H, hq, hkv, d, inter, vocab = 256, 2 * world, world, 128, 512, 1024
    torch.set_default_dtype(torch.bfloat16)
    qkv = QKVParallelLinear(H, d, hq, hkv)
    o = RowParallelLinear(hq * d, H)
    gu = MergedColumnParallelLinear(H, [inter, inter])
    down = RowParallelLinear(inter, H)
    emb = VocabParallelEmbedding(vocab, H)
    head = ParallelLMHead(vocab, H)
    torch.set_default_dtype(torch.float32)

    def full(*shape):
        return (torch.randn(*shape, generator=g) * 0.05).bfloat16()

    wq, wk, wv = full(hq * d, H), full(hkv * d, H), full(hkv * d, H)
    wo, wg, wu, wd, we = full(H, hq * d), full(inter, H), full(inter, H), full(H, inter), full(vocab, H)
    for shard, w in (("q", wq), ("k", wk), ("v", wv)):
        qkv.weight.weight_loader(qkv.weight, w, shard)
    o.weight.weight_loader(o.weight, wo)
    gu.weight.weight_loader(gu.weight, wg, 0)
    gu.weight.weight_loader(gu.weight, wu, 1)
    down.weight.weight_loader(down.weight, wd)
    emb.weight.weight_loader(emb.weight, we)
    head.weight.weight_loader(head.weight, we)

    # shards are the Megatron slices of linear.py:54-153 / embed_head.py:27-32
    hq_l, hkv_l = hq // world, hkv // world
    assert torch.equal(qkv.weight.data[: hq_l * d], wq[rank * hq_l * d:(rank + 1) * hq_l * d])
    assert torch.equal(qkv.weight.data[hq_l * d: (hq_l + hkv_l) * d], wk[rank * hkv_l * d:(rank + 1) * hkv_l * d])
    assert torch.equal(qkv.weight.data[(hq_l + hkv_l) * d:], wv[rank * hkv_l * d:(rank + 1) * hkv_l * d])
    assert torch.equal(o.weight.data, wo[:, rank * hq_l * d:(rank + 1) * hq_l * d])
    il = inter // world
    assert torch.equal(gu.weight.data[:il], wg[rank * il:(rank + 1) * il])
    assert torch.equal(gu.weight.data[il:], wu[rank * il:(rank + 1) * il])
    assert torch.equal(down.weight.data, wd[:, rank * il:(rank + 1) * il])
    vl = vocab // world
    assert torch.equal(emb.weight.data, we[rank * vl:(rank + 1) * vl])
    assert (emb.vocab_start_idx, emb.vocab_end_idx) == (rank * vl, (rank + 1) * vl)

    # column -> row parallel MLP: per-rank partial products summed over ranks == unsharded oracle
    x = full(5, H)
    part = oracle.linear(oracle.silu_and_mul(oracle.linear(x, gu.weight.data)), down.weight.data, keep_fp32=True)
    dist.all_reduce(part)
    want = oracle.linear(oracle.silu_and_mul(oracle.linear(x, torch.cat([wg, wu]))), wd, keep_fp32=True)
    assert (part - want).abs().max().item() < 2e-2  # bf16 intermediates, different split of the K sum

    # vocab-parallel embedding: masked partial rows sum to the full embedding (embed_head.py:34-42)
    ids = torch.tensor([0, 1, vl - 1, vl, vocab - 1, 17, vl + 5])
    y = oracle.embedding(ids, emb.weight.data, emb.vocab_start_idx).float()
    dist.all_reduce(y)
    assert torch.equal(y, we[ids].float())

    # vocab-parallel head: gather of shards to rank 0 == full logits (embed_head.py:56-66)
    h = full(3, H)
    local = oracle.linear(h, head.weight.data)
    parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, parts, 0)
    if rank == 0:
        assert torch.equal(torch.cat(parts, -1), oracle.linear(h, we))


def _body_rpc_channel(rank, world, port):
    from nanovllm.engine.rpc import StepChannel
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams

    ch = StepChannel(port + 1, world, rank)
    if rank == 0:
        seqs = []
        for i, n in enumerate((40, 7)):
            s = Sequence(list(range(100 * i, 100 * i + n)), SamplingParams(temperature=0.5 + i, max_tokens=9),
                         block_size=16)
            s.block_table = list(range(3 * i, 3 * i + s.num_blocks))
            seqs.append(s)
        ch.send("run", seqs, True)
        dist.barrier()
        for s in seqs:
            s.append_token(5)
        ch.send("run", seqs, False)
        dist.barrier()
        # two messages published back to back before anybody reads (the engine's lookahead queues one step behind
        # the running one): the ring keeps both, in order
        ch.send("launch_decode", seqs, False, extra=[1, -1])
        for s in seqs:
            s.append_token(6)
        ch.send("launch_decode", seqs, False, extra=[0, 1])
        dist.barrier()
        ch.send("exit")
        dist.barrier()
    else:
        method, seqs, is_prefill, _ = ch.recv()
        assert method == "run" and is_prefill and [len(s) for s in seqs] == [40, 7]
        assert seqs[0].token_ids == list(range(40)) and seqs[1].block_table == [3]
        assert seqs[1].temperature == 1.5 and seqs[0].num_blocks == 3 and seqs[0].last_block_num_tokens == 8
        dist.barrier()
        method, seqs, is_prefill, extra = ch.recv()
        assert method == "run" and not is_prefill and [len(s) for s in seqs] == [41, 8] and extra == []
        assert [s.last_token for s in seqs] == [5, 5]  # decode steps ship only the last token
        dist.barrier()
        dist.barrier()  # both lookahead messages are out before the first is read
        method, seqs, _, extra = ch.recv()
        assert method == "launch_decode" and extra == [1, -1] and [len(s) for s in seqs] == [41, 8]
        method, seqs, _, extra = ch.recv()
        assert method == "launch_decode" and extra == [0, 1] and [s.last_token for s in seqs] == [6, 6]
        method, seqs, _, _ = ch.recv()
        assert method == "exit" and seqs == []
        dist.barrier()
    ch.close()


def _body_rpc_channel_multipart(rank, world, port):
    """a prefill step with prefix-cache hits can be far larger than a slot (the scheduler's budget
    counts only the tokens behind the cached ones).  The slot is now the unit of transfer: the message travels in parts,
    rank 0 waits for the workers' acknowledgements before it re-uses a slot, and a prefix-aware engine ships only the
    tokens behind each sequence's cached prefix."""
    from nanovllm.engine.rpc import StepChannel, slot_words
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams

    def shared_prefix_step(n, length, cached, block):
        seqs = []
        for i in range(n):
            s = Sequence([7] * cached + [1000 * i + j for j in range(length - cached)], SamplingParams(max_tokens=4),
                         block_size=block)
            s.block_table = list(range(s.num_blocks))
            s.num_prefix_tokens = s.num_cached_tokens = cached
            seqs.append(s)
        return seqs

    # (a) the ADVICE's case at the engine's own slot size: 57 sequences of 2048 tokens sharing 1792
    cap = slot_words(16384, 100, 4096, 16)
    ch = StepChannel(port + 1, world, rank, cap)
    big = shared_prefix_step(57, 2048, 1792, 16)
    if rank == 0:
        assert sum(len(s.to_wire(True)) for s in big) > cap  # did not fit (and used to trip an assert) ...
        ch.send("run", big, True)                            # ... several parts, all tokens
        ch.skip_cached_prefix = True
        ch.send("run", big, True)                            # the prefix-aware form: one part
        assert ch.multipart_messages == 1 and ch.parts_sent == -(-sum(len(s.to_wire(True)) for s in big) // (cap - 8)) + 1
        ch.send("run", big[:2], False)
        dist.barrier()
    else:
        for full in (True, False):
            method, seqs, is_prefill, _ = ch.recv()
            assert method == "run" and is_prefill and len(seqs) == 57
            for i, s in enumerate(seqs):
                assert len(s) == 2048 and s.num_prefix_tokens == 1792 and len(s.block_table) == 128
                assert s.token_ids[1792:] == [1000 * i + j for j in range(256)]
                assert s.token_ids[:1792] == ([7] * 1792 if full else [0] * 1792)  # (the cached part is never read)
        method, seqs, is_prefill, _ = ch.recv()
        assert not is_prefill and [s.seq_id for s in seqs] == [big[0].seq_id, big[1].seq_id]
        dist.barrier()
    ch.close()
    dist.barrier()
    # (b) many parts through a tiny slot, the writer far ahead of the reader: flow control, not overwriting
    ch = StepChannel(port + 2, world, rank, 96)
    seqs = shared_prefix_step(6, 100, 64, 16)
    if rank == 0:
        ch.send("run", seqs, True)
        ch.send("launch_decode", seqs[:3], False, extra=[2, 1, 0])
        ch.send("exit")
        assert ch.multipart_messages == 1 and ch.parts_sent > 2 * 4
        dist.barrier()
    else:
        import time

        time.sleep(0.3)  # rank 0 fills the ring and has to wait for this reader
        method, got, is_prefill, _ = ch.recv()
        assert is_prefill and [s.token_ids for s in got] == [s.token_ids for s in seqs]
        assert [s.block_table for s in got] == [s.block_table for s in seqs]
        method, got, _, extra = ch.recv()
        assert method == "launch_decode" and extra == [2, 1, 0] and len(got) == 3
        assert ch.recv()[0] == "exit"
        dist.barrier()
    ch.close()


def _body_rpc_channel_reader_gone(rank, world, port):
    """A worker that stops reading: rank 0 must fail loudly when the ring is full, never overwrite an unread part."""
    from nanovllm.engine import rpc
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams

    rpc.ACK_TIMEOUT_S = 1.0
    ch = rpc.StepChannel(port + 1, world, rank, 96)
    if rank == 0:
        s = Sequence(list(range(900)), SamplingParams(max_tokens=4), block_size=16)
        s.block_table = list(range(s.num_blocks))
        try:
            ch.send("run", [s], True)  # eleven parts for a ring of four slots; nobody reads
            raise AssertionError("send() returned although no part was acknowledged")
        except RuntimeError as e:
            assert "has not read part" in str(e)
        dist.barrier()
    else:
        dist.barrier()  # (never calls recv)
    ch.close()


def _body_replicated_scheduling(rank, world, port):
    """Every rank can rebuild identical step metadata from the wire format (the reference's
    multi-rank story: deterministic replicated bookkeeping, ut/test_multi_rank_block_manager.py)."""
    import numpy as np

    from nanovllm.engine import batch_meta
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams

    s = Sequence(list(range(50)), SamplingParams(max_tokens=4), block_size=16)
    s.block_table = [9, 4, 7, 2]
    buf = np.array(s.to_wire(False), dtype=np.int64)
    r, _ = Sequence.from_wire(buf, 0)
    m = batch_meta.decode_meta([r])
    mine = torch.tensor([int(m.input_ids[0]), int(m.positions[0]), int(m.context_lens[0]), *m.slot_mapping[0].tolist(),
                         *m.block_tables[0].tolist()])
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    assert mine.tolist() == [49, 49, 50, 2, 1, 9, 4, 7, 2]


def _body_moe_expert_shards(rank, world, port):
    """Qwen3MoeSparseMoeBlock under TP (qwen3_moe.py:104-115): every rank holds all experts, sharded along the
    intermediate dimension; the router is replicated; per-rank partial expert outputs sum to the full expert."""
    from types import SimpleNamespace

    import oracle
    from nanovllm.models.qwen3_moe import Qwen3MoeSparseMoeBlock

    g = torch.Generator().manual_seed(1)
    H, E, K, I = 128, 8, 2, 64 * world
    cfg = SimpleNamespace(hidden_size=H, num_experts=E, num_experts_per_tok=K, moe_intermediate_size=I)
    torch.set_default_dtype(torch.bfloat16)
    blk = Qwen3MoeSparseMoeBlock(cfg)
    torch.set_default_dtype(torch.float32)

    def full(*shape):
        return (torch.randn(*shape, generator=g) * 0.05).bfloat16()

    wg, wu, wd, wr = full(E, I, H), full(E, I, H), full(E, H, I), full(E, H)
    blk.gate.weight.data.copy_(wr)
    for e in range(E):
        ex = blk.experts[e]
        ex.gate_up_proj.weight.weight_loader(ex.gate_up_proj.weight, wg[e], 0)
        ex.gate_up_proj.weight.weight_loader(ex.gate_up_proj.weight, wu[e], 1)
        ex.down_proj.weight.weight_loader(ex.down_proj.weight, wd[e])
    il = I // world
    assert blk.gate_up_stacked.shape == (E, 2 * il, H) and blk.down_stacked.shape == (E, H, il)
    for e in (0, E - 1):
        assert torch.equal(blk.gate_up_stacked[e, :il], wg[e, rank * il:(rank + 1) * il])
        assert torch.equal(blk.gate_up_stacked[e, il:], wu[e, rank * il:(rank + 1) * il])
        assert torch.equal(blk.down_stacked[e], wd[e][:, rank * il:(rank + 1) * il])
    x = full(6, H) * 20
    e = 3
    part = oracle.linear(oracle.silu_and_mul(oracle.linear(x, blk.gate_up_stacked[e])), blk.down_stacked[e], keep_fp32=True)
    dist.all_reduce(part)
    want = oracle.linear(oracle.silu_and_mul(oracle.linear(x, torch.cat([wg[e], wu[e]]))), wd[e], keep_fp32=True)
    assert (part - want).abs().max().item() < 2e-2
    # the replicated router takes the same decisions on every rank
    w, ids = oracle.layers.moe_route(oracle.linear(x, blk.gate.weight.data), K)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    assert all(torch.equal(t, gathered[0]) for t in gathered)


# ----------------------------------------------------------------------------- tests
def test_tp2_sharded_layers_match_oracle():
    _run("_body_sharded_layers")


@pytest.mark.parametrize("world", [4, 8])
def test_tp4_tp8_sharded_layers_match_oracle(world):
    """BASELINE configs[2]/[3] degrees: at 8 ranks every rank owns exactly one kv head"""
    _run("_body_sharded_layers", world)


def test_tp4_moe_expert_shards():
    _run("_body_moe_expert_shards", 4)


def test_tp2_rpc_channel():
    _run("_body_rpc_channel")


@pytest.mark.parametrize("world", [2, 4])
def test_rpc_channel_messages_larger_than_a_slot(world):
    _run("_body_rpc_channel_multipart", world)


def test_rpc_channel_fails_loudly_when_a_worker_stops_reading():
    _run("_body_rpc_channel_reader_gone", 2)


def test_tp2_replicated_metadata():
    _run("_body_replicated_scheduling")


def _body_replicas_are_not_tensor_parallel(rank, world, port):
    """A default process group without tensor parallelism (bench.py --mode replicas): once the runner has declared
    the TP group (size 1), the layers are whole and never call a collective - whatever the world size says."""
    from nanovllm.layers import parallel
    from nanovllm.layers.linear import QKVParallelLinear, RowParallelLinear
    from nanovllm.layers.embed_head import VocabParallelEmbedding

    assert parallel.tp_size() == world and parallel.tp_rank() == rank  # undeclared: the default group
    parallel.set_tp(0, 1)
    try:
        assert parallel.tp_size() == 1 and parallel.tp_rank() == 0
        qkv = QKVParallelLinear(64, 16, 4, 2)
        assert qkv.weight.shape == ((4 + 2 * 2) * 16, 64)
        assert RowParallelLinear(64, 32).weight.shape == (32, 64)
        emb = VocabParallelEmbedding(128, 32)
        assert emb.weight.shape == (128, 32) and (emb.vocab_start_idx, emb.vocab_end_idx) == (0, 128)
        t = torch.ones(4)
        assert parallel.all_reduce_sum(t) is t and float(t.sum()) == 4.0  # no collective
    finally:
        parallel.reset_tp()
    assert parallel.tp_size() == world


def test_replica_processes_share_a_group_but_not_their_layers():
    _run("_body_replicas_are_not_tensor_parallel")
