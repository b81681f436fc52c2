"""Unit tests of the CPU-side engine pieces, following the intent of the reference's ut/ suite
(ut/test_block_manager.py, ut/test_multi_rank_block_manager.py, ut/test_scheduler.py) but derived
from the reference CODE where its tests disagree with it (three ut/ cases assume a
"len(seq)+1" convention that block_manager.py:99-118 does not implement)."""
from types import SimpleNamespace

import numpy as np
import torch
import pytest

from nanovllm.config import Config
from nanovllm.engine import batch_meta
from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import FinishReason, Sequence, SequenceStatus
from nanovllm.sampling_params import SamplingParams


def seq(tokens, block_size=4, **sp):
    return Sequence(list(tokens), SamplingParams(**sp), block_size=block_size)


# ----------------------------------------------------------------------------- block manager
def test_allocate_and_deallocate_counts():
    bm = BlockManager(10, 4)
    s = seq(range(10))  # 3 blocks
    assert bm.can_allocate(s)
    bm.allocate(s)
    assert len(s.block_table) == 3 and len(bm.free_block_ids) == 7 and bm.used_block_ids == {0, 1, 2}
    bm.deallocate(s)
    assert s.block_table == [] and len(bm.free_block_ids) == 10 and not bm.used_block_ids
    assert list(bm.free_block_ids)[-3:] == [2, 1, 0]  # released last-block-first (block_manager.py:90-97)


def test_prefix_cache_hit_shares_full_blocks_only():
    bm = BlockManager(10, 4)
    a, b = seq([1, 2, 3, 4, 5, 6, 7, 8, 9]), seq([1, 2, 3, 4, 5, 6, 7, 8, 10])
    bm.allocate(a)
    bm.allocate(b)
    assert a.block_table[:2] == b.block_table[:2] and a.block_table[2] != b.block_table[2]
    assert (a.num_cached_tokens, b.num_cached_tokens) == (0, 8)
    assert bm.blocks[a.block_table[0]].ref_count == 2
    bm.deallocate(a)
    assert bm.blocks[b.block_table[0]].ref_count == 1 and b.block_table[0] in bm.used_block_ids
    # a divergent first block disables every later hit (block_manager.py:65,74-75)
    c = seq([9, 2, 3, 4, 5, 6, 7, 8])
    bm.allocate(c)
    assert c.num_cached_tokens == 0 and not set(c.block_table) & set(b.block_table)


def test_freed_block_is_revived_by_hash():
    bm = BlockManager(6, 4)
    a = seq([1, 2, 3, 4, 5])
    bm.allocate(a)
    first = a.block_table[0]
    bm.deallocate(a)
    b = seq([1, 2, 3, 4, 6])
    bm.allocate(b)
    assert b.block_table[0] == first and b.num_cached_tokens == 4  # revived, not re-taken from the head


def test_can_append_and_may_append_follow_the_code_convention():
    """len(seq) already counts the token whose KV this step writes (scheduler calls may_append after
    the previous step's append_token)."""
    bm = BlockManager(3, 4)
    s = seq([1, 2, 3, 4])  # exactly one full block
    bm.allocate(s)
    assert bm.blocks[s.block_table[0]].hash != -1  # sealed at allocate
    s.append_token(5)  # len 5 -> 5 % 4 == 1: this step needs a new block
    assert bm.can_append(s)
    bm.may_append(s)
    assert len(s.block_table) == 2 and bm.blocks[s.block_table[1]].hash == -1
    for t in (6, 7):
        s.append_token(t)
        bm.may_append(s)
        assert len(s.block_table) == 2
    s.append_token(8)  # len 8 -> 8 % 4 == 0: the tail block fills and is sealed with a chained hash
    bm.may_append(s)
    tail = bm.blocks[s.block_table[1]]
    assert tail.hash == BlockManager.compute_hash([5, 6, 7, 8], bm.blocks[s.block_table[0]].hash)
    assert bm.hash_to_block_id[tail.hash] == tail.block_id
    # no free block left for a sequence that needs one
    bm2 = BlockManager(1, 4)
    s2 = seq([1, 2, 3, 4])
    bm2.allocate(s2)
    s2.append_token(9)
    assert not bm2.can_append(s2)
    s2.append_token(9)  # len 6: no block needed
    assert bm2.can_append(s2)


def test_two_managers_stay_identical_and_exhaustion_raises():
    """Replicated bookkeeping is deterministic (the reference's whole multi-rank story)."""
    managers = [BlockManager(8, 4), BlockManager(8, 4)]
    tables = []
    for bm in managers:
        s1, s2 = seq(range(9)), seq(list(range(8)) + [99])
        bm.allocate(s1)
        bm.allocate(s2)
        s1.append_token(7)
        bm.may_append(s1)
        tables.append((list(s1.block_table), list(s2.block_table), dict(bm.hash_to_block_id)))
    assert tables[0] == tables[1]
    bm = BlockManager(1, 4)
    big = seq(range(9))
    assert not bm.can_allocate(big)
    with pytest.raises(IndexError):
        bm.allocate(big)  # free list runs empty (ut/test_multi_rank_block_manager.py:71-81)


# ----------------------------------------------------------------------------- scheduler
def sched(**kw):
    cfg = dict(max_num_seqs=4, max_num_batched_tokens=32, eos=99, num_kvcache_blocks=9, kvcache_block_size=4,
               max_model_len=64)
    cfg.update(kw)
    return Scheduler(SimpleNamespace(**cfg))


def test_prefill_first_and_token_budget():
    sc = sched(max_num_batched_tokens=12)
    a, b = seq(range(8)), seq(range(100, 108))
    sc.add(a)
    sc.add(b)
    assert not sc.is_finished()
    seqs, is_prefill = sc.schedule()
    assert is_prefill and seqs == [a] and b in sc.waiting  # 8 + 8 > 12: head-of-line stop
    assert a.status is SequenceStatus.RUNNING and len(sc.block_manager.free_block_ids) == 8 - 2
    seqs, is_prefill = sc.schedule()
    assert is_prefill and seqs == [b]  # prefill keeps priority over decoding a


def test_decode_preempts_from_the_right_and_requeues_at_the_front():
    sc = sched(num_kvcache_blocks=5)  # 4 usable blocks
    a, b = seq(range(8), max_tokens=8), seq(range(50, 58), max_tokens=8)
    sc.add(a)
    sc.add(b)
    seqs, _ = sc.schedule()
    assert seqs == [a, b] and not sc.block_manager.free_block_ids
    sc.postprocess(seqs, [1, 1])  # both now need a 3rd block, none is free
    seqs, is_prefill = sc.schedule()
    assert not is_prefill and seqs == [a]
    assert b.status is SequenceStatus.WAITING and b.finish_reason is FinishReason.PREEMPTED
    assert sc.waiting[0] is b and b.block_table == []
    assert len(a.block_table) == 3


def test_everything_preempted_returns_empty_decode_batch():
    sc = sched(num_kvcache_blocks=3)  # 2 usable blocks
    a = seq(range(8), max_tokens=4)
    sc.add(a)
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [1])
    seqs, is_prefill = sc.schedule()
    assert seqs == [] and not is_prefill and sc.waiting[0] is a  # the runner must tolerate an empty batch


def test_finish_conditions():
    sc = sched(max_model_len=10)
    a = seq(range(4), max_tokens=2)
    b = seq(range(4), max_tokens=50)
    c = seq(range(8), max_tokens=50)
    d = seq(range(4), max_tokens=50, ignore_eos=True)
    for s in (a, b, c, d):
        sc.add(s)
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [5, 99, 5, 99])
    assert b.is_finished and b.finish_reason is FinishReason.EOS and b not in sc.running
    assert not d.is_finished  # EOS ignored
    seqs, _ = sc.schedule()
    assert seqs == [a, c, d]
    sc.postprocess(seqs, [5, 5, 5])
    assert a.finish_reason is FinishReason.LENGTH and a.num_completion_tokens == 2
    assert c.finish_reason is FinishReason.LENGTH  # 8 + 2 == max_model_len
    assert not d.is_finished and list(sc.running) == [d]
    assert len(sc.block_manager.free_block_ids) == 8 - len(d.block_table)


def test_abort_request():
    sc = sched()
    a = Sequence([1, 2, 3], SamplingParams(), request_id="r1", block_size=4)
    b = Sequence([4, 5, 6], SamplingParams(), request_id="r2", block_size=4)
    sc.add(a)
    sc.add(b)
    sc.schedule()
    sc.abort_seq_group("r1")
    assert a.finish_reason is FinishReason.ABORTED and list(sc.running) == [b] and a.block_table == []


# ----------------------------------------------------------------------------- metadata / config / params
def test_decode_meta_padding_matches_reference_layout():
    a, b = seq(range(38), block_size=16), seq(range(43), block_size=16)
    a.block_table, b.block_table = [0, 1, 2], [3, 4, 5]
    m = batch_meta.decode_meta([a, b], pad_to=4, dummy_block=63, table_cols=8)
    assert m.slot_mapping.tolist() == [[2, 5], [5, 10], [63, 0], [63, 0]]  # SURVEY.md §8c probe values
    assert m.context_lens.tolist() == [38, 43, 0, 0] and m.positions.tolist() == [37, 42, 0, 0]
    assert m.block_tables.shape == (4, 8) and m.block_tables[2:].tolist() == [[-1] * 8] * 2
    assert m.block_tables.dtype == np.int32 and m.slot_mapping.dtype == np.int32 and m.input_ids.dtype == np.int64


def test_sampling_params_and_config_validation(tmp_path):
    with pytest.raises(AssertionError):
        SamplingParams(temperature=0.0)  # as sampling_params.py:10-11
    assert SamplingParams(greedy=True).temperature == 1.0
    with pytest.raises(AssertionError):
        Config(str(tmp_path / "missing"))
    import json

    (tmp_path / "config.json").write_text(json.dumps(dict(
        architectures=["Qwen3ForCausalLM"], model_type="qwen3", hidden_size=128, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, head_dim=128, intermediate_size=256, vocab_size=256,
        max_position_embeddings=512, eos_token_id=7)))
    cfg = Config(str(tmp_path), max_model_len=4096, max_num_batched_tokens=16384)
    assert cfg.max_model_len == 512 and cfg.eos == 7 and cfg.use_graphs  # clamped to max_position_embeddings
    with pytest.raises(AssertionError):
        Config(str(tmp_path), kvcache_block_size=24)
    with pytest.raises(AssertionError):
        Config(str(tmp_path), tensor_parallel_size=9)
    assert not Config(str(tmp_path), graph_mode="eager").use_graphs
    assert Config(str(tmp_path), graph_mode="max-autotune").use_graphs  # reference alias


def test_c_abi_exports_every_declared_symbol_and_rejects_bad_arguments():
    """The library loads without a GPU; every header symbol is exported; argument validation happens
    on the host (no compute calls here)."""
    import ctypes
    import re

    from conftest import REPO
    from nanovllm import _C

    header = open(f"{REPO}/include/mi355_nanovllm.h").read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", header)) - {"mi_bf16", "mi_stream"}
    assert declared and declared == set(_C.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(_C.lib, name)
    assert _C.lib.mi_rmsnorm(None, 0, None, None, 1, 1, 128, ctypes.c_float(1e-6), None) == -1  # MI_EINVAL
    assert _C.lib.mi_gemm_bf16_skinny(1 << 20, 1 << 20, None, 1 << 20, 513, 16, 32, None) == -2  # M > 512 rows
    assert _C.lib.mi_gemm_bf16(1 << 20, 96, 1 << 20, None, 1 << 20, 64, 128, 64, 96, 0, None, 0, None) == -2  # K % 64
    assert _C.lib.mi_gemm_bf16(1 << 20, 64, 1 << 20, None, 1 << 20, 64, 128, 64, 64, 2, None, 0, None) == -1  # epilogue
    assert _C.lib.mi_gemm_bf16_workspace(16384, 4096, 1024, 0) == 0           # plenty of tiles: no K slices
    ws = _C.lib.mi_gemm_bf16_workspace(1024, 1024, 3072, 0)  # 64 tiles of 128 x 128, 48 K steps: summed in K slices
    assert ws % (1024 * 1024 * 4) == 0 and ws // (1024 * 1024 * 4) in (2, 4, 8, 16)
    assert _C.lib.mi_paged_attn_decode_workspace(32, 16) == 32 * 16 * 16 * 130 * 4
    # the Python side asks the library for mi_gemm_bf16's shape contract instead of restating it.  M = 0
    # makes mi_gemm_bf16 return its contract check without launching: both must agree on the weight side ...
    from nanovllm.layers import linear

    for n, k in ((4096, 1024), (1024, 1376), (1022, 1024), (152064, 8192), (151936, 1024), (65536, 16320), (65536, 16384)):
        rc = _C.lib.mi_gemm_bf16(1 << 20, k, 1 << 20, None, 1 << 20, n, 0, n, k, 0, None, 0, None)
        assert (rc == 0) == (_C.lib.mi_gemm_bf16_max_rows(n, k, k) > 0) == linear.tile_gemm_takes(1, n, k), (n, k, rc)
    # ... and the row limit is exactly the 31-bit operand offset of the kernel's DMA descriptors
    for k in (64, 1024, 25600, 1 << 20):
        m = _C.lib.mi_gemm_bf16_max_rows(64, k, k)
        assert (m + 256) * k * 2 < 1 << 31 <= (m + 257) * k * 2
    assert linear.tile_gemm_takes(65536, 1024, 25600) and _C.lib.mi_gemm_bf16_max_rows(1024, 25600, 25600) < 65536
    assert not linear.tile_gemm_takes(512, 152064, 8192)  # a weight beyond the offsets: the streaming pieces take it
    assert b"unsupported" in _C.lib.mi_strerror(-2).lower() or b"not supported" in _C.lib.mi_strerror(-2).lower()
    assert _C.lib.mi_kv_elem_offset(0, 5, 1, 37, 2, 16) == 1 * 2048 + (37 // 32) * 512 + (((37 % 32) // 8) * 16 + 5) * 8 + 37 % 8
    with pytest.raises(_C.MiError):
        import torch

        from nanovllm import ops

        ops.silu_mul(torch.zeros(2, 16, dtype=torch.bfloat16))  # CPU tensor: no fallback


@pytest.mark.parametrize("n_stagers", [1, 2])
def test_decode_stager_equals_decode_meta_under_preemption_and_churn(n_stagers):
    """The incremental staging writer (what the runner uploads every step) must equal the plain
    decode_meta() at every step of a run with tight memory: preemptions rebuild block tables,
    finished sequences hand their rows to others.  n_stagers = 2: the runner's two alternating pinned buffers,
    each writer sees every other step (its rows may be several appended blocks behind)."""
    import random

    rng = random.Random(3)
    max_seqs, bs, cols, nblk = 6, 4, 16, 19
    sc = sched(max_num_seqs=max_seqs, max_num_batched_tokens=64, num_kvcache_blocks=nblk, kvcache_block_size=bs,
               max_model_len=cols * bs - 1)
    stages = [dict(ids=np.zeros(max_seqs, np.int64), pos=np.zeros(max_seqs, np.int64),
                   ctx=np.zeros(max_seqs, np.int32), slots=np.zeros((max_seqs, 2), np.int32),
                   tables=np.zeros((max_seqs, cols), np.int32)) for _ in range(n_stagers)]
    stagers = [batch_meta.DecodeStager(**st) for st in stages]
    pending = [seq([rng.randrange(1, 90) for _ in range(rng.randrange(2, 14))], block_size=bs,
                   max_tokens=rng.randrange(3, 25), ignore_eos=True) for _ in range(14)]
    decode_steps = preemptions = 0
    step = 0
    while pending or not sc.is_finished():
        if pending and step % 3 == 0:
            sc.add(pending.pop())
        step += 1
        if sc.is_finished():
            continue
        before = {id(s): s.table_gen for s in sc.running}
        seqs, is_prefill = sc.schedule()
        if not seqs:
            continue
        if is_prefill:
            preemptions += sum(1 for s in seqs if s.table_gen > 1)
        else:
            decode_steps += 1
            bucket = next(b for b in (1, 2, 4, max_seqs) if b >= len(seqs))
            stage, stager = stages[decode_steps % n_stagers], stagers[decode_steps % n_stagers]
            stager.fill(seqs, bucket, nblk - 1)
            want = batch_meta.decode_meta(seqs, pad_to=bucket, dummy_block=nblk - 1, table_cols=cols)
            assert stage["ids"][:bucket].tolist() == want.input_ids.tolist()
            assert stage["pos"][:bucket].tolist() == want.positions.tolist()
            assert stage["ctx"][:bucket].tolist() == want.context_lens.tolist()
            assert stage["slots"][:bucket].tolist() == want.slot_mapping.tolist()
            assert stage["tables"][:bucket].tolist() == want.block_tables.tolist(), step
        sc.postprocess(seqs, [rng.randrange(1, 90) for _ in seqs])
    assert decode_steps > 30 and preemptions > 0  # the run did exercise re-prefill after preemption


def test_prefill_meta_skipping_cached_prefix_blocks():
    """skip_cached=True: only the tokens behind the leading cache-hit blocks are fed; a fully cached
    prompt keeps its last token (for the logits) but does not rewrite its shared KV row."""
    bs = 4
    bm = BlockManager(16, bs)
    a = seq(list(range(10, 21)), block_size=bs)           # 11 tokens: 2 full blocks + 3
    b = seq(list(range(10, 18)) + [7, 7, 7], block_size=bs)  # shares the 2 full blocks
    c = seq(list(range(10, 18)), block_size=bs)           # exactly the 2 shared blocks
    for s_ in (a, b, c):
        bm.allocate(s_)
    assert (a.num_prefix_tokens, b.num_prefix_tokens, c.num_prefix_tokens) == (0, 8, 8)
    assert b.block_table[:2] == a.block_table[:2] == c.block_table
    full = batch_meta.prefill_meta([a, b, c], bs)
    m = batch_meta.prefill_meta([a, b, c], bs, skip_cached=True)
    assert m.cu_seqlens_q.tolist() == [0, 11, 14, 15] and m.cu_seqlens_k.tolist() == [0, 11, 22, 30]
    assert m.kv_lens.tolist() == [11, 11, 8] and (m.max_seqlen_q, m.max_seqlen_k) == (11, 11)
    assert m.input_ids.tolist() == a.token_ids + [7, 7, 7] + [17]
    assert m.positions.tolist() == list(range(11)) + [8, 9, 10] + [7]
    assert m.slot_mapping[:11].tolist() == full.slot_mapping[:11].tolist()
    assert m.slot_mapping[11:14].tolist() == full.slot_mapping[11 + 8:22].tolist()
    assert m.slot_mapping[14:].tolist() == [-1]
    assert (m.block_tables == full.block_tables).all()
    # the reference's reporting counter keeps growing over a preemption; the prefix count does not
    bm.deallocate(b)
    assert b.num_prefix_tokens == 0
    bm.allocate(b)
    assert b.num_prefix_tokens == 8 and b.num_cached_tokens == 16


@pytest.mark.parametrize("tied", [True, False])
def test_safetensors_checkpoint_lands_in_the_packed_parameters(tmp_path, tied):
    """utils/loader.py:12-59 with qwen3.py:189-195's packed_modules_mapping: a HF Qwen3 checkpoint
    written by transformers (safetensors) must land bit-equal in qkv_proj (rows q|k|v), gate_up_proj
    (gate|up), the norms and the (tied or separate) head."""
    from transformers import Qwen3Config
    from transformers import Qwen3ForCausalLM as HfQwen3

    from model_configs import TINY
    from nanovllm.models.qwen3 import Qwen3ForCausalLM
    from nanovllm.utils.loader import has_checkpoint, load_model

    cfg = Qwen3Config(**{k: v for k, v in dict(TINY, tie_word_embeddings=tied).items()
                         if k not in ("architectures", "model_type", "torch_dtype")})
    torch.manual_seed(0)
    hf = HfQwen3(cfg).to(torch.bfloat16)
    hf.save_pretrained(str(tmp_path), safe_serialization=True)
    assert has_checkpoint(str(tmp_path))
    model = Qwen3ForCausalLM(cfg)
    load_model(model, str(tmp_path))
    sd = hf.state_dict()
    for i, layer in enumerate(model.model.layers):
        p = f"model.layers.{i}."
        attn, mlp = layer.self_attn, layer.mlp
        assert torch.equal(attn.qkv_proj.weight, torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"]))
        assert torch.equal(mlp.gate_up_proj.weight,
                           torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]]))
        assert torch.equal(attn.o_proj.weight, sd[p + "self_attn.o_proj.weight"])
        assert torch.equal(mlp.down_proj.weight, sd[p + "mlp.down_proj.weight"])
        assert torch.equal(attn.q_norm.weight, sd[p + "self_attn.q_norm.weight"])
        assert torch.equal(attn.k_norm.weight, sd[p + "self_attn.k_norm.weight"])
        assert torch.equal(layer.input_layernorm.weight, sd[p + "input_layernorm.weight"])
        assert torch.equal(layer.post_attention_layernorm.weight, sd[p + "post_attention_layernorm.weight"])
    assert torch.equal(model.model.norm.weight, sd["model.norm.weight"])
    assert torch.equal(model.model.embed_tokens.weight, sd["model.embed_tokens.weight"])
    shared = model.lm_head.weight.data_ptr() == model.model.embed_tokens.weight.data_ptr()
    assert shared == tied  # qwen3.py:204-205
    assert torch.equal(model.lm_head.weight, sd["lm_head.weight"])


def test_moe_model_builds_only_its_own_layers_and_flags_ignored_config_fields():
    """models/qwen3_moe.py: a sparse layer allocates no dense MLP (and no dense stack is built and thrown away);
    config fields the reference ignores - norm_topk_prob = false (qwen3_moe.py:156-158 renormalises always),
    rope_scaling (rotary_embedding.py:52-69) - are reported, not silently dropped; expert widths without a grouped
    GEMM instantiation are known before the model is built"""
    from nanovllm import _C
    from nanovllm.models import qwen3
    from nanovllm.models.qwen3_moe import Qwen3MoeForCausalLM, Qwen3MoeSparseMoeBlock
    from model_configs import TINY_MOE

    built = []
    orig = qwen3.Qwen3MLP.__init__
    qwen3.Qwen3MLP.__init__ = lambda self, *a, **k: (built.append(a), orig(self, *a, **k))[1]
    try:
        cfg = SimpleNamespace(**dict(TINY_MOE, mlp_only_layers=[1]))
        with pytest.warns(UserWarning, match="norm_topk_prob"):
            model = Qwen3MoeForCausalLM(SimpleNamespace(**dict(vars(cfg), norm_topk_prob=False)))
        assert len(built) == 1  # layer 1 is dense, layer 0 sparse: ONE dense MLP in total
        assert isinstance(model.model.layers[0].mlp, Qwen3MoeSparseMoeBlock)
        assert isinstance(model.model.layers[1].mlp, qwen3.Qwen3MLP)
        qwen3.Qwen3Attention._warned_rope_scaling = False
        with pytest.warns(UserWarning, match="rope_scaling"):
            qwen3.Qwen3ForCausalLM(SimpleNamespace(**dict(TINY_MOE, rope_scaling={"rope_type": "llama3", "factor": 8.0})))
    finally:
        qwen3.Qwen3MLP.__init__ = orig
    assert _C.lib.mi_moe_shapes_supported(2048, 768) == 0 and _C.lib.mi_moe_shapes_supported(2048, 192) == 0
    assert _C.lib.mi_moe_shapes_supported(2048, 96) == -2  # Qwen3-30B-A3B experts at TP 8: no kernel, said at start-up


def test_sequence_ids_array_tracks_the_token_list():
    """Sequence.ids_array (the int64 copy the block hashes and the prefill staging read): the prompt part is converted
    once, the completion part on every call - it must equal token_ids after appends, after a pending token was resolved
    (lookahead) and for a sequence rebuilt from the rank wire format"""
    from array import array

    from nanovllm._C import xxh64_chain_blocks

    s = Sequence(list(range(100, 140)), SamplingParams(max_tokens=8), block_size=16)
    assert s.ids_array() is s.ids_array() and list(s.ids_array()) == s.token_ids  # cached, equal
    s.append_token(7)
    s.append_pending()
    assert list(s.ids_array()) == s.token_ids and len(s.ids_array()) == 42
    s.resolve_pending(9)
    assert list(s.ids_array()) == s.token_ids and s.ids_array()[-1] == 9
    assert xxh64_chain_blocks(s.ids_array(), 2, 16) == xxh64_chain_blocks(s.token_ids, 2, 16)
    wire = np.asarray(s.to_wire(True), dtype=np.int64)
    t, _ = Sequence.from_wire(wire)
    assert list(t.ids_array()) == s.token_ids and isinstance(t.ids_array(), array)
    m = batch_meta.prefill_meta([s], 16)  # no block table: slots -1, ids from the array
    assert m.input_ids.tolist() == s.token_ids


def test_chained_block_hashes_in_one_call_equal_the_per_block_chain():
    import random
    from array import array

    from nanovllm._C import lib, xxh64_chain, xxh64_chain_blocks

    rng = random.Random(1)
    for bs, n_tokens in ((4, 11), (16, 64), (16, 1030), (256, 700)):
        toks = [rng.randrange(0, 151936) for _ in range(n_tokens)]
        want, prev = [], -1
        for i in range(n_tokens // bs):
            prev = xxh64_chain(array("q", toks[i * bs:(i + 1) * bs]).tobytes(), prev)
            want.append(prev)
        assert xxh64_chain_blocks(toks, n_tokens // bs, bs) == want
        assert BlockManager.compute_hash(toks[:bs]) == (want[0] if want else BlockManager.compute_hash(toks[:bs]))
    assert xxh64_chain_blocks([1, 2, 3], 0, 4) == []
    assert lib.mi_xxh64_chain_blocks(None, 2, 4, 0, 0, None) == -1  # MI_EINVAL
    # the communicator entry points validate before touching the device
    assert lib.mi_comm_region_bytes(0, 1024) == 0 and lib.mi_comm_region_bytes(9, 1024) == 0
    assert lib.mi_comm_region_bytes(8, 65536) > 2 * 8 * 65536
    assert lib.mi_comm_create(0, 2, None, 1024, None) == -1
    assert lib.mi_allreduce_sum_bf16(None, None, None, 8, None) == -1


def test_block_manager_invariants_under_random_traffic():
    """Structural invariants of the allocator under seeded random admit / decode / preempt / finish traffic
    (the golden traces pin the exact ids; this pins what must hold for ANY stream):
      * every block id is either in the free list or in use, never both, never twice in the free list;
      * a block's ref_count equals the number of live block tables that contain it;
      * a live table never holds a block twice, and holds exactly ceil(len / block_size) blocks;
      * every full block of a live sequence carries the chained hash of its tokens, the partial tail none;
      * num_prefix_tokens counts only leading blocks whose stored tokens equal the sequence's."""
    import random
    from collections import Counter

    for seed in range(6):
        rng = random.Random(seed)
        bs, nblk = rng.choice([4, 16]), rng.randrange(24, 60)
        bm = BlockManager(nblk, bs)
        prefixes = [[rng.randrange(1, 50) for _ in range(bs * rng.randrange(1, 3))] for _ in range(3)]
        live: list = []

        def check():
            free = list(bm.free_block_ids)
            assert len(free) == len(set(free)) and not (set(free) & bm.used_block_ids)
            assert set(free) | bm.used_block_ids == set(range(nblk))
            refs = Counter(b for s_ in live for b in s_.block_table)
            for i, blk in enumerate(bm.blocks):
                assert blk.ref_count == refs.get(i, 0), (i, blk.ref_count, refs.get(i, 0))
            for s_ in live:
                t = s_.block_table
                assert len(t) == len(set(t)) == s_.num_blocks
                chain = -1
                for i, b in enumerate(t):
                    toks = s_.block(i)
                    if len(toks) == bs and (i < len(t) - 1 or len(s_) % bs == 0):
                        chain = BlockManager.compute_hash(toks, chain)
                        if bm.blocks[b].hash != -1:  # sealed (a decode-grown tail is sealed by may_append)
                            assert bm.blocks[b].hash == chain and bm.blocks[b].token_ids == toks
                assert s_.num_prefix_tokens % bs == 0 and s_.num_prefix_tokens <= len(s_)

        for _ in range(400):
            op = rng.random()
            if op < 0.35:  # admit
                body = [rng.randrange(1, 50) for _ in range(rng.randrange(1, 3 * bs))]
                toks = (rng.choice(prefixes) + body) if rng.random() < 0.6 else body
                s_ = seq(toks, block_size=bs, max_tokens=64, ignore_eos=True)
                if bm.can_allocate(s_):
                    bm.allocate(s_)
                    live.append(s_)
            elif op < 0.8 and live:  # one decode step for a random live sequence
                s_ = rng.choice(live)
                s_.append_token(rng.randrange(1, 50))
                if bm.can_append(s_):
                    bm.may_append(s_)
                else:  # no block for the new token: preempt it (scheduler.py:79-83)
                    s_.token_ids.pop()
                    s_.num_tokens -= 1
                    s_.last_token = s_.token_ids[-1]
                    bm.deallocate(s_)
                    live.remove(s_)
            elif live:  # finish / preempt
                s_ = live.pop(rng.randrange(len(live)))
                bm.deallocate(s_)
                assert not s_.block_table and s_.num_prefix_tokens == 0
            check()
        for s_ in live:
            bm.deallocate(s_)
        live.clear()
        check()
        assert len(bm.free_block_ids) == nblk


# --------------------------------------------------------------------------- serving harness
def test_serving_harness_metrics_against_scripted_engine():
    """bench/serving_bench.py: arrivals interleave with steps, TTFT / TPOT / latency follow the reference's
    definitions (serving_bench.py:35-58) - checked against a scripted engine with a virtual clock."""
    import importlib.util
    import os
    import sys

    from conftest import PKG

    spec = importlib.util.spec_from_file_location("serving_bench", os.path.join(PKG, "bench", "serving_bench.py"))
    sb = importlib.util.module_from_spec(spec)
    sys.modules["serving_bench"] = sb  # dataclasses resolve their module through sys.modules
    spec.loader.exec_module(sb)

    now = [0.0]
    clock = lambda: now[0]  # noqa: E731

    class Engine:
        """every step takes 10 ms; a request's first step is its prefill (first token), each later step one token"""
        def __init__(self):
            self.seqs, self.ttft, self.next_id = [], {}, 7

        def add_request(self, prompt, sp):
            s = SimpleNamespace(seq_id=self.next_id, arrival_time=clock(), out=[], max_tokens=sp.max_tokens)
            self.next_id += 1
            self.seqs.append(s)
            return s

        def is_finished(self):
            return not self.seqs

        def step(self):
            now[0] += 0.010
            finished = []
            for s in list(self.seqs):
                s.out.append(1)
                if len(s.out) == 1:
                    self.ttft[s.seq_id] = clock() - s.arrival_time
                if len(s.out) == s.max_tokens:
                    finished.append((s.seq_id, s.out, 3, 0))
                    self.seqs.remove(s)
            return finished, -1

    def sleep(dt):
        now[0] += max(dt, 1e-3)

    prompts = [[1, 2, 3]] * 3
    sps = [SimpleNamespace(max_tokens=n) for n in (4, 2, 1)]
    arrivals = np.array([0.0, 0.015, 0.5])  # the third arrives long after the others have finished
    res = sb.run_serving(Engine(), prompts, sps, arrivals, clock=clock, sleep=sleep)
    m = res.metrics
    assert sorted(m) == [7, 8, 9]
    assert abs(m[7].ttft - 0.010) < 1e-9 and abs(m[7].latency - 0.040) < 1e-9 and abs(m[7].tpot - 0.010) < 1e-9
    # the second request is picked up at the first loop turn after t = 15 ms, i.e. behind step 2 (t = 20 ms)
    assert abs(m[8].submission_time - 0.020) < 1e-9 and abs(m[8].ttft - 0.010) < 1e-9 and m[8].output_len == 2
    assert m[9].output_len == 1 and np.isnan(m[9].tpot) and m[9].submission_time >= 0.5
    s = res.summary()
    assert s["completed"] == 3 and s["output_tokens"] == 7 and s["engine_steps"] == 5
    assert abs(s["ttft_ms"]["p50"] - 10.0) < 1e-6
    # arrival processes: exponential gaps have the right mean; the reference's integer gaps are mostly zero
    rng = np.random.default_rng(0)
    a = sb.arrival_times(4000, 8.0, rng)
    assert abs(np.diff(a).mean() - 0.125) < 0.01 and (np.diff(a) > 0).all()
    b = sb.arrival_times(4000, 8.0, rng, reference_style=True)
    assert (np.diff(b) == 0).mean() > 0.8


# ----------------------------------------------------------------------------- lookahead decode
class _ScriptedRunner:
    """Stands in for ModelRunner on the host: a 'model' whose next token is a hash of (input id, position,
    first prompt token), evaluated at LAUNCH time from what the device would see - the previous step's token
    buffer for rows that name one (src >= 0), the host's last_token otherwise.  Records every decode step."""
    VOCAB = 23

    def __init__(self, max_rows=8):
        self.max_launch_rows = max_rows
        self.device_tokens = []  # tokens of the step launched last ("tokens_dev")
        self.steps = []          # (seq ids, positions, input ids, block tables) per decode step
        self.prefills = []       # seq ids per prefill step

    @classmethod
    def _next(cls, seq, input_id, position):
        return (input_id * 7 + position * 3 + seq.token_ids[0]) % cls.VOCAB

    def can_launch_decode(self, n):
        return 0 < n <= self.max_launch_rows

    def _decode(self, seqs, src):
        ids = [self.device_tokens[r] if r >= 0 else s.last_token for s, r in zip(seqs, src)]
        self.steps.append(([s.seq_id for s in seqs], [s.num_tokens - 1 for s in seqs], ids,
                           [list(s.block_table) for s in seqs]))
        self.device_tokens = [self._next(s, i, s.num_tokens - 1) for s, i in zip(seqs, ids)]
        return list(self.device_tokens)

    def launch_decode(self, seqs, src_rows=None):
        return self._decode(seqs, src_rows if src_rows is not None else [-1] * len(seqs))

    def collect(self, handle):
        return handle

    can_launch_prefill = True

    def launch_prefill(self, seqs):
        self.prefills.append([s.seq_id for s in seqs])
        self.device_tokens = [self._next(s, s.last_token, s.num_tokens - 1) for s in seqs]
        return list(self.device_tokens)

    def collect_prefill(self, handle):
        return handle

    def prefill_done(self, handle):
        return False

    def prefill_device_ms(self, handle):
        return 0.0

    def call(self, name, seqs, *args):
        if name == "launch_decode":
            return self.launch_decode(seqs, *args)
        if name == "launch_prefill":
            return self.launch_prefill(seqs)
        assert name == "run"
        is_prefill = args[0]
        if is_prefill:
            return self.launch_prefill(seqs)
        return self._decode(seqs, [-1] * len(seqs))


def _scripted_engine(lookahead, **cfg):
    from nanovllm.engine.llm_engine import LLMEngine

    cfg_gc = cfg.pop("gc_control", False)

    eng = object.__new__(LLMEngine)
    eng.scheduler = sched(**cfg)
    eng.model_runner = _ScriptedRunner()
    eng.block_size = eng.scheduler.block_manager.block_size
    eng.tokenizer, eng.ttft, eng.lookahead, eng._inflight = None, {}, lookahead, None
    eng._inflight_prefill, eng.prefill_lookahead_min_tokens, eng.prefill_lookahead_launches = None, 0, 0
    from nanovllm.engine.host_gc import HostGc

    eng.gc, eng.prefill_trace = HostGc(enabled=cfg_gc), []
    return eng


def _drain(eng, arrivals=()):
    """run to completion; arrivals = [(step index, prompt, SamplingParams)] added before that step"""
    done, step = {}, 0
    pending = sorted(arrivals, key=lambda a: a[0])
    while not eng.is_finished() or pending:
        while pending and pending[0][0] <= step:
            _, prompt, sp = pending.pop(0)
            eng.add_request(prompt, sp)
        for seq_id, toks, _, _ in eng.step()[0]:
            assert seq_id not in done
            done[seq_id] = list(toks)
        step += 1
    return done


@pytest.mark.parametrize("eos", [-1, 5])
def test_lookahead_decode_reproduces_the_synchronous_engine(eos):
    """Engine lookahead (step k+1 scheduled and launched before step k's tokens are on the host) against the
    plain loop on a scripted model: same tokens for every request; without EOS endings also the same decode
    steps (sequences, positions, input ids, block tables) and the same allocator state - block openings,
    length endings, deferred block seals and the free list all land where the synchronous order puts them."""
    rng = np.random.default_rng(3)
    prompts = [[int(t) for t in rng.integers(0, 23, n)] for n in (5, 9, 4, 13, 7, 3)]
    lens = (9, 17, 6, 12, 30, 8)
    Sequence.counter = __import__("itertools").count()
    outs, engines = [], []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=40, max_num_batched_tokens=64, max_num_seqs=4)
        for p, n in zip(prompts[:4], lens):
            eng.add_request(p, SamplingParams(max_tokens=n, ignore_eos=False, temperature=1.0))
        late = [(7, prompts[4], SamplingParams(max_tokens=lens[4], temperature=1.0)),
                (40, prompts[5], SamplingParams(max_tokens=lens[5], temperature=1.0))]
        outs.append(_drain(eng, late))
        engines.append(eng)
    sync, look = outs
    assert sorted(sync) == sorted(look) == list(range(6))
    assert sync == look
    if eos < 0:
        assert all(len(sync[i]) == lens[i] for i in range(6))
    else:
        assert any(len(sync[i]) < lens[i] for i in range(6))  # the scripted model does hit EOS
    a, b = (e.scheduler.block_manager for e in engines)
    assert not a.used_block_ids and not b.used_block_ids
    # arrivals land one decode step later under lookahead (that step was already queued): steps agree
    # exactly up to the first late arrival; afterwards the per-request streams above are the contract
    # (with EOS endings the queued step still carries a row for a sequence that has just ended)
    sa, sb_ = engines[0].model_runner.steps, engines[1].model_runner.steps
    if eos < 0:  # ... and the same registrations of sealed blocks (hash -> tokens), whatever the step they happened in
        assert sa[:5] == sb_[:5]
        seal = lambda bm: {h: tuple(bm.blocks[i].token_ids) for h, i in bm.hash_to_block_id.items()  # noqa: E731
                           if bm.blocks[i].hash == h}
        assert set(seal(a)) == set(seal(b))


def test_lookahead_without_arrivals_is_step_for_step_identical():
    """No arrivals, no EOS: every decode step of the lookahead engine equals the synchronous engine's, and
    the allocators end in the same state (free-list order included)."""
    rng = np.random.default_rng(5)
    prompts = [[int(t) for t in rng.integers(0, 23, n)] for n in (6, 11, 3, 8)]
    engines = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=-1, num_kvcache_blocks=30, max_num_batched_tokens=64)
        for p, n in zip(prompts, (14, 5, 22, 9)):
            eng.add_request(p, SamplingParams(max_tokens=n, ignore_eos=True, temperature=1.0))
        out = _drain(eng)
        engines.append((eng, out))
    (e0, o0), (e1, o1) = engines
    assert o0 == o1
    assert e0.model_runner.steps == e1.model_runner.steps
    b0, b1 = e0.scheduler.block_manager, e1.scheduler.block_manager
    assert list(b0.free_block_ids) == list(b1.free_block_ids)
    assert b0.hash_to_block_id == b1.hash_to_block_id
    assert [(b.hash, b.token_ids) for b in b0.blocks] == [(b.hash, b.token_ids) for b in b1.blocks]


def test_lookahead_falls_back_when_a_preemption_is_needed():
    """Too few blocks for every sequence to open its next block: the lookahead declines (None), the step is
    scheduled synchronously with the reference's preemption, and the streams still equal the plain loop's."""
    prompts = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]]
    outs = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=-1, num_kvcache_blocks=6, max_num_batched_tokens=64)  # 5 usable blocks
        for p in prompts:
            eng.add_request(p, SamplingParams(max_tokens=7, ignore_eos=True, temperature=1.0))
        outs.append((_drain(eng), eng.model_runner.steps))
    assert outs[0][0] == outs[1][0] and all(len(v) == 7 for v in outs[0][0].values())
    assert outs[0][1] == outs[1][1]  # preemption and re-prefill happen at the same steps


def test_lookahead_abort_drops_the_row_of_the_queued_step():
    """abort_request between two step() calls while the aborted sequence still has a row in the queued decode
    step: the row's token is discarded when the step is collected, the other streams are untouched."""
    outs = []
    for abort in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=30, max_num_batched_tokens=64)
        for i, p in enumerate(([1, 2, 3], [4, 5, 6, 7], [8, 9])):
            eng.add_request(p, SamplingParams(max_tokens=10, ignore_eos=True, temperature=1.0), request_id=f"r{i}")
        done = {}
        for step in range(40):
            if eng.is_finished():
                break
            if abort and step == 4:
                assert eng._inflight is not None  # a decode step is queued right now
                eng.abort_request("r1")
            for seq_id, toks, _, _ in eng.step()[0]:
                done[seq_id] = list(toks)
        outs.append(done)
        assert not eng.scheduler.block_manager.used_block_ids and eng._inflight is None
    full, aborted = outs
    assert sorted(full) == [0, 1, 2] and sorted(aborted) == [0, 2]
    assert aborted[0] == full[0] and aborted[2] == full[2]


def test_lookahead_seals_decode_blocks_for_later_prefix_hits():
    """Blocks filled DURING decode are sealed (hashed, registered) one step late under lookahead - when the token
    that completes them has reached the host.  A later request whose prompt is an earlier request's prompt + output
    must find exactly the blocks the synchronous engine would have registered."""
    results = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=-1, num_kvcache_blocks=40, max_num_batched_tokens=64)
        a = eng.add_request([3, 1, 4, 1, 5, 9], SamplingParams(max_tokens=12, ignore_eos=True, temperature=1.0))
        eng.add_request([2, 7, 1, 8], SamplingParams(max_tokens=40, ignore_eos=True, temperature=1.0))
        done, step, c = {}, 0, None
        while not eng.is_finished():
            if step == 25:  # A has long finished, B is still decoding
                assert a.is_finished
                c = eng.add_request(a.prompt_token_ids + a.completion_token_ids[:10],
                                    SamplingParams(max_tokens=5, ignore_eos=True, temperature=1.0))
            for seq_id, toks, _, cached in eng.step()[0]:
                done[seq_id] = (list(toks), cached)
            step += 1
        results.append((done, c.num_cached_tokens))
    (d0, c0), (d1, c1) = results
    assert c0 == c1 == 16  # four full blocks of A: one from its prompt, three sealed while it decoded
    assert {k: v[0] for k, v in d0.items()} == {k: v[0] for k, v in d1.items()}


@pytest.mark.parametrize("seed", range(12))
def test_lookahead_fuzz_against_the_synchronous_engine(seed):
    """Seeded random request streams (lengths straddling block boundaries, tight memory so that preemptions and
    lookahead refusals occur, EOS on in half of the seeds): token streams equal the synchronous engine's; without
    EOS endings and late arrivals the decode steps and the final allocator state are equal too."""
    import random

    rng = random.Random(seed)
    eos = 5 if seed % 2 else -1
    n_req = rng.randrange(3, 9)
    reqs = [([rng.randrange(0, 23) for _ in range(rng.randrange(1, 14))], rng.randrange(1, 30)) for _ in range(n_req)]
    nblk = rng.choice((9, 12, 40))  # 9 / 12: some requests wait or get preempted; 40: ample
    outs = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=nblk, max_num_batched_tokens=32, max_num_seqs=4)
        for p, n in reqs:
            if len(p) + n < 8 * 4 - 1:  # must fit the allocator even alone
                eng.add_request(p, SamplingParams(max_tokens=n, ignore_eos=False, temperature=1.0))
        done = _drain(eng)
        bm = eng.scheduler.block_manager
        assert not bm.used_block_ids and eng._inflight is None
        outs.append((done, eng.model_runner.steps, list(bm.free_block_ids), dict(bm.hash_to_block_id)))
    (d0, s0, f0, h0), (d1, s1, f1, h1) = outs
    assert d0 == d1
    if eos < 0:
        assert s0 == s1 and f0 == f1 and h0 == h1


# ----------------------------------------------------------------------------- lookahead prefill
def test_prefill_lookahead_rules():
    """Scheduler.lookahead_prefill admits the next prefill step behind the one in flight only when that is provably the
    admission schedule() would make after the step in flight was postprocessed."""
    def waiting_engine(n_wait, nblk=200, budget=64, max_num_seqs=8, first=(20, 20, 20)):
        Sequence.counter = __import__("itertools").count()
        s = sched(num_kvcache_blocks=nblk, max_num_batched_tokens=budget, max_num_seqs=max_num_seqs)
        for k, n in enumerate(first):
            s.add(Sequence([1000 * (k + 1) + j for j in range(n)],
                           SamplingParams(max_tokens=4, ignore_eos=True, temperature=1.0), block_size=4))
        for i in range(n_wait):
            s.add(Sequence([100 * (i + 1) + j for j in range(20)],
                           SamplingParams(max_tokens=4, ignore_eos=True, temperature=1.0), block_size=4))
        flying, is_prefill = s.schedule()
        assert is_prefill and len(flying) == 3
        return s, flying

    s, flying = waiting_engine(0)
    assert s.lookahead_prefill(flying, 0) is None                      # nothing waits
    s, flying = waiting_engine(5)
    assert s.lookahead_prefill(flying, 61) is None                     # the step in flight (60 tokens) is too short
    assert len(s.waiting) == 5 and len(s.running) == 3                 # ... and nothing was touched
    s, flying = waiting_engine(2)
    assert s.lookahead_prefill(flying, 0) is None                      # 40 of 64 tokens: a late request could still join
    s, flying = waiting_engine(5, nblk=3 * 5 + 4 * 5 + 1)              # one block short of covering all five candidates
    assert s.lookahead_prefill(flying, 0) is None and len(s.waiting) == 5
    s, flying = waiting_engine(5, nblk=3 * 5 + 5 * 5 + 1)
    nxt = s.lookahead_prefill(flying, 0)                               # closed by the budget (3 x 20 <= 64 < 4 x 20)
    assert [q.seq_id for q in nxt] == [3, 4, 5] and len(s.waiting) == 2
    assert [q.seq_id for q in s.running] == [0, 1, 2, 3, 4, 5]
    s, flying = waiting_engine(3, max_num_seqs=3, budget=1000)
    assert len(s.lookahead_prefill(flying, 0)) == 3                    # closed by the sequence count
    s, flying = waiting_engine(3, budget=60)
    assert len(s.lookahead_prefill(flying, 0)) == 3                    # exactly at the budget: nobody else fits


@pytest.mark.parametrize("eos", [-1, 5])
def test_prefill_lookahead_chain_reproduces_the_synchronous_engine(eos):
    """Many prompts behind a small token budget: the prefill steps are queued behind one another.  Same prefill steps,
    same token streams and the same allocator state as the plain loop - with requests that end at their first token
    (by length, and by EOS when it is on) between two queued steps."""
    rng = np.random.default_rng(11)
    reqs = [([int(t) for t in rng.integers(0, 23, n)], m) for n, m in
            ((14, 5), (9, 1), (11, 7), (13, 3), (12, 1), (10, 9), (15, 2), (8, 6), (9, 4), (14, 1), (7, 5), (12, 8))]
    outs = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=80, max_num_batched_tokens=40, max_num_seqs=6)
        for p, m in reqs:
            eng.add_request(p, SamplingParams(max_tokens=m, ignore_eos=False, temperature=1.0))
        done = _drain(eng)
        bm = eng.scheduler.block_manager
        assert not bm.used_block_ids and eng._inflight is None and eng._inflight_prefill is None
        outs.append((done, eng.model_runner.prefills, eng.model_runner.steps, list(bm.free_block_ids),
                     dict(bm.hash_to_block_id), eng.prefill_lookahead_launches))
    sync, look = outs
    assert sync[5] == 0 and look[5] >= 2
    assert sync[0] == look[0] and sync[1] == look[1]
    assert sync[3] == look[3] and sync[4] == look[4]
    if eos < 0:
        assert sync[2] == look[2]
    else:
        assert any(len(sync[0][i]) < m for i, (_, m) in enumerate(reqs))  # the scripted model does hit EOS


def test_prefill_lookahead_abort_drops_the_rows_of_the_queued_step():
    """abort_request while the aborted request's prompt is in the QUEUED prefill step: its first token is discarded
    when that step is collected, its blocks are free, the other streams are untouched."""
    outs = []
    for abort in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=60, max_num_batched_tokens=24, max_num_seqs=4)
        for i in range(6):
            eng.add_request([i + 1] * 11, SamplingParams(max_tokens=6, ignore_eos=True, temperature=1.0),
                            request_id=f"r{i}")
        done = {}
        for step in range(60):
            if eng.is_finished():
                break
            if step == 1:
                assert eng._inflight_prefill is not None  # requests 2 and 3 are in the queued step
                assert [s.request_id for s in eng._inflight_prefill[1]] == ["r2", "r3"]
                if abort:
                    eng.abort_request("r3")
            for seq_id, toks, _, _ in eng.step()[0]:
                done[seq_id] = list(toks)
        outs.append(done)
        assert not eng.scheduler.block_manager.used_block_ids and eng._inflight_prefill is None
    full, aborted = outs
    assert sorted(full) == list(range(6)) and sorted(aborted) == [0, 1, 2, 4, 5]
    assert all(aborted[i] == full[i] for i in aborted)


def test_add_request_refuses_prompts_longer_than_max_model_len():
    """max_model_len bounds the RoPE table and the static block-table width: a longer prompt fails at add_request
    (the reference indexes its rotary cache out of range instead); a prompt of exactly max_model_len tokens is served."""
    eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=40, max_num_batched_tokens=64, max_model_len=24)
    sp = SamplingParams(max_tokens=3, ignore_eos=True, temperature=1.0)
    with pytest.raises(ValueError, match="max_model_len"):
        eng.add_request(list(range(25)), sp)
    assert eng.is_finished()
    eng.add_request(list(range(24)), sp)
    done = _drain(eng)
    assert len(done) == 1 and len(next(iter(done.values()))) == 1  # ends by length with its first token


@pytest.mark.parametrize("seed", range(16))
def test_prefill_lookahead_fuzz_with_arrivals_and_aborts(seed):
    """Seeded request streams that keep several prefill steps' worth of prompts waiting (small token budget), with
    requests arriving and being aborted between steps, EOS on in half of the seeds, memory tight in a third: every
    request's token stream equals the synchronous engine's (the scripted model's tokens depend on a request's own
    history only), nothing is left allocated, and without arrivals / aborts / EOS the prefill steps themselves and the
    final allocator state are equal as well."""
    import random

    rng = random.Random(500 + seed)
    eos = 5 if seed % 2 else -1
    quiet = seed % 4 == 0  # no arrivals, no aborts: step-for-step comparable
    budget = rng.choice((24, 40, 64))
    nblk = rng.choice((14, 30, 120))
    n_req = rng.randrange(6, 16)
    reqs = [([rng.randrange(0, 23) for _ in range(rng.randrange(2, min(budget, 20)))], rng.randrange(1, 12))
            for _ in range(n_req)]
    arrive = [0 if quiet or rng.random() < 0.6 else rng.randrange(1, 12) for _ in reqs]
    aborts = {} if quiet else {rng.randrange(1, 10): f"r{rng.randrange(n_req)}" for _ in range(rng.randrange(0, 3))}
    outs = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=nblk, max_num_batched_tokens=budget, max_num_seqs=5,
                               max_model_len=48)
        pending = sorted(((a, i) for i, a in enumerate(arrive)), key=lambda t: (t[0], t[1]))
        ids, done, step = {}, {}, 0
        while pending or not eng.is_finished():
            while pending and pending[0][0] <= step:
                _, i = pending.pop(0)
                p, m = reqs[i]
                ids[i] = eng.add_request(p, SamplingParams(max_tokens=m, ignore_eos=False, temperature=1.0),
                                         request_id=f"r{i}").seq_id
            if step in aborts:
                eng.abort_request(aborts[step])
            if not eng.is_finished():
                for seq_id, toks, _, _ in eng.step()[0]:
                    done[seq_id] = list(toks)
            step += 1
            assert step < 4000
        bm = eng.scheduler.block_manager
        assert not bm.used_block_ids and eng._inflight is None and eng._inflight_prefill is None
        outs.append(({i: done.get(s) for i, s in ids.items()}, eng.model_runner.prefills, list(bm.free_block_ids),
                     eng.prefill_lookahead_launches))
    (d0, p0, f0, n0), (d1, p1, f1, n1) = outs
    assert n0 == 0
    never_aborted = [i for i in d0 if f"r{i}" not in aborts.values()]
    assert all(d0[i] == d1[i] and d0[i] is not None for i in never_aborted), (d0, d1)
    if quiet and eos < 0:
        assert p0 == p1 and f0 == f1
    _FUZZ_PREFILL_AHEAD.append(n1)


_FUZZ_PREFILL_AHEAD: list = []


def test_prefill_lookahead_fuzz_did_queue_steps_ahead():
    """(runs after the parametrised fuzz above) the fuzz exercised the lookahead, not only its refusals"""
    assert sum(_FUZZ_PREFILL_AHEAD) >= 10, _FUZZ_PREFILL_AHEAD


# ----------------------------------------------------------------------------- garbage collector placement
def _run_with_gc_pressure(gc_control):
    """The scripted engine under a collector that is eager to run full passes (thresholds 20 / 1 / 1: every other
    young collection escalates to generation 2), with the engine's policy on or off.  Returns HostGc.summary()."""
    import gc

    old = gc.get_threshold()
    Sequence.counter = __import__("itertools").count()
    eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=80, max_num_batched_tokens=48, max_num_seqs=4,
                           gc_control=gc_control)
    eng.gc.young_every, eng.gc.mid_every = 4, 16
    eng.gc.watch()
    eng.gc.settle()
    # CPython escalates to a full pass only when the objects pending since the last one exceed a quarter of what that one
    # left in generation 2: make that generation small (freeze everything, then one full pass over nothing), for both runs
    gc.freeze()
    gc.collect()
    gc.set_threshold(20, 1, 1)
    try:
        for i in range(8):
            eng.add_request([i + 1] * 13, SamplingParams(max_tokens=40, ignore_eos=True, temperature=1.0))
        done = _drain(eng)
    finally:
        gc.set_threshold(*old)
        summary, stats = eng.gc.summary(), dict(eng.gc.stats)
        eng.gc.release()
        gc.unfreeze()
    assert len(done) == 8
    return summary, stats


def test_no_full_collection_runs_inside_a_step():
    """a generation-2 pass costs ~100 ms in a process that holds torch + a model, a decode step 1.4 ms.
    With Config.gc_control the automatic collector is off inside step(); young collections run at the step's slack point,
    full ones only when the engine has drained.  The control run (policy off, same pressure) does see full passes
    inside steps - so the assertion is not vacuous."""
    import gc

    assert gc.isenabled()
    control, _ = _run_with_gc_pressure(False)
    assert control["full_in_step"] > 0, control
    guarded, stats = _run_with_gc_pressure(True)
    assert guarded["full_in_step"] == 0, guarded
    assert stats["young"] > 0 and stats["mid"] > 0 and stats["full_idle"] >= 1, stats
    assert stats["frozen_objects"] > 0
    assert gc.isenabled() and gc.get_freeze_count() == 0  # release() gave the collector back


def test_a_prefill_step_that_already_finished_is_stamped_before_the_next_one_is_launched():
    """_step_prefill: when the queued step's tokens are already on the host (prefill_done), its first tokens are stamped
    and postprocessed at once; the next prefill step is NOT queued first (its launch sequence costs the host milliseconds
    that would land in those requests' TTFT) but scheduled by the following step() call."""
    Sequence.counter = __import__("itertools").count()
    eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=60, max_num_batched_tokens=24, max_num_seqs=4)
    order = []
    runner = eng.model_runner
    launch = runner.launch_prefill
    runner.launch_prefill = lambda seqs: (order.append(("launch", [s.seq_id for s in seqs])), launch(seqs))[1]
    runner.prefill_done = lambda handle: True
    stamp = eng._stamp_first_tokens
    eng._stamp_first_tokens = lambda seqs: (order.append(("stamp", [s.seq_id for s in seqs])), stamp(seqs))[1]
    for i in range(4):
        eng.add_request([i + 1] * 11, SamplingParams(max_tokens=3, ignore_eos=True, temperature=1.0))
    eng.step()
    eng.step()
    assert order == [("launch", [0, 1]), ("stamp", [0, 1]), ("launch", [2, 3]), ("stamp", [2, 3])]
    assert eng.prefill_lookahead_launches == 0 and eng._inflight_prefill is None
    assert [r["queued_behind_previous"] for r in eng.prefill_trace] == [False, False]
    assert all(r["stamp"] >= r["launch_end"] >= r["launch_start"] for r in eng.prefill_trace)
    # ... and with the device still busy the next step IS queued first (the other tests of this section)
    runner.prefill_done = lambda handle: False
    order.clear()
    for i in range(6):  # (six: the queued step must be closed by the token budget, not by the end of the queue)
        eng.add_request([i + 9] * 11, SamplingParams(max_tokens=3, ignore_eos=True, temperature=1.0))
    while eng.scheduler.waiting:
        eng.step()
    assert [o[0] for o in order[:3]] == ["launch", "launch", "stamp"]


@pytest.mark.parametrize("eos", [-1, 5])
def test_first_decode_step_is_queued_behind_the_last_prefill_step(eos):
    """Round 5: when the last prefill step of a burst is in flight and nothing is waiting, the first decode step is
    scheduled from lengths alone and queued behind it (LLMEngine._queue_decode_behind_prefill) - the prefill step's rows
    read their input ids from its token buffer on the device.  Against the synchronous engine on the scripted model:
    the same decode steps (sequences, positions, input ids, block tables), token streams, TTFT stamps for every
    request, allocator state; a request whose FIRST token is its last (max_tokens 1, or EOS) never gets a decode row
    that counts."""
    rng = np.random.default_rng(11)
    prompts = [[int(t) for t in rng.integers(0, 23, n)] for n in (5, 8, 12, 3, 9, 16)]
    lens = (6, 1, 9, 4, 1, 7)
    outs, engines = [], []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=60, max_num_batched_tokens=32, max_num_seqs=8)
        for p, n in zip(prompts, lens):
            eng.add_request(p, SamplingParams(max_tokens=n, ignore_eos=False, temperature=1.0))
        outs.append(_drain(eng))
        engines.append(eng)
    sync, look = outs
    assert sync == look and sorted(look) == list(range(6))
    e_sync, e_look = engines
    assert getattr(e_sync, "decode_behind_prefill_launches", 0) == 0
    assert getattr(e_look, "decode_behind_prefill_launches", 0) == 1  # one burst: its last prefill step
    assert sorted(e_look.ttft) == sorted(e_sync.ttft) == list(range(6))
    if eos < 0:
        assert e_sync.model_runner.steps == e_look.model_runner.steps
        assert all(len(look[i]) == lens[i] for i in range(6))
    assert e_sync.model_runner.prefills == e_look.model_runner.prefills
    a, b = e_sync.scheduler.block_manager, e_look.scheduler.block_manager
    assert not a.used_block_ids and not b.used_block_ids
    if eos < 0:  # (an EOS ending is known one step later under lookahead: the freed blocks return in another order)
        assert list(a.free_block_ids) == list(b.free_block_ids)


_FUZZ_PREFILL_BEHIND_DECODE: list[int] = []


@pytest.mark.parametrize("seed", range(12))
def test_arrivals_are_prefilled_behind_the_running_decode_step(seed):
    """Round 5: a request that arrives while a decode step is queued is admitted at once and its prefill step - a graph
    replay on the device runner (`prefill_graph_takes`) - queued BEHIND that decode step, before the step's tokens have
    come back (Scheduler.lookahead_prefill_behind_decode).  Seeded streams of arrivals into a decoding engine, EOS on
    in half of the seeds, aborts in some, against the synchronous engine: every never-aborted request's token stream is
    equal, every request is prefilled exactly once per admission, nothing stays allocated."""
    import random

    rng = random.Random(900 + seed)
    eos = 5 if seed % 2 else -1
    n_req = rng.randrange(6, 14)
    reqs = [([rng.randrange(0, 23) for _ in range(rng.randrange(2, 20))], rng.randrange(2, 14)) for _ in range(n_req)]
    arrive = [0 if i < 2 else rng.randrange(1, 25) for i in range(n_req)]
    aborts = {rng.randrange(2, 20): f"r{rng.randrange(n_req)}" for _ in range(rng.randrange(0, 2))} if seed % 3 == 0 else {}
    outs = []
    for look in (False, True):
        Sequence.counter = __import__("itertools").count()
        eng = _scripted_engine(look, eos=eos, num_kvcache_blocks=120, max_num_batched_tokens=64, max_num_seqs=6,
                               max_model_len=48)
        if look:
            eng.model_runner.prefill_graph_takes = lambda n, t: n <= 2 and t <= 40
        pending = sorted(((a, i) for i, a in enumerate(arrive)), key=lambda t: (t[0], t[1]))
        ids, done, step = {}, {}, 0
        while pending or not eng.is_finished():
            while pending and pending[0][0] <= step:
                _, i = pending.pop(0)
                p, m = reqs[i]
                ids[i] = eng.add_request(p, SamplingParams(max_tokens=m, ignore_eos=False, temperature=1.0),
                                         request_id=f"r{i}").seq_id
            if step in aborts:
                eng.abort_request(aborts[step])
            if not eng.is_finished():
                for seq_id, toks, _, _ in eng.step()[0]:
                    assert seq_id not in done
                    done[seq_id] = list(toks)
            step += 1
            assert step < 4000
        bm = eng.scheduler.block_manager
        assert not bm.used_block_ids and eng._inflight is None and eng._inflight_prefill is None
        flat = [s for step_ids in eng.model_runner.prefills for s in step_ids]
        assert len(flat) == len(set(flat))  # (no preemption here: every request is prefilled once)
        outs.append(({i: done.get(s) for i, s in ids.items()}, getattr(eng, "prefill_behind_decode_launches", 0)))
    (d0, n0), (d1, n1) = outs
    assert n0 == 0
    never_aborted = [i for i in d0 if f"r{i}" not in aborts.values()]
    assert all(d0[i] == d1[i] and d0[i] is not None for i in never_aborted), (d0, d1)
    _FUZZ_PREFILL_BEHIND_DECODE.append(n1)


def test_some_arrivals_were_prefilled_behind_a_decode_step():
    """(runs after the fuzz above) the new admission path was taken, not only declined"""
    assert _FUZZ_PREFILL_BEHIND_DECODE and sum(_FUZZ_PREFILL_BEHIND_DECODE) >= 6, _FUZZ_PREFILL_BEHIND_DECODE


def test_generate_checks_every_prompt_before_queueing_any():
    """generate() with one over-long prompt in the batch used to raise after the earlier prompts were
    already queued - the next generate() call then ran and reported those orphans."""
    eng = _scripted_engine(True, eos=-1, num_kvcache_blocks=40, max_num_batched_tokens=64, max_model_len=24)
    sp = SamplingParams(max_tokens=2, ignore_eos=True, temperature=1.0)
    with pytest.raises(ValueError, match="max_model_len"):
        eng.generate([[1, 2, 3], list(range(30)), [4, 5]], sp, use_tqdm=False)
    assert eng.is_finished() and not eng.scheduler.waiting
    out = eng.generate([[1, 2, 3], [4, 5]], sp, use_tqdm=False)
    assert [o["prompt_len"] for o in out] == [3, 2]


def test_lazily_captured_prefill_graphs_are_keyed_by_buckets_that_pad_by_at_most_a_sixteenth():
    """ModelRunner._lazy_prefill_key (round 6: full-house prefill steps replay graphs): tokens round up to 256, sequences
    and the longest query to powers of two (queries from 256); a step that would be padded by more than 1/16 of its
    tokens stays eager - and the bench's full house of 16 x 1024 tokens is its own bucket."""
    from nanovllm.engine.model_runner import ModelRunner

    key = ModelRunner._lazy_prefill_key
    assert key(16384, 16, 1024) == (16384, 16, 1024)
    assert key(16000, 13, 1000) == (16128, 16, 1024)
    assert key(4100, 5, 900) == (4352, 8, 1024)
    assert key(2049, 2, 2049) is None            # 255 pad tokens on 2049: more than a sixteenth
    assert key(300, 1, 300) is None
    assert key(256, 1, 17) == (256, 1, 256)
    for tokens in range(1, 20000, 37):
        k = key(tokens, 1 + tokens % 40, 1 + tokens % 1500)
        if k is not None:
            tb, sb, mq = k
            assert tb >= tokens and tb % 256 == 0 and (tb - tokens) * 16 <= tokens
            assert sb >= 1 + tokens % 40 and sb & (sb - 1) == 0
            assert mq >= max(256, 1 + tokens % 1500) and mq & (mq - 1) == 0


def test_exchange_region_cap_for_eager_launches_and_the_scope_that_lifts_it():
    """XgmiComm.fits / fits_rows (round 6): an eager launch may put decode-sized rows through the exchange region only,
    however large the region is (it is also sized for captured prefill steps); `large()` lifts the cap for the capture of
    such a step, nests, and restores it when the capture raises."""
    from nanovllm.layers.xgmi_comm import XgmiComm

    comm = object.__new__(XgmiComm)  # (no device, no process group: the arithmetic only)
    comm.max_bytes, comm.eager_max_bytes, comm._large = 1024 * 2048 * 2, 64 * 2048 * 2, 0
    assert comm.fits_rows(64, 2048) and not comm.fits_rows(65, 2048) and not comm.fits_rows(112, 2048)
    with comm.large():
        assert comm.fits_rows(112, 2048) and comm.fits_rows(512, 2048) and not comm.fits_rows(513, 2048)
        with comm.large():
            assert comm.fits_rows(512, 2048)
        assert comm.fits_rows(512, 2048)
    assert not comm.fits_rows(112, 2048)
    with pytest.raises(RuntimeError):
        with comm.large():
            raise RuntimeError("capture failed")
    assert comm._large == 0 and not comm.fits_rows(112, 2048)
    comm.eager_max_bytes = comm.max_bytes  # (MI355_XGMI_EAGER_LARGE=1: the bring-up switch)
    assert comm.fits_rows(512, 2048)
