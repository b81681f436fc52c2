"""Pin the CPU oracle against golden vectors produced by the reference itself
(tools/gen_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, bf16_from_bits as bf
from oracle.model import OracleConfig, OracleQwen3


def _eq(a: torch.Tensor, b: torch.Tensor):
    assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (
        f"{(a.view(torch.int16) != b.view(torch.int16)).sum().item()} of {a.numel()} elements differ")


def test_rmsnorm_bit_exact(golden_layers):
    g = golden_layers
    for tag in ("h1024", "head128", "h256"):
        x, w, r = bf(g[f"rms_{tag}_x"]), bf(g[f"rms_{tag}_w"]), bf(g[f"rms_{tag}_r"])
        _eq(oracle.rms_norm(x, w, 1e-6), bf(g[f"rms_{tag}_y"]))
        y, r2 = oracle.add_rms_norm(x, r, w, 1e-6)
        _eq(y, bf(g[f"rms_{tag}_addy"]))
        _eq(r2, bf(g[f"rms_{tag}_addr"]))


def test_rope_bit_exact(golden_layers):
    g = golden_layers
    pos = torch.from_numpy(g["rope_pos"])
    table = oracle.build_cos_sin_cache(128, 40960, 1000000.0)
    assert torch.equal(table[pos], torch.from_numpy(g["rope_table_rows"]))
    _eq(oracle.apply_rope(pos, bf(g["rope_q"]), table), bf(g["rope_q_out"]))
    _eq(oracle.apply_rope(pos, bf(g["rope_k"]), table), bf(g["rope_k_out"]))


def test_silu_and_mul_bit_exact(golden_layers):
    g = golden_layers
    _eq(oracle.silu_and_mul(bf(g["silu_x"])), bf(g["silu_y"]))


def _caches(g, prefix):
    hq, hkv, d, bs, nblk = (int(v) for v in g["meta"])
    return hq, hkv, d, bs, nblk


def test_kv_scatter_bit_exact(golden_attention):
    g = golden_attention
    hq, hkv, d, bs, nblk = _caches(g, "pre")
    kc = torch.zeros(nblk, bs, hkv, d, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    slots = torch.from_numpy(g["pre_slots"])
    oracle.kv_scatter(bf(g["pre_k"]), bf(g["pre_v"]), kc, vc, slots)
    _eq(kc.view(-1, hkv, d)[slots.long()], bf(g["pre_krows"]))
    _eq(vc.view(-1, hkv, d)[slots.long()], bf(g["pre_vrows"]))
    assert int((kc.view(-1, hkv * d) != 0).any(dim=1).sum()) == int(g["pre_cache_nonzero_rows"])
    # decode-style scatter on a populated cache
    kc, vc = bf(g["dec_kcache_before"]).clone(), bf(g["dec_vcache_before"]).clone()
    kc0 = kc.clone()
    slots = torch.from_numpy(g["dec_slots"])
    oracle.kv_scatter(bf(g["dec_k"]), bf(g["dec_v"]), kc, vc, slots)
    _eq(kc.view(-1, hkv, d)[slots.long()], bf(g["dec_krows_after"]))
    _eq(vc.view(-1, hkv, d)[slots.long()], bf(g["dec_vrows_after"]))
    assert int((kc != kc0).view(-1, hkv * d).any(dim=1).sum()) == int(g["dec_cache_rows_changed"])


def test_attention_vs_reference_native(golden_attention):
    """The reference's CPU attention keeps S and P in bf16 with a bf16-rounded scale
    (attention_torch_native.py:80,139,189); the oracle is the exact fp32 softmax.
    Bound: a few bf16 ulps of the output magnitude (|out| <= ~1 here)."""
    g = golden_attention
    hq, hkv, d, bs, nblk = _caches(g, "")
    # prefill
    kc = torch.zeros(nblk, bs, hkv, d, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    oracle.kv_scatter(bf(g["pre_k"]), bf(g["pre_v"]), kc, vc, torch.from_numpy(g["pre_slots"]))
    cu = torch.from_numpy(g["pre_cu"])
    o = oracle.paged_attention_prefill(bf(g["pre_q"]), kc, vc, torch.from_numpy(g["pre_tables"]), cu,
                                       cu[1:] - cu[:-1])
    ref = bf(g["pre_out"])
    err = (o.float() - ref.float()).abs().max().item()
    assert err <= 3e-2, err
    # decode
    kc, vc = bf(g["dec_kcache_before"]).clone(), bf(g["dec_vcache_before"]).clone()
    oracle.kv_scatter(bf(g["dec_k"]), bf(g["dec_v"]), kc, vc, torch.from_numpy(g["dec_slots"]))
    o = oracle.paged_attention_decode(bf(g["dec_q"]), kc, vc, torch.from_numpy(g["dec_tables"]),
                                      torch.from_numpy(g["dec_ctx"]))
    ref = bf(g["dec_out"])
    err = (o.float() - ref.float()).abs().max().item()
    assert err <= 3e-2, err
    # ctx == 1 attends exactly one token: output equals that V row, bit for bit
    v0 = bf(g["dec_v"])[0]  # [hkv, d]
    expect = v0.repeat_interleave(hq // hkv, dim=0).reshape(-1)
    _eq(o[0], expect)


def test_sampler_probabilities(golden_layers):
    g = golden_layers
    logits, temps = bf(g["samp_logits"]), torch.from_numpy(g["samp_temps"])
    p = torch.softmax(logits.float() / temps.unsqueeze(-1), dim=-1)
    assert torch.equal(p, torch.from_numpy(g["samp_probs"]))


def _tiny_oracle(g, llama=False, moe=False, variant=None):
    from transformers import LlamaConfig, Qwen3Config, Qwen3MoeConfig

    weights = {k[3:]: bf(g[k]) for k in g.files if k.startswith("w::")}
    from model_configs import TINY, TINY_LLAMA, TINY_LLAMA_HD64, TINY_MOE, TINY_QWEN2_HD64

    tiny = dict(TINY, attention_bias=True) if "attention_bias" in g.files and int(g["attention_bias"]) else TINY
    tiny = TINY_LLAMA if llama else (TINY_MOE if moe else tiny)
    if variant in ("qwen2_hd64", "llama_hd64"):
        tiny, llama = {"qwen2_hd64": TINY_QWEN2_HD64, "llama_hd64": TINY_LLAMA_HD64}[variant], variant == "llama_hd64"
    cls = LlamaConfig if llama else (Qwen3MoeConfig if moe else Qwen3Config)
    hf = cls(**{k: v for k, v in tiny.items() if k not in ("architectures", "model_type", "torch_dtype")})
    cfg = OracleConfig.from_hf(hf)
    block_size, nblk = (int(v) for v in g["meta"])
    return OracleQwen3(cfg, weights, nblk, block_size), block_size


def test_moe_block_matches_reference(golden_moe_block):
    """oracle.moe_block vs the reference's Qwen3MoeSparseMoeBlock (models/qwen3_moe.py:150-185): the selected
    experts and their probabilities exactly (sets per token; the order inside a token's top-k is irrelevant to
    the result), the block output to <= 1 bf16 ulp on a few elements (the experts' F.linear accumulates in a
    different order on the reference's CPU GEMM)."""
    from oracle import layers as L

    g = golden_moe_block
    for tag in ("small", "wide"):
        T, H, E, K, I = (int(v) for v in g[f"{tag}_meta"])
        x, gw = bf(g[f"{tag}_x"]), bf(g[f"{tag}_gate_w"])
        gu, dn = bf(g[f"{tag}_gate_up_w"]).view(E, 2 * I, H), bf(g[f"{tag}_down_w"]).view(E, H, I)
        y, w, ids = L.moe_block(x, gw, gu, dn, K, return_routing=True)
        ref_ids = torch.from_numpy(g[f"{tag}_topk_ids"])
        assert torch.equal(ids.sort(-1).values, ref_ids.sort(-1).values)
        # the reference's top-k probabilities, renormalised and rounded as qwen3_moe.py:158-161 does
        p = torch.from_numpy(g[f"{tag}_topk_prob"])
        ref_w = (p / p.sum(-1, keepdim=True)).bfloat16()
        assert torch.equal(torch.gather(ref_w, 1, ref_ids.argsort(-1)).view(torch.int16),
                           torch.gather(w, 1, ids.argsort(-1)).view(torch.int16))
        ref_y = bf(g[f"{tag}_y"])
        d = (y.float() - ref_y.float()).abs()
        ulp = torch.exp2(torch.floor(torch.log2(ref_y.float().abs().clamp_min(1e-3))) - 7)
        assert float((d / ulp).max()) <= 2.0 and float((d > 0).float().mean()) <= 0.05, (float((d / ulp).max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("variant", ["qwen3", "qkv_bias", "llama", "moe", "qwen2_hd64", "llama_hd64"])
def test_tiny_model_matches_reference_run(golden_tiny, golden_tiny_bias, golden_tiny_llama, golden_tiny_moe, golden_tiny_hd64,
                                          variant):
    """Replay the reference's own greedy run (its scheduler, block manager, prepare_*,
    model) through the oracle model with the same token stream.  The reference
    pipeline is bf16 end to end with bf16 S/P in attention, so logits agree to a
    bf16-ulp-scale bound, and greedy tokens agree wherever the reference's top-2
    margin exceeds that bound."""
    g = {"qwen3": golden_tiny, "qkv_bias": golden_tiny_bias, "llama": golden_tiny_llama, "moe": golden_tiny_moe,
         **golden_tiny_hd64}[variant]
    model, bs = _tiny_oracle(g, llama=variant == "llama", moe=variant == "moe", variant=variant)
    lens = g["prompt_lens"].tolist()
    flat = g["prompts"].tolist()
    prompts, o = [], 0
    for n in lens:
        prompts.append(flat[o:o + n])
        o += n
    # the reference scheduler with ample blocks hands out block ids FIFO: 0.. per sequence
    tables, nxt = [], 0
    for n in lens:
        nb = (n + bs - 1) // bs
        tables.append(list(range(nxt, nxt + nb)))
        nxt += nb
    toks = [list(p) for p in prompts]
    worst = 0.0
    for step in range(int(g["n_steps"])):
        ref_logits = bf(g[f"s{step}_logits"]).float()
        ref_tokens = g[f"s{step}_tokens"].tolist()
        seqs = g[f"s{step}_seqs"].tolist()
        if int(g[f"s{step}_prefill"]):
            ids = torch.tensor(sum((toks[s] for s in seqs), []), dtype=torch.int64)
            pos = torch.tensor(sum((list(range(len(toks[s]))) for s in seqs), []), dtype=torch.int64)
            cu = torch.tensor([0] + list(np.cumsum([len(toks[s]) for s in seqs])), dtype=torch.int32)
            slots = torch.tensor(sum(([tables[s][i // bs] * bs + i % bs for i in range(len(toks[s]))]
                                      for s in seqs), []), dtype=torch.int32)
            width = max(len(tables[s]) for s in seqs)
            bt = torch.tensor([tables[s] + [-1] * (width - len(tables[s])) for s in seqs], dtype=torch.int32)
            logits = model.prefill(ids, pos, cu, slots, bt, fp32_logits=True)
        else:
            for s in seqs:  # may_append: new block when the new token starts one (block_manager.py:102-109)
                if len(toks[s]) % bs == 1 and len(tables[s]) * bs < len(toks[s]):
                    tables[s].append(nxt)
                    nxt += 1
            ids = torch.tensor([toks[s][-1] for s in seqs], dtype=torch.int64)
            pos = torch.tensor([len(toks[s]) - 1 for s in seqs], dtype=torch.int64)
            ctx = torch.tensor([len(toks[s]) for s in seqs], dtype=torch.int32)
            slot2d = torch.tensor([[tables[s][-1], (len(toks[s]) - 1) % bs] for s in seqs], dtype=torch.int32)
            width = max(len(tables[s]) for s in seqs)
            bt = torch.tensor([tables[s] + [-1] * (width - len(tables[s])) for s in seqs], dtype=torch.int32)
            logits = model.decode(ids, pos, slot2d, ctx, bt, fp32_logits=True)
        err = (logits - ref_logits).abs().max().item()
        worst = max(worst, err)
        # token agreement where the reference's decision margin is larger than the error bound
        top2 = ref_logits.topk(2, dim=-1).values
        margin = (top2[:, 0] - top2[:, 1])
        mine = logits.argmax(dim=-1).tolist()
        for i, s in enumerate(seqs):
            if margin[i] > 0.25:  # 4x the logit bound: both candidates may move
                assert mine[i] == ref_tokens[i], (step, i)
            toks[s].append(ref_tokens[i])  # follow the reference's token stream
    # the reference's bf16 pipeline (bf16 scores and probabilities) against fp32 internals: ~2 bf16 ulps of the largest
    # logits (4.6 on the 128-wide models: ulp 2^-5); the 256-wide llama_hd64 reaches logits of 5.5 over twice as long sums
    assert worst <= (1e-1 if variant == "llama_hd64" else 6e-2), worst
    assert sum(toks, []) == g["final_tokens"].tolist()


def test_hash_kats_against_xxhash_package():
    """block_manager.py:38-44 calls the `xxhash` pip package; pin the KATs the
    reference produced against the package installed here."""
    import xxhash

    with open(os.path.join(GOLDEN, "hash_kats.json")) as f:
        kats = json.load(f)
    for k in kats:
        data = np.array(k["tokens"]).tobytes()
        assert xxhash.xxh64(data).intdigest() == k["hash"]
        nxt = np.array(k.get("chained_tokens", k["tokens"])).tobytes()
        h = xxhash.xxh64()
        h.update(k["chained_with_prefix"].to_bytes(8, "little"))
        h.update(nxt)
        assert h.intdigest() == k["chained"]


def test_fp8_e4m3_restatement_and_product_quantiser_agree():
    """oracle.e4m3_* restates OCP E4M3 with integer arithmetic; the product's quantiser
    (nanovllm.ops.quantize_fp8, torch.float8_e4m3fn + ldexp) must produce the same bytes and scales,
    including ties, subnormals, zeros and the format's extremes."""
    from nanovllm import ops as product_ops

    from oracle import layers as L

    # every finite code decodes to what torch says, and re-encodes to itself
    codes = torch.tensor([c for c in range(256) if (c & 0x7F) != 0x7F], dtype=torch.uint8)
    vals = L.e4m3_decode(codes)
    assert torch.equal(vals, codes.view(torch.float8_e4m3fn).float())
    assert torch.equal(L.e4m3_encode(vals), codes)
    assert float(vals.abs().max()) == 448.0 and float(vals[vals > 0].min()) == 2.0 ** -9
    # exact midpoints between neighbours round to the even mantissa
    pos = vals[(vals > 0)].sort().values
    mids = (pos[:-1] + pos[1:]) / 2
    enc = L.e4m3_encode(mids)
    assert torch.equal(enc, mids.to(torch.float8_e4m3fn).view(torch.uint8))
    assert bool(((enc & 1) == 0).all())
    # row-wise quantiser: random weights, a zero row, a row of one huge value, tiny values
    g = torch.Generator().manual_seed(8)
    w = (torch.randn(64, 256, generator=g) * 0.02).bfloat16()
    w[3] = 0
    w[5, 7] = 300.0
    w[9] *= 1e-6
    q_o, s_o = L.quantize_fp8_rows(w)
    q_p, s_p = product_ops.quantize_fp8(w)
    assert torch.equal(s_o, s_p) and torch.equal(q_o, q_p)
    assert torch.equal(L.dequantize_fp8_rows(q_o, s_o), product_ops.dequantize_fp8(q_p, s_p))
    rel = ((L.dequantize_fp8_rows(q_o, s_o) - w.float()).abs() / w.float().abs().clamp_min(1e-30))[w.float().abs() > 0]
    amax = w.float().abs().amax(dim=1, keepdim=True).expand_as(w)[w.float().abs() > 0]
    normal = w.float().abs()[w.float().abs() > 0] >= amax * 2.0 ** -13  # above the subnormal range of the row
    assert float(rel[normal].max()) <= 2.0 ** -4 + 1e-6  # half a unit in the last of 3 mantissa bits
