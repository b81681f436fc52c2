"""Synthetic HF model directories (config.json only) for tests, smoke and bench."""
import json
import os
import tempfile

QWEN3_0_6B = dict(
    architectures=["Qwen3ForCausalLM"], model_type="qwen3", hidden_size=1024, num_hidden_layers=28,
    num_attention_heads=16, num_key_value_heads=8, head_dim=128, intermediate_size=3072, vocab_size=151936,
    max_position_embeddings=40960, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True,
    attention_bias=False, hidden_act="silu", torch_dtype="bfloat16", bos_token_id=151643, eos_token_id=151645,
)

TINY = dict(QWEN3_0_6B, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
            intermediate_size=256, vocab_size=256, max_position_embeddings=512, eos_token_id=255, bos_token_id=0)

# the reference's Llama wiring (models/llama.py) on the tiny shapes: no q/k norm, no biases, rope_theta 1e4
TINY_LLAMA = dict(TINY, architectures=["LlamaForCausalLM"], model_type="llama", rope_theta=10000.0, mlp_bias=False)

# the reference's Qwen3-MoE wiring (models/qwen3_moe.py): 8 experts, top-2, every layer sparse
TINY_MOE = dict(TINY, architectures=["Qwen3MoeForCausalLM"], model_type="qwen3_moe", num_experts=8,
                num_experts_per_tok=2, moe_intermediate_size=64, decoder_sparse_step=1, mlp_only_layers=[],
                norm_topk_prob=True)

# head geometries of the plain-layout attention kernels (csrc/attn_plain.hip): Qwen2-0.5B's (head_dim 64, 7 query
# heads per kv head, qkv bias, no q/k norm) and Llama-3.2-1B's (head_dim 64, 4 per kv head) on tiny widths
TINY_QWEN2_HD64 = dict(TINY, architectures=["Qwen2ForCausalLM"], model_type="qwen2", hidden_size=128, head_dim=64,
                       num_attention_heads=7, num_key_value_heads=1, attention_bias=True)
TINY_LLAMA_HD64 = dict(TINY_LLAMA, hidden_size=256, head_dim=64, num_attention_heads=4, num_key_value_heads=1)
# a head geometry that still belongs to the plain-layout family after round 4 (three query heads per kv head)
TINY_LLAMA_HD64_G3 = dict(TINY_LLAMA, hidden_size=384, head_dim=64, num_attention_heads=6, num_key_value_heads=2)

# the two small models the reference's README benchmarks next to Qwen3-0.6B (README.md:316-318), full shapes
QWEN2_0_5B = dict(
    architectures=["Qwen2ForCausalLM"], model_type="qwen2", hidden_size=896, num_hidden_layers=24,
    num_attention_heads=14, num_key_value_heads=2, intermediate_size=4864, vocab_size=151936,
    max_position_embeddings=32768, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True,
    hidden_act="silu", torch_dtype="bfloat16", bos_token_id=151643, eos_token_id=151645,
)
LLAMA_3_2_1B = dict(
    architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=2048, num_hidden_layers=16,
    num_attention_heads=32, num_key_value_heads=8, head_dim=64, intermediate_size=8192, vocab_size=128256,
    max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=True,
    hidden_act="silu", torch_dtype="bfloat16", bos_token_id=128000, eos_token_id=128001, attention_bias=False, mlp_bias=False,
)

MID = dict(QWEN3_0_6B, num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096, eos_token_id=4095,
           bos_token_id=0)

# a Llama-wired model with the plain-layout head geometry (head_dim 64, 8 query / 2 kv heads) that two TP ranks can
# share: one kv head per rank
MID_LLAMA_HD64 = dict(MID, architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=512, head_dim=64,
                      num_attention_heads=8, num_key_value_heads=2, intermediate_size=1024, rope_theta=10000.0,
                      attention_bias=False, mlp_bias=False)

# BASELINE.json configs[2]: Qwen3-32B widths (hidden 5120, 64 q / 8 kv heads, intermediate 25600), cut to
# 2 layers and a 4096 vocabulary so that the CPU oracle finishes in seconds
QWEN3_32B_2L = dict(QWEN3_0_6B, hidden_size=5120, num_hidden_layers=2, num_attention_heads=64,
                    num_key_value_heads=8, intermediate_size=25600, vocab_size=4096, max_position_embeddings=4096,
                    tie_word_embeddings=False, eos_token_id=4095, bos_token_id=0)


def make_model_dir(cfg: dict, root: str | None = None) -> str:
    d = tempfile.mkdtemp(prefix="mi355_model_", dir=root)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    return d
