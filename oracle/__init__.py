"""CPU oracle for the paged-KV Qwen3 decode path — TEST INFRASTRUCTURE ONLY.

This package is a plain torch-on-CPU restatement of the arithmetic the reference
(linzm1007/nano-vllm-ascend) performs on the hot path.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``nano-vllm-ascend_amd/`` imports it,
and the product path raises if the HIP extension is missing.

Pinning status
--------------
* Elementwise layers (RMSNorm, add-RMSNorm, RoPE, SiluAndMul), the KV scatter,
  the xxh64 prefix hash and the scheduler / block-manager / prepare_* index
  traces are PINNED: ``tools/gen_golden.py`` imports the Python reference from
  /root/reference (with in-memory stubs for the absent torch_npu / torchair
  modules), runs it on seeded inputs and commits inputs + outputs under
  ``tests/golden/``; ``tests/test_oracle_golden.py`` requires bit-equality.
* Attention: the reference's device op (torch_npu FIA v2) is a closed binary
  that is not in the reference tree and cannot run here; the in-tree CPU
  statement of the same contract, ``layers/attention_torch_native.py``, keeps
  scores/probabilities in bf16, so the oracle (fp32 softmax) is pinned to it
  within a bf16-rounding bound, not bit-exactly (tests state the bound).
* End-to-end: a tiny random-weight Qwen3 driven through the reference's own
  model / scheduler / block-manager classes gives golden greedy tokens and
  logits (``tests/golden/tiny_model.npz``).

* fp8 weights (no reference counterpart): ``e4m3_encode/decode`` restate the published OCP E4M3
  format with integer arithmetic; pinned against torch's float8_e4m3fn on all 254 finite codes and
  all midpoints (tests/test_oracle_golden.py).

Rounding points follow the reference file:line cited on each function; where the
reference delegates to torch's bf16 kernels (F.linear) the oracle accumulates in
fp32 and rounds once, which is what those kernels do.
"""
from oracle.layers import (  # noqa: F401
    dequantize_fp8_rows,
    e4m3_decode,
    e4m3_encode,
    quantize_fp8_rows,
    add_rms_norm,
    apply_rope,
    build_cos_sin_cache,
    embedding,
    kv_scatter,
    linear,
    paged_attention_decode,
    paged_attention_prefill,
    rms_norm,
    silu_and_mul,
)
from oracle.model import OracleQwen3  # noqa: F401
