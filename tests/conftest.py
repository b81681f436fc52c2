"""pytest configuration: paths, the `gpu` marker, shared fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "nano-vllm-ascend_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: a long variant of a GPU test whose faster sibling stays in `-m gpu`; "
                                       "run with -m 'gpu or gpu_slow' (tools/final_validation.sh does)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` is the driver's round-end run with a time limit: the long variants (gpu_slow: the eager twin of a
    hipGraph TP run, the 100-token twin of a 32-token MoE case ...) are skipped there - visibly, with this reason -
    and run whenever the marker expression names them."""
    wants_slow = "gpu_slow" in (config.getoption("-m") or "")
    slow = pytest.mark.skip(reason="gpu_slow variant: run with -m 'gpu or gpu_slow'")
    nogpu = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu_slow" in item.keywords and not wants_slow:
            item.add_marker(slow)
        elif ("gpu" in item.keywords or "gpu_slow" in item.keywords) and not torch.cuda.is_available():
            item.add_marker(nogpu)


def bf16_from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


@pytest.fixture(scope="session")
def golden_layers():
    return np.load(os.path.join(GOLDEN, "layers.npz"))


@pytest.fixture(scope="session")
def golden_attention():
    return np.load(os.path.join(GOLDEN, "attention.npz"))


@pytest.fixture(scope="session")
def golden_tiny():
    return np.load(os.path.join(GOLDEN, "tiny_model.npz"))


@pytest.fixture(scope="session")
def golden_tiny_bias():
    """The same run with attention_bias=True: qkv bias, no q/k norm (the Qwen2 wiring)."""
    return np.load(os.path.join(GOLDEN, "tiny_model_bias.npz"))


@pytest.fixture(scope="session")
def golden_tiny_llama():
    """The reference's LlamaForCausalLM (no q/k norm, no bias) on the tiny shapes."""
    return np.load(os.path.join(GOLDEN, "tiny_model_llama.npz"))


@pytest.fixture(scope="session")
def golden_tiny_moe():
    """The reference's Qwen3MoeForCausalLM (8 experts, top-2) on the tiny shapes."""
    return np.load(os.path.join(GOLDEN, "tiny_model_moe.npz"))


@pytest.fixture(scope="session")
def golden_tiny_hd64():
    """The reference's Qwen3ForCausalLM (qkv bias, no q/k norm: the Qwen2 wiring) and LlamaForCausalLM at head_dim 64
    with 7 / 4 query heads per kv head - Qwen2-0.5B's and Llama-3.2-1B's head geometry (csrc/attn_plain.hip)."""
    return {"qwen2_hd64": np.load(os.path.join(GOLDEN, "tiny_model_qwen2_hd64.npz")),
            "llama_hd64": np.load(os.path.join(GOLDEN, "tiny_model_llama_hd64.npz"))}


@pytest.fixture(scope="session")
def golden_moe_block():
    return np.load(os.path.join(GOLDEN, "moe_block.npz"))
