"""Engine configuration (reference: nanovllm/config.py:14-66) — same field names and
defaults, so `LLM(model_dir, **kwargs)` call sites carry over unchanged.

Differences, all forced by the platform:
  * `device` is "cuda" (PyTorch-ROCm's name for a HIP device), not "npu".
  * `graph_mode`: "hipgraph" captures decode steps into hipGraphs; the reference's
    torchair values "max-autotune" / "reduce-overhead" are accepted as aliases.
  * `hccl_port` keeps its name; it is the TCP rendezvous port of the RCCL group.
  * a model directory holding only `config.json` (no *.safetensors) selects
    synthetic random weights N(0, 0.02^2) seeded by `synthetic_seed`
    (there are no checkpoints on the GPU box).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from enum import Enum
from typing import Any


class GraphMode(Enum):
    EAGER = "eager"
    HIPGRAPH = "hipgraph"
    MAX_AUTOTUNE = "max-autotune"        # reference alias -> hipgraph
    REDUCE_OVERHEAD = "reduce-overhead"  # reference alias -> hipgraph


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 256
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.7
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: Any = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    use_graph_cache: bool = False  # accepted, unused: hipGraph capture takes milliseconds
    hccl_port: int = 28000
    graph_mode: str = GraphMode.HIPGRAPH.value
    is_multimodal: bool = False
    device: str = "cuda"
    trust_remote_code: bool = False
    synthetic_seed: int = 0
    decode_lookahead: bool = True  # one GPU: queue decode step k+1 before step k's tokens reach the host
    sampling_seed: int | None = None  # seed of the temperature sampler; None: drawn from os.urandom per engine
    quantization: str | None = None  # "fp8": e4m3 weights + per-row scales for the decode GEMMs (no reference counterpart)
    prefix_aware_prefill: bool = True  # skip the tokens of cache-hit prefix blocks in prefill (False: recompute, as the reference)
    prefill_graphs: bool = True  # prefill steps of up to 4 sequences / 4096 tokens replay bucketed hipGraphs (engine/model_runner.py)
    gc_control: bool = True  # engine/host_gc.py: freeze after warm-up, no automatic collection inside step(), full passes only when idle
    # a prefill step at least this many tokens long has the next one queued behind it (engine/llm_engine.py); < 0: derived
    # after warm-up from this host's launch time and this device's time per token
    prefill_lookahead_min_tokens: int = -1

    def __post_init__(self):
        assert os.path.isdir(self.model), f"model must be a directory with a HF config.json: {self.model}"
        assert self.kvcache_block_size % 16 == 0
        assert 1 <= self.tensor_parallel_size <= 8
        assert self.graph_mode in {m.value for m in GraphMode}, self.graph_mode
        assert self.quantization in (None, "fp8"), self.quantization
        if self.sampling_seed is None:
            self.sampling_seed = int.from_bytes(os.urandom(8), "little") >> 1
        if self.hf_config is None:
            from transformers import AutoConfig

            self.hf_config = AutoConfig.from_pretrained(self.model, trust_remote_code=self.trust_remote_code)
        text = getattr(self.hf_config, "text_config", self.hf_config)
        max_pos = getattr(text, "max_position_embeddings", None)
        if max_pos is not None:
            self.max_model_len = min(self.max_model_len, max_pos)
        eos = getattr(text, "eos_token_id", None)
        if eos is not None:
            self.eos = eos[0] if isinstance(eos, (list, tuple)) else eos
        assert self.max_num_batched_tokens >= self.max_model_len

    @property
    def use_graphs(self) -> bool:
        return not self.enforce_eager and self.graph_mode != GraphMode.EAGER.value

    def __repr__(self):
        attrs = {k: v for k, v in self.__dict__.items() if k != "hf_config"}
        attrs["hf_config"] = f"{type(self.hf_config).__name__}(...)"
        return "Config(" + ", ".join(f"{k}={v}" for k, v in attrs.items()) + ")"
