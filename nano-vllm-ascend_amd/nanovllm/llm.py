"""`LLM`, the user-facing entry point - the engine under the name the reference exports
(nanovllm/llm.py:4-6), so `from nanovllm import LLM, SamplingParams` call sites carry over.

    llm = LLM("/models/Qwen3-0.6B", kvcache_block_size=16, max_num_seqs=32)
    outs = llm.generate(["hello"], SamplingParams(temperature=0.6, max_tokens=64))

Keyword arguments are the fields of `nanovllm.config.Config`.  `from_config_dict` is the
no-checkpoint mode used by the tests and the benchmark: a HF `config.json` dictionary instead of
a model directory selects seeded random weights of that architecture.
"""
from __future__ import annotations

import json
import os
import tempfile

from nanovllm.engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    @classmethod
    def from_config_dict(cls, hf_config: dict, root: str | None = None, **kwargs) -> "LLM":
        model_dir = tempfile.mkdtemp(prefix="mi355_model_", dir=root)
        with open(os.path.join(model_dir, "config.json"), "w") as f:
            json.dump(hf_config, f)
        return cls(model_dir, **kwargs)
