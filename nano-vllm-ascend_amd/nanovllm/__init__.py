"""nanovllm — MI355X-native paged-KV Qwen3 decode path behind the nano-vLLM API.

Same import surface as the reference package (nanovllm/__init__.py:1-2):
    from nanovllm import LLM, SamplingParams
Heavy imports (transformers, the engine) are deferred until first use.
"""
__all__ = ["LLM", "SamplingParams"]


def __getattr__(name):
    if name == "LLM":
        from nanovllm.llm import LLM

        return LLM
    if name == "SamplingParams":
        from nanovllm.sampling_params import SamplingParams

        return SamplingParams
    raise AttributeError(name)
