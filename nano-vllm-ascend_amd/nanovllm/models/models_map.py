"""Architecture name -> model class (reference: nanovllm/models/models_map.py).
Only the Qwen3 / Qwen2 family is on the decode path this package covers."""
from nanovllm.models.qwen3 import Qwen3ForCausalLM

model_dict = {
    "Qwen3ForCausalLM": Qwen3ForCausalLM,
    "Qwen2ForCausalLM": Qwen3ForCausalLM,
}
