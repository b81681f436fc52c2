"""Architecture name -> model class (reference: nanovllm/models/models_map.py).
The dense decoder families of the reference that share the paged-KV decode path: Qwen3 / Qwen2 (one class,
qwen3.py:70-72 switches between q/k norm and qkv bias) and Llama (models/llama.py).  MiniCPM and Qwen3-VL
are outside this package's scope."""
from nanovllm.models.llama import LlamaForCausalLM
from nanovllm.models.qwen3 import Qwen3ForCausalLM

model_dict = {
    "Qwen3ForCausalLM": Qwen3ForCausalLM,
    "Qwen2ForCausalLM": Qwen3ForCausalLM,
    "LlamaForCausalLM": LlamaForCausalLM,
}
