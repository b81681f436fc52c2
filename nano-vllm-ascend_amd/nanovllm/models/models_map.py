"""Architecture name -> model class (reference: nanovllm/models/models_map.py).
The decoder families of the reference that share the paged-KV decode path: Qwen3 / Qwen2 (one class,
qwen3.py:70-72 switches between q/k norm and qkv bias), Llama (models/llama.py) and Qwen3-MoE
(models/qwen3_moe.py).  MiniCPM and Qwen3-VL are outside this package's scope."""
from nanovllm.models.llama import LlamaForCausalLM
from nanovllm.models.qwen3 import Qwen3ForCausalLM
from nanovllm.models.qwen3_moe import Qwen3MoeForCausalLM

model_dict = {
    "Qwen3ForCausalLM": Qwen3ForCausalLM,
    "Qwen2ForCausalLM": Qwen3ForCausalLM,
    "LlamaForCausalLM": LlamaForCausalLM,
    "Qwen3MoeForCausalLM": Qwen3MoeForCausalLM,
}
