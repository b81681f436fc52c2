"""Vocab-parallel embedding and LM head (reference: nanovllm/layers/embed_head.py)."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from nanovllm import ops
from nanovllm.layers.linear import pack_for_decode, linear_forward
from nanovllm.layers import parallel
from nanovllm.layers.parallel import all_reduce_sum, collectives_on, get_xgmi_comm, tp_rank, tp_size
from nanovllm.utils.context import get_context


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank = tp_rank()
        self.tp_size = tp_size()
        assert num_embeddings % self.tp_size == 0
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = num_embeddings // self.tp_size
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim))
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        rows = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * rows, rows))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """embed_head.py:34-42: the kernel applies the vocab-range mask (rows owned by other
        ranks come out zero), then the partial embeddings are summed over ranks."""
        ctx = get_context()
        if ctx.token_src is not None and not ctx.is_prefill:
            y = ops.embedding_from_prev(x, ctx.token_src, ctx.prev_tokens, self.weight, self.vocab_start_idx)
        else:
            y = ops.embedding(x, self.weight, self.vocab_start_idx)
        return all_reduce_sum(y)


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        assert not bias
        super().__init__(num_embeddings, embedding_dim)
        self.weight_packed = None

    def pack(self) -> None:
        self.weight_packed = pack_for_decode(self.weight.data, self.weight_packed)

    def local_logits(self, x: torch.Tensor) -> torch.Tensor:
        """This rank's vocabulary shard of the logits (kernels only: what a decode graph captures)."""
        context = get_context()
        if context.is_prefill:  # keep only each sequence's last token (embed_head.py:58-60)
            x = ops.gather_last_tokens(x, context.cu_seqlens_q)
        return linear_forward(x, self.weight, None, self.weight_packed)

    def can_pick(self) -> bool:
        if self.weight_packed is None:
            return False
        # tensor parallelism: the ranks pick among themselves over the exchange region (bf16 shards only)
        if self.tp_size == 1:
            return not collectives_on()  # (the one-rank bring-up hook behaves like ranks without the exchange region)
        return get_xgmi_comm() is not None and not isinstance(self.weight_packed, ops.Fp8Weight)

    def local_logits_pick(self, x: torch.Tensor, temperatures: torch.Tensor, rng: torch.Tensor,
                          out_tokens: torch.Tensor) -> torch.Tensor:
        """Decode: the (local) logits AND the sampler's tokens (sampler.py:9-17 on the key scheme of
        layers/sampler.py here) from one pass over the head weight; `rng` = {seed, step} on the device.
        Tensor parallelism: instead of gathering [B, vocab / n] logits to rank 0 for it to sample
        (embed_head.py:62-65, sampler.py:9-17), every rank reduces its shard to its best {key, token} per row, the
        ranks exchange those 8 bytes per row and every rank takes the same winner - the tokens of the step are then
        on every rank's device, inside the captured graph."""
        assert not get_context().is_prefill and self.can_pick()
        if self.tp_size == 1:
            logits, _ = ops.gemm_packed_pick(x, self.weight_packed, temperatures, rng, out_tokens)
            return logits
        rows = x.shape[0]
        pairs = torch.empty((rows, 2), dtype=torch.int32, device=x.device)
        logits, _ = ops.gemm_packed_pick(x, self.weight_packed, temperatures, rng, out_tokens,
                                         col_offset=self.vocab_start_idx, pairs_out=pairs)
        get_xgmi_comm().pick_exchange(pairs, out_tokens)
        return logits

    def gather(self, logits: torch.Tensor):
        """Vocabulary shards -> rank 0 (embed_head.py:62-65); None on the other ranks.  A collective of the
        process group (RCCL), kept outside captured graphs - as the reference keeps compute_logits
        outside its compiled graph (model_runner.py:394-396)."""
        if not collectives_on():
            return logits
        if dist.get_backend() == "nccl":
            parts = [torch.empty_like(logits) for _ in range(self.tp_size)] if self.tp_rank == 0 else None
            parallel.STATS["rccl_gather"] += 1
            dist.gather(logits, parts, 0)
            return torch.cat(parts, -1) if self.tp_rank == 0 else None
        # gloo has no device-side gather: place the shard in a zero buffer and sum (tests only)
        full = torch.zeros((logits.shape[0], self.num_embeddings), dtype=logits.dtype, device=logits.device)
        full[:, self.vocab_start_idx:self.vocab_end_idx] = logits
        dist.all_reduce(full)
        return full if self.tp_rank == 0 else None

    def forward(self, x: torch.Tensor):
        return self.gather(self.local_logits(x))
