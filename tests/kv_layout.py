"""Host-side model of the fragment-native KV layout (include/mi355_nanovllm.h),
used by the tests to move caches between the reference's logical layout
[nblk, block, Hkv, D] and the device layout [nblk, Hkv, block/16, 16 * D] (D = 128 or 64: the tile formulas do not
depend on the head width, a narrower head fills the first D / 32 blocks of 512 elements)."""
import torch


def _offsets(block_size: int, is_v: bool, head_dim: int = 128) -> torch.Tensor:
    """[block_size, head_dim] -> element offset inside one (block, head) slab."""
    s = torch.arange(block_size).unsqueeze(1)
    d = torch.arange(head_dim).unsqueeze(0)
    t, tile = s % 16, s // 16
    if is_v:
        off = (d // 32) * 512 + (((t // 4) * 16) + (d % 16)) * 8 + ((d // 16) % 2) * 4 + (t % 4)
    else:
        off = (d // 32) * 512 + ((((d // 8) % 4) * 16 + t) * 8) + (d % 8)
    return tile * (16 * head_dim) + off


def to_fragment(cache_logical: torch.Tensor, is_v: bool) -> torch.Tensor:
    nblk, bs, hkv, d = cache_logical.shape
    assert d in (64, 128) and bs % 16 == 0
    off = _offsets(bs, is_v, d).reshape(-1)  # [bs*d]
    src = cache_logical.permute(0, 2, 1, 3).reshape(nblk, hkv, bs * d)
    out = torch.empty_like(src)
    out[:, :, off] = src
    return out.view(nblk, hkv, bs // 16, 16 * d).contiguous()


def to_logical(cache_frag: torch.Tensor, block_size: int, is_v: bool) -> torch.Tensor:
    nblk, hkv = cache_frag.shape[:2]
    d = cache_frag.shape[-1] // 16
    off = _offsets(block_size, is_v, d).reshape(-1)
    flat = cache_frag.reshape(nblk, hkv, block_size * d)
    return flat[:, :, off].view(nblk, hkv, block_size, d).permute(0, 2, 1, 3).contiguous()
