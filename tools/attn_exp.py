#!/usr/bin/env python3
"""Decode-attention layout / split experiments (uses the experimental mi_paged_attn_decode_ex)."""
import ctypes, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R + "/nano-vllm-ascend_amd")
from nanovllm import ops, _C
from ctypes import c_int, c_int64, c_size_t, c_float, c_void_p as P
lib = ctypes.CDLL(_C.LIB_PATH)
fn = lib.mi_paged_attn_decode_ex
fn.restype = c_int
fn.argtypes = [P, c_int64, P, P, P, c_int, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int64, c_int64, c_int64, P]
DEV = torch.device("cuda:0")
B, bs, hq, hkv, L = 32, 16, 16, 8, 28
PEAK = 8e12

def timeit(fn_, n_layers, iters=5):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for l in range(n_layers): fn_(l)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in range(n_layers): fn_(l)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / (iters * n_layers)

def run(ctx, splits, layout, seq_tables):
    nb = (ctx + bs - 1) // bs
    nblk = B * nb
    if layout == "blk":      # [nblk][Hkv][2048]
        kc = [torch.randn(nblk, hkv, 2048, device=DEV).bfloat16() for _ in range(L)]
        vc = [torch.randn(nblk, hkv, 2048, device=DEV).bfloat16() for _ in range(L)]
        sb, sh = hkv * 2048, 2048
    elif layout == "head":   # [Hkv][nblk][2048]
        kc = [torch.randn(hkv, nblk, 2048, device=DEV).bfloat16() for _ in range(L)]
        vc = [torch.randn(hkv, nblk, 2048, device=DEV).bfloat16() for _ in range(L)]
        sb, sh = 2048, nblk * 2048
    else:                    # "headkv": [Hkv][nblk][K|V]
        kv = [torch.randn(hkv, nblk, 2, 2048, device=DEV).bfloat16() for _ in range(L)]
        kc = kv; vc = [t.view(-1)[2048:] for t in kv]
        sb, sh = 4096, nblk * 4096
    gen = torch.Generator().manual_seed(0)
    tab = (torch.arange(nblk, dtype=torch.int32).view(B, nb) if seq_tables
           else torch.randperm(nblk, generator=gen).to(torch.int32).view(B, nb)).to(DEV)
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    q = torch.randn(B, hq * 128, device=DEV).bfloat16()
    out = torch.empty_like(q)
    ws = ops.attn_workspace(DEV, B, hq)
    def call(l):
        rc = fn(q.data_ptr(), q.stride(0), kc[l].data_ptr(), vc[l].data_ptr(), tab.data_ptr(), tab.stride(0),
                ctxl.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, hq, hkv, 128, bs, 128 ** -0.5,
                splits, sb, sh, 2048, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    t = timeit(call, L)
    byt = B * 2 * ctx * hkv * 128 * 2
    print(f"ctx={ctx:5d} splits={splits:2d} layout={layout:6s} tables={'seq' if seq_tables else 'rand'}  {t*1e6:7.2f} us  {byt/t/1e9:7.1f} GB/s  frac={byt/t/PEAK:.3f}", flush=True)

for layout in ("blk", "head", "headkv"):
    for seq in (False, True):
        run(1024, 8, layout, seq)
for splits in (1, 2, 4):
    run(1024, splits, "head", True)
run(1056, 1, "blk", True); run(1100, 1, "blk", True); run(1152, 1, "blk", True)
run(2048, 1, "blk", True); run(1024, 1, "blk", True)
