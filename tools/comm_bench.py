"""Latency of the xGMI exchange kernels with WORLD rank processes sharing cuda:0 (no xGMI hop: this is
the kernels' own cost - launch, push, flag round trip, reduce - not the link's)."""
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-ascend_amd"))


def worker(rank, world, port):
    import torch.distributed as dist

    from nanovllm import ops
    from nanovllm.layers.xgmi_comm import XgmiComm

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = XgmiComm(rank, world, 64 * 5120 * 2, dev)
    rows, cols, reps = 32, 1024, 50
    x = torch.randn(rows, cols, device=dev).bfloat16()
    res = torch.randn(rows, cols, device=dev).bfloat16()
    w = torch.ones(cols, device=dev).bfloat16()

    def timed(fn, name):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                fn()
        dist.barrier()
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        s.record()
        for _ in range(5):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"world={world} {name}: {s.elapsed_time(e) * 1e3 / (5 * reps):.2f} us per call", flush=True)

    timed(lambda: comm.allreduce_add_rmsnorm(x, res, w, 1e-6), "(first graph: absorbs the processes' start-up skew)")
    timed(lambda: comm.allreduce_add_rmsnorm(x, res, w, 1e-6), "allreduce_add_rmsnorm [32x1024]")
    timed(lambda: comm.all_reduce(x), "allreduce_sum_bf16 in place [32x1024]")
    y = torch.empty_like(x)
    from nanovllm._C import check, lib, ptr, stream
    timed(lambda: check(lib.mi_allreduce_sum_bf16(comm._comm, ptr(res), ptr(y), y.numel(), stream()), "ar"),
          "allreduce_sum_bf16 out of place [32x1024]")
    timed(lambda: ops.add_rmsnorm(x, res, w, 1e-6), "add_rmsnorm alone (no exchange)")
    # a few wide rows (a Qwen3-32B TP-8 rank's decode step): the seam's multi-wave kernel against one wave per row
    from nanovllm import _C

    xw = torch.randn(32, 5120, device=dev).bfloat16()
    rw = torch.randn(32, 5120, device=dev).bfloat16()
    ww = torch.ones(5120, device=dev).bfloat16()
    for wpr, name in ((4, "eight waves per row"), (1, "one wave per row (MI_TUNE_NORM_WPR = 1)")):
        _C.set_tuning(_C.TUNE_NORM_WPR, wpr)
        timed(lambda: comm.allreduce_add_rmsnorm(xw, rw, ww, 1e-6), f"allreduce_add_rmsnorm [32x5120], {name}")
        timed(lambda: ops.add_rmsnorm(xw, rw, ww, 1e-6), f"add_rmsnorm alone [32x5120], {name}")
    _C.set_tuning(_C.TUNE_NORM_WPR, 4)
    assert not comm.timed_out()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    for world in [int(w) for w in sys.argv[1:]] or (1, 2, 4, 8):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(worker, args=(world, port), nprocs=world, join=True)
