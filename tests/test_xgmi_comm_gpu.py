"""mi_allreduce_sum_bf16 with several rank processes sharing cuda:0: HIP IPC mapping of the exchange
regions, the per-slice flag protocol, both epoch parities, replay from a hipGraph.  (One GPU cannot
show cross-device cache behaviour; the engine therefore re-runs XgmiComm.self_test on the real
topology at start-up and falls back to RCCL when it fails.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    import torch.distributed as dist

    from nanovllm.layers.xgmi_comm import XgmiComm

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = XgmiComm(rank, world, 1 << 20, dev)  # (1 MiB per slot: 65 x 5120 bf16 rows fit)
    try:
        ok = comm.self_test()
        # random values: the kernel sums the bf16 inputs in fp32 in rank order, one rounding at the end
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(32 * 1024, generator=g).bfloat16()
        everyone = [None] * world
        dist.all_gather_object(everyone, x)
        acc = torch.zeros_like(x, dtype=torch.float32)
        for t in everyone:
            acc += t.float()
        want = acc.bfloat16()
        y = x.to(dev)
        exact = True
        for _ in range(5):  # alternating parities, same input
            z = y.clone()
            comm.all_reduce(z)
            exact = exact and bool(torch.equal(z.cpu(), want))
        # fused all-reduce + add + RMSNorm == all-reduce kernel followed by mi_add_rmsnorm, bit for bit
        from nanovllm import ops

        fused_ok = True
        # (up to 64 rows of more than 1024 columns run the seam's multi-wave kernel, as mi_add_rmsnorm runs its own: a
        # Qwen3-32B TP-8 rank's 32 x 5120 decode rows; 65 rows of 5120 stay on one wave per row)
        for rows, cols in ((1, 1024), (32, 1024), (64, 1024), (7, 5120), (5, 512), (32, 5120), (64, 4096), (8, 8192),
                           (33, 2048), (65, 5120)):
            g2 = torch.Generator().manual_seed(rows * 10000 + cols)  # same residual / weight on every rank
            res = torch.randn(rows, cols, generator=g2).bfloat16().to(dev)
            w = (1 + 0.1 * torch.randn(cols, generator=g2)).bfloat16().to(dev)
            part = torch.randn(rows, cols, generator=torch.Generator().manual_seed(7 * rank + rows)).bfloat16().to(dev)
            want_y, want_r = ops.add_rmsnorm(comm.all_reduce(part.clone()), res, w, 1e-6)
            got_y, got_r = comm.allreduce_add_rmsnorm(part, res, w, 1e-6)
            fused_ok = fused_ok and bool(torch.equal(got_y, want_y)) and bool(torch.equal(got_r, want_r))
        out.put((rank, ok, exact and fused_ok, comm.timed_out()))
    finally:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_allreduce_ranks_sharing_one_gpu(world):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in range(world):
            results.append(out.get(timeout=180))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert len(results) == world
    for rank, ok, exact, timed_out in sorted(results):
        assert ok, f"rank {rank}: self-test failed"
        assert exact, f"rank {rank}: sum differs from the rank-ordered fp32 sum"
        assert not timed_out, f"rank {rank}: a peer timed out"


def _worker_missing_peer(rank, world, port, out):
    import ctypes

    import torch.distributed as dist

    from nanovllm._C import check, lib
    from nanovllm.layers.xgmi_comm import XgmiComm

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = XgmiComm(rank, world, 1 << 16, dev)
    try:
        x = torch.full((64,), float(rank + 1), dtype=torch.bfloat16, device=dev)
        comm.all_reduce(x)  # a complete exchange first (epoch 1)
        torch.cuda.synchronize()
        first_ok = bool((x == 3).all()) and not comm.timed_out()
        dist.barrier()
        info = ""
        if rank == 0:  # rank 1 never launches the second exchange
            check(lib.mi_comm_set_spin_limit(comm._comm, 1 << 12), "mi_comm_set_spin_limit")
            comm.all_reduce(torch.ones(64, dtype=torch.bfloat16, device=dev))
            comm.all_reduce(torch.ones(64, dtype=torch.bfloat16, device=dev))  # (behind a time-out: gives up at once)
            torch.cuda.synchronize()
            info = comm.timeout_info()
            raw = (ctypes.c_uint32 * 4)()
            check(lib.mi_comm_timeout_info(comm._comm, raw), "mi_comm_timeout_info")
            info = (info, [int(v) for v in raw])
        out.put((rank, first_ok, comm.timed_out(), info))
    finally:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


def test_a_missing_peer_raises_the_sticky_flag_and_says_who_was_missing():
    """A peer that never launches its side of an exchange: the waiting rank gives up after its spin limit instead of
    hanging the GPU, the sticky flag is set, mi_comm_timeout_info names the exchange (epoch 2), the slice and the missing
    rank with the stale flag value it saw (epoch 0 or 1: that rank is behind) - and the exchange queued behind the
    time-out does not wait its patience again."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_missing_peer, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = sorted(out.get(timeout=180) for _ in range(2))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    (r0, ok0, timed0, info0), (r1, ok1, timed1, _) = results
    assert ok0 and ok1 and not timed1
    assert timed0, "rank 0 waited for a peer that never came and did not report it"
    text, (epoch, slice_, peer, seen) = info0
    assert epoch == 2 and slice_ == 0 and peer == 1 and seen < epoch, info0
    assert "rank 1" in text and "behind" in text
