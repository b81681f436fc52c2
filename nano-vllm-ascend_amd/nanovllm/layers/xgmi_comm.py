"""One-shot SUM all-reduce over xGMI peer mappings (C ABI mi_comm_* / mi_allreduce_sum_bf16),
the MI355X stand-in for the HCCL all-reduce of linear.py:152-153 and embed_head.py:41-42.

Set-up (once per engine): every rank allocates an uncached exchange region, the 64-byte HIP IPC
handles travel through the already initialised torch.distributed group, peers map each other's
regions.  A self-test with integer-valued data (exact sums) decides - collectively - whether the
path is used; if it fails on any rank, every rank keeps the RCCL all-reduce.
"""
from __future__ import annotations

import ctypes
import os
import warnings

import torch
import torch.distributed as dist

from nanovllm._C import check, lib, ptr, stream


class XgmiComm:
    def __init__(self, rank: int, world: int, max_bytes: int, device: torch.device):
        self.rank, self.world, self.max_bytes, self.device = rank, world, int(max_bytes), device
        # What an EAGER launch may put through the region (ModelRunner sets it to the decode-sized rows): the region is
        # also sized for captured prefill steps, whose ranks replay the same graph together; an eager prefill step's ranks
        # arrive at their collectives host-skewed by whole launch sequences, and its prefill-sized all-reduces stay on the
        # process group (RCCL) as in rounds 1-5.  `large()` lifts the cap while a prefill graph is captured.
        self.eager_max_bytes = int(max_bytes)
        self._large = 0
        self._peers: list[int] = []
        self._own = ctypes.c_void_p()
        self._comm = ctypes.c_void_p()
        nbytes = lib.mi_comm_region_bytes(world, self.max_bytes)
        handle = ctypes.create_string_buffer(64)
        check(lib.mi_comm_region_alloc(nbytes, ctypes.byref(self._own), handle), "mi_comm_region_alloc")
        handles: list = [None] * world
        dist.all_gather_object(handles, handle.raw)
        regions = []
        for r in range(world):
            if r == rank:
                regions.append(self._own.value)
            else:
                p = ctypes.c_void_p()
                check(lib.mi_comm_region_open(handles[r], ctypes.byref(p)), "mi_comm_region_open")
                self._peers.append(p.value)
                regions.append(p.value)
        arr = (ctypes.c_void_p * world)(*regions)
        check(lib.mi_comm_create(rank, world, arr, self.max_bytes, ctypes.byref(self._comm)), "mi_comm_create")
        dist.barrier()  # every rank has mapped every region before the first push

    def _cap(self) -> int:
        return self.max_bytes if self._large else min(self.max_bytes, self.eager_max_bytes)

    def large(self):
        """context: launches inside may use the whole region (capture of a prefill step's graph and its warm-up run)"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self._large += 1
            try:
                yield self
            finally:
                self._large -= 1
        return scope()

    def fits(self, t: torch.Tensor) -> bool:
        return (t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous() and t.numel() % 8 == 0
                and 0 < t.numel() * 2 <= self._cap())

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        check(lib.mi_allreduce_sum_bf16(self._comm, ptr(t), ptr(t), t.numel(), stream()), "mi_allreduce_sum_bf16")
        return t

    def fits_rows(self, rows: int, cols: int) -> bool:
        return 0 < rows <= 512 and 512 <= cols <= 8192 and cols % 8 == 0 and rows * cols * 2 <= self._cap()

    def allreduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float):
        """sum over ranks of x, + residual, RMSNorm: one launch (mi_allreduce_add_rmsnorm)."""
        cols = x.shape[-1]
        rows = x.numel() // cols
        out, residual_out = torch.empty_like(x), torch.empty_like(x)
        check(lib.mi_allreduce_add_rmsnorm(self._comm, ptr(x), ptr(residual), ptr(w), ptr(out), ptr(residual_out),
                                           rows, cols, float(eps), stream()), "mi_allreduce_add_rmsnorm")
        return out, residual_out

    def pick_exchange(self, pairs: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """pairs [rows, 2] int32 = this rank's best {key bits, token} per row -> tokens[rows] int64, the same on
        every rank (mi_pick_exchange)."""
        rows = pairs.shape[0]
        assert pairs.dtype == torch.int32 and pairs.is_contiguous() and tokens.dtype == torch.int64
        assert tokens.numel() >= rows and rows * 8 <= self.max_bytes
        check(lib.mi_pick_exchange(self._comm, ptr(pairs), ptr(tokens), rows, stream()), "mi_pick_exchange")
        return tokens

    def status_async(self, landing: torch.Tensor) -> None:
        """queue a copy of the sticky timeout flag into `landing` (pinned int32) behind the stream's work"""
        check(lib.mi_comm_status_async(self._comm, landing.data_ptr(), stream()), "mi_comm_status_async")

    def timed_out(self) -> bool:
        flag = ctypes.c_int(0)
        check(lib.mi_comm_status(self._comm, ctypes.byref(flag)), "mi_comm_status")
        return bool(flag.value)

    def timeout_info(self) -> str:
        """what this rank's first timed-out exchange was waiting for (mi_comm_timeout_info), for the error message"""
        info = (ctypes.c_uint32 * 4)()
        try:
            check(lib.mi_comm_timeout_info(self._comm, info), "mi_comm_timeout_info")
        except Exception as e:  # noqa: BLE001 - diagnostics must not mask the time-out itself
            return f"(no details: {e!r})"
        epoch, slice_, peer, seen = (int(v) for v in info)
        return (f"exchange {epoch}, slice {slice_}: no flag from rank {peer} (saw {seen}: "
                f"{'that rank is behind' if seen < epoch else 'that rank is AHEAD'})")

    def self_test(self) -> bool:
        """Integer-valued inputs: the fp32 sum and its bf16 rounding are exact, so the expected
        result is known without a second collective.  Covers the smallest and the largest vector,
        consecutive launches (both epoch parities) and replay from a captured graph."""
        ok = True
        try:
            # a broken topology must not cost a minute per launch: ~4 s patience while testing, and the
            # first (tiny) exchange decides whether the rest is attempted at all
            torch.cuda.synchronize()
            check(lib.mi_comm_set_spin_limit(self._comm, 1 << 22), "mi_comm_set_spin_limit")
            dist.barrier()
            probe = torch.full((8,), float(self.rank + 1), dtype=torch.bfloat16, device=self.device)
            self.all_reduce(probe)
            first = not self.timed_out() and bool((probe == self.world * (self.world + 1) / 2).all())
            if not self._agree(first):  # every rank stops here together
                raise RuntimeError("first exchange failed")
            for n in (8, 4096, self.max_bytes // 2):
                idx = torch.arange(n, device=self.device, dtype=torch.int64)
                for it in range(3):
                    x = (((idx * 7 + (self.rank + it) * 3) % 17) - 8).to(torch.bfloat16)
                    want = sum((((idx * 7 + (r + it) * 3) % 17) - 8) for r in range(self.world)).to(torch.bfloat16)
                    self.all_reduce(x)
                    ok = ok and bool(torch.equal(x, want))
            idx = torch.arange(4096, device=self.device, dtype=torch.int64)
            src = ((idx * 5 + self.rank) % 13 - 6).to(torch.bfloat16)
            want = sum(((idx * 5 + r) % 13 - 6) for r in range(self.world)).to(torch.bfloat16)
            buf = torch.empty_like(src)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                buf.copy_(src)
                self.all_reduce(buf)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):  # (the process group's watchdog thread polls events)
                for _ in range(3):  # three dependent launches per replay
                    buf.copy_(src)
                    self.all_reduce(buf)
            for _ in range(2):
                graph.replay()
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(buf, want))
            # the sampler's pair exchange: rank r offers key = (row + r) % world (ties between ranks on purpose),
            # token = 1000 r + row; the winner of a row is the largest key, lowest token among equals
            rows = 70
            row = torch.arange(rows, device=self.device, dtype=torch.int64)
            key = ((row + self.rank) % self.world).to(torch.float32)
            pairs = torch.stack([key.view(torch.int32), (1000 * self.rank + row).to(torch.int32)], 1).contiguous()
            toks = torch.empty(rows, dtype=torch.int64, device=self.device)
            self.pick_exchange(pairs, toks)
            winner = (self.world - 1 - row) % self.world  # the rank whose key is world - 1
            ok = ok and bool(torch.equal(toks, 1000 * winner + row))
            ok = ok and not self.timed_out()
            # fault injection for tests: "the self-test failed on rank k" must end with EVERY rank on the RCCL path
            if os.environ.get("MI355_XGMI_SELFTEST_FAIL_RANK") == str(self.rank):
                ok = False
            # patience in production: ~a minute and more per exchange (MI355_XGMI_SPIN_LIMIT: bring-up / repro runs)
            check(lib.mi_comm_set_spin_limit(self._comm, int(os.environ.get("MI355_XGMI_SPIN_LIMIT", 1 << 26))),
                  "mi_comm_set_spin_limit")
        except Exception as e:  # noqa: BLE001 - any failure means "do not use this path"
            warnings.warn(f"xGMI all-reduce self-test raised {e!r}")
            ok = False
        return self._agree(ok)

    def _agree(self, ok: bool) -> bool:
        verdict = torch.tensor([1 if ok else 0], dtype=torch.int32,
                               device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        return bool(verdict.item())

    def close(self) -> None:
        if self._comm:
            lib.mi_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()
        for p in self._peers:
            lib.mi_comm_region_close(ctypes.c_void_p(p))
        self._peers = []
        if self._own:
            lib.mi_comm_region_free(self._own)
            self._own = ctypes.c_void_p()


LAST_STATUS = "not attempted"  # why the path is on or off (bench.py prints it with the N > 1 line)


def create_if_enabled(rank: int, world: int, max_bytes: int, device: torch.device) -> XgmiComm | None:
    """None when disabled by MI355_XGMI_ALLREDUCE=0, when set-up fails or when the self-test fails."""
    global LAST_STATUS
    if world < 2 or world > 8 or os.environ.get("MI355_XGMI_ALLREDUCE", "1") == "0":
        LAST_STATUS = "disabled"
        return None
    comm = None
    try:
        comm = XgmiComm(rank, world, max_bytes, device)
        made = True
    except Exception as e:  # noqa: BLE001
        warnings.warn(f"xGMI all-reduce set-up failed ({e!r}); using the RCCL all-reduce")
        made = False
    agree = torch.tensor([1 if made else 0], dtype=torch.int32,
                         device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    if not agree.item():
        if comm is not None:
            comm.close()
        LAST_STATUS = "set-up failed on some rank: RCCL all-reduce"
        return None
    if not comm.self_test():
        if rank == 0:
            warnings.warn("xGMI all-reduce self-test failed; using the RCCL all-reduce")
        comm.close()
        LAST_STATUS = "self-test failed: RCCL all-reduce"
        return None
    LAST_STATUS = f"self-test passed on {world} ranks"
    return comm
