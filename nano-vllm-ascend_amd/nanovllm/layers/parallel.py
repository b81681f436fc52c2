"""Tensor-parallel helpers shared by the layers (rank / world size of the TP group).

One process per GPU; the group is the default torch.distributed group whose "nccl"
backend is RCCL over xGMI on ROCm.  With tensor_parallel_size == 1 no process group
is needed at all.
"""
from __future__ import annotations

import torch.distributed as dist


_tp: tuple[int, int] | None = None  # (rank, size) of the tensor-parallel group once the runner has declared it


def set_tp(rank: int, size: int) -> None:
    """Declared by ModelRunner from Config.tensor_parallel_size.  A process group may exist WITHOUT tensor
    parallelism (bench.py --mode replicas: N independent engines that only meet in a barrier), so the size
    of the default group says nothing about how the layers are sharded."""
    global _tp
    assert 0 <= rank < size
    _tp = (rank, size)


def reset_tp() -> None:
    global _tp
    _tp = None


def tp_rank() -> int:
    if _tp is not None:
        return _tp[0]
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def tp_size() -> int:
    if _tp is not None:
        return _tp[1]
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator


# ---- collectives on a group of ONE rank (a hardware-bring-up hook) -------------------------
# Every branch that calls RCCL (`dist.get_backend() == "nccl"`) is otherwise reachable only with several GPUs.  With
# MI355_TP1_COLLECTIVES=1 a tensor_parallel_size == 1 engine creates an RCCL group of world size 1 and takes the
# collective code paths it would take as one of n ranks whose xGMI exchange is unavailable: RCCL all-reduce behind the
# embedding and every row-parallel projection (eager prefill AND inside the captured decode graphs), the logits
# gather to rank 0, the device-side seed broadcast / MIN all-reduce of the start-up handshake.  On one rank every
# collective is the identity, so the tokens are those of the plain engine - and the calls are real RCCL launches.
_force_collectives = False
STATS = {"rccl_all_reduce": 0, "rccl_gather": 0, "rccl_broadcast": 0}  # counted per CALL SITE execution (tests)


def set_force_collectives(on: bool) -> None:
    global _force_collectives
    _force_collectives = bool(on)


def collectives_on() -> bool:
    """True when the layers must exchange partial results: several TP ranks, or the one-rank bring-up hook."""
    return tp_size() > 1 or _force_collectives


_xgmi = None  # layers/xgmi_comm.XgmiComm once the runner has set it up and its self-test passed


def set_xgmi_comm(comm) -> None:
    global _xgmi
    _xgmi = comm


def get_xgmi_comm():
    return _xgmi


def all_reduce_sum(t):
    """C1/C2 of SURVEY.md §2.2: SUM over the TP ranks, in place.  Decode-sized bf16 vectors take
    the one-shot xGMI kernel (mi_allreduce_sum_bf16, hipGraph-capturable, one launch); everything
    else (prefill-sized activations) the RCCL all-reduce."""
    if collectives_on():
        if _xgmi is not None and _xgmi.fits(t):
            return _xgmi.all_reduce(t)
        STATS["rccl_all_reduce"] += 1
        dist.all_reduce(t)
    return t
