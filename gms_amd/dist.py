"""One-process-per-GPU plumbing for the sharded counts: rank/world discovery from the torchrun
environment and the single 8-byte all-reduce that replaces the reference's OpenMP `reduction(+:total)`
(triangle_count/parallel/total.h:12, k_clique_count_set_based.h:25).  torch.distributed's "nccl" backend is RCCL
on ROCm; on CPU (tests) the same code runs over "gloo"."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        # the control plane waits no longer for a dead peer than the library's own communicator does (include/gmsx.h: GMSX_COMM_TIMEOUT_S, default 180 s)
        try:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=max(10.0, float(os.environ.get("GMSX_COMM_TIMEOUT_S", "180")) * 2))
        except ValueError:
            pass
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def allreduce_count(partial: int, device=None) -> int:
    """Sum of the per-rank partial counts.  Counts travel as int64 two's complement, so sums that wrap mod 2^64
    (the reference's size_t arithmetic) survive the round trip."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return partial & 0xFFFFFFFFFFFFFFFF
    p = partial & 0xFFFFFFFFFFFFFFFF
    if p >= 1 << 63:
        p -= 1 << 64
    t = torch.tensor([p], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item()) & 0xFFFFFFFFFFFFFFFF


def allreduce_max(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
