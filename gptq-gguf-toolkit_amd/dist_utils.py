"""Rank helpers + the two multi-GPU decisions of the path (reference dist_utils.py:19-59,
quant.py:177-179, gptq.py:131-132): calibration sharding and, new here, the per-matrix
owner assignment that lets the independent Linears of a block quantize on different GPUs."""
from typing import Dict, List, Sequence

import torch.distributed as dist


def is_dist_available_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_available_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_available_and_initialized() else 0


def is_main() -> bool:
    return get_rank() == 0


def print_on_main(*args, **kwargs):
    if is_main():
        print(*args, **kwargs)


def barrier():
    if is_dist_available_and_initialized():
        dist.barrier()


def shard_calibration(data: Sequence, rank: int, world_size: int) -> list:
    """Contiguous slice per rank, remainder dropped (reference quant.py:177-179)."""
    n = len(data) // world_size
    return list(data[rank * n:(rank + 1) * n])


def allreduce_hessian(H):
    """One collective per distinct Hessian (reference gptq.py:131-132: all_reduce AVG).
    Backend 'nccl' is RCCL over xGMI on ROCm; 'gloo' in the CPU tests."""
    if is_dist_available_and_initialized() and get_world_size() > 1:
        dist.all_reduce(H, op=dist.ReduceOp.AVG)
    return H


def assign_owners(costs: Dict[str, float], world_size: int) -> Dict[str, int]:
    """Longest-processing-time-first assignment of the Linears of one block to ranks.
    cost ~ R*C*(C+128).  Deterministic: ties broken by name, so every rank
    computes the same map without communication."""
    loads: List[float] = [0.0] * world_size
    owners: Dict[str, int] = {}
    for name in sorted(costs, key=lambda k: (-costs[k], k)):
        r = min(range(world_size), key=lambda i: (loads[i], i))
        owners[name] = r
        loads[r] += costs[name]
    return owners
