"""Rank helpers + the two multi-GPU decisions of the path (reference dist_utils.py:19-59,
quant.py:177-179, gptq.py:131-132): calibration sharding and, new here, the per-matrix
owner assignment that lets the independent Linears of a block quantize on different GPUs."""
import os
from typing import Dict, List, Sequence, Set, Tuple

import torch.distributed as dist


# data-path collectives issued by this process (tests and bench.py read it: "one all-reduce per distinct Hessian and
# one all-gather per block" is asserted, not only claimed)
collective_calls: Dict[str, int] = {"all_reduce": 0, "reduce": 0, "all_gather": 0, "broadcast": 0, "small_all_reduce": 0}
# bytes this rank handed to them (the Hessian payloads and the packed results; "small_all_reduce" = the few-byte ones:
# per-expert sample counts, the MoE sharing agreement)
collective_bytes: Dict[str, int] = {"all_reduce": 0, "reduce": 0, "all_gather": 0, "broadcast": 0, "small_all_reduce": 0}


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


def _avg(payload, dst=None):
    """AVG over the ranks, to every rank or -- `dst` -- to that rank only (half the bytes on every link of a ring).
    The op is chosen from the BACKEND, once per call and identically on every rank (never by catching an exception around a
    collective: a rank-local failure would make one rank issue a second collective the others do not): nccl (= RCCL) has
    AVG; gloo does not -- SUM, divided where the result lives."""
    if dist.get_backend() != "gloo":
        if dst is None:
            dist.all_reduce(payload, op=dist.ReduceOp.AVG)
        else:
            dist.reduce(payload, dst=dst, op=dist.ReduceOp.AVG)
        return
    if dst is None:
        dist.all_reduce(payload, op=dist.ReduceOp.SUM)
        payload.div_(get_world_size())
    else:
        dist.reduce(payload, dst=dst, op=dist.ReduceOp.SUM)
        if get_rank() == dst:
            payload.div_(get_world_size())


def allreduce_hessian(H, num_samples=None, dst=None):
    """One collective per distinct Hessian (reference gptq.py:131-132: all_reduce AVG).
    Backend 'nccl' is RCCL over xGMI on ROCm; 'gloo' in the CPU tests.

    `dst`: REDUCE TO OWNER (SURVEY section 5 iii) -- once the Linears are fanned out, a Hessian whose Linears all belong to one
    rank is needed there only: `dist.reduce` instead of the reference's all_reduce; H is valid on `dst` alone afterwards.

    `num_samples` (this rank's sample count behind H) makes the reduction sample-weighted when the ranks saw
    DIFFERENT counts -- MoE experts receive a data-dependent number of tokens per rank, and
    H = (2/N) sum x x^T over all ranks' tokens is sum_r n_r H_r / sum_r n_r, not the plain mean.  With equal
    counts (every dense Linear) the reference's AVG is used unchanged.  Returns the total sample count (None when
    num_samples is None)."""
    if not (is_dist_available_and_initialized() and get_world_size() > 1):
        return num_samples
    import torch
    # H is exactly symmetric: only its upper 128x128 tiles travel (gq_h_pack_upper / gq_h_unpack_upper), half the
    # bytes on the xGMI links, and the reduced H is symmetric bit for bit on every rank
    packed = H.is_cuda and H.dtype == torch.float32 and H.is_contiguous() and _packs_upper(H.shape[0])
    if packed:
        from . import ops
        payload = ops.h_pack_upper(H)
    else:
        payload = H
    total = None
    kind = "all_reduce" if dst is None else "reduce"
    collective_calls[kind] += 1
    collective_bytes[kind] += payload.numel() * payload.element_size()
    if num_samples is None:
        _avg(payload, dst)
    else:
        counts = torch.zeros(get_world_size(), dtype=torch.float64, device=H.device)
        counts[get_rank()] = float(num_samples)
        collective_calls["small_all_reduce"] += 1
        collective_bytes["small_all_reduce"] += counts.numel() * 8
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)  # (a few bytes: the sample counts of an MoE expert)
        total = int(counts.sum().item())
        if bool((counts == counts[0]).all()):
            _avg(payload, dst)
        elif total > 0:
            payload.mul_(float(num_samples) / total)
            if dst is None:
                dist.all_reduce(payload, op=dist.ReduceOp.SUM)
            else:
                dist.reduce(payload, dst=dst, op=dist.ReduceOp.SUM)
    if packed and (dst is None or dst == get_rank()):
        ops.h_unpack_upper(payload, H)
    return total


def _packs_upper(C: int) -> bool:
    return C % 128 == 0 and C >= 1024


def hessian_payload_bytes(C: int) -> int:
    """Bytes one rank hands to the all-reduce of a C x C Hessian (upper 128x128 tiles only when packed)."""
    nt = C // 128
    return 4 * (128 * 128 * nt * (nt + 1) // 2 if _packs_upper(C) else C * C)


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


def row_split_names(costs: Dict[str, float], world_size: int) -> Set[str]:
    """Linears of a block that EVERY rank quantizes on its own slice of rows (rows are independent in
    the column loop given U, so a matrix that alone outweighs a fair share is split R/k rows per GPU with the
    factorisation replicated).  Replicating the Cholesky chain costs every rank its full time, so the split
    only pays from 4 ranks up (2 ranks: one takes the widest matrix, the other everything else -- already
    balanced); GQ_ROW_SPLIT=0 / 1 forces never / always, "all" splits every matrix (tests).  Deterministic: the
    same set on every rank."""
    env = os.environ.get("GQ_ROW_SPLIT")
    if world_size <= 1 or env == "0":
        return set()
    if env == "all":
        return set(costs)
    if env != "1" and world_size < 4:
        return set()
    total = sum(costs.values())
    return {n for n, c in costs.items() if c > 0.4 * total}


def row_slice(rows: int, rank: int, world_size: int, align: int = 128) -> Tuple[int, int, int]:
    """-> (r0, r1, chunk): rank's rows [r0, r1) of a row-split matrix; chunk (a multiple of `align`, so the
    whole-tile kernels apply) is the padded per-rank height used by the all-gather.  Late ranks may be empty."""
    chunk = -(-rows // world_size)
    chunk = -(-chunk // align) * align
    r0 = min(rank * chunk, rows)
    return r0, min(r0 + chunk, rows), chunk


def all_gather_rows(part, rows: int, chunk: int):
    """Concatenate the ranks' row slices (each padded to `chunk` rows) -> [rows, ...] on every rank.  (The handle-level
    exchange of a row-split matrix; a BlockSchedule gathers a whole block at once: all_gather_bytes.)"""
    import torch
    world = get_world_size()
    pad = part
    if part.shape[0] != chunk:
        pad = torch.zeros((chunk,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        pad[:part.shape[0]] = part
    out = torch.empty((world * chunk,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
    _all_gather_into(out, pad.contiguous())
    return out[:rows] if world * chunk != rows else out


def _all_gather_into(out, inp) -> None:
    """out[r * n : (r + 1) * n] = rank r's inp (n = inp.numel()), ONE collective, no list of temporaries."""
    collective_calls["all_gather"] += 1
    collective_bytes["all_gather"] += inp.numel() * inp.element_size()
    # (nccl = RCCL and gloo both have the tensor form; no fallback around a collective: a rank-local failure must not make one
    # rank issue a collective the others do not)
    dist.all_gather_into_tensor(out.view(-1), inp.view(-1))


def all_gather_bytes(mine, nbytes: int):
    """Every rank contributes `nbytes` bytes (uint8 tensor `mine`, padded by the caller to the common size); returns
    uint8 [world, nbytes] on every rank.  The result exchange of one transformer block (block_schedule.py)."""
    import torch
    assert mine.dtype == torch.uint8 and mine.numel() == nbytes
    out = torch.empty((get_world_size(), nbytes), dtype=torch.uint8, device=mine.device)
    _all_gather_into(out, mine)
    return out
