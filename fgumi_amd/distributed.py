"""Multi-GPU plumbing: one process per GPU (`torch.distributed`; backend "nccl" = RCCL on ROCm, "gloo"
on CPU for the tests).  Families are independent, so the family stream is cut into contiguous shards —
rank r owns molecules [r*F, (r+1)*F) for weak scaling, or `shard_range` of a fixed total for strong
scaling — and no collective runs on the data path.  Consensus output is `SO:unsorted GO:query` in input
MI-group order (reference: src/lib/commands/consensus_runner.rs:156-161), so reassembly is concatenation
of the shard payloads in rank order."""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` for `rank` (first `n_items % world` ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_shards(weights: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous shards of a weighted family stream (weight = record bytes or reads×length) with roughly
    equal total weight: long-tail family sizes make equal-count shards unbalanced.
    Shard k ends after the first family at which the running weight reaches k/world of the total."""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    n = int(w.shape[0])
    if n == 0:
        return [(0, 0)] * world
    acc = np.cumsum(w)
    total = float(acc[-1])
    cuts = [0]
    for k in range(1, world):
        cuts.append(max(cuts[-1], min(n, int(np.searchsorted(acc, total * k / world, side="left")) + 1)))
    cuts.append(n)
    return [(cuts[i], max(cuts[i], cuts[i + 1])) for i in range(world)]


def gather_sizes(values: Sequence[int], device) -> torch.Tensor:
    """all_gather of a few int64 per rank (payload bytes, record count, …) → tensor [world, len(values)]."""
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t.unsqueeze(0)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)


def gather_payload_to_root(local: torch.Tensor, root: int = 0) -> Optional[torch.Tensor]:
    """Variable-length gather of the consensus payload (uint8 tensor) to `root`, concatenated in rank order.
    Sizes travel by all_gather; payloads by point-to-point send/recv (xGMI is point-to-point: the root
    receives on all its links at once; an all-gather would move world× the bytes for nothing)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank()
    sizes = gather_sizes([local.numel()], local.device)[:, 0].tolist()
    # ONE batch of point-to-point operations per rank (`batch_isend_irecv`): on RCCL the root's receives are posted as one group —
    # posted one by one, each irecv is its own group and a sender can wait behind another peer's unfinished transfer
    if rank == root:
        out = torch.empty(sum(sizes), dtype=torch.uint8, device=local.device)
        offs, ops = 0, []
        for r, n in enumerate(sizes):
            if r == root:
                out[offs:offs + n] = local
            elif n:
                ops.append(dist.P2POp(dist.irecv, out[offs:offs + n], r))
            offs += n
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return out
    if local.numel():
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, root)]):
            q.wait()
    return None


def sum_over_ranks(values: Sequence[int], device) -> List[int]:
    """all_reduce(SUM) of a vector of int64 counters (the batch statistics: families are independent, counters add up)."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def max_over_ranks(seconds: float, device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
