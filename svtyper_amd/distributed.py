"""Multi-GPU: shard units across ranks, genotype locally, ONE gather of the result records.

Units are independent (svtyper/classic.py:279-513 keeps no cross-site state; the only cross-unit
quantity, QUAL = sum of SQ over a site's samples, stays local because all samples of a site are
kept on one rank), so the N-GPU path is: contiguous shards balanced by record count -> the same
kernel on every rank -> one gather of the fixed-size 128-byte result records onto rank 0
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  No reduction, no per-step
collective.

One process per GPU, `torch.distributed` for the rendezvous and the collective only.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import evidence as ev
from .evidence import EvidenceBatch, Results


def shard_bounds(rec_offset: np.ndarray, world: int, group: int = 1) -> List[Tuple[int, int]]:
    """[lo, hi) unit ranges, one per rank, balanced by the bytes a unit costs (16 F + 112) and cut
    only at multiples of `group` units (group = number of samples per site keeps a site's samples
    together)."""
    n = int(rec_offset.shape[0]) - 1
    if n <= 0:
        return [(0, 0)] * world
    cost = (rec_offset[1:] - rec_offset[:-1]).astype(np.float64) * 16.0 + 112.0
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(n, max(cuts[-1], (k // group) * group))
        cuts.append(k)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def local_shard(batch: EvidenceBatch, rank: int, world: int, group: int = 1) -> Tuple[EvidenceBatch, Tuple[int, int]]:
    lo, hi = shard_bounds(batch.rec_offset, world, group)[rank]
    return batch.slice(lo, hi), (lo, hi)


def gather_bytes(local, sizes: List[int], dst: int = 0):
    """Gather every rank's uint8 tensor (sizes[r] bytes on rank r, on the backend's device) onto `dst`
    with one collective.  Shards may differ in size, so the payload is padded to the largest one.
    Returns the concatenated uint8 tensor on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    width = max(max(sizes), 1)
    buf = local
    if local.numel() != width:
        buf = torch.zeros(width, dtype=torch.uint8, device=local.device)
        buf[: local.numel()] = local
    out = [torch.empty(width, dtype=torch.uint8, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:c] for o, c in zip(out, sizes)])


def gather_result_records(local, counts: List[int], dst: int = 0, rec_bytes: int = ev.RESULT_DTYPE.itemsize):
    """Gather every rank's result records (a uint8 torch tensor of n_local * rec_bytes bytes, on the
    backend's device) onto `dst`; `counts` are records per rank.  rec_bytes: 128 (svt_result) or 96
    (svt_result96, batches created with FLAG_RESULT96: a quarter fewer bytes through the collective)."""
    return gather_bytes(local, [c * rec_bytes for c in counts], dst)


def results_from_bytes(t, rec_bytes: int = ev.RESULT_DTYPE.itemsize) -> Results:
    """uint8 tensor of result records (any device) -> Results on the host (96-byte records are expanded:
    svt_results_expand96 restores the counts that follow from the tallies)."""
    a = t.cpu().numpy()
    if rec_bytes == ev.RESULT96_DTYPE.itemsize:
        from . import hip
        return hip.expand96(a)
    return Results(a.view(ev.RESULT_DTYPE).copy())
