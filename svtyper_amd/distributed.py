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
    """Every rank's uint8 tensor (sizes[r] bytes on rank r, on the backend's device) onto `dst`: the root posts one receive per
    rank straight into ITS slice of one preallocated buffer, every other rank one send of exactly its bytes (one batch of
    point-to-point operations: over RCCL one group, each peer over its own xGMI link) -- no padding to the largest shard and
    no second pass over the payload on the root.  Returns the uint8 tensor on `dst` (rank order), None elsewhere.
    SVT_GATHER=collective selects the padded `dist.gather` + concatenation this replaced (same bytes)."""
    import os
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    if local.numel() != sizes[rank]:
        raise ValueError("rank %d holds %d bytes, the sizes say %d" % (rank, local.numel(), sizes[rank]))
    if os.environ.get("SVT_GATHER") == "collective" or not _p2p_usable(local.device, dst):
        return _gather_bytes_collective(local, sizes, dst)
    if rank != dst:
        if sizes[rank]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst)]):
                req.wait()
        return None
    starts = [0]
    for c in sizes:
        starts.append(starts[-1] + c)
    out = torch.empty(starts[-1], dtype=torch.uint8, device=local.device)
    ops = [dist.P2POp(dist.irecv, out[starts[r]:starts[r + 1]], r) for r in range(world) if r != dst and sizes[r]]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    out[starts[dst]:starts[dst + 1]] = local       # the root's own records: one device copy, under the transfers
    for req in reqs:
        req.wait()
    return out


_P2P_OK = None


def _p2p_usable(device, dst: int) -> bool:
    """Once per process: one byte from every rank to `dst` over the same batched point-to-point calls; the ranks agree
    (all-reduce, minimum) on whether that worked, so that a backend that cannot do it sends EVERY rank to the collective route
    instead of leaving some of them behind."""
    global _P2P_OK
    if _P2P_OK is None:
        import torch
        import torch.distributed as dist
        ok = 1
        try:
            rank, world = dist.get_rank(), dist.get_world_size()
            if rank == dst:
                box = torch.zeros(max(world, 1), dtype=torch.uint8, device=device)
                ops = [dist.P2POp(dist.irecv, box[r:r + 1], r) for r in range(world) if r != dst]
            else:
                ops = [dist.P2POp(dist.isend, torch.ones(1, dtype=torch.uint8, device=device), dst)]
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _P2P_OK = bool(int(flag.item()))
    return _P2P_OK


def gather_route() -> str:
    """which route gather_bytes takes in this process: point-to-point into one buffer, or the padded collective"""
    import os
    if os.environ.get("SVT_GATHER") == "collective" or _P2P_OK is False:
        return "padded dist.gather + concatenation"
    return "point-to-point into one buffer"


def _gather_bytes_collective(local, sizes: List[int], dst: int = 0):
    """One `dist.gather` of payloads padded to the largest shard, then a concatenation on the root."""
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


def gather_result_records(local, counts: List[int], dst: int = 0):
    """Gather every rank's result records (a uint8 torch tensor of n_local * 128 bytes, on the
    backend's device) onto `dst`; `counts` are records per rank."""
    rec = ev.RESULT_DTYPE.itemsize
    return gather_bytes(local, [c * rec for c in counts], dst)


def gather_tagged_records(local, dst: int = 0):
    """The same for batches created with FLAG_RESULT96: a rank's device buffer holds result_slots() tagged 96-byte records in
    the order its kernel finished them (a quarter fewer bytes through the collective).  How many each rank holds is only known
    to that rank, so the sizes travel first (one tiny all_gather).  Returns (uint8 tensor, bytes per rank) on `dst`,
    (None, sizes) elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    mine = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(x.item()) for x in sizes]
    return gather_bytes(local, sizes, dst), sizes


def results_from_bytes(t) -> Results:
    """uint8 tensor of 128-byte result records (any device) -> Results on the host."""
    a = t.cpu().numpy()
    return Results(a.view(ev.RESULT_DTYPE).copy())


def results_from_tagged(t, sizes: List[int], counts: List[int]) -> Results:
    """The gathered tagged 96-byte records of all ranks (sizes[r] bytes from rank r, whose shard holds counts[r] units) ->
    Results in unit order: every rank's records are put where their tags say, behind the units of the ranks before it."""
    from . import hip
    a = t.cpu().numpy()
    out = Results.empty(sum(counts))
    at = base = 0
    for nbytes, n in zip(sizes, counts):
        hip.expand96(a[at:at + nbytes], n, out.rec[base:base + n])
        at += nbytes
        base += n
    return out


# ---- the compact gather record ------------------------------------------------------------------------------------------------
# What a consumer of GENOTYPES reads of a result record (parsers.py:375-399 prints GT, GQ, SQ, GL and, of the counts, QR / QA;
# the other counts are int() of sums of the tallies): 48 bytes instead of 96 through the collective, i.e. half the bytes the root
# has to take in -- the one gather onto rank 0 is bound by the root's xGMI ingest, not by the passes.
COMPACT_DTYPE = np.dtype([("gl", "<f8", (3,)), ("sq", "<f8"), ("qr", "<i4"), ("qa", "<i4"), ("gq", "<i2"), ("gt", "i1"), ("pad", "u1"),
                          ("unit", "<u4")])
assert COMPACT_DTYPE.itemsize == 48


def compact_tagged_records(t):
    """Tagged 96-byte device records (uint8 tensor, any device, slots * 96 bytes) -> slots * 48 bytes of COMPACT_DTYPE records
    (GL, SQ, QR, QA, GQ, GT, the unit tag) on the same device: four strided copies."""
    import torch
    x = t.view(-1, 96)
    out = torch.zeros((x.shape[0], 48), dtype=torch.uint8, device=t.device)
    out[:, 0:32] = x[:, 0:32]        # gl[3], sq
    out[:, 32:40] = x[:, 72:80]      # qr, qa
    out[:, 40:42] = x[:, 80:82]      # gq: the low half of an int32 in [-1, 200]
    out[:, 42] = x[:, 84]            # gt
    out[:, 44:48] = x[:, 88:92]      # unit
    return out.view(-1)


def results_from_compact(t, sizes: List[int], counts: List[int]) -> Results:
    """The gathered compact records of all ranks -> Results in unit order with GL, SQ, GT and the QR / QA / GQ counts filled
    (tallies and the other counts are not part of the compact record: zero)."""
    a = t.cpu().numpy()
    out = Results.empty(sum(counts))
    out.rec[:] = np.zeros((), ev.RESULT_DTYPE)
    seen = np.zeros(sum(counts), bool)
    at = base = 0
    for nbytes, n in zip(sizes, counts):
        c = a[at:at + nbytes].view(COMPACT_DTYPE)
        c = c[c["unit"] != ev.NO_UNIT]
        if len(c) != n or (n and (c["unit"].max() >= n or len(np.unique(c["unit"])) != n)):
            raise ValueError("compact records of a rank do not cover its units exactly once")
        u = c["unit"].astype(np.int64) + base
        r = out.rec
        r["gl"][u] = c["gl"]
        r["sq"][u] = c["sq"]
        r["counts"][u, ev.COUNT_NAMES.index("QR")] = c["qr"]
        r["counts"][u, ev.COUNT_NAMES.index("QA")] = c["qa"]
        r["counts"][u, ev.COUNT_NAMES.index("GQ")] = c["gq"]
        r["gt"][u] = c["gt"]
        seen[u] = True
        at += nbytes
        base += n
    assert seen.all()
    return out
