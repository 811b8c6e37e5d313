"""The drivers over several GPUs: `sv_genotype` / `sso_genotype` with the VCF's variants sharded across
the ranks of one `torch.distributed` job (one process per GPU), one gather of the output text at the end.

This is the multi-GPU form of what `svtyper-sso --cores N` does with a `multiprocessing.Pool`
(svtyper/singlesample.py:710-762): variants are independent (svtyper/classic.py:279-513 keeps no state
across sites), so rank r runs the unchanged driver over a contiguous slice of the body lines on its own
GPU and rank 0 writes header + the slices in rank order.  The one cross-line dependency is BND mate
pairing (svtyper/parsers.py:155-178: the first mate waits in a dict until its partner arrives and both
lines are written at the partner's position), so a pair belongs to the rank that owns its SECOND mate and
that rank is handed the first mate's line as well.  The pairing is found by replaying the reference's
dict logic over (ID, MATEID) of the BND lines before the split.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m svtyper_amd.singlesample ...

picks this path up on its own (WORLD_SIZE > 1).  Output is byte-identical to the single-process run.
"""
from __future__ import annotations

import io
import os
from typing import Callable, List, Optional, Sequence


class ShardingUnsupported(Exception):
    """The body cannot be split without changing the pairing (repeated BND ids, BND without MATEID):
    rank 0 then runs every line."""


def _info_value(info: str, key: str) -> Optional[str]:
    """Last `key=value` of an INFO column, as the dict of parsers.py:260-268 would hold it."""
    found = None
    for item in info.split(";"):
        kv = item.split("=")
        if kv[0] == key:
            found = kv[1] if len(kv) > 1 else ""
    return found


def bnd_pairs(body: Sequence[str]) -> dict:
    """{index of second mate: index of first mate}, by replaying parsers.py:155-178 over the BND lines."""
    pending: dict = {}
    seen: set = set()
    pairs: dict = {}
    for idx, line in enumerate(body):
        if "SVTYPE=BND" not in line:
            continue
        cols = line.split("\t", 8)
        if len(cols) < 8 or _info_value(cols[7].rstrip("\n"), "SVTYPE") != "BND":
            continue
        var_id = cols[2]
        mate_id = _info_value(cols[7].rstrip("\n"), "MATEID")
        if mate_id is None or var_id in seen:
            raise ShardingUnsupported(var_id)
        seen.add(var_id)
        first = pending.get(mate_id)
        if first is None:
            pending[var_id] = idx
        else:
            pairs[idx] = first
            del pending[mate_id]
    return pairs


def plan_shards(body: Sequence[str], world: int) -> List[List[int]]:
    """Body-line indices for each rank, in file order.  Rank r owns a contiguous range; first mates whose
    partner lies in a later range move to that range's rank, so every output line is produced exactly
    once and rank order is file order."""
    n = len(body)
    try:
        pairs = bnd_pairs(body)
    except ShardingUnsupported:
        return [list(range(n))] + [[] for _ in range(world - 1)]
    per = -(-n // world) if n else 0
    owner = lambda i: min(i // per, world - 1) if per else 0
    moved_to = {first: owner(second) for second, first in pairs.items() if owner(first) != owner(second)}
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in range(n):
        plan[moved_to.get(i, owner(i))].append(i)
    for p in plan:
        p.sort()
    if n and not plan[0]:
        # every line of rank 0's range was a first mate handed to a later rank: rank 0 would then never see a body
        # line, and the drivers write the VCF header when they meet the first one (classic.py:166-176) -- the gathered
        # output would have no header at all.  Inputs that small do not need sharding: one rank takes everything.
        return [list(range(n))] + [[] for _ in range(world - 1)]
    return plan


class _Lines(io.StringIO):
    """What the drivers need of an input file: iteration, read(), readline(), readlines(), close(), name."""

    def __init__(self, lines: List[str], name: str):
        super().__init__("".join(lines), newline="\n")
        self.name = name


class _Sink(io.StringIO):
    def close(self):        # sv_genotype closes its output; the text is still wanted afterwards
        pass


def _gather_text(text: str, rank: int, world: int) -> Optional[List[str]]:
    """One collective: every rank's text onto rank 0 (RCCL when the job runs on GPUs, gloo otherwise)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import distributed as D

    device = _backend_device()
    raw = np.frombuffer(text.encode("utf-8"), np.uint8)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[rank] = raw.size
    dist.all_reduce(sizes)
    sizes = [int(x) for x in sizes.tolist()]
    local = torch.from_numpy(raw.copy()).to(device)
    got = D.gather_bytes(local, sizes, dst=0)
    if rank != 0:
        return None
    flat = got.cpu().numpy().tobytes()
    out, at = [], 0
    for s in sizes:
        out.append(flat[at:at + s].decode("utf-8"))
        at += s
    return out


def _backend_device():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _require_same_input(lines: List[str], rank: int, world: int) -> None:
    """Every rank plans the split from its own copy of the VCF, so the copies must be the same text (a rank
    reading an empty stdin would otherwise drop its share silently)."""
    import zlib
    import torch
    import torch.distributed as dist

    crc = 0
    for line in lines:
        crc = zlib.crc32(line.encode("utf-8", "surrogateescape"), crc)
    mine = torch.tensor([len(lines), crc], dtype=torch.int64, device=_backend_device())
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if lo.tolist() != hi.tolist():
        raise RuntimeError("rank %d of %d: the ranks did not read the same VCF (%d lines here); give the input "
                           "as a file (-i), not on stdin, when running under torch.distributed.run" % (rank, world, len(lines)))


def run_sharded(driver: Callable, bam_string, vcf_in, vcf_out, *rest, rank: int, world: int,
                lib_info_index: int, **kw) -> None:
    """`driver(bam_string, vcf_in, vcf_out, *rest, **kw)` over this rank's share of `vcf_in`; rank 0
    writes everything to `vcf_out`.  Needs an initialised process group."""
    if vcf_in is None:
        return driver(bam_string, None, vcf_out, *rest, **kw)
    import sys
    import time
    trace = bool(os.environ.get("SVT_TRACE"))
    t0 = time.perf_counter()
    lines = vcf_in.readlines()
    n_head = 0
    while n_head < len(lines) and lines[n_head].startswith("#"):
        n_head += 1
    body = lines[n_head:]
    _require_same_input(lines, rank, world)
    mine = plan_shards(body, world)[rank]
    rest = list(rest)
    lib_info_path = rest[lib_info_index]
    if rank != 0 and lib_info_path is not None and not os.path.exists(lib_info_path):
        rest[lib_info_index] = None        # only rank 0 writes the library JSON; the others just scan
    sink = _Sink()
    share = _Lines(lines[:n_head] + [body[i] for i in mine], getattr(vcf_in, "name", "<stdin>"))
    t1 = time.perf_counter()
    driver(bam_string, share, sink, *rest, **kw)
    t2 = time.perf_counter()
    text = sink.getvalue()
    if rank != 0:                           # header comes from rank 0 only (body lines never start with '#')
        kept = [l for l in text.splitlines(True) if not l.startswith("#")]
        text = "".join(kept)
    parts = _gather_text(text, rank, world)
    if rank == 0:
        for part in parts:
            vcf_out.write(part)
        vcf_out.flush()
    if trace:
        sys.stderr.write("[sharded] rank %d/%d: %d of %d lines | read+plan %.2f s, driver %.2f s, gather+write %.2f s\n"
                         % (rank, world, len(mine), len(body), t1 - t0, t2 - t1, time.perf_counter() - t2))


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def job() -> Optional[tuple]:
    """(rank, world, local_rank) when launched by torch.distributed.run with more than one rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    return int(os.environ["RANK"]), world, int(os.environ.get("LOCAL_RANK", "0"))


def init(local_rank: int, backend: Optional[str] = None):
    """Process group + this rank's engine: one GPU per process and RCCL for the gather; when the node has
    fewer GPUs than local ranks the ranks share devices and the gather goes over gloo (RCCL wants one
    device per rank)."""
    import torch
    import torch.distributed as dist
    from . import hip
    from .pipeline import HipEngine

    hip.load()
    n_dev = hip.device_count()
    if n_dev <= 0:
        raise hip.SvtyperHipError("no MI355X visible: svtyper_amd has no CPU fallback for the likelihood path")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    device = local_rank % n_dev
    if backend is None:
        backend = "nccl" if local_world <= n_dev and torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts; RCCL needs it
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        dist.init_process_group(backend)
    os.environ.setdefault("SVT_READER_THREADS", str(max(1, usable_cpus() // max(local_world, 1))))
    return HipEngine(device)


def private_stdout(vcf_out):
    """When the VCF goes to stdout: keep the real stdout for it and point file descriptor 1 at stderr, because
    RCCL writes its version banner to stdout (C stdio, flushed at exit) and would end up inside the VCF."""
    import sys
    if vcf_out is not sys.stdout:
        return vcf_out
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def finish() -> None:
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def sv_genotype_sharded(bam_string, vcf_in, vcf_out, *rest, rank: int, world: int, **kw) -> None:
    """classic.sv_genotype's arguments (lib_info_path is the 8th positional) plus rank / world."""
    from .classic import sv_genotype
    run_sharded(sv_genotype, bam_string, vcf_in, vcf_out, *rest, rank=rank, world=world, lib_info_index=4, **kw)


def sso_genotype_sharded(bam_string, vcf_in, vcf_out, *rest, rank: int, world: int, **kw) -> None:
    from .singlesample import sso_genotype
    run_sharded(sso_genotype, bam_string, vcf_in, vcf_out, *rest, rank=rank, world=world, lib_info_index=4, **kw)
