#!/usr/bin/env python
"""bench.py -- breakpoints genotyped per second on N x MI355X (BASELINE.json metric).

    python bench.py                       (= --gpus 1 --steps 100 --warmup 3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling strong]

A *step* is ONE pass of the whole hot path over one synthetic batch whose canonical input (SURVEY.md 8d:
rec_offset[], 16-byte unit headers, 16-byte evidence records, exactly as the C ABI receives them) is already
resident in HBM: a single launch of svt_stream_kernel reads every record once and does the evidence tally, the
zeroing rules, QR/QA, bayes_gt and the GT/GQ/SQ decision, leaving the result records (tagged 96-byte ones by default) in HBM.  Nothing
is pre-digested outside the timed region: svt_batch_create is upload only (no scan, no tiling, no re-encoding),
so `roofline.achieved` = algorithmic bytes / kernel time cannot exceed the HBM peak.
Setup before the timed steps (reported, not timed): svt_batch_tune_placement -- the real pass over a handful of freshly allocated
device buffers for the result records and the records, the fastest kept (which physical blocks of HBM the two lie in moves the
pass by up to 8 %; `roofline.placement_tuned` = before / after, `--tune-placement 0,0` = none; `roofline.placement` = six fresh
allocations WITHOUT it, `roofline.no_spinup_kernel_ms` = the buffers svt_batch_create drew, cold) --, then spin_up's untimed passes.

Workload: BASELINE.json configs[2] -- 1 M mixed DEL/DUP/INV breakpoints, one library (the reference fixture's
empirical insert-size histogram, staged in LDS), ~100 fragment records (~200 reads) per breakpoint.
  --scaling weak   (default) that workload PER GPU: independent units, no data-path collective inside a step;
  --scaling strong configs[3] literally: ONE 1 M-unit workload cut into contiguous shards balanced by bytes
                   (svtyper_amd.distributed.shard_bounds; with --workload c5_multisample a site's 32 samples stay
                   together), every rank genotypes its shard.
After the timed region every rank's result records are gathered onto rank 0 with ONE RCCL gather over xGMI
(north_star: "a single RCCL gather ... at the end"); its time is reported separately under "gather".

Extra keys on the N=1 line (clearly labelled, never part of `value`; --legs selects them): `sso` (the same launch with
the singlesample association) and `c5_multisample` (the configs[4] shape: 32 samples, per-sample libraries; plus the
same batch site-major, without library hints -- windows read off the records at create -- and with every table in L2)
each with their own roofline object, `shard_of_8` (shard 0 of the 8-GPU cut of
the headline workload, timed alone), `one_shot` (host buffers -> results on the host, PCIe included),
`one_shot_packed` (the same from packed evidence, the host encoder's time included), `large_batch` (4 M units per
GPU: working set far beyond the 256 MiB Infinity Cache), `cpu_baseline`, `parity`.  N > 1 adds `value_with_gather`.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_SAMPLES_C5 = 32


def usable_cpus() -> int:
    """Host threads this process can actually keep busy: the affinity mask capped by the cgroup CPU
    quota (a container with `cpu.max = 1600000 100000` schedules 16 CPUs however many it can see)."""
    n = len(os.sched_getaffinity(0))
    quota, period = -1, 0
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            quota, period = (-1 if q == "max" else int(q)), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota > 0 and period > 0:
        n = min(n, max(1, -(-quota // period)))
    return max(1, n)


def fixture_library():
    from svtyper_amd.evidence import LibraryTable
    with open(os.path.join(ROOT, "tests", "data", "NA12878.bam.json")) as f:
        info = json.load(f)
    lib = info["NA12878"]["libraryArray"][0]
    return LibraryTable.from_counter({int(k): int(v) for k, v in lib["histogram"].items()},
                                     float(lib["mean"]), float(lib["sd"]), "NA12878")


def _gen_chunk(args):
    name, n, idx, rank = args
    from svtyper_amd import synth
    cfg = synth.CONFIGS[name]
    return synth.make_units(n, synth.BASE_SEED + cfg["config_no"] + 1000 * idx + 7919 * rank,
                            [fixture_library()], svtype_mix=cfg["svtype_mix"])


def generate(name: str, n_units: int, rank: int, workers: int, first_chunk: int = 0, layout: str = "site"):
    """The synthetic workload of this rank (chunks generated in parallel host processes)."""
    from svtyper_amd import evidence as ev
    if name == "c5_multisample":
        import multiprocessing as mp
        from svtyper_amd import synth
        with mp.get_context("fork").Pool(min(max(1, workers), 32)) as pool:
            lo, hi = (int(x) for x in os.environ.get("SVT_BENCH_C5_LIBS", "1,3").split(","))   # (probes: libraries per sample)
            return synth.make_multisample(max(1, n_units // N_SAMPLES_C5), N_SAMPLES_C5, synth.BASE_SEED + 5 + 7919 * rank,
                                          libs_per_sample=(lo, hi), pool_map=pool.map, layout=layout)
    chunk = 50_000
    jobs = [(name, min(chunk, n_units - i), first_chunk + i // chunk, rank) for i in range(0, n_units, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.map(_gen_chunk, jobs)
    else:
        parts = [_gen_chunk(j) for j in jobs]
    return parts[0] if len(parts) == 1 else ev.concat_batches(parts)


def _count_chunk(args):
    name, n, idx, rank = args
    from svtyper_amd import synth
    cfg = synth.CONFIGS[name]
    return synth.make_units(n, synth.BASE_SEED + cfg["config_no"] + 1000 * idx + 7919 * rank,
                            [fixture_library()], svtype_mix=cfg["svtype_mix"], counts_only=True)


def generate_shard(name: str, n_units: int, world: int, r: int, workers: int, group: int = 1):
    """Shard r of `world` of the workload generate(name, n_units, rank=0) under the shard rule (distributed.shard_bounds),
    without building the rest of it: the records per unit of every chunk are the first few draws of its generator
    (synth.make_units(counts_only=True)), the bounds follow from them, and only the chunks the shard overlaps are generated.
    Returns (the shard, the bounds of all ranks)."""
    from svtyper_amd import distributed as D
    from svtyper_amd import evidence as ev
    chunk = 50_000
    jobs = [(name, min(chunk, n_units - i), i // chunk, 0) for i in range(0, n_units, chunk)]
    counts = np.concatenate([_count_chunk(j) for j in jobs]) if jobs else np.zeros(0, np.int64)
    rec_offset = np.zeros(n_units + 1, np.uint64)
    np.cumsum(counts, out=rec_offset[1:])
    bounds = D.shard_bounds(rec_offset, world, group)
    lo, hi = bounds[r]
    if hi <= lo:
        return _gen_chunk(jobs[0]).slice(0, 0), bounds
    first, last = lo // chunk, (hi - 1) // chunk
    mine = jobs[first:last + 1]
    if workers > 1 and len(mine) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(mine))) as pool:
            parts = pool.map(_gen_chunk, mine)
    else:
        parts = [_gen_chunk(j) for j in mine]
    b = parts[0] if len(parts) == 1 else ev.concat_batches(parts)
    return b.slice(lo - first * chunk, hi - first * chunk), bounds


def pipelined_passes(dbatch, sizes, batches, backend, compact=False):
    """`batches` passes of the resident batch, each followed by the gather of ITS result records onto rank 0 -- with the gather
    of batch k running under the pass of batch k + 1: two result buffers per rank, the pass writes one while the collective
    reads the other (RCCL on its own stream; ranks that share a device: gloo from a host copy).  Every rank calls this between
    barriers; returns the seconds for all of it on this rank (the caller takes the maximum over the ranks)."""
    import torch
    from svtyper_amd import distributed as D
    cur = dbatch.result_slots() * dbatch.result_bytes()
    bufs = [torch.empty(cur + 128, dtype=torch.uint8, device="cuda") for _ in range(2)]
    base = [(b.data_ptr() + 127) // 128 * 128 - b.data_ptr() for b in bufs]
    # a result buffer is written by pass k and read by the gather of batch k (torch's / the backend's streams); pass k + 2 may
    # only be launched into it once that gather has finished: an event recorded behind the gather, waited for on the host
    # before the launch -- inside the timed region, as a pipeline over DIFFERENT batches would have to
    read_done = [None, None]
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(batches + 1):
            if k < batches:
                if read_done[k % 2] is not None:
                    read_done[k % 2].synchronize()
                dbatch.bind_device_results(bufs[k % 2].data_ptr() + base[k % 2], cur)
                dbatch.genotype(sync=False)
            if k >= 1:
                j = (k - 1) % 2
                src = bufs[j][base[j]: base[j] + cur]
                if compact:
                    src = D.compact_tagged_records(src)
                D.gather_bytes(src if backend == "nccl" else src.cpu(), sizes, dst=0)
                read_done[j] = torch.cuda.Event()
                read_done[j].record()      # (behind the collective: a synchronous c10d call makes the current stream wait for it)
            if k < batches:
                dbatch.synchronize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    finally:
        dbatch.bind_device_results(0)


def dist_leg(what, dbatch, counts, total_units, steps, warmup, rank, world, backend, coll_device, n_dev, order=0, alone=None, sites_per_unit=None):
    """One workload through the N-rank job, every rank calling this with ITS resident shard: the passes alone (barrier + device
    sync on both sides, maximum over the ranks), the single gather of the result records onto rank 0, the same with compact
    48-byte records, and the pipelined steady state (pipelined_passes).  Rank 0 returns the leg's dict; `alone` (rank 0): the
    result records of one rank's pass over the WHOLE workload -- what the gathered records must equal."""
    import torch
    import torch.distributed as dist
    from svtyper_amd import distributed as D
    from svtyper_amd import evidence as ev
    rec_bytes = dbatch.result_bytes()
    if order:
        dbatch.result_order(order)
    dbatch.genotype(sync=True)
    dist.barrier()          # (the ranks meet before they spin up: nobody idles in the barrier in front of the timed passes)
    spin_up(dbatch, 20.0)
    for _ in range(warmup):
        dbatch.genotype(sync=False)
    dbatch.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = dbatch.genotype_timed(steps) / steps
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    mine = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=coll_device)
    every = [torch.zeros(2, dtype=torch.float64, device=coll_device) for _ in range(world)]
    dist.all_gather(every, mine)
    elapsed = max(float(t[0].item()) for t in every)
    kern_all = [float(t[1].item()) for t in every]

    res = dbatch.device_results_tensor()
    cur = dbatch.result_slots() * rec_bytes

    def one_gather(compact):
        dist.barrier()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        src = res[:cur]
        if compact:
            src = D.compact_tagged_records(src)
        src = src if backend == "nccl" else src.cpu()
        if rec_bytes == 96:
            gathered, sizes = D.gather_tagged_records(src, dst=0)
        else:
            sizes = [c * rec_bytes for c in counts]
            gathered = D.gather_result_records(src, counts, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        return gathered, sizes, time.perf_counter() - g0

    gathered, sizes, g_s = one_gather(False)
    same = None
    if rank == 0 and alone is not None:
        joined = D.results_from_tagged(gathered, sizes, counts) if rec_bytes == 96 else D.results_from_bytes(gathered)
        same = bool(np.array_equal(joined.rec, alone))
        del joined
    gathered = None
    compact = None
    if rec_bytes == 96:
        c_gathered, c_sizes, c_s = one_gather(True)
        c_same = None
        if rank == 0 and alone is not None:
            j = D.results_from_compact(c_gathered, c_sizes, counts).rec
            c_same = bool(np.array_equal(j["gl"], alone["gl"]) and np.array_equal(j["sq"], alone["sq"]) and np.array_equal(j["gt"], alone["gt"])
                          and np.array_equal(j["counts"][:, :3], alone["counts"][:, :3]))
        c_gathered = None
        compact = {"record_bytes": 48, "ms": c_s * 1e3, "GB/s_into_root": sum(c_sizes[1:] or c_sizes) / c_s / 1e9,
                   "fields": "GL, SQ, QR, QA, GQ, GT + the unit tag (what parsers.py:375-399 prints of a genotype; DP / RO / AO / RS / AS / ASC / RP / AP "
                             "need the 40 bytes of tallies)", "genotype_fields_equal_single_rank_pass": c_same}
    # ---- the other way off the devices: every rank copies ITS records to page-locked host memory over its own PCIe link, all
    # ranks at once (what one process does in svt_genotype_multi, one host thread per device; what the sharded drivers do before
    # they format their share of the VCF: svtyper_amd/sharded.py gathers TEXT, not records) -- no root that has to take in N - 1 shards
    d2h = None
    try:
        host = torch.empty(cur, dtype=torch.uint8).pin_memory()
        host.copy_(res[:cur], non_blocking=True)          # (the first touch of the pinned pages)
        torch.cuda.synchronize()
        dist.barrier()
        h0 = time.perf_counter()
        host.copy_(res[:cur], non_blocking=True)
        torch.cuda.synchronize()
        dist.barrier()
        h_s = time.perf_counter() - h0
        t = torch.tensor([h_s, float(cur)], dtype=torch.float64, device=coll_device)
        every_h = [torch.zeros(2, dtype=torch.float64, device=coll_device) for _ in range(world)]
        dist.all_gather(every_h, t)
        h_s = max(float(x[0].item()) for x in every_h)
        d2h = {"ms": h_s * 1e3, "GB/s_aggregate": sum(float(x[1].item()) for x in every_h) / h_s / 1e9, "record_bytes": rec_bytes,
               "what": "every rank's result records device -> its own page-locked host buffer, all ranks at once (barrier on both sides, "
                       "maximum over the ranks)"}
        del host
    except Exception as e:
        d2h = {"error": repr(e)}
    res = None
    # ---- steady state: pass of batch k+1 over the gather of batch k
    batches = max(8, steps)
    pipe = {}
    for key, comp in (("value_pipelined", False),) + ((("value_pipelined_compact", True),) if rec_bytes == 96 else ()):
        use_sizes = (c_sizes if comp else sizes)
        dist.barrier()
        p_s = pipelined_passes(dbatch, use_sizes, batches, backend, compact=comp)
        dist.barrier()
        t = torch.tensor([p_s], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pipe[key] = total_units * batches / float(t[0].item())
        pipe[key.replace("value_", "ms_per_batch_")] = float(t[0].item()) / batches * 1e3
    if rank != 0:
        return None
    value = total_units * steps / elapsed
    leg = {"what": what, "total_units": int(total_units), "units_per_rank": [int(c) for c in counts], "steps": steps,
           "kernel_ms_per_rank": kern_all, "ms_per_step": elapsed / steps * 1e3, "value": value, "unit": "breakpoints/s",
           "gather": {"record_bytes": rec_bytes, "bytes_per_rank": int(cur), "ms": g_s * 1e3, "GB/s_into_root": sum(sizes[1:] or sizes) / g_s / 1e9,
                      "collective": "rccl gather" if backend == "nccl" else "gloo gather (ranks share %d device(s))" % n_dev},
           "gather_compact": compact,
           "d2h_parallel": d2h,
           "value_with_gather": total_units / (elapsed / steps + g_s),
           "value_with_d2h_parallel": total_units / (elapsed / steps + d2h["ms"] * 1e-3) if d2h and "ms" in d2h else None,
           "batches_pipelined": batches,
           "pipelined_note": "two result buffers per rank; the gather of batch k runs under the pass of batch k + 1 (steady state over "
                             "`batches_pipelined` batches, barrier on both sides, maximum over the ranks): the N-GPU throughput of pass + gather",
           "equals_single_rank_pass": same, "rccl_ranks": world if backend == "nccl" else 0}
    leg.update(pipe)
    if sites_per_unit:
        leg["sites_per_s"] = value / sites_per_unit
    return leg


def claim_stdout():
    """The contract is ONE JSON line on stdout, but libraries below us write there too (RCCL prints its version
    banner on stdout through C stdio, flushed at exit).  Keep the real stdout for the JSON line and point file
    descriptor 1 at stderr for everybody else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def library_stamp() -> str:
    """Identity of the kernel build the numbers belong to (first 16 hex digits of the .so's sha256)."""
    from svtyper_amd import hip
    h = hashlib.sha256()
    with open(hip.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


_COMPILER = None


def compiler_stamp() -> str:
    """the device compiler the in-tree library is built with (`hipcc --version`: HIP version + clang version lines)"""
    global _COMPILER
    if _COMPILER is None:
        import subprocess
        try:
            text = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True, timeout=60).stdout
            keep = [l.strip() for l in text.splitlines() if l.startswith("HIP version") or "clang version" in l]
            _COMPILER = "; ".join(keep) or "unknown"
        except Exception:
            _COMPILER = "unknown"
    return _COMPILER


def source_stamp() -> str:
    """Identity of the kernel SOURCES (sha256 over svtyper_amd/csrc/{*.hip,*.h,Makefile} and include/*.h, first 16 hex
    digits; the host-only *.cpp files -- BAM reader, formatter, packed-evidence encoder -- do not reach the device
    code).  hipcc does not produce the same bytes twice from the same sources, so a PMC entry is keyed by this and
    survives a rebuild by build() -- and is still refused once any kernel source has changed."""
    import glob
    h = hashlib.sha256()
    h.update(compiler_stamp().encode() + b"\0")   # (an entry must not survive a toolchain change either)
    files = sorted(glob.glob(os.path.join(ROOT, "svtyper_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "svtyper_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "svtyper_amd", "csrc", "Makefile")] +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(key: str, n_units: int, n_records: int):
    """HBM bytes per launch of a leg's kernel from the committed PMC passes (tools/profile.sh -> profiles/hbm_traffic.json;
    rocprofv3 cannot wrap this process from inside).  An entry only counts when it was measured on THESE kernel sources
    (source_stamp) and on this workload: a stale entry is refused, not reused."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            tj = json.load(f)[key]
    except (OSError, ValueError, KeyError):
        return None, "no PMC entry '%s' in profiles/hbm_traffic.json" % key
    if tj.get("units") != n_units or tj.get("records") != n_records:
        return None, "profiles/hbm_traffic.json['%s'] was measured on another workload (%s units): refused" % (key, tj.get("units"))
    if tj.get("source_sha16") != source_stamp():
        return None, ("profiles/hbm_traffic.json['%s'] was measured on other kernel sources (%s, these are %s): refused; re-run "
                      "tools/profile.sh" % (key, tj.get("source_sha16"), source_stamp()))
    return tj["traffic_bytes_per_launch"], ("rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) on these kernel sources, "
                                            "profiles/hbm_traffic.json['%s']" % key)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


LEGS = ("place", "sso", "r96", "c5", "c5x", "c5f", "shard", "one_shot", "packed", "large", "real")   # c5x: the c5 leg's comparison launches (site-major input, no hints)


def roofline_of(kernel_ms: float, alg_bytes: int, key: str, n_units: int, n_records: int) -> dict:
    """the roofline object of one leg: algorithmic bytes over the measured launch time, PMC traffic when this build has it"""
    ach = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, note = measured_traffic(key, n_units, n_records)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": "bytes per launch", "traffic_source": note,
            "traffic_kind": "committed PMC profile of these kernel sources (rocprofv3 cannot wrap this run from inside), not a measurement of this run",
            "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms}


SPINUP_MS = 40.0


def spin_up(dbatch, ms: float = None) -> int:
    """Keep the device busy with untimed passes for `ms` milliseconds so that the timed launches run at the clocks of a
    loaded GPU: after an idle period the first ~50 launches of this 0.35 ms kernel run ~6 % slower (measured: 20 timed
    steps after 3-5 warm-up steps 0.358-0.364 ms, after 50 warm-up steps 0.338 ms on the same box).  Returns the
    number of passes it ran; they are not steps and not warm-up steps, and nothing is skipped in the timed region."""
    ms = SPINUP_MS if ms is None else ms
    n, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        dbatch.genotype_n(8)
        dbatch.genotype(sync=True)
        n += 9
    return n


def time_passes(dbatch, steps: int) -> float:
    """average launch duration in ms: HIP events on the launch stream around `steps` back-to-back passes"""
    spin_up(dbatch)
    return dbatch.genotype_timed(steps) / steps


def _wgs_like_bam(path: str, genome: int = 1_200_000, coverage: float = 30.0, spacing: int = 4_000, seed: int = 1, sample: str = "smp"):
    """A bounded BAM that looks like whole-genome sequencing (tests/bamwriter.py): 150-bp pairs at ~30x with random bases and
    binned qualities (BGZF blocks inflate at a realistic cost), one DEL site every few kb -- every site touches blocks nobody
    has inflated yet.  Returns (library info dict, breakpoint dicts)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamwriter as bw
    rng = np.random.default_rng(seed)
    n_pairs = int(genome * coverage / 300)
    header = "@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:%d\n@RG\tID:rg\tSM:%s\tLB:lib\n" % (genome, sample)
    starts = rng.integers(0, genome - 1000, n_pairs)
    isz = np.clip(rng.normal(400, 60, n_pairs), 160, 900).astype(np.int64)
    seq_pool = rng.integers(0, 4, (4096, 75))
    seq_pool = ((1 << seq_pool) << 4 | (1 << rng.integers(0, 4, (4096, 75)))).astype(np.uint8)
    qual_pool = np.array([2, 11, 25, 37], np.uint8)[rng.choice(4, (4096, 150), p=[0.02, 0.08, 0.2, 0.7])]
    seqs = [seq_pool[i].tobytes() for i in range(4096)]
    quals = [qual_pool[i].tobytes() for i in range(4096)]
    mqs = [0, 20, 37, 60, 60, 60, 60]
    split = rng.random(n_pairs) < 0.02
    pick = rng.integers(0, 4096, (n_pairs, 2, 2))
    mq = rng.integers(0, len(mqs), (n_pairs, 2))
    nm = rng.integers(0, 3, (n_pairs, 2))
    recs = []
    for k in range(n_pairs):
        p1 = int(starts[k])
        p2 = p1 + int(isz[k]) - 150
        name = "r%08d" % k
        for mate, (p, mp, rev) in enumerate(((p1, p2, False), (p2, p1, True))):
            flag = 0x1 | 0x2 | (0x40 if mate == 0 else 0x80) | (0x10 if rev else 0x20)
            cigar = "150M" if not split[k] else ("100M50S" if mate == 0 else "50S100M")
            recs.append(dict(name=name, flag=flag, tid=0, pos=p, mapq=mqs[int(mq[k, mate])], cigar=cigar, mtid=0, mpos=mp,
                             tlen=(int(isz[k]) if mate == 0 else -int(isz[k])), tags=[("NM", "C", int(nm[k, mate])), ("RG", "Z", "rg")],
                             seq4=seqs[int(pick[k, mate, 0])], qual=quals[int(pick[k, mate, 1])]))
    recs.sort(key=lambda r: r["pos"])
    bw.write_bam(path, header, [("1", genome)], recs, block_bytes=65280)
    hist = {str(k): int(1000 * np.exp(-((k - 400) / 85.0) ** 2)) + 1 for k in range(160, 900)}
    info = {sample: {"mapped": len(recs), "unmapped": 0, "bam": path, "sample_name": sample, "libraryArray": [
        {"library_name": "lib", "readgroups": ["rg"], "read_length": 150, "histogram": hist, "mean": 400.0, "sd": 60.0, "prevalence": 1.0}]}}
    sites = []
    for pos in range(20_000, genome - 20_000, spacing):
        L = int(rng.integers(500, 3000))
        sites.append({"id": "d%d" % pos, "svtype": "DEL", "var_length": L,
                      "A": {"chrom": "1", "pos": pos, "ci": [-10, 10], "is_reverse": False},
                      "B": {"chrom": "1", "pos": pos + L + 1, "ci": [-10, 10], "is_reverse": True}})
    return info, sites, len(recs)


def real_data_leg(device: int) -> dict:
    """The f-rows end to end on real BAM bytes, reader="native", geometry="device": BGZF inflate + fetch +
    fragment summaries in C++ threads (svt_bam_summarise), 128-byte summaries over PCIe, the geometry kernel, the genotype
    pass, result records back, sample columns formatted (svt_format_results).  Replaces svtyper/classic.py:54-100,
    singlesample.py:187-205 + the per-variant loop.  Two inputs: the reference's fixture BAM (211 sites, repeated: every
    block is in the page cache and the same blocks are inflated again and again) and a bounded WGS-like BAM written here
    (every site touches blocks nobody has inflated before)."""
    import io
    import tempfile
    from svtyper_amd import bam as pybam, hip, library, pipeline, singlesample
    from svtyper_amd import native_reads as nr
    from svtyper_amd import evidence as ev
    from svtyper_amd.vcf import Variant, Vcf

    data = os.path.join(ROOT, "tests", "data")
    out = {"what": "reader=native, the geometry predicates in the reader's threads: svt_bam_evidence (inflate + fetch + fragment assembly + "
                   "svt_geometry_math.h -> 16-byte evidence records, C++ threads) -> H2D (svt_batch_create) -> svt_stream_kernel -> D2H -> "
                   "svt_format_results; stage times in ms, whole-run rates in sites/s.  `device_geometry`: the same sites with 128-byte fragment "
                   "summaries over PCIe and svt_geometry_kernel on the device (svt_bam_summarise -> svt_batch_create_from_fragments)",
           "bytes_per_fragment_over_pcie": 16, "bytes_per_fragment_over_pcie_device_geometry": 128, "canonical_record_bytes": 16}

    class Timed:
        """the HIP engine with a stopwatch on every stage behind the reader"""
        supports_site_qual = False

        def __init__(self):
            self.t = {"create_h2d_geometry": 0.0, "pass": 0.0, "results_d2h": 0.0}
            self.fragments = self.units = self.h2d_bytes = 0

        def _timed(self, make, n_fragments, n_units, h2d_bytes):
            t0 = time.perf_counter()
            d = make()
            t1 = time.perf_counter()
            d.genotype(sync=True)
            t2 = time.perf_counter()
            r = d.results()
            t3 = time.perf_counter()
            d.close()
            self.t["create_h2d_geometry"] += t1 - t0
            self.t["pass"] += t2 - t1
            self.t["results_d2h"] += t3 - t2
            self.fragments += n_fragments
            self.units += n_units
            self.h2d_bytes += h2d_bytes
            return hip.host_sq(r)

        def __call__(self, batch, flags=0, site_qual=None):          # geometry in the reader: canonical 16-byte records
            return self._timed(lambda: hip.DeviceBatch(batch, device, flags), batch.n_records, batch.n_units,
                               16 * batch.n_records + 24 * batch.n_units)

        def genotype_fragments(self, fb, flags=0, site_qual=None):  # geometry on the device: 128-byte summaries
            return self._timed(lambda: hip.DeviceBatch.from_fragments(fb, device, flags), fb.n_fragments, fb.n_units,
                               128 * fb.n_fragments + 56 * fb.n_units)

    def run(sample, nbam, header_vcf, lines, repeat, threads=0, geometry="reader"):
        """One block of variant lines through the stages the drivers' bulk route is made of, one after the other (driver_sso /
        driver_classic_8bam below are the same stages overlapped by blocks): svt_vcf_parse -> site arrays per sample -> the
        reader -> device -> svt_vcf_emit."""
        from svtyper_amd import bulk_vcf
        text = "".join(lines * repeat).encode()
        fields = sorted(pipeline.SVTYPER_FORMAT_KEYS, key=lambda k: header_vcf.format_rank[k])
        best = None
        for _ in range(2):     # (the first run pays the pooled device buffers; keep the better one)
            eng = Timed()
            coll = pipeline.NativeUnitCollector([sample], [nbam], 1.0, 1.0, 20, nr.COUNT_SSO, 1000, n_threads=threads, geometry=geometry)
            parser = bulk_vcf.VcfParser(header_vcf, 1e10, False, True)
            t0 = time.perf_counter()
            chunk, used = parser.parse(text)
            assert used == len(text) and not (chunk.line_kind == bulk_vcf.LINE_PYTHON).any()
            coll.add_site_arrays(chunk.sites)
            job = coll.take(eng, ev.FLAG_SSO_ASSOCIATION)
            t_prep = time.perf_counter()
            res = job()
            t1 = time.perf_counter()
            out_text, off = chunk.emit(res, 1, bulk_vcf.QUAL_SSO, fields, False, ":".join(fields))
            t2 = time.perf_counter()
            dev = sum(eng.t.values())
            n_sites = chunk.n_sites
            leg = {"sites": n_sites, "variant_lines": chunk.n_lines, "fragments": eng.fragments, "wall_ms": (t2 - t0) * 1e3,
                   "sites_per_s": n_sites / (t2 - t0),
                   "stage_ms": {"vcf_parse_and_site_arrays": (t_prep - t0) * 1e3, "inflate_fetch_summarise_host": (t1 - t_prep - dev) * 1e3,
                                "h2d_plus_geometry_kernel": eng.t["create_h2d_geometry"] * 1e3, "genotype_pass": eng.t["pass"] * 1e3,
                                "results_d2h": eng.t["results_d2h"] * 1e3, "vcf_emit_lines": (t2 - t1) * 1e3},
                   "geometry": geometry, "h2d_bytes": int(eng.h2d_bytes), "fragments_per_site": eng.fragments / max(1, n_sites),
                   "gt_histogram": {str(k): int(v) for k, v in zip(*np.unique(res.gt, return_counts=True))},
                   "lines_out": out_text.count(b"\n"), "vcf_bytes_out": len(out_text)}
            chunk.close()
            if best is None or leg["wall_ms"] < best["wall_ms"]:
                best = leg
        return best

    # ---- (1) the reference's fixture: breakpoints of tests/data/example.vcf, the driver itself first (byte check), then x R
    in_vcf, bam_path = os.path.join(data, "example.vcf"), os.path.join(data, "NA12878.target_loci.sorted.bam")
    lib_json = os.path.join(data, "NA12878.bam.json")
    t0 = time.perf_counter()
    buf = io.StringIO()
    with open(in_vcf) as inf, open(os.devnull, "w") as null:
        old, sys.stderr = sys.stderr, null
        try:
            singlesample.sso_genotype(bam_path, inf, buf, 20, 1, 1, 1000000, lib_json, False, None, False, 1000, 1e10, None, 1000,
                                      geometry="device", reader="native")
        finally:
            sys.stderr = old
    drv = time.perf_counter() - t0
    strip = lambda text: [l for l in text.splitlines() if not l.startswith("##fileDate")]
    with open(os.path.join(data, "example.gt.vcf")) as f:
        same = strip(buf.getvalue()) == strip(f.read())
    out["fixture_driver"] = {"what": "singlesample.sso_genotype(reader='native', geometry='device') over tests/data (211 breakpoints), VCF in -> VCF out",
                             "wall_ms": drv * 1e3, "output_equals_example_gt_vcf": bool(same)}
    vcf = Vcf()
    bps = []
    with open(in_vcf) as f:
        lines = f.readlines()
    vcf.add_header([l for l in lines if l.startswith("##")])
    for line in lines:
        if line.startswith("#"):
            continue
        v = Variant(line.rstrip().split("\t"), vcf)
        if not v.has_svtype() or not v.is_valid_svtype():
            continue
        bp = vcf.get_variant_breakpoints(v, 1e10)
        if bp is not None:
            bps.append(bp)
    with open(lib_json) as f:
        sample = library.Sample.from_lib_info(pybam.AlignmentFile(bam_path), json.load(f), 1e-3)
    nbam = nr.NativeBam(bam_path)
    vcf.add_custom_svtyper_headers()
    vcf.add_sample(sample.name)
    body_lines = [l for l in lines if not l.startswith("#")]
    out["fixture_x100"] = dict(run(sample, nbam, vcf, body_lines, 100), what="the fixture's %d variant lines (%d breakpoints) x 100 (cached blocks, repeated sites)" % (len(body_lines), len(bps)))
    dg = run(sample, nbam, vcf, body_lines, 100, geometry="device")
    out["fixture_x100"]["device_geometry"] = {k: dg[k] for k in ("wall_ms", "sites_per_s", "stage_ms", "h2d_bytes") if k in dg}
    out["fixture_x100"]["device_geometry"]["same_genotypes"] = dg["gt_histogram"] == out["fixture_x100"]["gt_histogram"]
    nbam.close()
    # ---- (2) a bounded WGS-like BAM: 1.2 Mbp at 30x, a DEL every 4 kb
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "wgs_like.bam")
        t0 = time.perf_counter()
        info, sites, n_rec = _wgs_like_bam(path)
        wrote = time.perf_counter() - t0
        sample = library.Sample.from_lib_info(pybam.AlignmentFile(path), info, 1e-3)
        nbam = nr.NativeBam(path)
        wvcf = Vcf()
        wvcf.add_header([l for l in lines if l.startswith("##")])
        wvcf.add_custom_svtyper_headers()
        wvcf.add_sample(sample.name)
        wlines = ["1\t%d\t%s\tN\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-%d;END=%d;STR=+-:8;CIPOS=-10,10;CIEND=-10,10;SU=8;PE=6;SR=2\n"
                  % (bp["A"]["pos"], bp["id"], bp["var_length"], bp["A"]["pos"] + bp["var_length"]) for bp in sites]
        dg = run(sample, nbam, wvcf, wlines, 1, geometry="device")
        out["wgs_like_30x"] = dict(run(sample, nbam, wvcf, wlines, 1), bam_records=n_rec, bam_bytes=os.path.getsize(path), bam_written_s=wrote,
                                   what="1.2 Mbp at 30x (150-bp pairs, random bases, binned qualities), %d DEL sites 4 kb apart: every "
                                        "site inflates blocks of its own" % len(sites))
        out["wgs_like_30x"]["device_geometry"] = {k: dg[k] for k in ("wall_ms", "sites_per_s", "stage_ms", "h2d_bytes") if k in dg}
        out["wgs_like_30x"]["device_geometry"]["same_genotypes"] = dg["gt_histogram"] == out["wgs_like_30x"]["gt_histogram"]
        nbam.close()
    return out


def driver_legs() -> dict:
    """The PUBLIC drivers end to end, called with the reference's own positional arguments (svtyper/singlesample.py:764-778,
    classic.py:107-120; no `reader=`, no `engine=`): VCF text in -> VCF text out through the bulk VCF route (svt_vcf_parse ->
    svt_bam_evidence -> the device pass -> svt_vcf_emit).  `driver_sso`: the fixture's 212 variant lines x 100 through
    sso_genotype; `driver_classic_8bam`: 8 WGS-like BAMs (300 kbp at 30x each, a sample of its own) x 10 500 DEL lines through
    sv_genotype = 84 000 (site, sample) units with QUAL over the samples.  Best of three walls after one warm-up call; the
    output is byte-compared with the per-line route's (SVT_BULK_VCF=0: one Variant object per line)."""
    import io
    import tempfile
    from svtyper_amd import classic, singlesample

    class Sink(io.StringIO):
        def close(self):
            pass

    def quiet(fn):
        with open(os.devnull, "w") as null:
            old, sys.stderr = sys.stderr, null
            try:
                return fn()
            finally:
                sys.stderr = old

    def timed(call, reps=3):
        quiet(call)
        best, text = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            text = quiet(call)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, text

    def per_line(call):
        os.environ["SVT_BULK_VCF"] = "0"
        try:
            t0 = time.perf_counter()
            text = quiet(call)
            return time.perf_counter() - t0, text
        finally:
            del os.environ["SVT_BULK_VCF"]

    strip = lambda text: [l for l in text.split("\n") if not l.startswith("##fileDate")]
    out = {}
    data = os.path.join(ROOT, "tests", "data")
    with open(os.path.join(data, "example.vcf")) as f:
        lines = f.readlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    text_in = "".join(head) + "".join(body * 100)
    bam_path, lib_json = os.path.join(data, "NA12878.target_loci.sorted.bam"), os.path.join(data, "NA12878.bam.json")

    laps = {}

    def sso():
        sink = Sink()
        laps.clear()
        singlesample.sso_genotype(bam_path, io.StringIO(text_in), sink, 20, 1, 1, 1000000, lib_json, False, None, False, 1000, 1e10, None, 1000,
                                  stats=laps)
        return sink.getvalue()

    def host_stages():
        """the caller's thread by stage (ms; pipeline.BulkFeeder.laps of the last call): what replaced `site_arrays_python` + `format_columns_host`"""
        return {"route": laps.get("route"), "blocks": laps.get("blocks"), "lines_handed_back_to_python": laps.get("lines_handed_back"),
                "vcf_parse_ms": laps.get("vcf_parse_s", 0.0) * 1e3, "site_arrays_ms": laps.get("site_arrays_s", 0.0) * 1e3,
                "vcf_emit_ms": laps.get("vcf_emit_s", 0.0) * 1e3, "decode_write_ms": laps.get("decode_write_s", 0.0) * 1e3}
    wall, text = timed(sso)
    sso_stages = host_stages()
    pl_wall, pl_text = per_line(sso)
    with open(os.path.join(data, "example.gt.vcf")) as f:
        want = [l for l in strip(f.read()) if l and not l.startswith("#")]
    got = [l for l in strip(text) if l and not l.startswith("#")]
    n = len(body) * 100
    out["driver_sso"] = {"what": "singlesample.sso_genotype(<the reference's 15 positional arguments>) over tests/data/example.vcf's %d variant lines x 100" % len(body),
                         "variant_lines": n, "wall_ms": wall * 1e3, "sites_per_s": n / wall, "vcf_bytes_out": len(text),
                         "every_repeat_equals_example_gt_vcf": bool(got == want * 100), "host_stage_ms": sso_stages,
                         "per_line_route": {"wall_ms": pl_wall * 1e3, "sites_per_s": n / pl_wall, "same_bytes": strip(pl_text) == strip(text)}}
    with tempfile.TemporaryDirectory() as tmp:
        paths, info, sites = [], {}, None
        for k in range(8):
            path = os.path.join(tmp, "s%d.bam" % k)
            inf, sites, _ = _wgs_like_bam(path, genome=300_000, seed=40 + k, sample="smp%d" % k)
            info.update(inf)
            paths.append(path)
        libs = os.path.join(tmp, "libs.json")
        with open(libs, "w") as f:
            json.dump(info, f)
        vlines = ["1\t%d\t%s\tN\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-%d;END=%d;STR=+-:8;CIPOS=-10,10;CIEND=-10,10;SU=8;PE=6;SR=2\n"
                  % (bp["A"]["pos"], bp["id"], bp["var_length"], bp["A"]["pos"] + bp["var_length"]) for bp in sites]
        reps = -(-10_500 // len(vlines))
        vtext = "".join(l for l in head if l.startswith("##")) + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "".join(vlines * reps)

        def joint():
            sink = Sink()
            laps.clear()
            classic.sv_genotype(",".join(paths), io.StringIO(vtext), sink, 20, 1, 1, 1000000, libs, False, None, None, False, None, 1e10, stats=laps)
            return sink.getvalue()
        wall, text = timed(joint)
        joint_stages = host_stages()
        pl_wall, pl_text = per_line(joint)
        n = len(vlines) * reps
        called = sum(1 for l in text.split("\n") if l and not l.startswith("#") and ("\t0/0:" in l or "\t0/1:" in l or "\t1/1:" in l))
        out["driver_classic_8bam"] = {"what": "classic.sv_genotype(<the reference's 14 positional arguments>), 8 BAMs (300 kbp at 30x each) x %d DEL lines" % n,
                                      "variant_lines": n, "samples": 8, "units": n * 8, "wall_ms": wall * 1e3, "sites_per_s": n / wall,
                                      "units_per_s": n * 8 / wall, "vcf_bytes_out": len(text), "lines_with_a_called_genotype": called, "host_stage_ms": joint_stages,
                                      "per_line_route": {"wall_ms": pl_wall * 1e3, "units_per_s": n * 8 / pl_wall, "same_bytes": strip(pl_text) == strip(text)}}
    return out


def main():
    json_out = claim_stdout()
    # the host driver only supports dmabuf IPC: RCCL across processes fails without this (already exported on the GPU boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (100 passes of ~0.29 ms: the fixed costs of the timed region -- 20 launches' enqueue, the wake-up after the last one, with N
    # ranks two barriers -- are ~0.13 ms + the barriers, 2 % of a 20-step region and 0.4 % of this one)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=None,
                    help="breakpoints per GPU (weak) or in total (strong) [1 000 000; c2_del_100k: 100 000]")
    ap.add_argument("--workload", default="c3_mixed_1m", choices=["c3_mixed_1m", "c2_del_100k", "c5_multisample"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--sso", action="store_true", help="singlesample.py floating-point association")
    ap.add_argument("--result-bytes", type=int, default=96, choices=[96, 128],
                    help="device result record: 96 = SVT_FLAG_RESULT96, the record of SURVEY 8(d) plus the index of its unit, written in "
                         "the order the kernel finishes the units (a wave's 64 records = 6 KB of whole lines; the host puts every record "
                         "where its tag says and restores the counts that follow from the tallies: the same svt_result records); "
                         "128 = svt_result in unit order, one cache line per unit (`result128` leg) [96]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline only (N=1: no sso / c5 / shard / one_shot / packed / large legs)")
    ap.add_argument("--legs", default="all", help="comma list of the extra N=1 legs to run: " + ",".join(LEGS) + " [all]")
    ap.add_argument("--no-dense-leg", action="store_true", help="(kept for old command lines) = --no-extra-legs")
    ap.add_argument("--large-units", type=int, default=4_000_000)
    ap.add_argument("--c5-units", type=int, default=None,
                    help="(site, sample) units of the c5_multisample leg [2 x --units: at the default that is configs[4]'s own per-GPU "
                         "share, 500 k sites x 32 samples over 8 GPUs = 62 500 sites x 32 = 2 M units x ~100 records]")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--tune-placement", default="32,8",
                    help="svt_batch_tune_placement before the timed region (resident legs): result,record candidates; 0,0 = none [32,8]")
    ap.add_argument("--spinup-ms", type=float, default=SPINUP_MS,
                    help="untimed passes for this many ms before the warm-up steps (device clocks; 0 = none)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the gather even with one rank")
    ap.add_argument("--no-dist-legs", action="store_true",
                    help="N > 1 (or --force-dist): the headline only -- no `strong` (configs[3]) / `c5` (configs[4]) legs, no pipelined steady state")
    ap.add_argument("--c5-units-per-rank", type=int, default=None,
                    help="(site, sample) units per rank of the N-rank `c5` leg [configs[4] literally: 16 M / N, at most 2 x --units]")
    args = ap.parse_args()
    if args.no_dense_leg:
        args.no_extra_legs = True
    legs = set() if args.no_extra_legs else set(LEGS) if args.legs == "all" else set(x for x in args.legs.split(",") if x)
    if legs - set(LEGS):
        sys.exit("unknown --legs entries: %s" % sorted(legs - set(LEGS)))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.units is None:
        args.units = 100_000 if args.workload == "c2_del_100k" else 1_000_000
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    group = N_SAMPLES_C5 if args.workload == "c5_multisample" else 1

    # generate on the host BEFORE importing torch (fork-safe, and no GPU context in the workers)
    n_cpu = usable_cpus()
    workers = max(1, n_cpu // max(1, min(world, 8)))
    t0 = time.time()
    if args.scaling == "strong":
        # every rank builds the SAME total workload (same seeds) and keeps its shard
        from svtyper_amd import distributed as D
        total = generate(args.workload, args.units, 0, workers)
        bounds = D.shard_bounds(total.rec_offset, world, group)
        lo, hi = bounds[rank]
        batch = total.slice(lo, hi)
        counts = [b[1] - b[0] for b in bounds]
        total_units = total.n_units
        if rank != 0 or world == 1:
            del total        # rank 0 keeps the whole workload: after the gather it runs it alone and compares the bytes
            total = None
    else:
        total = None
        batch = generate(args.workload, args.units, rank, workers)
        counts = [batch.n_units] * world
        total_units = batch.n_units * world
    gen_s = time.time() - t0
    if world != 1 or args.workload != "c3_mixed_1m" or args.scaling != "weak":
        legs = set()
    more = c5_batch = c5_sample_major = None
    if "c5" in legs:
        # the configs[4] shape at ITS per-GPU size (500 k sites x 32 samples over 8 GPUs = 62 500 sites x 32 = 2 M units of
        # ~100 records, 3.2 GB): sites x 32 samples with per-sample libraries (svt_unit.libs hints), generated sample-major --
        # the order a producer that reads BAM by BAM emits -- and, for the comparison launches, site-major as well
        c5_n = args.c5_units if args.c5_units else 2 * batch.n_units
        if "c5x" in legs:
            c5_batch, c5_sample_major = generate("c5_multisample", c5_n, rank, workers, layout="both")
        else:
            c5_sample_major = generate("c5_multisample", c5_n, rank, workers, layout="sample")
    # the N-rank legs (one invocation answers BASELINE.json: configs[3] = the SAME --units workload cut N ways, configs[4] = 500 k
    # sites x 32 samples over the ranks), generated like everything else before a GPU context exists
    dist_legs_on = (world > 1 or args.force_dist) and not args.no_dist_legs and args.workload == "c3_mixed_1m" and args.scaling == "weak"
    strong_shard = strong_bounds = c5_shard = None
    if dist_legs_on:
        t0 = time.time()
        strong_shard, strong_bounds = generate_shard(args.workload, args.units, world, rank, workers)
        c5_per_rank = args.c5_units_per_rank if args.c5_units_per_rank else min(2 * args.units, 16_000_000 // world)
        c5_per_rank = max(N_SAMPLES_C5, c5_per_rank // N_SAMPLES_C5 * N_SAMPLES_C5)
        c5_shard = generate("c5_multisample", c5_per_rank, rank, workers, layout="sample")
        gen_s += time.time() - t0
    if "large" in legs and args.large_units > batch.n_units:
        # the rest of the `large_batch` leg's workload (also generated before any GPU context exists)
        more = generate(args.workload, args.large_units - batch.n_units, rank, workers,
                        first_chunk=(batch.n_units + 49_999) // 50_000)

    import torch
    import torch.distributed as dist
    from svtyper_amd import evidence as ev
    from svtyper_amd import hip
    from svtyper_amd import synth

    hip.load()
    n_dev = hip.device_count()
    if n_dev < 1:
        sys.exit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    # One rank per GPU over RCCL.  On a box with fewer devices than ranks (the builder's and the test suite's one-GPU
    # lease) the ranks SHARE the devices and the gather goes over gloo from host copies -- the rule of
    # tests/test_multi_device.py -- so that every line of the N > 1 branch runs before it meets an 8-GPU node; such a line
    # says `"shared_devices": true` and is not a scaling measurement.
    shared_devices = n_dev < world
    if shared_devices:
        local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    backend = "gloo" if shared_devices else "nccl"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_device = "cuda" if backend == "nccl" else "cpu"   # where the collectives' tensors live

    r96 = ev.FLAG_RESULT96 if args.result_bytes == 96 else 0
    sso = (ev.FLAG_SSO_ASSOCIATION if args.sso else 0) | r96      # (every leg's create flags carry the record form)
    flags = sso
    t0 = time.time()
    dbatch = hip.DeviceBatch(batch, device=local_rank, flags=flags)
    upload_s = time.time() - t0
    n = batch.n_units
    alg_bytes, resident_bytes = dbatch.bytes()
    layout_name = dbatch.layout_name()

    tune_res, tune_rec = (int(x) for x in args.tune_placement.split(","))

    def tune(d):
        """svt_batch_tune_placement (setup, not timed): the pass over a handful of freshly allocated candidates for the result
        buffer and the record buffer, the fastest kept -- where the VRAM manager puts the two moves the pass by up to 8 %"""
        if not (tune_res or tune_rec):
            return None
        w0 = time.perf_counter()
        r = d.tune_placement(tune_res, tune_rec)
        r["wall_ms"] = (time.perf_counter() - w0) * 1e3
        # the audition is 0.3-0.4 s of uninterrupted passes, after which the device runs for a while at a level 4-7 % slower (1 M
        # units: groups of 20 passes alternate 0.289 / 0.300 ms, a steady 0.287 again after half a second of idling; 4 M units:
        # 1.20 instead of 1.127 ms through 1.5 s of idling and spin-ups, 1.127 again after two more seconds,
        # profiles/r04_placement_tuning.txt): let it idle before the spin-up and the timed steps, which are a burst of a few ms
        r["idle_after_s"] = max(2.0, 8.0 * r["wall_ms"] * 1e-3)
        time.sleep(r["idle_after_s"])
        return r

    # the same launches WITHOUT the spin-up and on the buffers svt_batch_create drew, for the record (a device coming out of
    # idle: DESIGN.md 5): one untimed pass, then `steps` passes between HIP events -- reported as `roofline.no_spinup_*`,
    # never as `value`
    dbatch.genotype(sync=True)
    cold_ms = dbatch.genotype_timed(args.steps) / args.steps
    tuned = tune(dbatch)

    # the batch's own result buffer as a torch tensor (zero-copy view): the final RCCL gather needs no extra copy, and the
    # result records stay where svt_batch_create put them (binding a tensor torch allocated costs 3-6 % of the pass: DESIGN.md 3.1)
    res_buf = dbatch.device_results_tensor()
    rec_bytes = dbatch.result_bytes()
    assert rec_bytes == args.result_bytes and res_buf.data_ptr() % 128 == 0 and res_buf.numel() >= n * rec_bytes
    cur = dbatch.result_slots() * rec_bytes        # (96-byte records: whole workgroups' worth of tagged slots, >= n)

    def barrier():
        if use_dist:
            dist.barrier()

    # the interpreter's collector stays out of the timed region (a collection of the heap is not part of a pass) -- switched
    # off HERE, in front of the spin-up: a full collection takes tens of milliseconds during which the device idles, and a
    # device that has idled that long runs the next passes ~10 % slower (this round's lines with the collection between the
    # warm-up and the timed steps: 0.317-0.325 ms where the audition had just measured 0.287-0.297)
    import gc
    gc.collect()
    gc.disable()
    # N ranks: meet BEFORE the spin-up.  The ranks get here at different times (generation, upload, audition), and the first
    # collective of a process sets the communicator up: a rank that waited for the others in the barrier of the timed region
    # itself would start its timed passes on a device that has idled -- the same ~10 % as above.  After this meeting the ranks
    # spin up together and the barrier in front of the timed steps returns at once.
    barrier()
    spun = spin_up(dbatch, args.spinup_ms)
    for _ in range(args.warmup):
        dbatch.genotype(sync=False)
    torch.cuda.synchronize()
    # which dispatches of the headline kernel the timed region is, counted from the process's first one: 1 (the pass after
    # create) + `steps` (the cold measurement) + the spin-up + the warm-up come first -- known exactly when no audition ran (its
    # launch count depends on the pass time).  tools/summarize_prof.py averages the kernel trace and the PMC passes over exactly these.
    first_timed_dispatch = (1 + args.steps + spun + args.warmup) if not tuned else None

    # ---- the timed region: EXACTLY `steps` passes, barrier + device sync on both sides
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # `steps` passes enqueued back to back on the batch stream, between two HIP events on that stream
    # (kern_ms: the dominant kernel's average launch duration over the timed region itself -- torch.cuda.Event
    # would only see torch's current stream)
    kern_ms = dbatch.genotype_timed(args.steps) / args.steps
    t_passes = time.perf_counter()
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # where the host spent the timed region: `steps` launches + the wait for the last one (the HIP events inside measure the
    # device's share, kernel_ms x steps), the device synchronisation, the barrier
    timed_region_host = {"launch_and_wait_ms": (t_passes - t0) * 1e3, "device_sync_ms": (t_sync - t_passes) * 1e3,
                         "barrier_ms": (elapsed - (t_sync - t0)) * 1e3, "device_ms_by_hip_events": kern_ms * args.steps,
                         "host_ms_outside_the_events": (elapsed * 1e3 - kern_ms * args.steps)}
    if use_dist:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms_max = float(t[0].item()), float(t[1].item())
    else:
        kern_ms_max = kern_ms

    # ---- the single RCCL gather of the result records onto rank 0
    gather = None
    if use_dist:
        from svtyper_amd import distributed as D
        barrier()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        # (ranks sharing a device: the records come down to the host first and travel over gloo)
        local_buf = res_buf[:cur] if backend == "nccl" else res_buf[:cur].cpu()
        if rec_bytes == 96:
            gathered, g_sizes = D.gather_tagged_records(local_buf, dst=0)
        else:
            gathered, g_sizes = D.gather_result_records(local_buf, counts, dst=0), [c * rec_bytes for c in counts]
        torch.cuda.synchronize()
        barrier()
        g_s = time.perf_counter() - g0
        if rank == 0:
            assert gathered.numel() == sum(g_sizes) >= sum(counts) * rec_bytes
        gather = {"bytes_per_rank": int(cur), "record_bytes": rec_bytes, "ms": g_s * 1e3,
                  "GB/s_into_root": sum(g_sizes[1:] or g_sizes) / g_s / 1e9,
                  "collective": "rccl gather" if backend == "nccl" else "gloo gather (ranks share %d device(s))" % n_dev,
                  "route": D.gather_route(), "backend": backend, "units_per_rank": counts,
                  # how many ranks the RCCL communicator of this run actually spanned (0: no RCCL in this run)
                  "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0}
        if rank == 0 and args.scaling == "strong" and total is not None:
            # the sharded job against the same workload on one rank: the gathered records must be the same bytes
            with hip.DeviceBatch(total, device=local_rank, flags=flags) as d_all:
                d_all.genotype(sync=True)
                alone = d_all.results().rec
            joined = D.results_from_tagged(gathered, g_sizes, counts) if rec_bytes == 96 else D.results_from_bytes(gathered)
            same = bool(np.array_equal(joined.rec, alone))
            del joined
            gather["equals_single_rank_pass"] = same
            assert same, "the gathered result records differ from the single-rank pass over the same workload"
            del alone
            total = None

    res_buf = None   # (the view keeps the batch alive; the legs below close and re-create it)
    got = dbatch.results() if rank == 0 else None
    # ---- the N-rank legs: every rank takes part, rank 0 keeps the dicts
    dist_out = {}
    if use_dist and dist_legs_on:
        try:
            # the headline's steady state: pass k+1 over the gather of batch k
            sizes = [0] * world
            mine = torch.tensor([cur], dtype=torch.int64, device=coll_device)
            every = [torch.zeros(1, dtype=torch.int64, device=coll_device) for _ in range(world)]
            dist.all_gather(every, mine)
            sizes = [int(x.item()) for x in every]
            batches = max(8, args.steps)
            barrier()
            p_s = pipelined_passes(dbatch, sizes, batches, backend)
            barrier()
            t = torch.tensor([p_s], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist_out["value_pipelined"] = total_units * batches / float(t[0].item())
            dist_out["value_pipelined_note"] = ("pass of batch k+1 over the RCCL gather of batch k, two result buffers per rank, %d batches, "
                                                "barrier on both sides, maximum over the ranks: units / s of pass + gather in steady state" % batches)
        except Exception as e:
            dist_out["value_pipelined"] = None
            dist_out["value_pipelined_error"] = repr(e)
        # configs[3]: the headline's own workload (rank 0's batch: same seeds) cut by the shard rule; rank 0 holds the whole of it
        # and its single-rank result records
        counts_s = [b[1] - b[0] for b in strong_bounds]
        with hip.DeviceBatch(strong_shard, device=local_rank, flags=flags) as ds:
            leg = dist_leg("BASELINE.json configs[3] literally: the %d-unit workload of the headline sharded over %d rank(s) by bytes "
                           "(distributed.shard_bounds), every rank one launch per step, ONE gather of the result records onto rank 0"
                           % (args.units, world), ds, counts_s, sum(counts_s), args.steps, args.warmup, rank, world, backend, coll_device, n_dev,
                           alone=got.rec if rank == 0 and batch.n_units == sum(counts_s) else None)
        if rank == 0:
            leg["speedup_vs_one_rank_pass"] = kern_ms / (leg["ms_per_step"]) if leg["ms_per_step"] else None
            leg["speedup_note"] = "the headline's kernel_ms (one rank, the whole workload) / this leg's ms_per_step (N ranks, passes alone)"
            dist_out["strong"] = leg
        strong_shard = None
        # configs[4]: sites x 32 samples, sample-major units with per-sample library windows, site-major result records; a site's
        # samples stay on one rank (group = 32), QUAL is local
        counts_c = [c5_shard.n_units] * world
        with hip.DeviceBatch(c5_shard, device=local_rank, flags=flags) as dc:
            leg = dist_leg("BASELINE.json configs[4]: %d sites x %d samples = %d (site, sample) units per rank x %d rank(s)%s, per-sample "
                           "libraries, sample-major units -> site-major records (svt_batch_result_order), QUAL on the device per rank"
                           % (c5_shard.n_units // N_SAMPLES_C5, N_SAMPLES_C5, c5_shard.n_units, world,
                              "" if c5_shard.n_units * world == 16_000_000 else " (configs[4] literally is 16 M units over 8 ranks = 2 M per rank)"),
                           dc, counts_c, c5_shard.n_units * world, max(5, args.steps // 2), args.warmup, rank, world, backend, coll_device, n_dev,
                           order=N_SAMPLES_C5, sites_per_unit=N_SAMPLES_C5)
            dc.genotype(sync=True)      # (the pipelined passes wrote into bound buffers: the records in the batch's own buffer again)
            q0 = time.perf_counter()
            qual = dc.site_qual(N_SAMPLES_C5)
            q_ms = (time.perf_counter() - q0) * 1e3
        if rank == 0:
            leg["site_qual_ms"] = q_ms
            leg["site_qual_sites"] = int(len(qual))
            leg["frac"] = (16 * c5_shard.n_records + 112 * c5_shard.n_units) / (leg["kernel_ms_per_rank"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
            dist_out["c5"] = leg
        c5_shard = None
    if rank == 0:
        value = total_units * args.steps / elapsed
        traffic_key = "c5_windows" if args.workload == "c5_multisample" else "stream_sso" if args.sso else "stream"
        roof = roofline_of(kern_ms, alg_bytes, traffic_key, n, batch.n_records)
        per_site = batch.n_records / max(1, n)
        workload = {
            "c3_mixed_1m": "BASELINE.json configs[2]: %d mixed DEL/DUP/INV breakpoints%s, 1 library (fixture insert-size "
                           "histogram in LDS), %.1f fragment records/site",
            "c5_multisample": "BASELINE.json configs[4] shape: %d (site, sample) units%s = sites x 32 samples, per-sample "
                              "libraries, %.1f fragment records/unit",
            "c2_del_100k": "BASELINE.json configs[1]: %d DEL breakpoints%s, 1 library, %.1f fragment records/site",
        }[args.workload] % (total_units if args.scaling == "strong" else n,
                            " in total, sharded over %d GPU(s)" % world if args.scaling == "strong" else " per GPU", per_site)
        out = {
            "metric": "breakpoints genotyped/sec",
            "value": value,
            "unit": "breakpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            # which number answers north_star's ">= 6 x at 8 GPUs" and what the other N-rank numbers are
            "scaling_answer": {
                "number": "value",
                "why": "north_star: the variants shard across the GPUs (no per-step collective) with a SINGLE gather at the END of the job, so the "
                       "job's rate over many passes is `value` (per-rank passes, barrier + device sync on both sides, maximum over the ranks) and "
                       "the gather is paid once -- reported separately under `gather`.  `value_with_gather` = one pass FOLLOWED BY its gather "
                       "every step: bound by the root taking in N - 1 shards over its xGMI links (DESIGN.md 6: about 1.3 x one GPU for any N >= 2), "
                       "it does not scale and is not the claim.  `value_pipelined` (in `strong` / `c5`) = pass of batch k + 1 over the gather of "
                       "batch k, the steady state of a job that does want every batch's records on one rank; `value_with_d2h_parallel` = every "
                       "rank bringing ITS records down over its own PCIe link instead (the sharded drivers format their share of the VCF locally "
                       "and gather text)",
            } if world > 1 else None,
            "config": {
                "workload": workload,
                "units_per_gpu": n,
                "records_per_gpu": batch.n_records,
                "total_units": total_units,
                "association": "sso" if args.sso else "classic",
                "device_result_record_bytes": rec_bytes,
                "device_result_records": ("SVT_FLAG_RESULT96: the 96-byte record of SURVEY 8(d) + the unit's index, in the order the kernel "
                                          "finishes the units; svt_batch_results returns svt_result[n] in unit order" if rec_bytes == 96
                                          else "svt_result, 128 bytes, unit order"),
                "step": "one launch of svt_stream_kernel over the canonical CSR records resident in HBM -> result records "
                        "in HBM (whole hot path; nothing pre-digested outside the timed region)",
                "device_layout": "the canonical CSR records as uploaded, streamed by the pass itself",
                "parallelism": "units sharded over %d GPU(s), no data-path collective per step" % world,
            },
            "roofline": dict(roof, **{
                "kernel": "svt_stream_kernel",
                "traffic_frac_of_peak": (roof["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if roof["traffic"] else None,
                "resident_bytes_per_launch": resident_bytes,
                "kernel_ms_max_over_ranks": kern_ms_max,
                "kernel_ms_note": "HIP events around the `steps` back-to-back launches of the timed region, divided by `steps`: "
                                  "includes the ~5-10 us between consecutive dispatches that rocprofv3's per-kernel duration leaves out; "
                                  "rocprofv3's per-kernel AVERAGE covers every call in the process (cold passes, the placement audition's "
                                  "candidates, spin-up, the untuned `placement` leg): the statistic of a kernel trace that corresponds to "
                                  "this number is the fastest run of `steps` consecutive dispatches (tools/summarize_prof.py prints both)",
                "no_spinup_kernel_ms": cold_ms, "no_spinup_frac": alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # the three numbers side by side: `frac` above = this run's timed steps (after svt_batch_tune_placement unless
                # --tune-placement 0,0, and after the spin-up); frac_cold = the same launches on the buffers svt_batch_create drew, device
                # out of idle; frac_untuned_median = median over six more fresh allocations without the audition (`placement` leg, N = 1)
                "frac_tuned": roof["frac"] if tuned else None,
                "frac_cold": alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_untuned_median": None,
                "timed_region_host": timed_region_host,
                "timed_region_dispatches": {"first": first_timed_dispatch, "count": args.steps,
                                            "from_end": 0 if (not legs and not use_dist and more is None) else None,
                                            "note": "`first`: 0-based index, among this process's dispatches of the headline kernel, of the timed region's "
                                                    "first launch (null after a placement audition, whose launch count is not fixed); `from_end`: dispatches "
                                                    "of that kernel BEHIND the timed region in this process (0 with --no-extra-legs: the timed region is the "
                                                    "last `count` dispatches -- how tools/summarize_prof.py finds it behind an audition; null: other legs follow)"},
                "placement_tuned": dict(tuned, what="svt_batch_tune_placement before the timed region (setup): the real pass over "
                                        "freshly allocated candidates for the result buffer and the record buffer, the fastest kept; "
                                        "before_ms / after_ms = the pass on the buffers svt_batch_create drew / on the kept ones. "
                                        "`roofline.placement` below: fresh allocations WITHOUT it") if tuned else None,
                "library_sha16": library_stamp(),
                "source_sha16": source_stamp(),
                "compiler": compiler_stamp(),
                "note": "`achieved` = ALGORITHMIC bytes (16 B per fragment record + 112 B per unit) over the time of the ONE "
                        "kernel that does all the work from the canonical input: <= peak by construction",
            }),
            "host": {"generate_s": gen_s, "first_create_s": upload_s},
            "spinup": {"ms": args.spinup_ms, "passes": spun,
                       "note": "untimed passes before the warm-up steps so that the timed steps run at a loaded GPU's clocks; "
                               "not steps, not warm-up steps (spin_up in bench.py)"},
        }
        out.update(dist_out)
        if gather:
            out["gather"] = gather
            out["rccl_ranks"] = gather["rccl_ranks"]
            out["shared_devices"] = bool(shared_devices)
            if shared_devices:
                out["shared_devices_note"] = ("%d ranks on %d device(s): the N > 1 code path end to end (shards, max-over-ranks timing, "
                                              "gather, byte equality), NOT a scaling measurement" % (world, n_dev))
            # the job as north_star states it -- every rank's pass, then ONE gather of the result records onto rank 0 --
            # per pass: `value` times the passes alone (the contract's timed region), this one adds the gather once per pass
            out["value_with_gather"] = total_units / (elapsed / args.steps + gather["ms"] * 1e-3)
            out["value_with_gather_note"] = "units / (ms_per_step + gather.ms): one pass followed by its RCCL gather"
        if args.workload == "c5_multisample":   # one breakpoint = one VCF site; a unit = (site, sample)
            out["sites_per_s"] = value / N_SAMPLES_C5
            out["units_per_s"] = value

        # the one-shot legs hand a page-locked svt_result[] over: with 128-byte device records the result DMA lands in it directly;
        # tagged 96-byte records would need a host pass to put them in order (measured: 35.2 instead of 31.5 ms per million units)
        os_flags = flags & ~ev.FLAG_RESULT96
        host_out = None
        if "one_shot" in legs:
            # ---- one shot, PCIe included: host arrays (pageable) -> svt_batch_create -> one pass -> result records
            # on the host, steady state (the first create pays the pinned ring and the pooled device buffers)
            host_out = hip.pinned_results(n)   # the caller's output array, page-locked (svt_pinned_alloc): D2H is one DMA
            try:
                dbatch.close()
                walls, parts = [], None
                for _ in range(3):
                    t0 = time.perf_counter()
                    d1 = hip.DeviceBatch(batch, device=local_rank, flags=os_flags)
                    t1 = time.perf_counter()
                    d1.genotype(sync=True)
                    t2 = time.perf_counter()
                    r1 = d1.results(out=host_out)
                    t3 = time.perf_counter()
                    d1.close()
                    if not walls or t3 - t0 < min(walls):
                        parts = (t1 - t0, t2 - t1, t3 - t2)
                    walls.append(t3 - t0)
                serial = min(walls)
                pipe = []
                for _ in range(3):     # the same through svt_genotype: upload || pass || download by unit ranges
                    t0 = time.perf_counter()
                    r1 = hip.genotype_batch(batch, device=local_rank, flags=os_flags, out=host_out)
                    pipe.append(time.perf_counter() - t0)
                best = min(pipe)
                out["one_shot"] = {
                    "what": "svt_genotype: host arrays in pageable memory -> result records in a page-locked output array; the "
                            "canonical CSR goes up through the pinned ring in 64 MB pieces, every piece's units are genotyped by "
                            "their own launch as soon as it has landed and their records come down on a third stream; best of 3. "
                            "`serial_*`: the same as svt_batch_create + pass + svt_batch_results one after the other",
                    "wall_ms": best * 1e3,
                    "serial_wall_ms": serial * 1e3,
                    "serial_create_ms": parts[0] * 1e3, "serial_pass_ms": parts[1] * 1e3, "serial_results_d2h_ms": parts[2] * 1e3,
                    "pcie_inclusive_breakpoints_per_s": n / best,
                    "h2d_bytes": int(16 * batch.n_records + 24 * n + 8), "d2h_bytes": int(128 * n),
                }
                assert np.array_equal(r1.rec, got.rec), "one-shot results differ from the resident batch's"
                dbatch = hip.DeviceBatch(batch, device=local_rank, flags=flags)
            except Exception as e:  # an extra leg must never break the bench line
                out["one_shot"] = {"error": repr(e)}

        if "packed" in legs:
            # ---- the same through PACKED evidence: what a host producer hands over when the bytes have to cross PCIe
            # (svt_pack_evidence: ~3 bytes per fragment record instead of 16; the pass is svt_packed_kernel on the slots)
            try:
                if host_out is None:
                    host_out = hip.pinned_results(n)
                pack_times = []
                for _ in range(6):          # (the first call pays the page-locked pool and the workers' arenas)
                    time.sleep(0.3)         # (the cgroup's CPU quota is per 100 ms: a burst right behind another one is throttled)
                    t0 = time.perf_counter()
                    packed = hip.PackedEvidence.try_pack(batch)
                    pack_times.append((time.perf_counter() - t0) * 1e3)
                    if packed is None:
                        break
                    if _ < 5:
                        packed.free()
                pack_ms = min(pack_times)
                pack_med = sorted(pack_times[1:] or pack_times)[len(pack_times[1:] or pack_times) // 2]
                back_to_back = []
                for _ in range(5 if packed is not None else 0):   # the encoder called back to back, no pauses (what a producer loop sees)
                    t0 = time.perf_counter()
                    p3 = hip.PackedEvidence.try_pack(batch)
                    back_to_back.append((time.perf_counter() - t0) * 1e3)
                    p3.free()
                if packed is None:
                    out["one_shot_packed"] = {"skipped": "this batch cannot be expressed as packed evidence (a histogram wider than 2047 bins, a geometry outside the format's range)"}
                else:
                    walls, parts = [], None
                    for _ in range(4):
                        t0 = time.perf_counter()
                        dp = hip.DeviceBatch.from_packed(packed, device=local_rank, flags=os_flags)
                        t1 = time.perf_counter()
                        dp.genotype(sync=True)
                        t2 = time.perf_counter()
                        rp = dp.results(out=host_out)
                        t3 = time.perf_counter()
                        if not walls or t3 - t0 < min(walls):
                            parts = (t1 - t0, t2 - t1, t3 - t2)
                        walls.append(t3 - t0)
                        if len(walls) < 4:
                            dp.close()
                    p_ms = time_passes(dp, args.steps)
                    dp.close()
                    serial = min(walls)
                    pipe = []
                    for _ in range(4):
                        t0 = time.perf_counter()
                        rp = hip.genotype_packed(packed, device=local_rank, flags=os_flags, out=host_out)
                        pipe.append(time.perf_counter() - t0)
                    best = min(pipe)
                    # the whole route from records in host memory as ONE call: svt_genotype_packed_from_records encodes the batch in
                    # ranges of whole units and uploads / genotypes / downloads every finished range while the host threads encode
                    # the next one (`serial`: svt_pack_evidence, then svt_genotype_packed, timed as one sequence -- round 3's route)
                    route, route_serial, route_b2b = [], [], []
                    for _ in range(6):
                        time.sleep(0.3)     # (the cgroup's CPU quota is per 100 ms: a burst right behind another one is throttled)
                        t0 = time.perf_counter()
                        rp = hip.genotype_packed_from_records(batch, device=local_rank, flags=os_flags, out=host_out)
                        route.append((time.perf_counter() - t0) * 1e3)
                    route_equal = bool(np.array_equal(rp.rec, got.rec))
                    for _ in range(5):      # the same called back to back, no pauses (what a producer loop sees under the CPU quota)
                        t0 = time.perf_counter()
                        rp = hip.genotype_packed_from_records(batch, device=local_rank, flags=os_flags, out=host_out)
                        route_b2b.append((time.perf_counter() - t0) * 1e3)
                    for _ in range(4):
                        time.sleep(0.3)
                        t0 = time.perf_counter()
                        p2 = hip.PackedEvidence.try_pack(batch)
                        rp = hip.genotype_packed(p2, device=local_rank, flags=os_flags, out=host_out)
                        route_serial.append((time.perf_counter() - t0) * 1e3)
                        p2.free()
                    out["one_shot_packed"] = {
                        "what": "svt_genotype_packed: packed slots (page-locked, written by svt_pack_evidence) -> result records in a "
                                "page-locked output array, upload || svt_packed_kernel || download by unit ranges, best of 4.  The "
                                "leg's headline is `from_records_*`: svt_genotype_packed_from_records over records that already exist in host "
                                "memory (best / median of 6; `from_records_serial_*`: svt_pack_evidence + svt_genotype_packed as one timed "
                                "sequence) -- the encoder's time belongs to the "
                                "route; compare with `one_shot.wall_ms`, the same records through the canonical upload.  `pack_ms`: "
                                "the encoder alone (best of 6, `pack_ms_median` over calls 2-6; host threads: %s); "
                                "`pcie_inclusive_breakpoints_per_s` alone is what a producer that emits slots directly would "
                                "see.  `serial_*`: svt_batch_create_packed + pass + svt_batch_results one after the other"
                                % (os.environ.get("SVT_PACK_THREADS") or "one per physical core of a socket, at most 64, for a call "
                                   "whose CPU time fits the cgroup's allowance of one accounting period, else the quota's %d" % n_cpu),
                        "from_records_wall_ms": min(route), "from_records_wall_ms_median": sorted(route)[len(route) // 2],
                        "from_records_breakpoints_per_s": n / (min(route) * 1e-3),
                        "from_records_breakpoints_per_s_median": n / (sorted(route)[len(route) // 2] * 1e-3),
                        "from_records_wall_ms_median_back_to_back": sorted(route_b2b)[len(route_b2b) // 2],
                        "from_records_serial_wall_ms": min(route_serial), "from_records_serial_wall_ms_median": sorted(route_serial)[len(route_serial) // 2],
                        "from_records_results_equal_headline": route_equal,
                        "from_records_what": "svt_genotype_packed_from_records: encode || upload || pass || download by unit ranges in one call",
                        "pack_ms_median": pack_med,
                        "pack_ms_median_back_to_back": sorted(back_to_back)[len(back_to_back) // 2],
                        "wall_ms": best * 1e3, "serial_wall_ms": serial * 1e3, "serial_create_ms": parts[0] * 1e3,
                        "serial_pass_ms": parts[1] * 1e3, "serial_results_d2h_ms": parts[2] * 1e3, "pack_ms": pack_ms,
                        "pack_inclusive_wall_ms": pack_ms + best * 1e3,
                        "pack_inclusive_breakpoints_per_s": n / (pack_ms * 1e-3 + best),
                        "pcie_inclusive_breakpoints_per_s": n / best,
                        "h2d_bytes": packed.nbytes, "d2h_bytes": int(128 * n),
                        "bytes_per_fragment_record": packed.nbytes / max(1, batch.n_records),
                        "resident_pass_ms": p_ms, "resident_breakpoints_per_s_pass_only": n / (p_ms * 1e-3),
                        "results_equal_headline": bool(np.array_equal(rp.rec, got.rec)),
                    }
                    packed.free()
            except Exception as e:
                out["one_shot_packed"] = {"error": repr(e)}

        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline: the C restatement (oracle/, a port of the reference's algorithm) on the
            # host cores over a bounded sample of the same workload, also used as parity check
            from oracle import c_oracle
            sample_n = n
            sample = batch.slice(0, sample_n)
            threads = min(c_oracle.max_threads(), n_cpu)   # more threads than the CPU quota only get throttled
            o_flags = sso & ev.FLAG_SSO_ASSOCIATION
            want = c_oracle.genotype_batch(sample, flags=o_flags, n_threads=threads)   # warm-up + parity reference
            t0 = time.perf_counter()
            reps = 0
            while True:
                c_oracle.genotype_batch(sample, flags=o_flags, n_threads=threads, out=want)
                reps += 1
                if time.perf_counter() - t0 >= args.cpu_seconds or reps >= 50:
                    break
            cpu_s = time.perf_counter() - t0
            n1 = min(sample_n, 200_000)               # the same restatement on one thread, bounded slice
            one = batch.slice(0, n1)
            c_oracle.genotype_batch(one, flags=o_flags, n_threads=1)
            t0 = time.perf_counter()
            c_oracle.genotype_batch(one, flags=o_flags, n_threads=1)
            one_thread = n1 / (time.perf_counter() - t0)
            c_port = {
                "value": sample_n * reps / cpu_s,
                "unit": "breakpoints/s",
                "cores": threads,
                "cpu_model": cpu_model(),
                "one_thread": one_thread,
                "kind": "port",
                "sample": "the workload's %d units x %d repetitions, oracle/svt_oracle.c "
                          "(OpenMP, %d threads = the host CPUs this process may use: %d visible, cgroup quota %d)"
                          % (sample_n, reps, threads, len(os.sched_getaffinity(0)), n_cpu),
            }
            # the closest stand-in for "the reference's own CPU path" that can run here: the pure-Python
            # restatement, 1 process and a multiprocessing.Pool over all cores with 1000-unit batches
            # (the structure of svtyper/singlesample.py:746-748), on bounded slices of the workload
            try:
                from oracle import py_oracle
                n1 = min(n, 6000)
                t0 = time.perf_counter()
                py_oracle.genotype_batch(batch.slice(0, n1), o_flags)
                one = n1 / (time.perf_counter() - t0)
                npool = min(n, 2000 * threads)
                t0 = time.perf_counter()
                py_oracle.genotype_batch_pool(batch.slice(0, npool), o_flags, processes=threads, batch_size=1000)
                pool = npool / (time.perf_counter() - t0)
                py = {"value": pool, "unit": "breakpoints/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
                      "one_process": one, "pool": pool,
                      "what": "SURVEY 8(d)(ii)'s denominator, the stand-in for `svtyper-sso --cores N` that can run on this box: the reference's own "
                              "structure (per-fragment object walk, log_choose loop, a multiprocessing.Pool with batch_size=1000: "
                              "svtyper/singlesample.py:246-473,723-751) restated in pure Python",
                      "sample": "oracle/py_oracle.py: the workload's first %d units through multiprocessing.Pool(%d), batch_size=1000 "
                                "(`value`); its first %d units in one process (`one_process`)" % (npool, threads, n1)}
            except Exception as e:  # never let a baseline break the bench line
                py = {"error": repr(e)}
            # parsed.cpu_baseline = the Python pool (what north_star's ">= 100 x svtyper-sso CPU" is judged against); the same algorithm in
            # C with OpenMP on the same cores -- a far stronger CPU line -- beside it as cpu_baseline_c
            out["cpu_baseline"] = py if "value" in py else dict(c_port, python_pool_error=py.get("error"))
            out["cpu_baseline_c"] = c_port
            # north_star: >= 100 x the svtyper-sso CPU path's breakpoints/s on one MI355X.  BASELINE.md publishes no number, so
            # `vs_baseline` stays null (the contract); the ratios against the two CPU restatements timed on THIS box's host cores:
            out["vs_cpu"] = {
                "python_restatement_pool": value / py["pool"] if py.get("pool") else None,
                "c_port": value / c_port["value"],
                "cores": threads,
                "note": "value / cpu_baseline.value (SURVEY 8d-ii's denominator: the reference's own structure -- a multiprocessing.Pool "
                        "of pure-Python workers, singlesample.py:723-751 -- restated in oracle/py_oracle.py) and value / cpu_baseline_c.value "
                        "(the same algorithm in C with OpenMP on the same cores); target >= 100",
            }
            ints_bad = int((got.counts[:sample_n] != want.counts).sum() + (got.gt[:sample_n] != want.gt).sum())
            out["parity"] = {
                "units_checked": sample_n,
                "integer_mismatches": ints_bad,
                "max_abs_dGL": float(np.max(np.abs(got.gl[:sample_n] - want.gl))),
                "max_abs_dSQ": float(np.max(np.abs(got.sq[:sample_n] - want.sq))),
            }

        if "sso" in legs and not args.sso:
            # ---- the same launch with the singlesample association of the split-read sums (svtyper/singlesample.py:246-276,367-372)
            try:
                with hip.DeviceBatch(batch, device=local_rank, flags=ev.FLAG_SSO_ASSOCIATION | r96) as ds:
                    ds.genotype(sync=True)
                    s_tuned = tune(ds)
                    s_ms = time_passes(ds, args.steps)
                    s_alg, _ = ds.bytes()
                out["sso"] = dict(roofline_of(s_ms, s_alg, "stream_sso", n, batch.n_records),
                                  what="the headline's workload and launch with SVT_FLAG_SSO_ASSOCIATION (svtyper-sso's summation order)",
                                  kernel="svt_stream_kernel<sso>", units=n, breakpoints_per_s=n / (s_ms * 1e-3), placement_tuned=s_tuned)
            except Exception as e:
                out["sso"] = {"error": repr(e)}

        if "place" in legs:
            # ---- where svt_batch_create's allocations land in HBM moves this pass by up to 8 % (a property of the physical blocks, not
            # of the offsets inside them, and on some boxes of no block at all: profiles/r04_placement_*.txt).  The headline above is ONE
            # draw; here are K more, each a batch made resident afresh while the earlier ones stay allocated.
            try:
                held, times = [], []
                for _ in range(6):
                    dk = hip.DeviceBatch(batch, device=local_rank, flags=flags)
                    held.append(dk)
                    dk.genotype(sync=True)
                    spin_up(dk, 15)
                    times.append(min(dk.genotype_timed(10) / 10 for _ in range(3)))
                for dk in held:
                    dk.close()
                hip.trim()
                ts = sorted(times)
                out["roofline"]["placement"] = {
                    "what": "the headline's pass over 6 more fresh allocations of the same batch (best of 3 x 10 launches each), in allocation order",
                    "kernel_ms": times, "min": ts[0], "median": ts[len(ts) // 2], "max": ts[-1],
                    "frac_min_median_max": [alg_bytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS for t in (ts[-1], ts[len(ts) // 2], ts[0])],
                    "spread_pct": (ts[-1] / ts[0] - 1.0) * 100.0, "headline_kernel_ms": kern_ms}
                out["roofline"]["frac_untuned_median"] = alg_bytes / (ts[len(ts) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception as e:
                out["roofline"]["placement"] = {"error": repr(e)}

        if "r96" in legs:
            # ---- the same launch with the other form of device result record (headline: 96-byte tagged records in the kernel's order
            # unless --result-bytes 128; here: the other one)
            other = 128 if r96 else 96
            key = "result%d" % other
            try:
                with hip.DeviceBatch(batch, device=local_rank, flags=(flags & ~ev.FLAG_RESULT96) | (0 if r96 else ev.FLAG_RESULT96)) as d9:
                    d9.genotype(sync=True)
                    same = bool(np.array_equal(d9.results().rec, got.rec))
                    r_ms = time_passes(d9, args.steps)
                    slots9 = d9.result_slots()
                out[key] = {"what": "the headline's launch writing %d-byte device result records (%s)" % (
                                other, "svt_result, unit order, one line per unit" if other == 128 else "SVT_FLAG_RESULT96: tagged, in the kernel's order"),
                            "kernel_ms": r_ms, "frac": alg_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "device_record_bytes": other,
                            "bytes_written_per_launch": other * slots9, "host_results_equal_headline": same}
            except Exception as e:
                out[key] = {"error": repr(e)}

        if c5_sample_major is not None:
            # ---- BASELINE.json configs[4] shape at its per-GPU size: (site, sample) units, 32 samples, per-sample libraries.
            # The producer hands the units over SAMPLE-MAJOR (it reads BAM by BAM; a sample's units, which share its library
            # window, are then contiguous in HBM) and svt_batch_result_order makes the pass write the records SITE-MAJOR, the
            # order QUAL, the shard rule and the VCF writer use.  `site_major_input`: the same units handed over site-major.
            try:
                sm_batch, c5_sample_major = c5_sample_major, None
                with hip.DeviceBatch(sm_batch, device=local_rank, flags=sso) as dc:
                    dc.result_order(N_SAMPLES_C5)
                    dc.genotype(sync=True)
                    c_tuned = tune(dc)
                    c_ms = time_passes(dc, args.steps)
                    c_alg, _ = dc.bytes()
                    c_mode = dc.table_mode()
                    c_res = dc.results().rec
                    c_qual = dc.site_qual(N_SAMPLES_C5)
                leg = roofline_of(c_ms, c_alg, "c5_windows", sm_batch.n_units, sm_batch.n_records)
                leg.update(what="configs[4] shape: %d sites x %d samples, %d libraries, every unit carries its sample's library window "
                                "(svt_unit.libs); units handed over sample-major, result records written site-major "
                                "(svt_batch_result_order); one launch of the library-window kernel"
                                % (sm_batch.n_units // N_SAMPLES_C5, N_SAMPLES_C5, len(sm_batch.libs)),
                           kernel="svt_stream_kernel<windows>", table_mode=c_mode, units=sm_batch.n_units, records=sm_batch.n_records,
                           placement_tuned=c_tuned,
                           units_per_s=sm_batch.n_units / (c_ms * 1e-3), sites_per_s=sm_batch.n_units / N_SAMPLES_C5 / (c_ms * 1e-3))
                out["c5_multisample"] = leg
                if "c5f" in legs:
                    # ---- configs[4] at ITS size on this one device: 16 M units = the leg's batch replicated along the sites (sample-major
                    # stays sample-major; the copies carry the same evidence, so the whole pass must repeat the 2 M-unit pass block by
                    # block): ~1.6 G records = 37 % of the 32-bit record index space, ~26 GB of HBM, ~31 k window chunks
                    try:
                        copies = max(1, min(8, 16_000_000 // sm_batch.n_units))     # (8 x the 2 M-unit leg = configs[4]; small --units: 8 x as well)
                        avail = 0.0
                        for line in open("/proc/meminfo"):
                            if line.startswith("MemAvailable:"):
                                avail = int(line.split()[1]) / 1e6
                        need_gb = (16 * sm_batch.n_records + 40 * sm_batch.n_units) * copies / 1e9 * 1.15 + 4
                        if avail and avail < need_gb:
                            raise MemoryError("host has %.0f GB available, the replicated batch needs %.0f" % (avail, need_gb))
                        t0 = time.perf_counter()
                        full = synth.replicate_sample_major(sm_batch, N_SAMPLES_C5, copies)
                        rep_s = time.perf_counter() - t0
                        t0 = time.perf_counter()
                        with hip.DeviceBatch(full, device=local_rank, flags=sso) as df:
                            create_s = time.perf_counter() - t0
                            df.result_order(N_SAMPLES_C5)
                            df.genotype(sync=True)
                            f_ms = time_passes(df, max(3, args.steps // 4))
                            f_alg, f_res = df.bytes()
                            t0 = time.perf_counter()
                            f_qual = df.site_qual(N_SAMPLES_C5)
                            q_ms = (time.perf_counter() - t0) * 1e3
                            f_rec = df.results().rec
                            f_slots = df.result_slots()
                        blocks = f_rec.reshape(copies, -1)
                        same = all(blocks[c].tobytes() == c_res.tobytes() for c in range(copies))
                        same_q = bool(np.array_equal(f_qual.reshape(copies, -1), np.broadcast_to(c_qual, (copies, len(c_qual)))))
                        out["c5_full"] = {
                            "what": "BASELINE.json configs[4] at its full size on ONE device: %d sites x %d samples = %d units, %d records (%.1f GB "
                                    "resident), sample-major units -> site-major tagged records, QUAL on the device; the batch is the "
                                    "c5_multisample leg's replicated %d x along the sites" % (
                                        full.n_units // N_SAMPLES_C5, N_SAMPLES_C5, full.n_units, full.n_records, f_res / 1e9, copies),
                            "units": full.n_units, "records": full.n_records, "kernel_ms": f_ms,
                            "frac": f_alg / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "units_per_s": full.n_units / (f_ms * 1e-3),
                            "sites_per_s": full.n_units / N_SAMPLES_C5 / (f_ms * 1e-3), "result_slots": f_slots,
                            "record_index_space_used": full.n_records / 2.0**32,
                            "site_qual_ms": q_ms, "replicate_host_s": rep_s, "create_s": create_s,
                            "every_site_block_equals_the_2M_unit_pass": bool(same), "site_qual_equals": same_q,
                        }
                        del full, f_rec, blocks, f_qual
                    except Exception as e:
                        out["c5_full"] = {"error": repr(e)}
                del sm_batch
                if c5_batch is None:
                    raise StopIteration
                with hip.DeviceBatch(c5_batch, device=local_rank, flags=sso) as dc:
                    dc.genotype(sync=True)
                    s_ms = time_passes(dc, args.steps)
                    same = bool(np.array_equal(dc.results().rec, c_res)) and bool(np.array_equal(dc.site_qual(N_SAMPLES_C5), c_qual))
                leg["site_major_input"] = {"kernel_ms": s_ms, "frac": c_alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "results_and_site_qual_equal": same}
                c_res_site = c_res
                del c_res
                # the same batch WITHOUT the hints: svt_batch_create reads every unit's library window off the uploaded
                # records (svt_window_scan_kernel, `create_scan_ms` = what that adds to the create) and the pass stages
                # windows as before; `general_tables`: SVT_FLAG_GENERAL_TABLES, every histogram look-up through L2
                nh_units = c5_batch.units.copy()
                nh_units["libs"] = 0
                nh = ev.EvidenceBatch(c5_batch.rec_offset, nh_units, c5_batch.records, c5_batch.libs, c5_batch.split_weight, c5_batch.disc_weight)
                t0 = time.perf_counter()
                with hip.DeviceBatch(nh, device=local_rank, flags=sso) as dn:
                    create_nh = time.perf_counter() - t0
                    dn.genotype(sync=True)
                    g_ms = time_passes(dn, max(3, args.steps // 2))
                    leg["hintless"] = {"table_mode": dn.table_mode(), "kernel_ms": g_ms, "frac": c_alg / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "results_equal": bool(np.array_equal(dn.results().rec, c_res_site)), "create_ms": create_nh * 1e3}
                t0 = time.perf_counter()
                with hip.DeviceBatch(nh, device=local_rank, flags=sso | ev.FLAG_GENERAL_TABLES) as dn:
                    create_gen = time.perf_counter() - t0
                    dn.genotype(sync=True)
                    g_ms = time_passes(dn, max(3, args.steps // 2))
                    leg["general_tables"] = {"table_mode": dn.table_mode(), "kernel_ms": g_ms, "frac": c_alg / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "create_ms": create_gen * 1e3}
                leg["hintless"]["create_scan_ms"] = (create_nh - create_gen) * 1e3
                # the one-shot of the same batch (svt_genotype, PCIe included), with and without the hints: both upload in one piece and
                # take the window kernel (the hint-less one after reading the windows off the records; it used to take the general mode)
                c5_out = hip.pinned_results(c5_batch.n_units)
                shots = {}
                for name, bb in (("hinted", c5_batch), ("hintless", nh)):
                    ts = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        r_os = hip.genotype_batch(bb, device=local_rank, flags=sso & ~ev.FLAG_RESULT96, out=c5_out)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    shots[name + "_wall_ms"] = min(ts)
                    shots[name + "_results_equal"] = bool(np.array_equal(r_os.rec, c_res_site))
                shots["hintless_over_hinted"] = shots["hintless_wall_ms"] / shots["hinted_wall_ms"]
                leg["one_shot"] = shots
                # the same batch as PACKED evidence (several libraries: library switches in the pair streams, the pass reads the
                # histogram tables through L2): the route from records in host memory in one call, and the pass alone over the
                # resident slots
                try:
                    ts = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        r_pk = hip.genotype_packed_from_records(c5_batch, device=local_rank, flags=sso & ~ev.FLAG_RESULT96, out=c5_out)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    pk = {"from_records_wall_ms": min(ts), "from_records_wall_ms_median": sorted(ts)[1],
                          "from_records_units_per_s": c5_batch.n_units / (min(ts) * 1e-3),
                          "results_equal": bool(np.array_equal(r_pk.rec, c_res_site)),
                          "over_canonical_one_shot": min(ts) / shots["hinted_wall_ms"]}
                    del r_pk
                    t0 = time.perf_counter()
                    with hip.PackedEvidence(c5_batch) as pe:
                        pk["encode_ms"] = (time.perf_counter() - t0) * 1e3
                        pk["bytes_per_record"] = pe.nbytes / max(c5_batch.n_records, 1)
                        with hip.DeviceBatch.from_packed(pe, device=local_rank, flags=sso) as dp:
                            dp.genotype(sync=True)
                            pk["pass_ms"] = time_passes(dp, max(3, args.steps // 2))
                            pk["pass_kernel"] = "svt_packed_kernel<several libraries>"
                    leg["packed"] = pk
                except Exception as e:
                    leg["packed"] = {"error": repr(e)}
                del c5_out, r_os
                del nh, nh_units, c_res_site
            except StopIteration:
                pass
            except Exception as e:
                out["c5_multisample"] = dict(out.get("c5_multisample", {}), error=repr(e))
            c5_batch = None

        if "shard" in legs:
            # ---- strong scaling, the part one GPU can measure: the headline workload cut by the 8-GPU shard rule
            # (svt_shard_bounds == distributed.shard_bounds), shard 0 timed alone = what every rank of configs[3] runs
            try:
                from svtyper_amd import distributed as D
                lo, hi = D.shard_bounds(batch.rec_offset, 8, 1)[0]
                sh = batch.slice(lo, hi)
                with hip.DeviceBatch(sh, device=local_rank, flags=flags) as dsb:
                    dsb.genotype(sync=True)
                    sh_ms = time_passes(dsb, max(args.steps, 20))
                    sh_alg, _ = dsb.bytes()
                    same = bool(np.array_equal(dsb.results().rec, got.rec[lo:hi]))
                out["shard_of_8"] = {
                    "what": "units [%d, %d) = shard 0 of the headline workload under the 8-GPU shard rule, one launch; "
                            "`speedup_vs_headline` is the per-GPU strong-scaling factor a rank of configs[3] can reach before the gather "
                            "(8 = ideal)" % (lo, hi),
                    "units": hi - lo, "records": sh.n_records, "kernel_ms": sh_ms,
                    "frac": sh_alg / (sh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "speedup_vs_headline": kern_ms / sh_ms,
                    "expected_speedup_model": "t(n) = t_fixed + n * t_unit fitted through this launch and the headline: t_fixed = %.4f ms"
                                              % max(0.0, (sh_ms * n - kern_ms * (hi - lo)) / max(1, n - (hi - lo))),
                    "results_equal_headline": same,
                }
            except Exception as e:
                out["shard_of_8"] = {"error": repr(e)}

        if "real" in legs:
            # ---- the rows either side of the path on real BAM bytes (f1 geometry on device, f3 native reader, f4 pipeline)
            try:
                out["real_data"] = real_data_leg(local_rank)
            except Exception as e:
                out["real_data"] = {"error": repr(e)}
            try:
                out["real_data"].update(driver_legs())
            except Exception as e:
                out["real_data"]["driver_legs_error"] = repr(e)

        if more is not None:
            # ---- the same step at 4 M units per GPU: 6.5 GB of records, far beyond the 256 MiB Infinity Cache
            try:
                dbatch.close()
                t0 = time.time()
                big = ev.concat_batches([batch, more])
                more = None
                with hip.DeviceBatch(big, device=local_rank, flags=flags) as db:
                    db.genotype(sync=True)
                    b_tuned = tune(db)
                    b_ms = time_passes(db, max(5, args.steps // 2))
                    b_alg, _ = db.bytes()
                    head = db.results().rec[:n]
                out["large_batch"] = {
                    "units": big.n_units, "records": big.n_records, "kernel_ms": b_ms, "placement_tuned": b_tuned,
                    "breakpoints_per_s": big.n_units / (b_ms * 1e-3),
                    "achieved_GBps": b_alg / (b_ms * 1e-3) / 1e9, "frac": b_alg / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "first_units_equal_headline": bool(np.array_equal(head, got.rec)),
                }
                del big
            except Exception as e:
                out["large_batch"] = {"error": repr(e)}
        print(json.dumps(out), file=json_out, flush=True)

    try:
        dbatch.close()
    except Exception:
        pass
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
