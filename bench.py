#!/usr/bin/env python
"""bench.py -- breakpoints genotyped per second on N x MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (evidence tally -> bayes_gt -> GT/GQ/SQ, one fused HIP
kernel launch) over one synthetic batch that is already resident in HBM.  Workload at every
N: BASELINE.json configs[2] -- 1 M mixed DEL/DUP/INV breakpoints, one library (the reference
fixture's empirical insert-size histogram, staged in LDS), ~100 fragment records (~200 reads)
per breakpoint -- PER GPU (weak scaling: independent units, no data-path collective inside a
step).  configs[1] (100 k sites = 160 MB) is not used for the headline because it fits the
256 MiB Infinity Cache and would not measure HBM.  After the timed region every rank's result
records are gathered onto rank 0 with ONE RCCL gather over xGMI (north_star: "a single RCCL
gather ... at the end"); its time is reported separately under "gather".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


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


def generate(name: str, n_units: int, rank: int, workers: int):
    """The synthetic workload of this rank (chunks generated in parallel host processes)."""
    from svtyper_amd import evidence as ev
    if name == "c5_multisample":
        import multiprocessing as mp
        from svtyper_amd import synth
        with mp.get_context("fork").Pool(min(max(1, workers), 32)) as pool:
            return synth.make_multisample(max(1, n_units // 32), 32, synth.BASE_SEED + 5 + 7919 * rank,
                                          pool_map=pool.map)
    chunk = 50_000
    jobs = [(name, min(chunk, n_units - i), i // chunk, rank) for i in range(0, n_units, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.map(_gen_chunk, jobs)
    else:
        parts = [_gen_chunk(j) for j in jobs]
    return parts[0] if len(parts) == 1 else ev.concat_batches(parts)


def claim_stdout():
    """The contract is ONE JSON line on stdout, but libraries below us write there too (RCCL prints its version
    banner on stdout through C stdio, flushed at exit).  Keep the real stdout for the JSON line and point file
    descriptor 1 at stderr for everybody else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    json_out = claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=None,
                    help="breakpoints per GPU [the workload's own size: 1 000 000; c2_del_100k: 100 000]")
    ap.add_argument("--workload", default="c3_mixed_1m", choices=["c3_mixed_1m", "c2_del_100k", "c5_multisample"])
    ap.add_argument("--sso", action="store_true", help="singlesample.py floating-point association")
    ap.add_argument("--layout", default="stream", choices=["stream", "short", "compact", "dense"],
                    help="stream (default): ONE kernel over the canonical CSR records as they lie in HBM; "
                         "short / compact / dense: the tiled layouts svt_batch_create builds once (re-run figures)")
    ap.add_argument("--dense", action="store_true", help="= --layout dense")
    ap.add_argument("--fixed-pair-entries", action="store_true", help="= --layout compact")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-leg", action="store_true",
                    help="skip the extra timing of the dense-record layout (N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the gather even with one rank")
    args = ap.parse_args()

    if args.dense:
        args.layout = "dense"
    if args.fixed_pair_entries:
        args.layout = "compact"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.units is None:
        args.units = 100_000 if args.workload == "c2_del_100k" else 1_000_000
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    # generate on the host BEFORE importing torch (fork-safe, and no GPU context in the workers)
    n_cpu = usable_cpus()
    t0 = time.time()
    batch = generate(args.workload, args.units, rank, max(1, n_cpu // max(1, min(world, 8))))
    gen_s = time.time() - t0

    import torch
    import torch.distributed as dist
    from svtyper_amd import evidence as ev
    from svtyper_amd import hip

    hip.load()
    if hip.device_count() <= local_rank:
        sys.exit("bench.py needs %d MI355X device(s); the HIP path has no CPU fallback" % (local_rank + 1))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    flags = ((ev.FLAG_SSO_ASSOCIATION if args.sso else 0)
             | {"stream": ev.FLAG_STREAM_LAYOUT, "short": 0, "compact": ev.FLAG_FIXED_PAIR_ENTRIES,
                "dense": ev.FLAG_DENSE_LAYOUT}[args.layout])
    t0 = time.time()
    dbatch = hip.DeviceBatch(batch, device=local_rank, flags=flags)
    upload_s = time.time() - t0
    if world == 1:
        # steady state of a chunked run: the second svt_batch_create finds the pinned ring, the device
        # scratch and the host work arrays of the first one (the first pays their allocation)
        dbatch.close()
        t0 = time.time()
        dbatch = hip.DeviceBatch(batch, device=local_rank, flags=flags)
        upload_steady_s = time.time() - t0
    else:
        upload_steady_s = upload_s
    n = batch.n_units
    alg_bytes, resident_bytes = dbatch.bytes()
    compact, table_mode = dbatch.layout()
    layout_name = dbatch.layout_name()

    # result records straight into a torch buffer (so the final RCCL gather needs no extra copy)
    res_buf = torch.zeros(max(n, 1) * ev.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    assert res_buf.data_ptr() % 128 == 0
    dbatch.bind_device_results(res_buf.data_ptr())
    cur = res_buf.numel()

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        dbatch.genotype(sync=False)
    torch.cuda.synchronize()

    # ---- the timed region: EXACTLY `steps` passes, barrier + device sync on both sides
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # `steps` passes enqueued back to back on the batch stream, between two HIP events on that stream
    kern_ms = dbatch.genotype_timed(args.steps) / args.steps
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # (kern_ms: the dominant kernel's average launch duration over the timed region itself, by HIP events on
    # the launch stream -- torch.cuda.Event would only see torch's current stream)

    # ---- the single RCCL gather of the result records onto rank 0
    gather = None
    if use_dist:
        from svtyper_amd import distributed as D
        barrier()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        gathered = D.gather_result_records(res_buf, [n] * world, dst=0)
        torch.cuda.synchronize()
        barrier()
        g_s = time.perf_counter() - g0
        if rank == 0:
            assert gathered.numel() == cur * world
        gather = {"bytes_per_rank": int(cur), "ms": g_s * 1e3,
                  "GB/s_into_root": cur * max(world - 1, 1) / g_s / 1e9, "collective": "rccl gather"}

    if rank == 0:
        got = dbatch.results()
        total_units = n * world
        value = total_units * args.steps / elapsed
        ach = alg_bytes / (kern_ms * 1e-3) / 1e9
        # HBM traffic of the same kernel on the same workload from the committed PMC passes
        # (tools/profile.sh -> profiles/hbm_traffic.json); rocprofv3 cannot wrap this process from inside
        traffic, traffic_note = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                tj = json.load(f)
            tj = tj[layout_name]
            if tj.get("units") == n and tj.get("records") == batch.n_records:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_note = "rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), profiles/hbm_traffic.json"
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "breakpoints genotyped/sec",
            "value": value,
            "unit": "breakpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2]: %d mixed DEL/DUP/INV breakpoints per GPU, 1 library "
                            "(fixture insert-size histogram in LDS), %.1f fragment records/site"
                            % (n, batch.n_records / max(1, n)) if args.workload == "c3_mixed_1m" else
                            ("BASELINE.json configs[4] shape: %d sites x 32 samples = %d units per GPU, %d libraries"
                             % (n // 32, n, len(batch.libs)) if args.workload == "c5_multisample" else
                             "BASELINE.json configs[1]: %d DEL breakpoints per GPU, 1 library" % n),
                "units_per_gpu": n,
                "records_per_gpu": batch.n_records,
                "association": "sso" if args.sso else "classic",
                "device_layout": {"dense": "dense 16-byte records, tiled once at svt_batch_create",
                                  "compact": "compact sparse 4-byte entry streams, re-encoded once at svt_batch_create",
                                  "short": "compact sparse entry streams, 2-byte pair entries for the common MAPQ pair, "
                                           "re-encoded once at svt_batch_create",
                                  "stream": "the canonical CSR records as uploaded, streamed by the pass itself"}[layout_name],
                "parallelism": "units sharded over %d GPU(s), no data-path collective per step" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "svt_stream_kernel" if layout_name == "stream" else "svt_genotype_kernel",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch",
                "traffic_source": traffic_note,
                "traffic_frac_of_peak": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "resident_bytes_per_launch": resident_bytes,
                "kernel_ms": kern_ms,
                "kernel_ms_note": "HIP events around the `steps` back-to-back launches of the timed region, divided by `steps`: "
                                  "includes the ~5-10 us between consecutive dispatches that rocprofv3's per-kernel duration leaves out",
                "note": ("`achieved` is ALGORITHMIC bytes (16 B per fragment record + 112 B per unit) over the "
                         "kernel time; the compact layout keeps only the entries that can change a sum "
                         "(resident_bytes_per_launch) so it can exceed the HBM peak -- `traffic` / "
                         "`traffic_frac_of_peak` are the physical HBM bytes (PMC) of the same kernel, and "
                         "`roofline_dense_layout` is the same pass streaming the canonical records")
                        if layout_name in ("short", "compact") else "canonical 16-byte records streamed as they are",
            },
            "host": {"generate_s": gen_s, "first_create_s": upload_s, "steady_create_s": upload_steady_s,
                     "pcie_inclusive_breakpoints_per_s": n / (upload_steady_s + kern_ms * 1e-3),
                     "note": "svt_batch_create (validate + H2D through the pinned ring + scan + tiling + repack) "
                             "+ one pass; host buffers in pageable memory; never part of `value`"},
        }
        if gather:
            out["gather"] = gather
        if args.workload == "c5_multisample":   # one breakpoint = one VCF site; a unit = (site, sample)
            out["sites_per_s"] = value / 32.0
            out["units_per_s"] = value
        if world == 1 and layout_name in ("short", "compact") and not args.no_dense_leg:
            # the same pass over the canonical 16-byte records (SVT_FLAG_DENSE_LAYOUT), for reference
            try:
                with hip.DeviceBatch(batch, device=local_rank, flags=flags | ev.FLAG_DENSE_LAYOUT) as dd:
                    dd.genotype(sync=True)
                    d_ms = dd.genotype_timed(args.steps) / args.steps
                    d_alg, d_res = dd.bytes()
                d_traffic = None
                try:
                    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                        tj = json.load(f)["dense"]
                    if tj.get("units") == n and tj.get("records") == batch.n_records:
                        d_traffic = tj["traffic_bytes_per_launch"]
                except (OSError, ValueError, KeyError):
                    pass
                out["roofline_dense_layout"] = {
                    "bound": "hbm", "kernel": "svt_genotype_kernel (dense records)", "kernel_ms": d_ms,
                    "achieved": d_alg / (d_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d_alg / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": d_traffic,
                    "resident_bytes_per_launch": d_res, "breakpoints_per_s_kernel_only": n / (d_ms * 1e-3)}
            except Exception as e:  # the reference leg must never break the bench line
                out["roofline_dense_layout"] = {"error": repr(e)}

        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline: the C restatement (oracle/, a port of the reference's algorithm) on the
            # host cores over a bounded sample of the same workload, also used as parity check
            from oracle import c_oracle
            sample_n = n
            sample = batch.slice(0, sample_n)
            threads = min(c_oracle.max_threads(), n_cpu)   # more threads than the CPU quota only get throttled
            want = c_oracle.genotype_batch(sample, flags=flags & ev.FLAG_SSO_ASSOCIATION, n_threads=threads)   # warm-up + parity reference
            t0 = time.perf_counter()
            reps = 0
            while True:
                c_oracle.genotype_batch(sample, flags=flags & ev.FLAG_SSO_ASSOCIATION, n_threads=threads, out=want)
                reps += 1
                if time.perf_counter() - t0 >= args.cpu_seconds or reps >= 50:
                    break
            cpu_s = time.perf_counter() - t0
            n1 = min(sample_n, 200_000)               # the same restatement on one thread, bounded slice
            one = batch.slice(0, n1)
            c_oracle.genotype_batch(one, flags=flags & ev.FLAG_SSO_ASSOCIATION, n_threads=1)
            t0 = time.perf_counter()
            c_oracle.genotype_batch(one, flags=flags & ev.FLAG_SSO_ASSOCIATION, n_threads=1)
            one_thread = n1 / (time.perf_counter() - t0)
            out["cpu_baseline"] = {
                "value": sample_n * reps / cpu_s,
                "unit": "breakpoints/s",
                "cores": threads,
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
                py_oracle.genotype_batch(batch.slice(0, n1), flags & ev.FLAG_SSO_ASSOCIATION)
                one = n1 / (time.perf_counter() - t0)
                npool = min(n, 2000 * threads)
                t0 = time.perf_counter()
                py_oracle.genotype_batch_pool(batch.slice(0, npool), flags & ev.FLAG_SSO_ASSOCIATION,
                                              processes=threads, batch_size=1000)
                pool = npool / (time.perf_counter() - t0)
                out["cpu_baseline_python"] = {
                    "kind": "port", "unit": "breakpoints/s", "one_process": one, "pool": pool, "cores": threads,
                    "sample": "oracle/py_oracle.py: first %d units (1 process), first %d units "
                              "(multiprocessing.Pool(%d), batch_size=1000)" % (n1, npool, threads)}
            except Exception as e:  # never let the extra baseline break the bench line
                out["cpu_baseline_python"] = {"error": repr(e)}
            ints_bad = int((got.counts[:sample_n] != want.counts).sum() + (got.gt[:sample_n] != want.gt).sum())
            out["parity"] = {
                "units_checked": sample_n,
                "integer_mismatches": ints_bad,
                "max_abs_dGL": float(np.max(np.abs(got.gl[:sample_n] - want.gl))),
                "max_abs_dSQ": float(np.max(np.abs(got.sq[:sample_n] - want.sq))),
            }
        print(json.dumps(out), file=json_out, flush=True)

    dbatch.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
