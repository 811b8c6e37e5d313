#!/usr/bin/env python
"""Condense the rocprofv3 output of tools/profile.sh (rocpd sqlite databases) into a text summary
that is small enough to commit under profiles/."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


# ---- the headline's timed region, isolated: a process without audition or other legs, whose bench line says which dispatches of
# the headline kernel its timed region was (roofline.timed_region_dispatches)
def _headline_line(name):
    try:
        import json as _j
        return _j.loads([l for l in open(os.path.join(out, name)) if l.startswith("{")][-1])
    except Exception:
        return None


def _timed_rows(sub, line, query):
    """rows of the headline kernel's dispatches in start order, cut to the timed region of `line`"""
    tr = line["roofline"]["timed_region_dispatches"]
    pat = "svt_stream_kernel<%s, 0, " % ("true" if line["config"]["association"] == "sso" else "false")
    rows = []
    for f in dbs(sub):
        c = sqlite3.connect(f)
        rows += [r for r in c.execute(query) if pat in r[0]]
    return rows, tr


def _first_of(tr, n):
    """index of the timed region's first dispatch among the n dispatches of the headline kernel in the trace"""
    if tr.get("first") is not None:
        return tr["first"]
    if tr.get("from_end") is not None:      # behind a placement audition: counted from the end of the process
        return n - tr["from_end"] - tr["count"]
    return None


_hl = _headline_line("stats_headline.json")
if _hl and _first_of(_hl["roofline"]["timed_region_dispatches"], 10 ** 9) is not None:
    rows, tr = _timed_rows("headstats", _hl, "select name, start, duration from kernels order by start")
    d = [r[2] / 1e3 for r in rows]
    a, k = _first_of(tr, len(d)), tr["count"]
    if a >= 0 and len(d) >= a + k:
        timed = d[a:a + k]
        avg = sum(timed) / k
        tuned = _hl["roofline"].get("placement_tuned")
        print("# TIMED REGION of the headline (own process, no other leg, %s; `bench.py --steps %d --no-extra-legs%s`)" % (
            "AFTER svt_batch_tune_placement -- the state the bench line is quoted in" if tuned else "no audition", k, "" if tuned else " --tune-placement 0,0"))
        print("# dispatches %d..%d of %d of the headline kernel = the %d launches between bench.py's HIP events" % (a, a + k - 1, len(d), k))
        print("timed_region_avg_us,%.3f" % avg)
        print("timed_region_min_us,%.3f" % min(timed))
        print("timed_region_max_us,%.3f" % max(timed))
        print("bench_kernel_ms_same_run,%.5f   (HIP events / steps: includes the gaps between consecutive dispatches)" % _hl["roofline"]["kernel_ms"])
        print("timed_region_avg_over_bench_kernel_ms,%.4f" % (avg / 1e3 / _hl["roofline"]["kernel_ms"]))
        alg = 16 * _hl["config"]["records_per_gpu"] + 112 * _hl["config"]["units_per_gpu"]
        print("frac_from_trace_avg,%.4f   frac_of_the_bench_line,%.4f   frac_cold_of_the_bench_line,%.4f" % (
            alg / (avg * 1e-6) / 8e12, _hl["roofline"]["frac"], _hl["roofline"]["frac_cold"]))
        print("all_dispatches_avg_us,%.3f   (cold passes + spin-up + warm-up + timed: what `top_kernels.average` shows)" % (sum(d) / len(d)))
    for sub, counter, fname in (("headfetch", "FETCH_SIZE", "fetch_headline.json"), ("headwrite", "WRITE_SIZE", "write_headline.json")):
        line = _headline_line(fname)
        if not line or _first_of(line["roofline"]["timed_region_dispatches"], 10 ** 9) is None:
            continue
        rows, tr = _timed_rows(sub, line, "select kernel_name, dispatch_id, value from counters_collection where counter_name = '%s' order by dispatch_id" % counter)
        v = [r[2] for r in rows]
        a, k = _first_of(tr, len(v)), tr["count"]
        if a >= 0 and len(v) >= a + k:
            print("timed_region_%s_KB_mean,%.1f   (dispatches %d..%d; all %d dispatches: %.1f)" % (counter, sum(v[a:a + k]) / k, a, a + k - 1, len(v), sum(v) / len(v)))

for f in dbs("stats"):
    c = sqlite3.connect(f)
    print("# rocprofv3 --kernel-trace --stats   (%s)" % os.path.relpath(f, out))
    print("name,total_calls,total_duration_us,average_us,percentage")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(",".join(str(x) for x in r))

# The same trace dispatch by dispatch: `top_kernels` averages EVERY call of a kernel in the process -- the cold passes right after
# svt_batch_create, the placement audition's candidate buffers (svt_batch_tune_placement), spin-up, the untuned `placement` leg --,
# the bench line's kernel_ms is the `steps` launches of the timed region.  The fastest run of `steps` consecutive dispatches is
# the statistic of the trace that corresponds to it (kernel_ms also holds the few microseconds between consecutive dispatches).
try:
    _steps = json_steps = None
    import json as _json
    _line = [l for l in open(os.path.join(out, "stats_bench.json")) if l.startswith("{")][-1]
    _steps = int(_json.loads(_line)["steps"])
except Exception:
    _steps = 10
for f in dbs("stats"):
    c = sqlite3.connect(f)
    try:
        rows = list(c.execute("select name, start, duration from kernels where name like '%svt_%' order by start"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("# kernel trace by dispatch: median, and the fastest run of %d consecutive dispatches of the kernel (the bench's timed region is %d launches)" % (_steps, _steps))
        print("name,calls,median_us,fastest_%d_consecutive_avg_us" % _steps)
        by = {}
        for name, start, dur in rows:
            by.setdefault(name, []).append(dur / 1e3)
        for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            if len(d) < _steps:
                continue
            best = min(sum(d[i:i + _steps]) / _steps for i in range(len(d) - _steps + 1))
            print("%s,%d,%.3f,%.3f" % (name, len(d), sorted(d)[len(d) // 2], best))

for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in dbs(sub):
        c = sqlite3.connect(f)
        print("# rocprofv3 --pmc pass %s   (per-dispatch mean over the svt_* kernels)" % sub)
        print("kernel,counter,mean,dispatches,vgpr_count,lds_block_size,sgpr_count,grid,workgroup")
        q = ("select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), max(lds_block_size), "
             "max(sgpr_count), max(grid_size), max(workgroup_size) from counters_collection "
             "where kernel_name like '%svt_%' group by kernel_name, counter_name")
        for r in c.execute(q):
            name = r[0].replace("(anonymous namespace)::", "")
            print(",".join([name] + [str(x) for x in r[1:]]))


# ---- the entries bench.py reads back (profiles/hbm_traffic.json), stamped with the kernel sources they were measured on
import json
import re
try:
    line = [l for l in open(os.path.join(out, "stats_bench.json")) if l.startswith("{")][-1]
    bj = json.loads(line)
    # leg -> (kernel-name pattern of its launches, units, records, workload text)
    legs = {"stream_sso" if bj["config"]["association"] == "sso" else "stream":
            (r"svt_stream_kernel<%s, 0, \d(, \d)?>" % ("true" if bj["config"]["association"] == "sso" else "false"),
             bj["config"]["units_per_gpu"], bj["config"]["records_per_gpu"], bj["config"]["workload"])}
    if bj["config"]["association"] != "sso" and "kernel_ms" in bj.get("sso", {}):
        legs["stream_sso"] = (r"svt_stream_kernel<true, 0, \d(, \d)?>", bj["sso"]["units"], bj["config"]["records_per_gpu"], bj["sso"]["what"])
    if "kernel_ms" in bj.get("c5_multisample", {}):
        c5 = bj["c5_multisample"]
        legs["c5_windows"] = (r"svt_stream_kernel<(true|false), 1, \d(, \d)?>", c5["units"], c5["records"], c5["what"])
        legs["c5_hintless"] = (r"svt_stream_kernel<(true|false), 2, \d(, \d)?>", c5["units"], c5["records"], "the same batch without svt_unit.libs hints")

    def mean_of(sub, counter, pat):
        tot = cnt = 0
        for f in dbs(sub):
            c = sqlite3.connect(f)
            for name, v, k in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                                        "group by kernel_name", (counter,)):
                if re.search(pat, name):
                    tot += v * k
                    cnt += k
        return tot / cnt if cnt else None

    def avg_us_of(pat):
        for f in dbs("stats"):
            c = sqlite3.connect(f)
            for name, avg in c.execute("select name, average from top_kernels"):
                if re.search(pat, name):
                    return avg / 1e3 if avg > 1e4 else avg
        return None

    entry = {}
    for key, (pat, units, records, what) in legs.items():
        fs, ws = mean_of("pmc_fetch", "FETCH_SIZE", pat), mean_of("pmc_write", "WRITE_SIZE", pat)
        if fs is None or ws is None:
            continue
        entry[key] = {
            "kernel": pat, "workload": what, "units": units, "records": records, "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws,
            # gfx950: FETCH_SIZE reports half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section)
            "traffic_bytes_per_launch": int(2 * fs * 1024 + ws * 1024),
            "kernel_trace_avg_us": avg_us_of(pat), "source_sha16": bj["roofline"]["source_sha16"],
            "library_sha16": bj["roofline"]["library_sha16"], "source": "tools/profile.sh %s" % os.path.basename(out.rstrip("/")),
        }
    with open(os.path.join(out, "hbm_traffic_entry.json"), "w") as f:
        json.dump(entry, f, indent=1)
    print("# hbm_traffic entries:", json.dumps(entry))
except Exception as e:   # the text summary above is still good
    print("# no hbm_traffic entry:", repr(e))
