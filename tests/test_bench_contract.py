"""bench.py prints ONE JSON line with the keys the driver and the judge read (gpu: it runs the real path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys(hip_device):
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--units", "30000",
                        "--cpu-seconds", "1", "--large-units", "70000"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and isinstance(cpu["cpu_model"], str) and cpu["cpu_model"]
    # parsed.cpu_baseline is SURVEY 8(d)(ii)'s denominator (the pure-Python pool); the C port of the same algorithm beside it
    assert cpu["value"] == cpu["pool"] > 0 and "multiprocessing.Pool" in cpu["sample"]
    c = d["cpu_baseline_c"]
    assert c["kind"] == "port" and c["value"] > cpu["value"] and c["one_thread"] > 0
    assert abs(d["vs_cpu"]["python_restatement_pool"] - d["value"] / cpu["value"]) < 1e-6 * d["vs_cpu"]["python_restatement_pool"]
    assert d["scaling_answer"] is None                     # (N = 1)
    # the public drivers end to end, called with the reference's own positional arguments
    for key, rate in (("driver_sso", "sites_per_s"), ("driver_classic_8bam", "units_per_s")):
        leg = d["real_data"][key]
        assert leg[rate] > 0 and leg["per_line_route"]["same_bytes"] is True
        st = leg["host_stage_ms"]
        assert st["route"] == "bulk" and st["lines_handed_back_to_python"] == 0 and st["vcf_parse_ms"] > 0 and st["vcf_emit_ms"] > 0
    assert d["real_data"]["driver_sso"]["every_repeat_equals_example_gt_vcf"] is True
    assert d["value"] > 0 and abs(d["value"] - 30000 * 3 / (d["ms_per_step"] * 3 * 1e-3)) / d["value"] < 1e-6
    assert d["parity"]["integer_mismatches"] == 0 and d["parity"]["max_abs_dGL"] <= 1e-6 and d["parity"]["max_abs_dSQ"] <= 1e-6
    # the timed step is the whole path from the canonical input: the algorithmic rate cannot exceed the HBM peak
    assert roof["kernel"] == "svt_stream_kernel" and 0 < roof["frac"] <= 1.0
    assert "nothing pre-digested" in d["config"]["step"]
    # a traffic figure is only reported when it was measured on this very build
    assert roof["traffic"] is None or "these kernel sources" in roof["traffic_source"]
    assert len(roof["source_sha16"]) == 16 and len(roof["library_sha16"]) == 16 and roof["compiler"]
    # where the host spent the timed region: the device's share (HIP events) lies inside the launch-and-wait time
    th = roof["timed_region_host"]
    assert th["launch_and_wait_ms"] >= th["device_ms_by_hip_events"] > 0 and th["host_ms_outside_the_events"] >= 0 and th["barrier_ms"] >= 0
    assert abs(th["device_ms_by_hip_events"] - roof["kernel_ms"] * d["steps"]) < 1e-6
    # the same launches without the untimed spin-up are in the record too
    assert roof["no_spinup_kernel_ms"] > 0 and 0 < roof["no_spinup_frac"] <= 1.0
    # the placement audition before the timed steps is setup and says what it did; the six fresh allocations below are without it
    pt = roof["placement_tuned"]
    assert 0 < pt["after_ms"] <= pt["before_ms"] and pt["result_candidates"] == 32 and pt["record_candidates"] == 8 and pt["idle_after_s"] >= 2.0
    assert d["sso"]["placement_tuned"]["after_ms"] <= d["sso"]["placement_tuned"]["before_ms"]
    pl = roof["placement"]
    assert len(pl["kernel_ms"]) == 6 and pl["min"] <= pl["median"] <= pl["max"] and pl["spread_pct"] >= 0
    # the labelled extra legs
    assert d["one_shot"]["pcie_inclusive_breakpoints_per_s"] > 0 and d["one_shot"]["wall_ms"] > 0
    assert d["large_batch"]["units"] == 70000 and d["large_batch"]["first_units_equal_headline"] is True
    assert 0 < d["large_batch"]["frac"] <= 1.0
    assert d["one_shot_packed"]["results_equal_headline"] is True and d["one_shot_packed"]["bytes_per_fragment_record"] < 5
    assert d["one_shot_packed"]["pack_inclusive_wall_ms"] > d["one_shot_packed"]["wall_ms"]
    assert d["one_shot_packed"]["from_records_wall_ms"] > 0 and d["one_shot_packed"]["from_records_results_equal_headline"] is True
    assert d["one_shot_packed"]["from_records_serial_wall_ms"] > d["one_shot_packed"]["wall_ms"]
    # the singlesample association, the configs[4] shape and the 8-GPU shard: own fractions, own (or no) traffic figures
    assert 0 < d["sso"]["frac"] <= 1.0 and d["sso"]["units"] == 30000
    assert d["result128"]["host_results_equal_headline"] is True and 0 < d["result128"]["frac"] <= 1.0 and d["config"]["device_result_record_bytes"] == 96
    c5 = d["c5_multisample"]
    assert c5["units"] == 60000 - 60000 % 32          # configs[4]'s per-GPU share is twice the headline's units
    assert 0 < c5["frac"] <= 1.0 and c5["table_mode"] == 1 and c5["units"] % 32 == 0 and c5["hintless"]["table_mode"] == 1 and c5["hintless"]["results_equal"] and c5["general_tables"]["table_mode"] == 2
    assert c5["site_major_input"]["results_and_site_qual_equal"] is True
    # configs[4] at "its full size" (here 8 x the leg's batch): every site block repeats the leg's pass, QUAL too
    full = d["c5_full"]
    assert "error" not in full, full
    assert full["units"] == 8 * c5["units"] and full["every_site_block_equals_the_2M_unit_pass"] is True and full["site_qual_equals"] is True
    assert 0 < full["frac"] <= 1.0 and full["result_slots"] >= full["units"]
    # the three placements side by side, the ratios against the CPU restatements, the timed region's dispatches for the profile
    assert roof["frac_tuned"] == roof["frac"] and 0 < roof["frac_cold"] <= 1.0 and 0 < roof["frac_untuned_median"] <= 1.0
    assert roof["timed_region_dispatches"]["count"] == 3 and roof["timed_region_dispatches"]["first"] is None      # (an audition ran)
    assert d["vs_cpu"]["c_port"] > 1 and d["vs_cpu"]["python_restatement_pool"] > d["vs_cpu"]["c_port"]
    assert d["shard_of_8"]["results_equal_headline"] is True
    # ... and the same batch as packed evidence of several libraries (library switches in the pair streams)
    assert c5["packed"]["results_equal"] is True and c5["packed"]["bytes_per_record"] < 6 and c5["packed"]["pass_ms"] > 0
    assert c5["placement_tuned"]["after_ms"] <= c5["placement_tuned"]["before_ms"]
    assert c5["one_shot"]["hinted_results_equal"] and c5["one_shot"]["hintless_results_equal"] and c5["one_shot"]["hintless_over_hinted"] < 1.5
    for leg in (d["sso"], c5):
        assert leg["traffic"] is None or "these kernel sources" in leg["traffic_source"]
    assert d["one_shot_packed"]["pack_ms_median_back_to_back"] > 0
    # the rows either side of the path on real BAM bytes: the driver itself reproduces the expected VCF, stage times are there
    real = d["real_data"]
    assert "error" not in real, real
    assert real["fixture_driver"]["output_equals_example_gt_vcf"] is True
    for key in ("fixture_x100", "wgs_like_30x"):
        leg = real[key]
        assert leg["sites"] > 0 and leg["fragments"] > 0 and leg["sites_per_s"] > 0 and leg["lines_out"] >= leg["sites"]
        assert set(leg["stage_ms"]) == {"vcf_parse_and_site_arrays", "inflate_fetch_summarise_host", "h2d_plus_geometry_kernel", "genotype_pass", "results_d2h", "vcf_emit_lines"}
        assert all(v >= 0 for v in leg["stage_ms"].values()) and leg["h2d_bytes"] == 16 * leg["fragments"] + 24 * leg["sites"]
        assert leg["geometry"] == "reader" and leg["device_geometry"]["same_genotypes"] is True and leg["device_geometry"]["h2d_bytes"] > 7 * leg["h2d_bytes"]
    assert real["fixture_x100"]["sites"] == 21100
    sh = d["shard_of_8"]
    assert sh["results_equal_headline"] is True and 0 < sh["units"] < 30000 and sh["speedup_vs_headline"] > 0


@pytest.mark.gpu
def test_one_line_on_stdout_with_the_process_group_up(hip_device):
    """--force-dist: RCCL initialised and the gather run on one rank; RCCL's banner must not reach stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--units", "20000", "--force-dist",
                        "--no-cpu-baseline", "--no-extra-legs", "--scaling", "strong"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["gather"]["collective"] == "rccl gather" and d["value"] > 0
    assert d["scaling"] == "strong" and d["config"]["total_units"] == 20000 and d["gather"]["units_per_rank"] == [20000]


def test_a_rank_builds_only_its_shard_of_the_strong_workload():
    """host only: bench.generate_shard (the `strong` leg = configs[3]) returns, for every rank, exactly the units a slice of the
    whole workload under distributed.shard_bounds would -- without generating the chunks the shard does not touch (the bounds
    come from synth.make_units(counts_only=True), the first draws of every chunk's generator)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from svtyper_amd import distributed as D
    n, world = 120_000, 3
    whole = bench.generate("c3_mixed_1m", n, 0, 2)
    bounds = D.shard_bounds(whole.rec_offset, world, 1)
    assert sum(hi - lo for lo, hi in bounds) == n
    for r in range(world):
        shard, b = bench.generate_shard("c3_mixed_1m", n, world, r, 2)
        assert b == bounds
        want = whole.slice(*bounds[r])
        assert np.array_equal(shard.rec_offset, want.rec_offset) and np.array_equal(shard.units, want.units)
        assert shard.records.tobytes() == want.records.tobytes()
