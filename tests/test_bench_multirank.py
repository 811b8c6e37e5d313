"""bench.py's N > 1 branch, launched exactly as the driver launches it (torch.distributed.run, one rank per GPU), on
whatever devices the box has: with fewer devices than ranks the ranks share them and the gather goes over gloo
(bench.py's rule, the same as tests/test_multi_device.py), with N devices the gather is RCCL.  Every line of the
multi-rank code -- shard cut, max-over-ranks timing, uneven gather, value_with_gather, the strong-scaling byte
equality -- has run before the first 8-GPU SCALE run.  Replaces the Pool fan-out of svtyper/singlesample.py:723-751."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(n, extra, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _common(d, n, total_units):
    from svtyper_amd import hip
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1
    g = d["gather"]
    assert len(g["units_per_rank"]) == n and sum(g["units_per_rank"]) == total_units == d["config"]["total_units"]
    rb = d["config"]["device_result_record_bytes"]
    assert g["record_bytes"] == rb and g["bytes_per_rank"] >= rb * g["units_per_rank"][0]      # (96: whole workgroups of tagged slots)
    shared = hip.device_count() < n
    assert d["shared_devices"] is shared
    assert g["backend"] == ("gloo" if shared else "nccl") and d["rccl_ranks"] == (0 if shared else n) == g["rccl_ranks"]
    assert 0 < d["value_with_gather"] < d["value"]
    assert abs(d["value"] - total_units * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-6
    assert 0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["kernel_ms_max_over_ranks"] >= d["roofline"]["kernel_ms"] * (1 - 1e-9)
    for leg in ("sso", "c5_multisample", "one_shot", "cpu_baseline", "large_batch"):
        assert leg not in d          # the extra legs and the CPU baseline are N = 1 only
    assert d["scaling_answer"]["number"] == "value" and "single gather" in d["scaling_answer"]["why"].lower()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_weak_scaling_line(hip_device, n):
    """the driver's own command: the workload PER rank, no collective inside a step, one gather at the end"""
    d = _launch(n, ["--units", "12000"])
    _common(d, n, 12000 * n)
    assert d["scaling"] == "weak" and d["gather"]["units_per_rank"] == [12000] * n
    assert d["config"]["units_per_gpu"] == 12000 and "equals_single_rank_pass" not in d["gather"]
    # ... and the SAME invocation answers BASELINE.json's multi-GPU configs: `strong` = configs[3] (the --units workload cut N
    # ways, gathered bytes equal to rank 0's own pass over all of it), `c5` = configs[4] (sites x 32 samples per rank, whole
    # sites per rank), each with the passes alone, the single gather (96-byte and compact 48-byte records) and the pipelined
    # steady state (pass of batch k+1 over the gather of batch k)
    assert d["value_pipelined"] > 0
    for key, total in (("strong", 12000), ("c5", None)):
        leg = d[key]
        per = leg["units_per_rank"]
        assert len(per) == n == len(leg["kernel_ms_per_rank"]) and sum(per) == leg["total_units"] and min(per) > 0
        assert leg["value"] > leg["value_with_gather"] > 0 and leg["value_pipelined"] > 0 and leg["value_pipelined_compact"] > 0
        assert leg["batches_pipelined"] >= 8 and leg["gather"]["record_bytes"] == 96 and leg["gather_compact"]["record_bytes"] == 48
        assert leg["rccl_ranks"] == d["rccl_ranks"]
        # the two ways off the devices: one gather onto rank 0 / every rank down its own PCIe link
        assert leg["d2h_parallel"]["ms"] > 0 and leg["d2h_parallel"]["GB/s_aggregate"] > 0 and 0 < leg["value_with_d2h_parallel"] < leg["value"]
        if total is not None:
            assert leg["total_units"] == total and leg["equals_single_rank_pass"] is True
            assert leg["gather_compact"]["genotype_fields_equal_single_rank_pass"] is True
            assert max(per) - min(per) < 0.2 * total / n + 64
        else:
            assert all(c % 32 == 0 for c in per) and len(set(per)) == 1 and leg["site_qual_sites"] == per[0] // 32
            assert abs(leg["sites_per_s"] * 32 - leg["value"]) < 1e-6 * leg["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,rec", [(2, 96), (8, 96), (2, 128)])
def test_strong_scaling_gathers_the_single_rank_bytes(hip_device, n, rec):
    """configs[3] literally: ONE workload cut by distributed.shard_bounds; rank 0 runs it alone afterwards and compares the bytes
    (rec = 96, the default: tagged SVT_FLAG_RESULT96 records through the gather, put in order and expanded on rank 0)"""
    d = _launch(n, ["--units", "40000", "--scaling", "strong", "--result-bytes", str(rec)])
    _common(d, n, 40000)
    assert d["gather"]["record_bytes"] == rec
    assert d["scaling"] == "strong" and d["gather"]["equals_single_rank_pass"] is True
    per = d["gather"]["units_per_rank"]
    assert min(per) > 0 and max(per) - min(per) < 0.2 * 40000 / n + 64      # balanced by bytes, not by count


@pytest.mark.gpu
@pytest.mark.parametrize("n,scaling", [(2, "weak"), (2, "strong"), (8, "strong")])
def test_c5_multisample_shards_at_whole_sites(hip_device, n, scaling):
    """the configs[4] shape: 32 samples per site, per-sample libraries (library windows); shards are cut at multiples of 32"""
    sites = 96
    d = _launch(n, ["--units", str(sites * 32), "--workload", "c5_multisample", "--scaling", scaling])
    total = sites * 32 * (n if scaling == "weak" else 1)
    _common(d, n, total)
    assert all(c % 32 == 0 for c in d["gather"]["units_per_rank"])
    assert abs(d["sites_per_s"] * 32 - d["units_per_s"]) < 1e-6 * d["units_per_s"]
    if scaling == "strong":
        assert d["gather"]["equals_single_rank_pass"] is True
