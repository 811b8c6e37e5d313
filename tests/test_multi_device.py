"""Several GPUs from one process through the C ABI (svt_genotype_multi / svt_shard_bounds), and ranks of a
torch.distributed job running the HIP path (shard -> HIP -> one gather).  On a one-GPU box the same device is listed
several times / shared by the ranks: every line of the multi-device code runs, only the devices coincide."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_shard_bounds_equal_the_python_rule():
    """host only: svt_shard_bounds == distributed.shard_bounds (the rule ranks use)"""
    from svtyper_amd import distributed as D, hip
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 1000, 10_000):
        F = rng.integers(0, 200, n)
        off = np.concatenate([[0], np.cumsum(F)]).astype(np.uint64)
        for world in (1, 2, 3, 8):
            for group in (1, 4, 32):
                assert hip.shard_bounds(off, world, group) == D.shard_bounds(off, world, group), (n, world, group)


@pytest.mark.gpu
@pytest.mark.parametrize("sso", [0, 1])
def test_genotype_multi_equals_one_device_and_the_oracle(hip_device, fixture_library, sso):
    from oracle import c_oracle
    from svtyper_amd import evidence as ev, hip, synth
    batch = synth.make_units(20_011, 41, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1), min_frags=0)
    one = hip.genotype_batch(batch, device=hip_device, flags=sso)
    n_dev = hip.device_count()
    for devices in ([0], [0, 0], [d % n_dev for d in range(5)], list(range(n_dev)) * 2):
        got = hip.genotype_multi(batch, devices, flags=sso)
        assert got.rec.tobytes() == one.rec.tobytes(), devices
    want = c_oracle.genotype_batch(batch, flags=sso)
    assert np.array_equal(one.gt, want.gt) and np.array_equal(one.counts, want.counts)
    assert np.array_equal(one.gl.view(np.uint64), want.gl.view(np.uint64))
    # a site's samples stay together: shards are cut at multiples of `group`
    ms = synth.make_multisample(60, 32, seed=9, mean_frags=30, sd_frags=10, min_frags=3, max_frags=70)
    assert hip.genotype_multi(ms, [0, 0, 0], group=32).rec.tobytes() == hip.genotype_batch(ms).rec.tobytes()
    assert all(lo % 32 == 0 for lo, _ in hip.shard_bounds(ms.rec_offset, 3, 32))
    # more devices than units, empty batch
    tiny = synth.make_units(2, 3, [fixture_library])
    assert hip.genotype_multi(tiny, [0, 0, 0, 0]).rec.tobytes() == hip.genotype_batch(tiny).rec.tobytes()
    assert hip.genotype_multi(synth.make_units(0, 3, [fixture_library]), [0, 0]).n_units == 0


_TORCH_VIEW = """
import sys, numpy as np, torch          # (torch first: its HIP runtime must be the one the process initialises)
sys.path.insert(0, %r); sys.path.insert(0, %r)
import bench
from svtyper_amd import hip, synth
batch = synth.make_units(5000, 3, [bench.fixture_library()], svtype_mix=(0.6, 0.2, 0.1, 0.1))
with hip.DeviceBatch(batch, %d) as d:
    d.genotype(sync=True)
    t = d.device_results_tensor()
    assert t.is_cuda and t.dtype == torch.uint8 and t.numel() == batch.n_units * 128 and t.data_ptr() == d.device_results_ptr()
    want = d.results().rec.tobytes()
    assert t.cpu().numpy().tobytes() == want
    t.zero_()
    torch.cuda.synchronize()
    d.genotype(sync=True)
    assert t.cpu().numpy().tobytes() == want
    # the storage owns a reference to the batch: a derived view alone keeps close() from handing the buffer back to the pool
    part = t[: 128 * 10].view(torch.int32)
    del t
    try:
        d.close()
        raise SystemExit("close() under a live view did not raise")
    except hip.SvtyperHipError as e:
        assert "still alive" in str(e)
    assert part.cpu().numpy().tobytes() == want[: 128 * 10]
    del part
print("view ok")
"""


@pytest.mark.gpu
def test_result_records_as_a_torch_view(hip_device):
    """DeviceBatch.device_results_tensor: the buffer the pass writes, handed to torch without a copy (what bench.py gathers over
    RCCL): same address, same bytes as svt_batch_results, and the next pass writes through it.  (Own process: torch has to
    be imported before the library initialises HIP.)"""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _TORCH_VIEW % (ROOT, os.path.join(ROOT, "tests"), hip_device)], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "view ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_genotype_multi_reports_the_failing_device(hip_device, fixture_library):
    from svtyper_amd import evidence as ev, hip, synth
    b = synth.make_units(3000, 3, [fixture_library])
    b.records["flags"][b.n_records - 5] |= 1 << 28          # an undefined flag bit in the last shard
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.genotype_multi(b, [0, 0])
    assert "reserved/undefined bits" in str(e.value) and "device 0" in str(e.value)
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_multi(b, [0, 99])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, out_path, scenario="c3"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from svtyper_amd import distributed as D, evidence as ev, hip, synth

    n_dev = hip.device_count()
    rccl = n_dev >= world                       # one GPU per rank: RCCL; otherwise the ranks share device 0 over gloo
    device = rank if rccl else 0
    torch.cuda.set_device(device)
    dist.init_process_group("nccl" if rccl else "gloo", rank=rank, world_size=world,
                            **({"device_id": torch.device("cuda", device)} if rccl else {}))
    group = 1
    flags = ev.FLAG_RESULT96 if scenario == "c3_r96" else 0    # c3_r96: 96-byte device records through the gather
    rec_bytes = 96 if flags else 128
    if scenario in ("c3", "c3_r96"):   # configs[3]: one library, mixed SV types
        lib = synth.normal_library(n=50000)
        batch = synth.make_units(30_001, 77, [lib], svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=40, sd_frags=20, min_frags=0)
    elif scenario == "c5":          # configs[4]: 32 samples with their own libraries (library windows), a site's samples on one rank
        group = 32
        batch = synth.make_multisample(301, 32, seed=21, mean_frags=30, sd_frags=12, min_frags=2, max_frags=80)
    else:                           # "empty": fewer sites than ranks -- one rank gets no unit at all
        group = 32
        batch = synth.make_multisample(1, 32, seed=22, mean_frags=30, sd_frags=12, min_frags=2, max_frags=80)
    bounds = D.shard_bounds(batch.rec_offset, world, group)
    assert all(lo % group == 0 for lo, _ in bounds)
    if scenario == "empty":
        assert any(hi == lo for lo, hi in bounds) and any(hi > lo for lo, hi in bounds)
    shard, (lo, hi) = D.local_shard(batch, rank, world, group)
    with hip.DeviceBatch(shard, device=device, flags=flags) as d:
        if scenario == "c5":
            assert d.table_mode() == 1
        assert d.result_bytes() == rec_bytes
        buf = torch.zeros(max(1, d.result_slots()) * rec_bytes, dtype=torch.uint8, device="cuda")
        d.bind_device_results(buf.data_ptr())        # the kernel writes straight into the tensor that is gathered
        d.genotype(sync=True)
        local = buf[: shard.n_units * rec_bytes] if rccl else buf[: shard.n_units * rec_bytes].cpu()
        counts = [b[1] - b[0] for b in bounds]
        if flags:     # tagged 96-byte records: a rank's buffer holds whole workgroups' worth of slots, sizes travel first
            assert d.result_slots() >= shard.n_units
            local = buf[: d.result_slots() * 96] if rccl else buf[: d.result_slots() * 96].cpu()
            gathered, sizes = D.gather_tagged_records(local, dst=0)
        else:
            gathered = D.gather_result_records(local, counts, dst=0)
    if rank == 0:
        from oracle import c_oracle
        got = D.results_from_tagged(gathered, sizes, counts) if flags else D.results_from_bytes(gathered)
        single = hip.genotype_batch(batch, device=device)
        want = c_oracle.genotype_batch(batch)
        if group > 1:     # QUAL over a site's samples from the gathered (site-major) records == the single-rank device pass
            with hip.DeviceBatch(batch, device=device) as d1:
                d1.genotype(sync=True)
                q1 = d1.site_qual(group)
            assert hip.site_qual_host(got, group).tobytes() == hip.site_qual_host(single, group).tobytes()
            assert np.allclose(q1, hip.site_qual_host(single, group), rtol=0, atol=1e-6)
        ok = (got.rec.tobytes() == single.rec.tobytes() and np.array_equal(got.gt, want.gt)
              and np.array_equal(got.counts, want.counts) and np.array_equal(got.gl.view(np.uint64), want.gl.view(np.uint64))
              and float(np.max(np.abs(got.sq - want.sq))) <= 1e-6)
        with open(out_path, "w") as f:
            f.write("ok %s" % ("rccl" if rccl else "gloo") if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", ["c3", "c3_r96", "c5", "empty"])
def test_two_ranks_shard_hip_gather(hip_device, tmp_path, scenario):
    """world 2: shard_bounds -> the HIP path on every rank -> ONE gather of the 128-byte records; byte-equal to the
    single-rank result and parity-equal to the oracle (RCCL when two devices are visible, gloo on one).  c3: one
    library; c5: the configs[4] shape (library windows, shards cut at whole sites: group = 32, QUAL per site from the
    gathered records); empty: fewer sites than ranks, so one rank contributes nothing to the gather."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_rank, args=(2, _free_port(), out, scenario), nprocs=2, join=True)
    assert open(out).read().startswith("ok")
