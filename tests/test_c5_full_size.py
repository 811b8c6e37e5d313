"""BASELINE.json configs[4] at ITS size on ONE MI355X: 500 000 sites x 32 samples = 16 M (site, sample) units, ~1.6 G fragment
records (37 % of the 32-bit record index space, ~26 GB of the 288 GB of HBM), ~31 k workgroups through the window chunks.
The batch is 32 copies of a 15 625-site batch (synth.replicate_sample_major: the host generates 500 k units in seconds, not 16 M
in minutes), so every property has an exact answer: the pass over the whole must repeat the pass over the base batch site block by
site block (split invariance at site boundaries), twice the same bytes (idempotence), the oracle's records on a sample of
sites, and QUAL summed on the device = QUAL summed on the host."""
import numpy as np
import pytest

from svtyper_amd import evidence as ev
from svtyper_amd import synth

N_SAMPLES = 32


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


@pytest.mark.gpu
def test_sixteen_million_units_on_one_device(hip_device):
    from oracle import c_oracle
    from svtyper_amd import hip
    if _mem_available_gb() < 48:
        pytest.skip("needs ~30 GB of host memory for the 1.6 G records")
    base_sites, copies = 15_625, 32
    base = synth.make_multisample(base_sites, N_SAMPLES, seed=77, layout="sample")
    assert base.n_units == 500_000
    flags = ev.FLAG_RESULT96
    with hip.DeviceBatch(base, hip_device, flags) as d:
        d.result_order(N_SAMPLES)
        d.genotype(sync=True)
        want = d.results().rec.copy()                  # site-major: index site * 32 + sample
    # the oracle on the first 700 sites (x 32 samples), sample-major slices of the base batch
    k = 700
    part = ev.concat_batches([base.slice(s * base_sites, s * base_sites + k) for s in range(N_SAMPLES)])
    part.libs = base.libs
    o = c_oracle.genotype_batch(part, flags=0).rec.reshape(N_SAMPLES, k)
    w = want.reshape(base_sites, N_SAMPLES)[:k]
    assert np.array_equal(w["gt"], o["gt"].T) and np.array_equal(w["counts"], o["counts"].transpose(1, 0, 2))
    assert np.array_equal(w["gl"].view(np.uint64), np.ascontiguousarray(o["gl"].transpose(1, 0, 2)).view(np.uint64))
    assert np.max(np.abs(w["sq"] - o["sq"].T)) <= 1e-6

    big = synth.replicate_sample_major(base, N_SAMPLES, copies)
    n_sites = base_sites * copies
    assert big.n_units == 16_000_000 and big.n_records > 1_500_000_000
    with hip.DeviceBatch(big, hip_device, flags) as d:
        assert d.table_mode() == 1                      # library windows
        d.result_order(N_SAMPLES)
        d.genotype(sync=True)
        got = d.results().rec
        slots = d.result_slots()
        assert big.n_units <= slots < 1.5 * big.n_units
        # split invariance: site block c of the whole batch = the base batch's own pass
        blocks = got.reshape(copies, base_sites * N_SAMPLES)
        for c in range(copies):
            assert blocks[c].tobytes() == want.tobytes(), "site block %d differs from the base batch's pass" % c
        # idempotence
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == got.tobytes()
        # QUAL (classic.py:216-217,485,498): the device's running sums = the host's
        q_dev = d.site_qual(N_SAMPLES)
        q_host = hip.site_qual_host(ev.Results(got), N_SAMPLES)
        assert q_dev.shape == (n_sites,) and np.array_equal(q_dev.view(np.uint64), np.asarray(q_host).view(np.uint64))
        ms = min(d.genotype_timed(3) / 3 for _ in range(2))
        alg, _ = d.bytes()
        print("configs[4] at full size: %.3f ms per pass, %.3f of 8 TB/s, %d result slots" % (ms, alg / (ms * 1e-3) / 8e12, slots))


@pytest.mark.gpu
def test_a_batch_beyond_the_32_bit_record_index_is_refused(hip_device, fixture_library):
    """>= 2^32 - 16 records in one batch: SVT_ERR_INVALID from svt_batch_create before anything is read or allocated (the
    kernels index records with 32 bits)."""
    from svtyper_amd import hip
    b = synth.make_units(4, 1, [fixture_library], mean_frags=3, sd_frags=1, min_frags=1, max_frags=5)
    b.rec_offset[-1] = np.uint64(2**32 - 16)          # (in place: the constructor checks the arrays against each other)
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.DeviceBatch(b, hip_device, 0)
    assert "too many records" in str(e.value)
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_batch(b, device=hip_device, flags=0)
