"""BASELINE.json configs[4] at ITS size on ONE MI355X: 500 000 sites x 32 samples = 16 M (site, sample) units, ~1.6 G fragment
records (37 % of the 32-bit record index space, ~26 GB of the 288 GB of HBM), ~31 k workgroups through the window chunks.
The batch is 32 copies of a 15 625-site batch (synth.replicate_sample_major: the host generates 500 k units in seconds, not 16 M
in minutes), so every property has an exact answer: the pass over the whole must repeat the pass over the base batch site block by
site block (split invariance at site boundaries), twice the same bytes (idempotence), the oracle's records on a sample of
sites, and QUAL summed on the device = QUAL summed on the host."""
import os

import numpy as np
import pytest

from svtyper_amd import evidence as ev
from svtyper_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_chunk_bounds_cut_at_whole_sites_within_the_bound(fixture_library):
    """svt_chunk_bounds (host only): the fewest contiguous chunks that fit a resident batch, cut at multiples of `group`"""
    from svtyper_amd import hip
    b = synth.make_units(3000, 5, [fixture_library])
    off = b.rec_offset
    total = int(off[-1])
    assert hip.chunk_bounds(off) == [(0, 3000)]
    for group, cap in ((1, 20_000), (32, 50_000), (7, total // 3 + 1)):
        bounds = hip.chunk_bounds(off, group, cap)
        assert bounds[0][0] == 0 and bounds[-1][1] == 3000 and all(a[1] == b2[0] for a, b2 in zip(bounds, bounds[1:]))
        sizes = [int(off[hi] - off[lo]) for lo, hi in bounds]
        assert max(sizes) <= cap and all(lo % group == 0 for lo, _ in bounds)
        # greedy = fewest: no chunk could have taken the next group as well
        for (lo, hi), _next in zip(bounds, bounds[1:]):
            assert int(off[min(3000, hi + group)] - off[lo]) > cap
    with pytest.raises(hip.SvtyperHipError):
        hip.chunk_bounds(off, 1, 10)          # a single unit beyond the bound


@pytest.mark.gpu
def test_a_batch_beyond_the_record_index_passes_through_the_one_shot_in_chunks(hip_device, fixture_library):
    """More records than one resident batch indexes (2^32 - 17; lowered to 30 000 for this process through
    SVT_MAX_BATCH_RECORDS, read once by the library): svt_batch_create refuses the batch and names the way out, svt_genotype
    runs it chunk after chunk (svt_chunk_bounds) -- the result records are those of the chunks genotyped one by one."""
    import subprocess
    import sys
    code = """
import numpy as np, sys
sys.path.insert(0, %r)
from svtyper_amd import hip, synth, evidence as ev
import bench
lib = bench.fixture_library()
b = synth.make_units(2500, 11, [lib])
assert b.n_records > 200_000
try:
    hip.DeviceBatch(b, 0, 0)
    raise SystemExit("svt_batch_create took a batch beyond the bound")
except hip.SvtyperHipError as e:
    assert "svt_chunk_bounds" in str(e), str(e)
bounds = hip.chunk_bounds(b.rec_offset)
assert len(bounds) >= 7 and all(int(b.rec_offset[hi] - b.rec_offset[lo]) <= 30000 for lo, hi in bounds)
for flags in (0, ev.FLAG_SSO_ASSOCIATION):
    whole = hip.genotype_batch(b, device=0, flags=flags)
    parts = np.concatenate([hip.genotype_batch(b.slice(lo, hi), device=0, flags=flags).rec for lo, hi in bounds])
    assert whole.rec.tobytes() == parts.tobytes()
    assert (whole.gt >= 0).sum() > 2000
print("ok")
""" % ROOT
    env = dict(os.environ, SVT_MAX_BATCH_RECORDS="30000")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_a_batch_beyond_the_32_bit_record_index_is_refused(hip_device, fixture_library):
    """>= 2^32 - 16 records in ONE RESIDENT batch: SVT_ERR_INVALID from svt_batch_create before anything is read or allocated
    (the kernels index records with 32 bits); the message names svt_chunk_bounds.  (svt_genotype, given arrays that really
    are that long, cuts them itself: the test above.)"""
    from svtyper_amd import hip
    b = synth.make_units(4, 1, [fixture_library], mean_frags=3, sd_frags=1, min_frags=1, max_frags=5)
    b.rec_offset[-1] = np.uint64(2**32 - 16)          # (in place: the constructor checks the arrays against each other)
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.DeviceBatch(b, hip_device, 0)
    assert "too many records" in str(e.value) and "svt_chunk_bounds" in str(e.value)
