"""N > 1 path on CPU: world_size-2 gloo run of shard -> genotype -> single gather (the GPU path uses
the same functions with the "nccl" = RCCL backend; here the engine seam is filled by the oracle)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, scenario="c3", route=""):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if route:
        os.environ["SVT_GATHER"] = route
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import c_oracle
    from svtyper_amd import distributed as D
    from svtyper_amd import synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = 1
    if scenario == "c3":
        lib = synth.normal_library(n=50000)
        batch = synth.make_units(3001, 77, [lib], svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=30, sd_frags=15,
                                 min_frags=0)
    else:       # the configs[4] shape; "empty": fewer sites than ranks, one rank has nothing to contribute
        group = 8
        batch = synth.make_multisample(1 if scenario == "empty" else 75, group, seed=4, mean_frags=20, sd_frags=8, min_frags=1, max_frags=50)
    bounds = D.shard_bounds(batch.rec_offset, world, group)
    assert all(lo % group == 0 for lo, _ in bounds)
    if scenario == "empty":
        assert sorted(hi - lo for lo, hi in bounds) == [0, group]
    shard, (lo, hi) = D.local_shard(batch, rank, world, group)
    local = c_oracle.genotype_batch(shard, n_threads=1)
    t = torch.from_numpy(local.rec.view(np.uint8).copy())
    gathered = D.gather_result_records(t, [b[1] - b[0] for b in bounds], dst=0)
    assert D.gather_route() == ("padded dist.gather + concatenation" if route == "collective" else "point-to-point into one buffer")
    if rank == 0:
        got = D.results_from_bytes(gathered)
        want = c_oracle.genotype_batch(batch, n_threads=1)
        ok = got.rec.tobytes() == want.rec.tobytes()
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,world,route", [("c3", 2, ""), ("c5", 2, ""), ("empty", 2, ""), ("c3", 3, ""), ("c3", 2, "collective"),
                                                  ("empty", 2, "collective")])
def test_shard_and_gather_over_gloo(tmp_path, scenario, world, route):
    """every rank's records onto rank 0: point-to-point into one preallocated buffer (the default), or the padded collective
    (SVT_GATHER=collective, also what the ranks agree to fall back to when the backend cannot do the first): same bytes"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out, scenario, route), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_balanced_and_grouped():
    from svtyper_amd import distributed as D
    rng = np.random.default_rng(3)
    F = rng.integers(0, 200, 10_000)
    off = np.concatenate([[0], np.cumsum(F)]).astype(np.uint64)
    for world in (1, 2, 4, 8):
        b = D.shard_bounds(off, world, group=4)
        assert b[0][0] == 0 and b[-1][1] == 10_000
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert all(lo % 4 == 0 for lo, _ in b)
        cost = [float((off[hi] - off[lo]) * 16 + 112 * (hi - lo)) for lo, hi in b]
        assert max(cost) / (sum(cost) / world) < 1.02
    assert D.shard_bounds(np.zeros(1, np.uint64), 4) == [(0, 0)] * 4
