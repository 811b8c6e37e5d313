"""The drivers over several ranks (svtyper_amd/sharded.py): variants sharded, BND pairs kept whole, one
gather of the output text -- the bytes must equal the single-process run.  CPU: gloo, world size 2 and 3,
engine seam filled by the oracle; the gpu-marked twin runs the CLI under torch.distributed.run on RCCL."""
import io
import os
import socket
import subprocess
import sys

import pytest

import test_host_pipeline as H
from svtyper_amd import classic, sharded, singlesample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bnd(chrom, pos, vid, mate, alt):
    return "%s\t%d\t%s\tN\t%s\t0\t.\tSVTYPE=BND;STRANDS=+-:5;CIPOS=-10,10;CIPOS95=-2,2;MATEID=%s;PE=3;SR=2\n" % (
        chrom, pos, vid, alt, mate)


def _input_with_spread_bnd_pairs(tmp_path):
    """The fixture's body with three BND pairs whose mates lie far apart, one never-paired mate, a line
    without SVTYPE and one of an unsupported type."""
    lines = open(H.IN_VCF).read().splitlines(True)
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    extra = {
        3: _bnd("1", 1000100, "b1_1", "b1_2", "N[1:1002000["),
        5: _bnd("1", 1000500, "b2_1", "b2_2", "N[1:1003000["),
        60: _bnd("1", 1003000, "b2_2", "b2_1", "]1:1000500]N"),
        100: _bnd("2", 5000, "lonely_1", "lonely_2", "N[2:9000["),
        150: _bnd("1", 1004000, "b3_1", "b3_2", "N[1:1004500["),
        151: _bnd("1", 1004500, "b3_2", "b3_1", "]1:1004000]N"),
        170: "1\t2000\tnotype\tN\t<DEL>\t0\t.\tEND=3000\n",
        171: "1\t2000\tins\tN\t<INS>\t0\t.\tSVTYPE=INS;END=2001\n",
        200: _bnd("1", 1002000, "b1_2", "b1_1", "]1:1000100]N"),
    }
    out = []
    for i, l in enumerate(body):
        if i in extra:
            out.append(extra[i])
        out.append(l)
    path = str(tmp_path / "in.vcf")
    open(path, "w").write("".join(head + out))
    return path, len(out)


def _classic_args(bam=H.IN_BAM):
    return (20, 1, 1, 1000000, H.LIB_JSON, False, None, None, False, None, 1e10)


def _sso_args():
    return (20, 1, 1, 1000000, H.LIB_JSON, False, None, False, 1000, 1e10, None, 1000)


def _single(driver, in_path, args):
    sink = sharded._Sink()
    with open(in_path) as f:
        driver(H.IN_BAM, f, sink, *args, engine=H.oracle_engine)
    return sink.getvalue()


def test_plan_keeps_pairs_whole_and_order():
    body = ["1\t%d\tv%d\tN\t<DEL>\t0\t.\tSVTYPE=DEL;END=%d\n" % (i, i, i + 100) for i in range(20)]
    body[2] = _bnd("1", 2, "a_1", "a_2", "N[1:50[")
    body[17] = _bnd("1", 50, "a_2", "a_1", "]1:2]N")
    body[9] = _bnd("1", 9, "c_1", "c_2", "N[1:10[")
    body[10] = _bnd("1", 10, "c_2", "c_1", "]1:9]N")
    assert sharded.bnd_pairs(body) == {17: 2, 10: 9}
    plan = sharded.plan_shards(body, 4)
    assert sorted(i for p in plan for i in p) == list(range(20))
    owner = {i: r for r, p in enumerate(plan) for i in p}
    assert owner[2] == owner[17] == 3 and owner[9] == owner[10]
    assert all(p == sorted(p) for p in plan)
    # repeated BND ids: the split could change the pairing, so rank 0 takes every line
    body[11] = _bnd("1", 11, "a_1", "zz", "N[1:12[")
    assert sharded.plan_shards(body, 4) == [list(range(20)), [], [], []]


def test_rank_zero_never_ends_up_without_a_body_line(tmp_path):
    """Fewer lines than ranks with BND pairs: every line of rank 0's range is a first mate handed to a later rank.
    The drivers write the header at their first body line, so rank 0 must keep one (here: the whole input)."""
    pair = [_bnd("1", 2, "a_1", "a_2", "N[1:50["), _bnd("1", 50, "a_2", "a_1", "]1:2]N")]
    assert sharded.plan_shards(pair, 2) == [[0, 1], []]
    assert sharded.plan_shards(pair, 5) == [[0, 1], [], [], [], []]
    three = pair + ["1\t90\tv\tN\t<DEL>\t0\t.\tSVTYPE=DEL;END=190\n"]
    for world in (2, 3, 4):
        plan = sharded.plan_shards(three, world)
        assert plan[0] and sorted(i for p in plan for i in p) == [0, 1, 2]
    # and the joined output of such an input has its header
    lines = open(H.IN_VCF).read().splitlines(True)
    head = [l for l in lines if l.startswith("#")]
    in_path = str(tmp_path / "pair.vcf")
    open(in_path, "w").write("".join(head + pair))
    want = _single(classic.sv_genotype, in_path, _classic_args())
    got = ""
    for rank, mine in enumerate(sharded.plan_shards(pair, 2)):
        sink = sharded._Sink()
        classic.sv_genotype(H.IN_BAM, sharded._Lines(head + [pair[i] for i in mine], in_path), sink, *_classic_args(),
                            engine=H.oracle_engine)
        text = sink.getvalue()
        got += text if rank == 0 else "".join(l for l in text.splitlines(True) if not l.startswith("#"))
    assert got == want and got.startswith("##fileformat")


@pytest.mark.parametrize("driver", ["classic", "sso"])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_concatenated_shares_equal_the_single_run(tmp_path, driver, world):
    """No process group: run the driver over each rank's share in turn and join the texts."""
    in_path, n_body = _input_with_spread_bnd_pairs(tmp_path)
    fn, args = (classic.sv_genotype, _classic_args()) if driver == "classic" else (singlesample.sso_genotype, _sso_args())
    want = _single(fn, in_path, args)
    lines = open(in_path).read().splitlines(True)
    head = [l for l in lines if l.startswith("#")]
    body = lines[len(head):]
    plan = sharded.plan_shards(body, world)
    assert sum(len(p) for p in plan) == n_body
    got = ""
    for rank, mine in enumerate(plan):
        sink = sharded._Sink()
        fn(H.IN_BAM, sharded._Lines(head + [body[i] for i in mine], in_path), sink, *args, engine=H.oracle_engine)
        text = sink.getvalue()
        if rank:
            text = "".join(l for l in text.splitlines(True) if not l.startswith("#"))
        got += text
    assert got == want
    assert "b1_2" in want and "lonely_1" not in want and "notype" in want


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, driver, in_path, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = open(out_path, "w") if rank == 0 else io.StringIO()
    with open(in_path) as f:
        if driver == "classic":
            sharded.sv_genotype_sharded(H.IN_BAM, f, out, *_classic_args(), rank=rank, world=world,
                                        engine=H.oracle_engine)
        else:
            sharded.sso_genotype_sharded(H.IN_BAM, f, out, *_sso_args(), rank=rank, world=world,
                                         engine=H.oracle_engine)
    if rank == 0:
        out.close()
    else:
        assert out.getvalue() == ""
    sharded.finish()


@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_two_ranks_gloo(tmp_path, driver):
    import torch.multiprocessing as mp
    in_path, _ = _input_with_spread_bnd_pairs(tmp_path)
    fn, args = (classic.sv_genotype, _classic_args()) if driver == "classic" else (singlesample.sso_genotype, _sso_args())
    want = _single(fn, in_path, args)
    out = str(tmp_path / "out.vcf")
    mp.spawn(_worker, args=(2, _free_port(), driver, in_path, out), nprocs=2, join=True)
    assert open(out).read() == want


def test_fixture_two_ranks_matches_expected_vcf(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "out.vcf")
    mp.spawn(_worker, args=(2, _free_port(), "classic", H.IN_VCF, out), nprocs=2, join=True)
    H.same_vcf(out, H.EXPECTED)


@pytest.mark.gpu
@pytest.mark.parametrize("module", ["svtyper_amd.classic", "svtyper_amd.singlesample"])
def test_cli_under_torch_distributed_run(tmp_path, hip_device, module):
    """Two ranks through the CLI (native reader, HIP engine).  On a one-GPU box the ranks share the device and
    the text is gathered over gloo; with two or more GPUs it is RCCL."""
    in_path, _ = _input_with_spread_bnd_pairs(tmp_path)
    single, multi = str(tmp_path / "single.vcf"), str(tmp_path / "multi.vcf")
    common = ["-i", in_path, "-B", H.IN_BAM, "-l", H.LIB_JSON]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    subprocess.run([sys.executable, "-m", module] + common + ["-o", single], check=True, env=env, cwd=ROOT, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", module]
                   + common + ["-o", multi], check=True, env=env, cwd=ROOT, timeout=900)
    assert open(multi).read() == open(single).read()


def _mismatch_worker(rank, world, port, in_path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = open(in_path).read() if rank == 0 else ""          # rank 1 sees an empty stdin
    try:
        sharded.sv_genotype_sharded(H.IN_BAM, sharded._Lines(text.splitlines(True), "<stdin>"), io.StringIO(),
                                    *_classic_args(), rank=rank, world=world, engine=H.oracle_engine)
        verdict = "ran"
    except RuntimeError as e:
        verdict = "refused" if "same VCF" in str(e) else "other: %s" % e
    open(os.path.join(out_dir, "rank%d" % rank), "w").write(verdict)
    dist.destroy_process_group()


def test_ranks_with_different_inputs_are_refused(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_mismatch_worker, args=(2, _free_port(), H.IN_VCF, str(tmp_path)), nprocs=2, join=True)
    assert [open(str(tmp_path / ("rank%d" % r))).read() for r in (0, 1)] == ["refused", "refused"]


def test_pairing_replay_matches_the_vcf_model_on_random_bodies():
    """bnd_pairs() replays parsers.py:155-178 on raw lines; the Vcf model is the same logic on parsed
    variants.  Random bodies: mates in either order, missing partners, several pairs interleaved."""
    import random
    from svtyper_amd.vcf import Variant, Vcf
    rng = random.Random(7)
    for trial in range(200):
        n_pairs = rng.randint(0, 6)
        lines = []
        for k in range(n_pairs):
            a, b = "p%d_1" % k, "p%d_2" % k
            mates = [_bnd("1", 1000 + k, a, b, "N[1:5000["), _bnd("1", 5000 + k, b, a, "]1:1000]N")]
            if rng.random() < 0.25:
                mates.pop(rng.randrange(2))                      # a mate whose partner never shows up
            lines += mates
        lines += ["1\t%d\tv%d\tN\t<DEL>\t0\t.\tSVTYPE=DEL;END=%d;CIPOS=0,0;CIEND=0,0\n" % (100 + i, i, 900 + i)
                  for i in range(rng.randint(0, 12))]
        rng.shuffle(lines)
        vcf = Vcf()
        want = {}
        first_index = {}
        for idx, line in enumerate(lines):
            var = Variant(line.rstrip().split("\t"), vcf)
            if var.get_svtype() != "BND":
                continue
            first_index[var.var_id] = idx
            bp = vcf.get_variant_breakpoints(var, 1e10)
            if bp is not None:
                want[idx] = first_index[bp["id"]]
                vcf._bnd_first.pop(bp["id"])
        assert sharded.bnd_pairs(lines) == want
        for world in (1, 2, 3, 7):
            plan = sharded.plan_shards(lines, world)
            assert sorted(i for p in plan for i in p) == list(range(len(lines)))
            owner = {i: r for r, p in enumerate(plan) for i in p}
            assert all(owner[s] == owner[f] for s, f in want.items())
            # apart from first mates that moved to their partner's rank, ranks hold file-order ranges
            stay = [i for i in range(len(lines)) if i not in set(want.values())]
            assert [owner[i] for i in stay] == sorted(owner[i] for i in stay)
