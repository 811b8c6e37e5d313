"""Shared host machinery of the two drivers (classic.sv_genotype, singlesample.sso_genotype).

The reference genotypes one (variant, sample) at a time inside its VCF loop; here the loop is
split so the likelihood path can run as ONE device batch:

    pass 1  VCF -> breakpoints -> BAM fetch -> fragments -> 16-byte evidence records (host)
    device  tally -> bayes_gt -> GT/GQ/SQ for every (breakpoint, sample) unit (HIP kernel)
    pass 2  results -> FORMAT fields / QUAL -> VCF text (host)

`engine` is the callable that turns an EvidenceBatch into Results.  The product engine is the
HIP library (`HipEngine`); there is no CPU engine in this package -- the tests plug the oracle in
through the same seam.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from itertools import chain
from operator import itemgetter, methodcaller

from typing import Callable, Dict, List, Optional, Tuple

from . import evidence as ev
from .evidence import EvidenceBatch, Results
from .fragments import SamFragment
from .library import Sample
from .packer import BatchBuilder, pack_fragments, unit_header

Engine = Callable[[EvidenceBatch, int], Results]

Z = 3            # fetch / straddle flank in standard deviations (classic.py:183)
SPLIT_SLOP = 3   # slop around the breakpoint for split reads (classic.py:184)
MIN_LIB_PREVALENCE = 1e-3
_READER_TURN = threading.Lock()   # svt_bam_summarise starts a pool of threads that fills the host: one call at a time


class HipEngine:
    """EvidenceBatch -> Results on one MI355X through the C ABI (include/svtyper_hip.h)."""

    def __init__(self, device: int = 0):
        from . import hip
        hip.load()   # raises if libsvtyper_hip.so is missing
        if hip.device_count() <= device:
            raise hip.SvtyperHipError(
                "no MI355X visible as device %d: svtyper_amd has no CPU fallback for the likelihood path" % device)
        self._hip = hip
        self.device = device

    # `site_qual=(n_samples, incoming QUAL)` is accepted: Results.site_qual = QUAL over a site's samples (classic.py:485,498).
    # The drivers' QUAL is summed on the HOST over the SQ the host libm refined from the bit-exact GL (hip.site_qual_host), so that
    # the printed %0.2f is the reference's byte for byte; the device-side sum (svt_batch_site_qual, over the device's own SQ,
    # <= 5e-13 away) serves callers that keep the records on the device (bench.py's configs[4] legs).
    supports_site_qual = True

    # a joint run may hand its units over SAMPLE-major (unit = sample * n_sites + site: the order the per-sample readers
    # produce them in); the pass writes the result records site-major (svt_batch_result_order), so nobody interleaves
    # 16 bytes per fragment on the host
    accepts_sample_major = True

    def __call__(self, batch: EvidenceBatch, flags: int = 0, site_qual=None, sample_major: int = 0) -> Results:
        if sample_major > 1:
            from .evidence import SegmentedBatch
            make = self._hip.DeviceBatch.from_segments if isinstance(batch, SegmentedBatch) else self._hip.DeviceBatch
            with make(batch, self.device, flags) as d:
                d.result_order(sample_major)
                d.genotype(sync=True)
                res = self._hip.host_sq(d.results())
            if site_qual is not None:      # classic.py:485,498 over the refined SQ, on the host, as hip._finish does
                res.site_qual = self._hip.site_qual_host(res, site_qual[0], site_qual[1])
            return res
        res = self._hip.genotype_batch(batch, device=self.device, flags=flags, site_qual=site_qual)
        return res if site_qual is not None else self._hip.host_sq(res)   # SQ as the reference's libm gives it from GL

    def genotype_fragments(self, fbatch, flags: int = 0, site_qual=None) -> Results:
        """geometry="device": fragment summaries in, both stages on the GPU."""
        res = self._hip.genotype_fragments(fbatch, device=self.device, flags=flags, site_qual=site_qual)
        return res if site_qual is not None else self._hip.host_sq(res)


def _site_qual_kw(engine, n_samples: int, site_quals) -> dict:
    if site_quals is None or not getattr(engine, "supports_site_qual", False):
        return {}
    import numpy as np
    return {"site_qual": (n_samples, np.asarray(site_quals, dtype=np.float64))}


def default_engine() -> Engine:
    return HipEngine(0)


MAX_BATCH_LIBS = 65536   # libraries one device batch can name: svt_record carries a 16-bit index into svt_evidence_batch.libs (ABI 18)


def library_groups(samples: List[Sample]) -> List[List[int]]:
    """Consecutive samples whose libraries fit ONE device batch.  The reference's `-B a.bam,b.bam,...` list is unbounded
    (svtyper/classic.py:145-158; read group -> library: parsers.py:432-447) while an evidence record names its library with
    sixteen bits (65 536 libraries per batch since ABI 18: a 130-sample cohort of 260 libraries is ONE batch), so a joint run
    over more libraries than that is several device batches -- one per group of samples, each with its own library table and
    indices local to it -- whose result records are put back site-major over all samples (units are independent; QUAL, the
    one quantity across a site's samples, is summed on the host afterwards)."""
    groups: List[List[int]] = []
    cur: List[int] = []
    n = 0
    for k, s in enumerate(samples):
        c = len(s.lib_dict)
        if c > MAX_BATCH_LIBS:
            raise ValueError("sample %s has %d libraries: more than the %d one evidence batch can name" % (s.name, c, MAX_BATCH_LIBS))
        if cur and n + c > MAX_BATCH_LIBS:
            groups.append(cur)
            cur, n = [], 0
        cur.append(k)
        n += c
    if cur:
        groups.append(cur)
    return groups


def _with_site_qual(res: Results, kw: dict) -> Results:
    """QUAL over the samples of every site (classic.py:216-217,485,498) from merged, SQ-refined records"""
    if "site_qual" in kw:
        from . import hip
        res.site_qual = hip.site_qual_host(res, kw["site_qual"][0], kw["site_qual"][1])
    return res


def resolve_reader(reader: Optional[str]) -> str:
    """`reader=None` (the drivers' default, i.e. what a caller with the reference's positional arguments gets): the C++
    reader of libsvtyper_hip.so when the library is there -- fetch, fragment assembly and the geometry predicates in its
    threads, VCF lines in bulk --, else the portable Python reader.  Same output bytes either way."""
    if reader is not None:
        return reader
    try:
        from . import hip
        return "native" if hasattr(hip.load(), "svt_bam_evidence") else "python"
    except Exception:
        return "python"


class UnitCollector:
    """Packs (breakpoint, sample) units of one chunk of variants and remembers where each went.

    geometry="host":   fragments -> evidence records here (packer.py), likelihood on the device;
    geometry="device": fragments -> breakpoint-independent summaries here (geometry.py), predicates
                       and likelihood on the device (needs an engine with genotype_fragments)."""

    def __init__(self, samples: List[Sample], split_weight: float, disc_weight: float, min_aligned: int,
                 geometry: str = "host"):
        if geometry not in ("host", "device"):
            raise ValueError("geometry must be 'host' or 'device'")
        self.geometry = geometry
        self.samples = samples
        self.min_aligned = min_aligned
        self.split_weight = split_weight
        self.disc_weight = disc_weight
        # one device batch per group of samples (library_groups): its library table, the index of every library in it, and per
        # sample the svt_unit.libs hint (a sample's libraries are contiguous in its group's table)
        self.groups = library_groups(samples)
        self.group_of = [g for g, members in enumerate(self.groups) for _ in members]
        self.group_tables: List[list] = []
        self.lib_index: Dict[int, int] = {}
        self.sample_libs = []
        for members in self.groups:
            tables = []
            for k in members:
                first = len(tables)
                for lib in samples[k].lib_dict.values():
                    self.lib_index[id(lib)] = len(tables)
                    tables.append(lib.table())
                self.sample_libs.append(ev.unit_libs(first, len(tables) - first))   # (0 = no hint when it does not fit)
            self.group_tables.append(tables)
        self.lib_tables = [t for tables in self.group_tables for t in tables]
        self._reset()

    def _reset(self):
        self.builders = [self._new_builder(tables) for tables in self.group_tables]
        self.slots: List[List[int]] = [[] for _ in self.groups]      # per group: where its units go in the order they were added
        self.n_units = 0

    def _new_builder(self, tables):
        if self.geometry == "device":
            from .geometry import FragmentBatchBuilder
            return FragmentBatchBuilder(tables, self.split_weight, self.disc_weight, self.min_aligned, SPLIT_SLOP)
        return BatchBuilder(tables, self.split_weight, self.disc_weight)

    def add(self, breakpoint: dict, sample_index: int, fragments: Optional[Dict[str, SamFragment]],
            skip: bool = False) -> int:
        g = self.group_of[sample_index]
        if self.geometry == "device":
            from .geometry import breakpoint_record, summarise_fragments
            tid_of = self.samples[sample_index].bam.gettid
            frs = None
            if fragments and not skip:
                frs = summarise_fragments(fragments, breakpoint, self.lib_index, tid_of)
            self.builders[g].add(breakpoint_record(breakpoint, tid_of, sample_index, skip, self.sample_libs[sample_index]), frs)
        else:
            unit = unit_header(breakpoint, sample_index, skip, self.sample_libs[sample_index])
            recs = None
            if fragments and not skip:
                recs = pack_fragments(fragments, breakpoint, self.lib_index, self.min_aligned, SPLIT_SLOP)
            self.builders[g].add(unit, recs)
        self.slots[g].append(self.n_units)
        self.n_units += 1
        return self.n_units - 1

    def __len__(self):
        return self.n_units

    def take(self, engine: Engine, flags: int, site_quals=None):
        """Detach the units collected so far as a job (a callable returning their Results); the collector is
        empty again and can be filled while the job runs on another thread (ChunkPipeline).  `site_quals`
        (incoming QUAL of every site, units site-major over self.samples) asks an engine that can for
        Results.site_qual."""
        builders, slots, n_units = self.builders, self.slots, self.n_units
        self._reset()
        geometry = self.geometry
        kw = _site_qual_kw(engine, len(self.samples), site_quals)

        def run_one(builder, **kw) -> Results:
            batch = builder.build()
            if batch.n_units == 0:
                return Results.empty(0)
            if geometry == "device":
                if not hasattr(engine, "genotype_fragments"):
                    raise TypeError("geometry='device' needs an engine with genotype_fragments (the HIP engine)")
                return engine.genotype_fragments(batch, flags, **kw)
            return engine(batch, flags, **kw)

        def job() -> Results:
            if len(builders) == 1:
                return run_one(builders[0], **kw)
            import numpy as np
            out = Results.empty(n_units)
            for builder, where in zip(builders, slots):      # one device batch per group of samples, results back in add() order
                if where:
                    out.rec[np.asarray(where, dtype=np.int64)] = run_one(builder).rec
            return _with_site_qual(out, kw)
        return job

    def run(self, engine: Engine, flags: int) -> Results:
        return self.take(engine, flags)()


class NativeUnitCollector:
    """reader="native": no per-read Python objects.  Sites are only recorded here; at run() the C++
    reader (svt_bam_summarise, one pool of threads per sample) fetches and condenses the fragments of
    every (site, sample) unit and the summaries go straight to the device geometry + likelihood stages."""

    def __init__(self, samples: List[Sample], native_bams, split_weight: float, disc_weight: float,
                 min_aligned: int, count_mode: int, max_reads, n_threads: int = 0, geometry: str = "reader"):
        """`geometry`: where the breakpoint-dependent predicates (parsers.py:785-857,1122-1215) are evaluated --
        "reader" (default): in the C++ reader's threads, which hands over 16-byte evidence records (svt_bam_evidence) for
        the canonical route of ANY engine; "device": 128-byte fragment summaries go to the device's geometry stage
        (svt_bam_summarise -> svt_batch_create_from_fragments, the HIP engine only).  Same records either way."""
        if geometry not in ("reader", "device"):
            raise ValueError("geometry must be 'reader' or 'device'")
        self.geometry = geometry
        self.samples = samples
        self.bams = native_bams
        self.min_aligned = min_aligned
        self.count_mode = count_mode
        self.max_reads = max_reads
        self.n_threads = n_threads or int(os.environ.get("SVT_READER_THREADS", "0"))   # 0 = the library's default
        self.split_weight = split_weight
        self.disc_weight = disc_weight
        # one device batch per group of samples (library_groups): per group its library table; per sample the read groups
        # with their library's index in the group's table (-1: not active) and the svt_unit.libs hint
        self.groups = library_groups(samples)
        self.group_of = [g for g, members in enumerate(self.groups) for _ in members]
        self.group_tables: List[list] = []
        self.rg_tables = []          # per sample: (read group ids, library index or -1)
        self.sample_libs = []        # per sample: svt_unit.libs hint
        for members in self.groups:
            tables = []
            for k in members:
                s = samples[k]
                base = len(tables)
                libs = list(s.lib_dict.values())
                tables.extend(lib.table() for lib in libs)
                self.sample_libs.append(ev.unit_libs(base, len(libs)))   # (0 = no hint when it does not fit)
                rgs = list(s.rg_to_lib.keys())
                idx = [base + libs.index(s.rg_to_lib[rg]) if s.rg_to_lib[rg].name in s.active_libs else -1 for rg in rgs]
                self.rg_tables.append((rgs, idx))
            self.group_tables.append(tables)
        self.lib_tables = [t for tables in self.group_tables for t in tables]
        self.sites: List[dict] = []
        self.site_arrays: list = []      # bulk_vcf.SiteArrays blocks, in front of the dict sites

    def add_site(self, breakpoint: dict) -> int:
        """Returns the index of the site's first unit (its samples follow in order)."""
        self.sites.append(breakpoint)
        return (self._n_array_sites() + len(self.sites) - 1) * len(self.samples)

    def add_site_arrays(self, arrays) -> int:
        """The sites of a parsed block as arrays (bulk_vcf.SiteArrays: no dict per site).  They come in front of whatever
        add_site() adds to the same batch; returns the index of their first unit."""
        if self.sites:
            raise ValueError("add_site_arrays() after add_site() in one batch")
        first = self._n_array_sites() * len(self.samples)
        self.site_arrays.append(arrays)
        return first

    def _n_array_sites(self) -> int:
        return sum(len(a) for a in self.site_arrays)

    def __len__(self):
        return (self._n_array_sites() + len(self.sites)) * len(self.samples)

    def take(self, engine: Engine, flags: int, site_quals=None):
        """Detach the sites recorded so far as a job (see UnitCollector.take).  The sites' fields become arrays HERE, on the
        caller's thread (Python: it holds the GIL) -- under ChunkPipeline that is while the reader and the device work on
        the chunk before; the job itself is C++ and HIP calls only."""
        from .bulk_vcf import SiteArrays
        sites, self.sites = self.sites, []
        blocks, self.site_arrays = self.site_arrays, []
        kw = _site_qual_kw(engine, len(self.samples), site_quals)
        if (sites or blocks) and self.geometry == "device" and not hasattr(engine, "genotype_fragments"):
            raise TypeError("reader='native' with geometry='device' needs an engine with genotype_fragments (the HIP engine)")
        t_begin = time.perf_counter()
        prepared = self._prepare(SiteArrays.concat(blocks + [SiteArrays.from_dicts(sites)]))
        prep_s = time.perf_counter() - t_begin
        return lambda: self._run_prepared(prepared, engine, flags, kw, prep_s)

    def run(self, engine: Engine, flags: int) -> Results:
        return self.take(engine, flags)()

    def _prepare(self, sites):
        """Per sample the (svt_breakpoint[], svt_fetch_unit[]) arrays of the sites (bulk_vcf.SiteArrays)."""
        import numpy as np
        from .geometry import BREAKPOINT_DTYPE
        from .native_reads import FETCH_DTYPE
        n_sites = len(sites)
        if n_sites == 0:
            return []
        pos, ci, rev, svt = sites.pos, sites.ci, sites.reverse, sites.svtype
        vlen = np.where(svt == ev.SVTYPE_CODE["DEL"], sites.var_length, 0)
        clip = lambda x: np.clip(x, -2**31, 2**31 - 1)
        prepared = []
        for k, (sample, nbam) in enumerate(zip(self.samples, self.bams)):
            tid = np.array([nbam.gettid(c) for c in sites.names], dtype=np.int64)[sites.chrom]      # [site, side]
            if (tid < 0).any():
                i, side = (int(x[0]) for x in np.nonzero(tid < 0))
                raise KeyError("chromosome %s of variant line %d of the batch is not in %s"
                               % (sites.names[int(sites.chrom[i, side])], i + 1, nbam.filename))
            bps = np.zeros(n_sites, BREAKPOINT_DTYPE)
            bps["tid_a"], bps["tid_b"] = tid[:, 0], tid[:, 1]
            bps["pos_a"], bps["pos_b"] = clip(pos[:, 0]), clip(pos[:, 1])
            bps["ci_a"], bps["ci_b"] = clip(ci[:, 0:2]), clip(ci[:, 2:4])
            bps["var_length"] = clip(vlen)
            bps["svtype"], bps["flags"], bps["sample"] = svt, rev, k
            bps["reserved"][:, 0] = self.sample_libs[k]
            # fetch windows: pos + ci -+ (mean + 3 sd), clamped to the chromosome (classic.py:73-81;
            # singlesample.py:139-156 truncates to int, pysam truncates classic's float bounds the same way)
            flank = sample.get_fetch_flank(Z)
            length = np.array(nbam.lengths, dtype=np.float64)[tid]
            lo = np.maximum(pos + ci[:, [0, 2]] - flank, 0.0)
            hi = np.minimum(pos + ci[:, [1, 3]] + flank, length)
            win = np.zeros(n_sites, FETCH_DTYPE)
            win["tid_a"], win["tid_b"] = tid[:, 0], tid[:, 1]
            win["lo_a"], win["lo_b"] = lo[:, 0].astype(np.int64), lo[:, 1].astype(np.int64)
            win["hi_a"], win["hi_b"] = hi[:, 0].astype(np.int64), hi[:, 1].astype(np.int64)
            prepared.append((bps, win))
        return prepared

    def _run_prepared(self, prepared, engine: Engine, flags: int, kw: dict, prep_s: float = 0.0) -> Results:
        import numpy as np
        from .geometry import FragmentBatch
        n_samp = len(self.samples)
        if not prepared:
            return Results.empty(0)
        per_sample = []
        trace = os.environ.get("SVT_TRACE") is not None
        t_begin = time.perf_counter()
        lap = (lambda what: sys.stderr.write("[NativeUnitCollector] %-22s %8.1f ms\n" % (what, (time.perf_counter() - t_begin) * 1e3))) if trace else (lambda what: None)
        if trace:
            sys.stderr.write("[NativeUnitCollector] %-22s %8.1f ms (at take(), on the caller's thread)\n" % ("site arrays (python)", prep_s * 1e3))
        if self.geometry == "reader":
            return self._run_records(prepared, engine, flags, kw, lap)
        for k, (nbam, (bps, win)) in enumerate(zip(self.bams, prepared)):
            rgs, idx = self.rg_tables[k]
            with _READER_TURN:      # (two chunks in flight under ChunkPipeline: one reads, the other is on the device)
                off, frags, skipped = nbam.summarise(win, bps, rgs, idx, self.max_reads, self.count_mode, self.n_threads)
            lap("svt_bam_summarise")
            bps["flags"] |= np.where(skipped != 0, 4, 0).astype(np.uint8)   # SVT_BP_SKIP
            fb = FragmentBatch(off, bps, frags, self.group_tables[self.group_of[k]], self.split_weight, self.disc_weight,
                               self.min_aligned, SPLIT_SLOP)
            if n_samp == 1:
                res = engine.genotype_fragments(fb, flags, **kw)
                lap("device stages")
                return res
            # one device batch per sample: the summaries go to the GPU as the reader produced them (no
            # re-interleaving of ~13 KB per unit on the host)
            per_sample.append(engine.genotype_fragments(fb, flags).rec)
        # units are site-major, sample-minor for the caller: interleave the 128-byte result records
        res = Results(np.stack(per_sample, axis=1).reshape(-1))
        if "site_qual" in kw:      # QUAL over the samples of a site, in -B order (classic.py:216-217,485,498):
            q = np.array(kw["site_qual"][1], dtype=np.float64, copy=True)   # the same adds, vectorised over the sites
            for rec in per_sample:
                gt = rec["gt"]
                q = np.where(gt >= 0, q + rec["sq"], np.where(gt == ev.GT_BLANK, 0.0, q))
            res.site_qual = q
        return res


    def _run_records(self, prepared, engine: Engine, flags: int, kw: dict, lap) -> Results:
        """geometry="reader": evidence records straight from the reader, ONE canonical batch over the samples of a library
        group (units site-major, sample-minor: what UnitCollector builds) -- one batch in all unless the run names more
        libraries than a batch can (library_groups)."""
        import numpy as np
        n_samp = len(self.samples)
        n_sites = int(prepared[0][0].shape[0])
        per_sample = []
        for k, (nbam, (bps, win)) in enumerate(zip(self.bams, prepared)):
            rgs, idx = self.rg_tables[k]
            flank = [float(t.mean) + float(t.sd) * 3 for t in self.group_tables[self.group_of[k]]]   # (svt_batch_create_from_fragments' v_nondel)
            with _READER_TURN:
                off, recs, skipped = nbam.evidence(win, bps, rgs, idx, self.max_reads, self.count_mode, flank, self.min_aligned,
                                                   SPLIT_SLOP, self.n_threads)
            lap("svt_bam_evidence")
            units = np.zeros(n_sites, ev.UNIT_DTYPE)
            units["var_length"] = np.where(bps["svtype"] == ev.SVTYPE_CODE["DEL"], bps["var_length"], 0)
            units["pos_delta"] = np.clip(bps["pos_b"].astype(np.int64) - bps["pos_a"].astype(np.int64), -2**31, 2**31 - 1)   # classic.py:339
            units["sample"], units["svtype"] = k, bps["svtype"]
            units["flags"] = np.where(skipped != 0, ev.UNIT_SKIP, 0)
            units["libs"] = self.sample_libs[k]
            per_sample.append((off, units, recs))
        if len(self.groups) == 1:
            return self._run_group(per_sample, self.group_tables[0], n_sites, engine, flags, kw, lap)
        out = Results.empty(n_sites * n_samp)
        grid = out.rec.reshape(n_sites, n_samp)
        for members, tables in zip(self.groups, self.group_tables):      # one device batch per group, its columns of the site x sample grid
            res = self._run_group([per_sample[k] for k in members], tables, n_sites, engine, flags, {}, lap)
            grid[:, members[0]:members[-1] + 1] = res.rec.reshape(n_sites, len(members))
        return _with_site_qual(out, kw)

    def _run_group(self, per_sample, tables, n_sites: int, engine: Engine, flags: int, kw: dict, lap) -> Results:
        """the samples of one library group -> their result records, site-major over the group's samples"""
        import numpy as np
        n_samp = len(per_sample)
        if n_samp == 1:
            off, units, recs = per_sample[0]
            batch = EvidenceBatch(off, units, recs, tables, self.split_weight, self.disc_weight)
        elif getattr(engine, "accepts_sample_major", False):
            # the engine takes the units as the readers left them, sample after sample: the unit arrays concatenated (24 bytes
            # per unit), every sample's records from where its reader left them (evidence.SegmentedBatch ->
            # svt_batch_create_segments), the result records written site-major by the pass -- no per-record work on the host
            from .evidence import SegmentedBatch
            counts = np.concatenate([np.diff(p[0].astype(np.int64)) for p in per_sample])
            off = np.zeros(n_sites * n_samp + 1, np.uint64)
            np.cumsum(counts, out=off[1:].view(np.int64))
            batch = SegmentedBatch(off, np.concatenate([p[1] for p in per_sample]), [p[2] for p in per_sample],
                                   tables, self.split_weight, self.disc_weight)
            lap("sample-major unit arrays")
            res = engine(batch, flags, sample_major=n_samp, **kw)
            lap("engine (canonical batch, sample-major)")
            return res
        else:
            # site-major, sample-minor: unit (site i, sample k) takes the k-th slice of site i -- counts interleaved, records
            # copied slice by slice through one fancy index (16 bytes per fragment, not the 128 of a summary)
            counts = np.stack([np.diff(p[0].astype(np.int64)) for p in per_sample], axis=1)          # [site, sample]
            off = np.zeros(n_sites * n_samp + 1, np.uint64)
            np.cumsum(counts.reshape(-1), out=off[1:].view(np.int64))
            units = np.stack([p[1] for p in per_sample], axis=1).reshape(-1)
            recs = np.empty(int(off[-1]), ev.RECORD_DTYPE)
            dst0 = off[:-1].astype(np.int64).reshape(n_sites, n_samp)
            for k, (soff, _u, srecs) in enumerate(per_sample):
                n_k = counts[:, k]
                if int(n_k.sum()):
                    dst = np.repeat(dst0[:, k] - soff[:-1].astype(np.int64), n_k) + np.arange(int(n_k.sum()), dtype=np.int64)
                    recs[dst] = srecs
            batch = EvidenceBatch(off, units, recs, tables, self.split_weight, self.disc_weight)
            lap("site-major interleave")
        res = engine(batch, flags, **kw)
        lap("engine (canonical batch)")
        return res


SVTYPER_FORMAT_KEYS = ("GT", "GQ", "SQ", "GL", "DP", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP", "AB")


class SampleColumnWriter:
    """Sample columns of a whole chunk as text in one native call (svt_format_results) instead of one dict,
    fifteen set_format calls and a join per sample.  It applies to a variant when every sample column of the
    VCF is written by this run, in this order, and carries nothing but a GT so far (sites-only input, the
    svtools workflow); anything else -- other samples' columns, FORMAT keys of other tools with values --
    keeps the general Genotype path, which prints the same bytes."""

    def __init__(self, vcf, sample_names, skipped_as_dots: bool):
        self._skipped_as_dots = skipped_as_dots
        self._ours = frozenset(SVTYPER_FORMAT_KEYS)
        self.enabled = (list(vcf.sample_list) == list(sample_names)
                        and all(k in vcf.format_rank for k in SVTYPER_FORMAT_KEYS))
        self.fields = sorted(SVTYPER_FORMAT_KEYS, key=lambda k: vcf.format_rank.get(k, 0))
        self.format_string = ":".join(self.fields)

    def eligible(self, variant) -> bool:
        if not self.enabled:
            return False
        ours = self._ours
        for f in variant.active_formats:
            if f not in ours:
                return False
        return variant.only_default_genotypes()

    def columns(self, results) -> list:
        """one string per unit of `results`"""
        from . import hip
        return hip.format_results(results, self.fields, self._skipped_as_dots)


# Variant lines per block of the bulk route = per device batch (SVT_BULK_BLOCK_SITES).  Small enough that the stages of
# neighbouring blocks overlap (parse / reader / device / text: ChunkPipeline), large enough that a block is not its fixed costs;
# counted in sites, not units: every sample of a joint run is a reader call of its own per block.  Measured on the fixture x 100
# through sso_genotype: 4 096 -> 46 ms, 16 384 -> 37 ms, 65 536 -> 46 ms, one block -> 49 ms (profiles/r06_block_sweep.txt).
BULK_BLOCK_SITES = 16_384


def text_blocks(first: str, source, n_samples: int):
    """Blocks of whole lines from `source` (a text file object) behind `first` (a line already read), each about
    BULK_BLOCK_SITES lines, judged by the length of the first one."""
    chars = block_chars(len(first))
    carry = first
    while True:
        data = source.read(chars)
        if not data:
            if carry:
                yield carry
            return
        if not data.endswith("\n"):
            data += source.readline()
        yield carry + data
        carry = ""


def block_chars(first_line_chars: int) -> int:
    sites = int(os.environ.get("SVT_BULK_BLOCK_SITES", BULK_BLOCK_SITES))
    return min(max(sites * max(64, first_line_chars), 1 << 10), 256 << 20)


def split_lines(text: str) -> List[str]:
    """file.readlines() of a text: lines end at '\\n' only (str.splitlines also cuts at \\r, \\x0b, \\x1c, \\u2028 ...)"""
    import io
    return io.StringIO(text, newline="\n").readlines()


class BulkFeeder:
    """The drivers' bulk route (reader="native"): a block of variant lines -> breakpoint arrays in one native call
    (bulk_vcf.VcfParser) -> the reader and the device -> the block's output lines as one text (VcfChunk.emit).  No Variant
    object, breakpoint dict or join per line: svtyper/classic.py:219-278 / singlesample.py:577-652 for a block at a time.
    Lines the parser hands back (no / unsupported SVTYPE, sample columns with FORMAT values, numbers it does not read
    exactly as Python would ...) go through the driver's own per-line code (`handle_line`, `render_actions`) and are written
    at their place; when the parser stops in front of a BND line it cannot express, run() returns the lines not consumed
    and the driver carries on per line (pairing is stateful: pending_lines() are the first mates still waiting)."""

    def __init__(self, bulk, vcf, collector, pipe, engine, flags: int, n_samples: int, fast, qual_mode: int, max_ci_dist,
                 sum_quals: bool, skip_hash_lines: bool, handle_line, render_actions, write):
        self._bulk = bulk
        self._parser = bulk.VcfParser(vcf, max_ci_dist, sum_quals, skip_hash_lines)
        self._collector, self._pipe, self._engine, self._flags = collector, pipe, engine, flags
        self._n_samp, self._fast, self._qual_mode = n_samples, fast, qual_mode
        self._handle_line, self._render_actions, self._write = handle_line, render_actions, write
        self._skipped_as_dots = qual_mode == bulk.QUAL_CLASSIC
        # seconds on the caller's thread, by stage (the drivers' `stats=` dict gets them): the native parse (+ encode), the
        # per-line code for lines handed back, the site arrays per sample (collector.take), the native emit, decode + write
        self.laps = {"blocks": 0, "lines": 0, "sites": 0, "lines_handed_back": 0, "vcf_parse_s": 0.0, "per_line_python_s": 0.0,
                     "site_arrays_s": 0.0, "vcf_emit_s": 0.0, "decode_write_s": 0.0}

    def n_pending(self) -> int:
        return len(self.pending_lines())

    def pending_lines(self) -> List[str]:
        return self._parser.pending_lines()

    def run(self, blocks) -> Optional[List[str]]:
        """Every block of `blocks` (an iterator of texts of whole lines); None, or the lines of the current block the parser
        did not consume (the iterator then stands behind that block)."""
        for block in blocks:
            rest = self._block(block)
            if rest is not None:
                return rest
        return None

    def _block(self, text: str) -> Optional[List[str]]:
        import numpy as np
        bulk = self._bulk
        enc = bulk.TEXT_ENCODING
        laps = self.laps
        t0 = time.perf_counter()
        raw = text.encode(*enc)
        chunk, used = self._parser.parse(raw)
        t1 = time.perf_counter()
        n_bulk_units = chunk.n_sites * self._n_samp
        if chunk.n_sites:
            self._collector.add_site_arrays(chunk.sites)
        actions = []          # (line index, action) of the lines handed back, in order
        begin = chunk.line_begin
        for i in np.nonzero(chunk.line_kind == bulk.LINE_PYTHON)[0].tolist():
            action = self._handle_line(raw[begin[i]:begin[i + 1]].decode(*enc), n_bulk_units)
            if action is not None:
                actions.append((i, action))
        t2 = time.perf_counter()
        job = self._collector.take(self._engine, self._flags)
        t3 = time.perf_counter()
        laps["blocks"] += 1
        laps["lines"] += chunk.n_lines
        laps["sites"] += chunk.n_sites
        laps["lines_handed_back"] += len(actions)
        laps["vcf_parse_s"] += t1 - t0
        laps["per_line_python_s"] += t2 - t1
        laps["site_arrays_s"] += t3 - t2
        self._pipe.submit(job, lambda results: self._write_block(chunk, actions, results))
        return None if used == len(raw) else split_lines(raw[used:].decode(*enc))

    def _write_block(self, chunk, actions, results) -> None:
        import numpy as np
        bulk = self._bulk
        enc = bulk.TEXT_ENCODING
        fast = self._fast
        t0 = time.perf_counter()
        text, off = chunk.emit(results, self._n_samp, self._qual_mode, fast.fields, self._skipped_as_dots, fast.format_string)
        t1 = time.perf_counter()
        self.laps["vcf_emit_s"] += t1 - t0
        if not actions:
            if text:
                self._write(text.decode(*enc))
            chunk.close()
            self.laps["decode_write_s"] += time.perf_counter() - t1
            return
        # lines handed back sit between the sites' lines: sites_before[i] = sites written by lines in front of line i
        sites_before = np.concatenate([[0], np.cumsum(chunk.line_kind == bulk.LINE_SITE)])
        own = Results(results.rec[chunk.n_sites * self._n_samp:])          # their units lie behind the parser's
        cursor = 0
        for (i, _), rendered in zip(actions, self._render_actions(own, [a for _, a in actions])):
            s = int(sites_before[i])
            if s > cursor:
                self._write(text[off[cursor]:off[s]].decode(*enc))
                cursor = s
            self._write(rendered)
        if cursor < chunk.n_sites:
            self._write(text[off[cursor]:off[chunk.n_sites]].decode(*enc))
        chunk.close()


class ChunkPipeline:
    """Chunk-level pipelining of the drivers (svtyper/singlesample.py:710-762 re-cast).  The job of chunk k -- C++
    fetch/summarise or packing, H2D, kernels, D2H, all outside the GIL -- runs on a worker thread while the caller's thread
    parses the VCF lines of chunk k+1, turns them into arrays and formats the columns of an earlier chunk; with two workers
    the device stages of chunk k also run under the reader of chunk k+1 (the readers themselves take turns:
    NativeUnitCollector holds a lock around svt_bam_summarise, whose own threads already fill the host).  `on_done(results)`
    is called on the caller's thread, in submission order: chunk k-depth's when chunk k is submitted, the rest at close()."""

    def __init__(self, overlap: bool = True, depth: Optional[int] = None):
        """`depth`: jobs in flight (default 2, SVT_PIPELINE_DEPTH): chunk k on the device while chunk k + 1 is with the reader;
        depth + 1 chunks' host and device buffers are alive at once."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        if depth is None:
            depth = int(os.environ.get("SVT_PIPELINE_DEPTH", "2"))
        self._depth = max(1, int(depth))
        self._pool = ThreadPoolExecutor(self._depth) if overlap else None
        self._pending = deque()

    def submit(self, job, on_done) -> None:
        if self._pool is None:
            on_done(job())
            return
        self._pending.append((self._pool.submit(job), on_done))
        while len(self._pending) > self._depth:
            self._drain_one()

    def _drain_one(self) -> None:
        fut, on_done = self._pending.popleft()
        on_done(fut.result())

    def close(self) -> None:
        try:
            while self._pending:
                self._drain_one()
        finally:
            if self._pool is not None:
                self._pool.shutdown(wait=True)


def add_read_to(fragments: Dict[str, SamFragment], read, lib):
    frag = fragments.get(read.query_name)
    if frag is None:
        fragments[read.query_name] = SamFragment(read, lib)
    else:
        frag.add_read(read)


def fetch_window(sample: Sample, chrom: str, pos: int, ci, as_int: bool) -> Tuple[str, float, float]:
    """Fetch region of one breakend: pos + ci +- (mean + 3 sd), clamped to the chromosome
    (classic.py:73-81; singlesample.py:139-156 truncates to int)."""
    flank = sample.get_fetch_flank(Z)
    chrom_length = sample.bam.lengths[sample.bam.gettid(chrom)]
    lo = max(pos + ci[0] - flank, 0)
    hi = min(pos + ci[1] + flank, chrom_length)
    if as_int:
        lo, hi = int(lo), int(hi)
    return chrom, lo, hi
