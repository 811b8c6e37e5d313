"""Synthetic read-evidence batches at the *evidence-record* level.

The generator draws, per (breakpoint, sample) unit, a true genotype and then per
read-fragment the evidence bits / MAPQs / outer-span lengths the reference's geometry
predicates would have produced, with the marginal frequencies measured on the
reference's own fixture: ~100 fragments per unit (18..183),
99.6 % two-primary fragments, MAPQ 60 for 94 %, one split candidate on 14.6 % of the
fragments (56 % soft-clip-only), outer spans drawn from the library's insert-size
histogram (shifted by the variant length for alt-supporting pairs).

Everything is vectorised numpy and deterministic in (seed, config).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import evidence as ev
from .evidence import EvidenceBatch, LibraryTable, RECORD_DTYPE, UNIT_DTYPE

BASE_SEED = 20260927


def normal_library(mu: float = 320.0, sigma: float = 80.0, n: int = 1_000_000, seed: int = 7,
                   name: str = "synthetic") -> LibraryTable:
    """A rounded N(mu, sigma) insert-size library truncated to [1, mu + 10 sigma]."""
    rng = np.random.default_rng(seed)
    x = np.rint(rng.normal(mu, sigma, n)).astype(np.int64)
    x = x[(x >= 1) & (x <= int(mu + 10 * sigma))]
    keys, counts = np.unique(x, return_counts=True)
    hist = {int(k): int(c) for k, c in zip(keys, counts)}
    tot = float(counts.sum())
    mean = float((keys * counts).sum() / tot)
    sd = float(np.sqrt((counts * (keys - mean) ** 2).sum() / tot))
    return LibraryTable.from_counter(hist, mean, sd, name)


def _mapq(rng, n):
    """MAPQ ~ {60: 94 %, 40: 1.75 %, 0: 0.3 %, rest uniform on 1..59}."""
    u = rng.random(n)
    q = np.full(n, 60, np.uint8)
    q[u < 0.0605] = 40  # overwritten below except for a 1.75 % band
    rest = u < 0.043
    q[rest] = rng.integers(1, 60, int(rest.sum()), dtype=np.uint8)
    q[u < 0.003] = 0
    return q


def _sample_hist(rng, lib: LibraryTable, n):
    cdf = np.cumsum(lib.hist, dtype=np.float64)
    cdf /= cdf[-1]
    return (np.searchsorted(cdf, rng.random(n), side="right") + lib.key_min).astype(np.int64)


def make_units(n_units: int, seed: int, libs: Sequence[LibraryTable], svtype_mix=(1.0, 0.0, 0.0, 0.0),
               mean_frags: float = 100.0, sd_frags: float = 25.0, min_frags: int = 18,
               max_frags: int = 183, sample: int = 0, lib_choices: Optional[Sequence[int]] = None,
               alt_af: Optional[np.ndarray] = None, split_weight: float = 1.0,
               disc_weight: float = 1.0, frac_empty: float = 0.0, frac_skip: float = 0.0,
               counts_only: bool = False) -> EvidenceBatch:
    """One chunk of synthetic units.  svtype_mix = probabilities of (DEL, DUP, INV, BND).
    counts_only: return just the records per unit (int64[n_units]) this seed would produce -- the first few draws of the
    generator, none of the per-record work -- so that a rank can find its shard of a workload without building all of it."""
    rng = np.random.default_rng(seed)
    libs = list(libs)
    if lib_choices is None:
        lib_choices = list(range(len(libs)))
    lib_choices = np.asarray(lib_choices, dtype=np.uint16)      # (sixteen bits of library index: ABI 18)

    # ---- units
    svtype = rng.choice(4, size=n_units, p=np.asarray(svtype_mix, float) / np.sum(svtype_mix)).astype(np.uint8)
    # var_length log-uniform on [50, 1e5] (includes the < 2*sd small-DEL gate)
    var_len = np.exp(rng.uniform(np.log(50.0), np.log(1e5), n_units)).astype(np.int64)
    if alt_af is None:
        genotype = rng.integers(0, 3, n_units)
    else:
        genotype = (rng.random(n_units) < alt_af).astype(np.int64) + (rng.random(n_units) < alt_af)
    F = np.clip(np.rint(rng.normal(mean_frags, sd_frags, n_units)), min_frags, max_frags).astype(np.int64)
    empty = rng.random(n_units) < frac_empty
    skip = rng.random(n_units) < frac_skip
    F[empty | skip] = 0
    if counts_only:
        return F
    units = np.zeros(n_units, UNIT_DTYPE)
    units["svtype"] = svtype
    units["var_length"] = np.where(svtype == 0, var_len, 0)
    # posB - posA after the +1 increments: DEL (o2 reverse) +1, DUP (o1 reverse) -1, INV 0
    units["pos_delta"] = var_len + np.where(svtype == 0, 1, np.where(svtype == 1, -1, 0))
    units["sample"] = sample
    units["flags"] = np.where(skip, ev.UNIT_SKIP, 0)
    rec_offset = np.zeros(n_units + 1, np.uint64)
    np.cumsum(F, out=rec_offset[1:])
    R = int(rec_offset[-1])

    # ---- per-record unit attributes
    unit_of = np.repeat(np.arange(n_units), F)
    g = genotype[unit_of]
    sv = svtype[unit_of]
    vlen = var_len[unit_of]
    is_del = sv == 0

    rec = np.zeros(R, RECORD_DTYPE)
    lib_idx = lib_choices[rng.integers(0, len(lib_choices), R)]
    two = rng.random(R) < 0.996
    mq_a = _mapq(rng, R)
    mq_b = np.where(two, _mapq(rng, R), 0).astype(np.uint8)
    rec["mapq_a"] = mq_a
    rec["mapq_b"] = mq_b
    flags = lib_idx.astype(np.uint32) << np.uint32(ev.REC_LIB_SHIFT)
    flags |= np.where(two, ev.REC_HAS_PAIR, 0).astype(np.uint32)

    # fragment class: alt-supporting with probability by genotype
    p_alt_frag = np.array([0.01, 0.45, 0.93])[g]
    alt_like = rng.random(R) < p_alt_frag

    # reference split-read evidence (is_ref_seq): 37 % of fragments, mostly ref-like ones
    u = rng.random(R)
    rs = u < np.where(alt_like, 0.08, 0.45)
    which = rng.random(R)
    rec["rs_a"] = np.where(rs & (which < 0.55), mq_a, 0)
    rec["rs_b"] = np.where(rs & two & (which > 0.45), mq_b, 0)

    # split candidates: one on 14.6 % of the fragments, two on 0.6 %; 56 % soft-clip-only.  A
    # record holds one candidate of each kind; MAPQs are gated by is_split_straddle()'s (L, R)
    u = rng.random(R)
    have = u < 0.152
    soft = have & (rng.random(R) < 0.56)
    second = (u < 0.006) & two          # second candidate of the OTHER kind in the same record
    for kind, present in (("seq", (have & ~soft) | (second & soft)), ("clip", soft | (second & have & ~soft))):
        sup_l = present & (rng.random(R) < np.where(alt_like, 0.85, 0.03))
        sup_r = present & (rng.random(R) < np.where(alt_like, 0.85, 0.03))
        ql = _mapq(rng, R)
        qr = _mapq(rng, R)
        if kind == "clip":  # the dummy other piece of a soft-clip-only candidate has MAPQ 0
            dummy_left = rng.random(R) < 0.5
            ql = np.where(dummy_left, 0, ql)
            qr = np.where(dummy_left, qr, 0)
        rec[kind + "_l"] = np.where(sup_l, ql, 0)
        rec[kind + "_r"] = np.where(sup_r, qr, 0)

    # paired-end evidence
    near = two & (rng.random(R) < 0.90)            # pair close enough to straddle something
    alt_st = near & alt_like & (rng.random(R) < 0.9)
    # ref-like pairs straddle one breakend (A or B); short variants are jumped by both
    ref_any = near & (~alt_like | (vlen < 500))
    side = rng.random(R)
    both = ref_any & ((vlen < 500) & (rng.random(R) < 0.6))
    ref_a = ref_any & (both | (side < 0.5))
    ref_b = ref_any & (both | (side >= 0.5))
    flags |= np.where(alt_st, ev.REC_ALT_STRADDLE, 0).astype(np.uint32)
    flags |= np.where(ref_a, ev.REC_REF_STRADDLE_A, 0).astype(np.uint32)
    flags |= np.where(ref_b, ev.REC_REF_STRADDLE_B, 0).astype(np.uint32)

    # outer span: concordant ~ hist, alt-supporting DEL pairs ~ hist + var_length; 14 % beyond the
    # histogram's range
    osp = np.zeros(R, np.int64)
    if len(libs) == 1:
        osp = _sample_hist(rng, libs[0], R)
    else:
        for li, lib in enumerate(libs):
            m = lib_idx == li
            if m.any():
                osp[m] = _sample_hist(rng, lib, int(m.sum()))
    osp = np.where(alt_like & is_del, osp + vlen, osp)
    far = rng.random(R) < 0.14
    osp = np.where(far, osp + rng.integers(400, 5000, R), osp)
    osp = np.where(two, osp, 0)
    rec["ospan_len"] = np.clip(osp, 0, 2**31 - 1).astype(np.int32)
    rec["flags"] = flags
    return EvidenceBatch(rec_offset, units, rec, libs, split_weight, disc_weight)


CONFIGS = {
    # BASELINE.json configs[1]: 100k synthetic DEL breakpoints, 1 library, ~200 reads/site
    "c2_del_100k": dict(n_units=100_000, svtype_mix=(1, 0, 0, 0), config_no=2),
    # configs[2]: 1M mixed DEL/DUP/INV breakpoints
    "c3_mixed_1m": dict(n_units=1_000_000, svtype_mix=(0.70, 0.15, 0.15, 0.0), config_no=3),
}


def make_config(name: str, libs: Sequence[LibraryTable], n_units: Optional[int] = None,
                chunk: int = 100_000, seed_offset: int = 0) -> EvidenceBatch:
    """A BASELINE.json configuration, generated in chunks (seeds BASE_SEED + config# ...)."""
    cfg = CONFIGS[name]
    n = int(n_units if n_units is not None else cfg["n_units"])
    parts: List[EvidenceBatch] = []
    done = 0
    i = 0
    while done < n:
        m = min(chunk, n - done)
        parts.append(make_units(m, BASE_SEED + cfg["config_no"] + 1000 * i + seed_offset, libs,
                                svtype_mix=cfg["svtype_mix"]))
        done += m
        i += 1
    return parts[0] if len(parts) == 1 else ev.concat_batches(parts)


def permute_units(batch: EvidenceBatch, order: np.ndarray) -> EvidenceBatch:
    """Units re-ordered as `order` (records move with their unit)."""
    order = np.asarray(order, dtype=np.int64)
    off = batch.rec_offset.astype(np.int64)
    F = (off[1:] - off[:-1])[order]
    new_off = np.zeros(len(order) + 1, np.int64)
    np.cumsum(F, out=new_off[1:])
    # source record index of every destination record
    src = np.repeat(off[:-1][order] - new_off[:-1], F) + np.arange(int(new_off[-1]))
    return EvidenceBatch(new_off.astype(np.uint64), batch.units[order], batch.records[src], batch.libs,
                         batch.split_weight, batch.disc_weight)


def _multisample_part(args):
    n_sites, seed, libs, lib_ids, s, af, frag = args
    return make_units(n_sites, seed + 17 * (s + 1), libs, svtype_mix=(0.70, 0.15, 0.15, 0.0), sample=s,
                      lib_choices=lib_ids, alt_af=af, **frag)


def make_multisample(n_sites: int, n_samples: int, seed: int, libs_per_sample=(1, 3), mean_frags: float = 100.0,
                     sd_frags: float = 25.0, min_frags: int = 18, max_frags: int = 183, pool_map=map, layout: str = "site"):
    """BASELINE.json configs[4] shape: (site, sample) units, site-major, every sample with its own 1..3
    libraries (rounded-normal insert-size histograms) and genotypes drawn per sample from a site allele
    frequency ~ Beta(0.5, 2).  `pool_map` may be a multiprocessing Pool.map to generate samples in parallel.
    layout: "site" = the site-major batch; "sample" = the same units sample-major (== to_sample_major(site-major batch)[0],
    which is how they are generated); "both" = (site-major, sample-major)."""
    rng = np.random.default_rng(seed)
    libs: List[LibraryTable] = []
    sample_libs = []
    for s in range(n_samples):
        k = int(rng.integers(libs_per_sample[0], libs_per_sample[1] + 1))
        ids = []
        for _ in range(k):
            ids.append(len(libs))
            libs.append(normal_library(float(rng.uniform(250, 550)), float(rng.uniform(40, 120)), n=200_000,
                                       seed=int(rng.integers(1 << 30)), name="s%d_l%d" % (s, len(ids))))
        sample_libs.append(ids)
    af = rng.beta(0.5, 2.0, n_sites)
    frag = dict(mean_frags=mean_frags, sd_frags=sd_frags, min_frags=min_frags, max_frags=max_frags)
    parts = list(pool_map(_multisample_part,
                          [(n_sites, seed, libs, sample_libs[s], s, af, frag) for s in range(n_samples)]))
    # the same site has the same svtype / length in every sample: copy sample 0's unit geometry
    for p in parts[1:]:
        for fld in ("svtype", "var_length", "pos_delta"):
            p.units[fld] = parts[0].units[fld]
    for s, p in enumerate(parts):      # every unit says which libraries its sample owns (svt_unit.libs)
        p.units["libs"] = ev.unit_libs(sample_libs[s][0], len(sample_libs[s]))
    allb = ev.concat_batches(parts)
    del parts
    allb.libs = libs
    if layout == "sample":
        return allb
    order = (np.arange(n_sites)[:, None] + n_sites * np.arange(n_samples)[None, :]).reshape(-1)
    out = permute_units(allb, order)
    out.libs = libs
    return (out, allb) if layout == "both" else out


def make_edge_cases(libs: Sequence[LibraryTable], seed: int = 1) -> EvidenceBatch:
    """Small adversarial batch: empty/skip units, ragged record counts 0..300, all svtypes,
    MAPQ values whose weights do not sum associatively (0.9, 0.99, ...), multi-record fragments
    (continuation records), duplications with very deep alt support (GT './.' by underflow)."""
    rng = np.random.default_rng(seed)
    parts = [
        make_units(700, seed + 1, libs, svtype_mix=(0.4, 0.2, 0.2, 0.2), mean_frags=60, sd_frags=60,
                   min_frags=0, max_frags=300, frac_empty=0.05, frac_skip=0.03),
    ]
    # low-MAPQ heavy units: sums of 0.9 / 0.99 / 0.5 that land next to integers
    b = make_units(300, seed + 2, libs, svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=20,
                   min_frags=1, max_frags=120)
    for fld in ("mapq_a", "mapq_b", "rs_a", "rs_b", "seq_l", "seq_r", "clip_l"):
        m = rng.random(b.n_records) < 0.7
        b.records[fld] = np.where(m & (b.records[fld] > 0), rng.choice([3, 10, 20, 30, 255], b.n_records),
                                  b.records[fld]).astype(np.uint8)
    parts.append(b)
    # continuation records (fragments with more than two primaries / candidates)
    c = make_units(200, seed + 3, libs, svtype_mix=(0.4, 0.2, 0.2, 0.2), mean_frags=30, sd_frags=10,
                   min_frags=2, max_frags=80)
    first = np.zeros(c.n_records, bool)
    first[c.rec_offset[:-1][c.rec_offset[:-1] < c.n_records].astype(np.int64)] = True
    cont = (~first) & (rng.random(c.n_records) < 0.15)
    fl = c.records["flags"].copy()
    # a continuation record carries no pair evidence
    fl[cont] &= ~np.uint32(ev.REC_ALT_STRADDLE | ev.REC_REF_STRADDLE_A | ev.REC_REF_STRADDLE_B | ev.REC_HAS_PAIR)
    fl[cont] |= np.uint32(ev.REC_CONTINUATION)
    c.records["flags"] = fl
    parts.append(c)
    # very deep duplications: QA > 679 with QR == 0 underflows sum(10**GL)
    d = make_units(64, seed + 4, libs, svtype_mix=(0, 1, 0, 0), mean_frags=900, sd_frags=150,
                   min_frags=600, max_frags=1300, alt_af=np.full(64, 1.0))
    fl = d.records["flags"]
    fl &= ~np.uint32(ev.REC_REF_STRADDLE_A | ev.REC_REF_STRADDLE_B)
    fl |= np.uint32(ev.REC_ALT_STRADDLE | ev.REC_HAS_PAIR)
    d.records["flags"] = fl
    d.records["rs_a"] = 0
    d.records["rs_b"] = 0
    d.records["mapq_a"] = 60
    d.records["mapq_b"] = 60
    parts.append(d)
    return ev.concat_batches(parts)


def to_sample_major(batch, n_samples: int):
    """The same (site, sample) units reordered from site-major (unit = site * n_samples + sample: what a joint
    caller walks, svtyper/classic.py:279) to sample-major (all sites of sample 0, then sample 1, ...: what a producer
    that reads BAM by BAM emits).  Returns (batch, order) with new unit k = old unit order[k].  Pair with
    hip.DeviceBatch.result_order(n_samples): the results then come back site-major."""
    n = batch.n_units
    if n % n_samples:
        raise ValueError("n_units must be a multiple of n_samples")
    n_sites = n // n_samples
    order = (np.arange(n_sites, dtype=np.int64)[None, :] * n_samples + np.arange(n_samples, dtype=np.int64)[:, None]).reshape(-1)
    off = batch.rec_offset.astype(np.int64)
    cnt = (off[1:] - off[:-1])[order]
    new_off = np.zeros(n + 1, np.uint64)
    new_off[1:] = np.cumsum(cnt)
    src = np.repeat(off[:-1][order] - new_off[:-1].astype(np.int64), cnt) + np.arange(int(new_off[-1]), dtype=np.int64)
    return ev.EvidenceBatch(new_off, batch.units[order], batch.records[src], batch.libs, batch.split_weight, batch.disc_weight), order


def replicate_sample_major(batch, n_samples: int, copies: int):
    """`copies` x the sites of a SAMPLE-MAJOR batch (unit = sample * n_sites + site), still sample-major: the sites of copy c
    follow those of copy c - 1 inside every sample's block.  One concatenation of contiguous slices -- the way to reach
    configs[4]'s full size (500 k sites x 32 samples = 16 M units, 1.6 G records, 26 GB) from a batch a host generates in
    seconds; the copies carry the same evidence (the pass does not care; results repeat with period n_sites)."""
    n = batch.n_units
    if n % n_samples:
        raise ValueError("n_units must be a multiple of n_samples")
    n_sites = n // n_samples
    off = batch.rec_offset.astype(np.int64)
    per_sample_recs = off[n_sites::n_sites] - off[:-1:n_sites][:n_samples]       # records of every sample's block
    total_recs = int(per_sample_recs.sum()) * copies
    units = np.empty(n * copies, batch.units.dtype)
    records = np.empty(total_recs, batch.records.dtype)
    counts = np.empty(n * copies, np.int64)
    u = r = 0
    cnt = off[1:] - off[:-1]
    for s in range(n_samples):
        lo, hi = s * n_sites, (s + 1) * n_sites
        r0, r1 = int(off[lo]), int(off[hi])
        for _ in range(copies):
            units[u:u + n_sites] = batch.units[lo:hi]
            counts[u:u + n_sites] = cnt[lo:hi]
            records[r:r + (r1 - r0)] = batch.records[r0:r1]
            u += n_sites
            r += r1 - r0
    new_off = np.zeros(n * copies + 1, np.uint64)
    np.cumsum(counts, out=new_off[1:])
    return ev.EvidenceBatch(new_off, units, records, batch.libs, batch.split_weight, batch.disc_weight)
