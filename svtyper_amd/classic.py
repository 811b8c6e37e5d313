"""`svtyper` (multi-sample) driver with the reference's call surface.

    sv_genotype(bam_string, vcf_in, vcf_out, min_aligned, split_weight, disc_weight, num_samp,
                lib_info_path, debug, alignment_outpath, ref_fasta, sum_quals, max_reads, max_ci_dist)

Same arguments, defaults and output bytes as svtyper/classic.py:107-533, but the per-sample
likelihood block (classic.py:286-513) does not run here: evidence is packed on the host and
genotyped in device batches (pipeline.py).  `engine` is an extra, keyword-only seam; it
defaults to the HIP library and there is no CPU implementation in this package.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
from typing import List

from . import __version__
from . import evidence as ev
from .bam import open_alignment_file
from .library import Sample, setup_sample, write_sample_json
from .pipeline import (MIN_LIB_PREVALENCE, BulkFeeder, ChunkPipeline, NativeUnitCollector, SampleColumnWriter, UnitCollector, add_read_to,
                       default_engine, fetch_window, resolve_reader, text_blocks)
from .results import results_to_dicts
from .vcf import VALID_SVTYPES, Variant, Vcf

CHUNK_UNITS = 50_000    # (breakpoint, sample) units per device batch: small enough to overlap chunks (ChunkPipeline)


def gather_all_reads(sample: Sample, bp: dict, max_reads):
    """Fragments of both breakends, or ({}, True) when a side has more than max_reads reads
    (classic.py:54-100: the counter runs over every fetched record of that side)."""
    fragments = {}
    for side in ("A", "B"):
        chrom, lo, hi = fetch_window(sample, bp[side]["chrom"], bp[side]["pos"], bp[side]["ci"], as_int=False)
        for i, read in enumerate(sample.bam.fetch(chrom, lo, hi)):
            if read.is_unmapped or read.is_duplicate:
                continue
            lib = sample.get_lib(read.get_tag("RG"))
            if lib.name not in sample.active_libs:
                continue
            if max_reads is not None and i > max_reads:
                return {}, True
            add_read_to(fragments, read, lib)
    return fragments, False


_FORMAT_KEYS = ("GL", "DP", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP", "AB", "GQ", "SQ", "GT")


def apply_result(var: Variant, sample_name: str, gt: int, res: dict) -> None:
    """Result of one unit -> FORMAT fields and QUAL of one sample (classic.py:454-513)."""
    g = var.genotype(sample_name)
    if gt == ev.GT_SKIPPED:                       # classic.py:282-284
        g.set_format("GT", "./.")
        return
    f = res["formats"]
    if gt == ev.GT_BLANK:                         # classic.py:496-513 (QUAL is reset, not kept)
        var.qual = 0
    g.set_formats([(key, f[key]) for key in _FORMAT_KEYS])
    if gt >= 0:
        var.qual += res["qual"]                   # classic.py:485


def sv_genotype(bam_string, vcf_in, vcf_out, min_aligned, split_weight, disc_weight, num_samp, lib_info_path,
                debug, alignment_outpath, ref_fasta, sum_quals, max_reads, max_ci_dist, *, engine=None, geometry="host",
                reader=None, stats=None):
    if alignment_outpath is not None:
        raise NotImplementedError("-w/--write_alignment (evidence BAM dump) is outside the MI355X hot path build")
    reader = resolve_reader(reader)
    bams = []
    for path in bam_string.split(","):
        if not (path.endswith(".bam") or path.endswith(".cram")):
            sys.stderr.write("Error: %s is not a valid alignment file (*.bam or *.cram)\n" % path)
            sys.exit(1)
        bams.append(open_alignment_file(path, ref_fasta))

    lib_info = None
    if lib_info_path is not None and os.path.isfile(lib_info_path):
        with open(lib_info_path) as f:
            lib_info = json.load(f)
    if vcf_in is None:
        sys.stderr.write("Warning: VCF not found.\n")
    native = None
    if reader == "native":      # C++ reader: library scans now, fetch + fragment summaries later
        from .native_reads import COUNT_CLASSIC, NativeBam
        native = [NativeBam(p) for p in bam_string.split(",")]
    samples: List[Sample] = [setup_sample(b, lib_info, num_samp, MIN_LIB_PREVALENCE, nb)
                             for b, nb in zip(bams, native or [None] * len(bams))]
    if lib_info_path is not None and not os.path.isfile(lib_info_path):
        logging.info("Writing library metrics to %s..." % lib_info_path)
        write_sample_json(samples, open(lib_info_path, "w"))
    if vcf_in is None:
        return

    if engine is None:
        engine = default_engine()
    vcf = Vcf()
    if reader == "native":      # C++ fetch + summariser; geometry in the reader's threads ("host") or on the device
        collector = NativeUnitCollector(samples, native, split_weight, disc_weight, min_aligned, COUNT_CLASSIC,
                                        max_reads, geometry="device" if geometry == "device" else "reader")
    elif reader == "python":
        collector = UnitCollector(samples, split_weight, disc_weight, min_aligned, geometry)
    else:
        raise ValueError("reader must be 'python' or 'native'")
    pending: list = []      # ordered output actions of the current chunk
    header_lines: list = []
    n_samp = len(samples)
    pipe = ChunkPipeline()
    fast: list = []     # SampleColumnWriter, made once the header is known

    def flush():
        actions = list(pending)
        pending.clear()
        quals = [float(a[1].qual) for a in actions if a[0] == "gt"]      # incoming QUAL (0 unless --sum_quals)
        pipe.submit(collector.take(engine, 0, site_quals=quals), lambda results: write_out(results, actions))

    def render_actions(results, actions):
        """the output text of every action, one string each (the lines of a variant, of a BND pair, of a line passed through)"""
        gts = results.gt.tolist()
        site_qual = None if results.site_qual is None else results.site_qual.tolist()
        columns = sqs = dicts = None
        for action in actions:
            if action[0] == "raw":
                yield action[1].get_var_string() + "\n"
                continue
            _, var, var2, first_unit = action
            unit_gts = gts[first_unit:first_unit + n_samp]
            if (not debug and fast and fast[0].eligible(var)
                    and any(g != ev.GT_SKIPPED for g in unit_gts)):
                # bulk path: the sample columns of the whole chunk were formatted in one native call
                if columns is None:
                    columns = fast[0].columns(results)
                    sqs = results.sq.tolist()
                if site_qual is not None:
                    var.qual = site_qual[first_unit // n_samp]
                else:
                    for k, g in enumerate(unit_gts):           # classic.py:485,498
                        if g >= 0:
                            var.qual += sqs[first_unit + k]
                        elif g == ev.GT_BLANK:
                            var.qual = 0
                cols = columns[first_unit:first_unit + n_samp]
                text = var.get_var_string_with(fast[0].format_string, cols) + "\n"
                if var2 is not None:               # BND: second mate carries the same QUAL and genotypes
                    var2.qual = var.qual
                    text += var2.get_var_string_with(fast[0].format_string, cols) + "\n"
                yield text
                continue
            if dicts is None:
                dicts = results_to_dicts(results)
            for k, sample in enumerate(samples):
                if debug:
                    _debug_print(results.rec[first_unit + k])
                apply_result(var, sample.name, gts[first_unit + k], dicts[first_unit + k])
            if site_qual is not None:      # the same running sum, over the refined SQ (hip.site_qual_host)
                var.qual = site_qual[first_unit // n_samp]
            text = var.get_var_string() + "\n"
            if var2 is not None:                   # BND: second mate carries the same genotypes
                var.share_genotypes_with(var2)
                text += var2.get_var_string() + "\n"
            yield text

    def write_out(results, actions):
        for text in render_actions(results, actions):
            vcf_out.write(text)

    def start_body():
        """the first variant line ends the header (classic.py:166-176)"""
        vcf.add_header(header_lines)
        vcf.add_custom_svtyper_headers()
        for sample in samples:
            if sample.name not in vcf.sample_list:
                vcf.add_sample(sample.name)
        vcf_out.write(vcf.get_header() + "\n")
        fast.append(SampleColumnWriter(vcf, [s.name for s in samples], skipped_as_dots=True))

    def handle_line(line, first_unit_base=0):
        """One variant line -> its output action (classic.py:219-278), or None for a first BND mate (it waits for its
        partner, classic.py:256-258); its units go to the collector."""
        var = Variant(line.rstrip().split("\t"), vcf)
        if not sum_quals:
            var.qual = 0
        if not var.has_svtype():
            sys.stderr.write("Warning: SVTYPE missing at variant %s. Skipping.\n" % var.var_id)
            return ("raw", var)
        if var.get_svtype() not in VALID_SVTYPES:
            sys.stderr.write("Warning: Unsupported SVTYPE at variant %s (%s). Skipping.\n"
                             % (var.var_id, var.get_svtype()))
            return ("raw", var)
        bp = vcf.get_variant_breakpoints(var, max_ci_dist)
        if bp is None:
            return None
        var2 = None
        if var.get_svtype() == "BND":
            var2 = var
            var = _take_first_mate(vcf, bp, var2)
        if reader == "native":
            first_unit = collector.add_site(bp)
        else:
            first_unit = len(collector)
            for k, sample in enumerate(samples):
                fragments, many = gather_all_reads(sample, bp, max_reads)
                collector.add(bp, k, fragments, skip=many)
        return ("gt", var, var2, first_unit - first_unit_base)

    def per_line(lines):
        """the general route: one Variant object per line, device batches of CHUNK_UNITS units"""
        for line in lines:
            action = handle_line(line)
            if action is not None:
                pending.append(action)
            if len(collector) >= CHUNK_UNITS:
                flush()

    # bulk route (reader="native"): blocks of lines -> breakpoint arrays -> output text in native calls (bulk_vcf.py); lines
    # it hands back, and everything once it stops in front of a BND line it cannot express, take the per-line route above
    bulk = None
    bulk_stats = None
    unpaired = False        # first BND mates left in the bulk parser at the end
    if (reader == "native" and not debug and hasattr(vcf_in, "readline") and hasattr(vcf_in, "read")
            and os.environ.get("SVT_BULK_VCF", "1") != "0"):
        from . import bulk_vcf
        if bulk_vcf.available():
            bulk = bulk_vcf
    if bulk is None:
        in_header = True
        for line in vcf_in:
            if in_header:
                if line[0] == "#":
                    header_lines.append(line)
                    continue
                in_header = False
                start_body()
            per_line((line,))
    else:
        first = vcf_in.readline()
        while first and first[0] == "#":
            header_lines.append(first)
            first = vcf_in.readline()
        if first:
            start_body()
            if not fast[0].enabled:       # other samples' columns in the VCF: every line keeps its Genotype objects
                per_line((first,))
                per_line(vcf_in)
            else:
                feeder = BulkFeeder(bulk, vcf, collector, pipe, engine, 0, n_samp, fast[0], bulk.QUAL_CLASSIC, max_ci_dist,
                                    sum_quals, False, handle_line, render_actions, vcf_out.write)
                rest = feeder.run(text_blocks(first, vcf_in, n_samp))
                if rest is not None:      # the per-line route from here on, with the BND mates the parser was holding
                    for held in feeder.pending_lines():
                        mate = Variant(held.split("\t"), vcf)
                        if not sum_quals:
                            mate.qual = 0
                        vcf._bnd_pending[mate.var_id] = mate
                    per_line(rest)
                    per_line(vcf_in)
                else:
                    unpaired = feeder.n_pending() > 0
                bulk_stats = (feeder.laps, "bulk" if rest is None else "bulk, then per line")

    flush()
    pipe.close()
    if stats is not None:       # (keyword-only extra: where the caller's thread spent its time, pipeline.BulkFeeder.laps)
        stats.update(bulk_stats[0] if bulk_stats else {}, route=bulk_stats[1] if bulk_stats else "per line")
    if vcf._bnd_pending or unpaired:
        logging.warning("Unpaired breakends found in file. These will not be present in output.")
    vcf_in.close()
    vcf_out.close()


# the first mate of a BND pair is kept by the Vcf until its partner shows up
def _take_first_mate(vcf: Vcf, bp: dict, second: Variant) -> Variant:
    return vcf._bnd_first.pop(bp["id"])


def _debug_print(rec):
    t = dict(zip(ev.TALLY_NAMES, (float(x) for x in rec["tallies"])))
    print("--------------------------")
    for key in ("ref_span", "alt_span", "ref_seq", "alt_seq", "alt_clip"):
        print("%s: %s" % (key, t[key]))
    if int(rec["gt"]) not in (ev.GT_BLANK, ev.GT_SKIPPED):
        print([float(x) for x in rec["gl"]])


# ------------------------------------------------------------------------------------------ CLI
def get_args():
    p = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description=(
        "svtyper (MI355X-native likelihood path)\nversion: %s\n"
        "description: Compute genotype of structural variants based on breakpoint depth" % __version__))
    p.add_argument("-i", "--input_vcf", metavar="FILE", type=argparse.FileType("r"), default=None,
                   help="VCF input (default: stdin)")
    p.add_argument("-o", "--output_vcf", metavar="FILE", type=argparse.FileType("w"), default=sys.stdout,
                   help="output VCF to write (default: stdout)")
    p.add_argument("-B", "--bam", metavar="FILE", type=str, required=True,
                   help="BAM or CRAM file(s), comma-separated if genotyping multiple samples")
    p.add_argument("-T", "--ref_fasta", metavar="FILE", type=str, default=None,
                   help="Indexed reference FASTA file (recommended for reading CRAM files)")
    p.add_argument("-S", "--split_bam", type=str, help=argparse.SUPPRESS)
    p.add_argument("-l", "--lib_info", metavar="FILE", dest="lib_info_path", type=str, default=None,
                   help="create/read JSON file of library information")
    p.add_argument("-m", "--min_aligned", metavar="INT", type=int, default=20,
                   help="minimum number of aligned bases to consider read as evidence [20]")
    p.add_argument("-n", dest="num_samp", metavar="INT", type=int, default=1000000,
                   help="number of reads to sample from BAM file for building insert size distribution [1000000]")
    p.add_argument("-q", "--sum_quals", action="store_true",
                   help="add genotyping quality to existing QUAL (default: overwrite QUAL field)")
    p.add_argument("--max_reads", metavar="INT", type=int, default=None,
                   help="maximum number of reads to assess at any variant (default: unlimited)")
    p.add_argument("--max_ci_dist", metavar="INT", type=int, default=1e10,
                   help="maximum size of a confidence interval before 95%% CI is used intead (default: 1e10)")
    p.add_argument("--split_weight", metavar="FLOAT", type=float, default=1, help="weight for split reads [1]")
    p.add_argument("--disc_weight", metavar="FLOAT", type=float, default=1,
                   help="weight for discordant paired-end reads [1]")
    p.add_argument("-w", "--write_alignment", metavar="FILE", dest="alignment_outpath", type=str, default=None,
                   help="write relevant reads to BAM file")
    p.add_argument("--debug", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--verbose", action="store_true", default=False, help="Report status updates")
    # not in the reference: where the host work runs (same output bytes either way)
    p.add_argument("--reader", choices=("python", "native"), default="native",
                   help="BAM access + fragment assembly: the C++ threads of libsvtyper_hip.so feeding the device "
                        "geometry stage, or the portable Python reader (same output bytes) [native]")
    p.add_argument("--geometry", choices=("host", "device"), default="host",
                   help="with --reader python: breakpoint-dependent read predicates on the host or on the GPU [host]")
    args = p.parse_args()
    if args.input_vcf is None and not sys.stdin.isatty():
        args.input_vcf = sys.stdin
    return args


def main():
    args = get_args()
    logging.basicConfig(format="%(message)s", level=logging.INFO if args.verbose else logging.WARNING)
    if args.split_bam is not None:
        sys.stderr.write("Warning: --split_bam (-S) is deprecated. Ignoring %s.\n" % args.split_bam)
    call = (args.bam, args.input_vcf, args.output_vcf, args.min_aligned, args.split_weight, args.disc_weight,
            args.num_samp, args.lib_info_path, args.debug, args.alignment_outpath, args.ref_fasta,
            args.sum_quals, args.max_reads, args.max_ci_dist)
    from . import sharded
    job = sharded.job()
    if job is None:
        return sv_genotype(*call, geometry=args.geometry, reader=args.reader)
    # launched by torch.distributed.run with several ranks: one GPU each, variants sharded, one gather
    rank, world, local_rank = job
    call = call[:2] + (sharded.private_stdout(call[2]),) + call[3:]
    engine = sharded.init(local_rank)
    sharded.sv_genotype_sharded(*call, rank=rank, world=world, engine=engine, geometry=args.geometry,
                                reader=args.reader)
    sharded.finish()


def cli():
    try:
        sys.exit(main())
    except IOError as e:
        if e.errno != 32:   # EPIPE
            raise


if __name__ == "__main__":
    cli()
