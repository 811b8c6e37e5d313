"""`svtyper-sso` (single sample) driver with the reference's call surface.

    sso_genotype(bam_string, vcf_in, vcf_out, min_aligned, split_weight, disc_weight, num_samp,
                 lib_info_path, debug, ref_fasta, sum_quals, max_reads, max_ci_dist, cores, batch_size)

Same arguments, defaults and output bytes as svtyper/singlesample.py:764-814.  The reference's
`--cores N` fans batches of breakpoints out to a multiprocessing.Pool (singlesample.py:710-762);
here every breakpoint of a chunk goes to the GPU in one batch instead, so `cores` only selects the
reference's two-pass bookkeeping and `batch_size` is accepted for compatibility.  The kernel runs
with the singlesample floating-point association (SVT_FLAG_SSO_ASSOCIATION).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

from . import __version__
from . import evidence as ev
from .bam import open_alignment_file
from .library import Sample, setup_sample, write_sample_json
from .pipeline import (MIN_LIB_PREVALENCE, BulkFeeder, block_chars, ChunkPipeline, NativeUnitCollector, split_lines, SampleColumnWriter, UnitCollector, add_read_to,
                       default_engine, fetch_window, resolve_reader)
from .results import results_to_dicts
from .vcf import Variant, Vcf

CHUNK_UNITS = 50_000    # (breakpoint, sample) units per device batch: small enough to overlap chunks (ChunkPipeline)
_ASSIGN_ORDER = ("GT", "GQ", "SQ", "GL", "DP", "AO", "RO", "AS", "ASC", "RS", "AP", "RP", "QR", "QA", "AB")


def logit(msg):
    import datetime
    import time
    ts = time.strftime("[ %Y-%m-%d %T ]", datetime.datetime.now().timetuple())
    print("%s %s" % (ts, msg), file=sys.stderr)
    sys.stderr.flush()


def gather_reads(sample: Sample, bp: dict, max_reads):
    """Fragments of both breakends, or ({}, True) when either region holds more than max_reads
    countable reads (singlesample.py:158-205: bam.count() of both regions first, then fetch)."""
    regions = [fetch_window(sample, bp[s]["chrom"], bp[s]["pos"], bp[s]["ci"], as_int=True) for s in ("A", "B")]
    if max_reads is not None:
        counts = [sample.bam.count(c, lo, hi, read_callback="all") for c, lo, hi in regions]
        if counts[0] > max_reads or counts[1] > max_reads:
            logit("SKIPPING -- Variant '%s' has a region with too many reads (> %s)" % (bp["id"], max_reads))
            return {}, True
    fragments = {}
    for chrom, lo, hi in regions:
        for read in sample.bam.fetch(chrom, lo, hi):
            if read.is_unmapped or read.is_duplicate:
                continue
            lib = sample.rg_to_lib[read.get_tag("RG")]
            if lib.name not in sample.active_libs:
                continue
            add_read_to(fragments, read, lib)
    return fragments, False


# ------------------------------------------------------------------------------------------
# the reference's inner operator seams, by name: same arguments, same dict shapes,
# the arithmetic on the MI355X through the C ABI (no CPU implementation lives here)
# ------------------------------------------------------------------------------------------
SPLIT_SLOP = 3   # singlesample.py:792 / classic.py:184
_TALLY_KEYS = ("ref_seq", "alt_seq", "alt_clip", "ref_span", "alt_span")   # SVT_TAL_* order


def blank_genotype_result():
    """svtyper/singlesample.py:207-227"""
    from .results import blank_result
    return blank_result()


def _library_table(lib):
    if hasattr(lib, "table"):
        return lib.table()
    return ev.LibraryTable.from_counter(dict(lib.hist), float(lib.mean), float(lib.sd), getattr(lib, "name", "lib"))


def tally_variant_read_fragments(split_slop, min_aligned, breakpoint, sam_fragments, debug, *, device=0):
    """svtyper/singlesample.py:355-404: the five evidence tallies of one breakpoint over its read-fragments
    (sorted by query name, fragment-local split-read sums, zeroing rules applied) as the reference's `counts`
    dict.  The fragments' yes/no geometry is asked here (packer.py), every weight, the insert-size test, the
    sums and the zeroing rules are evaluated by the streaming kernel (svt_genotype, SVT_FLAG_SSO_ASSOCIATION)."""
    from . import hip
    from .packer import BatchBuilder, pack_fragments, unit_header
    libs, lib_index = [], {}
    for name in sorted(sam_fragments.keys()):
        lib = sam_fragments[name].lib
        if id(lib) not in lib_index:
            lib_index[id(lib)] = len(libs)
            libs.append(_library_table(lib))
    if not libs:     # no fragments: the reference's loop body never runs and the initial integer zeros come back
        counts = {k: 0 for k in _TALLY_KEYS}
    else:
        builder = BatchBuilder(libs, 1.0, 1.0)
        builder.add(unit_header(breakpoint), pack_fragments(sam_fragments, breakpoint, lib_index, min_aligned, split_slop))
        res = hip.genotype_batch(builder.build(), device=device, flags=ev.FLAG_SSO_ASSOCIATION)
        counts = {k: float(res.tallies[0, i]) for i, k in enumerate(_TALLY_KEYS)}
    if debug:
        items = ("ref_span", "alt_span", "ref_seq", "alt_seq", "alt_clip")
        logit("{} -- read fragment tally counts:\n{}".format(
            breakpoint["id"], "\n".join("{}: {}".format(i, counts[i]) for i in items)))
    return counts


def bayesian_genotype(breakpoint, counts, split_weight, disc_weight, debug, *, device=0):
    """svtyper/singlesample.py:406-473: `counts` (as tally_variant_read_fragments returned them) -> the
    reference's result dict {'qual', 'formats': {GT, GQ, SQ, GL, DP, AO, RO, AS, ASC, RS, AP, RP, QR, QA, AB}}.
    QR/QA, bayes_gt, the GT/GQ decision and the counts come from svt_genotype_counts (device); SQ is then taken
    from the bit-exact GL with the host libm, as the reference does (svt_results_host_sq)."""
    from . import hip
    from .results import result_from_record
    res = hip.host_sq(hip.genotype_counts([[counts[k] for k in _TALLY_KEYS]], [breakpoint["svtype"] == "DUP"],
                                          split_weight, disc_weight, device))
    if debug:
        logit("{} -- log probabilities (homref, het, homalt) : {}".format(breakpoint["id"], [float(x) for x in res.gl[0]]))
    return result_from_record(res.rec[0])


def assign_genotype(variant: Variant, sample_name: str, res: dict) -> None:
    """singlesample.py:544-575: every FORMAT field is always written; QUAL accumulates."""
    variant.qual += res["qual"]
    f = res["formats"]
    variant.genotype(sample_name).set_formats([(key, f[key]) for key in _ASSIGN_ORDER])


def sso_genotype(bam_string, vcf_in, vcf_out, min_aligned, split_weight, disc_weight, num_samp, lib_info_path,
                 debug, ref_fasta, sum_quals, max_reads, max_ci_dist, cores, batch_size, *, engine=None, geometry="host",
                 reader=None, stats=None):
    if vcf_in is None:
        return
    reader = resolve_reader(reader)
    full_bam_path = os.path.abspath(bam_string)
    if not (full_bam_path.endswith(".bam") or full_bam_path.endswith(".cram")):
        sys.exit("Error: %s is not a valid alignment file (*.bam or *.cram)\n" % full_bam_path)
    bam = open_alignment_file(full_bam_path, ref_fasta)

    lib_info = None
    if lib_info_path is not None and os.path.exists(lib_info_path):
        logit("Reading library metrics from %s..." % lib_info_path)
        with open(lib_info_path) as f:
            lib_info = json.load(f)
    native = None
    if reader == "native":      # C++ reader: library scans now, fetch + fragment summaries later
        from .native_reads import COUNT_SSO, NativeBam
        native = NativeBam(full_bam_path)
    sample = setup_sample(bam, lib_info, num_samp, MIN_LIB_PREVALENCE, native)
    if lib_info_path is not None and not os.path.exists(lib_info_path):
        logit("Writing library metrics to %s..." % lib_info_path)
        write_sample_json([sample], open(lib_info_path, "w"))

    if engine is None:
        engine = default_engine()

    # bulk route (reader="native"): the body as blocks of text -> breakpoint arrays -> output text in native calls
    # (bulk_vcf.py); the per-line route below stays the general one (and takes over at a BND line the parser cannot express)
    bulk = None
    if reader == "native" and hasattr(vcf_in, "read") and os.environ.get("SVT_BULK_VCF", "1") != "0":
        from . import bulk_vcf
        if bulk_vcf.available():
            bulk = bulk_vcf

    # header: only the '##' lines are parsed, so sample columns of the input are not carried over and
    # the BAM's sample becomes the only column (singlesample.py:112-125)
    header = []
    input_samples = []
    if bulk is None:
        lines = vcf_in.readlines()
        for line in lines:
            if line.startswith("##"):
                header.append(line)
            else:
                break
        for line in lines:
            if line.startswith("#CHROM"):
                input_samples = line.rstrip().split("\t")[9:]
                break
    else:
        text = vcf_in.read()
        body_at = 0
        while text.startswith("##", body_at):
            nl = text.find("\n", body_at)
            end = len(text) if nl < 0 else nl + 1
            header.append(text[body_at:end])
            body_at = end
        at = 0 if text.startswith("#CHROM") else text.find("\n#CHROM") + 1
        if at > 0 or text.startswith("#CHROM"):
            nl = text.find("\n", at)
            input_samples = text[at:len(text) if nl < 0 else nl].rstrip().split("\t")[9:]
    vcf = Vcf()
    vcf.filename = getattr(vcf_in, "name", "<stdin>")
    vcf.add_header(header)
    vcf.add_custom_svtyper_headers()
    if sample.name not in input_samples:
        logit("Note: Did not find sample name : '%s' in input vcf: '%s' -- adding" % (sample.name, vcf.filename))
    vcf.add_sample(sample.name)
    vcf.write_header(vcf_out)

    logit("Genotyping Input VCF (%s Mode)" % ("Serial" if cores is None else "Parallel"))
    if reader == "native":      # C++ fetch + summariser (cores = its thread count); geometry in the reader's threads ("host") or on the device
        collector = NativeUnitCollector([sample], [native], split_weight, disc_weight, min_aligned,
                                        COUNT_SSO, max_reads, n_threads=cores or 0,
                                        geometry="device" if geometry == "device" else "reader")
    elif reader == "python":
        collector = UnitCollector([sample], split_weight, disc_weight, min_aligned, geometry)
    else:
        raise ValueError("reader must be 'python' or 'native'")
    pending: list = []
    pipe = ChunkPipeline()

    def flush():
        actions = list(pending)
        pending.clear()
        pipe.submit(collector.take(engine, ev.FLAG_SSO_ASSOCIATION), lambda results: write_out(results, actions))

    fast = SampleColumnWriter(vcf, [sample.name], skipped_as_dots=False)

    def render_actions(results, actions):
        """the output text of every action, one string each"""
        columns = gts = sqs = dicts = None
        for action in actions:
            if action[0] == "raw":
                yield action[1].get_var_string() + "\n"
                continue
            _, variant, variant2, unit = action
            if fast.eligible(variant):
                # bulk path: the sample column of every unit of the chunk was formatted in one native call
                if columns is None:
                    columns, gts, sqs = fast.columns(results), results.gt.tolist(), results.sq.tolist()
                if gts[unit] >= 0:
                    variant.qual += sqs[unit]          # singlesample.py:544-546
                cols = columns[unit:unit + 1]
                out = variant.get_var_string_with(fast.format_string, cols) + "\n"
                if variant2 is not None:
                    variant2.qual = variant.qual
                    out += variant2.get_var_string_with(fast.format_string, cols) + "\n"
                yield out
                continue
            if dicts is None:
                dicts = results_to_dicts(results)   # blank for "no evidence" and "too many reads" alike
            assign_genotype(variant, sample.name, dicts[unit])
            out = variant.get_var_string() + "\n"
            if variant2 is not None:
                variant.share_genotypes_with(variant2)
                out += variant2.get_var_string() + "\n"
            yield out

    def write_out(results, actions):
        for out in render_actions(results, actions):
            vcf_out.write(out)

    def handle_line(line, unit_base=0):
        """One variant line -> its output action (singlesample.py:577-652), None for a first BND mate"""
        variant = Variant(line.rstrip().split("\t"), vcf)
        if not sum_quals:
            variant.qual = 0
        if not variant.has_svtype():
            logit("Warning: SVTYPE missing at variant %s. Skipping.\n" % variant.var_id)
            return ("raw", variant)
        if not variant.is_valid_svtype():
            logit("Warning: Unsupported SVTYPE at variant %s (%s). Skipping.\n" % (variant.var_id, variant.get_svtype()))
            return ("raw", variant)
        bp = vcf.get_variant_breakpoints(variant, max_ci_dist)
        if bp is None:
            return None
        variant2 = None
        if variant.get_svtype() == "BND":
            variant2 = variant
            variant = vcf._bnd_first.pop(bp["id"])
        if reader == "native":
            unit = collector.add_site(bp)
        else:
            fragments, many = gather_reads(sample, bp, max_reads)
            unit = collector.add(bp, 0, fragments, skip=many)
        return ("gt", variant, variant2, unit - unit_base)

    def per_line(lines):
        for line in lines:
            if line.startswith("#"):
                continue
            action = handle_line(line)
            if action is not None:
                pending.append(action)
            if len(collector) >= CHUNK_UNITS:
                flush()

    bulk_stats = None
    if bulk is None:
        per_line(lines)
    else:
        def blocks():
            nl = text.find("\n", body_at)
            chars = block_chars((nl if nl >= 0 else len(text)) - body_at)
            at = body_at
            while at < len(text):
                cut = text.find("\n", at + chars)
                end = len(text) if cut < 0 else cut + 1
                yield text[at:end]
                at = end
        feeder = BulkFeeder(bulk, vcf, collector, pipe, engine, ev.FLAG_SSO_ASSOCIATION, 1, fast, bulk.QUAL_SSO, max_ci_dist,
                            sum_quals, True, handle_line, render_actions, vcf_out.write)
        source = blocks()
        rest = feeder.run(source)
        if rest is not None:          # the per-line route from here on, with the BND mates the parser was holding
            for held in feeder.pending_lines():
                mate = Variant(held.split("\t"), vcf)
                if not sum_quals:
                    mate.qual = 0
                vcf._bnd_pending[mate.var_id] = mate
            per_line(rest)
            for block in source:
                per_line(split_lines(block))
        bulk_stats = (feeder.laps, "bulk" if rest is None else "bulk, then per line")
    flush()
    pipe.close()
    if stats is not None:       # (keyword-only extra: where the caller's thread spent its time, pipeline.BulkFeeder.laps)
        stats.update(bulk_stats[0] if bulk_stats else {}, route=bulk_stats[1] if bulk_stats else "per line")
    sample.close()


# ------------------------------------------------------------------------------------------ CLI
def get_args():
    p = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description=(
        "svtyper-sso (MI355X-native likelihood path)\nversion: %s\n"
        "description: Compute genotype of structural variants based on breakpoint depth on a SINGLE sample"
        % __version__))
    p.add_argument("-i", "--input_vcf", metavar="FILE", type=argparse.FileType("r"), default=None,
                   help="VCF input (default: stdin)")
    p.add_argument("-o", "--output_vcf", metavar="FILE", type=argparse.FileType("w"), default=sys.stdout,
                   help="output VCF to write (default: stdout)")
    p.add_argument("-B", "--bam", metavar="FILE", type=str, required=True, help="BAM or CRAM file")
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
    p.add_argument("--max_reads", metavar="INT", type=int, default=1000,
                   help="maximum number of reads to assess at any variant (default: 1000)")
    p.add_argument("--max_ci_dist", metavar="INT", type=int, default=1e10,
                   help="maximum size of a confidence interval before 95%% CI is used intead (default: 1e10)")
    p.add_argument("--split_weight", metavar="FLOAT", type=float, default=1, help="weight for split reads [1]")
    p.add_argument("--disc_weight", metavar="FLOAT", type=float, default=1,
                   help="weight for discordant paired-end reads [1]")
    p.add_argument("--debug", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--cores", type=int, metavar="INT", default=None,
                   help="accepted for compatibility: breakpoints are batched onto the GPU instead of a worker pool")
    p.add_argument("--batch_size", type=int, metavar="INT", default=1000,
                   help="accepted for compatibility with the reference's worker batches")
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
    if args.split_bam is not None:
        sys.stderr.write("Warning: --split_bam (-S) is deprecated. Ignoring %s.\n" % args.split_bam)
    call = (args.bam, args.input_vcf, args.output_vcf, args.min_aligned, args.split_weight, args.disc_weight,
            args.num_samp, args.lib_info_path, args.debug, args.ref_fasta, args.sum_quals, args.max_reads,
            args.max_ci_dist, args.cores, args.batch_size)
    from . import sharded
    job = sharded.job()
    if job is None:
        return sso_genotype(*call, geometry=args.geometry, reader=args.reader)
    # launched by torch.distributed.run with several ranks: one GPU each, variants sharded, one gather
    rank, world, local_rank = job
    call = call[:2] + (sharded.private_stdout(call[2]),) + call[3:]
    engine = sharded.init(local_rank)
    sharded.sso_genotype_sharded(*call, rank=rank, world=world, engine=engine, geometry=args.geometry,
                                 reader=args.reader)
    sharded.finish()


def cli():
    try:
        sys.exit(main())
    except IOError as e:
        if e.errno != 32:
            raise


if __name__ == "__main__":
    cli()
