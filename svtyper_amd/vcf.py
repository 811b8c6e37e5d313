"""VCF model: header bookkeeping, record parsing / rendering, breakpoint extraction.

Written for this package after the behaviour of svtyper/parsers.py:11-399 (Vcf, Variant,
Genotype, confidence_interval): the output of the reference is byte-compared against
tests/data/example.gt.vcf, so the rendering rules are reproduced exactly --

  * the header is re-synthesised: fileformat, today's fileDate, reference, INFO / ALT / FORMAT
    lines in first-seen order, then unrecognised header lines, then the column line;
  * INFO is rendered in header order and keys the header does not declare are dropped;
  * FORMAT keys follow the header's declaration order; floats print as %0.2f, QUAL as %0.2f.
"""
from __future__ import annotations

import re
import sys
import time
from typing import Dict, List, Optional

# the FORMAT / INFO declarations SVTyper adds to every output (parsers.py:32-47); these are part of
# the output format
SVTYPER_INFO = (("SVTYPE", 1, "String", "Type of structural variant"),)
SVTYPER_FORMATS = (
    ("GQ", 1, "Integer", "Genotype quality"),
    ("SQ", 1, "Float", "Phred-scaled probability that this site is variant (non-reference in this sample"),
    ("GL", "G", "Float", "Genotype Likelihood, log10-scaled likelihoods of the data given the called genotype "
                         "for each possible genotype generated from the reference and alternate alleles given "
                         "the sample ploidy"),
    ("DP", 1, "Integer", "Read depth"),
    ("RO", 1, "Integer", "Reference allele observation count, with partial observations recorded fractionally"),
    ("AO", "A", "Integer", "Alternate allele observations, with partial observations recorded fractionally"),
    ("QR", 1, "Integer", "Sum of quality of reference observations"),
    ("QA", "A", "Integer", "Sum of quality of alternate observations"),
    ("RS", 1, "Integer", "Reference allele split-read observation count, with partial observations recorded "
                         "fractionally"),
    ("AS", "A", "Integer", "Alternate allele split-read observation count, with partial observations recorded "
                           "fractionally"),
    ("ASC", "A", "Integer", "Alternate allele clipped-read observation count, with partial observations "
                            "recorded fractionally"),
    ("RP", 1, "Integer", "Reference allele paired-end observation count, with partial observations recorded "
                         "fractionally"),
    ("AP", "A", "Integer", "Alternate allele paired-end observation count, with partial observations recorded "
                           "fractionally"),
    ("AB", "A", "Float", "Allele balance, fraction of observations from alternate allele, QA/(QR+QA)"),
)

_FIELD_RE = re.compile(r'(?:[^,"]|"[^"]*")+')
VALID_SVTYPES = ("BND", "DEL", "DUP", "INV")


def _unquote(s: str) -> str:
    s = str(s)
    return s[1:-1] if s.startswith('"') and s.endswith('"') else s


class HeaderLine:
    """One structured ##INFO / ##FORMAT / ##ALT declaration."""

    def __init__(self, kind: str, id, number=None, type=None, desc=""):
        self.kind = kind
        self.id = str(id)
        self.number = None if number is None else str(number)
        self.type = None if type is None else str(type)
        self.desc = _unquote(desc)
        if kind == "ALT":
            self.hstring = '##ALT=<ID=%s,Description="%s">' % (self.id, self.desc)
        else:
            self.hstring = '##%s=<ID=%s,Number=%s,Type=%s,Description="%s">' % (
                kind, self.id, self.number, self.type, self.desc)


def confidence_interval(var: "Variant", tag: str, alt_tag: str, max_ci_dist) -> List[int]:
    """CIPOS/CIEND, or the 95 % interval when the full one is wider than max_ci_dist
    (parsers.py:11-15)."""
    ci = [int(x) for x in var.info[tag].split(",")]
    if ci[1] - ci[0] > max_ci_dist:
        return [int(x) for x in var.info[alt_tag].split(",")]
    return ci


class Vcf:
    def __init__(self):
        self.file_format = "VCFv4.2"
        self.reference = ""
        self.sample_list: List[str] = []
        self.info_list: List[HeaderLine] = []
        self.format_list: List[HeaderLine] = []
        self.alt_list: List[HeaderLine] = []
        self.header_misc: List[str] = []
        self.filename: Optional[str] = None
        self.format_rank: Dict[str, int] = {}
        self.info_plans: Dict[tuple, tuple] = {}      # see Variant.get_info_string
        self._bnd_pending: Dict[str, "Variant"] = {}   # first mates waiting for their partner
        self._bnd_first: Dict[str, "Variant"] = {}     # first mates of completed pairs, by breakpoint id
        self.add_format("GT", 1, "String", "Genotype")

    # ---- declarations (first one wins: parsers.py:102-115)
    def add_info(self, id, number, type, desc):
        if str(id) not in [h.id for h in self.info_list]:
            self.info_list.append(HeaderLine("INFO", id, number, type, desc))
            self.info_plans.clear()

    def add_alt(self, id, desc):
        if str(id) not in [h.id for h in self.alt_list]:
            self.alt_list.append(HeaderLine("ALT", id, desc=desc))

    def add_format(self, id, number, type, desc):
        if str(id) not in [h.id for h in self.format_list]:
            self.format_list.append(HeaderLine("FORMAT", id, number, type, desc))
            self.format_rank = {h.id: i for i, h in enumerate(self.format_list)}   # header order of FORMAT keys

    def add_sample(self, name):
        self.sample_list.append(name)

    def add_custom_svtyper_headers(self):
        for spec in SVTYPER_INFO:
            self.add_info(*spec)
        for spec in SVTYPER_FORMATS:
            self.add_format(*spec)

    def add_header(self, header_lines):
        """parsers.py:49-72"""
        for line in header_lines:
            key = line.split("=")[0]
            if key == "##fileformat":
                self.file_format = line.rstrip().split("=")[1]
            elif key == "##reference":
                self.reference = line.rstrip().split("=")[1]
            elif key in ("##INFO", "##ALT", "##FORMAT"):
                body = line[line.find("<") + 1:line.rfind(">")]
                values = [f.split("=")[1] for f in _FIELD_RE.findall(body)]
                getattr(self, {"##INFO": "add_info", "##ALT": "add_alt", "##FORMAT": "add_format"}[key])(*values)
            elif line[0] == "#" and line[1] != "#":
                self.sample_list = line.rstrip().split("\t")[9:]
            elif line.startswith("##fileDate="):
                pass
            else:
                self.header_misc.append(line.rstrip())

    def get_header(self) -> str:
        lines = ["##fileformat=" + self.file_format, "##fileDate=" + time.strftime("%Y%m%d"),
                 "##reference=" + self.reference]
        lines += [h.hstring for h in self.info_list]
        lines += [h.hstring for h in self.alt_list]
        lines += [h.hstring for h in self.format_list]
        lines += self.header_misc
        lines.append("\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]
                               + self.sample_list))
        return "\n".join(lines)

    def write_header(self, fd=None):
        print(self.get_header(), file=fd if fd is not None else sys.stdout)

    def sample_to_col(self, sample) -> int:
        return self.sample_list.index(sample) + 9

    # ---- breakpoints (parsers.py:125-223)
    def get_variant_breakpoints(self, variant: "Variant", max_ci_dist) -> Optional[dict]:
        """{'id','svtype',['var_length'],'A','B'} with the +1 shift of reverse-strand sides applied;
        None for the first mate of a BND pair (it is kept until its partner arrives)."""
        svtype = variant.get_svtype()
        if svtype == "BND":
            mate = self._bnd_pending.get(variant.info["MATEID"])
            if mate is None:
                self._bnd_pending[variant.var_id] = variant
                return None
            first, second = mate, variant
            bp = {
                "id": first.var_id, "svtype": "BND",
                "A": {"chrom": first.chrom, "pos": first.pos,
                      "ci": confidence_interval(first, "CIPOS", "CIPOS95", max_ci_dist),
                      "is_reverse": first.alt[-1] not in "[]"},
                "B": {"chrom": second.chrom, "pos": second.pos,
                      "ci": confidence_interval(second, "CIPOS", "CIPOS95", max_ci_dist),
                      "is_reverse": second.alt[-1] not in "[]"},
            }
            del self._bnd_pending[first.var_id]
            self._bnd_first[first.var_id] = first
        else:
            strands = {"DEL": (False, True), "DUP": (True, False), "INV": (False, False)}[svtype]
            pos_b = int(variant.get_info("END"))
            bp = {
                "id": variant.var_id, "svtype": svtype,
                "A": {"chrom": variant.chrom, "pos": variant.pos,
                      "ci": confidence_interval(variant, "CIPOS", "CIPOS95", max_ci_dist), "is_reverse": strands[0]},
                "B": {"chrom": variant.chrom, "pos": pos_b,
                      "ci": confidence_interval(variant, "CIEND", "CIEND95", max_ci_dist), "is_reverse": strands[1]},
            }
            if svtype == "DEL":
                bp["var_length"] = pos_b - variant.pos
        for side in ("A", "B"):
            if bp[side]["is_reverse"]:
                bp[side]["pos"] += 1
        return bp


class Variant:
    def __init__(self, var_list: List[str], vcf: Vcf):
        if len(var_list) < 8:
            sys.stderr.write("Error: VCF file must have at least 8 columns\n")
            sys.exit(1)
        self.chrom = var_list[0]
        self.pos = int(var_list[1])
        self.var_id = var_list[2]
        self.ref = var_list[3]
        self.alt = var_list[4]
        self.qual = 0 if var_list[5] == "." else float(var_list[5])
        self.filter = var_list[6]
        self.sample_list = vcf.sample_list
        self.info_list = vcf.info_list
        self.format_list = vcf.format_list
        self.format_rank = vcf.format_rank
        self._info_plans = vcf.info_plans          # {tuple of INFO keys: ((key, is_flag), ...) in header order}
        self.active_formats: List[str] = []
        if len(var_list) <= 9:
            # no sample column on the line (a sites-only VCF): every sample starts as GT './.'
            # (parsers.py:296-307 via the IndexError branch); the Genotype objects are only made when
            # somebody asks for one -- the bulk writer of pipeline.SampleColumnWriter never does
            self._gts: Optional[Dict[str, Genotype]] = None
            if self.sample_list:
                self.active_formats.append("GT")
        else:
            self._gts = {}
            for s in self.sample_list:
                try:
                    col = var_list[vcf.sample_to_col(s)]
                    g = Genotype(self, s, col.split(":")[0])
                    self._gts[s] = g
                    for key, value in zip(var_list[8].split(":"), col.split(":")):
                        g.set_format(key, value)
                except IndexError:
                    self._gts[s] = Genotype(self, s, "./.")
        # key=value items keep what lies between the first and the second '=' (parsers.py:260-268 takes
        # item.split('=')[1]); a bare key is a flag
        info: Dict[str, object] = {}
        for item in var_list[7].split(";"):
            key, eq, value = item.partition("=")
            if not eq:
                info[key] = True
            elif "=" in value:
                info[key] = value[:value.index("=")]
            else:
                info[key] = value
        self.info = info

    def set_info(self, field, value):
        if field in [h.id for h in self.info_list]:
            self.info[field] = value
        else:
            sys.stderr.write('Error: invalid INFO field, "' + field + '"\n')
            sys.exit(1)

    def get_info(self, field):
        return self.info[field]

    def has_svtype(self) -> bool:
        return "SVTYPE" in self.info

    def get_svtype(self):
        return self.info["SVTYPE"]

    def is_valid_svtype(self) -> bool:
        return self.get_svtype() in VALID_SVTYPES

    @property
    def gts(self) -> Dict[str, "Genotype"]:
        if self._gts is None:
            self._gts = {s: Genotype(self, s, "./.") for s in self.sample_list}
        return self._gts

    def only_default_genotypes(self) -> bool:
        """no sample of this variant carries anything but its initial GT"""
        if self._gts is None:
            return True
        for g in self._gts.values():
            if len(g.format) != 1:
                return False
        return True

    def genotype(self, sample_name) -> "Genotype":
        if sample_name in self.sample_list:
            return self.gts[sample_name]
        sys.stderr.write('Error: invalid sample name, "' + sample_name + '"\n')

    def get_info_string(self) -> str:
        # header declaration order (parsers.py:346-355); the order for a given set of keys is worked out once
        keys = tuple(self.info)
        plan = self._info_plans.get(keys)
        if plan is None:     # (Vcf.add_info clears the plans when a declaration is added)
            plan = self._info_plans[keys] = tuple((h.id, h.type == "Flag") for h in self.info_list if h.id in self.info)
        info = self.info
        return ";".join([k if flag else "%s=%s" % (k, info[k]) for k, flag in plan])

    def get_format_string(self) -> str:
        # active_formats is kept in header declaration order by Genotype.set_format / set_formats
        return ":".join(self.active_formats)

    def get_var_string(self) -> str:
        return "\t".join(str(x) for x in (
            self.chrom, self.pos, self.var_id, self.ref, self.alt, "%0.2f" % self.qual, self.filter,
            self.get_info_string(), self.get_format_string(),
            "\t".join(self.genotype(s).get_gt_string() for s in self.sample_list)))

    def get_var_string_with(self, format_string: str, sample_columns) -> str:
        """get_var_string() with the FORMAT column and the sample columns supplied as ready text (the bulk
        formatter of pipeline.SampleColumnWriter); the eight fixed columns are printed as always."""
        return "\t".join((self.chrom, str(self.pos), self.var_id, self.ref, self.alt, "%0.2f" % self.qual, self.filter,
                          self.get_info_string(), format_string, "\t".join(sample_columns)))

    def write(self, fd=None):
        print(self.get_var_string(), file=fd if fd is not None else sys.stdout)

    def share_genotypes_with(self, other: "Variant"):
        """BND mates are written with the first mate's QUAL and genotype objects
        (classic.py:517-521, singlesample.py:647-652)."""
        other.qual = self.qual
        other.active_formats = self.active_formats
        other.genotype = self.genotype


class Genotype:
    def __init__(self, variant: Variant, sample_name: str, gt: str):
        self.format: Dict[str, object] = {}
        self.variant = variant
        self.set_format("GT", gt)

    def set_format(self, field, value):
        rank = self.variant.format_rank
        if field not in rank:
            sys.stderr.write('Error: invalid FORMAT field, "' + field + '"\n')
            sys.exit(1)
        self.format[field] = value
        active = self.variant.active_formats
        if field not in active:
            active.append(field)
            active.sort(key=rank.__getitem__)   # header declaration order (parsers.py:375-381)

    def set_formats(self, items):
        """set_format for several (field, value) pairs with one re-ordering of the active list"""
        rank = self.variant.format_rank
        fmt = self.format
        active = self.variant.active_formats
        grew = False
        for field, value in items:
            if field not in rank:
                sys.stderr.write('Error: invalid FORMAT field, "' + field + '"\n')
                sys.exit(1)
            fmt[field] = value
            if field not in active:
                active.append(field)
                grew = True
        if grew:
            active.sort(key=rank.__getitem__)

    def get_format(self, field):
        return self.format[field]

    def get_gt_string(self) -> str:
        cells = []
        for f in self.variant.active_formats:
            if f in self.format:
                v = self.format[f]
                cells.append("%0.2f" % v if type(v) == float else str(v))
            else:
                cells.append(".")
        return ":".join(cells)
