"""Per-sample library statistics: read-group -> library map, insert-size histogram, moments.

Host-side, runs once per BAM (svtyper/parsers.py:406-719, svtyper/statistics.py:40-121,
svtyper/utils.py:25-51).  The histogram and moments are what the device's p_concordant test and
small-deletion gate consume (LibraryTable); the JSON cache format is the reference's, so a
`-l lib_info.json` written by either tool is readable by the other.
"""
from __future__ import annotations

import json
import sys
from collections import Counter
from typing import Dict, List, Optional

from .evidence import LibraryTable


# ---- Counter statistics (statistics.py:40-121) ------------------------------------------------
def _count(hist) -> int:
    return sum(hist.values())


def hist_median(hist):
    """statistics.py:46-76: midpoint of the two middle values when the count is even."""
    limit = 0.5 * _count(hist)
    values = sorted(hist)
    seen, i, v = 0, 0, values[0]
    while seen < limit:
        v = values[i]
        seen += hist[v]
        i += 1
    return (v + values[i]) / 2.0 if seen == limit else v


def hist_upper_mad(hist, med):
    """statistics.py:79-84: median absolute deviation of the values above the median."""
    resid = Counter()
    for x in hist:
        if x > med:
            resid[abs(x - med)] += hist[x]
    return hist_median(resid)


def hist_mean(hist) -> float:
    total = 0.0
    for x in hist:
        total += x * float(hist[x])
    return total / _count(hist)


def hist_stdev(hist) -> float:
    u = hist_mean(hist)
    acc = 0.0
    for x in hist:
        acc += hist[x] * (x - u) ** 2
    return (float(acc) / _count(hist)) ** 0.5


class Library:
    """Insert-size model of one sequencing library (parsers.py:406-587)."""

    def __init__(self, name, readgroups, read_length, hist, mean, sd, prevalence):
        self.name = name
        self.readgroups = list(readgroups)
        self.read_length = read_length
        self.hist = hist
        self.mean = mean
        self.sd = sd
        self.prevalence = prevalence

    # ---- from the JSON cache (parsers.py:449-470)
    @classmethod
    def from_lib_info(cls, sample_name, lib_index, lib_info) -> "Library":
        lib = lib_info[sample_name]["libraryArray"][lib_index]
        hist = {int(k): int(v) for k, v in lib["histogram"].items()}
        return cls(lib["library_name"], lib["readgroups"], int(lib["read_length"]), hist,
                   float(lib["mean"]), float(lib["sd"]), float(lib["prevalence"]))

    # ---- empirically from a BAM (parsers.py:472-583)
    @classmethod
    def from_bam(cls, lib_name, bam, num_samp, native=None) -> "Library":
        readgroups = [rg["ID"] for rg in bam.header["RG"] if rg.get("LB", "") == lib_name]
        if native is not None:       # the same three scans in the C++ reader (native_reads.NativeBam.scan_library)
            read_length, counts, in_lib, total = native.scan_library(readgroups, num_samp)
            return cls._from_scan(lib_name, readgroups, read_length, Counter(counts), in_lib, total, bam)
        rgset = set(readgroups)
        primary = lambda r: not r.is_supplementary and not r.is_secondary

        read_length, seen = 0, 0                         # calc_read_length (:516-528)
        for read in bam.fetch():
            if read.get_tag("RG") not in rgset:
                continue
            read_length = max(read_length, read.infer_query_length())
            if seen == 10000:
                break
            seen += 1

        hist = Counter()                                 # calc_insert_hist (:534-576)
        n = 0
        for read in bam.fetch():
            if (read.is_reverse or not read.mate_is_reverse or read.is_unmapped or read.mate_is_unmapped
                    or not primary(read) or read.template_length <= 0 or read.get_tag("RG") not in rgset):
                continue
            hist[read.template_length] += 1
            n += 1
            if n == num_samp:
                break
        in_lib = total = 0                               # calc_lib_prevalence (:501-513)
        for read in bam.fetch():
            if total == 100000:
                break
            if read.get_tag("RG") in rgset:
                in_lib += 1
            total += 1
        return cls._from_scan(lib_name, readgroups, read_length, hist, in_lib, total, bam)

    @classmethod
    def _from_scan(cls, lib_name, readgroups, read_length, hist, in_lib, total, bam) -> "Library":
        """The arithmetic after the three BAM scans (parsers.py:549-576,512): outlier trimming, moments,
        prevalence.  `hist` keeps the order in which the template lengths first occurred."""
        if not hist:
            sys.stderr.write("Error: failed to build insert size histogram for paired-end reads.\n"
                             "Please ensure BAM file (%s) has inward facing, paired-end reads.\n" % bam.filename)
            sys.exit(1)
        med = hist_median(hist)
        cut = med + 10 * hist_upper_mad(hist, med)
        for x in [x for x in list(hist) if x > cut]:
            del hist[x]
        mean, sd = hist_mean(hist), hist_stdev(hist)
        return cls(lib_name, readgroups, read_length, hist, mean, sd, float(in_lib) / total)

    def table(self) -> LibraryTable:
        return LibraryTable.from_counter(self.hist, self.mean, self.sd, self.name)


class Sample:
    """A BAM and its libraries (parsers.py:594-719)."""

    def __init__(self, name, bam, lib_dict, rg_to_lib, min_lib_prevalence, bam_mapped, bam_unmapped):
        self.name = name
        self.bam = bam
        self.lib_dict: Dict[str, Library] = lib_dict
        self.rg_to_lib: Dict[str, Library] = rg_to_lib
        self.bam_mapped = bam_mapped
        self.bam_unmapped = bam_unmapped
        self.active_libs = [lib.name for lib in lib_dict.values() if lib.prevalence >= min_lib_prevalence]

    @classmethod
    def from_lib_info(cls, bam, lib_info, min_lib_prevalence) -> "Sample":
        name = bam.header["RG"][0]["SM"]
        lib_dict, rg_to_lib = {}, {}
        try:
            for i, lib in enumerate(lib_info[name]["libraryArray"]):
                lib_dict[lib["library_name"]] = Library.from_lib_info(name, i, lib_info)
                for rg in lib["readgroups"]:
                    rg_to_lib[rg] = lib_dict[lib["library_name"]]
        except KeyError:
            sys.stderr.write("Error: sample %s not found in JSON library file.\n" % name)
            sys.exit(1)
        return cls(name, bam, lib_dict, rg_to_lib, min_lib_prevalence,
                   lib_info[name]["mapped"], lib_info[name]["unmapped"])

    @classmethod
    def from_bam(cls, bam, num_samp, min_lib_prevalence, native=None) -> "Sample":
        name = bam.header["RG"][0]["SM"]
        lib_dict, rg_to_lib = {}, {}
        for rg in bam.header["RG"]:
            lib_name = rg.get("LB", "")
            if lib_name not in lib_dict:
                lib_dict[lib_name] = Library.from_bam(lib_name, bam, num_samp, native)
            rg_to_lib[rg["ID"]] = lib_dict[lib_name]
        return cls(name, bam, lib_dict, rg_to_lib, min_lib_prevalence, bam.mapped, bam.unmapped)

    def get_fetch_flank(self, z):
        """Widest mean + z sd over the sample's libraries (parsers.py:691-692)."""
        return max(lib.mean + lib.sd * z for lib in self.lib_dict.values())

    def get_lib(self, readgroup) -> Library:
        return self.rg_to_lib[readgroup]

    def close(self):
        self.bam.close()


def write_sample_json(sample_list: List[Sample], lib_info_file):
    """The `-l` cache (utils.py:25-51)."""
    info = {}
    for sample in sample_list:
        info[sample.name] = {
            "sample_name": sample.name,
            "bam": sample.bam.filename if isinstance(sample.bam.filename, str) else sample.bam.filename.decode(),
            "libraryArray": [
                {"library_name": lib.name, "readgroups": lib.readgroups, "read_length": lib.read_length,
                 "mean": lib.mean, "sd": lib.sd, "prevalence": lib.prevalence, "histogram": lib.hist}
                for lib in sample.lib_dict.values()],
            "mapped": sample.bam.mapped,
            "unmapped": sample.bam.unmapped,
        }
    json.dump(info, lib_info_file, indent=4)
    lib_info_file.close()


def setup_sample(bam, lib_info: Optional[dict], num_samp: int, min_lib_prevalence: float = 1e-3, native=None) -> Sample:
    """`native`: a native_reads.NativeBam of the same file; the library scans then run in the C++ reader."""
    if lib_info is not None:
        return Sample.from_lib_info(bam, lib_info, min_lib_prevalence)
    return Sample.from_bam(bam, num_samp, min_lib_prevalence, native)
