/*
 * svtyper_reads.h -- native (host, C++) BAM access + fragment summariser of libsvtyper_hip.so.
 *
 * The step BEFORE the device path: for every (breakpoint, sample) unit fetch the reads of the two
 * breakend regions from an indexed BAM, group them into read-fragments, run the
 * breakpoint-independent split-read QC and emit the fixed-size `svt_fragment` summaries
 * (include/svtyper_hip.h) that svt_batch_create_from_fragments consumes.  It replaces, for the
 * native pipeline, what the reference does with pysam objects in
 *   svtyper/classic.py:54-100   gather_all_reads / gather_reads            (count_mode 0)
 *   svtyper/singlesample.py:158-205  is_over_threshold / gather_reads      (count_mode 1)
 *   svtyper/parsers.py:729-768   SamFragment.__init__ / add_read
 *   svtyper/parsers.py:891-1058  SplitRead / SplitPiece / is_valid
 * and what svtyper_amd/bam.py + fragments.py + geometry.py do in Python (those stay the portable
 * implementation and are the checker of this one: tests/test_native_reads.py).
 *
 * Plain C ABI, host memory only; no GPU is needed for these calls.  BAM + .bai only (no CRAM).
 */
#ifndef SVTYPER_READS_H
#define SVTYPER_READS_H

#include <stdint.h>

#include "svtyper_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svt_bam svt_bam; /* opaque: header + index of one BAM file */

/* Open `path` (and `path`.bai or the .bai next to it).  0 or SVT_ERR_*; text via svt_last_error(). */
int svt_bam_open(const char* path, svt_bam** out);
void svt_bam_close(svt_bam* bam);

int32_t svt_bam_n_references(const svt_bam* bam);
const char* svt_bam_reference_name(const svt_bam* bam, int32_t tid);
int64_t svt_bam_reference_length(const svt_bam* bam, int32_t tid);
int32_t svt_bam_tid(const svt_bam* bam, const char* name); /* -1 when absent */
/* the @RG / other header text (NUL-terminated), owned by the handle */
const char* svt_bam_header_text(const svt_bam* bam);

/* the two fetch windows of one unit (already clamped to the chromosome, pysam semantics:
 * reads with pos < hi and end > lo) */
typedef struct svt_fetch_unit {
    int32_t tid_a, lo_a, hi_a;
    int32_t tid_b, lo_b, hi_b;
} svt_fetch_unit;

typedef struct svt_summarise_args {
    uint64_t n_units;
    const svt_fetch_unit* windows;      /* n_units */
    const svt_breakpoint* breakpoints;  /* n_units: only pos_a / pos_b are read (interval choice) */
    uint32_t n_read_groups;
    const char* const* read_groups;     /* RG ids */
    const int32_t* read_group_lib;      /* library index of each RG; -1 = library not active
                                           (prevalence below the cut, classic.py:85-87)          */
    int64_t max_reads;                  /* < 0: unlimited                                         */
    int32_t count_mode;                 /* 0: classic.py:79-93 (position of the read in the fetch of
                                           one side > max_reads); 1: singlesample.py:158-185
                                           (bam.count() of either region > max_reads; counted
                                           while the reads are gathered, same outcome)           */
    int32_t n_threads;                  /* <= 0: all hardware threads                             */
} svt_summarise_args;

typedef struct svt_summaries {
    uint64_t* frag_offset;    /* n_units + 1 */
    svt_fragment* fragments;  /* frag_offset[n_units]; owned by the library (a large one comes from its pool of
                                 huge-page mappings): release ONLY through svt_summaries_free */
    uint8_t* skipped;         /* n_units: 1 = too many reads (unit has no fragments) */
} svt_summaries;

/* Summarise all units (multi-threaded over units; every thread has its own file handle).
 * On success the three arrays of `out` are owned by the caller: release with svt_summaries_free. */
int svt_bam_summarise(const svt_bam* bam, const svt_summarise_args* args, svt_summaries* out);
void svt_summaries_free(svt_summaries* s);

/* ---- the same units as evidence records ------------------------------------------------------
 * svt_bam_evidence = svt_bam_summarise + the breakpoint-dependent predicates of the geometry stage
 * (svtyper/parsers.py:785-857,1122-1215; what svt_batch_create_from_fragments evaluates on the
 * device) in the reader's own threads: 16-byte svt_records leave the reader instead of 128-byte
 * summaries, and the batch goes to svt_batch_create / svt_genotype like any other.  The predicates
 * are ONE piece of source for both places (svtyper_amd/csrc/svt_geometry_math.h); the records are
 * those of the device stage, byte for byte (tests/test_hip_geometry.py).
 * Here every field of svt_summarise_args.breakpoints is read.                                      */
typedef struct svt_evidence_params {
    uint32_t n_libs;           /* 1..65536: size of lib_flank; a fragment of a library beyond it is an error */
    const double* lib_flank;   /* per library: mean + 3 sd, is_pair_straddle's flank (parsers.py:846-855)   */
    int32_t min_aligned;       /* -m / --min_aligned (classic.py:34)                                        */
    int32_t split_slop;        /* 3 (classic.py:184)                                                        */
} svt_evidence_params;

typedef struct svt_evidence {
    uint64_t* rec_offset;     /* n_units + 1 */
    svt_record* records;      /* rec_offset[n_units], in the units' order, sorted(query_name) inside a unit;
                                 owned by the library: release ONLY through svt_evidence_free               */
    uint8_t* skipped;         /* n_units: 1 = too many reads (unit has no records; set SVT_UNIT_SKIP)       */
} svt_evidence;

int svt_bam_evidence(const svt_bam* bam, const svt_summarise_args* args, const svt_evidence_params* geometry, svt_evidence* out);
void svt_evidence_free(svt_evidence* e);

/* Library statistics straight from the BAM (svtyper/parsers.py:501-576): what Library.from_bam scans
 * for, for ONE library given as its read-group ids, in three passes from the first record each --
 *   read_length : max query length (M/I/S/=/X) over the library's reads until 10 001 of them were seen
 *                 (calc_read_length, :516-528)
 *   hist        : Counter of template_length over the library's forward-strand, mate-reverse, mapped,
 *                 mate-mapped, primary reads with template_length > 0, until num_samp of them
 *                 (calc_insert_hist, :534-576; trimming and moments stay with the caller)
 *   in_lib/total: reads of the library among the first 100 000 records (calc_lib_prevalence, :501-513)
 * A read that has to be attributed but carries no RG tag is an error, as it is for the reference.   */
typedef struct svt_library_scan {
    int64_t read_length;
    uint64_t in_lib, total;
    uint64_t n_hist;          /* distinct template lengths                      */
    int64_t* hist_keys;       /* n_hist, in order of first occurrence (the order the
                                 reference's Counter sums in), malloc'ed        */
    uint64_t* hist_counts;    /* n_hist, malloc'ed                              */
} svt_library_scan;

int svt_bam_scan_library(const svt_bam* bam, uint32_t n_read_groups, const char* const* read_groups,
                         int64_t num_samp, svt_library_scan* out);
void svt_library_scan_free(svt_library_scan* s);

#ifdef __cplusplus
}
#endif
#endif /* SVTYPER_READS_H */
