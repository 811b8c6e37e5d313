/*
 * svtyper_hip.h -- C ABI of the MI355X-native SVTyper likelihood hot path.
 *
 * The library behind this header (svtyper_amd/csrc/libsvtyper_hip.so) is the
 * drop-in for ONE path of hall-lab/svtyper v0.7.1: per (breakpoint, sample)
 * "evidence tally -> bayes_gt likelihood -> GT/GQ/SQ decision".  Everything a
 * caller hands over is a plain pointer + size; nothing here depends on torch,
 * numpy or Python.  A Python binding (ctypes) lives in svtyper_amd/hip.py; the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *
 *   svt_batch_create + svt_batch_genotype + svt_batch_results
 *       == the per-sample block of svtyper/classic.py:286-513 (sv_genotype)
 *       == svtyper/singlesample.py:355-404 tally_variant_read_fragments()
 *          + svtyper/singlesample.py:406-473 bayesian_genotype()
 *          + svtyper/singlesample.py:207-243,478-496 blank results
 *   inside the kernels:
 *       svtyper/utils.py:74-75          prob_mapq()
 *       svtyper/parsers.py:861-882      SamFragment.p_concordant()
 *       svtyper/parsers.py:579-583      Library.calc_insert_density()
 *       svtyper/statistics.py:9-20      log_choose()
 *       svtyper/statistics.py:23-37     bayes_gt()
 *
 * There is NO CPU fallback in this library: every compute entry point needs a
 * gfx950 device and returns SVT_ERR_NO_DEVICE otherwise.  The CPU restatement
 * used to check it lives under oracle/ and is test infrastructure only.
 */
#ifndef SVTYPER_HIP_H
#define SVTYPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_ABI_VERSION 18

/* ---- error codes (0 = ok, <0 = error; text via svt_last_error()) ---------- */
#define SVT_OK 0
#define SVT_ERR_INVALID (-1)   /* malformed batch (bad offsets, lib index, ...) */
#define SVT_ERR_NO_DEVICE (-2) /* no HIP device / device index out of range   */
#define SVT_ERR_HIP (-3)       /* a HIP runtime call failed                   */
#define SVT_ERR_NOMEM (-4)
#define SVT_ERR_STATE (-5)     /* results requested before genotype, ...      */
#define SVT_ERR_INTERNAL (-6)  /* an unexpected failure inside the library     */
#define SVT_ERR_UNSUPPORTED (-7) /* svt_pack_evidence: the batch cannot be expressed as packed evidence
                                    (use the canonical records)                 */

/* ---- SV types (classic.py:228 accepts exactly these four) ----------------- */
#define SVT_SVTYPE_DEL 0
#define SVT_SVTYPE_DUP 1
#define SVT_SVTYPE_INV 2
#define SVT_SVTYPE_BND 3

/* ---- flags of svt_batch_create ------------------------------------------- */
/* floating-point association of the three split-read tallies:
 *   0 (classic): site total += each contribution directly (classic.py:306-328)
 *   1 (sso)    : contributions are first summed per fragment starting from 0,
 *                then added to the site total (singlesample.py:246-276,367-372) */
#define SVT_FLAG_SSO_ASSOCIATION 0x1u
/* keep every look-up table in L2 (the pass's general mode, which also evaluates the histogram keys in exact 64-bit
 * arithmetic) even where a workgroup could stage its libraries' histograms in LDS -- one library or many: for
 * measurements and for tests that compare the table paths.  Results are the same.                              */
#define SVT_FLAG_GENERAL_TABLES 0x10u
/* the pass writes 96-byte result records on the device (svt_result96: the record of SURVEY.md section 8(d) -- GL, SQ, the
 * five tallies, QR / QA / GQ, GT -- plus the index of the unit it belongs to) instead of 128-byte ones, IN THE ORDER THE KERNEL
 * FINISHES THEM: a workgroup sorts its units by length, and a wave's 64 records leave as 6 KB of whole cache lines, tagged,
 * instead of as 64 part-line writes at their units' positions (96-byte records in unit order were measured 4 % SLOWER than
 * 128-byte ones; in the kernel's order they are 2-4 % faster and a quarter fewer bytes).  The eight counts the record leaves
 * out are truncations of sums of the tallies (classic.py:455-469).  svt_batch_results / svt_genotype still fill
 * svt_result[n_units] in unit order, bit for bit the same: they put every record where its tag says and restore the counts.
 * Only a caller that reads the DEVICE records itself (svt_batch_device_results, svt_batch_bind_device_results, the RCCL
 * gather) sees this form: svt_batch_result_slots() records of svt_batch_result_bytes() bytes, some of them padding
 * (unit == SVT_NO_UNIT); svt_results_expand96 turns gathered records into svt_result[].                                */
#define SVT_FLAG_RESULT96 0x20u
/* (bits 1..3 selected round 1's tiled device layouts, which are gone: they are rejected as unknown bits.)
 * Device layout of a resident batch: nothing is re-tiled or re-encoded -- the CSR arrays go to HBM as the
 * caller packed them and ONE kernel (svt_stream_kernel) takes them to the result records, streaming every
 * record from HBM exactly once through per-wave LDS rings.  svt_batch_create is upload only; the record
 * contract is checked by the pass itself, so a malformed record is reported by svt_batch_genotype(sync) /
 * svt_batch_results / svt_genotype instead of svt_batch_create.  (svt_batch_create_packed: the same for
 * packed evidence, below.)                                                                            */

/* ---- evidence record: one per read-fragment (query name) of a unit, 16 B --
 * Records of a unit are stored in the order the reference walks them:
 * sorted(query_name) (classic.py:296, singlesample.py:364).
 *
 * The four "weight" byte pairs hold MAPQ values that are already gated by the
 * reference's geometry predicates, using prob_mapq(0) == 1 - 10**0 == 0.0
 * exactly (utils.py:74-75): a gated-off read carries MAPQ 0 and therefore adds
 * +0.0, which is what "not adding" means for the non-negative binary64 sums.  */
typedef struct svt_record {
    int32_t ospan_len; /* |readB.reference_end - readA.reference_start|
                          (parsers.py:792-796,866-869); 0 when < 2 primaries  */
    uint8_t mapq_a;    /* MAPQ of primary read A = primary_reads[0]
                          (parsers.py:756-768); 0 when absent                 */
    uint8_t mapq_b;    /* MAPQ of primary read B = primary_reads[1]           */
    uint8_t rs_a;      /* mapq_a if is_ref_seq(readA) at A or B else 0
                          (classic.py:306-311: ref_seq += prob_mapq(read))    */
    uint8_t rs_b;      /* same for read B                                     */
    uint8_t seq_l;     /* non-soft-clip split candidate (classic.py:317-328):
                          MAPQ of query_left  if is_split_straddle()[0] else 0 */
    uint8_t seq_r;     /* MAPQ of query_right if is_split_straddle()[1] else 0 */
    uint8_t clip_l;    /* soft-clip-only candidate (is_soft_clip, parsers.py:983):
                          same two gated MAPQs (the dummy piece has MAPQ 0,
                          parsers.py:976-981)                                  */
    uint8_t clip_r;
    uint32_t flags;    /* SVT_REC_* bits, library index in bits 8..23         */
} svt_record;

#define SVT_REC_ALT_STRADDLE (1u << 0)   /* is_pair_straddle(A,B,o1,o2) OR, for INV, the
                                            strand-flipped reciprocal (classic.py:342-359).
                                            Evaluated WITHOUT the small-DEL gate: the
                                            kernel applies classic.py:339-340 itself.    */
#define SVT_REC_REF_STRADDLE_A (1u << 1) /* is_pair_straddle(A,A,[0,0],F,T) (classic.py:387-391) */
#define SVT_REC_REF_STRADDLE_B (1u << 2) /* is_pair_straddle(B,B,[0,0],F,T) (classic.py:392-396) */
#define SVT_REC_CONTINUATION (1u << 3)   /* this record continues the previous record's
                                            fragment (a fragment with > 2 primaries, or two
                                            split candidates of the same kind: the extra
                                            read / candidate goes into an extra record);
                                            only affects the SVT_FLAG_SSO_ASSOCIATION
                                            summation order                                */
#define SVT_REC_HAS_PAIR (1u << 4)       /* num_primary == 2 (parsers.py:827): required by
                                            the three straddle bits                        */
#define SVT_REC_LIB_SHIFT 8              /* bits 8..23: index into svt_evidence_batch.libs
                                            (fragment.lib).  ABI 18 widened it from eight
                                            bits (bits 16..23 had to be 0 before, so records
                                            written for an older library read the same): the
                                            reference's `-B a.bam,b.bam,...` list is unbounded
                                            (classic.py:145-158, parsers.py:432-447)        */
#define SVT_REC_LIB(flags) (((flags) >> SVT_REC_LIB_SHIFT) & 0xffffu)
#define SVT_REC_FLAG_MASK 0x00ffff1fu    /* every other bit must be 0                      */

/* ---- unit header: one per (breakpoint, sample), 16 B ---------------------- */
typedef struct svt_unit {
    int32_t var_length; /* DEL: posB - posA before the strand increments
                           (classic.py:268, parsers.py:182); ignored otherwise:
                           non-DEL uses lib.mean + 3*lib.sd (parsers.py:874-875) */
    int32_t pos_delta;  /* posB - posA AFTER the +1 strand increments
                           (classic.py:276-277); the small-DEL gate
                           "posB - posA < 2 * lib.sd" (classic.py:339,383)      */
    uint16_t sample;    /* sample index (informational; libraries are per record) */
    uint8_t svtype;     /* SVT_SVTYPE_*                                          */
    uint8_t flags;      /* SVT_UNIT_* bits                                       */
    uint32_t libs;      /* optional hint, SVT_UNIT_LIBS(first, count): the libraries of
                           the unit's sample (sample.lib_dict, parsers.py:432-447) are
                           libs[first .. first + count) of the batch and every record
                           of the unit names one of them; 0 = no hint.  With the hint
                           on every unit of a several-library batch the pass stages
                           only a sample's histograms per workgroup;
                           a record outside its unit's window is a contract violation.
                           Without hints a batch whose libraries all fit LDS together
                           (a sample with a few read-group libraries) is treated as ONE
                           window; for a joint batch of many samples svt_batch_create
                           reads the windows off the uploaded records itself (one
                           streaming pass on the device, ~0.3 ms per 1.6 GB); only the
                           pipelined one-shot (svt_genotype), which uploads while it
                           runs, then takes the slow general mode.
                           first < 65536, count <= 255; bits 24..31 must be 0.    */
} svt_unit;
/* (first's low byte | count << 8 | first's high byte << 16: what ABI <= 17 wrote for first < 256, unchanged) */
#define SVT_UNIT_LIBS(first, count) (((uint32_t)(first) & 0xffu) | ((uint32_t)(count) & 0xffu) << 8 | (((uint32_t)(first) >> 8) & 0xffu) << 16)
#define SVT_UNIT_LIBS_FIRST(x) (((uint32_t)(x) & 0xffu) | (((uint32_t)(x) >> 16) & 0xffu) << 8)
#define SVT_UNIT_LIBS_COUNT(x) (((uint32_t)(x) >> 8) & 0xffu)

#define SVT_UNIT_SKIP (1u << 0) /* too many reads: GT './.' only (classic.py:282-284,
                                   singlesample.py:478-480)                      */

/* ---- library: insert-size histogram + moments (parsers.py:406-587) -------- */
typedef struct svt_library {
    const uint32_t* hist; /* dense counts: hist[k - key_min] = Library.hist[k]   */
    int32_t key_min;      /* smallest histogram key                              */
    uint32_t n_bins;      /* key_max - key_min + 1                               */
    double mean;          /* Library.mean                                        */
    double sd;            /* Library.sd                                          */
} svt_library;

/* ---- a batch of units in CSR form (host memory, caller-owned, read-only) --- */
typedef struct svt_evidence_batch {
    uint64_t n_units;
    const uint64_t* rec_offset; /* n_units + 1 entries, rec_offset[0] == 0        */
    const svt_unit* units;      /* n_units                                        */
    const svt_record* records;  /* rec_offset[n_units]                            */
    uint32_t n_libs;            /* 1..65536 (packed evidence: 1..256)             */
    const svt_library* libs;
    double split_weight;        /* --split_weight (classic.py:38)                 */
    double disc_weight;         /* --disc_weight  (classic.py:39)                 */
} svt_evidence_batch;

/* =====================================================================================
 * Geometry stage on the device: breakpoint-dependent predicates of
 * svtyper/parsers.py -- is_ref_seq :801-816, is_pair_straddle :821-857, get_ispan/ospan
 * :785-796, check_split_support :1122-1134, is_split_straddle :1136-1215 -- evaluated by
 * svt_geometry_kernel on fixed-size, breakpoint-independent fragment summaries, producing
 * the svt_record array above (one record per summary, same order).
 * ===================================================================================== */

/* one primary read of a fragment, 32 B */
typedef struct svt_read_summary {
    int32_t tid;        /* reference id of the read (index into the BAM header), -1 when absent */
    int32_t start;      /* reference_start                                                      */
    int32_t end;        /* reference_end (start + reference-consuming CIGAR ops)                */
    int32_t iv_start[2];/* up to two maximal reference intervals covered without a D/N gap by   */
    int32_t iv_end[2];  /* M/=/X operations (an insertion does not break an interval); a read   */
                        /* with more intervals keeps the two nearest to the unit's breakends    */
    uint8_t mapq;
    uint8_t flags;      /* SVT_READ_* */
    uint16_t reserved;
} svt_read_summary;
#define SVT_READ_PRESENT (1u << 0)
#define SVT_READ_REVERSE (1u << 1)

/* one piece of a split-read candidate (parsers.py:902-950), 16 B */
typedef struct svt_piece_summary {
    int32_t tid;        /* -2 for the dummy piece of a soft-clip-only candidate (chrom None)    */
    int32_t start;      /* reference_start */
    int32_t end;        /* reference_end   */
    uint8_t mapq;
    uint8_t flags;      /* SVT_READ_PRESENT (candidate exists), SVT_READ_REVERSE                */
    uint16_t reserved;
} svt_piece_summary;

/* one read-fragment (query name) of a unit, 128 B.  A fragment with more than two primaries, or
 * with two split candidates of the same kind, continues in a following summary that has
 * SVT_FRAG_CONTINUATION set (host side: svtyper_amd/geometry.py).                              */
typedef struct svt_fragment {
    svt_read_summary read[2];   /* primary_reads[0], primary_reads[1] (parsers.py:756-768)      */
    svt_piece_summary seq[2];   /* valid non-soft-clip candidate: query_left, query_right        */
    svt_piece_summary clip[2];  /* valid soft-clip-only candidate: query_left, query_right       */
} svt_fragment;
/* fragment-level bits live in read[0].reserved (the library index, 16 bits) and read[1].reserved */
#define SVT_FRAG_PAIR (1u << 0)         /* read[1].reserved: num_primary == 2 (parsers.py:827)   */
#define SVT_FRAG_CONTINUATION (1u << 1) /* read[1].reserved                                       */

/* one (breakpoint, sample) unit for the geometry stage, 48 B (parsers.py:149-154,190-203) */
typedef struct svt_breakpoint {
    int32_t tid_a, pos_a, ci_a[2];   /* side A: chromosome id, position (+1 applied on reverse   */
    int32_t tid_b, pos_b, ci_b[2];   /* sides, classic.py:276-277), confidence interval           */
    int32_t var_length;              /* DEL: END - POS (classic.py:268); else 0                   */
    uint16_t sample;
    uint8_t svtype;                  /* SVT_SVTYPE_* */
    uint8_t flags;                   /* bit0: side A is_reverse, bit1: side B is_reverse,
                                        bit2: SVT_UNIT_SKIP                                       */
    uint32_t reserved[2];            /* [0]: SVT_UNIT_LIBS(first, count) hint of the unit's sample, 0 = none (svt_unit.libs);
                                        [1]: must be 0                                            */
} svt_breakpoint;
#define SVT_BP_REV_A (1u << 0)
#define SVT_BP_REV_B (1u << 1)
#define SVT_BP_SKIP (1u << 2)

typedef struct svt_fragment_batch {
    uint64_t n_units;
    const uint64_t* frag_offset;     /* n_units + 1 */
    const svt_breakpoint* breakpoints;
    const svt_fragment* fragments;   /* frag_offset[n_units] */
    uint32_t n_libs;
    const svt_library* libs;
    double split_weight, disc_weight;
    int32_t min_aligned;             /* -m / --min_aligned (classic.py:34)          */
    int32_t split_slop;              /* 3 (classic.py:184)                          */
} svt_fragment_batch;

/* ---- genotype codes --------------------------------------------------------- */
#define SVT_GT_HOMREF 0    /* '0/0' */
#define SVT_GT_HET 1       /* '0/1' */
#define SVT_GT_HOMALT 2    /* '1/1' */
#define SVT_GT_MISSING (-1) /* evidence present but sum(10**GL) underflowed to 0:
                               GT './.', GQ '.', SQ '.' (classic.py:492-495); GL and
                               the counts are still valid                          */
#define SVT_GT_BLANK (-2)   /* all five tallies are 0: blank result (classic.py:496-513) */
#define SVT_GT_SKIPPED (-3) /* SVT_UNIT_SKIP                                        */

/* order of the ten integer FORMAT counts + GQ inside svt_result.counts */
enum {
    SVT_CNT_QR = 0, SVT_CNT_QA, SVT_CNT_GQ, SVT_CNT_DP, SVT_CNT_RO, SVT_CNT_AO,
    SVT_CNT_RS, SVT_CNT_AS, SVT_CNT_ASC, SVT_CNT_RP, SVT_CNT_AP, SVT_N_COUNTS
};
/* order inside svt_result.tallies (after the zeroing rules, classic.py:425-435) */
enum { SVT_TAL_REF_SEQ = 0, SVT_TAL_ALT_SEQ, SVT_TAL_ALT_CLIP, SVT_TAL_REF_SPAN,
       SVT_TAL_ALT_SPAN, SVT_N_TALLIES };

/* ---- result record: one per unit, 128 B (one L2 line, written by one lane) -- */
typedef struct svt_result {
    double gl[3];                 /* log10 likelihoods homref/het/homalt (statistics.py:33-37) */
    double sq;                    /* 0 when gt < 0                                             */
    double tallies[SVT_N_TALLIES];
    int32_t counts[SVT_N_COUNTS]; /* GQ = -1 when gt < 0                                       */
    int8_t gt;                    /* SVT_GT_*                                                  */
    uint8_t pad[11];              /* zero                                                      */
} svt_result;

/* ---- the same record without the counts that follow from the tallies, 96 B (SVT_FLAG_RESULT96): bytes 0..83 are
 * bytes 0..83 of svt_result (gl, sq, tallies, QR, QA, GQ), byte 84 is gt ------------------------------------------ */
#define SVT_NO_UNIT 0xFFFFFFFFu    /* svt_result96.unit of a padding record */
typedef struct svt_result96 {
    double gl[3];
    double sq;
    double tallies[SVT_N_TALLIES];
    int32_t qr, qa, gq;           /* counts[SVT_CNT_QR], counts[SVT_CNT_QA], counts[SVT_CNT_GQ] */
    int8_t gt;
    uint8_t pad[3];               /* zero                                                      */
    uint32_t unit;                /* where the record belongs: index into the results in unit order (with
                                     svt_batch_result_order: the site-major index), SVT_NO_UNIT = padding   */
    uint32_t pad2;                /* zero                                                      */
} svt_result96;

typedef struct svt_batch svt_batch; /* opaque: device-resident packed batch */

/* ---- entry points ---------------------------------------------------------- */

/* ABI version of the loaded library (== SVT_ABI_VERSION). */
int svt_version(void);

/* Number of usable HIP devices (0 when none; never fails). */
int svt_device_count(void);

/* Text of the last error raised on the calling thread ("" when none). */
const char* svt_last_error(void);

/* Make `in` resident in HBM on `device`: validates the unit arrays, builds the look-up tables and
 * uploads the CSR as it is (DESIGN.md "HBM layout").  `flags`: SVT_FLAG_*.
 * Replaces the hand-over of `read_batch` to the per-sample block of
 * classic.py:279-296 / the `sam_fragments` argument of singlesample.py:355.     */
int svt_batch_create(const svt_evidence_batch* in, int device, unsigned flags,
                     svt_batch** out);

/* The same with the RECORDS handed over in pieces (ABI 16): `in->records` is ignored; the record array of the batch is the
 * concatenation of `segments[0 .. n_segments)` (their lengths add up to in->rec_offset[in->n_units]; a segment may be empty,
 * unit boundaries need not coincide with segment boundaries).  For a joint run over several samples whose evidence comes from
 * one reader per sample (svt_bam_evidence: one record array per BAM): the units go in SAMPLE-major -- rec_offset / units
 * concatenated by the caller, 24 bytes per unit -- every sample's records go up from where its reader left them, and
 * svt_batch_result_order makes the pass write the result records site-major.  Nobody concatenates or interleaves 16 bytes per
 * fragment on the host (classic.py:279-296 walks samples inside sites; the evidence is produced sample by sample).
 * Same validation, same errors and the same batch as svt_batch_create over the concatenated array.                         */
typedef struct svt_record_segment {
    const svt_record* records;
    uint64_t n_records;
} svt_record_segment;
int svt_batch_create_segments(const svt_evidence_batch* in, const svt_record_segment* segments, uint32_t n_segments,
                              int device, unsigned flags, svt_batch** out);

/* One pass of the hot path over the resident batch: tally -> zeroing rules ->
 * QR/QA -> bayes_gt -> GT/GQ/SQ, results left in HBM.  Asynchronous on the
 * batch's stream unless `sync` != 0.   (classic.py:296-513)                     */
int svt_batch_genotype(svt_batch* b, int sync);

/* Enqueue `iters` back-to-back passes on the batch's stream without waiting (the caller
 * synchronises); equivalent to `iters` calls of svt_batch_genotype(b, 0).                  */
int svt_batch_genotype_n(svt_batch* b, int iters);

/* (ABI 14) Wait for the passes enqueued so far and report what they found (the record contract): what
 * svt_batch_genotype(b, 1) does after its launch, without a launch -- for a caller that overlaps an asynchronous pass with
 * other work (the gather of the previous batch's result records, bench.py `value_pipelined`).                        */
int svt_batch_sync(svt_batch* b);

/* Run `iters` back-to-back passes bracketed by HIP events on the batch's stream;
 * *ms_total receives the elapsed milliseconds of all `iters` launches.          */
int svt_batch_genotype_timed(svt_batch* b, int iters, float* ms_total);

/* Copy the result records of the last svt_batch_genotype to out[n_units] (blocking). */
int svt_batch_results(svt_batch* b, svt_result* out, uint64_t n_units);

/* Device pointer of the result records (svt_result[n_units]) the kernel currently
 * writes to, for a caller that keeps working on the GPU.  The record contract is checked by
 * the pass and reported by the SYNCHRONISING entry points (svt_batch_genotype(b, 1),
 * svt_batch_results, svt_batch_site_qual, svt_batch_genotype_timed): a consumer of the device
 * records runs one of them -- svt_batch_genotype(b, 1) is enough -- before trusting them.     */
int svt_batch_device_results(svt_batch* b, svt_result** dev_ptr);

/* Make the kernel write its result records straight into a caller-owned DEVICE buffer, 128-byte aligned (e.g. a torch
 * tensor that is then gathered over RCCL).  ONE size holds for both record forms:
 *     svt_batch_result_slots(b) * svt_batch_result_bytes(b)   bytes
 * -- n_units * 128 for svt_result records; under SVT_FLAG_RESULT96 the slots are whole workgroups of tagged 96-byte records
 * (padding included), which for many small window chunks can be up to about twice n_units.  The buffer must stay alive until
 * svt_batch_destroy or the next bind; pass NULL to return to the library's own buffer.
 * svt_batch_bind_device_results2 (ABI 14) takes the buffer's capacity and refuses one that is too small (SVT_ERR_INVALID);
 * the form without a capacity trusts the caller.  Either way a pass whose workgroup plan needs more slots than the bound
 * buffer was sized for fails with SVT_ERR_STATE instead of writing beyond it.                                              */
int svt_batch_bind_device_results(svt_batch* b, svt_result* dev_ptr);
int svt_batch_bind_device_results2(svt_batch* b, void* dev_ptr, uint64_t capacity_bytes);

/* Optional, for a resident batch that is passed over many times -- and for the first batch of a chunked run, whose buffers the
 * following batches inherit through the library's pool.  Where the VRAM manager puts the records and the result records
 * moves the pass by up to 8 % (levels that last as long as the allocation and that nothing at allocation time predicts,
 * DESIGN.md 3.1).  The call allocates `result_candidates` more result buffers and `record_candidates` more record buffers
 * (filled by device copies), runs the real pass over each once the clocks are up (~40 ms of passes first), keeps the fastest
 * combination and releases the others: ~0.1-0.4 s (the launches per candidate are chosen so that the audition stays a short
 * burst: a device under seconds of uninterrupted load is measured in another state) and, transiently, candidates x buffer size
 * of HBM; 32 and 8 are good values (a dozen candidates miss the fastest blocks about every other time).  *before_ms / *after_ms
 * (may be NULL): the pass time per launch before and after.  Not for a batch whose result records are bound to a caller's
 * buffer; record candidates only for canonical records resident in the batch's own buffer.  The result records in the
 * device buffer afterwards are those of the last pass (have_results as after svt_batch_genotype).                    */
int svt_batch_tune_placement(svt_batch* b, int result_candidates, int record_candidates, float* before_ms, float* after_ms);

/* Bytes of one DEVICE result record of this batch: sizeof(svt_result), or sizeof(svt_result96) under SVT_FLAG_RESULT96
 * (what svt_batch_device_results points at, what a buffer for svt_batch_bind_device_results must hold per unit). */
uint32_t svt_batch_result_bytes(const svt_batch* b);

/* Records in the device result buffer of this batch: n_units, or under SVT_FLAG_RESULT96 the slots the pass's workgroups
 * write (whole workgroups: >= n_units, padding records included).  A buffer for svt_batch_bind_device_results holds
 * svt_batch_result_slots(b) * svt_batch_result_bytes(b) bytes.                                                       */
uint64_t svt_batch_result_slots(const svt_batch* b);

/* 96-byte records (host memory; e.g. gathered from several devices) -> svt_result[n_units]: record i goes to out[in[i].unit]
 * (padding records are skipped); DP, RO, AO, RS, AS, ASC, RP, AP are the reference's own expressions over the tallies
 * (classic.py:455-469: int() of the sums, in its order of additions), zero for blank / skipped units as the 128-byte path
 * leaves them.  SVT_ERR_INVALID when a tag is >= n_units or the records do not cover every unit exactly once.
 * Host only, no device needed.  `in` and `out` may not overlap.                                                      */
int svt_results_expand96(const svt_result96* in, uint64_t n_records, svt_result* out, uint64_t n_units);

/* Result order for a SAMPLE-MAJOR batch.  A joint run over several samples has one unit per (site, sample).  The
 * reference walks them site-major (classic.py:279: for every variant, for every sample), and that is the order
 * svt_batch_site_qual, the shard rule's `group` and the VCF writer want the RESULTS in.  The EVIDENCE, however, is
 * produced BAM by BAM, and a sample's units -- which share that sample's library window (svt_unit.libs) -- are best
 * contiguous in HBM: the pass then streams them like a one-library batch (a unit's first and last 128-byte line
 * are shared with its neighbours of the same workgroup instead of with 31 other samples' units).  So a producer
 * may hand over the units sample-major -- all sites of sample 0, then all sites of sample 1, ...: unit index
 * i = sample * n_sites + site, n_units = n_samples * n_sites -- and call this before svt_batch_genotype: the pass
 * then writes the record of unit i to index (i % n_sites) * n_samples + i / n_sites, i.e. the result array
 * (svt_batch_results, svt_batch_device_results, svt_batch_site_qual) is site-major, exactly as if the units had
 * been handed over site-major.  n_samples = 0 or 1 restores unit order.  Canonical records only.               */
int svt_batch_result_order(svt_batch* b, uint32_t n_samples);

/* Bytes the genotype kernel must move per pass by the definition of SURVEY.md
 * section 8(d): sum_u (16*F(u) + 16 + 96); and what the batch really holds in
 * HBM (records or packed slots + offsets + unit headers).                       */
int svt_batch_bytes(const svt_batch* b, uint64_t* algorithmic, uint64_t* resident);

/* Which kernel flavour the batch got: *compact = 3 for the streamed CSR, 4 for streamed packed evidence
 * (0..2 were round 1's tiled layouts); *table_mode = 0 one library, tables in LDS; 1 several libraries,
 * per-workgroup library windows in LDS; 2 general geometry, tables read through L2.          */
int svt_batch_layout(const svt_batch* b, int* compact, int* table_mode);

/* The HIP stream (hipStream_t) the batch launches on, as an opaque pointer.    */
void* svt_batch_stream(svt_batch* b);

void svt_batch_destroy(svt_batch* b);

/* Release what the library keeps between calls: the per-device scratch of svt_batch_create (the device
 * copy of the canonical records; a large hipMalloc costs ~100 ms, so it is reused), the page-locked
 * staging buffers, the encoder's arenas and the reader's pooled gather buffers (svtyper_reads.h: up to
 * 1 GiB of idle huge-page mappings).  The parked worker threads of the host stages (at most 96, idle
 * on a condition variable) stay for the life of the process.                                      */
void svt_trim(void);

/* Geometry + packing on the device: evaluates the predicates for every fragment summary of `in`
 * (svt_geometry_kernel), leaves the evidence records in HBM and builds the resident batch from
 * them -- the result is the same svt_batch svt_batch_create would build from the records that
 * svtyper_amd/packer.py derives on the host.  If `records_out` is not NULL the canonical records
 * (frag_offset[n_units] entries) are also copied back to it (used by the parity tests).        */
int svt_batch_create_from_fragments(const svt_fragment_batch* in, int device, unsigned flags,
                                    svt_record* records_out, svt_batch** out);

/* QUAL of every site of a multi-sample batch (svtyper/classic.py:216-217,485,498): the running
 * binary64 sum of SQ over the site's samples in -B order, reset to 0 by a sample without evidence
 * (GT code SVT_GT_BLANK), untouched by skipped and './.' samples.  The batch's units must be
 * site-major (unit = site * n_samples + sample; n_units = n_sites * n_samples) and genotyped.
 * `initial` (host, n_sites doubles: the incoming QUAL under --sum_quals) may be NULL = all 0;
 * `qual_out` (host) receives n_sites doubles.                                                  */
int svt_batch_site_qual(svt_batch* b, uint32_t n_samples, const double* initial, double* qual_out,
                        uint64_t n_sites);

/* ---- host-side helper: result records -> text of VCF sample columns ---------------------------
 * The values the reference writes per sample (svtyper/classic.py:454-513, singlesample.py:207-227,
 * 430-471) joined with ':' in the order `fields` gives (the FORMAT order of the VCF header), for every
 * unit: GT "0/0|0/1|1/1|./.", GQ int or ".", SQ "%0.2f" or ".", GL "%.0f,%.0f,%.0f" or ".", the ten
 * integer counts, AB "%.2g" or ".".  SVT_FMT_ABSENT prints "." (a FORMAT key this sample has no value
 * for).  skipped_as_dots != 0: a unit with GT code SVT_GT_SKIPPED prints "./." and "." for everything
 * else (classic.py:282-284); 0: it prints the blank result like SVT_GT_BLANK (singlesample.py:207-227).
 * text_out / offsets_out (n_units + 1 offsets into the text, no terminators) are malloc'ed: release
 * with svt_format_free.  No GPU is involved.                                                        */
enum svt_format_field {
    SVT_FMT_GT = 0, SVT_FMT_GQ, SVT_FMT_SQ, SVT_FMT_GL, SVT_FMT_DP, SVT_FMT_RO, SVT_FMT_AO, SVT_FMT_QR,
    SVT_FMT_QA, SVT_FMT_RS, SVT_FMT_AS, SVT_FMT_ASC, SVT_FMT_RP, SVT_FMT_AP, SVT_FMT_AB,
    SVT_N_FORMAT_FIELDS, SVT_FMT_ABSENT = 255
};
int svt_format_results(const svt_result* res, uint64_t n_units, const uint8_t* fields, uint32_t n_fields,
                       int skipped_as_dots, char** text_out, uint64_t** offsets_out);
void svt_format_free(char* text, uint64_t* offsets);

/* SQ of every called unit (gt >= 0) recomputed IN PLACE from its GL with the host libm, i.e. with the very
 * calls CPython makes for svtyper/classic.py:473-481 (10 ** gl, math.log(gt_sum, 10)).  GL leaves the device
 * bit-identical to the reference's, so after this call SQ -- and every QUAL summed from it, and their '%0.2f'
 * renderings -- are bit-identical too (the device's own SQ goes through the GPU's exp10 / log: |dSQ| <= 5e-13).
 * Host only, multi-threaded; the drivers call it on every batch of results they format.                    */
int svt_results_host_sq(svt_result* res, uint64_t n_units);

/* Array form of the reference's inner operator seam statistics.bayes_gt(ref, alt, is_dup)
 * (svtyper/statistics.py:23-37) and log_choose(ref + alt, alt) (statistics.py:9-20):
 * out[4*i .. 4*i+3] = { lp_homref, lp_het, lp_homalt, log_choose } for item i.  All pointers
 * are host memory; ref[i], alt[i] >= 0 and ref[i] + alt[i] < 2^24.                           */
int svt_bayes_gt(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n,
                 double* out, int device);

/* Array form of the reference's inner operator seam bayesian_genotype(breakpoint, counts, split_weight,
 * disc_weight, debug) (svtyper/singlesample.py:406-473; the same lines as classic.py:437-495):
 * counts[5*i .. 5*i+4] = {ref_seq, alt_seq, alt_clip, ref_span, alt_span} of item i (SVT_TAL_* order) exactly as
 * tally_variant_read_fragments returned them, is_dup[i] = (breakpoint['svtype'] == 'DUP').  Like the reference
 * function it applies no zeroing rule and genotypes all-zero counts like any others (its callers take the blank
 * result before calling it), so out[i].gt is never SVT_GT_BLANK / SVT_GT_SKIPPED.  All pointers are host memory;
 * counts finite, >= 0, weighted totals < 2^24.                                                              */
int svt_genotype_counts(const double* counts, const uint8_t* is_dup, uint64_t n, double split_weight,
                        double disc_weight, svt_result* out, int device);

/* Convenience: create + genotype + results + destroy.                          */
int svt_genotype(const svt_evidence_batch* in, svt_result* out, int device,
                 unsigned flags);

/* ---- packed evidence: what a host producer emits when the bytes have to cross PCIe -------------------------
 * The canonical record spends 16 bytes on a fragment whose evidence is mostly "two MAPQ-60 reads and an insert
 * size".  Packed evidence is the same information as three sparse streams of small entries per unit, in 16-byte
 * slots, unit after unit in caller order:
 *   stream 0  pair entries (-> alt_span, ref_span): straddle bits + ospan_len translated into the index space of
 *             the library's histogram table; ONE half-word when the two MAPQs are the batch's most common pair
 *             (`common_mapq`; 60, 60 for bwa), two half-words otherwise; eight half-words per slot;
 *   stream 1  reference-read entries (-> ref_seq): the two gated MAPQs rs_a, rs_b; seven per slot + 7 first-of-
 *             fragment bits;
 *   stream 2  split / clip candidate entries (-> alt_seq / alt_clip): seq_l, seq_r or clip_l, clip_r; seven per
 *             slot + first-of-fragment and is-clip bits
 * (bit layouts: svtyper_amd/csrc/svt_entry_formats.h).  Entries that could only add +0.0 -- no straddle bit, a
 * zero MAPQ, a DEL below the small-deletion gate of classic.py:339,383, all-zero weight pairs -- are not stored;
 * the order inside every stream is the record order, so every tally receives the reference's additions in the
 * reference's order and the results are bit-identical to those of the canonical records.  About 3.2 bytes per
 * fragment record on BASELINE.json's workloads instead of 16.
 * Several libraries (a multi-sample batch, a sample sequenced more than once): the pair entries of a unit stay in
 * ONE stream in record order -- the sums are order-dependent -- and a library-switch half-word stands in front of
 * the first entry coded against another library than the one before (a sample with one library: one switch per
 * unit, 2 bytes; interleaved libraries: ~4.5 bytes per record).  The pass then reads the histogram tables through
 * L2 instead of LDS (svt_packed_kernel<several libraries>); results bit-identical as before.
 * Limits: libraries of at most 2047 histogram bins, DEL lengths >= 0, |var_length| <= 2^30, |key_min| <= 2^29,
 * mean + 3 sd of a library not within 4e-6 of an integer (svt_pack_evidence returns SVT_ERR_UNSUPPORTED
 * otherwise and the caller keeps the canonical records).                                                     */
typedef struct svt_packed_evidence {
    uint64_t n_units;
    uint64_t n_slots;            /* 16-byte slots of all units                                              */
    uint64_t n_records;          /* fragment records the slots were packed from (svt_batch_bytes)           */
    const uint32_t* slot_offset; /* 3 * n_units + 1 entries: stream k of unit u = slots [slot_offset[3u + k],
                                    slot_offset[3u + k + 1]); slot_offset[0] == 0                           */
    const svt_unit* units;       /* n_units                                                                 */
    const void* slots;           /* n_slots * 16 bytes                                                      */
    uint32_t common_mapq;        /* mapq_a | mapq_b << 8 of the one-half-word pair entries                  */
    uint32_t n_libs;             /* 1..256; several: a unit's pair stream starts in the context of libs[0] and
                                    carries a library-switch half-word (l + 1) << 3 in front of the first entry
                                    coded against libs[l]; every n_bins <= 2047                             */
    const svt_library* libs;
    double split_weight;
    double disc_weight;
} svt_packed_evidence;

/* Page-locked host memory from the library's pool (plain memory when no device is present): buffers a caller
 * fills or reads right before / after a transfer -- an output array for svt_batch_results, the arrays of a
 * svt_packed_evidence it writes itself -- then move by straight DMA instead of through the staging ring.
 * (The library does not page-lock CALLER memory: unmapping such pages later stalls the GPU queues.)          */
void* svt_pinned_alloc(size_t bytes);
void svt_pinned_free(void* p);

/* Encode a batch of canonical records (host only, multi-threaded; the record contract is checked here, so the
 * pass over packed evidence does not check it again).  The slots are placed in page-locked host memory when a
 * device is present, so svt_batch_create_packed can DMA them without a staging copy.  A producer that never
 * materialises canonical records (the host packer, svtyper_amd/packer.py; a reader) calls this per chunk of
 * units it has just produced -- or writes the slots itself.  Release with svt_packed_free.                    */
int svt_pack_evidence(const svt_evidence_batch* in, svt_packed_evidence** out);
void svt_packed_free(svt_packed_evidence* p);

/* svt_batch_create for packed evidence (flags: SVT_FLAG_SSO_ASSOCIATION only): upload + tables; the pass
 * (svt_batch_genotype) is one launch of svt_packed_kernel over the slots as they were uploaded.               */
int svt_batch_create_packed(const svt_packed_evidence* in, int device, unsigned flags, svt_batch** out);

/* create_packed + genotype + results + destroy.                                                              */
int svt_genotype_packed(const svt_packed_evidence* in, svt_result* out, int device, unsigned flags);

/* The same from canonical records in HOST memory, with the encoder running ahead of the wire: the batch is encoded in
 * ranges of whole units, and every finished range is uploaded, genotyped by its own launch and downloaded while the host
 * threads encode the next one -- the wall time of the route is the longer of encoding and transfer, not their sum.  Same
 * result bytes as svt_pack_evidence + svt_genotype_packed (and as svt_genotype over the same records).  Any number of
 * libraries (histograms wider than 2047 bins: SVT_ERR_UNSUPPORTED, the packed format's limit); small batches take the
 * plain sequence.  On any error `out` is UNDEFINED: ranges in front of a range that breaks the record contract have been
 * genotyped and written by then.
 * Replaces, for a producer that holds a batch of fragments in host memory, the hand-over at singlesample.py:355.      */
int svt_genotype_packed_from_records(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags);

/* ---- several GPUs of one node from ONE process (no torch, no MPI) -------------------------------------
 * The multi-device form of svt_genotype: the replacement for the multiprocessing.Pool of
 * svtyper/singlesample.py:723-751 (`svtyper-sso --cores N`) for a C caller.  The units are cut into
 * n_devices contiguous shards balanced by the bytes a unit costs (16 F + 112) and cut only at multiples of
 * `group` units (group = samples per site keeps all samples of a site on one GPU, so QUAL over a site's
 * samples stays local; 0 / 1 = any unit boundary); one host thread per entry of devices[] runs its shard
 * (upload, ONE pass of the hot path, download) and the result records land in out[] in unit order -- the
 * concatenation is the "gather".  The same device may be listed more than once.  Units are independent, so
 * out[] is byte-identical to what svt_genotype writes for the whole batch on one device.                  */
int svt_genotype_multi(const svt_evidence_batch* in, svt_result* out, const int* devices, int n_devices,
                       uint32_t group, unsigned flags);

/* The shard rule of svt_genotype_multi (and of svtyper_amd/distributed.py: shard_bounds, which ranks of a
 * torch.distributed job use): bounds[0 .. n_shards], shard r = units [bounds[r], bounds[r + 1]).          */
int svt_shard_bounds(const uint64_t* rec_offset, uint64_t n_units, int n_shards, uint32_t group, uint64_t* bounds);

/* Batches beyond one resident batch's 32-bit record index (2^32 - 17 records = 68 GB of the 288 GB of HBM; the same
 * bound for units): the fewest contiguous chunks of at most `max_records` records (0 = the library's bound) cut at
 * multiples of `group` units (samples per site: a site's samples, hence its QUAL, stay in one chunk; 0 / 1 = any unit).
 * bounds[0 .. *n_chunks], chunk c = units [bounds[c], bounds[c + 1]); call with bounds = NULL first for the count
 * (bounds needs *n_chunks + 1 entries; `max_chunks` = the chunks it has room for).  SVT_ERR_INVALID when one group of
 * units alone exceeds the bound.  svt_batch_create refuses a larger batch; svt_genotype cuts it this way itself
 * (group 1) and runs chunk after chunk, so its out[] never depends on the size.  Units are independent (the reference
 * keeps no state across (site, sample) pairs: svtyper/classic.py:279-513).                                           */
int svt_chunk_bounds(const uint64_t* rec_offset, uint64_t n_units, uint32_t group, uint64_t max_records, uint64_t* bounds,
                     uint32_t max_chunks, uint32_t* n_chunks);

#ifdef __cplusplus
}
#endif
#endif /* SVTYPER_HIP_H */
