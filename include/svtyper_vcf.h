/*
 * svtyper_vcf.h -- bulk VCF body parse + output-line emission of libsvtyper_hip.so (host, C++).
 *
 * The steps either side of the device path, for a whole chunk of variant lines at a time: what the
 * reference does per line with a `Variant` object, a breakpoint dict and fifteen `set_format` calls
 *   svtyper/parsers.py:256-310   Variant.__init__           (column split, INFO dict, QUAL)
 *   svtyper/parsers.py:11-15     confidence_interval        (CIPOS / CIEND, 95 % fallback)
 *   svtyper/parsers.py:125-223   Vcf.get_variant_breakpoints (DEL/DUP/INV strands, BND mates, +1 shift)
 *   svtyper/classic.py:219-278   the driver's per-line loop (SVTYPE checks, BND pairing)
 *   svtyper/singlesample.py:577-652  the same for svtyper-sso
 *   svtyper/parsers.py:346-399   get_info_string / get_format_string / get_var_string / get_gt_string
 *   svtyper/classic.py:485,498   QUAL = running sum of SQ over the samples of a site (reset by a blank one)
 * and what svtyper_amd/vcf.py does in Python (which stays the general implementation and is the checker
 * of this one: tests/test_bulk_vcf.py).  `Variant` objects are only made for the lines this parser
 * hands back (anything the fast route does not express exactly: see svt_vcf_view.line_kind).
 *
 * Plain C ABI, host memory only; no GPU is needed for these calls.
 */
#ifndef SVTYPER_VCF_H
#define SVTYPER_VCF_H

#include <stddef.h>
#include <stdint.h>

#include "svtyper_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svt_vcf_parser svt_vcf_parser; /* opaque: INFO declarations, chromosome table, unpaired BND mates */
typedef struct svt_vcf_chunk svt_vcf_chunk;   /* opaque: the parsed lines of one block of text */

#define SVT_VCF_SUM_QUALS 1u       /* keep the incoming QUAL (-q); default: QUAL starts at 0 (classic.py:227-228) */
#define SVT_VCF_SKIP_HASH_LINES 2u /* '#' lines inside the body are skipped (singlesample.py:589-590) */

/* what became of one input line (svt_vcf_view.line_kind) */
#define SVT_VCF_LINE_SITE 0   /* a genotypable site: DEL / DUP / INV, or the SECOND mate of a BND pair            */
#define SVT_VCF_LINE_HELD 1   /* the first mate of a BND pair: written together with its partner                  */
#define SVT_VCF_LINE_PYTHON 2 /* not expressed here (no / unsupported SVTYPE, sample columns with FORMAT values,
                                 numbers Python's int()/float() may read differently, missing INFO keys, ...):
                                 the caller runs its general per-line code on it; never a BND line (see below)   */
#define SVT_VCF_LINE_SKIPPED 3 /* a '#' line under SVT_VCF_SKIP_HASH_LINES */

/* `info_ids` / `info_is_flag`: the header's INFO declarations in declaration order (INFO is printed in that
 * order and undeclared keys are dropped, parsers.py:346-355; a declared Flag prints bare).
 * `max_ci_dist`: classic.py's --max_ci_dist. */
int svt_vcf_parser_create(const char* const* info_ids, const uint8_t* info_is_flag, uint32_t n_info, double max_ci_dist,
                          uint32_t flags, svt_vcf_parser** out);
void svt_vcf_parser_free(svt_vcf_parser* p);

/* chromosome names seen so far (svt_vcf_view.chrom_a / chrom_b index this table; it only grows) */
uint32_t svt_vcf_parser_n_chroms(const svt_vcf_parser* p);
const char* svt_vcf_parser_chrom(const svt_vcf_parser* p, uint32_t index);

/* first mates of BND pairs still waiting for their partner, in the order they were held: their lines as they
 * came in (NUL-terminated, no newline), owned by the parser until the next svt_vcf_parse / free */
uint32_t svt_vcf_parser_n_pending(const svt_vcf_parser* p);
const char* svt_vcf_parser_pending_line(const svt_vcf_parser* p, uint32_t index);

/* Parse `text` (whole lines; the last one may lack its newline).  Parsing stops in front of a BND line the fast
 * route cannot express (`*consumed` < len): BND pairing is stateful, so from that line on the caller continues
 * with its general per-line code, seeded with svt_vcf_parser_pending_line.  0 or SVT_ERR_*. */
int svt_vcf_parse(svt_vcf_parser* p, const char* text, size_t len, svt_vcf_chunk** out, size_t* consumed);
void svt_vcf_chunk_free(svt_vcf_chunk* c);

typedef struct svt_vcf_view {
    uint64_t n_lines;            /* input lines consumed                                                        */
    const uint8_t* line_kind;    /* n_lines: SVT_VCF_LINE_*                                                     */
    const uint64_t* line_begin;  /* n_lines + 1: byte offsets of the lines in `text`                            */
    const uint32_t* line_site;   /* n_lines: site index of a SVT_VCF_LINE_SITE line                             */
    uint64_t n_sites;
    /* per site, breakpoints as Vcf.get_variant_breakpoints gives them (+1 shift of reverse sides applied) */
    const int32_t* chrom_a;
    const int32_t* chrom_b;
    const int64_t* pos_a;
    const int64_t* pos_b;
    const int64_t* ci;           /* n_sites * 4: A lo, A hi, B lo, B hi                                         */
    const int64_t* var_length;   /* DEL: END - POS; 0 otherwise                                                 */
    const uint8_t* svtype;       /* SVT_SVTYPE_DEL / _DUP / _INV / _BND (svtyper_hip.h)                           */
    const uint8_t* strands;      /* bit 0: side A reverse, bit 1: side B reverse                                */
    const double* qual_in;       /* QUAL the site starts from                                                   */
} svt_vcf_view;

int svt_vcf_chunk_view(const svt_vcf_chunk* c, svt_vcf_view* out);

#define SVT_VCF_QUAL_SSO 0     /* QUAL += SQ of a called sample (singlesample.py:544-546)                        */
#define SVT_VCF_QUAL_CLASSIC 1 /* the same, and a blank sample resets QUAL to 0 (classic.py:485,498); a site whose
                                  samples were ALL skipped for too many reads prints FORMAT "GT" and "./."
                                  columns (classic.py:282-284)                                                    */

/* The output lines of every site of the chunk: chrom, pos, id, ref, alt, QUAL %0.2f, filter, INFO in header order,
 * `format_string`, one column per sample (a BND site: both mates' lines, same QUAL and columns).
 * `results`: n_sites * n_samples records, site-major (sample k of site i at i * n_samples + k), SQ already
 * refined (svt_results_host_sq).  `fields` / `skipped_as_dots`: as svt_format_results.
 * `*text_out`: malloc'ed text; `*site_offset_out`: n_sites + 1 offsets into it.  Free with svt_format_free. */
int svt_vcf_emit(const svt_vcf_chunk* c, const svt_result* results, uint32_t n_samples, int qual_mode,
                 const uint8_t* fields, uint32_t n_fields, int skipped_as_dots, const char* format_string,
                 char** text_out, uint64_t** site_offset_out);

#ifdef __cplusplus
}
#endif

#endif /* SVTYPER_VCF_H */
