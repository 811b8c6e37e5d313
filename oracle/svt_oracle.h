/*
 * svt_oracle.h -- CPU restatement of the SVTyper likelihood hot path.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.c).  Shares the plain-data structs of
 * include/svtyper_hip.h so that the oracle and the HIP library consume the very
 * same packed evidence.
 */
#ifndef SVT_ORACLE_H
#define SVT_ORACLE_H
#include "../include/svtyper_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

double svt_oracle_prob_mapq(int mapq);                       /* utils.py:74-75        */
double svt_oracle_log_choose(int64_t n, int64_t k);          /* statistics.py:9-20    */
void svt_oracle_bayes_gt(int64_t ref, int64_t alt, int is_dup, double out[3]); /* statistics.py:23-37 */
int svt_oracle_p_concordant(const svt_library* lib, uint64_t n_total, int32_t ospan_length,
                            int has_var_length, int32_t var_length); /* parsers.py:861-882 */
void svt_oracle_tally(const svt_unit* unit, const svt_record* recs, uint64_t n_recs,
                      const svt_library* libs, const uint64_t* lib_totals,
                      const double pmapq[256], int sso, double out[SVT_N_TALLIES]);
void svt_oracle_genotype(const double tallies[SVT_N_TALLIES], int svtype, double split_weight,
                         double disc_weight, double gl[3], double* sq,
                         int32_t counts[SVT_N_COUNTS], int8_t* gt);
/* whole batch; n_threads <= 0 keeps the OpenMP default */
int svt_oracle_batch(const svt_evidence_batch* in, svt_result* out, unsigned flags, int n_threads);
int svt_oracle_threads(void);

#ifdef __cplusplus
}
#endif
#endif
