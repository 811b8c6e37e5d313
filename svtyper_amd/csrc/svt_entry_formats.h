// svt_entry_formats.h -- the entries of packed evidence (include/svtyper_hip.h: svt_packed_evidence)
// Internal header of libsvtyper_hip.so.
//
// svt_pack_evidence (host, svt_pack.cpp) writes them, svt_packed_kernel.h / svt_unit_math.h consume them,
// oracle/py_packed.py decodes them independently on the CPU.
#ifndef SVT_ENTRY_FORMATS_H
#define SVT_ENTRY_FORMATS_H

#include "svt_device_types.h"

namespace svt {

// A unit's evidence becomes three sparse streams of entries packed into 16-byte slots, each in record order.
// The five tallies are independent sums, so evidence for different tallies can live in different streams
// without changing any of them.
//
//   pair entries (alt_span, ref_span), every library with n_bins <= 2047, in half-words:
//     an entry with the batch's most common MAPQ pair (60, 60 for bwa) is ONE half-word f3 | code << 3; any other
//     entry is a 4-byte-aligned pair of half-words, f3 | code << 3 | 0x8000 then mapq_a | mapq_b << 8.  A zero
//     half-word is a no-op (f3 = 0: both weights 0) and pads a wide entry to its alignment.
//     Several libraries (ABI 13+: svt_packed_evidence.n_libs > 1): a unit's pair stream starts in the context of library 0
//     of the batch; the half-word (l + 1) << 3 -- no straddle bit, not wide, not zero: nothing an entry can be -- is a
//     LIBRARY SWITCH: the entries behind it were coded against the tables of library l, until the next switch or the end
//     of the unit.  Entries stay in record order whatever their library (the sums are order-dependent), the gate of
//     classic.py:339,383 is the entry's own library's.  A batch of one library never holds a switch.
//     f3   = alt | refA << 1 | refB << 2 straddle bits
//     code = ospan_len translated into the index space of the library's histogram tables: with
//            r = ospan_len - key_min, the kernel needs thr[r] (parsers.py:870-872) and, for a
//            DEL, hist[r - var_length] (parsers.py:874-878), each replaced by the sentinel bin
//            n_bins when out of range.  With off2 = min(var_length, n_bins):
//                var_length <  n_bins:  code = r            for 0 <= r < var_length + n_bins
//                var_length >= n_bins:  code = r            for 0 <= r < n_bins
//                                       code = n_bins + (r - var_length)   for 0 <= r - var_length < n_bins
//                anything else / not a DEL window:  code = 2 * n_bins
//            so that i1 = min(code, n_bins) and i2 = min(code - off2, n_bins) (unsigned) are exactly the
//            two table indices.  Only the addressing is precomputed; the look-ups, the p_concordant
//            decision and every sum stay in the kernel.
//   reference-read entries (ref_seq)         2 bytes: mapq0, mapq1 -- seven per slot (bytes 0..13); byte 14
//                                            of the slot holds their seven first_of_fragment bits
//   candidate entries (alt_seq / alt_clip)   the same, plus their seven is_clip bits in byte 15
//     the two gated MAPQs of the reference reads (rs_a, rs_b), of the split candidate (seq_l, seq_r)
//     or of the clip candidate (clip_l, clip_r); first_of_fragment marks the first kept entry for its
//     tally in a read-fragment (sso association: fragment-local sums).
//
// Entries that can only add +0.0 to a sum are not stored: pair entries without a straddle bit, with a
// zero MAPQ on either read, or of a DEL smaller than 2 sd of the library (classic.py:339,383);
// weight entries whose two gated MAPQs are 0.  x + 0.0 == x bit-for-bit for these non-negative sums.
// Order inside a stream is record order, so every sum sees the reference's additions in the reference's order.

constexpr uint32_t kWideEntry = 0x8000u;      // pair stream: the next half-word holds this entry's MAPQs
constexpr uint32_t kDefaultCommonMapq = 60u | (60u << 8);
constexpr uint32_t kVoteRecords = 1u << 14;   // records the (host-side) MAPQ vote looks at

}  // namespace svt

#endif  // SVT_ENTRY_FORMATS_H
