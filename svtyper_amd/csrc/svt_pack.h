// svt_pack.h -- interface of the host encoder of packed evidence (svt_pack.cpp; plain C++, no HIP)
#ifndef SVT_PACK_H
#define SVT_PACK_H

#include <string>

#include "svt_device_types.h"

namespace svt {

// where the three arrays that cross PCIe are allocated: the library's page-locked pool (svtyper_hip.hip), plain
// malloc in the sanitizer build of the host code
struct PackAlloc {
    void* (*get)(uint64_t bytes);
    void (*put)(void* p);
};

struct PackedArrays {
    uint32_t* off = nullptr;     // 3 * n_units + 1 slot offsets
    svt_unit* units = nullptr;   // n_units
    void* slots = nullptr;       // n_slots * 16 bytes
    uint64_t n_slots = 0;
    uint64_t n_records = 0;
    uint32_t common = 0;         // mapq_a | mapq_b << 8 of the one-half-word pair entries
};

// release the worker arenas svt_pack_evidence keeps between calls (svt_trim)
void pack_trim();

// the text svt_last_error() gives for the record-contract bits kErr* (also what the streaming pass reports)
std::string record_error_text(uint32_t err_bits);

// Optional: the encoder hands the packed evidence over in ranges of whole units as it goes, so that a consumer can put range k
// on the wire while range k + 1 is still being encoded (svt_genotype_packed_from_records).  The output arrays are then
// allocated up front -- the slots for `slots_cap` slots, an estimate; SVT_ERR_PACK_OVERFLOW when the batch needs more (nothing
// was handed over that the caller may keep: fall back to the plain call).
struct PackSink {
    uint64_t range_units = 0;     // units per range (rounded up to whole encoder chunks); 0 = one range
    uint64_t slots_cap = 0;       // slots the output array is allocated for
    void* ctx = nullptr;
    // units [u0, u1) are final in out->off (entries 3 u0 .. 3 u1), out->units and out->slots [s0, s1); called on the calling
    // thread of encode_packed, ranges in order; a non-zero return stops the encoder and becomes its return value
    int (*ready)(void* ctx, const struct PackedArrays* out, uint64_t u0, uint64_t u1, uint64_t s0, uint64_t s1) = nullptr;
    // called before a FAILING encoder gives the output arrays back to the allocator: ranges handed over earlier may still be
    // read by the consumer (DMA out of the page-locked arrays) -- it has to be through with them first
    void (*drain)(void* ctx) = nullptr;
};
constexpr int SVT_ERR_PACK_OVERFLOW = -1000;   // internal: never leaves the library

// Encode `in` (canonical records, one library) as packed evidence.  On success the caller owns out->off / units / slots
// (allocated with A.get); on failure nothing is left allocated.  SVT_ERR_UNSUPPORTED: the batch cannot be expressed
// in the packed format (keep the canonical records); SVT_ERR_INVALID: it breaks the contract of include/svtyper_hip.h.
int encode_packed(const svt_evidence_batch* in, const PackAlloc& A, PackedArrays* out, const PackSink* sink = nullptr);

}  // namespace svt

#endif  // SVT_PACK_H
