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

// Encode `in` (canonical records, one library) as packed evidence.  On success the caller owns out->off / units / slots
// (allocated with A.get); on failure nothing is left allocated.  SVT_ERR_UNSUPPORTED: the batch cannot be expressed
// in the packed format (keep the canonical records); SVT_ERR_INVALID: it breaks the contract of include/svtyper_hip.h.
int encode_packed(const svt_evidence_batch* in, const PackAlloc& A, PackedArrays* out);

}  // namespace svt

#endif  // SVT_PACK_H
