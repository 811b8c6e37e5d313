// svt_host_tiling.h -- host-side sorting of units into 64-lane tiles
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_HOST_TILING_H
#define SVT_HOST_TILING_H

#include "svt_device_types.h"
#include "svt_error.h"
#include "svt_host_cpus.h"
#include "svt_prepare_kernels.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// host-side tiling
// ------------------------------------------------------------------------------------------
struct Tiling {
    std::vector<TileDesc> tiles;       // storage order
    std::vector<uint32_t> tile_lib_lo, tile_lib_hi;  // library range referenced by each tile
    std::vector<LaneHdr> hdr;
    std::vector<uint64_t> lane_src;
    std::vector<uint32_t> lane_nrec;
    uint64_t slots = 0;                // 16-byte row slots of all tiles
};

inline unsigned host_threads()
{
    return std::max(1u, std::min(usable_cpus(), 16u));
}

// run fn(i) for i in [0, n) on up to host_threads() threads
template <typename Fn>
inline void parallel_for(uint64_t n, Fn&& fn)
{
    const unsigned nt = (unsigned)std::min<uint64_t>(host_threads(), n);
    if (nt == 0) return;
    run_threads(nt, [&](unsigned t) { for (uint64_t i = t; i < n; i += nt) fn(i); });
}

// rows of 16-byte slots unit `u` needs in stream `k`
inline uint32_t stream_rows_of(const ScanOut& sc, uint32_t nrec, int layout, int k)
{
    if (layout == kLayoutShort && k == kPairs) return (sc.n_short + kHalfwordsPerRow - 1) / kHalfwordsPerRow;
    if (layout != kLayoutDense) return (sc.n[k] + kEntriesPerRow[k] - 1) / kEntriesPerRow[k];
    return k == 0 ? nrec : 0u;
}

// Bucket units by their first library (one sample's units end up together, whatever the input
// order: site-major batches interleave the samples), sort by stream length inside chunks of at
// most 16384 units of one bucket, cut into 64-unit tiles.  A chunk never crosses a bucket and its
// tile count is padded to whole workgroups, so the library window a workgroup stages in LDS is the
// one of a single sample.  Chunks are independent and are processed by several host threads.
inline void build_tiling(const svt_evidence_batch* in, const std::vector<uint32_t>& nrec,
                  const std::vector<ScanOut>& scan, int layout, Tiling& G)
{
    const uint64_t n = in->n_units;
    // stable counting sort by first library
    std::vector<uint32_t> by_lib;
    uint64_t bucket_end[256];
    int n_buckets = 1;
    bucket_end[0] = n;
    if (in->n_libs > 1) {
        uint64_t start[257] = {0};
        for (uint64_t u = 0; u < n; ++u) ++start[(scan[u].libs & 0xffu) + 1];
        for (int l = 0; l < 256; ++l) start[l + 1] += start[l];
        for (int l = 0; l < 256; ++l) bucket_end[l] = start[l + 1];
        n_buckets = 256;
        by_lib.resize(n);
        for (uint64_t u = 0; u < n; ++u) by_lib[start[scan[u].libs & 0xffu]++] = (uint32_t)u;
    }
    auto unit_at = [&](uint64_t i) -> uint64_t { return by_lib.empty() ? i : by_lib[i]; };
    struct Chunk { uint64_t begin; uint32_t len; uint64_t tile_base; };
    std::vector<Chunk> chunks;
    uint64_t n_tiles = 0;
    for (int bkt = 0; bkt < n_buckets; ++bkt) {
        const uint64_t lo = bkt ? bucket_end[bkt - 1] : 0, hi = bucket_end[bkt];
        for (uint64_t c0 = lo; c0 < hi; c0 += kChunkUnits) {
            const uint32_t len = (uint32_t)std::min<uint64_t>(kChunkUnits, hi - c0);
            chunks.push_back(Chunk{c0, len, n_tiles});
            const uint64_t t = (len + kWave - 1) / kWave;
            n_tiles += (t + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;   // whole workgroups
        }
    }
    // the last chunk of the batch needs no padding tiles (the dispatch list pads the last group itself)
    if (!chunks.empty()) n_tiles = chunks.back().tile_base + (chunks.back().len + kWave - 1) / kWave;
    G.tiles.assign(n_tiles, TileDesc{});
    G.tile_lib_lo.assign(n_tiles, 0xffffffffu);   // 0xffffffff: the tile references no library
    G.tile_lib_hi.assign(n_tiles, 0);
    G.hdr.assign(n_tiles * kWave, LaneHdr{0, 0, kPadUnit, 0});
    G.lane_src.assign(n_tiles * kWave, 0);
    G.lane_nrec.assign(n_tiles * kWave, 0);
    for (uint64_t ti = 0; ti < n_tiles; ++ti) G.tiles[ti].lane_base = (uint32_t)(ti * kWave);   // padding tiles too
    parallel_for(chunks.size(), [&](uint64_t c) {
        const uint64_t c0 = chunks[c].begin;
        const uint32_t cn = chunks[c].len;
        std::vector<uint32_t> order(cn);
        for (uint32_t i = 0; i < cn; ++i) order[i] = i;
        // longest first
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint64_t ux = unit_at(c0 + x), uy = unit_at(c0 + y);
            const uint64_t kx = ((uint64_t)stream_rows_of(scan[ux], nrec[ux], layout, 0) << 32) |
                                stream_rows_of(scan[ux], nrec[ux], layout, 1);
            const uint64_t ky = ((uint64_t)stream_rows_of(scan[uy], nrec[uy], layout, 0) << 32) |
                                stream_rows_of(scan[uy], nrec[uy], layout, 1);
            return kx > ky;
        });
        for (uint32_t t0 = 0; t0 < cn; t0 += kWave) {
            const uint64_t ti = chunks[c].tile_base + t0 / kWave;
            TileDesc td{};
            td.lane_base = (uint32_t)(ti * kWave);
            uint32_t lib_lo = 0xffffffffu, lib_hi = 0;
            for (uint32_t l = 0; l < (uint32_t)kWave; ++l) {
                LaneHdr h{};
                h.unit = kPadUnit;
                uint64_t src = 0;
                uint32_t f = 0;
                if (t0 + l < cn) {
                    const uint64_t u = unit_at(c0 + order[t0 + l]);
                    const svt_unit& U = in->units[u];
                    h.var_length = U.var_length;
                    h.pos_delta = U.pos_delta;
                    h.unit = (uint32_t)u;
                    h.packed = (uint32_t)U.svtype | ((uint32_t)U.flags << 8) | ((scan[u].libs & 0xffu) << 16);
                    src = in->rec_offset[u];
                    f = nrec[u];
                    for (int k = 0; k < kStreams; ++k)
                        td.rows[k] = std::max(td.rows[k], stream_rows_of(scan[u], f, layout, k));
                    if (f) {
                        lib_lo = std::min(lib_lo, scan[u].libs & 0xffu);
                        lib_hi = std::max(lib_hi, (scan[u].libs >> 8) & 0xffu);
                    }
                }
                G.hdr[td.lane_base + l] = h;
                G.lane_src[td.lane_base + l] = src;
                G.lane_nrec[td.lane_base + l] = f;
            }
            G.tiles[ti] = td;
            G.tile_lib_lo[ti] = lib_lo;
            G.tile_lib_hi[ti] = lib_hi;
        }
    });
    // slot offsets: the streams of a tile follow each other
    for (TileDesc& td : G.tiles) {
        td.base = G.slots;
        for (int k = 0; k < kStreams; ++k) G.slots += (uint64_t)td.rows[k] * kWave;
    }
}


}  // namespace svt

#endif  // SVT_HOST_TILING_H
