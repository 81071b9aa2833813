// lio_device.h -- device-only helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

// open-addressing probe of the cell table: packed cell key -> {start, count}; count = 0 when absent
__device__ __forceinline__ void table_find(const HashEntry* table_, uint32_t mask, uint64_t key, uint32_t& start,
                                           uint32_t& count) {
    const uint4* table = reinterpret_cast<const uint4*>(table_);
    uint32_t slot = hash_key(key) & mask;
    start = 0; count = 0;
    // tables are kept at most half full (apply_pending), so a probe sequence ends at an empty slot long before it
    // wraps; the bound only keeps a corrupted table from hanging the GPU
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        const uint4 e = table[slot];
        const uint64_t ek = ((uint64_t)e.y << 32) | (uint64_t)e.x;
        if (ek == key) { start = e.z; count = e.w; return; }
        if (ek == KEY_EMPTY) return;
        slot = (slot + 1) & mask;
    }
}

// The cell directory: level-1 hash entry (a cell of 4x4x4 grid cells) -> number of its child block; the block
// holds {start, count} of the 64 grid cells at position (x&3)<<4 | (y&3)<<2 | (z&3).  count = 0 when the cell
// does not exist; *index (optional) = position of the record in `child`, 0xffffffff without a level-1 cell.
__device__ __forceinline__ uint32_t child_pos(uint32_t cx, uint32_t cy, uint32_t cz) {
    return ((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u);
}
__device__ __forceinline__ void cell_find(const HashEntry* l1table, uint32_t l1mask, const uint2* child, uint32_t cx,
                                          uint32_t cy, uint32_t cz, uint32_t& start, uint32_t& count,
                                          uint32_t* index = nullptr) {
    uint32_t blk, one;
    table_find(l1table, l1mask, pack_key(cx >> 2, cy >> 2, cz >> 2), blk, one);
    start = 0; count = 0;
    if (index != nullptr) *index = 0xffffffffu;
    if (one == 0u || blk == 0xffffffffu) return;
    const uint32_t ci = blk * 64u + child_pos(cx, cy, cz);
    const uint2 e = child[ci];
    start = e.x; count = e.y;
    if (index != nullptr) *index = ci;
}

}  // namespace lio
