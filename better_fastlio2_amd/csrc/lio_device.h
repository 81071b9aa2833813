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
    for (;;) {
        const uint4 e = table[slot];
        const uint64_t ek = ((uint64_t)e.y << 32) | (uint64_t)e.x;
        if (ek == key) { start = e.z; count = e.w; return; }
        if (ek == KEY_EMPTY) return;
        slot = (slot + 1) & mask;
    }
}

}  // namespace lio
