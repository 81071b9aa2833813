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

// ---- the host's word to a pass that sits on the stream ahead of its pose, read by the FIRST KERNEL OF THE PASS itself ---------
// (lio_pass.hip gate_arm; the stand-alone gate_kernel of lio_knn.hip is the form for mail slots in host memory.)  One whole
// wavefront calls gate_poll: lanes 0..16 fetch one word of the slot each -- fine-grained device memory the CPU writes through
// the BAR, so a poll is a round trip to local memory -- until the sequence number the launch was made for is there with a
// matching check word.  Returns the lane's word; `go` (wave-uniform) is false when the host called the pass off or nobody
// answered for ten milliseconds.  The wait is over before most of the launch's workgroups have been dispatched: the host writes the
// word ~3 us after the previous pass's record, the dispatch of a dependent launch takes about as long.
__device__ __forceinline__ unsigned long long mail_word(unsigned long long w, int i) {   // word i of the mail as a wave-uniform value
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), i);
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
constexpr unsigned long long GATE_IN_KERNEL_WAIT_TICKS = 1000000ull;     // 100 MHz: 10 ms
__device__ __forceinline__ unsigned long long gate_poll(const GateMail* mail, unsigned long long expect, bool& go) {
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long t0 = wall_clock64();
    unsigned long long w = 0ull;
    bool ok = false;
    for (;;) {
        if (lane < 17) w = __hip_atomic_load(&mail->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (mail_word(w, 0) == expect) {
            unsigned long long all[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) all[i] = mail_word(w, i);
            if (gate_check(all) == mail_word(w, 16)) { ok = true; break; }
        }
        // 10 ms.  The host answers ~3 us after the previous pass's record; a host that has not answered after 10 ms is not
        // polling (pre-empted for scheduler quanta, stopped, in a debugger).  Every workgroup of the launch waits HERE -- a search of 100 k queries is a
        // resident grid that fills the GPU -- so the wait must be short: the launch stands down, the GPU is free for whoever else
        // uses it (tests/test_gpu_robustness.py: another process keeps stepping through a 3 s SIGSTOP), and the host, when it
        // comes back, finds an idle stream without a record and puts the pass on the stream directly (lio_measure_end,
        // "gate_reenqueued": ~25 us once).  The one-wavefront gate_kernel (lio_knn.hip) keeps its 2 s: it parks nothing.
        if (wall_clock64() - t0 > GATE_IN_KERNEL_WAIT_TICKS) break;
        __builtin_amdgcn_s_sleep(2);
    }
    go = ok && mail_word(w, 1) == GATE_GO;
    return w;
}
__device__ __forceinline__ Pose14 gate_pose(unsigned long long w) {                       // words 2..15, wave-uniform
    Pose14 p;
#pragma unroll
    for (int i = 0; i < 14; ++i) p.v[i] = __longlong_as_double((long long)mail_word(w, 2 + i));
    return p;
}

}  // namespace lio
