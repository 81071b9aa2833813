// lio_common.h -- data layout shared by the host side of liblio_hip.so and its gfx950 kernels.
//
// Map index ("hashed voxel grid over a bucketed point array"; replaces the pointer-chased
// KD_TREE_NODE of include/ikd-Tree/ikd_Tree.h:64-86):
//   pts    float4[slots] x, y, z, w   (w != 0 marks a dead slot, cf. point_deleted ikd_Tree.h:70; kinds below)
//                        a (re)build sorts by the Morton code of the point's neighbour-grid cell: every occupied
//                        cell is one contiguous BUCKET; Add_Points later re-writes touched buckets at the tail
//   ltable HashEntry[..] one open-addressing table per COARSE level (grid / 4, / 16, ...): packed cell key ->
//                        existence; level 1 -> the number of the cell's child block
//   child  uint2[B][64]  {start, count} of the 64 grid cells of a level-1 cell at a fixed position: together with
//                        level 1 of ltable this is the cell directory every look-up goes through (lio_device.h)
// The cell edge `cell` is a power of two, so floor(p * inv_cell) and the cell -> world box
// conversion are exact in fp32; this is what makes the "proven inside the searched block" test
// of the k-NN kernel sound.
#pragma once
#include <stdint.h>

#ifndef __HIPCC__            // host-only build of the shared math (tests/csrc/host_harness.cpp)
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#endif
#endif

#define LIO_HD __host__ __device__ __forceinline__

namespace lio {

constexpr int K = 5;                          // NUM_MATCH_POINTS, include/common_lib.h:149
constexpr int CELL_BITS = 21;                 // cell coordinate bits per axis
constexpr int CELL_OFF = 1 << 20;             // coordinate bias: cells are stored non-negative
constexpr uint32_t CELL_MAX = (1u << CELL_BITS) - 1u;
constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr uint64_t KEY64_MAX = ~0ull;         // "no candidate" in the (sqdist bits, index) ordering
constexpr uint32_t INF_BITS = 0x7f800000u;    // +inf as fp32 bits
// tomb-stones live in the w lane of a map point: 0 = live, TOMB_BOX = removed by Delete_Point_Boxes,
// TOMB_DOWNSAMPLE = replaced inside its down-sample voxel (point_downsample_deleted, ikd_Tree.cpp:662-700),
// TOMB_MOVED = not a deletion at all: the slot of a bucket that has been re-written elsewhere
constexpr float TOMB_BOX = 1.0f;
constexpr float TOMB_DOWNSAMPLE = 2.0f;
constexpr float TOMB_BOX_REPORTED = 4.0f;    // box-deleted and already handed out by acquire_removed_points
constexpr float TOMB_MOVED = 3.0f;           // the live copy is in the bucket re-written at the tail (lio_mapadd.hip)

struct HashEntry {                            // 16 bytes: one global_load_dwordx4
    uint32_t key_lo, key_hi;                  // packed cell key (x << 42 | y << 21 | z), KEY_EMPTY = free
    uint32_t start, count;                    // range of the cell in the sorted point array
};

struct Grid {
    float    cell;                            // power of two
    float    inv_cell;
    uint32_t n_points;                        // stored points (live + tomb-stoned)
};

// Levels of the hashed 64-ary tree over the grid cells (level L = grid coarsened by 4^L).  Levels 1 .. top-1 have a
// hash table each (level 1 doubles as the hashed half of the cell directory); the cells of level `top_level` are few
// (<= TREE_MAX_ROOTS) and are simply listed: the walk starts from all of them, so the depth is set by the extent of
// the map, not by where the map happens to sit relative to the power-of-two boundaries of the grid.
constexpr int TREE_MAX_LEVELS = 12;
constexpr int TREE_MAX_ROOTS = 32;
struct TreeLevels {
    uint32_t off[TREE_MAX_LEVELS];   // first slot of level L in ltable (1 <= L < top_level)
    uint32_t mask[TREE_MAX_LEVELS];  // slots - 1 of level L
    int      top_level;              // level of the root cells
    int      n_roots;
    uint32_t root_x[TREE_MAX_ROOTS], root_y[TREE_MAX_ROOTS], root_z[TREE_MAX_ROOTS];   // their coordinates at top_level
};

// pos3, rot xyzw, offR xyzw, offT3 -- passed to kernels by value (112 bytes of kernarg)
struct Pose14 { double v[14]; };

// A pass enqueued BEFORE its pose is known (the iterated update, lio_pass.hip): a one-wavefront gate kernel in front of the pass's
// kernels waits for the host's word in GateMail (host-mapped, coherent) and leaves the pose -- or the order to skip the pass -- in
// GateState (device memory), which the kernels behind it read instead of their by-value pose.
struct GateState { uint32_t skip; uint32_t pad[3]; Pose14 pose; };
struct GateMail { unsigned long long w[17]; };           // [0] seq, [1] cmd (1 = go, 2 = skip), [2..15] the pose's bits, [16] check word
constexpr unsigned long long GATE_GO = 1ull, GATE_SKIP = 2ull;
// the check word: the gate reads all 17 words in ONE round trip over PCIe, in no particular order -- a read that mixes the
// words of two messages fails the check and is repeated
LIO_HD unsigned long long gate_check(const unsigned long long w[16]) {
    unsigned long long c = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 16; ++i) c = (c ^ w[i]) * 0xff51afd7ed558ccdull + (unsigned long long)i;
    return c ^ (c >> 29);
}

// per-block partial of the measurement reduction: 78 + 12 unique sums, residual sum, count
constexpr int RED_NE_EXT = 78 + 12;           // extrinsic_est_en: 12 columns
constexpr int RED_NE_NOEXT = 21 + 6;          // otherwise columns 6..11 are zero (laserMapping.cpp:1997)
constexpr int RED_STRIDE = 96;                // doubles per block partial (>= 90 + 2)
constexpr int RED_GROUP = 16;                 // workgroups per first-level arrival group

struct DeviceOut {                            // written by the final reduction kernel
    double   HTH[144];
    double   HTh[12];
    double   res_sum;
    unsigned long long n_candidates;
    int      m;
    int      n_unresolved;
    unsigned long long seq;                   // written LAST (system-scope release): the host polls it
    unsigned long long dbg[8];                // wall_clock64 stamps of the last workgroup (100 MHz; diagnostics only)
    float    probe[8];                        // feats_down_body[0] and [nq - 1] as the measurement kernel read them: the host's check of a scan
                                              // a kernel pulled out of the caller's page-locked records (a stale registration reads other pages)
};

// ---- cell / key arithmetic (host + device) ----------------------------------------------------
LIO_HD int cell_coord(float v, float inv_cell) {
    // exact: inv_cell is a power of two.  Bias into [0, 2^21) and clamp (|p| beyond 2^20 cells is
    // outside the supported range: 524 km at a 0.5 m cell).
    float f = __builtin_floorf(v * inv_cell);
    f = f < -1048576.0f ? -1048576.0f : (f > 1048575.0f ? 1048575.0f : f);
    if (!(f == f)) f = 0.0f;
    return (int)f + CELL_OFF;
}
LIO_HD uint64_t pack_key(uint32_t cx, uint32_t cy, uint32_t cz) {
    return ((uint64_t)cx << 42) | ((uint64_t)cy << 21) | (uint64_t)cz;
}
LIO_HD uint64_t spread21(uint32_t v) {       // bit i -> bit 3i
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
LIO_HD uint32_t compact21(uint64_t x) {      // inverse of spread21
    x &= 0x1249249249249249ull;
    x = (x | (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x | (x >> 4)) & 0x100f00f00f00f00full;
    x = (x | (x >> 8)) & 0x001f0000ff0000ffull;
    x = (x | (x >> 16)) & 0x001f00000000ffffull;
    x = (x | (x >> 32)) & 0x1fffffull;
    return (uint32_t)x;
}
LIO_HD uint64_t morton3(uint32_t cx, uint32_t cy, uint32_t cz) {
    return (spread21(cx) << 2) | (spread21(cy) << 1) | spread21(cz);
}
// Slot hash of a cell: three 24-bit multiplies (full-rate v_mul_u32_u24 on gfx950; the biased
// coordinates have 21 bits) and an xor-shift fold -- 64-bit multiplies are quarter rate and showed
// up in the k-NN kernel's VALU time.
LIO_HD uint32_t hash_cell(uint32_t cx, uint32_t cy, uint32_t cz) {
    uint32_t h = ((cx & 0xffffffu) * 0x9e3779u) ^ ((cy & 0xffffffu) * 0x85ebcbu) ^ ((cz & 0xffffffu) * 0xc2b2afu);
    h ^= h >> 15;
    h ^= h << 7;
    h ^= h >> 11;
    return h;
}
LIO_HD uint32_t hash_key(uint64_t packed) {  // same hash from the packed key (x << 42 | y << 21 | z)
    return hash_cell((uint32_t)(packed >> 42) & 0x1fffffu, (uint32_t)(packed >> 21) & 0x1fffffu, (uint32_t)packed & 0x1fffffu);
}

// squared distance exactly as KD_TREE::calc_dist (include/ikd-Tree/ikd_Tree.cpp:1374-1378):
// fp32, (dx*dx + dy*dy) + dz*dz, no contraction (translation units are built -ffp-contract=off).
LIO_HD float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

}  // namespace lio
