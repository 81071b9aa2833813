// lio_rebuild.h -- the full index rebuild as a sort of CELL RUNS (lio_rebuild.hip), interface + the key frame both sides share.
//
// Build / reconstruct (ikd_Tree.cpp:353-364, 1394-1405) used to sort every point by the 63-bit Morton code of its grid cell
// (seven radix passes over 5 M pairs: 0.6 of the 1.0 ms a rebuild took at 5 M points).  But the compacted live array is a
// sequence of BUCKETS -- every cell's live points are contiguous (lio_mapadd.hip re-writes whole buckets) -- so a rebuild
// only has to put the runs of equal cells into Morton order: 1.6 M runs instead of 5 M points, and with keys of ~30 bits
// instead of 63 (below).  The point array is then copied run by run; within a cell the order is the array's (the pending
// points of a batch come last, in batch order): exactly what the stable point sort produced, bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

// Compressed sort key of a grid cell.  Cell coordinates are biased by 2^20 (lio_common.h), a map around the origin
// straddles that boundary on every axis and its Morton codes differ from bit 62 down.  But the map touches only a few
// cells of a coarse level `a` (grid coarsened by 4^a): their Morton codes (`parent`, ascending) are listed, a cell's key is
//      [rank of its level-a parent in that list | the low 6a bits of its own Morton code]
// -- the same order as the absolute Morton code, in 6a + log2(parents) bits.
constexpr int CELL_FRAME_PARENTS = 64;
struct CellKeyFrame {
    int      a;                             // parent level, 1..10
    int      n_parents;
    int      bits;                          // key bits: 6a + bits of (n_parents - 1)
    int      pad;
    uint64_t parent[CELL_FRAME_PARENTS];    // Morton codes >> 6a of the level-a cells of the map's bounding box, ascending
};
// the frame of the cells lo..hi (inclusive, biased coordinates): the finest level whose cells of the box number <= CELL_FRAME_PARENTS
inline CellKeyFrame cell_key_frame(const uint32_t lo[3], const uint32_t hi[3]) {
    CellKeyFrame f{};
    for (int a = 1; a <= 10; ++a) {
        uint64_t np = 1;
        uint32_t p0[3], cnt[3];
        for (int i = 0; i < 3; ++i) { p0[i] = lo[i] >> (2 * a); cnt[i] = (hi[i] >> (2 * a)) - p0[i] + 1u; np *= cnt[i]; }
        if (np > (uint64_t)CELL_FRAME_PARENTS && a < 10) continue;
        if (np > (uint64_t)CELL_FRAME_PARENTS) np = CELL_FRAME_PARENTS;      // (cannot happen: 21-bit coordinates leave <= 2 parents per axis at a = 10)
        f.a = a; f.n_parents = 0;
        for (uint32_t x = 0; x < cnt[0]; ++x)
            for (uint32_t y = 0; y < cnt[1]; ++y)
                for (uint32_t z = 0; z < cnt[2]; ++z)
                    if (f.n_parents < CELL_FRAME_PARENTS) f.parent[f.n_parents++] = morton3(p0[0] + x, p0[1] + y, p0[2] + z);
        for (int i = 1; i < f.n_parents; ++i) {                               // insertion sort: <= 64 codes
            const uint64_t v = f.parent[i];
            int j = i - 1;
            while (j >= 0 && f.parent[j] > v) { f.parent[j + 1] = f.parent[j]; --j; }
            f.parent[j + 1] = v;
        }
        for (int i = f.n_parents; i < CELL_FRAME_PARENTS; ++i) f.parent[i] = ~0ull;   // (above every code: cell_key_compress)
        int rb = 0;
        while ((1 << rb) < f.n_parents) ++rb;
        f.bits = 6 * a + rb;
        break;
    }
    return f;
}
// Morton code -> compressed key; *ok = false when the cell's parent is not one of the frame's.  (Fixed trip counts over the whole
// list -- the unused entries hold ~0, above every code -- so that a kernel fetches the list from its arguments once, up front; and a
// loop of selects, not f.parent[rank]: a per-lane index into a by-value kernel argument would send the struct to scratch.)
LIO_HD uint64_t cell_key_compress(const CellKeyFrame& f, uint64_t mk, bool* ok) {
    const uint64_t pm = mk >> (6 * f.a);
    uint32_t r = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < CELL_FRAME_PARENTS; ++i) { r += f.parent[i] < pm ? 1u : 0u; found = found || f.parent[i] == pm; }
    *ok = found;
    return ((uint64_t)r << (6 * f.a)) | (mk & ((1ull << (6 * f.a)) - 1ull));
}
LIO_HD uint64_t cell_key_expand(const CellKeyFrame& f, uint64_t ck) {
    const uint32_t r = (uint32_t)(ck >> (6 * f.a));
    uint64_t pm = 0;
#pragma unroll
    for (int i = 0; i < CELL_FRAME_PARENTS; ++i) pm = r == (uint32_t)i ? f.parent[i] : pm;
    return (pm << (6 * f.a)) | (ck & ((1ull << (6 * f.a)) - 1ull));
}

constexpr uint32_t RUN_TILE = 2048;          // points per workgroup of the run kernels
constexpr uint32_t PLACE_WORKGROUPS = 1024;  // workgroups of runs_place_kernel (each leaves TREE_MAX_LEVELS partial counts)

// runs of equal grid cells in pts[0..n): tile_cnt[n_tiles + 1] run heads per tile (the last entry: 0, the scan's total),
// tile_box[n_tiles][6] the cells' bounding box per tile {min x, y, z, max x, y, z}
hipError_t launch_cell_runs_count(const float4* pts, uint32_t n, Grid grid, uint32_t* tile_cnt, uint32_t* tile_box, hipStream_t s);
// run_key[r] = Morton code of the run's cell, run_start[r] = its first point; *n_runs = tile_off[n_tiles]
hipError_t launch_cell_runs_emit(const float4* pts, uint32_t n, Grid grid, const uint32_t* tile_off, uint64_t* run_key, uint32_t* run_start,
                                 uint32_t* n_runs, hipStream_t s);
// dst_mapped[0] = *n_runs, [1..6] = the box of all tiles, then `seq` at dst_mapped[32] (system scope): the host polls (wait_published)
hipError_t launch_cell_box_publish(const uint32_t* tile_box, uint32_t n_tiles, const uint32_t* n_runs, uint32_t* dst_mapped, uint32_t seq, hipStream_t s);
// run_key[r] -> compressed key, in place; run_val[r] = first point << 32 | points of run r; *err |= 2 when a cell lies outside the frame
hipError_t launch_compress_run_keys(uint64_t* run_key, const uint32_t* run_start, uint32_t n_runs, uint32_t n, const CellKeyFrame& f, uint64_t* run_val,
                                    uint32_t* err, hipStream_t s);
// stable sort of the runs by compressed key (the values ride along: nothing is gathered through an index afterwards)
hipError_t sort_runs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in, uint64_t* vals_out, uint32_t n_runs,
                     int bits, hipStream_t s);
// first destination slot of every sorted run (exclusive scan of the run lengths in sorted order)
hipError_t scan_run_lengths(void* temp, size_t temp_bytes, const uint64_t* sorted_val, uint32_t n_runs, uint32_t* dst_start, hipStream_t s);
// the points of sorted run j to dst[dst_start[j]..); abs_key[j] = Morton code of its cell; level_parts[PLACE_WORKGROUPS][TREE_MAX_LEVELS]:
// per workgroup, the sorted runs that start a new cell of level L
hipError_t launch_runs_place(const float4* src, float4* dst, const uint64_t* sorted_key, const uint64_t* sorted_val, const uint32_t* dst_start,
                             uint32_t n_runs, const CellKeyFrame& f, uint64_t* abs_key, uint32_t* level_parts, hipStream_t s);
// cells = runs of equal abs_key: cell_keys, cell_counts (points), *n_cells
hipError_t reduce_runs_to_cells(void* temp, size_t temp_bytes, const uint64_t* abs_key, const uint64_t* sorted_val, uint32_t n_runs, uint64_t* cell_keys,
                                uint32_t* cell_counts, uint32_t* n_cells, hipStream_t s);
// dst_mapped[L] = cells of level L (sum of the partials), L < TREE_MAX_LEVELS, [TREE_MAX_LEVELS] = *err; then `seq` as above
hipError_t launch_level_counts_publish(const uint32_t* level_parts, const uint32_t* err, uint32_t* dst_mapped, uint32_t seq, hipStream_t s);

}  // namespace lio
