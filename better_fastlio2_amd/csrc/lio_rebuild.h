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
// cells of a coarse level `a` (grid coarsened by 4^a): a box of n[0] x n[1] x n[2] <= 64 of them from p0 on.  Their Morton
// codes (`parent`, ascending) are listed, a cell's key is
//      [rank of its level-a parent in that list | the low 6a bits of its own Morton code]
// -- the same order as the absolute Morton code, in 6a + log2(parents) bits.
constexpr int CELL_FRAME_PARENTS = 64;
struct CellKeyFrame {
    int      a;                             // parent level, 1..10
    int      n_parents;
    int      bits;                          // key bits: 6a + bits of (n_parents - 1)
    uint32_t p0[3], n[3];                   // the parents' box: coordinates >> 2a from p0[i], n[i] of them
    uint32_t pad;
    uint64_t rank_of[CELL_FRAME_PARENTS / 8];   // byte (dx * n[1] + dy) * n[2] + dz: the rank of that parent
    uint64_t parent[CELL_FRAME_PARENTS];    // Morton codes >> 6a of the parents, ascending (unused entries: ~0)
};
// the frame of the cells lo..hi (inclusive, biased coordinates): the finest level whose cells of the box number <= CELL_FRAME_PARENTS
inline CellKeyFrame cell_key_frame(const uint32_t lo[3], const uint32_t hi[3]) {
    CellKeyFrame f{};
    for (int a = 1; a <= 10; ++a) {
        uint64_t np = 1;
        for (int i = 0; i < 3; ++i) { f.p0[i] = lo[i] >> (2 * a); f.n[i] = (hi[i] >> (2 * a)) - f.p0[i] + 1u; np *= f.n[i]; }
        if (np > (uint64_t)CELL_FRAME_PARENTS && a < 10) continue;
        // (a = 10: 21-bit coordinates leave at most 2 parents per axis)
        f.a = a; f.n_parents = (int)np;
        uint64_t code[CELL_FRAME_PARENTS];
        int at[CELL_FRAME_PARENTS];
        for (uint32_t x = 0; x < f.n[0]; ++x)
            for (uint32_t y = 0; y < f.n[1]; ++y)
                for (uint32_t z = 0; z < f.n[2]; ++z) {
                    const int idx = (int)((x * f.n[1] + y) * f.n[2] + z);
                    code[idx] = morton3(f.p0[0] + x, f.p0[1] + y, f.p0[2] + z); at[idx] = idx;
                }
        for (int i = 1; i < f.n_parents; ++i) {                               // insertion sort: <= 64 codes
            const uint64_t v = code[i]; const int w = at[i];
            int j = i - 1;
            while (j >= 0 && code[j] > v) { code[j + 1] = code[j]; at[j + 1] = at[j]; --j; }
            code[j + 1] = v; at[j + 1] = w;
        }
        for (int r = 0; r < CELL_FRAME_PARENTS; ++r) f.parent[r] = r < f.n_parents ? code[r] : ~0ull;
        for (int r = 0; r < f.n_parents; ++r) f.rank_of[at[r] >> 3] |= (uint64_t)r << (8 * (at[r] & 7));
        int rb = 0;
        while ((1 << rb) < f.n_parents) ++rb;
        f.bits = 6 * a + rb;
        break;
    }
    return f;
}
// cell coordinates -> compressed key; *ok = false when the cell's parent is not one of the frame's.  (Selects over the words of the
// rank table, not rank_of[i]: a per-lane index into a by-value kernel argument would send the struct to scratch.)
LIO_HD uint64_t cell_key_compress(const CellKeyFrame& f, uint32_t cx, uint32_t cy, uint32_t cz, bool* ok) {
    const uint32_t dx = (cx >> (2 * f.a)) - f.p0[0], dy = (cy >> (2 * f.a)) - f.p0[1], dz = (cz >> (2 * f.a)) - f.p0[2];
    *ok = dx < f.n[0] && dy < f.n[1] && dz < f.n[2];
    const uint32_t idx = *ok ? (dx * f.n[1] + dy) * f.n[2] + dz : 0u;
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < CELL_FRAME_PARENTS / 8; ++k) w = (idx >> 3) == (uint32_t)k ? f.rank_of[k] : w;
    const uint64_t r = (w >> (8u * (idx & 7u))) & 255ull;
    const uint32_t m = (1u << (2 * f.a)) - 1u;
    return (r << (6 * f.a)) | morton3(cx & m, cy & m, cz & m);
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
// run_key[r] = packed cell coordinates (pack_key) of the run's cell, run_start[r] = its first point (tile_off: the scanned tile_cnt)
hipError_t launch_cell_runs_emit(const float4* pts, uint32_t n, Grid grid, const uint32_t* tile_off, uint64_t* run_key, uint32_t* run_start, hipStream_t s);
// dst_mapped[0] = *n_runs, [1..6] = the box of all tiles, [7] = *check (optional), then `seq` at dst_mapped[32] (system scope): the host polls
hipError_t launch_cell_box_publish(const uint32_t* tile_box, uint32_t n_tiles, const uint32_t* n_runs, const uint32_t* check, uint32_t* dst_mapped, uint32_t seq,
                                   hipStream_t s);
// sort_key[r] = compressed key of run_key[r] (uint32_t[] when f.bits <= 32, uint64_t[] otherwise), run_val[r] = first point << 32 | points
// of run r; *err |= 2 when a cell lies outside the frame
hipError_t launch_compress_run_keys(const uint64_t* run_key, const uint32_t* run_start, uint32_t n_runs, uint32_t n, const CellKeyFrame& f, void* sort_key,
                                    uint64_t* run_val, uint32_t* err, hipStream_t s);
// stable sort of the runs by compressed key, 32- or 64-bit keys by `bits` (the values ride along: nothing is gathered through an index afterwards)
hipError_t sort_runs(void* temp, size_t temp_bytes, const void* keys_in, void* keys_out, const uint64_t* vals_in, uint64_t* vals_out, uint32_t n_runs, int bits,
                     hipStream_t s);
// first destination slot of every sorted run (exclusive scan of the run lengths in sorted order)
hipError_t scan_run_lengths(void* temp, size_t temp_bytes, const uint64_t* sorted_val, uint32_t n_runs, uint32_t* dst_start, hipStream_t s);
// the points of sorted run j to dst[dst_start[j]..); abs_key[j] = Morton code of its cell; level_parts[PLACE_WORKGROUPS][TREE_MAX_LEVELS]:
// per workgroup, the sorted runs that start a new cell of level L
hipError_t launch_runs_place(const float4* src, float4* dst, const void* sorted_key, const uint64_t* sorted_val, const uint32_t* dst_start,
                             uint32_t n_runs, const CellKeyFrame& f, uint64_t* abs_key, uint32_t* level_parts, hipStream_t s);
// cells = runs of equal abs_key: cell_keys, cell_val (first point << 32 | points), *n_cells
hipError_t reduce_runs_to_cells(void* temp, size_t temp_bytes, const uint64_t* abs_key, const uint64_t* sorted_val, const uint32_t* dst_start, uint32_t n_runs,
                                uint64_t* cell_keys, uint64_t* cell_val, uint32_t* n_cells, hipStream_t s);
// tree level 1 from the packed cell records (keys >> 6, points summed)
hipError_t coarsen_cells_packed(void* temp, size_t temp_bytes, const uint64_t* cell_keys, const uint64_t* cell_val, size_t n, uint64_t* keys_out, uint32_t* counts_out,
                                uint32_t* n_out, hipStream_t s);
// dst_mapped[L] = cells of level L (sum of the partials), L < TREE_MAX_LEVELS, [TREE_MAX_LEVELS] = *err; then `seq` as above
hipError_t launch_level_counts_publish(const uint32_t* level_parts, const uint32_t* err, uint32_t* dst_mapped, uint32_t seq, hipStream_t s);

}  // namespace lio
