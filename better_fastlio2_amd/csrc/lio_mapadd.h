// lio_mapadd.h -- launch interface of the Add_Points kernels (lio_mapadd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

struct FoldArgs {
    float4*          map_pts;       // tomb-stones are written here
    const HashEntry* l1table;       // cell directory (lio_device.h)
    uint32_t         l1mask;
    const uint2*     child;
    Grid             grid;
    float            ds;            // downsample_size (ikd_Tree.h:197)
    const float4*    new_pts;       // the batch, in the caller's order
    const uint32_t*  order;         // batch indices sorted by voxel key (stable)
    const uint32_t*  group_start;   // per voxel group: first position in `order`
    const uint32_t*  group_count;
    uint32_t         n_groups;      // number of voxel groups (used when n_groups_dev is null)
    const uint32_t*  n_groups_dev;  // device-resident number of voxel groups (no host round trip)
    uint32_t         n_groups_max;  // launch bound (>= the real number); groups past the real count emit "nothing"
    float4*          out_pts;       // per group: survivor to append (w == 0) or nothing (w == 1)
    uint32_t*        n_added;       // += tmp_counter of ikd_Tree.cpp:448
    uint32_t*        n_replaced;    // += points of the map tomb-stoned by the fold (validnum bookkeeping)
    const uint8_t*   dirty;         // per group: 1 = left to the sequential pass (lio_mapadd.hip, "boundary points")
};
struct IncrArgs {
    const float4*  body;            // feats_down_body
    int            nq, nq_stride;
    Pose14         pose;            // posterior state (state_point, laserMapping.cpp:2381)
    const float4*  nbr;             // Nearest_Points of the last search pass
    const uint8_t* cnt;
    double         filter_size_map_min;
    int            flg_EKF_inited;
    float4*        world;           // feats_down_world at the posterior pose
    uint8_t*       cls;             // 0 skip, 1 PointToAdd, 2 PointNoNeedDownsample
};
// ---- incremental index update: appended points enter the index without re-sorting the map ---------------
// The batch is grouped by grid cell (stable sort: batch order inside a cell).  A touched cell's bucket is
// re-written at the tail of the point array -- its live points in their old order, then the new ones --
// the old slots become TOMB_MOVED, and only that cell's table entry changes.  New cells are inserted into
// the cell directory and, as existence marks, into the coarser levels of the tree.
struct CellPlan { uint32_t old_start, old_count, index, found; };   // index = position of the cell's record in the directory
struct ApplyArgs {
    float4*          map_pts;
    Grid             grid;
    HashEntry*       ltable;        // levels >= 1
    uint2*           child;         // cell directory: [blocks][64] {start, count}
    uint32_t*        block_bump;    // next free child block
    uint32_t         blocks_cap;
    TreeLevels       levels;
    const float4*    pend;          // appended points, batch order
    const uint32_t*  order;         // pending indices sorted by cell key (stable)
    const uint64_t*  cell_key;      // per touched cell: packed key
    const uint32_t*  group_start;   // per touched cell: first position in `order`
    const uint32_t*  group_count;
    const uint32_t*  n_groups;      // device-resident count of touched cells
    CellPlan*        plan;          // [touched cells]
    uint32_t*        new_size;      // [touched cells] live old points + new points
    const uint32_t*  new_off;       // exclusive scan of new_size
    uint32_t         tail;          // first free slot of the point array
    uint32_t         slots_cap;     // physical slots of the point array
    uint32_t         n_scan;        // new_off[n_scan] = total number of slots the re-written buckets need
    uint32_t*        stats;         // [0] new cells, [1] old slots retired, [2] flags (APPLY_FLAG_*), [3] total new slots,
                                    // [APPLY_LEVEL_NEW + L] new entries of level L (not reset per batch)
};
constexpr int APPLY_LEVEL_NEW = 16;                // stats[16 + L]: entries hashed level L gained since the last full rebuild
constexpr uint32_t APPLY_FLAG_OUTSIDE_TOP = 1u;   // a new cell lies outside the root node of the tree
constexpr uint32_t APPLY_FLAG_TABLE_FULL = 2u;    // an insertion probe wrapped around a table
constexpr uint32_t APPLY_FLAG_NO_BLOCKS = 8u;     // the pool of child blocks is exhausted
constexpr uint32_t APPLY_FLAG_NO_ROOM = 4u;       // the tail cannot hold the re-written buckets: nothing was changed
// (n_head: the list is [*n_head][n_after] points in n slots, the rest get the key ~0 -- see the kernel)
hipError_t launch_pend_keys(const float4* pend, uint32_t n, Grid grid, uint64_t* keys, uint32_t* order, const uint32_t* n_head,
                            uint32_t n_after, hipStream_t s);
hipError_t launch_append_points(float4* dst_base, const uint32_t* offset, const float4* src, uint32_t n, hipStream_t s);
hipError_t launch_cell_plan(const ApplyArgs& a, uint32_t max_groups, hipStream_t s);
hipError_t launch_cell_apply(const ApplyArgs& a, uint32_t max_groups, hipStream_t s);
// Boundary points of a down-sampled batch.  The reference's down-sample box is computed in fp32
// (floor(p / ds) * ds, + ds): next to a voxel boundary a point can lie outside its own box, or inside the box of
// the neighbouring voxel.  Such points make the inserts of neighbouring voxels depend on each other, so their
// voxel groups (and the neighbours they reach) are taken out of the parallel fold and replayed one point after the
// other, in batch order, with the reference's rule -- exact, and rare (a fraction ~ulp(p)/ds of the points).
struct BoundaryArgs {
    float4*          map_pts;
    const HashEntry* l1table;
    uint32_t         l1mask;
    const uint2*     child;
    Grid             grid;
    float            ds;
    const float4*    new_pts;       // the batch, caller's order
    const uint32_t*  order;         // batch indices sorted by voxel key
    const uint64_t*  ukeys;         // sorted voxel keys of the groups
    const uint32_t*  group_start;
    const uint32_t*  group_count;
    const uint32_t*  n_groups;      // device-resident
    uint32_t         n;             // batch size
    uint8_t*         dirty;         // [n] per group
    uint32_t*        dlist;         // [n] batch indices of the points of dirty groups (0xffffffff = unused)
    uint32_t*        n_dirty;
    float4*          seq_out;       // [n] points appended by the sequential pass (w == 0 live)
    uint32_t*        n_added;
    uint32_t*        n_replaced;
};
hipError_t launch_classify_boundary(const BoundaryArgs& a, hipStream_t s);
hipError_t launch_detect_shared_points(const BoundaryArgs& a, hipStream_t s);
hipError_t launch_gather_dirty(const BoundaryArgs& a, hipStream_t s);
hipError_t launch_fold_sequential(const BoundaryArgs& a, const uint32_t* dlist_sorted, hipStream_t s);
hipError_t launch_map_incremental(const IncrArgs& a, hipStream_t s);
// ---- single-launch compaction primitives (lio_compact.hip) for inputs up to COMPACT_MAX_ITEMS; agg = COMPACT_AGG_WORDS
// 8-byte words of scratch shared by all of them, stamp = a number no earlier launch on this buffer used, *err |= 1 on a
// timed-out wait
constexpr uint32_t COMPACT_MAX_ITEMS = 262144;    // 1024 tiles of 256: all resident at once
constexpr size_t COMPACT_AGG_WORDS = 2 * 1024;
// map_incremental's classification + PointToAdd / PointNoNeedDownsample in scan order + their counts, one launch
hipError_t launch_map_incremental_lists(const IncrArgs& a, float4* add_out, float4* nodown_out, uint32_t* counts,
                                        unsigned long long* agg, uint32_t stamp, uint32_t* err, hipStream_t s);
// live points (w == 0) to the front of dst, order kept
hipError_t launch_compact_live(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, unsigned long long* agg, uint32_t stamp,
                               uint32_t* err, hipStream_t s);
// entries != 0xffffffff to the front of dst, order kept
hipError_t launch_compact_marked(const uint32_t* src, uint32_t n, uint32_t* dst, uint32_t* n_out, unsigned long long* agg, uint32_t stamp,
                                 uint32_t* err, hipStream_t s);
// runs of a sorted key array: unique keys, run lengths, run starts (exclusive scan of the lengths), number of runs
// (n_head != nullptr: only the first *n_head + n_after keys are real, the rest are ~0 and form no run)
hipError_t launch_run_lengths(const uint64_t* keys, uint32_t n, uint64_t* unique_out, uint32_t* counts_out, uint32_t* starts_out,
                              uint32_t* n_runs_out, unsigned long long* agg, uint32_t stamp, uint32_t* err, hipStream_t s,
                              const uint32_t* n_head = nullptr, uint32_t n_after = 0);
// stable sort of (u64 key, u32 value) pairs, one launch, no workspace (rank by counting: O(n^2))
constexpr uint32_t SORT_MAX_ITEMS = 16384;
hipError_t launch_sort_pairs_small(const uint64_t* keys_in, const uint32_t* vals_in, uint32_t n, uint64_t* keys_out, uint32_t* vals_out,
                                   hipStream_t s);
// run starts only, runs of equal keys >> shift (the voxel filter's leaves: the leaf index is the upper half of the key)
hipError_t launch_run_heads(const uint64_t* keys, uint32_t n, int shift, uint32_t* starts_out, uint32_t* n_runs_out, unsigned long long* agg,
                            uint32_t stamp, uint32_t* err, hipStream_t s);
hipError_t launch_exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, unsigned long long* agg, uint32_t stamp, uint32_t* err,
                                 hipStream_t s);
hipError_t launch_voxel_keys(const float4* pts, uint32_t n, float ds, uint64_t* vkeys, uint32_t* order, uint32_t* zero2,
                             uint8_t* dirty, uint32_t* dlist, float4* out2n, hipStream_t s);
hipError_t launch_fold_groups(const FoldArgs& a, hipStream_t s);

}  // namespace lio
