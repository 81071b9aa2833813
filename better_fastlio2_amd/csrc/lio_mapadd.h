// lio_mapadd.h -- launch interface of the Add_Points kernels (lio_mapadd.hip), the single-launch compaction primitives
// (lio_compact.hip) and the group sort (lio_gsort.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

// ---- compressed keys ----------------------------------------------------------------------------------------------
// A batch is grouped twice: by down-sample voxel (Add_Points' fold) and by grid cell (the index update).  The key of a
// point is its integer voxel / cell coordinate RELATIVE to the batch's own bounding box -- the host knows that box (it
// arrives with the list lengths) -- so a key has as many bits as the batch has extent: 20-30 for a scan, where the
// packed 3 x 21-bit key had 63.  Fewer bits = fewer radix passes (lio_gsort.hip).  A coordinate outside the frame
// cannot belong to a batch point: look-ups of such keys answer "no such group".
struct KeyFrame {
    int ox, oy, oz;                 // smallest coordinate of the batch per axis
    int bx, by, bz;                 // bits per axis: coordinates are in [o, o + 2^b)
};
LIO_HD int frame_bits(const KeyFrame& f) { return f.bx + f.by + f.bz; }
LIO_HD bool frame_holds(const KeyFrame& f, int x, int y, int z) {
    const long long dx = (long long)x - f.ox, dy = (long long)y - f.oy, dz = (long long)z - f.oz;
    return dx >= 0 && dx < (1ll << f.bx) && dy >= 0 && dy < (1ll << f.by) && dz >= 0 && dz < (1ll << f.bz);
}
// The key of a coordinate triple inside its frame.  NOT x | y | z from the top: the group sort (lio_gsort.hip) splits a batch by the
// top eight key bits first, and with the high coordinate bits on top those are slabs of space -- a scan's inserts crowd around
// the sensor, eight ranks' lists put thousands of records into a few of the 256 buckets and one workgroup sorts each (61 us of a
// 76 k-point batch, profiles/r05z_emulated_world8_timeline.txt).  The top digit is made of MIDDLE bits instead -- bits 2..4 of x
// and y, 2..3 of z (fewer where an axis has fewer): space is cut into blocks of 4 x 4 x 4 coordinates, and the 256 buckets take
// the blocks in turn with a period of 32 x 32 x 16 -- whatever the shape of the cloud every bucket gets its share, and a bucket
// still holds whole blocks, so that neighbouring positions of the sorted batch are neighbouring voxels / cells (the kernels behind
// the sort read the map around them: with the LOWEST bits on top -- tried first -- cell_apply_kernel went from 14 to 36 us).
// Layout, top to bottom: [digit x y z | high x y z | low x y z].  Nothing needs the order of the keys to mean anything -- groups
// are runs of equal keys, look-ups are binary searches for an exact key, the decode below is the inverse -- only that every party
// uses the same one; with frame origins that are multiples of 32 (frame_of) it does not depend on the frame either.
constexpr int KEY_BLOCK_BITS = 2;
struct KeyLayout { int l[3], d[3], h[3]; };       // low / digit / high bits per axis
LIO_HD KeyLayout frame_layout(const KeyFrame& f) {
    const int b[3] = {f.bx, f.by, f.bz}, dmax[3] = {3, 3, 2};
    KeyLayout k;
    for (int i = 0; i < 3; ++i) {
        k.l[i] = b[i] < KEY_BLOCK_BITS ? b[i] : KEY_BLOCK_BITS;
        const int rest = b[i] - k.l[i];
        k.d[i] = rest < dmax[i] ? rest : dmax[i];
        k.h[i] = rest - k.d[i];
    }
    return k;
}
LIO_HD uint64_t frame_key(const KeyFrame& f, int x, int y, int z) {
    const KeyLayout k = frame_layout(f);
    const uint32_t u[3] = {(uint32_t)(x - f.ox), (uint32_t)(y - f.oy), (uint32_t)(z - f.oz)};
    uint64_t dig = 0, hi = 0, lo = 0;
    for (int i = 0; i < 3; ++i) {
        lo = (lo << k.l[i]) | (uint64_t)(u[i] & ((1u << k.l[i]) - 1u));
        dig = (dig << k.d[i]) | (uint64_t)((u[i] >> k.l[i]) & ((1u << k.d[i]) - 1u));
        hi = (hi << k.h[i]) | (uint64_t)(u[i] >> (k.l[i] + k.d[i]));
    }
    const int lb = k.l[0] + k.l[1] + k.l[2], hb = k.h[0] + k.h[1] + k.h[2];
    return (((dig << hb) | hi) << lb) | lo;
}
LIO_HD void frame_decode(const KeyFrame& f, uint64_t key, int& x, int& y, int& z) {
    const KeyLayout k = frame_layout(f);
    const int lb = k.l[0] + k.l[1] + k.l[2], hb = k.h[0] + k.h[1] + k.h[2];
    uint64_t lo = key & ((1ull << lb) - 1ull), hi = (key >> lb) & ((1ull << hb) - 1ull), dig = key >> (lb + hb);
    uint32_t u[3];
    for (int i = 2; i >= 0; --i) {
        const uint32_t l = (uint32_t)(lo & ((1ull << k.l[i]) - 1ull)); lo >>= k.l[i];
        const uint32_t d = (uint32_t)(dig & ((1ull << k.d[i]) - 1ull)); dig >>= k.d[i];
        const uint32_t h = (uint32_t)(hi & ((1ull << k.h[i]) - 1ull)); hi >>= k.h[i];
        u[i] = (h << (k.l[i] + k.d[i])) | (d << k.l[i]) | l;
    }
    x = f.ox + (int)u[0]; y = f.oy + (int)u[1]; z = f.oz + (int)u[2];
}
// integer down-sample voxel of a coordinate: floor(v / ds) (ikd_Tree.cpp:424), clamped like the grid cells
LIO_HD int voxel_coord(float v, float ds) {
    float f = __builtin_floorf(v / ds);
    f = f < -1048576.0f ? -1048576.0f : (f > 1048575.0f ? 1048575.0f : f);
    if (!(f == f)) f = 0.0f;
    return (int)f;
}

// frame of the integer coordinates lo .. hi (inclusive) of a batch: one spare coordinate per axis above hi, so that the
// all-ones key is never a point's (the group of the non-finite points, lio_mapadd.hip)
// The origin is rounded down to a multiple of 32: the low five bits of a coordinate inside the frame are then those of the
// coordinate itself, and the ORDER of the keys (frame_key below: a digit of middle bits leads) does not depend on which box the
// frame was derived from -- the list kernel's box of the list points and the library path's box of the whole scan sort a batch alike.
LIO_HD KeyFrame frame_of(const int lo[3], const int hi[3]) {
    KeyFrame f{};
    const int o[3] = {lo[0] & ~31, lo[1] & ~31, lo[2] & ~31};
    f.ox = o[0]; f.oy = o[1]; f.oz = o[2];
    int bits[3];
    for (int i = 0; i < 3; ++i) {
        const long long extent = (long long)hi[i] - (long long)o[i] + 2;
        // smallest b >= 1 with 2^b >= extent (no loop: the list kernel's last lane runs this at the tail of its launch)
        bits[i] = extent <= 2 ? 1 : 64 - __builtin_clzll((unsigned long long)(extent - 1));
    }
    f.bx = bits[0]; f.by = bits[1]; f.bz = bits[2];
    return f;
}
// of the down-sample voxel keys of a batch inside the box blo .. bhi
LIO_HD KeyFrame voxel_frame_of(const float blo[3], const float bhi[3], float ds) {
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { lo[i] = voxel_coord(blo[i], ds); hi[i] = voxel_coord(bhi[i], ds); }   // floor(v / ds) is monotone in v
    return frame_of(lo, hi);
}
// of the grid cells the pending points of a batch can fall into: survivors are batch points or map points inside the
// down-sample boxes of batch points, i.e. within one box edge of the batch's bounding box
LIO_HD KeyFrame cell_frame_of(const float blo[3], const float bhi[3], float ds, float inv_cell) {
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
        lo[i] = cell_coord(blo[i] - 2.0f * ds, inv_cell) - 1;
        hi[i] = cell_coord(bhi[i] + 2.0f * ds, inv_cell) + 1;
    }
    return frame_of(lo, hi);
}

// What the kernel that makes map_incremental's two insert lists leaves ON THE DEVICE for the kernels of the map update behind
// it (round 6: no host round trip between the lists and their Add_Points -- the host used to read the two counts and the
// bounding box back, derive the frames and only then launch): the list lengths and both key frames, derived by the last tile
// of that kernel with the very functions the host uses (frame_of above: same integers either way).
struct BatchDesc {
    uint32_t n_add, n_plain;        // PointToAdd / PointNoNeedDownsample as the update is to see them (0 / 0 when flags != 0)
    uint32_t flags;                 // 1: a bounded wait of the list kernel ran out; 2: a frame too wide for the group sort; 4: a list longer
                                    // than the launches behind were sized for (2, 4: the host's own path takes the lists -- they are
                                    // intact, their counts and box where they always were)
    uint32_t kb_v, kb_c;            // key bits of the two sorts: frame_bits(vf), frame_bits(cf) + 1
    uint32_t pad[3];
    KeyFrame vf;                    // of the down-sample voxel keys
    KeyFrame cf;                    // of the grid-cell keys
};
constexpr int BATCH_KB_MAX = 46;    // 64 - GSORT_INDEX_BITS (lio_gsort.hip: the record is key << 18 | input position)

// ---- a batch grouped by key: what lio_gsort.hip leaves --------------------------------------------------------------
// Sorted keys + the batch index of the record at every sorted position.  Equal keys are contiguous and in batch order;
// a GROUP is named by the position of its first record (its HEAD): position s is a head iff s == 0 or the keys at s-1 and s differ.
// No run-length arrays, no dense group numbers: kernels take one lane (group) per POSITION and only heads act.
struct GroupView {
    const uint64_t* skey;
    const uint32_t* sord;
    const uint32_t* n_dev;          // optional: *n_dev + n_add records are real (the count is still on the device)
    uint32_t        n_add;
    uint32_t        n;              // records, or the launch bound when n_dev != nullptr
    int             shift;          // groups are runs of equal (skey >> shift): 1 for the cell view, whose low key bit only orders
                                    // the two insert lists INSIDE a cell
};

struct FoldArgs {
    float4*          map_pts;       // tomb-stones are written here
    const HashEntry* l1table;       // cell directory (lio_device.h)
    uint32_t         l1mask;
    const uint2*     child;
    Grid             grid;
    float            ds;            // downsample_size (ikd_Tree.h:197)
    const float4*    new_pts;       // the batch, in the caller's order
    GroupView        view;          // the batch grouped by down-sample voxel
    float4*          out_pts;       // per POSITION: a head's survivor to append (w == 0) or nothing (w == 1); pre-set to "nothing"
    uint32_t*        n_added;       // += tmp_counter of ikd_Tree.cpp:448
    uint32_t*        n_replaced;    // += points of the map tomb-stoned by the fold (validnum bookkeeping)
    const uint8_t*   dirty;         // per head position: 1 = left to the sequential replay ("boundary points")
    float4*          plan;          // [2 n] per head position: {survivor xyz, tmp_counter bits (0: nothing to do)}, {box min xyz, -}
    uint2*           ranges;        // [8 n] per head position and lane: the bucket {start, count} that lane scanned
    uint2*           count_parts;   // [workgroups of fold_commit] {replaced map points, tmp_counter} per workgroup: summed by pend_build
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
// Bounding box of a batch as six order-preserving unsigned images of floats (min x, y, z, max x, y, z): accumulated
// with atomicMin / atomicMax by the kernels that produce the insert lists, read back with the list lengths, reset by
// the first kernel that consumes the lists.  A stale (wider) box is harmless: keys only get more bits.
constexpr uint32_t BBOX_MIN_INIT = 0xffffffffu, BBOX_MAX_INIT = 0u;
LIO_HD uint32_t float_order_image(float v) {
    union { float f; uint32_t u; } c; c.f = v;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
LIO_HD float float_from_order_image(uint32_t u) {
    union { float f; uint32_t u; } c;
    c.u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return c.f;
}

// ---- incremental index update: appended points enter the index without re-sorting the map ---------------
// The batch is grouped by grid cell (stable: batch order inside a cell).  A touched cell's bucket is
// re-written at the tail of the point array -- its live points in their old order, then the new ones --
// the old slots become TOMB_MOVED, and only that cell's table entry changes.  New cells are inserted into
// the cell directory and, as existence marks, into the coarser levels of the tree.
struct CellPlan { uint32_t old_start, old_count, index, found, live, n_new; };   // index = position of the cell's record in the directory; live = live points of the old bucket, n_new = pending points of the cell
struct ApplyArgs {
    float4*          map_pts;
    Grid             grid;
    HashEntry*       ltable;        // levels >= 1
    uint2*           child;         // cell directory: [blocks][64] {start, count}
    uint32_t*        block_bump;    // next free child block
    uint32_t         blocks_cap;
    TreeLevels       levels;
    const float4*    pend;          // appended points: [PointNoNeedDownsample ...][survivors of the fold ...]
    GroupView        view;          // pending points grouped by (grid cell, list): key = cell key * 2 + (1 for the plain list), so
                                    // that inside a cell the survivors of Add_Points(.., true) precede the plain inserts (:1492-1493)
    KeyFrame         frame;         // of the cell keys
    const KeyFrame*  frame_dev;     // nullptr, or where the frame is on the device (BatchDesc: it replaces `frame`)
    CellPlan*        plan;          // [position] (heads)
    uint32_t*        new_size;      // [position] live old points + new points of a head, 0 elsewhere
    uint32_t*        new_off;       // [position] exclusive scan of new_size (written by the plan kernel)
    uint32_t         tail;          // first free slot of the point array
    uint32_t         slots_cap;     // physical slots of the point array
    uint32_t*        stats;         // [0] new cells, [1] old slots retired, [2] flags (APPLY_FLAG_*), [3] total new slots,
                                    // [APPLY_LEVEL_NEW + L] new entries of level L (not reset per batch)
    unsigned long long* agg;        // stamped tile sums of the plan kernel's scan (lio_compact.hip)
    uint32_t         stamp;
    uint32_t*        err;
    // (round 6) the statistics leave with the apply kernel: its last workgroup to finish copies stats[0..32) into host-mapped memory
    // and publishes `publish_seq` there (what a publish_words launch behind it did until round 5); nullptr: nobody is waiting
    uint32_t*        publish_dst;
    uint32_t         publish_seq;
    uint32_t*        ticket;        // arrival counter of the workgroups (zero on entry, zero again on exit)
};
constexpr int APPLY_LEVEL_NEW = 16;                // stats[16 + L]: entries hashed level L gained since the last full rebuild
constexpr uint32_t APPLY_FLAG_OUTSIDE_TOP = 1u;   // a new cell lies outside the root node of the tree
constexpr uint32_t APPLY_FLAG_TABLE_FULL = 2u;    // an insertion probe wrapped around a table
constexpr uint32_t APPLY_FLAG_NO_BLOCKS = 8u;     // the pool of child blocks is exhausted
constexpr uint32_t APPLY_FLAG_NO_ROOM = 4u;       // the tail cannot hold the re-written buckets: nothing was changed
hipError_t launch_cell_plan(const ApplyArgs& a, hipStream_t s);
hipError_t launch_cell_apply(const ApplyArgs& a, hipStream_t s);
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
    GroupView        view;          // grouped by voxel
    KeyFrame         frame;         // of the voxel keys
    const KeyFrame*  frame_dev;     // nullptr, or where the frame is on the device (BatchDesc: it replaces `frame`)
    uint8_t*         dirty;         // [n] per head position
    uint8_t*         dirty_pt;      // [n] per batch index: the point belongs to a dirty group
    uint32_t*        dlist;         // [n] batch indices of the points of dirty groups, ascending (compact_dirty)
    uint32_t*        n_dirty;
    bool             list_here;     // the replaying wavefront of fold_commit_kernel makes dlist / n_dirty itself (small batches: one
                                    // launch less; launch_compact_dirty is then not called)
    float4*          seq_out;       // [n] points appended by the sequential pass (w == 0 live)
    uint32_t*        n_added;
    uint32_t*        n_replaced;
};
// boundary marks (dirty / dirty_pt) AND the speculative fold of every group (FoldArgs.plan / .ranges), one launch
hipError_t launch_fold_plan(const FoldArgs& fa, const BoundaryArgs& ba, hipStream_t s);
// the replay list: batch indices with dirty_pt set, in batch order (single-launch compaction)
hipError_t launch_compact_dirty(const BoundaryArgs& a, unsigned long long* agg, uint32_t stamp, uint32_t* err, hipStream_t s);
// the plans of the clean groups carried out AND (its first workgroup) the sequential replay of the dirty points
hipError_t launch_fold_commit(const FoldArgs& fa, const BoundaryArgs& ba, hipStream_t s);
uint32_t fold_commit_workgroups(uint32_t n);            // its grid (= entries of FoldArgs.count_parts)
hipError_t launch_map_incremental(const IncrArgs& a, hipStream_t s);

// first kernel of a down-sampled batch: voxel keys (compressed, frame) + order = iota, the partition histogram of the
// sort that follows (hist: 256 words, zero on entry; nullptr = none), and the scratch of the later kernels reset
// (counters, dirty marks, "nothing" in every survivor slot, the bounding box accumulators for the next batch)
struct VoxelKeyArgs {
    const float4* pts;
    uint32_t      n;                // points, or (desc != nullptr) the launch bound
    const BatchDesc* desc;          // nullptr, or the device-resident count (n_add) and frame (vf) that replace `n` / `frame`
    float         ds;
    KeyFrame      frame;
    uint64_t*     keys;
    uint32_t*     order;
    uint32_t*     hist;
    uint32_t*     zero_counters;    // [0..1] fold counters, [6] number of boundary points  (d_apply + 4)
    uint8_t*      dirty;
    uint8_t*      dirty_pt;
    float4*       out2n;            // [2n] survivor slots of the fold and of the replay (n = the launch bound with desc)
    uint32_t*     bbox;             // [6] reset to "empty"
    uint32_t*     flags;            // |= 1 when a point lies outside the frame (must not happen: reported)
};
hipError_t launch_voxel_keys(const VoxelKeyArgs& a, hipStream_t s);

// The pending list of an index update in ONE launch: [the plain inserts, copied][the survivors of the fold and of the replay,
// compacted in order]; the (cell, list) key of every pending point for the sort by grid cell, its partition histogram, the
// number of survivors (*n_surv) -- the count stays on the device.
struct PendBuildArgs {
    const float4* surv;             // [n_surv_slots] w == 0: a survivor
    uint32_t      n_surv_slots;
    const float4* plain;            // PointNoNeedDownsample (may be nullptr)
    uint32_t      n_plain;
    const BatchDesc* desc;          // nullptr, or the device-resident counts and cell frame: n_plain is then the launch bound of the
                                    // plain list, the survivor slots are two halves of n_surv_slots / 2 of which the first n_add
                                    // each are real, and `frame` is replaced by desc->cf
    uint32_t*     n_pend;           // (with desc) out: n_plain + survivors, the length of the pending list
    uint32_t*     desc_copy;        // (with desc) out: [0] n_add, [1] n_plain, [2] flags -- where the statistics read-back finds them
    float4*       pend;             // out: [n_plain][survivors]
    Grid          grid;
    KeyFrame      frame;            // cell frame
    uint64_t*     keys;             // out: per pending point
    uint32_t*     order;            // out: iota
    uint32_t*     hist;             // partition histogram of the cell sort (nullptr = none)
    uint32_t*     n_surv;
    const uint2*  count_parts;      // the fold's per-workgroup counter pairs (nullptr: no fold ran) ...
    uint32_t      n_count_parts;
    uint32_t*     fold_counts;      // ... their sums: [0] replaced map points, [1] tmp_counter
    uint32_t*     flags;
    unsigned long long* agg;
    uint32_t      stamp;
    uint32_t*     err;
};
hipError_t launch_pend_build(const PendBuildArgs& a, hipStream_t s);

// ---- single-launch compaction primitives (lio_compact.hip) for inputs up to COMPACT_MAX_ITEMS; agg = COMPACT_AGG_WORDS
// 8-byte words of scratch shared by all of them, stamp = a number no earlier launch on this buffer used, *err |= 1 on a
// timed-out wait
constexpr uint32_t DIRTY_LIST_HERE_MAX = 16384;  // batches up to here: fold_commit_kernel lists the dirty points itself
constexpr uint32_t COMPACT_MAX_ITEMS = 262144;    // 1024 tiles of 256: all resident at once
constexpr size_t COMPACT_AGG_WORDS = 8 * 1024;   // (the list kernel publishes eight words per tile: two counts, the six faces of its bounding box)
// map_incremental's classification + PointToAdd / PointNoNeedDownsample in scan order + their counts + their bounding box, one launch
// desc (optional): the BatchDesc of the two lists for a map update that follows without a host round trip (ds / inv_cell: what its frames are derived with)
hipError_t launch_map_incremental_lists(const IncrArgs& a, float4* add_out, float4* nodown_out, uint32_t* counts, uint32_t* bbox,
                                        unsigned long long* agg, uint32_t stamp, uint32_t* err, hipStream_t s,
                                        BatchDesc* desc = nullptr, float ds = 0.0f, float inv_cell = 0.0f, int kb_max = BATCH_KB_MAX,
                                        uint32_t bound_add = 0xffffffffu, uint32_t bound_plain = 0xffffffffu);
// live points (w == 0) to the front of dst, order kept
hipError_t launch_compact_live(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, unsigned long long* agg, uint32_t stamp,
                               uint32_t* err, hipStream_t s);
// stable sort of (u64 key, u32 value) pairs, one launch, no workspace (rank by counting: O(n^2))
constexpr uint32_t SORT_MAX_ITEMS = 16384;
hipError_t launch_sort_pairs_small(const uint64_t* keys_in, const uint32_t* vals_in, uint32_t n, uint64_t* keys_out, uint32_t* vals_out,
                                   hipStream_t s, const uint32_t* n_dev = nullptr, uint32_t n_dev_add = 0, int kb = 64, const uint32_t* kb_dev = nullptr);
// (kb / kb_dev: every key is below 2^kb -- from 32 bits down the kernel compares 32-bit words)
// run starts only, runs of equal keys >> shift (the voxel filter's leaves: the leaf index is the upper half of the key)
hipError_t launch_run_heads(const uint64_t* keys, uint32_t n, int shift, uint32_t* starts_out, uint32_t* n_runs_out, unsigned long long* agg,
                            uint32_t stamp, uint32_t* err, hipStream_t s);
hipError_t launch_exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, unsigned long long* agg, uint32_t stamp, uint32_t* err,
                                 hipStream_t s);
// live points (w == 0) of a large array to the front of dst, order kept (three launches: tile counts, their scan, scatter);
// tile_cnt / tile_off: n / 2048 + 2 words each
hipError_t launch_compact_live_large(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, uint32_t* tile_cnt, uint32_t* tile_off,
                                     unsigned long long* agg, uint32_t stamp, uint32_t* err, hipStream_t s);
// bounding box (order images, six atomics per workgroup) of n float4 points
hipError_t launch_bbox(const float4* pts, uint32_t n, uint32_t* bbox, hipStream_t s);

// ---- group sort (lio_gsort.hip): stable sort of (key, value) records with keys of `kb` bits, two launches --------------
constexpr int GSORT_INDEX_BITS = 18;
constexpr uint32_t GSORT_MAX = 1u << GSORT_INDEX_BITS;      // records (the input position rides in the low bits of the record)
constexpr uint32_t GSORT_RANK_MAX = 8192;                   // up to here the one-launch rank sort is used instead (round 6, with 32-bit compares:
                                                            // up to 16384 measured 10 us slower per map update at 6.3 k inserts, profiles/r06x_*)
constexpr size_t GSORT_AGG_WORDS = (size_t)(GSORT_MAX / 2048u) * 256u;
LIO_HD uint32_t gsort_digit(uint64_t key, int kb) { return (uint32_t)(key >> (kb > 8 ? kb - 8 : 0)) & 255u; }   // the partition's digit
struct GsortArgs {
    const uint64_t* keys;           // [n] keys (< 2^kb), input order = the order inside equal keys
    const uint32_t* vals;           // [n] values
    uint32_t        n;              // records, or the launch bound with n_dev
    const uint32_t* n_dev;          // optional device-resident count: *n_dev + n_dev_add records are real
    uint32_t        n_dev_add;
    int             kb;
    const uint32_t* kb_dev;         // optional: the key bits are still on the device (BatchDesc), `kb` is ignored
    uint64_t*       keys_out;
    uint32_t*       vals_out;
    uint64_t*       part;           // [n] scratch
    uint64_t*       part2;          // [n] scratch (oversized buckets only)
    const uint32_t* hist;           // [256] histogram of gsort_digit over the real records (accumulated by the key kernel)
    uint32_t*       hist_other;     // [256] zeroed here: the histogram buffer of the NEXT sort
    unsigned long long* agg;        // [GSORT_AGG_WORDS]
    uint32_t        stamp;
    uint32_t*       err;
};
// LSD form (lio_gsort.hip): ceil(kb / 8) <= 4 stable splits, least significant digit first; for skewed key distributions
struct GsortLsdArgs {
    const uint64_t* keys;           // [n] keys (< 2^kb), input order = the order inside equal keys
    uint32_t        n;              // <= GSORT_MAX
    int             kb;             // <= 32
    int             passes;         // 0: ceil(kb / 8); otherwise that many 8-bit splits (>= ceil(kb / 8): a split by a digit
                                    // that is zero in every key changes nothing)
    uint64_t*       keys_out;       // [n] sorted keys
    uint32_t*       vals_out;       // [n] input positions in sorted order
    uint64_t*       part;           // [n] two record buffers
    uint64_t*       part2;
    const uint32_t* hist;           // [passes][256] digit histograms of the keys (the key kernel accumulates them)
    unsigned long long* agg;        // [GSORT_AGG_WORDS]
    uint32_t        stamp0;         // pass p is stamped stamp0 + p (none of them 0)
    uint32_t*       err;
};
hipError_t launch_gsort_lsd(const GsortLsdArgs& g, hipStream_t s);
hipError_t launch_gsort_large(const GsortArgs& a, hipStream_t s);
hipError_t launch_gsort_hist(const uint64_t* keys, uint32_t n, const uint32_t* n_dev, uint32_t n_dev_add, int kb, uint32_t* hist, hipStream_t s);

}  // namespace lio
