// lio_kernels.h -- launch interface between the host side of liblio_hip.so (lio_ctx.hip, lio_update.hip, lio_scan.hip, lio_pull.hip, lio_pass.hip) and the
// gfx950 kernels (lio_knn.hip, lio_measure.hip, lio_map.hip).  Internal; the public boundary is
// include/lio_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "lio_common.h"

namespace lio {

// ---- k-NN (replaces KD_TREE::Nearest_Search / Search, ikd_Tree.cpp:367-397, 869-1013) ---------
struct KnnArgs {
    const float4*    map_pts;       // Morton-sorted map
    const HashEntry* l1table;       // level-1 table (cells of 4x4x4 level-0 cells): key -> child block number
    uint32_t         l1mask;
    const uint2*     child;         // [blocks][64] {start, count} of the level-0 cells of a level-1 cell
    Grid             grid;
    const float4*    world_in;      // queries in the world frame (body == nullptr)
    const float4*    body;          // or: feats_down_body -- the kernel transforms with `pose` itself (one wavefront per
    float4*          world_out;     //     workgroup, fp64 -> fp32, laserMapping.cpp:1894-1899) and writes feats_down_world
    // the first search of a scan that was handed over as page-locked HOST records (lio_scan_set_async): the transforming wavefront
    // pulls x, y, z of its queries over PCIe itself (device alias of the host records, `body_raw_stride` bytes apart) and leaves
    // them in `body_fill` (= body) for every later reader -- the pull of the whole scan runs beside the search instead of in front
    // of it (a stand-alone pull kernel: ~105 us for 100 k records of 48 bytes, then a 23 us search; fused: the longer of the two)
    const char*      body_raw;      // nullptr: the scan is resident, read `body`
    int              body_raw_stride;
    int              body_raw_aligned;   // records start on 16-byte boundaries: one 16-byte read each
    float4*          body_fill;
    // ... and the host helps (lio_pull.hip, assisted pull): while the workgroups dispatched first pull the BACK of the scan as raw
    // records, the host thread that would otherwise only wait for the pass packs the FRONT [0, pack_front) into 16-byte records
    // (`pack_src`: device alias of its page-locked staging buffer) and publishes how far it has come in `pack_progress`
    // (stamp << 32 | records done; fine-grained device memory where the BAR maps it).  A third of the bytes cross PCIe for every
    // record the host gets to first.  A workgroup whose records are not there in time reads them raw: same values either way.
    const float4*    pack_src;      // nullptr: everything raw
    const unsigned long long* pack_progress;
    uint32_t         pack_stamp;
    int              pack_front;    // a multiple of 64 records (whole workgroups whatever the lanes per query)
    int              pack_mid;      // > 0: a second host thread packs [pack_mid, pack_front) and announces it in pack_progress[1]
                                    // (the absolute record number it has reached); [0, pack_mid) belongs to pack_progress[0]
    int              pack_rot;      // workgroup b takes query block (b + pack_rot) mod blocks: the raw part is dispatched first
    unsigned long long* pack_wait_out;   // host-mapped: stamp << 32 | ticks (100 MHz) the LAST workgroup of the front waited for the host's
                                    // word, 0xffffffff = it gave up and read raw records (the host then takes a smaller share, assist_share)
    Pose14           pose;
    const GateState* gate;          // non-null: the pose (and the order to skip the launch) come from there (lio_common.h)
    // non-null: this launch is the first kernel of a pass enqueued ahead of its pose and there is NO gate kernel in front of it --
    // the transforming wavefront of every workgroup waits for the host's word itself (gate_poll, lio_device.h); workgroup 0
    // leaves {skip, pose} in gate_out for the measurement kernel behind this launch
    const GateMail*  mail;
    unsigned long long expect;
    GateState*       gate_out;
    int              nq;
    int              nq_stride;     // leading dimension of nbr
    float4*          nbr;           // [5][nq_stride]: x, y, z, sqdist of the k-th neighbour
    uint8_t*         cnt;           // [nq] neighbours found (0..5)
    float4*          ulist;         // queries the 27-cell search could not prove: (x, y, z, query number as bits),
    uint32_t*        n_unres;       // appended here (one atomic per unproven query: tens per 100 k) for the tree walk
    unsigned long long* useeds;     // [list entry][8]: the five (sqdist bits << 32 | map index) keys of the unproven 27-cell result
    unsigned long long* cand_total; // device counter: sum of C(q)
    int              use_hint;      // nbr/cnt hold this scan's previous neighbours: bound the search with them
    int              tune;          // experiment switches (lio_debug_set "knn_tune"): bit 0 = do NOT raise the transforming wavefront's priority
    int              walk;          // candidate walk: 0 = list of {start, count}, one load in flight per lane; 1 / 2 / 4 = prefix-form
                                    // list with that many candidates per lane in flight (4, 8 or 16 lanes per query; lio_knn.hip)
    // diagnostics (lio_debug_set "knn_stamps"; 8 lanes per query, walk 4 only): non-null = the STAMPED instantiation of the kernel --
    // wavefront 0 of every workgroup leaves eight wall_clock64 stamps (100 MHz) at [workgroup][8]: entry, query known (transform +
    // barrier), directory records arrived, bucket list in prefix form, candidates walked, merged + proven, results stored, spare.
    // The launch ramp and the rungs of the kernel's dependent chain, per workgroup (scripts/knn_stamps.py).
    unsigned long long* stamps;
};
constexpr int KNN_STAMP_WORDS = 8;
// feats_down_world = pose (x) feats_down_body as a launch of its own (large scans only; otherwise inside knn27_kernel)
hipError_t launch_transform(const float4* body, int n, const Pose14& pose, float4* world, hipStream_t s);
// lanes_per_query in {4, 8, 16, 32, 64}
// optional per-launch timing: when `start` is set the kernel is dispatched through hipExtLaunchKernelGGL,
// which stamps the two events with the dispatch's own begin/end (no marker packets around the kernel)
struct KernelTimer { hipEvent_t start = nullptr, stop = nullptr; };
hipError_t launch_knn27(const KnnArgs& a, int lanes_per_query, hipStream_t s, KernelTimer t = KernelTimer());
// the gate in front of a pre-enqueued pass: waits (bounded) until mail->seq == expect, then st = {skip, pose}
hipError_t launch_gate(const GateMail* mail, unsigned long long expect, GateState* st, hipStream_t s);

// hashed 64-ary tree: level L = grid coarsened by 4^L, tables of levels 1.. live in one buffer
// or-ed into cnt[q] by knn27_kernel: the exact tree walk owes / has answered this query (the walk keeps the mark, every
// reader of a count masks it)
constexpr uint8_t CNT_UNPROVEN = 0x80;
struct TreeArgs {
    const float4*    map_pts;
    const uint2*     child;         // cell directory blocks (level-0 cells), see lio_device.h
    const HashEntry* ltable;        // levels >= 1
    TreeLevels       levels;
    Grid             grid;
    const float4*    ulist;         // the unproven queries: (x, y, z, query number as bits)
    const unsigned long long* useeds;  // [list entry][8] keys of their 27-cell results (nullptr: the walk starts from nothing)
    uint32_t*        n_unres;       // how many (written by knn27_kernel; consumed and zeroed by whoever reads the result;
                                    // bit 31 = TREE_OVERFLOW report)
    int              nq;
    int              nq_stride;
    float4*          nbr;
    uint8_t*         cnt;
};
hipError_t launch_knn_tree(const TreeArgs& a, hipStream_t s, KernelTimer t = KernelTimer());

// ---- measurement model (replaces h_share_model's per-point body, laserMapping.cpp:1888-2002) --
struct ResidualArgs {
    const float4*  body;            // feats_down_body
    int            nq;
    int            nq_stride;
    Pose14         pose;
    const GateState* gate;          // non-null: the pose (and the order to skip the launch) come from there (lio_common.h)
    const GateMail*  mail;          // non-null (passes that re-use neighbours only): no gate kernel in front -- the first wavefront of
    unsigned long long expect;      //           every workgroup waits for the host's word itself (gate_poll, lio_device.h)
    const float4*  nbr;             // Nearest_Points
    const uint8_t* cnt;
    uint8_t*       sel;             // point_selected_surf (in: previous pass, out: this pass)
    float4*        world;           // feats_down_world
    float4*        normvec;         // normvec (n^, pd2)
    float4*        plane;           // (n^, d) of the plane fitted in the last search pass, valid where sel
    double*        rows;            // optional [nq][13]: h_x row (12) + h, valid where sel
    int*           m_dev;           // optional device word: receives m (the callback path's row compaction reads it)
    unsigned long long* units;      // [nblocks][RED_STRIDE][2] stamped halves of the workgroup partial sums
    unsigned long long* gunits;     // [groups][RED_STRIDE][2] the same for the group sums
    uint32_t*      n_unresolved;    // read into the record and reset (search passes)
    unsigned long long* cand_total; // ditto
    DeviceOut*     out;             // host-mapped result record
    unsigned long long seq;         // published last (system-scope release) in out->seq
    int            debug_fail;      // test hooks (lio_debug_set): bit 0 "force_reduction_timeout" reports a timed-out wait, bit 1 "stamps"
    // search passes: the exact tree walk of the queries the 27-cell kernel could not prove runs INSIDE this launch, on
    // the first n_walkers workgroups (one walking wavefront each); a model workgroup that owns such a query (marked in
    // cnt by CNT_UNPROVEN) waits until every listed query has been walked (walk_done == number of listed queries)
    int            n_walkers;       // 0: no walk in this launch (the stand-alone knn_tree_kernel ran, or a no-search pass)
    uint32_t*      walk_done;       // device counter: walked queries of this pass (reset by the last workgroup)
    TreeArgs       tree;            // map index + the list (tree.nbr / tree.cnt = the scan's neighbour arrays)
    // Few listed queries (<= FIX_MAX, the normal case): nobody waits for a walk.  The walker workgroup of a listed query
    // walks, evaluates the model on the exact neighbours and publishes the query's term of every sum as a stamped partial
    // in lunits[rank of the query among the listed ones]; it also writes the query's outputs.  The owner of the query
    // contributes a zero row.  The last model workgroup folds the listed slots like the group partials.  More listed
    // queries: the owners wait for the walk (see above).
    unsigned long long* lunits;     // [FIX_MAX][RED_STRIDE][2] stamped halves of the listed queries' terms
};
constexpr int FIX_MAX = 256;                 // listed queries up to which their walkers evaluate the model
int residual_blocks(int nq);
int residual_walkers(int nq);       // walking workgroups a search pass adds in front of the model workgroups
hipError_t launch_residual(const ResidualArgs& a, bool do_search, bool extrinsic_est_en, hipStream_t s, KernelTimer t = KernelTimer());

// h_x (m x 12) and h (m) of a pass, compacted in scan order and COLUMN-major with leading dimension m -- the layout of
// ekfom_data.h_x / .h the unmodified update_iterated_dyn_share_modified reads (esekfom.hpp:1790,1810): dst = [12 columns of
// h_x | h], one launch (stamped tile prefix, lio_compact.hip).  dst may be device memory or host-mapped pinned memory.
struct RowsCompactArgs {
    const double*  rows;            // [nq][13] staged by residual_kernel (valid where sel)
    const uint8_t* sel;
    int            nq;
    const int*     m_dev;           // m, written by the residual kernel of the same pass
    double*        dst;             // 13 * m doubles
    unsigned long long* agg;
    uint32_t       stamp;
    uint32_t*      err;
};
hipError_t launch_rows_compact(const RowsCompactArgs& a, hipStream_t s);

// ---- map construction / mutation ----------------------------------------------------------------
// mkeys[i] = Morton key of pts[i], vals[i] = base_index + i
hipError_t launch_point_keys(const float4* pts, uint32_t n, Grid grid, uint64_t* mkeys, uint32_t* vals,
                             hipStream_t s, uint32_t base_index = 0);
hipError_t launch_gather_points(const float4* src, const uint32_t* vals, uint32_t n, float4* dst,
                                hipStream_t s);
hipError_t launch_table_insert(const uint64_t* cell_mkeys, const uint32_t* cell_start,
                               const uint32_t* cell_count, uint32_t n_cells, HashEntry* table,
                               uint32_t table_mask, hipStream_t s);
// cell directory of the k-NN kernel: level-1 entries carry the number of their child block, the block holds
// {start, count} of the 64 level-0 cells at a fixed position (no hashing, no collisions on the 27 look-ups)
hipError_t launch_iota(uint32_t* out, uint32_t n, hipStream_t s);
hipError_t launch_publish_words(const uint32_t* src, int n, uint32_t* dst_mapped, uint32_t seq, hipStream_t s);
hipError_t launch_fill_child_blocks(const uint64_t* cell_mkeys, const uint32_t* cell_start, const uint32_t* cell_count,
                                    uint32_t n_cells, const HashEntry* l1table, uint32_t l1mask, uint2* child, hipStream_t s);
hipError_t launch_fill_child_blocks_packed(const uint64_t* cell_mkeys, const uint64_t* cell_val, uint32_t n_cells, const HashEntry* l1table, uint32_t l1mask,
                                           uint2* child, uint32_t* block_bump, uint32_t n_blocks, uint32_t* level_new, hipStream_t s);
// w != 0 for every live point inside any of the nb half-open boxes; *counter += points tomb-stoned
hipError_t launch_delete_boxes(float4* pts, uint32_t n, const float* boxes6, int nb, uint32_t* counter,
                               hipStream_t s);
// stream compaction of live points (w == 0) into dst; *n_out = count.  Order-preserving.
hipError_t launch_pack_xyz(const float4* src, uint32_t n, float* dst_xyz, hipStream_t s);
hipError_t launch_mark_reported(float4* pts, uint32_t n, hipStream_t s);
hipError_t launch_unpack_neighbors(const float4* nbr, const uint8_t* cnt, int nq, int nq_stride,
                                   float* out_xyz, float* out_d, int* out_cnt, hipStream_t s);

// ---- replica exchange (lio_replicate.hip) -------------------------------------------------
constexpr int REP_HEADER_F4 = 8;             // float4s in front of the lists: {n_add, n_nodown, n_boxes, flags}, 24 floats of boxes, pad
constexpr int REP_MAX_BOXES = 4;             // delete boxes per rank and exchange (cub_needrm holds <= 3, laserMapping.cpp:1183-1198)
constexpr int REP_MAX_BOXES_TOTAL = 64;      // all ranks together (the context's box buffer)
constexpr uint32_t REP_FLAG_OVERFLOW = 1u;   // a rank had more points than the message holds
struct RepPackArgs {
    const uint32_t* counts;         // device: [0] points in add, [1] points in nodown
    uint32_t        cap;            // points per list and message
    const float4*   add;
    const float4*   nodown;
    float4*         msg;            // this rank's message
    float           boxes[6 * REP_MAX_BOXES];
    int             n_boxes;
};
struct RepCompactArgs {
    const float4*   gathered;       // [world] messages
    int             world;
    uint32_t        cap;
    float4*         add_out;        // concatenation in rank order
    float4*         nodown_out;
    float*          boxes_out;      // [<= REP_MAX_BOXES_TOTAL][6]
    uint32_t*       totals;         // [0] add, [1] nodown, [2] boxes, [3] flags
    uint32_t*       bbox;           // [6] bounding box of every gathered point (order images, lio_mapadd.h): += 
};
hipError_t launch_replicate_pack(const RepPackArgs& a, hipStream_t s);
hipError_t launch_replicate_compact(const RepCompactArgs& a, hipStream_t s);
// loop-back test mode: park the lists (<= cap points each) and the two counts of one virtual rank
hipError_t launch_replicate_stage(const uint32_t* counts, const float4* add, const float4* nodown, uint32_t cap, float4* stage_add,
                                  float4* stage_nodown, uint32_t* stage_counts, hipStream_t s);

// ---- rocPRIM-backed helpers (lio_sort.hip) ------------------------------------------------------
size_t sort_pairs_temp_bytes(size_t n);
hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                              const uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t s);
hipError_t sort_keys_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, hipStream_t s);
size_t rle_temp_bytes(size_t n);
hipError_t run_length_encode_u64(void* temp, size_t temp_bytes, const uint64_t* keys, size_t n,
                                 uint64_t* unique_out, uint32_t* counts_out, uint32_t* n_runs_out,
                                 hipStream_t s);
// same on (keys[i] >> shift): the cells of a coarser tree level
hipError_t run_length_encode_shifted(void* temp, size_t temp_bytes, const uint64_t* keys, size_t n, int shift,
                                     uint64_t* unique_out, uint32_t* counts_out, uint32_t* n_runs_out,
                                     hipStream_t s);
size_t scan_temp_bytes(size_t n);
hipError_t exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, size_t n,
                              hipStream_t s);
size_t select_temp_bytes(size_t n);
// order-preserving compaction of the live (w == 0) points of src into dst; *n_out = live count
hipError_t select_live_points(void* temp, size_t temp_bytes, const float4* src, float4* dst,
                              uint32_t* n_out, size_t n, hipStream_t s);
hipError_t select_box_deleted_points(void* temp, size_t temp_bytes, const float4* src, float4* dst, uint32_t* n_out,
                                     size_t n, hipStream_t s);
hipError_t select_run_heads_u64(void* temp, size_t temp_bytes, const uint64_t* sorted_keys, size_t n, uint32_t* starts_out,
                                uint32_t* n_runs_out, hipStream_t s);

hipError_t coarsen_level(void* temp, size_t temp_bytes, const uint64_t* keys_in, const uint32_t* counts_in, size_t n,
                         uint64_t* keys_out, uint32_t* counts_out, uint32_t* n_out, hipStream_t s);
size_t merge_temp_bytes(size_t n1, size_t n2);
hipError_t merge_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* k1, const uint64_t* k2, uint64_t* kout,
                               const uint32_t* v1, const uint32_t* v2, uint32_t* vout, size_t n1, size_t n2, hipStream_t s);
hipError_t select_live_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, const uint32_t* vals_in,
                             const float4* pts, uint64_t* keys_out, uint32_t* vals_out, uint32_t* n_out, size_t n,
                             hipStream_t s);
hipError_t merge_pairs_identity(void* temp, size_t temp_bytes, const uint64_t* k1, const uint64_t* k2, uint64_t* kout,
                                const uint32_t* v2, uint32_t* vout, size_t n1, size_t n2, hipStream_t s);
hipError_t select_points_by_class(void* temp, size_t temp_bytes, const float4* src, const uint8_t* cls, uint8_t want,
                                  float4* dst, uint32_t* n_out, size_t n, hipStream_t s);

}  // namespace lio
