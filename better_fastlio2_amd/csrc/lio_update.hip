// lio_update.hip -- the map side of liblio_hip.so: index (re)builds (ikd_Tree.cpp:353-364, 1394-1405), the drivers of Add_Points
// (:414-489), Delete_Point_Boxes (:536-556), map_incremental (laserMapping.cpp:1440-1496), lasermap_fov_segment (:1136-1200), flatten /
// acquire_removed_points, the stand-alone Nearest_Search, and the replica exchange of SURVEY.md section 8e (lio_replicate_*: one
// ncclAllGather per scan on the library's stream, immediate or deferred schedule).  Kernels: lio_map.hip, lio_mapadd.hip, lio_gsort.hip,
// lio_compact.hip, lio_sort.hip, lio_replicate.hip.
#include "lio_ctx.h"

namespace {
// the frames of a batch's compressed keys from its bounding box (lio_mapadd.h: the same functions the list kernel runs on the device)
static KeyFrame voxel_frame(const HostBox& box, float ds) { return voxel_frame_of(box.lo, box.hi, ds); }
static KeyFrame cell_frame(const HostBox& box, float ds, float inv_cell) { return cell_frame_of(box.lo, box.hi, ds, inv_cell); }
// the bounding-box accumulators of the insert lists (d_counters[24..29]) back to "empty": by the host when no kernel that
// consumes the lists (voxel_keys_kernel resets them itself) is going to run
hipError_t reset_list_box(lio_ctx* c) {
    return hipMemcpyAsync(c->d_counters + 24, BBOX_RESET, sizeof(BBOX_RESET), hipMemcpyHostToDevice, c->stream);
}
static void box_from_images(const uint32_t img[6], HostBox& box) {
    box.any = img[0] != BBOX_MIN_INIT;
    for (int i = 0; i < 3; ++i) { box.lo[i] = float_from_order_image(img[i]); box.hi[i] = float_from_order_image(img[3 + i]); }
}
static void box_of_host_points(HostBox& box, const float* pts, size_t n, int stride_bytes) {
    const char* src = reinterpret_cast<const char*>(pts);
    for (size_t i = 0; i < n; ++i) {
        const float* p = reinterpret_cast<const float*>(src + i * (size_t)stride_bytes);
        box.add(p[0], p[1], p[2]);
    }
}

// Last part of every (re)build: c->d_pts[0..n) is Morton-sorted and live, c->d_cell_keys / d_cell_counts [0..n_cells) hold
// the Morton code and the number of points of every occupied grid cell, in order.  Derive the coarser tree levels (hash
// tables), fill the child blocks of the cell directory.  level_cells (optional): the number of cells of every level, already
// known to the host (the rebuild by cell runs counts them on the way) -- nothing is read back between the levels then.
// cell_val (optional, with level_cells): the cells as packed records (first point << 32 | points) in place of d_cell_counts / a scan for d_cell_starts.
static int finish_index_levels(lio_ctx* c, uint32_t n, uint32_t n_cells, const uint32_t* level_cells, const uint64_t* cell_val = nullptr) {
    hipStream_t s = c->stream;
    c->n_points = n; c->n_live = n; c->n_cells = 0;
    c->n_retired = 0; c->n_dead = 0; c->n_new_cells = 0; c->n_cells_built = 0;
    c->grid.n_points = n;
    if (cell_val == nullptr) {                  // (with packed records the child-block kernel clears the level counters and no scan is needed)
        HIPCHK(c, hipMemsetAsync(c->d_apply + APPLY_LEVEL_NEW, 0, TREE_MAX_LEVELS * sizeof(uint32_t), s));
        HIPCHK(c, exclusive_scan_u32(c->d_temp, c->temp_bytes, c->d_cell_counts, c->d_cell_starts, n_cells, s));
    }
    c->n_cells = n_cells; c->n_cells_built = n_cells;
    bool drained = false;                      // the last thing this function did was wait for the stream
    // every level's size known up front: the hash tables of all levels are sized, (re)allocated and cleared at once
    bool tables_ready = false;
    if (level_cells != nullptr) {
        size_t total = 0;
        for (int L = 1; L < TREE_MAX_LEVELS; ++L) {
            if (L >= 2 && level_cells[L] <= (uint32_t)TREE_MAX_ROOTS) break;
            size_t sl = 64;
            while (sl < (size_t)level_cells[L] * 4) sl <<= 1;
            total += sl;
        }
        if (total > c->ltable_slots) {
            size_t cap = c->ltable_slots ? c->ltable_slots : 4096;
            while (cap < total) cap <<= 1;
            HashEntry* nt = nullptr;
            HIPCHK(c, hipMalloc((void**)&nt, cap * sizeof(HashEntry)));
            HIPCHK(c, stream_wait(c));
            if (c->d_ltable) (void)hipFree(c->d_ltable);
            c->d_ltable = nt; c->ltable_slots = cap;
        }
        HIPCHK(c, hipMemsetAsync(c->d_ltable, 0xFF, total * sizeof(HashEntry), s));
        tables_ready = true;
    }

    // ---- coarser levels of the hashed 64-ary tree (exact fallback of the k-NN): level L+1 is a
    // reduce-by-key of level L (keys >> 6, counts summed) -- the point array is not touched again
    TreeLevels lv{};
    size_t used = 0;
    uint32_t n_prev = n_cells;
    // ping-pong between the cell arrays and the (now free) build scratch
    uint64_t* kin = c->d_cell_keys; uint32_t* cin = c->d_cell_counts;
    uint64_t* kout = c->d_mkeys_tmp; uint32_t* cout = c->d_vals_tmp;
    // levels 0 and 1 always exist (the cell directory hangs off level 1); the first level from 2 up with at most
    // TREE_MAX_ROOTS cells is not hashed but listed: its cells are the roots of the walk
    for (int L = 1; L < TREE_MAX_LEVELS; ++L) {
        if (L == 1 && cell_val != nullptr) HIPCHK(c, coarsen_cells_packed(c->d_temp, c->temp_bytes, kin, cell_val, n_prev, kout, cout, c->d_counters + 4, s));
        else HIPCHK(c, coarsen_level(c->d_temp, c->temp_bytes, kin, cin, n_prev, kout, cout, c->d_counters + 4, s));
        uint32_t n_l = 0;
        if (level_cells != nullptr) n_l = level_cells[L];
        else HIPCHK(c, read_back(c, &n_l, c->d_counters + 4, sizeof(uint32_t)));
        if (L >= 2 && n_l <= (uint32_t)TREE_MAX_ROOTS) {
            // root keys through the host-mapped counter block (a copy into pageable memory may be completed by the
            // runtime after the stream is seen idle): at most TREE_MAX_ROOTS x 8 bytes = 2 x 128 bytes
            uint64_t rk[TREE_MAX_ROOTS];
            for (uint32_t o = 0; o < n_l; o += 16) {
                const uint32_t part = n_l - o < 16 ? n_l - o : 16;
                HIPCHK(c, read_back(c, rk + o, kout + o, (size_t)part * sizeof(uint64_t)));
            }
            lv.top_level = L; lv.n_roots = (int)n_l;
            for (uint32_t i = 0; i < n_l; ++i) {
                lv.root_x[i] = compact21(rk[i] >> 2); lv.root_y[i] = compact21(rk[i] >> 1); lv.root_z[i] = compact21(rk[i]);
            }
            drained = n_l != 0u;
            break;
        }
        size_t sl = 64;
        while (sl < (size_t)n_l * 4) sl <<= 1;
        if (!tables_ready && used + sl > c->ltable_slots) {
            // grow (keeps the levels already inserted)
            size_t cap = c->ltable_slots ? c->ltable_slots : 4096;
            while (cap < used + sl) cap <<= 1;
            HashEntry* nt = nullptr;
            HIPCHK(c, hipMalloc((void**)&nt, cap * sizeof(HashEntry)));
            if (used > 0) HIPCHK(c, hipMemcpyAsync(nt, c->d_ltable, used * sizeof(HashEntry), hipMemcpyDeviceToDevice, s));
            HIPCHK(c, stream_wait(c));
            if (c->d_ltable) (void)hipFree(c->d_ltable);
            c->d_ltable = nt; c->ltable_slots = cap;
        }
        if (!tables_ready) HIPCHK(c, hipMemsetAsync(c->d_ltable + used, 0xFF, sl * sizeof(HashEntry), s));
        // `start` of a coarse entry is its rank among the cells of its level (cell_start == nullptr); level 1 uses it as the number
        // of its child block, the levels above only need to exist
        HIPCHK(c, launch_table_insert(kout, nullptr, cout, n_l, c->d_ltable + used, (uint32_t)(sl - 1), s));
        lv.off[L] = (uint32_t)used; lv.mask[L] = (uint32_t)(sl - 1);
        c->level_entries[L] = n_l;
        if (L == 1) {
            // cell directory: one block of 64 {start, count} per level-1 cell, with room for the level-1 cells that
            // incremental updates will add
            size_t need = (size_t)n_l + (size_t)n_l / 4 + 4096;
            if (need > c->child_blocks_cap) {
                // (grown by half at least: a re-allocation is a device synchronisation plus ~0.3 ms, not something to pay at every rebuild of a growing map)
                if (need < c->child_blocks_cap + c->child_blocks_cap / 2) need = c->child_blocks_cap + c->child_blocks_cap / 2;
                if (c->d_child) (void)hipFree(c->d_child);
                c->d_child = nullptr; c->child_blocks_cap = 0;
                HIPCHK(c, hipMalloc((void**)&c->d_child, need * 64 * sizeof(uint2)));
                c->child_blocks_cap = need;
            }
            HIPCHK(c, hipMemsetAsync(c->d_child, 0, c->child_blocks_cap * 64 * sizeof(uint2), s));
            c->n_child_blocks = n_l;
            if (cell_val != nullptr) {
                HIPCHK(c, launch_fill_child_blocks_packed(c->d_cell_keys, cell_val, n_cells, c->d_ltable + used, (uint32_t)(sl - 1), c->d_child, c->d_apply + 9, n_l,
                                                          c->d_apply + APPLY_LEVEL_NEW, s));
            } else {
                HIPCHK(c, launch_fill_child_blocks(c->d_cell_keys, c->d_cell_starts, c->d_cell_counts, n_cells,
                                                   c->d_ltable + used, (uint32_t)(sl - 1), c->d_child, s));
                HIPCHK(c, hipMemcpyAsync(c->d_apply + 9, &c->n_child_blocks, sizeof(uint32_t), hipMemcpyHostToDevice, s));
            }
        }
        used += sl;
        n_prev = n_l;
        // next level reads what this one wrote; alternate between two scratch pairs
        if (kout == c->d_mkeys_tmp) { kin = c->d_mkeys_tmp; cin = c->d_vals_tmp; kout = c->d_cell_keys; cout = c->d_cell_counts; }
        else { kin = c->d_cell_keys; cin = c->d_cell_counts; kout = c->d_mkeys_tmp; cout = c->d_vals_tmp; }
    }
    if (!drained) HIPCHK(c, stream_wait(c));
    c->levels = lv;
    c->have_map = true; c->map_broken = false;
    ++c->n_full_rebuilds;
    return LIO_OK;
}
// ... behind the POINT sort (c->d_mkeys[0..n): the sorted keys): the cells are the runs of equal keys
int finish_index(lio_ctx* c, uint32_t n) {
    HIPCHK(c, run_length_encode_u64(c->d_temp, c->temp_bytes, c->d_mkeys, n, c->d_cell_keys, c->d_cell_counts, c->d_counters + 4, c->stream));
    uint32_t n_cells = 0;
    HIPCHK(c, read_back(c, &n_cells, c->d_counters + 4, sizeof(uint32_t)));
    return finish_index_levels(c, n, n_cells, nullptr);
}

int build_index(lio_ctx* c, const float4* src, uint32_t n, const uint32_t* live_dev = nullptr, uint32_t live_expected = 0u, uint32_t* live_found = nullptr);
constexpr int LIO_RETRY_COUNT = -1000;         // internal: build_index found another number of live points than it was told
// does a (re)build of n points go through the sort of cell runs?
static bool by_cell_runs(const lio_ctx* c, size_t n) {
    return c->rebuild_by_cells && !c->dbg_library_sort && n > 0 && n / RUN_TILE + 2u <= COMPACT_MAX_ITEMS && (n / RUN_TILE + 4u) * 8u * sizeof(uint32_t) <= c->temp_bytes;
}
// the POINT sort (points in no order; lio_debug_set "library_sort", LIO_REBUILD_BY_CELLS=0): Morton code per point, 63-bit radix sort, gather
static int build_index_by_points(lio_ctx* c, const float4* src, uint32_t n) {
    hipStream_t s = c->stream;
    HIPCHK(c, launch_point_keys(src, n, c->grid, c->d_mkeys_tmp, c->d_vals_tmp, s));
    HIPCHK(c, sort_pairs_u64_u32(c->d_temp, c->temp_bytes, c->d_mkeys_tmp, c->d_mkeys, c->d_vals_tmp, c->d_vals, n, s));
    HIPCHK(c, launch_gather_points(src, c->d_vals, n, c->d_pts, s));
    return finish_index(c, n);
}

// (Re)build by CELL RUNS (lio_rebuild.h): the n points at src are a sequence of runs of equal grid cells -- the buckets of the old
// index in array order, then the pending points -- and only the runs are sorted, by a compressed key.  Two waits on the way (the
// number of runs with the cells' bounding box; the cells per level), one at the end (the roots).  Same d_pts, cell table and
// tree as the point sort, bit for bit (tests/test_gpu_parity.py::test_rebuild_by_cell_runs_equals_the_point_sort).
// Second half: the runs are known (d_mkeys_tmp: packed cell per run, d_vals_tmp: first point), box6 = the cells' bounding box.
static int build_index_from_runs(lio_ctx* c, const float4* src, uint32_t n, uint32_t n_runs, const uint32_t box6[6]) {
    hipStream_t s = c->stream;
    uint64_t* run_key = c->d_mkeys_tmp;        // packed cell coordinates per run; later the Morton code per SORTED run
    void* sort_key = c->d_cell_keys;           // compressed keys, in and out of the sort (32-bit when the frame allows; free until the cells are made)
    void* sorted_key = c->d_mkeys;
    uint32_t* run_start = c->d_vals_tmp;       // [n] at most (before the number of runs is known)
    // (first point, points) per run, in and out of the sort: 8 bytes per run in the 4-byte-per-point arrays -- at most n / 2 runs get here
    uint64_t* run_val = reinterpret_cast<uint64_t*>(c->d_vals);
    uint64_t* sorted_val = reinterpret_cast<uint64_t*>(c->d_rank);
    uint32_t* dst_start = c->d_cell_starts;
    if (n_runs == 0u || n_runs > n) return c->fail(LIO_ERR_HIP, "index rebuild", "the run kernels left an impossible number of cell runs");
    // points in no particular order (a Build from the caller's cloud): as many runs as points, the run sort would only add to the work
    if ((size_t)n_runs * 2u > (size_t)n) return build_index_by_points(c, src, n);
    const CellKeyFrame f = cell_key_frame(box6, box6 + 3);
    HIPCHK(c, launch_compress_run_keys(run_key, run_start, n_runs, n, f, sort_key, run_val, c->d_apply + 15, s));
    HIPCHK(c, sort_runs(c->d_temp, c->temp_bytes, sort_key, sorted_key, run_val, sorted_val, n_runs, f.bits, s));
    HIPCHK(c, scan_run_lengths(c->d_temp, c->temp_bytes, sorted_val, n_runs, dst_start, s));
    HIPCHK(c, launch_runs_place(src, c->d_pts, sorted_key, sorted_val, dst_start, n_runs, f, run_key, c->d_level_parts, s));
    // (the cells per level travel to the host beside the reduction of the runs to cells)
    const uint32_t seq = ++c->small_seq;
    HIPCHK(c, launch_level_counts_publish(c->d_level_parts, c->d_apply + 15, c->d_small, seq, s));
    uint64_t* cell_val = reinterpret_cast<uint64_t*>(c->d_cell_counts);       // (first point, points) per cell: at most n / 2 of them
    HIPCHK(c, reduce_runs_to_cells(c->d_temp, c->temp_bytes, run_key, sorted_val, dst_start, n_runs, c->d_cell_keys, cell_val, c->d_counters + 4, s));
    HIPCHK(c, wait_published(c, seq));
    uint32_t lv[TREE_MAX_LEVELS + 1];
    std::memcpy(lv, c->h_small, sizeof(lv));
    if (lv[TREE_MAX_LEVELS] != 0u) {            // d_apply[15]: a bounded wait of the tile scan ran out (bit 0), a cell outside the key frame (bit 1)
        (void)hipMemsetAsync(c->d_apply + 15, 0, sizeof(uint32_t), s);
        return c->fail(LIO_ERR_HIP, "index rebuild", lv[TREE_MAX_LEVELS] & 2u ? "a grid cell lies outside the key frame derived from the cells' bounding box"
                                                                              : "a bounded wait of a compaction kernel ran out");
    }
    if (lv[0] == 0u || lv[0] > n_runs) return c->fail(LIO_ERR_HIP, "index rebuild", "the run kernels left an impossible number of cells");
    return finish_index_levels(c, n, lv[0], lv, cell_val);
}
// First half: run heads per tile of the n (live) points
static int build_index_by_cells(lio_ctx* c, const float4* src, uint32_t n, const uint32_t* live_dev, uint32_t live_expected, uint32_t* live_found) {
    hipStream_t s = c->stream;
    const uint32_t n_tiles = (n + RUN_TILE - 1u) / RUN_TILE;
    uint32_t* tile_cnt = reinterpret_cast<uint32_t*>(c->d_temp);                 // (the sort workspace is idle until sort_runs)
    uint32_t* tile_off = tile_cnt + (n_tiles + 4u);
    uint32_t* tile_box = tile_off + (n_tiles + 4u);
    HIPCHK(c, launch_cell_runs_count(src, n, c->grid, tile_cnt, tile_box, s));
    HIPCHK(c, launch_exclusive_scan(tile_cnt, n_tiles + 1u, tile_off, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, s));
    // the number of runs (the scan's total) and the cells' box are known from here: they travel to the host BESIDE the emit kernel
    const uint32_t seq = ++c->small_seq;
    HIPCHK(c, launch_cell_box_publish(tile_box, n_tiles, tile_off + n_tiles, live_dev, c->d_small, seq, s));
    HIPCHK(c, launch_cell_runs_emit(src, n, c->grid, tile_off, c->d_mkeys_tmp, c->d_vals_tmp, s));
    HIPCHK(c, wait_published(c, seq));
    uint32_t w[8];
    std::memcpy(w, c->h_small, sizeof(w));
    if (live_dev != nullptr && w[7] != live_expected) { *live_found = w[7]; return LIO_RETRY_COUNT; }
    return build_index_from_runs(c, src, n, w[0], w + 1);
}
// (re)build the index from n points at src (device, float4, must not alias c->d_pts)
// live_dev (optional): a device word that must equal `live_expected` -- read with the rebuild's first wait; LIO_RETRY_COUNT and
// *live_found when it does not (nothing of the index has been touched then)
int build_index(lio_ctx* c, const float4* src, uint32_t n, const uint32_t* live_dev, uint32_t live_expected, uint32_t* live_found) {
    hipStream_t s = c->stream;
    c->have_map = false;
    c->hint_valid = false;                     // neighbours of an older map state no longer bound a search
    if (n == 0) {
        c->n_points = 0; c->n_live = 0; c->n_cells = 0;
        c->n_retired = 0; c->n_dead = 0; c->n_new_cells = 0; c->n_cells_built = 0;
        c->grid.n_points = 0;
        // empty map: an all-empty level-1 table keeps the kernels well defined
        HIPCHK(c, hipMemsetAsync(c->d_ltable, 0xFF, 64 * sizeof(HashEntry), s));
        HIPCHK(c, stream_wait(c));
        c->levels = TreeLevels{};                  // level-1 table = slot 0 of d_ltable, empty
        c->have_map = true; c->map_broken = false;
        return LIO_OK;
    }
    if (by_cell_runs(c, n)) return build_index_by_cells(c, src, n, live_dev, live_expected, live_found);
    return build_index_by_points(c, src, n);
}

// box-deleted points not handed out yet -> the tail of d_removed (they stay tomb-stones in d_pts, re-marked)
int collect_box_deleted(lio_ctx* c) {
    hipStream_t s = c->stream;
    if (c->box_deleted_pending == 0 || c->n_points == 0) { c->box_deleted_pending = 0; return LIO_OK; }
    const size_t need = (size_t)c->n_removed + (size_t)c->box_deleted_pending;
    if (need > c->cap_removed) {
        size_t cap = c->cap_removed ? c->cap_removed : 65536;
        while (cap < need) cap <<= 1;
        float4* nb = nullptr;
        HIPCHK(c, hipMalloc((void**)&nb, cap * sizeof(float4)));
        if (c->n_removed > 0)
            HIPCHK(c, hipMemcpyAsync(nb, c->d_removed, (size_t)c->n_removed * sizeof(float4), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, stream_wait(c));
        if (c->d_removed) (void)hipFree(c->d_removed);
        c->d_removed = nb; c->cap_removed = cap;
    }
    HIPCHK(c, select_box_deleted_points(c->d_temp, c->temp_bytes, c->d_pts, c->d_removed + c->n_removed, c->d_counters + 7, c->n_points, s));
    HIPCHK(c, launch_mark_reported(c->d_pts, c->n_points, s));
    uint32_t got = 0;
    HIPCHK(c, read_back(c, &got, c->d_counters + 7, sizeof(uint32_t)));
    c->n_removed += got;
    c->box_deleted_pending = 0;
    return LIO_OK;
}

// ---- Add_Points machinery (ikd_Tree.cpp:414-489) --------------------------------------------------
int apply_pending(lio_ctx* c);
static int deferred_settle_fwd(lio_ctx* c) { return deferred_settle(c, nullptr, nullptr); }
static int ensure_add_buffers(lio_ctx* c, size_t n) {
    if (n <= c->cap_add) return LIO_OK;
    // (grown by half at least: two dozen re-allocations are a millisecond, not something a growing batch should pay step by step)
    if (c->cap_add != 0 && n < c->cap_add + c->cap_add / 2) n = c->cap_add + c->cap_add / 2;
    if (c->def.pending) {                      // (its plan points into the buffers about to be replaced)
        const int rc = deferred_settle_fwd(c);
        if (rc != LIO_OK) return rc;
    }
    if (c->fold_n != 0 || c->plain_n != 0) {   // left behind by a call that failed half-way: index them first
        const int rc = apply_pending(c);
        if (rc != LIO_OK) return rc;
    }
    void* old[] = {c->d_add_pts, c->d_add_pts2, c->d_add_out, c->d_add_keys, c->d_add_keys2, c->d_add_order, c->d_add_order2,
                   c->d_add_cls, c->d_add_dirty, c->d_add_dirty_pt, c->d_add_dlist, c->d_add_plan, c->d_add_ranges, c->d_pend, c->d_pkeys, c->d_pkeys2, c->d_porder,
                   c->d_porder2, c->d_pnewoff, c->d_plan, c->d_gs_part, c->d_gs_part2, c->d_cagg};
    for (void* p : old) if (p) (void)hipFree(p);
    c->d_add_pts = c->d_add_pts2 = c->d_add_out = nullptr; c->d_add_keys = c->d_add_keys2 = nullptr;
    c->d_add_order = c->d_add_order2 = nullptr; c->d_add_cls = nullptr; c->d_add_dirty = c->d_add_dirty_pt = nullptr; c->d_add_dlist = nullptr;
    c->d_add_plan = nullptr; c->d_add_ranges = nullptr;
    c->cap_add = 0;
    c->d_pend = nullptr; c->d_pkeys = c->d_pkeys2 = c->d_gs_part = c->d_gs_part2 = nullptr;
    c->d_porder = c->d_porder2 = c->d_pnewoff = nullptr; c->d_plan = nullptr; c->d_cagg = nullptr;
    c->cap_pend = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_add_pts, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_pts2, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_out, 2 * n * sizeof(float4)));      // [fold survivors by position | sequential replay]
    HIPCHK(c, hipMalloc((void**)&c->d_add_dirty, n));
    HIPCHK(c, hipMalloc((void**)&c->d_add_dirty_pt, n));
    HIPCHK(c, hipMalloc((void**)&c->d_add_dlist, n * sizeof(uint32_t)));
    if (!c->d_add_parts) HIPCHK(c, hipMalloc((void**)&c->d_add_parts, 16400 * sizeof(uint2)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_plan, 2 * n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_ranges, 8 * n * sizeof(uint2)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_keys, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_keys2, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_order, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_order2, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_add_cls, n));
    c->cap_add = n;
    // the pair Add_Points(add, true) + Add_Points(nodown, false) is applied together: two batches pending
    const size_t np = 2 * n + 2;
    HIPCHK(c, hipMalloc((void**)&c->d_pend, np * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_pkeys, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_pkeys2, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_gs_part, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_gs_part2, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_porder, np * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_porder2, np * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_pnewoff, np * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_plan, np * sizeof(CellPlan)));
    c->cap_pend = np;
    // stamped tile counts of the single-launch scans: one word per tile of 256 (the survivor slots are 2 n), two streams at most
    c->cagg_words = 2 * (np / 256 + 2);
    if (c->cagg_words < COMPACT_AGG_WORDS) c->cagg_words = COMPACT_AGG_WORDS;
    HIPCHK(c, hipMalloc((void**)&c->d_cagg, c->cagg_words * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->d_cagg, 0, c->cagg_words * sizeof(unsigned long long), c->stream));
    return LIO_OK;
}

// The pending list d_pend = [plain inserts][survivors of the fold and of the replay, compacted in order] with the (cell, list)
// key of every pending point, one launch (lio_compact.hip); the number of survivors stays on the device (d_apply[6]).
static int build_pending(lio_ctx* c, const KeyFrame& cf, SortKind kind) {
    hipStream_t s = c->stream;
    PendBuildArgs pb{};
    pb.surv = c->d_add_out; pb.n_surv_slots = 2u * c->fold_n; pb.plain = c->plain_ptr; pb.n_plain = c->plain_n;
    pb.pend = c->d_pend; pb.grid = c->grid; pb.frame = cf; pb.keys = c->d_pkeys; pb.order = c->d_porder;
    {
        const int rc = sort_hist(c, kind, &pb.hist);
        if (rc != LIO_OK) return rc;
    }
    // (the fold's two counters arrive as one pair per workgroup of fold_commit_kernel: this kernel's extra workgroup adds them up)
    pb.count_parts = c->fold_n != 0 ? c->d_add_parts : nullptr; pb.n_count_parts = c->fold_n != 0 ? fold_commit_workgroups(c->fold_n) : 0u;
    pb.fold_counts = c->d_apply + 4;
    pb.n_surv = c->d_apply + 6; pb.flags = c->d_apply + 14; pb.agg = c->d_cagg; pb.stamp = next_compact_stamp(c); pb.err = c->d_apply + 15;
    if (kind == SORT_LIBRARY)                   // the library sorts every slot: the unused ones must sort last
        HIPCHK(c, hipMemsetAsync(c->d_pkeys, 0xFF, ((size_t)c->plain_n + (size_t)c->fold_n) * sizeof(uint64_t), s));
    HIPCHK(c, launch_pend_build(pb, s));
    c->pend_built = true;
    return LIO_OK;
}

// the counts a down-sampled Add_Points left on the device: replaced map points, tmp_counter, survivors (d_apply[4..6]),
// with the flags of the kernels that ran (frame violations [14], compaction time-outs [15])
static int fetch_batch_counts(lio_ctx* c, uint32_t st[32]) {
    HIPCHK(c, read_back(c, st, c->d_apply, 32 * sizeof(uint32_t)));
    if (st[14] != 0u) {
        (void)hipMemsetAsync(c->d_apply + 14, 0, sizeof(uint32_t), c->stream);
        return c->fail(LIO_ERR_HIP, "Add_Points", "a batch point lies outside the key frame derived from the batch's bounding box");
    }
    return check_compact_error(c, st[15]);
}

// Full rebuild: the live points in array order (a bucket is contiguous, so the order inside a cell is kept),
// then the pending points, re-sorted and re-hashed.  Run when the tail is full, when the retired slots or
// the new cells have grown past their budget, or when a new cell lies outside the root of the tree.
int rebuild_full(lio_ctx* c) {
    hipStream_t s = c->stream;
    uint32_t n_pend = 0;
    if (c->fold_n != 0 || c->plain_n != 0) {
        // the pending points as one list (the frame is only needed for keys nobody will look at: any frame that holds them)
        if (!c->pend_built) {
            const int rc = build_pending(c, cell_frame(c->batch_box, c->cfg.filter_size_map_min, c->grid.inv_cell), SORT_RANK);
            if (rc != LIO_OK) return rc;
        }
        uint32_t st[32] = {};
        const int rc = fetch_batch_counts(c, st);
        if (rc != LIO_OK) return rc;
        if (c->fold_n != 0) { c->n_live -= st[4]; c->last_add_counter = (int)st[5]; c->n_boundary_points += st[10]; }
        n_pend = c->plain_n + (c->fold_n != 0 ? st[6] : 0u);
        c->fold_n = 0; c->plain_n = 0; c->plain_ptr = nullptr; c->pend_built = false; c->batch_box = HostBox{};
    }
    c->have_map = false;
    c->hint_valid = false;
    const uint32_t n_all = c->n_points;
    // Points_deleted (ikd_Tree.cpp:1338-1341): box-deleted points leave the storage in this rebuild
    {
        const int rc = collect_box_deleted(c);
        if (rc != LIO_OK) return rc;
    }
    // The host keeps count of the live points (n_live): the rebuild by cell runs goes ahead on that number and has the compaction's
    // own count delivered with its first wait; should the two ever differ, it starts over from the device's.
    uint32_t live = 0;
    bool counted = true;
    if (n_all > 0) {
        HIPCHK(c, compact_live_any(c, c->d_pts, n_all, c->d_pts_tmp, c->d_counters + 3));
        counted = !by_cell_runs(c, c->n_live + n_pend) || c->n_live == 0u || c->n_live > n_all;
        if (counted) HIPCHK(c, read_back(c, &live, c->d_counters + 3, sizeof(uint32_t)));
        else live = c->n_live + (c->test_live_miscount ? 1u : 0u);   // (LIO_TEST_LIVE_MISCOUNT=1: the tests' way into the restart below)
    }
    for (;;) {
        if ((size_t)live + (size_t)n_pend > c->cap_map)
            return c->fail(LIO_ERR_CAPACITY, "Add_Points", "live points + batch exceed max_map_points");
        if (n_pend > 0)
            HIPCHK(c, hipMemcpyAsync(c->d_pts_tmp + live, c->d_pend, (size_t)n_pend * sizeof(float4), hipMemcpyDeviceToDevice, s));
        uint32_t live_dev = live;
        const int rc = build_index(c, c->d_pts_tmp, live + n_pend, counted ? nullptr : c->d_counters + 3, live, &live_dev);
        if (rc != LIO_RETRY_COUNT) return rc;
        if (c->trace_rebuilds) std::fprintf(stderr, "lio: full rebuild: the host counted %u live points, the compaction %u\n", live, live_dev);
        live = live_dev; counted = true;
    }
}

// how many retired slots (copies of re-written buckets) the map may carry before a full rebuild compacts them away
}  // namespace
extern "C" uint32_t retired_slot_budget(const lio_ctx* c) {
    if (c->dbg_retired_budget > 0) return (uint32_t)c->dbg_retired_budget;
    const uint64_t b = RETIRED_PER_LIVE * (uint64_t)c->n_live;
    return (uint32_t)(b > 1048576ull ? (b < 0xfffffff0ull ? b : 0xfffffff0ull) : 1048576ull);
}
namespace {

// The pending points enter the index: O(batch), the rest of the map is not touched (lio_mapadd.hip).
// Launches: pend_build, the sort by (cell, list) (1 or 2), cell_plan (with the scan of the new bucket sizes), cell_apply,
// and ONE read-back that brings the statistics of this update together with the counts the fold left on the device.
// First half: the launches (on c->stream).  *launched = false: there was nothing to do, or the full rebuild did it.
static int apply_pending_enqueue(lio_ctx* c, bool* launched, uint32_t* publish_dst = nullptr, uint32_t publish_seq = 0u) {
    hipStream_t s = c->stream;
    *launched = false;
    if (c->map_broken) return c->fail(LIO_ERR_NO_MAP, "Add_Points", "the map index is undefined after a failed update: Build / reconstruct it");
    if (c->fold_n == 0 && c->plain_n == 0) return LIO_OK;
    c->hint_valid = false;
    if (!c->have_map || c->n_live == 0 || c->n_points == 0) return rebuild_full(c);
    const KeyFrame cf = cell_frame(c->batch_box, c->cfg.filter_size_map_min, c->grid.inv_cell);
    const int kb = frame_bits(cf) + 1;                            // + the bit that orders the two lists inside a cell
    if (kb > 63) return c->fail(LIO_ERR_CAPACITY, "Add_Points", "the batch spans more grid cells than a 63-bit key can number");
    const uint32_t n_bound = c->plain_n + c->fold_n;              // (a batch leaves at most as many survivors as it has points)
    const SortKind kind = sort_kind(c, n_bound, kb);
    int rc = build_pending(c, cf, kind);
    if (rc != LIO_OK) return rc;
    const uint32_t* n_dev = c->fold_n != 0 ? c->d_apply + 6 : nullptr;
    rc = run_sort(c, kind, c->d_pkeys, c->d_porder, c->d_pkeys2, c->d_porder2, n_bound, n_dev, c->plain_n, kb);
    if (rc != LIO_OK) return rc;
    ApplyArgs a{};
    a.map_pts = c->d_pts; a.grid = c->grid; a.ltable = c->d_ltable;
    a.child = c->d_child; a.block_bump = c->d_apply + 9; a.blocks_cap = (uint32_t)c->child_blocks_cap;
    a.levels = c->levels;
    a.pend = c->d_pend;
    a.view.skey = c->d_pkeys2; a.view.sord = c->d_porder2; a.view.n_dev = n_dev; a.view.n_add = c->plain_n; a.view.n = n_bound; a.view.shift = 1;
    a.frame = cf; a.plan = c->d_plan; a.new_off = c->d_pnewoff; a.tail = c->n_points; a.stats = c->d_apply;
    a.slots_cap = (uint32_t)c->slots_cap;
    a.agg = c->d_cagg; a.stamp = next_compact_stamp(c); a.err = c->d_apply + 15;
    a.publish_dst = publish_dst; a.publish_seq = publish_seq; a.ticket = c->d_apply + 8;   // (the statistics leave with the apply kernel)
    HIPCHK(c, launch_cell_plan(a, s));
    HIPCHK(c, launch_cell_apply(a, s));
    *launched = true;
    return LIO_OK;
}
// the 32 statistics words the apply kernel published under `seq` (+ the flags of the kernels in front of it)
static int collect_batch_counts(lio_ctx* c, uint32_t seq, uint32_t st[32]) {
    HIPCHK(c, wait_published(c, seq));
    std::memcpy(st, c->h_small, 32 * sizeof(uint32_t));
    if (st[14] != 0u) {
        (void)hipMemsetAsync(c->d_apply + 14, 0, sizeof(uint32_t), c->stream);
        return c->fail(LIO_ERR_HIP, "Add_Points", "a batch point lies outside the key frame derived from the batch's bounding box");
    }
    return check_compact_error(c, st[15]);
}
// Second half: the 32 statistics words of the update (st; rc = how reading them went) -> the host's view of the map, the budgets,
// a full rebuild when one is due.
static int apply_pending_finish(lio_ctx* c, const uint32_t st[32], int rc) {
    if (rc != LIO_OK) {
        // cell_apply may already have re-written buckets at the tail and retired their old slots: the counts that say how far
        // it got are what could not be read.  The batch is dropped and the index declared undefined -- loudly, for every call
        // that would search or extend it -- instead of being re-applied over a half-updated state by the next call.
        c->fold_n = 0; c->plain_n = 0; c->plain_ptr = nullptr; c->pend_built = false; c->batch_box = HostBox{};
        c->map_broken = true; c->have_map = false; c->hint_valid = false;
        return rc;
    }
    if (st[2] & APPLY_FLAG_NO_ROOM) return rebuild_full(c);    // nothing was changed: the pending points go in there
    // the counts of the down-sampled Add_Points arrive with this read-back (no host round trip of their own)
    const uint32_t n_real = c->plain_n + (c->fold_n != 0 ? st[6] : 0u);
    if (c->trace_rebuilds && c->fold_n != 0 && st[10] > 32u)
        std::fprintf(stderr, "lio: index update %llu: %u points folded, %u of them replayed one by one (voxel-boundary groups), %u survivors\n",
                     (unsigned long long)c->n_incremental_updates + 1ull, c->fold_n, st[10], st[6]);
    if (c->fold_n != 0) { c->n_live -= st[4]; c->last_add_counter = (int)st[5]; c->n_boundary_points += st[10]; }
    c->fold_n = 0; c->plain_n = 0; c->plain_ptr = nullptr; c->pend_built = false; c->batch_box = HostBox{};
    const uint32_t total = st[3];
    c->n_points += total;
    c->grid.n_points = c->n_points;
    c->n_live += n_real;
    c->n_cells += st[0]; c->n_new_cells += st[0]; c->n_retired += st[1];
    ++c->n_incremental_updates;
    // Retired copies of re-written buckets (TOMB_MOVED) cost memory and nothing else -- no search ever reads them, and 16 bytes
    // a slot is cheap on a 288 GB part: they may grow to RETIRED_PER_LIVE times the live points (the array is sized for that; a
    // full tail asks for the rebuild by itself, APPLY_FLAG_NO_ROOM).  Box-deleted points sit INSIDE live buckets and are
    // streamed by every search that passes: their budget stays tight.
    const uint32_t retired_budget = retired_slot_budget(c);
    const uint32_t dead_budget = c->n_live / 8 > 65536u ? c->n_live / 8 : 65536u;
    // Keep every hashed level at most half full (they are built a quarter full; the apply kernel counts the entries a
    // level gains), so that probe sequences stay short and a look-up of an absent key always meets an empty slot --
    // small maps have 64-slot tables that the cell budget below would let run full.
    bool table_crowded = false;
    for (int L = 1; L < c->levels.top_level; ++L)
        table_crowded = table_crowded || ((size_t)c->level_entries[L] + (size_t)st[APPLY_LEVEL_NEW + L]) * 2 > (size_t)c->levels.mask[L] + 1;
    if (st[2] != 0 || table_crowded || (size_t)c->n_retired > (size_t)retired_budget || (size_t)c->n_dead > (size_t)dead_budget ||
        (size_t)c->n_new_cells * 4 > (size_t)c->n_cells_built + 1024) {
        if (c->trace_rebuilds)
            std::fprintf(stderr, "lio: full rebuild after %llu incremental updates: flags %u crowded %d retired %u/%u dead %u/%u new cells %u of %u built, live %u slots %u\n",
                         (unsigned long long)c->n_incremental_updates, st[2], (int)table_crowded, c->n_retired, retired_budget, c->n_dead, dead_budget,
                         c->n_new_cells, c->n_cells_built, c->n_live, c->n_points);
        return rebuild_full(c);
    }
    return LIO_OK;
}
int apply_pending(lio_ctx* c) {
    bool launched = false;
    if (!c->dbg_publish_in_kernel) {            // (LIO_PUBLISH_IN_KERNEL=0, A/B runs of scripts: a publish launch behind the apply kernel, as until round 5)
        int rc0 = apply_pending_enqueue(c, &launched);
        if (rc0 != LIO_OK || !launched) return rc0;
        uint32_t st0[32] = {};
        rc0 = fetch_batch_counts(c, st0);
        return apply_pending_finish(c, st0, rc0);
    }
    const uint32_t seq = ++c->small_seq;
    int rc = apply_pending_enqueue(c, &launched, c->d_small, seq);
    if (rc != LIO_OK || !launched) return rc;
    uint32_t st[32] = {};
    rc = collect_batch_counts(c, seq, st);
    return apply_pending_finish(c, st, rc);
}

}  // namespace

extern "C" {
// ---- map ---------------------------------------------------------------------------------------
int lio_map_build(lio_ctx* c, const float* pts, long long n, int stride_bytes) {
    if (!c || n < 0 || (n > 0 && !pts) || stride_bytes < 12) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if ((size_t)n > c->cap_map) return c->fail(LIO_ERR_CAPACITY, "lio_map_build", "n exceeds max_map_points");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_points(c, pts, (size_t)n, stride_bytes, c->d_pts_tmp);
    if (rc != LIO_OK) return rc;
    // Build / reconstruct drop the old nodes (ikd_Tree.cpp:353-358, 1396) but Points_deleted survives them: what the
    // reference may still hand out afterwards is a subset of the box-deleted points, so they are kept for
    // acquire_removed_points (every box-deleted point is reported exactly once, lio_hip.h)
    rc = collect_box_deleted(c);
    if (rc != LIO_OK) return rc;
    return build_index(c, c->d_pts_tmp, (uint32_t)n);
}
int lio_map_build_device(lio_ctx* c, const void* d_pts_f4, long long n) {
    if (!c || n < 0 || (n > 0 && !d_pts_f4)) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if ((size_t)n > c->cap_map) return c->fail(LIO_ERR_CAPACITY, "lio_map_build_device", "n exceeds max_map_points");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0)
        HIPCHK(c, hipMemcpyAsync(c->d_pts_tmp, d_pts_f4, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    {
        const int rc = collect_box_deleted(c);
        if (rc != LIO_OK) return rc;
    }
    return build_index(c, c->d_pts_tmp, (uint32_t)n);
}
int lio_map_reconstruct(lio_ctx* c, const float* pts, long long n, int stride_bytes) {
    if (c) enter(c);
    return lio_map_build(c, pts, n, stride_bytes);       // ikd_Tree.cpp:1394-1405: delete all, Build
}

// One Add_Points(.., true) call on a device-resident batch (float4, batch order = insertion order), WITHOUT touching the
// index: the batch is grouped by down-sample voxel, every group folds against the indexed points (tomb-stoning what it
// replaces) and leaves its survivor in d_add_out; the counts stay on the device until the index update's read-back.
// At most one down-sampled batch per index update (pending points are not in the index yet).
// First half: everything that only READS the map -- voxel keys, the grouping sort, boundary classification, the speculative
// fold (fold_plan_kernel), the list of dirty groups -- on the context's current stream (`c->stream`: the deferred schedule of
// the replica exchange points it at the side stream for the duration of this call).  fa / ba: what the second half needs.
static int fold_batch_plan(lio_ctx* c, const float4* batch, uint32_t n, const HostBox& box, FoldArgs& fa, BoundaryArgs& ba) {
    hipStream_t s = c->stream;
    if ((size_t)n > c->cap_add) return c->fail(LIO_ERR_CAPACITY, "Add_Points", "batch buffers too small");
    const float ds = c->cfg.filter_size_map_min;
    const KeyFrame vf = voxel_frame(box, ds);
    const int kb = frame_bits(vf);
    if (kb > 63) return c->fail(LIO_ERR_CAPACITY, "Add_Points", "the batch spans more down-sample voxels than a 63-bit key can number");
    const SortKind kind = sort_kind(c, n, kb);
    VoxelKeyArgs vk{};
    vk.pts = batch; vk.n = n; vk.ds = ds; vk.frame = vf; vk.keys = c->d_add_keys; vk.order = c->d_add_order;
    {
        const int rc = sort_hist(c, kind, &vk.hist);
        if (rc != LIO_OK) return rc;
    }
    vk.zero_counters = c->d_apply + 4; vk.dirty = c->d_add_dirty; vk.dirty_pt = c->d_add_dirty_pt; vk.out2n = c->d_add_out;
    vk.bbox = c->d_counters + 24; vk.flags = c->d_apply + 14;
    HIPCHK(c, launch_voxel_keys(vk, s));
    int rc = run_sort(c, kind, c->d_add_keys, c->d_add_order, c->d_add_keys2, c->d_add_order2, n, nullptr, 0, kb);
    if (rc != LIO_OK) return rc;
    GroupView view{};
    view.skey = c->d_add_keys2; view.sord = c->d_add_order2; view.n_dev = nullptr; view.n_add = 0; view.n = n; view.shift = 0;
    // boundary points (outside their own fp32 box / inside a neighbour's): their groups leave the parallel fold
    // and are replayed point by point
    ba = BoundaryArgs{};
    ba.map_pts = c->d_pts; ba.l1table = c->d_ltable + c->levels.off[1]; ba.l1mask = c->levels.mask[1]; ba.child = c->d_child;
    ba.grid = c->grid; ba.ds = ds; ba.new_pts = batch; ba.view = view; ba.frame = vf;
    ba.dirty = c->d_add_dirty; ba.dirty_pt = c->d_add_dirty_pt; ba.dlist = c->d_add_dlist; ba.n_dirty = c->d_apply + 10;
    ba.seq_out = c->d_add_out + n; ba.n_added = c->d_apply + 5; ba.n_replaced = c->d_apply + 4;
    fa = FoldArgs{};
    fa.map_pts = c->d_pts; fa.l1table = ba.l1table; fa.l1mask = ba.l1mask; fa.child = c->d_child;
    fa.grid = c->grid; fa.ds = ds; fa.new_pts = batch; fa.view = view;
    fa.out_pts = c->d_add_out; fa.n_added = c->d_apply + 5; fa.n_replaced = c->d_apply + 4; fa.dirty = c->d_add_dirty;
    fa.plan = c->d_add_plan; fa.ranges = c->d_add_ranges; fa.count_parts = c->d_add_parts;
    ba.list_here = n <= DIRTY_LIST_HERE_MAX;  // (small batches: the replaying wavefront of fold_commit_kernel lists the dirty points itself)
    HIPCHK(c, launch_fold_plan(fa, ba, s));
    if (!ba.list_here) HIPCHK(c, launch_compact_dirty(ba, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, s));
    return LIO_OK;
}
// Second half: the plans of the clean groups are carried out, the dirty groups replayed (tomb-stones in the map, survivors in
// d_add_out); the batch is then pending for the index update.  The map must not have changed since the plan.
static int fold_batch_commit(lio_ctx* c, const FoldArgs& fa, const BoundaryArgs& ba, uint32_t n, const HostBox& box) {
    HIPCHK(c, launch_fold_commit(fa, ba, c->stream));
    c->fold_n = n;
    c->batch_box.merge(box);
    return LIO_OK;
}
static int fold_batch_device(lio_ctx* c, const float4* batch, uint32_t n, const HostBox& box) {
    if (n == 0) return LIO_OK;
    if (c->fold_n != 0) {
        const int rc = apply_pending(c);
        if (rc != LIO_OK) return rc;
    }
    if ((size_t)c->n_live + (size_t)c->plain_n + (size_t)n > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "Add_Points", "live points + batch exceed max_map_points");
    c->last_add_counter = 0;
    FoldArgs fa{};
    BoundaryArgs ba{};
    const int rc = fold_batch_plan(c, batch, n, box, fa, ba);
    if (rc != LIO_OK) return rc;
    return fold_batch_commit(c, fa, ba, n, box);
}
// One Add_Points(.., false) call: the points are simply pending (they must stay where they are until the index update)
static int plain_batch_device(lio_ctx* c, const float4* batch, uint32_t n, const HostBox& box) {
    if (n == 0) return LIO_OK;
    if (c->plain_n != 0) {
        const int rc = apply_pending(c);
        if (rc != LIO_OK) return rc;
    }
    if ((size_t)c->n_live + (size_t)c->fold_n + (size_t)n > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "Add_Points", "live points + batch exceed max_map_points");
    if ((size_t)c->fold_n + (size_t)n > c->cap_pend) return c->fail(LIO_ERR_CAPACITY, "Add_Points", "pending buffer too small");
    c->plain_ptr = batch; c->plain_n = n;
    c->batch_box.merge(box);
    return LIO_OK;
}

// the batches enter the index (incrementally, or through a full rebuild when budgets are exceeded)
// (*add_counter: tmp_counter of the down-sampled batch of this call: it arrives with the index update's read-back)
static int rebuild_from_stored(lio_ctx* c, int* add_counter = nullptr) {
    const int rc = apply_pending(c);
    if (rc == LIO_OK && add_counter) *add_counter = c->last_add_counter;
    return rc;
}

// Add_Points (ikd_Tree.cpp:414-489).  Returns through *n_added what the reference returns: the number
// of down-sampled inserts that changed the tree (tmp_counter), 0 when downsample_on is false.
int lio_map_add(lio_ctx* c, const float* pts, int n, int stride_bytes, int downsample_on, int* n_added) {
    if (!c || n < 0 || (n > 0 && !pts) || stride_bytes < 12) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (n_added) *n_added = 0;
    if (n == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)c->n_live + (size_t)n > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "lio_map_add", "live points + batch exceed max_map_points");
    int rc = ensure_add_buffers(c, (size_t)n);
    if (rc != LIO_OK) return rc;
    rc = upload_points(c, pts, (size_t)n, stride_bytes, c->d_add_pts);
    if (rc != LIO_OK) return rc;
    HostBox box;
    box_of_host_points(box, pts, (size_t)n, stride_bytes);
    c->last_add_counter = 0;
    rc = downsample_on ? fold_batch_device(c, c->d_add_pts, (uint32_t)n, box) : plain_batch_device(c, c->d_add_pts, (uint32_t)n, box);
    if (rc != LIO_OK) return rc;
    return rebuild_from_stored(c, downsample_on ? n_added : nullptr);
}

// Add_Points(add, true) followed by Add_Points(nodown, false) -- the pair map_incremental issues
// (laserMapping.cpp:1492-1493) -- with ONE index rebuild.  Used to apply replicated insert lists.
int lio_map_add_pair(lio_ctx* c, const float* add_pts, int n_add, const float* nodown_pts, int n_nodown,
                     int stride_bytes, int* add_counter) {
    if (!c || n_add < 0 || n_nodown < 0 || (n_add > 0 && !add_pts) || (n_nodown > 0 && !nodown_pts) || stride_bytes < 12)
        return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (add_counter) *add_counter = 0;
    if (n_add + n_nodown == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)c->n_live + (size_t)n_add + (size_t)n_nodown > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "lio_map_add_pair", "live points + batches exceed max_map_points");
    int rc = ensure_add_buffers(c, (size_t)(n_add > n_nodown ? n_add : n_nodown));
    if (rc != LIO_OK) return rc;
    c->last_add_counter = 0;
    if (n_add > 0) {
        rc = upload_points(c, add_pts, (size_t)n_add, stride_bytes, c->d_add_pts);
        if (rc != LIO_OK) return rc;
        HostBox box;
        box_of_host_points(box, add_pts, (size_t)n_add, stride_bytes);
        rc = fold_batch_device(c, c->d_add_pts, (uint32_t)n_add, box);
        if (rc != LIO_OK) return rc;
    }
    if (n_nodown > 0) {
        rc = upload_points(c, nodown_pts, (size_t)n_nodown, stride_bytes, c->d_add_pts2);
        if (rc != LIO_OK) return rc;
        HostBox box;
        box_of_host_points(box, nodown_pts, (size_t)n_nodown, stride_bytes);
        rc = plain_batch_device(c, c->d_add_pts2, (uint32_t)n_nodown, box);
        if (rc != LIO_OK) return rc;
    }
    return rebuild_from_stored(c, add_counter);
}

// a bounded wait of map_incremental_lists_kernel ran out (d_counters[12] != 0): its counts are zero, its lists have holes
static int list_kernel_failed(lio_ctx* c) {
    (void)hipMemsetAsync(c->d_counters + 12, 0, sizeof(uint32_t), c->stream);
    return c->fail(LIO_ERR_HIP, "map_incremental", "a bounded wait of the list compaction kernel ran out");
}
// the block of device words a list-producing kernel leaves: d_counters[5..6] the two list lengths, [12] its error word,
// [24..29] the bounding box of both lists (order images)
struct ListWords { uint32_t n_add = 0, n_plain = 0; HostBox box; };
static int fetch_list_words(lio_ctx* c, ListWords& w) {
    uint32_t v[25] = {};                                        // d_counters[5 .. 29]
    HIPCHK(c, read_back(c, v, c->d_counters + 5, sizeof(v)));
    if (v[7] != 0u) return list_kernel_failed(c);
    w.n_add = v[0]; w.n_plain = v[1];
    box_from_images(v + 19, w.box);
    return LIO_OK;
}
// map_incremental's classification + the two order-preserving lists, left on the device
// (c->d_add_pts = PointToAdd, c->d_add_pts2 = PointNoNeedDownsample), their lengths and bounding box too.
static int map_incremental_lists(lio_ctx* c, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                                 ListWords* words /* nullptr: everything stays on the device */) {
    if (words) *words = ListWords{};
    if (!c->have_scan) return c->fail(LIO_ERR_NO_SCAN, "lio_map_incremental", "no scan set");
    const int nq = c->nq;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (nq == 0) {
        if (!words) HIPCHK(c, hipMemsetAsync(c->d_counters + 5, 0, 2 * sizeof(uint32_t), s));
        return LIO_OK;
    }
    int rc = ensure_add_buffers(c, (size_t)nq);
    if (rc != LIO_OK) return rc;
    if (!c->have_search) {
        // no neighbour lists at all (first scans): every point is a PointToAdd (:1487)
        HIPCHK(c, hipMemsetAsync(c->d_cnt, 0, (size_t)nq, s));
    }
    IncrArgs ia{};
    ia.body = c->d_body; ia.nq = nq; ia.nq_stride = c->cap_q;
    for (int i = 0; i < 14; ++i) ia.pose.v[i] = state14[i];
    ia.nbr = c->d_nbr; ia.cnt = c->d_cnt; ia.filter_size_map_min = filter_size_map_min;
    ia.flg_EKF_inited = flg_EKF_inited; ia.world = c->d_world; ia.cls = c->d_add_cls;
    if (compact_small(c, (size_t)nq)) {
        // classification + both lists + their counts + their bounding box in ONE launch
        // (its error word is d_counters[12]: it travels with every read-back of the two counts, here and in the exchange)
        unsigned long long* agg = c->d_cagg;
        uint32_t stamp;
        if (c->settle_side) {                       // (the commit on the side stream runs its scans on d_cagg right now)
            agg = c->d_cagg_lists;
            if (++c->compact_seq_lists == 0u) {
                HIPCHK(c, hipMemsetAsync(c->d_cagg_lists, 0, COMPACT_AGG_WORDS * sizeof(unsigned long long), s));
                ++c->compact_seq_lists;
            }
            stamp = c->compact_seq_lists;
        } else {
            stamp = next_compact_stamp(c);
        }
        HIPCHK(c, launch_map_incremental_lists(ia, c->d_add_pts, c->d_add_pts2, c->d_counters + 5, c->d_counters + 24, agg,
                                               stamp, c->d_counters + 12, s));
    } else {
        HIPCHK(c, launch_map_incremental(ia, s));
        HIPCHK(c, select_points_by_class(c->d_temp, c->temp_bytes, c->d_world, c->d_add_cls, 1, c->d_add_pts, c->d_counters + 5, (size_t)nq, s));
        HIPCHK(c, select_points_by_class(c->d_temp, c->temp_bytes, c->d_world, c->d_add_cls, 2, c->d_add_pts2, c->d_counters + 6, (size_t)nq, s));
        HIPCHK(c, launch_bbox(c->d_world, (uint32_t)nq, c->d_counters + 24, s));    // (of every scan point: a superset, which is all a frame needs)
    }
    if (words) return fetch_list_words(c, *words);
    return LIO_OK;
}

// ---- round 6: map_incremental + both Add_Points + the index update with NO host round trip in between ----------------------
// The list kernel leaves the list lengths and both key frames on the device (BatchDesc, d_apply[32..]); every kernel behind it takes
// its real size and its frame from there; the one read-back at the end brings the statistics of the index update together with the
// two list lengths.  Same kernels, same values, same order as the path with the read-back in the middle (fold_batch_device /
// apply_pending): the frames are derived by the same functions from the same box.
// What the host cannot know is how LARGE to launch.  The scan size is a bound, but a scan inserts a twelfth of its points and a
// launch pays for the workgroups it does not need (measured, profiles/r06b_*: with every kernel sized for the 100 k points of the
// scan the update was 18 us slower than with the round trip it saved).  So the launches are sized by the lists of the last eight
// updates (their maximum + 25 % + 256: consecutive scans of a drive insert alike); the list kernel itself checks the lists against these bounds and, if
// one is longer, leaves a descriptor that makes every kernel behind it a no-op -- the host then finds the lists intact and takes
// its own path with them (one wasted chain of empty launches; `fast_update_overflows`).
static bool fast_update_possible(const lio_ctx* c) {
    const size_t nq = (size_t)c->nq;
    return c->dbg_fast_update && c->have_scan && nq > 0 && compact_small(c, nq) && !c->map_broken && c->have_map && c->n_live != 0 && c->n_points != 0 &&
           c->fold_n == 0 && c->plain_n == 0 && !c->def.pending && !c->settle_side && c->pred_lists_valid &&
           (size_t)c->n_live + nq <= c->cap_map;
}
static uint32_t list_bound(uint32_t last, uint32_t nq) {
    const uint64_t b = ((uint64_t)last + (uint64_t)last / 4u + 256u + 255u) / 256u * 256u;
    return (uint32_t)(b < (uint64_t)nq ? b : (uint64_t)nq);
}
static int map_incremental_fast(lio_ctx* c, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                                int* n_add_out, int* n_nodown_out, int* add_counter, bool* done) {
    *done = false;
    const uint32_t nq = (uint32_t)c->nq;
    hipStream_t s = c->stream;
    int rc = ensure_add_buffers(c, (size_t)nq);
    if (rc != LIO_OK) return rc;
    const uint32_t b_add = list_bound(c->pred_max(c->pred_add_ring), nq), b_plain = list_bound(c->pred_max(c->pred_plain_ring), nq);
    const uint32_t b_pend = b_add + b_plain;                      // (every survivor is one of the b_add points)
    if ((size_t)b_pend > c->cap_pend) return LIO_OK;
    const SortKind kind_v = sort_kind(c, b_add, 0), kind_c = sort_kind(c, b_pend, 0);
    if (kind_v == SORT_LIBRARY || kind_c == SORT_LIBRARY) return LIO_OK;
    BatchDesc* desc = reinterpret_cast<BatchDesc*>(c->d_apply + 32);
    const float ds = c->cfg.filter_size_map_min;
    if (!c->have_search) HIPCHK(c, hipMemsetAsync(c->d_cnt, 0, (size_t)nq, s));   // no neighbour lists at all (first scans): every point is a PointToAdd (:1487)
    IncrArgs ia{};
    ia.body = c->d_body; ia.nq = (int)nq; ia.nq_stride = c->cap_q;
    for (int i = 0; i < 14; ++i) ia.pose.v[i] = state14[i];
    ia.nbr = c->d_nbr; ia.cnt = c->d_cnt; ia.filter_size_map_min = filter_size_map_min;
    ia.flg_EKF_inited = flg_EKF_inited; ia.world = c->d_world; ia.cls = c->d_add_cls;
    HIPCHK(c, launch_map_incremental_lists(ia, c->d_add_pts, c->d_add_pts2, c->d_counters + 5, c->d_counters + 24, c->d_cagg, next_compact_stamp(c),
                                           c->d_counters + 12, s, desc, ds, c->grid.inv_cell, BATCH_KB_MAX, b_add, b_plain));
    c->last_add_counter = 0;
    // ---- Add_Points(PointToAdd, true): plan + commit (fold_batch_plan / fold_batch_commit with device-resident n and frame)
    VoxelKeyArgs vk{};
    vk.pts = c->d_add_pts; vk.n = b_add; vk.desc = desc; vk.ds = ds; vk.keys = c->d_add_keys; vk.order = c->d_add_order;
    rc = sort_hist(c, kind_v, &vk.hist);
    if (rc != LIO_OK) return rc;
    vk.zero_counters = c->d_apply + 4; vk.dirty = c->d_add_dirty; vk.dirty_pt = c->d_add_dirty_pt; vk.out2n = c->d_add_out;
    vk.bbox = c->d_counters + 24; vk.flags = c->d_apply + 14;
    HIPCHK(c, launch_voxel_keys(vk, s));
    rc = run_sort(c, kind_v, c->d_add_keys, c->d_add_order, c->d_add_keys2, c->d_add_order2, b_add, &desc->n_add, 0, 0, &desc->kb_v);
    if (rc != LIO_OK) return rc;
    GroupView view{};
    view.skey = c->d_add_keys2; view.sord = c->d_add_order2; view.n_dev = &desc->n_add; view.n_add = 0; view.n = b_add; view.shift = 0;
    BoundaryArgs ba{};
    ba.map_pts = c->d_pts; ba.l1table = c->d_ltable + c->levels.off[1]; ba.l1mask = c->levels.mask[1]; ba.child = c->d_child;
    ba.grid = c->grid; ba.ds = ds; ba.new_pts = c->d_add_pts; ba.view = view; ba.frame_dev = &desc->vf;
    ba.dirty = c->d_add_dirty; ba.dirty_pt = c->d_add_dirty_pt; ba.dlist = c->d_add_dlist; ba.n_dirty = c->d_apply + 10;
    ba.seq_out = c->d_add_out + b_add; ba.n_added = c->d_apply + 5; ba.n_replaced = c->d_apply + 4;
    FoldArgs fa{};
    fa.map_pts = c->d_pts; fa.l1table = ba.l1table; fa.l1mask = ba.l1mask; fa.child = c->d_child;
    fa.grid = c->grid; fa.ds = ds; fa.new_pts = c->d_add_pts; fa.view = view;
    fa.out_pts = c->d_add_out; fa.n_added = c->d_apply + 5; fa.n_replaced = c->d_apply + 4; fa.dirty = c->d_add_dirty;
    fa.plan = c->d_add_plan; fa.ranges = c->d_add_ranges; fa.count_parts = c->d_add_parts;
    ba.list_here = b_add <= DIRTY_LIST_HERE_MAX;
    HIPCHK(c, launch_fold_plan(fa, ba, s));
    if (!ba.list_here) HIPCHK(c, launch_compact_dirty(ba, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, s));
    HIPCHK(c, launch_fold_commit(fa, ba, s));
    // ---- the index update (build_pending / apply_pending_enqueue with device-resident sizes and frame)
    PendBuildArgs pb{};
    pb.surv = c->d_add_out; pb.n_surv_slots = 2u * b_add; pb.plain = c->d_add_pts2; pb.n_plain = b_plain; pb.desc = desc;
    pb.pend = c->d_pend; pb.grid = c->grid; pb.keys = c->d_pkeys; pb.order = c->d_porder;
    rc = sort_hist(c, kind_c, &pb.hist);
    if (rc != LIO_OK) return rc;
    pb.count_parts = c->d_add_parts; pb.n_count_parts = fold_commit_workgroups(b_add);
    pb.fold_counts = c->d_apply + 4;
    pb.n_surv = c->d_apply + 6; pb.n_pend = c->d_apply + 7; pb.desc_copy = c->d_apply + 11;
    pb.flags = c->d_apply + 14; pb.agg = c->d_cagg; pb.stamp = next_compact_stamp(c); pb.err = c->d_apply + 15;
    HIPCHK(c, launch_pend_build(pb, s));
    rc = run_sort(c, kind_c, c->d_pkeys, c->d_porder, c->d_pkeys2, c->d_porder2, b_pend, c->d_apply + 7, 0, 0, &desc->kb_c);
    if (rc != LIO_OK) return rc;
    ApplyArgs a{};
    a.map_pts = c->d_pts; a.grid = c->grid; a.ltable = c->d_ltable;
    a.child = c->d_child; a.block_bump = c->d_apply + 9; a.blocks_cap = (uint32_t)c->child_blocks_cap;
    a.levels = c->levels;
    a.pend = c->d_pend;
    a.view.skey = c->d_pkeys2; a.view.sord = c->d_porder2; a.view.n_dev = c->d_apply + 7; a.view.n_add = 0; a.view.n = b_pend; a.view.shift = 1;
    a.frame_dev = &desc->cf; a.plan = c->d_plan; a.new_off = c->d_pnewoff; a.tail = c->n_points; a.stats = c->d_apply;
    a.slots_cap = (uint32_t)c->slots_cap;
    a.agg = c->d_cagg; a.stamp = next_compact_stamp(c); a.err = c->d_apply + 15;
    const uint32_t seq = ++c->small_seq;
    a.publish_dst = c->d_small; a.publish_seq = seq; a.ticket = c->d_apply + 8;
    HIPCHK(c, launch_cell_plan(a, s));
    HIPCHK(c, launch_cell_apply(a, s));
    // ---- the ONE wait: statistics of the index update, counts of the fold, lengths of the two lists -- published by the apply
    // kernel's last workgroup, no launch behind it
    uint32_t st[32] = {};
    const hipError_t he = wait_published(c, seq);
    std::memcpy(st, c->h_small, sizeof(st));
    if (he != hipSuccess) {
        // cell_apply may have re-written buckets: the index is undefined (as apply_pending_finish says of its own failed read-back)
        c->map_broken = true; c->have_map = false; c->hint_valid = false;
        return c->fail(LIO_ERR_HIP, "lio_map_incremental", hipGetErrorString(he));
    }
    if (st[13] & 1u) return list_kernel_failed(c);               // (lists with holes, counts zero: every kernel behind it ran on nothing)
    if (st[13] & 6u) {                                            // a frame too wide for the group sort, or a list longer than its bound:
        if (st[13] & 4u) ++c->n_fast_update_overflows;            // nothing ran; *done stays false, the lists wait on the device
        return LIO_OK;
    }
    *done = true;
    ++c->n_fast_updates;
    const uint32_t n_add = st[11], n_plain = st[12];
    c->pred_record(n_add, n_plain);
    if (n_add_out) *n_add_out = (int)n_add;
    if (n_nodown_out) *n_nodown_out = (int)n_plain;
    if (n_add + n_plain == 0u) return LIO_OK;                     // (an update of nothing: the map is as it was)
    c->hint_valid = false;
    rc = LIO_OK;
    if (st[14] != 0u) {
        (void)hipMemsetAsync(c->d_apply + 14, 0, sizeof(uint32_t), s);
        rc = c->fail(LIO_ERR_HIP, "Add_Points", "a batch point lies outside the key frame derived from the batch's bounding box");
    } else {
        rc = check_compact_error(c, st[15]);
    }
    // the host's view of what has just happened: a down-sampled batch of n_add points and a plain one of n_plain, pending list built
    c->fold_n = n_add; c->plain_n = n_plain; c->plain_ptr = c->d_add_pts2; c->pend_built = true;
    rc = apply_pending_finish(c, st, rc);
    if (rc == LIO_OK && add_counter) *add_counter = c->last_add_counter;
    return rc;
}

// map_incremental (src/laserMapping.cpp:1440-1496) on the scan of the last measurement passes.
int lio_map_incremental(lio_ctx* c, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                        int apply, float* out_add_xyz, int* n_add, float* out_nodown_xyz, int* n_nodown,
                        int* add_counter) {
    if (!c || !state14 || !(filter_size_map_min > 0.0)) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (n_add) *n_add = 0;
    if (n_nodown) *n_nodown = 0;
    if (add_counter) *add_counter = 0;
    HIPCHK(c, hipSetDevice(c->device));
    ListWords w;
    int rc;
    bool lists_made = false;
    if (apply && !out_add_xyz && !out_nodown_xyz && fast_update_possible(c)) {
        // the node's own call (laserMapping.cpp:2401): nobody wants the lists on the host -- no round trip in the middle
        bool done = false;
        rc = map_incremental_fast(c, state14, filter_size_map_min, flg_EKF_inited, n_add, n_nodown, add_counter, &done);
        if (rc != LIO_OK || done) return rc;
        // (frames too wide for the group sort, or a list longer than predicted: nothing but the list kernel ran -- its lists, their
        // counts and their box are in place)
        rc = fetch_list_words(c, w);
        if (rc != LIO_OK) return rc;
        lists_made = true;
    }
    if (!lists_made) {
        rc = map_incremental_lists(c, state14, filter_size_map_min, flg_EKF_inited, &w);
        if (rc != LIO_OK) return rc;
    }
    hipStream_t s = c->stream;
    c->pred_record(w.n_add, w.n_plain);         // (what the next updates' launches are sized by)
    if (n_add) *n_add = (int)w.n_add;
    if (n_nodown) *n_nodown = (int)w.n_plain;
    // the lists, for the caller that replicates them to other GPUs
    for (int which = 0; which < 2; ++which) {
        float* dst = which == 0 ? out_add_xyz : out_nodown_xyz;
        const uint32_t m = which == 0 ? w.n_add : w.n_plain;
        if (!dst || m == 0) continue;
        rc = ensure_scratch(c, (size_t)m * 3 * sizeof(float));
        if (rc != LIO_OK) return rc;
        HIPCHK(c, launch_pack_xyz(which == 0 ? c->d_add_pts : c->d_add_pts2, m, (float*)c->d_scratch, s));
        HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCHK(c, stream_wait(c));
        std::memcpy(dst, c->h_scratch, (size_t)m * 3 * sizeof(float));
    }
    if (!apply || w.n_add == 0) HIPCHK(c, reset_list_box(c));     // (no voxel_keys_kernel will do it)
    if (!apply) return LIO_OK;
    // ikdtree.Add_Points(PointToAdd, true); ikdtree.Add_Points(PointNoNeedDownsample, false);   :1492-1493
    c->last_add_counter = 0;
    rc = fold_batch_device(c, c->d_add_pts, w.n_add, w.box);
    if (rc != LIO_OK) return rc;
    rc = plain_batch_device(c, c->d_add_pts2, w.n_plain, w.box);
    if (rc != LIO_OK) return rc;
    return rebuild_from_stored(c, add_counter);
}

// Device-resident variants for replicated maps: the lists stay in HBM (float4 records), are exchanged
// by the caller (RCCL all-gather over xGMI) and applied from device memory.
int lio_map_incremental_device(lio_ctx* c, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                               void** d_add_f4, int* n_add, void** d_nodown_f4, int* n_nodown) {
    if (!c || !state14 || !(filter_size_map_min > 0.0) || !d_add_f4 || !n_add || !d_nodown_f4 || !n_nodown)
        return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    ListWords w;
    int rc = map_incremental_lists(c, state14, filter_size_map_min, flg_EKF_inited, &w);
    if (rc != LIO_OK) return rc;
    *d_add_f4 = c->d_add_pts; *n_add = (int)w.n_add;
    *d_nodown_f4 = c->d_add_pts2; *n_nodown = (int)w.n_plain;
    HIPCHK(c, reset_list_box(c));
    return LIO_OK;
}
int lio_map_add_pair_device(lio_ctx* c, const void* d_add_f4, int n_add, const void* d_nodown_f4, int n_nodown,
                            int* add_counter) {
    if (!c || n_add < 0 || n_nodown < 0 || (n_add > 0 && !d_add_f4) || (n_nodown > 0 && !d_nodown_f4)) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (add_counter) *add_counter = 0;
    if (n_add + n_nodown == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)c->n_live + (size_t)n_add + (size_t)n_nodown > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "lio_map_add_pair_device", "live points + batches exceed max_map_points");
    // stage into the context's own batch buffers (the caller's may be ours)
    const size_t need = (size_t)(n_add > n_nodown ? n_add : n_nodown);
    if (need > c->cap_add) {
        // growing would free buffers the caller may be pointing into: copy out first
        float4 *ta = nullptr, *tb = nullptr;
        if (n_add) { HIPCHK(c, hipMalloc((void**)&ta, (size_t)n_add * sizeof(float4))); HIPCHK(c, hipMemcpy(ta, d_add_f4, (size_t)n_add * sizeof(float4), hipMemcpyDeviceToDevice)); }
        if (n_nodown) { HIPCHK(c, hipMalloc((void**)&tb, (size_t)n_nodown * sizeof(float4))); HIPCHK(c, hipMemcpy(tb, d_nodown_f4, (size_t)n_nodown * sizeof(float4), hipMemcpyDeviceToDevice)); }
        int rc0 = ensure_add_buffers(c, need);
        if (rc0 == LIO_OK && n_add) HIPCHK(c, hipMemcpy(c->d_add_pts, ta, (size_t)n_add * sizeof(float4), hipMemcpyDeviceToDevice));
        if (rc0 == LIO_OK && n_nodown) HIPCHK(c, hipMemcpy(c->d_add_pts2, tb, (size_t)n_nodown * sizeof(float4), hipMemcpyDeviceToDevice));
        if (ta) (void)hipFree(ta);
        if (tb) (void)hipFree(tb);
        if (rc0 != LIO_OK) return rc0;
    } else {
        hipStream_t s = c->stream;
        if (n_add && d_add_f4 != c->d_add_pts)
            HIPCHK(c, hipMemcpyAsync(c->d_add_pts, d_add_f4, (size_t)n_add * sizeof(float4), hipMemcpyDeviceToDevice, s));
        if (n_nodown && d_nodown_f4 != c->d_add_pts2)
            HIPCHK(c, hipMemcpyAsync(c->d_add_pts2, d_nodown_f4, (size_t)n_nodown * sizeof(float4), hipMemcpyDeviceToDevice, s));
    }
    // the bounding box of lists the host has never seen: one small kernel and one read-back (this entry point is not on
    // the per-scan path: lio_replicate_step carries the box in its own read-back)
    hipStream_t s = c->stream;
    HIPCHK(c, reset_list_box(c));
    if (n_add) HIPCHK(c, launch_bbox(c->d_add_pts, (uint32_t)n_add, c->d_counters + 24, s));
    if (n_nodown) HIPCHK(c, launch_bbox(c->d_add_pts2, (uint32_t)n_nodown, c->d_counters + 24, s));
    uint32_t img[6] = {};
    HIPCHK(c, read_back(c, img, c->d_counters + 24, sizeof(img)));
    HostBox box;
    box_from_images(img, box);
    c->last_add_counter = 0;
    int rc = fold_batch_device(c, c->d_add_pts, (uint32_t)n_add, box);
    if (rc != LIO_OK) return rc;
    rc = plain_batch_device(c, c->d_add_pts2, (uint32_t)n_nodown, box);
    if (rc != LIO_OK) return rc;
    return rebuild_from_stored(c, add_counter);
}

// ---- replica exchange behind the C ABI ------------------------------------------------
static int delete_boxes_resident(lio_ctx* c, int nbb, int* total, const float* d_boxes = nullptr);
static void account_deleted(lio_ctx* c, int total);
// RCCL is loaded at run time (dlopen): a single-GPU user of this library needs no librccl.  `backend "nccl" IS RCCL on
// ROCm`: a process that already carries one (PyTorch bundles its own librccl.so) shares it through the SONAME.
namespace {
typedef struct { char internal[128]; } rep_unique_id;                           // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(rep_unique_id*) = nullptr;
    int (*CommInitRank)(void**, int, rep_unique_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api(std::string* why) {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // an RCCL the process already carries comes first (PyTorch-ROCm loads its bundled copy under the name
        // "librccl.so"): one RCCL instance per process, whoever brought it
        const char* loaded[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : loaded) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (api.lib) break;
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (api.lib) {
            api.GetUniqueId = reinterpret_cast<int (*)(rep_unique_id*)>(dlsym(api.lib, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<int (*)(void**, int, rep_unique_id, int)>(dlsym(api.lib, "ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(api.lib, "ncclCommDestroy"));
            api.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(api.lib, "ncclAllGather"));
            api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(api.lib, "ncclGetErrorString"));
        }
    }
    if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
        if (why) *why = api.lib ? "librccl lacks an entry point" : (std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : ""));
        return nullptr;
    }
    return &api;
}
constexpr int RCCL_FLOAT32 = 7;                                               // ncclFloat32 (rccl.h)
}  // namespace

int lio_replicate_unique_id(void* id128) {
    if (!id128) return LIO_ERR_INVALID_ARG;
    RcclApi* r = rccl_api(nullptr);
    if (!r) return LIO_ERR_HIP;
    rep_unique_id id;
    if (r->GetUniqueId(&id) != 0) return LIO_ERR_HIP;
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return LIO_OK;
}

int lio_replicate_init(lio_ctx* c, const void* id128, int rank, int world, int list_capacity) {
    if (!c || world < 1 || rank < 0 || rank >= world || world * REP_MAX_BOXES > REP_MAX_BOXES_TOTAL || list_capacity < 0 ||
        (world > 1 && !id128 && !c->dbg_rep_loopback))
        return LIO_ERR_INVALID_ARG;
    enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    lio_replicate_finalize(c);
    const bool loopback = world > 1 && !id128;
    // points per list and rank in one message: a scan inserts a few thousand points (5-30 % of N_q in the reference's
    // runs); 32 Ki points = 1 MiB per rank and message keeps an 8-rank all-gather at ~8 MiB.  A rank with more points
    // is flagged, the exchange is repeated with the full scan capacity (lio_replicate_step does that itself).
    uint32_t cap = list_capacity > 0 ? (uint32_t)list_capacity : 32768u;
    if (cap > (uint32_t)c->cap_q) cap = (uint32_t)c->cap_q;
    const size_t msg_f4 = (size_t)REP_HEADER_F4 + 2 * (size_t)c->cap_q;        // buffers sized for the fallback capacity
    HIPCHK(c, hipMalloc((void**)&c->d_rep_send, msg_f4 * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_rep_recv, msg_f4 * (size_t)world * sizeof(float4)));
    int rc = ensure_add_buffers(c, (size_t)world * (size_t)c->cap_q);
    if (rc != LIO_OK) return rc;
    if (!loopback && (world > 1 || id128 != nullptr)) {   // (a single rank WITH an id runs the same collective: tests)
        std::string why;
        RcclApi* r = rccl_api(&why);
        if (!r) return c->fail(LIO_ERR_HIP, "lio_replicate_init", why.c_str());
        rep_unique_id id;
        std::memcpy(id.internal, id128, sizeof(id.internal));
        const int e = r->CommInitRank(&c->rep_comm, world, id, rank);
        if (e != 0) return c->fail(LIO_ERR_HIP, "ncclCommInitRank", r->GetErrorString ? r->GetErrorString(e) : "failed");
    }
    if (loopback) {
        HIPCHK(c, hipMalloc((void**)&c->d_rep_stage, (size_t)(world - 1) * 2 * (size_t)c->cap_q * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&c->d_rep_stage_cnt, (size_t)(world - 1) * 2 * sizeof(uint32_t)));
    }
    c->rep_loopback = loopback; c->rep_staged = 0;
    c->rep_rank = rank; c->rep_world = world; c->rep_cap = cap; c->rep_n_boxes = 0;
    return LIO_OK;
}

void lio_replicate_finalize(lio_ctx* c) {
    if (!c) return;
    (void)enter_map(c);
    if (c->rep_comm) {
        RcclApi* r = rccl_api(nullptr);
        if (r) (void)r->CommDestroy(c->rep_comm);
        c->rep_comm = nullptr;
    }
    if (c->d_rep_send) (void)hipFree(c->d_rep_send);
    if (c->d_rep_recv) (void)hipFree(c->d_rep_recv);
    if (c->d_rep_stage) (void)hipFree(c->d_rep_stage);
    if (c->d_rep_stage_cnt) (void)hipFree(c->d_rep_stage_cnt);
    c->d_rep_send = nullptr; c->d_rep_recv = nullptr; c->d_rep_stage = nullptr; c->d_rep_stage_cnt = nullptr;
    c->rep_world = 0; c->rep_n_boxes = 0; c->rep_loopback = false; c->rep_staged = 0;
}

int lio_replicate_defer(lio_ctx* c, int on) {
    if (!c) return LIO_ERR_INVALID_ARG;
    { const int rc = enter_map(c); if (rc != LIO_OK) return rc; }      // (a batch held back under the old setting enters the map first)
    c->rep_defer = on != 0;
    return LIO_OK;
}
int lio_replicate_queue_boxes(lio_ctx* c, const float* minmax6, int nb) {
    if (!c || nb < 0 || (nb > 0 && !minmax6)) return LIO_ERR_INVALID_ARG;
    if (c->rep_world == 0) return c->fail(LIO_ERR_INVALID_ARG, "lio_replicate_queue_boxes", "lio_replicate_init has not been called");
    if (c->rep_n_boxes + nb > REP_MAX_BOXES) return c->fail(LIO_ERR_CAPACITY, "lio_replicate_queue_boxes", "more than 4 boxes queued for one exchange");
    std::memcpy(c->rep_boxes + 6 * c->rep_n_boxes, minmax6, (size_t)nb * 6 * sizeof(float));
    c->rep_n_boxes += nb;
    return LIO_OK;
}

// One exchange: this rank's two lists (device, counts in d_counters[5..6]) + queued boxes -> message -> ncclAllGather on
// the library's stream -> concatenation in rank order -> one read-back of the four totals -> Delete_Point_Boxes, then
// Add_Points(all PointToAdd, true) and Add_Points(all PointNoNeedDownsample, false) on this replica.
// loop-back: the message of parked virtual rank r into its slot of the gathered buffer
static int pack_parked_rank(lio_ctx* c, int r, uint32_t cap) {
    const size_t msg_f4 = (size_t)REP_HEADER_F4 + 2 * (size_t)cap;
    RepPackArgs pr{};
    pr.counts = c->d_rep_stage_cnt + 2 * (size_t)r; pr.cap = cap;
    pr.add = c->d_rep_stage + (size_t)(2 * r) * (size_t)c->cap_q; pr.nodown = c->d_rep_stage + (size_t)(2 * r + 1) * (size_t)c->cap_q;
    pr.msg = c->d_rep_recv + (size_t)r * msg_f4;
    std::memcpy(pr.boxes, c->rep_stage_boxes[r], sizeof(pr.boxes)); pr.n_boxes = c->rep_stage_nb[r];
    HIPCHK(c, launch_replicate_pack(pr, c->stream));
    return LIO_OK;
}
// the side stream, its event, the second mail box and the two sets of gathered-list buffers of the deferred schedule
static int ensure_deferred(lio_ctx* c) {
    if (!c->side_stream) {
        // the LOWEST priority the device offers: what runs there is meant to fill the gaps of the scan step (measured: no
        // difference to the default priority on this part, scan step 144.6-148.3 vs 146.1 us; LIO_SIDE_PRIORITY=0 switches back)
        int least = 0, greatest = 0;
        const char* e = std::getenv("LIO_SIDE_PRIORITY");
        if ((e != nullptr && e[0] == '0') || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
        // (round 6: confining the side stream to every 2nd / 4th / 8th compute unit -- hipExtStreamCreateWithCUMask -- changed neither the
        // scan step beside the plan nor the update, profiles/r06w_side_stream_cu_mask_ab.log: the step is slower in a world of eight
        // because the map grows eight times as fast, not because the plan takes its compute units)
        HIPCHK(c, hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
    }
    if (!c->ev_side) HIPCHK(c, hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    if (!c->h_small2) {
        HIPCHK(c, hipHostMalloc(&c->h_small2, 256, hipHostMallocMapped));
        std::memset(c->h_small2, 0, 256);
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_small2, c->h_small2, 0));
    }
    if (!c->h_small4) {
        HIPCHK(c, hipHostMalloc(&c->h_small4, 256, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(c->h_small4, 0, 256);
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_small4, c->h_small4, 0));
    }
    if (!c->d_cagg_lists) {
        HIPCHK(c, hipMalloc((void**)&c->d_cagg_lists, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->d_cagg_lists, 0, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
    }
    if (c->cap_def < c->cap_add) {
        for (int i = 0; i < 2; ++i) {
            if (c->d_def_add[i]) (void)hipFree(c->d_def_add[i]);
            if (c->d_def_plain[i]) (void)hipFree(c->d_def_plain[i]);
            c->d_def_add[i] = c->d_def_plain[i] = nullptr;
        }
        c->cap_def = 0;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipMalloc((void**)&c->d_def_add[i], c->cap_add * sizeof(float4)));
            HIPCHK(c, hipMalloc((void**)&c->d_def_plain[i], c->cap_add * sizeof(float4)));
            if (!c->d_def_boxes[i]) HIPCHK(c, hipMalloc((void**)&c->d_def_boxes[i], 6 * REP_MAX_BOXES_TOTAL * sizeof(float)));
        }
        c->cap_def = c->cap_add;
    }
    return LIO_OK;
}
// the plan of the deferred batch on the SIDE stream (called from the first measurement pass of the next scan step, behind its
// launches): nothing on the main stream waits for it until the next exchange
int deferred_launch_plan(lio_ctx* c) {
    lio_ctx::DeferredBatch& d = c->def;
    d.plan_owed = false;
    if (!d.pending || d.planned || d.n_add == 0) return LIO_OK;
    hipStream_t main_stream = c->stream;
    c->stream = c->side_stream;                 // (every launch helper of the fold takes the context's stream)
    const int rc = fold_batch_plan(c, c->d_def_add[c->def_cur], d.n_add, d.box, d.fa, d.ba);
    c->stream = main_stream;
    if (rc != LIO_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev_side, c->side_stream));
    d.planned = true; d.joined = false;
    ++c->n_plans_on_side_stream;
    return LIO_OK;
}
// main stream <- side stream: nothing that follows on the main stream starts before the plan is done (usually it finished
// during the scan step: then the host has seen it complete and no dependency is enqueued at all)
int deferred_join_side(lio_ctx* c) {
    if (!c->def.planned || c->def.joined) return LIO_OK;
    const hipError_t q = hipEventQuery(c->ev_side);
    if (q == hipErrorNotReady) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_side, 0));
    else if (q != hipSuccess) return c->fail(LIO_ERR_HIP, "deferred exchange: side stream", hipGetErrorString(q));
    c->def.joined = true;                       // (the main stream is ordered behind this plan: nothing to do until the next one)
    return LIO_OK;
}
// The deferred batch enters the map: Delete_Point_Boxes(all boxes), Add_Points(all PointToAdd, true),
// Add_Points(all PointNoNeedDownsample, false), rank order inside each -- exactly what the immediate schedule runs, one
// exchange later.  A batch whose plan has not run (no scan step came in between, or it carries delete boxes: their points must
// be gone before the fold looks at the voxels) is planned here, on the main stream.
int deferred_settle(lio_ctx* c, int* add_counter, int* n_deleted) {
    lio_ctx::DeferredBatch& d = c->def;
    if (add_counter) *add_counter = 0;
    if (n_deleted) *n_deleted = 0;
    if (!d.pending) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = deferred_join_side(c);
    if (rc != LIO_OK) return rc;
    const uint32_t n_add = d.n_add, n_plain = d.n_plain, n_boxes = d.n_boxes;
    const HostBox box = d.box;
    const bool planned = d.planned;
    const int cur = c->def_cur;
    d.pending = false; d.plan_owed = false; d.planned = false; d.joined = false;      // (whatever happens below, the batch is not applied twice)
    ++c->n_settled;
    int deleted = 0;
    if (n_boxes > 0 && c->n_points > 0) {
        c->hint_valid = false;
        rc = delete_boxes_resident(c, (int)n_boxes, &deleted, c->d_def_boxes[cur]);
        if (rc != LIO_OK) return rc;
        account_deleted(c, deleted);
    }
    c->last_add_counter = 0;
    if (n_add > 0) {
        if (c->fold_n != 0) {                   // (left behind by a call that failed half-way)
            rc = apply_pending(c);
            if (rc != LIO_OK) return rc;
        }
        if ((size_t)c->n_live + (size_t)c->plain_n + (size_t)n_add > c->cap_map)
            return c->fail(LIO_ERR_CAPACITY, "lio_replicate_step", "live points + gathered lists exceed max_map_points");
        if (!planned) {
            rc = fold_batch_plan(c, c->d_def_add[cur], n_add, box, d.fa, d.ba);
            if (rc != LIO_OK) return rc;
        }
        rc = fold_batch_commit(c, d.fa, d.ba, n_add, box);
        if (rc != LIO_OK) return rc;
    }
    rc = plain_batch_device(c, c->d_def_plain[cur], n_plain, box);
    if (rc != LIO_OK) return rc;
    rc = rebuild_from_stored(c, add_counter);
    if (n_deleted) *n_deleted = deleted;
    return rc;
}

// The commit half of the deferred batch on the SIDE stream (behind its plan, which ran there beside the scan step): enqueued, not
// waited for -- lio_replicate_step then runs everything of the next exchange that does not touch the map on the main stream beside
// it and collects the statistics with deferred_settle_side_end.  Only the plain case: planned, no delete boxes, an index that takes
// incremental updates, every sort / scan on the single-launch kernels (the library's workspace is shared).  Returns LIO_OK with
// c->settle_side == false when the case is another one (the caller then settles the blocking way).
static int deferred_settle_side_begin(lio_ctx* c) {
    lio_ctx::DeferredBatch& d = c->def;
    if (!c->dbg_settle_beside || !d.pending || !d.planned || d.n_boxes != 0 || d.n_add == 0 || c->fold_n != 0 || c->plain_n != 0 ||
        c->map_broken || !c->have_map || c->n_live == 0 || c->n_points == 0 || c->dbg_library_sort || !compact_small(c, (size_t)c->nq) ||
        (size_t)d.n_add + (size_t)d.n_plain > (size_t)GSORT_MAX || c->prep_active || c->pass_in_flight || c->gate_n != 0 ||
        (size_t)c->n_live + (size_t)d.n_add + (size_t)d.n_plain > c->cap_map)
        return LIO_OK;
    const uint32_t n_add = d.n_add, n_plain = d.n_plain;
    const HostBox box = d.box;
    const int cur = c->def_cur;
    d.pending = false; d.plan_owed = false; d.planned = false; d.joined = false;      // (whatever happens below, the batch is not applied twice)
    ++c->n_settled;
    c->last_add_counter = 0;
    hipStream_t main_stream = c->stream;
    // The commit writes the map (tomb-stones, directory records) and shares d_apply / the sort scratch with the main stream, with no
    // event between the two streams: by the call graph the main stream holds nothing that reads the map at this moment -- every pass
    // was collected by the host before lio_replicate_step, armed passes were called off by enter() (they return without looking at
    // the map).  The invariant is CHECKED on the host (pass_in_flight / gate_n in the condition above: anything else settles the
    // blocking way); an event recorded on the main stream for the side stream to wait on -- the round-5 review's suggestion, built
    // first -- cost 8 us per exchange on the emulated world of eight (profiles/r06v_*: update 131 -> 123 us) for a dependency that is
    // always already satisfied.
    c->stream = c->side_stream;                 // (every launch helper of the update takes the context's stream)
    bool launched = false;
    int rc = fold_batch_commit(c, d.fa, d.ba, n_add, box);
    if (rc == LIO_OK) rc = plain_batch_device(c, c->d_def_plain[cur], n_plain, box);
    c->settle_seq = ++c->small4_seq;            // (published by the apply kernel itself, into the mail box of the side stream)
    if (c->dbg_publish_in_kernel) {
        if (rc == LIO_OK) rc = apply_pending_enqueue(c, &launched, c->d_small4, c->settle_seq);
    } else {
        if (rc == LIO_OK) rc = apply_pending_enqueue(c, &launched);
        if (rc == LIO_OK && launched) {
            const hipError_t e = launch_publish_words(c->d_apply, 32, c->d_small4, c->settle_seq, c->side_stream);
            if (e != hipSuccess) rc = c->fail(LIO_ERR_HIP, "deferred exchange: publishing the statistics", hipGetErrorString(e));
        }
    }
    c->stream = main_stream;
    if (rc != LIO_OK) {
        // part of the chain may be on the side stream: nothing of the error handling on the main stream may run beside it
        (void)hipStreamSynchronize(c->side_stream);
        return rc;
    }
    c->settle_side = launched;                  // (not launched: a full rebuild took the batch in, synchronously)
    if (launched) ++c->n_settled_beside_exchange;
    return LIO_OK;
}
static int deferred_settle_side_end(lio_ctx* c, int* add_counter) {
    if (add_counter) *add_counter = 0;
    if (!c->settle_side) return LIO_OK;
    c->settle_side = false;
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(c->h_small4) + 32;
    int rc = LIO_OK;
    for (uint64_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != c->settle_seq; ++spins) {
        if ((spins & 0xfffffu) == 0xfffffu) {
            const hipError_t q = hipStreamQuery(c->side_stream);
            if (q != hipErrorNotReady) {          // idle (or broken) without the word: the chain never got there
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == c->settle_seq) break;
                rc = c->fail(LIO_ERR_HIP, "deferred exchange: the commit on the side stream", q == hipSuccess ? "stream idle but its statistics never arrived" : hipGetErrorString(q));
                break;
            }
        }
        __builtin_ia32_pause();
    }
    uint32_t st[32] = {};
    if (rc == LIO_OK) {
        std::memcpy(st, c->h_small4, sizeof(st));
        if (st[14] != 0u) {
            (void)hipMemsetAsync(c->d_apply + 14, 0, sizeof(uint32_t), c->stream);
            rc = c->fail(LIO_ERR_HIP, "Add_Points", "a batch point lies outside the key frame derived from the batch's bounding box");
        } else {
            rc = check_compact_error(c, st[15]);
        }
    }
    // (the side stream is idle: whatever the second half launches -- a full rebuild -- runs on the main stream behind nothing of it)
    rc = apply_pending_finish(c, st, rc);
    if (rc == LIO_OK && add_counter) *add_counter = c->last_add_counter;
    return rc;
}

static int replicate_exchange_apply(lio_ctx* c, uint32_t cap, uint32_t tot[4], HostBox& box, bool defer = false) {
    hipStream_t s = c->stream;
    const size_t msg_f4 = (size_t)REP_HEADER_F4 + 2 * (size_t)cap;
    RepPackArgs pa{};
    pa.counts = c->d_counters + 5; pa.cap = cap; pa.add = c->d_add_pts; pa.nodown = c->d_add_pts2;
    pa.msg = c->rep_comm ? c->d_rep_send : c->d_rep_recv + (size_t)c->rep_rank * msg_f4;   // no communicator: straight into this rank's slot of the gathered buffer
    std::memcpy(pa.boxes, c->rep_boxes, sizeof(pa.boxes)); pa.n_boxes = c->rep_n_boxes;
    HIPCHK(c, launch_replicate_pack(pa, s));
    if (c->rep_loopback) {
        // the messages of the parked virtual ranks: packed when they parked (their own GPUs would do it concurrently);
        // here only when the exchange is repeated at a larger capacity -- by the same kernel, from the parked lists
        for (int r = 0; r + 1 < c->rep_world; ++r) {
            if (!(c->rep_staged & (1u << r))) return c->fail(LIO_ERR_INVALID_ARG, "lio_replicate_step", "loop-back: a virtual rank has not parked its lists");
            if (cap == c->rep_cap) continue;
            const int rc = pack_parked_rank(c, r, cap);
            if (rc != LIO_OK) return rc;
        }
    }
    if (c->rep_comm) {
        RcclApi* r = rccl_api(nullptr);
        const int e = r->AllGather(c->d_rep_send, c->d_rep_recv, msg_f4 * 4, RCCL_FLOAT32, c->rep_comm, s);
        if (e != 0) return c->fail(LIO_ERR_HIP, "ncclAllGather", r->GetErrorString ? r->GetErrorString(e) : "failed");
    }
    RepCompactArgs ca{};
    ca.gathered = c->d_rep_recv; ca.world = c->rep_world; ca.cap = cap; ca.add_out = c->d_add_pts; ca.nodown_out = c->d_add_pts2;
    ca.boxes_out = c->d_boxes; ca.totals = c->d_counters + 8; ca.bbox = c->d_counters + 24;
    if (defer) {                                // the gathered lists wait in buffers of their own until the next exchange
        const int g = c->def_cur ^ 1;
        ca.add_out = c->d_def_add[g]; ca.nodown_out = c->d_def_plain[g]; ca.boxes_out = c->d_def_boxes[g];
    }
    HIPCHK(c, launch_replicate_compact(ca, s));
    uint32_t w[22] = {};                                        // d_counters[8 .. 29]: the four totals, the list kernel's error word, ... the bounding box
    if (defer && (c->def.pending || c->settle_side)) {
        // The totals go into the second mail box WITHOUT a wait; the previous batch enters the map behind them on the stream, and
        // its own read-back (the statistics of the index update) is the one host wait of this exchange: by then the totals are there.
        // (Round 5: in the plain case that batch's commit has been running on the side stream since this exchange began -- here its
        // statistics are collected.)
        const uint32_t seq2 = ++c->small2_seq;
        HIPCHK(c, launch_publish_words(c->d_counters + 8, 22, c->d_small2, seq2, s));
        int rc;
        if (c->settle_side) { c->def_last_deleted = 0; rc = deferred_settle_side_end(c, &c->def_last_counter); }
        else rc = deferred_settle(c, &c->def_last_counter, &c->def_last_deleted);
        if (rc != LIO_OK) return rc;
        volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(c->h_small2) + 32;
        for (uint64_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq2; ++spins) {
            if (spins > (1ull << 28)) { HIPCHK(c, hipStreamSynchronize(s)); break; }
        }
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq2) return c->fail(LIO_ERR_HIP, "lio_replicate_step", "the totals of the exchange never arrived");
        std::memcpy(w, c->h_small2, sizeof(w));
    } else {
        HIPCHK(c, read_back(c, w, c->d_counters + 8, sizeof(w)));
    }
    if (w[4] != 0u) return list_kernel_failed(c);
    std::memcpy(tot, w, 4 * sizeof(uint32_t));
    box_from_images(w + 16, box);
    return LIO_OK;
}

int lio_replicate_step(lio_ctx* c, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                       int* n_add_total, int* n_nodown_total, int* n_deleted, int* add_counter) {
    if (!c || !state14 || !(filter_size_map_min > 0.0)) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (c->rep_world == 0) return c->fail(LIO_ERR_INVALID_ARG, "lio_replicate_step", "lio_replicate_init has not been called");
    if (n_add_total) *n_add_total = 0;
    if (n_nodown_total) *n_nodown_total = 0;
    if (n_deleted) *n_deleted = 0;
    if (add_counter) *add_counter = 0;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = deferred_join_side(c);
    if (rc != LIO_OK) return rc;
    // the deferred schedule, plain case: the commit of the batch held back starts NOW on the side stream, beside this rank's list
    // kernel, the message, the all-gather and the compaction below (none of them reads the map).  A virtual rank that only parks
    // its lists (loop-back) leaves the batch where it is: the exchange proper is the last rank's call.
    if (c->rep_defer && c->def.pending && !(c->rep_loopback && c->rep_rank + 1 < c->rep_world)) {
        rc = ensure_deferred(c);
        if (rc == LIO_OK) rc = deferred_settle_side_begin(c);
        if (rc != LIO_OK) return rc;
    }
    struct SettleGuard {                        // an error return below must not leave the commit's statistics uncollected
        lio_ctx* c;
        ~SettleGuard() { if (c->settle_side) (void)deferred_settle_side_end(c, nullptr); }
    } settle_guard{c};
    // map_incremental's two lists of this rank stay on the device, and so do their counts
    rc = map_incremental_lists(c, state14, filter_size_map_min, flg_EKF_inited, nullptr);
    if (rc != LIO_OK) return rc;
    if (c->rep_loopback && c->rep_rank + 1 < c->rep_world) {
        // loop-back: this call speaks for a virtual rank other than the last one -- park what it would have sent
        const int r = c->rep_rank;
        HIPCHK(c, launch_replicate_stage(c->d_counters + 5, c->d_add_pts, c->d_add_pts2, (uint32_t)c->cap_q,
                                         c->d_rep_stage + (size_t)(2 * r) * (size_t)c->cap_q, c->d_rep_stage + (size_t)(2 * r + 1) * (size_t)c->cap_q,
                                         c->d_rep_stage_cnt + 2 * (size_t)r, c->stream));
        std::memcpy(c->rep_stage_boxes[r], c->rep_boxes, sizeof(c->rep_boxes));
        c->rep_stage_nb[r] = c->rep_n_boxes;
        c->rep_n_boxes = 0;
        c->rep_staged |= 1u << r;
        return pack_parked_rank(c, r, c->rep_cap);
    }
    uint32_t tot[4] = {0, 0, 0, 0};
    HostBox box;
    const bool defer = c->rep_defer;
    if (defer) {
        rc = ensure_deferred(c);
        if (rc != LIO_OK) return rc;
        c->def_last_counter = 0; c->def_last_deleted = 0;
    }
    rc = replicate_exchange_apply(c, c->rep_cap, tot, box, defer);
    if (rc != LIO_OK) return rc;
    if ((tot[3] & REP_FLAG_OVERFLOW) != 0 && c->rep_cap < (uint32_t)c->cap_q) {
        // some rank had more points than a message holds: every rank sees the flag, every rank repeats the exchange at
        // the full scan capacity (the lists and counters of this rank are still in place: compact wrote elsewhere only
        // if ... the first round's outputs are overwritten by the second)
        // Immediate schedule: the first round's compaction wrote the gathered lists over this rank's own (d_add_pts / d_add_pts2):
        // they are made again, on the same map, with the same result.  Deferred schedule: the gathered lists went to buffers of their
        // own and the first round has already let the HELD batch into the map (deferred_settle) -- lists made now would see batch
        // k - 1, which the documented schedule (and every rank's twin) does not: this rank's lists and counts of the first round are
        // still in place (nothing of the settle touches d_add_pts / d_add_pts2 / d_counters[5..6]) and are packed again as they are.
        if (!defer) {
            rc = map_incremental_lists(c, state14, filter_size_map_min, flg_EKF_inited, nullptr);
            if (rc != LIO_OK) return rc;
        }
        rc = replicate_exchange_apply(c, (uint32_t)c->cap_q, tot, box, defer);
        if (rc != LIO_OK) return rc;
    }
    c->rep_n_boxes = 0;
    c->rep_staged = 0;
    if (defer) {
        // (a first exchange finds nothing to settle: its read-back was the plain one)
        if ((size_t)c->n_live + (size_t)tot[0] + (size_t)tot[1] > c->cap_map)
            return c->fail(LIO_ERR_CAPACITY, "lio_replicate_step", "live points + gathered lists exceed max_map_points");
        lio_ctx::DeferredBatch& d = c->def;
        c->def_cur ^= 1;                        // what was gathered is now the deferred batch
        d.pending = tot[0] + tot[1] + tot[2] > 0;
        d.planned = false; d.joined = false;
        d.n_add = tot[0]; d.n_plain = tot[1]; d.n_boxes = tot[2]; d.box = box;
        // its plan may run beside the next scan step unless delete boxes have to go first (or nothing is down-sampled at all)
        d.plan_owed = d.pending && tot[0] > 0 && tot[2] == 0;
        if (!d.plan_owed) HIPCHK(c, reset_list_box(c));   // (no voxel_keys_kernel will do it before the next lists are made)
        if (n_add_total) *n_add_total = (int)tot[0];
        if (n_nodown_total) *n_nodown_total = (int)tot[1];
        if (n_deleted) *n_deleted = c->def_last_deleted;
        if (add_counter) *add_counter = c->def_last_counter;
        return LIO_OK;
    }
    if ((size_t)c->n_live + (size_t)tot[0] + (size_t)tot[1] > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "lio_replicate_step", "live points + gathered lists exceed max_map_points");
    // every replica: the boxes of all ranks (rank order), then the insert lists of all ranks (rank order)
    int deleted = 0;
    if (tot[2] > 0 && c->n_points > 0) {
        c->hint_valid = false;
        rc = delete_boxes_resident(c, (int)tot[2], &deleted);
        if (rc != LIO_OK) return rc;
        account_deleted(c, deleted);
    }
    c->last_add_counter = 0;
    if (tot[0] == 0) HIPCHK(c, reset_list_box(c));                // (no voxel_keys_kernel will do it)
    rc = fold_batch_device(c, c->d_add_pts, tot[0], box);
    if (rc != LIO_OK) return rc;
    rc = plain_batch_device(c, c->d_add_pts2, tot[1], box);
    if (rc != LIO_OK) return rc;
    rc = rebuild_from_stored(c, add_counter);
    if (n_add_total) *n_add_total = (int)tot[0];
    if (n_nodown_total) *n_nodown_total = (int)tot[1];
    if (n_deleted) *n_deleted = deleted;
    return rc;
}

int lio_map_set_downsample_param(lio_ctx* c, float downsample_size) {
    if (!c || !(downsample_size > 0.0f)) return LIO_ERR_INVALID_ARG;
    if (c->n_points == 0 && c->cfg.cell_size == 0.0f) {       // nothing indexed yet: the grid may still follow
        c->grid.cell = auto_cell_size(downsample_size);
        c->grid.inv_cell = 1.0f / c->grid.cell;
    } else if (downsample_size > c->grid.cell) {
        // the Add_Points fold looks at the 2x2x2 grid cells a down-sample box can touch: the box must not exceed a cell
        return c->fail(LIO_ERR_INVALID_ARG, "lio_map_set_downsample_param", "down-sample box larger than the neighbour-grid cell of the built map");
    }
    c->cfg.filter_size_map_min = downsample_size;
    return LIO_OK;
}
// (a call that failed between an Add_Points and its index update leaves the batch pending, its counts on the device: it is
// indexed first, so that the numbers are exact)
// Both read HOST counters: they touch the stream only when a batch is really owed to the map (a deferred exchange, a call that
// failed half-way).  In particular they do NOT call armed passes off and do NOT flush a pending scan pull: the node calls them
// between lio::scan_changed and the update (laserMapping.cpp:2343-2344), and the fused / assisted pull of the scan's first search
// must survive that (round 4's enter_map() here made the assisted pull dead code in the node's own call order).
static void settle_owed_batches(lio_ctx* c) {
    if (!c->def.pending && c->fold_n == 0 && c->plain_n == 0) return;
    (void)enter_map(c);
    if ((c->fold_n != 0 || c->plain_n != 0) && hipSetDevice(c->device) == hipSuccess) (void)apply_pending(c);
}
long long lio_map_size(lio_ctx* c) {
    if (!c) return 0;
    settle_owed_batches(c);
    return (long long)c->n_points - (long long)c->n_retired;
}
long long lio_map_validnum(lio_ctx* c) {
    if (!c) return 0;
    settle_owed_batches(c);
    return (long long)c->n_live;
}

// Delete_by_range over the boxes already sitting in c->d_boxes (<= 64)
static int delete_boxes_resident(lio_ctx* c, int nbb, int* total, const float* d_boxes) {
    if (d_boxes == nullptr) d_boxes = c->d_boxes;
    HIPCHK(c, hipMemsetAsync(c->d_counters + 2, 0, sizeof(uint32_t), c->stream));
    HIPCHK(c, launch_delete_boxes(c->d_pts, c->n_points, d_boxes, nbb, c->d_counters + 2, c->stream));
    uint32_t cnt = 0;
    HIPCHK(c, read_back(c, &cnt, c->d_counters + 2, sizeof(uint32_t)));
    *total += (int)cnt;
    return LIO_OK;
}
static void account_deleted(lio_ctx* c, int total) {
    c->n_live -= (uint32_t)total;
    c->box_deleted_pending += (uint32_t)total;
    c->n_dead += (uint32_t)total;              // dead slots count towards the budget that triggers the next full rebuild
}
int lio_map_delete_boxes(lio_ctx* c, const float* minmax6, int nb, int* n_deleted) {
    if (!c || nb < 0 || (nb > 0 && !minmax6)) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (n_deleted) *n_deleted = 0;
    if (nb == 0 || c->n_points == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    c->hint_valid = false;
    {
        const int rc = apply_pending(c);                          // (a batch left behind by a call that failed half-way)
        if (rc != LIO_OK) return rc;
    }
    int total = 0;
    for (int b0 = 0; b0 < nb; b0 += 64) {
        const int nbb = nb - b0 < 64 ? nb - b0 : 64;
        HIPCHK(c, hipMemcpyAsync(c->d_boxes, minmax6 + 6 * (size_t)b0, (size_t)nbb * 6 * sizeof(float),
                                 hipMemcpyHostToDevice, c->stream));
        const int rc = delete_boxes_resident(c, nbb, &total);
        if (rc != LIO_OK) return rc;
    }
    account_deleted(c, total);
    if (n_deleted) *n_deleted = total;
    return LIO_OK;
}

// lasermap_fov_segment (src/laserMapping.cpp:1136-1200): keep the map inside a cube of edge cube_len that
// follows the LiDAR; when the LiDAR comes within MOV_THRESHOLD * DET_RANGE of a face the cube moves by
// mov_dist and the slab left behind is deleted (Delete_Point_Boxes).  Types follow the reference: the
// cube corners are float (BoxPointType), pos_LiD / cube_len double, DET_RANGE / MOV_THRESHOLD float.
int lio_fov_segment(lio_ctx* c, const double pos_lid[3], double cube_len, float det_range, float* boxes_out,
                    int* n_boxes, int* n_deleted) {
    if (!c || !pos_lid) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (n_boxes) *n_boxes = 0;
    if (n_deleted) *n_deleted = 0;
    const float MOV_THRESHOLD = 1.5f;                                        // :53
    if (!c->localmap_initialized) {                                           // :1144-1153
        for (int i = 0; i < 3; ++i) {
            c->localmap_min[i] = (float)(pos_lid[i] - cube_len / 2.0);
            c->localmap_max[i] = (float)(pos_lid[i] + cube_len / 2.0);
        }
        c->localmap_initialized = true;
        return LIO_OK;
    }
    float dist_to_map_edge[3][2];
    bool need_move = false;
    const float thr = MOV_THRESHOLD * det_range;
    for (int i = 0; i < 3; ++i) {                                             // :1158-1166
        dist_to_map_edge[i][0] = (float)std::fabs(pos_lid[i] - (double)c->localmap_min[i]);
        dist_to_map_edge[i][1] = (float)std::fabs(pos_lid[i] - (double)c->localmap_max[i]);
        if (dist_to_map_edge[i][0] <= thr || dist_to_map_edge[i][1] <= thr) need_move = true;
    }
    if (!need_move) return LIO_OK;
    float new_min[3], new_max[3];
    for (int i = 0; i < 3; ++i) { new_min[i] = c->localmap_min[i]; new_max[i] = c->localmap_max[i]; }
    const double m1 = (cube_len - 2.0 * (double)MOV_THRESHOLD * (double)det_range) * 0.5 * 0.9;
    const double m2 = (double)(det_range * (MOV_THRESHOLD - 1));
    const float mov_dist = (float)(m1 > m2 ? m1 : m2);                        // :1173
    float boxes[3 * 6];
    int nb = 0;
    for (int i = 0; i < 3; ++i) {                                             // :1174-1192
        float tmin[3], tmax[3];
        for (int k = 0; k < 3; ++k) { tmin[k] = c->localmap_min[k]; tmax[k] = c->localmap_max[k]; }
        if (dist_to_map_edge[i][0] <= thr) {
            new_max[i] -= mov_dist; new_min[i] -= mov_dist;
            tmin[i] = c->localmap_max[i] - mov_dist;
        } else if (dist_to_map_edge[i][1] <= thr) {
            new_max[i] += mov_dist; new_min[i] += mov_dist;
            tmax[i] = c->localmap_min[i] + mov_dist;
        } else {
            continue;
        }
        for (int k = 0; k < 3; ++k) { boxes[6 * nb + k] = tmin[k]; boxes[6 * nb + 3 + k] = tmax[k]; }
        ++nb;
    }
    for (int i = 0; i < 3; ++i) { c->localmap_min[i] = new_min[i]; c->localmap_max[i] = new_max[i]; }
    if (boxes_out) std::memcpy(boxes_out, boxes, sizeof(float) * 6 * (size_t)nb);
    if (n_boxes) *n_boxes = nb;
    if (nb > 0) return lio_map_delete_boxes(c, boxes, nb, n_deleted);         // :1196-1198
    return LIO_OK;
}

int lio_map_flatten(lio_ctx* c, float* out_xyz, long long cap, long long* n) {
    if (!c || cap < 0 || (cap > 0 && !out_xyz)) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    HIPCHK(c, hipSetDevice(c->device));
    if (n) *n = (long long)c->n_live;
    if (c->n_points == 0 || cap == 0) return LIO_OK;
    // compact the live points (order-preserving), pack xyz, download
    HIPCHK(c, compact_live_any(c, c->d_pts, c->n_points, c->d_pts_tmp, c->d_counters + 3));
    uint32_t live = 0;
    HIPCHK(c, read_back(c, &live, c->d_counters + 3, sizeof(uint32_t)));
    if (n) *n = (long long)live;
    const size_t m = (size_t)live < (size_t)cap ? (size_t)live : (size_t)cap;
    if (m == 0) return LIO_OK;
    int rc = ensure_scratch(c, m * 3 * sizeof(float));
    if (rc != LIO_OK) return rc;
    HIPCHK(c, launch_pack_xyz(c->d_pts_tmp, (uint32_t)m, (float*)c->d_scratch, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, m * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c));
    std::memcpy(out_xyz, c->h_scratch, m * 3 * sizeof(float));
    return LIO_OK;
}

int lio_map_acquire_removed_points(lio_ctx* c, float* out_xyz, long long cap, long long* n) {
    if (!c || !n || cap < 0) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    HIPCHK(c, hipSetDevice(c->device));
    {
        const int rc0 = collect_box_deleted(c);
        if (rc0 != LIO_OK) return rc0;
    }
    *n = (long long)c->n_removed;
    if (!out_xyz) return LIO_OK;                              // count only
    if (cap < (long long)c->n_removed) return c->fail(LIO_ERR_CAPACITY, "lio_map_acquire_removed_points", "cap < removed points");
    const size_t m = c->n_removed;
    if (m == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, m * 3 * sizeof(float));
    if (rc != LIO_OK) return rc;
    HIPCHK(c, launch_pack_xyz(c->d_removed, (uint32_t)m, (float*)c->d_scratch, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, m * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c));
    std::memcpy(out_xyz, c->h_scratch, m * 3 * sizeof(float));
    c->n_removed = 0;                                         // Points_deleted.clear()  (ikd_Tree.cpp:567)
    return LIO_OK;
}

int lio_map_get_info(lio_ctx* c, lio_map_info* info) {
    if (!c || !info) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    info->n_points = c->n_points; info->n_live = c->n_live; info->n_cells = c->n_cells;
    info->table_slots = (long long)c->levels.mask[1] + 1;
    info->cell_size = c->grid.cell; info->downsample_size = c->cfg.filter_size_map_min;
    info->n_retired = c->n_retired; info->n_incremental_updates = c->n_incremental_updates; info->n_full_rebuilds = c->n_full_rebuilds;
    info->n_boundary_points = c->n_boundary_points;
    info->bytes_resident = (long long)c->n_points * (long long)(sizeof(float4) + sizeof(uint64_t)) +
                           (long long)c->ltable_slots * (long long)sizeof(HashEntry) + (long long)c->child_blocks_cap * 512ll;
    return LIO_OK;
}

int lio_map_nearest_search(lio_ctx* c, const float* q_xyz, int n, int stride_bytes, float* out_xyz,
                           float* out_sqdist, int* out_count) {
    if (!c || n < 0 || (n > 0 && !q_xyz) || stride_bytes < 12) return LIO_ERR_INVALID_ARG;
    { const int rc_settle = enter_map(c); if (rc_settle != LIO_OK) return rc_settle; }
    if (n == 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->cap_qs) {
        void* old[] = {c->d_qs_world, c->d_qs_nbr, c->d_qs_cnt, c->d_qs_ulist, c->d_qs_useeds};
        for (void* p : old) if (p) (void)hipFree(p);
        c->d_qs_world = nullptr; c->d_qs_nbr = nullptr; c->d_qs_cnt = nullptr; c->d_qs_ulist = nullptr; c->d_qs_useeds = nullptr; c->cap_qs = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_qs_world, (size_t)n * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&c->d_qs_nbr, (size_t)n * K * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&c->d_qs_cnt, (size_t)n));
        HIPCHK(c, hipMalloc((void**)&c->d_qs_ulist, (size_t)n * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&c->d_qs_useeds, (size_t)n * 8 * sizeof(unsigned long long)));
        c->cap_qs = n;
    }
    int rc = upload_points(c, q_xyz, (size_t)n, stride_bytes, c->d_qs_world);
    if (rc != LIO_OK) return rc;
    Pose14 ident{};
    rc = run_knn(c, nullptr, c->d_qs_world, ident, n, n, c->d_qs_nbr, c->d_qs_cnt, c->d_qs_ulist, c->d_qs_useeds, false, true);
    if (rc != LIO_OK) return rc;
    const size_t b_xyz = (size_t)n * K * 3 * sizeof(float), b_d = (size_t)n * K * sizeof(float), b_c = (size_t)n * sizeof(int);
    rc = ensure_scratch(c, b_xyz + b_d + b_c);
    if (rc != LIO_OK) return rc;
    float* dx = (float*)c->d_scratch;
    float* dd = (float*)((char*)c->d_scratch + b_xyz);
    int* dc = (int*)((char*)c->d_scratch + b_xyz + b_d);
    HIPCHK(c, launch_unpack_neighbors(c->d_qs_nbr, c->d_qs_cnt, n, n, dx, dd, dc, c->stream));
    uint32_t walked = 0;
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, b_xyz + b_d + b_c, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, read_back(c, &walked, c->d_counters, sizeof(uint32_t)));      // through host-mapped memory; waits for the stream
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_cand, 0, sizeof(unsigned long long), c->stream));
    if (walked & 0x80000000u) return c->fail(LIO_ERR_HIP, "lio_map_nearest_search", "tree walk exceeded its stack bound");
    if (out_xyz) std::memcpy(out_xyz, c->h_scratch, b_xyz);
    if (out_sqdist) std::memcpy(out_sqdist, (char*)c->h_scratch + b_xyz, b_d);
    if (out_count) std::memcpy(out_count, (char*)c->h_scratch + b_xyz + b_d, b_c);
    return LIO_OK;
}

}  // extern "C"
