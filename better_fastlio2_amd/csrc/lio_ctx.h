// lio_ctx.h -- PRIVATE to liblio_hip.so: the context object behind the C ABI of include/lio_hip.h and the internal entry points its
// translation units share.  One context = one GPU = one HIP stream; every device buffer is allocated once at lio_create (capacities
// from lio_config) so that a scan step performs no allocation.
//   lio_ctx.hip        create / destroy, memory and read-back helpers, the single-launch sort / compaction front ends, profiling, debug keys
//   lio_update.hip     the map: index (re)builds, Add_Points / Delete_Point_Boxes / map_incremental drivers, the replica exchange
//   lio_scan.hip       scan hand-over (set / upload_async / swap) and the raw-scan preparation drivers (undistort, voxel filter)
//   lio_pull.hip       lio_scan_set_async: the fused pull of page-locked host clouds, the assisted pull and its packing thread
//   lio_pass.hip       measurement passes: launches, passes armed ahead of their pose (mail slots, gates), rows, result read-out
//   lio_capi_eskf.hip  the C API of the host-side filter and IMU front end (lio_eskf_*, lio_imu_*, lio_state_*)
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <immintrin.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lio_hip.h"
#include "lio_esekf.hpp"
#include "lio_imu.hpp"
#include "lio_kernels.h"
#include "lio_mapadd.h"
#include "lio_rebuild.h"
#include "lio_scanprep.h"

using namespace lio;

// bounding box of an insert batch on the host (it arrives with the list lengths): the frames of the compressed keys
struct HostBox {
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    bool any = false;
    void add(float x, float y, float z) {
        const float v[3] = {std::isfinite(x) ? x : 0.0f, std::isfinite(y) ? y : 0.0f, std::isfinite(z) ? z : 0.0f};   // non-finite: filed under 0
        for (int i = 0; i < 3; ++i) {
            if (!any || v[i] < lo[i]) lo[i] = v[i];
            if (!any || v[i] > hi[i]) hi[i] = v[i];
        }
        any = true;
    }
    void merge(const HostBox& o) {
        if (!o.any) return;
        add(o.lo[0], o.lo[1], o.lo[2]);
        add(o.hi[0], o.hi[1], o.hi[2]);
    }
};

// The optional second packing thread of the assisted pull (lio_scan_pack_threads(ctx, 2); none by default: a library that starts
// threads should be asked to).  It sleeps on a condition variable; lio_scan_set_async wakes it when a cloud worth the help has been
// handed over, it then spins (<= 400 us) for the job the scan's first pass posts and packs the SECOND half of the host's share
// while the calling thread packs the first.  Whoever gets there first claims that half: a helper that wakes up late finds it taken.
constexpr double PCIE_BYTES_PER_NS = 57.0;               // GPU-initiated reads of page-locked host memory (profiles/r04a_upload_probe.log)
struct PackJob {
    const char* src = nullptr; size_t stride = 0; float* dst = nullptr;
    int lo = 0, hi = 0, chunk = 0; uint32_t stamp = 0; volatile unsigned long long* word = nullptr;
};
static inline void pack_records(const PackJob& j) {
    const __m128 keep_xyz = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1));
    for (int done = j.lo; done < j.hi;) {
        const int end = std::min(j.hi, done + j.chunk);
        // four records per round, streaming stores (whole 64-byte lines leave through the write-combining buffers: no read for
        // ownership of the staging buffer, nothing of it displaces the cloud in the caches -- scripts/micro/pack_probe.cpp on the
        // bench's host: 0.48 -> 0.40 ns per record warm, 1.29 -> 0.94 from DRAM)
        for (int i = done; i < end; i += 4) {                 // (chunks and ranges are multiples of 64)
            const __m128 r0 = _mm_loadu_ps(reinterpret_cast<const float*>(j.src + (size_t)i * j.stride));
            const __m128 r1 = _mm_loadu_ps(reinterpret_cast<const float*>(j.src + (size_t)(i + 1) * j.stride));
            const __m128 r2 = _mm_loadu_ps(reinterpret_cast<const float*>(j.src + (size_t)(i + 2) * j.stride));
            const __m128 r3 = _mm_loadu_ps(reinterpret_cast<const float*>(j.src + (size_t)(i + 3) * j.stride));
            _mm_stream_ps(j.dst + 4 * (size_t)i, _mm_and_ps(r0, keep_xyz)); _mm_stream_ps(j.dst + 4 * (size_t)i + 4, _mm_and_ps(r1, keep_xyz));
            _mm_stream_ps(j.dst + 4 * (size_t)i + 8, _mm_and_ps(r2, keep_xyz)); _mm_stream_ps(j.dst + 4 * (size_t)i + 12, _mm_and_ps(r3, keep_xyz));
        }
        done = end;
        __builtin_ia32_sfence();                            // the records before the word that announces them
        *j.word = ((unsigned long long)j.stamp << 32) | (unsigned long long)(uint32_t)done;
        __builtin_ia32_sfence();                            // (write-combined BAR mapping: nothing lingers)
    }
}
struct PackHelper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    unsigned wake_seq = 0;                               // (under mu) bumped by lio_scan_set_async
    bool quit = false;                                   // (under mu)
    std::atomic<unsigned long long> job_seq{0};          // a job with this number is posted (`job` is complete)
    std::atomic<unsigned long long> claimed{0};          // number of the last job whose second half somebody took
    std::atomic<unsigned long long> helper_busy{0};      // number of the job the helper is packing (0: none)
    PackJob job;
    bool claim(unsigned long long id) {
        unsigned long long cur = claimed.load(std::memory_order_relaxed);
        while (cur < id) if (claimed.compare_exchange_weak(cur, id, std::memory_order_acq_rel)) return true;
        return false;
    }
    void run() {
        unsigned seen_wake = 0;
        unsigned long long seen_job = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || wake_seq != seen_wake; });
                if (quit) return;
                seen_wake = wake_seq;
            }
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400)) {
                const unsigned long long id = job_seq.load(std::memory_order_acquire);
                if (id != seen_job) {
                    seen_job = id;
                    helper_busy.store(id, std::memory_order_release);
                    const bool mine = claim(id);            // (a job the calling thread finished itself long ago: keep looking for this scan's)
                    if (mine) pack_records(job);            // (`job` does not change while helper_busy names it: assist_share waits)
                    helper_busy.store(0, std::memory_order_release);
                    if (mine) break;
                }
                __builtin_ia32_pause();
            }
        }
    }
};

// one attempt of the voxel filter (lio_scan.hip): what was predicted, which record it filled
struct VoxelRun {
    int n = 0;
    float leaf = 0.0f;
    bool time_minor = false, may_compact = false, compact = false;
    int passes = 0;
    uint32_t max_leaves = 0;
    VoxelGridDev* vox = nullptr;               // the record this attempt filled (its summary is what the host reads)
};

struct lio_ctx {
    lio_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::string arch;

    // ---- map --------------------------------------------------------------------------------
    Grid grid{};
    size_t cap_map = 0;
    float4 *d_pts = nullptr, *d_pts_tmp = nullptr;
    uint64_t *d_mkeys = nullptr, *d_mkeys_tmp = nullptr;
    uint32_t *d_vals = nullptr, *d_vals_tmp = nullptr;
    uint64_t* d_cell_keys = nullptr;
    uint32_t *d_cell_counts = nullptr, *d_cell_starts = nullptr;
    HashEntry* d_ltable = nullptr;             // tables of the coarser tree levels (>= 1), back to back
    size_t ltable_slots = 0;
    TreeLevels levels{};
    // cell directory of the k-NN kernel: one block of 64 {start, count} per level-1 cell
    uint2* d_child = nullptr;
    size_t child_blocks_cap = 0;
    uint32_t n_child_blocks = 0;               // blocks handed out (host view; the device bump counter is d_apply[9])
    uint32_t* d_rank = nullptr;                // iota scratch of the index build
    uint32_t* d_level_parts = nullptr;         // [PLACE_WORKGROUPS][TREE_MAX_LEVELS] partial counts of the rebuild by cell runs (lio_rebuild.h)
    bool test_live_miscount = false;           // LIO_TEST_LIVE_MISCOUNT=1 (tests): a full rebuild starts from a live count that is off by one
    bool rebuild_by_cells = true;              // LIO_REBUILD_BY_CELLS=0 (A/B runs of scripts) or lio_debug_set("library_sort"): the point sort
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    uint32_t* d_counters = nullptr;            // [0] n_unresolved, [2..3] delete / build scratch, [4] run counts, [7] removed points, [16] walked queries of a search pass
    unsigned long long* d_cand = nullptr;
    float* d_boxes = nullptr;                  // up to 64 boxes
    uint32_t n_points = 0, n_live = 0, n_cells = 0;   // slots in use (live + tomb-stones + retired copies), live points, cells
    size_t slots_cap = 0;                      // physical slots of d_pts / d_pts_tmp (cap_map + room for re-written buckets)
    uint32_t n_retired = 0;                    // TOMB_MOVED slots since the last full rebuild
    uint32_t n_dead = 0;                       // box-deleted slots since the last full rebuild
    long long n_incremental_updates = 0, n_full_rebuilds = 0, n_boundary_points = 0;
    // LIO_TRACE_HOST=1 in the environment: host time stamps of the measurement passes (begin entry / launches issued / end entry /
    // record seen, with the number of polls: 0 polls = the GPU had finished before the host asked), dumped by lio_destroy
    bool trace_host = false;
    struct HostStamp { char tag; long long ns; unsigned polls; };
    std::vector<HostStamp> host_trace;
    void stamp(char tag, unsigned polls = 0) {
        if (!trace_host) return;
        if (host_trace.size() >= 4096) host_trace.erase(host_trace.begin(), host_trace.begin() + 2048);   // (keeps the most recent stamps)
        host_trace.push_back({tag, (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(), polls});
    }
    bool trace_rebuilds = false;               // LIO_TRACE_REBUILDS=1 in the environment: one stderr line per budget-triggered rebuild
    long long dbg_retired_budget = 0;          // lio_debug_set("retired_budget"): overrides the budget of retired slots (tests)
    uint32_t n_new_cells = 0, n_cells_built = 0;   // cells inserted since / present at the last full rebuild
    uint32_t level_entries[TREE_MAX_LEVELS]{};     // entries of each hashed level at the last full rebuild
    // An index update in the making (Add_Points called, the index not yet updated): the survivor slots of ONE down-sampled
    // batch (d_add_out[0 .. 2 fold_n): the fold's, then the sequential replay's; their number is still on the device) and ONE
    // plain batch (plain_ptr, plain_n points).  apply_pending builds the pending list d_pend = [plain][survivors] from them.
    float4* d_pend = nullptr;
    size_t cap_pend = 0;
    uint32_t fold_n = 0;                       // points of the down-sampled batch (0: none); survivors <= fold_n
    const float4* plain_ptr = nullptr;
    uint32_t plain_n = 0;
    bool pend_built = false;                   // d_pend holds [plain_n][d_apply[6] survivors]
    HostBox batch_box{};                       // bounding box of the batch(es): the frames of the compressed keys
    int last_add_counter = 0;                  // tmp_counter of the last down-sampled Add_Points (ikd_Tree.cpp:414-489)
    uint64_t *d_pkeys = nullptr, *d_pkeys2 = nullptr, *d_gs_part = nullptr, *d_gs_part2 = nullptr;
    uint32_t *d_porder = nullptr, *d_porder2 = nullptr, *d_pnewoff = nullptr;
    CellPlan* d_plan = nullptr;
    uint32_t* d_apply = nullptr;               // [0..2] apply stats, [3] new slots, [4] fold: replaced, [5] fold: tmp_counter,
                                               // [6] survivors, [7] pending points, [10] boundary points, [11..13] n_add / n_plain / flags of the
                                               // BatchDesc, [14] key-frame violations, [15] compaction time-outs; [32..47] the BatchDesc itself
    bool dbg_fast_update = true;               // LIO_FAST_UPDATE=0: lio_map_incremental takes the host round trip between its lists and their
                                               // Add_Points again (A/B runs of scripts)
    bool dbg_publish_in_kernel = true;         // LIO_PUBLISH_IN_KERNEL=0 (A/B runs of scripts)
    long long n_fast_updates = 0;              // map updates that ran without it
    long long n_fast_update_overflows = 0;     // ... that found a list longer than their launches were sized for (redone with the round trip)
    // list lengths of the last eight lio_map_incremental calls: the next one's launches are sized by their maxima (a drive that
    // alternates between places -- bench.py's rotation of eight scans -- predicts badly from the previous scan alone)
    uint32_t pred_add_ring[8] = {}, pred_plain_ring[8] = {};
    int pred_ring_pos = 0;
    bool pred_lists_valid = false;
    void pred_record(uint32_t n_add, uint32_t n_plain) {
        pred_add_ring[pred_ring_pos] = n_add; pred_plain_ring[pred_ring_pos] = n_plain;
        pred_ring_pos = (pred_ring_pos + 1) & 7; pred_lists_valid = true;
    }
    uint32_t pred_max(const uint32_t (&r)[8]) const { uint32_t m = 0; for (uint32_t v : r) m = v > m ? v : m; return m; }
    uint32_t* d_ghist = nullptr;               // two partition histograms of the group sort (lio_gsort.hip), used alternately
    int ghist_cur = 0;
    bool ghist_dirty = false;                  // a key kernel filled a histogram whose sort never ran
    unsigned long long* d_gagg = nullptr;      // stamped tile counts of the group sort's partition
    uint32_t gsort_seq = 0;
    size_t cagg_words = 0;                     // size of d_cagg
    bool have_map = false;
    bool map_broken = false;                   // an index update failed half-way: every map / search call fails until Build / reconstruct
    unsigned long long* d_cagg = nullptr;      // scratch of the single-launch compaction primitives (lio_compact.hip)
    uint32_t compact_seq = 0;                  // their stamp: one number per launch
    // acquire_removed_points: box-deleted points whose storage was reclaimed by a rebuild, not yet handed out
    float4* d_removed = nullptr;
    size_t cap_removed = 0;
    uint32_t n_removed = 0;
    uint32_t box_deleted_pending = 0;          // box tomb-stones still inside d_pts

    // local-map cube of lasermap_fov_segment (laserMapping.cpp:1132-1133)
    float localmap_min[3] = {0.f, 0.f, 0.f}, localmap_max[3] = {0.f, 0.f, 0.f};
    bool localmap_initialized = false;

    // ---- scan -------------------------------------------------------------------------------
    int cap_q = 0, nq = 0;
    float4 *d_body = nullptr, *d_world = nullptr, *d_nbr = nullptr, *d_normvec = nullptr, *d_plane = nullptr;
    // pipelined hand-over of the next scan (lio_scan_upload_async / lio_scan_swap): a second scan buffer filled on a
    // copy stream while the current scan is being processed
    float4* d_body_back = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy = nullptr;
    int nq_back = -1;                          // points in the back buffer (-1: nothing uploaded)
    bool upload_pending = false;               // lio_scan_upload_async was called, the copy has not been issued yet
    const void* upload_src = nullptr;
    int upload_n = 0;
    bool pass_in_flight = false, pass_empty = false;   // between lio_measure_begin and lio_measure_end
    unsigned long long pass_seq = 0;           // sequence number of the pass in flight (its record carries it)
    // ---- passes enqueued ahead of their pose (the iterated update only; lio_common.h GateState) ----------------------
    static constexpr int GATE_SLOTS = 4;       // (a power of two > GATE_AHEAD)
    static constexpr int GATE_AHEAD = 2;       // passes armed at most: a short pass is too short to arm the one behind it
    GateMail* h_gate_mail = nullptr;           // [GATE_SLOTS] host-mapped, coherent: the host's word to an armed gate
    GateMail* d_gate_mail = nullptr;           // their device address
    bool gate_mail_in_vram = false;            // the slots are fine-grained device memory the CPU writes through the BAR (lio_create)
    GateState* d_gate = nullptr;               // [GATE_SLOTS] device: what a gate leaves for the kernels behind it
    struct ArmedPass {
        bool search, ext; unsigned long long mail_seq, pass_seq;
        bool hint = false;                     // the bound its search was enqueued with (hint_valid at that moment)
        bool across = false;                   // armed across a scan boundary (stays set; next_scan is cleared by lio_scan_swap)
        bool next_scan = false;                // the FIRST pass of the scan in the back buffer (gate_arm_next_scan): survives the end of
        const float4* body = nullptr; int nq = 0;   // the update and lio_scan_swap (which checks that it swaps in this buffer, this many points)
    };
    ArmedPass gate_q[GATE_AHEAD];              // gates + passes that sit on the stream waiting for the host, in stream order
    int gate_n = 0;
    unsigned long long gate_mail_seq = 0;      // sequence number of the gate armed last (its slot: seq % GATE_SLOTS)
    int dbg_pre_enqueue = 1;                   // lio_debug_set("pre_enqueue", 0): never enqueue ahead
    int dbg_gate_mispredict = 0;               // lio_debug_set("gate_mispredict", 1): predict the WRONG kind every time (tests of the call-off path)
    long long n_gate_released = 0, n_gate_called_off = 0;
    std::vector<signed char> pattern_prev, pattern_cur;   // converge flag per call of the previous / the running iterated update
    bool pass_searched = false;                // the pass in flight is a search pass
    bool pass_gated = false;                   // ... and was released through a gate (re-enqueued directly once if its gate gave up)
    bool pass_hint = false;                    // the bound the pass in flight was armed / launched with: what a re-enqueue must use again
    long long n_gate_reenqueued = 0;
    int gate_in_kernel = 3;                    // bit 0: passes that re-use neighbours, bit 1: search passes (bit 2: also the one armed across a scan boundary) -- wait for the host's word inside
                                               // their first kernel instead of behind a gate kernel (mail slots in device memory only;
                                               // LIO_GATE_IN_KERNEL / lio_debug_set("gate_in_kernel"))
    long long n_gate_in_kernel = 0;
    bool arm_across_scans = false;             // lio_eskf_scan_step_pipelined: the first search of the NEXT scan (already uploaded) goes onto the
    int dbg_arm_next_scan = 1;                 // stream behind the last pass of this one; lio_debug_set("arm_next_scan", 0) / LIO_ARM_NEXT_SCAN=0: never
    long long n_next_scan_armed = 0, n_next_scan_released = 0;
    bool across_off_this_scan = false;         // this scan's update already ran past an end the previous one predicted: no second guess
    long long n_rows_direct_dma = 0;           // lio_measure_rows calls whose rows the copy engine wrote straight into the caller's page-locked buffers
    // a scan handed over as page-locked host records (lio_scan_set_async) whose pull has not been issued yet: the first search
    // pass of the scan pulls it inside knn27_kernel; anything else that needs d_body first launches the stand-alone pull
    bool pull_pending = false;
    const void* pull_alias = nullptr;          // device alias of the host records
    int pull_stride = 0, pull_n = 0;
    // assisted pull (enqueue_pass): while the first search of a scan pulls the back of the host cloud as raw records, this thread
    // packs the front into 16-byte records the search reads instead -- a third of the PCIe bytes for every record packed in time
    const char* pull_host = nullptr;           // the host address behind pull_alias
    float4* h_pack = nullptr;                  // [cap_q] page-locked, coherent staging of the packed front
    const float4* d_pack = nullptr;            // its device alias
    unsigned long long* h_pack_progress = nullptr;   // stamp << 32 | records packed: the word behind the gates' mail slots
    const unsigned long long* d_pack_progress = nullptr;
    uint32_t pack_stamp = 0;
    double assist_frac = 0.78;                 // share of a scan the host packs: steered by the measured time of the scans' first passes (assist_share)
    int assist_side = -1, assist_next_side = 0; // this scan's share: a little below (0) / above (1) assist_frac; -1 = not steered
    double assist_acc[2] = {0.0, 0.0};         // ns per record from the launch of an assisted first pass to its record, per side
    int assist_cnt[2] = {0, 0};
    double assist_t0_ns = 0.0;                 // launch of the assisted pass in flight (0 = none)
    double pack_ns_per_record = 1.0;           // this host's packing speed (reported; warm caches ~0.45, from DRAM ~1.0)
    int dbg_assist_pull = 1;                   // lio_debug_set("assist_pull"): 0 = off, 1 = share by the running estimate, 2..90 = that share of the scan in per cent
    long long n_assisted_pulls = 0, last_pack_front = 0;
    int dbg_assist_chunk = 0;                  // (0 = by scan size) lio_debug_set("assist_pull_chunk"): records per progress announcement (a multiple of 64)
    long long n_helper_late = 0;               // scans whose second half the calling thread packed itself (the helper had not come)
    int assist_threads = 1;                    // lio_scan_pack_threads: 1 = the calling thread alone, 2 = + the helper
    PackHelper* pack_helper = nullptr;
    unsigned long long pack_job_id = 0;
    int dbg_assist_stall_us = 0;               // test hook: this thread sleeps that long before it packs (the search falls back to raw records)
    double last_pack_us = 0.0;
    int dbg_fused_pull = 1;                    // lio_debug_set("fused_pull", 0): always the stand-alone pull kernel (A/B switch)
    // the pass in flight pulled its scan out of the caller's records: its record carries the first and the last point the GPU saw
    // (DeviceOut::probe) and lio_measure_end compares them with the host's copy -- a registration that went stale (the node freed and
    // re-allocated its cloud at the same address: the old pages stay pinned, the GPU reads THEM) is caught, the scan uploaded again
    // through the staging buffer and the pass repeated ("stale_pulls")
    bool scan_pull_failed = false;             // the stand-alone pull of a scan could not be launched: the scan is gone (the next pass says so)
    bool pull_check = false;
    const char* pull_check_host = nullptr;
    int pull_check_stride = 0, pull_check_n = 0;
    long long n_stale_pulls = 0;
    int dbg_pull_alias_shift = 0;              // test hook: the fused pull reads this many records further on (what a stale registration looks like)
    long long n_scan_pulled = 0;               // scans a kernel pulled out of page-locked host records (lio_scan_set_async)
    // page-locked destinations the rows of the callback path are copied into directly (lio_measure_rows)
    int listed_last_search = 0;                // unproven queries of the most recent search pass: predicts the next one's
    uint8_t *d_cnt = nullptr, *d_sel = nullptr;
    float4* d_ulist = nullptr;                 // queries the 27-cell kernel could not prove (count: d_counters[0])
    unsigned long long* d_useeds = nullptr;    // [list entry][8]: keys of their 27-cell results, the seeds of the walk; [5] = its verdict
    unsigned long long* d_lunits = nullptr;    // [FIX_MAX][RED_STRIDE][2]: stamped terms of the listed queries (search passes with few of them)
    double* d_rows = nullptr;
    // the unmodified-callback path (lio_measure_want_rows): every pass also leaves h_x / h compacted and column-major
    bool want_rows = false;
    bool rows_staged = false;                  // the last pass did
    double* d_rows_c = nullptr;                // device copy of the compacted rows (large scans: moved by the copy engine)
    double* h_rows = nullptr;                  // pinned + mapped: the kernel writes small results straight into it
    double* d_rows_mapped = nullptr;           // device alias of h_rows
    int dbg_rows_direct = -1;                  // lio_debug_set("rows_direct"): -1 auto, 0 always through d_rows_c, 1 always direct
    unsigned long long *d_units = nullptr, *d_gunits = nullptr;   // stamped partial sums of the residual kernel's reduction
    DeviceOut* h_out = nullptr;                // pinned + mapped: the final reduction writes it directly
    DeviceOut* d_out = nullptr;                // device alias of h_out
    unsigned long long seq = 0;                // sequence number of the last pass (polled in h_out->seq)
    bool have_scan = false, have_search = false, have_pass = false;
    bool hint_valid = false;                   // d_nbr holds live map points of THIS map state for THIS scan
    Pose14 last_pose{};
    bool last_ext = false;
    int last_m = 0;

    // ---- staging / scratch --------------------------------------------------------------------
    float4* h_stage = nullptr;                 // pinned upload staging
    hipEvent_t ev_stage = nullptr;             // recorded behind an asynchronous copy out of h_stage
    void* h_small = nullptr;                   // 256 host-mapped bytes: 32 words of counters + the sequence word the host polls
    uint32_t* d_small = nullptr;               // device alias of h_small
    uint32_t small_seq = 0;
    bool stage_busy = false;
    hipStream_t stage_stream = nullptr;        // the stream ev_stage was last recorded on (one event covers the buffer: lio_scan_compensate)
    size_t stage_busy_slots = 0;               // float4 slots at the head of h_stage the copy in flight still reads
    bool undist_sorted = true;                 // d_undist is in time order (else: input order, sorted on demand)
    bool raw_time_keyed = false;               // the raw scan came with per-point times
    bool raw_uploaded = false;                 // lio_scan_upload_raw ran (lio_scan_compensate may follow)
    size_t h_stage_cap = 0;
    void* d_scratch = nullptr;                 // download staging (device side)
    size_t d_scratch_bytes = 0;
    void* h_scratch = nullptr;                 // pinned download staging
    size_t h_scratch_bytes = 0;
    // query buffers of lio_map_nearest_search
    int cap_qs = 0;
    float4 *d_qs_world = nullptr, *d_qs_nbr = nullptr;
    uint8_t* d_qs_cnt = nullptr;
    float4* d_qs_ulist = nullptr;
    unsigned long long* d_qs_useeds = nullptr;

    // raw-scan preparation (lio_scan_undistort / lio_scan_voxel_filter)
    size_t cap_raw = 0;
    int n_raw = 0;
    bool have_raw = false;
    float4 *d_raw = nullptr, *d_undist = nullptr, *d_leafpts = nullptr;
    uint64_t *d_raw_keys = nullptr, *d_raw_keys2 = nullptr, *d_leaf_keys = nullptr;
    uint32_t *d_raw_order = nullptr, *d_raw_order2 = nullptr, *d_leaf_count = nullptr, *d_leaf_start = nullptr;
    double* d_imu_poses = nullptr;             // up to 512 Pose6D records (22 doubles each)
    uint32_t* d_undist_parts = nullptr;        // [ceil(cap_raw / 256)][8] extremes + time-order flag per workgroup of undistort_kernel (lio_scanprep.hip)
    uint32_t undist_parts_n = 0;               // records that describe d_undist as it is now (0: none, the filter runs minmax_kernel)
    bool prep_fused = true;                    // LIO_PREP_FUSED=0 (A/B runs of scripts): minmax / run heads as launches of their own
    VoxelGridDev* d_vox = nullptr;             // two records, used alternately: the centroid kernel of a call clears the other one
    int vox_cur = 0;
    bool vox_clean[2] = {false, false};
    bool vox_saw_unsorted = false;             // the previous time-keyed sweep had a point earlier than its predecessor
    int vox_passes_guess = 3;                  // 8-bit splits the previous sweep's leaf index needed (22-24 bits at 0.5 m leaves)
    long long n_voxel_redone = 0;              // sweeps filtered twice (a wrong prediction)

    // asynchronous scan preparation (lio_scan_prefetch_raw / lio_scan_prep_begin / _end, lio_scan.hip)
    float4* d_raw_back = nullptr;              // destination of a prefetched sweep (copy stream)
    const void* raw_prefetch_src = nullptr;
    int raw_prefetch_n = 0;
    bool raw_prefetched = false;
    hipEvent_t ev_raw = nullptr;
    hipStream_t prep_stream = nullptr;         // = copy_stream (ensure_prep, lio_scan.hip)
    bool prep_active = false, prep_filtered = false, prep_enqueued = false;
    VoxelRun prep_run;                         // the filter enqueued there: what lio_scan_prep_end judges
    unsigned long long *d_cagg_prep = nullptr, *d_gagg_prep = nullptr;   // the preparation stream's own stamped scratch
    uint32_t compact_seq_prep = 0, gsort_seq_prep = 0;
    void* h_small3 = nullptr;                  // its mail box (the filter's summary)
    uint32_t* d_small3 = nullptr;
    uint32_t small3_seq = 0;
    long long n_raw_prefetch_hits = 0, n_prep_overlapped = 0;

    // batch buffers of lio_map_add
    size_t cap_add = 0;
    float4 *d_add_pts = nullptr, *d_add_pts2 = nullptr, *d_add_out = nullptr;
    uint8_t* d_add_cls = nullptr;
    uint8_t *d_add_dirty = nullptr, *d_add_dirty_pt = nullptr;   // boundary points of a down-sampled batch (lio_mapadd.hip)
    uint32_t* d_add_dlist = nullptr;
    uint2* d_add_parts = nullptr;              // [16385] the fold's counters, one pair per workgroup of fold_commit_kernel
    float4* d_add_plan = nullptr;              // [2 n] the speculative fold's plan per head position (lio_mapadd.hip)
    uint2* d_add_ranges = nullptr;             // [8 n] the bucket ranges of its eight lanes
    uint64_t *d_add_keys = nullptr, *d_add_keys2 = nullptr;
    uint32_t *d_add_order = nullptr, *d_add_order2 = nullptr;

    // ---- replica exchange (lio_replicate_*) ---------------------------------------------
    void* rep_comm = nullptr;                  // ncclComm_t (RCCL), created by lio_replicate_init
    int rep_rank = 0, rep_world = 0;           // rep_world == 0: not initialised
    uint32_t rep_cap = 0;                      // points per list and message
    float4 *d_rep_send = nullptr, *d_rep_recv = nullptr;
    float rep_boxes[6 * REP_MAX_BOXES]{};      // delete boxes queued for the next exchange
    int rep_n_boxes = 0;
    // ---- the DEFERRED schedule of the exchange (lio_replicate_defer): the gathered lists of exchange k enter the map at the
    // START of exchange k + 1 -- after scan step k + 1 has searched the map without them -- so that the read-only half of their
    // Add_Points (voxel keys, the grouping sort, the speculative fold: fold_batch_plan) runs on a SIDE stream beside scan step
    // k + 1 and only the half that writes the map (fold_commit, the index update) stays between two scan steps.
    bool rep_defer = false;
    struct DeferredBatch {
        bool pending = false;                  // a gathered batch waits to enter the map
        bool plan_owed = false;                // its plan may run on the side stream: the next scan step launches it (lio_measure_begin)
        bool planned = false;                  // ... and has been launched there (ev_side says when it is done)
        bool joined = false;                   // ... and the main stream has been ordered behind it (enter / the settle)
        uint32_t n_add = 0, n_plain = 0, n_boxes = 0;
        HostBox box;
        FoldArgs fa{};
        BoundaryArgs ba{};
    } def;
    float4* d_def_add[2] = {nullptr, nullptr};     // gathered PointToAdd lists: [def_cur] = the deferred batch, [def_cur ^ 1] = the next gather
    float4* d_def_plain[2] = {nullptr, nullptr};   // gathered PointNoNeedDownsample lists
    float* d_def_boxes[2] = {nullptr, nullptr};    // gathered delete boxes
    size_t cap_def = 0;
    int def_cur = 0;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_side = nullptr;
    void* h_small2 = nullptr;                  // a second host-mapped mail box (the totals of an exchange, read after the settle's own wait)
    uint32_t* d_small2 = nullptr;
    uint32_t small2_seq = 0;
    long long n_plans_on_side_stream = 0, n_settled = 0;
    // Round 5: the COMMIT half of the deferred batch (fold_commit ... cell_apply) on the side stream too -- right behind its plan --
    // beside the part of the NEXT exchange that does not touch the map: this rank's list kernel, the message, the all-gather (its wire
    // time on real hardware), the compaction of the gathered lists.  lio_replicate_step starts it, reads its statistics from a mail
    // box of their own where the blocking settle used to sit, and only then lets anything else onto the main stream.
    bool settle_side = false;                  // the chain is on the side stream, its 32 statistics words on their way to mail box 4
    uint32_t settle_seq = 0;
    void* h_small4 = nullptr;
    uint32_t* d_small4 = nullptr;
    uint32_t small4_seq = 0;
    unsigned long long* d_cagg_lists = nullptr;   // stamped scratch of the list kernel while the commit's scans use d_cagg
    uint32_t compact_seq_lists = 0;
    long long n_settled_beside_exchange = 0;
    int dbg_settle_beside = 1;                 // lio_debug_set("settle_beside_exchange", 0): the blocking settle of round 4 (A/B switch)
    int def_last_counter = 0, def_last_deleted = 0;   // tmp_counter / deleted points of the batch the last exchange let into the map
    // loop-back test mode (lio_debug_set "replicate_loopback", then lio_replicate_init without an id): rep_world VIRTUAL
    // ranks on this one GPU.  lio_replicate_step as virtual rank r < world - 1 ("replicate_loopback_rank") parks that
    // rank's lists, counts and boxes; as rank world - 1 it packs every parked rank's message into its slot of the gathered
    // buffer (what ncclAllGather would deliver) and goes on exactly like a real rank: compact, read-back, rank-ordered apply.
    int dbg_rep_loopback = 0;
    bool rep_loopback = false;
    float4* d_rep_stage = nullptr;             // [world - 1][2][cap_q] parked lists
    uint32_t* d_rep_stage_cnt = nullptr;       // [world - 1][2] their counts
    float rep_stage_boxes[REP_MAX_BOXES_TOTAL / REP_MAX_BOXES][6 * REP_MAX_BOXES]{};
    int rep_stage_nb[REP_MAX_BOXES_TOTAL / REP_MAX_BOXES]{};
    unsigned rep_staged = 0;                   // bit r: virtual rank r has parked its lists for the next exchange

    // ---- profiling ----------------------------------------------------------------------------
    // one HIP-event pair per timed launch, stamped by the dispatch (hipExtLaunchKernelGGL); resolved when
    // the totals are read, so that timing a run does not add a stream synchronisation per pass
    unsigned prof = 0;                         // bit w set: time kernel family w
    bool census = false;                       // count C(q) in the k-NN kernel (one atomic per wavefront)
    int dbg_force_timeout = 0;                 // lio_debug_set("force_reduction_timeout")
    int dbg_knn_tune = 0;                      // lio_debug_set("knn_tune")
    long long n_voxel_fast = 0;                // voxel-filter calls that took the compact-key path (lio_debug_get("voxel_fast_runs"))
    int dbg_voxel_fast = 1;                    // lio_debug_set("voxel_fast_path"): the compact-key LSD sort of time-ordered sweeps (A/B switch)
    int dbg_knn_walk = -1;                     // lio_debug_set("knn_walk") / LIO_KNN_WALK: -1 = by scan size (auto_walk), 0 / 1 / 2 / 4 = KnnArgs.walk
    int dbg_library_sort = 0;                  // lio_debug_set("library_sort"): rocPRIM for every sort / compaction (A/B switch)
    int dbg_stamps = 0;                        // lio_debug_set("stamps"): the last workgroup takes wall_clock64 stamps (lio_debug_stamps)
    int dbg_knn_stamps = 0;                    // lio_debug_set("knn_stamps"): searches run the stamped instantiation of knn27_kernel (lio_debug_knn_stamps)
    unsigned long long* d_knn_stamps = nullptr;   // [workgroups][KNN_STAMP_WORDS]
    int knn_stamps_wgs = 0;                    // workgroups of the last stamped launch
    int dbg_separate_walk = 0;                 // lio_debug_set("separate_tree_walk"): the stand-alone knn_tree_kernel per search pass
    std::vector<hipEvent_t> ev_pool[LIO_PROF_COUNT];   // start, stop, start, stop, ...
    size_t ev_next[LIO_PROF_COUNT]{};          // events of the pool in use
    double prof_ms[LIO_PROF_COUNT]{};
    long long prof_n[LIO_PROF_COUNT]{};

    int fail(int code, const char* what, const char* detail) {
        err = std::string(what) + ": " + (detail ? detail : "");
        return code;
    }
};

#define HIPCHK(ctx, expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return (ctx)->fail(LIO_ERR_HIP, #expr, hipGetErrorString(_e));  \
    } while (0)

#define LIO_INTERNAL __attribute__((visibility("hidden")))

// Shape of a knn27_kernel launch: lanes per query and candidate walk (KnnArgs.walk, lio_knn.hip).  Measured on MI355X
// (scripts/dbg/knn_walk_ab.py: 5 M-point map, scans of 3 k ... 100 k points, first search at the prior and bounded re-search at the
// converged pose; profiles/r03z_walk_ab.log, r03z_walk_ab_lanes4.log; first / bounded search in us).  The kernel is bound by
// instruction issue when the launch fills the GPU and by its chain of dependent round trips when it does not:
//   lanes  8 per query from ~22 k queries up (C2, C5; at 24 k: 10.3 / 11.1 with 8 lanes, 10.6 / 11.8 with 16), 16 below (8.6 k:
//          7.8 / 8.7 vs 8.6 / 9.3), 32 for very small scans.  4 lanes lose the first search everywhere (100 k: 23.2 vs 21.9) and
//          both searches below ~60 k queries; at 100 k they win the BOUNDED search by 0.7 us with four candidates in flight (21.2 vs
//          21.9) and lost it with the plain walk inside bench.py's drive (24.7 vs 23.3): not adopted (0.6 % of a scan step)
//   walk   the prefix-form list with four candidates per lane in flight wins the first search at every size (100 k: 22.8 -> 22.0;
//          50 k: 14.8 -> 14.0; 24 k: 11.1 -> 10.3; 8.6 k, 16 lanes: 8.2 -> 7.8) and the bounded one from 50 k down (14.9 -> 14.6;
//          11.9 -> 11.1; 8.9 -> 8.7; at 100 k it is a wash: 22.2 -> 22.1), and C5's 1 M queries against a map beyond the Infinity
//          Cache (bench.py --config c5, A/B on one box: 207.4 -> 196.3 and 164.9 -> 160.0 us).
static inline int auto_walk(int nq, int lanes) {
    (void)nq;
    return (lanes == 4 || lanes == 8 || lanes == 16) ? 4 : 0;
}
static inline int auto_lanes(int nq) {
    if (nq >= 22000) return 8;
    if (nq >= 2000) return 16;
    return 32;
}

constexpr int FUSED_TRANSFORM_MAX_QUERIES = 262144;   // scans beyond it: transform / pull as launches of their own (run_knn, lio_scan_set_async)
// listed (unproven) queries of the last search pass above which the next one walks them in a launch of its own
constexpr int INLINE_WALK_MAX_LISTED = 1024;
constexpr size_t POSE_STAGE_SLOTS = 512 * sizeof(lio_pose6d) / sizeof(float4);   // room for IMUpose behind the points
constexpr uint64_t RETIRED_PER_LIVE = 4;   // retired slots (copies of re-written buckets) the map may carry per live point before a full rebuild

// grouping sorts of map maintenance (lio_gsort.hip): which implementation
enum SortKind { SORT_LIBRARY, SORT_RANK, SORT_LARGE };
// the bounding-box accumulators of the insert lists (d_counters[24..29]) in their "empty" state
static const uint32_t BBOX_RESET[6] = {BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT};

// ---- internal entry points shared by the translation units (hidden: not part of the ABI) ----------------------------------
extern "C" {
// lio_ctx.hip
LIO_INTERNAL hipError_t publish_and_wait(lio_ctx* c, const uint32_t* d_src, int n_words);
LIO_INTERNAL hipError_t stream_wait(lio_ctx* c);
LIO_INTERNAL hipError_t wait_published(lio_ctx* c, uint32_t seq);
LIO_INTERNAL hipError_t read_back(lio_ctx* c, void* host_dst, const void* d_src, size_t bytes);
LIO_INTERNAL int ensure_stage(lio_ctx* c, size_t n);
LIO_INTERNAL int ensure_scratch(lio_ctx* c, size_t bytes);
LIO_INTERNAL int upload_points(lio_ctx* c, const float* pts, size_t n, int stride_bytes, float4* d_dst);
LIO_INTERNAL uint32_t next_compact_stamp(lio_ctx* c);
LIO_INTERNAL hipError_t sort_pairs_any(lio_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                                       size_t n, hipStream_t s, uint32_t* err = nullptr);
LIO_INTERNAL bool compact_small(const lio_ctx* c, size_t n);
LIO_INTERNAL int check_compact_error(lio_ctx* c, uint32_t flag_word);
LIO_INTERNAL SortKind sort_kind(const lio_ctx* c, size_t n_bound, int kb);
LIO_INTERNAL int sort_hist(lio_ctx* c, SortKind k, uint32_t** hist);
LIO_INTERNAL int run_sort(lio_ctx* c, SortKind k, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                          size_t n_bound, const uint32_t* n_dev, uint32_t n_dev_add, int kb, const uint32_t* kb_dev = nullptr);
LIO_INTERNAL hipError_t compact_live_any(lio_ctx* c, const float4* src, uint32_t n, float4* dst, uint32_t* n_out);
LIO_INTERNAL bool is_pinned_host(const void* p);
LIO_INTERNAL void prof_collect(lio_ctx* c);
LIO_INTERNAL void dump_host_trace(lio_ctx* c, size_t last);
LIO_INTERNAL float auto_cell_size(float ds);
LIO_INTERNAL bool is_pow2f(float v);
// lio_update.hip
LIO_INTERNAL int deferred_join_side(lio_ctx* c);
LIO_INTERNAL int deferred_settle(lio_ctx* c, int* add_counter, int* n_deleted);
LIO_INTERNAL int deferred_launch_plan(lio_ctx* c);
LIO_INTERNAL uint32_t retired_slot_budget(const lio_ctx* c);
// lio_scan.hip
LIO_INTERNAL int scan_reset(lio_ctx* c, int n);
LIO_INTERNAL int issue_pending_upload(lio_ctx* c);
// lio_pull.hip
LIO_INTERNAL void flush_pull(lio_ctx* c);
LIO_INTERNAL void pack_helper_stop(lio_ctx* c);
LIO_INTERNAL int assist_share(lio_ctx* c, int nq);
LIO_INTERNAL int assist_post(lio_ctx* c, int front);
LIO_INTERNAL void assist_pack(lio_ctx* c, int front, int mid);
// lio_pass.hip
LIO_INTERNAL void gate_call_off(lio_ctx* c);
LIO_INTERNAL void gate_call_off_this_scan(lio_ctx* c);
LIO_INTERNAL int run_knn(lio_ctx* c, const float4* body, float4* queries_world, const Pose14& pose, int nq, int stride,
                         float4* nbr, uint8_t* cnt, float4* ulist, unsigned long long* useeds, bool use_hint, bool walk_now, const GateState* gate = nullptr,
                         bool pull_raw = false, const GateMail* mail = nullptr, unsigned long long expect = 0ull, int pack_front = 0, int pack_mid = 0);
}  // extern "C"
LIO_INTERNAL KernelTimer prof_timer(lio_ctx* c, int which);     // the next event pair of a family (empty timer when the family is not being timed)

// Passes armed ahead of their pose (gate_arm) sit on the stream behind a gate kernel that waits for the host: whoever else wants
// the stream -- every entry point that enqueues on it or waits for it -- calls them off first.
// ... and a scan whose pull out of the caller's host records is still owed (lio_scan_set_async) is pulled before anything but
// its first search pass looks at the scan buffer.
static inline void enter(lio_ctx* c) {
    if (c->gate_n > 0) gate_call_off(c);
    if (c->pull_pending) flush_pull(c);
    // the plan of a deferred batch on the side stream (deferred_launch_plan) uses the context-wide stamped scratch of the sorts and
    // compactions (d_cagg / d_gagg / d_ghist / d_temp / d_apply): whatever an entry point puts on the main stream from here on is
    // ordered behind it (usually the plan finished long ago: one event query, no dependency enqueued)
    if (c->def.planned) (void)deferred_join_side(c);
}
// ... and whoever reads or changes the MAP (everything but the measurement passes and the scan hand-over) first lets a gathered
// batch the deferred exchange schedule still holds back enter it (lio_replicate_defer): outside lio_replicate_step the map is
// always the map with every exchanged list applied.
static inline int enter_map(lio_ctx* c) {
    enter(c);
    if (c->def.pending) return deferred_settle(c, nullptr, nullptr);
    return LIO_OK;
}
