// lio_ctx.hip -- host side of liblio_hip.so: the context object and the C ABI of include/lio_hip.h.
// One context = one GPU = one HIP stream; every device buffer is allocated once at lio_create
// (capacities from lio_config) so that a scan step performs no allocation.
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
static void pack_records(const PackJob& j) {
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
                                               // [6] survivors, [10] boundary points, [14] key-frame violations, [15] compaction time-outs
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
    VoxelGridDev* d_vox = nullptr;             // two records, used alternately: the centroid kernel of a call clears the other one
    int vox_cur = 0;
    bool vox_clean[2] = {false, false};
    bool vox_saw_unsorted = false;             // the previous time-keyed sweep had a point earlier than its predecessor
    int vox_passes_guess = 3;                  // 8-bit splits the previous sweep's leaf index needed (22-24 bits at 0.5 m leaves)
    long long n_voxel_redone = 0;              // sweeps filtered twice (a wrong prediction)

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

namespace {

float auto_cell_size(float ds) {
    // largest power of two <= 2.5 * ds
    const float target = 2.5f * ds;
    float c = 1.0f;
    while (c > target) c *= 0.5f;
    while (c * 2.0f <= target) c *= 2.0f;
    return c;
}
bool is_pow2f(float v) {
    int e;
    return v > 0.0f && std::frexp(v, &e) == 0.5f;
}
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
int auto_walk(int nq, int lanes) {
    (void)nq;
    return (lanes == 4 || lanes == 8 || lanes == 16) ? 4 : 0;
}
int auto_lanes(int nq) {
    if (nq >= 22000) return 8;
    if (nq >= 2000) return 16;
    return 32;
}

// A few counters from the device: through pinned memory (a copy into pageable memory takes the runtime's slow
// staged path, ~3x the latency), then the one synchronisation the caller needs anyway.
// Waiting for the stream without entering the runtime's wait: a one-wavefront kernel at the end of the stream copies
// the counters the host wants (if any) into host-mapped memory and publishes a sequence number there; the host polls
// that word.  hipStreamSynchronize may park the thread on an interrupt (tens of microseconds to wake up) and
// hipStreamQuery costs a runtime call per poll; the mapped word costs a cache miss.  Bounded: after ~1 s of polling
// the runtime's wait takes over (and reports what went wrong, if something did).
static hipError_t publish_and_wait(lio_ctx* c, const uint32_t* d_src, int n_words) {
    const uint32_t seq = ++c->small_seq;
    hipError_t e = launch_publish_words(d_src, n_words, c->d_small, seq, c->stream);
    if (e != hipSuccess) return e;
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(c->h_small) + 32;
    for (uint64_t spins = 0; spins < (1ull << 28); ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        if ((spins & 0xfffffu) == 0xfffffu) {                                  // every ~1 M polls: is the stream still healthy?
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return q;
        }
    }
    return hipStreamSynchronize(c->stream);
}
static hipError_t stream_wait(lio_ctx* c) { return publish_and_wait(c, nullptr, 0); }
// A few counters from the device (bytes: a multiple of 4, at most 128)
static hipError_t read_back(lio_ctx* c, void* host_dst, const void* d_src, size_t bytes) {
    const hipError_t e = publish_and_wait(c, reinterpret_cast<const uint32_t*>(d_src), (int)(bytes / 4));
    if (e != hipSuccess) return e;
    std::memcpy(host_dst, c->h_small, bytes);
    return hipSuccess;
}
constexpr size_t POSE_STAGE_SLOTS = 512 * sizeof(lio_pose6d) / sizeof(float4);   // room for IMUpose behind the points
int ensure_stage(lio_ctx* c, size_t n) {
    if (c->stage_busy) {                       // an upload that did not wait for its own copy (lio_scan_undistort)
        HIPCHK(c, hipEventSynchronize(c->ev_stage));
        c->stage_busy = false;
    }
    if (n <= c->h_stage_cap) return LIO_OK;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, n * sizeof(float4), hipHostMallocDefault));
    c->h_stage_cap = n;
    return LIO_OK;
}
int ensure_scratch(lio_ctx* c, size_t bytes) {
    if (bytes > c->d_scratch_bytes) {
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr; c->d_scratch_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_scratch, bytes));
        c->d_scratch_bytes = bytes;
    }
    if (bytes > c->h_scratch_bytes) {
        if (c->h_scratch) (void)hipHostFree(c->h_scratch);
        c->h_scratch = nullptr; c->h_scratch_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->h_scratch, bytes, hipHostMallocDefault));
        c->h_scratch_bytes = bytes;
    }
    return LIO_OK;
}
// host records (x,y,z first, stride bytes) -> pinned float4 staging -> device.  The host records are consumed before the
// call returns; the copy out of the staging buffer is NOT waited for (whatever follows on the stream is ordered behind it, the
// next writer of the staging buffer waits for the event: ensure_stage) -- a blocking wait here cost every lio_scan_set the
// copy engine's ~22 us floor plus the transfer (scripts/micro/upload_probe.hip).
static void pack_points(const float* pts, size_t n, int stride_bytes, float4* dst) {
    const char* src = reinterpret_cast<const char*>(pts);
    if (stride_bytes == 16) {
        for (size_t i = 0; i < n; ++i) { float4 v = reinterpret_cast<const float4*>(src)[i]; v.w = 0.0f; dst[i] = v; }
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        const float* p = reinterpret_cast<const float*>(src + i * (size_t)stride_bytes);
        dst[i] = make_float4(p[0], p[1], p[2], 0.0f);
    }
}
int upload_points(lio_ctx* c, const float* pts, size_t n, int stride_bytes, float4* d_dst) {
    if (n == 0) return LIO_OK;
    int rc = ensure_stage(c, n);
    if (rc != LIO_OK) return rc;
    pack_points(pts, n, stride_bytes, c->h_stage);
    HIPCHK(c, hipMemcpyAsync(d_dst, c->h_stage, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage, c->stream));
    c->stage_busy = true; c->stage_busy_slots = n;
    return LIO_OK;
}

// stamp of the next single-launch compaction (lio_compact.hip): never 0 (the scratch starts zeroed), never repeated within
// 2^32 launches; *err lives in d_apply[15]
static uint32_t next_compact_stamp(lio_ctx* c) {
    if (++c->compact_seq == 0u) {
        // the 32-bit stamp has gone round: words stamped 2^32 launches ago must not pass for this launch's -- start over from
        // clean scratch (once in 2^32 launches; the stream orders the clear before the launch that follows)
        (void)hipMemsetAsync(c->d_cagg, 0, c->cagg_words * sizeof(unsigned long long), c->stream);
        ++c->compact_seq;
    }
    return c->compact_seq;
}
static int check_compact_error(lio_ctx* c, uint32_t flag_word) {
    if (flag_word == 0u) return LIO_OK;
    (void)hipMemsetAsync(c->d_apply + 15, 0, sizeof(uint32_t), c->stream);
    return c->fail(LIO_ERR_HIP, "map maintenance", "a bounded wait of a compaction kernel ran out");
}

// stable sort of (key, value) pairs: the single-launch kernel for the batch sizes of map maintenance and scan
// preparation, rocPRIM above them
static hipError_t sort_pairs_any(lio_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                                 size_t n, hipStream_t s, uint32_t* err = nullptr) {
    (void)err;
    if (!c->dbg_library_sort && n <= SORT_MAX_ITEMS) return launch_sort_pairs_small(keys_in, vals_in, (uint32_t)n, keys_out, vals_out, s);
    return sort_pairs_u64_u32(c->d_temp, c->temp_bytes, keys_in, keys_out, vals_in, vals_out, n, s);
}
static bool compact_small(const lio_ctx* c, size_t n) { return !c->dbg_library_sort && n <= COMPACT_MAX_ITEMS; }

// ---- grouping sorts of map maintenance (lio_gsort.hip): which implementation, and its partition histogram ----------------
enum SortKind { SORT_LIBRARY, SORT_RANK, SORT_LARGE };
static SortKind sort_kind(const lio_ctx* c, size_t n_bound, int kb) {
    if (c->dbg_library_sort) return SORT_LIBRARY;
    if (n_bound <= GSORT_RANK_MAX) return SORT_RANK;
    if (n_bound <= GSORT_MAX && kb + GSORT_INDEX_BITS <= 64) return SORT_LARGE;
    return SORT_LIBRARY;
}
// the histogram the key-generating kernel of a SORT_LARGE sort has to accumulate (zero on entry); nullptr otherwise
static int sort_hist(lio_ctx* c, SortKind k, uint32_t** hist) {
    *hist = nullptr;
    if (k != SORT_LARGE) return LIO_OK;
    if (c->ghist_dirty) {                       // a key kernel filled it and its sort never ran (an error path)
        HIPCHK(c, hipMemsetAsync(c->d_ghist, 0, 512 * sizeof(uint32_t), c->stream));
        c->ghist_dirty = false;
    }
    *hist = c->d_ghist + 256 * c->ghist_cur;
    c->ghist_dirty = true;
    return LIO_OK;
}
// stable sort by key of n_bound records of which *n_dev + n_dev_add are real (n_dev == nullptr: all); with SORT_LIBRARY the
// unused slots must hold keys that sort last.  Result in keys_out / vals_out.
static int run_sort(lio_ctx* c, SortKind k, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                    size_t n_bound, const uint32_t* n_dev, uint32_t n_dev_add, int kb) {
    hipStream_t s = c->stream;
    if (n_bound == 0) return LIO_OK;
    if (k == SORT_RANK) {
        HIPCHK(c, launch_sort_pairs_small(keys_in, vals_in, (uint32_t)n_bound, keys_out, vals_out, s, n_dev, n_dev_add));
    } else if (k == SORT_LARGE) {
        GsortArgs g{};
        g.keys = keys_in; g.vals = vals_in; g.n = (uint32_t)n_bound; g.n_dev = n_dev; g.n_dev_add = n_dev_add; g.kb = kb;
        g.keys_out = keys_out; g.vals_out = vals_out; g.part = c->d_gs_part; g.part2 = c->d_gs_part2;
        g.hist = c->d_ghist + 256 * c->ghist_cur; g.hist_other = c->d_ghist + 256 * (c->ghist_cur ^ 1);
        if (++c->gsort_seq == 0u) {                        // (stamp wrap-around: as next_compact_stamp)
            HIPCHK(c, hipMemsetAsync(c->d_gagg, 0, GSORT_AGG_WORDS * sizeof(unsigned long long), s));
            ++c->gsort_seq;
        }
        g.agg = c->d_gagg; g.stamp = c->gsort_seq; g.err = c->d_apply + 15;
        HIPCHK(c, launch_gsort_large(g, s));
        c->ghist_cur ^= 1;
        c->ghist_dirty = false;
    } else {
        HIPCHK(c, sort_pairs_u64_u32(c->d_temp, c->temp_bytes, keys_in, keys_out, vals_in, vals_out, n_bound, s));
    }
    return LIO_OK;
}
// the live points of the point array to the front of dst, order kept: three streaming launches (lio_compact.hip), or the library
static hipError_t compact_live_any(lio_ctx* c, const float4* src, uint32_t n, float4* dst, uint32_t* n_out) {
    if (c->dbg_library_sort) return select_live_points(c->d_temp, c->temp_bytes, src, dst, n_out, n, c->stream);
    // (tile counts and offsets live in the sort workspace, which is idle here)
    uint32_t* tile_cnt = reinterpret_cast<uint32_t*>(c->d_temp);
    uint32_t* tile_off = tile_cnt + (n / 2048u + 4u);
    if (((size_t)n / 2048u + 4u) * 2u * sizeof(uint32_t) > c->temp_bytes || (size_t)n / 2048u + 2u > COMPACT_MAX_ITEMS) return select_live_points(c->d_temp, c->temp_bytes, src, dst, n_out, n, c->stream);
    return launch_compact_live_large(src, n, dst, n_out, tile_cnt, tile_off, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, c->stream);
}
// frame of the integer coordinates lo .. hi (inclusive) of a batch: one spare coordinate per axis above hi, so that the
// all-ones key is never a point's (the group of the non-finite points, lio_mapadd.hip)
static KeyFrame frame_of(const int lo[3], const int hi[3]) {
    KeyFrame f{};
    f.ox = lo[0]; f.oy = lo[1]; f.oz = lo[2];
    int* b[3] = {&f.bx, &f.by, &f.bz};
    for (int i = 0; i < 3; ++i) {
        const long long extent = (long long)hi[i] - (long long)lo[i] + 2;
        int bits = 1;
        while ((1ll << bits) < extent) ++bits;
        *b[i] = bits;
    }
    return f;
}
static KeyFrame voxel_frame(const HostBox& box, float ds) {
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { lo[i] = voxel_coord(box.lo[i], ds); hi[i] = voxel_coord(box.hi[i], ds); }   // floor(v / ds) is monotone in v
    return frame_of(lo, hi);
}
// the grid cells the pending points of a batch can fall into: survivors are batch points or map points inside the
// down-sample boxes of batch points, i.e. within one box edge of the batch's bounding box
static KeyFrame cell_frame(const HostBox& box, float ds, float inv_cell) {
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
        lo[i] = cell_coord(box.lo[i] - 2.0f * ds, inv_cell) - 1;
        hi[i] = cell_coord(box.hi[i] + 2.0f * ds, inv_cell) + 1;
    }
    return frame_of(lo, hi);
}
static const uint32_t BBOX_RESET[6] = {BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT};
// the bounding-box accumulators of the insert lists (d_counters[24..29]) back to "empty": by the host when no kernel that
// consumes the lists (voxel_keys_kernel resets them itself) is going to run
static hipError_t reset_list_box(lio_ctx* c) {
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

// Second half of every (re)build: c->d_pts[0..n) is Morton-sorted and live, c->d_mkeys[0..n) holds the
// keys.  Run-length encode the cells, derive the coarser tree levels (hash tables), fill the child blocks of the
// cell directory.
int finish_index(lio_ctx* c, uint32_t n) {
    hipStream_t s = c->stream;
    c->n_points = n; c->n_live = n; c->n_cells = 0;
    c->n_retired = 0; c->n_dead = 0; c->n_new_cells = 0; c->n_cells_built = 0;
    c->grid.n_points = n;
    HIPCHK(c, hipMemsetAsync(c->d_apply + APPLY_LEVEL_NEW, 0, TREE_MAX_LEVELS * sizeof(uint32_t), s));
    HIPCHK(c, run_length_encode_u64(c->d_temp, c->temp_bytes, c->d_mkeys, n, c->d_cell_keys, c->d_cell_counts,
                                    c->d_counters + 4, s));
    uint32_t n_cells = 0;
    HIPCHK(c, read_back(c, &n_cells, c->d_counters + 4, sizeof(uint32_t)));
    HIPCHK(c, exclusive_scan_u32(c->d_temp, c->temp_bytes, c->d_cell_counts, c->d_cell_starts, n_cells, s));
    c->n_cells = n_cells; c->n_cells_built = n_cells;

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
        HIPCHK(c, coarsen_level(c->d_temp, c->temp_bytes, kin, cin, n_prev, kout, cout, c->d_counters + 4, s));
        uint32_t n_l = 0;
        HIPCHK(c, read_back(c, &n_l, c->d_counters + 4, sizeof(uint32_t)));
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
            break;
        }
        size_t sl = 64;
        while (sl < (size_t)n_l * 4) sl <<= 1;
        if (used + sl > c->ltable_slots) {
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
        HIPCHK(c, hipMemsetAsync(c->d_ltable + used, 0xFF, sl * sizeof(HashEntry), s));
        // `start` of a coarse entry is its rank among the cells of its level; level 1 uses it as the number of its
        // child block, the levels above only need to exist
        HIPCHK(c, launch_iota(c->d_rank, n_l, s));
        HIPCHK(c, launch_table_insert(kout, c->d_rank, cout, n_l, c->d_ltable + used, (uint32_t)(sl - 1), s));
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
            HIPCHK(c, launch_fill_child_blocks(c->d_cell_keys, c->d_cell_starts, c->d_cell_counts, n_cells,
                                               c->d_ltable + used, (uint32_t)(sl - 1), c->d_child, s));
            c->n_child_blocks = n_l;
            HIPCHK(c, hipMemcpyAsync(c->d_apply + 9, &c->n_child_blocks, sizeof(uint32_t), hipMemcpyHostToDevice, s));
        }
        used += sl;
        n_prev = n_l;
        // next level reads what this one wrote; alternate between two scratch pairs
        if (kout == c->d_mkeys_tmp) { kin = c->d_mkeys_tmp; cin = c->d_vals_tmp; kout = c->d_cell_keys; cout = c->d_cell_counts; }
        else { kin = c->d_cell_keys; cin = c->d_cell_counts; kout = c->d_mkeys_tmp; cout = c->d_vals_tmp; }
    }
    HIPCHK(c, stream_wait(c));
    c->levels = lv;
    c->have_map = true; c->map_broken = false;
    ++c->n_full_rebuilds;
    return LIO_OK;
}

// (re)build the index from n points at src (device, float4, must not alias c->d_pts): full sort
int build_index(lio_ctx* c, const float4* src, uint32_t n) {
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
    HIPCHK(c, launch_point_keys(src, n, c->grid, c->d_mkeys_tmp, c->d_vals_tmp, s));
    HIPCHK(c, sort_pairs_u64_u32(c->d_temp, c->temp_bytes, c->d_mkeys_tmp, c->d_mkeys, c->d_vals_tmp, c->d_vals, n, s));
    HIPCHK(c, launch_gather_points(src, c->d_vals, n, c->d_pts, s));
    return finish_index(c, n);
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
static int deferred_settle_fwd(lio_ctx* c);
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
    uint32_t live = 0;
    if (n_all > 0) {
        HIPCHK(c, compact_live_any(c, c->d_pts, n_all, c->d_pts_tmp, c->d_counters + 3));
        HIPCHK(c, read_back(c, &live, c->d_counters + 3, sizeof(uint32_t)));
    }
    if ((size_t)live + (size_t)n_pend > c->cap_map)
        return c->fail(LIO_ERR_CAPACITY, "Add_Points", "live points + batch exceed max_map_points");
    if (n_pend > 0)
        HIPCHK(c, hipMemcpyAsync(c->d_pts_tmp + live, c->d_pend, (size_t)n_pend * sizeof(float4), hipMemcpyDeviceToDevice, s));
    return build_index(c, c->d_pts_tmp, live + n_pend);
}

// how many retired slots (copies of re-written buckets) the map may carry before a full rebuild compacts them away
constexpr uint64_t RETIRED_PER_LIVE = 4;
static uint32_t retired_slot_budget(const lio_ctx* c) {
    if (c->dbg_retired_budget > 0) return (uint32_t)c->dbg_retired_budget;
    const uint64_t b = RETIRED_PER_LIVE * (uint64_t)c->n_live;
    return (uint32_t)(b > 1048576ull ? (b < 0xfffffff0ull ? b : 0xfffffff0ull) : 1048576ull);
}

// The pending points enter the index: O(batch), the rest of the map is not touched (lio_mapadd.hip).
// Launches: pend_build, the sort by (cell, list) (1 or 2), cell_plan (with the scan of the new bucket sizes), cell_apply,
// and ONE read-back that brings the statistics of this update together with the counts the fold left on the device.
int apply_pending(lio_ctx* c) {
    hipStream_t s = c->stream;
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
    HIPCHK(c, launch_cell_plan(a, s));
    HIPCHK(c, launch_cell_apply(a, s));
    uint32_t st[32] = {};
    rc = fetch_batch_counts(c, st);
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

// resolve every recorded pair (needs the stream drained) and recycle the pool
void prof_collect(lio_ctx* c) {
    bool any = false;
    for (int w = 0; w < LIO_PROF_COUNT; ++w) any = any || c->ev_next[w] > 0;
    if (!any) return;
    (void)hipStreamSynchronize(c->stream);
    for (int w = 0; w < LIO_PROF_COUNT; ++w) {
        for (size_t i = 0; i + 1 < c->ev_next[w]; i += 2) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->ev_pool[w][i], c->ev_pool[w][i + 1]) == hipSuccess) {
                c->prof_ms[w] += ms;
                c->prof_n[w] += 1;
            }
        }
        c->ev_next[w] = 0;
    }
}
// the next event pair of a family (empty timer when the family is not being timed)
KernelTimer prof_timer(lio_ctx* c, int which) {
    KernelTimer t;
    if (!(c->prof & (1u << which))) return t;
    if (c->ev_next[which] + 2 > 8192) prof_collect(c);            // bounded pool
    while (c->ev_pool[which].size() < c->ev_next[which] + 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return t;
        c->ev_pool[which].push_back(e);
    }
    t.start = c->ev_pool[which][c->ev_next[which]];
    t.stop = c->ev_pool[which][c->ev_next[which] + 1];
    c->ev_next[which] += 2;
    return t;
}

// k-NN of nq world/body queries into (nbr, cnt); queries_world must hold the world points when body == nullptr, and
// receives them otherwise (the kernel transforms).  walk_now: settle the unproven queries with the stand-alone tree
// kernel; otherwise they stay listed (count in d_counters[0], cnt marked CNT_UNPROVEN) for the measurement kernel of
// the pass, which carries the walk itself.
constexpr int FUSED_TRANSFORM_MAX_QUERIES = 262144;
// listed (unproven) queries of the last search pass above which the next one walks them in a launch of its own
constexpr int INLINE_WALK_MAX_LISTED = 1024;
TreeArgs tree_args(lio_ctx* c, int nq, int stride, float4* nbr, uint8_t* cnt, float4* ulist, unsigned long long* useeds) {
    TreeArgs ta{};
    ta.map_pts = c->d_pts; ta.child = c->d_child; ta.ltable = c->d_ltable; ta.levels = c->levels; ta.grid = c->grid;
    ta.ulist = ulist; ta.useeds = useeds; ta.n_unres = c->d_counters;
    ta.nq = nq; ta.nq_stride = stride; ta.nbr = nbr; ta.cnt = cnt;
    return ta;
}
int run_knn(lio_ctx* c, const float4* body, float4* queries_world, const Pose14& pose, int nq, int stride,
            float4* nbr, uint8_t* cnt, float4* ulist, unsigned long long* useeds, bool use_hint, bool walk_now, const GateState* gate = nullptr,
            bool pull_raw = false, const GateMail* mail = nullptr, unsigned long long expect = 0ull, int pack_front = 0, int pack_mid = 0) {
    hipStream_t s = c->stream;
    // d_counters[0] / d_cand are zero here: zeroed at creation and re-zeroed by whoever consumes them
    // (the residual kernel's last workgroup after a measurement pass, lio_map_nearest_search explicitly)
    KnnArgs ka{};
    ka.map_pts = c->d_pts; ka.grid = c->grid;
    ka.l1table = c->d_ltable + c->levels.off[1]; ka.l1mask = c->levels.mask[1]; ka.child = c->d_child;
    ka.world_in = queries_world; ka.body = body; ka.world_out = queries_world; ka.pose = pose;
    if (body != nullptr && nq > FUSED_TRANSFORM_MAX_QUERIES) {       // (see transform_kernel, lio_knn.hip)
        HIPCHK(c, launch_transform(body, nq, pose, queries_world, s));
        ka.body = nullptr;
    }
    if (pull_raw && ka.body != nullptr) {
        ka.body_raw = static_cast<const char*>(c->pull_alias); ka.body_raw_stride = c->pull_stride; ka.body_fill = c->d_body;
        ka.body_raw_aligned = (c->pull_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(c->pull_alias) & 15u) == 0u) ? 1 : 0;
    }
    ka.nq = nq; ka.nq_stride = stride;
    ka.nbr = nbr; ka.cnt = cnt; ka.ulist = ulist; ka.useeds = useeds; ka.n_unres = c->d_counters; ka.cand_total = c->census ? c->d_cand : nullptr;
    ka.use_hint = use_hint ? 1 : 0;
    ka.tune = c->dbg_knn_tune;
    ka.gate = gate;
    if (mail != nullptr && ka.body != nullptr) {                      // (no gate kernel in front: the launch reads the host's word itself)
        ka.mail = mail; ka.expect = expect; ka.gate_out = const_cast<GateState*>(gate); ka.gate = nullptr;
    }
    const int lanes = c->cfg.lanes_per_query > 0 ? c->cfg.lanes_per_query : auto_lanes(nq);
    ka.walk = (lanes == 4 || lanes == 8 || lanes == 16) ? (c->dbg_knn_walk >= 0 ? c->dbg_knn_walk : auto_walk(nq, lanes)) : 0;
    if (pack_front > 0 && ka.body_raw != nullptr) {                   // (assisted pull: the front comes packed from the host, enqueue_pass)
        ka.pack_src = c->d_pack; ka.pack_progress = c->d_pack_progress; ka.pack_stamp = c->pack_stamp; ka.pack_front = pack_front;
        ka.pack_rot = pack_front / (4 * (64 / lanes)); ka.pack_mid = pack_mid;
        ka.pack_wait_out = reinterpret_cast<unsigned long long*>(c->d_small + 48);     // (bytes 192..199 of the host-mapped words)
    }
    HIPCHK(c, launch_knn27(ka, lanes, s, prof_timer(c, use_hint ? LIO_PROF_KNN_HINT : LIO_PROF_KNN)));
    if (walk_now) HIPCHK(c, launch_knn_tree(tree_args(c, nq, stride, nbr, cnt, ulist, useeds), s, prof_timer(c, LIO_PROF_FALLBACK)));
    return LIO_OK;
}

}  // namespace

extern "C" {
// Passes armed ahead of their pose (gate_arm) sit on the stream behind a gate kernel that waits for the host: whoever else wants
// the stream -- every entry point that enqueues on it or waits for it -- calls them off first.
static void gate_call_off(lio_ctx* c);
static int gate_arm_next_scan(lio_ctx* c, bool ext);
// ... and a scan whose pull out of the caller's host records is still owed (lio_scan_set_async) is pulled before anything but
// its first search pass looks at the scan buffer.
static void flush_pull(lio_ctx* c) {
    c->pull_pending = false;
    if (c->pull_n > 0 && hipSetDevice(c->device) == hipSuccess) {
        const hipError_t e = launch_scan_pull(c->pull_alias, c->pull_stride, c->pull_n, c->d_body, c->stream);
        if (e != hipSuccess) c->err = std::string("launch_scan_pull: ") + hipGetErrorString(e);
    }
}
static int deferred_join_side(lio_ctx* c);
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
static int deferred_settle(lio_ctx* c, int* add_counter, int* n_deleted);
namespace { int deferred_settle_fwd(lio_ctx* c) { return deferred_settle(c, nullptr, nullptr); } }
static inline int enter_map(lio_ctx* c) {
    enter(c);
    if (c->def.pending) return deferred_settle(c, nullptr, nullptr);
    return LIO_OK;
}

const char* lio_version(void) { return "lio_hip 0.2 (gfx950)"; }
int lio_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int lio_create(const lio_config* cfg, lio_ctx** out) {
    if (!cfg || !out) return LIO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LIO_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return LIO_ERR_INVALID_ARG;
    if (!(cfg->filter_size_map_min > 0.0f) || cfg->max_scan_points <= 0 || cfg->max_map_points <= 0 ||
        cfg->max_map_points > 0x7fffff00ll)
        return LIO_ERR_INVALID_ARG;
    if (cfg->lanes_per_query != 0 && cfg->lanes_per_query != 4 && cfg->lanes_per_query != 8 && cfg->lanes_per_query != 16 &&
        cfg->lanes_per_query != 32 && cfg->lanes_per_query != 64)
        return LIO_ERR_INVALID_ARG;
    if (cfg->cell_size != 0.0f && (!is_pow2f(cfg->cell_size) || cfg->cell_size < cfg->filter_size_map_min)) return LIO_ERR_INVALID_ARG;

    lio_ctx* c = new lio_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
#define CREATE_CHK(expr)                                                         \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) {                                                  \
            fprintf(stderr, "lio_create: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            lio_destroy(c);                                                      \
            return LIO_ERR_HIP;                                                  \
        }                                                                        \
    } while (0)
    CREATE_CHK(hipSetDevice(c->device));
    { const char* e = std::getenv("LIO_TRACE_REBUILDS"); c->trace_rebuilds = e != nullptr && e[0] == '1'; }
    { const char* e = std::getenv("LIO_TRACE_HOST"); c->trace_host = e != nullptr && e[0] == '1'; }
    { const char* e = std::getenv("LIO_PRE_ENQUEUE"); if (e != nullptr && e[0] == '0') c->dbg_pre_enqueue = 0; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_KNN_TUNE"); if (e != nullptr && e[0] >= '0' && e[0] <= '9') c->dbg_knn_tune = std::atoi(e); }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_ARM_NEXT_SCAN"); if (e != nullptr && e[0] == '0') c->dbg_arm_next_scan = 0; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_GATE_IN_KERNEL"); if (e != nullptr && e[0] >= '0' && e[0] <= '7' && e[1] == 0) c->gate_in_kernel = e[0] - '0'; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_ASSIST_PULL"); if (e != nullptr && e[0] >= '0' && e[0] <= '9') c->dbg_assist_pull = std::atoi(e); }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_KNN_WALK"); if (e != nullptr && (e[0] == '0' || e[0] == '1' || e[0] == '2' || e[0] == '4') && e[1] == 0) c->dbg_knn_walk = e[0] - '0'; }
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, c->device));
    c->arch = prop.gcnArchName;
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    CREATE_CHK(hipHostMalloc(&c->h_small, 256, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_small, 0, 256);
    CREATE_CHK(hipHostGetDevicePointer((void**)&c->d_small, c->h_small, 0));
    // The mail slots of the gates.  Where the whole of device memory is mapped into the CPU's address space (large BAR: every
    // MI300-class box) they live in FINE-GRAINED DEVICE memory: the host's 17 words travel once, as posted writes through the
    // BAR, and the gate polls local memory instead of paying a PCIe read round trip per poll (scripts/micro/mail_in_vram.hip:
    // host word -> first workgroup of the pass -> host 3.6 -> 3.2 us).  Otherwise: host-mapped, coherent memory, as before.
    {
        int large_bar = 0;
        const char* e = std::getenv("LIO_GATE_MAIL_VRAM");
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) != hipSuccess) { (void)hipGetLastError(); large_bar = 0; }
        if (e != nullptr && e[0] == '0') large_bar = 0;
        void* vram = nullptr;
        if (large_bar && hipExtMallocWithFlags(&vram, (lio_ctx::GATE_SLOTS + 1) * sizeof(GateMail), hipDeviceMallocFinegrained) == hipSuccess && vram != nullptr) {
            c->h_gate_mail = static_cast<GateMail*>(vram); c->d_gate_mail = c->h_gate_mail; c->gate_mail_in_vram = true;
            CREATE_CHK(hipMemset(vram, 0, (lio_ctx::GATE_SLOTS + 1) * sizeof(GateMail)));
        } else {
            (void)hipGetLastError();
            CREATE_CHK(hipHostMalloc((void**)&c->h_gate_mail, (lio_ctx::GATE_SLOTS + 1) * sizeof(GateMail), hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(c->h_gate_mail, 0, (lio_ctx::GATE_SLOTS + 1) * sizeof(GateMail));
            CREATE_CHK(hipHostGetDevicePointer((void**)&c->d_gate_mail, c->h_gate_mail, 0));
        }
    }
    // (one slot more than the gates use: its first word is the progress word of the assisted pull)
    c->h_pack_progress = &c->h_gate_mail[lio_ctx::GATE_SLOTS].w[0];
    c->d_pack_progress = &c->d_gate_mail[lio_ctx::GATE_SLOTS].w[0];
    CREATE_CHK(hipMalloc((void**)&c->d_gate, lio_ctx::GATE_SLOTS * sizeof(GateState)));
    CREATE_CHK(hipMemset(c->d_gate, 0, lio_ctx::GATE_SLOTS * sizeof(GateState)));

    c->grid.cell = cfg->cell_size != 0.0f ? cfg->cell_size : auto_cell_size(cfg->filter_size_map_min);
    c->grid.inv_cell = 1.0f / c->grid.cell;
    c->grid.n_points = 0;

    const size_t nm = (size_t)cfg->max_map_points;
    c->cap_map = nm;
    // buckets that receive points are re-written at the tail of the array; the room for that is ours, not the
    // caller's (max_map_points bounds the live points)
    c->slots_cap = (1 + RETIRED_PER_LIVE) * nm + 1048576;   // live points + the budget of retired slots (apply_pending): 80 B per map point
    if (c->slots_cap > 0xfffffff0ull) c->slots_cap = 0xfffffff0ull;
    CREATE_CHK(hipMalloc((void**)&c->d_pts, c->slots_cap * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_pts_tmp, c->slots_cap * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_apply, 32 * sizeof(uint32_t)));
    CREATE_CHK(hipMemset(c->d_apply, 0, 32 * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_mkeys, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_mkeys_tmp, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_vals, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_vals_tmp, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_keys, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_counts, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_starts, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_rank, nm * sizeof(uint32_t)));
    size_t tb = sort_pairs_temp_bytes(nm);
    size_t t2 = rle_temp_bytes(nm); if (t2 > tb) tb = t2;
    t2 = scan_temp_bytes(nm); if (t2 > tb) tb = t2;
    t2 = select_temp_bytes(c->slots_cap); if (t2 > tb) tb = t2;
    c->temp_bytes = tb + 256;
    CREATE_CHK(hipMalloc(&c->d_temp, c->temp_bytes));
    // an empty map still has a (one-slot, empty) level-1 table and one zero child block: look-ups stay defined
    c->ltable_slots = 4096;
    CREATE_CHK(hipMalloc((void**)&c->d_ltable, c->ltable_slots * sizeof(HashEntry)));
    CREATE_CHK(hipMemset(c->d_ltable, 0xFF, c->ltable_slots * sizeof(HashEntry)));
    c->child_blocks_cap = 4096;
    CREATE_CHK(hipMalloc((void**)&c->d_child, c->child_blocks_cap * 64 * sizeof(uint2)));
    CREATE_CHK(hipMemset(c->d_child, 0, c->child_blocks_cap * 64 * sizeof(uint2)));
    c->cagg_words = COMPACT_AGG_WORDS;
    CREATE_CHK(hipMalloc((void**)&c->d_cagg, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
    CREATE_CHK(hipMemset(c->d_cagg, 0, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void**)&c->d_ghist, 512 * sizeof(uint32_t)));
    CREATE_CHK(hipMemset(c->d_ghist, 0, 512 * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_gagg, GSORT_AGG_WORDS * sizeof(unsigned long long)));
    CREATE_CHK(hipMemset(c->d_gagg, 0, GSORT_AGG_WORDS * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void**)&c->d_counters, 32 * sizeof(uint32_t)));
    CREATE_CHK(hipMemset(c->d_counters, 0, 32 * sizeof(uint32_t)));
    CREATE_CHK(hipMemcpy(c->d_counters + 24, BBOX_RESET, sizeof(BBOX_RESET), hipMemcpyHostToDevice));   // [24..29] bounding box of the insert lists: empty
    CREATE_CHK(hipMalloc((void**)&c->d_cand, sizeof(unsigned long long)));
    CREATE_CHK(hipMemset(c->d_cand, 0, sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void**)&c->d_boxes, 64 * 6 * sizeof(float)));

    const size_t nq = (size_t)cfg->max_scan_points;
    c->cap_q = cfg->max_scan_points;
    CREATE_CHK(hipMalloc((void**)&c->d_body, nq * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_body_back, nq * sizeof(float4)));
    CREATE_CHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    CREATE_CHK(hipMalloc((void**)&c->d_world, nq * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_nbr, nq * K * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_normvec, nq * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_plane, nq * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_cnt, nq));
    CREATE_CHK(hipMalloc((void**)&c->d_sel, nq));
    CREATE_CHK(hipMalloc((void**)&c->d_ulist, nq * sizeof(float4)));
    CREATE_CHK(hipMalloc((void**)&c->d_useeds, nq * 8 * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void**)&c->d_lunits, (size_t)FIX_MAX * RED_STRIDE * 2 * sizeof(unsigned long long)));
    CREATE_CHK(hipMemset(c->d_lunits, 0, (size_t)FIX_MAX * RED_STRIDE * 2 * sizeof(unsigned long long)));
    {
        const size_t blocks = (size_t)residual_blocks((int)nq), groups = (blocks + RED_GROUP - 1) / RED_GROUP;
        CREATE_CHK(hipMalloc((void**)&c->d_units, blocks * RED_STRIDE * 2 * sizeof(unsigned long long)));
        CREATE_CHK(hipMemset(c->d_units, 0, blocks * RED_STRIDE * 2 * sizeof(unsigned long long)));     // stamp 0: no pass has it
        CREATE_CHK(hipMalloc((void**)&c->d_gunits, groups * RED_STRIDE * 2 * sizeof(unsigned long long)));
        CREATE_CHK(hipMemset(c->d_gunits, 0, groups * RED_STRIDE * 2 * sizeof(unsigned long long)));
    }
    CREATE_CHK(hipHostMalloc((void**)&c->h_out, sizeof(DeviceOut), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_out, 0, sizeof(DeviceOut));
    CREATE_CHK(hipHostGetDevicePointer((void**)&c->d_out, c->h_out, 0));
#undef CREATE_CHK
    c->have_map = true;                         // an empty map is a valid map (Build not yet called)
    *out = c;
    return LIO_OK;
}

// LIO_TRACE_HOST=1: the last `last` stamps of the host side (tag, microseconds since the previous one) to stderr; the trace starts over
static void dump_host_trace(lio_ctx* c, size_t last) {
    const size_t n = c->host_trace.size(), from = n > last ? n - last : 0;
    for (size_t i = from; i < n && n > 1; ++i)
        std::fprintf(stderr, "lio host trace %c %+9.2f us%s\n", c->host_trace[i].tag,
                     i > from ? (double)(c->host_trace[i].ns - c->host_trace[i - 1].ns) * 1e-3 : 0.0,
                     c->host_trace[i].tag == 'S' ? (c->host_trace[i].polls == 0 ? "  (no poll: the record was there)" : "  (polled)") : "");
    c->host_trace.clear();
}
static void pack_helper_stop(lio_ctx* c) {
    PackHelper* h = c->pack_helper;
    if (h == nullptr) return;
    { std::lock_guard<std::mutex> lk(h->mu); h->quit = true; }
    h->cv.notify_one();
    if (h->th.joinable()) h->th.join();
    delete h;
    c->pack_helper = nullptr;
}
int lio_scan_pack_threads(lio_ctx* c, int n) {
    if (!c || n < 1) return LIO_ERR_INVALID_ARG;
    if (n == 1) { pack_helper_stop(c); c->assist_threads = 1; return LIO_OK; }
    if (c->pack_helper == nullptr) {
        PackHelper* h = new PackHelper();
        try { h->th = std::thread([h] { h->run(); }); }
        catch (const std::exception& e) { delete h; return c->fail(LIO_ERR_HIP, "lio_scan_pack_threads", e.what()); }
        c->pack_helper = h;
    }
    c->assist_threads = 2;                                 // (more than one helper: the calling thread's half would have to shrink below what the launches take)
    return LIO_OK;
}
void lio_destroy(lio_ctx* c) {
    if (c && c->trace_host) dump_host_trace(c, 80);
    if (!c) return;
    (void)hipSetDevice(c->device);
    lio_replicate_finalize(c);
    if (c->gate_n > 0) gate_call_off(c);                       // (a pass armed across a scan boundary may still wait for its pose)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* dev[] = {c->d_pts, c->d_pts_tmp, c->d_mkeys, c->d_mkeys_tmp, c->d_vals, c->d_vals_tmp, c->d_cell_keys,
                   c->d_cell_counts, c->d_cell_starts, c->d_rank, c->d_child, c->d_ltable, c->d_pend, c->d_pkeys, c->d_pkeys2, c->d_gs_part, c->d_gs_part2, c->d_porder, c->d_porder2, c->d_pnewoff, c->d_plan, c->d_ghist, c->d_gagg, c->d_apply, c->d_temp, c->d_counters, c->d_cagg, c->d_cand, c->d_boxes,
                   c->d_body, c->d_body_back, c->d_world, c->d_nbr, c->d_normvec, c->d_plane, c->d_cnt, c->d_sel, c->d_ulist, c->d_useeds, c->d_lunits, c->d_rows, c->d_rows_c,
                   c->d_units, c->d_gunits, c->d_removed, c->d_scratch, c->d_qs_world, c->d_qs_nbr, c->d_qs_cnt, c->d_qs_ulist, c->d_qs_useeds,
                   c->d_raw, c->d_undist, c->d_leafpts, c->d_raw_keys, c->d_raw_keys2, c->d_leaf_keys, c->d_raw_order,
                   c->d_raw_order2, c->d_leaf_count, c->d_leaf_start, c->d_imu_poses, c->d_vox,
                   c->d_add_pts, c->d_add_pts2, c->d_add_cls, c->d_add_out, c->d_add_keys, c->d_add_keys2,
                   c->d_add_order, c->d_add_order2, c->d_add_dirty, c->d_add_dirty_pt, c->d_add_dlist, c->d_add_plan, c->d_add_ranges, c->d_add_parts,
                   c->d_rep_send, c->d_rep_recv};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_rows) (void)hipHostFree(c->h_rows);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    for (int w = 0; w < LIO_PROF_COUNT; ++w)
        for (hipEvent_t e : c->ev_pool[w]) (void)hipEventDestroy(e);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->h_small2) (void)hipHostFree(c->h_small2);
    for (int i = 0; i < 2; ++i) {
        if (c->d_def_add[i]) (void)hipFree(c->d_def_add[i]);
        if (c->d_def_plain[i]) (void)hipFree(c->d_def_plain[i]);
        if (c->d_def_boxes[i]) (void)hipFree(c->d_def_boxes[i]);
    }
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_small) (void)hipHostFree(c->h_small);
    pack_helper_stop(c);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    if (c->h_gate_mail) { if (c->gate_mail_in_vram) (void)hipFree(c->h_gate_mail); else (void)hipHostFree(c->h_gate_mail); }
    if (c->d_gate) (void)hipFree(c->d_gate);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* lio_last_error(lio_ctx* c) { return c ? c->err.c_str() : "null context"; }
void* lio_stream(lio_ctx* c) { return c ? (void*)c->stream : nullptr; }
int lio_synchronize(lio_ctx* c) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_wait(c));
    if (c->def.planned) HIPCHK(c, hipStreamSynchronize(c->side_stream));       // (the plan of a deferred batch, beside the scan step)
    return LIO_OK;
}

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
    HIPCHK(c, launch_fold_plan(fa, ba, s));
    HIPCHK(c, launch_compact_dirty(ba, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, s));
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
        HIPCHK(c, launch_map_incremental_lists(ia, c->d_add_pts, c->d_add_pts2, c->d_counters + 5, c->d_counters + 24, c->d_cagg,
                                               next_compact_stamp(c), c->d_counters + 12, s));
    } else {
        HIPCHK(c, launch_map_incremental(ia, s));
        HIPCHK(c, select_points_by_class(c->d_temp, c->temp_bytes, c->d_world, c->d_add_cls, 1, c->d_add_pts, c->d_counters + 5, (size_t)nq, s));
        HIPCHK(c, select_points_by_class(c->d_temp, c->temp_bytes, c->d_world, c->d_add_cls, 2, c->d_add_pts2, c->d_counters + 6, (size_t)nq, s));
        HIPCHK(c, launch_bbox(c->d_world, (uint32_t)nq, c->d_counters + 24, s));    // (of every scan point: a superset, which is all a frame needs)
    }
    if (words) return fetch_list_words(c, *words);
    return LIO_OK;
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
    int rc = map_incremental_lists(c, state14, filter_size_map_min, flg_EKF_inited, &w);
    if (rc != LIO_OK) return rc;
    hipStream_t s = c->stream;
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
        HIPCHK(c, hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
    }
    if (!c->ev_side) HIPCHK(c, hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    if (!c->h_small2) {
        HIPCHK(c, hipHostMalloc(&c->h_small2, 256, hipHostMallocMapped));
        std::memset(c->h_small2, 0, 256);
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_small2, c->h_small2, 0));
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
static int deferred_launch_plan(lio_ctx* c) {
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
static int deferred_join_side(lio_ctx* c) {
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
static int deferred_settle(lio_ctx* c, int* add_counter, int* n_deleted) {
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
    if (defer && c->def.pending) {
        // The totals go into the second mail box WITHOUT a wait; the previous batch enters the map behind them on the stream, and
        // its own read-back (the statistics of the index update) is the one host wait of this exchange: by then the totals are there.
        const uint32_t seq2 = ++c->small2_seq;
        HIPCHK(c, launch_publish_words(c->d_counters + 8, 22, c->d_small2, seq2, s));
        const int rc = deferred_settle(c, &c->def_last_counter, &c->def_last_deleted);
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

// ---- scan + measurement ----------------------------------------------------------------------
static int scan_reset(lio_ctx* c, int n) {
    // the passes of the scan that ends here (the converge flag of every lio_measure_begin) predict the next scan's: what
    // lio_measure_begin arms ahead of its pose (gate_arm)
    if (!c->pattern_cur.empty()) { c->pattern_prev.swap(c->pattern_cur); c->pattern_cur.clear(); }
    c->nq = n;
    c->have_scan = true; c->have_search = false; c->have_pass = false; c->last_m = 0; c->hint_valid = false;
    c->across_off_this_scan = false;
    // point_selected_surf starts all-true (laserMapping.cpp:2131), but nothing reads it before the first pass of
    // the scan has written it: that pass must search (lio_measure enforces it) and a search pass decides from the
    // neighbours alone (:1911) -- so there is no fill launch here
    return LIO_OK;
}
int lio_scan_set(lio_ctx* c, const float* body_pts, int n, int stride_bytes) {
    if (!c || n < 0 || (n > 0 && !body_pts) || stride_bytes < 12) return LIO_ERR_INVALID_ARG;
    c->pull_pending = false;
    enter(c);
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_set", "n exceeds max_scan_points");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_points(c, body_pts, (size_t)n, stride_bytes, c->d_body);
    if (rc != LIO_OK) return rc;
    return scan_reset(c, n);
}
int lio_scan_set_device(lio_ctx* c, const void* d_body_f4, int n) {
    if (!c || n < 0 || (n > 0 && !d_body_f4)) return LIO_ERR_INVALID_ARG;
    c->pull_pending = false;
    enter(c);
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_set_device", "n exceeds max_scan_points");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0) HIPCHK(c, hipMemcpyAsync(c->d_body, d_body_f4, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    return scan_reset(c, n);
}
// page-locked host memory known to the HIP runtime (lio_host_alloc, lio_host_register, hipHostMalloc, hipHostRegister)?
static bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
// feats_down_body handed over WITHOUT waiting for it to arrive (the drop-in callback path: lio::scan_changed right behind the
// node's voxel filter, laserMapping.cpp:2322-2325).  Page-locked records the runtime knows: a kernel pulls x, y, z over PCIe
// itself (launch_scan_pull; no host work at all, the call returns after one launch) -- the records must then stay untouched until
// the first measurement pass of the scan has returned.  Any other memory: packed into the staging buffer before the call
// returns, the copy is not waited for.  Either way the passes that follow are ordered behind the upload on the stream.
int lio_scan_set_async(lio_ctx* c, const float* body_pts, int n, int stride_bytes) {
    if (!c || n < 0 || (n > 0 && !body_pts) || stride_bytes < 12 || (stride_bytes & 3) != 0) return LIO_ERR_INVALID_ARG;
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_set_async", "n exceeds max_scan_points");
    c->pull_pending = false;                    // (a scan nobody looked at is simply replaced)
    enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0 && is_pinned_host(body_pts) && is_pinned_host(reinterpret_cast<const char*>(body_pts) + (size_t)(n - 1) * (size_t)stride_bytes + 11)) {
        void* alias = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(&alias, const_cast<float*>(body_pts), 0));
        ++c->n_scan_pulled;
        if (c->dbg_fused_pull && n <= FUSED_TRANSFORM_MAX_QUERIES) {
            // not yet: the first search pass of the scan pulls its queries itself (KnnArgs.body_raw), the search runs beside the
            // transfer; whatever else wants the scan buffer first makes up for it (enter())
            c->pull_pending = true; c->pull_alias = alias; c->pull_stride = stride_bytes; c->pull_n = n; c->pull_host = reinterpret_cast<const char*>(body_pts);
            if (c->pack_helper != nullptr && stride_bytes >= 32 && (double)n * stride_bytes / PCIE_BYTES_PER_NS >= 15000.0) {
                // the second packing thread: awake and looking for its job by the time the scan's first pass posts it
                { std::lock_guard<std::mutex> lk(c->pack_helper->mu); ++c->pack_helper->wake_seq; }
                c->pack_helper->cv.notify_one();
            }
        } else {
            HIPCHK(c, launch_scan_pull(alias, stride_bytes, n, c->d_body, c->stream));
        }
        return scan_reset(c, n);
    }
    const int rc = upload_points(c, body_pts, (size_t)n, stride_bytes, c->d_body);
    if (rc != LIO_OK) return rc;
    return scan_reset(c, n);
}
int lio_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return LIO_ERR_INVALID_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return LIO_ERR_HIP; }
    return LIO_OK;
}
int lio_host_unregister(void* p) {
    if (!p) return LIO_ERR_INVALID_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return LIO_ERR_HIP; }
    return LIO_OK;
}
int lio_scan_size(lio_ctx* c) { return c ? c->nq : 0; }

// ---- pipelined hand-over: the next scan travels while the current one is being processed --------------------
static int issue_pending_upload(lio_ctx* c) {
    c->upload_pending = false;
    // the back buffer is not read by anything in flight: the scan step that used it last has returned to the caller
    if (c->upload_n > 0)
        HIPCHK(c, hipMemcpyAsync(c->d_body_back, c->upload_src, (size_t)c->upload_n * sizeof(float4), hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy_stream));
    c->nq_back = c->upload_n;
    return LIO_OK;
}
int lio_scan_upload_async(lio_ctx* c, const void* host_f4, int n) {
    if (!c || n < 0 || (n > 0 && !host_f4)) return LIO_ERR_INVALID_ARG;
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_upload_async", "n exceeds max_scan_points");
    for (int i = 0; i < c->gate_n; ++i) if (c->gate_q[i].next_scan) { gate_call_off(c); break; }   // (armed on the buffer this upload replaces)
    // The copy is issued by the next measurement pass right behind its kernel launches (the runtime calls then run
    // beside the GPU's work), or by lio_scan_swap if no pass comes first.
    c->upload_pending = true; c->upload_src = host_f4; c->upload_n = n;
    return LIO_OK;
}
int lio_scan_swap(lio_ctx* c) {
    if (!c) return LIO_ERR_INVALID_ARG;
    // the first pass of the scan that is swapped in may already sit on the stream (gate_arm_next_scan): it stays
    const bool armed_for_it = c->gate_n == 1 && c->gate_q[0].next_scan && !c->upload_pending && !c->pull_pending &&
                              c->gate_q[0].body == c->d_body_back && c->gate_q[0].nq == c->nq_back;
    if (!armed_for_it) enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->upload_pending) {
        const int rc = issue_pending_upload(c);
        if (rc != LIO_OK) return rc;
    }
    if (c->nq_back < 0) return c->fail(LIO_ERR_NO_SCAN, "lio_scan_swap", "no scan was uploaded with lio_scan_upload_async");
    // usually the copy finished long ago (it ran beside the previous scan step): then the host has seen it complete
    // and everything launched from here on is ordered behind it; otherwise the compute stream waits for the event
    if (!armed_for_it) {                                       // (an armed pass already sits behind the copy)
        const hipError_t q = hipEventQuery(c->ev_copy);
        if (q == hipErrorNotReady) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
        else if (q != hipSuccess) return c->fail(LIO_ERR_HIP, "lio_scan_swap", hipGetErrorString(q));
    }
    float4* t = c->d_body; c->d_body = c->d_body_back; c->d_body_back = t;
    const int n = c->nq_back;
    c->nq_back = -1;
    if (armed_for_it) c->gate_q[0].next_scan = false;         // from here on: the first pass of the CURRENT scan
    return scan_reset(c, n);
}
void* lio_host_alloc(size_t bytes) {
    void* p = nullptr;
    // portable: pinned for every device of the process (one rank per GPU allocates before or after it picks its device)
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
void lio_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---- raw-scan preparation: UndistortPcl's compensation + pcl::VoxelGrid ----
static int ensure_raw_buffers(lio_ctx* c, size_t n) {
    if (!c->d_imu_poses) HIPCHK(c, hipMalloc((void**)&c->d_imu_poses, 512 * 22 * sizeof(double)));
    if (!c->d_vox) HIPCHK(c, hipMalloc((void**)&c->d_vox, 2 * sizeof(VoxelGridDev)));
    if (n <= c->cap_raw) return LIO_OK;
    void* old[] = {c->d_raw, c->d_undist, c->d_leafpts, c->d_raw_keys, c->d_raw_keys2, c->d_leaf_keys, c->d_raw_order,
                   c->d_raw_order2, c->d_leaf_count, c->d_leaf_start};
    for (void* p : old) if (p) (void)hipFree(p);
    c->d_raw = c->d_undist = c->d_leafpts = nullptr; c->d_raw_keys = c->d_raw_keys2 = c->d_leaf_keys = nullptr;
    c->d_raw_order = c->d_raw_order2 = c->d_leaf_count = c->d_leaf_start = nullptr; c->cap_raw = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_raw, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_undist, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_leafpts, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_keys, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_keys2, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_keys, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_order, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_order2, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_count, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_start, n * sizeof(uint32_t)));
    c->cap_raw = n;
    return LIO_OK;
}

// ImuProcess::UndistortPcl, point side (src/IMU_Processing.hpp:250-251, 334-386): sort the raw scan by its
// per-point time offset, then compensate every point to the scan-end LiDAR frame from the forward-propagated
// IMU poses (IMUpose, :258/:322) and the propagated end state.  n_poses == 0: upload only (no sort, no motion).
// The two halves of it, so that a caller can put host work (the IMU forward propagation) between them while the
// copy is in flight.  First half: the raw records -> pinned staging -> device (asynchronous).
// time_offset_bytes < 0: the scan carries no per-point time (w = 0).
int lio_scan_upload_raw(lio_ctx* c, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes) {
    if (!c || n < 0 || (n > 0 && !raw_pts) || stride_bytes < 12 ||
        (time_offset_bytes >= 0 && (time_offset_bytes < 12 || time_offset_bytes + 4 > stride_bytes)))
        return LIO_ERR_INVALID_ARG;
    enter(c);
    if ((size_t)n > c->temp_bytes / 16) return c->fail(LIO_ERR_CAPACITY, "lio_scan_undistort", "raw scan larger than the sort workspace");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = ensure_raw_buffers(c, (size_t)(n > 0 ? n : 1));
    if (rc != LIO_OK) return rc;
    c->n_raw = n; c->have_raw = false; c->raw_uploaded = true;
    c->raw_time_keyed = time_offset_bytes >= 0;
    if (n == 0) return LIO_OK;
    rc = ensure_stage(c, (size_t)n + POSE_STAGE_SLOTS);                        // the IMU poses travel behind the points
    if (rc != LIO_OK) return rc;
    const char* src = reinterpret_cast<const char*>(raw_pts);
    if (time_offset_bytes == 12 && stride_bytes == 16 && is_pinned_host(raw_pts)) {
        // x, y, z, t records in page-locked memory (lio_host_alloc): the copy engine reads them where they are -- no
        // staging copy (60 us of host time for a 120 k-point sweep).  The caller leaves them alone until a call that
        // waits for the stream has returned (lio_scan_voxel_filter does).
        HIPCHK(c, hipMemcpyAsync(c->d_raw, raw_pts, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s));
        return LIO_OK;
    }
    if (time_offset_bytes == 12 && stride_bytes == 16) {
        std::memcpy(c->h_stage, raw_pts, (size_t)n * sizeof(float4));         // the records already are x, y, z, t
    } else {
        for (int i = 0; i < n; ++i) {
            const char* rec = src + (size_t)i * (size_t)stride_bytes;
            const float* p = reinterpret_cast<const float*>(rec);
            const float t = time_offset_bytes >= 0 ? *reinterpret_cast<const float*>(rec + time_offset_bytes) : 0.0f;
            c->h_stage[i] = make_float4(p[0], p[1], p[2], t);
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->d_raw, c->h_stage, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->ev_stage, s));
    c->stage_busy = true; c->stage_busy_slots = (size_t)n;                     // waited for by the next writer of h_stage
    return LIO_OK;
}

// Second half: compensation of the uploaded scan from the forward-propagated poses.  n_poses == 0: the cloud is
// taken as it is.
int lio_scan_compensate(lio_ctx* c, const lio_pose6d* imu_poses, int n_poses, const double end_state14[14]) {
    if (!c || n_poses < 0 || n_poses > 512 || (n_poses > 0 && (!imu_poses || !end_state14))) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (!c->raw_uploaded) return c->fail(LIO_ERR_NO_SCAN, "lio_scan_compensate", "no raw scan uploaded");
    if (n_poses > 0 && !c->raw_time_keyed) return LIO_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = c->n_raw;
    c->have_raw = true;
    if (n == 0) return LIO_OK;
    if (n_poses == 0) {
        HIPCHK(c, hipMemcpyAsync(c->d_undist, c->d_raw, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, s));
        c->undist_sorted = true;                                               // no sort, no compensation
        c->raw_time_keyed = false;
        return LIO_OK;
    }
    // the poses go through the tail of the pinned staging buffer (bytes the point copy does not read)
    static_assert(sizeof(lio_pose6d) == 22 * sizeof(double), "Pose6D layout");
    if (c->stage_busy && c->stage_busy_slots > c->h_stage_cap - POSE_STAGE_SLOTS) {   // (a copy in flight reads the tail too: not the raw upload's)
        HIPCHK(c, hipEventSynchronize(c->ev_stage));
        c->stage_busy = false;
    }
    lio_pose6d* h_poses = reinterpret_cast<lio_pose6d*>(c->h_stage + (c->h_stage_cap - POSE_STAGE_SLOTS));
    std::memcpy(h_poses, imu_poses, (size_t)n_poses * sizeof(lio_pose6d));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_poses, h_poses, (size_t)n_poses * sizeof(lio_pose6d), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->ev_stage, s));
    c->stage_busy = true; c->stage_busy_slots = c->h_stage_cap;
    // The compensation of a point depends on nothing but the point: it runs in input order.  The order
    // sort(pcl_out.points.begin(), pcl_out.points.end(), time_list) (:251) establishes is produced where it is
    // consumed -- as the minor key of the voxel filter's sort, or on demand by lio_scan_get_cloud (stable: ties keep
    // the input order).
    Pose14 end;
    for (int i = 0; i < 14; ++i) end.v[i] = end_state14[i];
    HIPCHK(c, launch_undistort(c->d_raw, n, c->d_imu_poses, n_poses, end, c->d_undist, s));
    c->undist_sorted = false;
    return LIO_OK;
}

int lio_scan_undistort(lio_ctx* c, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes,
                       const lio_pose6d* imu_poses, int n_poses, const double end_state14[14]) {
    if (!c || n_poses < 0 || n_poses > 512 ||
        (n_poses > 0 && (!imu_poses || !end_state14 || time_offset_bytes < 12 || time_offset_bytes + 4 > stride_bytes)))
        return LIO_ERR_INVALID_ARG;
    const int rc = lio_scan_upload_raw(c, raw_pts, n, stride_bytes, n_poses > 0 ? time_offset_bytes : -1);
    if (rc != LIO_OK) return rc;
    return lio_scan_compensate(c, imu_poses, n_poses, end_state14);
}

// feats_undistort in time order (on demand: the voxel filter does not need it)
static int sort_undistorted_by_time(lio_ctx* c) {
    if (c->undist_sorted || c->n_raw == 0) { c->undist_sorted = true; return LIO_OK; }
    hipStream_t s = c->stream;
    const int n = c->n_raw;
    HIPCHK(c, launch_time_keys(c->d_undist, n, c->d_raw_keys, c->d_raw_order, s));
    HIPCHK(c, sort_pairs_any(c, c->d_raw_keys, c->d_raw_keys2, c->d_raw_order, c->d_raw_order2, (size_t)n, s));
    HIPCHK(c, launch_gather_points(c->d_undist, c->d_raw_order2, (uint32_t)n, c->d_leafpts, s));
    float4* t = c->d_undist; c->d_undist = c->d_leafpts; c->d_leafpts = t;
    c->undist_sorted = true;
    return LIO_OK;
}

// downSizeFilterSurf.setInputCloud(feats_undistort); downSizeFilterSurf.filter(*feats_down_body)
// (src/laserMapping.cpp:2322-2323): pcl::VoxelGrid on the cloud left by lio_scan_undistort; the centroids
// become the current scan (as lio_scan_set would).
int lio_scan_voxel_filter(lio_ctx* c, float leaf_size, int* n_out) {
    if (!c || !(leaf_size > 0.0f)) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (n_out) *n_out = 0;
    if (!c->have_raw) return c->fail(LIO_ERR_NO_SCAN, "lio_scan_voxel_filter", "no raw scan (lio_scan_undistort first)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = c->n_raw;
    if (n == 0) return scan_reset(c, 0);
    // One pass over the stream, one read-back at the end: getMinMax3D, min_b_/divb_mul_, the leaf index of every
    // point, a stable sort by leaf, the run heads and the centroids all stay on the device (VoxelGridDev).
    const bool time_minor = c->raw_time_keyed && !c->undist_sorted;     // the cloud is not KNOWN to be in time order
    // A sweep between the single-launch sort's range and the group sort's (16 k ... 256 k points: every real one) whose points
    // already come in time order -- most drivers deliver them so -- needs no time key: a STABLE sort by the leaf index alone is
    // the sort by (leaf, time, input order).  That key has ~22 bits instead of 64: three stable 8-bit splits (lio_gsort.hip, LSD
    // form) instead of the library's ten merge launches.  Whether THIS sweep is in order, and how wide its leaf index is, only
    // the device knows (minmax_kernel checks the order while it reads the points): the host PREDICTS both from the previous sweep,
    // enqueues the whole filter without waiting, and finds the verdict in the read-back at the end; a wrong prediction (the first
    // out-of-order sweep of a driver, a sweep whose extent needs another digit) is followed by the filter on the general path.
    const bool may_compact = !c->dbg_library_sort && c->dbg_voxel_fast && (size_t)n > SORT_MAX_ITEMS && (uint32_t)n <= GSORT_MAX;
    bool compact = may_compact && !(time_minor && c->vox_saw_unsorted);
    const uint32_t max_leaves = (uint32_t)n < (uint32_t)c->cap_q + 1u ? (uint32_t)n : (uint32_t)c->cap_q + 1u;
    uint32_t summary[6] = {0, 0, 0, 0, 0, 0};
    for (int attempt = 0; ; ++attempt) {
        VoxelGridDev* vox = c->d_vox + c->vox_cur;
        VoxelGridDev* vox_next = c->d_vox + (c->vox_cur ^ 1);
        if (!c->vox_clean[c->vox_cur]) HIPCHK(c, hipMemsetAsync(vox, 0, sizeof(VoxelGridDev), s));
        c->vox_clean[c->vox_cur] = false;
        c->vox_clean[c->vox_cur ^ 1] = false;                           // (until the centroid launch below is known to be enqueued)
        // inverse_leaf_size_ = 1 / leaf_size_; the order of the points is checked whenever it matters for the next prediction
        HIPCHK(c, launch_minmax_params(c->d_undist, n, vox, 1.0f / leaf_size, may_compact && time_minor, s));
        int passes = 0;
        if (compact) {
            passes = c->vox_passes_guess;
            ++c->n_voxel_fast;
            if (c->gsort_seq + (uint32_t)passes < c->gsort_seq || c->gsort_seq + (uint32_t)passes == 0xffffffffu) {   // (stamp wrap-around: as run_sort)
                HIPCHK(c, hipMemsetAsync(c->d_gagg, 0, GSORT_AGG_WORDS * sizeof(unsigned long long), s));
                c->gsort_seq = 0u;
            }
            HIPCHK(c, launch_voxel_index_compact(c->d_undist, n, vox, c->d_raw_keys, s));
            GsortLsdArgs g{};
            g.keys = c->d_raw_keys; g.n = (uint32_t)n; g.kb = 8 * passes; g.passes = passes; g.keys_out = c->d_raw_keys2; g.vals_out = c->d_raw_order2;
            g.part = c->d_leaf_keys; g.part2 = reinterpret_cast<uint64_t*>(c->d_leafpts);  // (scratch of this call: 8 and 16 bytes per point)
            g.hist = &vox->hist[0][0]; g.agg = c->d_gagg; g.stamp0 = c->gsort_seq + 1u; g.err = &vox->summary[3];
            c->gsort_seq += (uint32_t)passes;
            HIPCHK(c, launch_gsort_lsd(g, s));
            HIPCHK(c, launch_run_heads(c->d_raw_keys2, (uint32_t)n, 0, c->d_leaf_start, &vox->n_leaves, c->d_cagg, next_compact_stamp(c), &vox->summary[3], s));
        } else {
            HIPCHK(c, launch_voxel_index(c->d_undist, n, vox, time_minor, c->d_raw_keys, c->d_raw_order, s));
            HIPCHK(c, sort_pairs_any(c, c->d_raw_keys, c->d_raw_keys2, c->d_raw_order, c->d_raw_order2, (size_t)n, s, &vox->summary[3]));
            if (compact_small(c, (size_t)n))
                HIPCHK(c, launch_run_heads(c->d_raw_keys2, (uint32_t)n, 32, c->d_leaf_start, &vox->n_leaves, c->d_cagg, next_compact_stamp(c), &vox->summary[3], s));
            else
                HIPCHK(c, select_run_heads_u64(c->d_temp, c->temp_bytes, c->d_raw_keys2, (size_t)n, c->d_leaf_start, &vox->n_leaves, s));
        }
        HIPCHK(c, launch_centroids(c->d_undist, c->d_raw_order2, c->d_raw_keys2, c->d_leaf_start, vox, (uint32_t)n, max_leaves,
                                   (uint32_t)c->cap_q, c->d_body, s, compact, vox_next));
        c->vox_clean[c->vox_cur ^ 1] = true;                            // (cleared by the centroid kernel)
        c->vox_cur ^= 1;
        static_assert(offsetof(VoxelGridDev, probe) == offsetof(VoxelGridDev, summary) + 4 * sizeof(uint32_t), "summary and probe are read together");
        HIPCHK(c, read_back(c, summary, vox->summary, sizeof(summary)));
        if (summary[3] != 0u) return c->fail(LIO_ERR_HIP, "lio_scan_voxel_filter", "a bounded wait of a sort / compaction kernel ran out");
        if (may_compact && time_minor) c->vox_saw_unsorted = summary[4] != 0u;            // what the next sweep is taken for
        if (!compact || summary[2] != 0u) break;                        // (an empty / overflowing grid: nothing was sorted, nothing to redo)
        const int need = ((int)summary[5] + 7) / 8;
        const bool wrong = (time_minor && summary[4] != 0u) || need > passes;
        if (need >= 1 && need <= 4) c->vox_passes_guess = need;
        if (!wrong) break;
        ++c->n_voxel_redone;
        if (attempt >= 2) return c->fail(LIO_ERR_HIP, "lio_scan_voxel_filter", "the sort prediction failed three times on one sweep");
        compact = !(time_minor && summary[4] != 0u);                    // out of order: the general path; too few digits: again with `need`
    }
    if (summary[2] & VOX_OVERFLOW)
        return c->fail(LIO_ERR_CAPACITY, "lio_scan_voxel_filter", "leaf size too small for the extent of the scan (index overflow, as in PCL)");
    if (summary[2] & VOX_EMPTY) return scan_reset(c, 0);                // no finite point
    const uint32_t n_real = summary[0] - summary[1];                     // the run of dropped (non-finite) points is last
    if ((int)n_real > c->cap_q) {
        c->have_scan = false;                                            // the scan buffer holds a truncated result
        return c->fail(LIO_ERR_CAPACITY, "lio_scan_voxel_filter", "down-sampled scan exceeds max_scan_points");
    }
    if (n_out) *n_out = (int)n_real;
    return scan_reset(c, (int)n_real);
}

// which = 0: feats_undistort (time-sorted, compensated); 1: the current scan (feats_down_body)
int lio_scan_get_cloud(lio_ctx* c, int which, float* out_xyz, int cap, int* n) {
    if (!c || (which != 0 && which != 1) || cap < 0 || (cap > 0 && !out_xyz)) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (which == 0 && c->have_raw) {
        HIPCHK(c, hipSetDevice(c->device));
        const int rc0 = sort_undistorted_by_time(c);
        if (rc0 != LIO_OK) return rc0;
    }
    const float4* src = which == 0 ? c->d_undist : c->d_body;
    const int cnt = which == 0 ? (c->have_raw ? c->n_raw : 0) : (c->have_scan ? c->nq : 0);
    if (n) *n = cnt;
    const int m = cnt < cap ? cnt : cap;
    if (m <= 0) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, (size_t)m * 3 * sizeof(float));
    if (rc != LIO_OK) return rc;
    HIPCHK(c, launch_pack_xyz(src, (uint32_t)m, (float*)c->d_scratch, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c));
    std::memcpy(out_xyz, c->h_scratch, (size_t)m * 3 * sizeof(float));
    return LIO_OK;
}

// buffers of the callback path's rows (lio_measure_want_rows): staged [nq][13], compacted 13 x m on the device and in pinned
// host memory the kernels can write
static int ensure_rows_buffers(lio_ctx* c) {
    const size_t bytes = (size_t)c->cap_q * 13 * sizeof(double);
    if (!c->d_rows) HIPCHK(c, hipMalloc((void**)&c->d_rows, bytes));
    if (!c->d_rows_c) HIPCHK(c, hipMalloc((void**)&c->d_rows_c, bytes));
    if (!c->h_rows) {
        HIPCHK(c, hipHostMalloc((void**)&c->h_rows, bytes, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_rows_mapped, c->h_rows, 0));
    }
    return LIO_OK;
}
// small results are written over PCIe by the compaction kernel itself (they arrive with the pass, no copy to wait for);
// large ones go through device memory and the copy engine (twice the rate of a kernel's stores to host memory)
static bool rows_direct(const lio_ctx* c, int nq) {
    if (c->dbg_rows_direct >= 0) return c->dbg_rows_direct != 0;
    return (size_t)nq * 13 * sizeof(double) <= (2u << 20);
}

// One pass = lio_measure_begin (enqueue the kernels, return at once) + lio_measure_end (wait for the result record).
// The host work of the caller that does not depend on the result -- in the iterated update: x [-] x_prop, the
// Jacobian-corrected P and its first inversion (esekfom.hpp:1655-1703, :1788) -- runs between the two, beside the GPU.
// ---- assisted pull ---------------------------------------------------------------------------------------------------------
// The first search of a scan handed over as page-locked host records pulls them over PCIe itself; 100 k records of 48 bytes are
// 4.8 MB = 85 us at 57 GB/s, behind which a 23 us search waits.  The host thread that launched the pass would only wait for it:
// instead it packs the FRONT of the cloud into 16-byte records (page-locked staging) and announces its progress in a word the
// search polls; workgroups of the back -- dispatched first -- pull raw records meanwhile.  Share of the front: where packing
// (pack_ns_per_record, measured on every scan: the node's cloud may be warm in L3 or not) and the PCIe bytes of both parts take
// equally long.  Same values by either route: nothing about a result depends on who got to a record first.

static int assist_share(lio_ctx* c, int nq) {
    if (c->dbg_assist_pull == 0 || !c->gate_mail_in_vram || c->pull_host == nullptr || c->pull_stride < 32) return 0;
    {   // (the search kernel has the assisted form where it walks the prefix-form list: 4, 8 or 16 lanes per query, no census)
        const int lanes = c->cfg.lanes_per_query > 0 ? c->cfg.lanes_per_query : auto_lanes(nq);
        const int walk = (lanes == 4 || lanes == 8 || lanes == 16) ? (c->dbg_knn_walk >= 0 ? c->dbg_knn_walk : auto_walk(nq, lanes)) : 0;
        if (c->census || (walk != 1 && walk != 2 && walk != 4)) return 0;
    }
    const double raw_ns = (double)nq * c->pull_stride / PCIE_BYTES_PER_NS;
    if (raw_ns < 15000.0) return 0;                       // (a pull a search hides anyway)
    // The share is steered by what it is meant to shorten: the time from the launch of the scan's first pass to its record
    // (per record of the scan), taken on every assisted scan.  Scans alternate between a share a little above and a little below the
    // current one; every few pairs the share moves towards the faster side.  (A model -- the host's packing speed against the PCIe
    // rate -- does not hold: the speed depends on the share itself, both sides read the same cloud, and on what the caches hold;
    // neither does "host and raw pull finish together": the best share has the host finish later, the search still has the
    // packed front to read and to search.)  A workgroup that gave up on the host (stalled: pre-empted, stopped) cuts the share at once.
    double share;
    if (c->dbg_assist_pull >= 2) {
        share = std::min(c->dbg_assist_pull, 92) / 100.0;
        c->assist_side = -1;
    } else {
        const unsigned long long w = __atomic_load_n(reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<char*>(c->h_small) + 192), __ATOMIC_ACQUIRE);
        if (c->pack_stamp != 0u && (uint32_t)(w >> 32) == c->pack_stamp && (uint32_t)w == 0xffffffffu) {     // (the previous assisted scan's word)
            c->assist_frac = std::max(0.25, c->assist_frac * 0.7);
            c->assist_acc[0] = c->assist_acc[1] = 0.0; c->assist_cnt[0] = c->assist_cnt[1] = 0;
        }
        c->assist_side = c->assist_next_side; c->assist_next_side ^= 1;
        share = std::max(0.2, std::min(0.94, c->assist_frac + (c->assist_side ? 0.03 : -0.03)));
    }
    int front = (int)(share * nq) / 64 * 64;
    if (front < 4096) return 0;
    if (!c->h_pack) {
        if (hipHostMalloc((void**)&c->h_pack, (size_t)c->cap_q * sizeof(float4), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&c->d_pack, c->h_pack, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (c->h_pack) { (void)hipHostFree(c->h_pack); c->h_pack = nullptr; }
            c->dbg_assist_pull = 0;                           // (no staging: every scan is pulled raw)
            return 0;
        }
    }
    if (c->pack_helper != nullptr) {
        // (a helper still packing the previous cloud -- it was pre-empted for longer than a scan -- writes into the staging buffer:
        // this scan goes without help)
        const auto t0 = std::chrono::steady_clock::now();
        while (c->pack_helper->helper_busy.load(std::memory_order_acquire) != 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) return 0;
            __builtin_ia32_pause();
        }
    }
    if (++c->pack_stamp == 0u) ++c->pack_stamp;
    return front;
}
static int assist_chunk(const lio_ctx* c) {
    // records per announcement: a twelfth of the scan, 1024 ... 8192 (announcing costs ~0.1 us, but with 2048-record chunks the
    // packing itself ran at 0.75 instead of 0.43 ns per record beside the search's reads of what was just announced --
    // profiles/r04y_assisted_pull.log; finer than that only delays nothing: a workgroup waits for at most one chunk = ~3.5 us)
    int chunk = c->dbg_assist_chunk;
    if (chunk <= 0) { chunk = (c->pull_n / 12 + 63) / 64 * 64; chunk = std::max(1024, std::min(8192, chunk)); }
    return chunk;
}
// before the launches of the pass: where the calling thread's part ends (0 = it packs everything), the helper's job posted
static int assist_post(lio_ctx* c, int front) {
    PackHelper* h = c->pack_helper;
    if (h == nullptr || front < 16384) return 0;
    // the calling thread still has the launches of the pass in front of it (~6 us): its part is the smaller one
    const int mid = (int)(0.45 * front) / 64 * 64;
    h->job.src = c->pull_host; h->job.stride = (size_t)c->pull_stride; h->job.dst = reinterpret_cast<float*>(c->h_pack);
    h->job.lo = mid; h->job.hi = front; h->job.chunk = assist_chunk(c); h->job.stamp = c->pack_stamp; h->job.word = c->h_pack_progress + 1;
    h->job_seq.store(++c->pack_job_id, std::memory_order_release);
    return mid;
}
static void assist_pack(lio_ctx* c, int front, int mid) {
    const auto t0 = std::chrono::steady_clock::now();
    if (c->dbg_assist_stall_us > 0) ::usleep((useconds_t)c->dbg_assist_stall_us);
    PackJob mine;
    mine.src = c->pull_host; mine.stride = (size_t)c->pull_stride; mine.dst = reinterpret_cast<float*>(c->h_pack);
    mine.lo = 0; mine.hi = mid > 0 ? mid : front; mine.chunk = assist_chunk(c); mine.stamp = c->pack_stamp; mine.word = c->h_pack_progress;
    pack_records(mine);
    if (mid > 0 && c->pack_helper->claim(c->pack_job_id)) {      // the helper has not come for its half (woke up late, or not at all)
        mine.lo = mid; mine.hi = front; mine.word = c->h_pack_progress + 1;
        pack_records(mine);
        ++c->n_helper_late;
    }
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    c->pack_ns_per_record = 0.7 * c->pack_ns_per_record + 0.3 * std::min(ns / std::max(1, mine.hi - mine.lo), 4.0);
    c->last_pack_us = ns * 1e-3; c->last_pack_front = front;
    ++c->n_assisted_pulls;
}

// The kernels of one pass on the stream: [k-NN (search passes)] [measurement kernel].  gate != nullptr: the pose is not known
// yet -- the kernels take it from there (and stand down if told to).  Nothing of the context's pass state changes here
// except the sequence number the pass consumes (*seq_out).
static int enqueue_pass(lio_ctx* c, const Pose14& pose, bool do_search, bool ext, bool separate_walk, const GateState* gate,
                        unsigned long long* seq_out, const GateMail* mail = nullptr, unsigned long long expect = 0ull) {
    hipStream_t s = c->stream;
    const int nq = c->nq;
    // the scan is still in the caller's host records (lio_scan_set_async): its first search pulls it; anything else pulls first
    const bool pull_here = c->pull_pending && do_search && gate == nullptr && nq == c->pull_n && nq <= FUSED_TRANSFORM_MAX_QUERIES;
    if (c->pull_pending && !pull_here) flush_pull(c);
    const int pack_front = pull_here ? assist_share(c, nq) : 0;
    const int pack_mid = pack_front > 0 ? assist_post(c, pack_front) : 0;
    if (pack_front > 0) c->assist_t0_ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count();   // (passes armed behind this one come through here too)
    if (do_search) {
        // a later search of the same scan re-uses the neighbours of the previous one as a bound
        const int rc = run_knn(c, c->d_body, c->d_world, pose, nq, c->cap_q, c->d_nbr, c->d_cnt, c->d_ulist, c->d_useeds, c->hint_valid, separate_walk, gate, pull_here, mail, expect, pack_front, pack_mid);
        if (rc != LIO_OK) return rc;
        if (pull_here) c->pull_pending = false;
    }
    ResidualArgs ra{};
    ra.body = c->d_body; ra.nq = nq; ra.nq_stride = c->cap_q; ra.pose = pose; ra.gate = gate; ra.nbr = c->d_nbr; ra.cnt = c->d_cnt;
    if (mail != nullptr && !do_search) { ra.mail = mail; ra.expect = expect; ra.gate = nullptr; }   // (the first kernel of the pass reads the host's word)
    ra.sel = c->d_sel; ra.world = c->d_world; ra.normvec = c->d_normvec; ra.plane = c->d_plane; ra.rows = nullptr; ra.units = c->d_units; ra.gunits = c->d_gunits;
    if ((uint32_t)(c->seq + 1ull) == 0u) {
        // The partial sums carry the low 32 bits of the sequence number as their stamp (lio_measure.hip): when those go round, a
        // unit left by a pass 2^32 passes ago -- a larger scan's grid had workgroups this one lacks -- must not pass for this pass's.
        // Stamp 0 is never used; the units start over from zero (once in 2^32 passes: ~39 h at 30 k passes/s).
        const size_t blocks = (size_t)residual_blocks(c->cap_q), groups = (blocks + RED_GROUP - 1) / RED_GROUP;
        (void)hipMemsetAsync(c->d_units, 0, blocks * RED_STRIDE * 2 * sizeof(unsigned long long), s);
        (void)hipMemsetAsync(c->d_gunits, 0, groups * RED_STRIDE * 2 * sizeof(unsigned long long), s);
        (void)hipMemsetAsync(c->d_lunits, 0, (size_t)FIX_MAX * RED_STRIDE * 2 * sizeof(unsigned long long), s);
        ++c->seq;
    }
    const unsigned long long seq = ++c->seq;
    *seq_out = seq;
    ra.n_unresolved = c->d_counters; ra.cand_total = c->d_cand; ra.out = c->d_out; ra.seq = seq;
    ra.debug_fail = (c->dbg_force_timeout ? 1 : 0) | (c->dbg_stamps ? 2 : 0);
    if (c->want_rows && gate == nullptr) {
        const int rc = ensure_rows_buffers(c);
        if (rc != LIO_OK) return rc;
        ra.rows = c->d_rows; ra.m_dev = reinterpret_cast<int*>(c->d_counters + 18);
    }
    if (do_search && !separate_walk) {
        // (walker workgroups sized by the previous pass's list instead of the scan -- 64-128 instead of 390 at C2 -- changed nothing:
        // A/B on one box, three alternations, median step 115.8-116.7 vs 115.3-117.1 us; idle workgroups are not what the ramp costs)
        ra.n_walkers = residual_walkers(nq);
        ra.walk_done = c->d_counters + 16;
        ra.tree = tree_args(c, nq, c->cap_q, c->d_nbr, c->d_cnt, c->d_ulist, c->d_useeds);
        ra.lunits = c->d_lunits;
    }
    const hipError_t le = launch_residual(ra, do_search, ext, s, prof_timer(c, LIO_PROF_RESIDUAL));
    if (pack_front > 0) assist_pack(c, pack_front, pack_mid);   // (whatever the launch said: the search is waiting for these records)
    if (le != hipSuccess) return c->fail(LIO_ERR_HIP, "launch_residual", hipGetErrorString(le));
    return LIO_OK;
}
// The exact walk of the unproven queries normally rides inside the measurement kernel (tens of queries: nobody waits
// for it).  When the LAST search pass listed many (a scan at the map's edge, C5's 1 M-point scan: thousands), nearly
// every model workgroup would sit waiting for 1-wavefront walkers: then the walk gets the whole GPU in a launch of
// its own, as in round 1 -- and so it does for scans far beyond the node's 100 k points, where the measurement kernel
// is throughput-bound and the lean variant (without the walk's registers) runs at nearly twice the occupancy.
static bool walk_in_its_own_launch(const lio_ctx* c) {
    return c->dbg_separate_walk != 0 || c->listed_last_search > INLINE_WALK_MAX_LISTED || c->nq > FUSED_TRANSFORM_MAX_QUERIES;
}

// ---- passes enqueued ahead of their pose -----------------------------------------------------------------------------------
// Between "the host has the new pose" and "the first wavefront of the next pass runs" lie the launch calls of the pass and
// the dispatch of its first kernel (~6 us, scripts/micro/gate_latency.hip).  Inside the iterated update the next pass is
// predictable (the converge pattern of the previous scan's update), so it is put on the stream while the GPU still works on
// the current one, behind a gate kernel; when the pose is known the host only writes it, with a sequence number, into the
// gate's mail.  A wrong prediction (or the end of the update) calls the pass off: its kernels see GateState.skip and return.
// lio_measure_begin arms the passes it expects next (arm_predicted); every other entry point that needs the stream calls them
// off first (enter()), and a gate nobody answers gives up by itself after two seconds.
static void gate_write(lio_ctx* c, unsigned long long mail_seq, unsigned long long cmd, const Pose14* pose) {
    unsigned long long w[16];
    w[0] = mail_seq; w[1] = cmd;
    for (int i = 0; i < 14; ++i) { double v = pose ? pose->v[i] : 0.0; std::memcpy(&w[2 + i], &v, 8); }
    GateMail* mail = c->h_gate_mail + (mail_seq % lio_ctx::GATE_SLOTS);
    volatile unsigned long long* m = mail->w;
    for (int i = 1; i < 16; ++i) m[i] = w[i];
    m[16] = gate_check(w);
    if (c->gate_mail_in_vram) __builtin_ia32_sfence();      // (write-combined BAR mapping: the sixteen words leave before the seventeenth)
    __atomic_store_n(&mail->w[0], w[0], __ATOMIC_RELEASE);
    if (c->gate_mail_in_vram) __builtin_ia32_sfence();      // ... and nothing lingers in a write-combining buffer
}
// every armed pass stands down (a prediction failed, the update ended, or something else needs the stream)
static void gate_call_off(lio_ctx* c) {
    for (int i = 0; i < c->gate_n; ++i) {
        gate_write(c, c->gate_q[i].mail_seq, GATE_SKIP, nullptr);
        ++c->n_gate_called_off;
    }
    c->gate_n = 0;
}
// the update of the current scan is over: what was armed for IT stands down; the first pass of the next scan stays
static void gate_call_off_this_scan(lio_ctx* c) {
    int kept = 0;
    for (int i = 0; i < c->gate_n; ++i) {
        if (c->gate_q[i].next_scan) { c->gate_q[kept++] = c->gate_q[i]; continue; }
        gate_write(c, c->gate_q[i].mail_seq, GATE_SKIP, nullptr);
        ++c->n_gate_called_off;
    }
    c->gate_n = kept;
}
// enqueue [gate][pass of the given kind] behind whatever is on the stream; LIO_OK also when the context declines (nothing armed)
static int gate_arm(lio_ctx* c, bool do_search, bool ext) {
    if (c->gate_n >= lio_ctx::GATE_AHEAD || !c->dbg_pre_enqueue || c->want_rows || c->nq == 0 || !c->have_search) return LIO_OK;
    if (walk_in_its_own_launch(c)) return LIO_OK;               // (the stand-alone walk and the transform launch read no gate)
    // (kernel timing on: a full event pool is drained with a stream synchronisation -- never behind an armed gate)
    for (int w = 0; c->prof != 0u && w < LIO_PROF_COUNT; ++w) if (c->ev_next[w] + 16 > 8192) return LIO_OK;
    const unsigned long long mail_seq = ++c->gate_mail_seq;
    GateState* st = c->d_gate + (mail_seq % lio_ctx::GATE_SLOTS);
    const GateMail* slot = c->d_gate_mail + (mail_seq % lio_ctx::GATE_SLOTS);
    // Mail slots in device memory: the first kernel of the pass waits for the host's word itself (gate_poll, lio_device.h) --
    // one dependent dispatch less between the host's word and the pass.  Slots in host memory (no large BAR): a one-wavefront
    // gate kernel in front does the waiting (hundreds of workgroups polling over PCIe starve each other).
    const bool in_kernel = c->gate_mail_in_vram && (c->gate_in_kernel & (do_search ? 2 : 1)) != 0;
    if (!in_kernel) {
        const hipError_t ge = launch_gate(slot, mail_seq, st, c->stream);
        if (ge != hipSuccess) return c->fail(LIO_ERR_HIP, "launch_gate", hipGetErrorString(ge));
    } else {
        ++c->n_gate_in_kernel;
    }
    lio_ctx::ArmedPass& ap = c->gate_q[c->gate_n++];
    ap = lio_ctx::ArmedPass{};
    ap.search = do_search; ap.ext = ext; ap.mail_seq = mail_seq; ap.pass_seq = 0; ap.hint = c->hint_valid;
    Pose14 none{};
    const int rc = enqueue_pass(c, none, do_search, ext, false, st, &ap.pass_seq, in_kernel ? slot : nullptr, mail_seq);
    if (rc != LIO_OK) { gate_call_off(c); return rc; }
    c->stamp('G');
    return LIO_OK;
}

static int gate_arm_next_scan(lio_ctx* c, bool ext) {
    for (int i = 0; i < c->gate_n; ++i) if (c->gate_q[i].next_scan) return LIO_OK;          // (once)
    if (c->gate_n >= lio_ctx::GATE_AHEAD || !c->dbg_pre_enqueue || c->want_rows || c->upload_pending || c->nq_back <= 0 || c->pull_pending) return LIO_OK;
    if (c->dbg_separate_walk != 0 || c->listed_last_search > INLINE_WALK_MAX_LISTED || c->nq_back > FUSED_TRANSFORM_MAX_QUERIES) return LIO_OK;
    if (c->prof != 0u) return LIO_OK;          // (kernel timing on: an event pair around a pass that may be called off would count as a launch)
    // the copy of that scan: usually long finished (it ran beside this scan's first pass); otherwise the stream waits for it here
    const hipError_t q = hipEventQuery(c->ev_copy);
    if (q == hipErrorNotReady) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
    else if (q != hipSuccess) return c->fail(LIO_ERR_HIP, "gate_arm_next_scan", hipGetErrorString(q));
    c->stamp(q == hipErrorNotReady ? '!' : '1');
    const unsigned long long mail_seq = ++c->gate_mail_seq;
    GateState* st = c->d_gate + (mail_seq % lio_ctx::GATE_SLOTS);
    const GateMail* slot = c->d_gate_mail + (mail_seq % lio_ctx::GATE_SLOTS);
    // This wait is as long as the host takes from one scan to the next (~10 us in a replay loop, the scan period in a live node):
    // by default it is a gate kernel's -- one wavefront -- so that the search kernel's own duration stays its work (what rocprofv3
    // and the event stamps report for it) and the GPU is not full of polling workgroups in between; bit 2 of "gate_in_kernel"
    // moves it into the search kernel like the waits inside a scan (one dependent dispatch less, A/B switch).
    const bool in_kernel = c->gate_mail_in_vram && (c->gate_in_kernel & 4) != 0;
    if (!in_kernel) {
        const hipError_t ge = launch_gate(slot, mail_seq, st, c->stream);
        if (ge != hipSuccess) return c->fail(LIO_ERR_HIP, "launch_gate", hipGetErrorString(ge));
    } else {
        ++c->n_gate_in_kernel;
    }
    c->stamp('2');
    lio_ctx::ArmedPass& ap = c->gate_q[c->gate_n++];
    ap = lio_ctx::ArmedPass{};
    ap.search = true; ap.ext = ext; ap.mail_seq = mail_seq; ap.pass_seq = 0;
    ap.next_scan = true; ap.across = true; ap.body = c->d_body_back; ap.nq = c->nq_back; ap.hint = false;
    // the pass is enqueued as the first pass of THAT scan: its points, its count, no neighbours of an earlier search to bound it
    float4* const body_now = c->d_body; const int nq_now = c->nq; const bool hint_now = c->hint_valid;
    c->d_body = c->d_body_back; c->nq = c->nq_back; c->hint_valid = false;
    Pose14 none{};
    const int rc = enqueue_pass(c, none, true, ext, false, st, &ap.pass_seq, in_kernel ? slot : nullptr, mail_seq);
    c->d_body = body_now; c->nq = nq_now; c->hint_valid = hint_now;
    if (rc != LIO_OK) { gate_call_off(c); return rc; }
    ++c->n_next_scan_armed;
    c->stamp('N');
    return LIO_OK;
}

// The passes that follow the one just put on the stream, as the previous scan's update ran them (pattern_prev): enqueued now,
// each behind a gate (gate_arm) -- during a search pass there is time to arm two, during a short one there is not.  Any caller
// of lio_measure / lio_measure_begin gets this (the reference's callback included: h_share_model -> lio_measure); a prediction
// that fails, or any other entry point that needs the stream, calls the armed passes off (enter()).
static int arm_predicted(lio_ctx* c, bool searching, bool ext) {
    const int arm_up_to = searching ? lio_ctx::GATE_AHEAD : 1;
    for (size_t next = c->pattern_cur.size() + (size_t)c->gate_n; c->gate_n < arm_up_to && next < c->pattern_prev.size(); ++next) {
        bool kind = c->pattern_prev[next] != 0;
        if (c->dbg_gate_mispredict) kind = !kind;
        const int before = c->gate_n;
        const int rc = gate_arm(c, kind, ext);
        if (rc != LIO_OK) return rc;
        if (c->gate_n == before) return LIO_OK;                 // the context declined
    }
    // Everything the previous update predicts for THIS scan is running or armed: the scan that waits in the back buffer comes next,
    // and its first pass always searches -- onto the stream behind them (lio_eskf_scan_step_pipelined only: between two of its calls
    // nothing else uses the stream).  The launch calls and the dispatch of that search then lie beside this scan's last passes
    // instead of between the scans (scripts/step_timeline.py: 12-16 us of an idle GPU per scan boundary before).  lio_scan_swap
    // keeps the pass armed; the first lio_measure_begin of the scan hands it its pose; anything else calls it off.
    if (c->arm_across_scans && c->dbg_arm_next_scan && !c->across_off_this_scan && c->gate_n < arm_up_to &&
        c->pattern_cur.size() + (size_t)c->gate_n >= c->pattern_prev.size() && !c->pattern_prev.empty())
        return gate_arm_next_scan(c, ext);
    return LIO_OK;
}

int lio_measure_begin(lio_ctx* c, const double state14[14], int do_search, int extrinsic_est_en) {
    if (!c || !state14) return LIO_ERR_INVALID_ARG;
    c->stamp(do_search ? 'B' : 'b');
    if (c->pass_in_flight) return c->fail(LIO_ERR_INVALID_ARG, "lio_measure_begin", "the previous pass has not been collected (lio_measure_end)");
    if (!c->have_scan) return c->fail(LIO_ERR_NO_SCAN, "lio_measure", "no scan set");
    if (!do_search && !c->have_search) return c->fail(LIO_ERR_NO_SCAN, "lio_measure", "first pass of a scan must search");
    if (c->map_broken) return c->fail(LIO_ERR_NO_MAP, "lio_measure", "the map index is undefined after a failed update: Build / reconstruct it");
    HIPCHK(c, hipSetDevice(c->device));
    const int nq = c->nq;
    Pose14 pose;
    for (int i = 0; i < 14; ++i) pose.v[i] = state14[i];
    const bool ext = extrinsic_est_en != 0;
    bool released = false;
    if (c->gate_n > 0) {
        const lio_ctx::ArmedPass ap = c->gate_q[0];
        if (ap.search == (do_search != 0) && ap.ext == ext && !c->want_rows && nq != 0 && !ap.next_scan) {
            // the pass is already on the stream: hand it its pose
            gate_write(c, ap.mail_seq, GATE_GO, &pose);
            for (int i = 1; i < c->gate_n; ++i) c->gate_q[i - 1] = c->gate_q[i];
            --c->gate_n;
            ++c->n_gate_released;
            if (ap.across) ++c->n_next_scan_released;
            c->pass_seq = ap.pass_seq;
            c->pass_hint = ap.hint;
            released = true;
        } else {
            if (ap.next_scan) c->across_off_this_scan = true;   // (the update runs longer than the previous one did)
            gate_call_off(c);
        }
    }
    if (released) c->stamp('r');
    c->last_pose = pose; c->last_ext = ext;
    c->pass_in_flight = true; c->pass_empty = nq == 0; c->pass_gated = released;
    if (nq == 0) return LIO_OK;                               // "No Effective Points" laserMapping.cpp:1956
    hipStream_t s = c->stream;
    c->pass_searched = do_search != 0;
    c->rows_staged = false;
    c->pattern_cur.push_back(do_search ? 1 : 0);
    if (!released) {
        c->pass_hint = c->hint_valid;
        const int rc = enqueue_pass(c, pose, do_search != 0, ext, walk_in_its_own_launch(c), nullptr, &c->pass_seq);
        if (rc != LIO_OK) { c->pass_in_flight = false; return rc; }
    }
    if (do_search) { c->have_search = true; c->hint_valid = true; }
    c->stamp('L');
    if (c->want_rows) {
        // right behind the pass: the rows of the selected points, compacted in scan order, column-major (m read on the device)
        RowsCompactArgs rc{};
        rc.rows = c->d_rows; rc.sel = c->d_sel; rc.nq = nq; rc.m_dev = reinterpret_cast<const int*>(c->d_counters + 18);
        rc.dst = rows_direct(c, nq) ? c->d_rows_mapped : c->d_rows_c;
        rc.agg = c->d_cagg; rc.stamp = next_compact_stamp(c); rc.err = c->d_apply + 15;
        const hipError_t ce = launch_rows_compact(rc, s);
        if (ce != hipSuccess) { lio_out dropped; (void)lio_measure_end(c, &dropped); return c->fail(LIO_ERR_HIP, "launch_rows_compact", hipGetErrorString(ce)); }
        c->rows_staged = true;
    }
    // the passes behind this one, as predicted: onto the stream now (their launch calls run beside the GPU's work)
    {
        const int rc = arm_predicted(c, do_search != 0, ext);
        if (rc != LIO_OK) {                                   // the pass itself is fine: collect it, report
            const std::string why = c->err;
            lio_out dropped;
            (void)lio_measure_end(c, &dropped);
            return c->fail(rc, "lio_measure_begin (arming the next pass)", why.c_str());
        }
    }
    // the plan of a gathered batch the deferred exchange holds back: onto the side stream now, beside this scan step (loop-back:
    // beside the scan step of the LAST virtual rank -- the one a real rank's would run beside)
    if (c->def.plan_owed && !c->want_rows && (!c->rep_loopback || c->rep_rank + 1 == c->rep_world)) {
        const int rc = deferred_launch_plan(c);
        if (rc != LIO_OK) {
            const std::string why = c->err;
            lio_out dropped;
            (void)lio_measure_end(c, &dropped);
            return c->fail(rc, "lio_measure_begin (planning the deferred batch)", why.c_str());
        }
    }
    // the next scan's upload, if one was handed over (lio_scan_upload_async): issued here, behind this pass's
    // launches, so that the runtime calls of the copy run while the GPU is busy with the pass
    if (c->upload_pending) {
        const int rc = issue_pending_upload(c);
        if (rc != LIO_OK) {                                   // the pass itself is fine: collect it, report the copy
            const std::string why = c->err;
            lio_out dropped;
            (void)lio_measure_end(c, &dropped);
            return c->fail(rc, "lio_scan_upload_async", why.c_str());
        }
    }
    return LIO_OK;
}

static int lio_measure_end_again(lio_ctx* c, lio_out* out) { c->pass_in_flight = true; return lio_measure_end(c, out); }
int lio_measure_end(lio_ctx* c, lio_out* out) {
    if (!c || !out) return LIO_ERR_INVALID_ARG;
    if (!c->pass_in_flight) return c->fail(LIO_ERR_INVALID_ARG, "lio_measure_end", "no pass in flight (lio_measure_begin)");
    c->pass_in_flight = false;
    std::memset(out, 0, sizeof(*out));
    if (c->pass_empty) { out->valid = 0; return LIO_OK; }
    hipStream_t s = c->stream;
    const unsigned long long seq = c->pass_seq;
    c->stamp('E');
    // The record lands in host-mapped memory; poll its sequence number instead of paying for a
    // device-to-host copy plus a stream synchronisation.  hipStreamQuery guards against a fault.
    {
        volatile unsigned long long* pseq = &c->h_out->seq;
        unsigned spins = 0;
        for (;;) {
            if (__atomic_load_n(pseq, __ATOMIC_ACQUIRE) == seq) break;
            if ((++spins & 0x3fff) == 0) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) {
                    if (__atomic_load_n(pseq, __ATOMIC_ACQUIRE) == seq) break;
                    if (c->pass_gated) {
                        // the pass sat behind a gate that gave up before the host's word came (a host stalled for seconds:
                        // debugger, SIGSTOP, VM pause) and stood down: put it on the stream directly, once
                        c->pass_gated = false;
                        ++c->n_gate_reenqueued;
                        gate_call_off(c);
                        // with the bound it was ARMED with: lio_measure_begin has since set hint_valid for every search pass, and the
                        // first search of a scan armed across the scan boundary must not bound on the previous scan's neighbours
                        const bool hint_now = c->hint_valid;
                        c->hint_valid = c->pass_hint;
                        const int rc = enqueue_pass(c, c->last_pose, c->pass_searched, c->last_ext, walk_in_its_own_launch(c), nullptr, &c->pass_seq);
                        c->hint_valid = hint_now;
                        if (rc != LIO_OK) return rc;
                        return lio_measure_end_again(c, out);
                    }
                    return c->fail(LIO_ERR_HIP, "lio_measure", "stream idle but result record not published");
                }
                if (q != hipErrorNotReady) return c->fail(LIO_ERR_HIP, "lio_measure", hipGetErrorString(q));
            }
            __builtin_ia32_pause();
        }
        c->stamp('S', spins);
    }
    if (c->assist_t0_ns > 0.0) {                             // the first pass of an assisted pull: its time steers the share (assist_share)
        const double t = (std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->assist_t0_ns) / std::max(1, c->nq);
        c->assist_t0_ns = 0.0;
        if (c->assist_side >= 0) {
            c->assist_acc[c->assist_side] += t; ++c->assist_cnt[c->assist_side];
            if (c->assist_cnt[0] >= 3 && c->assist_cnt[1] >= 3) {
                const double below = c->assist_acc[0] / c->assist_cnt[0], above = c->assist_acc[1] / c->assist_cnt[1];
                // (1.5 %: at C2 six points of share are worth ~6 % of a first pass; where they are worth less -- C3 -- the share stays put)
                // Steering stays within 55 ... 90 %: every host and cloud measured has its best share there (cold DRAM 70, warm L3 90);
                // below that only a workgroup that gave up takes it (assist_share), and steering brings it back up.
                if (above < below * 0.985) c->assist_frac = std::min(0.9, c->assist_frac + 0.03);
                else if (below < above * 0.985) c->assist_frac = std::max(std::min(c->assist_frac, 0.55), c->assist_frac - 0.03);
                c->assist_acc[0] = c->assist_acc[1] = 0.0; c->assist_cnt[0] = c->assist_cnt[1] = 0;
            }
        }
    }

    const DeviceOut& d = *c->h_out;
    if (d.n_unresolved < 0) return c->fail(LIO_ERR_HIP, "lio_measure", "tree walk exceeded its stack bound");
    if (d.m < 0) return c->fail(LIO_ERR_HIP, "lio_measure", "a bounded wait inside the measurement kernel ran out (sums incomplete)");
    out->m = d.m;
    out->valid = d.m >= 1 ? 1 : 0;                        // laserMapping.cpp:1956-1961
    out->n_unresolved = d.n_unresolved;
    out->res_mean = d.m >= 1 ? d.res_sum / (double)d.m : 0.0;   // :1963
    std::memcpy(out->HTH, d.HTH, sizeof(out->HTH));
    std::memcpy(out->HTh, d.HTh, sizeof(out->HTh));
    out->n_candidates = d.n_candidates;
    c->last_m = d.m; c->have_pass = true;
    if (c->pass_searched) c->listed_last_search = d.n_unresolved;
    return LIO_OK;
}

int lio_measure(lio_ctx* c, const double state14[14], int do_search, int extrinsic_est_en, lio_out* out) {
    if (!c || !state14 || !out) return LIO_ERR_INVALID_ARG;
    const int rc = lio_measure_begin(c, state14, do_search, extrinsic_est_en);
    if (rc != LIO_OK) return rc;
    return lio_measure_end(c, out);
}

int lio_measure_want_rows(lio_ctx* c, int on) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);                                   // (an armed pass was built for the old setting)
    c->want_rows = on != 0;
    return LIO_OK;
}

int lio_measure_rows(lio_ctx* c, double* h_x, double* h, int cap_rows, int col_major) {
    if (!c || !h_x || !h) return LIO_ERR_INVALID_ARG;
    if (!c->have_pass) return c->fail(LIO_ERR_NO_SCAN, "lio_measure_rows", "no measurement pass yet");
    HIPCHK(c, hipSetDevice(c->device));
    gate_call_off(c);                           // (inside the iterated update: nothing may queue behind a waiting gate)
    const int nq = c->nq, m = c->last_m;
    if (cap_rows < m) return c->fail(LIO_ERR_CAPACITY, "lio_measure_rows", "cap_rows < m");
    if (m == 0) return LIO_OK;
    hipStream_t s = c->stream;
    if (c->rows_staged) {
        // the pass left them compacted and column-major (lio_measure_want_rows): nothing is re-evaluated, nothing but the m
        // rows crosses PCIe -- written by the kernel itself (small scans) or moved by the copy engine (large ones)
        if (!rows_direct(c, nq)) {
            // destinations the runtime knows as page-locked (the adaptor registers ekfom_data.h_x / .h, lio_host_register): the copy
            // engine writes them where they are -- no pass through h_rows, no 9.6 MB host copy per pass at 100 k points
            if (col_major && is_pinned_host(h_x) && is_pinned_host(reinterpret_cast<char*>(h_x) + (size_t)m * 12 * sizeof(double) - 1) &&
                is_pinned_host(h) && is_pinned_host(reinterpret_cast<char*>(h) + (size_t)m * sizeof(double) - 1)) {
                HIPCHK(c, hipMemcpyAsync(h_x, c->d_rows_c, (size_t)m * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipMemcpyAsync(h, c->d_rows_c + (size_t)12 * m, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
                HIPCHK(c, stream_wait(c));
                ++c->n_rows_direct_dma;
                return LIO_OK;
            }
            HIPCHK(c, hipMemcpyAsync(c->h_rows, c->d_rows_c, (size_t)m * 13 * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(c, stream_wait(c));
        const double* src = c->h_rows;
        if (col_major) {
            std::memcpy(h_x, src, (size_t)m * 12 * sizeof(double));
        } else {
            for (int cidx = 0; cidx < 12; ++cidx)
                for (int r = 0; r < m; ++r) h_x[(size_t)r * 12 + cidx] = src[(size_t)cidx * m + r];
        }
        std::memcpy(h, src + (size_t)12 * m, (size_t)m * sizeof(double));
        return LIO_OK;
    }
    if (!c->d_rows) HIPCHK(c, hipMalloc((void**)&c->d_rows, (size_t)c->cap_q * 13 * sizeof(double)));
    // Re-evaluate the rows of the selected points with the pose of the last pass: selection flags
    // are a fixed point of the no-search pass (same neighbours, same pose -> same decisions).
    ResidualArgs ra{};
    ra.body = c->d_body; ra.nq = nq; ra.nq_stride = c->cap_q; ra.pose = c->last_pose; ra.nbr = c->d_nbr; ra.cnt = c->d_cnt;
    ra.sel = c->d_sel; ra.world = c->d_world; ra.normvec = c->d_normvec; ra.plane = c->d_plane; ra.rows = c->d_rows; ra.units = c->d_units; ra.gunits = c->d_gunits;
    ra.n_unresolved = c->d_counters; ra.cand_total = c->d_cand; ra.out = c->d_out;
    ra.seq = ++c->seq;                          // the record is rewritten with identical content
    HIPCHK(c, launch_residual(ra, false, c->last_ext, s));
    const size_t b_rows = (size_t)nq * 13 * sizeof(double);
    int rc = ensure_scratch(c, b_rows + (size_t)nq);
    if (rc != LIO_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_rows, b_rows, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync((char*)c->h_scratch + b_rows, c->d_sel, (size_t)nq, hipMemcpyDeviceToHost, s));
    HIPCHK(c, stream_wait(c));
    const double* rows = (const double*)c->h_scratch;
    const uint8_t* sel = (const uint8_t*)c->h_scratch + b_rows;
    int r = 0;
    for (int i = 0; i < nq && r < m; ++i) {               // stable compaction, laserMapping.cpp:1944-1954
        if (!sel[i]) continue;
        const double* src = rows + (size_t)i * 13;
        for (int cidx = 0; cidx < 12; ++cidx) {
            if (col_major) h_x[(size_t)cidx * m + r] = src[cidx];
            else h_x[(size_t)r * 12 + cidx] = src[cidx];
        }
        h[r] = src[12];
        ++r;
    }
    if (r != m) return c->fail(LIO_ERR_HIP, "lio_measure_rows", "row count changed between passes");
    return LIO_OK;
}

int lio_get_neighbors(lio_ctx* c, float* out_xyz, float* out_sqdist, int* out_count) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (!c->have_search) return c->fail(LIO_ERR_NO_SCAN, "lio_get_neighbors", "no search pass yet");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->nq;
    if (n == 0) return LIO_OK;
    const size_t b_xyz = (size_t)n * K * 3 * sizeof(float), b_d = (size_t)n * K * sizeof(float), b_c = (size_t)n * sizeof(int);
    int rc = ensure_scratch(c, b_xyz + b_d + b_c);
    if (rc != LIO_OK) return rc;
    float* dx = (float*)c->d_scratch;
    float* dd = (float*)((char*)c->d_scratch + b_xyz);
    int* dc = (int*)((char*)c->d_scratch + b_xyz + b_d);
    HIPCHK(c, launch_unpack_neighbors(c->d_nbr, c->d_cnt, n, c->cap_q, dx, dd, dc, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_scratch, c->d_scratch, b_xyz + b_d + b_c, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c));
    if (out_xyz) std::memcpy(out_xyz, c->h_scratch, b_xyz);
    if (out_sqdist) std::memcpy(out_sqdist, (char*)c->h_scratch + b_xyz, b_d);
    if (out_count) std::memcpy(out_count, (char*)c->h_scratch + b_xyz + b_d, b_c);
    return LIO_OK;
}

int lio_get_scan_state(lio_ctx* c, float* world_xyz, uint8_t* selected, float* normvec) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (!c->have_pass) return c->fail(LIO_ERR_NO_SCAN, "lio_get_scan_state", "no measurement pass yet");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->nq;
    if (n == 0) return LIO_OK;
    const size_t b4 = (size_t)n * sizeof(float4);
    int rc = ensure_scratch(c, 2 * b4 + (size_t)n);
    if (rc != LIO_OK) return rc;
    char* hs = (char*)c->h_scratch;
    HIPCHK(c, hipMemcpyAsync(hs, c->d_world, b4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + b4, c->d_normvec, b4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + 2 * b4, c->d_sel, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c));
    const float4* w = (const float4*)hs;
    const float4* nv = (const float4*)(hs + b4);
    const uint8_t* sl = (const uint8_t*)(hs + 2 * b4);
    for (int i = 0; i < n; ++i) {
        if (world_xyz) { world_xyz[3 * i] = w[i].x; world_xyz[3 * i + 1] = w[i].y; world_xyz[3 * i + 2] = w[i].z; }
        if (normvec) {
            // normvec is only written for selected points (laserMapping.cpp:1932-1936)
            if (sl[i]) { normvec[4 * i] = nv[i].x; normvec[4 * i + 1] = nv[i].y; normvec[4 * i + 2] = nv[i].z; normvec[4 * i + 3] = nv[i].w; }
            else { normvec[4 * i] = normvec[4 * i + 1] = normvec[4 * i + 2] = normvec[4 * i + 3] = 0.0f; }
        }
        if (selected) selected[i] = sl[i];
    }
    return LIO_OK;
}

// ---- profiling -----------------------------------------------------------------------------------
int lio_profile_enable(lio_ctx* c, int on) {
    if (!c) return LIO_ERR_INVALID_ARG;
    // (no enter(): what is armed stays armed, with or without its event pair -- gate_arm / gate_arm_next_scan look at the pools and
    // at `prof` when THEY launch, and a direct launch never happens with anything armed)
    HIPCHK(c, hipSetDevice(c->device));
    // (the recorded pairs are resolved by lio_profile_get / lio_profile_reset -- that takes a drained stream -- or when a pool
    // runs full; switching the timers on and off costs nothing, so a caller can time every n-th step of a loop)
    // on == 1: every family; otherwise a bit mask (1 << LIO_PROF_x) | ... ; 0 = off
    c->prof = on == 1 ? 0xffu : ((unsigned)on & 0xffu);
    return LIO_OK;
}
int lio_debug_set(lio_ctx* c, const char* key, long long value) {
    if (!c || !key) return LIO_ERR_INVALID_ARG;
    enter(c);                                   // (switches change what a pass does: none stays armed across one)
    if (std::strcmp(key, "force_reduction_timeout") == 0) { c->dbg_force_timeout = value != 0; return LIO_OK; }
    if (std::strcmp(key, "separate_tree_walk") == 0) { c->dbg_separate_walk = value != 0; return LIO_OK; }
    if (std::strcmp(key, "library_sort") == 0) { c->dbg_library_sort = value != 0; return LIO_OK; }
    if (std::strcmp(key, "stamps") == 0) { c->dbg_stamps = value != 0; return LIO_OK; }
    if (std::strcmp(key, "knn_tune") == 0) { c->dbg_knn_tune = (int)value; return LIO_OK; }
    if (std::strcmp(key, "voxel_fast_path") == 0) { c->dbg_voxel_fast = value != 0; return LIO_OK; }
    if (std::strcmp(key, "knn_walk") == 0) {
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4) return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_set", "knn_walk: -1, 0, 1, 2 or 4");
        c->dbg_knn_walk = (int)value;
        return LIO_OK;
    }
    if (std::strcmp(key, "lanes_per_query") == 0) {            // (0 = by scan size again)
        if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_set", "lanes_per_query: 0, 4, 8, 16, 32 or 64");
        c->cfg.lanes_per_query = (int)value;
        return LIO_OK;
    }
    if (std::strcmp(key, "rows_direct") == 0) { c->dbg_rows_direct = (int)value; return LIO_OK; }
    if (std::strcmp(key, "assist_pull") == 0) { c->dbg_assist_pull = (int)value; return LIO_OK; }
    if (std::strcmp(key, "assist_pull_chunk") == 0) {
        if (value < 0 || value % 64 != 0 || value > 65536) return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_set", "assist_pull_chunk: 0 (by scan size) or a multiple of 64 up to 65536");
        c->dbg_assist_chunk = (int)value; return LIO_OK;
    }
    if (std::strcmp(key, "assist_pull_threads") == 0) return lio_scan_pack_threads(c, (int)value);
    if (std::strcmp(key, "assist_pull_stall_us") == 0) { c->dbg_assist_stall_us = (int)value; return LIO_OK; }
    if (std::strcmp(key, "fused_pull") == 0) { c->dbg_fused_pull = value != 0; return LIO_OK; }
    if (std::strcmp(key, "retired_budget") == 0) { c->dbg_retired_budget = value; return LIO_OK; }
    if (std::strcmp(key, "pre_enqueue") == 0) { c->dbg_pre_enqueue = value != 0; return LIO_OK; }
    if (std::strcmp(key, "gate_mispredict") == 0) { c->dbg_gate_mispredict = value != 0; return LIO_OK; }
    if (std::strcmp(key, "gate_in_kernel") == 0) { enter(c); c->gate_in_kernel = (int)(value & 7); return LIO_OK; }
    if (std::strcmp(key, "dump_host_trace") == 0) { if (c->trace_host) dump_host_trace(c, (size_t)(value > 0 ? value : 80)); return LIO_OK; }
    if (std::strcmp(key, "arm_next_scan") == 0) { enter(c); c->dbg_arm_next_scan = value != 0; return LIO_OK; }
    if (std::strcmp(key, "sequence_numbers") == 0) {           // test hook: the next pass / compaction launch take these numbers + 1 (stamp wrap-around)
        c->seq = (unsigned long long)value; c->compact_seq = (uint32_t)value; c->gsort_seq = (uint32_t)value;
        return LIO_OK;
    }
    if (std::strcmp(key, "replicate_loopback") == 0) { c->dbg_rep_loopback = value != 0; return LIO_OK; }
    if (std::strcmp(key, "replicate_loopback_rank") == 0) {
        if (!c->rep_loopback || value < 0 || value >= c->rep_world) return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_set", "replicate_loopback_rank outside a loop-back world");
        c->rep_rank = (int)value;
        return LIO_OK;
    }
    return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_set", "unknown key");
}
int lio_debug_get(lio_ctx* c, const char* key, long long* value) {
    if (!c || !key || !value) return LIO_ERR_INVALID_ARG;
    if (std::strcmp(key, "gate_released") == 0) { *value = c->n_gate_released; return LIO_OK; }
    if (std::strcmp(key, "gate_called_off") == 0) { *value = c->n_gate_called_off; return LIO_OK; }
    if (std::strcmp(key, "retired_budget") == 0) { *value = (long long)retired_slot_budget(c); return LIO_OK; }
    if (std::strcmp(key, "voxel_fast_runs") == 0) { *value = c->n_voxel_fast; return LIO_OK; }
    if (std::strcmp(key, "voxel_redone") == 0) { *value = c->n_voxel_redone; return LIO_OK; }
    if (std::strcmp(key, "gate_mail_in_vram") == 0) { *value = c->gate_mail_in_vram ? 1 : 0; return LIO_OK; }
    if (std::strcmp(key, "plans_on_side_stream") == 0) { *value = c->n_plans_on_side_stream; return LIO_OK; }
    if (std::strcmp(key, "deferred_settled") == 0) { *value = c->n_settled; return LIO_OK; }
    if (std::strcmp(key, "gate_reenqueued") == 0) { *value = c->n_gate_reenqueued; return LIO_OK; }
    if (std::strcmp(key, "gate_in_kernel") == 0) { *value = c->n_gate_in_kernel; return LIO_OK; }
    if (std::strcmp(key, "next_scan_armed") == 0) { *value = c->n_next_scan_armed; return LIO_OK; }
    if (std::strcmp(key, "next_scan_released") == 0) { *value = c->n_next_scan_released; return LIO_OK; }
    if (std::strcmp(key, "assisted_pulls") == 0) { *value = c->n_assisted_pulls; return LIO_OK; }
    if (std::strcmp(key, "assist_share_permille") == 0) { *value = (long long)(c->assist_frac * 1000.0 + 0.5); return LIO_OK; }
    if (std::strcmp(key, "last_pack_front") == 0) { *value = c->last_pack_front; return LIO_OK; }
    if (std::strcmp(key, "pack_helper_late") == 0) { *value = c->n_helper_late; return LIO_OK; }
    if (std::strcmp(key, "last_pack_ns") == 0) { *value = (long long)(c->last_pack_us * 1e3); return LIO_OK; }
    if (std::strcmp(key, "scan_pulled") == 0) { *value = c->n_scan_pulled; return LIO_OK; }
    if (std::strcmp(key, "rows_direct_dma") == 0) { *value = c->n_rows_direct_dma; return LIO_OK; }
    return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_get", "unknown key");
}
int lio_debug_stamps(lio_ctx* c, unsigned long long out8[8]) {
    if (!c || !out8) return LIO_ERR_INVALID_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = c->h_out->dbg[i];
    return LIO_OK;
}
int lio_census_enable(lio_ctx* c, int on) { if (!c) return LIO_ERR_INVALID_ARG; enter(c); c->census = on != 0; return LIO_OK; }
int lio_profile_reset(lio_ctx* c) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    prof_collect(c);
    for (int w = 0; w < LIO_PROF_COUNT; ++w) { c->prof_ms[w] = 0.0; c->prof_n[w] = 0; }
    return LIO_OK;
}
int lio_profile_get(lio_ctx* c, int which, double* total_ms, long long* launches) {
    if (!c || which < 0 || which >= LIO_PROF_COUNT) return LIO_ERR_INVALID_ARG;
    enter(c);
    prof_collect(c);
    if (total_ms) *total_ms = c->prof_ms[which];
    if (launches) *launches = c->prof_n[which];
    return LIO_OK;
}

// ---- host-side ESKF -----------------------------------------------------------------------------
struct lio_eskf {
    lio::esekf kf;
};

lio_eskf* lio_eskf_create(void) { return new lio_eskf(); }
void lio_eskf_destroy(lio_eskf* kf) { delete kf; }
int lio_eskf_init_dyn_share(lio_eskf* kf, int maximum_iter, const double limit23[23]) {
    if (!kf || maximum_iter < 0 || !limit23) return LIO_ERR_INVALID_ARG;
    kf->kf.init_dyn_share(nullptr, maximum_iter, limit23);
    return LIO_OK;
}
int lio_eskf_change_x(lio_eskf* kf, const double x26[26]) {
    if (!kf || !x26) return LIO_ERR_INVALID_ARG;
    kf->kf.change_x(lio::state_ikfom::from_array(x26));
    return LIO_OK;
}
int lio_eskf_change_P(lio_eskf* kf, const double P[23 * 23]) {
    if (!kf || !P) return LIO_ERR_INVALID_ARG;
    kf->kf.change_P(P);
    return LIO_OK;
}
int lio_eskf_get_x(lio_eskf* kf, double x26[26]) {
    if (!kf || !x26) return LIO_ERR_INVALID_ARG;
    kf->kf.get_x().to_array(x26);
    return LIO_OK;
}
int lio_eskf_get_P(lio_eskf* kf, double P[23 * 23]) {
    if (!kf || !P) return LIO_ERR_INVALID_ARG;
    std::memcpy(P, kf->kf.get_P(), sizeof(double) * 23 * 23);
    return LIO_OK;
}
int lio_eskf_update_iterated_dyn_share_modified(lio_eskf* kf, lio_ctx* ctx, double R, int extrinsic_est_en,
                                                lio_update_log* log) {
    if (!kf || !ctx || !(R > 0.0)) return LIO_ERR_INVALID_ARG;
    int status = LIO_OK;
    // (the converge flag of every call -- recorded by lio_measure_begin -- predicts the NEXT update's passes)
    if (!ctx->pattern_cur.empty()) { ctx->pattern_prev.swap(ctx->pattern_cur); ctx->pattern_cur.clear(); }
    // the reference's h_share_model, bound to the GPU path (laserMapping.cpp:1876 <-> lio_measure), in two halves so
    // that the filter's result-independent algebra runs while the GPU works on the pass
    kf->kf.set_measurement_model_async(
        [&](lio::state_ikfom& s, const lio::dyn_share_datastruct& d) {
            if (status != LIO_OK) return;
            double pose[14];
            s.pose14(pose);
            // (lio_measure_begin also puts the passes after this one onto the stream, each behind a gate, as the previous update ran them)
            const int rc = lio_measure_begin(ctx, pose, d.converge ? 1 : 0, extrinsic_est_en);
            if (rc != LIO_OK) { status = rc; return; }
        },
        [&](lio::dyn_share_datastruct& d) {
            if (status != LIO_OK) { d.valid = false; d.m = 0; return; }
            lio_out o;
            const int rc = lio_measure_end(ctx, &o);
            if (rc != LIO_OK) { status = rc; d.valid = false; d.m = 0; return; }
            d.valid = o.valid != 0;
            d.m = o.m;
            d.res_mean = o.res_mean;
            std::memcpy(d.HTH, o.HTH, sizeof(d.HTH));
            std::memcpy(d.HTh, o.HTh, sizeof(d.HTh));
            if (d.valid && o.m < lio::DOF) {                    // dense-gain branch needs the rows
                d.h_x.assign((size_t)o.m * 12, 0.0);
                d.h.assign((size_t)o.m, 0.0);
                const int rc2 = lio_measure_rows(ctx, d.h_x.data(), d.h.data(), o.m, 0);
                if (rc2 != LIO_OK) { status = rc2; d.valid = false; }
            }
        });
    double solve_time = 0.0;
    kf->kf.update_iterated_dyn_share_modified(R, solve_time);
    gate_call_off_this_scan(ctx);                // the update ended earlier than the previous one: nothing of THIS scan stays armed
    ctx->pattern_prev.swap(ctx->pattern_cur);
    ctx->pattern_cur.clear();
    if (log) {
        const lio::update_stats& st = kf->kf.stats();
        log->n_calls = st.n_calls; log->n_search = st.n_search; log->converged = st.converged ? 1 : 0;
        log->last_m = st.last_m; log->last_res_mean = st.last_res_mean;
        log->solve_seconds = st.solve_seconds; log->measure_seconds = st.measure_seconds;
    }
    return status;
}
// esekf with a caller-supplied h_dyn_share (esekfom.hpp:239-258): the callback answers with rows, the products
// h_x^T h_x / h_x^T h are formed here (esekfom.hpp:1790,1810) -- column blocks that stay in the L1 / L2 of one core
int lio_eskf_update_iterated_dyn_share_callback(lio_eskf* kf, double R, lio_h_dyn_share_fn h, void* user, lio_update_log* log) {
    if (!kf || !h || !(R > 0.0)) return LIO_ERR_INVALID_ARG;
    int status = LIO_OK;
    kf->kf.set_measurement_model([&](lio::state_ikfom& s, lio::dyn_share_datastruct& d) {
        double x26[26];
        s.to_array(x26);
        lio_dyn_share ds{};
        ds.converge = d.converge ? 1 : 0;
        ds.valid = 1;
        h(user, x26, &ds);
        d.valid = ds.valid != 0;
        d.m = ds.m;
        d.res_mean = 0.0;
        if (!d.valid) return;
        // what h_share_model guarantees (laserMapping.cpp:1956-1961: valid = false when effct_feat_num < 1) is checked, not trusted
        if (ds.m < 1) { d.valid = false; d.m = 0; return; }
        if (ds.h_x != nullptr && ds.h == nullptr) { status = LIO_ERR_INVALID_ARG; d.valid = false; d.m = 0; return; }
        if (ds.h_x == nullptr) {
            std::memcpy(d.HTH, ds.HTH, sizeof(d.HTH));
            std::memcpy(d.HTh, ds.HTh, sizeof(d.HTh));
        } else {
            lio::hth_from_rows_colmajor(ds.h_x, ds.h, (size_t)ds.m, d.HTH, d.HTh);
        }
        if (ds.m < lio::DOF && ds.h_x != nullptr) {            // dense-gain branch (esekfom.hpp:1720-1750): rows, row-major
            d.h_x.assign((size_t)ds.m * 12, 0.0);
            d.h.assign((size_t)ds.m, 0.0);
            for (int r = 0; r < ds.m; ++r) {
                for (int cidx = 0; cidx < 12; ++cidx) d.h_x[(size_t)r * 12 + cidx] = ds.h_x[(size_t)cidx * ds.m + r];
                d.h[r] = ds.h[r];
            }
        } else if (ds.m < lio::DOF) {
            d.valid = false;                                    // the reduced form cannot feed the dense gain
        }
    });
    double solve_time = 0.0;
    kf->kf.update_iterated_dyn_share_modified(R, solve_time);
    if (log) {
        const lio::update_stats& st = kf->kf.stats();
        log->n_calls = st.n_calls; log->n_search = st.n_search; log->converged = st.converged ? 1 : 0;
        log->last_m = st.last_m; log->last_res_mean = st.last_res_mean;
        log->solve_seconds = st.solve_seconds; log->measure_seconds = st.measure_seconds;
    }
    return status;
}
int lio_eskf_scan_step_device(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                              const void* d_body_f4, int n, double R, int extrinsic_est_en, lio_update_log* log) {
    if (!kf || !ctx) return LIO_ERR_INVALID_ARG;
    int rc = LIO_OK;
    if (x26 != nullptr) rc = lio_eskf_change_x(kf, x26);
    if (rc == LIO_OK && P != nullptr) rc = lio_eskf_change_P(kf, P);
    if (rc == LIO_OK && d_body_f4 != nullptr) rc = lio_scan_set_device(ctx, d_body_f4, n);
    if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, R, extrinsic_est_en, log);
    return rc;
}
int lio_eskf_scan_step_pipelined(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                                 const void* next_host_f4, int n_next, double R, int extrinsic_est_en, lio_update_log* log) {
    if (!kf || !ctx) return LIO_ERR_INVALID_ARG;
    ctx->stamp('P');
    int rc = lio_scan_swap(ctx);                                   // the scan uploaded during the previous step
    ctx->stamp('w');
    if (rc == LIO_OK && next_host_f4 != nullptr) rc = lio_scan_upload_async(ctx, next_host_f4, n_next);   // travels during this one
    if (rc == LIO_OK && x26 != nullptr) rc = lio_eskf_change_x(kf, x26);
    if (rc == LIO_OK && P != nullptr) rc = lio_eskf_change_P(kf, P);
    // (the next call of this function is what follows on the stream: the first search of the scan just handed over may be put
    // there behind this scan's last passes -- arm_predicted)
    ctx->arm_across_scans = next_host_f4 != nullptr && n_next > 0;
    if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, R, extrinsic_est_en, log);
    ctx->arm_across_scans = false;
    if (rc != LIO_OK) enter(ctx);
    ctx->stamp('U');
    return rc;
}
int lio_state_boxplus(double x26[26], const double dx23[23]) {
    if (!x26 || !dx23) return LIO_ERR_INVALID_ARG;
    lio::state_ikfom s = lio::state_ikfom::from_array(x26);
    s.boxplus(dx23);
    s.to_array(x26);
    return LIO_OK;
}
int lio_eskf_predict(lio_eskf* kf, double dt, const double Q[12 * 12], const double acc[3], const double gyro[3]) {
    if (!kf || !Q || !acc || !gyro) return LIO_ERR_INVALID_ARG;
    lio::input_ikfom in;
    for (int i = 0; i < 3; ++i) { in.acc[i] = acc[i]; in.gyro[i] = gyro[i]; }
    kf->kf.predict(dt, Q, in);
    return LIO_OK;
}
int lio_process_noise_cov(double Q[12 * 12]) {
    if (!Q) return LIO_ERR_INVALID_ARG;
    lio::process_noise_cov(Q);
    return LIO_OK;
}

// ---- host-side IMU front end ----------------------------------------------------------------------
struct lio_imu {
    lio::ImuProcess p;
};
lio_imu* lio_imu_create(void) { return new lio_imu(); }
void lio_imu_destroy(lio_imu* imu) { delete imu; }
int lio_imu_reset(lio_imu* imu) {
    if (!imu) return LIO_ERR_INVALID_ARG;
    imu->p.Reset();
    return LIO_OK;
}
int lio_imu_set_extrinsic(lio_imu* imu, const double T[3], const double R[9]) {
    if (!imu || !T || !R) return LIO_ERR_INVALID_ARG;
    imu->p.set_extrinsic(T, R);
    return LIO_OK;
}
int lio_imu_set_cov(lio_imu* imu, const double gyr_cov[3], const double acc_cov[3], const double b_gyr_cov[3],
                    const double b_acc_cov[3]) {
    if (!imu) return LIO_ERR_INVALID_ARG;
    if (gyr_cov) imu->p.set_gyr_cov(gyr_cov);
    if (acc_cov) imu->p.set_acc_cov(acc_cov);
    if (b_gyr_cov) imu->p.set_gyr_bias_cov(b_gyr_cov);
    if (b_acc_cov) imu->p.set_acc_bias_cov(b_acc_cov);
    return LIO_OK;
}
int lio_imu_process(lio_imu* imu, lio_eskf* kf, lio_ctx* ctx, double lidar_beg_time, double lidar_end_time,
                    const float* raw_pts, int n, int stride_bytes, int time_offset_bytes,
                    const lio_imu_sample* samples, int n_samples, int* undistorted) {
    if (!imu || !kf || n_samples < 0 || (n_samples > 0 && !samples)) return LIO_ERR_INVALID_ARG;
    if (ctx && (n < 0 || (n > 0 && !raw_pts))) return LIO_ERR_INVALID_ARG;
    lio::MeasureGroup meas;
    meas.lidar_beg_time = lidar_beg_time;
    meas.lidar_end_time = lidar_end_time;
    meas.lidar = raw_pts;
    meas.n_points = n;
    meas.stride_bytes = stride_bytes;
    meas.time_offset_bytes = time_offset_bytes;
    meas.imu.resize((size_t)n_samples);
    for (int i = 0; i < n_samples; ++i) {
        meas.imu[i].stamp = samples[i].stamp;
        for (int k = 0; k < 3; ++k) { meas.imu[i].acc[k] = samples[i].acc[k]; meas.imu[i].gyr[k] = samples[i].gyr[k]; }
    }
    return imu->p.Process(meas, kf->kf, ctx, undistorted);
}
int lio_imu_get_poses(lio_imu* imu, lio_pose6d* out, int cap, int* n) {
    if (!imu || !n) return LIO_ERR_INVALID_ARG;
    const std::vector<lio_pose6d>& v = imu->p.poses();
    *n = (int)v.size();
    if (out) {
        if (cap < (int)v.size()) return LIO_ERR_CAPACITY;
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    }
    return LIO_OK;
}
int lio_imu_get_info(lio_imu* imu, lio_imu_info* info) {
    if (!imu || !info) return LIO_ERR_INVALID_ARG;
    info->need_init = imu->p.need_init() ? 1 : 0;
    info->init_iter_num = imu->p.init_iter();
    for (int i = 0; i < 3; ++i) {
        info->mean_acc[i] = imu->p.mean_acc[i]; info->mean_gyr[i] = imu->p.mean_gyr[i];
        info->cov_acc[i] = imu->p.cov_acc[i]; info->cov_gyr[i] = imu->p.cov_gyr[i];
    }
    info->last_lidar_end_time = imu->p.last_lidar_end_time();
    return LIO_OK;
}

int lio_state_boxminus(const double x26[26], const double y26[26], double dx23[23]) {
    if (!x26 || !y26 || !dx23) return LIO_ERR_INVALID_ARG;
    lio::state_ikfom::from_array(x26).boxminus(dx23, lio::state_ikfom::from_array(y26));
    return LIO_OK;
}

}  // extern "C"
