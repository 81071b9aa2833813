// lio_scan.hip -- scan hand-over of liblio_hip.so: lio_scan_set / _set_device, the pipelined upload of the NEXT scan on a copy stream
// (lio_scan_upload_async / lio_scan_swap), and the drivers of the raw-scan preparation: UndistortPcl's
// point compensation (src/IMU_Processing.hpp:334-386) and pcl::VoxelGrid (laserMapping.cpp:2322-2323).  Kernels: lio_scanprep.hip.
#include "lio_ctx.h"

extern "C" {
// ---- scan + measurement ----------------------------------------------------------------------
int scan_reset(lio_ctx* c, int n) {
    // the passes of the scan that ends here (the converge flag of every lio_measure_begin) predict the next scan's: what
    // lio_measure_begin arms ahead of its pose (gate_arm)
    if (!c->pattern_cur.empty()) { c->pattern_prev.swap(c->pattern_cur); c->pattern_cur.clear(); }
    c->nq = n;
    c->scan_pull_failed = false;
    c->pull_check = false;                      // (the check of a pass that pulled the PREVIOUS scan: its records may be gone by now)
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
    c->raw_prefetched = false;
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
    c->raw_prefetched = false;
    enter(c);
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_set_device", "n exceeds max_scan_points");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0) HIPCHK(c, hipMemcpyAsync(c->d_body, d_body_f4, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    return scan_reset(c, n);
}
int lio_scan_size(lio_ctx* c) { return c ? c->nq : 0; }

// ---- pipelined hand-over: the next scan travels while the current one is being processed --------------------
int issue_pending_upload(lio_ctx* c) {
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
    if (c->prep_active) return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_upload_async", "the back scan buffer belongs to a running preparation (lio_scan_prep_end first)");
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
// ---- raw-scan preparation: UndistortPcl's compensation + pcl::VoxelGrid ----
static int ensure_raw_buffers(lio_ctx* c, size_t n) {
    if (!c->d_imu_poses) HIPCHK(c, hipMalloc((void**)&c->d_imu_poses, 512 * 22 * sizeof(double)));
    if (!c->d_vox) HIPCHK(c, hipMalloc((void**)&c->d_vox, 2 * sizeof(VoxelGridDev)));
    if (n <= c->cap_raw) return LIO_OK;
    void* old[] = {c->d_raw, c->d_raw_back, c->d_undist, c->d_leafpts, c->d_raw_keys, c->d_raw_keys2, c->d_leaf_keys, c->d_raw_order,
                   c->d_raw_order2, c->d_leaf_count, c->d_leaf_start, c->d_undist_parts};
    for (void* p : old) if (p) (void)hipFree(p);           // (hipFree waits for the device: nothing in flight reads them)
    c->raw_prefetched = false;
    c->d_raw = c->d_raw_back = c->d_undist = c->d_leafpts = nullptr; c->d_raw_keys = c->d_raw_keys2 = c->d_leaf_keys = nullptr;
    c->d_raw_order = c->d_raw_order2 = c->d_leaf_count = c->d_leaf_start = nullptr; c->cap_raw = 0;
    c->d_undist_parts = nullptr; c->undist_parts_n = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_raw, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_back, n * sizeof(float4)));     // destination of lio_scan_prefetch_raw
    HIPCHK(c, hipMalloc((void**)&c->d_undist, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_leafpts, n * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_keys, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_keys2, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_keys, n * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_order, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_raw_order2, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_count, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_leaf_start, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_undist_parts, ((n + 255) / 256) * 8 * sizeof(uint32_t)));
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
    hipStream_t s = c->prep_active ? c->prep_stream : c->stream;      // (lio_scan_prep_begin: the preparation runs on a stream of its own)
    int rc = ensure_raw_buffers(c, (size_t)(n > 0 ? n : 1));
    if (rc != LIO_OK) return rc;
    c->n_raw = n; c->have_raw = false; c->raw_uploaded = true; c->undist_parts_n = 0;
    c->raw_time_keyed = time_offset_bytes >= 0;
    if (n == 0) return LIO_OK;
    rc = ensure_stage(c, (size_t)n + POSE_STAGE_SLOTS);                        // the IMU poses travel behind the points (lio_scan_compensate)
    if (rc != LIO_OK) return rc;
    if (c->raw_prefetched) {
        // the sweep is already on its way (lio_scan_prefetch_raw, copy stream): take the buffer it went to
        const bool hit = c->raw_prefetch_src == raw_pts && c->raw_prefetch_n == n && time_offset_bytes == 12 && stride_bytes == 16;
        c->raw_prefetched = false;
        if (hit) {
            const hipError_t q = hipEventQuery(c->ev_raw);
            if (q == hipErrorNotReady) HIPCHK(c, hipStreamWaitEvent(s, c->ev_raw, 0));
            else if (q != hipSuccess) return c->fail(LIO_ERR_HIP, "lio_scan_upload_raw (prefetched sweep)", hipGetErrorString(q));
            float4* t = c->d_raw; c->d_raw = c->d_raw_back; c->d_raw_back = t;
            ++c->n_raw_prefetch_hits;
            return LIO_OK;
        }
    }
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
    c->stage_busy = true; c->stage_busy_slots = (size_t)n; c->stage_stream = s;   // waited for by the next writer of h_stage
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
    hipStream_t s = c->prep_active ? c->prep_stream : c->stream;
    const int n = c->n_raw;
    c->have_raw = true;
    if (n == 0) return LIO_OK;
    if (n_poses == 0) {
        HIPCHK(c, hipMemcpyAsync(c->d_undist, c->d_raw, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, s));
        c->undist_sorted = true;                                               // no sort, no compensation
        c->raw_time_keyed = false;
        c->undist_parts_n = 0;
        return LIO_OK;
    }
    // the poses go through the tail of the pinned staging buffer (bytes the point copy does not read)
    static_assert(sizeof(lio_pose6d) == 22 * sizeof(double), "Pose6D layout");
    // (a copy in flight reads the tail too: not the raw upload's.  ONE event covers the buffer: a copy recorded on the other stream
    // -- the main stream's upload of host points beside a preparation -- would no longer be covered once the event is re-recorded here)
    if (c->stage_busy && (c->stage_busy_slots > c->h_stage_cap - POSE_STAGE_SLOTS || c->stage_stream != s)) {
        HIPCHK(c, hipEventSynchronize(c->ev_stage));
        c->stage_busy = false;
    }
    lio_pose6d* h_poses = reinterpret_cast<lio_pose6d*>(c->h_stage + (c->h_stage_cap - POSE_STAGE_SLOTS));
    std::memcpy(h_poses, imu_poses, (size_t)n_poses * sizeof(lio_pose6d));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_poses, h_poses, (size_t)n_poses * sizeof(lio_pose6d), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->ev_stage, s));
    c->stage_busy = true; c->stage_busy_slots = c->h_stage_cap; c->stage_stream = s;
    // The compensation of a point depends on nothing but the point: it runs in input order.  The order
    // sort(pcl_out.points.begin(), pcl_out.points.end(), time_list) (:251) establishes is produced where it is
    // consumed -- as the minor key of the voxel filter's sort, or on demand by lio_scan_get_cloud (stable: ties keep
    // the input order).
    Pose14 end;
    for (int i = 0; i < 14; ++i) end.v[i] = end_state14[i];
    // (the kernel also leaves the extremes of every workgroup's compensated points and whether the sweep is in time order: the voxel
    // filter's first pass over the cloud, getMinMax3D, is then a fold of ~500 records inside its key kernel)
    HIPCHK(c, launch_undistort(c->d_raw, n, c->d_imu_poses, n_poses, end, c->d_undist, s, c->prep_fused ? c->d_undist_parts : nullptr));
    c->undist_parts_n = c->prep_fused ? (uint32_t)((n + 255) / 256) : 0u;
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
    c->undist_parts_n = 0;                     // (their time-order flag describes the cloud as it was)
    return LIO_OK;
}

// downSizeFilterSurf.setInputCloud(feats_undistort); downSizeFilterSurf.filter(*feats_down_body)
// (src/laserMapping.cpp:2322-2323): pcl::VoxelGrid on the cloud left by lio_scan_undistort; the centroids
// become the current scan (as lio_scan_set would).
//
// One pass over the stream, one read-back at the end: getMinMax3D, min_b_/divb_mul_, the leaf index of every
// point, a stable sort by leaf, the run heads and the centroids all stay on the device (VoxelGridDev).
// A sweep between the single-launch sort's range and the group sort's (16 k ... 256 k points: every real one) whose points
// already come in time order -- most drivers deliver them so -- needs no time key: a STABLE sort by the leaf index alone is
// the sort by (leaf, time, input order).  That key has ~22 bits instead of 64: three stable 8-bit splits (lio_gsort.hip, LSD
// form) instead of the library's ten merge launches.  Whether THIS sweep is in order, and how wide its leaf index is, only
// the device knows (minmax_kernel checks the order while it reads the points): the host PREDICTS both from the previous sweep,
// enqueues the whole filter without waiting, and finds the verdict in the read-back at the end; a wrong prediction (the first
// out-of-order sweep of a driver, a sweep whose extent needs another digit) is followed by the filter on the general path.
static void voxel_plan(lio_ctx* c, VoxelRun& r, float leaf_size) {
    r.n = c->n_raw; r.leaf = leaf_size;
    r.time_minor = c->raw_time_keyed && !c->undist_sorted;     // the cloud is not KNOWN to be in time order
    r.may_compact = !c->dbg_library_sort && c->dbg_voxel_fast && (size_t)r.n > SORT_MAX_ITEMS && (uint32_t)r.n <= GSORT_MAX;
    r.compact = r.may_compact && !(r.time_minor && c->vox_saw_unsorted);
    r.max_leaves = (uint32_t)r.n < (uint32_t)c->cap_q + 1u ? (uint32_t)r.n : (uint32_t)c->cap_q + 1u;
}
// the stamped scratch of the single-launch scans / the LSD sort: the context's own, or the preparation stream's (lio_scan_prep_begin:
// the map update may run its own stamped launches on the main stream at the same time)
static uint32_t voxel_compact_stamp(lio_ctx* c, bool prep, hipStream_t s, unsigned long long** agg) {
    if (!prep) { *agg = c->d_cagg; return next_compact_stamp(c); }
    *agg = c->d_cagg_prep;
    if (++c->compact_seq_prep == 0u) {
        (void)hipMemsetAsync(c->d_cagg_prep, 0, COMPACT_AGG_WORDS * sizeof(unsigned long long), s);
        ++c->compact_seq_prep;
    }
    return c->compact_seq_prep;
}
// one attempt of the filter on stream s, centroids into out_body; nothing waits
static int voxel_enqueue(lio_ctx* c, VoxelRun& r, hipStream_t s, bool prep, float4* out_body) {
    const int n = r.n;
    VoxelGridDev* vox = c->d_vox + c->vox_cur;
    VoxelGridDev* vox_next = c->d_vox + (c->vox_cur ^ 1);
    r.vox = vox;
    if (!c->vox_clean[c->vox_cur]) HIPCHK(c, hipMemsetAsync(vox, 0, sizeof(VoxelGridDev), s));
    c->vox_clean[c->vox_cur] = false;
    c->vox_clean[c->vox_cur ^ 1] = false;                           // (until the centroid launch below is known to be enqueued)
    // inverse_leaf_size_ = 1 / leaf_size_; the order of the points is checked whenever it matters for the next prediction
    const bool folded = r.compact && c->undist_parts_n != 0u && c->undist_parts_n == (uint32_t)((n + 255) / 256);   // (the compensation kernel left the extremes)
    if (!folded) HIPCHK(c, launch_minmax_params(c->d_undist, n, vox, 1.0f / r.leaf, r.may_compact && r.time_minor, s));
    unsigned long long* agg = nullptr;
    if (r.compact) {
        r.passes = c->vox_passes_guess;
        ++c->n_voxel_fast;
        unsigned long long* gagg = prep ? c->d_gagg_prep : c->d_gagg;
        uint32_t& gseq = prep ? c->gsort_seq_prep : c->gsort_seq;
        if (gseq + (uint32_t)r.passes < gseq || gseq + (uint32_t)r.passes == 0xffffffffu) {   // (stamp wrap-around: as run_sort)
            HIPCHK(c, hipMemsetAsync(gagg, 0, GSORT_AGG_WORDS * sizeof(unsigned long long), s));
            gseq = 0u;
        }
        HIPCHK(c, launch_voxel_index_compact(c->d_undist, n, vox, c->d_raw_keys, s, folded ? c->d_undist_parts : nullptr, c->undist_parts_n, 1.0f / r.leaf,
                                             r.may_compact && r.time_minor));
        GsortLsdArgs g{};
        g.keys = c->d_raw_keys; g.n = (uint32_t)n; g.kb = 8 * r.passes; g.passes = r.passes; g.keys_out = c->d_raw_keys2; g.vals_out = c->d_raw_order2;
        g.part = c->d_leaf_keys; g.part2 = reinterpret_cast<uint64_t*>(c->d_leafpts);  // (scratch of this call: 8 and 16 bytes per point)
        g.hist = &vox->hist[0][0]; g.agg = gagg; g.stamp0 = gseq + 1u; g.err = &vox->summary[3];
        gseq += (uint32_t)r.passes;
        HIPCHK(c, launch_gsort_lsd(g, s));
        const uint32_t stamp = voxel_compact_stamp(c, prep, s, &agg);
        if (c->prep_fused) {                                        // run heads inside the centroid launch
            HIPCHK(c, launch_leaf_centroids(c->d_undist, c->d_raw_order2, c->d_raw_keys2, vox, (uint32_t)n, (uint32_t)c->cap_q, out_body, vox_next, agg, stamp, s));
            c->vox_clean[c->vox_cur ^ 1] = true;
            c->vox_cur ^= 1;
            return LIO_OK;
        }
        HIPCHK(c, launch_run_heads(c->d_raw_keys2, (uint32_t)n, 0, c->d_leaf_start, &vox->n_leaves, agg, stamp, &vox->summary[3], s));
    } else {
        HIPCHK(c, launch_voxel_index(c->d_undist, n, vox, r.time_minor, c->d_raw_keys, c->d_raw_order, s));
        HIPCHK(c, sort_pairs_any(c, c->d_raw_keys, c->d_raw_keys2, c->d_raw_order, c->d_raw_order2, (size_t)n, s, &vox->summary[3]));
        if (compact_small(c, (size_t)n)) {
            const uint32_t stamp = voxel_compact_stamp(c, prep, s, &agg);
            HIPCHK(c, launch_run_heads(c->d_raw_keys2, (uint32_t)n, 32, c->d_leaf_start, &vox->n_leaves, agg, stamp, &vox->summary[3], s));
        } else {
            HIPCHK(c, select_run_heads_u64(c->d_temp, c->temp_bytes, c->d_raw_keys2, (size_t)n, c->d_leaf_start, &vox->n_leaves, s));
        }
    }
    HIPCHK(c, launch_centroids(c->d_undist, c->d_raw_order2, c->d_raw_keys2, c->d_leaf_start, vox, (uint32_t)n, r.max_leaves,
                               (uint32_t)c->cap_q, out_body, s, r.compact, vox_next));
    c->vox_clean[c->vox_cur ^ 1] = true;                            // (cleared by the centroid kernel)
    c->vox_cur ^= 1;
    static_assert(offsetof(VoxelGridDev, probe) == offsetof(VoxelGridDev, summary) + 4 * sizeof(uint32_t), "summary and probe are read together");
    return LIO_OK;
}
// the verdict of an attempt: *redo = the prediction was wrong, r is set up for the next attempt
static int voxel_judge(lio_ctx* c, VoxelRun& r, const uint32_t summary[6], int attempt, bool* redo) {
    *redo = false;
    if (summary[3] != 0u) return c->fail(LIO_ERR_HIP, "lio_scan_voxel_filter", "a bounded wait of a sort / compaction kernel ran out");
    if (r.may_compact && r.time_minor) c->vox_saw_unsorted = summary[4] != 0u;            // what the next sweep is taken for
    if (!r.compact || summary[2] != 0u) return LIO_OK;             // (an empty / overflowing grid: nothing was sorted, nothing to redo)
    const int need = ((int)summary[5] + 7) / 8;
    const bool wrong = (r.time_minor && summary[4] != 0u) || need > r.passes;
    if (need >= 1 && need <= 4) c->vox_passes_guess = need;
    if (!wrong) return LIO_OK;
    ++c->n_voxel_redone;
    if (attempt >= 2) return c->fail(LIO_ERR_HIP, "lio_scan_voxel_filter", "the sort prediction failed three times on one sweep");
    r.compact = !(r.time_minor && summary[4] != 0u);               // out of order: the general path; too few digits: again with `need`
    *redo = true;
    return LIO_OK;
}
// the filtered cloud in `out_body` (== c->d_body, or the back buffer that is then swapped in) becomes the current scan
static int voxel_adopt(lio_ctx* c, const uint32_t summary[6], float4* out_body, int* n_out) {
    if (summary[2] & VOX_OVERFLOW)
        return c->fail(LIO_ERR_CAPACITY, "lio_scan_voxel_filter", "leaf size too small for the extent of the scan (index overflow, as in PCL)");
    if (out_body != c->d_body) { float4* t = c->d_body; c->d_body = c->d_body_back; c->d_body_back = t; }
    if (summary[2] & VOX_EMPTY) return scan_reset(c, 0);            // no finite point
    const uint32_t n_real = summary[0] - summary[1];                 // the run of dropped (non-finite) points is last
    if ((int)n_real > c->cap_q) {
        c->have_scan = false;                                        // the scan buffer holds a truncated result
        return c->fail(LIO_ERR_CAPACITY, "lio_scan_voxel_filter", "down-sampled scan exceeds max_scan_points");
    }
    if (n_out) *n_out = (int)n_real;
    return scan_reset(c, (int)n_real);
}
// the whole filter on the main stream, attempts and all (out_body: c->d_body, or c->d_body_back at the end of an asynchronous preparation)
static int voxel_filter_sync(lio_ctx* c, VoxelRun& r, float4* out_body, int first_attempt, int* n_out) {
    uint32_t summary[6] = {0, 0, 0, 0, 0, 0};
    for (int attempt = first_attempt; ; ++attempt) {
        int rc = voxel_enqueue(c, r, c->stream, false, out_body);
        if (rc != LIO_OK) return rc;
        HIPCHK(c, read_back(c, summary, r.vox->summary, sizeof(summary)));
        bool redo = false;
        rc = voxel_judge(c, r, summary, attempt, &redo);
        if (rc != LIO_OK) return rc;
        if (!redo) break;
    }
    return voxel_adopt(c, summary, out_body, n_out);
}

int lio_scan_voxel_filter(lio_ctx* c, float leaf_size, int* n_out) {
    if (!c || !(leaf_size > 0.0f)) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (n_out) *n_out = 0;
    if (!c->have_raw) return c->fail(LIO_ERR_NO_SCAN, "lio_scan_voxel_filter", "no raw scan (lio_scan_undistort first)");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->n_raw;
    if (c->prep_active) {
        // asynchronous preparation: the filter is enqueued on the preparation (= copy) stream (its summary goes to a mail box of its own,
        // its centroids to the back scan buffer) and the call returns; lio_scan_prep_end reads the verdict and swaps the scan in.
        // The general path (tiny sweeps, sweeps out of time order) may use the library's workspace, which the map update on the main
        // stream uses too: such a sweep is filtered by lio_scan_prep_end instead.
        if (c->prep_filtered) return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_voxel_filter", "one filter per lio_scan_prep_begin");
        c->prep_filtered = true; c->prep_enqueued = false;
        voxel_plan(c, c->prep_run, leaf_size);
        if (n_out) *n_out = -1;                                     // (known at lio_scan_prep_end)
        if (n == 0 || !c->prep_run.compact) return LIO_OK;
        const int rc = voxel_enqueue(c, c->prep_run, c->prep_stream, true, c->d_body_back);
        if (rc != LIO_OK) return rc;
        HIPCHK(c, launch_publish_words(c->prep_run.vox->summary, 6, c->d_small3, ++c->small3_seq, c->prep_stream));
        c->prep_enqueued = true;
        return LIO_OK;
    }
    if (n == 0) return scan_reset(c, 0);
    VoxelRun r;
    voxel_plan(c, r, leaf_size);
    return voxel_filter_sync(c, r, c->d_body, 0, n_out);
}

// ---- asynchronous scan preparation: scan k + 1 is compensated and filtered BESIDE the map update of scan k -----------------------
// (the node's loop body, laserMapping.cpp:2256-2401: Process -> voxel filter -> update -> map_incremental -> lasermap_fov_segment;
// the last two of scan k and the first two of scan k + 1 depend on nothing of each other but the posterior of scan k.)
//   lio_scan_prefetch_raw(ctx, sweep)      the moment the driver has the sweep: copy engine -> a second raw buffer (copy stream)
//   lio_scan_prep_begin(ctx)               from here on lio_scan_upload_raw / _compensate / _voxel_filter (through lio_imu_process or
//                                          directly) only ENQUEUE, on the copy stream, with their own stamped scratch
//   ... lio_map_incremental / lio_fov_segment of scan k on the main stream, as ever (they read scan k: untouched) ...
//   lio_scan_prep_end(ctx, &n)             the filter's verdict (already there), the filtered cloud becomes the current scan
// Same kernels, same order per stream, same values: the preparation writes the back scan buffer, the map update reads the front one.
static int ensure_prep(lio_ctx* c) {
    // The preparation runs on the context's COPY stream, behind the prefetch of its own sweep: a stream of its own may end up sharing
    // a hardware queue with the main stream (the runtime hands queues out round-robin over ALL streams of the process -- inside
    // bench.py, with the side stream and torch's streams alive, the overlapped loop body ran at 312 us instead of 215 us: nothing
    // overlapped), and the copy stream has nothing else to do between lio_scan_prep_begin and _end (lio_scan_upload_async is excluded).
    if (!c->prep_stream) c->prep_stream = c->copy_stream;
    if (!c->ev_raw) HIPCHK(c, hipEventCreateWithFlags(&c->ev_raw, hipEventDisableTiming));
    if (!c->h_small3) {
        HIPCHK(c, hipHostMalloc(&c->h_small3, 256, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(c->h_small3, 0, 256);
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_small3, c->h_small3, 0));
    }
    if (!c->d_cagg_prep) {
        HIPCHK(c, hipMalloc((void**)&c->d_cagg_prep, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->d_cagg_prep, 0, COMPACT_AGG_WORDS * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc((void**)&c->d_gagg_prep, GSORT_AGG_WORDS * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->d_gagg_prep, 0, GSORT_AGG_WORDS * sizeof(unsigned long long)));
    }
    return LIO_OK;
}
int lio_scan_prefetch_raw(lio_ctx* c, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes) {
    if (!c || n < 0 || (n > 0 && !raw_pts)) return LIO_ERR_INVALID_ARG;
    // A prefetched sweep is recognised by (records, count) alone, and a driver re-uses one buffer for sweeps of one size: a prefetch
    // nobody consumed (lio_imu_process returns before UndistortPcl while the IMU initialises) must not answer for a LATER sweep.  Every
    // call here -- also one that does nothing -- drops what was prefetched before, and so do lio_imu_process when it returns without
    // having uploaded its sweep and lio_scan_set / _set_device / _set_async (a scan that did not come from a raw sweep).  A caller
    // prefetches every sweep or none; a prefetch nobody consumed is simply issued again.
    if (!c->prep_active) c->raw_prefetched = false;
    // a hint: only x, y, z, t records in page-locked memory can be read in place by the copy engine; anything else is uploaded
    // by lio_scan_upload_raw as before
    if (n == 0 || stride_bytes != 16 || time_offset_bytes != 12 || (size_t)n > c->temp_bytes / 16 || c->prep_active || !is_pinned_host(raw_pts)) return LIO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_prep(c);
    if (rc != LIO_OK) return rc;
    if ((size_t)n > c->cap_raw) {
        enter(c);                               // (the buffers are re-allocated: nothing may be armed on them)
        rc = ensure_raw_buffers(c, (size_t)n);
        if (rc != LIO_OK) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_raw_back, raw_pts, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_raw, c->copy_stream));
    c->raw_prefetched = true; c->raw_prefetch_src = raw_pts; c->raw_prefetch_n = n;
    return LIO_OK;
}
int lio_scan_prep_begin(lio_ctx* c) {
    if (!c) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (c->prep_active) return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_prep_begin", "a preparation is already running (lio_scan_prep_end)");
    if (c->upload_pending || c->nq_back >= 0)
        return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_prep_begin", "the back scan buffer holds a scan handed over with lio_scan_upload_async");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = ensure_prep(c);
    if (rc != LIO_OK) return rc;
    c->prep_active = true; c->prep_filtered = false; c->prep_enqueued = false;
    return LIO_OK;
}
static int scan_prep_end_body(lio_ctx* c, int* n_out) {
    if (!c->prep_filtered) {                    // (only the upload / the compensation ran there)
        HIPCHK(c, hipStreamSynchronize(c->prep_stream));
        return LIO_OK;
    }
    c->prep_filtered = false;
    if (!c->have_raw) return c->fail(LIO_ERR_NO_SCAN, "lio_scan_prep_end", "no raw scan (lio_scan_undistort first)");
    VoxelRun& r = c->prep_run;
    if (r.n == 0) { HIPCHK(c, hipStreamSynchronize(c->prep_stream)); return scan_reset(c, 0); }
    if (!c->prep_enqueued) {
        // the general path was predicted: the filter runs here, on the main stream, behind everything of the preparation stream
        HIPCHK(c, hipStreamSynchronize(c->prep_stream));
        return voxel_filter_sync(c, r, c->d_body_back, 0, n_out);
    }
    // the summary of the filter is the last thing the preparation stream does: when its word is there, everything before it is done
    {
        const uint32_t seq = c->small3_seq;
        volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(c->h_small3) + 32;
        for (uint64_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
            if ((spins & 0xfffffu) == 0xfffffu) {
                const hipError_t q = hipStreamQuery(c->prep_stream);
                if (q != hipSuccess && q != hipErrorNotReady) return c->fail(LIO_ERR_HIP, "lio_scan_prep_end", hipGetErrorString(q));
                if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return c->fail(LIO_ERR_HIP, "lio_scan_prep_end", "the preparation stream is idle but its summary never arrived");
            }
            __builtin_ia32_pause();
        }
    }
    uint32_t summary[6];
    std::memcpy(summary, c->h_small3, sizeof(summary));
    bool redo = false;
    int rc = voxel_judge(c, r, summary, 0, &redo);
    if (rc != LIO_OK) return rc;
    if (redo) return voxel_filter_sync(c, r, c->d_body_back, 1, n_out);   // (a wrong prediction: once more, here)
    ++c->n_prep_overlapped;
    return voxel_adopt(c, summary, c->d_body_back, n_out);
}
int lio_scan_prep_end(lio_ctx* c, int* n_out) {
    if (!c) return LIO_ERR_INVALID_ARG;
    if (n_out) *n_out = 0;
    if (!c->prep_active) return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_prep_end", "no preparation is running (lio_scan_prep_begin)");
    enter(c);
    HIPCHK(c, hipSetDevice(c->device));
    c->prep_active = false;
    const int rc = scan_prep_end_body(c, n_out);
    // an aborted preparation may still be writing the raw / key / back-scan buffers on its stream: the caller's next (synchronous)
    // upload or filter re-uses them on the main stream -- nothing of the preparation outlives a failed lio_scan_prep_end
    if (rc != LIO_OK && c->prep_stream) (void)hipStreamSynchronize(c->prep_stream);
    return rc;
}

// which = 0: feats_undistort (time-sorted, compensated); 1: the current scan (feats_down_body)
int lio_scan_get_cloud(lio_ctx* c, int which, float* out_xyz, int cap, int* n) {
    if (!c || (which != 0 && which != 1) || cap < 0 || (cap > 0 && !out_xyz)) return LIO_ERR_INVALID_ARG;
    enter(c);
    if (which == 0 && c->prep_active) return c->fail(LIO_ERR_INVALID_ARG, "lio_scan_get_cloud", "feats_undistort is being prepared (lio_scan_prep_end first)");
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

}  // extern "C"
