// lio_pull.hip -- feats_down_body straight out of the node's host cloud (lio_scan_set_async; the drop-in callback path, DESIGN.md
// section 4.3): page-locked records are pulled over PCIe by the scan's first search itself (fused pull, lio_knn.hip), large clouds with the
// HOST'S help -- the calling thread (and an opt-in second thread, lio_scan_pack_threads) packs the front of the cloud into 16-byte
// records while the search pulls the back raw (assisted pull).
#include "lio_ctx.h"

extern "C" {
// a scan whose pull out of the caller's host records is still owed (lio_scan_set_async) is pulled before anything but its first search
// pass looks at the scan buffer (enter(), lio_ctx.h)
void flush_pull(lio_ctx* c) {
    c->pull_pending = false;
    if (c->pull_n > 0 && hipSetDevice(c->device) == hipSuccess) {
        const hipError_t e = launch_scan_pull(c->pull_alias, c->pull_stride, c->pull_n, c->d_body, c->stream);
        if (e != hipSuccess) {
            // nobody reads enter()'s status: the scan is declared missing, so that the next pass fails loudly (LIO_ERR_NO_SCAN)
            // instead of measuring whatever the scan buffer holds
            c->err = std::string("launch_scan_pull: ") + hipGetErrorString(e);
            c->have_scan = false;
            c->scan_pull_failed = true;
        }
    }
}
void pack_helper_stop(lio_ctx* c) {
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
// feats_down_body handed over WITHOUT waiting for it to arrive (the drop-in callback path: lio::scan_changed right behind the
// node's voxel filter, laserMapping.cpp:2322-2325).  Page-locked records the runtime knows: a kernel pulls x, y, z over PCIe
// itself (launch_scan_pull; no host work at all, the call returns after one launch) -- the records must then stay untouched until
// the first measurement pass of the scan has returned.  Any other memory: packed into the staging buffer before the call
// returns, the copy is not waited for.  Either way the passes that follow are ordered behind the upload on the stream.
int lio_scan_set_async(lio_ctx* c, const float* body_pts, int n, int stride_bytes) {
    if (!c || n < 0 || (n > 0 && !body_pts) || stride_bytes < 12 || (stride_bytes & 3) != 0) return LIO_ERR_INVALID_ARG;
    if (n > c->cap_q) return c->fail(LIO_ERR_CAPACITY, "lio_scan_set_async", "n exceeds max_scan_points");
    c->pull_pending = false;                    // (a scan nobody looked at is simply replaced)
    c->raw_prefetched = false;
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
// ---- assisted pull ---------------------------------------------------------------------------------------------------------
// The first search of a scan handed over as page-locked host records pulls them over PCIe itself; 100 k records of 48 bytes are
// 4.8 MB = 85 us at 57 GB/s, behind which a 23 us search waits.  The host thread that launched the pass would only wait for it:
// instead it packs the FRONT of the cloud into 16-byte records (page-locked staging) and announces its progress in a word the
// search polls; workgroups of the back -- dispatched first -- pull raw records meanwhile.  Share of the front: where packing
// (pack_ns_per_record, measured on every scan: the node's cloud may be warm in L3 or not) and the PCIe bytes of both parts take
// equally long.  Same values by either route: nothing about a result depends on who got to a record first.

int assist_share(lio_ctx* c, int nq) {
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
int assist_post(lio_ctx* c, int front) {
    PackHelper* h = c->pack_helper;
    if (h == nullptr || front < 16384) return 0;
    // the calling thread still has the launches of the pass in front of it (~6 us): its part is the smaller one
    const int mid = (int)(0.45 * front) / 64 * 64;
    h->job.src = c->pull_host; h->job.stride = (size_t)c->pull_stride; h->job.dst = reinterpret_cast<float*>(c->h_pack);
    h->job.lo = mid; h->job.hi = front; h->job.chunk = assist_chunk(c); h->job.stamp = c->pack_stamp; h->job.word = c->h_pack_progress + 1;
    h->job_seq.store(++c->pack_job_id, std::memory_order_release);
    return mid;
}
void assist_pack(lio_ctx* c, int front, int mid) {
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

}  // extern "C"
