// lio_ctx.hip -- the context object of liblio_hip.so: create / destroy, device and host memory, the read-back helpers every other
// translation unit uses (a mapped word polled by the host instead of a stream synchronisation), the front ends of the single-launch
// sorts / compactions, kernel timing (HIP event pairs) and the debug keys.  The struct itself and the map of the files: lio_ctx.h.
#include "lio_ctx.h"

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
// A few counters from the device: through pinned memory (a copy into pageable memory takes the runtime's slow
// staged path, ~3x the latency), then the one synchronisation the caller needs anyway.
// Waiting for the stream without entering the runtime's wait: a one-wavefront kernel at the end of the stream copies
// the counters the host wants (if any) into host-mapped memory and publishes a sequence number there; the host polls
// that word.  hipStreamSynchronize may park the thread on an interrupt (tens of microseconds to wake up) and
// hipStreamQuery costs a runtime call per poll; the mapped word costs a cache miss.  Bounded: after ~1 s of polling
// the runtime's wait takes over (and reports what went wrong, if something did).
hipError_t publish_and_wait(lio_ctx* c, const uint32_t* d_src, int n_words) {
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
hipError_t stream_wait(lio_ctx* c) { return publish_and_wait(c, nullptr, 0); }
// the polling half alone: a kernel already on the stream publishes `seq` itself (cell_apply_kernel's last workgroup)
hipError_t wait_published(lio_ctx* c, uint32_t seq) {
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(c->h_small) + 32;
    for (uint64_t spins = 0; spins < (1ull << 28); ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        if ((spins & 0xfffffu) == 0xfffffu) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipErrorNotReady) {                              // idle (or broken): the word is there by now, or it never will be
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
                return q == hipSuccess ? hipErrorUnknown : q;
            }
        }
    }
    return hipErrorUnknown;
}
// A few counters from the device (bytes: a multiple of 4, at most 128)
hipError_t read_back(lio_ctx* c, void* host_dst, const void* d_src, size_t bytes) {
    const hipError_t e = publish_and_wait(c, reinterpret_cast<const uint32_t*>(d_src), (int)(bytes / 4));
    if (e != hipSuccess) return e;
    std::memcpy(host_dst, c->h_small, bytes);
    return hipSuccess;
}
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
    c->stage_busy = true; c->stage_busy_slots = n; c->stage_stream = c->stream;
    return LIO_OK;
}

// stamp of the next single-launch compaction (lio_compact.hip): never 0 (the scratch starts zeroed), never repeated within
// 2^32 launches; *err lives in d_apply[15]
uint32_t next_compact_stamp(lio_ctx* c) {
    if (++c->compact_seq == 0u) {
        // the 32-bit stamp has gone round: words stamped 2^32 launches ago must not pass for this launch's -- start over from
        // clean scratch (once in 2^32 launches; the stream orders the clear before the launch that follows)
        (void)hipMemsetAsync(c->d_cagg, 0, c->cagg_words * sizeof(unsigned long long), c->stream);
        ++c->compact_seq;
    }
    return c->compact_seq;
}
int check_compact_error(lio_ctx* c, uint32_t flag_word) {
    if (flag_word == 0u) return LIO_OK;
    (void)hipMemsetAsync(c->d_apply + 15, 0, sizeof(uint32_t), c->stream);
    return c->fail(LIO_ERR_HIP, "map maintenance", "a bounded wait of a compaction kernel ran out");
}

// stable sort of (key, value) pairs: the single-launch kernel for the batch sizes of map maintenance and scan
// preparation, rocPRIM above them
hipError_t sort_pairs_any(lio_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                                 size_t n, hipStream_t s, uint32_t* err) {
    (void)err;
    if (!c->dbg_library_sort && n <= SORT_MAX_ITEMS) return launch_sort_pairs_small(keys_in, vals_in, (uint32_t)n, keys_out, vals_out, s);
    return sort_pairs_u64_u32(c->d_temp, c->temp_bytes, keys_in, keys_out, vals_in, vals_out, n, s);
}
bool compact_small(const lio_ctx* c, size_t n) { return !c->dbg_library_sort && n <= COMPACT_MAX_ITEMS; }

// ---- grouping sorts of map maintenance (lio_gsort.hip): which implementation, and its partition histogram ----------------
SortKind sort_kind(const lio_ctx* c, size_t n_bound, int kb) {
    if (c->dbg_library_sort) return SORT_LIBRARY;
    if (n_bound <= GSORT_RANK_MAX) return SORT_RANK;
    if (n_bound <= GSORT_MAX && kb + GSORT_INDEX_BITS <= 64) return SORT_LARGE;
    return SORT_LIBRARY;
}
// the histogram the key-generating kernel of a SORT_LARGE sort has to accumulate (zero on entry); nullptr otherwise
int sort_hist(lio_ctx* c, SortKind k, uint32_t** hist) {
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
int run_sort(lio_ctx* c, SortKind k, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                    size_t n_bound, const uint32_t* n_dev, uint32_t n_dev_add, int kb, const uint32_t* kb_dev) {
    hipStream_t s = c->stream;
    if (n_bound == 0) return LIO_OK;
    if (k == SORT_RANK) {
        HIPCHK(c, launch_sort_pairs_small(keys_in, vals_in, (uint32_t)n_bound, keys_out, vals_out, s, n_dev, n_dev_add, kb_dev != nullptr ? 64 : (kb > 0 ? kb : 64), kb_dev));
    } else if (k == SORT_LARGE) {
        GsortArgs g{};
        g.keys = keys_in; g.vals = vals_in; g.n = (uint32_t)n_bound; g.n_dev = n_dev; g.n_dev_add = n_dev_add; g.kb = kb; g.kb_dev = kb_dev;
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
hipError_t compact_live_any(lio_ctx* c, const float4* src, uint32_t n, float4* dst, uint32_t* n_out) {
    if (c->dbg_library_sort) return select_live_points(c->d_temp, c->temp_bytes, src, dst, n_out, n, c->stream);
    // (tile counts and offsets live in the sort workspace, which is idle here)
    uint32_t* tile_cnt = reinterpret_cast<uint32_t*>(c->d_temp);
    uint32_t* tile_off = tile_cnt + (n / 2048u + 4u);
    if (((size_t)n / 2048u + 4u) * 2u * sizeof(uint32_t) > c->temp_bytes || (size_t)n / 2048u + 2u > COMPACT_MAX_ITEMS) return select_live_points(c->d_temp, c->temp_bytes, src, dst, n_out, n, c->stream);
    return launch_compact_live_large(src, n, dst, n_out, tile_cnt, tile_off, c->d_cagg, next_compact_stamp(c), c->d_apply + 15, c->stream);
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


extern "C" {
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
    { const char* e = std::getenv("LIO_PREP_FUSED"); if (e != nullptr && e[0] == '0') c->prep_fused = false; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_TEST_LIVE_MISCOUNT"); c->test_live_miscount = e != nullptr && e[0] == '1'; }
    { const char* e = std::getenv("LIO_REBUILD_BY_CELLS"); if (e != nullptr && e[0] == '0') c->rebuild_by_cells = false; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_PRE_ENQUEUE"); if (e != nullptr && e[0] == '0') c->dbg_pre_enqueue = 0; }   // (A/B runs of scripts)
    // Environments in which a launch call returns only when its kernel has ended (HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL: debugging
    // aids of the runtime): a pass put on the stream AHEAD of its pose would wait for a word the host can only write after the launch
    // call has returned -- it could only time out.  Nothing is armed there; every pass is launched when its pose is known.
    {
        const char* b = std::getenv("HIP_LAUNCH_BLOCKING");
        const char* k = std::getenv("AMD_SERIALIZE_KERNEL");
        if ((b != nullptr && b[0] != 0 && b[0] != '0') || (k != nullptr && k[0] != 0 && k[0] != '0')) { c->dbg_pre_enqueue = 0; c->dbg_arm_next_scan = 0; }
    }
    { const char* e = std::getenv("LIO_SETTLE_BESIDE"); if (e != nullptr && e[0] == '0') c->dbg_settle_beside = 0; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_KNN_TUNE"); if (e != nullptr && e[0] >= '0' && e[0] <= '9') c->dbg_knn_tune = std::atoi(e); }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_ARM_NEXT_SCAN"); if (e != nullptr && e[0] == '0') c->dbg_arm_next_scan = 0; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_GATE_IN_KERNEL"); if (e != nullptr && e[0] >= '0' && e[0] <= '7' && e[1] == 0) c->gate_in_kernel = e[0] - '0'; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_ASSIST_PULL"); if (e != nullptr && e[0] >= '0' && e[0] <= '9') c->dbg_assist_pull = std::atoi(e); }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_PUBLISH_IN_KERNEL"); if (e != nullptr && e[0] == '0') c->dbg_publish_in_kernel = false; }   // (A/B runs of scripts)
    { const char* e = std::getenv("LIO_FAST_UPDATE"); if (e != nullptr && e[0] == '0') c->dbg_fast_update = false; }   // (A/B runs of scripts)
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
    CREATE_CHK(hipMalloc((void**)&c->d_apply, 64 * sizeof(uint32_t)));     // (words 32..: the BatchDesc of the insert lists, lio_mapadd.h)
    CREATE_CHK(hipMemset(c->d_apply, 0, 64 * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_mkeys, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_mkeys_tmp, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_vals, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_vals_tmp, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_keys, nm * sizeof(uint64_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_counts, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_cell_starts, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_rank, nm * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc((void**)&c->d_level_parts, (size_t)PLACE_WORKGROUPS * TREE_MAX_LEVELS * sizeof(uint32_t)));
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
void dump_host_trace(lio_ctx* c, size_t last) {
    const size_t n = c->host_trace.size(), from = n > last ? n - last : 0;
    for (size_t i = from; i < n && n > 1; ++i)
        std::fprintf(stderr, "lio host trace %c %+9.2f us%s\n", c->host_trace[i].tag,
                     i > from ? (double)(c->host_trace[i].ns - c->host_trace[i - 1].ns) * 1e-3 : 0.0,
                     c->host_trace[i].tag == 'S' ? (c->host_trace[i].polls == 0 ? "  (no poll: the record was there)" : "  (polled)") : "");
    c->host_trace.clear();
}
void lio_destroy(lio_ctx* c) {
    if (c && c->trace_host) dump_host_trace(c, 80);
    if (!c) return;
    (void)hipSetDevice(c->device);
    lio_replicate_finalize(c);
    if (c->gate_n > 0) gate_call_off(c);                       // (a pass armed across a scan boundary may still wait for its pose)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* dev[] = {c->d_pts, c->d_pts_tmp, c->d_mkeys, c->d_mkeys_tmp, c->d_vals, c->d_vals_tmp, c->d_cell_keys,
                   c->d_cell_counts, c->d_cell_starts, c->d_rank, c->d_level_parts, c->d_child, c->d_ltable, c->d_pend, c->d_pkeys, c->d_pkeys2, c->d_gs_part, c->d_gs_part2, c->d_porder, c->d_porder2, c->d_pnewoff, c->d_plan, c->d_ghist, c->d_gagg, c->d_apply, c->d_temp, c->d_counters, c->d_cagg, c->d_cand, c->d_boxes,
                   c->d_body, c->d_body_back, c->d_world, c->d_nbr, c->d_normvec, c->d_plane, c->d_cnt, c->d_sel, c->d_ulist, c->d_useeds, c->d_lunits, c->d_rows, c->d_rows_c,
                   c->d_units, c->d_gunits, c->d_knn_stamps, c->d_removed, c->d_scratch, c->d_qs_world, c->d_qs_nbr, c->d_qs_cnt, c->d_qs_ulist, c->d_qs_useeds,
                   c->d_raw, c->d_raw_back, c->d_undist, c->d_leafpts, c->d_raw_keys, c->d_raw_keys2, c->d_leaf_keys, c->d_raw_order,
                   c->d_raw_order2, c->d_leaf_count, c->d_leaf_start, c->d_undist_parts, c->d_imu_poses, c->d_vox,
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
    if (c->h_small4) (void)hipHostFree(c->h_small4);
    if (c->d_cagg_lists) (void)hipFree(c->d_cagg_lists);
    for (int i = 0; i < 2; ++i) {
        if (c->d_def_add[i]) (void)hipFree(c->d_def_add[i]);
        if (c->d_def_plain[i]) (void)hipFree(c->d_def_plain[i]);
        if (c->d_def_boxes[i]) (void)hipFree(c->d_def_boxes[i]);
    }
    // (prep_stream is the copy stream: destroyed below)
    if (c->ev_raw) (void)hipEventDestroy(c->ev_raw);
    if (c->h_small3) (void)hipHostFree(c->h_small3);
    if (c->d_cagg_prep) (void)hipFree(c->d_cagg_prep);
    if (c->d_gagg_prep) (void)hipFree(c->d_gagg_prep);
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

// page-locked host memory known to the HIP runtime (lio_host_alloc, lio_host_register, hipHostMalloc, hipHostRegister)?
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
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
void* lio_host_alloc(size_t bytes) {
    void* p = nullptr;
    // portable: pinned for every device of the process (one rank per GPU allocates before or after it picks its device)
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
void lio_host_free(void* p) { if (p) (void)hipHostFree(p); }

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
    if (std::strcmp(key, "knn_stamps") == 0) { c->dbg_knn_stamps = value != 0; return LIO_OK; }
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
    if (std::strcmp(key, "assist_pull_stall_us") == 0) { c->dbg_assist_stall_us = (int)value; return LIO_OK; }
    if (std::strcmp(key, "fused_pull") == 0) { c->dbg_fused_pull = value != 0; return LIO_OK; }
    if (std::strcmp(key, "fast_update") == 0) { c->dbg_fast_update = value != 0; return LIO_OK; }   // (round 6; replaces the unused keys "settle_beside_exchange": LIO_SETTLE_BESIDE, and "assist_pull_threads": lio_scan_pack_threads)
    if (std::strcmp(key, "pull_alias_shift") == 0) { c->dbg_pull_alias_shift = (int)value; return LIO_OK; }
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
    if (std::strcmp(key, "settled_beside_exchange") == 0) { *value = c->n_settled_beside_exchange; return LIO_OK; }
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
    if (std::strcmp(key, "stale_pulls") == 0) { *value = c->n_stale_pulls; return LIO_OK; }
    if (std::strcmp(key, "raw_prefetch_hits") == 0) { *value = c->n_raw_prefetch_hits; return LIO_OK; }
    if (std::strcmp(key, "prep_overlapped") == 0) { *value = c->n_prep_overlapped; return LIO_OK; }
    if (std::strcmp(key, "rows_direct_dma") == 0) { *value = c->n_rows_direct_dma; return LIO_OK; }
    if (std::strcmp(key, "fast_updates") == 0) { *value = c->n_fast_updates; return LIO_OK; }
    if (std::strcmp(key, "fast_update_overflows") == 0) { *value = c->n_fast_update_overflows; return LIO_OK; }
    return c->fail(LIO_ERR_INVALID_ARG, "lio_debug_get", "unknown key");
}
int lio_debug_stamps(lio_ctx* c, unsigned long long out8[8]) {
    if (!c || !out8) return LIO_ERR_INVALID_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = c->h_out->dbg[i];
    return LIO_OK;
}
int lio_debug_knn_stamps(lio_ctx* c, unsigned long long* out, long long cap_words, int* n_workgroups) {
    if (!c || !n_workgroups || cap_words < 0 || (cap_words > 0 && !out)) return LIO_ERR_INVALID_ARG;
    enter(c);
    *n_workgroups = c->d_knn_stamps ? c->knn_stamps_wgs : 0;
    const size_t words = (size_t)*n_workgroups * KNN_STAMP_WORDS;
    if (words == 0 || cap_words == 0) return LIO_OK;
    if ((size_t)cap_words < words) return c->fail(LIO_ERR_CAPACITY, "lio_debug_knn_stamps", "cap_words < workgroups * 8");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_wait(c));
    HIPCHK(c, hipMemcpy(out, c->d_knn_stamps, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
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

}  // extern "C"
