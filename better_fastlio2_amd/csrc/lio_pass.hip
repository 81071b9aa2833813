// lio_pass.hip -- the measurement passes of liblio_hip.so (h_share_model, laserMapping.cpp:1876-2004 <-> lio_measure): the launches of
// one pass (k-NN + fused model kernel), passes put on the stream AHEAD of their pose (mail slots in device memory, gates, the first
// search of the next scan across the scan boundary -- DESIGN.md section 4.2), the result record polled in host-mapped memory, the rows
// of the callback path, and the read-out of the per-point globals.  Kernels: lio_knn.hip, lio_measure.hip, lio_compact.hip.
#include "lio_ctx.h"

// k-NN of nq world/body queries into (nbr, cnt); queries_world must hold the world points when body == nullptr, and
// receives them otherwise (the kernel transforms).  walk_now: settle the unproven queries with the stand-alone tree
// kernel; otherwise they stay listed (count in d_counters[0], cnt marked CNT_UNPROVEN) for the measurement kernel of
// the pass, which carries the walk itself.
static TreeArgs tree_args(lio_ctx* c, int nq, int stride, float4* nbr, uint8_t* cnt, float4* ulist, unsigned long long* useeds) {
    TreeArgs ta{};
    ta.map_pts = c->d_pts; ta.child = c->d_child; ta.ltable = c->d_ltable; ta.levels = c->levels; ta.grid = c->grid;
    ta.ulist = ulist; ta.useeds = useeds; ta.n_unres = c->d_counters;
    ta.nq = nq; ta.nq_stride = stride; ta.nbr = nbr; ta.cnt = cnt;
    return ta;
}
int run_knn(lio_ctx* c, const float4* body, float4* queries_world, const Pose14& pose, int nq, int stride,
            float4* nbr, uint8_t* cnt, float4* ulist, unsigned long long* useeds, bool use_hint, bool walk_now, const GateState* gate,
            bool pull_raw, const GateMail* mail, unsigned long long expect, int pack_front, int pack_mid) {
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
        ka.body_raw = static_cast<const char*>(c->pull_alias) + (size_t)c->dbg_pull_alias_shift * (size_t)c->pull_stride; ka.body_raw_stride = c->pull_stride; ka.body_fill = c->d_body;
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
    ka.stamps = nullptr;
    if (c->dbg_knn_stamps && lanes == 8 && ka.walk == 4 && !c->census && ka.pack_src == nullptr) {   // diagnostics: the stamped instantiation
        if (!c->d_knn_stamps) HIPCHK(c, hipMalloc((void**)&c->d_knn_stamps, ((size_t)c->cap_q / 32 + 1) * KNN_STAMP_WORDS * sizeof(unsigned long long)));
        ka.stamps = c->d_knn_stamps;
        c->knn_stamps_wgs = (nq + 31) / 32;
    }
    HIPCHK(c, launch_knn27(ka, lanes, s, prof_timer(c, use_hint ? LIO_PROF_KNN_HINT : LIO_PROF_KNN)));
    if (walk_now) HIPCHK(c, launch_knn_tree(tree_args(c, nq, stride, nbr, cnt, ulist, useeds), s, prof_timer(c, LIO_PROF_FALLBACK)));
    return LIO_OK;
}


extern "C" {
static int gate_arm_next_scan(lio_ctx* c, bool ext);
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
        if (pull_here) {
            c->pull_pending = false;
            c->pull_check = c->pull_host != nullptr; c->pull_check_host = c->pull_host; c->pull_check_stride = c->pull_stride; c->pull_check_n = nq;
        }
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
// off first (enter()), and a wait nobody answers gives up by itself: after 10 ms inside a pass's own first kernel (every workgroup
// of the launch waits there: it must not park the GPU), after two seconds in the one-wavefront gate kernel.
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
void gate_call_off(lio_ctx* c) {
    for (int i = 0; i < c->gate_n; ++i) {
        gate_write(c, c->gate_q[i].mail_seq, GATE_SKIP, nullptr);
        ++c->n_gate_called_off;
    }
    c->gate_n = 0;
}
// the update of the current scan is over: what was armed for IT stands down; the first pass of the next scan stays
void gate_call_off_this_scan(lio_ctx* c) {
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
    if (!c->have_scan && c->scan_pull_failed) return c->fail(LIO_ERR_NO_SCAN, "lio_measure", ("the pull of the scan out of the caller's records failed: " + c->err).c_str());
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
        if (rc != LIO_OK) { c->pass_in_flight = false; c->pull_check = false; return rc; }
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
    if (c->pull_check) {
        // this pass pulled its scan out of the caller's page-locked records: are the points the GPU saw the points that are there?
        c->pull_check = false;
        // (only the scan this pass ran on: scan_reset drops the check with the scan, and a count that differs is never dereferenced)
        const bool mine = c->pull_check_n == c->nq && c->pull_check_n > 0 && c->pull_check_host == c->pull_host;
        const float* first = reinterpret_cast<const float*>(c->pull_check_host);
        const float* last = mine ? reinterpret_cast<const float*>(c->pull_check_host + (size_t)(c->pull_check_n - 1) * (size_t)c->pull_check_stride) : nullptr;
        const bool same = !mine || (std::memcmp(first, d.probe, 12) == 0 && std::memcmp(last, d.probe + 4, 12) == 0);
        if (!same) {
            ++c->n_stale_pulls;
            gate_call_off(c);                           // (passes armed behind this one read the same scan buffer)
            const int rcu = upload_points(c, first, (size_t)c->pull_check_n, c->pull_check_stride, c->d_body);   // the CPU reads the real cloud
            if (rcu != LIO_OK) return rcu;
            const bool hint_now = c->hint_valid;
            c->hint_valid = c->pass_hint;
            const int rc = enqueue_pass(c, c->last_pose, c->pass_searched, c->last_ext, walk_in_its_own_launch(c), nullptr, &c->pass_seq);
            c->hint_valid = hint_now;
            if (rc != LIO_OK) return rc;
            return lio_measure_end_again(c, out);
        }
    }
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

}  // extern "C"
