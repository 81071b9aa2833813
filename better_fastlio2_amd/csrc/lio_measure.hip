// lio_measure.hip -- fused plane fit + residual + Jacobian + reduction, gfx950 only.
//
// Replaces the per-point body and the two serial loops of h_share_model
// (src/laserMapping.cpp:1888-1940 OMP loop, :1942-1954 compaction, :1967-2002 Jacobian rows) and
// the h_x^T h_x / h_x^T h products of its consumer (esekfom.hpp:1790,1810): the filter needs only
// those 12x12 + 12 numbers, so the m x 12 matrix never leaves the GPU unless asked for.
//
//   residual_kernel<EXT, SEARCH>  one lane per scan point: world transform (fp64 -> fp32),
//       neighbour gate (:1911), esti_plane (fp32 Householder QR, lio_math.h), pd2 and the
//       s > 0.9 test (:1925-1930), the fp64 Jacobian row (:1976-2001); the rows of a 256-point
//       workgroup are staged in LDS and reduced to the unique entries of H^T H, H^T h, the
//       residual sum and the row count in a FIXED order (run-to-run bit-stable, no atomics).
//       Designated workgroups fold the per-workgroup sums in two levels (stamped 8-byte units, the poll is
//       the read: no arrival counters), the last one expands the symmetric matrix and writes the DeviceOut record straight into
//       host-mapped memory, publishing it with a system-scope release of a sequence number: one
//       launch per pass, no separate reduction kernel, no device-to-host copy.
//       A search pass also carries the exact tree walk (lio_tree.h) of the queries the 27-cell kernel could not
//       prove: the first n_walkers workgroups of the grid walk the listed queries (one wavefront each) while the
//       model workgroups already work; only a model workgroup that OWNS such a query (cnt marked CNT_UNPROVEN)
//       waits for the walk -- this replaced a separate 7 us launch for ~40 of 100 000 queries.
#include "lio_tree.h"

namespace lio {

template <bool EXT, int BS = 256> struct RedShape {
    static constexpr int NC = EXT ? 12 : 6;            // non-zero columns of h_x
    static constexpr int NP = NC * (NC + 1) / 2;       // unique entries of H^T H
    static constexpr int COL_H = NC;                   // staged columns: J[0..NC), h, |pd2|, 1
    static constexpr int COL_RES = NC + 1;
    static constexpr int COL_ONE = NC + 2;
    static constexpr int NCOLS = NC + 3;
    static constexpr int NE = NP + NC + 2;             // + H^T h, residual sum, count
    static constexpr int NEP = EXT ? 128 : 32;         // lanes per row segment (>= NE)
    static constexpr int SEG = BS / NEP;               // row segments per workgroup
    static constexpr int RS = BS / SEG;                // rows per segment
};

// entry e -> the two staged columns whose product is summed
template <bool EXT>
__device__ __forceinline__ void entry_columns(int e, int& ca, int& cb) {
    typedef RedShape<EXT> S;
    if (e < S::NP) {
        // row-major upper triangle: (0,0),(0,1)...(0,NC-1),(1,1)...
        int a = 0, rem = e, len = S::NC;
        while (rem >= len) { rem -= len; --len; ++a; }
        ca = a; cb = a + rem;
    } else if (e < S::NP + S::NC) {
        ca = e - S::NP; cb = S::COL_H;
    } else if (e == S::NP + S::NC) {
        ca = S::COL_RES; cb = S::COL_ONE;
    } else {
        ca = S::COL_ONE; cb = S::COL_ONE;
    }
}

template <bool EXT, int BS> struct ModelSmem {
    typedef RedShape<EXT, BS> S;
    double rows[BS][S::NCOLS];
    double part[S::SEG][S::NEP];
    double own[S::NEP];
    double fin[S::NEP];
};

// WALK: a search pass that carries the exact tree walk of the listed queries (walker workgroups in front, the last
// workgroup adds their rows).  Without it (the walk ran as its own launch, or a pass that re-uses neighbours) the kernel
// is the lean one: 62 instead of 104 VGPRs -- at 1 M points (C5) that is 45 us instead of 160.
template <bool EXT, bool SEARCH, bool WALK, int BS>
__global__ __launch_bounds__(BS) void residual_kernel(ResidualArgs a) {
    static_assert(SEARCH || !WALK, "only a search pass walks");
    typedef RedShape<EXT, BS> S;
    // a workgroup is either a walker (TreeScratch, search passes only) or a model workgroup: one LDS allocation
    constexpr size_t SMEM = WALK && sizeof(TreeScratch) > sizeof(ModelSmem<EXT, BS>) ? sizeof(TreeScratch) : sizeof(ModelSmem<EXT, BS>);
    __shared__ __align__(16) unsigned char s_raw[SMEM];
    __shared__ uint32_t s_failed;                  // a bounded wait ran out (reported, never silent)
    if (a.gate != nullptr && a.gate->skip != 0u) return;      // a pre-enqueued pass the host called off (nothing is published)
    const int tid = threadIdx.x;
    // A pass that re-uses neighbours, enqueued ahead of its pose with no gate kernel in front: the first wavefront of every
    // workgroup waits for the host's word (gate_poll, lio_device.h) and hands it to the others through LDS.
    Pose14 pose_in{};                             // (search passes read their pose where they use it: no 28 live scalar registers)
    if (!SEARCH) pose_in = a.pose;
    if (!SEARCH && a.mail != nullptr) {
        __shared__ unsigned long long s_mail[17];
        if (tid < 64) {
            bool go;
            unsigned long long w = gate_poll(a.mail, a.expect, go);
            if (tid == 1 && !go) w = GATE_SKIP;               // (a timed-out wait reads as an order to stand down)
            if (tid < 17) s_mail[tid] = w;
        }
        __syncthreads();
        const unsigned long long w = s_mail[(tid & 63) < 17 ? (tid & 63) : 0];
        if (mail_word(w, 1) != GATE_GO) return;
        pose_in = gate_pose(w);
    } else if (!SEARCH && a.gate != nullptr) {
        pose_in = a.gate->pose;
    }
    auto pose_now = [&]() -> Pose14 {
        if (SEARCH) return a.gate != nullptr ? a.gate->pose : a.pose;
        return pose_in;
    };
    // bounded waits: seconds, not microseconds -- a pathological search (every query inside a slab of deleted points:
    // ~1 ms per walk) must still finish; the bound only turns a hang into a reported error
    constexpr unsigned SPIN_LIMIT = 1u << 22;
    constexpr double POISON = -1.0e300;           // row count of a workgroup whose bounded wait ran out

    // ---- walkers: the exact tree walk of the listed queries -----------------------------------------------------------
    const int n_walkers = WALK ? a.n_walkers : 0;
    // few listed queries (the normal case, FIX_MAX in lio_kernels.h): their walkers also evaluate the model, nobody waits
    const uint32_t n_listed = WALK && n_walkers > 0 ? (*a.n_unresolved & ~TREE_OVERFLOW) : 0u;       // complete: written by the k-NN launch before this one
    const bool fix_mode = WALK && n_walkers > 0 && n_listed <= (uint32_t)FIX_MAX && n_listed <= (uint32_t)n_walkers;
    if (WALK && (int)blockIdx.x < n_walkers) {
        TreeScratch& ts = *reinterpret_cast<TreeScratch*>(s_raw);
        if (!fix_mode) {
            // many listed queries: one walking wavefront per workgroup, entries strided; the OWNERS of the queries wait (below)
            if (tid >= 64) return;
            __builtin_amdgcn_s_setprio(3);         // the walk is on the critical path of the pass: ahead of the model waves
            uint32_t walked = 0;
            for (uint32_t e = blockIdx.x; e < n_listed; e += (uint32_t)n_walkers) {
                const float4 q = a.tree.ulist[e];
                tree_search_one(a.tree, __float_as_uint(q.w), q.x, q.y, q.z, tid, ts, nullptr, a.tree.useeds ? a.tree.useeds + (size_t)e * 8u : nullptr);
                ++walked;
            }
            if (walked != 0 && tid == 0) {
                // results are written THROUGH (agent-scope stores, lio_tree.h): drained = in memory; then the counter the
                // waiters poll (no L2 write-back needed, and buffer_wbl2 costs ~1.7 us)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.walk_done, walked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // Fix mode: entry e = this workgroup's ONE listed query, finished here; nobody waits for a walk and the last model
        // workgroup never orders or evaluates a listed query (before: ~6.5 us of one-workgroup work at the end of every
        // search pass).  Wavefront 0 walks, then evaluates the model on the exact neighbours it holds in registers (at C2
        // ~85 % of the walks do change the unproven result) and publishes the PRODUCTS of the query's row -- its term of
        // every sum -- as a stamped partial in the slot of the query's RANK among the listed queries (by query number: a
        // fixed order whatever order the list was filled in); wavefront 1 computes that rank while the walk runs.  The last
        // model workgroup folds the n_listed slots like any other partials (the poll is the read).  The query's owner
        // contributes a zero row and writes no outputs: one writer per query.
        const uint32_t e_w = blockIdx.x;
        const int wave = tid >> 6, lane = tid & 63;
        if (wave >= 2 || e_w >= n_listed) return;
        __shared__ uint32_t s_rank;
        __shared__ double s_lrow[16];
        __builtin_amdgcn_s_setprio(3);             // the walk is on the critical path of the pass: ahead of the model waves
        const float4 q = a.tree.ulist[e_w];
        const uint32_t qi = __float_as_uint(q.w);
        if (wave == 1) {
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t t = 0; t < (uint32_t)FIX_MAX / 64u; ++t) {
                const uint32_t j = (uint32_t)lane + 64u * t;
                const uint32_t kj = j < n_listed ? __float_as_uint(a.tree.ulist[j].w) : 0xffffffffu;
                rank += (uint32_t)__popcll(__ballot(kj < qi));
            }
            if (lane == 0) s_rank = rank;
            __syncthreads();
            return;
        }
        bool changed = true;
        int nv = 0;
        float4 own = tree_search_one(a.tree, qi, q.x, q.y, q.z, lane, ts, &nv, a.tree.useeds ? a.tree.useeds + (size_t)e_w * 8u : nullptr, &changed);
        if (!changed) {
            // the walk confirmed knn27_kernel's result and stored nothing: Nearest_Points[q] is that launch's (plain loads)
            if (lane < K) own = a.nbr[(size_t)lane * a.nq_stride + qi];
            nv = (int)(a.cnt[qi] & ~CNT_UNPROVEN);
        }
        float px[5], py[5], pz[5];
#pragma unroll
        for (int k = 0; k < K; ++k) { px[k] = __shfl(own.x, k, 64); py[k] = __shfl(own.y, k, 64); pz[k] = __shfl(own.z, k, 64); }
        const float d4 = __shfl(own.w, 4, 64);
        // every lane evaluates the same query (uniform; lane 0 stores)
        const PoseQ sw = pose_unpack(pose_now());
        const float4 fb = a.body[qi];
        bool fsel = false;
        float fnx = 0.f, fny = 0.f, fnz = 0.f, fpd = 0.f, fpd2 = 0.f;
        double jrow[12], fh = 0.0;
#pragma unroll
        for (int c = 0; c < 12; ++c) jrow[c] = 0.0;
        if (nv >= K && !(d4 > 5.0f))                                 // :1911
            fsel = point_residual(px, py, pz, q.x, q.y, q.z, fb.x, fb.y, fb.z, fnx, fny, fnz, fpd2, &fpd);
        if (fsel) jacobian_row<EXT>(sw, fb.x, fb.y, fb.z, fnx, fny, fnz, fpd2, jrow, fh);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < S::NC; ++c) s_lrow[c] = jrow[c];
            s_lrow[S::COL_H] = fh;
            s_lrow[S::COL_RES] = fsel ? (double)__builtin_fabsf(fpd2) : 0.0;
            s_lrow[S::COL_ONE] = fsel ? 1.0 : 0.0;
        }
        __syncthreads();                                            // s_lrow and s_rank (wavefront 1 is long done)
        {
            const unsigned long long st_hi = (unsigned long long)(uint32_t)a.seq << 32;
            unsigned long long* slot = a.lunits + (size_t)s_rank * RED_STRIDE * 2;
            for (int en = lane; en < S::NE; en += 64) {
                int ca, cb;
                entry_columns<EXT>(en, ca, cb);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(s_lrow[ca] * s_lrow[cb]);
                __hip_atomic_store(slot + (size_t)en * 2 + 0, (bits & 0xffffffffull) | st_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + (size_t)en * 2 + 1, (bits >> 32) | st_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the query's outputs, behind the hand-off (nobody inside this launch reads them)
        if (lane == 0) {
            if (fsel) {
                a.plane[qi] = make_float4(fnx, fny, fnz, fpd);
                a.normvec[qi] = make_float4(fnx, fny, fnz, fpd2);
                if (a.rows != nullptr) {
                    double* dst = a.rows + (size_t)qi * 13;
#pragma unroll
                    for (int c = 0; c < 12; ++c) dst[c] = jrow[c];
                    dst[12] = fh;
                }
            }
            a.sel[qi] = fsel ? 1 : 0;
        }
        return;
    }
    ModelSmem<EXT, BS>& sm = *reinterpret_cast<ModelSmem<EXT, BS>*>(s_raw);
    double (&s_rows)[BS][S::NCOLS] = sm.rows;
    double (&s_part)[S::SEG][S::NEP] = sm.part;
    const int bid = (int)blockIdx.x - n_walkers;   // index among the model workgroups
    const bool stamps = (a.debug_fail & 2) != 0;   // lio_debug_set("stamps"): s_memrealtime answers through the scalar memory path,
                                                   // every stamp delays the next lgkmcnt wait -- off unless asked for
    const unsigned long long t_dbg0 = stamps ? wall_clock64() : 0ull;
    unsigned long long t_dbg1 = 0, t_dbg2 = 0, t_dbg3 = 0, t_dbg4 = 0, t_dbg5 = 0, t_dbg6 = 0;
    const int i = bid * BS + tid;
    if (tid == 0) s_failed = 0u;
    bool sel = false;
    double row[12];
    double h = 0.0;
    float pd2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 12; ++c) row[c] = 0.0;

    // Every load of a point is issued up front (one memory round trip instead of a dependent chain -- the kernel is
    // latency-bound at ~1.5 waves per SIMD).  A search pass fits the plane through the five neighbours; the passes
    // after it re-evaluate only the points it accepted (:1916), on the SAME neighbours, so the plane they would re-fit
    // is the one stored here (16 bytes instead of 80, no QR).
    const bool in = i < a.nq;
    bool walked_elsewhere = false;                 // a listed query in fix mode: its walker evaluates it and writes its outputs
    const PoseQ s = pose_unpack(pose_now());
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float wx = 0.f, wy = 0.f, wz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, plane_d = 0.f;
    if (SEARCH) {
        float4 nb[K];
        uint8_t cnt_i = 0;
        float4 wpt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
            b = a.body[i];
#pragma unroll
            for (int k = 0; k < K; ++k) nb[k] = a.nbr[(size_t)k * a.nq_stride + i];
            cnt_i = a.cnt[i];
            wpt = a.world[i];                     // a search pass already holds feats_down_world (the k-NN launch wrote it)
        }
        // Does this workgroup own a query the walkers still owe (marked by the k-NN kernel)?  Then wait for the walk
        // (bounded; a timeout is reported like a reduction timeout), acquire, and read those neighbours again.  Walkers
        // have the lowest workgroup indices, i.e. they are dispatched first.
        if (WALK && n_walkers > 0 && fix_mode) {
            // a listed query belongs to its walker (above): no outputs from here, a zero row in this workgroup's sums
            walked_elsewhere = in && (cnt_i & CNT_UNPROVEN) != 0;
        } else if (WALK && n_walkers > 0) {
            const bool owes = __syncthreads_or((in && (cnt_i & CNT_UNPROVEN) != 0) ? 1 : 0) != 0;
            if (owes) {
                if (tid == 0) {
                    const uint32_t n = *a.n_unresolved & ~TREE_OVERFLOW;
                    unsigned spins = 0;
                    while (__hip_atomic_load(a.walk_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n && ++spins < SPIN_LIMIT)
                        __builtin_amdgcn_s_sleep(2);
                    if (spins >= SPIN_LIMIT) s_failed = 1u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (in && (cnt_i & CNT_UNPROVEN) != 0) {   // the walked queries only; written through by the walkers
#pragma unroll
                    for (int k = 0; k < K; ++k) nb[k] = load_walked_neighbor(a.nbr, (size_t)k * a.nq_stride + i);
                    cnt_i = __hip_atomic_load(a.cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        wx = wpt.x; wy = wpt.y; wz = wpt.z;
        // points_near.size() < 5 ? false : sqdis[4] > 5 ? false : true          :1911
        if (in && ((cnt_i & ~CNT_UNPROVEN) >= K) && !(nb[4].w > 5.0f)) {
            float px[5], py[5], pz[5];
#pragma unroll
            for (int k = 0; k < K; ++k) { px[k] = nb[k].x; py[k] = nb[k].y; pz[k] = nb[k].z; }
            sel = point_residual(px, py, pz, wx, wy, wz, b.x, b.y, b.z, nx, ny, nz, pd2, &plane_d);
            if (sel && !walked_elsewhere) a.plane[i] = make_float4(nx, ny, nz, plane_d);
        }
    } else if (in) {
        b = a.body[i];
        const uint8_t sel_prev = a.sel[i];                             // :1916 (previous flag)
        const float4 pl = a.plane[i];
        body_to_world(s, b.x, b.y, b.z, wx, wy, wz);                   // :1894-1899
        a.world[i] = make_float4(wx, wy, wz, b.w);
        if (sel_prev != 0) {
            nx = pl.x; ny = pl.y; nz = pl.z;
            sel = residual_from_plane(pl.x, pl.y, pl.z, pl.w, wx, wy, wz, b.x, b.y, b.z, pd2);
        }
    }
    if (in) {
        if (sel) {
            if (!walked_elsewhere) a.normvec[i] = make_float4(nx, ny, nz, pd2);   // :1933-1936
            jacobian_row<EXT>(s, b.x, b.y, b.z, nx, ny, nz, pd2, row, h);
        }
        if (!walked_elsewhere) a.sel[i] = sel ? 1 : 0;
        if (a.rows != nullptr && sel && !walked_elsewhere) {
            double* dst = a.rows + (size_t)i * 13;
#pragma unroll
            for (int c = 0; c < 12; ++c) dst[c] = row[c];
            dst[12] = h;
        }
    }

    // ---- stage the row (zeros when not selected) ---------------------------------------------
    if (WALK && walked_elsewhere) {
#pragma unroll
        for (int c = 0; c < 12; ++c) row[c] = 0.0;
        h = 0.0; sel = false;
    }
#pragma unroll
    for (int c = 0; c < S::NC; ++c) s_rows[tid][c] = row[c];
    s_rows[tid][S::COL_H] = h;
    s_rows[tid][S::COL_RES] = sel ? (double)__builtin_fabsf(pd2) : 0.0;     // res_last[i] :1937
    s_rows[tid][S::COL_ONE] = sel ? 1.0 : 0.0;
    __syncthreads();

    // ---- workgroup reduction: lane (seg, e) sums rows [seg*RS, (seg+1)*RS) of entry e ---------
    const int e = tid % S::NEP, seg = tid / S::NEP;
    double acc = 0.0;
    if (e < S::NE) {
        int ca, cb;
        entry_columns<EXT>(e, ca, cb);
        const int r0 = seg * S::RS;
        for (int r = 0; r < S::RS; ++r) acc += s_rows[r0 + r][ca] * s_rows[r0 + r][cb];
    }
    s_part[seg][e] = acc;
    __syncthreads();
    // ---- two-level reduction across workgroups, no arrival counters ---------------------------------------
    // A partial sum is published as two 8-byte units per entry, {low 32 bits | stamp} and {high 32 bits | stamp},
    // stamp = this pass's sequence number: an 8-byte store is single-copy atomic, so whoever reads both units with
    // the stamp of THIS pass has the value -- the poll IS the read, there is no ticket, no fence, no second round
    // trip.  The LAST workgroup (by index) of every group of RED_GROUP folds the group: it spins on the units of
    // the others; the last workgroup of the grid then folds the group sums the same way and writes the record.
    // Workgroups are dispatched in index order and a reducer only ever waits for lower indices, so what it
    // waits for is running or done.  Every sum has a fixed order; the waits are bounded (a timeout is reported).
    typedef unsigned long long u64;
    double (&s_own)[S::NEP] = sm.own;
    const int nblocks = (int)gridDim.x - n_walkers;
    const int grp = bid / RED_GROUP;
    const int ngroups = (nblocks + RED_GROUP - 1) / RED_GROUP;
    const int grp_first = grp * RED_GROUP;
    const int grp_size = (grp_first + RED_GROUP <= nblocks) ? RED_GROUP : nblocks - grp_first;
    const bool grp_reducer = bid == grp_first + grp_size - 1;
    const bool final_reducer = bid == nblocks - 1;
    const u64 stamp_hi = (u64)(uint32_t)a.seq << 32;
    constexpr u64 HI_MASK = 0xffffffff00000000ull;
    if (tid < S::NE) {
        double t = 0.0;
#pragma unroll
        for (int sg = 0; sg < S::SEG; ++sg) t += s_part[sg][tid];
        // a wait of THIS workgroup that ran out (the walk above) poisons its row count: the sum of the counts then
        // comes out negative at the last workgroup, which reports the pass as failed
        if (tid == S::NE - 1 && s_failed != 0u) t = POISON;
        if (grp_reducer) {
            s_own[tid] = t;                                 // its own partial never leaves the workgroup
        } else {
            const u64 bits = (u64)__double_as_longlong(t);
            u64* dst = a.units + ((size_t)bid * RED_STRIDE + tid) * 2;
            __hip_atomic_store(dst + 0, (bits & 0xffffffffull) | stamp_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, (bits >> 32) | stamp_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (stamps) t_dbg1 = wall_clock64();
    if (!grp_reducer) return;

    // poll-and-read `count` stamped partials starting at `src` (stride RED_STRIDE entries), fixed order per lane
    auto fold = [&](const u64* src, int count) -> double {
        constexpr int PER = (32 + S::SEG - 1) / S::SEG;                // entries per lane and batch: 32 partials in flight
        double f = 0.0;
        if (e < S::NE) {
            for (int j0 = seg; j0 < count; j0 += S::SEG * PER) {
                u64 lo[PER], hi[PER];
                unsigned spins = 0;
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int u = 0; u < PER; ++u) {
                        const int j = j0 + u * S::SEG;
                        lo[u] = stamp_hi; hi[u] = stamp_hi;
                        if (j < count) {
                            const u64* p = src + ((size_t)j * RED_STRIDE + e) * 2;
                            lo[u] = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hi[u] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < PER; ++u) ok = ok && (lo[u] & HI_MASK) == stamp_hi && (hi[u] & HI_MASK) == stamp_hi;
                } while (!ok && ++spins < SPIN_LIMIT);
                if (!ok) s_failed = 1u;
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    if (j0 + u * S::SEG < count) f += __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xffffffffull)));
            }
        }
        return f;
    };

    double f = fold(a.units + (size_t)grp_first * RED_STRIDE * 2, grp_size - 1);
    __syncthreads();                               // s_part is reused
    s_part[seg][e] = f;
    __syncthreads();
    if (tid < S::NE) {
        double t = 0.0;
#pragma unroll
        for (int sg = 0; sg < S::SEG; ++sg) t += s_part[sg][tid];
        t += s_own[tid];                                    // workgroups of the group in index order, the reducer's own last
        if (tid == S::NE - 1 && s_failed != 0u) t = POISON;  // (a fold of this group reducer that ran out)
        if (final_reducer) {
            s_own[tid] = t;
        } else {
            const u64 bits = (u64)__double_as_longlong(t);
            u64* dst = a.gunits + ((size_t)grp * RED_STRIDE + tid) * 2;
            __hip_atomic_store(dst + 0, (bits & 0xffffffffull) | stamp_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, (bits >> 32) | stamp_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (stamps) t_dbg2 = wall_clock64();
    if (!final_reducer) return;

    // ---- the last workgroup of the grid: the group sums in index order, its own group's last ---------------
    f = fold(a.gunits, ngroups - 1);
    __syncthreads();
    s_part[seg][e] = f;
    __syncthreads();
    double (&s_fin)[S::NEP] = sm.fin;
    if (tid < S::NE) {
        double t = 0.0;
#pragma unroll
        for (int sg = 0; sg < S::SEG; ++sg) t += s_part[sg][tid];
        s_fin[tid] = t + s_own[tid];
    }
    __syncthreads();
    if (stamps) t_dbg3 = wall_clock64();
    // ---- fix mode: the listed queries' terms, published by their walkers in query order (above), folded like partials
    if (WALK && fix_mode && n_listed > 0u) {
        f = fold(a.lunits, (int)n_listed);
        __syncthreads();
        s_part[seg][e] = f;
        __syncthreads();
        if (stamps) t_dbg4 = wall_clock64();
        if (tid < S::NE) {
            double t = 0.0;
#pragma unroll
            for (int sg = 0; sg < S::SEG; ++sg) t += s_part[sg][tid];
            s_fin[tid] += t;
        }
        __syncthreads();
    }
    if (stamps) t_dbg6 = wall_clock64();
    const bool failed = s_failed != 0u || (a.debug_fail & 1) != 0 || s_fin[S::NE - 1] < -0.5;
    // One wavefront writes the whole record into host-mapped memory, fences at system scope and only
    // then publishes the sequence number the host is polling.
    if (tid < 64) {
        DeviceOut* out = a.out;
        for (int i = tid; i < 144; i += 64) {
            const int r = i / 12, c = i % 12;
            double v = 0.0;
            if (r < S::NC && c < S::NC) {
                const int lo = r < c ? r : c, hi = r < c ? c : r;
                // index of (lo, hi) in the row-major upper triangle
                const int idx = lo * S::NC - (lo * (lo - 1)) / 2 + (hi - lo);
                v = s_fin[idx];
            }
            out->HTH[i] = v;
        }
        if (tid < 12) out->HTh[tid] = tid < S::NC ? s_fin[S::NP + tid] : 0.0;
        if (tid == 62 || tid == 63) {               // the first and the last scan point as this launch sees them (DeviceOut::probe)
            const float4 p = a.body[tid == 62 ? 0 : a.nq - 1];
            float* dst = out->probe + (tid == 62 ? 0 : 4);
            dst[0] = p.x; dst[1] = p.y; dst[2] = p.z; dst[3] = p.w;
        }
        if (tid == 0) {
            out->res_sum = s_fin[S::NP + S::NC];
            out->m = failed ? -1 : (int)(s_fin[S::NP + S::NC + 1] + 0.5);
            if (a.m_dev != nullptr) *a.m_dev = failed ? 0 : (int)(s_fin[S::NP + S::NC + 1] + 0.5);
            out->n_unresolved = SEARCH ? (int)*a.n_unresolved : 0;
            out->n_candidates = SEARCH ? *a.cand_total : 0ull;
            out->dbg[0] = t_dbg0; out->dbg[1] = t_dbg1; out->dbg[2] = t_dbg2; out->dbg[3] = t_dbg3; out->dbg[4] = t_dbg4;
            out->dbg[5] = t_dbg5; out->dbg[6] = t_dbg6; out->dbg[7] = stamps ? wall_clock64() : 0ull;
            // consumed: zero the counters for the next pass
            if (SEARCH) { *a.n_unresolved = 0u; *a.cand_total = 0ull; if (n_walkers > 0) { a.walk_done[0] = 0u; a.walk_done[1] = 0u; } }
        }
        __threadfence_system();
        if (tid == 0) __hip_atomic_store(&out->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int residual_blocks(int nq) { return (nq + 255) / 256; }
// walking workgroups of a search pass: one wavefront each (the stand-alone kernel's sizing); unproven queries are tens
// per 100 k in-map queries and a few per thousand at the map's edge -- entries beyond the walkers are strided.  An idle
// walker exits at once.
int residual_walkers(int nq) {
    int w = nq / 256;
    if (w < 256) w = 256;
    if (w > 2048) w = 2048;
    return w;
}

// 256 lanes per workgroup: 512 measured the same and 1024 slower on MI355X (the kernel is a chain of dependent round trips, not
// throughput-bound); 128 (round 6, A/B on one box, profiles/r06a_residual_128_lanes_ab.log): search pass 14.0-14.3 vs 12.7-13.0 us
// at C2, equal at C3
template <bool EXT, bool SEARCH, bool WALK>
static void launch_residual_v(const ResidualArgs& a, hipStream_t s, KernelTimer t) {
    const dim3 grid(residual_blocks(a.nq) + (WALK ? a.n_walkers : 0)), block(256);
    if (t.start != nullptr) hipExtLaunchKernelGGL((residual_kernel<EXT, SEARCH, WALK, 256>), grid, block, 0, s, t.start, t.stop, 0, a);
    else hipLaunchKernelGGL((residual_kernel<EXT, SEARCH, WALK, 256>), grid, block, 0, s, a);
}
hipError_t launch_residual(const ResidualArgs& a, bool do_search, bool ext, hipStream_t s, KernelTimer t) {
    if (a.nq <= 0) return hipSuccess;
    const bool walk = do_search && a.n_walkers > 0;
    if (ext) {
        if (walk)           launch_residual_v<true, true, true>(a, s, t);
        else if (do_search) launch_residual_v<true, true, false>(a, s, t);
        else                launch_residual_v<true, false, false>(a, s, t);
    } else {
        if (walk)           launch_residual_v<false, true, true>(a, s, t);
        else if (do_search) launch_residual_v<false, true, false>(a, s, t);
        else                launch_residual_v<false, false, false>(a, s, t);
    }
    return hipGetLastError();
}

}  // namespace lio
