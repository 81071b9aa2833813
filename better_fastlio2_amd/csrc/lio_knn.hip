// lio_knn.hip -- exact k-NN (k = 5) of a scan against the GPU-resident map, gfx950 only.
//
// Replaces KD_TREE::Nearest_Search -> KD_TREE::Search (include/ikd-Tree/ikd_Tree.cpp:367-397,
// 869-1013): same fp32 distance expression (ikd_Tree.cpp:1374-1378), same visible point set
// (tomb-stoned points are skipped like point_deleted nodes, ikd_Tree.cpp:885), unbounded search
// radius (max_dist = INFINITY, ikd_Tree.h:236), results ascending by squared distance.
//
// Two kernels:
//   knn27_kernel<G>  G lanes (a DPP-addressable sub-group of a 64-lane wavefront) per query.
//                    The lanes look the 27 cells of the 3x3x3 block up in the cell directory (<= 8
//                    hashed level-1 probes, then fixed-position 8-byte records), stage the non-empty
//                    buckets {start,count} in LDS, walk them as one concatenated candidate list
//                    (16-byte loads, consecutive lanes -> consecutive addresses inside a bucket),
//                    keep a per-lane sorted top-5 of (sqdist bits << 32 | map index) keys (carried as
//                    doubles: v_min_f64 / v_max_f64 ladder) and merge the G lists with DPP min all-reduces.  A result is PROVEN when the 5th
//                    distance is below the squared distance from the query to the faces of the
//                    searched block (exact in fp32 because the cell edge is a power of two);
//                    everything else is flagged for
//   knn_tree_kernel  one wavefront per unresolved query (lio_tree.h): the 5x5x5 block first, then a best-first
//                    branch-and-bound over the hashed 64-ary tree (grid coarsened by 4^L), 64 children
//                    probed per round trip, level-0 buckets as leaves.
//
// Ordering of equal squared distances: by map index (the reference's choice among exact ties
// is traversal-order dependent, ikd_Tree.cpp:887).
#include "lio_tree.h"

namespace lio {

// maximum of a 32-bit value over each aligned group of G lanes, result in every lane of the group
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_move_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
template <int G>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
    if (G >= 2) v = max_u32(v, dpp_move_u32<0xB1>(v));
    if (G >= 4) v = max_u32(v, dpp_move_u32<0x4E>(v));
    if (G >= 8) v = max_u32(v, dpp_move_u32<0x141>(v));
    if (G >= 16) v = max_u32(v, dpp_move_u32<0x140>(v));
    if (G >= 32) v = max_u32(v, (uint32_t)__shfl_xor((int)v, 16, 64));
    if (G >= 64) v = max_u32(v, (uint32_t)__shfl_xor((int)v, 32, 64));
    return v;
}

// The (sqdist bits << 32 | map index) key of a candidate, carried as an IEEE double: for sign-clear,
// non-NaN bit patterns the double order IS the unsigned 64-bit order, and gfx950 has full-rate
// v_min_f64 / v_max_f64 where a 64-bit integer min costs a compare and two selects.  The high word is
// clamped to +inf's fp32 pattern (0x7f800000), which as a double exponent is finite, so no key is a NaN;
// distances below 2^-126 make denormal doubles, which the f64 denorm mode of HIP kernels preserves.
__device__ __forceinline__ double key_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double key_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double key_make(uint32_t dist_bits, uint32_t idx) { return __hiloint2double((int)dist_bits, (int)idx); }
__device__ __forceinline__ uint32_t key_dist_bits(double k) { return (uint32_t)__double2hiint(k); }
__device__ __forceinline__ uint32_t key_index(double k) { return (uint32_t)__double2loint(k); }
__device__ __forceinline__ bool key_same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// sorted (ascending) list of five keys in registers; insertion = a 9-instruction min/max ladder
struct Top5 {
    double k0, k1, k2, k3, k4;
    __device__ __forceinline__ static double none() { return key_make(INF_BITS, 0xffffffffu); }
    __device__ __forceinline__ void init() { k0 = k1 = k2 = k3 = k4 = none(); }
    __device__ __forceinline__ void offer(double x) {
        double t;
        t = key_max(k0, x); k0 = key_min(k0, x); x = t;
        t = key_max(k1, x); k1 = key_min(k1, x); x = t;
        t = key_max(k2, x); k2 = key_min(k2, x); x = t;
        t = key_max(k3, x); k3 = key_min(k3, x); x = t;
        k4 = key_min(k4, x);
    }
    __device__ __forceinline__ void pop() { k0 = k1; k1 = k2; k2 = k3; k3 = k4; k4 = none(); }
};

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // the controls used here are full permutations of their rows: every lane receives a value, `old` is never kept
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double group_min_key(double v) {
    if (G >= 2) v = key_min(v, dpp_move_f64<0xB1>(v));       // quad_perm [1,0,3,2]
    if (G >= 4) v = key_min(v, dpp_move_f64<0x4E>(v));       // quad_perm [2,3,0,1]
    if (G >= 8) v = key_min(v, dpp_move_f64<0x141>(v));      // row_half_mirror
    if (G >= 16) v = key_min(v, dpp_move_f64<0x140>(v));     // row_mirror
    if (G >= 32) v = key_min(v, __shfl_xor(v, 16, 64));
    if (G >= 64) v = key_min(v, __shfl_xor(v, 32, 64));
    return v;
}

// candidate of the 3x3x3 search: tomb-stoned points (w != 0) and NaN distances (non-finite query) take
// the +inf pattern, sort after every real candidate and are dropped by the `< INF_BITS` tests
__device__ __forceinline__ double candidate_key_f(float qx, float qy, float qz, float4 p, uint32_t idx) {
    const float d = sqdist3(qx, qy, qz, p.x, p.y, p.z);
    const uint32_t bits = __float_as_uint(d);
    const uint32_t db = p.w != 0.0f ? INF_BITS : (bits < INF_BITS ? bits : INF_BITS);
    return key_make(db, idx);
}

// The second chance of a query whose 3x3x3 result stayed unproven, by the whole wavefront: the 98 shell cells of the 5x5x5 block
// within the 5th distance `top` already holds (the best five of the inner block), two cells per lane one after the other, then
// their buckets one after the other, 64 points per round trip -- the work of block5_search (lio_tree.h, phase A of the tree walk:
// same cells, same candidates, same keys, same proof => the same result) written for FEW REGISTERS instead of few round trips:
// inlined into knn27_kernel the batched form took the kernel from 56 to 88-92 VGPRs (8 -> 5 wavefronts per SIMD) for the sake
// of ~45 of 100 000 queries.  Returns true when the result is proven inside the 5x5x5 block.
__device__ __forceinline__ bool shell_second_chance(const KnnArgs& a, float qx, float qy, float qz, int lane, uint2* s_bk, WaveTop5& top) {
    const float cell = a.grid.cell;
    const int cx = cell_coord(qx, a.grid.inv_cell), cy = cell_coord(qy, a.grid.inv_cell), cz = cell_coord(qz, a.grid.inv_cell);
    const bool inrange = cx >= 2 && cy >= 2 && cz >= 2 && (uint32_t)cx + 2u <= CELL_MAX && (uint32_t)cy + 2u <= CELL_MAX &&
                         (uint32_t)cz + 2u <= CELL_MAX;
    if (!inrange) return false;                    // (wave-uniform: one query)
    int n_ne = 0;
#pragma unroll 1
    for (int sl = 0; sl < 2; ++sl) {
        const int j = lane + 64 * sl;
        const int ox = j / 25 - 2, oy = (j / 5) % 5 - 2, oz = j % 5 - 2;
        const bool inner = ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1;
        const bool want = j < 125 && !inner &&
                          cell_box_sqdist(qx, qy, qz, (uint32_t)(cx + ox), (uint32_t)(cy + oy), (uint32_t)(cz + oz), 0, cell) <= top.tau();
        uint32_t st = 0, ct = 0;
        if (want) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)(cx + ox), (uint32_t)(cy + oy), (uint32_t)(cz + oz), st, ct);
        const bool ne = ct != 0;
        const u64 b = __ballot(ne);
        if (ne) s_bk[n_ne + __popcll(b & ((1ull << lane) - 1ull))] = make_uint2(st, ct);
        n_ne += __popcll(b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
    for (int b = 0; b < n_ne; ++b) {
        const uint2 bk = s_bk[b];
#pragma unroll 1
        for (uint32_t base = 0; base < bk.y; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            u64 key = KEY64_MAX;
            if (i < bk.y) key = candidate_key(qx, qy, qz, a.map_pts[bk.x + i], bk.x + i);
            top.merge(key);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if ((uint32_t)(top.g4 >> 32) >= INF_BITS) return false;
    const float lox = (float)(cx - CELL_OFF - 2) * cell, hix = (float)(cx - CELL_OFF + 3) * cell;
    const float loy = (float)(cy - CELL_OFF - 2) * cell, hiy = (float)(cy - CELL_OFF + 3) * cell;
    const float loz = (float)(cz - CELL_OFF - 2) * cell, hiz = (float)(cz - CELL_OFF + 3) * cell;
    const float gmin = fminf(fminf(qx - lox, hix - qx), fminf(fminf(qy - loy, hiy - qy), fminf(qz - loz, hiz - qz)));
    return top.tau() < gmin * gmin;
}

// ---------------------------------------------------------------------------------------------
// U = 0: the candidate walk of rounds 1-3 (list of {start, count}, one load in flight per lane) -- the form every throughput-bound
//        launch uses (C2, C5: the kernel is VALU-bound there and extra loads in flight bought nothing).
// U >= 1 (G = 8 or 16): the list is turned into PREFIX form {start - candidates before, candidates up to and including} -- a lane
//        finds the bucket of its t-th candidate with one compare per bucket and the map index with one add -- and U candidates per
//        lane are loaded together before any of them is looked at: launches too small to fill the SIMDs (C3's 24 k queries, the
//        node's ~9 k-point scans) are a chain of dependent round trips, one per candidate of a lane; this makes it one per U.
//        Same candidates, same keys, same ladder: bit-identical results (the order in which a lane offers its candidates to its
//        sorted list does not change the list).
// PACK: the first search of an assisted pull (KnnArgs.pack_*) -- an instantiation of its own: compiled into every launch, the few
// live values it adds took the kernel from eight to seven wavefronts per SIMD (+2 us per launch at C2, measured).
// STAMP: the diagnostic instantiation (KnnArgs.stamps): same code, seven wall-clock stamps per workgroup, a full wait for memory in
// front of the stamps that close a memory phase -- never launched unless asked for (lio_debug_set "knn_stamps").
template <int G, bool CENSUS, bool HINT, int U = 0, bool PACK = false, bool STAMP = false>
__global__ __launch_bounds__(256) void knn27_kernel(KnnArgs a) {
    constexpr int WPB = 4;
    unsigned long long tst[KNN_STAMP_WORDS] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    auto stamp = [&](int i, bool drain) {
        if constexpr (STAMP) {
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            tst[i] = wall_clock64();
        }
    };
    stamp(0, false);
    constexpr int GPW = 64 / G;                    // queries per wavefront
    static_assert(U == 0 || ((G == 4 || G == 8 || G == 16) && !CENSUS), "prefix-form walk: 4, 8 or 16 lanes per query, no census");
    __shared__ uint2 s_bucket[WPB][GPW][U > 0 ? 32 : 28];   // LDS-staged non-empty buckets {start, count}

    if (a.gate != nullptr && a.gate->skip != 0u) return;      // a pre-enqueued pass the host called off
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, r = lane % G;
    // (assisted pull: the query blocks are handed out rotated -- the part pulled raw goes first, then the front in the host's order)
    int qb = (int)blockIdx.x;
    if constexpr (PACK) {
        qb += a.pack_rot;
        if (qb >= (int)gridDim.x) qb -= (int)gridDim.x;
    }
    const int q = (qb * WPB + wave) * GPW + g;
    const bool active = q < a.nq;

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (a.body != nullptr) {
        // feats_down_world = pose (x) feats_down_body (laserMapping.cpp:1894-1899) for the WPB * GPW (<= 64) queries of
        // this workgroup, by its first wavefront: the fp64 transform is issued once per workgroup instead of once per
        // G-lane group (a separate 3 us launch before; every lane repeating it cost 2.6 us of this kernel's VALU time)
        __shared__ float s_q[WPB * GPW][3];
        __shared__ uint32_t s_called_off;
        if (wave == 0) {
            // everyone else in the workgroup waits for this wavefront: ahead of the other workgroups' wavefronts on its SIMD
            // (A/B on one box, scripts/dbg/knn_ab.py: -0.3 ... -1.0 us per launch)
            if (!(a.tune & 1)) __builtin_amdgcn_s_setprio(3);
            Pose14 pose_w = a.pose;
            bool go = true;
            if (a.mail != nullptr) {
                // a pass enqueued ahead of its pose, no gate kernel in front: the host's word is read here
                const unsigned long long w = gate_poll(a.mail, a.expect, go);
                pose_w = gate_pose(w);
                if (blockIdx.x == 0) {                     // ... and left for the measurement kernel behind this launch
                    if (lane >= 2 && lane < 16) a.gate_out->pose.v[lane - 2] = go ? __longlong_as_double((long long)w) : 0.0;
                    if (lane == 0) a.gate_out->skip = go ? 0u : 1u;
                }
                if (lane == 0) s_called_off = go ? 0u : 1u;
            } else if (a.gate != nullptr) {
                pose_w = a.gate->pose;
            }
            const int qi = qb * (WPB * GPW) + lane;
            // assisted pull: a workgroup of the front waits (bounded) until the host has packed its records; out of time -> raw
            bool packed_by_host = false;
            if (PACK && go && a.body_raw != nullptr && a.pack_src != nullptr && qb * (WPB * GPW) < a.pack_front) {
                const uint32_t need = (uint32_t)min((qb + 1) * (WPB * GPW), a.pack_front);
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    // (workgroups of [pack_mid, pack_front) listen to the second packing thread's word, lio_pull.hip)
                    const unsigned long long w = __hip_atomic_load(a.pack_progress + (a.pack_mid > 0 && qb * (WPB * GPW) >= a.pack_mid ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((uint32_t)(w >> 32) == a.pack_stamp && (uint32_t)w >= need) { packed_by_host = true; break; }
                    if (wall_clock64() - t0 > 30000ull) break;           // 100 MHz: 300 us -- a host that was pre-empted
                    __builtin_amdgcn_s_sleep(8);
                }
                if (need == (uint32_t)a.pack_front && lane == 0 && a.pack_wait_out != nullptr) {
                    // the last workgroup of the front (the last of the launch): how long did it wait for the host -- or did it give up?
                    unsigned long long dt = wall_clock64() - t0;
                    if (dt > 0xfffffffeull) dt = 0xfffffffeull;
                    __hip_atomic_store(a.pack_wait_out, ((unsigned long long)a.pack_stamp << 32) | (packed_by_host ? dt : 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                // (the records were written before the word that announces them: x86 store order + sfence, lio_pull.hip)
            }
            if (go && lane < WPB * GPW && qi < a.nq) {
                float4 b;
                if (PACK && a.body_raw != nullptr && packed_by_host) {
                    b = a.pack_src[qi];                                  // (16 bytes instead of the whole record; w = 0 already)
                    a.body_fill[qi] = b;
                } else if (a.body_raw != nullptr) {
                    // (host memory through its device alias: one PCIe read per record, thousands of workgroups' worth in flight)
                    const char* rec = a.body_raw + (size_t)qi * (size_t)a.body_raw_stride;
                    if (a.body_raw_aligned) {
                        b = *reinterpret_cast<const float4*>(rec);
                    } else {
                        const float* p = reinterpret_cast<const float*>(rec);
                        b.x = p[0]; b.y = p[1]; b.z = p[2];
                    }
                    b.w = 0.0f;
                    a.body_fill[qi] = b;
                } else {
                    b = a.body[qi];
                }
                const PoseQ s = pose_unpack(pose_w);
                float wx, wy, wz;
                body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
                a.world_out[qi] = make_float4(wx, wy, wz, b.w);
                s_q[lane][0] = wx; s_q[lane][1] = wy; s_q[lane][2] = wz;
            }
            if (!(a.tune & 1)) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        if (a.mail != nullptr && s_called_off != 0u) return;   // the host called the pass off (or never answered): nothing is published
        if (active) { qx = s_q[wave * GPW + g][0]; qy = s_q[wave * GPW + g][1]; qz = s_q[wave * GPW + g][2]; }
    } else if (active) {
        const float4 w = a.world_in[q];            // world-frame query (the caller's)
        qx = w.x; qy = w.y; qz = w.z;
    }
    stamp(1, true);
    const float inv_cell = a.grid.inv_cell;
    // cell_coord (lio_common.h) with the clamp as one v_med3_f32: the same cell for every finite coordinate and for +-inf;
    // a NaN lands in SOME valid cell instead of the middle one, which is all it needs (none of its distances is a number:
    // no candidate counts, the query goes to the tree walk, which does its own arithmetic)
    auto query_cell = [inv_cell](float v) {
        return (int)__builtin_amdgcn_fmed3f(__builtin_floorf(v * inv_cell), -1048576.0f, 1048575.0f) + CELL_OFF;
    };
    const int cx = query_cell(qx), cy = query_cell(qy), cz = query_cell(qz);

    // ---- HINT (second and later searches of a scan): the previous neighbours of this scan point,
    // re-measured from the new query, bound the new 5th distance from above: tau = max_k d(q, nbr_k).
    // Every point that can enter the result then lies in a cell whose box distance is <= tau, so the
    // other cells are neither probed nor streamed (exactness: see DESIGN.md section 3.1).
    // Round 5: the loads of the previous neighbours are ISSUED here and looked at behind the level-1 probes of the directory
    // below -- one memory round trip for both instead of two in a row (the stamped ladder, scripts/knn_stamps.py: the directory
    // rung of a bounded search took 1.84 us against 1.04 us in a first search at 24 k queries).
    uint32_t tau_bits = INF_BITS;
    float glo[3] = {0.f, 0.f, 0.f}, ghi[3] = {0.f, 0.f, 0.f};
    constexpr int HK = (K + G - 1) / G;            // previous neighbours per lane (G < K: lane 0 measures two)
    float4 hint_p[HK];
    uint8_t hint_cnt = 0;
    if (HINT) {
#pragma unroll
        for (int j = 0; j < HK; ++j) hint_p[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            hint_cnt = a.cnt[q];
#pragma unroll
            for (int j = 0; j < HK; ++j)
                if (r + j * G < K) hint_p[j] = a.nbr[(size_t)(r + j * G) * a.nq_stride + q];
        }
    }
    auto hint_bound = [&]() {
        uint32_t db = 0u;
        const bool have = active && (hint_cnt & ~CNT_UNPROVEN) >= K;
        if (active) {
#pragma unroll
            for (int j = 0; j < HK; ++j) {
                if (r + j * G < K) {
                    const float d = sqdist3(qx, qy, qz, hint_p[j].x, hint_p[j].y, hint_p[j].z);
                    const uint32_t b = d == d ? __float_as_uint(d) : INF_BITS;
                    db = b > db ? b : db;
                }
            }
        }
        db = group_max_u32<G>(db);
        tau_bits = have ? (db < INF_BITS ? db : INF_BITS) : INF_BITS;
        // distances from the query to the faces of its own cell (exact differences of exact numbers)
        const float cell = a.grid.cell;
        const float lox = (float)(cx - CELL_OFF) * cell, loy = (float)(cy - CELL_OFF) * cell, loz = (float)(cz - CELL_OFF) * cell;
        glo[0] = qx - lox; glo[1] = qy - loy; glo[2] = qz - loz;
        ghi[0] = (lox + cell) - qx; ghi[1] = (loy + cell) - qy; ghi[2] = (loz + cell) - qz;
    };
    float tau = __uint_as_float(INF_BITS);

    // ---- cell directory look-up.  The 3x3x3 block touches at most 2x2x2 level-1 cells (a level-1 cell = 4x4x4
    // grid cells): lanes 0..7 of the group hash-probe those (the only hashing, the only collisions), every level-1
    // entry names a block of 64 {start, count} records, and the 27 cells are then read at FIXED positions of those
    // blocks -- one 8-byte load per cell, no key compare, no probe sequence.
    constexpr int L1SLOTS = (8 + G - 1) / G;
    // z-columns per lane: all nine in one round when the group has the lanes; with 8 or 4 lanes the first eight columns
    // in full rounds and the ninth (x + 1, y + 1) cell by cell on lanes 0..2 -- a ninth column round would cost every
    // lane three more records to compact for one busy lane
    constexpr int CSLOTS = G >= 9 ? 1 : 8 / G;
    constexpr bool EXTRA = G < 9;
    constexpr int NREC = 3 * CSLOTS + (EXTRA ? 1 : 0);   // cell records per lane
    uint32_t st[NREC], ct[NREC];
    {
        const uint4* l1 = reinterpret_cast<const uint4*>(a.l1table);
        const uint32_t l1mask = a.l1mask;
        const uint32_t ucx = (uint32_t)cx, ucy = (uint32_t)cy, ucz = (uint32_t)cz;     // cell_coord clamps into [0, 2^21)
        const uint32_t b0x = (ucx > 0u ? ucx - 1u : 0u) >> 2, b0y = (ucy > 0u ? ucy - 1u : 0u) >> 2, b0z = (ucz > 0u ? ucz - 1u : 0u) >> 2;
        const uint32_t h1x = (ucx < CELL_MAX ? ucx + 1u : CELL_MAX) >> 2, h1y = (ucy < CELL_MAX ? ucy + 1u : CELL_MAX) >> 2,
                       h1z = (ucz < CELL_MAX ? ucz + 1u : CELL_MAX) >> 2;
        uint32_t blk[L1SLOTS];
        {
            u64 k1[L1SLOTS];
            uint32_t sl1[L1SLOTS];
            uint4 e1[L1SLOTS];
            bool pend[L1SLOTS];
            bool any_pend = false;
#pragma unroll
            for (int s1 = 0; s1 < L1SLOTS; ++s1) {
                const int li = r + s1 * G;
                const uint32_t x1 = b0x + (uint32_t)((li >> 2) & 1), y1 = b0y + (uint32_t)((li >> 1) & 1), z1 = b0z + (uint32_t)(li & 1);
                const bool want = active && li < 8 && x1 <= h1x && y1 <= h1y && z1 <= h1z;
                k1[s1] = pack_key(x1, y1, z1);
                sl1[s1] = hash_cell(x1, y1, z1) & l1mask;
                e1[s1] = l1[want ? sl1[s1] : 0u];                   // unconditional load, discarded when not wanted
            }
            if (HINT) { hint_bound(); tau = __uint_as_float(tau_bits); }     // (the probes are in flight: the previous neighbours arrive beside them)
#pragma unroll
            for (int s1 = 0; s1 < L1SLOTS; ++s1) {
                const int li = r + s1 * G;
                const uint32_t x1 = b0x + (uint32_t)((li >> 2) & 1), y1 = b0y + (uint32_t)((li >> 1) & 1), z1 = b0z + (uint32_t)(li & 1);
                const bool want = active && li < 8 && x1 <= h1x && y1 <= h1y && z1 <= h1z;
                if (!want) e1[s1] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
            }
#pragma unroll
            for (int s1 = 0; s1 < L1SLOTS; ++s1) {
                const u64 ek = ((u64)e1[s1].y << 32) | (u64)e1[s1].x;
                const bool hit = ek == k1[s1];
                blk[s1] = hit ? e1[s1].z : 0xffffffffu;
                pend[s1] = !hit && ek != KEY_EMPTY;
                any_pend = any_pend || pend[s1];
            }
            // collision rounds (load factor <= 0.5, apply_pending); the round bound only matters for a corrupted table
            for (uint32_t rounds = 0; __any(any_pend) && rounds <= l1mask; ++rounds) {
                any_pend = false;
#pragma unroll
                for (int s1 = 0; s1 < L1SLOTS; ++s1) {
                    sl1[s1] = (sl1[s1] + 1u) & l1mask;
                    if (pend[s1]) e1[s1] = l1[sl1[s1]];
                    const u64 ek = ((u64)e1[s1].y << 32) | (u64)e1[s1].x;
                    const bool hit = pend[s1] && ek == k1[s1];
                    blk[s1] = hit ? e1[s1].z : blk[s1];
                    pend[s1] = pend[s1] && !hit && ek != KEY_EMPTY;
                    any_pend = any_pend || pend[s1];
                }
            }
        }
        const int gbase = lane - r;                                  // first lane of this group
        // block number held by lane li of the group (li = which of the up to eight level-1 cells)
        auto block_of = [&](uint32_t li) {
            uint32_t bv = 0xffffffffu;
#pragma unroll
            for (int s1 = 0; s1 < L1SLOTS; ++s1) {
                const uint32_t v = (uint32_t)__shfl((int)blk[s1], gbase + (int)(li % (uint32_t)G), 64);
                if ((int)(li / (uint32_t)G) == s1) bv = v;
            }
            return bv;
        };
        // One lane per z-COLUMN of three cells (9 columns instead of 27 cells): x, y, their level-1 cell, the bounds
        // and the position inside the block are worked out once per column, the column's cells lie in at most two
        // level-1 cells (z below / above a multiple of four) whose block numbers come with two shuffles, and the
        // three records differ in the two low address bits only.
        uint2 ch[NREC];
        bool ok[NREC];
#pragma unroll
        for (int s = 0; s < CSLOTS; ++s) {
            const int j = r + s * G;
            const int ox = (j / 3) - 1, oy = (j % 3) - 1;
            const int nx = cx + ox, ny = cy + oy;
            const bool col = active && j < (EXTRA ? 8 : 9) && (uint32_t)nx <= CELL_MAX && (uint32_t)ny <= CELL_MAX;
            const uint32_t li0 = col ? (((((uint32_t)nx >> 2) - b0x) << 2) | ((((uint32_t)ny >> 2) - b0y) << 1)) : 0u;
            const uint32_t bv_lo = block_of(li0), bv_hi = block_of(li0 + 1u);
            const uint32_t posxy = (((uint32_t)nx & 3u) << 4) | (((uint32_t)ny & 3u) << 2);
            float bxy = 0.0f;
            if (HINT) {
                // box distance of the neighbour cell, same fp32 form as sqdist3 (a lower bound of every
                // computed point distance inside it)
                const float bx = ox < 0 ? glo[0] : (ox > 0 ? ghi[0] : 0.0f);
                const float by = oy < 0 ? glo[1] : (oy > 0 ? ghi[1] : 0.0f);
                bxy = bx * bx + by * by;
            }
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                const int e = 3 * s + z;
                const int nz = cz + z - 1;
                bool live = col && (uint32_t)nz <= CELL_MAX;
                if (HINT) {
                    const float bz = z == 0 ? glo[2] : (z == 2 ? ghi[2] : 0.0f);
                    live = live && (bxy + bz * bz <= tau);
                }
                const uint32_t bv = (live && ((uint32_t)nz >> 2) != b0z) ? bv_hi : bv_lo;
                ok[e] = live && bv != 0xffffffffu;
                ch[e] = a.child[ok[e] ? (size_t)bv * 64u + (posxy | ((uint32_t)nz & 3u)) : 0u];   // unconditional load: the loads overlap
            }
        }
        if (EXTRA) {
            // the ninth column, one cell per lane
            constexpr int e = 3 * CSLOTS;
            const int nx = cx + 1, ny = cy + 1, nz = cz + r - 1;
            bool live = active && r < 3 && (uint32_t)nx <= CELL_MAX && (uint32_t)ny <= CELL_MAX && (uint32_t)nz <= CELL_MAX;
            if (HINT) {
                const float bz = r == 0 ? glo[2] : (r == 2 ? ghi[2] : 0.0f);
                live = live && ((ghi[0] * ghi[0] + ghi[1] * ghi[1]) + bz * bz <= tau);
            }
            const uint32_t li = live ? (((((uint32_t)nx >> 2) - b0x) << 2) | ((((uint32_t)ny >> 2) - b0y) << 1) | (((uint32_t)nz >> 2) - b0z)) : 0u;
            const uint32_t bv = block_of(li);
            ok[e] = live && bv != 0xffffffffu;
            const uint32_t pos = (((uint32_t)nx & 3u) << 4) | (((uint32_t)ny & 3u) << 2) | ((uint32_t)nz & 3u);
            ch[e] = a.child[ok[e] ? (size_t)bv * 64u + pos : 0u];
        }
#pragma unroll
        for (int e = 0; e < NREC; ++e) {
            st[e] = ok[e] ? ch[e].x : 0u;
            ct[e] = ok[e] ? ch[e].y : 0u;
        }
    }

    stamp(2, true);
    // ---- compact the non-empty buckets of each group into LDS ---------------------------------
    int n_ne = 0;
#pragma unroll
    for (int s = 0; s < NREC; ++s) {
        const bool ne = ct[s] != 0;
        const u64 b = __ballot(ne);
        const u64 gb = (G == 64) ? b : ((b >> (g * G)) & ((1ull << (G & 63)) - 1ull));
        const int pos = n_ne + __popcll(gb & ((1ull << r) - 1ull));
        if (ne) s_bucket[wave][g][pos] = make_uint2(st[s], ct[s]);
        n_ne += __popcll(gb);
    }
    Top5 top; top.init();
    uint32_t csum = 0;
    if constexpr (U == 0) {
        // a sentinel bucket that no candidate number reaches ends the walk below
        if (r == 0) s_bucket[wave][g][n_ne] = make_uint2(0u, 0xffffffffu);
        // LDS is in-order per wavefront; the fence keeps the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- stream the candidates: the buckets of a query are walked as ONE concatenated list and lane r
        // takes its candidates r, r+G, r+2G, ... -- every iteration keeps all G lanes busy whatever the
        // bucket sizes are (points of a bucket are contiguous: consecutive lanes, consecutive addresses)
        if (CENSUS) for (int i = 0; i < n_ne; ++i) csum += s_bucket[wave][g][i].y;
        {
            uint32_t b = 0, base = 0, t = (uint32_t)r;
            uint2 bk = s_bucket[wave][g][0];
            while (t - base >= bk.y) { base += bk.y; ++b; bk = s_bucket[wave][g][b]; }
            while ((int)b < n_ne) {
                const uint32_t idx = bk.x + (t - base);
                top.offer(candidate_key_f(qx, qy, qz, a.map_pts[idx], idx));
                t += (uint32_t)G;
                while (t - base >= bk.y) { base += bk.y; ++b; bk = s_bucket[wave][g][b]; }
            }
        }
    } else {
        // ---- prefix form of the list: lane r owns entries [r * EPL, (r + 1) * EPL) (32 slots >= the 27 + 1 a group can fill)
        constexpr int EPL = 32 / G;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint2* lst = s_bucket[wave][g];
        uint2 ent[EPL];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            ent[k] = lst[r * EPL + k];                                   // (slots at and beyond n_ne hold leftovers: masked)
            if (r * EPL + k >= n_ne) ent[k].y = 0u;
            run += ent[k].y;
        }
        uint32_t inc = run;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, o, G);
            if (r >= o) inc += up;
        }
        const uint32_t total = (uint32_t)__shfl((int)inc, lane - r + G - 1, 64);   // candidates of this query
        uint32_t before = inc - run;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // every lane has read its entries
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            if (r * EPL + k < n_ne) lst[r * EPL + k] = make_uint2(ent[k].x - before, before + ent[k].y);
            before += ent[k].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- the walk: candidate t of the query lies in the first bucket whose cumulative count exceeds t, at map index
        // (start - candidates before) + t; lane r takes t = r, r + G, ...; U of them are fetched together
        stamp(3, true);
        uint32_t t = (uint32_t)r;
        const uint2* cur = lst;
        uint2 bk = *cur;                                                 // (total == 0: never used)
        while (t < total) {
            uint32_t idx[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t tu = t + (uint32_t)(u * G);
                ok[u] = tu < total;
                idx[u] = u > 0 ? idx[0] : 0u;                            // (beyond the list: a load nobody looks at)
                if (ok[u]) {
                    while (tu >= bk.y) { ++cur; bk = *cur; }              // ends inside the list: bucket n_ne - 1 reaches `total`
                    idx[u] = bk.x + tu;
                }
            }
            float4 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = a.map_pts[idx[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double key = candidate_key_f(qx, qy, qz, p[u], idx[u]);
                top.offer(ok[u] ? key : Top5::none());                   // offering "none" changes nothing
            }
            t += (uint32_t)(U * G);
        }
    }

    stamp(4, true);
    // ---- merge the G sorted lists: five rounds of group-min + pop ------------------------------
    // lane r keeps result r (and result r + G when G < 5)
    double mine0 = Top5::none(), mine1 = Top5::none();
    int nvalid = 0;
    uint32_t d5bits = INF_BITS;
#pragma unroll
    for (int rd = 0; rd < K; ++rd) {
        const double m = group_min_key<G>(top.k0);
        if (key_same(top.k0, m)) top.pop();                // real keys are unique (distinct map indices);
                                                           // popping the "none" key leaves "none"
        if (r == rd % G) { if (rd < G) mine0 = m; else mine1 = m; }
        nvalid += (key_dist_bits(m) < INF_BITS) ? 1 : 0;
        if (rd == K - 1) d5bits = key_dist_bits(m);
    }

    // ---- proof: every point outside the searched block is farther than the 5th neighbour ------
    bool proven = false;
    if (active && r == 0) {
        const float cell = a.grid.cell;
        if (nvalid == K && cx >= 1 && cy >= 1 && cz >= 1 && (uint32_t)cx < CELL_MAX &&
            (uint32_t)cy < CELL_MAX && (uint32_t)cz < CELL_MAX) {
            // block faces (exact: integer * power of two)
            const float lox = (float)(cx - CELL_OFF - 1) * cell, hix = (float)(cx - CELL_OFF + 2) * cell;
            const float loy = (float)(cy - CELL_OFF - 1) * cell, hiy = (float)(cy - CELL_OFF + 2) * cell;
            const float loz = (float)(cz - CELL_OFF - 1) * cell, hiz = (float)(cz - CELL_OFF + 2) * cell;
            float gmin = fminf(fminf(qx - lox, hix - qx), fminf(fminf(qy - loy, hiy - qy), fminf(qz - loz, hiz - qz)));
            // a point outside has |delta| >= gmin on one axis -> its computed sqdist >= fl(gmin*gmin)
            proven = __uint_as_float(d5bits) < gmin * gmin;
        }
    }
    // ---- second chance (round 4, OPT-IN: a.tune bit 1 -- measured and not adopted): the rare group whose
    // result stayed unproven (41-55 of 100 000 queries at C2: sparse places, fewer than five points in the 27 cells) is finished
    // HERE by the whole wavefront -- the 98 shell cells of the 5x5x5 block, pruned by the 5th distance the group already holds --
    // instead of being listed for the walker workgroups of the measurement kernel.  Bit-identical results, an empty list at
    // every BASELINE config, the search pass's measurement kernel 2.2 us shorter at C2 (1.0 at C3) -- and this kernel 25 us
    // LONGER at C2 (6 at C3): a wavefront that takes the second chance runs a chain of dependent round trips (two per cell
    // pair, one per shell bucket) while the launch is bound by its last wavefronts; the batched form that needs few round
    // trips (block5_search) takes the kernel from 56 to 81-92 VGPRs.  The walk stays where it overlaps with the model.
    {
        const u64 need = (a.tune & 2) ? __ballot(active && r == 0 && !proven) : 0ull;
        if (need != 0ull) {
            __shared__ uint2 s_bk5[WPB][100];
            for (u64 mm = need; mm != 0ull; mm &= mm - 1ull) {
                const int gl = __ffsll((long long)mm) - 1;           // first lane of the unproven group
                const float sx = __shfl(qx, gl, 64), sy = __shfl(qy, gl, 64), sz = __shfl(qz, gl, 64);
                WaveTop5 t5;
                {
                    u64 kk[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const long long v = __shfl(__double_as_longlong(k < G ? mine0 : mine1), gl + (k % G), 64);
                        kk[k] = (u64)v;
                    }
                    t5.g0 = kk[0]; t5.g1 = kk[1]; t5.g2 = kk[2]; t5.g3 = kk[3]; t5.g4 = kk[4];
                }
                const bool p2 = shell_second_chance(a, sx, sy, sz, lane, s_bk5[wave], t5);
                const u64 res[K] = {t5.g0, t5.g1, t5.g2, t5.g3, t5.g4};
                int nv2 = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) nv2 += ((uint32_t)(res[k] >> 32) < INF_BITS) ? 1 : 0;
                // back to the group's lanes: lane gl + j keeps result j (and j + G when G < 5)
                const int jj = lane - gl;
                if (jj >= 0 && jj < G) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (k % G == jj) { if (k < G) mine0 = __longlong_as_double((long long)res[k]); else mine1 = __longlong_as_double((long long)res[k]); }
                    }
                    if (jj == 0) { proven = p2; nvalid = nv2; }
                }
            }
        }
    }

    stamp(5, false);
    // ---- write Nearest_Points[q] ---------------------------------------------------------------
#pragma unroll
    for (int half = 0; half < (G < K ? 2 : 1); ++half) {
        const int slot = r + half * G;
        if (active && slot < K) {
            const double mine = half == 0 ? mine0 : mine1;
            const uint32_t db = key_dist_bits(mine);
            float4 o;
            if (db < INF_BITS) {
                const float4 p = a.map_pts[key_index(mine)];
                o = make_float4(p.x, p.y, p.z, __uint_as_float(db));
            } else {
                const float nanv = __uint_as_float(0x7fc00000u);
                o = make_float4(nanv, nanv, nanv, __uint_as_float(INF_BITS));
            }
            a.nbr[(size_t)slot * a.nq_stride + q] = o;
        }
    }

    uint32_t slot = 0xffffffffu;                   // list entry of this query when it stays unproven (lane r == 0)
    if (active && r == 0) {
        // unproven queries are redone by the exact tree walk (inside the measurement kernel of the pass, or
        // knn_tree_kernel); until then their count carries the mark the owner of the query looks for
        a.cnt[q] = (uint8_t)nvalid | (proven ? (uint8_t)0 : CNT_UNPROVEN);
        if (!proven) {
            slot = atomicAdd(a.n_unres, 1u);
            a.ulist[slot] = make_float4(qx, qy, qz, __uint_as_float((uint32_t)q));
        }
    }
    // The walk does not start from nothing: the five keys of this result are the best five of the 27 cells (of the
    // cells the bound left, in a bounded search -- the others hold nothing that can enter), so the walk seeds its list
    // with them and only looks at what lies OUTSIDE the block and within their 5th distance.
    {
        const uint32_t gslot = (uint32_t)__shfl((int)slot, lane - r, 64);
        if (gslot != 0xffffffffu) {
            unsigned long long* dst = a.useeds + (size_t)gslot * 8u;
            if (r < K) dst[r] = (unsigned long long)__double_as_longlong(mine0);
            if (G < K && r + G < K) dst[r + G] = (unsigned long long)__double_as_longlong(mine1);
        }
    }
    stamp(6, true);
    if constexpr (STAMP) {
        if (a.stamps != nullptr && threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < KNN_STAMP_WORDS; ++i) a.stamps[(size_t)blockIdx.x * KNN_STAMP_WORDS + i] = tst[i];
        }
    }
    // candidate census for the byte model (opt-in: one atomic per wavefront)
    if (CENSUS) {
        uint32_t c = (active && r == 0) ? csum : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0 && c != 0) atomicAdd(a.cand_total, (unsigned long long)c);
    }
}

template <int G, int U>
static void launch_knn27_gu(const KnnArgs& a, hipStream_t s, KernelTimer t) {
    const int qpb = 4 * (64 / G);                  // queries per 256-thread workgroup
    const dim3 grid((a.nq + qpb - 1) / qpb), block(256);
    if constexpr (G == 8 && U == 4) {
        if (a.stamps != nullptr && a.pack_src == nullptr) {      // diagnostics: the stamped instantiation
            if (a.use_hint) hipLaunchKernelGGL((knn27_kernel<G, false, true, U, false, true>), grid, block, 0, s, a);
            else            hipLaunchKernelGGL((knn27_kernel<G, false, false, U, false, true>), grid, block, 0, s, a);
            return;
        }
    }
    if (t.start != nullptr) {
        if (a.use_hint)               hipExtLaunchKernelGGL((knn27_kernel<G, false, true, U>), grid, block, 0, s, t.start, t.stop, 0, a);
        else if (a.pack_src != nullptr) hipExtLaunchKernelGGL((knn27_kernel<G, false, false, U, true>), grid, block, 0, s, t.start, t.stop, 0, a);
        else                          hipExtLaunchKernelGGL((knn27_kernel<G, false, false, U>), grid, block, 0, s, t.start, t.stop, 0, a);
        return;
    }
    if (a.use_hint)               hipLaunchKernelGGL((knn27_kernel<G, false, true, U>), grid, block, 0, s, a);
    else if (a.pack_src != nullptr) hipLaunchKernelGGL((knn27_kernel<G, false, false, U, true>), grid, block, 0, s, a);
    else                          hipLaunchKernelGGL((knn27_kernel<G, false, false, U>), grid, block, 0, s, a);
}
template <int G>
static void launch_knn27_g(const KnnArgs& a, hipStream_t s, KernelTimer t) {
    if constexpr (G == 4 || G == 8 || G == 16) {
        // the prefix-form walk with U candidates per lane in flight (KnnArgs.walk; the census keeps the plain walk)
        if (a.cand_total == nullptr && a.walk == 1) { launch_knn27_gu<G, 1>(a, s, t); return; }
        if (a.cand_total == nullptr && a.walk == 2) { launch_knn27_gu<G, 2>(a, s, t); return; }
        if (a.cand_total == nullptr && a.walk == 4) { launch_knn27_gu<G, 4>(a, s, t); return; }
    }
    const int qpb = 4 * (64 / G);                  // queries per 256-thread workgroup
    const dim3 grid((a.nq + qpb - 1) / qpb), block(256);
    if (t.start != nullptr) {
        if (a.cand_total != nullptr) hipExtLaunchKernelGGL((knn27_kernel<G, true, false>), grid, block, 0, s, t.start, t.stop, 0, a);
        else if (a.use_hint)         hipExtLaunchKernelGGL((knn27_kernel<G, false, true>), grid, block, 0, s, t.start, t.stop, 0, a);
        else                         hipExtLaunchKernelGGL((knn27_kernel<G, false, false>), grid, block, 0, s, t.start, t.stop, 0, a);
        return;
    }
    if (a.cand_total != nullptr) hipLaunchKernelGGL((knn27_kernel<G, true, false>), grid, block, 0, s, a);
    else if (a.use_hint)         hipLaunchKernelGGL((knn27_kernel<G, false, true>), grid, block, 0, s, a);
    else                         hipLaunchKernelGGL((knn27_kernel<G, false, false>), grid, block, 0, s, a);   // (never with pack_src: launch_knn27)
}
// feats_down_world = pose (x) feats_down_body (laserMapping.cpp:1894-1899) as its own launch, one lane per point: for scans
// far beyond the 100 k of the node (C5: 1 M queries against a map that does not fit the Infinity Cache) the transform
// inside knn27_kernel -- one wavefront per workgroup, the others waiting at a barrier for a body-point load that now
// comes from HBM -- costs more than this launch (+35 us vs ~12 us at 1 M queries); at 100 k it is the other way round.
__global__ __launch_bounds__(256) void transform_kernel(const float4* body, int n, Pose14 pose, float4* world) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 b = body[i];
    const PoseQ s = pose_unpack(pose);
    float wx, wy, wz;
    body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
    world[i] = make_float4(wx, wy, wz, b.w);
}
hipError_t launch_transform(const float4* body, int n, const Pose14& pose, float4* world, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, body, n, pose, world);
    return hipGetLastError();
}

// ---- the gate of a pre-enqueued pass --------------------------------------------------------------------------------
// One wavefront polls the host's mail (system-scope loads of host-mapped memory: one PCIe read per poll) until the sequence
// number it was launched for appears, then copies the pose / the order into device memory; the kernels behind it on the
// stream start when it ends.  Bounded (~2 s of wall clock): a host that never answers turns into a skipped pass, which the
// host then reports (the record of that pass never appears), not into a hung queue.
__global__ __launch_bounds__(64) void gate_kernel(const GateMail* mail, unsigned long long expect, GateState* st) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned long long w = 0ull;
    bool ok = false;
    for (;;) {
        // lanes 0..16 fetch one word each: one round trip
        if (lane < 17) w = __hip_atomic_load(&mail->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long seq = __shfl(w, 0, 64);
        if (seq == expect) {
            unsigned long long all[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) all[i] = __shfl(w, i, 64);
            if (gate_check(all) == __shfl(w, 16, 64)) { ok = true; break; }
        }
        if (wall_clock64() - t0 > 200000000ull) break;             // 100 MHz: 2 s
        __builtin_amdgcn_s_sleep(1);
    }
    const bool go = ok && __shfl(w, 1, 64) == GATE_GO;
    if (lane >= 2 && lane < 16) st->pose.v[lane - 2] = go ? __longlong_as_double((long long)w) : 0.0;
    if (lane == 0) st->skip = go ? 0u : 1u;
}
hipError_t launch_gate(const GateMail* mail, unsigned long long expect, GateState* st, hipStream_t s) {
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, mail, expect, st);
    return hipGetLastError();
}

hipError_t launch_knn27(const KnnArgs& a, int lanes_per_query, hipStream_t s, KernelTimer t) {
    if (a.nq <= 0) return hipSuccess;
    switch (lanes_per_query) {
        case 4:  launch_knn27_g<4>(a, s, t); break;
        case 8:  launch_knn27_g<8>(a, s, t); break;
        case 16: launch_knn27_g<16>(a, s, t); break;
        case 32: launch_knn27_g<32>(a, s, t); break;
        case 64: launch_knn27_g<64>(a, s, t); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void knn_tree_kernel(TreeArgs a) {
    __shared__ TreeScratch ts;
    const int lane = threadIdx.x;
    // wavefront w takes entries w, w + W, ... of the list (unproven queries cluster in scan order; the list order is
    // arrival order, the stride spreads them over wavefronts).  The first entry is fetched together with the count.
    float4 e = a.ulist[blockIdx.x];
    const uint32_t n = __hip_atomic_load(a.n_unres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~TREE_OVERFLOW;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (i != blockIdx.x) e = a.ulist[i];
        tree_search_one(a, __float_as_uint(e.w), e.x, e.y, e.z, lane, ts, nullptr, a.useeds ? a.useeds + (size_t)i * 8u : nullptr);
    }
}

hipError_t launch_knn_tree(const TreeArgs& a, hipStream_t s, KernelTimer t) {
    if (a.nq <= 0) return hipSuccess;
    // one wavefront per workgroup; the list is short (tens of entries per 100 k queries), a few hundred wavefronts
    // cover it with one entry each, larger lists are strided
    int blocks = a.nq / 256;
    if (blocks < 256) blocks = 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks > a.nq) blocks = a.nq;                 // the list holds at most nq entries (and a.ulist[blockIdx.x] is read unconditionally)
    if (t.start != nullptr) hipExtLaunchKernelGGL(knn_tree_kernel, dim3(blocks), dim3(64), 0, s, t.start, t.stop, 0, a);
    else hipLaunchKernelGGL(knn_tree_kernel, dim3(blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace lio
