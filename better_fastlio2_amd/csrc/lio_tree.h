// lio_tree.h -- the exact tree walk of ONE query by ONE 64-lane wavefront (device code shared by the stand-alone
// knn_tree_kernel, lio_knn.hip, and by the measurement kernel of a search pass, lio_measure.hip, which settles the
// few queries its own workgroup's points left unproven instead of waiting for a separate launch).
#pragma once
#include "lio_device.h"
#include "lio_kernels.h"
#include "lio_math.h"

namespace lio {

typedef unsigned long long u64;

// ---- wavefront helpers (wave64, gfx9 DPP controls) -------------------------------------------
template <int CTRL>
__device__ __forceinline__ u64 dpp_move_u64(u64 v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    // the controls used here are full permutations of their rows: every lane receives a value, `old` is never kept
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}
__device__ __forceinline__ u64 min_u64(u64 a, u64 b) { return a < b ? a : b; }

// minimum over each aligned group of G lanes, result in every lane of the group
template <int G>
__device__ __forceinline__ u64 group_min_u64(u64 v) {
    if (G >= 2) v = min_u64(v, dpp_move_u64<0xB1>(v));       // quad_perm [1,0,3,2]
    if (G >= 4) v = min_u64(v, dpp_move_u64<0x4E>(v));       // quad_perm [2,3,0,1]
    if (G >= 8) v = min_u64(v, dpp_move_u64<0x141>(v));      // row_half_mirror
    if (G >= 16) v = min_u64(v, dpp_move_u64<0x140>(v));     // row_mirror
    if (G >= 32) v = min_u64(v, __shfl_xor(v, 16, 64));
    if (G >= 64) v = min_u64(v, __shfl_xor(v, 32, 64));
    return v;
}

__device__ __forceinline__ u64 candidate_key(float qx, float qy, float qz, float4 p, uint32_t idx) {
    const float d = sqdist3(qx, qy, qz, p.x, p.y, p.z);
    // tomb-stoned candidates (w != 0) sort after every real one and are dropped; a NaN distance
    // (non-finite query) has bits above +inf and is dropped by the same `< INF_BITS` tests
    const uint32_t db = p.w != 0.0f ? INF_BITS : __float_as_uint(d);
    return ((u64)db << 32) | (u64)idx;
}

// ---------------------------------------------------------------------------------------------
// Exact fallback: best-first branch-and-bound over the hashed 64-ary tree of the map.
//
// Level L of the tree = the neighbour grid coarsened by 4^L (cell edge = cell * 4^L).  Levels 1 .. top-1 have a
// hash table each (packed coarse cell -> existence; level 1 -> number of its child block); the level-0 cells of
// a level-1 node are the 64 records of its child block, and their buckets -- contiguous runs of the point
// array -- are the leaves.  The walk starts from the few cells of level `top_level` (TreeLevels).
// One 64-lane wavefront per query: the 64 lanes probe the 4x4x4 children of a node in ONE
// round trip, rank the non-empty children by box distance (ballot + readlane), push them on an
// LDS stack nearest-on-top, and scan leaf ranges 64 points at a time; the running top-5 is
// wave-uniform and candidates are merged with DPP min reductions.
// ---------------------------------------------------------------------------------------------
// Only level-0 cells are leaves: their buckets are the one thing that is contiguous in the point array
// (a bucket moves to the tail of the array when points are added to its cell, lio_mapadd.hip).  The walk
// starts with at most TREE_MAX_ROOTS entries, a pop pushes at most 63 net entries and the tree is at most 11
// levels deep (21-bit cell coordinates): 32 + 63 * 11 < 736.
constexpr int TREE_STACK = 736;
constexpr uint32_t TREE_OVERFLOW = 0x80000000u;  // or-ed into the unproven-query counter if the stack bound were ever violated

// squared distance from q to the box of cell (cx,cy,cz) at `level` (edge = cell * 4^level); same
// fp32 expression form as sqdist3, so it never exceeds the computed distance of a point inside.
__device__ __forceinline__ float cell_box_sqdist(float qx, float qy, float qz, uint32_t cx, uint32_t cy, uint32_t cz,
                                                 int level, float cell) {
    const int sh = 2 * level;
    const float lox = (float)((long long)((u64)cx << sh) - CELL_OFF) * cell;
    const float loy = (float)((long long)((u64)cy << sh) - CELL_OFF) * cell;
    const float loz = (float)((long long)((u64)cz << sh) - CELL_OFF) * cell;
    const float edge = (float)(1ull << sh) * cell;
    const float hix = lox + edge, hiy = loy + edge, hiz = loz + edge;
    const float dx = qx < lox ? lox - qx : (qx > hix ? qx - hix : 0.0f);
    const float dy = qy < loy ? loy - qy : (qy > hiy ? qy - hiy : 0.0f);
    const float dz = qz < loz ? loz - qz : (qz > hiz ? qz - hiz : 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

// wave-uniform sorted top-5 merged from per-lane candidate keys
struct WaveTop5 {
    u64 g0, g1, g2, g3, g4;
    __device__ __forceinline__ void init() { g0 = g1 = g2 = g3 = g4 = KEY64_MAX; }
    __device__ __forceinline__ float tau() const {
        const uint32_t tb = (uint32_t)(g4 >> 32);
        return __uint_as_float(tb < INF_BITS ? tb : INF_BITS);
    }
    // every lane contributes one key (KEY64_MAX = nothing); at most five extractions.  Keys already
    // held are ignored, so a point may be offered twice (ring phase, then tree phase).
    __device__ __forceinline__ void merge(u64 key) {
        if (key == g0 || key == g1 || key == g2 || key == g3 || key == g4) key = KEY64_MAX;
        for (int it = 0; it < K; ++it) {
            const bool better = key < g4;
            if (!__any(better)) break;
            const u64 m = group_min_u64<64>(better ? key : KEY64_MAX);
            g4 = m;
            u64 t;
            if (g4 < g3) { t = g3; g3 = g4; g4 = t; }
            if (g3 < g2) { t = g2; g2 = g3; g3 = t; }
            if (g2 < g1) { t = g1; g1 = g2; g2 = t; }
            if (g1 < g0) { t = g0; g0 = g1; g1 = t; }
            if (key == m) key = KEY64_MAX;
        }
    }
};

// lanes 0..4 write the five results (ascending) of query q; lane 0 the count.  Returns the lane's own result (lane k < 5:
// neighbour k as x, y, z, sqdist) and the number of valid results, for a caller that goes on with them.
__device__ __forceinline__ float4 write_wave_result(const TreeArgs& a, uint32_t q, int lane, const WaveTop5& top, int& nvalid_out) {
    const u64 res[K] = {top.g0, top.g1, top.g2, top.g3, top.g4};
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) nvalid += ((uint32_t)(res[k] >> 32) < INF_BITS) ? 1 : 0;
    nvalid_out = nvalid;
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < K) {
        u64 mine = res[0];
#pragma unroll
        for (int k = 1; k < K; ++k) if (lane == k) mine = res[k];
        const uint32_t db = (uint32_t)(mine >> 32);
        float4 o;
        if (db < INF_BITS) {
            const float4 p = a.map_pts[(uint32_t)mine];
            o = make_float4(p.x, p.y, p.z, __uint_as_float(db));
        } else {
            const float nanv = __uint_as_float(0x7fc00000u);
            o = make_float4(nanv, nanv, nanv, __uint_as_float(INF_BITS));
        }
        // Results are written THROUGH (agent-scope 8-byte stores; the owner of the query reads them back with agent-scope
        // loads): a walk may run inside the same launch as its consumer (lio_measure.hip), on another XCD, whose L2 may
        // still hold the line from before the walk -- plain stores + a release fence were observed stale there.
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.nbr + (size_t)lane * a.nq_stride + q);
        const unsigned long long lo = ((unsigned long long)__float_as_uint(o.y) << 32) | (unsigned long long)__float_as_uint(o.x);
        const unsigned long long hi = ((unsigned long long)__float_as_uint(o.w) << 32) | (unsigned long long)__float_as_uint(o.z);
        __hip_atomic_store(dst + 0, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        own = o;
    }
    // the mark stays: whoever owns the query decides from the MARK (written by the k-NN launch, i.e. reliably visible)
    // that it has to wait for the walk
    if (lane == 0) __hip_atomic_store(a.cnt + q, (uint8_t)((uint8_t)nvalid | CNT_UNPROVEN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return own;
}
// the owner's side of the hand-off: neighbour k of query q / its count, read past the caches the walk did not write
__device__ __forceinline__ float4 load_walked_neighbor(const float4* nbr, size_t index) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(nbr + index);
    const unsigned long long lo = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                       __uint_as_float((uint32_t)(hi >> 32)));
}

// Phase A of the exact walk as a function of its own: the 5x5x5 cell block around q, looked up in the cell directory by the 64
// lanes of a wavefront (two cells each), its candidates streamed as one flat list and merged into `top`.  seeded: `top` already
// holds the best five of the inner 3x3x3 block (knn27_kernel's result) -- then only the 98 shell cells within their 5th distance
// are looked at: a handful of cells and candidates instead of 125 cells and ~300.  Returns true when the result is PROVEN: the 5th
// distance lies below the distance from q to the faces of the 5x5x5 block.  s_bk / s_pre: 128 entries of LDS each, private to the
// wavefront; all 64 lanes call it together.  Used by tree_search_one below and by knn27_kernel (lio_knn.hip) as the SECOND
// CHANCE of a query whose 3x3x3 result stayed unproven -- at C2 every such query is settled here.
__device__ __forceinline__ bool block5_search(const float4* map_pts, const HashEntry* l1table, uint32_t l1mask, const uint2* child,
                                          float cell, float inv_cell, float qx, float qy, float qz, int lane, uint2* s_bk,
                                          uint32_t* s_pre, WaveTop5& top, bool seeded) {
    const int cx = cell_coord(qx, inv_cell), cy = cell_coord(qy, inv_cell), cz = cell_coord(qz, inv_cell);
    const bool inrange = cx >= 2 && cy >= 2 && cz >= 2 && (uint32_t)cx + 2u <= CELL_MAX &&
                         (uint32_t)cy + 2u <= CELL_MAX && (uint32_t)cz + 2u <= CELL_MAX;
    int n_ne = 0;
    if (inrange) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int j = lane + 64 * sl;
            uint32_t st = 0, ct = 0;
            const int ox = j / 25 - 2, oy = (j / 5) % 5 - 2, oz = j % 5 - 2;
            bool want = j < 125;
            if (seeded) {
                const bool inner = ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1;
                want = want && !inner &&
                       cell_box_sqdist(qx, qy, qz, (uint32_t)(cx + ox), (uint32_t)(cy + oy), (uint32_t)(cz + oz), 0, cell) <= top.tau();
            }
            if (want)
                cell_find(l1table, l1mask, child, (uint32_t)(cx + ox),
                          (uint32_t)(cy + oy), (uint32_t)(cz + oz), st, ct);
            const bool ne = ct != 0;
            const u64 b = __ballot(ne);
            const int pos = n_ne + __popcll(b & ((1ull << lane) - 1ull));
            if (ne) s_bk[pos] = make_uint2(st, ct);
            n_ne += __popcll(b);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // exclusive prefix of the bucket counts (<= 125 buckets, two per lane)
    uint32_t c0 = lane < n_ne ? s_bk[lane].y : 0u, c1 = lane + 64 < n_ne ? s_bk[lane + 64].y : 0u;
    uint32_t i0 = c0, i1 = c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u0 = __shfl_up(i0, o, 64), u1 = __shfl_up(i1, o, 64);
        if (lane >= o) { i0 += u0; i1 += u1; }
    }
    const uint32_t tot0 = __shfl(i0, 63, 64), tot1 = __shfl(i1, 63, 64);
    s_pre[lane] = i0 - c0;
    s_pre[lane + 64] = tot0 + i1 - c1;
    const uint32_t total = tot0 + tot1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // four 64-candidate chunks per round: their index searches and 16-byte gathers are all in
    // flight before the first merge (one memory round trip instead of four)
    for (uint32_t base = 0; base < total; base += 256) {
        u64 key[4];
        uint32_t idx[4];
        float4 pt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t f = base + 64u * c + (uint32_t)lane;
            idx[c] = 0xffffffffu;
            if (f < total) {
                int lo = 0, hi = n_ne - 1;               // largest b with s_pre[b] <= f
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_pre[mid] <= f) lo = mid; else hi = mid - 1;
                }
                idx[c] = s_bk[lo].x + (f - s_pre[lo]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            pt[c] = make_float4(0.f, 0.f, 0.f, 1.f);
            if (idx[c] != 0xffffffffu) pt[c] = map_pts[idx[c]];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            key[c] = idx[c] != 0xffffffffu ? candidate_key(qx, qy, qz, pt[c], idx[c]) : KEY64_MAX;
            if (base + 64u * c < total) top.merge(key[c]);
        }
    }
    bool proven = false;
    if (inrange && (uint32_t)(top.g4 >> 32) < INF_BITS) {
        const float lox = (float)(cx - CELL_OFF - 2) * cell, hix = (float)(cx - CELL_OFF + 3) * cell;
        const float loy = (float)(cy - CELL_OFF - 2) * cell, hiy = (float)(cy - CELL_OFF + 3) * cell;
        const float loz = (float)(cz - CELL_OFF - 2) * cell, hiz = (float)(cz - CELL_OFF + 3) * cell;
        const float gmin = fminf(fminf(qx - lox, hix - qx), fminf(fminf(qy - loy, hiy - qy), fminf(qz - loz, hiz - qz)));
        proven = top.tau() < gmin * gmin;
    }
    __builtin_amdgcn_wave_barrier();
    return proven;
}

// LDS of one walking wavefront
struct TreeScratch {
    uint4 node[TREE_STACK];          // {start, count, level, bd bits}
    uint4 cell[TREE_STACK];          // {cx, cy, cz, -} at the node's level
    uint2 bk[128];                   // phase A: non-empty buckets {start, count}
    uint32_t pre[128];               //          exclusive prefix of their counts
};

// Exact 5-NN of query q = (qx, qy, qz); all 64 lanes of the wavefront must call it together.  Returns what
// write_wave_result returns (lane k < 5: neighbour k), *nvalid = neighbours found.
__device__ __forceinline__ float4 tree_search_one(const TreeArgs& a, uint32_t q, float qx, float qy, float qz, int lane,
                                                  TreeScratch& ts, int* nvalid = nullptr, const u64* seeds = nullptr,
                                                  bool* changed = nullptr) {
    int nv_local = 0;
    u64 s0 = KEY64_MAX, s1 = KEY64_MAX, s2 = KEY64_MAX, s3 = KEY64_MAX, s4 = KEY64_MAX;
    uint4* const s_node = ts.node;
    uint4* const s_cell = ts.cell;
    uint2* const s_bk = ts.bk;
    uint32_t* const s_pre = ts.pre;
    const float cell = a.grid.cell;
    WaveTop5 top; top.init();
    // seeds (knn27_kernel's result for this query): the best five of the 3x3x3 block, so phase A skips the block and
    // prunes the shell with their 5th distance -- a handful of cells and candidates instead of 125 cells and ~300
    const bool seeded = seeds != nullptr;
    if (seeded) {
        const u64 mine = lane < K ? seeds[lane] : KEY64_MAX;
        top.g0 = __shfl(mine, 0, 64); top.g1 = __shfl(mine, 1, 64); top.g2 = __shfl(mine, 2, 64);
        top.g3 = __shfl(mine, 3, 64); top.g4 = __shfl(mine, 4, 64);
        s0 = top.g0; s1 = top.g1; s2 = top.g2; s3 = top.g3; s4 = top.g4;
    }
    // seeded and nothing better found: Nearest_Points[q] already holds exactly this result (the k-NN launch wrote it,
    // the mark stays in its count) -- nothing to load, nothing to store
    auto unchanged = [&]() { return seeded && top.g0 == s0 && top.g1 == s1 && top.g2 == s2 && top.g3 == s3 && top.g4 == s4; };

    // ---- phase A: the 5x5x5 cell block around the query (125 probes, 2 per lane).  Most queries
    // the 27-cell kernel could not prove are settled here in a handful of round trips.
    {
        const bool proven = block5_search(a.map_pts, a.ltable + a.levels.off[1], a.levels.mask[1], a.child, cell, a.grid.inv_cell, qx, qy, qz,
                                          lane, s_bk, s_pre, top, seeded);
        if (proven) {
            if (changed) *changed = !unchanged();
            if (unchanged()) { __builtin_amdgcn_wave_barrier(); return make_float4(0.f, 0.f, 0.f, 0.f); }
            const float4 own = write_wave_result(a, q, lane, top, nv_local);
            if (nvalid) *nvalid = nv_local;
            __builtin_amdgcn_wave_barrier();
            return own;
        }
    }

    // ---- phase B: best-first walk of the whole tree, pruned by what phase A already holds
    // the walk starts from every root cell that can still hold a better point, nearest on top of the stack
    int sp = 0;
    if (a.grid.n_points > 0) {
        float bd = 0.0f;
        bool want = false;
        uint32_t rx = 0, ry = 0, rz = 0;
        if (lane < a.levels.n_roots) {
            rx = a.levels.root_x[lane]; ry = a.levels.root_y[lane]; rz = a.levels.root_z[lane];
            bd = cell_box_sqdist(qx, qy, qz, rx, ry, rz, a.levels.top_level, cell);
            want = bd <= top.tau();
        }
        const u64 wm = __ballot(want);
        int rank = 0;
        for (u64 mm = wm; mm != 0; mm &= mm - 1) {
            const int j = __ffsll((long long)mm) - 1;
            const float bj = __shfl(bd, j, 64);
            rank += (bj > bd || (bj == bd && j < lane)) ? 1 : 0;
        }
        if (want) {
            s_node[rank] = make_uint4(0u, 1u, (uint32_t)a.levels.top_level, __float_as_uint(bd));
            s_cell[rank] = make_uint4(rx, ry, rz, 0u);
        }
        sp = __popcll(wm);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    while (sp > 0) {
        --sp;
        const uint4 nd = s_node[sp];
        const uint4 nc = s_cell[sp];
        __builtin_amdgcn_wave_barrier();
        if (__uint_as_float(nd.w) > top.tau()) continue;
        const int level = (int)nd.z;
        if (level == 0) {
            // leaf: stream the bucket, 64 points per round
            for (uint32_t base = 0; base < nd.y; base += 64) {
                const uint32_t i = base + (uint32_t)lane;
                u64 key = KEY64_MAX;
                if (i < nd.y) {
                    const uint32_t idx = nd.x + i;
                    key = candidate_key(qx, qy, qz, a.map_pts[idx], idx);
                }
                top.merge(key);
            }
            continue;
        }
        // the 64 children: lane j <-> (j >> 4, (j >> 2) & 3, j & 3)
        const int cl = level - 1;
        const uint32_t ccx = nc.x * 4u + (uint32_t)(lane >> 4), ccy = nc.y * 4u + (uint32_t)((lane >> 2) & 3),
                       ccz = nc.z * 4u + (uint32_t)(lane & 3);
        uint32_t st = 0, ct = 0;
        const uint32_t lim = CELL_MAX >> (2 * cl);
        if (ccx <= lim && ccy <= lim && ccz <= lim) {
            if (cl == 0) {
                // the 64 children of a level-1 node ARE its child block (nd.x = block number), lane <-> position
                const uint2 e = a.child[(size_t)nd.x * 64u + (uint32_t)lane];
                st = e.x; ct = e.y;
            } else {
                table_find(a.ltable + a.levels.off[cl], a.levels.mask[cl], pack_key(ccx, ccy, ccz), st, ct);
            }
        }
        const float bd = cell_box_sqdist(qx, qy, qz, ccx, ccy, ccz, cl, cell);
        const bool want = ct != 0 && bd <= top.tau();
        u64 wm = __ballot(want);
        const int nw = __popcll(wm);
        if (nw == 0) continue;
        // descending rank by (bd, lane): the nearest child ends on top of the stack
        int rank = 0;
        for (u64 mm = wm; mm != 0; mm &= mm - 1) {
            const int j = __ffsll((long long)mm) - 1;
            const float bj = __shfl(bd, j, 64);
            rank += (bj > bd || (bj == bd && j < lane)) ? 1 : 0;
        }
        if (sp + nw > TREE_STACK) {                    // unreachable (see TREE_STACK); reported, never silent
            if (lane == 0) atomicOr(a.n_unres, TREE_OVERFLOW);
            continue;
        }
        if (want) {
            s_node[sp + rank] = make_uint4(st, ct, (uint32_t)cl, __float_as_uint(bd));
            s_cell[sp + rank] = make_uint4(ccx, ccy, ccz, 0u);
        }
        sp += nw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (changed) *changed = !unchanged();
    if (unchanged()) { __builtin_amdgcn_wave_barrier(); return make_float4(0.f, 0.f, 0.f, 0.f); }
    const float4 own = write_wave_result(a, q, lane, top, nv_local);
    if (nvalid) *nvalid = nv_local;
    __builtin_amdgcn_wave_barrier();
    return own;
}

}  // namespace lio
