// lio_knn.hip -- exact k-NN (k = 5) of a scan against the GPU-resident map, gfx950 only.
//
// Replaces KD_TREE::Nearest_Search -> KD_TREE::Search (include/ikd-Tree/ikd_Tree.cpp:367-397,
// 869-1013): same fp32 distance expression (ikd_Tree.cpp:1374-1378), same visible point set
// (tomb-stoned points are skipped like point_deleted nodes, ikd_Tree.cpp:885), unbounded search
// radius (max_dist = INFINITY, ikd_Tree.h:236), results ascending by squared distance.
//
// Two kernels:
//   knn27_kernel<G>  G lanes (a DPP-addressable sub-group of a 64-lane wavefront) per query.
//                    The lanes split the 27 hash probes of the 3x3x3 cell block, stage the
//                    non-empty buckets {start,count} in LDS, stream the candidate points
//                    (16-byte loads, consecutive lanes -> consecutive addresses inside a bucket),
//                    keep a per-lane sorted top-5 of (sqdist bits << 32 | map index) keys and
//                    merge the G lists with DPP min all-reduces.  A result is PROVEN when the 5th
//                    distance is below the squared distance from the query to the faces of the
//                    searched block (exact in fp32 because the cell edge is a power of two);
//                    everything else goes to
//   knn_tree_kernel  one lane per unresolved query: branch-and-bound walk of the implicit octree
//                    formed by the Morton-sorted point array (children = sub-ranges found by
//                    binary search on the Morton keys; LCA jump skips single-child levels).
//
// Ordering of equal squared distances: by map index (the reference's choice among exact ties
// is traversal-order dependent, ikd_Tree.cpp:887).
#include "lio_device.h"
#include "lio_kernels.h"
#include "lio_math.h"

namespace lio {

typedef unsigned long long u64;

// ---- wavefront helpers (wave64, gfx9 DPP controls) -------------------------------------------
template <int CTRL>
__device__ __forceinline__ u64 dpp_move_u64(u64 v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}
__device__ __forceinline__ u64 min_u64(u64 a, u64 b) { return a < b ? a : b; }

// minimum over each aligned group of G lanes, result in every lane of the group
template <int G>
__device__ __forceinline__ u64 group_min_u64(u64 v) {
    v = min_u64(v, dpp_move_u64<0xB1>(v));                   // quad_perm [1,0,3,2]
    v = min_u64(v, dpp_move_u64<0x4E>(v));                   // quad_perm [2,3,0,1]
    if (G >= 8) v = min_u64(v, dpp_move_u64<0x141>(v));      // row_half_mirror
    if (G >= 16) v = min_u64(v, dpp_move_u64<0x140>(v));     // row_mirror
    if (G >= 32) v = min_u64(v, __shfl_xor(v, 16, 64));
    if (G >= 64) v = min_u64(v, __shfl_xor(v, 32, 64));
    return v;
}

// sorted (ascending) list of five keys in registers
struct Top5 {
    u64 k0, k1, k2, k3, k4;
    __device__ __forceinline__ void init() { k0 = k1 = k2 = k3 = k4 = KEY64_MAX; }
    __device__ __forceinline__ void offer(u64 key) {
        if (key < k4) {
            k4 = key;
            u64 t;
            if (k4 < k3) { t = k3; k3 = k4; k4 = t; }
            if (k3 < k2) { t = k2; k2 = k3; k3 = t; }
            if (k2 < k1) { t = k1; k1 = k2; k2 = t; }
            if (k1 < k0) { t = k0; k0 = k1; k1 = t; }
        }
    }
    __device__ __forceinline__ void pop() { k0 = k1; k1 = k2; k2 = k3; k3 = k4; k4 = KEY64_MAX; }
};

__device__ __forceinline__ u64 candidate_key(float qx, float qy, float qz, float4 p, uint32_t idx) {
    const float d = sqdist3(qx, qy, qz, p.x, p.y, p.z);
    uint32_t db = __float_as_uint(d);
    // tomb-stoned or non-comparable (NaN) candidates sort after every real one and are dropped
    if (p.w != 0.0f || !(d == d)) db = INF_BITS;
    return ((u64)db << 32) | (u64)idx;
}

// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void knn27_kernel(KnnArgs a) {
    constexpr int GPW = 64 / G;                    // queries per wavefront
    constexpr int SLOTS = (27 + G - 1) / G;        // hash probes per lane
    __shared__ uint2 s_bucket[4][GPW][28];         // LDS-staged non-empty buckets {start, count}

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, r = lane % G;
    const int q = (blockIdx.x * 4 + wave) * GPW + g;
    const bool active = q < a.nq;

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) {
        if (a.body != nullptr) {
            const float4 b = a.body[q];
            const PoseQ s = pose_unpack(a.pose);
            body_to_world(s, b.x, b.y, b.z, qx, qy, qz);
            if (r == 0 && a.world_out != nullptr) a.world_out[q] = make_float4(qx, qy, qz, b.w);
        } else {
            const float4 w = a.world_in[q];
            qx = w.x; qy = w.y; qz = w.z;
        }
    }
    const float inv_cell = a.grid.inv_cell;
    const int cx = cell_coord(qx, inv_cell), cy = cell_coord(qy, inv_cell), cz = cell_coord(qz, inv_cell);

    // ---- 27 hash probes, SLOTS per lane -----------------------------------------------------
    uint32_t st[SLOTS], ct[SLOTS];
    const uint32_t mask = a.grid.table_mask;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = r + s * G;
        st[s] = 0; ct[s] = 0;
        if (active && j < 27) {
            const int nx = cx + (j / 9) - 1, ny = cy + ((j / 3) % 3) - 1, nz = cz + (j % 3) - 1;
            if ((uint32_t)nx <= CELL_MAX && (uint32_t)ny <= CELL_MAX && (uint32_t)nz <= CELL_MAX) {
                table_find(a.table, mask, pack_key((uint32_t)nx, (uint32_t)ny, (uint32_t)nz), st[s], ct[s]);
            }
        }
    }

    // ---- compact the non-empty buckets of each group into LDS ---------------------------------
    int n_ne = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const bool ne = ct[s] != 0;
        const u64 b = __ballot(ne);
        const u64 gb = (G == 64) ? b : ((b >> (g * G)) & ((1ull << (G & 63)) - 1ull));
        const int pos = n_ne + __popcll(gb & ((1ull << r) - 1ull));
        if (ne) s_bucket[wave][g][pos] = make_uint2(st[s], ct[s]);
        n_ne += __popcll(gb);
    }
    // LDS is in-order per wavefront; the fence keeps the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stream the candidates: lane r takes points r, r+G, ... of every bucket ---------------
    Top5 top; top.init();
    uint32_t csum = 0;
    for (int i = 0; i < n_ne; ++i) {
        const uint2 bk = s_bucket[wave][g][i];
        csum += bk.y;
        for (uint32_t t = (uint32_t)r; t < bk.y; t += G) {
            const uint32_t idx = bk.x + t;
            const float4 p = a.map_pts[idx];
            top.offer(candidate_key(qx, qy, qz, p, idx));
        }
    }

    // ---- merge the G sorted lists: five rounds of group-min + pop ------------------------------
    u64 mine = KEY64_MAX;
    int nvalid = 0;
    uint32_t d5bits = INF_BITS;
#pragma unroll
    for (int rd = 0; rd < K; ++rd) {
        const u64 m = group_min_u64<G>(top.k0);
        if (top.k0 == m && m != KEY64_MAX) top.pop();      // keys are unique (distinct map indices)
        if (r == rd) mine = m;
        nvalid += ((uint32_t)(m >> 32) < INF_BITS) ? 1 : 0;
        if (rd == K - 1) d5bits = (uint32_t)(m >> 32);
    }

    // ---- write Nearest_Points[q] ---------------------------------------------------------------
    if (active && r < K) {
        const uint32_t db = (uint32_t)(mine >> 32);
        float4 o;
        if (db < INF_BITS) {
            const float4 p = a.map_pts[(uint32_t)mine];
            o = make_float4(p.x, p.y, p.z, __uint_as_float(db));
        } else {
            const float nanv = __uint_as_float(0x7fc00000u);
            o = make_float4(nanv, nanv, nanv, __uint_as_float(INF_BITS));
        }
        a.nbr[(size_t)r * a.nq_stride + q] = o;
    }

    // ---- proof: every point outside the searched block is farther than the 5th neighbour ------
    bool unres = false;
    if (active && r == 0) {
        a.cnt[q] = (uint8_t)nvalid;
        const float cell = a.grid.cell;
        bool proven = false;
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
        unres = !proven;
    }
    // wave-aggregated append to the unresolved list
    const u64 ub = __ballot(unres);
    if (ub != 0) {
        const int leader = __ffsll((long long)ub) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(a.n_unresolved, (uint32_t)__popcll(ub));
        base = __shfl(base, leader, 64);
        if (unres) a.unresolved[base + __popcll(ub & ((1ull << lane) - 1ull))] = (uint32_t)q;
    }
    // candidate census for the byte model (one atomic per wavefront)
    if (a.cand_total != nullptr) {
        uint32_t c = (active && r == 0) ? csum : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0 && c != 0) atomicAdd(a.cand_total, (unsigned long long)c);
    }
}

hipError_t launch_knn27(const KnnArgs& a, int lanes_per_query, hipStream_t s) {
    if (a.nq <= 0) return hipSuccess;
    const int qpb = 256 / lanes_per_query;         // queries per 256-thread workgroup
    const dim3 grid((a.nq + qpb - 1) / qpb), block(256);
    switch (lanes_per_query) {
        case 8:  hipLaunchKernelGGL(knn27_kernel<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(knn27_kernel<16>, grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL(knn27_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(knn27_kernel<64>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Exact fallback: branch-and-bound over the implicit octree of the Morton-sorted array.
// ---------------------------------------------------------------------------------------------
struct TStackEntry { uint32_t lo, hi; float bd; };

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t* keys, uint32_t lo, uint32_t hi, uint64_t target) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// squared distance from q to the axis-aligned box of the octree node `code` (a Morton prefix at
// `level`, i.e. 2^level cells per edge); same fp32 expression form as sqdist3, so it never exceeds
// the computed distance of any point stored inside the node.
__device__ __forceinline__ float node_box_sqdist(float qx, float qy, float qz, uint64_t code, int level, float cell) {
    const uint32_t nx = compact21(code >> 2), ny = compact21(code >> 1), nz = compact21(code);
    const float lox = (float)((long long)((u64)nx << level) - CELL_OFF) * cell;
    const float loy = (float)((long long)((u64)ny << level) - CELL_OFF) * cell;
    const float loz = (float)((long long)((u64)nz << level) - CELL_OFF) * cell;
    const float edge = (float)(1ull << level) * cell;
    const float hix = lox + edge, hiy = loy + edge, hiz = loz + edge;
    const float dx = qx < lox ? lox - qx : (qx > hix ? qx - hix : 0.0f);
    const float dy = qy < loy ? loy - qy : (qy > hiy ? qy - hiy : 0.0f);
    const float dz = qz < loz ? loz - qz : (qz > hiz ? qz - hiz : 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

__global__ __launch_bounds__(64) void knn_tree_kernel(TreeArgs a) {
    const uint32_t n_list = *a.n_list;
    const uint32_t n = a.grid.n_points;
    const float cell = a.grid.cell;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_list; t += gridDim.x * blockDim.x) {
        const uint32_t q = a.list[t];
        const float4 qw = a.queries[q];
        const float qx = qw.x, qy = qw.y, qz = qw.z;
        Top5 top; top.init();

        TStackEntry stack[176];                     // <= 7 siblings per level * 21 levels + 1
        int sp = 0;
        if (n > 0) { stack[0] = {0u, n, 0.0f}; sp = 1; }
        while (sp > 0) {
            const TStackEntry nd = stack[--sp];
            // squared distance of the current 5th candidate; +inf until five real points are held
            const uint32_t tb = (uint32_t)(top.k4 >> 32);
            const float tau = __uint_as_float(tb < INF_BITS ? tb : INF_BITS);
            if (nd.bd > tau) continue;
            const uint32_t cntn = nd.hi - nd.lo;
            const uint64_t kf = a.mkeys[nd.lo], kl = a.mkeys[nd.hi - 1];
            const uint64_t diff = kf ^ kl;
            if (cntn <= 8 || diff == 0) {
                for (uint32_t i = nd.lo; i < nd.hi; ++i) top.offer(candidate_key(qx, qy, qz, a.map_pts[i], i));
                continue;
            }
            // lowest common ancestor of the range: children live at level (lvl - 1)
            const int msb = 63 - __clzll((long long)diff);
            const int lvl = msb / 3 + 1;
            const int shift = 3 * (lvl - 1);
            const uint64_t parent = kf >> (3 * lvl);
            uint32_t bnd[9];
            bnd[0] = nd.lo; bnd[8] = nd.hi;
            for (int c = 1; c < 8; ++c)
                bnd[c] = lower_bound_u64(a.mkeys, bnd[c - 1], nd.hi, ((parent << 3) | (uint64_t)c) << shift);
            // children with points, box distance, sorted far -> near so that the nearest is popped first
            float cbd[8]; int cid[8]; int nc = 0;
            for (int c = 0; c < 8; ++c) {
                if (bnd[c + 1] > bnd[c]) {
                    const float bd = node_box_sqdist(qx, qy, qz, (parent << 3) | (uint64_t)c, lvl - 1, cell);
                    if (bd <= tau) {
                        int p = nc++;
                        while (p > 0 && cbd[p - 1] < bd) { cbd[p] = cbd[p - 1]; cid[p] = cid[p - 1]; --p; }
                        cbd[p] = bd; cid[p] = c;
                    }
                }
            }
            for (int i = 0; i < nc; ++i) {
                if (sp < 176) stack[sp++] = {bnd[cid[i]], bnd[cid[i] + 1], cbd[i]};
            }
        }
        // write the result (ascending)
        const u64 res[K] = {top.k0, top.k1, top.k2, top.k3, top.k4};
        int nvalid = 0;
        for (int k = 0; k < K; ++k) {
            const uint32_t db = (uint32_t)(res[k] >> 32);
            float4 o;
            if (db < INF_BITS) {
                const float4 p = a.map_pts[(uint32_t)res[k]];
                o = make_float4(p.x, p.y, p.z, __uint_as_float(db));
                ++nvalid;
            } else {
                const float nanv = __uint_as_float(0x7fc00000u);
                o = make_float4(nanv, nanv, nanv, __uint_as_float(INF_BITS));
            }
            a.nbr[(size_t)k * a.nq_stride + q] = o;
        }
        a.cnt[q] = (uint8_t)nvalid;
    }
}

hipError_t launch_knn_tree(const TreeArgs& a, hipStream_t s) {
    if (a.nq <= 0) return hipSuccess;
    int blocks = (a.nq + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(knn_tree_kernel, dim3(blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace lio
