// lio_mapadd.hip -- KD_TREE::Add_Points (include/ikd-Tree/ikd_Tree.cpp:414-489) on the GPU index.
//
// The reference inserts the points of a batch one after another; with down-sampling on, every
// insert (a) box-searches the down-sample voxel of the point, (b) picks, among the point itself
// and the points found, the one nearest the voxel centre (the new point wins ties, strict `<`
// :440), and (c) if the voxel held more than one point or the winner is the new point, deletes
// the voxel's points and adds the winner (:445-448).  Inserts interact only through their voxel,
// so the batch is grouped by voxel (stable radix sort keeps the batch order inside a group) and
// one lane folds each group sequentially -- the result equals the sequential one.
//   voxel_keys_kernel     voxel key of each new point
//   fold_groups_kernel    per voxel: existing points in the box (directory look-ups of the <= 8 grid
//                         cells the box touches), sequential fold, tomb-stone + emit the survivor
// The caller then appends the survivors and rebuilds the index (sort + hash), see lio_ctx.hip.
#include "lio_device.h"
#include "lio_kernels.h"
#include "lio_mapadd.h"
#include "lio_math.h"
#include "lio_incr.h"

namespace lio {

__device__ __forceinline__ float voxel_floor(float v, float ds) { return floorf(v / ds); }   // :424

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float4* pts, uint32_t n, float ds, uint64_t* vkeys,
                                                         uint32_t* order, uint32_t* zero2, uint8_t* dirty, uint32_t* dlist,
                                                         float4* out2n) {
    // first kernel of an Add_Points(downsample) batch: it also clears what the later kernels accumulate into
    // (zero2[0..1] fold counters, zero2[6] number of boundary points) and marks every output slot "nothing"
    if (zero2 != nullptr && blockIdx.x == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0u;
    if (zero2 != nullptr && blockIdx.x == 0 && threadIdx.x == 2) zero2[6] = 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        dirty[i] = 0;
        dlist[i] = 0xffffffffu;
        out2n[i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        out2n[n + i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        const float4 p = pts[i];
        float kx = voxel_floor(p.x, ds), ky = voxel_floor(p.y, ds), kz = voxel_floor(p.z, ds);
        kx = fminf(fmaxf(kx, -1048576.0f), 1048575.0f);
        ky = fminf(fmaxf(ky, -1048576.0f), 1048575.0f);
        kz = fminf(fmaxf(kz, -1048576.0f), 1048575.0f);
        vkeys[i] = pack_key((uint32_t)((int)kx + CELL_OFF), (uint32_t)((int)ky + CELL_OFF), (uint32_t)((int)kz + CELL_OFF));
        order[i] = i;
    }
}

__device__ __forceinline__ bool same_point(float4 a, float4 b) {         // ikd_Tree.cpp:1365-1367, EPSS 1e-6
    return (double)fabsf(a.x - b.x) < 1e-6 && (double)fabsf(a.y - b.y) < 1e-6 && (double)fabsf(a.z - b.z) < 1e-6;
}

// Eight lanes per voxel group: the down-sample box (edge ds < cell edge) touches at most 2x2x2 grid cells, lane
// t = 4 dx + 2 dy + dz takes one of them -- probe, bucket scan and tomb-stoning of the eight cells run side by
// side instead of one after the other.  The reference visits the cells in (x, y, z) order and keeps the FIRST
// minimum (strict `<`), which is the minimum over (distance, t) here.
__global__ __launch_bounds__(64) void fold_groups_kernel(FoldArgs a) {
    const uint32_t n_groups = a.n_groups_dev != nullptr ? *a.n_groups_dev : a.n_groups;
    const int t = threadIdx.x & 7;
    const uint32_t gpw = 8u;                                    // groups per wavefront
    for (uint32_t g0 = blockIdx.x * gpw; g0 < a.n_groups_max; g0 += gridDim.x * gpw) {
        const uint32_t g = g0 + (threadIdx.x >> 3);
        if (g >= n_groups) {
            if (g < a.n_groups_max && t == 0) a.out_pts[g] = make_float4(0.f, 0.f, 0.f, 1.0f);   // nothing to append
            continue;                                           // whole sub-group leaves together
        }
        if (a.dirty[g] != 0) continue;                         // boundary points around: replayed by fold_sequential_kernel
        const uint32_t gs = a.group_start[g], gc = a.group_count[g];
        const float ds = a.ds;
        const float4 p0 = a.new_pts[a.order[gs]];
        // Box_of_Point and mid_point (:424-432): fp32 box, centre through double
        float bmin[3], bmax[3], mid[3];
        const float pc[3] = {p0.x, p0.y, p0.z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            bmin[ax] = voxel_floor(pc[ax], ds) * ds;
            bmax[ax] = bmin[ax] + ds;
            mid[ax] = (float)((double)bmin[ax] + (double)(bmax[ax] - bmin[ax]) / 2.0);
        }
        int c0[3], c1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            c0[ax] = cell_coord(bmin[ax], a.grid.inv_cell);
            c1[ax] = cell_coord(bmax[ax], a.grid.inv_cell);
        }
        const int cx = c0[0] + ((t >> 2) & 1), cy = c0[1] + ((t >> 1) & 1), cz = c0[2] + (t & 1);
        const bool mine = cx <= c1[0] && cy <= c1[1] && cz <= c1[2];
        // ---- pass 1: existing live points inside the box (Search_by_range :1015-1045) -------------
        uint32_t st = 0, ct = 0;
        if (mine) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, st, ct);
        uint32_t n_e = 0;
        float d_e = __uint_as_float(INF_BITS);
        float4 e_best = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t i = st; i < st + ct; ++i) {
            const float4 sp = a.map_pts[i];
            if (sp.w != 0.0f) continue;
            if (bmin[0] <= sp.x && bmax[0] > sp.x && bmin[1] <= sp.y && bmax[1] > sp.y && bmin[2] <= sp.z && bmax[2] > sp.z) {
                ++n_e;
                const float d = sqdist3(sp.x, sp.y, sp.z, mid[0], mid[1], mid[2]);
                if (d < d_e) { d_e = d; e_best = sp; }
            }
        }
        // combine the eight cells: count, and the first minimum in cell order
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            n_e += __shfl_xor(n_e, o, 64);
            const float od = __shfl_xor(d_e, o, 64);
            const float ox = __shfl_xor(e_best.x, o, 64), oy = __shfl_xor(e_best.y, o, 64), oz = __shfl_xor(e_best.z, o, 64);
            // the partner holds the lower cell numbers iff its lane number is lower
            const bool partner_first = ((t ^ o) < t);
            const bool take = partner_first ? (od <= d_e) : (od < d_e);
            if (take) { d_e = od; e_best = make_float4(ox, oy, oz, 0.0f); }
        }
        // ---- sequential fold over the group's new points (:433-449), identical in all eight lanes ------
        bool touched = false;
        float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
        float d_cur = 0.0f;
        uint32_t counter = 0;
        for (uint32_t j = 0; j < gc; ++j) {
            const float4 p = a.new_pts[a.order[gs + j]];
            const float d_p = sqdist3(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
            if (!touched) {
                const bool best_is_p = !(n_e > 0 && d_e < d_p);
                const float4 best = best_is_p ? p : e_best;
                if (n_e > 1 || same_point(p, best)) {
                    touched = true; cur = best; d_cur = best_is_p ? d_p : d_e; ++counter;
                }
            } else {
                const bool best_is_p = !(d_cur < d_p);
                const float4 best = best_is_p ? p : cur;
                if (same_point(p, best)) { cur = best; d_cur = best_is_p ? d_p : d_cur; ++counter; }
            }
        }
        // ---- pass 2: Delete_by_range of the box + Add_by_point of the survivor --------------------
        uint32_t replaced = 0;
        if (touched) {
            for (uint32_t i = st; i < st + ct; ++i) {
                float4 sp = a.map_pts[i];
                if (sp.w != 0.0f) continue;
                if (bmin[0] <= sp.x && bmax[0] > sp.x && bmin[1] <= sp.y && bmax[1] > sp.y && bmin[2] <= sp.z && bmax[2] > sp.z) {
                    // exchange, not store: the bookkeeping stays exact even if two boxes ever claimed the same point
                    const uint32_t was = atomicExch(reinterpret_cast<uint32_t*>(&a.map_pts[i].w), __float_as_uint(TOMB_DOWNSAMPLE));
                    if (was == 0u) ++replaced;
                }
            }
        }
        if (replaced != 0) atomicAdd(a.n_replaced, replaced);
        if (t == 0) {
            a.out_pts[g] = touched ? make_float4(cur.x, cur.y, cur.z, 0.0f) : make_float4(0.f, 0.f, 0.f, 1.0f);
            if (counter != 0) atomicAdd(a.n_added, counter);
        }
    }
}

// ---- boundary points -----------------------------------------------------------------------------------
// index of `key` in the sorted group keys, or 0xffffffff
__device__ __forceinline__ uint32_t find_group(const uint64_t* ukeys, uint32_t n_groups, uint64_t key) {
    uint32_t lo = 0, hi = n_groups;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n_groups && ukeys[lo] == key) ? lo : 0xffffffffu;
}

// one lane per batch point: is it outside its own fp32 box, or inside a neighbour's?  Then its group and the groups
// of the voxels it reaches are dirty.
__global__ __launch_bounds__(256) void classify_boundary_kernel(BoundaryArgs a) {
    const uint32_t n_groups = *a.n_groups;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        const float4 p = a.new_pts[i];
        const float pc[3] = {p.x, p.y, p.z};
        int dir[3];
        float kf[3];
        bool irregular = false;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float ds = a.ds;
            kf[ax] = voxel_floor(pc[ax], ds);
            const float bmin = kf[ax] * ds, bmax = bmin + ds;
            const float up_min = (kf[ax] + 1.0f) * ds;              // lower bound of the upper neighbour's box
            const float dn_max = (kf[ax] - 1.0f) * ds + ds;         // upper bound of the lower neighbour's box
            const bool up = !(pc[ax] < bmax) || pc[ax] >= up_min;
            const bool dn = !(bmin <= pc[ax]) || pc[ax] < dn_max;
            dir[ax] = up ? 1 : (dn ? -1 : 0);
            irregular = irregular || up || dn;
        }
        if (!irregular) continue;
        for (int m = 0; m < 8; ++m) {                              // own voxel (m = 0) and every combination of the reached sides
            bool use = true;
            float k[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const bool bit = (m >> ax) & 1;
                if (bit && dir[ax] == 0) use = false;
                k[ax] = fminf(fmaxf(kf[ax] + (bit ? (float)dir[ax] : 0.0f), -1048576.0f), 1048575.0f);
            }
            if (!use) continue;
            const uint64_t key = pack_key((uint32_t)((int)k[0] + CELL_OFF), (uint32_t)((int)k[1] + CELL_OFF), (uint32_t)((int)k[2] + CELL_OFF));
            const uint32_t g = find_group(a.ukeys, n_groups, key);
            if (g != 0xffffffffu) a.dirty[g] = 1;
        }
    }
}

// Pre-pass of the fold, same lane layout: a LIVE MAP POINT inside the box of a group that also lies inside the box
// of a neighbouring voxel (fp32 boxes of neighbours can overlap by an ulp) couples the two groups -- if the
// neighbour has points in this batch, both go to the sequential pass.
__global__ __launch_bounds__(64) void detect_shared_points_kernel(BoundaryArgs a) {
    const uint32_t n_groups = *a.n_groups;
    const int t = threadIdx.x & 7;
    for (uint32_t g0 = blockIdx.x * 8u; g0 < a.n; g0 += gridDim.x * 8u) {
        const uint32_t g = g0 + (threadIdx.x >> 3);
        if (g >= n_groups) continue;
        const float ds = a.ds;
        const float4 p0 = a.new_pts[a.order[a.group_start[g]]];
        const float pc[3] = {p0.x, p0.y, p0.z};
        float kf[3], bmin[3], bmax[3], up_min[3], dn_max[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            kf[ax] = voxel_floor(pc[ax], ds);
            bmin[ax] = kf[ax] * ds; bmax[ax] = bmin[ax] + ds;
            up_min[ax] = (kf[ax] + 1.0f) * ds; dn_max[ax] = (kf[ax] - 1.0f) * ds + ds;
        }
        int c0[3], c1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { c0[ax] = cell_coord(bmin[ax], a.grid.inv_cell); c1[ax] = cell_coord(bmax[ax], a.grid.inv_cell); }
        const int cx = c0[0] + ((t >> 2) & 1), cy = c0[1] + ((t >> 1) & 1), cz = c0[2] + (t & 1);
        uint32_t st = 0, ct = 0;
        if (cx <= c1[0] && cy <= c1[1] && cz <= c1[2]) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, st, ct);
        for (uint32_t i = st; i < st + ct; ++i) {
            const float4 sp = a.map_pts[i];
            if (sp.w != 0.0f) continue;
            const float xc[3] = {sp.x, sp.y, sp.z};
            if (!(bmin[0] <= sp.x && bmax[0] > sp.x && bmin[1] <= sp.y && bmax[1] > sp.y && bmin[2] <= sp.z && bmax[2] > sp.z)) continue;
            int dir[3];
            bool shared = false;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                dir[ax] = xc[ax] >= up_min[ax] ? 1 : (xc[ax] < dn_max[ax] ? -1 : 0);
                shared = shared || dir[ax] != 0;
            }
            if (!shared) continue;
            for (int m = 1; m < 8; ++m) {
                bool use = true;
                float k[3];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const bool bit = (m >> ax) & 1;
                    if (bit && dir[ax] == 0) use = false;
                    k[ax] = fminf(fmaxf(kf[ax] + (bit ? (float)dir[ax] : 0.0f), -1048576.0f), 1048575.0f);
                }
                if (!use) continue;
                const uint64_t key = pack_key((uint32_t)((int)k[0] + CELL_OFF), (uint32_t)((int)k[1] + CELL_OFF), (uint32_t)((int)k[2] + CELL_OFF));
                const uint32_t h = find_group(a.ukeys, n_groups, key);
                if (h != 0xffffffffu) { a.dirty[h] = 1; a.dirty[g] = 1; }
            }
        }
    }
}

// one lane per group: the points of dirty groups are marked with their own batch index (dlist[i] = i, 0xffffffff
// otherwise); the marks, compacted in order, are the replay list
__global__ __launch_bounds__(256) void gather_dirty_kernel(BoundaryArgs a) {
    const uint32_t n_groups = *a.n_groups;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        if (a.dirty[g] == 0) continue;
        const uint32_t gs = a.group_start[g], gc = a.group_count[g];
        atomicAdd(a.n_dirty, gc);
        for (uint32_t j = 0; j < gc; ++j) { const uint32_t pt = a.order[gs + j]; a.dlist[pt] = pt; }
    }
}

// Add_Points(downsample) for the points of dirty groups, replayed in batch order by ONE wavefront with the
// reference's rule (ikd_Tree.cpp:422-449): Storage = live points inside the fp32 box (map points through the cell
// directory, lanes 0..7 one cell each; points added earlier in this pass from a list), nearest to the box centre
// wins (the new point on ties), delete-and-re-add when Storage held more than one point or the new point wins.
// Everything that later iterations read back goes through agent-scope atomics (no stale L1 lines).
__global__ __launch_bounds__(64) void fold_sequential_kernel(BoundaryArgs a, const uint32_t* dlist_sorted) {
    const int lane = threadIdx.x;
    const uint32_t n_dirty = *a.n_dirty;
    uint32_t n_seq = 0, counter = 0, replaced = 0;
    float* seq = reinterpret_cast<float*>(a.seq_out);
    float* mp = reinterpret_cast<float*>(a.map_pts);
    auto ld = [](const float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [](float* q, float v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (uint32_t it = 0; it < n_dirty; ++it) {
        const float4 p = a.new_pts[dlist_sorted[it]];
        const float ds = a.ds;
        float bmin[3], bmax[3], mid[3];
        const float pc[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            bmin[ax] = voxel_floor(pc[ax], ds) * ds;
            bmax[ax] = bmin[ax] + ds;
            mid[ax] = (float)((double)bmin[ax] + (double)(bmax[ax] - bmin[ax]) / 2.0);
        }
        auto inbox = [&](float x, float y, float z) {
            return bmin[0] <= x && bmax[0] > x && bmin[1] <= y && bmax[1] > y && bmin[2] <= z && bmax[2] > z;
        };
        // ---- Storage, part 1: map points (lanes 0..7)
        int c0[3], c1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { c0[ax] = cell_coord(bmin[ax], a.grid.inv_cell); c1[ax] = cell_coord(bmax[ax], a.grid.inv_cell); }
        const int cx = c0[0] + ((lane >> 2) & 1), cy = c0[1] + ((lane >> 1) & 1), cz = c0[2] + (lane & 1);
        const bool mine = lane < 8 && cx <= c1[0] && cy <= c1[1] && cz <= c1[2];
        uint32_t bst = 0, bct = 0;
        if (mine) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, bst, bct);
        uint32_t cnt = 0;
        float dbest = __uint_as_float(INF_BITS), bx = 0.f, by = 0.f, bz = 0.f;
        for (uint32_t i = bst; i < bst + bct; ++i) {
            const float* q = mp + 4 * (size_t)i;
            if (ld(q + 3) != 0.0f) continue;
            const float x = ld(q), y = ld(q + 1), z = ld(q + 2);
            if (!inbox(x, y, z)) continue;
            ++cnt;
            const float d = sqdist3(x, y, z, mid[0], mid[1], mid[2]);
            if (d < dbest) { dbest = d; bx = x; by = y; bz = z; }
        }
        // ---- Storage, part 2: points added earlier in this pass (all lanes, list order); they rank after the map's
        uint32_t ord = (uint32_t)lane;                             // tie order: map cells first (lane < 8), then list index + 8
        for (uint32_t j = (uint32_t)lane; j < n_seq; j += 64) {
            const float* q = seq + 4 * (size_t)j;
            if (ld(q + 3) != 0.0f) continue;
            const float x = ld(q), y = ld(q + 1), z = ld(q + 2);
            if (!inbox(x, y, z)) continue;
            ++cnt;
            const float d = sqdist3(x, y, z, mid[0], mid[1], mid[2]);
            if (d < dbest) { dbest = d; bx = x; by = y; bz = z; ord = 8u + j; }
        }
        // combine: count, and the first minimum in (distance, order)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            cnt += __shfl_xor(cnt, o, 64);
            const float od = __shfl_xor(dbest, o, 64);
            const uint32_t oo = __shfl_xor(ord, o, 64);
            const float ox = __shfl_xor(bx, o, 64), oy = __shfl_xor(by, o, 64), oz = __shfl_xor(bz, o, 64);
            if (od < dbest || (od == dbest && oo < ord)) { dbest = od; ord = oo; bx = ox; by = oy; bz = oz; }
        }
        const float d_p = sqdist3(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
        const bool best_is_p = !(cnt > 0 && dbest < d_p);
        const float4 best = best_is_p ? p : make_float4(bx, by, bz, 0.0f);
        if (cnt > 1 || same_point(p, best)) {
            ++counter;
            if (cnt > 0) {                                          // Delete_by_range of the box
                for (uint32_t i = bst; i < bst + bct; ++i) {
                    float* q = mp + 4 * (size_t)i;
                    if (ld(q + 3) != 0.0f) continue;
                    if (inbox(ld(q), ld(q + 1), ld(q + 2))) { st(q + 3, TOMB_DOWNSAMPLE); ++replaced; }
                }
                for (uint32_t j = (uint32_t)lane; j < n_seq; j += 64) {
                    float* q = seq + 4 * (size_t)j;
                    if (ld(q + 3) != 0.0f) continue;
                    if (inbox(ld(q), ld(q + 1), ld(q + 2))) st(q + 3, 1.0f);
                }
            }
            if (lane == 0) {                                        // Add_by_point of the winner
                float* q = seq + 4 * (size_t)n_seq;
                st(q, best.x); st(q + 1, best.y); st(q + 2, best.z); st(q + 3, 0.0f);
            }
            ++n_seq;
        }
        __builtin_amdgcn_s_waitcnt(0);                             // the appends / tomb-stones are out before the next point reads
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) replaced += __shfl_xor(replaced, o, 64);
    if (lane == 0) {
        if (counter != 0) atomicAdd(a.n_added, counter);
        if (replaced != 0) atomicAdd(a.n_replaced, replaced);
    }
}

// ---- incremental index update -------------------------------------------------------------------------
// n_head != nullptr: the list is [*n_head points][n_after points] inside a buffer of n slots (the first count is still on
// the device, lio_ctx.hip: add_batch_device); the slots behind them get a key no cell has, sort last and are dropped
// by the run detection
__global__ __launch_bounds__(256) void pend_keys_kernel(const float4* pend, uint32_t n, Grid grid, uint64_t* keys,
                                                        uint32_t* order, const uint32_t* n_head, uint32_t n_after) {
    const uint32_t n_real = n_head != nullptr ? min(n, *n_head + n_after) : n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k = ~0ull;
        if (i < n_real) {
            const float4 p = pend[i];
            k = pack_key((uint32_t)cell_coord(p.x, grid.inv_cell), (uint32_t)cell_coord(p.y, grid.inv_cell),
                         (uint32_t)cell_coord(p.z, grid.inv_cell));
        }
        keys[i] = k;
        order[i] = i;
    }
}
// dst_base[*offset + i] = src[i]: a batch appended behind a list whose length is still on the device
__global__ __launch_bounds__(256) void append_points_kernel(float4* dst_base, const uint32_t* offset, const float4* src, uint32_t n) {
    const uint32_t off = *offset;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst_base[off + i] = src[i];
}

// one lane per touched cell: where the cell is now and how large its re-written bucket will be
__global__ __launch_bounds__(64) void cell_plan_kernel(ApplyArgs a) {
    const uint32_t n_groups = *a.n_groups;
    // the statistics the apply kernel accumulates start from zero (nothing reads them before it)
    if (blockIdx.x == 0 && threadIdx.x < 3) a.stats[threadIdx.x] = 0u;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= a.n_scan; g += gridDim.x * blockDim.x) {
        if (g >= n_groups) { a.new_size[g] = 0u; continue; }      // the scan of new_size runs over n_scan + 1 entries
        const uint64_t key = a.cell_key[g];
        const uint32_t cx = (uint32_t)(key >> 42) & 0x1fffffu, cy = (uint32_t)(key >> 21) & 0x1fffffu, cz = (uint32_t)key & 0x1fffffu;
        CellPlan pl{0u, 0u, 0xffffffffu, 0u};
        cell_find(a.ltable + a.levels.off[1], a.levels.mask[1], a.child, cx, cy, cz, pl.old_start, pl.old_count, &pl.index);
        pl.found = pl.old_count != 0u ? 1u : 0u;                  // records are only ever written with count >= 1
        uint32_t live = 0;
        for (uint32_t i = 0; i < pl.old_count; ++i) live += a.map_pts[pl.old_start + i].w == 0.0f ? 1u : 0u;
        a.plan[g] = pl;
        a.new_size[g] = live + a.group_count[g];
    }
}

// find-or-insert of a cell key; returns the slot, or 0xffffffff when the table is full
__device__ __forceinline__ uint32_t table_claim(HashEntry* table, uint32_t mask, uint64_t key, bool& is_new) {
    uint32_t slot = hash_key(key) & mask;
    is_new = false;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&table[slot]);
        const unsigned long long prev = atomicCAS(kp, (unsigned long long)KEY_EMPTY, (unsigned long long)key);
        if (prev == KEY_EMPTY || prev == key) { is_new = prev == KEY_EMPTY; return slot; }
        slot = (slot + 1) & mask;
    }
    return 0xffffffffu;
}

__global__ __launch_bounds__(64) void cell_apply_kernel(ApplyArgs a) {
    const uint32_t n_groups = *a.n_groups;
    // room check on the device (every lane sees the same total): either the whole batch is applied or nothing is
    const uint32_t total = a.new_off[a.n_scan];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.stats[3] = total;
    if ((uint64_t)a.tail + (uint64_t)total > (uint64_t)a.slots_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.stats + 2, APPLY_FLAG_NO_ROOM);
        return;
    }
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        const CellPlan pl = a.plan[g];
        const uint32_t dst0 = a.tail + a.new_off[g];
        uint32_t dst = dst0;
        uint32_t retired = 0;
        for (uint32_t i = 0; i < pl.old_count; ++i) {           // live points keep their order
            float4 p = a.map_pts[pl.old_start + i];
            if (p.w != 0.0f) continue;
            a.map_pts[dst++] = p;
            p.w = TOMB_MOVED;
            a.map_pts[pl.old_start + i] = p;
            ++retired;
        }
        const uint32_t gs = a.group_start[g], gc = a.group_count[g];
        for (uint32_t j = 0; j < gc; ++j) {                      // then the new ones, in batch order
            const float4 p = a.pend[a.order[gs + j]];
            a.map_pts[dst++] = make_float4(p.x, p.y, p.z, 0.0f);
        }
        const uint32_t count = dst - dst0;
        if (retired != 0) atomicAdd(a.stats + 1, retired);
        const uint64_t key = a.cell_key[g];
        const uint32_t cx = (uint32_t)(key >> 42) & 0x1fffffu, cy = (uint32_t)(key >> 21) & 0x1fffffu, cz = (uint32_t)key & 0x1fffffu;
        const uint32_t pos = ((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u);        // place inside the child block
        if (pl.found) {
            a.child[pl.index] = make_uint2(dst0, count);
            continue;
        }
        // ---- a cell the map did not have ----------------------------------------------------------
        atomicAdd(a.stats + 0, 1u);
        // the cell must hang under one of the root cells of the tree
        {
            const int sh_top = 2 * a.levels.top_level;
            const uint32_t ax = cx >> sh_top, ay = cy >> sh_top, az = cz >> sh_top;
            bool under_root = false;
            for (int i = 0; i < a.levels.n_roots; ++i)
                under_root = under_root || (a.levels.root_x[i] == ax && a.levels.root_y[i] == ay && a.levels.root_z[i] == az);
            if (!under_root) { atomicOr(a.stats + 2, APPLY_FLAG_OUTSIDE_TOP); continue; }
        }
        // level 1: the entry names the child block of the cell directory (a new level-1 cell takes a block from the
        // pool; if two lanes race for the same new cell one block number is simply left unused).  Levels above:
        // existence marks (the tree walk only asks "is this child non-empty").
        for (int L = 1; L < a.levels.top_level; ++L) {
            const uint64_t ck = pack_key(cx >> (2 * L), cy >> (2 * L), cz >> (2 * L));
            HashEntry* lt = a.ltable + a.levels.off[L];
            bool fresh;
            const uint32_t ls = table_claim(lt, a.levels.mask[L], ck, fresh);
            if (ls == 0xffffffffu) { atomicOr(a.stats + 2, APPLY_FLAG_TABLE_FULL); break; }
            if (fresh) atomicAdd(a.stats + APPLY_LEVEL_NEW + L, 1u);   // entries this level gained since the last full rebuild
            if (L == 1) {
                uint32_t* sp = &lt[ls].start;
                uint32_t blk = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (blk == 0xffffffffu) {
                    const uint32_t nb = atomicAdd(a.block_bump, 1u);
                    if (nb >= a.blocks_cap) { atomicOr(a.stats + 2, APPLY_FLAG_NO_BLOCKS); break; }
                    const uint32_t prev = atomicCAS(sp, 0xffffffffu, nb);
                    blk = prev == 0xffffffffu ? nb : prev;
                }
                a.child[(size_t)blk * 64u + pos] = make_uint2(dst0, count);
            } else {
                lt[ls].start = 0u;
            }
            lt[ls].count = 1u;
        }
    }
}

// map_incremental's per-point decision (src/laserMapping.cpp:1448-1489): lio_incr.h; this kernel is the classification
// alone (scans beyond the single-launch compaction's size, lio_compact.hip)
__global__ __launch_bounds__(256) void map_incremental_kernel(IncrArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nq) return;
    float4 wp;
    a.cls[i] = map_incremental_classify_point(a, i, wp);
    a.world[i] = wp;
}
hipError_t launch_map_incremental(const IncrArgs& a, hipStream_t s) {
    if (a.nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(map_incremental_kernel, dim3((a.nq + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_classify_boundary(const BoundaryArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    uint32_t b = (a.n + 255u) / 256u; if (b > 4096u) b = 4096u;
    hipLaunchKernelGGL(classify_boundary_kernel, dim3(b), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_detect_shared_points(const BoundaryArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    uint32_t b = (a.n + 7u) / 8u; if (b > 16384u) b = 16384u;
    hipLaunchKernelGGL(detect_shared_points_kernel, dim3(b), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gather_dirty(const BoundaryArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    uint32_t b = (a.n + 255u) / 256u; if (b > 4096u) b = 4096u;
    hipLaunchKernelGGL(gather_dirty_kernel, dim3(b), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fold_sequential(const BoundaryArgs& a, const uint32_t* dlist_sorted, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(fold_sequential_kernel, dim3(1), dim3(64), 0, s, a, dlist_sorted);
    return hipGetLastError();
}
hipError_t launch_pend_keys(const float4* pend, uint32_t n, Grid grid, uint64_t* keys, uint32_t* order, const uint32_t* n_head,
                            uint32_t n_after, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t b = (n + 255u) / 256u; if (b > 4096u) b = 4096u;
    hipLaunchKernelGGL(pend_keys_kernel, dim3(b), dim3(256), 0, s, pend, n, grid, keys, order, n_head, n_after);
    return hipGetLastError();
}
hipError_t launch_append_points(float4* dst_base, const uint32_t* offset, const float4* src, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t b = (n + 255u) / 256u; if (b > 1024u) b = 1024u;
    hipLaunchKernelGGL(append_points_kernel, dim3(b), dim3(256), 0, s, dst_base, offset, src, n);
    return hipGetLastError();
}
hipError_t launch_cell_plan(const ApplyArgs& a, uint32_t max_groups, hipStream_t s) {
    if (max_groups == 0) return hipSuccess;
    uint32_t b = (max_groups + 1u + 63u) / 64u; if (b > 8192u) b = 8192u;
    hipLaunchKernelGGL(cell_plan_kernel, dim3(b), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_cell_apply(const ApplyArgs& a, uint32_t max_groups, hipStream_t s) {
    if (max_groups == 0) return hipSuccess;
    uint32_t b = (max_groups + 63u) / 64u; if (b > 8192u) b = 8192u;
    hipLaunchKernelGGL(cell_apply_kernel, dim3(b), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_voxel_keys(const float4* pts, uint32_t n, float ds, uint64_t* vkeys, uint32_t* order, uint32_t* zero2,
                             uint8_t* dirty, uint32_t* dlist, float4* out2n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t b = (n + 255u) / 256u; if (b > 4096u) b = 4096u;
    hipLaunchKernelGGL(voxel_keys_kernel, dim3(b), dim3(256), 0, s, pts, n, ds, vkeys, order, zero2, dirty, dlist, out2n);
    return hipGetLastError();
}
hipError_t launch_fold_groups(const FoldArgs& a, hipStream_t s) {
    if (a.n_groups_max == 0) return hipSuccess;
    uint32_t b = (a.n_groups_max + 7u) / 8u; if (b > 16384u) b = 16384u;
    hipLaunchKernelGGL(fold_groups_kernel, dim3(b), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace lio
