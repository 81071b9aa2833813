// lio_mapadd.hip -- KD_TREE::Add_Points (include/ikd-Tree/ikd_Tree.cpp:414-489) on the GPU index.
//
// The reference inserts the points of a batch one after another; with down-sampling on, every
// insert (a) box-searches the down-sample voxel of the point, (b) picks, among the point itself
// and the points found, the one nearest the voxel centre (the new point wins ties, strict `<`
// :440), and (c) if the voxel held more than one point or the winner is the new point, deletes
// the voxel's points and adds the winner (:445-448).  Inserts interact only through their voxel,
// so the batch is grouped by voxel (stable sort: batch order inside a group, lio_gsort.hip) and every
// group is folded on its own -- the result equals the sequential one.
//
// Round 3: groups are named by the POSITION of their first record in the sorted batch (GroupView, lio_mapadd.h); kernels
// take one lane (or eight) per position and only heads act.  That removed the run-length kernels (two launches per
// sort), the dense group numbering and the per-group count arrays; the keys are compressed (KeyFrame).  Launches of a
// down-sampled batch:  voxel_keys -> sort (1 or 2) -> fold_plan (boundary marks + the speculative fold) -> compact_dirty ->
// fold_commit (the plans carried out + the sequential replay, side by side); of the index update:  pend_build -> sort -> cell_plan (with its scan) -> cell_apply.
#include "lio_device.h"
#include "lio_kernels.h"
#include "lio_mapadd.h"
#include "lio_math.h"
#include "lio_incr.h"

namespace lio {

__device__ __forceinline__ float voxel_floor(float v, float ds) { return floorf(v / ds); }   // :424

__device__ __forceinline__ uint32_t view_count(const GroupView& v) { return v.n_dev != nullptr ? min(v.n, *v.n_dev + v.n_add) : v.n; }
__device__ __forceinline__ bool view_is_head(const GroupView& v, uint32_t s) { return s == 0u || (v.skey[s - 1u] >> v.shift) != (v.skey[s] >> v.shift); }
// head position of the group with this key, or 0xffffffff
__device__ __forceinline__ uint32_t view_find(const GroupView& v, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v.skey[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && v.skey[lo] == key) ? lo : 0xffffffffu;
}
__device__ __forceinline__ bool finite3(float4 p) {
    return (__float_as_uint(p.x) & 0x7f800000u) != 0x7f800000u && (__float_as_uint(p.y) & 0x7f800000u) != 0x7f800000u &&
           (__float_as_uint(p.z) & 0x7f800000u) != 0x7f800000u;
}

__global__ __launch_bounds__(256) void voxel_keys_kernel(VoxelKeyArgs a) {
    __shared__ uint32_t s_hist[256];
    // first kernel of an Add_Points(downsample) batch: it also clears what the later kernels accumulate into
    // (fold counters, number of boundary points, the bounding box of the NEXT batch) and marks every output slot "nothing"
    if (blockIdx.x == 0 && threadIdx.x < 2) a.zero_counters[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 2) a.zero_counters[6] = 0u;
    // (a BatchDesc that sends the lists back to the host's own path -- flags -- leaves their box where that path reads it)
    const bool keep_box = a.desc != nullptr && a.desc->flags != 0u;
    if (blockIdx.x == 0 && threadIdx.x >= 8 && threadIdx.x < 14 && !keep_box) a.bbox[threadIdx.x - 8] = threadIdx.x < 11 ? BBOX_MIN_INIT : BBOX_MAX_INIT;
    s_hist[threadIdx.x] = 0u;
    __syncthreads();
    // (round 6) count and frame may still be on the device: a.n is then the launch bound, the survivor slots are laid out by it
    const uint32_t n_slots = a.n;
    uint32_t n = a.n;
    if (a.desc != nullptr) n = min(a.n, a.desc->n_add);
    const KeyFrame frame = a.desc != nullptr ? a.desc->vf : a.frame;
    const int kb = frame_bits(frame);
    const uint64_t junk = kb >= 64 ? ~0ull : (1ull << kb) - 1ull;      // a key no point of the frame has (lio_mapadd.h)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        a.dirty[i] = 0;
        a.dirty_pt[i] = 0;
        a.out2n[i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        a.out2n[n_slots + i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        const float4 p = a.pts[i];
        const int ix = voxel_coord(p.x, a.ds), iy = voxel_coord(p.y, a.ds), iz = voxel_coord(p.z, a.ds);
        // a point with a non-finite coordinate never enters the map through Add_Points(.., true) (its box holds nothing and
        // same_point(p, p) is false, :440-445): all of them form one group of their own, behind every real key
        const bool fin = finite3(p), held = frame_holds(frame, ix, iy, iz);
        if (fin && !held) atomicOr(a.flags, 1u);                   // the host's frame is wrong: reported, never silent
        const uint64_t key = (fin && held) ? frame_key(frame, ix, iy, iz) : junk;
        a.keys[i] = key;
        a.order[i] = i;
        if (a.hist != nullptr) atomicAdd(&s_hist[gsort_digit(key, kb)], 1u);    // (aggregating equal digits inside the wavefront first -- neighbouring
    }                                                                          //  points share their digit -- measured the same: 7.4 vs 7.1 us, round 6)
    if (a.hist != nullptr) {
        __syncthreads();
        const uint32_t v = s_hist[threadIdx.x];
        if (v != 0u) atomicAdd(a.hist + threadIdx.x, v);
    }
}

__device__ __forceinline__ bool same_point(float4 a, float4 b) {         // ikd_Tree.cpp:1365-1367, EPSS 1e-6
    return (double)fabsf(a.x - b.x) < 1e-6 && (double)fabsf(a.y - b.y) < 1e-6 && (double)fabsf(a.z - b.z) < 1e-6;
}

// Eight lanes per voxel group: the down-sample box (edge ds < cell edge) touches at most 2x2x2 grid cells, lane
// t = 4 dx + 2 dy + dz takes one of them -- probe, bucket scan and tomb-stoning of the eight cells run side by
// side instead of one after the other.  The reference visits the cells in (x, y, z) order and keeps the FIRST
// minimum (strict `<`), which is the minimum over (distance, t) here.
//
// The fold over the group's new points (ikd_Tree.cpp:433-449, one insert after the other) is a running minimum:
// once the voxel has been "touched" (Delete_by_range + Add_by_point happened) it holds exactly one point, cur, and a later
// point p replaces it iff !(d_cur < d_p).  Small groups run that loop as it stands; a group with many points -- an
// un-down-sampled 100 k-point scan puts hundreds of returns of a near wall into one 0.2 m voxel, and eight replicas'
// lists make it worse -- would hold its wavefront for a dependent load per point (836 points: 200 us, measured), so
// the whole wavefront folds it 64 points at a time: prefix-min of the distances (who sets a new record), prefix-max of
// the record positions (which point is cur when a later one is compared with it), the counter from two ballots.
constexpr uint32_t FOLD_WAVE_MIN = 16;          // group sizes from which the 64-lane fold takes over

struct FoldResult { bool touched; float x, y, z, d; uint32_t counter; };

// all 64 lanes, wave-uniform arguments and result; the group = the records from position gs on that carry `key`
__device__ __forceinline__ FoldResult fold_members_wave(const FoldArgs& a, uint32_t n, uint32_t gs, uint64_t key, uint32_t n_e, float d_e,
                                                        float ex, float ey, float ez, float mx, float my, float mz, int lane) {
    FoldResult r{false, 0.f, 0.f, 0.f, 0.f, 0u};
    const float inf = __uint_as_float(INF_BITS);
    for (uint32_t base = 0;; base += 64u) {
        const uint32_t s = gs + base + (uint32_t)lane;
        const bool has = s < n && a.view.skey[s] == key;           // members are contiguous: the first miss ends the group
        const bool more = __ballot(has) == ~0ull;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) p = a.new_pts[a.view.sord[s]];
        const float d = sqdist3(p.x, p.y, p.z, mx, my, mz);
        const bool selfsame = has && same_point(p, p);          // false for a point with a NaN coordinate: it never enters
        int first_after = 0;
        bool skip = false;
        if (!r.touched) {
            const bool best_is_p = !(n_e > 0 && d_e < d);
            const float4 best = best_is_p ? p : make_float4(ex, ey, ez, 0.0f);
            const bool tc = has && (n_e > 1 || same_point(p, best));
            const unsigned long long m = __ballot(tc);
            if (m == 0ull) {
                skip = true;
            } else {
                const int lt = __ffsll((long long)m) - 1;
                const bool bip = __shfl((int)best_is_p, lt, 64) != 0;
                const float px = __shfl(p.x, lt, 64), py = __shfl(p.y, lt, 64), pz = __shfl(p.z, lt, 64), pd = __shfl(d, lt, 64);
                r.touched = true; r.counter = 1u;
                r.x = bip ? px : ex; r.y = bip ? py : ey; r.z = bip ? pz : ez; r.d = bip ? pd : d_e;
                first_after = lt + 1;
            }
        }
        if (!skip) {
            const bool in = has && lane >= first_after;
            // who sets a new record: d <= the minimum of everything before it (cur's distance included)
            float pm = (in && selfsame) ? d : inf;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float u = __shfl_up(pm, o, 64);
                if (lane >= o) pm = fminf(pm, u);
            }
            float em = __shfl_up(pm, 1, 64);
            if (lane == 0) em = inf;
            em = fminf(em, r.d);
            const bool rec = in && selfsame && !(em < d);
            // cur as this lane sees it: the latest record before it, else the state carried in
            int pr = rec ? lane : -1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(pr, o, 64);
                if (lane >= o) pr = pr > u ? pr : u;
            }
            int er = __shfl_up(pr, 1, 64);
            if (lane == 0) er = -1;
            const int src = er < 0 ? 0 : er;
            float bx = __shfl(p.x, src, 64), by = __shfl(p.y, src, 64), bz = __shfl(p.z, src, 64);
            if (er < 0) { bx = r.x; by = r.y; bz = r.z; }
            const bool dup = in && !rec && same_point(p, make_float4(bx, by, bz, 0.0f));
            r.counter += (uint32_t)__popcll(__ballot(rec || dup));
            const unsigned long long rm = __ballot(rec);
            if (rm != 0ull) {
                const int last = 63 - __clzll((long long)rm);
                r.x = __shfl(p.x, last, 64); r.y = __shfl(p.y, last, 64); r.z = __shfl(p.z, last, 64); r.d = __shfl(d, last, 64);
            }
        }
        if (!more) break;
    }
    return r;
}

// ---- boundary points -----------------------------------------------------------------------------------
// the group at head position h is dirty: itself (the parallel fold skips it) and every one of its points (the replay list)
__device__ __forceinline__ void mark_group_dirty(const BoundaryArgs& a, uint32_t n, uint32_t h) {
    a.dirty[h] = 1;
    const uint64_t key = a.view.skey[h];
    for (uint32_t s = h; s < n && a.view.skey[s] == key; ++s) a.dirty_pt[a.view.sord[s]] = 1;
}
__device__ __forceinline__ uint32_t find_voxel_group(const BoundaryArgs& a, const KeyFrame& f, uint32_t n, int x, int y, int z) {
    if (!frame_holds(f, x, y, z)) return 0xffffffffu;             // no batch point lives outside the frame
    return view_find(a.view, n, frame_key(f, x, y, z));
}

// Add_Points(downsample) for the points of dirty groups, replayed in batch order by ONE wavefront with the
// reference's rule (ikd_Tree.cpp:422-449): Storage = live points inside the fp32 box (map points through the cell
// directory, lanes 0..7 one cell each; points added earlier in this pass from a list), nearest to the box centre
// wins (the new point on ties), delete-and-re-add when Storage held more than one point or the new point wins.
// What an iteration reads may have been written by the iteration before (tomb-stones, appended points): every iteration
// starts with an agent-scope acquire (the L1 forgets its lines) and ends with a release (its stores are in L2), so the
// reads in between are plain 16-byte loads, four in flight -- round 2 read every coordinate with an agent-scope atomic
// of its own, 17 us per replayed point (measured: seven points were most of the fold's 120 us).
// the replay list of a small batch made by the replaying wavefront itself (what compact_dirty_kernel makes for large ones): the
// batch indices whose dirty_pt mark is set, ascending.  4096 marks per round (four 16-byte loads per lane in flight); marked
// points are a handful per batch, so the slow part -- one lane's sixteen marks at a time -- hardly ever runs.
__device__ __forceinline__ uint32_t list_dirty_points(const BoundaryArgs& a, int lane) {
    const uint32_t n = view_count(a.view);
    const uint4* marks = reinterpret_cast<const uint4*>(a.dirty_pt);
    uint32_t cnt = 0;
    for (uint32_t base = 0; base < n; base += 4096u) {
        uint4 v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t i0 = base + u * 1024u + (uint32_t)lane * 16u;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (i0 < n) v[u] = marks[i0 >> 4];                  // (the buffer is a whole number of 16-byte words: hipMalloc)
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t i0 = base + u * 1024u + (uint32_t)lane * 16u;
            for (unsigned long long m = __ballot((v[u].x | v[u].y | v[u].z | v[u].w) != 0u); m != 0ull; m &= m - 1ull) {
                const int l = __builtin_ctzll(m);
                const uint32_t w[4] = {(uint32_t)__shfl((int)v[u].x, l, 64), (uint32_t)__shfl((int)v[u].y, l, 64),
                                       (uint32_t)__shfl((int)v[u].z, l, 64), (uint32_t)__shfl((int)v[u].w, l, 64)};
                const uint32_t j0 = (uint32_t)__shfl((int)i0, l, 64);
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) {
                    const bool set = ((w[j >> 2] >> (8u * (j & 3u))) & 0xffu) != 0u && j0 + j < n;   // (marks beyond n were never initialised)
                    if (set) { if (lane == 0) a.dlist[cnt] = j0 + j; ++cnt; }
                }
            }
        }
    }
    if (lane == 0) *a.n_dirty = cnt;
    // (the list is read back by this very wavefront: stores drained, L1 lines dropped)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return cnt;
}

__device__ __forceinline__ void fold_sequential(const BoundaryArgs& a, int lane, uint2* count_parts) {
    const uint32_t n_dirty = a.list_here ? list_dirty_points(a, lane) : *a.n_dirty;
    if (n_dirty == 0u) { if (lane == 0) count_parts[0] = make_uint2(0u, 0u); return; }
    uint32_t n_seq = 0, counter = 0, replaced = 0;
    // What one point of the replay writes (tomb-stones, the list of points added so far) is read by THIS wavefront at the
    // next point and by nobody else in the launch (a map point that a clean group could touch makes that group dirty): the
    // ordering needed is the wavefront's own, stores drained before the next point's loads -- not an agent-scope release
    // with its L2 write-back (that and the reload of the bucket for the deletes were most of ~9.5 us per replayed point).
    float4 p_next = a.new_pts[a.dlist[0]];
    for (uint32_t it = 0; it < n_dirty; ++it) {
        const float4 p = p_next;
        if (it + 1u < n_dirty) p_next = a.new_pts[a.dlist[it + 1u]];   // (the batch is not written here: in flight during this point)
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
        // ---- Storage, part 1: map points.  Lanes 0..7 look the (up to eight) grid cells of the box up; then EIGHT lanes
        // per cell read its bucket side by side (lane = cell * 8 + r takes slots r, r + 8, ...: the dirty groups of an 8-rank
        // batch sit where the map is dense, a lane per cell walked 40-point buckets four slots per round trip)
        int c0[3], c1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { c0[ax] = cell_coord(bmin[ax], a.grid.inv_cell); c1[ax] = cell_coord(bmax[ax], a.grid.inv_cell); }
        uint32_t bst = 0, bct = 0;
        {
            const int cx = c0[0] + ((lane >> 2) & 1), cy = c0[1] + ((lane >> 1) & 1), cz = c0[2] + (lane & 1);
            const bool mine = lane < 8 && cx <= c1[0] && cy <= c1[1] && cz <= c1[2];
            uint32_t fst = 0, fct = 0;
            if (mine) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, fst, fct);
            bst = (uint32_t)__shfl((int)fst, lane >> 3, 64);
            bct = (uint32_t)__shfl((int)fct, lane >> 3, 64);
        }
        const uint32_t r8 = (uint32_t)lane & 7u;
        uint32_t cnt = 0;
        float dbest = __uint_as_float(INF_BITS), bx = 0.f, by = 0.f, bz = 0.f;
        // tie order: map points first, by cell, then by slot (a lane meets its slots in ascending order and keeps the first
        // minimum); points added earlier in the replay after them, by list index
        uint32_t ord = 0x7fffffffu;
        unsigned long long in_mask = 0ull;                         // this lane's in-box live slots: bit k = slot r + 8 k (k < 64)
        for (uint32_t k0 = 0; r8 + 8u * k0 < bct; k0 += 2u) {
            float4 sp[2];
#pragma unroll
            for (uint32_t u = 0; u < 2u; ++u) {
                sp[u] = make_float4(0.f, 0.f, 0.f, 1.f);
                const uint32_t sl = r8 + 8u * (k0 + u);
                if (sl < bct) sp[u] = a.map_pts[bst + sl];
            }
#pragma unroll
            for (uint32_t u = 0; u < 2u; ++u) {
                if (sp[u].w != 0.0f || !inbox(sp[u].x, sp[u].y, sp[u].z)) continue;
                ++cnt;
                if (k0 + u < 64u) in_mask |= 1ull << (k0 + u);
                const float d = sqdist3(sp[u].x, sp[u].y, sp[u].z, mid[0], mid[1], mid[2]);
                if (d < dbest) { dbest = d; bx = sp[u].x; by = sp[u].y; bz = sp[u].z; ord = ((uint32_t)(lane >> 3) << 24) | (r8 + 8u * (k0 + u)); }
            }
        }
        // ---- Storage, part 2: points added earlier in this pass (all lanes, list order); they rank after the map's
        for (uint32_t j = (uint32_t)lane; j < n_seq; j += 64) {
            const float4 q = a.seq_out[j];
            if (q.w != 0.0f || !inbox(q.x, q.y, q.z)) continue;
            ++cnt;
            const float d = sqdist3(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
            if (d < dbest) { dbest = d; bx = q.x; by = q.y; bz = q.z; ord = 0x80000000u + j; }
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
                // (the slots the scan above found; a bucket beyond 64 slots is looked at again)
                for (unsigned long long m = in_mask; m != 0ull; m &= m - 1ull) {
                    a.map_pts[bst + r8 + 8u * (uint32_t)__builtin_ctzll(m)].w = TOMB_DOWNSAMPLE;
                    ++replaced;
                }
                for (uint32_t sl = r8 + 8u * 64u; sl < bct; sl += 8u) {          // (a bucket beyond 512 slots)
                    const float4 sp = a.map_pts[bst + sl];
                    if (sp.w == 0.0f && inbox(sp.x, sp.y, sp.z)) { a.map_pts[bst + sl].w = TOMB_DOWNSAMPLE; ++replaced; }
                }
                for (uint32_t j = (uint32_t)lane; j < n_seq; j += 64) {
                    const float4 q = a.seq_out[j];
                    if (q.w == 0.0f && inbox(q.x, q.y, q.z)) a.seq_out[j].w = 1.0f;
                }
            }
            if (lane == 0) a.seq_out[n_seq] = make_float4(best.x, best.y, best.z, 0.0f);   // Add_by_point of the winner
            ++n_seq;
        }
        // the appends / tomb-stones are out of this wavefront before the next point's loads are issued
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) replaced += __shfl_xor(replaced, o, 64);
    if (lane == 0) count_parts[0] = make_uint2(replaced, counter);
}

// live points of a bucket range, four loads in flight at a time (a lane's range is one contiguous run of 16-byte points)
template <uint32_t N = 4u, class F>
__device__ __forceinline__ void for_each_live_point(const float4* map_pts, uint32_t st, uint32_t ct, F&& f) {
    for (uint32_t i0 = 0; i0 < ct; i0 += N) {
        float4 sp[N];
#pragma unroll
        for (uint32_t u = 0; u < N; ++u) {
            sp[u] = make_float4(0.f, 0.f, 0.f, 1.f);
            if (i0 + u < ct) sp[u] = map_pts[st + i0 + u];
        }
#pragma unroll
        for (uint32_t u = 0; u < N; ++u)
            if (sp[u].w == 0.0f) f(st + i0 + u, sp[u]);
    }
}
// how many of them: only the tomb-stone lane of every slot is read, sixteen in flight (a bucket of a few dozen points is two
// round trips instead of one per four points)
__device__ __forceinline__ uint32_t count_live_points(const float4* map_pts, uint32_t st, uint32_t ct) {
    uint32_t live = 0;
    for (uint32_t i0 = 0; i0 < ct; i0 += 16u) {
        float w[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) {
            w[u] = 1.0f;
            if (i0 + u < ct) w[u] = map_pts[st + i0 + u].w;
        }
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) live += w[u] == 0.0f ? 1u : 0u;
    }
    return live;
}

// ---- the PLAN of a down-sampled batch, one launch ----------------------------------------------------------------------
// (1) one lane per batch point: is it outside its own fp32 box, or inside a neighbour's?  Then its group and the groups
// of the voxels it reaches are dirty (replayed point by point).  (2) eight lanes per group, one per grid cell the box can
// touch: the live map points inside the box are looked at ONCE -- a point that also lies inside a neighbouring voxel's box
// (fp32 boxes of neighbours can overlap by an ulp) couples the two groups: if the neighbour has points in this batch, both
// are dirty; what the fold needs of the others (how many, the nearest to the centre) stays in registers -- and the fold
// over the group's new points runs right here, SPECULATIVELY: nothing is written to the map.  What a clean group will
// do is parked (survivor, tmp_counter, box, the bucket ranges of its eight lanes); fold_commit_kernel carries it out once
// every dirty mark of the batch is known.  The replay only touches the voxels of dirty groups, and a map point inside two
// boxes makes both groups dirty, so nothing a clean group's plan was computed from changes before it is carried out.
__global__ __launch_bounds__(64) void fold_plan_kernel(FoldArgs a, BoundaryArgs ba) {
    const KeyFrame vframe = ba.frame_dev != nullptr ? *ba.frame_dev : ba.frame;   // (round 6: the frame may still be on the device, BatchDesc; a local: see cell_apply_kernel)
    const uint32_t n = view_count(a.view);
    const float ds = a.ds;
    const int lane = threadIdx.x, t = lane & 7;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = a.new_pts[i];
        if (!finite3(p)) continue;
        const float pc[3] = {p.x, p.y, p.z};
        int dir[3], iv[3];
        bool irregular = false;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            iv[ax] = voxel_coord(pc[ax], ds);
            const float kf = (float)iv[ax];
            const float bmin = kf * ds, bmax = bmin + ds;
            const float up_min = (kf + 1.0f) * ds;              // lower bound of the upper neighbour's box
            const float dn_max = (kf - 1.0f) * ds + ds;         // upper bound of the lower neighbour's box
            const bool up = !(pc[ax] < bmax) || pc[ax] >= up_min;
            const bool dn = !(bmin <= pc[ax]) || pc[ax] < dn_max;
            dir[ax] = up ? 1 : (dn ? -1 : 0);
            irregular = irregular || up || dn;
        }
        if (!irregular) continue;
        for (int m = 0; m < 8; ++m) {                              // own voxel (m = 0) and every combination of the reached sides
            bool use = true;
            int k[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const bool bit = (m >> ax) & 1;
                if (bit && dir[ax] == 0) use = false;
                k[ax] = iv[ax] + (bit ? dir[ax] : 0);
            }
            if (!use) continue;
            const uint32_t g = find_voxel_group(ba, vframe, n, k[0], k[1], k[2]);
            if (g != 0xffffffffu) mark_group_dirty(ba, n, g);
        }
    }
    const uint32_t spw = 8u;                                    // positions per wavefront
    for (uint32_t s0 = blockIdx.x * spw; s0 < a.view.n; s0 += gridDim.x * spw) {    // (wave-uniform: all 64 lanes iterate together)
        const uint32_t s = s0 + ((uint32_t)lane >> 3);
        const bool valid = s < n && view_is_head(a.view, s);
        const uint64_t key = valid ? a.view.skey[s] : 0ull;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) p0 = a.new_pts[a.view.sord[s]];
        const bool fin0 = valid && finite3(p0);
        // Box_of_Point and mid_point (:424-432): fp32 box, centre through double
        float bmin[3], bmax[3], mid[3], up_min[3], dn_max[3];
        int iv[3];
        const float pc[3] = {p0.x, p0.y, p0.z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            bmin[ax] = voxel_floor(pc[ax], ds) * ds;
            bmax[ax] = bmin[ax] + ds;
            mid[ax] = (float)((double)bmin[ax] + (double)(bmax[ax] - bmin[ax]) / 2.0);
            iv[ax] = voxel_coord(pc[ax], ds);
            const float kf = (float)iv[ax];
            up_min[ax] = (kf + 1.0f) * ds; dn_max[ax] = (kf - 1.0f) * ds + ds;
        }
        int c0[3], c1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            c0[ax] = cell_coord(bmin[ax], a.grid.inv_cell);
            c1[ax] = cell_coord(bmax[ax], a.grid.inv_cell);
        }
        const int cx = c0[0] + ((t >> 2) & 1), cy = c0[1] + ((t >> 1) & 1), cz = c0[2] + (t & 1);
        const bool mine = valid && cx <= c1[0] && cy <= c1[1] && cz <= c1[2];
        // ---- existing live points inside the box (Search_by_range :1015-1045), looked at once -------------
        uint32_t st = 0, ct = 0;
        if (mine) cell_find(a.l1table, a.l1mask, a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, st, ct);
        uint32_t n_e = 0;
        float d_e = __uint_as_float(INF_BITS);
        float4 e_best = make_float4(0.f, 0.f, 0.f, 0.f);
        for_each_live_point(a.map_pts, st, ct, [&](uint32_t, float4 sp) {
            if (!(bmin[0] <= sp.x && bmax[0] > sp.x && bmin[1] <= sp.y && bmax[1] > sp.y && bmin[2] <= sp.z && bmax[2] > sp.z)) return;
            ++n_e;
            const float d = sqdist3(sp.x, sp.y, sp.z, mid[0], mid[1], mid[2]);
            if (d < d_e) { d_e = d; e_best = sp; }
            if (!fin0) return;
            // inside a neighbouring voxel's box as well?
            const float xc[3] = {sp.x, sp.y, sp.z};
            int dir[3];
            bool shared = false;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                dir[ax] = xc[ax] >= up_min[ax] ? 1 : (xc[ax] < dn_max[ax] ? -1 : 0);
                shared = shared || dir[ax] != 0;
            }
            if (!shared) return;
            for (int m = 1; m < 8; ++m) {
                bool use = true;
                int k[3];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const bool bit = (m >> ax) & 1;
                    if (bit && dir[ax] == 0) use = false;
                    k[ax] = iv[ax] + (bit ? dir[ax] : 0);
                }
                if (!use) continue;
                const uint32_t h = find_voxel_group(ba, vframe, n, k[0], k[1], k[2]);
                if (h != 0xffffffffu) { mark_group_dirty(ba, n, h); mark_group_dirty(ba, n, s); }
            }
        });
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
        // ---- the fold over the group's new points (:433-449) ------------------------------------------------------
        bool touched = false;
        float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
        float d_cur = 0.0f;
        uint32_t counter = 0;
        // a large group: its FOLD_WAVE_MIN-th record still carries the key
        const bool big = valid && s + (FOLD_WAVE_MIN - 1u) < n && a.view.skey[s + (FOLD_WAVE_MIN - 1u)] == key;
        if (valid && !big) {
            // small group: point after point, identical in all eight lanes
            // (round 6: fetching the members three at a time -- keys and batch indices of three positions together, then their points --
            // computed the same plans on the host model and wrong tomb-stones on the device (test_gpu_knn_reproduces_reference_golden);
            // not chased further, the loop stays as it was)
            for (uint32_t j = 0; j < FOLD_WAVE_MIN - 1u; ++j) {
                if (j != 0u && (s + j >= n || a.view.skey[s + j] != key)) break;
                const float4 p = a.new_pts[a.view.sord[s + j]];
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
        }
        // large groups of this wavefront, one after the other, by all 64 lanes
        for (unsigned long long bm = __ballot(big && t == 0); bm != 0ull; bm &= bm - 1ull) {
            const int l0 = __ffsll((long long)bm) - 1;
            const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)key, l0, 64), khi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), l0, 64);
            const FoldResult fr = fold_members_wave(a, n, (uint32_t)__shfl((int)s, l0, 64), ((uint64_t)khi << 32) | (uint64_t)klo,
                                                    (uint32_t)__shfl((int)n_e, l0, 64), __shfl(d_e, l0, 64), __shfl(e_best.x, l0, 64),
                                                    __shfl(e_best.y, l0, 64), __shfl(e_best.z, l0, 64), __shfl(mid[0], l0, 64),
                                                    __shfl(mid[1], l0, 64), __shfl(mid[2], l0, 64), lane);
            if ((lane >> 3) == (l0 >> 3)) {
                touched = fr.touched; cur = make_float4(fr.x, fr.y, fr.z, 0.0f); d_cur = fr.d; counter = fr.counter;
            }
        }
        // ---- the plan of this group: the survivor and tmp_counter (0: the voxel stays as it is), the box, the lanes' buckets
        if (valid) {
            if (t == 0) {
                a.plan[2 * (size_t)s] = make_float4(cur.x, cur.y, cur.z, __uint_as_float(touched ? counter : 0u));
                a.plan[2 * (size_t)s + 1] = make_float4(bmin[0], bmin[1], bmin[2], 0.0f);
            }
            if (touched) a.ranges[(size_t)s * 8u + (uint32_t)t] = make_uint2(st, n_e > 0 ? ct : 0u);
        }
    }
}

// Carries the plans of the clean groups out (workgroups 1 ..): Delete_by_range of the box + Add_by_point of the survivor
// (:445-448) -- and, side by side (workgroup 0), replays the dirty points one after the other.  The two touch disjoint
// voxels and disjoint map points.
__global__ __launch_bounds__(64) void fold_commit_kernel(FoldArgs a, BoundaryArgs ba) {
    const int lane = threadIdx.x, t = lane & 7;
    if (blockIdx.x == 0) { fold_sequential(ba, lane, a.count_parts); return; }
    const uint32_t n = view_count(a.view);
    const float ds = a.ds;
    const uint32_t spw = 8u;
    uint32_t replaced_sum = 0, added_sum = 0;
    for (uint32_t s0 = (blockIdx.x - 1u) * spw; s0 < a.view.n; s0 += (gridDim.x - 1u) * spw) {
        const uint32_t s = s0 + ((uint32_t)lane >> 3);
        const bool valid = s < n && view_is_head(a.view, s) && a.dirty[s] == 0;
        float4 pl = make_float4(0.f, 0.f, 0.f, 0.f), bx = pl;
        if (valid) { pl = a.plan[2 * (size_t)s]; bx = a.plan[2 * (size_t)s + 1]; }
        const uint32_t counter = __float_as_uint(pl.w);
        const bool touched = valid && counter != 0u;
        uint32_t replaced = 0;
        if (touched) {
            const uint2 rg = a.ranges[(size_t)s * 8u + (uint32_t)t];
            const float bmin[3] = {bx.x, bx.y, bx.z};
            const float bmax[3] = {bmin[0] + ds, bmin[1] + ds, bmin[2] + ds};
            for_each_live_point(a.map_pts, rg.x, rg.y, [&](uint32_t i, float4 sp) {
                if (bmin[0] <= sp.x && bmax[0] > sp.x && bmin[1] <= sp.y && bmax[1] > sp.y && bmin[2] <= sp.z && bmax[2] > sp.z) {
                    // (a plain store: no other clean group's box holds this point -- it would have made both dirty -- and the
                    // replay stays inside the dirty voxels)
                    a.map_pts[i].w = TOMB_DOWNSAMPLE;
                    ++replaced;
                }
            });
            if (t == 0) a.out_pts[s] = make_float4(pl.x, pl.y, pl.z, 0.0f);      // (every slot starts as "nothing")
        }
        replaced_sum += replaced;
        added_sum += (touched && t == 0) ? counter : 0u;
    }
    // The two counters of the batch (replaced map points, tmp_counter) leave as ONE pair per workgroup; pend_build_kernel, the
    // next launch, adds the pairs up.  Atomics on one address serialise at the memory side of an eight-XCD part (~7 ns each,
    // measured): one per voxel group was 170 us of this kernel with eight replicas' lists, one per wavefront still 120 us.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        replaced_sum += (uint32_t)__shfl_xor((int)replaced_sum, o, 64);
        added_sum += (uint32_t)__shfl_xor((int)added_sum, o, 64);
    }
    if (lane == 0) a.count_parts[blockIdx.x] = make_uint2(replaced_sum, added_sum);
}

// ---- incremental index update -------------------------------------------------------------------------
// exclusive prefix over the tiles before tile b of one stamped count (lio_compact.hip's scheme), the tile's own count published first
__device__ __forceinline__ bool plan_tile_prefix(uint32_t mine, uint32_t b, unsigned long long* agg, uint32_t stamp, uint32_t& before) {
    __shared__ uint32_t s_sum[4];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    if (tid == 0) __hip_atomic_store(agg + b, (unsigned long long)mine | ((unsigned long long)stamp << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc = 0u;
    bool bad = false;
    for (uint32_t p = tid; p < b; p += 256u) {
        unsigned long long v;
        unsigned spins = 0;
        do { v = __hip_atomic_load(agg + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        while ((uint32_t)(v >> 32) != stamp && ++spins < (1u << 22));
        bad = bad || (uint32_t)(v >> 32) != stamp;
        acc += (uint32_t)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
    if ((tid & 63) == 0) s_sum[tid >> 6] = acc;
    if (bad) s_bad = 1u;
    __syncthreads();
    before = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    return s_bad == 0u;
}

// one lane per position of the cell-sorted pending list; heads: where the cell is now and how large its re-written bucket
// will be -- and, in the same launch, the exclusive scan of those sizes (tiles of 256 positions, stamped tile sums)
__global__ __launch_bounds__(256) void cell_plan_kernel(ApplyArgs a) {
    __shared__ uint32_t s_w[4];
    __shared__ uint8_t s_head[256];
    const KeyFrame frame = a.frame_dev != nullptr ? *a.frame_dev : a.frame;   // (round 6: the frame may still be on the device, BatchDesc)
    const uint32_t n = view_count(a.view);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the statistics the apply kernel accumulates start from zero (nothing reads them before it)
    if (blockIdx.x == 0 && tid < 3) a.stats[tid] = 0u;
    const uint32_t s = blockIdx.x * 256u + (uint32_t)tid;
    // One lane per pending point (they are sorted by cell).  EVERY point takes one slot of its cell's new bucket, the head of
    // a cell also the cell's live old points in front of it: the exclusive prefix of these sizes is then every point's own
    // destination (the head's: + live) -- no lane walks over its cell's new points, neither here nor in the apply kernel.
    const bool head = s < n && view_is_head(a.view, s);
    s_head[tid] = head ? 1 : 0;
    __syncthreads();
    uint32_t size = s < n ? 1u : 0u;
    if (head) {
        const uint64_t key = a.view.skey[s] >> 1;                  // (the low bit tells the two lists apart, not the cells)
        int cx, cy, cz;
        frame_decode(frame, key, cx, cy, cz);
        CellPlan pl{0u, 0u, 0xffffffffu, 0u, 0u, 0u};
        cell_find(a.ltable + a.levels.off[1], a.levels.mask[1], a.child, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, pl.old_start, pl.old_count, &pl.index);
        pl.found = pl.old_count != 0u ? 1u : 0u;                  // records are only ever written with count >= 1
        const uint32_t live = count_live_points(a.map_pts, pl.old_start, pl.old_count);
        // pending points of the cell = distance to the next head: inside the tile from the flags in LDS, beyond it from the keys
        uint32_t gc = 1u;
        while (tid + gc < 256u && s + gc < n && s_head[tid + gc] == 0) ++gc;
        if (tid + gc == 256u)
            while (s + gc < n && (a.view.skey[s + gc] >> 1) == key) ++gc;
        pl.live = live; pl.n_new = gc;
        a.plan[s] = pl;
        size += live;
    }
    uint32_t inc = size;                                         // inclusive scan inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    uint32_t before;
    const bool ok = plan_tile_prefix(tot, blockIdx.x, a.agg, a.stamp, before);
    if (s < a.view.n) a.new_off[s] = ok ? before + off + inc - size : 0u;
    if (!ok && tid == 0) atomicOr(a.err, 1u);
    if (blockIdx.x == gridDim.x - 1 && tid == 0) a.stats[3] = ok ? before + tot : 0xffffffffu;   // (a failed scan applies nothing)
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

__global__ __launch_bounds__(256) void cell_apply_kernel(ApplyArgs a) {
    __shared__ uint32_t s_stat[4][2];
    __shared__ uint32_t s_last;
    // (a LOCAL frame: assigning into the by-value argument struct makes the compiler keep a private copy of all ~700 bytes of it --
    // 720 bytes of scratch per lane, every access of a.levels through it: this kernel went from 12.5 to 33 us that way)
    const KeyFrame frame = a.frame_dev != nullptr ? *a.frame_dev : a.frame;
    const uint32_t n = view_count(a.view);
    // room check on the device (every lane sees the same total): either the whole batch is applied or nothing is
    const uint32_t total = a.stats[3];
    const bool room = (uint64_t)a.tail + (uint64_t)total <= (uint64_t)a.slots_cap;
    if (!room && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.stats + 2, APPLY_FLAG_NO_ROOM);
    uint32_t retired_sum = 0, new_cells = 0;                   // summed over the wavefront at the end: one atomic each, not one per cell
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; room && s < n; s += gridDim.x * blockDim.x) {
        // every pending point goes to its own slot (cell_plan_kernel's prefix): behind the cell's live old points, survivors
        // first, then plain inserts, batch order -- the order of the sort
        const bool head = view_is_head(a.view, s);
        const float4 np = a.pend[a.view.sord[s]];
        const uint32_t off_s = a.new_off[s];
        if (!head) { a.map_pts[a.tail + off_s] = make_float4(np.x, np.y, np.z, 0.0f); continue; }
        const CellPlan pl = a.plan[s];
        const uint32_t dst0 = a.tail + off_s;
        a.map_pts[dst0 + pl.live] = make_float4(np.x, np.y, np.z, 0.0f);
        uint32_t dst = dst0;
        uint32_t retired = 0;
        for_each_live_point<8u>(a.map_pts, pl.old_start, pl.old_count, [&](uint32_t i, float4 p) {   // live points keep their order; eight loads in flight
            a.map_pts[dst++] = p;
            a.map_pts[i].w = TOMB_MOVED;
            ++retired;
        });
        const uint64_t key = a.view.skey[s] >> 1;
        const uint32_t count = pl.live + pl.n_new;
        retired_sum += retired;
        int icx, icy, icz;
        frame_decode(frame, key, icx, icy, icz);
        const uint32_t cx = (uint32_t)icx, cy = (uint32_t)icy, cz = (uint32_t)icz;
        const uint32_t pos = ((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u);        // place inside the child block
        if (pl.found) {
            a.child[pl.index] = make_uint2(dst0, count);
            continue;
        }
        // ---- a cell the map did not have ----------------------------------------------------------
        ++new_cells;
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        retired_sum += (uint32_t)__shfl_xor((int)retired_sum, o, 64);
        new_cells += (uint32_t)__shfl_xor((int)new_cells, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_stat[threadIdx.x >> 6][0] = new_cells; s_stat[threadIdx.x >> 6][1] = retired_sum; }
    __syncthreads();
    if (threadIdx.x < 2) {                                        // (few workgroups: a few hundred atomics, not one per cell)
        const uint32_t v = s_stat[0][threadIdx.x] + s_stat[1][threadIdx.x] + s_stat[2][threadIdx.x] + s_stat[3][threadIdx.x];
        if (v != 0) atomicAdd(a.stats + threadIdx.x, v);
    }
    if (a.publish_dst == nullptr) return;
    // ---- the statistics leave with this kernel: the workgroup that finishes LAST copies them into host-mapped memory and publishes
    // the sequence number the host polls.  Every statistic is accumulated by device-scope atomics (performed at the memory side);
    // a workgroup takes its ticket only when its own have been acknowledged (vmcnt), so the last one reads them complete.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last == 0u) return;
    // NO system-scope fence / release here: it would write the whole L2 back first -- this kernel has just filled it with re-written
    // buckets (measured: the kernel 12.5 -> 34 us on an eight-rank batch).  The words go to host-mapped (uncached) memory as
    // system-scope stores, are acknowledged (vmcnt), and only then is the sequence number stored the same way: posted writes to
    // one destination do not pass each other.  The host reads nothing of this kernel but these words.
    if (threadIdx.x < 32)
        __hip_atomic_store(a.publish_dst + threadIdx.x, __hip_atomic_load(a.stats + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.publish_dst + 32, a.publish_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

hipError_t launch_voxel_keys(const VoxelKeyArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    uint32_t b = (a.n + 255u) / 256u; if (b > 1024u) b = 1024u;
    hipLaunchKernelGGL(voxel_keys_kernel, dim3(b), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fold_plan(const FoldArgs& fa, const BoundaryArgs& ba, hipStream_t s) {
    if (fa.view.n == 0) return hipSuccess;
    uint32_t b = (fa.view.n + 7u) / 8u; if (b > 16384u) b = 16384u;
    hipLaunchKernelGGL(fold_plan_kernel, dim3(b), dim3(64), 0, s, fa, ba);
    return hipGetLastError();
}
uint32_t fold_commit_workgroups(uint32_t n) {
    uint32_t b = (n + 7u) / 8u; if (b > 16384u) b = 16384u;
    return b + 1u;
}
hipError_t launch_fold_commit(const FoldArgs& fa, const BoundaryArgs& ba, hipStream_t s) {
    if (fa.view.n == 0) return hipSuccess;
    hipLaunchKernelGGL(fold_commit_kernel, dim3(fold_commit_workgroups(fa.view.n)), dim3(64), 0, s, fa, ba);
    return hipGetLastError();
}
hipError_t launch_cell_plan(const ApplyArgs& a, hipStream_t s) {
    if (a.view.n == 0) return hipSuccess;
    hipLaunchKernelGGL(cell_plan_kernel, dim3((a.view.n + 255u) / 256u), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_cell_apply(const ApplyArgs& a, hipStream_t s) {
    if (a.view.n == 0 && a.publish_dst == nullptr) return hipSuccess;
    uint32_t b = (a.view.n + 255u) / 256u; if (b > 512u) b = 512u;
    if (b == 0u) b = 1u;                         // (somebody waits for the statistics this kernel publishes)
    hipLaunchKernelGGL(cell_apply_kernel, dim3(b), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace lio
