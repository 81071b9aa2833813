// lio_hip_adaptor.hpp -- the C++ side of the drop-in: what a maintainer of Yixin-F/better_fastlio2 includes in
// src/laserMapping.cpp to run the per-scan measurement path on liblio_hip.so (include/lio_hip.h) while the rest of the
// node -- IKFoM, ROS plumbing, publishers, loop closure -- stays as it is.
//
// Two pieces, both header-only and free of Eigen / PCL / ROS includes (they are templates over the node's own types,
// so the same header type-checks against the stand-ins of tests/csrc/adaptor_stubs and against the real headers):
//
//   lio::KdTreeFacade<PointType>   replaces `KD_TREE<PointType> ikdtree;`                 (src/laserMapping.cpp:116)
//        same member names and argument lists as the public surface the node uses
//        (include/ikd-Tree/ikd_Tree.h:225-249): set_downsample_param, Build, reconstruct, Add_Points,
//        Delete_Point_Boxes, Nearest_Search, size, validnum, acquire_removed_points, flatten + PCL_Storage,
//        Root_Node (comparable with nullptr, laserMapping.cpp:2328).
//
//   lio::h_share_model(tree, s, ekfom_data, g)   the body of h_share_model             (src/laserMapping.cpp:1876-2004)
//        `g` (lio::NodeGlobals) holds REFERENCES to the file-scope globals the reference's callback reads and writes
//        (laserMapping.cpp:44,52,64-65,95,102-109): the measurement runs on the GPU, `effct_feat_num`, `res_mean_last`,
//        `ekfom_data.valid / h_x / h` are set exactly like the reference does; the bulky per-point globals
//        (feats_down_world, Nearest_Points, point_selected_surf, normvec, res_last, laserCloudOri, corr_normvect) are
//        refreshed LAZILY -- lio::sync_globals(tree, g) downloads them once per scan, only when the node is about to
//        read them (map_incremental :1451-1481, publish_effect_world); a node that also switches map_incremental to
//        lio_map_incremental never pays for that download.
//
// Usage in src/laserMapping.cpp (the complete diff is in INTEGRATION.md):
//     #include <lio_hip_adaptor.hpp>
//     lio::KdTreeFacade<PointType> ikdtree;                                        // was: KD_TREE<PointType> ikdtree;
//     static lio::NodeGlobals<PointType, PointCloudXYZI::Ptr> G{feats_down_body, feats_down_world, normvec, laserCloudOri,
//            corr_normvect, Nearest_Points, point_selected_surf, res_last, feats_down_size, effct_feat_num, res_mean_last,
//            extrinsic_est_en};
//     void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) { lio::h_share_model(ikdtree, s, d, G); }
//     ... lio::scan_changed(ikdtree, G); after downSizeFilterSurf.filter(*feats_down_body)              (:2322-2325)
//     ... lio::sync_globals(ikdtree, G); at the top of map_incremental()                                (:1440)
#ifndef LIO_HIP_ADAPTOR_HPP
#define LIO_HIP_ADAPTOR_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lio_hip.h"

namespace lio {

// Page-locked (hipHostRegister'ed) ranges of the node's memory: the cloud feats_down_body's points live in, ekfom_data.h_x,
// ekfom_data.h.  ensure() registers [p, p + bytes) unless exactly that range already is -- the node's buffers keep their
// address from scan to scan (PCL's VoxelGrid resizes the same vector, Eigen's resize() keeps an allocation whose size does not
// change) -- and first drops every registered range it overlaps (the memory was re-allocated: a stale registration must never be
// read through).  What an exact (address, size) match cannot see -- the node freed the buffer and got the same address back for a new
// one: the old PAGES stay registered -- is caught behind the transfer: rows by a marker (h_share_model), the cloud by the library's
// comparison of the first and the last point the GPU saw with the host's records (lio_measure, "stale_pulls").  A few ranges are kept (a node that alternates between clouds), the oldest goes first.  Registration costs
// tens of microseconds (78 us for 4.8 MB), so ranges below `min_bytes` are not worth it.  Contract: memory handed to ensure()
// stays allocated until it is handed to ensure() again with another size, or drop_all() is called.  The destructor does NOT
// unregister: it may run after the HIP runtime has gone.
struct HostPin {
    struct Range { char* p; size_t bytes; unsigned long long used; };
    std::vector<Range> ranges;
    unsigned long long clock = 0;
    size_t max_ranges = 16;
    // contained_ok: a range that starts where a registered one starts and ends inside it counts as registered (a matrix that
    // shrank inside its allocation).  ONLY for destinations whose arrival the caller verifies (h_share_model plants a marker
    // the copy must overwrite): an allocator may have unmapped and re-mapped the pages behind an old registration.
    bool ensure(void* q, size_t n, size_t min_bytes = 65536, bool contained_ok = false) {
        if (q == nullptr || n < min_bytes) return false;
        char* a = static_cast<char*>(q);
        for (size_t i = 0; i < ranges.size();) {
            Range& r = ranges[i];
            if (r.p == a && (r.bytes == n || (contained_ok && n < r.bytes))) { r.used = ++clock; return true; }
            if (a < r.p + r.bytes && r.p < a + n) { (void)lio_host_unregister(r.p); ranges.erase(ranges.begin() + (long)i); continue; }
            ++i;
        }
        if (ranges.size() >= max_ranges) {
            size_t oldest = 0;
            for (size_t i = 1; i < ranges.size(); ++i) if (ranges[i].used < ranges[oldest].used) oldest = i;
            (void)lio_host_unregister(ranges[oldest].p);
            ranges.erase(ranges.begin() + (long)oldest);
        }
        if (lio_host_register(q, n) != LIO_OK) return false;
        ranges.push_back(Range{a, n, ++clock});
        return true;
    }
    void drop_all() { for (const Range& r : ranges) (void)lio_host_unregister(r.p); ranges.clear(); }
};

// ikd_Tree.h:14 `enum delete_point_storage_set { NOT_RECORD, DELETE_POINTS_REC, MULTI_THREAD_REC }` -- the node passes
// NOT_RECORD to flatten (:2364); the facade accepts whatever enum/int arrives there.

template <class PointT>
class KdTreeFacade {
public:
    typedef std::vector<PointT> HostPoints;

    // `ikdtree.Root_Node == nullptr` (laserMapping.cpp:2328): true until the map holds a point
    struct RootProxy {
        const KdTreeFacade* t;
        bool operator==(std::nullptr_t) const { return t->size() == 0; }
        bool operator!=(std::nullptr_t) const { return t->size() != 0; }
    };
    RootProxy Root_Node{this};
    HostPoints PCL_Storage;                                           // ikd_Tree.h:247

    // KD_TREE(delete_param, balance_param, box_length) (ikd_Tree.h:226): the criteria of the re-balancing thread have no
    // counterpart (the index is rebuilt by sorting); box_length is the down-sample box.  The context is created on
    // first use, so that a file-scope `ikdtree` object costs nothing before the node has read its parameters.
    explicit KdTreeFacade(float /*delete_param*/ = 0.5f, float /*balance_param*/ = 0.6f, float box_length = 0.2f,
                          int device = 0, int max_scan_points = 100000 /* laserMapping.cpp:52,65 */,
                          long long max_map_points = 20000000)
        : box_length_(box_length), device_(device), max_scan_(max_scan_points), max_map_(max_map_points) {}
    ~KdTreeFacade() { if (ctx_ && owns_) lio_destroy(ctx_); }
    // a process that created its context elsewhere (e.g. next to other users of the same GPU) hands it over; the facade
    // then neither creates nor destroys one
    void adopt(lio_ctx* ctx) {
        if (ctx_ && owns_) lio_destroy(ctx_);
        ctx_ = ctx; owns_ = false;
        if (ctx_) (void)lio_measure_want_rows(ctx_, 0);               // a known starting point whatever the previous owner asked for
        rows_requested = false; scan_dirty = true; scan_failed = false; globals_stale = false;
    }
    KdTreeFacade(const KdTreeFacade&) = delete;
    KdTreeFacade& operator=(const KdTreeFacade&) = delete;

    // Without a GPU there is nothing to fall back to: a failed lio_create throws whatever throw_on_error says.
    lio_ctx* ctx() {
        if (!ctx_) {
            lio_config cfg{device_, box_length_, 0.0f, max_scan_, max_map_, 0};
            const int rc = lio_create(&cfg, &ctx_);
            if (rc != LIO_OK) throw std::runtime_error("lio_create failed (a gfx950 GPU is required; there is no CPU path)");
        }
        return ctx_;
    }
    // Error behaviour.  The reference's measurement path reports trouble the way ROS nodes do: ekfom_data.valid = false + a
    // warning (laserMapping.cpp:1956-1961) and the update skips the pass (esekfom.hpp:1641-1644); ikd-Tree printf()s.  Default
    // here: the same -- a failed library call prints `lio_hip: <what>: <lio_last_error>` on stderr, h_share_model answers
    // valid = false, map calls return 0 / leave their outputs empty, and the node stays alive.  throw_on_error = true turns
    // every failure into a std::runtime_error instead (tests, nodes that prefer to die loudly).
    bool throw_on_error = false;
    int last_status = LIO_OK;     // status of the last failed call (LIO_OK: none since reset_status())
    void reset_status() { last_status = LIO_OK; }
    bool report(int rc, const char* what) const {
        if (rc == LIO_OK) return true;
        const_cast<KdTreeFacade*>(this)->last_status = rc;
        const std::string msg = std::string(what) + ": " + (ctx_ ? lio_last_error(ctx_) : "no context");
        if (throw_on_error) throw std::runtime_error(msg);
        std::fprintf(stderr, "lio_hip: %s (status %d)\n", msg.c_str(), rc);
        return false;
    }
    // page-lock the node's host buffers (the cloud the scan is pulled from, h_x / h the rows are copied into) with
    // lio_host_register; false: every transfer goes through the library's own staging buffers
    bool pin_host_buffers = true;
    // Nodes that push un-down-sampled sweeps (~100 k points) through the callback: a second packing thread inside the library for
    // the assisted pull of the cloud (lio_scan_pack_threads; the reference's own h_share_model keeps MP_PROC_NUM OpenMP threads
    // busy at this point, laserMapping.cpp:1888).  1 = none (default).
    bool set_pack_threads(int n) { return report(lio_scan_pack_threads(ctx(), n), "lio_scan_pack_threads"); }
    size_t pin_rows_min_bytes = (size_t)2 << 20;   // row blocks (m x 13 doubles) from this size on are copied straight into page-locked h_x / h
    HostPin pins;
    long n_stale_pins = 0;        // row blocks that did not arrive through a registration and were fetched again (see h_share_model)
    long long n_stale_scan_pulls = 0;   // scans the library found to differ from the host's cloud after a pull through a registration (it
                                  // uploaded them again through its staging buffer: "stale_pulls"); every registration is dropped then
    bool check_pull = false;      // the scan was handed over through a registration: look at "stale_pulls" behind its first pass
    // forget every registration (before the node frees or re-allocates buffers it has handed over)
    void unpin_all() { if (ctx_) (void)lio_synchronize(ctx_); pins.drop_all(); }
    lio_ctx* ctx() const { return const_cast<KdTreeFacade*>(this)->ctx(); }

    void set_downsample_param(float box_length) {                      // ikd_Tree.cpp:186-189; laserMapping.cpp:2332
        box_length_ = box_length;
        if (ctx_) check(lio_map_set_downsample_param(ctx_, box_length), "set_downsample_param");
    }
    void Set_delete_criterion_param(float) {}                          // re-balancing criteria: nothing to balance
    void Set_balance_criterion_param(float) {}
    void InitializeKDTree(float = 0.5f, float = 0.7f, float box_length = 0.2f) { set_downsample_param(box_length); }

    template <class Points> void Build(const Points& pts) {            // ikd_Tree.cpp:353-364; laserMapping.cpp:2339
        check(lio_map_build(ctx(), xyz_of(pts), (long long)pts.size(), (int)sizeof(PointT)), "Build");
    }
    template <class Points> void reconstruct(const Points& pts) {      // ikd_Tree.cpp:1394-1405; laserMapping.cpp:656
        check(lio_map_reconstruct(ctx(), xyz_of(pts), (long long)pts.size(), (int)sizeof(PointT)), "reconstruct");
    }
    template <class Points> int Add_Points(Points& pts, bool downsample_on) {   // ikd_Tree.cpp:414-489; :1492-1493
        int n = 0;
        check(lio_map_add(ctx(), xyz_of(pts), (int)pts.size(), (int)sizeof(PointT), downsample_on ? 1 : 0, &n), "Add_Points");
        return n;
    }
    template <class Boxes> int Delete_Point_Boxes(Boxes& boxes) {      // ikd_Tree.cpp:536-556; laserMapping.cpp:1198
        std::vector<float> mm(6 * boxes.size());
        for (size_t i = 0; i < boxes.size(); ++i)
            for (int a = 0; a < 3; ++a) { mm[6 * i + a] = boxes[i].vertex_min[a]; mm[6 * i + 3 + a] = boxes[i].vertex_max[a]; }
        int n = 0;
        check(lio_map_delete_boxes(ctx(), mm.data(), (int)boxes.size(), &n), "Delete_Point_Boxes");
        return n;
    }
    // ikd_Tree.cpp:367-397.  One query per call is a GPU round trip -- the measurement path never calls this (it
    // searches the whole scan inside lio_measure); kept for the node's other, occasional callers.  k_nearest <= 5.
    template <class Points> void Nearest_Search(const PointT& point, int k_nearest, Points& Nearest_Points,
                                                std::vector<float>& Point_Distance, double max_dist = INFINITY) {
        float xyz[15], d[5];
        int cnt = 0;
        const float q[3] = {point.x, point.y, point.z};
        check(lio_map_nearest_search(ctx(), q, 1, 12, xyz, d, &cnt), "Nearest_Search");
        Nearest_Points.clear(); Point_Distance.clear();
        for (int k = 0; k < cnt && k < k_nearest; ++k) {
            if ((double)d[k] > max_dist * max_dist) break;             // the reference prunes with max_dist (ikd_Tree.cpp:871)
            PointT p = point;
            p.x = xyz[3 * k]; p.y = xyz[3 * k + 1]; p.z = xyz[3 * k + 2];
            Nearest_Points.push_back(p);
            Point_Distance.push_back(d[k]);
        }
    }
    int size() const { return ctx_ ? (int)lio_map_size(ctx_) : 0; }              // ikd_Tree.cpp:79
    int validnum() const { return ctx_ ? (int)lio_map_validnum(ctx_) : 0; }      // ikd_Tree.cpp:129
    template <class Points> void acquire_removed_points(Points& removed_points) {   // ikd_Tree.cpp:559-571; :1127
        long long n = 0;
        check(lio_map_acquire_removed_points(ctx(), nullptr, 0, &n), "acquire_removed_points");
        if (n == 0) return;
        std::vector<float> xyz(3 * (size_t)n);
        check(lio_map_acquire_removed_points(ctx(), xyz.data(), n, &n), "acquire_removed_points");
        append_xyz(removed_points, xyz, n);
    }
    // `ikdtree.flatten(ikdtree.Root_Node, ikdtree.PCL_Storage, NOT_RECORD)` (laserMapping.cpp:2364; ikd_Tree.cpp:1326-1352)
    template <class Points, class StorageType> void flatten(const RootProxy&, Points& Storage, StorageType) { flatten(Storage); }
    template <class Points> void flatten(Points& Storage) {
        long long n = validnum();
        std::vector<float> xyz(3 * (size_t)(n > 0 ? n : 1));
        check(lio_map_flatten(ctx(), xyz.data(), n, &n), "flatten");
        append_xyz(Storage, xyz, n);
    }

    // ---- per-scan state of the adaptor (not part of the reference's class) ---------------------------------------
    bool scan_dirty = true;       // feats_down_body changed since the last upload
    bool scan_failed = false;     // ... and the library refused it (reported once; every pass of this scan answers valid = false)
    bool rows_requested = false;  // lio_measure_want_rows(ctx, 1) has been issued (the unmodified IKFoM reads h_x / h)
    bool globals_stale = false;   // a measurement pass ran since the last sync_globals

private:
    template <class Points> static const float* xyz_of(const Points& pts) {
        return pts.size() ? &pts[0].x : nullptr;
    }
    template <class Points> static void append_xyz(Points& out, const std::vector<float>& xyz, long long n) {
        for (long long i = 0; i < n; ++i) {
            PointT p{};
            p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2];
            out.push_back(p);
        }
    }
    void check(int rc, const char* what) const { (void)report(rc, what); }
    lio_ctx* ctx_ = nullptr;
    bool owns_ = true;
    float box_length_;
    int device_, max_scan_;
    long long max_map_;
};

// References to the file-scope globals of src/laserMapping.cpp the callback communicates through.
// CloudPtr: pcl::PointCloud<PointType>::Ptr (anything with ->points, a std::vector-like of PointT).
template <class PointT, class CloudPtr, class NearestVec = std::vector<std::vector<PointT>>>
struct NodeGlobals {
    CloudPtr& feats_down_body;        // :102   in
    CloudPtr& feats_down_world;       // :103   out (lazy)
    CloudPtr& normvec;                // :104   out (lazy): n^ in x,y,z, pd2 in intensity (:1933-1936)
    CloudPtr& laserCloudOri;          // :105   out (lazy): selected body points, compacted (:1948)
    CloudPtr& corr_normvect;          // :106   out (lazy): their normvec, compacted (:1949)
    NearestVec& Nearest_Points;       // :95    out (lazy): consumed by map_incremental (:1453-1481)
    bool* point_selected_surf;        // :65    out (lazy)
    float* res_last;                  // :52    out (lazy): |pd2| of selected points (:1937)
    int& feats_down_size;             // :64    in
    int& effct_feat_num;              // :63    out
    double& res_mean_last;            // :48    out
    bool& extrinsic_est_en;           // :44    in
};

// feats_down_body -> the GPU, without waiting for it to arrive.  Page-locks the cloud's buffer once (its whole capacity, so that
// a scan of another size inside the same allocation needs no new registration) and lets a kernel pull x, y, z out of the
// 48-byte records where they lie; with pin_host_buffers off (or a registration that fails) the records are packed into the
// library's staging buffer instead.  Returns false after reporting (KdTreeFacade::report) when the library refuses the scan.
template <class PointT, class G>
inline bool upload_scan(KdTreeFacade<PointT>& tree, G& g) {
    lio_ctx* ctx = tree.ctx();
    auto& pts = g.feats_down_body->points;
    const int n = g.feats_down_size;
    tree.check_pull = false;
    if (tree.pin_host_buffers && n > 0) tree.check_pull = tree.pins.ensure(&pts[0], pts.capacity() * sizeof(PointT));
    const bool ok = tree.report(lio_scan_set_async(ctx, n > 0 ? &pts[0].x : nullptr, n, (int)sizeof(PointT)), "lio_scan_set_async");
    tree.scan_dirty = !ok;
    tree.scan_failed = !ok;
    return ok;
}
// call after feats_down_body has been refilled (downSizeFilterSurf.filter, laserMapping.cpp:2322-2325).  With the globals at
// hand the upload starts HERE, beside the node's own work between the voxel filter and the update (:2326-2379: validnum(),
// size(), the resizes, the log line); the one-argument form only marks the scan and the first callback uploads it.
template <class PointT> inline void scan_changed(KdTreeFacade<PointT>& tree) { tree.scan_dirty = true; tree.scan_failed = false; }
template <class PointT, class G> inline void scan_changed(KdTreeFacade<PointT>& tree, G& g) {
    tree.scan_dirty = true; tree.scan_failed = false;
    (void)upload_scan(tree, g);
}

// h_share_model (src/laserMapping.cpp:1876-2004).  reduced != nullptr: the "two-line patch" of INTEGRATION.md section 3 -- the
// caller's update_iterated_dyn_share_modified takes h_x^T h_x / h_x^T h from *reduced (lio_out::HTH, ::HTh) instead of forming
// them from ekfom_data.h_x / .h, which then stay untouched (unless m < 23: the dense-gain branch needs the rows).
template <class PointT, class State, class DynShare, class G>
inline void h_share_model(KdTreeFacade<PointT>& tree, State& s, DynShare& ekfom_data, G& g, lio_out* reduced = nullptr) {
    lio_ctx* ctx = tree.ctx();
    // any failure below: what the reference does when it has no effective points (:1956-1961) -- the pass is invalid, the update
    // skips it (esekfom.hpp:1641-1644), the node lives
    auto invalid = [&]() { ekfom_data.valid = false; g.effct_feat_num = 0; };
    const bool want_rows = reduced == nullptr;
    if (tree.rows_requested != want_rows) {
        // the unmodified update_iterated_dyn_share_modified reads ekfom_data.h_x / .h: every pass stages them on the device,
        // compacted and column-major, so that lio_measure_rows below only moves m x 13 doubles
        if (!tree.report(lio_measure_want_rows(ctx, want_rows ? 1 : 0), "lio_measure_want_rows")) { invalid(); return; }
        tree.rows_requested = want_rows;
    }
    if (tree.scan_failed) { invalid(); return; }                       // (reported when it happened)
    if (tree.scan_dirty && !upload_scan(tree, g)) { invalid(); return; }   // once per scan, before its first pass
    const double st[14] = {s.pos[0], s.pos[1], s.pos[2],
                           s.rot.x(), s.rot.y(), s.rot.z(), s.rot.w(),
                           s.offset_R_L_I.x(), s.offset_R_L_I.y(), s.offset_R_L_I.z(), s.offset_R_L_I.w(),
                           s.offset_T_L_I[0], s.offset_T_L_I[1], s.offset_T_L_I[2]};
    lio_out local;
    lio_out& o = reduced ? *reduced : local;
    if (!tree.report(lio_measure(ctx, st, ekfom_data.converge ? 1 : 0, g.extrinsic_est_en ? 1 : 0, &o), "lio_measure")) { invalid(); return; }
    tree.globals_stale = true;
    if (tree.check_pull) {
        // The first pass of a scan that a kernel pulled out of the registered cloud: the library compares the first and the last point
        // the GPU saw with the host's records (a buffer freed and re-allocated at the same address keeps its OLD pages registered) and,
        // on a difference, uploads the scan again through its staging buffer and repeats the pass -- the results above are right either
        // way; here the stale registrations go, so that the next scan is registered afresh.
        tree.check_pull = false;
        long long stale = 0;
        if (lio_debug_get(ctx, "stale_pulls", &stale) == LIO_OK && stale != tree.n_stale_scan_pulls) { tree.n_stale_scan_pulls = stale; tree.unpin_all(); }
    }
    g.effct_feat_num = o.m;                                            // :1942-1954
    if (!o.valid) {                                                    // :1956-1961
        ekfom_data.valid = false;
        std::fprintf(stderr, "No Effective Points! \n");
        return;
    }
    g.res_mean_last = o.res_mean;                                      // :1963
    if (reduced != nullptr && o.m >= 23) return;                      // the patched filter needs nothing else
    // the unmodified update_iterated_dyn_share_modified reads h_x (m x 12, Eigen column-major) and h (:1967-2002)
    ekfom_data.h_x.resize(o.m, 12);
    ekfom_data.h.resize(o.m);
    double* hx = ekfom_data.h_x.data();
    double* hh = ekfom_data.h.data();
    // Large row blocks (100 k points: 9.6 MB per pass) are written by the copy engine straight into page-locked h_x / h.  m
    // changes from pass to pass and Eigen re-allocates on every change of size, usually at the same address: a block that
    // starts where a registered one starts and fits inside it is taken as registered (a registration costs ~150 us for 9.6
    // MB) -- and because an allocator may have re-mapped those pages since, the arrival is VERIFIED: the last element of both
    // blocks carries a marker no row can hold until the copy overwrites it; a marker that survives means the registration
    // was stale: everything is unpinned and the rows come again through the library's own staging buffer.
    bool pinned = false;
    if (tree.pin_host_buffers && (size_t)o.m * 13 * sizeof(double) >= tree.pin_rows_min_bytes)
        pinned = tree.pins.ensure(hx, (size_t)o.m * 12 * sizeof(double), 65536, true) && tree.pins.ensure(hh, (size_t)o.m * sizeof(double), 4096, true);
    const unsigned long long MARK = 0x7ff8dead5ca1ab1eull;              // a quiet NaN with a payload of its own
    auto bits_of = [](const double& v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; };
    if (pinned) { std::memcpy(&hx[(size_t)o.m * 12 - 1], &MARK, 8); std::memcpy(&hh[(size_t)o.m - 1], &MARK, 8); }
    if (!tree.report(lio_measure_rows(ctx, hx, hh, o.m, /*col_major=*/1), "lio_measure_rows")) { invalid(); return; }
    if (pinned && (bits_of(hx[(size_t)o.m * 12 - 1]) == MARK || bits_of(hh[(size_t)o.m - 1]) == MARK)) {
        ++tree.n_stale_pins;
        tree.unpin_all();
        if (!tree.report(lio_measure_rows(ctx, hx, hh, o.m, /*col_major=*/1), "lio_measure_rows")) { invalid(); return; }
    }
}

// Brings the reference's per-point globals up to date with the last measurement pass (no-op when they are).
template <class PointT, class G>
inline void sync_globals(KdTreeFacade<PointT>& tree, G& g) {
    if (!tree.globals_stale) return;
    lio_ctx* ctx = tree.ctx();
    const int n = lio_scan_size(ctx);
    std::vector<float> world(3 * (size_t)n + 3), nv(4 * (size_t)n + 4), nbr(15 * (size_t)n + 15), nd(5 * (size_t)n + 5);
    std::vector<uint8_t> sel((size_t)n + 1);
    std::vector<int> cnt((size_t)n + 1);
    if (!tree.report(lio_get_scan_state(ctx, world.data(), sel.data(), nv.data()), "lio::sync_globals (lio_get_scan_state)") ||
        !tree.report(lio_get_neighbors(ctx, nbr.data(), nd.data(), cnt.data()), "lio::sync_globals (lio_get_neighbors)"))
        return;                                                        // the globals keep the previous scan's values
    auto& body = g.feats_down_body->points;
    auto& wpts = g.feats_down_world->points;
    if ((int)wpts.size() < n) wpts.resize(n);
    if ((int)g.Nearest_Points.size() < n) g.Nearest_Points.resize(n);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        PointT pw = body[i];                                           // intensity etc. carried over (:1899)
        pw.x = world[3 * i]; pw.y = world[3 * i + 1]; pw.z = world[3 * i + 2];
        wpts[i] = pw;
        auto& near = g.Nearest_Points[i];
        near.clear();
        for (int k = 0; k < cnt[i]; ++k) {
            PointT p{};
            p.x = nbr[15 * i + 3 * k]; p.y = nbr[15 * i + 3 * k + 1]; p.z = nbr[15 * i + 3 * k + 2];
            near.push_back(p);
        }
        g.point_selected_surf[i] = sel[i] != 0;
        if (sel[i]) {
            PointT q{};
            q.x = nv[4 * i]; q.y = nv[4 * i + 1]; q.z = nv[4 * i + 2]; q.intensity = nv[4 * i + 3];   // :1933-1936
            g.normvec->points[i] = q;
            g.res_last[i] = std::fabs(nv[4 * i + 3]);                                                 // :1937
            g.laserCloudOri->points[m] = body[i];                                                     // :1948
            g.corr_normvect->points[m] = q;                                                           // :1949
            ++m;
        }
    }
    tree.globals_stale = false;
}

}  // namespace lio
#endif  // LIO_HIP_ADAPTOR_HPP
