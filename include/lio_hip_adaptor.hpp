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
//     ... lio::scan_changed(ikdtree);   after downSizeFilterSurf.filter(*feats_down_body)               (:2322-2325)
//     ... lio::sync_globals(ikdtree, G); at the top of map_incremental()                                (:1440)
#ifndef LIO_HIP_ADAPTOR_HPP
#define LIO_HIP_ADAPTOR_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "lio_hip.h"

namespace lio {

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
    void adopt(lio_ctx* ctx) { if (ctx_ && owns_) lio_destroy(ctx_); ctx_ = ctx; owns_ = false; rows_requested = false; }
    KdTreeFacade(const KdTreeFacade&) = delete;
    KdTreeFacade& operator=(const KdTreeFacade&) = delete;

    lio_ctx* ctx() {
        if (!ctx_) {
            lio_config cfg{device_, box_length_, 0.0f, max_scan_, max_map_, 0};
            const int rc = lio_create(&cfg, &ctx_);
            if (rc != LIO_OK) throw std::runtime_error("lio_create failed (a gfx950 GPU is required; there is no CPU path)");
        }
        return ctx_;
    }
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
    bool scan_dirty = true;       // feats_down_body changed since the last lio_scan_set
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
    void check(int rc, const char* what) const {
        if (rc != LIO_OK) throw std::runtime_error(std::string(what) + ": " + lio_last_error(ctx_));
    }
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

// call after feats_down_body has been refilled (downSizeFilterSurf.filter, laserMapping.cpp:2322-2325)
template <class PointT> inline void scan_changed(KdTreeFacade<PointT>& tree) { tree.scan_dirty = true; }

// h_share_model (src/laserMapping.cpp:1876-2004)
template <class PointT, class State, class DynShare, class G>
inline void h_share_model(KdTreeFacade<PointT>& tree, State& s, DynShare& ekfom_data, G& g) {
    lio_ctx* ctx = tree.ctx();
    if (!tree.rows_requested) {
        // the unmodified update_iterated_dyn_share_modified reads ekfom_data.h_x / .h: every pass stages them on the device,
        // compacted and column-major, so that lio_measure_rows below only moves m x 13 doubles
        if (lio_measure_want_rows(ctx, 1) != LIO_OK) throw std::runtime_error("lio_measure_want_rows failed");
        tree.rows_requested = true;
    }
    if (tree.scan_dirty) {                                             // once per scan, before its first pass
        const auto& pts = g.feats_down_body->points;
        const int rc = lio_scan_set(ctx, pts.size() ? &pts[0].x : nullptr, g.feats_down_size, (int)sizeof(PointT));
        if (rc != LIO_OK) throw std::runtime_error(std::string("lio_scan_set: ") + lio_last_error(ctx));
        tree.scan_dirty = false;
    }
    const double st[14] = {s.pos[0], s.pos[1], s.pos[2],
                           s.rot.x(), s.rot.y(), s.rot.z(), s.rot.w(),
                           s.offset_R_L_I.x(), s.offset_R_L_I.y(), s.offset_R_L_I.z(), s.offset_R_L_I.w(),
                           s.offset_T_L_I[0], s.offset_T_L_I[1], s.offset_T_L_I[2]};
    lio_out o;
    const int rc = lio_measure(ctx, st, ekfom_data.converge ? 1 : 0, g.extrinsic_est_en ? 1 : 0, &o);
    if (rc != LIO_OK) throw std::runtime_error(std::string("lio_measure: ") + lio_last_error(ctx));
    tree.globals_stale = true;
    g.effct_feat_num = o.m;                                            // :1942-1954
    if (!o.valid) {                                                    // :1956-1961
        ekfom_data.valid = false;
        std::fprintf(stderr, "No Effective Points! \n");
        return;
    }
    g.res_mean_last = o.res_mean;                                      // :1963
    // the unmodified update_iterated_dyn_share_modified reads h_x (m x 12, Eigen column-major) and h (:1967-2002)
    ekfom_data.h_x.resize(o.m, 12);
    ekfom_data.h.resize(o.m);
    const int rc2 = lio_measure_rows(ctx, ekfom_data.h_x.data(), ekfom_data.h.data(), o.m, /*col_major=*/1);
    if (rc2 != LIO_OK) throw std::runtime_error(std::string("lio_measure_rows: ") + lio_last_error(ctx));
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
    if (lio_get_scan_state(ctx, world.data(), sel.data(), nv.data()) != LIO_OK ||
        lio_get_neighbors(ctx, nbr.data(), nd.data(), cnt.data()) != LIO_OK)
        throw std::runtime_error(std::string("lio::sync_globals: ") + lio_last_error(ctx));
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
