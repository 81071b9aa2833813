// callback_bench.cpp -- the ZERO-PATCH drop-in, timed: the reference's own call chain
//     kf.update_iterated_dyn_share_modified(..)  ->  h_dyn_share(x_, dyn_share) = h_share_model(state_ikfom&, dyn_share_datastruct&)
// (src/laserMapping.cpp:2380 -> esekfom.hpp:1639 -> laserMapping.cpp:1876) with include/lio_hip_adaptor.hpp behind the
// callback and a filter that really forms h_x^T h_x / h_x^T h from the returned m x 12 rows (esekfom.hpp:1790,1810;
// lio_eskf_update_iterated_dyn_share_callback).  Built as a small shared library (better_fastlio2_amd/build.py) that
// bench.py and the tests drive through ctypes; the node-side types are the stand-ins of tests/csrc/adaptor_stubs (Eigen and
// PCL are not in the image), laid out like the real ones (48-byte PointXYZINormal, column-major h_x).
//   patched != 0: the "two-line patch" of INTEGRATION.md section 3 -- the callback hands over lio_measure's reduced form
//                 (HTH / HTh) instead of rows; everything else (host-resident scan, synchronous callback) stays.
// What the adaptor does on its own in both forms: lio::scan_changed(ikdtree, G) starts the upload of the 48-byte records the
// moment the node has them (the first search pulls x, y, z out of the page-locked cloud, the calling thread packs the front of a
// large cloud beside it), and lio_measure puts the passes it expects next onto the stream ahead of their pose (the previous scan's
// converge pattern), so a callback mostly just mails a pose.
#include "use-ikfom.hpp"            // stand-ins: state_ikfom, esekfom::dyn_share_datastruct
#include <ikd-Tree/ikd_Tree.h>     // stand-ins: PointType, PointVector, StubCloud
#include <memory>
typedef std::shared_ptr<StubCloud> CloudPtr;
typedef esekfom::dyn_share_datastruct<double> DynShare;
#include <lio_hip_adaptor.hpp>

#include <chrono>
#include <cstring>

namespace {
// ---- the file-scope globals of src/laserMapping.cpp the measurement path touches (:44-116) -------------------------
bool extrinsic_est_en = false;
double res_mean_last = 0.05;
std::vector<float> res_last;
int effct_feat_num = 0, feats_down_size = 0;
std::unique_ptr<bool[]> point_selected_surf;
std::vector<PointVector> Nearest_Points;
CloudPtr feats_down_body = std::make_shared<StubCloud>(), feats_down_world = std::make_shared<StubCloud>(),
         normvec = std::make_shared<StubCloud>(), laserCloudOri = std::make_shared<StubCloud>(),
         corr_normvect = std::make_shared<StubCloud>();
lio::KdTreeFacade<PointType> ikdtree;                                   // :116
std::unique_ptr<lio::NodeGlobals<PointType, CloudPtr, std::vector<PointVector>>> G;
std::vector<CloudPtr> scans;                                            // what downSizeFilterSurf.filter left, one cloud per scan of the drive
DynShare dyn;                                                           // the filter's dyn_share (esekfom.hpp:1636)
bool patched_mode = false;
bool legacy_scan_changed = false;
int node_valid_num = 0, node_tree_size = 0;                              // featsFromMapNum / kdtree_size_st (:2343-2344)
double seconds_in_model = 0.0;                                           // inside the callback (library + adaptor), summed over the passes

void h_share_model(state_ikfom& s, DynShare& ekfom_data) { lio::h_share_model(ikdtree, s, ekfom_data, *G); }   // :1876

state_ikfom state_of(const double x[26]) {
    state_ikfom s{};
    s.pos = {{x[0], x[1], x[2]}}; s.rot = {{x[3], x[4], x[5], x[6]}};
    s.offset_R_L_I = {{x[7], x[8], x[9], x[10]}}; s.offset_T_L_I = {{x[11], x[12], x[13]}};
    return s;
}

// h_dyn_share of the filter = the node's callback, unmodified signature
struct ModelClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~ModelClock() { seconds_in_model += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
void model(void*, const double x26[26], lio_dyn_share* d) {
    ModelClock clock;
    state_ikfom s = state_of(x26);
    dyn.converge = d->converge != 0; dyn.valid = true;
    if (!patched_mode) {
        h_share_model(s, dyn);
        d->valid = dyn.valid ? 1 : 0; d->m = effct_feat_num;
        d->h_x = dyn.h_x.data(); d->h = dyn.h.data();
        return;
    }
    // two-line patch: same callback, same adaptor; the filter takes the reduced form (lio_out::HTH / ::HTh) instead of rows
    lio_out o;
    lio::h_share_model(ikdtree, s, dyn, *G, &o);
    d->valid = dyn.valid ? 1 : 0; d->m = effct_feat_num;
    if (!dyn.valid) return;
    if (o.m < 23) { d->h_x = dyn.h_x.data(); d->h = dyn.h.data(); return; }   // the dense-gain branch works on rows (esekfom.hpp:1720-1750)
    d->h_x = nullptr; d->h = nullptr;
    std::memcpy(d->HTH, o.HTH, sizeof(o.HTH)); std::memcpy(d->HTh, o.HTh, sizeof(o.HTh));
}
}  // namespace

extern "C" {
// the facade works on an existing context (the bench built the 5 M-point map in it)
int cbb_setup(lio_ctx* ctx, int max_scan_points) {
    ikdtree.pins.drop_all();                                            // (the clouds of a previous set-up are about to be freed; its context may be gone)
    ikdtree.adopt(ctx);
    res_last.assign((size_t)max_scan_points, 0.0f);
    point_selected_surf.reset(new bool[(size_t)max_scan_points]());
    normvec->points.resize((size_t)max_scan_points); laserCloudOri->points.resize((size_t)max_scan_points); corr_normvect->points.resize((size_t)max_scan_points);
    G.reset(new lio::NodeGlobals<PointType, CloudPtr, std::vector<PointVector>>{
        feats_down_body, feats_down_world, normvec, laserCloudOri, corr_normvect, Nearest_Points, point_selected_surf.get(), res_last.data(),
        feats_down_size, effct_feat_num, res_mean_last, extrinsic_est_en});
    scans.clear();
    return 0;
}
// one down-sampled scan of the drive as the node holds it: a PCL cloud of 48-byte points on the host
int cbb_add_scan(const float* xyz, int n, int stride_floats) {
    CloudPtr c = std::make_shared<StubCloud>();
    c->points.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        PointType p{};
        p.x = xyz[(size_t)i * stride_floats]; p.y = xyz[(size_t)i * stride_floats + 1]; p.z = xyz[(size_t)i * stride_floats + 2];
        c->points[(size_t)i] = p;
    }
    scans.push_back(c);
    return (int)scans.size() - 1;
}
// one scan step: feats_down_body = scan k (laserMapping.cpp:2322-2325), kf.change_x / change_P, the iterated update
int cbb_step(lio_eskf* kf, int k, const double* x26, const double* P, double R, int ext, int patched, lio_update_log* log) {
    if (k < 0 || k >= (int)scans.size()) return LIO_ERR_INVALID_ARG;
    try {
        feats_down_body = scans[(size_t)k];
        feats_down_size = (int)feats_down_body->points.size();
        extrinsic_est_en = ext != 0;
        patched_mode = patched != 0;
        ikdtree.reset_status();
        if (legacy_scan_changed) lio::scan_changed(ikdtree);        // round 3's form: the first callback uploads the cloud
        else lio::scan_changed(ikdtree, *G);                         // :2325 -- the upload starts here, nobody waits for it
        // what the node does between the voxel filter and the update (laserMapping.cpp:2343-2344): `int featsFromMapNum =
        // ikdtree.validnum(); kdtree_size_st = ikdtree.size();` -- both read host counters and must leave the pending pull alone
        node_valid_num = ikdtree.validnum();
        node_tree_size = ikdtree.size();
        int rc = lio_eskf_change_x(kf, x26);
        if (rc == LIO_OK) rc = lio_eskf_change_P(kf, P);
        if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_callback(kf, R, model, nullptr, log);
        return rc;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "callback_bench: %s\n", e.what());
        return LIO_ERR_HIP;
    }
}
int cbb_last_m(void) { return effct_feat_num; }
int cbb_last_validnum(void) { return node_valid_num; }
// status of the last library call the adaptor reported as failed during the last step (LIO_OK: none) -- in the default,
// non-throwing mode a failure costs the pass (ekfom_data.valid = false), not the node
int cbb_last_status(void) { return ikdtree.last_status; }
// knobs: 0 = page-locking of the node's buffers off (staging copies), 1 = exceptions instead of valid = false,
// 2 = the one-argument scan_changed (upload inside the first callback), 3 = size in bytes from which h_x / h are page-locked,
// 4 = packing threads of the assisted pull (1 = the calling thread alone)
void cbb_option(int which, int value) {
    if (which == 0) ikdtree.pin_host_buffers = value != 0;
    if (which == 1) ikdtree.throw_on_error = value != 0;
    if (which == 2) legacy_scan_changed = value != 0;
    if (which == 3) ikdtree.pin_rows_min_bytes = (size_t)value;
    if (which == 4) (void)ikdtree.set_pack_threads(value);
}
// seconds spent inside the measurement-model callback since the last call (the rest of a step is the filter's own dense algebra)
double cbb_take_model_seconds(void) { const double t = seconds_in_model; seconds_in_model = 0.0; return t; }
}
