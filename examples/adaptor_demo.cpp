// adaptor_demo.cpp -- include/lio_hip_adaptor.hpp used the way src/laserMapping.cpp would use it: the reference's
// globals, its call sites (line numbers of the reference in the comments) and its own host-side map_incremental, with
// stand-in types for Eigen / PCL (tests/csrc/adaptor_stubs) -- or the real ones when the node's headers are there
// (-DLIO_ADAPTOR_REAL_HEADERS with Eigen, PCL and the reference's include/ on the include path).
//
//   g++ -std=c++17 -O2 -Iinclude -Itests/csrc/adaptor_stubs examples/adaptor_demo.cpp -o adaptor_demo
//       -Lbetter_fastlio2_amd/lib -llio_hip -Wl,-rpath,$PWD/better_fastlio2_amd/lib
#if defined(LIO_ADAPTOR_REAL_HEADERS) && __has_include(<Eigen/Core>) && __has_include(<pcl/point_types.h>)
#include "use-ikfom.hpp"
#include <ikd-Tree/ikd_Tree.h>
#include <pcl/point_cloud.h>
typedef pcl::PointXYZINormal PointType;
typedef pcl::PointCloud<PointType>::Ptr CloudPtr;
typedef esekfom::dyn_share_datastruct<double> DynShare;
#define NEW_CLOUD(n) CloudPtr(new pcl::PointCloud<PointType>(n, 1))
#else
#include "use-ikfom.hpp"            // stand-ins: state_ikfom, esekfom::dyn_share_datastruct
#include <ikd-Tree/ikd_Tree.h>     // stand-ins: PointType, PointVector, BoxPointType, StubCloud
#include <memory>
typedef std::shared_ptr<StubCloud> CloudPtr;
typedef esekfom::dyn_share_datastruct<double> DynShare;
enum delete_point_storage_set { NOT_RECORD, DELETE_POINTS_REC, MULTI_THREAD_REC };   // ikd_Tree.h:14
static CloudPtr NEW_CLOUD(size_t n) { auto c = std::make_shared<StubCloud>(); c->points.resize(n); return c; }
#endif
#include <lio_hip_adaptor.hpp>

#include <cmath>
#include <cstdio>
#include <random>

// ---- the file-scope globals of src/laserMapping.cpp the measurement path touches (:44-116) -------------------------
bool extrinsic_est_en = false;
double res_mean_last = 0.05, filter_size_map_min = 0.2;
float res_last[100000] = {0.0};
int effct_feat_num = 0, feats_down_size = 0;
bool point_selected_surf[100000] = {0};
bool flg_EKF_inited = true;
std::vector<PointVector> Nearest_Points;
CloudPtr feats_down_body = NEW_CLOUD(0), feats_down_world = NEW_CLOUD(0), normvec = NEW_CLOUD(100000),
         laserCloudOri = NEW_CLOUD(100000), corr_normvect = NEW_CLOUD(100000), featsFromMap = NEW_CLOUD(0);
lio::KdTreeFacade<PointType> ikdtree;                                   // :116  was KD_TREE<PointType> ikdtree;
static lio::NodeGlobals<PointType, CloudPtr, std::vector<PointVector>> G{
    feats_down_body, feats_down_world, normvec, laserCloudOri, corr_normvect, Nearest_Points, point_selected_surf, res_last,
    feats_down_size, effct_feat_num, res_mean_last, extrinsic_est_en};

void h_share_model(state_ikfom& s, DynShare& ekfom_data) { lio::h_share_model(ikdtree, s, ekfom_data, G); }   // :1876

static float calc_dist(const PointType& a, const PointType& b) {      // common_lib.h calc_dist
    return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
}
// map_incremental (:1440-1496) left on the host, exactly the reference's loop: it reads Nearest_Points and
// feats_down_world, which the adaptor refreshes on demand
static int map_incremental() {
    lio::sync_globals(ikdtree, G);
    PointVector PointToAdd, PointNoNeedDownsample;
    for (int i = 0; i < feats_down_size; i++) {
        const PointType& pw = feats_down_world->points[i];
        if (!Nearest_Points[i].empty() && flg_EKF_inited) {
            const PointVector& points_near = Nearest_Points[i];
            bool need_add = true;
            PointType mid_point{};
            mid_point.x = std::floor(pw.x / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min;
            mid_point.y = std::floor(pw.y / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min;
            mid_point.z = std::floor(pw.z / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min;
            const float dist = calc_dist(pw, mid_point);
            if (std::fabs(points_near[0].x - mid_point.x) > 0.5 * filter_size_map_min && std::fabs(points_near[0].y - mid_point.y) > 0.5 * filter_size_map_min &&
                std::fabs(points_near[0].z - mid_point.z) > 0.5 * filter_size_map_min) { PointNoNeedDownsample.push_back(pw); continue; }
            for (int k = 0; k < 5; k++) {
                if (points_near.size() < 5) break;
                if (calc_dist(points_near[k], mid_point) < dist) { need_add = false; break; }
            }
            if (need_add) PointToAdd.push_back(pw);
        } else {
            PointToAdd.push_back(pw);
        }
    }
    const int add_point_size = ikdtree.Add_Points(PointToAdd, true);                     // :1492
    ikdtree.Add_Points(PointNoNeedDownsample, false);                                    // :1493
    std::printf("map_incremental: %zu + %zu points offered, Add_Points returned %d\n", PointToAdd.size(), PointNoNeedDownsample.size(), add_point_size);
    return (int)(PointToAdd.size() + PointNoNeedDownsample.size());
}

int main() {
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> jit(-0.08f, 0.08f), zn(-0.02f, 0.02f);
    // a ground patch and a wall, one point per 0.2 m voxel
    PointVector world;
    for (int i = 0; i < 250; ++i)
        for (int j = 0; j < 250; ++j) { PointType p{}; p.x = -25.f + 0.2f * i + jit(rng); p.y = -25.f + 0.2f * j + jit(rng); p.z = zn(rng); world.push_back(p); }
    for (int i = 0; i < 200; ++i)
        for (int k = 0; k < 15; ++k) { PointType p{}; p.x = 8.0f + zn(rng); p.y = -20.f + 0.2f * i + jit(rng); p.z = 0.1f + 0.2f * k + jit(rng); world.push_back(p); }
    try {
        if (ikdtree.Root_Node == nullptr) {                                               // :2328
            ikdtree.set_downsample_param((float)filter_size_map_min);                     // :2332
            PointVector half;                                                             // every second point: the scan adds the rest
            for (size_t i = 0; i < world.size(); i += 2) half.push_back(world[i]);
            ikdtree.Build(half);                                                          // :2339
        }
        if (ikdtree.Root_Node == nullptr) { std::printf("map is empty after Build\n"); return 1; }
        std::printf("validnum %d size %d\n", ikdtree.validnum(), ikdtree.size());        // :2343-2344

        // feats_down_body: map surfaces seen from a LiDAR 1.7 m above the origin (identity extrinsics)
        feats_down_body->points.clear();
        for (size_t i = 1; i < world.size(); i += 7) {
            PointType p = world[i];
            if (std::fabs(p.x) > 20 || std::fabs(p.y) > 20) continue;
            p.z -= 1.7f; p.intensity = (float)(i % 255);
            feats_down_body->points.push_back(p);
        }
        feats_down_size = (int)feats_down_body->points.size();                            // :2325
        lio::scan_changed(ikdtree);
        Nearest_Points.resize(feats_down_size);

        state_ikfom s{};
        s.pos = {{0.03, -0.02, 1.71}}; s.rot = {{0, 0, 0, 1}}; s.offset_R_L_I = {{0, 0, 0, 1}}; s.offset_T_L_I = {{0, 0, 0}};
        DynShare d;
        d.converge = true; d.valid = true;
        h_share_model(s, d);                                                              // esekfom.hpp:1639, search pass
        if (!d.valid || effct_feat_num < feats_down_size / 2) { std::printf("too few effective points: %d of %d\n", effct_feat_num, feats_down_size); return 1; }
        std::printf("pass 1: effct_feat_num %d of %d, res_mean_last %.4f, h_x %ld x %ld\n", effct_feat_num, feats_down_size, res_mean_last, (long)d.h_x.rows(), (long)d.h_x.cols());
        if (d.h_x.rows() != effct_feat_num || d.h_x.cols() != 12 || d.h.rows() != effct_feat_num) return 1;
        d.converge = false;
        s.pos = {{0.0, 0.0, 1.70}};
        h_share_model(s, d);                                                              // a pass that re-uses the neighbours
        std::printf("pass 2: effct_feat_num %d, res_mean_last %.4f\n", effct_feat_num, res_mean_last);
        if (!(res_mean_last < 0.03)) return 1;

        const int offered = map_incremental();                                            // reads the refreshed globals
        int sel = 0;
        for (int i = 0; i < feats_down_size; ++i) sel += point_selected_surf[i] ? 1 : 0;
        if (sel != effct_feat_num || offered <= 0) { std::printf("globals out of date: %d selected vs %d\n", sel, effct_feat_num); return 1; }
        const PointType& w0 = feats_down_world->points[0];
        const PointType& b0 = feats_down_body->points[0];
        if (std::fabs(w0.z - (b0.z + 1.70f)) > 1e-4f || w0.intensity != b0.intensity) { std::printf("feats_down_world not refreshed\n"); return 1; }

        std::vector<BoxPointType> cub_needrm(1);                                          // :1183-1198
        cub_needrm[0].vertex_min[0] = -25; cub_needrm[0].vertex_min[1] = -25; cub_needrm[0].vertex_min[2] = -1;
        cub_needrm[0].vertex_max[0] = -22; cub_needrm[0].vertex_max[1] = 25; cub_needrm[0].vertex_max[2] = 3;
        const int kdtree_delete_counter = ikdtree.Delete_Point_Boxes(cub_needrm);
        PointVector points_history;
        ikdtree.acquire_removed_points(points_history);                                   // :1127
        if (kdtree_delete_counter <= 0 || (int)points_history.size() != kdtree_delete_counter) { std::printf("delete/acquire mismatch\n"); return 1; }

        PointVector().swap(ikdtree.PCL_Storage);                                          // :2363
        ikdtree.flatten(ikdtree.Root_Node, ikdtree.PCL_Storage, NOT_RECORD);              // :2364
        featsFromMap->points = ikdtree.PCL_Storage;                                       // :2366
        if ((int)featsFromMap->points.size() != ikdtree.validnum()) { std::printf("flatten size mismatch\n"); return 1; }

        PointVector near; std::vector<float> sq;
        ikdtree.Nearest_Search(feats_down_world->points[5], 5, near, sq);                 // :1909 (single query form)
        if (near.size() != 5 || !(sq[0] <= sq[4])) { std::printf("Nearest_Search failed\n"); return 1; }

        ikdtree.reconstruct(featsFromMap->points);                                        // :656
        if (ikdtree.validnum() != (int)featsFromMap->points.size()) return 1;
    } catch (const std::exception& e) {
        std::printf("adaptor: %s\n", e.what());
        return 2;                                                                         // no GPU: lio_create fails loudly
    }
    std::printf("ADAPTOR DEMO OK\n");
    return 0;
}
