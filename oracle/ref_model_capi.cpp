// TEST INFRASTRUCTURE ONLY (oracle/).  Pins the MODEL half of the hot path on the reference itself, wherever Eigen (and
// Boost headers) exist -- they do not in the build image, so this file is compiled by `make -C oracle ref_model
// EIGEN=/path/to/eigen3 [BOOST=/path/to/boost/include]` on a machine that has them (Eigen 3.3.7 is what the reference's
// Dockerfile installs: docker/Dockerfile:2,19 -> ros-noetic-desktop-full on Ubuntu 20.04).
//
// What is the reference's OWN code here, compiled from where it lies under $(REF) and never copied into this repository:
//   - include/IKFoM_toolkit/**, include/use-ikfom.hpp      #included as they are (esekf::update_iterated_dyn_share_modified,
//                                                           the manifolds, boxplus / boxminus)
//   - include/ikd-Tree/ikd_Tree.{h,cpp}                     compiled as for libikd_ref.so
//   - esti_plane (include/common_lib.h:506-536), the macros it needs (:149-172) and the WHOLE function
//     h_share_model (src/laserMapping.cpp:1876-2004)        extracted by oracle/Makefile (awk on the function's first /
//                                                           last line) into oracle/_ref/ref_extract_*.inc at build time
// What this file adds: the file-scope globals those functions reach (same names and types as src/laserMapping.cpp:44-116,
// with a 20-line stand-in for pcl::PointCloud) and a C entry point per quantity the tests compare.
#include <Eigen/Core>
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

using namespace std;       // include/common_lib.h does both (the extracted code relies on them)
using namespace Eigen;

#include <use-ikfom.hpp>   // state_ikfom, esekfom::esekf, process model (reference header, as it is)
#include <ikd_Tree.h>      // KD_TREE<PointType> (reference header, as it is)

typedef pcl::PointXYZINormal PointType;                                       // common_lib.h:161
typedef vector<PointType, Eigen::aligned_allocator<PointType>> PointVector;   // common_lib.h:163
typedef Vector3d V3D;                                                         // common_lib.h:164-167
typedef Matrix3d M3D;
typedef Vector3f V3F;
typedef Matrix3f M3F;
#include "_ref/ref_extract_macros.inc"      // #define NUM_MATCH_POINTS / SKEW_SYM_MATRX / VEC_FROM_ARRAY / MD / VD / MF / VF
#include "_ref/ref_extract_esti_plane.inc"  // template<typename T> bool esti_plane(...)

// stand-in for pcl::PointCloud<PointType>: what h_share_model uses of it (->points[i], ->clear())
namespace refm {
struct Cloud {
    vector<PointType, Eigen::aligned_allocator<PointType>> points;
    explicit Cloud(size_t n = 0) : points(n) {}
    void clear() {}      // the reference clears and then writes points[k] below the old size (:1878, :1948): keep the storage
    typedef std::shared_ptr<Cloud> Ptr;
};
}  // namespace refm
#define ROS_WARN(...) std::fprintf(stderr, __VA_ARGS__)

// ---- the globals of src/laserMapping.cpp the extracted function reaches (:19,26,44-116) ---------------------------
bool extrinsic_est_en = true;
double res_mean_last = 0.05, total_residual = 0.0;
double match_time = 0, solve_time = 0, solve_const_H_time = 0, kdtree_search_time = 0.0;
float res_last[100000] = {0.0};
int effct_feat_num = 0, feats_down_size = 0;
bool point_selected_surf[100000] = {0};
vector<PointVector> Nearest_Points;
refm::Cloud::Ptr feats_down_body(new refm::Cloud()), feats_down_world(new refm::Cloud()), normvec(new refm::Cloud(100000)),
    laserCloudOri(new refm::Cloud(100000)), corr_normvect(new refm::Cloud(100000));
KD_TREE<PointType> ikdtree;

#include "_ref/ref_extract_h_share_model.inc"   // void h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&)

static esekfom::esekf<state_ikfom, 12, input_ikfom> kf;                    // laserMapping.cpp:136

static state_ikfom state_from(const double* x) {   // x26: pos3, rot xyzw, offR xyzw, offT3, vel3, bg3, ba3, grav3
    state_ikfom s;
    for (int i = 0; i < 3; ++i) { s.pos[i] = x[i]; s.offset_T_L_I[i] = x[11 + i]; s.vel[i] = x[14 + i]; s.bg[i] = x[17 + i]; s.ba[i] = x[20 + i]; }
    s.rot = SO3(x[6], x[3], x[4], x[5]);                                    // (w, x, y, z), mtk/types/SOn.hpp:207
    s.offset_R_L_I = SO3(x[10], x[7], x[8], x[9]);
    s.grav.vec[0] = x[23]; s.grav.vec[1] = x[24]; s.grav.vec[2] = x[25];    // no re-normalisation (S2.hpp:111)
    return s;
}
static void state_to(const state_ikfom& s, double* x) {
    for (int i = 0; i < 3; ++i) { x[i] = s.pos[i]; x[11 + i] = s.offset_T_L_I[i]; x[14 + i] = s.vel[i]; x[17 + i] = s.bg[i]; x[20 + i] = s.ba[i]; x[23 + i] = s.grav.vec[i]; }
    x[3] = s.rot.x(); x[4] = s.rot.y(); x[5] = s.rot.z(); x[6] = s.rot.w();
    x[7] = s.offset_R_L_I.x(); x[8] = s.offset_R_L_I.y(); x[9] = s.offset_R_L_I.z(); x[10] = s.offset_R_L_I.w();
}

extern "C" {

// esti_plane<float> (include/common_lib.h:506-536): pts = 5 x (x, y, z)
int refm_esti_plane(const float* pts5x3, float threshold, float* pabcd) {
    PointVector pts(NUM_MATCH_POINTS);
    for (int j = 0; j < NUM_MATCH_POINTS; ++j) { pts[j].x = pts5x3[3 * j]; pts[j].y = pts5x3[3 * j + 1]; pts[j].z = pts5x3[3 * j + 2]; }
    Matrix<float, 4, 1> r;
    const bool ok = esti_plane(r, pts, threshold);
    for (int k = 0; k < 4; ++k) pabcd[k] = r(k);
    return ok ? 1 : 0;
}

// x [+] dx and x [-] y of state_ikfom (MTK_BUILD_MANIFOLD)
void refm_boxplus(double* x26, const double* dx23) {
    state_ikfom s = state_from(x26);
    Matrix<double, 23, 1> d;
    for (int i = 0; i < 23; ++i) d(i) = dx23[i];
    s.boxplus(d);
    state_to(s, x26);
}
void refm_boxminus(const double* x26, const double* y26, double* dx23) {
    state_ikfom a = state_from(x26), b = state_from(y26);
    Matrix<double, 23, 1> d;
    a.boxminus(d, b);
    for (int i = 0; i < 23; ++i) dx23[i] = d(i);
}

void refm_map_build(const float* xyz, int n, float downsample) {            // ikdtree.Build (laserMapping.cpp:2332-2339)
    PointVector v(n);
    for (int i = 0; i < n; ++i) { v[i].x = xyz[3 * i]; v[i].y = xyz[3 * i + 1]; v[i].z = xyz[3 * i + 2]; }
    ikdtree.set_downsample_param(downsample);
    ikdtree.Build(v);
}
void refm_scan_set(const float* body_xyz, int n) {                          // feats_down_body (:2322-2325, :2131)
    feats_down_body->points.assign(n, PointType());
    feats_down_world->points.assign(n, PointType());
    for (int i = 0; i < n; ++i) { feats_down_body->points[i].x = body_xyz[3 * i]; feats_down_body->points[i].y = body_xyz[3 * i + 1]; feats_down_body->points[i].z = body_xyz[3 * i + 2]; }
    feats_down_size = n;
    Nearest_Points.assign(n, PointVector());
    for (int i = 0; i < n; ++i) point_selected_surf[i] = true;
}

// One call of the reference's h_share_model.  h_x [m * 12] ROW-major, h [m]; world [n*3], normvec [n*4], selected [n],
// nbr_xyz [n*5*3] / nbr_cnt [n] = Nearest_Points.  Returns ekfom_data.valid.
int refm_h_share_model(const double* x26, int converge, int ext, double* h_x, double* h, int* m_out, double* res_mean,
                       float* world, float* normvec_out, uint8_t* selected, float* nbr_xyz, int* nbr_cnt) {
    extrinsic_est_en = ext != 0;
    state_ikfom s = state_from(x26);
    esekfom::dyn_share_datastruct<double> d;
    d.valid = true; d.converge = converge != 0;
    h_share_model(s, d);
    *m_out = effct_feat_num;
    *res_mean = res_mean_last;
    const int n = feats_down_size;
    for (int i = 0; i < n; ++i) {
        const PointType& w = feats_down_world->points[i];
        world[3 * i] = w.x; world[3 * i + 1] = w.y; world[3 * i + 2] = w.z;
        const PointType& q = normvec->points[i];
        normvec_out[4 * i] = q.x; normvec_out[4 * i + 1] = q.y; normvec_out[4 * i + 2] = q.z; normvec_out[4 * i + 3] = q.intensity;
        selected[i] = point_selected_surf[i] ? 1 : 0;
        nbr_cnt[i] = (int)Nearest_Points[i].size();
        for (int k = 0; k < nbr_cnt[i] && k < 5; ++k) {
            nbr_xyz[(i * 5 + k) * 3] = Nearest_Points[i][k].x; nbr_xyz[(i * 5 + k) * 3 + 1] = Nearest_Points[i][k].y; nbr_xyz[(i * 5 + k) * 3 + 2] = Nearest_Points[i][k].z;
        }
    }
    if (!d.valid) return 0;
    for (int r = 0; r < effct_feat_num; ++r) {
        for (int c = 0; c < 12; ++c) h_x[r * 12 + c] = d.h_x(r, c);
        h[r] = d.h(r);
    }
    return 1;
}

// kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_H_time) (laserMapping.cpp:2380) from prior (x26, P):
// the reference's own filter around the reference's own h_share_model.  P 23 x 23 ROW-major in and out.
void refm_update_iterated(double* x26, double* P, int max_iter, int ext, double R) {
    extrinsic_est_en = ext != 0;
    double epsi[23];
    for (int i = 0; i < 23; ++i) epsi[i] = 0.001;                           // laserMapping.cpp:2148-2149
    kf.init_dyn_share(get_f, df_dx, df_dw, h_share_model, max_iter, epsi); // :2151
    state_ikfom s = state_from(x26);
    kf.change_x(s);
    esekfom::esekf<state_ikfom, 12, input_ikfom>::cov Pm;
    for (int r = 0; r < 23; ++r) for (int c = 0; c < 23; ++c) Pm(r, c) = P[r * 23 + c];
    kf.change_P(Pm);
    for (int i = 0; i < feats_down_size; ++i) point_selected_surf[i] = true;
    double solve_H_time = 0;
    kf.update_iterated_dyn_share_modified(R, solve_H_time);
    state_to(kf.get_x(), x26);
    const auto& Po = kf.get_P();
    for (int r = 0; r < 23; ++r) for (int c = 0; c < 23; ++c) P[r * 23 + c] = Po(r, c);
}

const char* refm_eigen_version() {
    static char v[32];
    std::snprintf(v, sizeof(v), "%d.%d.%d", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
    return v;
}

}  // extern "C"
