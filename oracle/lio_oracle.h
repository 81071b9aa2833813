/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the LIO hot path.
 *
 * A plain C++ (no Eigen / Boost / PCL / ROS: none are installed) restatement of the reference's
 * per-scan measurement path, each function citing the reference file:line it follows
 * (paths relative to /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (better_fastlio2_amd/, liblio_hip.so)
 * never links, imports or calls it.
 *
 * PINNING STATUS
 *   - k-NN / map live set (ikd-Tree): PINNED against the reference itself -- the verbatim
 *     reference ikd_Tree.cpp is compiled by oracle/Makefile into oracle/_ref/libikd_ref.so and
 *     tests/test_oracle_vs_ref.py checks this restatement against it bit-for-bit.
 *   - esti_plane / h_share_model / IKFoM update: **parity unpinned** -- they need Eigen (un-vendored,
 *     unpinned: CMakeLists.txt:51, Eigen 3.3.7 by distro) and cannot be compiled here; the reference
 *     has no tests or golden vectors.  The restatement follows Eigen 3.3.7's
 *     published ColPivHouseholderQR / PartialPivLU algorithms and is cross-checked against
 *     scipy.linalg.qr(pivoting=True) and numpy.linalg in tests/test_oracle_math.py.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- map live-set model: restates the visible semantics of KD_TREE<PointType>
 *      (include/ikd-Tree/ikd_Tree.h:225-249) -------------------------------------------------- */
void*  orc_map_create(float downsample_size);                       /* ikd_Tree.cpp:10,40  */
void   orc_map_destroy(void* m);
void   orc_map_build(void* m, const float* xyz, int n);             /* ikd_Tree.cpp:353    */
int    orc_map_add(void* m, const float* xyz, int n, int downsample_on); /* ikd_Tree.cpp:414 */
int    orc_map_delete_boxes(void* m, const float* minmax6, int nb); /* ikd_Tree.cpp:536    */
int    orc_map_validnum(void* m);                                   /* ikd_Tree.cpp:129    */
int    orc_map_flatten(void* m, float* out_xyz, int cap);           /* ikd_Tree.cpp:1326   */
/* Nearest_Search (ikd_Tree.cpp:367) for nq queries. method 0 = brute force over the live set,
 * 1 = restated k-d tree branch-and-bound (ikd_Tree.cpp:574,869).  Returns seconds spent. */
double orc_map_knn(void* m, const float* q_xyz, int nq, int k, float* out_xyz, float* out_d,
                   int* out_cnt, int method);

/* knn_method == 2 routes the k-NN of orc_h_share_model / orc_update_iterated through an external
 * provider with the signature of ref_ikd_nearest (oracle/ref_ikd_capi.cpp), i.e. the reference's
 * own ikd-Tree; `map` may then be any (even empty) oracle map. */
void   orc_set_external_knn(void* fn, void* handle);

/* ---- measurement model ------------------------------------------------------------------------ */
/* esti_plane<float> (include/common_lib.h:506-536); pts = 5 x (x,y,z); returns 1 if plane ok. */
int    orc_esti_plane(const float* pts5x3, float threshold, float* pabcd);
/* TEST SWITCH -- how much can "parity unpinned" matter?  Order in which the reductions (squaredNorm / dot) inside the
 * QR of esti_plane add their fp32 terms: 0 sequential (default; what the GPU path is bit-equal to), 1 Eigen's SSE2
 * packet order (a0+a2)+(a1+a3)+tail, 2 its SSE3 hadd order (a0+a1)+(a2+a3)+tail, 3 a random permutation and association
 * per reduction (seeded).  Not thread-safe: set it while no other thread runs the model. */
void   orc_set_sum_order(int mode, unsigned long long seed);
/* decision-flip audit: distance of the fp32 plane-inlier (0.1 m) and weight (s > 0.9) decisions to their thresholds */
void   orc_decision_margins(const float* nbr_xyz, const float* nbr_d, const int* nbr_cnt, const float* world_xyz,
                            const float* body_xyz, int n, float* margin_plane, float* margin_s);
/* world transform of laserMapping.cpp:1894-1898; state14 = pos3, rot xyzw, offR xyzw, offT3. */
void   orc_body_to_world(const double* state14, const float* body_xyz, int n, float* world_xyz);

/* One call of h_share_model (src/laserMapping.cpp:1876-2004).
 * Persistent per-scan arrays (the reference's globals): nbr_xyz [n*5*3], nbr_d [n*5], nbr_cnt [n]
 * (Nearest_Points), selected [n] (point_selected_surf, initialised to 1 by the caller,
 * laserMapping.cpp:2131).  Outputs: world [n*3], normvec [n*4] (n^, pd2), res_last [n],
 * h_x [m*12] ROW-major, h [m], *m_out, *res_mean.  Returns ekfom_data.valid (0/1).
 * knn_method as orc_map_knn. */
int    orc_h_share_model(void* map, const double* state14, const float* body_xyz, int n,
                         int converge, int extrinsic_est_en, int knn_method,
                         float* nbr_xyz, float* nbr_d, int* nbr_cnt, uint8_t* selected,
                         float* world, float* normvec, float* res_last,
                         double* h_x, double* h, int* m_out, double* res_mean);

/* ---- filter -------------------------------------------------------------------------------------
 * state x[26] = pos3, rot xyzw, offR xyzw, offT3, vel3, bg3, ba3, grav3 (S2, |g| = 9.809,
 * include/use-ikfom.hpp:8,21-30); P[23*23] row-major; DOF order pos,rot,offR,offT,vel,bg,ba,grav. */
void   orc_state_boxplus(double* x26, const double* dx23);                       /* MTK boxplus  */
void   orc_state_boxminus(const double* x26, const double* y26, double* dx23);   /* x [-] y      */
/* update_iterated_dyn_share_modified (esekfom.hpp:1620-1938) around orc_h_share_model.
 * log (optional, may be NULL): per call of h_share_model writes [converge_in, valid, m, res_mean]
 * (4 doubles), up to max_iter+1 entries; *n_calls receives the count.
 * Also returns the LAST pass's nbr_* / selected through the persistent arrays. */
void   orc_update_iterated(void* map, double* x26, double* P, const float* body_xyz, int n,
                           double R, int max_iter, const double* limit23, int extrinsic_est_en,
                           int knn_method, float* nbr_xyz, float* nbr_d, int* nbr_cnt,
                           uint8_t* selected, double* log, int* n_calls);

/* Information-form pieces of the update for cross-checks: HTH (144, row-major), HTh (12). */
void   orc_hth(const double* h_x, const double* h, int m, double* HTH, double* HTh);

/* map_incremental's classification (src/laserMapping.cpp:1440-1496) given the last pass's
 * neighbours: cls[i] = 0 skip, 1 PointToAdd (down-sampled insert), 2 PointNoNeedDownsample. */
void   orc_map_incremental_classify(const float* world_xyz, int n, const float* nbr_xyz,
                                    const int* nbr_cnt, double filter_size_map_min,
                                    int flg_EKF_inited, uint8_t* cls);

/* UndistortPcl point side (src/IMU_Processing.hpp:250-251, 334-386) and pcl::VoxelGrid (PCL 1.10, xyz only):
 * both "parity unpinned" (Eigen / PCL are un-vendored third-party code). */
void   orc_undistort(const float* xyzt, int n, const double* poses22, int K, const double* end14, float* out_xyz, float* out_t);
int    orc_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap);
/* lasermap_fov_segment (src/laserMapping.cpp:1136-1200); state7 = LocalMap_Points min, max, initialised flag */
int    orc_fov_segment(float* state7, const double* pos_lid, double cube_len, float det_range, float* boxes18);

#ifdef __cplusplus
}
#endif
