/* lio_hip.h -- C ABI of liblio_hip.so: the MI355X-native (gfx950, hand-written HIP) replacement for
 * the per-scan LIO measurement path of Yixin-F/better_fastlio2.
 *
 * What it replaces (paths relative to the reference repository):
 *   - KD_TREE<PointType> ikdtree            include/ikd-Tree/ikd_Tree.h:225-249   -> lio_map_*
 *   - h_share_model(state_ikfom&, dyn_share_datastruct<double>&)
 *                                           src/laserMapping.cpp:1876-2004        -> lio_scan_set + lio_measure
 *   - esti_plane<float>                     include/common_lib.h:506-536          -> inside lio_measure
 *   - the consumer side, update_iterated_dyn_share_modified
 *                                           include/IKFoM_toolkit/esekfom/esekfom.hpp:1620-1938
 *     stays on the host; this library ships a host C++ mirror of it (lio_eskf_*) so that a
 *     scan step can be run, tested and timed without Eigen/IKFoM.
 *
 * Conventions: plain C, int status returns (LIO_OK == 0, negative = error), caller-owned host
 * buffers, one context per GPU, a context is not re-entrant.  No torch / Eigen / PCL types cross
 * this boundary.  There is NO CPU fallback: every entry point that computes needs a gfx950 device
 * and fails with LIO_ERR_NO_DEVICE / LIO_ERR_HIP otherwise.
 *
 * Point input: `const float* pts` + `stride_bytes` addresses x,y,z as the first three floats of
 * each record, so the reference's pcl::PointXYZINormal (48-byte records, include/common_lib.h:161)
 * can be passed without repacking (stride 48) as well as packed xyz (stride 12) or float4 (16).
 *
 * Streams: every kernel and copy of a context runs on ITS OWN non-blocking stream (lio_stream); nothing is ordered
 * against the null stream or anybody else's.  Entry points that take DEVICE pointers (lio_scan_set_device,
 * lio_map_build_device, lio_map_add_pair_device, lio_eskf_scan_step_device) enqueue their reads on that stream and may
 * return before the reads have happened: (1) whatever produced the buffer (another stream, torch, RCCL) must have
 * FINISHED before the call -- synchronise the producer, or make lio_stream wait for it (hipStreamWaitEvent on
 * (hipStream_t)lio_stream(ctx)); (2) the buffer must stay untouched until lio_synchronize(ctx) or the next call on the
 * context that returns results to the host (lio_measure / lio_eskf_update...: they wait for the stream).  Host pointers
 * are consumed before the call returns unless the entry point says otherwise (lio_scan_upload_async).
 *
 * Pose block `state14`: pos(3), rot quaternion x,y,z,w (4), offset_R_L_I quaternion x,y,z,w (4),
 * offset_T_L_I (3) -- the members of state_ikfom read by h_share_model
 * (include/use-ikfom.hpp:21-30; Eigen::Quaternion coefficient order).
 */
#ifndef LIO_HIP_H
#define LIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIO_OK                 0
#define LIO_ERR_INVALID_ARG   (-1)
#define LIO_ERR_NO_DEVICE     (-2)
#define LIO_ERR_HIP           (-3)
#define LIO_ERR_CAPACITY      (-4)
#define LIO_ERR_NO_SCAN       (-5)
#define LIO_ERR_NO_MAP        (-6)

#define LIO_NUM_MATCH_POINTS   5      /* NUM_MATCH_POINTS, include/common_lib.h:149 */

typedef struct lio_ctx lio_ctx;

typedef struct lio_config {
    int       device;               /* HIP device ordinal                                               */
    float     filter_size_map_min;  /* map down-sample box (ikdtree.set_downsample_param,
                                       src/laserMapping.cpp:2332; default 0.2 laserMapping.cpp:2067)      */
    float     cell_size;            /* neighbour-grid cell edge; 0 = auto = largest power of two that is
                                       <= 2.5 * filter_size_map_min (0.5 m for 0.2, 0.25 m for 0.1)       */
    int       max_scan_points;      /* N_q capacity (the node's static cap is 100000,
                                       src/laserMapping.cpp:52,65)                                        */
    long long max_map_points;       /* N_m capacity of the resident map                                   */
    int       lanes_per_query;      /* kNN sub-group width: 4, 8, 16, 32 or 64 lanes; 0 = auto           */
} lio_config;

/* Result of one pass of the measurement model (what update_iterated_dyn_share_modified consumes,
 * esekfom.hpp:1790,1810: h_x^T h_x and h_x^T h; h_x has 12 non-zero columns, laserMapping.cpp:1967). */
typedef struct lio_out {
    int      m;                 /* effct_feat_num (laserMapping.cpp:1942-1954)                            */
    int      valid;             /* ekfom_data.valid (laserMapping.cpp:1956-1961): 0 when m < 1            */
    int      n_unresolved;      /* queries of this pass that left the 27-cell search unproven and were
                                   answered by the exact tree walk (0 when do_search == 0)                */
    int      reserved;
    double   res_mean;          /* res_mean_last (laserMapping.cpp:1963)                                  */
    double   HTH[144];          /* h_x^T h_x, 12x12 row-major (symmetric)                                 */
    double   HTh[12];           /* h_x^T h                                                                */
    unsigned long long n_candidates; /* sum over queries of C(q) = live map points in the 27 cells
                                   searched (SURVEY.md section 8(d) byte model); 0 unless lio_census_enable  */
} lio_out;

typedef struct lio_map_info {
    long long n_points;         /* slots in use: live + tomb-stoned + retired copies of re-written buckets */
    long long n_live;           /* validnum()                                                             */
    long long n_cells;          /* occupied neighbour-grid cells                                          */
    long long table_slots;      /* hash-table capacity                                                    */
    float     cell_size;
    float     downsample_size;
    long long bytes_resident;   /* device bytes held by the map index                                     */
    long long n_retired;        /* slots whose bucket has been re-written at the tail since the last full rebuild */
    long long n_incremental_updates;  /* Add_Points batches indexed in O(batch) since lio_create          */
    long long n_full_rebuilds;  /* full re-sort + re-hash passes since lio_create (Build included)        */
    long long n_boundary_points; /* points of down-sampled batches that were replayed one after the other since lio_create
                                    (they, or a point of their voxel, sit within an ulp of a voxel face)  */
} lio_map_info;

/* ---- context -------------------------------------------------------------------------------- */
int  lio_create(const lio_config* cfg, lio_ctx** out);
void lio_destroy(lio_ctx* ctx);
const char* lio_last_error(lio_ctx* ctx);           /* text of the last failure on this context        */
const char* lio_version(void);
int  lio_device_count(void);                          /* HIP devices visible to this process (0 without a GPU)   */
void* lio_stream(lio_ctx* ctx);                     /* the hipStream_t every kernel of ctx runs on     */
int  lio_synchronize(lio_ctx* ctx);

/* ---- map: replaces KD_TREE<PointType> (ikd_Tree.h:225-249) ----------------------------------- */
/* KD_TREE::Build (ikd_Tree.cpp:353-364): the map becomes exactly these points, no down-sampling.  */
int  lio_map_build(lio_ctx* ctx, const float* pts, long long n, int stride_bytes);
/* same, points already resident in device memory (float4 records x,y,z,_)                          */
int  lio_map_build_device(lio_ctx* ctx, const void* d_pts_f4, long long n);
/* KD_TREE::Add_Points (ikd_Tree.cpp:414-489); *n_added = its return value: the number of
 * down-sampled inserts that changed the tree (tmp_counter), 0 when downsample_on == 0.             */
int  lio_map_add(lio_ctx* ctx, const float* pts, int n, int stride_bytes, int downsample_on, int* n_added);
/* map_incremental (src/laserMapping.cpp:1440-1496) for the scan of the last measurement passes:
 * world points at the posterior pose `state14` (pointBodyToWorld, :1451), per-point decision from the
 * neighbours of the last search pass (skip / PointToAdd / PointNoNeedDownsample, :1453-1488), then --
 * if apply != 0 -- Add_Points(PointToAdd, true) and Add_Points(PointNoNeedDownsample, false)
 * (:1492-1493) on this replica.  filter_size_map_min is the reference's double (:56, :2067).
 * out_add_xyz / out_nodown_xyz (may be NULL, capacity lio_scan_size * 3 floats) receive the two lists in
 * scan order, e.g. to replicate them to other GPUs; *add_counter = Add_Points' return value.           */
int  lio_map_incremental(lio_ctx* ctx, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                         int apply, float* out_add_xyz, int* n_add, float* out_nodown_xyz, int* n_nodown,
                         int* add_counter);
/* Device-resident forms for replicated maps (one rank per GPU): the two lists of map_incremental stay in
 * HBM as float4 records (pointers valid until the next lio_map_* call on ctx), are exchanged by the caller
 * (RCCL all-gather) and applied from device memory.                                                      */
int  lio_map_incremental_device(lio_ctx* ctx, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                                void** d_add_f4, int* n_add, void** d_nodown_f4, int* n_nodown);
int  lio_map_add_pair_device(lio_ctx* ctx, const void* d_add_f4, int n_add, const void* d_nodown_f4, int n_nodown,
                             int* add_counter);
/* Add_Points(add, true) then Add_Points(nodown, false) (the pair of laserMapping.cpp:1492-1493) with a
 * single index rebuild; *add_counter = return value of the first call.  Applies replicated insert lists. */
int  lio_map_add_pair(lio_ctx* ctx, const float* add_pts, int n_add, const float* nodown_pts, int n_nodown,
                      int stride_bytes, int* add_counter);
/* ---- replicated maps: one rank per GPU, every rank its own scan stream and a full replica of the map (BASELINE.json
 * configs[3]).  The reference has ONE map and ONE stream (globals `ikdtree`, `kf`,
 * src/laserMapping.cpp:116,136); what a multi-stream node adds after each scan step is the exchange below: every
 * rank's map_incremental lists (PointToAdd / PointNoNeedDownsample, :1442-1493) and delete boxes (cub_needrm,
 * :1183-1198) travel in ONE ncclAllGather (RCCL over xGMI, enqueued on lio_stream -- no host copy of the lists, no
 * device-wide synchronisation) and every replica applies all of them in rank order, so the replicas stay bit-identical.
 * RCCL is loaded with dlopen at lio_replicate_init (world > 1 only); world == 1 runs the same code without it.
 *   rank 0:  lio_replicate_unique_id(id)  -> hand the 128 bytes to the other ranks (any side channel)
 *   all   :  lio_replicate_init(ctx, id, rank, world, 0)
 *   per scan, after the iterated update:  [lio_replicate_queue_boxes(ctx, boxes, nb)]  lio_replicate_step(ctx, ...)   */
int  lio_replicate_unique_id(void* id128);          /* ncclGetUniqueId: 128 bytes                                        */
/* list_capacity: points per list, rank and message (0 = 32768, clamped to max_scan_points); a scan step with more
 * points on some rank repeats the exchange at max_scan_points by itself.                                           */
int  lio_replicate_init(lio_ctx* ctx, const void* id128, int rank, int world, int list_capacity);
void lio_replicate_finalize(lio_ctx* ctx);
/* Delete boxes of THIS rank for the next exchange (<= 4 per exchange): applied by every replica, this one included, at
 * the exchange -- call it INSTEAD of lio_map_delete_boxes / with the boxes lio_fov_segment would delete.             */
int  lio_replicate_queue_boxes(lio_ctx* ctx, const float* minmax6, int nb);
/* map_incremental of this rank's scan at its posterior pose (as lio_map_incremental), the all-gather, and on every
 * replica: Delete_Point_Boxes(all boxes), Add_Points(all PointToAdd, true), Add_Points(all PointNoNeedDownsample,
 * false), rank order inside each.  Totals over all ranks are returned; *add_counter as Add_Points' return value.   */
int  lio_replicate_step(lio_ctx* ctx, const double state14[14], double filter_size_map_min, int flg_EKF_inited,
                        int* n_add_total, int* n_nodown_total, int* n_deleted, int* add_counter);
/* The DEFERRED schedule of the exchange (off by default; every rank of a job must choose the same).  on != 0: the lists and
 * boxes gathered by exchange k enter every replica at the START of exchange k + 1 instead of at the end of exchange k -- scan
 * step k + 1 searches the map without them (all replicas alike: they stay bit-identical, and the sequence of Delete_Point_Boxes /
 * Add_Points calls a replica sees is unchanged), and the half of their Add_Points that only READS the map (voxel keys, grouping
 * sort, the speculative fold) runs on a second stream beside scan step k + 1; only the half that writes stays between two scan
 * steps.  lio_replicate_step then returns the totals of the exchange it ran and *n_deleted / *add_counter of the batch that
 * entered the map during the call (the previous exchange's).  Any other map call (lio_map_*, lio_fov_segment, ...) lets a batch
 * that is held back enter first, so outside lio_replicate_step the map always holds every exchanged list.                     */
int  lio_replicate_defer(lio_ctx* ctx, int on);
/* KD_TREE::Delete_Point_Boxes (ikd_Tree.cpp:536-556): boxes = nb x {min xyz, max xyz}, half-open.  */
int  lio_map_delete_boxes(lio_ctx* ctx, const float* minmax6, int nb, int* n_deleted);
/* lasermap_fov_segment (src/laserMapping.cpp:1136-1200): the local-map cube (edge cube_len) follows the
 * LiDAR position pos_lid; when it moves, the slabs left behind (<= 3 boxes, written to boxes_out[18] if
 * not NULL) are removed with Delete_Point_Boxes.  The cube state (LocalMap_Points) lives in ctx.        */
int  lio_fov_segment(lio_ctx* ctx, const double pos_lid[3], double cube_len, float det_range, float* boxes_out,
                     int* n_boxes, int* n_deleted);
/* KD_TREE::reconstruct (ikd_Tree.cpp:1394-1405) == Build on a new cloud.                           */
int  lio_map_reconstruct(lio_ctx* ctx, const float* pts, long long n, int stride_bytes);
/* KD_TREE::set_downsample_param (ikd_Tree.cpp:186-189, caller laserMapping.cpp:2332): the box edge of
 * Add_Points' down-sampling.  lio_config.filter_size_map_min is the same parameter at creation; as long as no
 * map has been built the neighbour-grid cell follows it (largest power of two <= 2.5 * value).          */
int  lio_map_set_downsample_param(lio_ctx* ctx, float downsample_size);
long long lio_map_size(lio_ctx* ctx);               /* KD_TREE::size()     (ikd_Tree.cpp:79)           */
long long lio_map_validnum(lio_ctx* ctx);           /* KD_TREE::validnum() (ikd_Tree.cpp:129)          */
/* KD_TREE::flatten (ikd_Tree.cpp:1326-1352): live points, xyz packed; *n = live count (may exceed
 * cap, in which case only cap points were written).                                                */
int  lio_map_flatten(lio_ctx* ctx, float* out_xyz, long long cap, long long* n);
/* KD_TREE::acquire_removed_points (ikd_Tree.cpp:559-571, caller laserMapping.cpp:1127): the points removed
 * by Delete_Point_Boxes since the last call (down-sample replacements are not reported, ikd_Tree.cpp:1339).
 * The reference reports a point when the rebuild thread happens to flatten its sub-tree; here every
 * box-deleted point is reported exactly once, by the first call after its deletion.  out_xyz == NULL: only
 * *n (nothing is consumed); cap < *n: LIO_ERR_CAPACITY (nothing is consumed).                          */
int  lio_map_acquire_removed_points(lio_ctx* ctx, float* out_xyz, long long cap, long long* n);
int  lio_map_get_info(lio_ctx* ctx, lio_map_info* info);
/* KD_TREE::Nearest_Search (ikd_Tree.cpp:367-397) for a batch of world-frame queries: out_xyz
 * [n*5*3], out_sqdist [n*5] ascending, out_count [n].  Unfilled slots: NaN / +inf.               */
int  lio_map_nearest_search(lio_ctx* ctx, const float* q_xyz, int n, int stride_bytes,
                            float* out_xyz, float* out_sqdist, int* out_count);

/* ---- scan + measurement model: replaces h_share_model (laserMapping.cpp:1876-2004) ----------- */
/* feats_down_body (laserMapping.cpp:2322-2325): uploads the body-frame scan and resets
 * point_selected_surf to all-true (laserMapping.cpp:2131).                                         */
int  lio_scan_set(lio_ctx* ctx, const float* body_pts, int n, int stride_bytes);
int  lio_scan_set_device(lio_ctx* ctx, const void* d_body_f4, int n);
/* lio_scan_set without the wait: for the caller that hands feats_down_body over the moment its voxel filter has written it
 * (lio::scan_changed, laserMapping.cpp:2322-2325) and goes on with its own work -- the passes that follow are ordered behind the
 * upload on the stream.  Records in page-locked memory the HIP runtime knows (lio_host_alloc, lio_host_register) are PULLED by a
 * kernel over PCIe where they lie -- no host-side pack, no staging copy, the call returns after one kernel launch -- and must then
 * stay untouched until the first measurement pass of the scan has returned; any other memory is packed into the library's staging
 * buffer before the call returns.  stride_bytes: a multiple of 4 (pcl::PointXYZINormal: 48).
 * Large clouds of fat records (>= 32 bytes apart, >= 15 us of PCIe time) are pulled WITH THE CALLER'S HELP: inside the scan's first
 * lio_measure[_begin] the calling thread -- which would only wait for the pass -- packs the front of the cloud into 16-byte records
 * the search reads instead of the raw ones, while the search pulls the back as it is (a third of the PCIe bytes for every record
 * packed in time; the share is steered by the measured time of the scans' first passes; lio_debug_set "assist_pull" = 0 switches
 * it off).  Results do not depend on who got to a record first.                                                                  */
int  lio_scan_set_async(lio_ctx* ctx, const float* body_pts, int n, int stride_bytes);
/* The assisted pull with a SECOND packing thread inside the library (n = 2; n = 1, the default: the calling thread alone, no thread
 * is ever started): it sleeps until lio_scan_set_async hands over a cloud worth the help, then packs the second half of the host's
 * share while the calling thread packs the first.  For nodes that push un-down-sampled sweeps (~100 k points) through the callback:
 * 166 -> ~150 us per scan at C2; nothing below ~50 k points.  Joined by lio_destroy / n = 1.                */
int  lio_scan_pack_threads(lio_ctx* ctx, int n);
/* hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime themselves (the adaptor header): page-locks
 * [p, p + bytes) and maps it for the GPU, so that lio_scan_set_async can pull a cloud and lio_measure_rows can have the copy
 * engine write h_x / h in place.  The range must stay allocated while it is registered.                                        */
int  lio_host_register(void* p, size_t bytes);
int  lio_host_unregister(void* p);
/* Pipelined hand-over (the node receives scan k+1 while scan k is being processed): lio_scan_upload_async enqueues
 * the host-to-device copy of n float4 records (x, y, z, _) into the context's BACK scan buffer on a copy stream of
 * its own and returns at once (host_f4 should be pinned -- lio_host_alloc -- and must stay untouched until
 * lio_scan_swap; pageable memory works but copies synchronously); lio_scan_swap makes the back buffer the current
 * scan (as lio_scan_set would) -- the compute stream waits for the copy only if it is still in flight.  At most one
 * upload may be outstanding; the back buffer is the scan that was current before the last swap.             */
int  lio_scan_upload_async(lio_ctx* ctx, const void* host_f4, int n);
int  lio_scan_swap(lio_ctx* ctx);
void* lio_host_alloc(size_t bytes);                 /* pinned host memory (hipHostMalloc); NULL on failure     */
void lio_host_free(void* p);
/* ---- raw-scan preparation (the step before the path) ---------------------- */
/* Pose6D (msg/Pose6D.msg, include/common_lib.h:446-460): one forward-propagated IMU pose of IMUpose.      */
typedef struct lio_pose6d {
    double offset_time;      /* seconds after the first LiDAR point                                          */
    double acc[3], gyr[3];   /* world-frame acceleration, unbiased body angular rate                          */
    double vel[3], pos[3];   /* world-frame velocity / position                                              */
    double rot[9];           /* world rotation, row-major                                                    */
} lio_pose6d;
/* ImuProcess::UndistortPcl, point side (src/IMU_Processing.hpp:250-251, 334-386): time-sorts the raw scan
 * (the per-point offset in MILLISECONDS is the float at time_offset_bytes inside each record -- `curvature`
 * of pcl::PointXYZINormal, byte 36) and compensates every point to the scan-end LiDAR frame from the IMU
 * poses and the propagated end state (pos, rot, offset_R_L_I, offset_T_L_I as in state14).  The result
 * (feats_undistort) stays in device memory; the call returns when the work is enqueued (the time order is
 * materialised where it is consumed: by the voxel filter's sort, or by lio_scan_get_cloud).  n_poses == 0:
 * upload only, no sort, no compensation.                                                                   */
int  lio_scan_undistort(lio_ctx* ctx, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes,
                        const lio_pose6d* imu_poses, int n_poses, const double end_state14[14]);
/* The two halves of lio_scan_undistort for callers that have host work to do while the copy is in flight
 * (lio_imu_process runs the IMU forward propagation between them): upload of the raw records (asynchronous;
 * time_offset_bytes < 0: no per-point time), then the compensation from the propagated poses.
 * Records that already are {x, y, z, t} floats (stride 16, time at byte 12) in page-locked memory the HIP
 * runtime knows (lio_host_alloc) are read by the copy engine IN PLACE -- no staging copy, 60 us less host time
 * for a 120 k-point sweep -- and must then be left untouched until a call that waits for the stream has
 * returned (lio_scan_voxel_filter, lio_synchronize).  Any other memory is copied before the call returns.   */
int  lio_scan_upload_raw(lio_ctx* ctx, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes);
int  lio_scan_compensate(lio_ctx* ctx, const lio_pose6d* imu_poses, int n_poses, const double end_state14[14]);
/* pcl::VoxelGrid (downSizeFilterSurf, src/laserMapping.cpp:2135, 2322-2323) on feats_undistort: centroid
 * of every occupied leaf, ascending leaf index; the result becomes the current scan (feats_down_body),
 * exactly as if it had been passed to lio_scan_set.  *n_out = feats_down_size.                             */
int  lio_scan_voxel_filter(lio_ctx* ctx, float leaf_size, int* n_out);
/* The node's loop body overlapped (laserMapping.cpp:2256-2401): Process + voxel filter of scan k + 1 depend on nothing of
 * map_incremental / lasermap_fov_segment of scan k but the posterior both start from.
 *   lio_scan_prefetch_raw   a hint, the moment the driver has the sweep: x, y, z, t records (stride 16, time at byte 12) in page-locked
 *                           memory start their way to the GPU on the copy stream; the lio_scan_upload_raw / lio_imu_process call that
 *                           names the same records then takes them where they are.  Anything else: LIO_OK, nothing happens.
 *                           A prefetched sweep is recognised by (records, count) and answers for the NEXT upload only: every call of
 *                           lio_scan_prefetch_raw drops an earlier prefetch nobody consumed, and so do lio_imu_process when it returns
 *                           without undistorting (IMU initialisation) and lio_scan_set / _set_device / _set_async.  Prefetch every
 *                           sweep or none.
 *   lio_scan_prep_begin     from here to lio_scan_prep_end, lio_scan_upload_raw / lio_scan_compensate / lio_scan_undistort /
 *                           lio_imu_process / lio_scan_voxel_filter only ENQUEUE -- on the context's copy stream, into the back scan buffer;
 *                           lio_scan_voxel_filter answers *n_out = -1.  Map calls (lio_map_incremental, lio_fov_segment, ...) and
 *                           read-outs of the CURRENT scan go on as ever on the main stream meanwhile.  Not together with
 *                           lio_scan_upload_async / lio_eskf_scan_step_pipelined (they use the same back buffer).
 *   lio_scan_prep_end       waits for the preparation (usually done), the filtered cloud becomes the current scan (as the synchronous
 *                           lio_scan_voxel_filter leaves it), *n_out = its points.  Same values as the synchronous calls.               */
int  lio_scan_prefetch_raw(lio_ctx* ctx, const float* raw_pts, int n, int stride_bytes, int time_offset_bytes);
int  lio_scan_prep_begin(lio_ctx* ctx);
int  lio_scan_prep_end(lio_ctx* ctx, int* n_out);
/* which = 0: feats_undistort, 1: the current scan; xyz packed, *n = number of points available.            */
int  lio_scan_get_cloud(lio_ctx* ctx, int which, float* out_xyz, int cap, int* n);

/* One pass.  do_search = ekfom_data.converge (laserMapping.cpp:1906); out->valid = ekfom_data.valid.*/
int  lio_measure(lio_ctx* ctx, const double state14[14], int do_search, int extrinsic_est_en,
                 lio_out* out);
/* The same pass in two halves: lio_measure_begin enqueues the kernels and returns at once, lio_measure_end waits for
 * the result.  Host work that does not depend on the result (in update_iterated_dyn_share_modified: x [-] x_prop, the
 * Jacobian-corrected P_ and (P_/R).inverse(), esekfom.hpp:1655-1703, :1788) can run between the two, beside the GPU.
 * Exactly one pass may be in flight per context; every begin must be followed by its end.                  */
int  lio_measure_begin(lio_ctx* ctx, const double state14[14], int do_search, int extrinsic_est_en);
int  lio_measure_end(lio_ctx* ctx, lio_out* out);
/* h_x (m x 12) and h (m) of the LAST lio_measure, compacted in scan order exactly like
 * laserCloudOri / corr_normvect (laserMapping.cpp:1942-2002).  col_major != 0 writes h_x in
 * Eigen's default (column-major, leading dimension m).  cap_rows >= m required.                    */
int  lio_measure_rows(lio_ctx* ctx, double* h_x, double* h, int cap_rows, int col_major);
/* For a node that keeps the UNMODIFIED update_iterated_dyn_share_modified, which reads ekfom_data.h_x / .h
 * (esekfom.hpp:1790,1810): on != 0 makes every following pass also leave its rows compacted in scan order and
 * column-major on the device (one more launch behind the pass), so that lio_measure_rows only moves m x 13 doubles --
 * written by the kernel straight into pinned host memory for scans whose rows fit 2 MB, through the copy engine above.
 * Off (default): lio_measure_rows re-evaluates the pass (the m < 23 dense-gain branch of the library's own filter).   */
int  lio_measure_want_rows(lio_ctx* ctx, int on);
/* Nearest_Points of the last search pass (consumed by map_incremental, laserMapping.cpp:1453-1481):
 * xyz [n*5*3], sqdist [n*5], count [n] (any of them may be NULL).                                  */
int  lio_get_neighbors(lio_ctx* ctx, float* out_xyz, float* out_sqdist, int* out_count);
/* feats_down_world (xyz [n*3]), point_selected_surf ([n]), normvec ([n*4] = n^ xyz, pd2) after the
 * last pass (any may be NULL).                                                                     */
int  lio_get_scan_state(lio_ctx* ctx, float* world_xyz, uint8_t* selected, float* normvec);
int  lio_scan_size(lio_ctx* ctx);

/* ---- timing hooks (a HIP event pair per launch, stamped by the dispatch itself on lio_stream) -------- */
#define LIO_PROF_KNN       0    /* 27-cell k-NN kernel, first search of a scan                      */
#define LIO_PROF_FALLBACK  1    /* exact tree walk for unresolved queries                           */
#define LIO_PROF_RESIDUAL  2    /* plane fit + residual + Jacobian + H^T H reduction + result record */
#define LIO_PROF_REDUCE    3    /* (fused into LIO_PROF_RESIDUAL since the last-arriver reduction; 0) */
#define LIO_PROF_KNN_HINT  4    /* 27-cell k-NN kernel, later searches of the same scan (bounded by the previous neighbours) */
#define LIO_PROF_COUNT     5
/* on = 0: off; 1: every family; otherwise a bit mask of (1 << LIO_PROF_x) | 0x100 (0x100 marks a mask,
 * e.g. 0x100 | 1 << LIO_PROF_KNN times only the k-NN kernel: one event pair per search pass)              */
int  lio_profile_enable(lio_ctx* ctx, int on);
/* candidate census: when on, search passes also report lio_out.n_candidates (costs one device
 * atomic per wavefront of the k-NN kernel, so it is off by default and never on in timed runs)    */
int  lio_census_enable(lio_ctx* ctx, int on);
int  lio_profile_reset(lio_ctx* ctx);
/* test hooks (never needed by a caller): key "force_reduction_timeout" (value != 0) makes the next measurement
 * passes report what a timed-out wait of the cross-workgroup reduction reports -> lio_measure fails with
 * LIO_ERR_HIP instead of returning incomplete sums; "separate_tree_walk" runs the exact tree walk of a search pass as
 * its own launch (knn_tree_kernel) instead of inside the measurement kernel (an A/B switch for timing; identical
results); "library_sort" sends every sort / compaction of map maintenance and scan preparation to rocPRIM
 * instead of the single-launch kernels of lio_compact.hip (identical results); "stamps" makes the last workgroup of a measurement kernel take the time stamps lio_debug_stamps returns
 * (they perturb what they measure: ~0.5 us each); "replicate_loopback" (before lio_replicate_init with id128 == NULL and
 * world > 1) runs the replica exchange with `world` VIRTUAL ranks on this one GPU: lio_replicate_step as virtual rank
 * r < world - 1 ("replicate_loopback_rank" = r) parks that rank's lists and boxes, as rank world - 1 it packs every rank's
 * message into its slot of the gathered buffer -- the place of ncclAllGather -- and goes on like a real rank (compaction
 * in rank order, overflow repeat, Delete_Point_Boxes + the Add_Points pair); "pre_enqueue" = 0 stops
 * lio_eskf_update_iterated_dyn_share_modified from putting the next pass on the stream ahead of its pose (identical results),
 * "gate_mispredict" = 1 makes it predict the wrong kind of pass every time (every pre-enqueued pass is called off; identical
 * results), "gate_in_kernel" (bit 0: passes that re-use neighbours, bit 1: search passes; default 3) chooses where such a pass
 * waits for its pose: inside its own first kernel (mail slots in device memory, i.e. a large-BAR GPU) or, 0, behind a
 * one-wavefront gate kernel (always the form when the slots are host memory; identical results; bit 2: also the search
 * lio_eskf_scan_step_pipelined arms across a scan boundary, which waits behind a gate kernel by default), "arm_next_scan" = 0
 * keeps lio_eskf_scan_step_pipelined from arming it; "knn_walk" = 0 / 1 / 2 / 4 forces the candidate walk of the k-NN kernel (plain list / prefix-form list with that
 * many candidates per lane in flight; -1 = by scan size again) and "lanes_per_query" = 4 ... 64 its lanes per query (0 = by
 * scan size) -- identical results, A/B switches; "voxel_fast_path" = 0 keeps lio_scan_voxel_filter on the general sort even
 * for sweeps in time order (identical results); "retired_budget" overrides the number of retired slots that triggers a full
 * index rebuild (0 = the default, four times the live points); "assist_pull" = 0 / 1 / 2..92: the host's share of an assisted
 * pull off / steered / fixed in per cent, "assist_pull_chunk" = records per progress announcement (0 = by scan size),
 * "assist_pull_stall_us" = test hook: the host sleeps that long before it packs; "knn_stamps" = 1: searches of 8 lanes per query run
 * the stamped instantiation of the k-NN kernel (lio_debug_knn_stamps); "pull_alias_shift" = n: test hook, the fused pull of a host
 * cloud reads n records further on (what a stale registration looks like: the check behind the pull must catch it);
 * "fast_update" = 0: lio_map_incremental(apply = 1, no lists wanted) reads its list lengths back before it launches their
 * Add_Points, as until round 5 (identical results; environment: LIO_FAST_UPDATE=0).
 * Unknown key: LIO_ERR_INVALID_ARG.                                                                                        */
int  lio_debug_set(lio_ctx* ctx, const char* key, long long value);
/* counters for tests: "gate_released" / "gate_called_off" = pre-enqueued passes that ran / were called off; "voxel_fast_runs" /
 * "voxel_redone" = voxel-filter attempts on the compact-key path / sweeps filtered twice after a wrong prediction;
 * "retired_budget" = the retired-slot budget in force; "plans_on_side_stream" / "deferred_settled" = batches of the deferred exchange schedule whose plan ran beside a scan
 * step / that entered the map; "gate_reenqueued" = passes put on the stream again because their gate had
 * given up (a host stalled for seconds); "gate_in_kernel" = passes armed without a gate kernel; "next_scan_armed" / "next_scan_released" = first searches
 * armed across a scan boundary / of those, the ones that ran; "scan_pulled" = scans a kernel pulled out of page-locked host records
 * (lio_scan_set_async); "rows_direct_dma" = lio_measure_rows calls served by copies straight into page-locked h_x / h;
 * "assisted_pulls" = scans whose first search had the host pack the front of the cloud beside it, "last_pack_front" / "last_pack_ns" /
 * "assist_share_permille" = records packed, time it took, and the share in force; "stale_pulls" = scans whose first and last point,
 * as the GPU saw them through a registration, differed from the host's records -- uploaded again through the staging buffer, the pass
 * repeated (a caller drops its registrations then: the adaptor does); "raw_prefetch_hits" = sweeps lio_scan_upload_raw took from the
 * prefetch buffer, "prep_overlapped" = preparations whose filter ran on the preparation stream (lio_scan_prep_begin / _end);
 * "fast_updates" = lio_map_incremental calls whose whole map update ran with no host round trip in the middle               */
int  lio_debug_get(lio_ctx* ctx, const char* key, long long* value);
/* wall_clock64 stamps (100 MHz) the last workgroup of the last measurement kernel took: [0] its start, [1] own partial
 * published, [2] group folded, [3] grid folded, [4] the listed queries' terms folded (search passes that list any),
 * [5] unused, [6] sums final, [7] record written (diagnostics)                                                          */
int  lio_debug_stamps(lio_ctx* ctx, unsigned long long out8[8]);
/* lio_debug_set("knn_stamps", 1): searches of 8 lanes per query run a STAMPED instantiation of the k-NN kernel (diagnostics only: it
 * drains memory in front of every stamp).  Returns the stamps of the last such launch, eight wall_clock64 words (100 MHz) per
 * workgroup: [0] entry, [1] query known (transform + barrier), [2] directory records arrived, [3] bucket list in prefix form,
 * [4] candidates walked, [5] merged + proven, [6] results stored, [7] unused.  out == NULL / cap_words == 0: the count only.   */
int  lio_debug_knn_stamps(lio_ctx* ctx, unsigned long long* out, long long cap_words, int* n_workgroups);
/* accumulated milliseconds and launch count of one kernel family since the last reset             */
int  lio_profile_get(lio_ctx* ctx, int which, double* total_ms, long long* launches);

/* ---- host-side iterated ESKF (mirror of esekfom::esekf, esekfom.hpp:106-2011) ----------------- */
/* State x26 = pos3, rot xyzw, offset_R_L_I xyzw, offset_T_L_I 3, vel3, bg3, ba3, grav3 (S2, |g| =
 * 9.809, use-ikfom.hpp:8,21-30).  P is 23x23 row-major in DOF order pos,rot,offR,offT,vel,bg,ba,grav.*/
typedef struct lio_eskf lio_eskf;
typedef struct lio_update_log {
    int    n_calls;             /* calls of the measurement model (<= maximum_iter + 1)             */
    int    n_search;            /* of which with do_search = 1                                      */
    int    converged;           /* t > 1 reached before the last iteration                          */
    int    last_m;
    double last_res_mean;
    double solve_seconds;       /* host time in the 23x23 algebra                                   */
    double measure_seconds;     /* host wall time inside lio_measure                                */
} lio_update_log;

lio_eskf* lio_eskf_create(void);
void lio_eskf_destroy(lio_eskf* kf);
/* init_dyn_share (esekfom.hpp:239-258): maximum_iter and the 23 convergence limits.               */
int  lio_eskf_init_dyn_share(lio_eskf* kf, int maximum_iter, const double limit23[23]);
int  lio_eskf_change_x(lio_eskf* kf, const double x26[26]);       /* esekfom.hpp:1948             */
int  lio_eskf_change_P(lio_eskf* kf, const double P[23 * 23]);    /* esekfom.hpp:1953             */
int  lio_eskf_get_x(lio_eskf* kf, double x26[26]);                /* esekfom.hpp:1940             */
int  lio_eskf_get_P(lio_eskf* kf, double P[23 * 23]);             /* esekfom.hpp:1944             */
/* update_iterated_dyn_share_modified(R, solve_time) (esekfom.hpp:1620-1938) with the measurement
 * model bound to lio_measure on ctx (the scan must have been set with lio_scan_set).             */
int  lio_eskf_update_iterated_dyn_share_modified(lio_eskf* kf, lio_ctx* ctx, double R,
                                                 int extrinsic_est_en, lio_update_log* log);
/* The same update around a measurement model SUPPLIED BY THE CALLER -- esekf::init_dyn_share(.., h_dyn_share, ..)
 * (esekfom.hpp:239-258, laserMapping.cpp:2151): h is called like the reference calls h_dyn_share(x_, dyn_share)
 * (esekfom.hpp:1639) with the iterate's state and ekfom_data.converge set, and answers like h_share_model does: valid, m
 * and the rows h_x (m x 12, Eigen's column-major default, leading dimension m) and h (m) -- the filter then forms
 * h_x^T h_x and h_x^T h itself, as esekfom.hpp:1790,1810 do.  A callback that already holds the reduced form (lio_measure's
 * HTH / HTh: the "two-line patch" of INTEGRATION.md) leaves h_x NULL and fills HTH / HTh instead.  The pointers must stay
 * valid until h is called again or the update returns.                                                            */
typedef struct lio_dyn_share {
    int           converge;         /* in:  ekfom_data.converge (search again)                                  */
    int           valid;            /* out: ekfom_data.valid                                                    */
    int           m;                /* out: rows (effct_feat_num)                                               */
    int           reserved;
    const double* h_x;              /* out: m x 12 column-major, or NULL                                        */
    const double* h;                /* out: m                                                                   */
    double        HTH[144];         /* out when h_x == NULL: h_x^T h_x, 12 x 12 row-major                       */
    double        HTh[12];          /* out when h_x == NULL: h_x^T h                                            */
} lio_dyn_share;
typedef void (*lio_h_dyn_share_fn)(void* user, const double x26[26], lio_dyn_share* ekfom_data);
int  lio_eskf_update_iterated_dyn_share_callback(lio_eskf* kf, double R, lio_h_dyn_share_fn h, void* user, lio_update_log* log);
/* One scan of the node's loop body (laserMapping.cpp:2320-2325 + :2380) in one call: the propagated state and
 * covariance go in (kf.change_x / kf.change_P; NULL = keep the filter's), the down-sampled scan goes in from
 * device memory (lio_scan_set_device; d_body_f4 == NULL: the current scan, e.g. after lio_scan_swap, is kept), the
 * iterated update runs.  Equivalent to the four calls it names.                                             */
int  lio_eskf_scan_step_device(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                               const void* d_body_f4, int n, double R, int extrinsic_est_en, lio_update_log* log);
/* The same for a node that receives scan k+1 while scan k is being processed: the scan handed over with
 * lio_scan_upload_async during the previous step becomes the current one (lio_scan_swap), the copy of the next scan
 * (next_host_f4, n_next float4 records in pinned host memory; NULL = none) is started behind this step's first
 * kernels, then the update runs.  Before the first step: lio_scan_upload_async(ctx, first_scan, n).
 * With a next scan handed over, the call also leaves that scan's FIRST SEARCH on the stream, behind this scan's last passes,
 * waiting for its pose (a one-wavefront gate kernel in front of it): the next call of this function hands the pose over instead
 * of launching -- the launch calls and the dispatch lie beside this scan's work, not between the scans.  Every other lio_* call
 * on the context (lio_synchronize included) calls that search off first -- it is then launched when its scan's update begins,
 * same results -- and an unanswered wait ends by itself after 2 s.  A caller that synchronises the DEVICE by other means
 * (hipDeviceSynchronize, a collective on another stream that needs the whole GPU) calls lio_synchronize(ctx) first.
 * lio_debug_set("arm_next_scan", 0) switches it off.                                                                    */
int  lio_eskf_scan_step_pipelined(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                                  const void* next_host_f4, int n_next, double R, int extrinsic_est_en, lio_update_log* log);
/* esekf::predict(dt, Q, i_in) (esekfom.hpp:280-383) with the process model f / df_dx / df_dw of
 * use-ikfom.hpp:56-99: propagates x and P over dt with the IMU input (acc, gyro); Q = 12 x 12 process noise
 * (ng, na, nbg, nba), row-major.                                                                             */
int  lio_eskf_predict(lio_eskf* kf, double dt, const double Q[12 * 12], const double acc[3], const double gyro[3]);
/* process_noise_cov() (use-ikfom.hpp:44-52)                                                                  */
int  lio_process_noise_cov(double Q[12 * 12]);

/* ---- host-side IMU front end (mirror of ImuProcess, src/IMU_Processing.hpp:30-441) ------------------------ */
typedef struct lio_imu lio_imu;
typedef struct lio_imu_sample {      /* the fields of sensor_msgs::Imu the reference reads                    */
    double stamp;                    /* header.stamp.toSec()                                                  */
    double acc[3];                   /* linear_acceleration                                                   */
    double gyr[3];                   /* angular_velocity                                                      */
} lio_imu_sample;
typedef struct lio_imu_info {
    int    need_init;                /* imu_need_init_                                                        */
    int    init_iter_num;
    double mean_acc[3], mean_gyr[3], cov_acc[3], cov_gyr[3];
    double last_lidar_end_time;
} lio_imu_info;
lio_imu* lio_imu_create(void);
void lio_imu_destroy(lio_imu* imu);
int  lio_imu_reset(lio_imu* imu);                                                  /* Reset, :91-103          */
int  lio_imu_set_extrinsic(lio_imu* imu, const double T[3], const double R[9]);    /* :131-135, R row-major   */
/* set_gyr_cov / set_acc_cov / set_gyr_bias_cov / set_acc_bias_cov (:141-171; laserMapping.cpp:2143-2146);
 * NULL leaves a value as it is.                                                                              */
int  lio_imu_set_cov(lio_imu* imu, const double gyr_cov[3], const double acc_cov[3],
                     const double b_gyr_cov[3], const double b_acc_cov[3]);
/* ImuProcess::Process(Measures, kf, feats_undistort) (:392-441; laserMapping.cpp:2256): the first calls
 * initialise (IMU_init, :174-231: gravity, gyroscope bias, extrinsics, P) until more than MAX_INI_COUNT
 * samples were seen; later calls run UndistortPcl (:237-386): forward propagation of kf through the IMU
 * samples on the host (lio_eskf_predict per sample), then the per-point compensation on the GPU
 * (lio_scan_undistort on ctx with the propagated poses).  *undistorted = 1 when ctx now holds
 * feats_undistort.  ctx == NULL: propagate only (no device work).  raw_pts as for lio_scan_undistort.         */
int  lio_imu_process(lio_imu* imu, lio_eskf* kf, lio_ctx* ctx, double lidar_beg_time, double lidar_end_time,
                     const float* raw_pts, int n, int stride_bytes, int time_offset_bytes,
                     const lio_imu_sample* samples, int n_samples, int* undistorted);
/* IMUpose of the last UndistortPcl (*n = number available)                                                   */
int  lio_imu_get_poses(lio_imu* imu, lio_pose6d* out, int cap, int* n);
int  lio_imu_get_info(lio_imu* imu, lio_imu_info* info);

/* manifold helpers exposed for tests: x [+] dx and x [-] y on the 23-DOF compound manifold        */
int  lio_state_boxplus(double x26[26], const double dx23[23]);
int  lio_state_boxminus(const double x26[26], const double y26[26], double dx23[23]);

#ifdef __cplusplus
}
#endif
#endif /* LIO_HIP_H */
