/*
 * oracle/fastlivo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU oracle: a dependency-free restatement of the FAST-LIVO ESKF hot path
 * (reference snapshot 2024-11-08). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (fast-livo_amd/) never does.
 *
 * Parity status.  The reference ships no tests, golden vectors or fixtures for this path and needs
 * Eigen / PCL / ROS / OpenCV / vikit / Sophus / Boost, none of which exist in this environment.  What the
 * oracle is held to instead:
 *   - its k-NN and map rows: the reference's own ikd-Tree compiled unmodified (oracle/ref_ikdtree) -- PINNED;
 *   - the Mode-18 loop, the VIO update, the patch selection and the visual map, the local map, the IMU undistortion, h_share_model,
 *     the IKFoM updater and the toolkit's manifold types (vect / SO3 / S2 / mtkmath): the reference's own
 *     TEXT, read from /root/reference at build time, compiled and run over a stand-in for Eigen's API (oracle/ref_eigen,
 *     tests/test_ref_eigen_cpu.py: bit for bit) -- the reference's logic is pinned, Eigen's own arithmetic is not;
 *   - third-party arithmetic (Eigen's summation orders and its Quaternion, vikit, Sophus, PCL, OpenCV):
 *     PARITY UNPINNED -- restated from the published sources, cross-checked by independent numpy restatements
 *     (oracle/np_oracle.py, oracle/np_ikfom.py), sensitivity studies and analytic properties (tests/).
 */
#ifndef FASTLIVO_ORACLE_H
#define FASTLIVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DIM18 18

/* StatesGroup, /root/reference/include/common_lib.h:296-381. Matrices are row-major. */
typedef struct orc_state18 {
    double rot[9];
    double pos[3];
    double vel[3];
    double bg[3];
    double ba[3];
    double grav[3];
    double cov[ORC_DIM18 * ORC_DIM18];
} orc_state18;

typedef struct orc_lio18_iter_out {
    double HTH[36];       /* Hsub^T Hsub                        laserMapping.cpp:1666 */
    double HTz[6];        /* Hsub^T meas_vec                    laserMapping.cpp:1665 */
    double solution[18];  /* delta applied to the state         laserMapping.cpp:1672 */
    double total_residual;/* sum |pd2| over effective points    laserMapping.cpp:1597 */
    int32_t effct_feat_num;
    int32_t converged;    /* flg_EKF_converged                  laserMapping.cpp:1688-1691 */
    int32_t status;       /* 0 ok, 1 singular, 2 non-finite */
    int32_t pad;
} orc_lio18_iter_out;

/* One pass of the body of the Mode-18 loop, laserMapping.cpp:1506-1695 ([A]-[D] of SURVEY 3.2),
 * with the kNN already done: nbr = 5 neighbours per point (ascending distance), and on entry
 * sel[i] = point_selected_surf[i] (on a search pass: the kNN validity sq[4]<=5 && count==5).
 * sel, normvec, res_last persist between calls exactly like the reference's globals.
 * G (18x18) is the reference's persistent G matrix (columns 0..5 rewritten). */
int orc_lio18_iterate(orc_state18 *x, const orc_state18 *x_prop, const float *body_xyz,
                      const float *nbr_xyz, uint8_t *sel, int n, const double *R_LI,
                      const double *t_LI, double laser_point_cov, int nthreads, float *world_xyz,
                      float *normvec, double *res_last, double *G, orc_lio18_iter_out *out);

/* kNN provider: fills nbr (n x 5 x 3 floats, ascending) and valid (n) for the given world points. */
typedef void (*orc_knn_fn)(void *ctx, const float *world_xyz, int n, float *nbr_xyz, uint8_t *valid);

typedef struct orc_lio_frame_out {
    int32_t iterations;   /* passes of the loop body executed */
    int32_t searches;     /* passes with nearest_search_en */
    int32_t effct_feat_num;
    int32_t converged_last;
    double total_residual;
} orc_lio_frame_out;

/* Whole per-frame loop incl. rematch/stop logic and covariance update, laserMapping.cpp:1472-1732.
 * sel_out (n) / normvec_out (n x 4) receive the last pass's selection (publish_effect_world input). */
int orc_lio18_frame(orc_state18 *x, const float *body_xyz, int n, const double *R_LI,
                    const double *t_LI, double laser_point_cov, int max_iterations, orc_knn_fn knn,
                    void *knn_ctx, int nthreads, uint8_t *sel_out, float *normvec_out,
                    orc_lio_frame_out *out);

/* Exact brute-force 5-NN with the ikd-Tree's float distance (ikd_Tree.cpp:1291-1295), ascending, ties by
 * lower map index; valid = 5 found && sqdist[4] <= 5 (laserMapping.cpp:1549,1567). */
int orc_knn5(const float *map_xyz, int k, const float *query_xyz, int n, float *nbr_xyz, float *sqdist, uint8_t *valid,
             int32_t *nbr_idx, int nthreads);

/* Map maintenance as the reference drives the ikd-Tree (orc_map.c): KD_TREE::Add_Points(points, downsample) ikd_Tree.cpp:382-457,
 * Delete_Point_Boxes :501-520, lasermap_fov_segment laserMapping.cpp:363-417 -- on a flat point array, sequentially. */
typedef struct orc_map_info { int32_t n_before, n_after, n_added, n_removed, n_ambiguous; } orc_map_info;
int orc_map_add_points(const float *map_xyz, int n_map, const float *new_xyz, int n_new, float downsample_size, float *out_xyz,
                       orc_map_info *info);
int orc_map_delete_boxes(const float *map_xyz, int n_map, const float *boxes, int nb, float *out_xyz, orc_map_info *info);
int orc_fov_segment(float *win, int *initialized, const double *pos_lid, double cube_len, float det_range, float mov_threshold,
                    float *boxes_out);

/* pcl::VoxelGrid::applyFilter as used at laserMapping.cpp:1398-1399 / lidar_selection.cpp:352-353 (orc_voxel.c).
 * xyzi: n x 4 floats (x, y, z, intensity); out_xyzi has room for n points; centroids in ascending voxel index. */
int orc_voxel_grid(const float *xyzi, int n, float leaf_x, float leaf_y, float leaf_z, float *out_xyzi, int32_t *out_n,
                   int32_t *leaf_too_small);

/* ------------------------------------------------------------ IMU propagation + undistortion (orc_imu.c) */
typedef struct orc_imu_sample { double t; double gyr[3]; double acc[3]; } orc_imu_sample;   /* sensor_msgs::Imu fields used */
typedef struct orc_pose6d { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; } orc_pose6d;  /* Pose6D, common_lib.h:396-412 */
typedef struct orc_imu_proc {          /* members of ImuProcess read/written by UndistortPcl (IMU_Processing.h) */
    double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3];
    double mean_acc[3];
    double Lid_rot_to_IMU[9], Lid_offset_to_IMU[3];
    double acc_s_last[3], angvel_last[3];
    orc_imu_sample last_imu;
    double last_lidar_end_time;
} orc_imu_proc;
/* ImuProcess::UndistortPcl, IMU_Processing.cpp:611-809. imu = meas.imu (n_imu samples, without last_imu_);
 * pts_xyzt: n x 4 floats (x, y, z, curvature = offset in ms), compensated in place; poses_out (nullable) has
 * room for n_imu + 1 poses. state: rot/pos/vel = *_end, cov propagated. */
int orc_imu_undistort(orc_imu_proc *proc, orc_state18 *state, const orc_imu_sample *imu, int n_imu, double pcl_beg_time,
                      double pcl_end_time, float *pts_xyzt, int n, orc_pose6d *poses_out, int32_t *n_poses_out);

/* ---------------------------------------------------------------- VIO (lidar_selection.cpp) */
typedef struct orc_vio_config {
    double Rcl[9], Pcl[3];    /* camera <- lidar extrinsic (avia.yaml:42-45)            */
    double R_LI[9], t_LI[3];  /* imu <- lidar extrinsic (set_extrinsic, :35-39)         */
    double fx, fy, cx, cy;    /* pinhole intrinsics                                      */
    double d[5];              /* radtan distortion k1,k2,p1,p2,k3 (vikit; 0 => off)      */
    int32_t width, height;
    int32_t max_iterations;   /* NUM_MAX_ITERATIONS                                      */
    int32_t patch_size;       /* 8 */
    double img_point_cov;
} orc_vio_config;

typedef struct orc_vio_level_out {
    double HTH[36];
    double HTz[6];
    double solution[18];
    float error;        /* returned last_error */
    int32_t iterations; /* iterations executed at this level */
    int32_t n_meas;
    int32_t accepted;   /* accepted solves */
    int32_t fragile;    /* test infrastructure: an accept test was decided within float-rounding distance (3e-5 relative) */
} orc_vio_level_out;

/* LidarSelector::UpdateState(img, total_residual, level), lidar_selection.cpp:743-902.
 * ref_patch: m x 3 x 64 floats (index 64*level + 8*x + y); pos: m x 3 doubles (world);
 * search_level: m ints; errors: m floats out. G persists (18x18). */
float orc_vio_update_state(const orc_vio_config *cfg, orc_state18 *x, const orc_state18 *x_prop,
                           const uint8_t *img, const float *ref_patch, const double *pos,
                           const int32_t *search_level, int m, float total_residual, int level,
                           float *errors, double *G, orc_vio_level_out *out);

/* LidarSelector::ComputeJ, lidar_selection.cpp:967-983: levels 2,1,0 then cov -= G*cov. */
int orc_vio_compute_j(const orc_vio_config *cfg, orc_state18 *x, const orc_state18 *x_prop,
                      const uint8_t *img, const float *ref_patch, const double *pos,
                      const int32_t *search_level, int m, float *errors,
                      orc_vio_level_out *out3 /* [3], index = level */);

/* Pixel-level part of LidarSelector::addFromSparseMap (orc_select.c). A candidate = the map point a grid cell
 * chose (:441-466) plus the reference observation Point::getCloseViewObs picked for it (point.cpp:141-178). */
typedef struct orc_patch_candidate {
    double pos[3];        /* pt->pos_ */
    double px_ref[2];     /* ref_ftr->px */
    double f_ref[3];      /* ref_ftr->f */
    double R_ref[9], t_ref[3]; /* ref_ftr->T_f_w_ */
    int32_t keyframe_id;  /* which image is ref_ftr->img */
    int32_t level_ref;    /* ref_ftr->level (ignored by warpAffine, :258-296) */
    int32_t grid_index;
    int32_t reserved;
} orc_patch_candidate;
void orc_vio_depth_image(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const float *scan_world_xyz, int n,
                         float *depth);
/* patches: room for m x 3 x 64; accepted_idx/errors/search_levels: room for m; reason (nullable, m):
 * 0 accepted, 1 depth discontinuity, 3 NCC gate, 4 outlier gate. Returns -2 if cfg has distortion. */
int orc_vio_select(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const uint8_t *cur_img,
                   const uint8_t *const *keyframes, const float *depth, const orc_patch_candidate *cand, int m,
                   int ncc_en, double ncc_thre, double outlier_threshold, int32_t *accepted_idx, float *patches,
                   float *errors, int32_t *search_levels, int32_t *n_accepted, int32_t *reason);

/* lidar_selection.cpp:412-466 on a flat list of map points (orc_select.c); arrays of length (W/grid)*(H/grid); returns it. */
int orc_vio_grid_select(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const double *pos, const float *value, int k,
                        int grid_size, int32_t *winner, float *map_dist, float *map_value, int32_t *grid_num);

/* The visual map and its three per-frame steps (orc_vmap.c): addSparseMap :142-230, addFromSparseMap :346-587 (whole),
 * addObservation :913-965, Point::getCloseViewObs / getFurthestViewObs (point.cpp). Opaque handle. */
#define ORC_VMAP_MAX_OBS 20
typedef struct orc_vmap orc_vmap;
typedef struct orc_vmap_obs { double px[2], f[3], R[9], t[3]; float score; int32_t level, kf_id, frame_id; } orc_vmap_obs;
orc_vmap *orc_vmap_create(const orc_vio_config *cfg, int grid_size);
void orc_vmap_destroy(orc_vmap *m);
int orc_vmap_size(const orc_vmap *m);
int orc_vmap_get_point(const orc_vmap *m, int i, double *pos, float *value, int32_t *n_obs, orc_vmap_obs *obs);
void orc_vmap_get_grid(const orc_vmap *m, float *map_value, int32_t *grid_num);
float orc_shi_tomasi(const uint8_t *img, int width, int height, int u, int v);
int orc_vmap_add_sparse(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *img, const float *scan_world_xyz, int n,
                        int kf_id, int frame_id);
int orc_vmap_select(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *cur_img, const uint8_t *const *keyframes,
                    const float *scan_down_world_xyz, int n, int ncc_en, double ncc_thre, double outlier_threshold,
                    int32_t *sel_point, float *errors, int32_t *search_levels, float *patches, int32_t *n_selected);
int orc_vmap_add_observation(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *img, const int32_t *sel_point,
                             const int32_t *search_levels, int n_sel, int kf_id, int frame_id);

/* vk::PinholeCamera::world2cam (rpg_vikit, unpinned master; restated from memory). */
void orc_world2cam(const orc_vio_config *cfg, const double *xyz_c, double *px);
/* vk::PinholeCamera::cam2world; with distortion = cv::undistortPoints (five sweeps), restated in orc_vio.c */
void orc_cam2world(const orc_vio_config *cfg, double u, double v, double *f);

/* ------------------------------------------------------------- Mode-23 (IKFoM, dormant path) */
typedef struct orc_state23 {
    double pos[3];
    double rot[4];          /* Eigen quaternion coeffs order x,y,z,w */
    double offset_R_L_I[4]; /* x,y,z,w */
    double offset_T_L_I[3];
    double vel[3];
    double bg[3];
    double ba[3];
    double grav[3];         /* S2, |grav| = 9.809 */
} orc_state23;

typedef struct orc_ikfom_out {
    double HTH[144];      /* last h_x^T h_x */
    double HTh[12];       /* last h_x^T h */
    double dx[23];        /* last dx_ */
    int32_t iterations;   /* calls of h_share_model */
    int32_t searches;     /* calls with converge == true */
    int32_t effct_feat_num;
    int32_t status;
} orc_ikfom_out;

/* h_share_model, laserMapping.cpp:961-1093: fills h_x (n_eff x 12) and h (n_eff); returns n_eff.
 * converge != 0 => caller has refreshed nbr/sel from a kNN pass at the current state. */
int orc_h_share_model(const orc_state23 *s, const float *body_xyz, const float *nbr_xyz,
                      uint8_t *sel, int n, int nthreads, float *world_xyz, float *normvec,
                      double *res_last, double *h_x /* n x 12 */, double *h /* n */,
                      double *total_residual);

/* esekf::update_iterated_dyn_share_modified(R, solve_time), esekfom.hpp:1619-1928, driving
 * orc_h_share_model + the kNN provider. P is 23x23 row-major, in/out. */
int orc_ikfom_update_iterated(orc_state23 *x, double *P, const float *body_xyz, int n, double R,
                              int maximum_iter, const double *limit /*23*/, orc_knn_fn knn,
                              void *knn_ctx, int nthreads, uint8_t *sel_out, float *normvec_out,
                              orc_ikfom_out *out);

/* bench.py's "generous" CPU baseline: threads for the VIO patch loop / column sums (1 = the reference's single thread; results
 * are bit-identical for any value, see orc_vio.c). */
void orc_vio_set_threads(int n);
/* sensitivity study only: alternative operation orders of the unpinned radtan projection (orc_vio.c); 0 = the restatement */
void orc_vio_set_radtan_mode(int mode);
void orc_vio_cam_pose(const orc_vio_config *cfg, const orc_state18 *x, double *Rcw, double *Pcw);

/* The same update around ANY measurement callback of the reference's shape -- measurementModel_dyn_share, esekfom.hpp:129:
 * `void (state &, dyn_share_datastruct<scalar_type> &)`, registered by init_dyn_share (:238-254), invoked at :1636.  The callback
 * receives the state and the in/out flags valid (true on entry) / converge, and returns h_x (rows x 12 row-major) and h (rows).
 * Used by tests to run a replacement h_share_model through the unmodified updater. */
typedef void (*orc_h_dyn_share_fn)(void *ctx, orc_state23 *x, int *valid, int *converge, int *rows, const double **h_x, const double **h);
int orc_ikfom_update_dyn_share(orc_state23 *x, double *P, double R, int maximum_iter, const double *limit /*23*/,
                               orc_h_dyn_share_fn h_dyn_share, void *h_ctx, orc_ikfom_out *out);

/* state_ikfom boxplus / boxminus (build_manifold.hpp:192-200) exposed for unit tests. */
void orc_state23_boxplus(orc_state23 *x, const double *dx /*23*/);
void orc_state23_boxminus(const orc_state23 *x, const orc_state23 *other, double *dx /*23*/);

/* Eigen::Quaternion applied to a vector (QuaternionBase::_transformVector) and as a rotation matrix (toRotationMatrix), as orc_ikfom.c restates them */
void orc_unit_q_rot(const double *q /*x,y,z,w*/, const double *v /*3*/, double *o /*3*/);
void orc_unit_q_to_R(const double *q /*x,y,z,w*/, double *R /*3x3*/);
/* MTK::A_matrix (mtkmath.hpp:236-247), S2::S2_Nx_yy / S2_Mx (S2.hpp:259-280); matrices row-major. */
void orc_unit_A_matrix(const double *v /*3*/, double *res /*3x3*/);
void orc_unit_s2_Nx_yy(const double *vec /*3*/, double *Nx /*2x3*/);
void orc_unit_s2_Mx(const double *vec /*3*/, const double *delta /*2*/, double *Mx /*3x2*/);

/* Unit entry points (one restated reference function each) for the cross-oracle and Eigen-pinning tests:
 * esti_plane<float> common_lib.h:448-493 ; StatesGroup += / - common_lib.h:343-365 ; Exp / Log so3_math.h:54-81. */
int orc_unit_esti_plane(const float *near /*5x3*/, float threshold, float *pabcd /*4*/);
void orc_unit_state18_plus(orc_state18 *x, const double *d /*18*/);
void orc_unit_state18_minus(const orc_state18 *a, const orc_state18 *b, double *out /*18*/);
void orc_unit_so3_exp(const double *v /*3*/, double *R /*9 row-major*/);
void orc_unit_so3_log(const double *R /*9 row-major*/, double *out /*3*/);

#ifdef __cplusplus
}
#endif
#endif
