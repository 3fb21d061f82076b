/*
 * include/fastlivo_hip.h -- C ABI of libfastlivo_hip.so, the MI355X (gfx950) implementation of
 * FAST-LIVO's per-frame residual/Jacobian assembly + iterated error-state Kalman update (Mode-18 LIO, VIO, Mode-23) and, section
 * by section below, of the steps either side of it: the map's 5-NN search and its updates, scan undistortion and down-sampling,
 * the VIO patch selection and the visual map, and the sharded forms for several GPUs.
 *
 * Drop-in boundary.  Every entry point names the reference code whose body
 * it replaces (paths relative to the reference tree, snapshot 2024-11-08).  The reference-side
 * bindings (what a maintainer pastes into laserMapping.cpp / lidar_selection.cpp) are shown in
 * INTEGRATION.md and compiled against mock Eigen-free types in fast-livo_amd/host/.
 *
 * Conventions: plain C types only; all matrices row-major doubles; caller owns every host
 * pointer and the library retains none past the call; the library owns all device memory; a
 * handle is NOT thread-safe (one per caller thread, like the reference's single main thread,
 * src/laserMapping.cpp:1260-1264); every call returns an int32 status (0 ok, <0 HIP/usage
 * error, >0 numerical) and never throws.  There is no CPU fallback: without a HIP device
 * fl_create fails.
 */
#ifndef FASTLIVO_HIP_H
#define FASTLIVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FL_OK 0
#define FL_ERR_HIP (-1)       /* a HIP runtime call failed (see fl_last_error_string) */
#define FL_ERR_ARG (-2)       /* bad argument / call order */
#define FL_ERR_NODEVICE (-3)  /* no usable gfx950 device */
#define FL_NUM_SINGULAR 1     /* gain solve met a zero pivot */
#define FL_NUM_NONFINITE 2    /* NaN/Inf in the state delta */
#define FL_NUM_FEWPOINTS 4    /* no effective measurement */
#define FL_NUM_TIMEOUT 8      /* a bounded in-kernel hand-off wait expired: the pass was ABANDONED -- the state did not move, nothing enqueued
                                 behind it ran. The synchronous entry points (frame drivers, fl_*_iterate with info) clear the bit and re-run
                                 what was left with one launch per pass, up to three times, before it reaches the caller; status bits are
                                 sticky on the device from fl_*_begin to the read-back */
#define FL_NUM_FRAGILE 16     /* VIO: an accept test `error <= last_error` (lidar_selection.cpp:859) fell inside the rounding noise of the
                                 reference's float running sum of res^2 and was decided by replaying that sum in the reference's own
                                 arithmetic (informational; also with the patches spread over ranks by fl_p2p_*: the sum runs through the
                                 ranks; in the collective forms fl_vio_iterate_sharded / fl_vio_solve_exact every rank replays it over the
                                 all-gathered per-patch floats on every pass). Only where the per-patch errors of the other ranks are
                                 not available -- fl_vio_solve without the gather -- or under FL_ITER_FORCE it means "the reference may
                                 take the other branch here". */

#define FL_DIM18 18           /* DIM_STATE, include/common_lib.h:34 */
#define FL_DIM23 23           /* state_ikfom::DOF, include/use-ikfom.hpp:12-21 */
#define FL_NUM_MATCH_POINTS 5 /* include/common_lib.h:39 */
#define FL_SUMS18 32          /* doubles in a Mode-18 / VIO reduction record */
#define FL_SUMS23 96          /* doubles in a Mode-23 reduction record */

typedef struct fl_context *fl_handle;

/* Run-time parameters the path reads (src/laserMapping.cpp:1096-1137 readParameters,
 * src/lidar_selection.cpp:35-59 init/set_extrinsic, config/ yaml files). */
typedef struct fl_config {
    int32_t device;           /* HIP device ordinal */
    int32_t max_iterations;   /* NUM_MAX_ITERATIONS ("max_iteration") */
    int32_t img_width, img_height;
    int32_t patch_size;       /* must be 8 ("patch_size") */
    int32_t reserved0;
    double R_LI[9], t_LI[3];  /* Lidar_rot_to_IMU / Lidar_offset_to_IMU ("mapping/extrinsic_R,T") */
    double Rcl[9], Pcl[3];    /* "camera/Rcl", "camera/Pcl" */
    double fx, fy, cx, cy;    /* cam_fx .. cam_cy */
    double d[5];              /* cam_d0..d3 (+k3): radtan distortion of vk::PinholeCamera; 0 = off */
    double laser_point_cov;   /* LASER_POINT_COV */
    double img_point_cov;     /* IMG_POINT_COV */
} fl_config;

/* StatesGroup (include/common_lib.h:296-381). */
typedef struct fl_state18 {
    double rot[9];  /* rot_end, row-major */
    double pos[3];  /* pos_end */
    double vel[3];  /* vel_end */
    double bg[3];   /* bias_g */
    double ba[3];   /* bias_a */
    double grav[3]; /* gravity */
    double cov[FL_DIM18 * FL_DIM18];
} fl_state18;

/* state_ikfom (include/use-ikfom.hpp:12-21); quaternions in Eigen coeffs order x,y,z,w. */
typedef struct fl_state23 {
    double pos[3];
    double rot[4];
    double offset_R_L_I[4];
    double offset_T_L_I[3];
    double vel[3];
    double bg[3];
    double ba[3];
    double grav[3];           /* S2 of length 9.809 */
} fl_state23;

/* What one ESKF iteration reports back (the reference keeps these in globals:
 * effct_feat_num, total_residual, flg_EKF_converged, solution; laserMapping.cpp:1588-1695). */
typedef struct fl_iter_info {
    double solution[FL_DIM23];  /* state delta applied (18 used in Mode-18 / VIO) */
    double total_residual;      /* LIO: sum |pd2| ; VIO: mean squared photometric error */
    int32_t effct_feat_num;     /* LIO: effective points ; VIO: n_meas_ */
    int32_t converged;          /* LIO: flg_EKF_converged ; VIO: EKF_end */
    int32_t status;             /* FL_OK or FL_NUM_* bits */
    int32_t iterations;         /* iterations executed so far in this frame / level */
    int32_t need_search;        /* LIO: nearest_search_en for the next pass */
    int32_t stop;               /* LIO: EKF_stop_flg */
    int32_t accepted;           /* VIO: accepted (error <= last_error) solves at this level */
    int32_t reserved;
} fl_iter_info;

/* ------------------------------------------------------------------------------------------------
 * Common
 * ---------------------------------------------------------------------------------------------- */
int32_t fl_create(const fl_config *cfg, fl_handle *out);
int32_t fl_destroy(fl_handle h);
const char *fl_last_error_string(fl_handle h);
/* Run all work of this handle on an existing hipStream_t (e.g. torch's current stream, so that an
 * RCCL all-reduce enqueued by the caller is ordered with the kernels). NULL selects HIP's default
 * (null) stream. Without this call a handle runs on a private non-blocking stream. */
int32_t fl_set_stream(fl_handle h, void *hip_stream);
/* Handles on different streams of one device run concurrently. The multi-pass kernels behind fl_*_iterate(count > 1) and the
 * frame drivers need all their workgroups (<= one per compute unit) resident at once: keep at most 4 such launches in flight per
 * device. Exceeding that cannot hang -- every in-kernel wait is bounded and reports status bit 8 -- but it is slow. */
int32_t fl_sync(fl_handle h);
/* Page-locked host memory for the arrays the caller hands over every frame (scan points, neighbours, patches):
 * a staging call on such memory is a true asynchronous DMA (fl_lio_set_points of 50 k points costs the host
 * ~5 us instead of ~40 us from pageable memory). The reference fills these arrays in a conversion loop anyway
 * (PointCloudXYZI -> float xyz); it can write straight into a buffer obtained here. */
int32_t fl_host_alloc(fl_handle h, size_t bytes, void **out);
int32_t fl_host_free(fl_handle h, void *p);
/* Times the last fl_lio_iterate18 / fl_vio_iterate batch with HIP events on the handle's stream. */
int32_t fl_set_timing(fl_handle h, int32_t enable);
int32_t fl_get_last_kernel_ms(fl_handle h, float *ms);
/* The reference's per-frame timers (match_time / solve_time, src/laserMapping.cpp:1604,1729, printed at :1805) for the last
 * fl_lio_frame18_dev run under fl_set_timing(h, 1): GPU time of the search + plane-fit launches (match) and of the pass launches +
 * covariance update (solve), from HIP events behind each stage of the frame. */
typedef struct fl_frame_timing {
    float match_ms;      /* k-NN searches + plane fits */
    float solve_ms;      /* passes (residuals, rows, normal equations, gain solve, state update) + covariance update */
    float total_ms;
    int32_t searches;    /* search launches that were enqueued (a search that was not asked for is a no-op launch) */
} fl_frame_timing;
int32_t fl_get_frame_timing(fl_handle h, fl_frame_timing *out);
/* Per-handle options (all have working defaults; none is read from the environment):
 *   FL_OPT_MULTIPASS      1 (default): fl_*_iterate(count > 1) and the frame drivers run the passes of a frame segment as ONE multi-pass
 *                         launch when its workgroups fit the device; 0: always one launch per pass.
 *   FL_OPT_MAX_PRODUCERS  cap of the producer workgroups of a Mode-18 LIO pass (0 = the built-in size policy).
 *   FL_OPT_IK_PRODUCERS   the same for the Mode-23 pass (0 = 128).
 *   FL_OPT_MP_CAPACITY    workgroups of a multi-pass kernel the device is assumed to hold at once (0 = occupancy query x CUs);
 *                         lowering it sends concurrent launches down the per-pass path earlier.
 *   FL_OPT_VIO_WHOLE_CU   1 (default): a VIO multi-pass launch that has the device to itself uses the one-workgroup-per-CU register
 *                         budget; 0: always the co-resident form. Results are bit-identical either way.
 *   FL_OPT_MAILBOX        bit 0 (1): fl_vio_compute_j, bit 1 (2): fl_lio_frame18_dev get their results through a word the frame's last
 *                         kernel writes into page-locked host memory, polled by the calling thread (profiles/HISTORY.md section 4.1.1), instead
 *                         of a device-to-host copy and a stream synchronisation (-4 .. -5 us per call). Default 3. (A large copy
 *                         command enqueued right behind a call that ended this way starts ~6 us later than behind a synchronised
 *                         stream: callers that upload scans with copy commands may prefer 2.) Results are bit-identical either way.
 *   FL_OPT_SCAN_PULL      1 (default): fl_lio_frame18_dev does not copy a scan that lies in memory of fl_host_alloc -- the frame's first
 *                         search kernel fetches it over the host link (and leaves the device copy behind); 0: always a copy command.
 *                         Results are bit-identical either way.
 *   FL_OPT_INCR_SEARCH    1 (default): a k-NN search over a scan and a map that an earlier search already covered (the rematch of a
 *                         frame, KD_TREE::Nearest_Search at laserMapping.cpp:1543 for the second time) walks only the map cells within
 *                         reach of the earlier winners' new distances; 0: every search walks all 27 cells. Results are identical either way.
 *   FL_OPT_DEMOTE_AFTER   2 (default): the multi-pass kernels wait for their own workgroups, which the library admits only when they fit
 *                         beside its OWN launches; a foreign compute client on the device can still keep some of them off the chip, the
 *                         bounded wait then abandons the pass (FL_NUM_TIMEOUT, resumed per pass by the synchronous entry points). After
 *                         this many driver calls IN A ROW that ended so, the handle is demoted to one launch per pass for
 *   FL_OPT_DEMOTE_CALLS   64 (default) driver calls (doubling on every relapse, at most 1024), then it tries the multi-pass form again.
 *                         0 for FL_OPT_DEMOTE_AFTER: never demote. fl_get_diagnostics reports demotions / demoted_calls_left. Results
 *                         are bit-identical in either launch form. */
#define FL_OPT_MULTIPASS 1
#define FL_OPT_MAX_PRODUCERS 2
#define FL_OPT_IK_PRODUCERS 3
#define FL_OPT_MP_CAPACITY 4
#define FL_OPT_VIO_WHOLE_CU 5
#define FL_OPT_MAILBOX 6
#define FL_OPT_SCAN_PULL 7
#define FL_OPT_INCR_SEARCH 8
#define FL_OPT_DEMOTE_AFTER 9
#define FL_OPT_DEMOTE_CALLS 10
#define FL_OPT_VOXEL_SORT 11    /* 0 (default): fl_scan_voxel_filter orders the voxels through an occupancy bitmap over the grid's cells (8 small
                                 * launches, no sort); 1: the round-1 form (hipCUB radix sort of (voxel index, point) + scan, 13 launches), which clouds
                                 * above FL_VX_SORT_ABOVE points always take. Results are bit-identical either way. */
#define FL_OPT_MAP_INCREMENTAL 12  /* 1 (default): fl_map_add_points / fl_map_delete_boxes update the device map IN PLACE -- only the cells of the new
                                    * and of the deleted points are touched, nothing waits for the host (csrc/mapinc_kernels.h); the whole map is
                                    * compacted and re-indexed only when its arrays fill up or its cell size goes out of tune. 0: every update
                                    * compacts the map array and rebuilds the index (rounds 1-4). The map's content (fl_map_get_points) and every
                                    * search result are identical either way. */
#define FL_OPT_VIO_SPECULATE 13    /* 1: in fl_vio_compute_j / fl_vio_update_state a pass whose accept test is decided inside float rounding
                                    * noise AND accepted by the fp64 test does not wait for the reference's float running sum (~6 us): it goes ahead
                                    * and the sum's verdict is applied a pass later, with a roll-back on the (rare) disagreement. 2 (default, round
                                    * 6): as 1, and ComputeJ's three pyramid levels are ONE launch (fl_vio_compute_j, fl_vio_detect) -- a fragile
                                    * accept that ends a level goes ahead too, the next level begins on its state and takes the verdict behind its
                                    * first pass (a rejection re-opens the finished level and starts the begun one again). 0: every such pass
                                    * waits, one launch per level (round 2-4). Accept / revert sequences, states, per-level results and per-patch
                                    * errors are bit-identical in all three forms. */
#define FL_OPT_VIO_WIDE 14         /* 1 (default): VIO passes over >= 16 384 patches (any pyramid level) run with ONE PATCH PER LANE (csrc/vio_kernels.h
                                    * vio_produce_wide: shared taps and bilinear values, no cross-lane reductions) instead of a patch per 16 lanes -- at
                                    * these sizes a pass is bound by instruction issue and memory requests, not by hand-offs. 2: every pass does, at
                                    * every level and size, one launch per pass (tests); 0: never. Per-patch errors are bit-identical either way, the
                                    * fp64 sums differ in their order only. */
#define FL_OPT_DETECT_FUSED 15     /* 1 (default): fl_vio_detect is ONE enqueue -- the candidate / accepted-patch / founded / observed counts stay on the
                                    * device, launches that depend on them are sized for the number of grid cells, the result comes back through one
                                    * mailbox (csrc/api_vmap.inc). 0: the six staged calls sequenced inside the library (round 5: three returns to the
                                    * host). The visual map, the state and every count are bit-identical either way. */
int32_t fl_set_option(fl_handle h, int32_t option, int32_t value);
/* Counters of the resident-grid machinery of the multi-pass kernels.
 * ABI note: the struct carries no size member; it is 24 bytes since ABI revision 4 (16 before: the two demotion fields were appended)
 * and fl_get_diagnostics writes all of it. A caller built against an older header must be rebuilt -- fl_abi_revision() (below) is
 * the run-time check: it changes whenever a struct of this header grows or a signature changes. */
#define FL_ABI_REVISION 5
int32_t fl_abi_revision(void);
typedef struct fl_diagnostics {
    int32_t multipass_fallbacks;   /* multi-pass launches the admission check sent down the one-launch-per-pass path */
    int32_t frames_resumed;        /* frames / iterate calls resumed per pass after an ABANDONED pass (FL_NUM_TIMEOUT) */
    int32_t multipass_capacity;    /* workgroups of a multi-pass kernel the device can hold at once */
    int32_t compute_units;
    int32_t demotions;             /* times the handle was demoted to one launch per pass after repeated time-outs (FL_OPT_DEMOTE_AFTER) */
    int32_t demoted_calls_left;    /* > 0: driver calls the handle still serves per pass before it tries the multi-pass form again */
} fl_diagnostics;
int32_t fl_get_diagnostics(fl_handle h, fl_diagnostics *out);
/* Flag bit 4 of the iterate calls (FL_ITER_STAMP) is reserved for the instrumented build (include/fastlivo_hip_debug.h); this
 * library ignores it. */

/* ------------------------------------------------------------------------------------------------
 * LIO staging (both modes)
 * ---------------------------------------------------------------------------------------------- */
/* feats_down_body after downSizeFilterSurf (src/laserMapping.cpp:1398-1399): n x (x,y,z) float. */
int32_t fl_lio_set_points(fl_handle h, const float *body_xyz, int32_t n);
/* Output of the host ikd-Tree search pass, KD_TREE::Nearest_Search
 * (include/ikd-Tree/ikd_Tree.cpp:350-380, call site src/laserMapping.cpp:1543): nbr_xyz = n x 5 x 3
 * floats ascending by distance; valid[i] = (5 neighbours returned && sqdist[4] <= 5), i.e.
 * point_selected_surf[i] of src/laserMapping.cpp:1549,1567.  Fits the n planes (esti_plane,
 * include/common_lib.h:448-493) once on the device and re-arms the per-point selection flags. */
int32_t fl_lio_set_neighbours(fl_handle h, const float *nbr_xyz, const uint8_t *valid, int32_t n);
/* point_selected_surf && res_last<=2 mask and normvec (n x 4: nx,ny,nz,pd2) of the last iteration:
 * the inputs of publish_effect_world (src/laserMapping.cpp:871-885). Either pointer may be NULL. */
int32_t fl_lio_get_selection(fl_handle h, uint8_t *mask, float *normvec);
/* feats_down_world at the current device state (pointBodyToWorld, src/laserMapping.cpp:272-286):
 * what the host kNN needs on a rematch pass. */
int32_t fl_lio_get_world_points(fl_handle h, float *world_xyz);

/* ------------------------------------------------------------------------------------------------
 * LIO Mode-18: the inline ESKF loop of main(), src/laserMapping.cpp:1504-1733
 * ---------------------------------------------------------------------------------------------- */
/* Start a frame: state and state_propagat (= state, src/laserMapping.cpp:1292); zeroes G and the
 * loop counters (iterCount=-1, rematch_num=0, nearest_search_en=true; :1472-1473,1506). */
int32_t fl_lio_begin18(fl_handle h, const fl_state18 *state, const fl_state18 *state_propagat);
/* Enqueue `count` passes of the loop body (:1506-1731: residuals, Hsub^T Hsub, gain solve, state
 * += solution, rematch/stop judgement) without host round trips. Passes issued after the device
 * raised need_search or stop are no-ops, exactly where the reference would search or break.
 * info (nullable) is filled after a stream sync with the last executed pass.
 * flags: FL_ITER_FORCE ignores need_search/stop (benchmark: identical work every pass). */
#define FL_ITER_FORCE 1
#define FL_ITER_KEEP_NORMVEC 2  /* also store per-point normvec for fl_lio_get_selection */
int32_t fl_lio_iterate18(fl_handle h, int32_t count, int32_t flags, fl_iter_info *info);
/* state.cov = (I - G) * state.cov (src/laserMapping.cpp:1715) then read the state back. */
int32_t fl_lio_finish18(fl_handle h, fl_state18 *state_out);
int32_t fl_lio_get_state18(fl_handle h, fl_state18 *state_out);

/* Host kNN provider for the whole-frame driver (stands in for ikdtree.Nearest_Search). */
typedef void (*fl_knn_fn)(void *ctx, const float *world_xyz, int32_t n, float *nbr_xyz, uint8_t *valid);
/* The whole `if(lidar_en){ for(iterCount=-1; ...) }` block, src/laserMapping.cpp:1504-1733. */
int32_t fl_lio_frame18(fl_handle h, fl_state18 *state_io, const float *body_xyz, int32_t n,
                       fl_knn_fn knn, void *knn_ctx, fl_iter_info *info);

/* Sharded form: this rank's points only. accumulate writes FL_SUMS18 doubles
 * [HTH upper 21 | HTz 6 | n_eff | sum|res| | sum res^2 | 0 0] to a DEVICE buffer the caller then
 * all-reduces (sum); solve consumes the reduced record and runs the gain solve + state update
 * redundantly on every rank. */
int32_t fl_lio_accumulate18(fl_handle h, double *d_sums, int32_t flags);
int32_t fl_lio_solve18(fl_handle h, const double *d_sums, int32_t flags, fl_iter_info *info);

/* ------------------------------------------------------------------------------------------------
 * VIO: LidarSelector::ComputeJ / UpdateState, src/lidar_selection.cpp:743-983
 * ---------------------------------------------------------------------------------------------- */
/* cv::Mat img (CV_8UC1) as read at src/lidar_selection.cpp:821. stride in bytes. */
int32_t fl_vio_set_frame(fl_handle h, const uint8_t *gray, int32_t width, int32_t height, int32_t stride);
/* sub_sparse_map (include/common_lib.h:263-292): patch[i] = 3 x 64 floats indexed
 * 64*level + 8*x + y (src/lidar_selection.cpp:837), voxel_points[i]->pos_, search_levels[i]. */
int32_t fl_vio_set_patches(fl_handle h, const float *ref_patch, const double *pos,
                           const int32_t *search_level, int32_t m);
int32_t fl_vio_begin(fl_handle h, const fl_state18 *state, const fl_state18 *state_propagat);
/* UpdateState(img, total_residual, level): up to max_iterations accept/revert iterations at one
 * pyramid level, fully on the device; returns last_error through *error_out. */
int32_t fl_vio_update_state(fl_handle h, float total_residual, int32_t level, float *error_out,
                            fl_iter_info *info);
/* ComputeJ(img): levels 2,1,0, then cov -= G*cov (src/lidar_selection.cpp:967-983). */
int32_t fl_vio_compute_j(fl_handle h, fl_state18 *state_io, const fl_state18 *state_propagat,
                         fl_iter_info *info3 /* [3] by level, nullable */);
/* sub_sparse_map->errors (src/lidar_selection.cpp:851). */
int32_t fl_vio_get_errors(fl_handle h, float *errors);
/* Benchmark/sharded form: one iteration's sums at `level` for this rank's patches, and the solve. */
int32_t fl_vio_iterate(fl_handle h, int32_t level, int32_t count, int32_t flags, fl_iter_info *info);
int32_t fl_vio_accumulate(fl_handle h, int32_t level, double *d_sums);
int32_t fl_vio_solve(fl_handle h, const double *d_sums, int32_t flags, fl_iter_info *info);
/* The collective form with the reference's accept test (lidar_selection.cpp:849-861: `error` is a FLOAT running sum over the patches
 * in order): after fl_vio_accumulate every rank exports its per-patch floats as one chunk of `stride` floats
 * [patch count (int bits), patch_error[0..m-1], zero padding] (stride >= the largest patch count of any rank + 1, equal on all
 * ranks), the caller all-gathers the chunks in rank order with its transport (and all-reduces the sums), and fl_vio_solve_exact
 * decides every pass on the float chain over ALL patches, replayed on every rank (bit-identical results on all ranks, equal to the
 * single-GPU and to the in-kernel-exchange forms). fl_vio_iterate_sharded does exactly this with its RCCL communicator. */
int32_t fl_vio_errors_chunk(fl_handle h, float *d_chunk /* device, stride floats */, int32_t stride);
int32_t fl_vio_solve_exact(fl_handle h, const double *d_sums, int32_t flags, const float *d_all_chunks /* device, world x stride */,
                           int32_t stride, int32_t world, fl_iter_info *info);
int32_t fl_vio_get_state18(fl_handle h, fl_state18 *state_out);

/* ------------------------------------------------------------------------------------------------
 * LIO Mode-23 (IKFoM): h_share_model, src/laserMapping.cpp:961-1093, and
 * esekf::update_iterated_dyn_share_modified, include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928
 * ---------------------------------------------------------------------------------------------- */
/* kf.change_x / change_P + the x_propagated/P_propagated copies made at esekfom.hpp:1625-1626. */
int32_t fl_ikfom_begin(fl_handle h, const fl_state23 *x, const double *P /* 23x23 */,
                       const double *limit /* 23, init_dyn_share's limit[] */);
/* "sum-compat" body of h_share_model: residuals + 12-wide rows reduced on the device to
 * h_x^T h_x (12x12 row-major), h_x^T h (12) and effct_feat_num at state s. The caller shim turns
 * them into the surrogate h_x/h described in INTEGRATION.md so esekfom.hpp runs unmodified. */
int32_t fl_h_share_model_sums(fl_handle h, const fl_state23 *s, double *HTH12, double *HTh12,
                              int32_t *effct_feat_num, double *total_residual);
/* "row-compat" body (parity/debug): fills h_x (n_eff x 12, row-major) and h (n_eff) in the
 * reference's order (ascending point index). h_x/h must hold n rows. */
int32_t fl_h_share_model_rows(fl_handle h, const fl_state23 *s, double *h_x, double *hvec,
                              int32_t *effct_feat_num);
/* feats_down_world at state s (the transform at src/laserMapping.cpp:980-984): what the host kNN of a
 * `converge` pass searches with. Does not touch the filter state of the handle. */
int32_t fl_ikfom_world_points(fl_handle h, const fl_state23 *s, float *world_xyz);
/* `count` passes of the body of update_iterated_dyn_share_modified's loop on the device. */
int32_t fl_ikfom_iterate(fl_handle h, int32_t count, int32_t flags, fl_iter_info *info);
/* The final covariance block (esekfom.hpp:1831-1924) runs inside the pass that finishes; this
 * reads x_ and P_ back. */
int32_t fl_ikfom_get(fl_handle h, fl_state23 *x_out, double *P_out);
/* The whole update_iterated_dyn_share_modified(R, solve_time) with a host kNN provider. */
int32_t fl_ikfom_update_iterated(fl_handle h, fl_state23 *x_io, double *P_io, const float *body_xyz,
                                 int32_t n, double R, const double *limit, fl_knn_fn knn,
                                 void *knn_ctx, fl_iter_info *info);
int32_t fl_ikfom_accumulate(fl_handle h, double *d_sums, int32_t flags);
int32_t fl_ikfom_solve(fl_handle h, const double *d_sums, int32_t flags, fl_iter_info *info);

/* ------------------------------------------------------------------------------------------------
 * Device k-NN: the LiDAR-map search of the 2 search passes per frame
 * on the GPU instead of the host ikd-Tree (KD_TREE::Nearest_Search, include/ikd-Tree/ikd_Tree.cpp:350-380;
 * call sites src/laserMapping.cpp:1543 and :1002). Exact 5-NN with the tree's float distance
 * (ikd_Tree.cpp:1291-1295), ascending; exact ties are broken by the lower map index.
 * ---------------------------------------------------------------------------------------------- */
/* Mirror of the host map on the device: call after ikdtree.Build (:1411-1419) and after map_incremental
 * (:692-706) with the current map points (k x 3 floats). cell_size (m) is the voxel edge of the device
 * grid: 2-3x the map's point spacing. cell_size <= 0: automatic -- the library keeps the points per occupied cell
 * between 5 and 20 by moving the edge in steps of 1.5x whenever the map is (re)built; results do not depend on it
 * (the search is exact for any cell), its cost does (a 0.5 m cell on a map thinned to 0.5 m spacing: 10x slower).
 * An explicit edge below 0.02 m is raised to 0.02 m. */
int32_t fl_map_set_points(fl_handle h, const float *map_xyz, int32_t k, float cell_size);
/* The map kept ON the device between frames (the map side of rows N1/N3): instead of re-staging the host map after every
 * map_incremental, update the device copy in place. The array order is part of the contract (ties of the k-NN go to the lower
 * index): surviving points keep their order, added points follow in input order.
 *   fl_map_clear         an empty map with the given k-NN cell size, <= 0: automatic (first frame, before ikdtree.Build's points arrive)
 *   fl_map_add_points    map_incremental (src/laserMapping.cpp:692-706): KD_TREE::Add_Points(points, downsample_on = true)
 *                        (include/ikd-Tree/ikd_Tree.cpp:382-457) with downsample_size = the tree's (filter_size_map_min,
 *                        laserMapping.cpp:1410). Per down-sampling box [floor(p/ds)*ds, +ds) a new point falls into, the box ends
 *                        with exactly one point: the old point closest to the box centre if strictly closer than every new point
 *                        of the box, else the closest new point, the latest among equals (the sequential loop's outcome).
 *                        downsample_size <= 0: every point is appended (Add_Points(.., false) / Build, :1411-1419).
 *                        world_xyz == NULL: the scan staged on the device, body -> world under the filter state the device holds
 *                        (pointBodyToWorld, :695-698; the 18-state or, after fl_ikfom_*, the state_ikfom) -- after
 *                        fl_lio_frame18_dev / fl_ikfom_update_iterated_dev that is the updated state, as in the reference.
 *   fl_map_delete_boxes  lasermap_fov_segment (:363-417): KD_TREE::Delete_Point_Boxes (ikd_Tree.cpp:501-520); boxes = nb x
 *                        {min x,y,z, max x,y,z} floats (BoxPointType), a point goes iff min <= v && max > v on every axis (:650).
 *   fl_map_get_points    read the map array back (parity checks, publishing); cap = 0 queries the size.
 * Each update rebuilds the k-NN index of the whole map and costs one host synchronisation (the new size). */
typedef struct fl_map_info {
    int32_t n_before, n_after;
    int32_t n_added;          /* new points that went in */
    int32_t n_removed;        /* old points that left */
    int32_t n_ambiguous;      /* points whose box membership depends on the rounding of floor(v/ds)*ds (the reference tests
                                 coordinates against the box, this library partitions by floor(v/ds)): results may differ from the
                                 sequential reference for these points only. 0 for power-of-two ds; ~1e-7 per coordinate otherwise */
    int32_t status;           /* FL_NUM_NONFINITE: a coordinate outside +-2^20 boxes */
    float cell_size;          /* k-NN cell edge the index was rebuilt with (moves only in automatic mode) */
} fl_map_info;
int32_t fl_map_clear(fl_handle h, float cell_size);
int32_t fl_map_add_points(fl_handle h, const float *world_xyz, int32_t n, float downsample_size, fl_map_info *info);
int32_t fl_map_delete_boxes(fl_handle h, const float *boxes, int32_t nb, fl_map_info *info);
int32_t fl_map_get_points(fl_handle h, float *xyz_out, int32_t cap, int32_t *n_out);
/* The in-place updates (FL_OPT_MAP_INCREMENTAL 1) leave tombstones in the map array and holes in the index's point pool; when those fill
 * up, the update that finds them full first compacts the array and re-indexes the map -- an O(map) step inside that frame's
 * fl_map_add_points. fl_map_compact does that step NOW (no-op if nothing is dead), so that a caller can pay for it where the frame has
 * slack (e.g. while the camera half runs) instead of in the frame that happens to hit it. Map content and search results are unchanged.
 * bench.py's map_scale section times it per map size (the worst-case frame of the in-place form). */
int32_t fl_map_compact(fl_handle h);
/* One search pass at the current device state (after fl_lio_begin18 / fl_ikfom_begin): neighbours ->
 * planes -> selection flags, nothing leaves the device. nbr_xyz_out (n x 5 x 3) / valid_out (n) are
 * optional read-backs for parity checks. */
int32_t fl_lio_search18(fl_handle h, float *nbr_xyz_out, uint8_t *valid_out);
int32_t fl_ikfom_search(fl_handle h, float *nbr_xyz_out, uint8_t *valid_out);
/* fl_lio_frame18 / fl_ikfom_update_iterated with the search on the device: the whole frame is enqueued
 * at once ([search-if-asked, pass] x (max_iterations + 1)), one host synchronisation per frame.
 * body_xyz == NULL: use the scan already staged on the device (fl_lio_set_points / fl_scan_voxel_filter). */
int32_t fl_lio_frame18_dev(fl_handle h, fl_state18 *state_io, const float *body_xyz, int32_t n, fl_iter_info *info);
int32_t fl_ikfom_update_iterated_dev(fl_handle h, fl_state23 *x_io, double *P_io, const float *body_xyz, int32_t n, double R,
                                     const double *limit, fl_iter_info *info);

/* ------------------------------------------------------------------------------------------------
 * Scan voxel down-sampling on the device: pcl::VoxelGrid<PointType>::filter as called at
 * src/laserMapping.cpp:1398-1399 (downSizeFilterSurf, leaf = filter_size_surf, :1186) and
 * src/lidar_selection.cpp:352-353 (leaf 0.2, :7). xyzi: n x 4 floats (x, y, z, intensity). One output point per
 * occupied voxel, in ascending voxel index (PCL's output order): the centroid of its points, summed in float in
 * ascending input index (PCL's std::sort leaves that order unspecified) and divided by the count.
 * stage_as_scan != 0: the centroids become the staged scan of this handle, exactly as if fl_lio_set_points had been
 * called with them (feats_down_body never visits the host). out_xyzi (nullable) has room for n points.
 * xyzi == NULL: filter the n-point cloud that fl_imu_undistort left on the device.
 * leaf_too_small (nullable) reports PCL's "Leaf size is too small for the input dataset" case (output = input).
 * Two deviations from PCL a caller should know: (1) in that pass-through case PCL copies EVERY input point, non-finite ones included;
 * this library drops non-finite points there as it does everywhere else; (2) inside a voxel PCL adds the points in the order its
 * (unstable) std::sort left them, here in ascending input index -- a centroid can differ from PCL's in its last float bit.
 * ---------------------------------------------------------------------------------------------- */
int32_t fl_scan_voxel_filter(fl_handle h, const float *xyzi, int32_t n, float leaf_x, float leaf_y, float leaf_z,
                             int32_t stage_as_scan, float *out_xyzi, int32_t *out_n, int32_t *leaf_too_small);

/* ------------------------------------------------------------------------------------------------
 * IMU forward propagation + point undistortion on the device:
 * ImuProcess::UndistortPcl(LidarMeasureGroup&, StatesGroup&, PointCloudXYZI&), src/IMU_Processing.cpp:611-809,
 * as called from Process2 (:875). The choice of the frame's points and of pcl_beg_time / pcl_end_time (:620-648)
 * is bookkeeping on LidarMeasureGroup and stays with the caller.
 * fl_imu_proc holds the ImuProcess members the function reads and writes (include/IMU_Processing.h); imu = meas.imu
 * (n_imu >= 1 samples, last_imu_ is taken from proc). pts_xyzt: n x 4 floats (x, y, z, curvature = offset in ms).
 * On return: state (rot/pos/vel = *_end, cov propagated), proc (acc_s_last, angvel_last, last_imu_,
 * last_lidar_end_time_), out_xyzt (nullable) the compensated cloud, poses_out (nullable, room for n_imu + 1) the
 * IMUpose list. The compensated cloud also stays on the device: fl_scan_voxel_filter(h, NULL, n, ...) continues
 * from it. Unsorted clouds, points not later than IMUpose[0] and the repeated compensation of the first point
 * behave exactly as the reference's backward loops do (see imu_kernels.h).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fl_imu_sample { double t; double gyr[3]; double acc[3]; } fl_imu_sample;   /* sensor_msgs::Imu fields used */
typedef struct fl_pose6d { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; } fl_pose6d;  /* Pose6D, common_lib.h:396-412 */
typedef struct fl_imu_proc {
    double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3];   /* IMU_Processing.cpp:15-20, after IMU_init scaling (:829-836) */
    double mean_acc[3];                                                /* :21, :113 */
    double Lid_rot_to_IMU[9], Lid_offset_to_IMU[3];                    /* set_extrinsic :59-63 */
    double acc_s_last[3], angvel_last[3];
    fl_imu_sample last_imu;
    double last_lidar_end_time;
} fl_imu_proc;
int32_t fl_imu_undistort(fl_handle h, fl_imu_proc *proc_io, fl_state18 *state_io, const fl_imu_sample *imu, int32_t n_imu,
                         double pcl_beg_time, double pcl_end_time, const float *pts_xyzt, int32_t n, float *out_xyzt,
                         fl_pose6d *poses_out, int32_t *n_poses_out);

/* ------------------------------------------------------------------------------------------------
 * The LiDAR half of a frame in ONE enqueue (round 5): what laserMapping.cpp does between the sync of a measurement group and the
 * map update -- p_imu->Process2 / UndistortPcl (src/IMU_Processing.cpp:611-809), downSizeFilterSurf (src/laserMapping.cpp:1398-1399,
 * leaf = filter_size_surf), the Mode-18 iterated update over the device map (:1504-1733) -- i.e. the three calls
 *     fl_imu_undistort(h, proc, state, imu, n_imu, beg, end, pts, n, NULL, NULL, &k);
 *     fl_scan_voxel_filter(h, NULL, n, leaf, leaf, leaf, 1, NULL, &m, &small);
 *     fl_lio_frame18_dev(h, state, NULL, 0, info);
 * with identical results (every bit of the state, the covariance and the selection), but nothing returns to the host in between:
 * the propagated state goes into the update's state block on the device, the filtered scan's size stays on the device, the result
 * comes back through the mailbox of the frame's last kernel (csrc/api_front.inc). scan_points_out (nullable): feats_down_size.
 * Needs a map (fl_map_set_points / fl_map_add_points). flags: FL_FRONT_STAGED runs the three calls above instead (A/B, tests); the
 * library also falls back to them by itself when the frame cannot be fused (no multi-pass admission, FL_OPT_VOXEL_SORT, a sharded handle).
 * ---------------------------------------------------------------------------------------------- */
#define FL_FRONT_STAGED 1
int32_t fl_lidar_front(fl_handle h, fl_imu_proc *proc_io, fl_state18 *state_io, const fl_imu_sample *imu, int32_t n_imu,
                       double pcl_beg_time, double pcl_end_time, const float *pts_xyzt, int32_t n, float leaf_size, int32_t flags,
                       fl_iter_info *info, int32_t *scan_points_out);

/* ------------------------------------------------------------------------------------------------
 * VIO patch selection + affine warp on the device (pixel-level part of
 * LidarSelector::addFromSparseMap, src/lidar_selection.cpp:346-587): depth image of the scan (:376-410) and, per
 * candidate, depth-continuity test (:484-506), getWarpMatrixAffine (:232-256), getBestSearchLevel (:315-329),
 * warpAffine x 3 levels (:258-296), getpatch of the current image (:119-140), NCC gate (:298-313, :559-563),
 * squared-error gate (:565-570). The caller keeps the walk over the visual map: voxel lookups + grid competition
 * (:412-466) and Point::getCloseViewObs (src/point.cpp:141-178) produce one candidate per winning grid cell, in
 * ascending grid index. Reference images (Feature::img) are registered once with fl_vio_add_keyframe and stay in
 * HBM until dropped. The current image is the one staged by fl_vio_set_frame; Rcw/Pcw = new_frame_->T_f_w_.
 * Accepted candidates (ascending input order, :572-579) become the staged VIO patch set, exactly as after
 * fl_vio_set_patches(ref = warped patches, pos = pt->pos_, search_level): fl_vio_compute_j can follow directly and
 * the 768-byte patches never visit the host. Outputs (all nullable except n_accepted): accepted_idx / errors /
 * search_levels (room for m), reason (m: 0 accepted, 1 depth discontinuity, 3 NCC, 4 outlier), patches_out
 * (m x 192 floats, parity checks), depth_out (width x height floats). With distortion coefficients in fl_config, cam2world is
 * cv::undistortPoints as vikit calls it (five fixed-point sweeps on the float pixel).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fl_patch_candidate {
    double pos[3];             /* pt->pos_ */
    double px_ref[2];          /* ref_ftr->px */
    double f_ref[3];           /* ref_ftr->f */
    double R_ref[9], t_ref[3]; /* ref_ftr->T_f_w_ */
    int32_t keyframe_id;       /* id of ref_ftr->img from fl_vio_add_keyframe; also the key of the reference's Warp_map (ref_ftr->id_:
                                  one image per frame), i.e. candidates of one keyframe share the warp of the first of them (:530-546) */
    int32_t level_ref;         /* ref_ftr->level (not read by warpAffine) */
    int32_t grid_index;        /* caller's bookkeeping */
    int32_t reserved;
} fl_patch_candidate;
/* Projection + grid competition of addFromSparseMap (:412-466) over a flat list of the map points of the visible voxels, in the
 * order the reference's loops visit them (pos k x 3 doubles, value k floats = Point::value). Arrays of
 * length (width/grid_size)*(height/grid_size): winner = index into the list of the point a cell keeps (voxel_points_[index],
 * the closest one, the later one on equal float distances; -1: none), map_dist, map_value, grid_num (1 TYPE_MAP, 3 TYPE_UNKNOWN). */
int32_t fl_vio_grid_select(fl_handle h, const double *Rcw, const double *Pcw, const double *pos, const float *value, int32_t k,
                           int32_t grid_size, int32_t *winner, float *map_dist, float *map_value, int32_t *grid_num, int32_t *length_out);
/* gray == NULL: the current image staged by fl_vio_set_frame becomes the keyframe (device-to-device copy, no second upload) */
int32_t fl_vio_add_keyframe(fl_handle h, const uint8_t *gray, int32_t width, int32_t height, int32_t stride, int32_t *keyframe_id);
/* refused (FL_ERR_ARG) while observations of the device visual map (fl_vmap_*) still refer to the keyframe */
int32_t fl_vio_drop_keyframe(fl_handle h, int32_t keyframe_id);
int32_t fl_vio_select_patches(fl_handle h, const double *Rcw, const double *Pcw, const float *scan_world_xyz, int32_t n_scan,
                              const fl_patch_candidate *cand, int32_t m, int32_t ncc_en, double ncc_thre, double outlier_threshold,
                              int32_t *accepted_idx, float *errors, int32_t *search_levels, int32_t *n_accepted, int32_t *reason,
                              float *patches_out, float *depth_out);

/* ------------------------------------------------------------------------------------------------
 * The visual map on the device: the caller's side of addFromSparseMap, so that no per-point visual-map data crosses PCIe.
 * A map point = position, Shi-Tomasi score and up to 20 observations {px, f, T_f_w, score, level, keyframe id, frame id} in the
 * order of the reference's list (front = newest). Per frame, in the order of LidarSelector::detect (:1050-1064), with the
 * current image staged by fl_vio_set_frame and registered once with fl_vio_add_keyframe (its id is what new observations refer to):
 *   fl_vmap_select           addFromSparseMap (src/lidar_selection.cpp:346-587), whole: scan_down_world_xyz = the scan after
 *                            downSizeFilter (:352-353, fl_scan_voxel_filter); voxels the scan touches (:383-391), depth image,
 *                            the map points of those voxels projected + grid competition (:412-466), Point::getCloseViewObs
 *                            (src/point.cpp:141-178) per winning cell, then warp / NCC / outlier gates as fl_vio_select_patches.
 *                            The accepted patches are the staged VIO patch set (fl_vio_compute_j can follow); sel_point (nullable,
 *                            room for (W/grid)*(H/grid)) = their map indices (sub_sparse_map->voxel_points), errors / search_levels
 *                            alike, patches_out n_selected x 192 floats. Rcw/Pcw = new_frame_->T_f_w_ at that moment (LIO posterior).
 *   fl_vmap_add_sparse       addSparseMap (:142-197) + AddPoint (:199-230): scan_world_xyz = the scan `pg` itself; a scan point founds
 *                            a map point where its score beats every map point that projected into its grid cell during
 *                            fl_vmap_select (map_value is not reset in between, :81-90).
 *   fl_vmap_add_observation  addObservation (:913-965) for the points selected by the last fl_vmap_select, with the pose AFTER
 *                            fl_vio_compute_j.
 * fl_vmap_clear(grid_size) creates the (empty) map; grid_size as LidarSelector::grid_size.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fl_vmap_obs { double px[2], f[3], R[9], t[3]; float score; int32_t level, kf_id, frame_id; } fl_vmap_obs;
int32_t fl_vmap_clear(fl_handle h, int32_t grid_size);
int32_t fl_vmap_size(fl_handle h, int32_t *n);
int32_t fl_vmap_get_point(fl_handle h, int32_t i, double *pos, float *value, int32_t *n_obs, fl_vmap_obs *obs /* room for 20 */);
int32_t fl_vmap_select(fl_handle h, const double *Rcw, const double *Pcw, const float *scan_down_world_xyz, int32_t n, int32_t ncc_en,
                       double ncc_thre, double outlier_threshold, int32_t *n_selected, int32_t *sel_point, float *errors, int32_t *search_levels,
                       float *patches_out);
int32_t fl_vmap_add_sparse(fl_handle h, const double *Rcw, const double *Pcw, const float *scan_world_xyz, int32_t n, int32_t keyframe_id,
                           int32_t frame_id, int32_t *n_added);
int32_t fl_vmap_add_observation(fl_handle h, const double *Rcw, const double *Pcw, int32_t keyframe_id, int32_t frame_id, int32_t *n_added);

/* LidarSelector::detect in one call (src/lidar_selection.cpp:1027-1076): addFromSparseMap -> addSparseMap -> ComputeJ -> addObservation
 * on the device's visual map, i.e. fl_vio_set_frame + fl_vio_add_keyframe (the staged image) + fl_vmap_select + fl_vmap_add_sparse +
 * fl_vio_compute_j + fl_vmap_add_observation, the frame pose taken from state_io before and after ComputeJ as updateFrameState does
 * (:904-911; Rci, Pci: camera extrinsics of the state frame, lidar_selection.cpp:35-52). pg_world_xyz: the registered scan (n_pg x 3),
 * pg_down_world_xyz: its 0.2 m down-sampled form (:352-353). state_io: in = state_propagat = the LIO result, out = after ComputeJ.
 * ONE enqueue and one wait (FL_OPT_DETECT_FUSED, csrc/api_vmap.inc): the counts that size the steps stay on the device and come back with
 * the state; an image in fl_host_alloc memory (rows contiguous, 16-byte aligned) is fetched by the frame's first kernel instead of copied. */
/* n_pg = FL_DETECT_SCAN_ON_DEVICE (pg_world_xyz, pg_down_world_xyz ignored): the registered scan never leaves the device -- pg = the scan this
 * handle holds (fl_lio_set_points / fl_lio_frame18_dev / fl_lidar_front: feats_down_body) under state_io (pointBodyToWorld, laserMapping.cpp:695-698,
 * exactly what fl_lio_get_world_points returns), its 0.2 m down-sampling (downSizeFilter, lidar_selection.cpp:7,352-353) by the device voxel filter
 * with the count kept on the device. Same results as handing both clouds in. */
#define FL_DETECT_SCAN_ON_DEVICE (-1)
#define FL_DETECT_DOWN_LEAF 0.2f
int32_t fl_vio_detect(fl_handle h, const uint8_t *gray, int32_t width, int32_t height, int32_t stride, const float *pg_world_xyz, int32_t n_pg,
                      const float *pg_down_world_xyz, int32_t n_down, const double *Rci, const double *Pci, fl_state18 *state_io,
                      int32_t frame_id, int32_t ncc_en, double ncc_thre, double outlier_threshold, int32_t *n_selected, int32_t *n_founded,
                      int32_t *n_observed);
/* Frees the keyframe images (fl_vio_add_keyframe) that no observation of the map refers to any more -- what the reference's
 * reference counting of Feature::img does; call now and then on long runs (one image per frame is 0.3 MB). */
int32_t fl_vmap_release_keyframes(fl_handle h, int32_t *n_released);

/* ------------------------------------------------------------------------------------------------
 * Sharded form with the exchange done natively: each rank stages its contiguous range of the scan
 * points / patches; a pass = accumulate (this rank's range) -> ncclAllReduce of the 32-double (Mode-23: 96) record
 * on the handle's stream (RCCL over xGMI) -> solve, replicated on bitwise-identical inputs. Three enqueues, no host
 * work in between. fl_comm_unique_id on one rank, distributed to all by the caller (e.g. torch.distributed
 * broadcast), then fl_comm_init on every rank (collective: blocks until all ranks have called it).
 * RCCL is bound at run time (the librccl.so.1 already loaded in the process, else /opt/rocm/lib).
 * ---------------------------------------------------------------------------------------------- */
int32_t fl_comm_unique_id(fl_handle h, void *id128 /* 128 bytes out */);
int32_t fl_comm_init(fl_handle h, const void *id128, int32_t rank, int32_t world);
int32_t fl_comm_destroy(fl_handle h);
int32_t fl_lio_iterate18_sharded(fl_handle h, int32_t count, int32_t flags, fl_iter_info *info);
int32_t fl_vio_iterate_sharded(fl_handle h, int32_t level, int32_t count, int32_t flags, fl_iter_info *info);
int32_t fl_ikfom_iterate_sharded(fl_handle h, int32_t count, int32_t flags, fl_iter_info *info);

/* ------------------------------------------------------------------------------------------------
 * Sharded form with the exchange INSIDE the pass kernels ("one-shot P2P all-gather ... each rank writes its
 * record into the peers' mapped buffers + flag; fixed-order local sum"). After connecting, the ordinary Mode-18 / VIO entry
 * points (fl_lio_iterate18, fl_lio_frame18_dev, fl_vio_iterate, fl_vio_compute_j, ...) run on this rank's range of the
 * points / patches: the solver workgroup of every pass publishes its 32 sums into every peer's exchange buffer (fine-grained
 * device memory, 8-byte self-validating words, over xGMI between GPUs) and adds up what the peers sent in rank order, so all
 * ranks solve on bitwise-identical totals -- and the passes of a frame remain ONE launch per rank. The Mode-23 entry points
 * (fl_ikfom_begin after the connect, fl_ikfom_iterate, fl_ikfom_update_iterated_dev) exchange their 96 sums the same way, as three
 * exchanges of 32 per pass. Every rank must issue the
 * same sequence of passes (it is a collective). State, covariance, configuration replicated; map / image replicated or sharded
 * by the caller. The VIO accept test stays the reference's: on the fragile passes the float running sum over the patches is handed
 * from rank to rank (rank r continues from the float rank r-1 ended with; the last rank sends the total back), so accept / revert
 * sequences equal the single-GPU ones. (The collective forms fl_vio_iterate_sharded / fl_vio_solve_exact all-gather the per-patch
 * floats and replay the chain on every rank: the same decisions. Only fl_vio_solve without the gather keeps the fp64 comparison:
 * FL_NUM_FRAGILE = "may differ" there.)
 *   separate processes:  fl_p2p_export on every rank -> exchange the 64-byte handles (any transport) -> fl_p2p_connect
 *   one process:         fl_p2p_connect_local(h, rank, world, all_handles) on every handle
 * Connect before fl_*_begin of the frame, and put a barrier of the caller's transport between the connects and the first pass
 * (connecting clears this rank's buffer and restarts the exchange epochs: no peer may still be writing an older session into it).
 * Waiting for a peer is bounded (seconds): FL_NUM_TIMEOUT in the status, no hang. A timed-out pass is abandoned (state untouched)
 * but NOT resumed automatically in the sharded form -- one rank may have completed the pass its peer abandoned -- so after
 * FL_NUM_TIMEOUT the ranks connect again (fl_p2p_connect) and restart the frame together.
 * A pass kernel waits for its peers' kernels, so every rank must issue its passes in the same order and nothing a rank's kernel
 * waits for may be queued BEHIND it: give all connected handles of a process one stream (fl_set_stream; bench.py does), and when
 * several ranks share one device (tests) one stream per rank -- HIP multiplexes streams onto a few hardware queues.
 * A replicated device map stays replicated only if every rank adds the WHOLE registered scan: fl_map_add_points(h, NULL, ..)
 * adds this rank's range only -- pass the full world scan explicitly there.
 * ---------------------------------------------------------------------------------------------- */
int32_t fl_p2p_export(fl_handle h, int32_t world /* 2..8 */, void *handle64_out /* 64 bytes */);
int32_t fl_p2p_connect(fl_handle h, int32_t rank, int32_t world, const void *handles64 /* world x 64 bytes, own entry ignored */);
int32_t fl_p2p_connect_local(fl_handle h, int32_t rank, int32_t world, const fl_handle *all /* world handles, all[rank] == h */);
int32_t fl_p2p_disconnect(fl_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FASTLIVO_HIP_H */
