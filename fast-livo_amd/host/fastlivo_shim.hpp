// fastlivo_shim.hpp -- host-side bodies for the reference's entry points, over the C ABI.
//
// These are the functions whose bodies a maintainer replaces in the reference:
//   h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&)   src/laserMapping.cpp:961
//   the Mode-18 LIO block of main()                                       src/laserMapping.cpp:1504-1733
//   LidarSelector::ComputeJ(cv::Mat)                                      src/lidar_selection.cpp:967
// Names, argument meaning and error behaviour follow the reference (void / silent in the reference;
// here the int32 status of the C ABI is kept in last_status and never thrown).
#pragma once

#include "../../include/fastlivo_hip.h"
#include "fastlivo_types.hpp"

#include <cmath>
#include <cstdint>
#include <vector>

namespace fastlivo_host {

inline void to_abi(const StatesGroup &s, fl_state18 &o)
{
    std::memcpy(o.rot, s.rot_end.m, sizeof o.rot);
    std::memcpy(o.pos, s.pos_end.v, sizeof o.pos);
    std::memcpy(o.vel, s.vel_end.v, sizeof o.vel);
    std::memcpy(o.bg, s.bias_g.v, sizeof o.bg);
    std::memcpy(o.ba, s.bias_a.v, sizeof o.ba);
    std::memcpy(o.grav, s.gravity.v, sizeof o.grav);
    std::memcpy(o.cov, s.cov, sizeof o.cov);
}
inline void from_abi(const fl_state18 &o, StatesGroup &s)
{
    std::memcpy(s.rot_end.m, o.rot, sizeof o.rot);
    std::memcpy(s.pos_end.v, o.pos, sizeof o.pos);
    std::memcpy(s.vel_end.v, o.vel, sizeof o.vel);
    std::memcpy(s.bias_g.v, o.bg, sizeof o.bg);
    std::memcpy(s.bias_a.v, o.ba, sizeof o.ba);
    std::memcpy(s.gravity.v, o.grav, sizeof o.grav);
    std::memcpy(s.cov, o.cov, sizeof o.cov);
}
inline void to_abi(const state_ikfom &s, fl_state23 &o)
{
    std::memcpy(o.pos, s.pos.v, sizeof o.pos);
    o.rot[0] = s.rot.x; o.rot[1] = s.rot.y; o.rot[2] = s.rot.z; o.rot[3] = s.rot.w;
    o.offset_R_L_I[0] = s.offset_R_L_I.x; o.offset_R_L_I[1] = s.offset_R_L_I.y; o.offset_R_L_I[2] = s.offset_R_L_I.z; o.offset_R_L_I[3] = s.offset_R_L_I.w;
    std::memcpy(o.offset_T_L_I, s.offset_T_L_I.v, sizeof o.offset_T_L_I);
    std::memcpy(o.vel, s.vel.v, sizeof o.vel);
    std::memcpy(o.bg, s.bg.v, sizeof o.bg);
    std::memcpy(o.ba, s.ba.v, sizeof o.ba);
    std::memcpy(o.grav, s.grav.v, sizeof o.grav);
}

// Symmetric eigen-decomposition (cyclic Jacobi) of a 12x12 matrix: A = V diag(w) V^T.
inline void jacobi_eig12(const double *A_in, double *w, double *V)
{
    const int n = 12;
    double A[144];
    std::memcpy(A, A_in, sizeof A);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

// ------------------------------------------------------------------------------------------------
// h_share_model, "sum-compat" body: signature unchanged; laserMapping.cpp and
// esekfom.hpp stay untouched. The IKFoM updater consumes h_x only through h_x^T h_x and h_x^T h when
// rows >= 23 (esekfom.hpp:1781,1801,1806), so the callback returns a 23x12 surrogate S with
// S^T S = H^T H (rows 0..11 = sqrt(Lambda) V^T, rows 12..22 zero) and h with S^T h = H^T z.
// The kNN of the `converge` pass stays on the host (ikd-Tree) and is staged through the handle.
// ------------------------------------------------------------------------------------------------
struct HShareContext {
    fl_handle handle = nullptr;
    fl_knn_fn knn = nullptr;          // stands for ikdtree.Nearest_Search over all points
    void *knn_ctx = nullptr;
    int n = 0;
    std::vector<float> world, nbr;
    std::vector<uint8_t> valid;
    int32_t last_status = 0;
    int effct_feat_num = 0;
    double total_residual = 0.0;
};

inline void h_share_model(state_ikfom &s, esekfom::dyn_share_datastruct<double> &ekfom_data, HShareContext &ctx)
{
    fl_state23 st;
    to_abi(s, st);
    double HTH[144], HTh[12];
    if (ekfom_data.converge) {   // laserMapping.cpp:994-1013: redo the kNN at the current state
        ctx.world.resize((size_t)ctx.n * 3); ctx.nbr.resize((size_t)ctx.n * 15); ctx.valid.resize((size_t)ctx.n);
        ctx.last_status = fl_ikfom_world_points(ctx.handle, &st, ctx.world.data());       // :980-984 on the device
        ctx.knn(ctx.knn_ctx, ctx.world.data(), ctx.n, ctx.nbr.data(), ctx.valid.data());  // ikdtree.Nearest_Search
        ctx.last_status |= fl_lio_set_neighbours(ctx.handle, ctx.nbr.data(), ctx.valid.data(), ctx.n);
    }
    int32_t neff = 0;
    ctx.last_status |= fl_h_share_model_sums(ctx.handle, &st, HTH, HTh, &neff, &ctx.total_residual);
    ctx.effct_feat_num = neff;
    // FAST-LIVO's h_share_model has NO early-out for effct_feat_num < 1 (FAST-LIO's sets valid = false there; laserMapping.cpp:1040-1060
    // resizes h_x to 0 x 12 and returns with valid untouched): the updater then runs its N x N branch over empty matrices, K_h = 0,
    // K_x = 0, dx = -dx_new. The all-zero surrogate drives the rows >= 23 branch to exactly that (H^T H = 0, H^T z = 0).
    ekfom_data.h_x.resize(23, 12);
    ekfom_data.h.resize(23);
    if (neff < 1) return;
    double w[12], V[144];
    jacobi_eig12(HTH, w, V);
    for (int k = 0; k < 12; k++) {
        const double lam = w[k] > 0 ? w[k] : 0.0, sq = std::sqrt(lam);
        double proj = 0.0;                       // (V^T HTz)_k
        for (int j = 0; j < 12; j++) {
            ekfom_data.h_x(k, j) = sq * V[j * 12 + k];
            proj += V[j * 12 + k] * HTh[j];
        }
        ekfom_data.h(k) = (sq > 1e-150) ? proj / sq : 0.0;
    }
}

// The callback with the reference's exact signature -- `typedef void measurementModel_dyn_share(state &,
// dyn_share_datastruct<scalar_type> &)` (esekfom.hpp:129) -- as laserMapping.cpp defines it (:961) and registers it
// (`kf.init_dyn_share(get_f, df_dx, df_dw, h_share_model, NUM_MAX_ITERATIONS, epsi)`, :1233-1235).  The reference's body reads
// file-scope globals (feats_down_body, ikdtree, Nearest_Points, point_selected_surf, ...); this one reads the file-scope context
// below, which the frame loop fills where the reference fills those globals (handle + kNN provider once, n per frame after
// fl_lio_set_points).
inline HShareContext g_hshare;
inline void h_share_model(state_ikfom &s, esekfom::dyn_share_datastruct<double> &ekfom_data) { h_share_model(s, ekfom_data, g_hshare); }

// ------------------------------------------------------------------------------------------------
// Mode-18 LIO block of main(): `if(lidar_en){ for(iterCount=-1; ...) {...} }`, laserMapping.cpp:1504-1733
// ------------------------------------------------------------------------------------------------
struct LioMode18 {
    fl_handle handle = nullptr;
    fl_knn_fn knn = nullptr;
    void *knn_ctx = nullptr;
    int32_t last_status = 0;
    int effct_feat_num = 0;
    double total_residual = 0.0;
    int iterCount = 0;

    // state is updated in place like the reference's global `state`; state_propagat = state on entry
    void update(StatesGroup &state, const float *feats_down_body_xyz, int feats_down_size)
    {
        fl_state18 st;
        to_abi(state, st);
        fl_iter_info info;
        last_status = fl_lio_frame18(handle, &st, feats_down_body_xyz, feats_down_size, knn, knn_ctx, &info);
        if (last_status < 0) return;                      // HIP/usage error: leave the state untouched
        from_abi(st, state);
        effct_feat_num = info.effct_feat_num;
        total_residual = info.total_residual;
        iterCount = info.iterations - 1;
        last_status = info.status;
    }
};

// ------------------------------------------------------------------------------------------------
// The same block with the map search on the device: ikdtree.Nearest_Search leaves the loop. `set_map` is
// called where the reference's tree content changes (ikdtree.Build :1416, map_incremental :692-706 / :1758).
// update(state, nullptr, 0) consumes a scan already staged on the device (VoxelGridDev::filter_to_scan below).
// ------------------------------------------------------------------------------------------------
struct LioMode18Dev {
    fl_handle handle = nullptr;
    int32_t last_status = 0;
    int effct_feat_num = 0;
    double total_residual = 0.0;
    int iterCount = 0;

    int32_t set_map(const float *map_xyz, int k, float cell_size = 0.5f) { return last_status = fl_map_set_points(handle, map_xyz, k, cell_size); }

    void update(StatesGroup &state, const float *feats_down_body_xyz, int feats_down_size)
    {
        fl_state18 st;
        to_abi(state, st);
        fl_iter_info info;
        last_status = fl_lio_frame18_dev(handle, &st, feats_down_body_xyz, feats_down_size, &info);
        if (last_status < 0) return;
        from_abi(st, state);
        effct_feat_num = info.effct_feat_num;
        total_residual = info.total_residual;
        iterCount = info.iterations - 1;
        last_status = info.status;
    }
};

// ------------------------------------------------------------------------------------------------
// pcl::VoxelGrid<PointType> as the reference drives it (setLeafSize / setInputCloud / filter), laserMapping.cpp:1186,1398-1399
// ------------------------------------------------------------------------------------------------
struct VoxelGridDev {
    fl_handle handle = nullptr;
    float leaf[3] = {0.5f, 0.5f, 0.5f};
    const float *input = nullptr;     // x, y, z, intensity per point; nullptr: the cloud fl_imu_undistort left on the device
    int input_n = 0;
    int32_t last_status = 0;
    bool leaf_too_small = false;

    void setLeafSize(float lx, float ly, float lz) { leaf[0] = lx; leaf[1] = ly; leaf[2] = lz; }
    void setInputCloud(const float *xyzi, int n) { input = xyzi; input_n = n; }
    void setInputCloudOnDevice(int n) { input = nullptr; input_n = n; }
    // filter(*out): centroids to the host
    int filter(std::vector<float> &out_xyzi)
    {
        out_xyzi.resize((size_t)input_n * 4);
        int32_t m = 0, small = 0;
        last_status = fl_scan_voxel_filter(handle, input, input_n, leaf[0], leaf[1], leaf[2], 0, out_xyzi.data(), &m, &small);
        leaf_too_small = small != 0;
        out_xyzi.resize(last_status < 0 ? 0 : (size_t)m * 4);
        return m;
    }
    // filter(*feats_down_body) without the round trip: the centroids become the staged scan of the LIO block
    int filter_to_scan()
    {
        int32_t m = 0, small = 0;
        last_status = fl_scan_voxel_filter(handle, input, input_n, leaf[0], leaf[1], leaf[2], 1, nullptr, &m, &small);
        leaf_too_small = small != 0;
        return last_status < 0 ? 0 : m;
    }
};

// ------------------------------------------------------------------------------------------------
// The LiDAR map kept on the device: lasermap_fov_segment (laserMapping.cpp:363-417) + map_incremental (:692-706) +
// the first frame's ikdtree.Build (:1411-1419). The window arithmetic is the reference's (floats of BoxPointType, doubles of
// pos_LiD / cube_len); the point work is fl_map_delete_boxes / fl_map_add_points.
// ------------------------------------------------------------------------------------------------
struct LocalMapDev {
    fl_handle handle = nullptr;
    double cube_len = 200.0;                  // cube_side_length (:1118)
    float DET_RANGE = 300.0f;                 // :83
    float MOV_THRESHOLD = 1.5f;               // :90
    float downsample_size = 0.5f;             // filter_size_map_min -> ikdtree.set_downsample_param (:1410)
    float vertex_min[3] = {0, 0, 0}, vertex_max[3] = {0, 0, 0};      // LocalMap_Points
    bool Localmap_Initialized = false;
    int kdtree_delete_counter = 0;
    fl_map_info last{};
    int32_t last_status = 0;

    // returns the number of boxes that were cut off (cub_needrm.size())
    int lasermap_fov_segment(const V3D &pos_LiD)
    {
        kdtree_delete_counter = 0;
        if (!Localmap_Initialized) {
            for (int i = 0; i < 3; i++) {
                vertex_min[i] = (float)(pos_LiD.v[i] - cube_len / 2.0);
                vertex_max[i] = (float)(pos_LiD.v[i] + cube_len / 2.0);
            }
            Localmap_Initialized = true;
            return 0;
        }
        float edge[3][2];
        const float lim = MOV_THRESHOLD * DET_RANGE;
        bool need_move = false;
        for (int i = 0; i < 3; i++) {
            edge[i][0] = (float)std::fabs(pos_LiD.v[i] - (double)vertex_min[i]);
            edge[i][1] = (float)std::fabs(pos_LiD.v[i] - (double)vertex_max[i]);
            if (edge[i][0] <= lim || edge[i][1] <= lim) need_move = true;
        }
        if (!need_move) return 0;
        const double a = (cube_len - 2.0 * (double)MOV_THRESHOLD * (double)DET_RANGE) * 0.5 * 0.9;
        const double b = (double)(DET_RANGE * (MOV_THRESHOLD - 1));
        const float mov_dist = (float)(a > b ? a : b);
        float nmin[3], nmax[3], boxes[18];
        for (int i = 0; i < 3; i++) { nmin[i] = vertex_min[i]; nmax[i] = vertex_max[i]; }
        int nb = 0;
        for (int i = 0; i < 3; i++) {
            float *bmin = boxes + nb * 6, *bmax = bmin + 3;
            for (int k = 0; k < 3; k++) { bmin[k] = vertex_min[k]; bmax[k] = vertex_max[k]; }
            if (edge[i][0] <= lim) {
                nmax[i] -= mov_dist; nmin[i] -= mov_dist;
                bmin[i] = vertex_max[i] - mov_dist;
                nb++;
            } else if (edge[i][1] <= lim) {
                nmax[i] += mov_dist; nmin[i] += mov_dist;
                bmax[i] = vertex_min[i] + mov_dist;
                nb++;
            }
        }
        for (int i = 0; i < 3; i++) { vertex_min[i] = nmin[i]; vertex_max[i] = nmax[i]; }
        if (nb > 0) {
            last_status = fl_map_delete_boxes(handle, boxes, nb, &last);
            if (last_status >= 0) kdtree_delete_counter = last.n_removed;
        }
        return nb;
    }
    // feats_down_world of the scan staged on the device, under the state the LIO block left there
    void map_incremental(bool first_frame_build = false)
    {
        last_status = fl_map_add_points(handle, nullptr, 0, first_frame_build ? 0.0f : downsample_size, &last);
    }
    // the in-place map's O(map) compaction + re-index, where the frame has slack (it happens by itself when the arrays fill up otherwise)
    void compact() { last_status = fl_map_compact(handle); }
};

// ------------------------------------------------------------------------------------------------
// ImuProcess: the members UndistortPcl uses + the call, IMU_Processing.cpp:611-809 (Process2 :875)
// ------------------------------------------------------------------------------------------------
struct ImuProcessDev {
    fl_handle handle = nullptr;
    fl_imu_proc proc{};               // cov_gyr, cov_acc, cov_bias_*, mean_acc, Lid_*_to_IMU, acc_s_last, angvel_last, last_imu_, last_lidar_end_time_
    int32_t last_status = 0;

    void set_extrinsic(const double *transl, const double *rot)   // IMU_Processing.cpp:59-63
    {
        for (int i = 0; i < 3; i++) proc.Lid_offset_to_IMU[i] = transl[i];
        for (int i = 0; i < 9; i++) proc.Lid_rot_to_IMU[i] = rot[i];
    }
    // pcl_out in/out: x, y, z, curvature(ms). keep_on_device: skip the read-back (the voxel filter continues on the device)
    void UndistortPcl(const std::vector<fl_imu_sample> &imu, double pcl_beg_time, double pcl_end_time, StatesGroup &state_inout,
                      std::vector<float> &pcl_out_xyzt, bool keep_on_device)
    {
        fl_state18 st;
        to_abi(state_inout, st);
        const int n = (int)(pcl_out_xyzt.size() / 4);
        last_status = fl_imu_undistort(handle, &proc, &st, imu.data(), (int)imu.size(), pcl_beg_time, pcl_end_time, pcl_out_xyzt.data(), n,
                                       keep_on_device ? nullptr : pcl_out_xyzt.data(), nullptr, nullptr);
        if (last_status < 0) return;
        from_abi(st, state_inout);
    }
};

// ------------------------------------------------------------------------------------------------
// The LiDAR half of the frame loop in one call (round 5): `p_imu->Process2(LidarMeasures, state, feats_undistort)` (laserMapping.cpp:1359) ->
// `downSizeFilterSurf.setInputCloud(feats_undistort); downSizeFilterSurf.filter(*feats_down_body)` (:1398-1399) -> the Mode-18 block
// `if(lidar_en){ for(iterCount = -1; ...) }` (:1504-1733) with the map search on the device. One enqueue, one wait (fl_lidar_front);
// `imu` holds the ImuProcess members exactly as ImuProcessDev does, feats_down_size comes back with the state.
// ------------------------------------------------------------------------------------------------
struct LidarFrontDev {
    fl_handle handle = nullptr;
    ImuProcessDev *imu = nullptr;
    float filter_size_surf = 0.5f;
    int32_t last_status = 0;
    int feats_down_size = 0, effct_feat_num = 0, iterCount = 0;
    double total_residual = 0.0;

    void update(const std::vector<fl_imu_sample> &samples, double pcl_beg_time, double pcl_end_time, StatesGroup &state,
                const std::vector<float> &pcl_xyzt, bool staged = false)
    {
        fl_state18 st;
        to_abi(state, st);
        fl_iter_info info;
        int32_t m = 0;
        last_status = fl_lidar_front(handle, &imu->proc, &st, samples.data(), (int)samples.size(), pcl_beg_time, pcl_end_time, pcl_xyzt.data(),
                                     (int)(pcl_xyzt.size() / 4), filter_size_surf, staged ? FL_FRONT_STAGED : 0, &info, &m);
        if (last_status < 0) return;
        from_abi(st, state);
        feats_down_size = m;
        effct_feat_num = info.effct_feat_num;
        total_residual = info.total_residual;
        iterCount = info.iterations - 1;
        last_status = info.status;
    }
};

// ------------------------------------------------------------------------------------------------
// LidarSelector::ComputeJ(cv::Mat img) -> UpdateState(img, err, level) x 3, lidar_selection.cpp:967-983
// ------------------------------------------------------------------------------------------------
struct VioUpdater {
    fl_handle handle = nullptr;
    int32_t last_status = 0;

    void ComputeJ(const uint8_t *img, int width, int height, SubSparseMapView &sub_sparse_map, StatesGroup &state,
                  const StatesGroup &state_propagat)
    {
        const int total_points = (int)sub_sparse_map.search_levels.size();
        if (total_points == 0) return;                    // lidar_selection.cpp:969-970
        last_status = fl_vio_set_frame(handle, img, width, height, width);
        last_status |= fl_vio_set_patches(handle, sub_sparse_map.patch.data(), sub_sparse_map.pos.data(),
                                          sub_sparse_map.search_levels.data(), total_points);
        if (last_status < 0) return;
        fl_state18 st, sp;
        to_abi(state, st);
        to_abi(state_propagat, sp);
        last_status = fl_vio_compute_j(handle, &st, &sp, nullptr);
        if (last_status < 0) return;
        from_abi(st, state);
        sub_sparse_map.errors.resize((size_t)total_points);
        last_status |= fl_vio_get_errors(handle, sub_sparse_map.errors.data());
    }
};

// ------------------------------------------------------------------------------------------------
// LidarSelector::detect(cv::Mat img, PointCloudXYZI::Ptr pg), lidar_selection.cpp:1027-1075, with the visual map on the device:
// addFromSparseMap -> addSparseMap -> ComputeJ -> addObservation become one call each
// ------------------------------------------------------------------------------------------------
struct LidarSelectorDev {
    fl_handle handle = nullptr;
    int grid_size = 40;
    bool ncc_en = false;
    double ncc_thre = 0.0, outlier_threshold = 300.0;
    int32_t frame_id = 0;
    int32_t last_status = 0;
    int n_selected = 0, n_founded = 0, n_observed = 0;

    int32_t init() { return last_status = fl_vmap_clear(handle, grid_size); }            // LidarSelector::init (:61-79)

    // T_f_w of the frame from the state, as updateFrameState does (:904-911): Rcw = Rci * R^T, Pcw = -Rci * R^T * p + Pci
    static void frame_pose(const double *Rci, const double *Pci, const StatesGroup &s, double *Rcw, double *Pcw)
    {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double a = 0.0;
                for (int k = 0; k < 3; k++) a += Rci[i * 3 + k] * s.rot_end.m[j * 3 + k];
                Rcw[i * 3 + j] = a;
            }
        for (int i = 0; i < 3; i++) {
            double a = 0.0;
            for (int k = 0; k < 3; k++) a += Rcw[i * 3 + k] * s.pos_end.v[k];
            Pcw[i] = -a + Pci[i];
        }
    }

    // img: grey image of the frame; pg / pg_down: the registered scan and its 0.2 m down-sampled form (world frame, xyz floats)
    void detect(const uint8_t *img, int width, int height, int stride, const float *pg, int n_pg, const float *pg_down, int n_down,
                const double *Rci, const double *Pci, StatesGroup &state)
    {
        fl_state18 st;
        to_abi(state, st);
        int32_t ns = 0, na = 0, no = 0;
        // the whole body of LidarSelector::detect (:1027-1076) in one ABI call
        last_status = fl_vio_detect(handle, img, width, height, stride, pg, n_pg, pg_down, n_down, Rci, Pci, &st, frame_id, ncc_en ? 1 : 0, ncc_thre,
                                    outlier_threshold, &ns, &na, &no);
        if (last_status < 0) return;
        from_abi(st, state);
        n_selected = ns; n_founded = na; n_observed = no;
        frame_id++;
    }
    // The same with pg left where the LiDAR frame put it (round 6): the scan this handle holds (fl_lidar_front / fl_lio_frame18_dev:
    // feats_down_body) is registered under `state` on the device (laserMapping.cpp:695-698) and down-sampled there (lidar_selection.cpp:352-353);
    // nothing but the image goes up -- and an image in fl_host_alloc memory is fetched by the frame's first kernel.
    void detect(const uint8_t *img, int width, int height, int stride, const double *Rci, const double *Pci, StatesGroup &state)
    {
        detect(img, width, height, stride, nullptr, FL_DETECT_SCAN_ON_DEVICE, nullptr, 0, Rci, Pci, state);
    }
};

}  // namespace fastlivo_host
