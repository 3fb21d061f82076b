// imu_kernels.h -- SURVEY.md section 8(f) row N4: ImuProcess::UndistortPcl (src/IMU_Processing.cpp:611-809) on the device.
//
//  imu_forward_kernel      one workgroup: forward propagation of state and 18x18 covariance over the IMU samples
//                          (:656-741: F_x, cov_w, cov = F cov F^T + cov_w -- dense 18-term dot products in the order of
//                          the oracle, one covariance element per lane), the IMUpose list, the frame-end prediction
//                          (:743-759) and the extrinsic products (:764-765).
//  undistort_heads_kernel  per point: the latest IMU interval (head >= 1) that starts before the point's time
//  undistort_scan_kernel   suffix minimum of those heads over the cloud = the interval the reference's backward
//                          double loop (:778-808) is in when it reaches the point. The loop walks the cloud from the
//                          last point and only ever moves to EARLIER intervals, so for an unsorted cloud a point can be
//                          handled by an earlier interval than its own time asks for; the suffix minimum reproduces
//                          that exactly. It also ends for good at the first point (from the back) that is not later
//                          than IMUpose[0] (offset 0): every point before it stays uncompensated (`term`).
//  undistort_apply_kernel  per point compensation (:792-800) in double, stored as float. Point 0 is special: after
//                          compensating it the reference breaks the inner loop WITHOUT stepping back (:803), so each
//                          earlier interval compensates it again from its already-rewritten coordinates; replicated.
#pragma once

#include "fl_device.h"

struct FlImuSample { double t, gyr[3], acc[3]; };
struct FlPose6 { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };

struct FlImuDev {
    // in/out state (StatesGroup)
    double rot[9], pos[3], vel[3], bg[3], ba[3], grav[3];
    double P[324];
    // ImuProcess members
    double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3];
    double mean_acc_norm;
    double R_LI[9], t_LI[3];           // Lid_rot_to_IMU, Lid_offset_to_IMU
    double acc_s_last[3], angvel_last[3];
    double last_lidar_end_time, pcl_beg_time, pcl_end_time;
    // products for the backward pass
    double extR_Ri[9], exrR_extT[3];
    int32_t n_poses;
    int32_t term;                      // highest point index the backward loop does not reach (-1: none)
};

__device__ __forceinline__ void fl_m3_mul(const double *A, const double *B, double *C)
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = T[i];
}
__device__ __forceinline__ void fl_m3_mv(const double *A, const double *x, double *o)
{
    double t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
// include/so3_math.h:31-52, Exp(ang_vel, dt)
__device__ __forceinline__ void fl_so3_exp_dt(const double *w, double dt, double *R)
{
    const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n > 0.0000001) {
        const double a[3] = {w[0] / n, w[1] / n, w[2] / n};
        const double K[9] = {0.0, -a[2], a[1], a[2], 0.0, -a[0], -a[1], a[0], 0.0};
        const double ang = n * dt;
        double s, c1;
        fl_sin_omc(ang, &s, &c1);          // Taylor form for |ang| <= 0.5 (always, for IMU intervals), library beyond
        double cK[9], cKK[9];
#pragma unroll
        for (int i = 0; i < 9; i++) cK[i] = c1 * K[i];
        fl_m3_mul(cK, K, cKK);
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + cKK[i];
    }
}

#define FL_IMU_NT 384
// the 3x3 pieces of F_x (:700-709) and cov_w (:701, :711-714) of one interval
struct FlImuStep {
    double Exp_m[9], RAdt[9], Rdt[9], Cacc[9];   // Exp(w,-dt), R*[a]x*dt, R*dt, R diag(cov_acc) R^T dt^2
    double dt, dt2;
};
// sum over k of F_x(a, k) * v[k * stride], the terms of row a that are not structurally zero, in ascending k, starting from 0.0: the
// value (and the bits) of the dense 18-term loop, whose other terms are exact zeros
__device__ __forceinline__ double fl_imu_row_dot(const FlImuStep &S, int a, const double *v, int stride)
{
    const int ba = a / 3, r = a % 3;
    double q = 0.0;
    if (ba == 0) {                      // rot:  Exp(w, -dt) | -dt I at bias_g
        q = q + S.Exp_m[r * 3 + 0] * v[0 * stride];
        q = q + S.Exp_m[r * 3 + 1] * v[1 * stride];
        q = q + S.Exp_m[r * 3 + 2] * v[2 * stride];
        q = q + (-S.dt) * v[(9 + r) * stride];
    } else if (ba == 1) {               // pos:  I | dt I at vel
        q = q + 1.0 * v[a * stride];
        q = q + S.dt * v[(6 + r) * stride];
    } else if (ba == 2) {               // vel:  -R [a]x dt at rot | I | -R dt at bias_a | dt I at gravity
        q = q + (-S.RAdt[r * 3 + 0]) * v[0 * stride];
        q = q + (-S.RAdt[r * 3 + 1]) * v[1 * stride];
        q = q + (-S.RAdt[r * 3 + 2]) * v[2 * stride];
        q = q + 1.0 * v[a * stride];
        q = q + (-S.Rdt[r * 3 + 0]) * v[12 * stride];
        q = q + (-S.Rdt[r * 3 + 1]) * v[13 * stride];
        q = q + (-S.Rdt[r * 3 + 2]) * v[14 * stride];
        q = q + S.dt * v[(15 + r) * stride];
    } else {                            // bias_g, bias_a, gravity: I
        q = q + 1.0 * v[a * stride];
    }
    return q;
}
// cov_w(i, j) (:701, :711-714)
__device__ __forceinline__ double fl_imu_cov_w(const FlImuStep &S, const double *cg, const double *cbg, const double *cba, int i, int j)
{
    const int bi = i / 3, bj = j / 3, r = i % 3, c = j % 3;
    if (bi != bj) return 0.0;
    if (bi == 2) return S.Cacc[r * 3 + c];
    if (r != c) return 0.0;
    if (bi == 0) return cg[r] * S.dt * S.dt;
    if (bi == 3) return cbg[r] * S.dt * S.dt;
    if (bi == 4) return cba[r] * S.dt * S.dt;
    return 0.0;
}

// The loop over the IMU intervals (:658-741) carries three chains of different weight: the per-interval quantities that depend
// on the samples only (mean rates, dt, Exp(w, +-dt): two square roots, six divisions, two sine/cosine pairs), the state chain
// (R <- R Exp, acc, pos, vel: ~100 flops per interval) and the covariance chain (two 18x18x18 products per interval). Run one after
// the other by one lane, as a first version did, an interval cost 3.6 us (72 us per 20-sample frame, the longest kernel of the
// LiDAR front). Here, per chunk of FL_IMU_CH intervals: (A) one lane per interval does the sample-only part, (B) one lane walks the
// state chain, (C) one lane per interval forms the pieces of F_x / cov_w that need that interval's R, (D) the workgroup walks the
// covariance chain, one element per lane. Every quantity is computed by the same expression as before, so the results are
// bit-identical; only the schedule changed.
// (Round 6, measured and rejected: (B) and (C) on a seventh wavefront one interval AHEAD of (D), joined at (D)'s two barriers per interval. The
// chain's step on one lane -- ~1 us with its 22 pose stores -- is as long as both covariance phases together, so an interval became
// max(B, phase 1) + phase 2 instead of B + phases: 100 samples 149 -> 183 us, 20 samples 59-63 us either way, and fl_lidar_front did not move
// at all: there the launch is bound by the fetch of the raw scan over the host link (~33 us for 100 k points) that runs beside the chain.)
#define FL_IMU_CH 64
// out18 / tail (fl_lidar_front, nullable): the propagated state goes straight into the LIO state block on the device -- state,
// state_propagat (:1292 of laserMapping.cpp: state_propagat = state) and cov -- and the members the next frame starts from into the
// tail behind it; nothing of it visits the host between the propagation and the update.
// pull (fl_lidar_front with everything in page-locked host memory, nullable): the launch then needs no copy command in front of it --
//   in_host     the FlImuDev block as the host filled it (its device address): workgroup 0 copies it into D first
//   x18_host    the update's state block in the handle's mirror: workgroup 0 copies it into out18 (then overwrites state and cov)
//   v           may itself point into page-locked host memory (each sample is read once)
//   scan_host   the raw scan: workgroups 1 .. gridDim.x - 1 copy it into scan_dev BESIDE the propagation (the upload used to go through a
//               second stream and two events; ~33 us for 100 k points either way, now inside this launch)
struct FlImuPull {
    const FlImuDev *in_host;
    const FlDev18 *x18_host;
    const float4 *scan_host;
    float4 *scan_dev;
    int n_scan;
};
__global__ __launch_bounds__(FL_IMU_NT) void imu_forward_kernel(FlImuDev *__restrict__ D, const FlImuSample *__restrict__ v, int nv,
                                                               FlPose6 *__restrict__ poses, FlDev18 *__restrict__ out18 = nullptr,
                                                               FlFrontTail *__restrict__ tail = nullptr, FlImuPull pull = FlImuPull{})
{
    if (blockIdx.x > 0) {                       // scan fetch: 16 bytes per lane and load, every byte once
        const int stride = ((int)gridDim.x - 1) * FL_IMU_NT;
        const fl_u4 *src = reinterpret_cast<const fl_u4 *>(pull.scan_host);
        fl_u4 *dst = reinterpret_cast<fl_u4 *>(pull.scan_dev);
        for (int i = ((int)blockIdx.x - 1) * FL_IMU_NT + (int)threadIdx.x; i < pull.n_scan; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
        return;
    }
    // (every load of both blocks in flight before the first store: fl_device.h FlPull)
    FlPull<FL_IMU_NT, (int)sizeof(FlImuDev)> blk_in;
    FlPull<FL_IMU_NT, (int)sizeof(FlDev18)> blk_x;
    const bool pull_in = pull.in_host != nullptr, pull_x = pull.x18_host != nullptr && out18 != nullptr;      // (uniform)
    if (pull_in) blk_in.load(pull.in_host);
    if (pull_x) blk_x.load(pull.x18_host);
    // the first chunk's samples travel with the blocks above (one trip over the host link for all of them when v is host memory too)
    FlImuSample pre_head = {}, pre_tail = {};
    if ((int)threadIdx.x < min(FL_IMU_CH, nv - 1)) { pre_head = v[threadIdx.x]; pre_tail = v[threadIdx.x + 1]; }
    if (pull_in) blk_in.store(D);
    if (pull_x) blk_x.store(out18);
    if (pull_in || pull_x) { __threadfence_block(); __syncthreads(); }
    __shared__ double sP[324], sT[324];        // the covariance; (F cov)^T of the interval in flight
    __shared__ FlImuStep sS[FL_IMU_CH];
    __shared__ double sExpF[FL_IMU_CH][9], sW[FL_IMU_CH][3], sA[FL_IMU_CH][3], sRb[FL_IMU_CH][9];
    __shared__ int sGo[FL_IMU_CH];
    __shared__ double sTail[FL_IMU_CH];        // tail.t of the interval (a global load inside the serial chain costs ~1 us)
    __shared__ double s_R[9], s_vel[3], s_pos[3], s_acc[3], s_w[3];
    __shared__ int s_K;
    const int t = (int)threadIdx.x;
    const int ti = t / 18, tj = t % 18;
    if (t < 324) sP[t] = D->P[t];
    // per-frame constants in registers (every lane: the F/C construction below reads them)
    double bg[3], ba[3], grav[3], cg[3], ca[3], cbg[3], cba[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        bg[k] = D->bg[k]; ba[k] = D->ba[k]; grav[k] = D->grav[k];
        cg[k] = D->cov_gyr[k]; ca[k] = D->cov_acc[k]; cbg[k] = D->cov_bias_gyr[k]; cba[k] = D->cov_bias_acc[k];
    }
    const double mean_acc_norm = D->mean_acc_norm, last_end = D->last_lidar_end_time, beg = D->pcl_beg_time;
    if (t == 0) {
        for (int k = 0; k < 9; k++) s_R[k] = D->rot[k];
        for (int k = 0; k < 3; k++) { s_vel[k] = D->vel[k]; s_pos[k] = D->pos[k]; s_acc[k] = D->acc_s_last[k]; s_w[k] = D->angvel_last[k]; }
        FlPose6 p;
        p.offset_time = 0.0;
        for (int k = 0; k < 3; k++) { p.acc[k] = s_acc[k]; p.gyr[k] = s_w[k]; p.vel[k] = s_vel[k]; p.pos[k] = s_pos[k]; }
        for (int k = 0; k < 9; k++) p.rot[k] = s_R[k];
        poses[0] = p;                                                              // :656
        s_K = 1;
    }
    __syncthreads();
    for (int c0 = 0; c0 + 1 < nv; c0 += FL_IMU_CH) {
        const int ns = min(FL_IMU_CH, nv - 1 - c0);
        // (A) sample-only part of interval c0 + t
        if (t < ns) {
            const FlImuSample head = c0 == 0 ? pre_head : v[c0 + t], tail = c0 == 0 ? pre_tail : v[c0 + t + 1];
            const int go = !(tail.t < last_end);                                   // :666
            sGo[t] = go;
            sTail[t] = tail.t;
            if (go) {
                double w[3], a[3];
                for (int k = 0; k < 3; k++) {
                    w[k] = 0.5 * (head.gyr[k] + tail.gyr[k]);
                    a[k] = 0.5 * (head.acc[k] + tail.acc[k]);
                }
                for (int k = 0; k < 3; k++) {
                    w[k] -= bg[k];                                                 // :684
                    a[k] = a[k] * 9.81 / mean_acc_norm - ba[k];                    // :685
                }
                const double dt = (head.t < last_end) ? (tail.t - last_end) : (tail.t - head.t);   // :687-694
                double Exp_f[9], Exp_m[9];
                fl_so3_exp_dt(w, dt, Exp_f);
                fl_so3_exp_dt(w, -dt, Exp_m);                                      // :703
                for (int k = 0; k < 9; k++) { sExpF[t][k] = Exp_f[k]; sS[t].Exp_m[k] = Exp_m[k]; }
                for (int k = 0; k < 3; k++) { sW[t][k] = w[k]; sA[t][k] = a[k]; }
                sS[t].dt = dt;
            }
        }
        __syncthreads();
        // (B) the state chain
        if (t == 0) {
            double R[9], vel[3], pos[3], acc[3] = {s_acc[0], s_acc[1], s_acc[2]}, wl[3] = {s_w[0], s_w[1], s_w[2]};
            for (int k = 0; k < 9; k++) R[k] = s_R[k];
            for (int k = 0; k < 3; k++) { vel[k] = s_vel[k]; pos[k] = s_pos[k]; }
            int K = s_K;
            for (int st = 0; st < ns; st++) {
                if (!sGo[st]) continue;
                const double dt = sS[st].dt;
                double a[3], Exp_f[9];
                for (int k = 0; k < 3; k++) { a[k] = sA[st][k]; wl[k] = sW[st][k]; }
                for (int k = 0; k < 9; k++) { Exp_f[k] = sExpF[st][k]; sRb[st][k] = R[k]; }     // R before the step: F_x / cov_w use it (:707-712)
                fl_m3_mul(R, Exp_f, R);                                            // :719
                double Ra[3];
                fl_m3_mv(R, a, Ra);
                for (int k = 0; k < 3; k++) acc[k] = Ra[k] + grav[k];              // :722
                for (int k = 0; k < 3; k++) pos[k] = pos[k] + vel[k] * dt + 0.5 * acc[k] * dt * dt;   // :725
                for (int k = 0; k < 3; k++) vel[k] = vel[k] + acc[k] * dt;         // :728
                FlPose6 p;
                p.offset_time = sTail[st] - beg;                                   // :733
                for (int k = 0; k < 3; k++) { p.acc[k] = acc[k]; p.gyr[k] = wl[k]; p.vel[k] = vel[k]; p.pos[k] = pos[k]; }
                for (int k = 0; k < 9; k++) p.rot[k] = R[k];
                poses[K] = p;
                K++;
            }
            for (int k = 0; k < 9; k++) s_R[k] = R[k];
            for (int k = 0; k < 3; k++) { s_vel[k] = vel[k]; s_pos[k] = pos[k]; s_acc[k] = acc[k]; s_w[k] = wl[k]; }   // :731-732
            s_K = K;
        }
        __syncthreads();
        // (C) the pieces of F_x and cov_w that need the interval's R
        if (t < ns && sGo[t]) {
            double R[9], RA[9];
            for (int k = 0; k < 9; k++) R[k] = sRb[t][k];
            const double a[3] = {sA[t][0], sA[t][1], sA[t][2]};
            const double dt = sS[t].dt;
            const double askew[9] = {0.0, -a[2], a[1], a[2], 0.0, -a[0], -a[1], a[0], 0.0};
            fl_m3_mul(R, askew, RA);
            for (int k = 0; k < 9; k++) { sS[t].RAdt[k] = RA[k] * dt; sS[t].Rdt[k] = R[k] * dt; }   // :707-708
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {                                      // :712
                    double q = 0.0;
                    for (int k = 0; k < 3; k++) q += R[i * 3 + k] * ca[k] * R[j * 3 + k];
                    sS[t].Cacc[i * 3 + j] = q * dt * dt;
                }
        }
        __syncthreads();
        // (D) the covariance chain: cov = F cov F^T + cov_w (:716). F_x is the identity plus five 3x3 blocks (:700-709): every row has
        // 1, 2, 4 or 8 entries that are not structurally zero, and the dense 18-term dot products of the reference add an exact 0 * x for
        // all the others -- skipping those terms (same order for the rest, same start from 0.0) gives the same bits with less than half
        // the dependent additions, no F / cov_w matrices in LDS and two barriers per interval instead of three (round 5; the dense form
        // was 1.6 us per interval, 32 of the kernel's 43 us). Phase 1: T = F P, thread (i, j) = row i of F against column j of P, T kept
        // TRANSPOSED; phase 2: thread (j, i) -- consecutive lanes share row j of F, i.e. the branch -- T^T's column against that row, + cov_w.
        for (int st = 0; st < ns; st++) {
            if (!sGo[st]) continue;                                                // uniform over the workgroup
            if (t < 324) sT[tj * 18 + ti] = fl_imu_row_dot(sS[st], ti, sP + tj, 18);      // T(i, j) = sum_k F(i, k) P(k, j)
            __syncthreads();
            if (t < 324) {
                const int i2 = tj, j2 = ti;                                        // this thread's element of the new covariance: (i2, j2)
                const double q = fl_imu_row_dot(sS[st], j2, sT + i2, 18);          // sum_k T(i2, k) F(j2, k), T(i2, k) at sT[k * 18 + i2]
                sP[i2 * 18 + j2] = q + fl_imu_cov_w(sS[st], cg, cbg, cba, i2, j2);
            }
            __syncthreads();
        }
    }
    if (t < 324) { D->P[t] = sP[t]; if (out18) out18->P[t] = sP[t]; }
    if (t == 0) {
        const double imu_end_time = v[nv - 1].t, end = D->pcl_end_time;
        double note, dt;                                                           // :743-759
        if (imu_end_time > beg) { note = end > imu_end_time ? 1.0 : -1.0; dt = note * (end - imu_end_time); }
        else { note = end > beg ? 1.0 : -1.0; dt = note * (end - beg); }
        double w[3] = {note * s_w[0], note * s_w[1], note * s_w[2]}, E[9], R[9], Rend[9];
        for (int k = 0; k < 9; k++) R[k] = s_R[k];
        fl_so3_exp_dt(w, dt, E);
        fl_m3_mul(R, E, Rend);
        for (int k = 0; k < 3; k++) D->vel[k] = s_vel[k] + note * s_acc[k] * dt;
        for (int k = 0; k < 9; k++) D->rot[k] = Rend[k];
        for (int k = 0; k < 3; k++) D->pos[k] = s_pos[k] + note * s_vel[k] * dt + note * 0.5 * s_acc[k] * dt * dt;
        for (int k = 0; k < 3; k++) { D->acc_s_last[k] = s_acc[k]; D->angvel_last[k] = s_w[k]; }
        double LT[9], RT[9], X[9], o[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { LT[i * 3 + j] = D->R_LI[j * 3 + i]; RT[i * 3 + j] = Rend[j * 3 + i]; }
        fl_m3_mul(LT, RT, X);                                                      // :764
        fl_m3_mv(LT, D->t_LI, o);                                                  // :765
        for (int k = 0; k < 9; k++) D->extR_Ri[k] = X[k];
        for (int k = 0; k < 3; k++) D->exrR_extT[k] = o[k];
        D->n_poses = s_K;
        D->term = -1;
        if (out18) {
            double x[24];
            for (int k = 0; k < 9; k++) x[k] = Rend[k];
            for (int k = 0; k < 3; k++) { x[9 + k] = D->pos[k]; x[12 + k] = D->vel[k]; x[15 + k] = bg[k]; x[18 + k] = ba[k]; x[21 + k] = grav[k]; }
            for (int k = 0; k < 24; k++) { out18->x[k] = x[k]; out18->xprop[k] = x[k]; out18->xold[k] = x[k]; }
        }
        if (tail) {
            for (int k = 0; k < 3; k++) { tail->acc_s_last[k] = s_acc[k]; tail->angvel_last[k] = s_w[k]; }
            tail->n_poses = s_K;
            tail->unsorted = 0;
        }
    }
}

// latest head k in [1, K-2] with offset_time[k] < t (those offsets increase with k), 0 if none
__device__ __forceinline__ int fl_imu_head(const FlPose6 *poses, int K, double t)
{
    int lo = 1, hi = K - 1;            // count of k in [1, K-2] with offset < t, found by bisection
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (poses[mid].offset_time < t) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

__global__ __launch_bounds__(FL_BLOCK) void undistort_heads_kernel(const float4 *__restrict__ pts, int n, const FlImuDev *__restrict__ D,
                                                                  const FlPose6 *__restrict__ poses, int *__restrict__ head,
                                                                  int *__restrict__ blockmin)
{
    __shared__ int s_min[FL_BLOCK / 64];
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    const int K = D->n_poses;
    int v = 0x7fffffff;
    if (i < n) {
        v = fl_imu_head(poses, K, (double)pts[i].w / 1000.0);
        head[i] = v;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s));
    if ((threadIdx.x & 63u) == 0) s_min[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < FL_BLOCK / 64; w++) v = min(v, s_min[w]);
        blockmin[blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(FL_BLOCK) void undistort_scan_kernel(const float4 *__restrict__ pts, int n, FlImuDev *__restrict__ D,
                                                                 const FlPose6 *__restrict__ poses, int *__restrict__ head,
                                                                 const int *__restrict__ blockmin, int nblocks)
{
    __shared__ int s_v[FL_BLOCK];
    __shared__ int s_later[FL_BLOCK / 64];
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    // minimum over all later workgroups
    int later = 0x7fffffff;
    for (int b = blockIdx.x + 1 + (int)threadIdx.x; b < nblocks; b += FL_BLOCK) later = min(later, blockmin[b]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) later = min(later, __shfl_xor(later, s));
    if ((threadIdx.x & 63u) == 0) s_later[threadIdx.x >> 6] = later;
    s_v[threadIdx.x] = (i < n) ? head[i] : 0x7fffffff;
    __syncthreads();
    later = s_later[0];
    for (int w = 1; w < FL_BLOCK / 64; w++) later = min(later, s_later[w]);
    // suffix minimum inside the workgroup (Hillis-Steele from the right)
    for (int d = 1; d < FL_BLOCK; d <<= 1) {
        const int mine = s_v[threadIdx.x];
        const int other = (threadIdx.x + d < FL_BLOCK) ? s_v[threadIdx.x + d] : 0x7fffffff;
        __syncthreads();
        s_v[threadIdx.x] = min(mine, other);
        __syncthreads();
    }
    if (i < n) {
        const int m = min(s_v[threadIdx.x], later);
        head[i] = m;
        // at head 0 the point must be later than IMUpose[0] (offset 0.0); otherwise the reference's loops end here
        if (m == 0 && !((double)pts[i].w / 1000.0 > poses[0].offset_time)) atomicMax(&D->term, i);
    }
}

__device__ __forceinline__ void fl_undistort_point(const FlImuDev *D, const FlPose6 &hd, double t, float &x, float &y, float &z)
{
    const double dt = t - hd.offset_time;                                          // :790
    double E[9], R_i[9], T_ei[3], P_i[3] = {(double)x, (double)y, (double)z}, a[3], b[3], c[3];
    fl_so3_exp_dt(hd.gyr, dt, E);
    fl_m3_mul(hd.rot, E, R_i);                                                     // :796
#pragma unroll
    for (int k = 0; k < 3; k++) T_ei[k] = hd.pos[k] + hd.vel[k] * dt + 0.5 * hd.acc[k] * dt * dt - D->pos[k];   // :797
    fl_m3_mv(D->R_LI, P_i, a);
#pragma unroll
    for (int k = 0; k < 3; k++) a[k] += D->t_LI[k];
    fl_m3_mv(R_i, a, b);
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] += T_ei[k];
    fl_m3_mv(D->extR_Ri, b, c);                                                    // :800
    x = (float)(c[0] - D->exrR_extT[0]); y = (float)(c[1] - D->exrR_extT[1]); z = (float)(c[2] - D->exrR_extT[2]);
}

// box (fl_lidar_front, nullable): the workgroup's bounding box of the FINAL points for the voxel filter that follows (voxel_kernels.h
// FlVxPartial; what vx_minmax_kernel would compute in a launch of its own)
__global__ __launch_bounds__(FL_BLOCK) void undistort_apply_kernel(float4 *__restrict__ pts, int n, const FlImuDev *__restrict__ D,
                                                                  const FlPose6 *__restrict__ poses, const int *__restrict__ head,
                                                                  FlVxPartial *__restrict__ box = nullptr)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    const int K = D->n_poses;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const bool moves = !(K < 2 || i <= D->term);
        if (moves || box) p = pts[i];
        if (moves) {
            const double t = (double)p.w / 1000.0;
            const int hd = head[i];
            fl_undistort_point(D, poses[hd], t, p.x, p.y, p.z);
            if (i == 0) {
                for (int k = hd - 1; k >= 0; k--)                                  // :803: no step back after the first point
                    if (t > poses[k].offset_time) fl_undistort_point(D, poses[k], t, p.x, p.y, p.z);
            }
            pts[i] = p;
        }
    }
    if (!box) return;
    static_assert(FL_BLOCK == 256, "fl_vx_block_box");
    __shared__ unsigned s_red[4][7];
    unsigned mn[3] = {0u, 0u, 0u}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
    if (i < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        mn[0] = ~fl_vx_enc(p.x); mn[1] = ~fl_vx_enc(p.y); mn[2] = ~fl_vx_enc(p.z);
        mx[0] = fl_vx_enc(p.x); mx[1] = fl_vx_enc(p.y); mx[2] = fl_vx_enc(p.z);
        cnt = 1;
    }
    const FlVxPartial r = fl_vx_block_box(mn, mx, cnt, s_red);
    if (threadIdx.x == 0) box[blockIdx.x] = r;
}

// The three kernels above in ONE launch for clouds whose times do not decrease (what a LiDAR driver delivers; the reference's own
// UndistortPcl has its sort commented out, :647, and therefore handles any order -- the general kernels replicate that): then the latest
// interval that starts before a point is never later than its successor's, the suffix minimum of undistort_scan_kernel is the point's own
// head, and "the loops ended before they reached this point" (term) is the point's own test (head 0 and not later than IMUpose[0]). Every
// point checks its predecessor's time; a decrease (or a NaN time) anywhere raises tail->unsorted and the host runs the frame again with the
// general kernels (and keeps to them for a while). Same arithmetic per point: same bits. box as in undistort_apply_kernel.
__global__ __launch_bounds__(FL_BLOCK) void undistort_sorted_kernel(float4 *__restrict__ pts, int n, const FlImuDev *__restrict__ D,
                                                                   const FlPose6 *__restrict__ poses, FlVxPartial *__restrict__ box,
                                                                   FlFrontTail *__restrict__ tail)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    const int K = D->n_poses;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    bool disorder = false;
    if (i < n) {
        p = pts[i];
        const float w_prev = (i > 0) ? pts[i - 1].w : p.w;       // (the predecessor's time is never rewritten: only x, y, z are)
        disorder = !(w_prev <= p.w);
        const double t = (double)p.w / 1000.0;
        const int hd = fl_imu_head(poses, K, t);
        const bool stays = (hd == 0) && !(t > poses[0].offset_time);
        if (!(K < 2 || stays)) {
            fl_undistort_point(D, poses[hd], t, p.x, p.y, p.z);
            if (i == 0) {
                for (int k = hd - 1; k >= 0; k--)                                  // :803: no step back after the first point
                    if (t > poses[k].offset_time) fl_undistort_point(D, poses[k], t, p.x, p.y, p.z);
            }
        }
    }
    if (__ballot(disorder) != 0ull && (threadIdx.x & 63u) == 0u) tail->unsorted = 1;      // (same value from everybody)
    // every lane has read its predecessor's time before anybody's store lands on it? Stores touch x, y, z AND w of the own point only (a
    // float4 store): w is rewritten with the value it had, so a neighbour that reads it late sees the same bits.
    if (i < n) pts[i] = p;
    static_assert(FL_BLOCK == 256, "fl_vx_block_box");
    __shared__ unsigned s_red[4][7];
    unsigned mn[3] = {0u, 0u, 0u}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
    if (i < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        mn[0] = ~fl_vx_enc(p.x); mn[1] = ~fl_vx_enc(p.y); mn[2] = ~fl_vx_enc(p.z);
        mx[0] = fl_vx_enc(p.x); mx[1] = fl_vx_enc(p.y); mx[2] = fl_vx_enc(p.z);
        cnt = 1;
    }
    const FlVxPartial r = fl_vx_block_box(mn, mx, cnt, s_red);
    if (threadIdx.x == 0) box[blockIdx.x] = r;
}
