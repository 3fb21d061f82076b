/*
 * oracle/orc_imu.c -- TEST INFRASTRUCTURE ONLY (see fastlivo_oracle.h).
 *
 * CPU restatement of ImuProcess::UndistortPcl(LidarMeasureGroup&, StatesGroup&, PointCloudXYZI&),
 * /root/reference/src/IMU_Processing.cpp:611-809 (the variant Process2 calls, :875), without the ROS/PCL containers:
 *   - the selection of the frame's points and of pcl_beg_time / pcl_end_time (:620-648) is host bookkeeping on
 *     LidarMeasureGroup and stays with the caller; both times are inputs here;
 *   - forward propagation of state and covariance over the IMU samples   :656-741
 *   - extrapolation to the frame end                                      :743-759
 *   - backward propagation (undistortion) of every point                 :778-809, loops restated literally,
 *     including what they do when the first point is reached before the first IMU interval (it is compensated
 *     again by every earlier interval, :803) and when a point is not later than IMUpose[0] (the loop ends).
 * Exp() is include/so3_math.h:31-52; set_pose6d include/common_lib.h:396-412. Held to the text of
 * ImuProcess::UndistortPcl since round 4 (oracle/ref_eigen, tests/test_ref_eigen_cpu.py: bit for bit); no reference tests exist.
 */
#include "fastlivo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static void m3_mul(const double *A, const double *B, double *C)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, T, sizeof T);
}
static void m3_mv(const double *A, const double *x, double *o)
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    memcpy(o, t, sizeof t);
}
static void m3_T(const double *A, double *O)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
    memcpy(O, T, sizeof T);
}

/* so3_math.h:31-52: Exp(ang_vel, dt) */
static void so3_exp_dt(const double *w, double dt, double *R)
{
    const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n > 0.0000001) {
        const double a[3] = {w[0] / n, w[1] / n, w[2] / n};
        const double K[9] = {0.0, -a[2], a[1], a[2], 0.0, -a[0], -a[1], a[0], 0.0};
        const double ang = n * dt;
        const double s = sin(ang), c1 = 1.0 - cos(ang);
        double cK[9], cKK[9];
        for (int i = 0; i < 9; i++) cK[i] = c1 * K[i];
        m3_mul(cK, K, cKK);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + cKK[i];
    }
}

int orc_imu_undistort(orc_imu_proc *proc, orc_state18 *st, const orc_imu_sample *imu, int n_imu, double pcl_beg_time,
                      double pcl_end_time, float *pts_xyzt, int n, orc_pose6d *poses_out, int32_t *n_poses_out)
{
    const int nv = n_imu + 1;                                   /* v_imu = last_imu_ + meas.imu   :617-618 */
    orc_imu_sample *v = (orc_imu_sample *)malloc(sizeof(orc_imu_sample) * (size_t)nv);
    orc_pose6d *pose = (orc_pose6d *)malloc(sizeof(orc_pose6d) * (size_t)(nv + 1));
    if (!v || !pose) { free(v); free(pose); return -1; }
    v[0] = proc->last_imu;
    memcpy(v + 1, imu, sizeof(orc_imu_sample) * (size_t)n_imu);
    const double imu_end_time = v[nv - 1].t;
    int K = 0;
#define PUSH_POSE(T_, A_, G_, V_, P_, R_)                                                            \
    do {                                                                                             \
        pose[K].offset_time = (T_);                                                                  \
        memcpy(pose[K].acc, (A_), 24); memcpy(pose[K].gyr, (G_), 24); memcpy(pose[K].vel, (V_), 24); \
        memcpy(pose[K].pos, (P_), 24); memcpy(pose[K].rot, (R_), 72);                                \
        K++;                                                                                         \
    } while (0)
    PUSH_POSE(0.0, proc->acc_s_last, proc->angvel_last, st->vel, st->pos, st->rot);                      /* :656 */
    double acc_imu[3], angvel_avr[3], acc_avr[3], vel_imu[3], pos_imu[3], R_imu[9];
    memcpy(acc_imu, proc->acc_s_last, 24); memcpy(angvel_avr, proc->angvel_last, 24);
    memcpy(vel_imu, st->vel, 24); memcpy(pos_imu, st->pos, 24); memcpy(R_imu, st->rot, 72);
    const double mean_acc_norm = sqrt(proc->mean_acc[0] * proc->mean_acc[0] + proc->mean_acc[1] * proc->mean_acc[1] +
                                      proc->mean_acc[2] * proc->mean_acc[2]);
    double dt = 0;
    double *F = (double *)malloc(sizeof(double) * 324 * 3), *Tm = F + 324, *Cw = F + 648;
    for (int it = 0; it + 1 < nv; it++) {
        const orc_imu_sample *head = &v[it], *tail = &v[it + 1];
        if (tail->t < proc->last_lidar_end_time) continue;                                               /* :666 */
        for (int k = 0; k < 3; k++) {
            angvel_avr[k] = 0.5 * (head->gyr[k] + tail->gyr[k]);
            acc_avr[k] = 0.5 * (head->acc[k] + tail->acc[k]);
        }
        for (int k = 0; k < 3; k++) {
            angvel_avr[k] -= st->bg[k];                                                                  /* :684 */
            acc_avr[k] = acc_avr[k] * 9.81 / mean_acc_norm - st->ba[k];                                   /* :685 */
        }
        if (head->t < proc->last_lidar_end_time) dt = tail->t - proc->last_lidar_end_time;               /* :687-694 */
        else dt = tail->t - head->t;
        double Exp_f[9], Exp_m[9];
        so3_exp_dt(angvel_avr, dt, Exp_f);
        const double askew[9] = {0.0, -acc_avr[2], acc_avr[1], acc_avr[2], 0.0, -acc_avr[0], -acc_avr[1], acc_avr[0], 0.0};
        for (int i = 0; i < 324; i++) { F[i] = (i % 19 == 0) ? 1.0 : 0.0; Cw[i] = 0.0; }
        so3_exp_dt(angvel_avr, -dt, Exp_m);                                                              /* :703 */
        double RA[9];
        m3_mul(R_imu, askew, RA);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                F[i * 18 + j] = Exp_m[i * 3 + j];
                F[i * 18 + 9 + j] = (i == j) ? -dt : -0.0 * dt;                                          /* -Eye3d*dt       :704 */
                F[(3 + i) * 18 + 6 + j] = (i == j) ? dt : 0.0;                                           /* :706 */
                F[(6 + i) * 18 + j] = -RA[i * 3 + j] * dt;                                               /* :707 */
                F[(6 + i) * 18 + 12 + j] = -R_imu[i * 3 + j] * dt;                                       /* :708 */
                F[(6 + i) * 18 + 15 + j] = (i == j) ? dt : 0.0;                                          /* :709 */
            }
        for (int i = 0; i < 3; i++) {
            Cw[i * 18 + i] = proc->cov_gyr[i] * dt * dt;                                                 /* :711 */
            Cw[(9 + i) * 18 + 9 + i] = proc->cov_bias_gyr[i] * dt * dt;                                  /* :713 */
            Cw[(12 + i) * 18 + 12 + i] = proc->cov_bias_acc[i] * dt * dt;                                /* :714 */
            for (int j = 0; j < 3; j++) {                                                                /* R diag(cov_acc) R^T dt^2  :712 */
                double s = 0.0;
                for (int k = 0; k < 3; k++) s += R_imu[i * 3 + k] * proc->cov_acc[k] * R_imu[j * 3 + k];
                Cw[(6 + i) * 18 + 6 + j] = s * dt * dt;
            }
        }
        for (int i = 0; i < 18; i++)                                                                     /* :716 */
            for (int j = 0; j < 18; j++) {
                double s = 0.0;
                for (int k = 0; k < 18; k++) s += F[i * 18 + k] * st->cov[k * 18 + j];
                Tm[i * 18 + j] = s;
            }
        for (int i = 0; i < 18; i++)
            for (int j = 0; j < 18; j++) {
                double s = 0.0;
                for (int k = 0; k < 18; k++) s += Tm[i * 18 + k] * F[j * 18 + k];
                st->cov[i * 18 + j] = s + Cw[i * 18 + j];
            }
        m3_mul(R_imu, Exp_f, R_imu);                                                                     /* :719 */
        double Ra[3];
        m3_mv(R_imu, acc_avr, Ra);
        for (int k = 0; k < 3; k++) acc_imu[k] = Ra[k] + st->grav[k];                                    /* :722 */
        for (int k = 0; k < 3; k++) pos_imu[k] = pos_imu[k] + vel_imu[k] * dt + 0.5 * acc_imu[k] * dt * dt;   /* :725 */
        for (int k = 0; k < 3; k++) vel_imu[k] = vel_imu[k] + acc_imu[k] * dt;                           /* :728 */
        memcpy(proc->angvel_last, angvel_avr, 24);                                                       /* :731-732 */
        memcpy(proc->acc_s_last, acc_imu, 24);
        PUSH_POSE(tail->t - pcl_beg_time, acc_imu, angvel_avr, vel_imu, pos_imu, R_imu);                  /* :733-735 */
    }
    /* frame-end prediction :743-759 */
    {
        double note;
        if (imu_end_time > pcl_beg_time) { note = pcl_end_time > imu_end_time ? 1.0 : -1.0; dt = note * (pcl_end_time - imu_end_time); }
        else { note = pcl_end_time > pcl_beg_time ? 1.0 : -1.0; dt = note * (pcl_end_time - pcl_beg_time); }
        double w[3] = {note * angvel_avr[0], note * angvel_avr[1], note * angvel_avr[2]}, E[9];
        so3_exp_dt(w, dt, E);
        for (int k = 0; k < 3; k++) st->vel[k] = vel_imu[k] + note * acc_imu[k] * dt;
        m3_mul(R_imu, E, st->rot);
        for (int k = 0; k < 3; k++) st->pos[k] = pos_imu[k] + note * vel_imu[k] * dt + note * 0.5 * acc_imu[k] * dt * dt;
    }
    proc->last_imu = v[nv - 1];                                                                          /* :761-762 */
    proc->last_lidar_end_time = pcl_end_time;
    double LT[9], RT[9], extR_Ri[9], exrR_extT[3];
    m3_T(proc->Lid_rot_to_IMU, LT); m3_T(st->rot, RT);
    m3_mul(LT, RT, extR_Ri);                                                                             /* :764 */
    m3_mv(LT, proc->Lid_offset_to_IMU, exrR_extT);                                                       /* :765 */
    if (poses_out) memcpy(poses_out, pose, sizeof(orc_pose6d) * (size_t)K);
    if (n_poses_out) *n_poses_out = K;
    /* backward propagation :776-808 */
    if (n >= 1) {
        int ip = n - 1;
        for (int kp = K - 1; kp != 0; kp--) {
            const orc_pose6d *head = &pose[kp - 1];
            for (; (double)pts_xyzt[4 * (size_t)ip + 3] / 1000.0 > head->offset_time; ip--) {
                float *p = pts_xyzt + 4 * (size_t)ip;
                dt = (double)p[3] / 1000.0 - head->offset_time;
                double E[9], R_i[9], T_ei[3], P_i[3] = {p[0], p[1], p[2]}, a[3], b[3], c[3];
                so3_exp_dt(head->gyr, dt, E);
                m3_mul(head->rot, E, R_i);
                for (int k = 0; k < 3; k++) T_ei[k] = head->pos[k] + head->vel[k] * dt + 0.5 * head->acc[k] * dt * dt - st->pos[k];
                m3_mv(proc->Lid_rot_to_IMU, P_i, a);
                for (int k = 0; k < 3; k++) a[k] += proc->Lid_offset_to_IMU[k];
                m3_mv(R_i, a, b);
                for (int k = 0; k < 3; k++) b[k] += T_ei[k];
                m3_mv(extR_Ri, b, c);
                for (int k = 0; k < 3; k++) p[k] = (float)(c[k] - exrR_extT[k]);
                if (ip == 0) break;
            }
        }
    }
    free(F); free(v); free(pose);
    return 0;
}
