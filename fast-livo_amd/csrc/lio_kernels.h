// lio_kernels.h -- gfx950 kernels for the LiDAR point-to-plane ESKF iteration (Mode-18).
//
//  K0  lio_fit_planes_kernel   once per neighbour staging: 5-NN -> plane (n,d) + selection flag.
//                              The plane depends only on the neighbours, not on the state, so the
//                              reference's per-iteration esti_plane (laserMapping.cpp:1571) is
//                              hoisted out of the iteration loop with bit-identical results.
//  K1  lio18_iterate_kernel    one ESKF pass: per-point residual + gates + 1x6 row, fp64 wave /
//                              block reduction of the 32-double record, write-through hand-off of
//                              per-block partials, and -- in the last-arriving workgroup -- the
//                              fixed-order final reduce, the gain solve, the state update and the
//                              rematch/stop judgement. One launch per pass, no host round trip.
//  K1a lio18_accumulate_kernel sharded form: same but stops after the final reduce (sums out).
//  K3  eskf18_solve_kernel     sharded form: gain solve from an (all-reduced) record.
//  K4  lio18_finish_kernel     P <- (I - G) P   (laserMapping.cpp:1715)
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "solve18.h"

#define FL_ITER_FORCE 1
#define FL_ITER_KEEP_NORMVEC 2

// -------------------------------------------------------------------------------------------- K0
__global__ __launch_bounds__(FL_BLOCK) void lio_fit_planes_kernel(const float *__restrict__ nbr, const uint8_t *__restrict__ valid,
                                                                 float4 *__restrict__ plane, uint8_t *__restrict__ sel, int n)
{
    // Coalesced staging of this block's 256 x 15 floats through LDS, then one point per lane.
    __shared__ float s_nb[FL_BLOCK * 15];
    const int base = blockIdx.x * FL_BLOCK;
    const int cnt = min(FL_BLOCK, n - base);
    const float *src = nbr + (size_t)base * 15;
    for (int i = threadIdx.x; i < cnt * 15; i += FL_BLOCK) s_nb[i] = src[i];
    __syncthreads();
    const int i = base + threadIdx.x;
    if (threadIdx.x >= cnt) return;
    float nb[15];
#pragma unroll
    for (int k = 0; k < 15; k++) nb[k] = s_nb[threadIdx.x * 15 + k];   // stride 15 dwords: conflict-free
    float pl[4];
    const int ok = fl_esti_plane(nb, pl);
    plane[i] = make_float4(pl[0], pl[1], pl[2], pl[3]);
    sel[i] = (uint8_t)((valid[i] != 0) && ok);
}

// ------------------------------------------------------------------------------------ epilogues
// Rematch / stop judgement of the Mode-18 loop, laserMapping.cpp:1688-1728 (one thread).
__device__ __forceinline__ void lio18_judge(FlDev18 *D, const double *delta, const double *sums, int st)
{
    const double rn = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    const double tn = sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]);
    const int converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
    int rematch = D->rematch_num, need_search = 0, stop = 0;
    const int it = D->iterCount;
    if (converged || ((rematch == 0) && (it == (D->max_iter - 2)))) { need_search = 1; rematch++; }
    if (rematch >= 2 || (it == D->max_iter - 1)) stop = 1;
    D->converged = converged;
    D->rematch_num = rematch;
    D->need_search = need_search;
    D->stop = stop;
    D->iterCount = it + 1;
    D->iters_run = D->iters_run + 1;
    D->neff = (int)sums[FL_S_NEFF];
    D->total_residual = sums[FL_S_RES];
    D->status = st | ((sums[FL_S_NEFF] < 1.0) ? 4 : 0);
}

// Per-frame prepare: Q and T of fl_math.h (depends on P and the measurement covariance only).
__global__ void eskf18_prepare_kernel(FlDev18 *__restrict__ D)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double Q[36], T[108];
    const int st = fl_prepare18(D->P, D->meas_cov, Q, T);
    for (int i = 0; i < 36; i++) D->Q[i] = Q[i];
    for (int i = 0; i < 108; i++) D->T[i] = T[i];
    D->status = st;
}

// Gain solve + state update executed by one thread of the final workgroup.
__device__ __forceinline__ void eskf18_solve_serial(FlDev18 *D, const double *sums, double sign)
{
    double x[24], xp[24], delta[18];
#pragma unroll
    for (int i = 0; i < 24; i++) { x[i] = D->x[i]; xp[i] = D->xprop[i]; }
    const int st = fl_solve18_fast(x, xp, D->Q, D->T, sums, sign, delta);
#pragma unroll
    for (int i = 0; i < 24; i++) D->x[i] = x[i];
#pragma unroll
    for (int i = 0; i < 18; i++) D->solution[i] = delta[i];
#pragma unroll
    for (int i = 0; i < FL_SUMS18; i++) { D->sums[i] = sums[i]; D->sums_acc[i] = sums[i]; }
    lio18_judge(D, delta, sums, st);
}

// -------------------------------------------------------------------------------------------- K1
// mode 0: fused (accumulate + final reduce + solve).  mode 1: accumulate only, sums -> sums_out.
template <int MODE>
__global__ __launch_bounds__(FL_BLOCK) void lio18_iterate_kernel(const float *__restrict__ body, const float4 *__restrict__ plane,
                                                                uint8_t *__restrict__ sel, float4 *__restrict__ normvec, int n,
                                                                FlDev18 *__restrict__ D, double *__restrict__ partials,
                                                                unsigned *__restrict__ ticket, double *__restrict__ sums_out, int flags)
{
    if (!(flags & FL_ITER_FORCE) && (D->stop || D->need_search)) return;
    __shared__ double s_red[4 * FL_SUMS18];
    __shared__ double s_fin[FL_FIN_LDS];
    __shared__ double s_sums[FL_SUMS18];
    __shared__ FlSolveLds s_solve;

    double R[9], p[3], RLI[9], tLI[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = D->x[i]; RLI[i] = D->R_LI[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { p[i] = D->x[9 + i]; tLI[i] = D->t_LI[i]; }

    double v[FL_SUMS18];
#pragma unroll
    for (int k = 0; k < FL_SUMS18; k++) v[k] = 0.0;
    if (blockIdx.x == 0) fl_stamp(flags, 0);

    for (int i = blockIdx.x * FL_BLOCK + threadIdx.x; i < n; i += gridDim.x * FL_BLOCK) {
        if (!sel[i]) continue;
        const float pb[3] = {body[i * 3 + 0], body[i * 3 + 1], body[i * 3 + 2]};
        const float4 plq = plane[i];
        const float pl[4] = {plq.x, plq.y, plq.z, plq.w};
        double p_i[3];
        float pw[3], pd2;
        int eff;
        const int s = fl_point_gates(pb, pl, R, p, RLI, tLI, p_i, pw, &pd2, &eff);
        if (!s) sel[i] = 0;
        if ((flags & FL_ITER_KEEP_NORMVEC) && s) normvec[i] = make_float4(pl[0], pl[1], pl[2], pd2);
        if (eff) {
            double row[6], z;
            fl_row18(p_i, pl, pd2, R, row, &z);
            fl_accum6(v, row, z);
            v[FL_S_NEFF] += 1.0;
            v[FL_S_RES] += (double)fabsf(pd2);
            v[FL_S_RES2] += (double)pd2 * (double)pd2;
        }
    }

    if (blockIdx.x == 0) fl_stamp(flags, 1);
    const bool last = block_publish<FL_SUMS18>(v, partials, ticket, s_red);
    if (blockIdx.x == 0) fl_stamp(flags, 2);
    if (!last) return;
    fl_stamp(flags, 8);
    final_reduce<FL_SUMS18>(partials, gridDim.x, s_fin, s_sums);
    fl_stamp(flags, 9);
    if (threadIdx.x == 0) *ticket = 0u;
    if (MODE == 0) {
        eskf18_epilogue_block<FL_EPI_LIO>(D, s_sums, s_solve);
        fl_stamp(flags, 10);
    } else {
        if (threadIdx.x < FL_SUMS18) sums_out[threadIdx.x] = s_sums[threadIdx.x];
    }
}

// -------------------------------------------------------------------------------------------- K3
__global__ void eskf18_solve_kernel(FlDev18 *__restrict__ D, const double *__restrict__ sums_in, double sign, int vio, int flags);

// -------------------------------------------------------------------------------------------- K4
// P <- P - G6 * P[0:6,:]  (== (I - G) P with G's columns 6..17 zero)
__global__ __launch_bounds__(384) void eskf18_cov_update_kernel(FlDev18 *__restrict__ D)
{
    __shared__ double sP[324];
    __shared__ double sG[108];
    const int t = threadIdx.x;
    if (t < 324) sP[t] = D->P[t];
    if (t == 0) {   // G[:,0:6] of the last executed pass
        double G6[108];
        fl_gain18(D->Q, D->T, D->sums_acc, G6);
        for (int i = 0; i < 108; i++) { sG[i] = G6[i]; D->G6[i] = G6[i]; }
    }
    __syncthreads();
    if (t < 324) {
        const int r = t / 18, c = t % 18;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += sG[r * 6 + k] * sP[k * 18 + c];
        D->P[t] = sP[t] - s;
    }
}

// world points at the current device state (pointBodyToWorld) for the host kNN on rematch passes
__global__ __launch_bounds__(FL_BLOCK) void lio_world_points_kernel(const float *__restrict__ body, float *__restrict__ world, int n,
                                                                   const FlDev18 *__restrict__ D)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    double R[9], p[3], RLI[9], tLI[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { R[k] = D->x[k]; RLI[k] = D->R_LI[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { p[k] = D->x[9 + k]; tLI[k] = D->t_LI[k]; }
    const double b0 = (double)body[i * 3], b1 = (double)body[i * 3 + 1], b2 = (double)body[i * 3 + 2];
    const double q0 = (RLI[0] * b0 + RLI[1] * b1 + RLI[2] * b2) + tLI[0];
    const double q1 = (RLI[3] * b0 + RLI[4] * b1 + RLI[5] * b2) + tLI[1];
    const double q2 = (RLI[6] * b0 + RLI[7] * b1 + RLI[8] * b2) + tLI[2];
    world[i * 3 + 0] = (float)((R[0] * q0 + R[1] * q1 + R[2] * q2) + p[0]);
    world[i * 3 + 1] = (float)((R[3] * q0 + R[4] * q1 + R[5] * q2) + p[1]);
    world[i * 3 + 2] = (float)((R[6] * q0 + R[7] * q1 + R[8] * q2) + p[2]);
}
