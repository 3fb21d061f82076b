// vio_kernels.h -- gfx950 kernels for the 8x8-patch photometric ESKF update
// (LidarSelector::UpdateState / ComputeJ, src/lidar_selection.cpp:743-983).
//
//  K2  vio_iterate_kernel   one iteration at one pyramid level: ONE WAVEFRONT PER PATCH, lane =
//                           8*x + y = one pixel. Each lane gathers its 12 u8 taps (bilinear value +
//                           central-difference gradient), forms the 1x6 Jacobian row in fp64,
//                           the wave reduces the 32-double record with the transposing butterfly,
//                           the workgroup publishes a partial (write-through), and the last
//                           workgroup does the fixed-order final reduce + accept/revert + gain
//                           solve + state update.  One launch per iteration, no host round trip.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "lio_kernels.h"

// Photometric measurement for one pixel of one patch. All wave-uniform inputs are precomputed by
// the caller (same values in every lane). Operation order mirrors lidar_selection.cpp:826-837 so the
// float part rounds identically to the reference.
struct FlPatchGeom {
    double Jdpi[6];     // dpi(pf), :92-103
    double pf[3];
    float wtl, wtr, wbl, wbr;
    int u_i, v_i, scale;
};

FL_HD void fl_world2cam(const FlVioConst &c, const double *pf, double *pc)
{
    const double u = pf[0] / pf[2], v = pf[1] / pf[2];
    if (!c.distort) {
        pc[0] = c.fx * u + c.cx;
        pc[1] = c.fy * v + c.cy;
    } else {   // vk::PinholeCamera radtan (rpg_vikit, restated from memory -- see oracle/orc_vio.c)
        const double r2 = u * u + v * v, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * u * v, a2 = r2 + 2 * u * u, a3 = r2 + 2 * v * v;
        const double cdist = 1 + c.d[0] * r2 + c.d[1] * r4 + c.d[4] * r6;
        const double xd = u * cdist + c.d[2] * a1 + c.d[3] * a2;
        const double yd = v * cdist + c.d[3] * a1 + c.d[2] * a3;
        pc[0] = xd * c.fx + c.cx;
        pc[1] = yd * c.fy + c.cy;
    }
}

FL_HD void fl_patch_geom(const FlVioConst &c, const double *Rcw, const double *Pcw, const double *pos, int scale, FlPatchGeom &g)
{
    g.pf[0] = (Rcw[0] * pos[0] + Rcw[1] * pos[1] + Rcw[2] * pos[2]) + Pcw[0];
    g.pf[1] = (Rcw[3] * pos[0] + Rcw[4] * pos[1] + Rcw[5] * pos[2]) + Pcw[1];
    g.pf[2] = (Rcw[6] * pos[0] + Rcw[7] * pos[1] + Rcw[8] * pos[2]) + Pcw[2];
    double pc[2];
    fl_world2cam(c, g.pf, pc);
    const double z_inv = 1. / g.pf[2], z_inv_2 = z_inv * z_inv;
    g.Jdpi[0] = c.fx_abs * z_inv; g.Jdpi[1] = 0.0; g.Jdpi[2] = -c.fx_abs * g.pf[0] * z_inv_2;
    g.Jdpi[3] = 0.0; g.Jdpi[4] = c.fy_abs * z_inv; g.Jdpi[5] = -c.fy_abs * g.pf[1] * z_inv_2;
    const float u_ref = (float)pc[0], v_ref = (float)pc[1];
    g.scale = scale;
    g.u_i = (int)(floorf((float)(pc[0] / scale)) * scale);
    g.v_i = (int)(floorf((float)(pc[1] / scale)) * scale);
    const float su = (u_ref - g.u_i) / scale;
    const float sv = (v_ref - g.v_i) / scale;
    g.wtl = (float)((1.0 - su) * (1.0 - sv));
    g.wtr = (float)(su * (1.0 - sv));
    g.wbl = (float)((1.0 - su) * sv);
    g.wbr = su * sv;
}

// taps: t[r][c] = img[(row0 + (r-1)*scale) , (col0 + (c-1)*scale)], r,c in 0..3 (only 12 used)
FL_HD void fl_pixel_row(const FlPatchGeom &g, const float t[4][4], float ref, const double *Jdphi_dR, const double *Jdp_dR,
                        const double *Jdp_dt, double *row /*6*/, double *res_out)
{
    const float wtl = g.wtl, wtr = g.wtr, wbl = g.wbl, wbr = g.wbr;
    // t[1][1] = img_ptr[0]; columns: [0]=-s [1]=0 [2]=+s [3]=+2s ; rows: [0]=-sW [1]=0 [2]=+sW [3]=+2sW
    const float du = 0.5f * ((wtl * t[1][2] + wtr * t[1][3] + wbl * t[2][2] + wbr * t[2][3])
                           - (wtl * t[1][0] + wtr * t[1][1] + wbl * t[2][0] + wbr * t[2][1]));
    const float dv = 0.5f * ((wtl * t[2][1] + wtr * t[2][2] + wbl * t[3][1] + wbr * t[3][2])
                           - (wtl * t[0][1] + wtr * t[0][2] + wbl * t[1][1] + wbr * t[1][2]));
    const double inv_s = 1.0 / g.scale;
    const double J0 = (double)du * inv_s, J1 = (double)dv * inv_s;
    double JJ[3], Jdphi[3], Jdp[3];
#pragma unroll
    for (int c = 0; c < 3; c++) JJ[c] = J0 * g.Jdpi[c] + J1 * g.Jdpi[3 + c];
    // p_hat = skew(pf): [0,-z,y; z,0,-x; -y,x,0]
    const double px = g.pf[0], py = g.pf[1], pz = g.pf[2];
    Jdphi[0] = JJ[0] * 0.0 + JJ[1] * pz + JJ[2] * (-py);
    Jdphi[1] = JJ[0] * (-pz) + JJ[1] * 0.0 + JJ[2] * px;
    Jdphi[2] = JJ[0] * py + JJ[1] * (-px) + JJ[2] * 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) Jdp[c] = (-J0) * g.Jdpi[c] + (-J1) * g.Jdpi[3 + c];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double a = Jdphi[0] * Jdphi_dR[c] + Jdphi[1] * Jdphi_dR[3 + c] + Jdphi[2] * Jdphi_dR[6 + c];
        const double b = Jdp[0] * Jdp_dR[c] + Jdp[1] * Jdp_dR[3 + c] + Jdp[2] * Jdp_dR[6 + c];
        row[c] = a + b;
        row[3 + c] = Jdp[0] * Jdp_dt[c] + Jdp[1] * Jdp_dt[3 + c] + Jdp[2] * Jdp_dt[6 + c];
    }
    *res_out = (double)(wtl * t[1][1] + wtr * t[1][2] + wbl * t[2][1] + wbr * t[2][2] - ref);
}

#define FL_VIO_NT 512

// grid = producers + 1 ; MODE 0: fused pass ; MODE 1: accumulate only (sums -> sums_out).
// 512-thread workgroups (8 patches in flight per workgroup): 2000 patches -> 250 records, which the
// solver workgroup gathers in a single sweep.
template <int MODE>
__global__ __launch_bounds__(FL_VIO_NT) void vio_pass_kernel(const uint8_t *__restrict__ img, const float *__restrict__ ref,
                                                            const double *__restrict__ pos, const int32_t *__restrict__ slevel,
                                                            float *__restrict__ errors, int m, int level_arg,
                                                            const FlVioConst *__restrict__ VC, FlDev18 *__restrict__ D,
                                                            void *__restrict__ records, unsigned *__restrict__ epoch_ptr,
                                                            double *__restrict__ sums_out, int flags)
{
    constexpr int NT = FL_VIO_NT;
    constexpr int WPB = NT / 64;
    if (!(flags & FL_ITER_FORCE) && D->stop) return;
    const unsigned epoch = *epoch_ptr;
    const int nprod = gridDim.x - 1;

    if (blockIdx.x == nprod) {
        // ------------------------------------------------------------------ solver workgroup
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS18];
        __shared__ FlSolveLds s_solve;
        fl_stamp(flags, 8);
        if (MODE == 0) eskf18_prefetch(D, s_solve);
        fl_stamp(flags, 9);
        const int gst = gather_records<NT, FL_SUMS18>(records, nprod, epoch, s_fin, s_sums);
        fl_stamp(flags, 10);
        if (threadIdx.x == 0) *epoch_ptr = epoch + 1u;
        if (MODE == 0) {
            eskf18_solve_block<FL_EPI_VIO>(D, s_sums, s_solve, gst);
        } else {
            if (threadIdx.x < FL_SUMS18) sums_out[threadIdx.x] = s_sums[threadIdx.x];
        }
        fl_stamp(flags, 11);
        return;
    }

    // -------------------------------------------------------------------- producer workgroups
    __shared__ double s_red[WPB * FL_SUMS18];
    const int level = (level_arg >= 0) ? level_arg : D->level;

    // wave-uniform camera pose: Rcw = Rci Rwi^T, Pcw = -Rci Rwi^T Pwi + Pci  (:780-784)
    const FlVioConst vc = *VC;
    double Rwi[9], Rwit[9], Rcw[9], nRci[9], T[9], Pcw[3], Pwi[3];
#pragma unroll
    for (int i = 0; i < 9; i++) Rwi[i] = D->x[i];
#pragma unroll
    for (int i = 0; i < 3; i++) Pwi[i] = D->x[9 + i];
    m3_tr(Rwi, Rwit);
    m3_mul(vc.Rci, Rwit, Rcw);
#pragma unroll
    for (int i = 0; i < 9; i++) nRci[i] = -vc.Rci[i];
    m3_mul(nRci, Rwit, T);
    m3_vec(T, Pwi, Pcw);
#pragma unroll
    for (int i = 0; i < 3; i++) Pcw[i] = Pcw[i] + vc.Pci[i];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xr = lane >> 3, yc = lane & 7;
    const int W = vc.stride, Hm1 = vc.height - 1, Wm1 = vc.width - 1;

    double v[FL_SUMS18];
#pragma unroll
    for (int k = 0; k < FL_SUMS18; k++) v[k] = 0.0;

    if (blockIdx.x == 0) fl_stamp(flags, 0);
    for (int i = blockIdx.x * WPB + wave; i < m; i += nprod * WPB) {
        const int scale = 1 << (level + slevel[i]);
        const double ps[3] = {pos[i * 3 + 0], pos[i * 3 + 1], pos[i * 3 + 2]};
        FlPatchGeom g;
        fl_patch_geom(vc, Rcw, Pcw, ps, scale, g);
        const int row0 = g.v_i + (xr - 4) * scale;
        const int col0 = g.u_i + (yc - 4) * scale;
        float t[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int rr = row0 + (r - 1) * scale;
            rr = rr < 0 ? 0 : (rr > Hm1 ? Hm1 : rr);          // the reference reads unchecked; clamp instead of faulting
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                if (used) {
                    int cc = col0 + (c - 1) * scale;
                    cc = cc < 0 ? 0 : (cc > Wm1 ? Wm1 : cc);
                    t[r][c] = (float)img[rr * W + cc];
                } else {
                    t[r][c] = 0.f;
                }
            }
        }
        const float refv = ref[(size_t)i * 192 + 64 * level + lane];
        double row[6], res;
        fl_pixel_row(g, t, refv, vc.Jdphi_dR, vc.Jdp_dR, Rcw, row, &res);
        fl_accum6(v, row, res);
        v[FL_S_NEFF] += 1.0;
        const double r2 = res * res;
        v[FL_S_RES] += r2;
        const double pe = wave_sum(r2);
        if (lane == 0) errors[i] = (float)pe;
    }
    if (blockIdx.x == 0) fl_stamp(flags, 1);
    const double mine = block_reduce_record<NT, FL_SUMS18>(v, s_red);
    publish_record<FL_SUMS18>(mine, epoch, records, nprod);
    if (blockIdx.x == 0) fl_stamp(flags, 2);
}

// UpdateState prologue: old_state = *state, last_error = total_residual (:747,756); per-level counters.
__global__ void vio_level_begin_kernel(FlDev18 *__restrict__ D, int level, float total_residual)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 24; i++) D->xold[i] = D->x[i];
    D->last_error = total_residual;
    D->level = level;
    D->stop = 0;
    D->converged = 0;
    D->iters_run = 0;
    D->accepted = 0;
    D->status = 0;
}

struct FlVioLevelInfo {
    double solution[18];
    float error;
    int32_t iterations, n_meas, accepted, status, converged;
};
__global__ void vio_level_end_kernel(const FlDev18 *__restrict__ D, FlVioLevelInfo *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 18; i++) out->solution[i] = D->solution[i];
    out->error = D->last_error;
    out->iterations = D->iters_run;
    out->n_meas = D->neff;
    out->accepted = D->accepted;
    out->status = D->status;
    out->converged = D->converged;
}

// ComputeJ tail: if (now_error < error) state->cov -= G*state->cov  (:978-981)
__global__ __launch_bounds__(384) void vio_cov_update_kernel(FlDev18 *__restrict__ D)
{
    __shared__ double sP[324];
    __shared__ double sG[108];
    const int t = threadIdx.x;
    const bool apply = D->last_error < 1e10f;
    if (t < 324) sP[t] = D->P[t];
    if (t == 0) {   // G[:,0:6] of the last ACCEPTED iteration (G is only rewritten on acceptance, :877)
        double G6[108];
        fl_gain18(D->Q, D->T, D->sums_acc, G6);
        for (int i = 0; i < 108; i++) { sG[i] = G6[i]; D->G6[i] = G6[i]; }
    }
    __syncthreads();
    if (apply && t < 324) {
        const int r = t / 18, c = t % 18;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += sG[r * 6 + k] * sP[k * 18 + c];
        D->P[t] = sP[t] - s;
    }
}
