// ikfom_solve_block.h -- workgroup-cooperative form of one iteration of
// esekf::update_iterated_dyn_share_modified (esekfom.hpp:1644-1926): same mathematics as the one-thread
// fl_ikfom_iterate (fl_ikfom_math.h, unit-tested on the host), with the O(n^2 k) / O(n^3) loops spread over
// the 256 threads of the solver workgroup and every matrix in LDS. ~290 us -> ~20 us per pass.
//
// The reference applies the SO3/S2 projection Jacobians block by block, in place; the blocks are
// disjoint, so the result equals Jf P Jf^T with the block-diagonal Jf = diag(I3, J_rot, J_off, I12, J_S2)
// -- formed here element-wise (rounding-level differences only, compared by tolerance).
#pragma once

#include "fl_device.h"
#include "fl_ikfom_math.h"

struct FlIkLds {
    double P[529];       // projected P_ (then the final P_ on the finishing pass)
    double Pc[529];      // P_ with columns re-projected (final block)
    double L[529];       // L_
    double S[144], A12[144], SA[144], M[144], X[144];
    double Kx[276];      // 23 x 12
    double x[FL_X23_LEN], xp[FL_X23_LEN];
    double dx[23], dxn[23], dxo[23];
    double rhs[12], y[12];
    double J[3][9];      // J_rot, J_off (3x3), J_S2 (2x2 in the first 4 entries)
    int ctl[8];          // 0 t_count, 1 converge, 2 finished, 3 status
};

// block-diagonal Jf: element (r, a); blk = start index of r's block, bs = its size
__device__ __forceinline__ void ik_blk(int r, int &blk, int &bs, int &which)
{
    if (r >= 3 && r < 6) { blk = 3; bs = 3; which = 0; }
    else if (r >= 6 && r < 9) { blk = 6; bs = 3; which = 1; }
    else if (r >= 21) { blk = 21; bs = 2; which = 2; }
    else { blk = r; bs = 1; which = -1; }
}
__device__ __forceinline__ double ik_J(const FlIkLds &L, int which, int bs, int r_in, int a_in)
{
    return (which < 0) ? 1.0 : L.J[which][r_in * bs + a_in];
}
// thread 0..2: the three projection Jacobians for segments of `seg` (dx or dx_)
__device__ __forceinline__ void ik_make_J(FlIkLds &L, const double *seg, int tid)
{
    if (tid == 0) fl_ikfom_J_so3(seg + 3, L.J[0]);
    else if (tid == 1) fl_ikfom_J_so3(seg + 6, L.J[1]);
    else if (tid == 2) fl_ikfom_J_s2(L.x + FL_X23_GRAV, L.xp + FL_X23_GRAV, seg + 21, L.J[2]);
}

__device__ __forceinline__ void ikfom_solver_block(FlDev23 *__restrict__ D, const double *s_sums, FlIkLds &L, int gst)
{
    const int tid = threadIdx.x, NTH = blockDim.x, n = FL_N23;
    // ---- stage 0: state to LDS
    if (tid < FL_X23_LEN) { L.x[tid] = D->x[tid]; L.xp[tid] = D->xprop[tid]; }
    __syncthreads();
    // ---- stage 1: dx = x (-) x_prop  (serial, small)
    if (tid == 0) {
        double x[FL_X23_LEN], xp[FL_X23_LEN], dx[23];
        for (int i = 0; i < FL_X23_LEN; i++) { x[i] = L.x[i]; xp[i] = L.xp[i]; }
        fl_x23_boxminus(x, xp, dx);
        for (int i = 0; i < 23; i++) { L.dx[i] = dx[i]; L.dxn[i] = dx[i]; }
    }
    __syncthreads();
    ik_make_J(L, L.dx, tid);
    __syncthreads();
    // ---- stage 2: P = Jf Pprop Jf^T ; dx_new = Jf dx
    for (int e = tid; e < n * n; e += NTH) {
        const int r = e / n, c = e % n;
        int rb, rs, rw, cb, cs, cw;
        ik_blk(r, rb, rs, rw);
        ik_blk(c, cb, cs, cw);
        double s = 0.0;
        for (int a = 0; a < rs; a++) {
            const double jr = ik_J(L, rw, rs, r - rb, a);
            for (int b = 0; b < cs; b++) s += jr * D->Pprop[(rb + a) * n + (cb + b)] * ik_J(L, cw, cs, c - cb, b);
        }
        L.P[e] = s;
    }
    if (tid < n) {
        int rb, rs, rw;
        ik_blk(tid, rb, rs, rw);
        double s = 0.0;
        for (int a = 0; a < rs; a++) s += ik_J(L, rw, rs, tid - rb, a) * L.dx[rb + a];
        L.dxn[tid] = s;
    }
    if (tid >= 64 && tid < 64 + 78) {   // unpack S (upper triangle, row-major)
        int k = tid - 64, i = 0, rowlen = 12;
        while (k >= rowlen) { k -= rowlen; rowlen--; i++; }
        const int j = i + k;
        const double v = s_sums[tid - 64];
        L.S[i * 12 + j] = v;
        L.S[j * 12 + i] = v;
    }
    __syncthreads();
    const double R = D->meas_cov;
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        L.A12[tid] = 0.5 * (L.P[i * n + j] + L.P[j * n + i]) / R;
    }
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += L.S[i * 12 + k] * L.A12[k * 12 + j];
        L.SA[tid] = s;
    }
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        double s = L.A12[tid];
        for (int k = 0; k < 12; k++) s += L.A12[i * 12 + k] * L.SA[k * 12 + j];
        L.X[tid] = s;                       // unsymmetrised M, staged in X
    }
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        L.M[tid] = (i == j) ? L.X[tid] : 0.5 * (L.X[i * 12 + j] + L.X[j * 12 + i]);
    }
    if (tid >= 160 && tid < 172) {          // rhs = HTh + S dx_new12
        const int i = tid - 160;
        double s = s_sums[FL_S23_HTZ + i];
        for (int k = 0; k < 12; k++) s += L.S[i * 12 + k] * L.dxn[k];
        L.rhs[i] = s;
    }
    __syncthreads();
    // ---- stage 3: Cholesky of M (12x12), column by column
    int bad = 0;
    for (int j = 0; j < 12; j++) {
        if (tid == 0) {
            double d = L.M[j * 12 + j];
            for (int k = 0; k < j; k++) d -= L.M[j * 12 + k] * L.M[j * 12 + k];
            if (!(d > 0.0)) bad = 1;
            L.M[j * 12 + j] = sqrt(d);
        }
        __syncthreads();
        if (tid > j && tid < 12) {
            double v = L.M[tid * 12 + j];
            for (int k = 0; k < j; k++) v -= L.M[tid * 12 + k] * L.M[j * 12 + k];
            L.M[tid * 12 + j] = v / L.M[j * 12 + j];
        }
        __syncthreads();
    }
    if (tid < 12) {                         // y0 = A12 rhs
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += L.A12[tid * 12 + k] * L.rhs[k];
        L.y[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double y[12];
        for (int i = 0; i < 12; i++) y[i] = L.y[i];
        fl_chol_solve(L.M, 12, y);
        for (int i = 0; i < 12; i++) L.y[i] = y[i];
    }
    __syncthreads();
    if (tid < n) {                          // dx_ = A[:,0:12] y - dx_new
        double s = 0.0;
        for (int c = 0; c < 12; c++) s += (L.P[tid * n + c] / R) * L.y[c];
        L.dxo[tid] = s - L.dxn[tid];
    }
    __syncthreads();
    // ---- stage 4: boxplus, convergence bookkeeping (serial, small)
    if (tid == 0) {
        double x[FL_X23_LEN], dxo[23];
        for (int i = 0; i < FL_X23_LEN; i++) x[i] = L.x[i];
        for (int i = 0; i < 23; i++) dxo[i] = L.dxo[i];
        fl_x23_boxplus(x, dxo);
        int converge = 1, st = gst | bad;
        for (int i = 0; i < n; i++) {
            if (fabs(dxo[i]) > D->limit[i]) { converge = 0; break; }
        }
        for (int i = 0; i < n; i++)
            if (!(fabs(dxo[i]) <= DBL_MAX)) st |= 2;
        int t = D->t_count;
        const int i_loop = D->iter_i, max_iter = D->max_iter;
        if (converge) t++;
        if (!t && i_loop == max_iter - 2) converge = 1;
        const int finishing = (t > 1 || i_loop == max_iter - 1) ? 1 : 0;
        for (int i = 0; i < FL_X23_LEN; i++) { L.x[i] = x[i]; D->x[i] = x[i]; }
        for (int i = 0; i < 23; i++) D->solution[i] = dxo[i];
        L.ctl[0] = t; L.ctl[1] = converge; L.ctl[2] = finishing; L.ctl[3] = st;
        D->t_count = t;
        D->need_search = converge;
        D->converged = converge;
        D->iter_i = i_loop + 1;
        D->stop = (finishing || (i_loop + 1) >= max_iter) ? 1 : 0;
        D->neff = (int)s_sums[FL_S23_NEFF];
        D->total_residual = s_sums[FL_S23_RES];
        D->status = st;
        D->iters_run = D->iters_run + 1;
    }
    if (tid >= 64 && tid < 64 + FL_SUMS23) D->sums[tid - 64] = s_sums[tid - 64];
    __syncthreads();
    if (!L.ctl[2]) {                        // not finishing: publish the projected P_ and return
        for (int e = tid; e < n * n; e += NTH) D->P[e] = L.P[e];
        return;
    }
    // ---- stage 5: final covariance block, esekfom.hpp:1831-1924
    ik_make_J(L, L.dxo, tid);               // Jacobians at dx_ (S2: Nx at the UPDATED state, L.x)
    if (tid >= 32 && tid < 44) {            // X[:,c] = M^-1 (A12 S)[:,c]
        const int c = tid - 32;
        double col[12];
        for (int i = 0; i < 12; i++) {
            double s = 0.0;
            for (int k = 0; k < 12; k++) s += L.A12[i * 12 + k] * L.S[k * 12 + c];
            col[i] = s;
        }
        fl_chol_solve(L.M, 12, col);
        for (int i = 0; i < 12; i++) L.X[i * 12 + c] = col[i];
    }
    __syncthreads();
    for (int e = tid; e < n * 12; e += NTH) {   // Kx = A[:,0:12] X
        const int r = e / 12, c = e % 12;
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += (L.P[r * n + k] / R) * L.X[k * 12 + c];
        L.Kx[e] = s;
    }
    for (int e = tid; e < n * n; e += NTH) {    // Pc = P Jf^T (columns)
        const int r = e / n, c = e % n;
        int cb, cs, cw;
        ik_blk(c, cb, cs, cw);
        double s = 0.0;
        for (int b = 0; b < cs; b++) s += L.P[r * n + cb + b] * ik_J(L, cw, cs, c - cb, b);
        L.Pc[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NTH) {    // L_ = Jf Pc (rows)
        const int r = e / n, c = e % n;
        int rb, rs, rw;
        ik_blk(r, rb, rs, rw);
        double s = 0.0;
        for (int a = 0; a < rs; a++) s += ik_J(L, rw, rs, r - rb, a) * L.Pc[(rb + a) * n + c];
        L.L[e] = s;
    }
    for (int e = tid; e < n * 12; e += NTH) {   // Kx' = Jf Kx (rows), staged in P (free from here on)
        const int r = e / 12, c = e % 12;
        int rb, rs, rw;
        ik_blk(r, rb, rs, rw);
        double s = 0.0;
        for (int a = 0; a < rs; a++) s += ik_J(L, rw, rs, r - rb, a) * L.Kx[(rb + a) * 12 + c];
        L.P[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NTH) {    // P_ = L_ - Kx'[:,0:12] Pc[0:12,:]
        const int r = e / n, c = e % n;
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += L.P[r * 12 + k] * L.Pc[k * n + c];
        D->P[e] = L.L[e] - s;
    }
}
