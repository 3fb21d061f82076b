// ikfom_solve_block.h -- workgroup-cooperative form of one iteration of
// esekf::update_iterated_dyn_share_modified (esekfom.hpp:1644-1926): same mathematics as the one-thread
// fl_ikfom_iterate (fl_ikfom_math.h, unit-tested on the host), with the O(n^2 k) / O(n^3) loops spread over
// the 256 threads of the solver workgroup and every matrix in LDS.
//
// The reference applies the SO3/S2 projection Jacobians block by block, in place; the blocks are
// disjoint, so the result equals Jf P Jf^T with the block-diagonal Jf = diag(I3, J_rot, J_off, I12, J_S2)
// -- formed here element-wise (rounding-level differences only, compared by tolerance).
//
// Round 2: the iteration is split where the measurement enters.
//   ikfom_pre   everything that depends on the state alone -- dx = x (-) x_prop with its three manifold logarithms, the
//               projection Jacobians, P = Jf P_prop Jf^T (23x23), dx_new, A12 = sym(P[0:12,0:12]) / R -- runs BEFORE the
//               records arrive, while the producers are still working (it was 6 of the solver's 14 us, all of it on the pass's
//               critical path);
//   ikfom_post  S = h_x^T h_x enters: M = A12 + A12 S A12, rhs; the 12x12 SPD solve as a right-looking LDL^T held in the
//               registers of every lane of wavefront 0 (reciprocal + Newton steps; no square roots, no divisions, no cross-lane
//               traffic: the v_readlane Cholesky with 12 sqrt and 36 divisions took 4.6 us), dx_, the three boxplus segments,
//               judgement, and -- on the finishing pass only -- the final covariance block (esekfom.hpp:1831-1924).
// A pass whose hand-off timed out is ABANDONED as in solve18.h: state untouched, FL_NUM_TIMEOUT sticky, the host resumes.
#pragma once

#include "fl_device.h"
#include "fl_ikfom_math.h"

struct FlIkLds {
    double P[529];       // projected P_ (then the final P_ on the finishing pass)
    double Pd[276];      // P_[:, 0:12] / R (23 x 12), formed once per pass by the threads the record unpacking leaves idle: the
                         // twelve IEEE divisions per row sat on wavefront 0's path behind the LDL^T (0.75 us) and again in the final block
    double Pc[529];      // P_ with columns re-projected (final block)
    double L[529];       // L_ (final block)
    double Pp[529];      // P_propagated, staged once per launch
    double S[144], A12[144], SA[144], M[144], X[144];
    double Kx[276];      // 23 x 12
    double x[FL_X23_LEN], xp[FL_X23_LEN];
    double dx[23], dxn[23], dxo[23];
    double rhs[12], y[12], y0[12];
    double J[3][9];      // J_rot, J_off (3x3), J_S2 (2x2 in the first 4 entries)
    int ctl[8];          // 0 t_count, 1 converge, 2 finishing, 3 status of this pass, 4 abandoned, 5 stop
    double limit[23];    // convergence limits, staged once
    int cnt[4];          // t_count, iter_i, max_iter, iters_run: kept current across the passes of a multi-pass launch
    int sticky;          // status bits accumulated since fl_ikfom_begin
    double R;
    double Jw[23][3];    // row r of the block-diagonal projection Jacobian Jf as three weights on the states jb[r] .. jb[r] + 2 (identity rows:
    int jb[24];          //   (1, 0, 0) on r itself): P = Jf Pprop Jf^T becomes a branch-free 9-term sum per element (ikfom_pre)
    double Rm[9];        // rotation matrix of x.rot (ikfom_pre): rebuilds the C block of h_x^T h_x from the internal record
    double htz[12];      // h_x^T h
    double scal[4];      // n_eff, sum |pd2|, sum pd2^2
};

// value `idx` of the PUBLIC 96-double record (78 upper-triangle entries of the 12 x 12, 12 x h_x^T h, n_eff, sum |pd2|, sum pd2^2)
// out of the internal 64-double one
__device__ __forceinline__ double ikfom_public_value(const double *s64, const double *Rm, int idx)
{
    if (idx < 78) {
        int k = idx, i = 0, rowlen = 12;
        while (k >= rowlen) { k -= rowlen; rowlen--; i++; }
        return fl_s12_from_s9(s64, Rm, i, i + k);
    }
    if (idx < 90) return fl_htz12_from_s9(s64, Rm, idx - 78);
    if (idx == FL_S23_NEFF) return s64[FL_S23I_NEFF];
    if (idx == FL_S23_RES) return s64[FL_S23I_RES];
    if (idx == FL_S23_RES2) return s64[FL_S23I_RES2];
    return 0.0;
}

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
// threads 0, 64, 128 (three different wavefronts: divergent lanes of one wave would run one after the other): the three
// projection Jacobians for segments of `seg` (dx or dx_)
__device__ __forceinline__ void ik_make_J(FlIkLds &L, const double *seg, int tid)
{
    if (tid == 0) fl_ikfom_J_so3(seg + 3, L.J[0]);
    else if (tid == 64) fl_ikfom_J_so3(seg + 6, L.J[1]);
    else if (tid == 128) fl_ikfom_J_s2(L.x + FL_X23_GRAV, L.xp + FL_X23_GRAV, seg + 21, L.J[2]);
}
__device__ __forceinline__ constexpr int ik_tri(int i, int j) { return i * (i + 1) / 2 + j; }   // j <= i
__device__ __forceinline__ double ik_rcp_nr(double d)
{
#pragma clang fp contract(fast)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}
// M y = w for the SPD 12x12 M = sym(X) (LDS), entirely in the registers of the calling lane: right-looking LDL^T with w as a 13th
// row (the forward substitution falls out of the elimination), reciprocal + Newton steps for the pivots, back substitution.
// Every lane may bring its own right-hand side. Returns 1 if a pivot is not positive.
__device__ __forceinline__ int ik_ldl_solve_regs(const double *X /* LDS, 144 */, const double (&w_in)[12], double (&y)[12])
{
#pragma clang fp contract(fast)
    double c[12][12];       // lower triangle used
    double w[12];
    int bad = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
#pragma unroll
        for (int j = 0; j < 12; j++)
            if (j <= i) c[i][j] = (i == j) ? X[i * 12 + i] : 0.5 * (X[i * 12 + j] + X[j * 12 + i]);
        w[i] = w_in[i];
    }
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const double dj = c[j][j];
        if (!(dj > 0.0)) bad = 1;
        const double inv = ik_rcp_nr(dj);
        double u[12];
#pragma unroll
        for (int i = 0; i < 12; i++)
            if (i > j) { u[i] = c[i][j]; c[i][j] = u[i] * inv; }
        const double wj = w[j] * inv;
        w[j] = wj;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            if (i > j) {
#pragma unroll
                for (int k = 0; k < 12; k++)
                    if (k > j && k <= i) c[i][k] = fma(-c[i][j], u[k], c[i][k]);
                w[i] = fma(-wj, u[i], w[i]);
            }
        }
    }
#pragma unroll
    for (int i = 11; i >= 0; i--) {
        double yi = w[i];
#pragma unroll
        for (int k = 0; k < 12; k++)
            if (k > i) yi = fma(-c[k][i], y[k], yi);
        y[i] = yi;
    }
    return bad;
}

// The same solve with ONE right-hand side, spread over the wavefront: lane i < 12 holds row i of sym(X) and w_i as a 13th column.
// Step j: row j (pivot, the row's tail and its right-hand side) is broadcast from lane j through scalar registers (v_readlane with a
// constant lane), every lane i > j scales its element of column j and updates its whole row tail -- the Schur complement stays
// symmetric, so the column a lane needs for the update IS row j. Lane j's row is final after step j: it holds d_j and d_j l_kj, i.e.
// column j of L, which is what the back substitution needs in lane j. 13 doubles per lane instead of 90, a dependent chain of
// 12 x (reciprocal + 2 fused multiply-adds) instead of 12 x (reciprocal + up to 11 updates deep): ~1.1 us instead of ~3 us (the full
// register version spilled into AGPRs). All 64 lanes call it; y (the whole solution) is returned in every lane.
__device__ __forceinline__ double ik_bcast_lane(double v, int src_lane /* compile-time constant after unrolling */)
{
    const int lo = __builtin_amdgcn_readlane((int)f64_lo(v), src_lane), hi = __builtin_amdgcn_readlane((int)f64_hi(v), src_lane);
    return f64_make((unsigned)lo, (unsigned)hi);
}
__device__ __forceinline__ int ik_ldl_solve_rows(const double *X /* LDS, 144 */, const double *w_lds /* LDS, 12 */, double (&y)[12])
{
#pragma clang fp contract(fast)
    const int lane = threadIdx.x & 63;
    const int r = lane < 12 ? lane : 0;
    double c[13];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const double v = (k == r) ? X[r * 12 + r] : 0.5 * (X[r * 12 + k] + X[k * 12 + r]);
        c[k] = lane < 12 ? v : 0.0;                      // (lanes >= 12 carry zero rows: their updates are no-ops)
    }
    c[12] = lane < 12 ? w_lds[r] : 0.0;
    int bad = 0;
    double myinv = 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const double dj = ik_bcast_lane(c[j], j);
        if (!(dj > 0.0)) bad = 1;
        const double inv = ik_rcp_nr(dj);
        if (lane == j) myinv = inv;
        const double l = (lane > j) ? c[j] * inv : 0.0;  // l_ij
#pragma unroll
        for (int k = j + 1; k < 13; k++) {
            const double u = ik_bcast_lane(c[k], j);     // a_jk = a_kj (k = 12: the right-hand side of row j)
            c[k] = fma(-l, u, c[k]);
        }
    }
    // back substitution L^T y = D^-1 z: lane j holds z_j (c[12]) and column j of d_j L (c[k], k > j)
    double acc = c[12] * myinv;
#pragma unroll
    for (int i = 11; i >= 0; i--) {
        const double yi = ik_bcast_lane(acc, i);
        y[i] = yi;
        const double lt = (lane < i) ? c[i] * myinv : 0.0;   // l_i,lane
        acc = fma(-lt, yi, acc);
    }
    return bad;
}

// once per launch: state, propagated covariance, limits, counters
__device__ __forceinline__ void ikfom_stage_once(const FlDev23 *__restrict__ D, FlIkLds &L)
{
    const int tid = threadIdx.x, NTH = blockDim.x, n = FL_N23;
    if (tid < FL_X23_LEN) { L.x[tid] = D->x[tid]; L.xp[tid] = D->xprop[tid]; }
    for (int e = tid; e < n * n; e += NTH) L.Pp[e] = D->Pprop[e];
    if (tid >= 64 && tid < 64 + 23) L.limit[tid - 64] = D->limit[tid - 64];
    if (tid >= 128 && tid < 128 + 23) {          // the weight table's constant part (identity rows) and every row's first state
        const int r = tid - 128;
        int rb, rs, rw;
        ik_blk(r, rb, rs, rw);
        L.jb[r] = rb;
        if (rw < 0) { L.Jw[r][0] = 1.0; L.Jw[r][1] = 0.0; L.Jw[r][2] = 0.0; }
        else if (rs == 2) L.Jw[r][2] = 0.0;
    }
    if (tid == 96) { L.cnt[0] = D->t_count; L.cnt[1] = D->iter_i; L.cnt[2] = D->max_iter; L.cnt[3] = D->iters_run; L.sticky = D->status; L.R = D->meas_cov; }
    __syncthreads();
}

// before the records arrive: dx, Jacobians, projected covariance, dx_new, A12. All threads; ends with a barrier.
__device__ __forceinline__ void ikfom_pre(FlIkLds &L)
{
    const int tid = threadIdx.x, NTH = blockDim.x, n = FL_N23;
    // dx = x (-) x_prop and the projection Jacobians at dx. The three manifold segments (SO3 rot, SO3 offset_R, S2 grav -- each a Log
    // with acos/atan2 and a Jacobian with sin/cos) are independent: one lane of three different waves each
    if (tid == 192) {
        double q[4] = {L.x[FL_X23_ROT], L.x[FL_X23_ROT + 1], L.x[FL_X23_ROT + 2], L.x[FL_X23_ROT + 3]}, Rm[9];
        flq_to_R(q, Rm);
        for (int k = 0; k < 9; k++) L.Rm[k] = Rm[k];
    }
    if (tid == 0 || tid == 64 || tid == 128) {
        const double *x = L.x, *o = L.xp;
        double oc[4], r[4], d3[3];
        if (tid == 0) {
            for (int i = 0; i < 3; i++) { const double v = x[FL_X23_POS + i] - o[FL_X23_POS + i]; L.dx[i] = v; L.dxn[i] = v; }
            oc[0] = -o[FL_X23_ROT]; oc[1] = -o[FL_X23_ROT + 1]; oc[2] = -o[FL_X23_ROT + 2]; oc[3] = o[FL_X23_ROT + 3];
            flq_mul(oc, x + FL_X23_ROT, r);
            fl_so3_log(r, d3);
            for (int i = 0; i < 3; i++) { L.dx[3 + i] = d3[i]; L.dxn[3 + i] = d3[i]; }
            fl_ikfom_J_so3(d3, L.J[0]);
            for (int i = 0; i < 9; i++) L.Jw[3 + i / 3][i % 3] = L.J[0][i];
        } else if (tid == 64) {
            oc[0] = -o[FL_X23_ORLI]; oc[1] = -o[FL_X23_ORLI + 1]; oc[2] = -o[FL_X23_ORLI + 2]; oc[3] = o[FL_X23_ORLI + 3];
            flq_mul(oc, x + FL_X23_ORLI, r);
            fl_so3_log(r, d3);
            for (int i = 0; i < 3; i++) {
                L.dx[6 + i] = d3[i]; L.dxn[6 + i] = d3[i];
                const double v = x[FL_X23_OTLI + i] - o[FL_X23_OTLI + i]; L.dx[9 + i] = v; L.dxn[9 + i] = v;
            }
            fl_ikfom_J_so3(d3, L.J[1]);
            for (int i = 0; i < 9; i++) L.Jw[6 + i / 3][i % 3] = L.J[1][i];
        } else {
            for (int i = 0; i < 3; i++) {
                const double v = x[FL_X23_VEL + i] - o[FL_X23_VEL + i], g = x[FL_X23_BG + i] - o[FL_X23_BG + i], a = x[FL_X23_BA + i] - o[FL_X23_BA + i];
                L.dx[12 + i] = v; L.dxn[12 + i] = v; L.dx[15 + i] = g; L.dxn[15 + i] = g; L.dx[18 + i] = a; L.dxn[18 + i] = a;
            }
            double d2[2];
            fl_s2_boxminus(x + FL_X23_GRAV, o + FL_X23_GRAV, d2);
            L.dx[21] = d2[0]; L.dx[22] = d2[1]; L.dxn[21] = d2[0]; L.dxn[22] = d2[1];
            double seg[23];
            seg[21] = d2[0]; seg[22] = d2[1];
            fl_ikfom_J_s2(L.x + FL_X23_GRAV, L.xp + FL_X23_GRAV, seg + 21, L.J[2]);
            L.Jw[21][0] = L.J[2][0]; L.Jw[21][1] = L.J[2][1]; L.Jw[22][0] = L.J[2][2]; L.Jw[22][1] = L.J[2][3];
        }
    }
    __syncthreads();
#ifdef FL_IK_STAMPS
    if (tid == 0) g_fl_stamps[44] = (long long)wall_clock64();
#endif
    // P = Jf Pprop Jf^T ; dx_new = Jf dx ; A12 = sym(P[0:12,0:12]) / R -- every element a branch-free 9-term sum over the weight table
    // (round 3: the generic block loops with their run-time trip counts made this phase 3 us of a 5 us ikfom_pre, LONGER than the
    // producers' pass and therefore on the critical path)
    auto pelem = [&](int r, int c) -> double {
        const int rb = L.jb[r], cb = L.jb[c];
        double s2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int ra = (rb + a < n) ? rb + a : n - 1;
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const int cbb = (cb + b < n) ? cb + b : n - 1;
                t += L.Pp[ra * n + cbb] * L.Jw[c][b];
            }
            s2 += L.Jw[r][a] * t;
        }
        return s2;
    };
    for (int e = tid; e < n * n; e += NTH) L.P[e] = pelem(e / n, e % n);
#ifdef FL_IK_STAMPS
    if (tid == 0) g_fl_stamps[45] = (long long)wall_clock64();
#endif
    if (tid < n) {
        const int rb = L.jb[tid];
        double s2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++) s2 += L.Jw[tid][a] * L.dx[(rb + a < n) ? rb + a : n - 1];
        L.dxn[tid] = s2;
    }
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        L.A12[tid] = 0.5 * (pelem(i, j) + pelem(j, i)) / L.R;
    }
    __syncthreads();
}

// state words of the multi-pass broadcast: 26 doubles + control = 53 words (handoff.h)
#define FL_IK_BCAST_DOUBLES FL_X23_LEN

// after the gather. All threads call it; on return (after the caller's barrier) L.ctl is valid for everybody.
// bcast != nullptr (multi-pass kernel): the new state and the control word are published for the producers' next pass.
// INTERNAL: s_sums is the 64-double record of the pass kernels (the C block is rebuilt here); else the public 96-double record.
template <bool INTERNAL = true>
__device__ __forceinline__ void ikfom_post(FlDev23 *__restrict__ D, const double *s_sums, FlIkLds &L, int gst,
                                           unsigned long long *bcast = nullptr, unsigned bepoch = 0u, bool write_P = true)
{
    const int tid = threadIdx.x, NTH = blockDim.x, n = FL_N23;
    const double R = L.R;
    if (gst) {            // hand-off timed out: abandon the pass (see solve18.h)
        if (tid < FL_IK_BCAST_DOUBLES && bcast) fl_bcast_store(bcast, tid, L.x[tid], bepoch);
        if (tid == 64) {
            L.sticky |= FL_NUM_TIMEOUT;
            D->status = L.sticky;
            L.ctl[2] = 0; L.ctl[4] = 1; L.ctl[5] = 1;
            if (bcast) fl_bcast_ctrl_at(bcast, 2 * FL_IK_BCAST_DOUBLES, 1 | 4, bepoch);
            else D->resume_count = 1;                 // one launch per pass: THIS pass is still to do (the multi-pass loop writes count - p)
        }
        return;
    }
    if (tid >= 144) {                   // (same quotients, same products as before: bit-identical)
        for (int e = tid - 144; e < n * 12; e += NTH - 144) L.Pd[e] = L.P[(e / 12) * n + (e % 12)] / R;
    }
    if (INTERNAL) {                     // S = h_x^T h_x (12 x 12), h_x^T h and the scalars out of the internal record
        if (tid < 144) L.S[tid] = fl_s12_from_s9(s_sums, L.Rm, tid / 12, tid % 12);
        else if (tid >= 160 && tid < 172) L.htz[tid - 160] = fl_htz12_from_s9(s_sums, L.Rm, tid - 160);
        else if (tid >= 192 && tid < 195) L.scal[tid - 192] = s_sums[FL_S23I_NEFF + (tid - 192)];
    } else {
        if (tid >= 64 && tid < 64 + 78) {   // unpack S (upper triangle, row-major)
            int k = tid - 64, i = 0, rowlen = 12;
            while (k >= rowlen) { k -= rowlen; rowlen--; i++; }
            const int j = i + k;
            const double v = s_sums[tid - 64];
            L.S[i * 12 + j] = v;
            L.S[j * 12 + i] = v;
        } else if (tid >= 160 && tid < 172) L.htz[tid - 160] = s_sums[FL_S23_HTZ + (tid - 160)];
        else if (tid >= 192 && tid < 195) L.scal[tid - 192] = s_sums[FL_S23_NEFF + (tid - 192)];
    }
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) s += L.S[i * 12 + k] * L.A12[k * 12 + j];
        L.SA[tid] = s;
    } else if (tid >= 160 && tid < 172) {          // rhs = HTh + S dx_new12
        const int i = tid - 160;
        double s = L.htz[i];
#pragma unroll
        for (int k = 0; k < 12; k++) s += L.S[i * 12 + k] * L.dxn[k];
        L.rhs[i] = s;
    }
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        double s = L.A12[tid];
#pragma unroll
        for (int k = 0; k < 12; k++) s += L.A12[i * 12 + k] * L.SA[k * 12 + j];
        L.X[tid] = s;                       // unsymmetrised M, staged in X
    } else if (tid >= 160 && tid < 172) {   // y0 = A12 rhs
        const int i = tid - 160;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) s += L.A12[i * 12 + k] * L.rhs[k];
        L.y0[i] = s;
    }
    __syncthreads();
#ifdef FL_IK_STAMPS
    if (tid == 0) g_fl_stamps[36] = (long long)wall_clock64();
#endif
    // ---- wavefront 0: M (symmetrised on the way in) and y0 into registers, right-looking LDL^T with y0 as a 13th row, back
    // substitution; lane r < 23 then forms dx_[r].  No barrier, no cross-lane traffic until the results are written.
    int bad = 0;
    if (tid < 64) {
        double y[12];
        bad = ik_ldl_solve_rows(L.X, L.y0, y);
        if (tid < n) {                          // dx_ = A[:,0:12] y - dx_new
            double s2 = 0.0;
#pragma unroll
            for (int cc = 0; cc < 12; cc++) s2 += L.Pd[tid * 12 + cc] * y[cc];
            L.dxo[tid] = s2 - L.dxn[tid];
        }
    }
    bad = __syncthreads_or(bad);
#ifdef FL_IK_STAMPS
    if (tid == 0) g_fl_stamps[37] = (long long)wall_clock64();
#endif
    // ---- boxplus on three lanes of three waves (SO3 rot | SO3 offset_R | additive + S2), judgement on a fourth
    if (tid == 0 || tid == 64 || tid == 128) {
        double e[4], q[4];
        if (tid == 0) {
            for (int i = 0; i < 3; i++) { const double v = L.x[FL_X23_POS + i] + L.dxo[i]; L.x[FL_X23_POS + i] = v; }
            fl_mtk_exp3(L.dxo + 3, 0.5, e);
            for (int i = 0; i < 4; i++) q[i] = L.x[FL_X23_ROT + i];
            flq_mul(q, e, q);
            for (int i = 0; i < 4; i++) L.x[FL_X23_ROT + i] = q[i];
        } else if (tid == 64) {
            fl_mtk_exp3(L.dxo + 6, 0.5, e);
            for (int i = 0; i < 4; i++) q[i] = L.x[FL_X23_ORLI + i];
            flq_mul(q, e, q);
            for (int i = 0; i < 4; i++) L.x[FL_X23_ORLI + i] = q[i];
            for (int i = 0; i < 3; i++) { const double v = L.x[FL_X23_OTLI + i] + L.dxo[9 + i]; L.x[FL_X23_OTLI + i] = v; }
        } else {
            for (int i = 0; i < 3; i++) {
                const double v = L.x[FL_X23_VEL + i] + L.dxo[12 + i], g = L.x[FL_X23_BG + i] + L.dxo[15 + i], a = L.x[FL_X23_BA + i] + L.dxo[18 + i];
                L.x[FL_X23_VEL + i] = v; L.x[FL_X23_BG + i] = g; L.x[FL_X23_BA + i] = a;
            }
            double gv[3] = {L.x[FL_X23_GRAV], L.x[FL_X23_GRAV + 1], L.x[FL_X23_GRAV + 2]};
            fl_s2_boxplus(gv, L.dxo + 21);
            for (int i = 0; i < 3; i++) L.x[FL_X23_GRAV + i] = gv[i];
        }
    } else if (tid >= 192) {                 // wavefront 3 judges: lane i < 23 looks at dx_[i], lane 0 decides
        const int ln = tid - 192;
        const double di = (ln < n) ? L.dxo[ln] : 0.0;
        const bool over = (ln < n) && (fabs(di) > L.limit[ln < n ? ln : 0]);
        const bool nonfin = (ln < n) && !(fabs(di) <= DBL_MAX);
        const unsigned long long m_over = __ballot(over), m_nf = __ballot(nonfin);
        if (ln == 0) {
        int converge = (m_over == 0ull) ? 1 : 0, st = bad;
        if (m_nf != 0ull) st |= 2;
        int t = L.cnt[0];
        const int i_loop = L.cnt[1], max_iter = L.cnt[2];
        if (converge) t++;
        if (!t && i_loop == max_iter - 2) converge = 1;
        const int finishing = (t > 1 || i_loop == max_iter - 1) ? 1 : 0;
        const int stop = (finishing || (i_loop + 1) >= max_iter) ? 1 : 0;
        L.ctl[0] = t; L.ctl[1] = converge; L.ctl[2] = finishing; L.ctl[3] = st; L.ctl[4] = 0; L.ctl[5] = stop;
        L.cnt[0] = t; L.cnt[1] = i_loop + 1; L.cnt[3] = L.cnt[3] + 1;
        D->t_count = t;
        D->need_search = converge;
        D->converged = converge;
        D->iter_i = i_loop + 1;
        D->stop = stop;
        D->neff = (int)L.scal[0];
        D->total_residual = L.scal[1];
        L.sticky |= st;
        D->status = L.sticky;
        D->iters_run = L.cnt[3];
        }
    }
    if (tid >= 96 && tid < 96 + 23) D->solution[tid - 96] = L.dxo[tid - 96];
    if (tid >= 64 && tid < 64 + FL_SUMS23) {      // the public record for the host
        const int idx = tid - 64;
        double v = 0.0;
        if (idx < 78) { int k = idx, i = 0, rowlen = 12; while (k >= rowlen) { k -= rowlen; rowlen--; i++; } v = L.S[i * 12 + i + k]; }
        else if (idx < 90) v = L.htz[idx - 78];
        else if (idx < 93) v = L.scal[idx - 90];
        D->sums[idx] = v;
    }
    __syncthreads();
#ifdef FL_IK_STAMPS
    if (tid == 0) g_fl_stamps[38] = (long long)wall_clock64();
#endif
    // the new state: to the device block, and to the producers of the next pass
    if (tid < FL_X23_LEN) {
        const double v = L.x[tid];
        D->x[tid] = v;
        if (bcast) fl_bcast_store(bcast, tid, v, bepoch);
    }
    if (tid == 64 && bcast) fl_bcast_ctrl_at(bcast, 2 * FL_IK_BCAST_DOUBLES, (L.ctl[5] ? 1 : 0) | (L.ctl[1] ? 2 : 0), bepoch);
    if (!L.ctl[2]) {                        // not finishing: the projected P_ is what a reader of the device block finds
        if (write_P)
            for (int e = tid; e < n * n; e += NTH) D->P[e] = L.P[e];
        return;
    }
    // ---- final covariance block, esekfom.hpp:1831-1924
    ik_make_J(L, L.dxo, tid);               // Jacobians at dx_ (S2: Nx at the UPDATED state, L.x)
    if (tid < 64) {                         // column c of M^-1 (A12 S): lane c < 12 of wavefront 0, the register solve again
        const int c = tid < 12 ? tid : 0;
        double col[12], sol[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 12; k++) s2 += L.A12[i * 12 + k] * L.S[k * 12 + c];
            col[i] = s2;
        }
        (void)ik_ldl_solve_regs(L.X, col, sol);
        if (tid < 12) {
#pragma unroll
            for (int i = 0; i < 12; i++) L.M[i * 12 + c] = sol[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < n * 12; e += NTH) {   // Kx = A[:,0:12] X
        const int r = e / 12, c = e % 12;
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += L.Pd[r * 12 + k] * L.M[k * 12 + c];
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

// one whole pass for a single launch (per-pass kernel, solve kernel): stage + pre + post
__device__ __forceinline__ void ikfom_solver_block(FlDev23 *__restrict__ D, const double *s_sums, FlIkLds &L, int gst)
{
    ikfom_post<true>(D, s_sums, L, gst);
}
