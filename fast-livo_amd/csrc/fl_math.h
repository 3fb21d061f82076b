// fl_math.h -- per-measurement math of the ESKF kernels (plane fit, point-to-plane row, photometric
// row, 18-state gain solve).  Pure functions on registers/arrays, no memory traffic: the kernels in
// fastlivo_hip.hip do the loading, reduction and publication around them.
//
// FL_HD is __host__ __device__ under hipcc; tests/host_emul compiles the same header with g++ to
// unit-test the arithmetic without a GPU (never part of libfastlivo_hip.so).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FL_HD __host__ __device__ __forceinline__
#else
#define FL_HD static inline
#endif
#include <float.h>
#include <math.h>
#include <stdint.h>

// FMA contraction is OFF for everything that feeds a gate (world point, plane fit, pd2, s, bilinear
// taps: they must round like the reference's FMA-less x86-64 build) and switched ON, function by
// function, only for fp64 arithmetic whose results are compared by tolerance (Jacobian rows, their
// accumulation, the gain solve): half the instructions for the dominant fp64 work.
#if defined(__HIPCC__)
#define FL_FP_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define FL_FP_CONTRACT
#endif

// ------------------------------------------------------------------------------------------------
// esti_plane<float> (include/common_lib.h:448-493): least-squares plane through 5 neighbours by
// column-pivoted Householder QR (Eigen ColPivHouseholderQR, float), normal n = x/|x|, d = 1/|x|,
// rejected if any |n.p_j + d| > 0.1.  Same operation order as the CPU oracle so that the plane and
// therefore the gates are bit-identical; all indices static so everything stays in registers.
// ------------------------------------------------------------------------------------------------
FL_HD void fl_swapf(float &a, float &b) { float t = a; a = b; b = t; }

FL_HD void fl_qr_solve_5x3(const float *nb /*15: 5 x (x,y,z)*/, float *x /*3*/)
{
    float q[5][3];
    float hc[3], nu[3], nd[3];
    int tr[3];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) q[r][c] = nb[r * 3 + c];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 5; r++) s += q[r][k] * q[r][k];
        nd[k] = sqrtf(s);
        nu[k] = nd[k];
    }
    float mx = nu[0];
    if (nu[1] > mx) mx = nu[1];
    if (nu[2] > mx) mx = nu[2];
    const float th = mx * FLT_EPSILON;
    const float threshold_helper = (th * th) / 5.0f;
    const float downdate_thr = sqrtf(FLT_EPSILON);
    int np = 3;

#pragma unroll
    for (int k = 0; k < 3; k++) {
        int big = k;
        float bigv = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; j++)
            if (nu[j] > bigv) { bigv = nu[j]; big = j; }
        const float big_sq = bigv * bigv;
        if (np == 3 && big_sq < threshold_helper * (float)(5 - k)) np = k;
        tr[k] = big;
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            if (big == j) {
#pragma unroll
                for (int r = 0; r < 5; r++) fl_swapf(q[r][k], q[r][j]);
                fl_swapf(nu[k], nu[j]);
                fl_swapf(nd[k], nd[j]);
            }
        }
        const float c0 = q[k][k];
        float tail = 0.f;
#pragma unroll
        for (int r = k + 1; r < 5; r++) tail += q[r][k] * q[r][k];
        float tau, beta;
        if (tail <= FLT_MIN) {
            tau = 0.f; beta = c0;
#pragma unroll
            for (int r = k + 1; r < 5; r++) q[r][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int r = k + 1; r < 5; r++) q[r][k] = q[r][k] / den;
            tau = (beta - c0) / beta;
        }
        q[k][k] = beta;
        hc[k] = tau;
        if (tau != 0.f) {
#pragma unroll
            for (int j = k + 1; j < 3; j++) {
                float tmp = 0.f;
#pragma unroll
                for (int r = k + 1; r < 5; r++) tmp += q[r][k] * q[r][j];
                tmp += q[k][j];
                q[k][j] -= tau * tmp;
#pragma unroll
                for (int r = k + 1; r < 5; r++) q[r][j] -= (tau * q[r][k]) * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            if (nu[j] != 0.f) {
                float temp = fabsf(q[k][j]) / nu[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                const float ratio = nu[j] / nd[j];
                const float temp2 = temp * (ratio * ratio);
                if (temp2 <= downdate_thr) {
                    float s = 0.f;
#pragma unroll
                    for (int r = k + 1; r < 5; r++) s += q[r][j] * q[r][j];
                    nd[j] = sqrtf(s);
                    nu[j] = nd[j];
                } else {
                    nu[j] *= sqrtf(temp);
                }
            }
        }
    }

    float c[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
    x[0] = 0.f; x[1] = 0.f; x[2] = 0.f;
    if (np == 0) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k < np) {
            const float tau = hc[k];
            if (tau != 0.f) {
                float tmp = 0.f;
#pragma unroll
                for (int r = k + 1; r < 5; r++) tmp += q[r][k] * c[r];
                tmp += c[k];
                c[k] -= tau * tmp;
#pragma unroll
                for (int r = k + 1; r < 5; r++) c[r] -= (tau * q[r][k]) * tmp;
            }
        }
    }
#pragma unroll
    for (int i = 2; i >= 0; i--) {
        if (i < np) {
            c[i] = c[i] / q[i][i];
#pragma unroll
            for (int r = 0; r < i; r++) c[r] -= c[i] * q[r][i];
        }
    }
    // column permutation: perm = identity with transpositions (k, tr[k]) applied on the right
    int p0 = 0, p1 = 1, p2 = 2;
    {   // k = 0
        if (tr[0] == 1) { int t = p0; p0 = p1; p1 = t; }
        else if (tr[0] == 2) { int t = p0; p0 = p2; p2 = t; }
        // k = 1
        if (tr[1] == 2) { int t = p1; p1 = p2; p2 = t; }
    }
    const int perm[3] = {p0, p1, p2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i < np) {
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (perm[i] == j) x[j] = c[i];
        }
    }
}

// returns 1 if the plane passes the 0.1 planarity check; pabcd = (n, d)
FL_HD int fl_esti_plane(const float *nb, float *pabcd)
{
    float nv[3];
    fl_qr_solve_5x3(nb, nv);
    const float n = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    pabcd[0] = nv[0] / n;
    pabcd[1] = nv[1] / n;
    pabcd[2] = nv[2] / n;
    pabcd[3] = 1.0f / n;   // == (float)(1.0 / (double)n): division is innocuous under double rounding
    int ok = 1;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const float v = pabcd[0] * nb[j * 3 + 0] + pabcd[1] * nb[j * 3 + 1] + pabcd[2] * nb[j * 3 + 2] + pabcd[3];
        if (fabsf(v) > 0.1f) ok = 0;
    }
    return ok;
}

// ------------------------------------------------------------------------------------------------
// Point-to-plane residual and gates (src/laserMapping.cpp:1527,1573-1584,1593).
//   p_i  = R_LI p_b + t_LI ; p_w = (float)(R p_i + p)           (pointBodyToWorld, :272-286)
//   pd2  = n.p_w + d (float) ; s = 1 - 0.9 |pd2| / sqrt(|p_b|)  ; keep iff s > 0.9
//   effective iff also |pd2| <= 2.0
// Returns the new point_selected_surf flag; *eff = contributes to the normal equations.
// ------------------------------------------------------------------------------------------------
FL_HD int fl_point_gates(const float *pb, const float *pl /*plane n,d*/, const double *R /*9*/, const double *p /*3*/,
                         const double *R_LI, const double *t_LI, double *p_i /*3 out*/, float *pw /*3 out*/,
                         float *pd2_out, int *eff)
{
    const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
    p_i[0] = (R_LI[0] * b0 + R_LI[1] * b1 + R_LI[2] * b2) + t_LI[0];
    p_i[1] = (R_LI[3] * b0 + R_LI[4] * b1 + R_LI[5] * b2) + t_LI[1];
    p_i[2] = (R_LI[6] * b0 + R_LI[7] * b1 + R_LI[8] * b2) + t_LI[2];
    const double g0 = R[0] * p_i[0] + R[1] * p_i[1] + R[2] * p_i[2];
    const double g1 = R[3] * p_i[0] + R[4] * p_i[1] + R[5] * p_i[2];
    const double g2 = R[6] * p_i[0] + R[7] * p_i[1] + R[8] * p_i[2];
    pw[0] = (float)(g0 + p[0]);
    pw[1] = (float)(g1 + p[1]);
    pw[2] = (float)(g2 + p[2]);
    const float pd2 = pl[0] * pw[0] + pl[1] * pw[1] + pl[2] * pw[2] + pl[3];
    const double pbn = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
    const float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(pbn));
    *pd2_out = pd2;
    const int sel = ((double)s > 0.9) ? 1 : 0;
    *eff = (sel && ((double)fabsf(pd2) <= 2.0)) ? 1 : 0;
    return sel;
}

// ------------------------------------------------------------------------------------------------
// The selection gate as a per-point threshold.  `s > 0.9` with s = (float)(1 - 0.9*|pd2| / sqrt(|p_b|))
// (laserMapping.cpp:1574-1576) costs two fp64 square roots and an fp64 division per point and pass, although sqrt(|p_b|)
// depends on the body point only and the expression is monotone in |pd2|: 0.9*a rounds monotonically, so do the division
// by a positive number, 1 - x, and the cast to float.  Hence {a >= 0 : gate(a)} is a down-set [0, T] of floats, and
//     gate(|pd2|)  <=>  |pd2| <= T          (NaN fails both; T = -1 when even a = 0 fails: |p_b| = 0 or non-finite)
// with T found ONCE per point (per frame) by evaluating the reference's own expression: bracket around q/9 and bisection on
// the float's bit pattern.  The per-pass test is one float compare -- bit-identical selections by construction
// (tests/test_lio18_gpu.py: 0 flips), ~40 fp64 instructions fewer per point and pass.
// ------------------------------------------------------------------------------------------------
FL_HD int fl_gate_expr(float a, double q)     // the reference's expression, verbatim
{
    const float s = (float)(1 - 0.9 * fabs((double)a) / q);
    return ((double)s > 0.9) ? 1 : 0;
}
FL_HD float fl_bits_float(unsigned u) { union { unsigned u; float f; } c; c.u = u; return c.f; }
FL_HD unsigned fl_float_bits(float f) { union { unsigned u; float f; } c; c.f = f; return c.u; }
FL_HD float fl_gate_threshold(const float *pb)
{
    const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
    const double pbn = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
    const double q = sqrt(pbn);
    if (!fl_gate_expr(0.0f, q)) return -1.0f;
    const unsigned top = 0x7F7FFFFFu;                       // FLT_MAX
    if (fl_gate_expr(fl_bits_float(top), q)) return fl_bits_float(top);
    unsigned lo = 0u, hi = top;                             // gate(lo) true, gate(hi) false
    {   // narrow the bracket around the analytic crossing a ~ q / 9 (a few ulps off at most)
        const float est = (float)(q * (1.0 / 9.0));
        const unsigned e = fl_float_bits(est);
        if (e > 64u && e < top - 64u) {
            if (fl_gate_expr(fl_bits_float(e - 64u), q)) lo = e - 64u;
            if (!fl_gate_expr(fl_bits_float(e + 64u), q)) hi = e + 64u;
        }
    }
    while (hi - lo > 1u) {
        const unsigned mid = lo + (hi - lo) / 2u;
        if (fl_gate_expr(fl_bits_float(mid), q)) lo = mid; else hi = mid;
    }
    return fl_bits_float(lo);
}
// fl_point_gates with the gate as a threshold (gate_T = fl_gate_threshold(pb)): same outputs, bit for bit.
FL_HD int fl_point_gates_T(const float *pb, float gate_T, const float *pl /*plane n,d*/, const double *R /*9*/, const double *p /*3*/,
                           const double *R_LI, const double *t_LI, double *p_i /*3 out*/, float *pw /*3 out*/, float *pd2_out, int *eff)
{
    const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
    p_i[0] = (R_LI[0] * b0 + R_LI[1] * b1 + R_LI[2] * b2) + t_LI[0];
    p_i[1] = (R_LI[3] * b0 + R_LI[4] * b1 + R_LI[5] * b2) + t_LI[1];
    p_i[2] = (R_LI[6] * b0 + R_LI[7] * b1 + R_LI[8] * b2) + t_LI[2];
    const double g0 = R[0] * p_i[0] + R[1] * p_i[1] + R[2] * p_i[2];
    const double g1 = R[3] * p_i[0] + R[4] * p_i[1] + R[5] * p_i[2];
    const double g2 = R[6] * p_i[0] + R[7] * p_i[1] + R[8] * p_i[2];
    pw[0] = (float)(g0 + p[0]);
    pw[1] = (float)(g1 + p[1]);
    pw[2] = (float)(g2 + p[2]);
    const float pd2 = pl[0] * pw[0] + pl[1] * pw[1] + pl[2] * pw[2] + pl[3];
    *pd2_out = pd2;
    const float a = fabsf(pd2);
    const int sel = (a <= gate_T) ? 1 : 0;
    *eff = (sel && (a <= 2.0f)) ? 1 : 0;
    return sel;
}

// Mode-18 Jacobian row (src/laserMapping.cpp:1611-1629): row = [ [p_i]x R^T n , n ], z = -pd2.
FL_HD void fl_row18(const double *p_i, const float *pl, float pd2, const double *R, double *row /*6*/, double *z)
{
    // explicit fma(): a fixed operation tree (half the instructions of mul+add), so every kernel that inlines this -- the
    // per-pass, multi-pass and accumulate-only forms -- produces bit-identical partial sums whatever the surrounding code is
    // (with `fp contract(fast)` the compiler's choice of what to fuse depended on the inlining context)
    const double n0 = (double)pl[0], n1 = (double)pl[1], n2 = (double)pl[2];
    const double c0 = fma(R[6], n2, fma(R[3], n1, R[0] * n0));   // C = R^T n
    const double c1 = fma(R[7], n2, fma(R[4], n1, R[1] * n0));
    const double c2 = fma(R[8], n2, fma(R[5], n1, R[2] * n0));
    row[0] = fma(p_i[1], c2, -(p_i[2] * c1));                    // A = p_i x C
    row[1] = fma(p_i[2], c0, -(p_i[0] * c2));
    row[2] = fma(p_i[0], c1, -(p_i[1] * c0));
    row[3] = n0; row[4] = n1; row[5] = n2;
    *z = -(double)pd2;
}

// Accumulate one measurement row into a reduction record (layout in fl_device.h).
FL_HD void fl_accum6(double *v /*32*/, const double *row, double z)
{
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { v[k] = fma(row[i], row[j], v[k]); k++; }
#pragma unroll
    for (int i = 0; i < 6; i++) v[21 + i] = fma(row[i], z, v[21 + i]);
}

// ------------------------------------------------------------------------------------------------
// 6x6 (or NxN, N<=12) linear solves with partial pivoting on register arrays.
// ------------------------------------------------------------------------------------------------
// Solves M X = B for NR right-hand sides in place (B -> X). M is destroyed. Returns 1 if singular.
template <int N, int NR>
FL_HD int fl_gauss_solve(double (&M)[N][N], double (&B)[N][NR])
{
    int singular = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        int p = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int r = k + 1; r < N; r++) {
            const double a = fabs(M[r][k]);
            if (a > best) { best = a; p = r; }
        }
        if (best == 0.0) singular = 1;
#pragma unroll
        for (int r = k + 1; r < N; r++) {
            if (p == r) {
#pragma unroll
                for (int c = 0; c < N; c++) { double t = M[k][c]; M[k][c] = M[r][c]; M[r][c] = t; }
#pragma unroll
                for (int c = 0; c < NR; c++) { double t = B[k][c]; B[k][c] = B[r][c]; B[r][c] = t; }
            }
        }
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int r = k + 1; r < N; r++) {
            const double l = M[r][k] * inv;
#pragma unroll
            for (int c = k + 1; c < N; c++) M[r][c] -= l * M[k][c];
#pragma unroll
            for (int c = 0; c < NR; c++) B[r][c] -= l * B[k][c];
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        const double inv = 1.0 / M[i][i];
#pragma unroll
        for (int c = 0; c < NR; c++) {
            double s = B[i][c];
#pragma unroll
            for (int j = i + 1; j < N; j++) s -= M[i][j] * B[j][c];
            B[i][c] = s * inv;
        }
    }
    return singular;
}

// ------------------------------------------------------------------------------------------------
// 18-state gain solve + state update (src/laserMapping.cpp:1664-1695, lidar_selection.cpp:871-886).
// The reference forms K_1 = (H^T H + (P/R)^-1)^-1 with two dense 18x18 inverses and uses only
// K_1[:,0:6]. With A = P/R, S = H^T H (6x6) and E = [I6;0]:  K_1 E = A E (I6 + S A66)^-1, so the
// same delta needs one 6x6 solve:
//     W = (I + S A66)^-1,   y = W (sign*HTz - S vec6),   delta = A[:,0:6] y + vec,
//     G[:,0:6] = A[:,0:6] W S.
// (algebraically identical; agreement with the two-inverse CPU oracle is asserted in tests/).
// x: 24 doubles rot(9) pos vel bg ba grav.  sums: reduction record.  Returns FL_NUM_* status bits.
// ------------------------------------------------------------------------------------------------
FL_HD void fl_state18_minus(const double *a /*x_prop*/, const double *b /*x*/, double *out /*18*/)
{
    // Log(b.rot^T a.rot), common_lib.h:354-365
    double rd[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) rd[i * 3 + j] = b[0 * 3 + i] * a[0 * 3 + j] + b[1 * 3 + i] * a[1 * 3 + j] + b[2 * 3 + i] * a[2 * 3 + j];
    const double tr = rd[0] + rd[4] + rd[8];
    const double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
    const double f = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
    out[0] = f * (rd[7] - rd[5]);
    out[1] = f * (rd[2] - rd[6]);
    out[2] = f * (rd[3] - rd[1]);
#pragma unroll
    for (int i = 0; i < 15; i++) out[3 + i] = a[9 + i] - b[9 + i];
}

FL_HD void fl_state18_plus(double *x, const double *d /*18*/)
{
    // rot <- rot * Exp(d0,d1,d2), others add (common_lib.h:343-352, so3_math.h:54-72)
    const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nrm > 0.00001) {
        const double r0 = d[0] / nrm, r1 = d[1] / nrm, r2 = d[2] / nrm;
        const double K[9] = {0.0, -r2, r1, r2, 0.0, -r0, -r1, r0, 0.0};
        double cK[9], cKK[9], E[9], Rn[9];
        const double s = sin(nrm), c = 1.0 - cos(nrm);
        // the reference's `(1.0 - cos) * K * K` groups as ((1 - cos) K) K (so3_math.h:66)
#pragma unroll
        for (int i = 0; i < 9; i++) cK[i] = c * K[i];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) cKK[i * 3 + j] = cK[i * 3 + 0] * K[0 * 3 + j] + cK[i * 3 + 1] * K[1 * 3 + j] + cK[i * 3 + 2] * K[2 * 3 + j];
#pragma unroll
        for (int i = 0; i < 9; i++) E[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + cKK[i];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Rn[i * 3 + j] = x[i * 3 + 0] * E[0 * 3 + j] + x[i * 3 + 1] * E[1 * 3 + j] + x[i * 3 + 2] * E[2 * 3 + j];
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] = Rn[i];
    }
#pragma unroll
    for (int i = 0; i < 15; i++) x[9 + i] += d[3 + i];
}

// Serial reference form of the solve (one thread). G6: 18x6 row-major out.
FL_HD int fl_solve18_serial(double *x, const double *xprop, const double *P, double meas_cov, const double *sums,
                            double sign, double *G6, double *delta)
{
    double S[6][6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) { S[i][j] = sums[k]; S[j][i] = sums[k]; k++; }
    }
    double vec[18];
    fl_state18_minus(xprop, x, vec);
    double M[6][6], B[6][7];
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += S[i][k] * (P[k * 18 + j] / meas_cov);
            M[i][j] = s;
            B[i][j] = (i == j) ? 1.0 : 0.0;
        }
        double b = sign * sums[21 + i];
#pragma unroll
        for (int k = 0; k < 6; k++) b -= S[i][k] * vec[k];
        B[i][6] = b;
    }
    int st = fl_gauss_solve<6, 7>(M, B) ? 1 : 0;   // B[:,0:6] = W, B[:,6] = y
    double WS[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += B[i][k] * S[k][j];
            WS[i][j] = s;
        }
    for (int r = 0; r < 18; r++) {
        double a[6];
#pragma unroll
        for (int c = 0; c < 6; c++) a[c] = P[r * 18 + c] / meas_cov;
        double dl = vec[r];
#pragma unroll
        for (int c = 0; c < 6; c++) dl += a[c] * B[c][6];
        delta[r] = dl;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += a[k] * WS[k][c];
            G6[r * 6 + c] = s;
        }
    }
    fl_state18_plus(x, delta);
    for (int r = 0; r < 18; r++)
        if (!(fabs(delta[r]) <= DBL_MAX)) st |= 2;
    return st;
}


// ------------------------------------------------------------------------------------------------
// Fast per-iteration form of the same gain solve.  P (hence A = P/R) is constant while a frame
// iterates (the reference only rewrites state.cov after the loop, laserMapping.cpp:1715;
// lidar_selection.cpp:980), so the part that depends on P alone is hoisted to a per-frame prepare:
//     Q = A66^-1 (6x6, SPD)            T = A[:,0:6] Q   (18x6; its top block is I)
// and with (I + S A66)^-1 = Q^-1 ... : W = (I + S A66)^-1 = A66^-1 (A66^-1 + S)^-1 = Q (Q+S)^-1, i.e.
//     A[:,0:6] W = T' with  K_1[:,0:6] = A[:,0:6] Q (Q+S)^-1 = T (Q+S)^-1.
// Per iteration only the SPD 6x6 system (Q + S) z = sign*HTz - S vec6 is factorised (LDL^T, no
// pivoting needed for SPD) and  delta = T z + vec.   G[:,0:6] = T (Q+S)^-1 S is needed only for the
// covariance update after the last pass and is formed there (fl_gain18).
// ------------------------------------------------------------------------------------------------
FL_HD int fl_prepare18(const double *P, double meas_cov, double *Q /*36*/, double *T /*108*/)
{
    double M[6][6], B[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            M[i][j] = 0.5 * (P[i * 18 + j] + P[j * 18 + i]) / meas_cov;   // symmetrised A66
            B[i][j] = (i == j) ? 1.0 : 0.0;
        }
    const int st = fl_gauss_solve<6, 6>(M, B);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) Q[i * 6 + j] = 0.5 * (B[i][j] + B[j][i]);
    for (int r = 0; r < 18; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += (P[r * 18 + k] / meas_cov) * Q[k * 6 + c];
            T[r * 6 + c] = s;
        }
    return st;
}

// LDL^T factorisation of the SPD 6x6 C (full storage, lower part used). Returns 1 if a pivot <= 0.
struct FlLdl6 {
    double L[6][6];
    double dinv[6];
};
FL_HD int fl_ldl6(const double (&C)[6][6], FlLdl6 &f)
{
    FL_FP_CONTRACT   // the solve is compared to the oracle by tolerance, never bitwise
    int bad = 0;
    double d[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double dj = C[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= f.L[j][k] * f.L[j][k] * d[k];
        d[j] = dj;
        if (!(dj > 0.0)) bad = 1;
        f.dinv[j] = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = C[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= f.L[i][k] * f.L[j][k] * d[k];
            f.L[i][j] = v * f.dinv[j];
        }
    }
    return bad;
}
FL_HD void fl_ldl6_solve(const FlLdl6 &f, double *b /*6, in: rhs, out: solution*/)
{
    FL_FP_CONTRACT
#pragma unroll
    for (int i = 1; i < 6; i++)
#pragma unroll
        for (int k = 0; k < i; k++) b[i] -= f.L[i][k] * b[k];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] *= f.dinv[i];
#pragma unroll
    for (int i = 4; i >= 0; i--)
#pragma unroll
        for (int k = i + 1; k < 6; k++) b[i] -= f.L[k][i] * b[k];
}

FL_HD void fl_unpack_S(const double *sums, double (&S)[6][6])
{
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { S[i][j] = sums[k]; S[j][i] = sums[k]; k++; }
}

// z = (Q+S)^-1 (sign*HTz - S vec6). Returns status bits (1 = not positive definite).
FL_HD int fl_solve18_z(const double *Q, const double *sums, const double *vec /*18*/, double sign, double *z /*6*/)
{
    double S[6][6], C[6][6];
    fl_unpack_S(sums, S);
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j < 6; j++) C[i][j] = Q[i * 6 + j] + S[i][j];
        double b = sign * sums[21 + i];
#pragma unroll
        for (int k = 0; k < 6; k++) b -= S[i][k] * vec[k];
        z[i] = b;
    }
    FlLdl6 f;
    const int bad = fl_ldl6(C, f);
    fl_ldl6_solve(f, z);
    return bad;
}

// Whole fast solve on one thread: delta = T z + vec ; x (+)= delta.
FL_HD int fl_solve18_fast(double *x, const double *xprop, const double *Q, const double *T, const double *sums, double sign,
                          double *delta)
{
    double vec[18], z[6];
    fl_state18_minus(xprop, x, vec);
    int st = fl_solve18_z(Q, sums, vec, sign, z);
    for (int r = 0; r < 18; r++) {
        double dl = vec[r];
#pragma unroll
        for (int c = 0; c < 6; c++) dl += T[r * 6 + c] * z[c];
        delta[r] = dl;
    }
    fl_state18_plus(x, delta);
    for (int r = 0; r < 18; r++)
        if (!(fabs(delta[r]) <= DBL_MAX)) st |= 2;
    return st;
}

// Column c of X = (Q+S)^-1 S -- what fl_gain18 does for its six columns one after the other, for the covariance kernels that give
// each column to a thread of its own (the same factorisation six times in parallel: identical values).
FL_HD int fl_gain18_column(const double *Q, const double *sums, int c, double *xcol /*6*/)
{
    double S[6][6], C[6][6];
    fl_unpack_S(sums, S);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) C[i][j] = Q[i * 6 + j] + S[i][j];
    FlLdl6 f;
    const int bad = fl_ldl6(C, f);
    double b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = S[i][0];
#pragma unroll
        for (int k = 1; k < 6; k++) v = (c == k) ? S[i][k] : v;      // (static indices: S stays in registers)
        b[i] = v;
    }
    fl_ldl6_solve(f, b);
#pragma unroll
    for (int i = 0; i < 6; i++) xcol[i] = b[i];
    return bad;
}

// G[:,0:6] = T (Q+S)^-1 S  (18x6), from the record of the last executed/accepted pass.
FL_HD int fl_gain18(const double *Q, const double *T, const double *sums, double *G6)
{
    double S[6][6], C[6][6];
    fl_unpack_S(sums, S);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) C[i][j] = Q[i * 6 + j] + S[i][j];
    FlLdl6 f;
    const int bad = fl_ldl6(C, f);
    double X[6][6];   // X = (Q+S)^-1 S, column by column
    for (int c = 0; c < 6; c++) {
        double b[6];
#pragma unroll
        for (int i = 0; i < 6; i++) b[i] = S[i][c];
        fl_ldl6_solve(f, b);
#pragma unroll
        for (int i = 0; i < 6; i++) X[i][c] = b[i];
    }
    for (int r = 0; r < 18; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += T[r * 6 + k] * X[k][c];
            G6[r * 6 + c] = s;
        }
    return bad;
}
