// fl_ikfom_math.h -- Mode-23 (IKFoM) arithmetic: state_ikfom manifold operations, the 1x12
// measurement row of h_share_model and one iteration of the iterated update, written once for the
// device (FL_HD) and unit-tested on the host through tests/host_emul.
//
// Reference (paths under the reference tree):
//   state_ikfom                 include/use-ikfom.hpp:12-21  (DOF 23: pos 0-2, rot 3-5, offset_R_L_I 6-8,
//                               offset_T_L_I 9-11, vel 12-14, bg 15-17, ba 18-20, grav(S2) 21-22)
//   h_share_model rows          src/laserMapping.cpp:1063-1089
//   iterated update             include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928
//   SO3 / S2 / vect manifolds   include/IKFoM_toolkit/mtk/types/SOn.hpp:233-297, S2.hpp:97-280, vect.hpp:117-122
//   A_matrix, exp, log          include/IKFoM_toolkit/mtk/src/mtkmath.hpp:142-174,236-288
//
// Gain: the reference forms P_inv = ((P_/R)^-1 + E HTH E^T)^-1 (two 23x23 inverses, or the N x N
// form when fewer than 23 rows) and uses only P_inv[:,0:12].  With A = P_/R (symmetric), S = HTH:
//     P_inv E = A E (I + S A12)^-1          and  A12 (I + S A12) = A12 + A12 S A12   (SPD)
// so   y = (A12 + A12 S A12)^-1 A12 (HTh + S dx_new[0:12]),   dx_ = A[:,0:12] y - dx_new
// and for the final covariance block  K_x[:,0:12] = A[:,0:12] (A12 + A12 S A12)^-1 A12 S.
// One SPD 12x12 factorisation per iteration, no pivoting; the same formula covers the reference's
// N<23 branch (esekfom.hpp:1712-1741), which is algebraically identical.
#pragma once

#include "fl_math.h"

#define FL_N23 23
#define FL_S2_LEN (98090.0 / 10000.0)   // use-ikfom.hpp:8
#define FL_MTK_TOL 1e-11                 // mtkmath.hpp:122

// state_ikfom as 26 doubles: pos(3) rot(4: x,y,z,w) offset_R_L_I(4) offset_T_L_I(3) vel(3) bg(3) ba(3) grav(3)
#define FL_X23_POS 0
#define FL_X23_ROT 3
#define FL_X23_ORLI 7
#define FL_X23_OTLI 11
#define FL_X23_VEL 14
#define FL_X23_BG 17
#define FL_X23_BA 20
#define FL_X23_GRAV 23
#define FL_X23_LEN 26

// ---- small-angle forms for the DEVICE (round 3). The increments of an update are tiny (|x (-) x_prop| and |dx_| << 0.5 rad), and the
// library sqrt / sin / cos / atan / atan2 behind the manifold operations ran on single lanes of the solver workgroup, on the pass's
// critical path (ikfom_pre 5 us, the three boxplus segments 1.6 us). Power series in the squared argument (truncation < 1e-17
// relative inside the stated range, no cancellation) replace them on the device; larger arguments and the host build (the unit
// tests against the oracle) keep the library form. The update is compared with the oracle by tolerance (1e-9), never bitwise.
#if defined(__HIP_DEVICE_COMPILE__) || defined(FL_IK_SERIES_ON_HOST)      /* FL_IK_SERIES_ON_HOST: tests/host_emul builds the SHIPPED forms too */
#define FL_IK_SERIES 1
#else
#define FL_IK_SERIES 0
#endif
// cos x and sin x / x from x^2 <= 0.25
FL_HD void fl_series_cos_sinc(double x2, double *co, double *si)
{
    double c = 1.0 / 87178291200.0;
    c = c * (-x2) + 1.0 / 479001600.0;
    c = c * (-x2) + 1.0 / 3628800.0;
    c = c * (-x2) + 1.0 / 40320.0;
    c = c * (-x2) + 1.0 / 720.0;
    c = c * (-x2) + 1.0 / 24.0;
    c = c * (-x2) + 0.5;
    *co = 1.0 - x2 * c;
    double a = 1.0 / 1307674368000.0;
    a = a * (-x2) + 1.0 / 6227020800.0;
    a = a * (-x2) + 1.0 / 39916800.0;
    a = a * (-x2) + 1.0 / 362880.0;
    a = a * (-x2) + 1.0 / 5040.0;
    a = a * (-x2) + 1.0 / 120.0;
    a = a * (-x2) + 1.0 / 6.0;
    *si = 1.0 - x2 * a;
}
// (1 - cos t) / t^2 and (1 - sin t / t) / t^2 from t^2 <= 0.25
FL_HD void fl_series_A_coeffs(double t2, double *a, double *b)
{
    double c = 1.0 / 87178291200.0;
    c = c * (-t2) + 1.0 / 479001600.0;
    c = c * (-t2) + 1.0 / 3628800.0;
    c = c * (-t2) + 1.0 / 40320.0;
    c = c * (-t2) + 1.0 / 720.0;
    c = c * (-t2) + 1.0 / 24.0;
    *a = c * (-t2) + 0.5;
    double d = 1.0 / 1307674368000.0;
    d = d * (-t2) + 1.0 / 6227020800.0;
    d = d * (-t2) + 1.0 / 39916800.0;
    d = d * (-t2) + 1.0 / 362880.0;
    d = d * (-t2) + 1.0 / 5040.0;
    d = d * (-t2) + 1.0 / 120.0;
    *b = d * (-t2) + 1.0 / 6.0;
}
// atan(r) / r from r^2 <= 1/16
FL_HD double fl_series_atan_over(double r2)
{
    double p = 1.0 / 29.0;
#pragma unroll
    for (int k = 13; k >= 0; k--) p = p * (-r2) + 1.0 / (2 * k + 1);
    return p;
}

// Mode-23 reduction record (FL_SUMS23 = 96 doubles): [0..77] upper triangle of the 12x12 h_x^T h_x
// (row-major, i<=j), [78..89] h_x^T h, [90] n_eff, [91] sum|pd2|, [92] sum pd2^2, [93..95] zero.
#define FL_S23_HTH 0
#define FL_S23_HTZ 78
#define FL_S23_NEFF 90
#define FL_S23_RES 91
#define FL_S23_RES2 92

// ---- Eigen::Quaternion helpers (coeff order x,y,z,w) ------------------------------------------
FL_HD void flq_mul(const double *a, const double *b, double *o)
{
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
// QuaternionBase::_transformVector: v + w*(2 q x v) + q x (2 q x v). conj != 0 rotates by q^-1.
FL_HD void flq_rot(const double *q, int conj, const double *v, double *o)
{
    const double sgn = conj ? -1.0 : 1.0;
    const double q0 = sgn * q[0], q1 = sgn * q[1], q2 = sgn * q[2], w = q[3];
    double u0 = q1 * v[2] - q2 * v[1], u1 = q2 * v[0] - q0 * v[2], u2 = q0 * v[1] - q1 * v[0];
    u0 += u0; u1 += u1; u2 += u2;
    const double c0 = q1 * u2 - q2 * u1, c1 = q2 * u0 - q0 * u2, c2 = q0 * u1 - q1 * u0;
    o[0] = v[0] + w * u0 + c0;
    o[1] = v[1] + w * u1 + c1;
    o[2] = v[2] + w * u2 + c2;
}
FL_HD void flq_to_R(const double *q, double *R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// mtkmath.hpp:142-174
FL_HD void fl_cos_sinc_sqrt(double x2, double *co, double *si)
{
    const double taylor_n_bound = 1.220703125e-4;      // sqrt(sqrt(DBL_EPSILON))
#if FL_IK_SERIES
    if (x2 >= taylor_n_bound && x2 <= 0.25) { fl_series_cos_sinc(x2, co, si); return; }
#endif
    if (x2 >= taylor_n_bound) {
        const double x = sqrt(x2);
        *co = cos(x); *si = sin(x) / x;
        return;
    }
    const double inv[7] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
    *co = cosi; *si = sinc;
}
// MTK::exp<scalar,3> (mtkmath.hpp:249-256): quaternion (vec, w) of a rotation vector at scale
FL_HD void fl_mtk_exp3(const double *vec, double scale, double *q /*x,y,z,w*/)
{
    const double norm2 = vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2];
    double c, s;
    fl_cos_sinc_sqrt(scale * scale * norm2, &c, &s);
    const double mult = s * scale;
    q[0] = mult * vec[0]; q[1] = mult * vec[1]; q[2] = mult * vec[2]; q[3] = c;
}
// SO3::log (SOn.hpp:293-297; mtkmath.hpp:268-288 with plus_minus_periodicity)
FL_HD void fl_so3_log(const double *q, double *res)
{
#if FL_IK_SERIES
    {
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
        if (n2 >= FL_MTK_TOL * FL_MTK_TOL && w > 0.0 && n2 <= 0.0625 * w * w) {       // 2 / nv * atan(nv / w) = 2 / w * (atan r / r), r = nv / w
            const double inv_w = 1.0 / w;
            const double s = 2.0 * inv_w * fl_series_atan_over(n2 * inv_w * inv_w);
            res[0] = s * q[0]; res[1] = s * q[1]; res[2] = s * q[2];
            return;
        }
    }
#endif
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < FL_MTK_TOL) nv = FL_MTK_TOL;
    const double s = 2.0 / nv * atan(nv / q[3]);
    res[0] = s * q[0]; res[1] = s * q[1]; res[2] = s * q[2];
}
FL_HD void fl_skew(const double *v, double *K)
{
    K[0] = 0.0;   K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2];  K[4] = 0.0;   K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0];  K[8] = 0.0;
}
FL_HD void fl_m3mul(const double *A, const double *B, double *C)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = T[i];
}
// A_matrix (mtkmath.hpp:236-247)
FL_HD void fl_A_matrix(const double *v, double *res)
{
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0) ? 1.0 : 0.0;
#if FL_IK_SERIES
    if (sq >= FL_MTK_TOL * FL_MTK_TOL && sq <= 0.25) {
        double K[9], bK[9], bKK[9], a, b;
        fl_skew(v, K);
        fl_series_A_coeffs(sq, &a, &b);
        for (int i = 0; i < 9; i++) bK[i] = b * K[i];      // `b * hat(v) * hat(v)` groups as (b K) K (mtkmath.hpp:244)
        fl_m3mul(bK, K, bKK);
        for (int i = 0; i < 9; i++) res[i] = res[i] + a * K[i] + bKK[i];
        return;
    }
#endif
    const double norm = sqrt(sq);
    if (!(norm < FL_MTK_TOL)) {
        double K[9], bK[9], bKK[9];
        fl_skew(v, K);
        const double a = (1 - cos(norm)) / sq, b = (1 - sin(norm) / norm) / sq;
        for (int i = 0; i < 9; i++) bK[i] = b * K[i];
        fl_m3mul(bK, K, bKK);
        for (int i = 0; i < 9; i++) res[i] = res[i] + a * K[i] + bKK[i];
    }
}
// S2_Bx, S2_typ == 1 (S2.hpp:215-231), 3x2 row-major
FL_HD void fl_s2_Bx(const double *vec, double *Bx)
{
    const double L = FL_S2_LEN;
#if FL_IK_SERIES
    if (vec[0] + L > FL_MTK_TOL) {       // device: one reciprocal (+ two Newton steps, ~1e-16) instead of nine IEEE divisions
        const double den = L + vec[0];
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(den);
#else
        double r = (double)(1.0f / (float)den);      // host build of the shipped form (tests): any 2^-20 seed, the two Newton steps below do the rest
#endif
        double e = fma(-den, r, 1.0);
        r = fma(r, e, r);
        e = fma(-den, r, 1.0);
        r = fma(r, e, r);
        const double iL = 1.0 / FL_S2_LEN;           // (a compile-time constant)
        Bx[0] = -vec[1] * iL;                        Bx[1] = -vec[2] * iL;
        Bx[2] = (L - vec[1] * vec[1] * r) * iL;      Bx[3] = (-vec[2] * vec[1] * r) * iL;
        Bx[4] = Bx[3];                               Bx[5] = (L - vec[2] * vec[2] * r) * iL;
        return;
    }
#endif
    if (vec[0] + L > FL_MTK_TOL) {
        Bx[0] = -vec[1];                              Bx[1] = -vec[2];
        Bx[2] = L - vec[1] * vec[1] / (L + vec[0]);   Bx[3] = -vec[2] * vec[1] / (L + vec[0]);
        Bx[4] = -vec[2] * vec[1] / (L + vec[0]);      Bx[5] = L - vec[2] * vec[2] / (L + vec[0]);
        for (int i = 0; i < 6; i++) Bx[i] /= L;
    } else {
        for (int i = 0; i < 6; i++) Bx[i] = 0.0;
        Bx[3] = -1;
        Bx[4] = 1;
    }
}
FL_HD void fl_s2_boxplus(double *vec, const double *delta)
{
    double Bx[6], Bu[3], q[4], R[9];
    fl_s2_Bx(vec, Bx);
    for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
    fl_mtk_exp3(Bu, 0.5, q);
    flq_to_R(q, R);
    const double o0 = R[0] * vec[0] + R[1] * vec[1] + R[2] * vec[2];
    const double o1 = R[3] * vec[0] + R[4] * vec[1] + R[5] * vec[2];
    const double o2 = R[6] * vec[0] + R[7] * vec[1] + R[8] * vec[2];
    vec[0] = o0; vec[1] = o1; vec[2] = o2;
}
FL_HD void fl_s2_boxminus(const double *vec, const double *other, double *res)
{
    double K[9], hv[3];
    fl_skew(vec, K);
    for (int i = 0; i < 3; i++) hv[i] = K[i * 3] * other[0] + K[i * 3 + 1] * other[1] + K[i * 3 + 2] * other[2];
    const double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
#if FL_IK_SERIES
    {
        const double s2 = hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2];
        if (s2 >= FL_MTK_TOL * FL_MTK_TOL && v_cos > 0.0 && s2 <= 0.0625 * v_cos * v_cos) {   // theta / v_sin = (atan t / t) / v_cos, t = v_sin / v_cos
            double Bx[6], Ko[9], w[3];
            fl_s2_Bx(other, Bx);
            fl_skew(other, Ko);
            for (int i = 0; i < 3; i++) w[i] = Ko[i * 3] * vec[0] + Ko[i * 3 + 1] * vec[1] + Ko[i * 3 + 2] * vec[2];
            const double inv_c = 1.0 / v_cos;
            const double f = fl_series_atan_over(s2 * inv_c * inv_c) * inv_c;
            for (int r = 0; r < 2; r++) res[r] = f * (Bx[0 * 2 + r] * w[0] + Bx[1 * 2 + r] * w[1] + Bx[2 * 2 + r] * w[2]);
            return;
        }
    }
#endif
    const double v_sin = sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    const double theta = atan2(v_sin, v_cos);
    if (v_sin < FL_MTK_TOL) {
        res[0] = (fabs(theta) > FL_MTK_TOL) ? 3.1415926 : 0.0;
        res[1] = 0;
    } else {
        double Bx[6], Ko[9], w[3];
        fl_s2_Bx(other, Bx);
        fl_skew(other, Ko);
        for (int i = 0; i < 3; i++) w[i] = Ko[i * 3] * vec[0] + Ko[i * 3 + 1] * vec[1] + Ko[i * 3 + 2] * vec[2];
        const double f = theta / v_sin;
        for (int r = 0; r < 2; r++) res[r] = f * (Bx[0 * 2 + r] * w[0] + Bx[1 * 2 + r] * w[1] + Bx[2 * 2 + r] * w[2]);
    }
}
// Nx (2x3) = Bx^T hat(vec) / L^2  (S2.hpp:259-264)
FL_HD void fl_s2_Nx_yy(const double *vec, double *Nx)
{
    double Bx[6], K[9];
    const double f = 1 / FL_S2_LEN / FL_S2_LEN;
    fl_s2_Bx(vec, Bx);
    fl_skew(vec, K);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) Nx[r * 3 + c] = f * (Bx[0 * 2 + r] * K[0 * 3 + c] + Bx[1 * 2 + r] * K[1 * 3 + c] + Bx[2 * 2 + r] * K[2 * 3 + c]);
}
// Mx (3x2), S2.hpp:266-280 -- exp_delta is the identity there (scalar(1/2) == 0), kept on purpose
FL_HD void fl_s2_Mx(const double *vec, const double *delta, double *Mx)
{
    double Bx[6], K[9], T[9];
    fl_s2_Bx(vec, Bx);
    fl_skew(vec, K);
    if (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < FL_MTK_TOL) {
        for (int i = 0; i < 9; i++) T[i] = -K[i];
    } else {
        double Bu[3], A[9], At[9], nK[9];
        for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
        fl_A_matrix(Bu, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) At[j * 3 + i] = A[i * 3 + j];
        for (int i = 0; i < 9; i++) nK[i] = -K[i];
        fl_m3mul(nK, At, T);
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 2; c++) Mx[r * 2 + c] = T[r * 3] * Bx[0 * 2 + c] + T[r * 3 + 1] * Bx[1 * 2 + c] + T[r * 3 + 2] * Bx[2 * 2 + c];
}

// state_ikfom boxplus / boxminus (build_manifold.hpp:192-200)
FL_HD void fl_x23_boxplus(double *x, const double *dx)
{
    double e[4];
    for (int i = 0; i < 3; i++) x[FL_X23_POS + i] += dx[i];
    fl_mtk_exp3(dx + 3, 0.5, e);
    flq_mul(x + FL_X23_ROT, e, x + FL_X23_ROT);
    fl_mtk_exp3(dx + 6, 0.5, e);
    flq_mul(x + FL_X23_ORLI, e, x + FL_X23_ORLI);
    for (int i = 0; i < 3; i++) {
        x[FL_X23_OTLI + i] += dx[9 + i];
        x[FL_X23_VEL + i] += dx[12 + i];
        x[FL_X23_BG + i] += dx[15 + i];
        x[FL_X23_BA + i] += dx[18 + i];
    }
    fl_s2_boxplus(x + FL_X23_GRAV, dx + 21);
}
FL_HD void fl_x23_boxminus(const double *x, const double *o, double *dx)
{
    double oc[4], r[4];
    for (int i = 0; i < 3; i++) dx[i] = x[FL_X23_POS + i] - o[FL_X23_POS + i];
    oc[0] = -o[FL_X23_ROT]; oc[1] = -o[FL_X23_ROT + 1]; oc[2] = -o[FL_X23_ROT + 2]; oc[3] = o[FL_X23_ROT + 3];
    flq_mul(oc, x + FL_X23_ROT, r);
    fl_so3_log(r, dx + 3);
    oc[0] = -o[FL_X23_ORLI]; oc[1] = -o[FL_X23_ORLI + 1]; oc[2] = -o[FL_X23_ORLI + 2]; oc[3] = o[FL_X23_ORLI + 3];
    flq_mul(oc, x + FL_X23_ORLI, r);
    fl_so3_log(r, dx + 6);
    for (int i = 0; i < 3; i++) {
        dx[9 + i] = x[FL_X23_OTLI + i] - o[FL_X23_OTLI + i];
        dx[12 + i] = x[FL_X23_VEL + i] - o[FL_X23_VEL + i];
        dx[15 + i] = x[FL_X23_BG + i] - o[FL_X23_BG + i];
        dx[18 + i] = x[FL_X23_BA + i] - o[FL_X23_BA + i];
    }
    fl_s2_boxminus(x + FL_X23_GRAV, o + FL_X23_GRAV, dx + 21);
}

// ---- per-point model at state_ikfom s -------------------------------------------------------
// World point (laserMapping.cpp:980-984), same quaternion arithmetic as Eigen so that the float
// world point -- and with it every gate -- rounds like the CPU path.
FL_HD void fl_world_point23(const double *x, const float *pb, double *p_i /*3*/, float *pw /*3*/)
{
    const double b[3] = {(double)pb[0], (double)pb[1], (double)pb[2]};
    double g[3];
    flq_rot(x + FL_X23_ORLI, 0, b, p_i);
    p_i[0] += x[FL_X23_OTLI]; p_i[1] += x[FL_X23_OTLI + 1]; p_i[2] += x[FL_X23_OTLI + 2];
    flq_rot(x + FL_X23_ROT, 0, p_i, g);
    pw[0] = (float)(g[0] + x[FL_X23_POS]); pw[1] = (float)(g[1] + x[FL_X23_POS + 1]); pw[2] = (float)(g[2] + x[FL_X23_POS + 2]);
}
// gates shared with Mode-18 (laserMapping.cpp:1023-1034,1043)
FL_HD int fl_gates_from_pw(const float *pb, const float *pl, const float *pw, float *pd2_out, int *eff)
{
    const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
    const float pd2 = pl[0] * pw[0] + pl[1] * pw[1] + pl[2] * pw[2] + pl[3];
    const double pbn = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
    const float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(pbn));
    *pd2_out = pd2;
    const int sel = ((double)s > 0.9) ? 1 : 0;
    *eff = (sel && ((double)fabsf(pd2) <= 2.0)) ? 1 : 0;
    return sel;
}
// row = [n, A, B, C] (laserMapping.cpp:1063-1089): C = R^T n, A = p_i x C, B = p_b x (R_LI^T C)
FL_HD void fl_row23(const double *x, const float *pb, const double *p_i, const float *pl, float pd2, double *row /*12*/, double *z)
{
    FL_FP_CONTRACT
    const double nv[3] = {(double)pl[0], (double)pl[1], (double)pl[2]};
    const double b[3] = {(double)pb[0], (double)pb[1], (double)pb[2]};
    double C[3], D[3];
    flq_rot(x + FL_X23_ROT, 1, nv, C);
    flq_rot(x + FL_X23_ORLI, 1, C, D);
    row[0] = nv[0]; row[1] = nv[1]; row[2] = nv[2];
    row[3] = p_i[1] * C[2] - p_i[2] * C[1];
    row[4] = p_i[2] * C[0] - p_i[0] * C[2];
    row[5] = p_i[0] * C[1] - p_i[1] * C[0];
    row[6] = b[1] * D[2] - b[2] * D[1];
    row[7] = b[2] * D[0] - b[0] * D[2];
    row[8] = b[0] * D[1] - b[1] * D[0];
    row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
    *z = -(double)pd2;
}
// The record the pass kernels hand over (FL_SUMS23I = 64 doubles, round 3): the row is [n, A, B, C] with C = R^T n (R = rotation of
// the state, the same for every point of a pass), so every sum that involves C follows from the sums over n: C C^T = R^T (n n^T) R,
// X C^T = (X n^T) R, sum C z = R^T sum n z. The producers therefore accumulate only the 9 x 9 block over [n, A, B] -- 45 + 9 + 3
// numbers instead of 78 + 12 + 3: a third less to accumulate, to reduce, to publish and to gather -- and the solver rebuilds the
// 12 x 12 matrix (ikfom_solve_block.h ikfom_unpack). Layout: [0..44] upper triangle of the 9 x 9, row-major (i <= j);
// [45..53] H^T z; [54] n_eff; [55] sum |pd2|; [56] sum pd2^2; rest zero. Rounding-level differences from summing C per point
// (compared by tolerance, like every fp64 sum).
#define FL_S23I_HTZ 45
#define FL_S23I_NEFF 54
#define FL_S23I_RES 55
#define FL_S23I_RES2 56
FL_HD void fl_accum9(double *v /*64*/, const double *row /* 9 used */, double z)
{
    FL_FP_CONTRACT
    int k = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = i; j < 9; j++) { v[k] += row[i] * row[j]; k++; }
#pragma unroll
    for (int i = 0; i < 9; i++) v[FL_S23I_HTZ + i] += row[i] * z;
}
FL_HD constexpr int fl_tri9(int i, int j) { return (i <= j) ? (i * 9 - i * (i - 1) / 2 + (j - i)) : (j * 9 - j * (j - 1) / 2 + (i - j)); }
// entry (i, j) of the 12 x 12 h_x^T h_x and entry i of h_x^T h out of the internal record; Rm = flq_to_R(x.rot), row-major
FL_HD double fl_s12_from_s9(const double *s /*64*/, const double *Rm, int i, int j)
{
    FL_FP_CONTRACT
    if (i > j) { const int t = i; i = j; j = t; }
    if (j < 9) return s[fl_tri9(i, j)];
    const int c = j - 9;
    if (i < 9) return Rm[c] * s[fl_tri9(i, 0)] + Rm[3 + c] * s[fl_tri9(i, 1)] + Rm[6 + c] * s[fl_tri9(i, 2)];
    const int a = i - 9;
    double acc = 0.0;
    for (int k = 0; k < 3; k++)
        acc += Rm[3 * k + a] * (Rm[c] * s[fl_tri9(k, 0)] + Rm[3 + c] * s[fl_tri9(k, 1)] + Rm[6 + c] * s[fl_tri9(k, 2)]);
    return acc;
}
FL_HD double fl_htz12_from_s9(const double *s /*64*/, const double *Rm, int i)
{
    FL_FP_CONTRACT
    if (i < 9) return s[FL_S23I_HTZ + i];
    const int c = i - 9;
    return Rm[c] * s[FL_S23I_HTZ] + Rm[3 + c] * s[FL_S23I_HTZ + 1] + Rm[6 + c] * s[FL_S23I_HTZ + 2];
}
FL_HD void fl_accum12(double *v /*96*/, const double *row, double z)
{
    FL_FP_CONTRACT
    int k = 0;
#pragma unroll
    for (int i = 0; i < 12; i++)
#pragma unroll
        for (int j = i; j < 12; j++) { v[k] += row[i] * row[j]; k++; }
#pragma unroll
    for (int i = 0; i < 12; i++) v[FL_S23_HTZ + i] += row[i] * z;
}

// ---- one iteration of update_iterated_dyn_share_modified given the reduced record ---------------
// Apply J (k x k) to rows idx..idx+k of dst from src, and J^T on the right to columns (n x n, row-major)
FL_HD void fl_rows_apply(double *dst, const double *src, int n, int idx, int k, const double *J)
{
    for (int i = 0; i < n; i++) {
        double t[3];
        for (int r = 0; r < k; r++) {
            double s = 0.0;
            for (int c = 0; c < k; c++) s += J[r * k + c] * src[(idx + c) * n + i];
            t[r] = s;
        }
        for (int r = 0; r < k; r++) dst[(idx + r) * n + i] = t[r];
    }
}
FL_HD void fl_cols_apply(double *P, int n, int idx, int k, const double *J)
{
    for (int i = 0; i < n; i++) {
        double t[3];
        for (int r = 0; r < k; r++) {
            double s = 0.0;
            for (int c = 0; c < k; c++) s += P[i * n + idx + c] * J[r * k + c];
            t[r] = s;
        }
        for (int r = 0; r < k; r++) P[i * n + idx + r] = t[r];
    }
}
// SO3 / S2 projection Jacobians of esekfom.hpp:1656-1696 (and :1833-1897 with seg = dx_)
FL_HD void fl_ikfom_J_so3(const double *seg, double *J)
{
    double A[9];
    fl_A_matrix(seg, A);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) J[j * 3 + i] = A[i * 3 + j];
}
FL_HD void fl_ikfom_J_s2(const double *grav_x, const double *grav_prop, const double *seg, double *J2)
{
    double Nx[6], Mx[6];
    fl_s2_Nx_yy(grav_x, Nx);
    fl_s2_Mx(grav_prop, seg, Mx);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) J2[r * 2 + c] = Nx[r * 3] * Mx[0 * 2 + c] + Nx[r * 3 + 1] * Mx[1 * 2 + c] + Nx[r * 3 + 2] * Mx[2 * 2 + c];
}

// Cholesky of an SPD n x n (n <= 12) in place (lower), returns 1 if not positive definite.
FL_HD int fl_chol(double *M, int n)
{
    int bad = 0;
    for (int j = 0; j < n; j++) {
        double d = M[j * n + j];
        for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k];
        if (!(d > 0.0)) bad = 1;
        const double l = sqrt(d);
        M[j * n + j] = l;
        for (int i = j + 1; i < n; i++) {
            double v = M[i * n + j];
            for (int k = 0; k < j; k++) v -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = v / l;
        }
    }
    return bad;
}
FL_HD void fl_chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

struct FlIkfomCtl {
    int iter_i;       // loop index i of esekfom.hpp:1633 (starts at -1)
    int t_count;      // t
    int converge;     // dyn_share.converge (kNN wanted for the NEXT call of h_share_model)
    int finished;     // the final covariance block has run
    int max_iter;
    int status;
};

// One pass of the loop body esekfom.hpp:1644-1926 after h_share_model produced `sums`.
//   x (26) in/out, xprop (26), Pprop (23x23), P (23x23 out: projected P_, or the final P_ when finishing),
//   limit[23], R = LASER_POINT_COV.  work: >= FL_IKFOM_WORK doubles of scratch (LDS on the device).
#define FL_IKFOM_WORK (5 * 144 + 529 + 2 * 276)
FL_HD void fl_ikfom_iterate(double *x, const double *xprop, const double *Pprop, double *P, const double *limit, double R,
                            const double *sums, FlIkfomCtl *ctl, double *dx_out /*23*/, double *work)
{
    const int n = FL_N23;
    double dx[FL_N23], dx_new[FL_N23];
    fl_x23_boxminus(x, xprop, dx);
    for (int i = 0; i < n; i++) dx_new[i] = dx[i];
    for (int i = 0; i < n * n; i++) P[i] = Pprop[i];
    double J[9], J2[4];
    for (int b = 0; b < 2; b++) {
        const int idx = b ? 6 : 3;
        double tv[3];
        fl_ikfom_J_so3(dx + idx, J);
        for (int r = 0; r < 3; r++) tv[r] = J[r * 3] * dx_new[idx] + J[r * 3 + 1] * dx_new[idx + 1] + J[r * 3 + 2] * dx_new[idx + 2];
        for (int r = 0; r < 3; r++) dx_new[idx + r] = tv[r];
        fl_rows_apply(P, P, n, idx, 3, J);
        fl_cols_apply(P, n, idx, 3, J);
    }
    {
        double tv[2];
        fl_ikfom_J_s2(x + FL_X23_GRAV, xprop + FL_X23_GRAV, dx + 21, J2);
        tv[0] = J2[0] * dx_new[21] + J2[1] * dx_new[22];
        tv[1] = J2[2] * dx_new[21] + J2[3] * dx_new[22];
        dx_new[21] = tv[0]; dx_new[22] = tv[1];
        fl_rows_apply(P, P, n, 21, 2, J2);
        fl_cols_apply(P, n, 21, 2, J2);
    }

    // ---- gain through the SPD 12x12 system  (A12 + A12 S A12) y = A12 (HTh + S dx_new12)
    double *S = work + 144, *A12 = work + 288, *M = work + 432, *X = work + 576;
    double rhs[12], y[12];
    {
        int k = 0;
        for (int i = 0; i < 12; i++)
            for (int j = i; j < 12; j++) { S[i * 12 + j] = sums[k]; S[j * 12 + i] = sums[k]; k++; }
    }
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) A12[i * 12 + j] = 0.5 * (P[i * n + j] + P[j * n + i]) / R;
    double *SA = work;                       // 144: S * A12
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            double s = 0.0;
            for (int k = 0; k < 12; k++) s += S[i * 12 + k] * A12[k * 12 + j];
            SA[i * 12 + j] = s;
        }
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            double s = A12[i * 12 + j];
            for (int k = 0; k < 12; k++) s += A12[i * 12 + k] * SA[k * 12 + j];
            M[i * 12 + j] = s;
        }
    for (int i = 0; i < 12; i++)            // symmetrise before the factorisation
        for (int j = 0; j < i; j++) { const double m = 0.5 * (M[i * 12 + j] + M[j * 12 + i]); M[i * 12 + j] = m; M[j * 12 + i] = m; }
    int st = fl_chol(M, 12) ? 1 : 0;
    for (int i = 0; i < 12; i++) {
        double s = sums[FL_S23_HTZ + i];
        for (int k = 0; k < 12; k++) s += S[i * 12 + k] * dx_new[k];
        rhs[i] = s;
    }
    for (int i = 0; i < 12; i++) {
        double s = 0.0;
        for (int k = 0; k < 12; k++) s += A12[i * 12 + k] * rhs[k];
        y[i] = s;
    }
    fl_chol_solve(M, 12, y);
    // dx_ = A[:,0:12] y - dx_new   (= K_h + (K_x - I) dx_new)
    double dx_[FL_N23];
    for (int r = 0; r < n; r++) {
        double s = 0.0;
        for (int c = 0; c < 12; c++) s += (P[r * n + c] / R) * y[c];
        dx_[r] = s - dx_new[r];
    }
    fl_x23_boxplus(x, dx_);
    int converge = 1;
    for (int i = 0; i < n; i++) {
        if (fabs(dx_[i]) > limit[i]) { converge = 0; break; }
    }
    for (int i = 0; i < n; i++)
        if (!(fabs(dx_[i]) <= DBL_MAX)) st |= 2;
    int t = ctl->t_count;
    const int i_loop = ctl->iter_i;
    if (converge) t++;
    if (!t && i_loop == ctl->max_iter - 2) converge = 1;
    ctl->t_count = t;
    ctl->converge = converge;
    ctl->status |= st;
    for (int i = 0; i < n; i++) dx_out[i] = dx_[i];

    if (t > 1 || i_loop == ctl->max_iter - 1) {
        // ---- final covariance block, esekfom.hpp:1831-1924
        double *L_ = work + 720;            // 529
        double *Kx = work + 720 + 529;      // 23 x 12 : K_x[:,0:12] = A[:,0:12] Minv A12 S
        double *top = work + 720 + 529 + 276;
        // X = Minv (A12 S)  (12x12), column by column ; A12 S = (S A12)^T
        for (int c = 0; c < 12; c++) {
            double col[12];
            for (int i = 0; i < 12; i++) {
                double s = 0.0;
                for (int k = 0; k < 12; k++) s += A12[i * 12 + k] * S[k * 12 + c];
                col[i] = s;
            }
            fl_chol_solve(M, 12, col);
            for (int i = 0; i < 12; i++) X[i * 12 + c] = col[i];
        }
        for (int r = 0; r < n; r++)
            for (int c = 0; c < 12; c++) {
                double s = 0.0;
                for (int k = 0; k < 12; k++) s += (P[r * n + k] / R) * X[k * 12 + c];
                Kx[r * 12 + c] = s;
            }
        for (int i = 0; i < n * n; i++) L_[i] = P[i];
        for (int b = 0; b < 2; b++) {
            const int idx = b ? 6 : 3;
            fl_ikfom_J_so3(dx_ + idx, J);
            fl_rows_apply(L_, P, n, idx, 3, J);
            for (int c = 0; c < 12; c++) {
                double tv[3];
                for (int r = 0; r < 3; r++) tv[r] = J[r * 3] * Kx[idx * 12 + c] + J[r * 3 + 1] * Kx[(idx + 1) * 12 + c] + J[r * 3 + 2] * Kx[(idx + 2) * 12 + c];
                for (int r = 0; r < 3; r++) Kx[(idx + r) * 12 + c] = tv[r];
            }
            fl_cols_apply(L_, n, idx, 3, J);
            fl_cols_apply(P, n, idx, 3, J);
        }
        {
            fl_ikfom_J_s2(x + FL_X23_GRAV, xprop + FL_X23_GRAV, dx_ + 21, J2);
            fl_rows_apply(L_, P, n, 21, 2, J2);
            for (int c = 0; c < 12; c++) {
                const double t0 = J2[0] * Kx[21 * 12 + c] + J2[1] * Kx[22 * 12 + c];
                const double t1 = J2[2] * Kx[21 * 12 + c] + J2[3] * Kx[22 * 12 + c];
                Kx[21 * 12 + c] = t0; Kx[22 * 12 + c] = t1;
            }
            fl_cols_apply(L_, n, 21, 2, J2);
            fl_cols_apply(P, n, 21, 2, J2);
        }
        // P_ = L_ - K_x[:,0:12] * P_[0:12,:]  (P rows 0..11 are read while rows are rewritten: stage them)
        for (int i = 0; i < 12 * n; i++) top[i] = P[i];
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) {
                double s = 0.0;
                for (int k = 0; k < 12; k++) s += Kx[r * 12 + k] * top[k * n + c];
                P[r * n + c] = L_[r * n + c] - s;
            }
        ctl->finished = 1;
    }
    ctl->iter_i = i_loop + 1;
}
