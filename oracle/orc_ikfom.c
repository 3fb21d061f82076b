/*
 * oracle/orc_ikfom.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Mode-23: the IKFoM variant (USE_IKFOM) FAST-LIVO ships disabled -- h_share_model plus
 * esekf::update_iterated_dyn_share_modified on the 23-DOF state_ikfom manifold.
 *
 * Reference lines restated (file:line under /root/reference):
 *   state_ikfom layout                 include/use-ikfom.hpp:6-21
 *   h_share_model                      src/laserMapping.cpp:961-1093
 *   update_iterated_dyn_share_modified include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928
 *   boxplus / boxminus macro           include/IKFoM_toolkit/mtk/build_manifold.hpp:192-200
 *   SO3 boxplus/boxminus/exp/log       include/IKFoM_toolkit/mtk/types/SOn.hpp:233-239,284-297
 *   S2 boxplus/boxminus/Bx/Nx_yy/Mx    include/IKFoM_toolkit/mtk/types/S2.hpp:136-280
 *   vect boxplus/boxminus              include/IKFoM_toolkit/mtk/types/vect.hpp:117-122
 *   cos_sinc_sqrt, exp, log, A_matrix  include/IKFoM_toolkit/mtk/src/mtkmath.hpp:142-174,236-288
 *   Eigen quaternion product / rotate / toRotationMatrix (third party, unpinned >=3.3.4)
 * Quirk kept on purpose: S2_Mx builds exp_delta with scalar(1/2) == 0 (integer division,
 * S2.hpp:277) so exp_delta is the identity.
 * Held to the reference's own text since round 4 (h_share_model, the updater, the toolkit's vect / SO3 / S2 / mtkmath: oracle/ref_eigen,
 * tests/test_ref_eigen_cpu.py -- bit for bit); Eigen's own arithmetic stays unpinned -- see fastlivo_oracle.h. (This configuration does not even compile in the
 * reference: SURVEY.md fact 1.)
 */
#include "fastlivo_oracle.h"
#include "orc_lio_common.h"
#include "orc_math.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define N23 23
#define S2_LEN (98090.0 / 10000.0) /* use-ikfom.hpp:8 : scalar(den)/scalar(num) */
#define MTK_TOL 1e-11              /* mtkmath.hpp:122 */

/* ---- Eigen::Quaternion helpers (coeff order x,y,z,w) ---------------------------------------- */
static void q_mul(const double *a, const double *b, double *o)
{
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void q_conj(const double *a, double *o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
/* QuaternionBase::_transformVector: v + w*(2 q x v) + q x (2 q x v) */
static void q_rot(const double *q, const double *v, double *o)
{
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    o[0] = v[0] + q[3] * uv[0] + c[0];
    o[1] = v[1] + q[3] * uv[1] + c[1];
    o[2] = v[2] + q[3] * uv[2] + c[2];
}
static void q_to_R(const double *q, double *R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* mtkmath.hpp:142-174 */
static void cos_sinc_sqrt(double x2, double *co, double *si)
{
    const double taylor_0_bound = DBL_EPSILON;
    const double taylor_2_bound = sqrt(taylor_0_bound);
    const double taylor_n_bound = sqrt(taylor_2_bound);
    if (x2 >= taylor_n_bound) {
        double x = sqrt(x2);
        *co = cos(x); *si = sin(x) / x;
        return;
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
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
/* mtkmath.hpp:249-256 : returns w, writes the vector part */
static double mtk_exp3(double *res, const double *vec, double scale)
{
    double norm2 = vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2];
    double c, s;
    cos_sinc_sqrt(scale * scale * norm2, &c, &s);
    double mult = s * scale;
    res[0] = mult * vec[0]; res[1] = mult * vec[1]; res[2] = mult * vec[2];
    return c;
}
/* mtkmath.hpp:268-288 with plus_minus_periodicity = true (SOn.hpp:295) */
static void mtk_log3(double *res, double w, const double *vec, double scale)
{
    double nv = sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
    if (nv < MTK_TOL) nv = MTK_TOL;
    double s = scale / nv * atan(nv / w);
    res[0] = s * vec[0]; res[1] = s * vec[1]; res[2] = s * vec[2];
}
/* mtkmath.hpp:236-247 */
static void A_matrix(const double *v, double *res)
{
    double squaredNorm = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double norm = sqrt(squaredNorm);
    for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (!(norm < MTK_TOL)) {
        /* `Identity() + a * hat(v) + b * hat(v) * hat(v)` groups as (I + a K) + ((b K) K): the scalar goes into the LEFT factor (until round 4
         * this computed b (K K): one unit in the last place in a quarter of random vectors, found by running the toolkit's own text,
         * oracle/ref_eigen ikf) */
        double K[9], bK[9], bKK[9];
        skew3(v, K);
        double a = (1 - cos(norm)) / squaredNorm, b = (1 - sin(norm) / norm) / squaredNorm;
        for (int i = 0; i < 9; i++) bK[i] = b * K[i];
        m3_mul(bK, K, bKK);
        for (int i = 0; i < 9; i++) res[i] = (res[i] + a * K[i]) + bKK[i];
    }
}
/* SO3::boxplus, SOn.hpp:233-236 */
static void so3_boxplus(double *q, const double *d)
{
    double e[4];
    e[3] = mtk_exp3(e, d, 1.0 / 2);
    q_mul(q, e, q);
}
/* SO3::boxminus, SOn.hpp:237-239 */
static void so3_boxminus(const double *q, const double *other, double *res)
{
    double oc[4], r[4];
    q_conj(other, oc);
    q_mul(oc, q, r);
    mtk_log3(res, r[3], r, 2.0);
}
/* S2::S2_Bx with S2_typ == 1 (use-ikfom.hpp:8), S2.hpp:215-231. Bx is 3x2 row-major. */
static void s2_Bx(const double *vec, double *Bx)
{
    const double L = S2_LEN;
    if (vec[0] + L > MTK_TOL) {
        Bx[0] = -vec[1];                         Bx[1] = -vec[2];
        Bx[2] = L - vec[1] * vec[1] / (L + vec[0]); Bx[3] = -vec[2] * vec[1] / (L + vec[0]);
        Bx[4] = -vec[2] * vec[1] / (L + vec[0]);    Bx[5] = L - vec[2] * vec[2] / (L + vec[0]);
        for (int i = 0; i < 6; i++) Bx[i] /= L;
    } else {
        for (int i = 0; i < 6; i++) Bx[i] = 0.0;
        Bx[1 * 2 + 1] = -1;
        Bx[2 * 2 + 0] = 1;
    }
}
/* S2::boxplus, S2.hpp:136-142 */
static void s2_boxplus(double *vec, const double *delta)
{
    double Bx[6], Bu[3], q[4], R[9], o[3];
    s2_Bx(vec, Bx);
    for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
    q[3] = mtk_exp3(q, Bu, 1.0 / 2);
    q_to_R(q, R);
    m3_vec(R, vec, o);
    vec[0] = o[0]; vec[1] = o[1]; vec[2] = o[2];
}
/* S2::boxminus, S2.hpp:144-167 : res = this [-] other */
static void s2_boxminus(const double *vec, const double *other, double *res)
{
    double K[9], hv[3];
    skew3(vec, K);
    m3_vec(K, other, hv);
    double v_sin = norm3(hv);
    double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
    double theta = atan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
        if (fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[6], Ko[9], M[6], f = theta / v_sin;
        s2_Bx(other, Bx);
        skew3(other, Ko);
        /* (f * Bx^T) * hat(other) * vec, left to right */
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++)
                M[r * 3 + c] = (f * Bx[0 * 2 + r]) * Ko[0 * 3 + c] + (f * Bx[1 * 2 + r]) * Ko[1 * 3 + c] + (f * Bx[2 * 2 + r]) * Ko[2 * 3 + c];
        for (int r = 0; r < 2; r++) res[r] = M[r * 3] * vec[0] + M[r * 3 + 1] * vec[1] + M[r * 3 + 2] * vec[2];
    }
}
/* S2::S2_Nx_yy, S2.hpp:259-264 : Nx (2x3) */
static void s2_Nx_yy(const double *vec, double *Nx)
{
    double Bx[6], K[9];
    const double f = 1 / S2_LEN / S2_LEN;
    s2_Bx(vec, Bx);
    skew3(vec, K);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            Nx[r * 3 + c] = (f * Bx[0 * 2 + r]) * K[0 * 3 + c] + (f * Bx[1 * 2 + r]) * K[1 * 3 + c] + (f * Bx[2 * 2 + r]) * K[2 * 3 + c];
}
/* S2::S2_Mx, S2.hpp:266-280 : Mx (3x2) */
static void s2_Mx(const double *vec, const double *delta, double *Mx)
{
    double Bx[6], K[9];
    s2_Bx(vec, Bx);
    skew3(vec, K);
    if (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < MTK_TOL) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 2; c++)
                Mx[r * 2 + c] = (-K[r * 3 + 0]) * Bx[0 * 2 + c] + (-K[r * 3 + 1]) * Bx[1 * 2 + c] + (-K[r * 3 + 2]) * Bx[2 * 2 + c];
    } else {
        double Bu[3], q[4], R[9], A[9], At[9], T1[9], T2[9];
        for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
        q[3] = mtk_exp3(q, Bu, (double)(1 / 2)); /* integer division: scale 0 => identity */
        q_to_R(q, R);
        for (int i = 0; i < 9; i++) R[i] = -R[i];
        A_matrix(Bu, A);
        m3_tr(A, At);
        m3_mul(R, K, T1);
        m3_mul(T1, At, T2);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 2; c++)
                Mx[r * 2 + c] = T2[r * 3 + 0] * Bx[0 * 2 + c] + T2[r * 3 + 1] * Bx[1 * 2 + c] + T2[r * 3 + 2] * Bx[2 * 2 + c];
    }
}

void orc_state23_boxplus(orc_state23 *x, const double *dx)
{
    for (int i = 0; i < 3; i++) x->pos[i] += 1.0 * dx[0 + i];
    so3_boxplus(x->rot, dx + 3);
    so3_boxplus(x->offset_R_L_I, dx + 6);
    for (int i = 0; i < 3; i++) {
        x->offset_T_L_I[i] += 1.0 * dx[9 + i];
        x->vel[i] += 1.0 * dx[12 + i];
        x->bg[i] += 1.0 * dx[15 + i];
        x->ba[i] += 1.0 * dx[18 + i];
    }
    s2_boxplus(x->grav, dx + 21);
}
void orc_state23_boxminus(const orc_state23 *x, const orc_state23 *o, double *dx)
{
    for (int i = 0; i < 3; i++) dx[i] = x->pos[i] - o->pos[i];
    so3_boxminus(x->rot, o->rot, dx + 3);
    so3_boxminus(x->offset_R_L_I, o->offset_R_L_I, dx + 6);
    for (int i = 0; i < 3; i++) {
        dx[9 + i] = x->offset_T_L_I[i] - o->offset_T_L_I[i];
        dx[12 + i] = x->vel[i] - o->vel[i];
        dx[15 + i] = x->bg[i] - o->bg[i];
        dx[18 + i] = x->ba[i] - o->ba[i];
    }
    s2_boxminus(x->grav, o->grav, dx + 21);
}

/* unit entry points (one restated toolkit function each; used by oracle/ref_eigen's text unit of the updater and by tests) */
void orc_unit_q_rot(const double *q /*x,y,z,w*/, const double *v /*3*/, double *o /*3*/) { q_rot(q, v, o); }
void orc_unit_q_to_R(const double *q /*x,y,z,w*/, double *R /*3x3*/) { q_to_R(q, R); }
void orc_unit_A_matrix(const double *v /*3*/, double *res /*3x3*/) { A_matrix(v, res); }
void orc_unit_s2_Nx_yy(const double *vec /*3*/, double *Nx /*2x3*/) { s2_Nx_yy(vec, Nx); }
void orc_unit_s2_Mx(const double *vec /*3*/, const double *delta /*2*/, double *Mx /*3x2*/) { s2_Mx(vec, delta, Mx); }

/* world point at state s: laserMapping.cpp:980-984 */
static void world_point23(const orc_state23 *s, const float *pb, float *pw)
{
    double p_body[3] = {(double)pb[0], (double)pb[1], (double)pb[2]};
    double a[3], g[3];
    q_rot(s->offset_R_L_I, p_body, a);
    a[0] += s->offset_T_L_I[0]; a[1] += s->offset_T_L_I[1]; a[2] += s->offset_T_L_I[2];
    q_rot(s->rot, a, g);
    pw[0] = (float)(g[0] + s->pos[0]); pw[1] = (float)(g[1] + s->pos[1]); pw[2] = (float)(g[2] + s->pos[2]);
}

int orc_h_share_model(const orc_state23 *s, const float *body_xyz, const float *nbr_xyz,
                      uint8_t *sel, int n, int nthreads, float *world_xyz, float *normvec,
                      double *res_last, double *h_x, double *h, double *total_residual)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for
#endif
    for (int i = 0; i < n; i++) {
        double p_body[3] = {(double)body_xyz[i * 3], (double)body_xyz[i * 3 + 1], (double)body_xyz[i * 3 + 2]};
        float pw[3];
        world_point23(s, body_xyz + (size_t)i * 3, pw);
        if (world_xyz) { world_xyz[i * 3] = pw[0]; world_xyz[i * 3 + 1] = pw[1]; world_xyz[i * 3 + 2] = pw[2]; }
        if (!sel[i]) continue;
        sel[i] = (uint8_t)orc_point_residual(nbr_xyz + (size_t)i * 15, pw, p_body, normvec + (size_t)i * 4, res_last + i);
    }
    int neff = 0;
    double tr = 0.0;
    double rc[4], orc_[4];
    q_conj(s->rot, rc);
    q_conj(s->offset_R_L_I, orc_);
    for (int i = 0; i < n; i++) {
        if (!(sel[i] && res_last[i] <= 2.0)) continue;
        tr += res_last[i];
        double pbe[3] = {(double)body_xyz[i * 3], (double)body_xyz[i * 3 + 1], (double)body_xyz[i * 3 + 2]};
        double bcm[9], pcm[9], pt[3];
        skew3(pbe, bcm);
        q_rot(s->offset_R_L_I, pbe, pt);
        pt[0] += s->offset_T_L_I[0]; pt[1] += s->offset_T_L_I[1]; pt[2] += s->offset_T_L_I[2];
        skew3(pt, pcm);
        double nv[3] = {(double)normvec[i * 4], (double)normvec[i * 4 + 1], (double)normvec[i * 4 + 2]};
        double C[3], A[3], B[3], t[3];
        q_rot(rc, nv, C);                 /* C = s.rot.conjugate() * norm_vec */
        m3_vec(pcm, C, A);                /* A = point_crossmat * C */
        /* B = point_be_crossmat * s.offset_R_L_I.conjugate() * C : (matrix * quaternion) is a
         * matrix product with the quaternion's rotation matrix, then times C */
        {
            double Rc[9], M[9];
            q_to_R(orc_, Rc);
            m3_mul(bcm, Rc, M);
            m3_vec(M, C, B);
        }
        (void)t;
        double *row = h_x + (size_t)neff * 12;
        row[0] = nv[0]; row[1] = nv[1]; row[2] = nv[2];
        row[3] = A[0]; row[4] = A[1]; row[5] = A[2];
        row[6] = B[0]; row[7] = B[1]; row[8] = B[2];
        row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
        h[neff] = -(double)normvec[i * 4 + 3];
        neff++;
    }
    if (total_residual) *total_residual = tr;
    return neff;
}

/* helpers: apply J (k x k) to rows / columns idx..idx+k of an n x n row-major matrix */
static void rows_apply(double *P, int n, int idx, int k, const double *J, const double *src)
{
    for (int i = 0; i < n; i++) {
        double t[3];
        for (int r = 0; r < k; r++) {
            double s = 0.0;
            for (int c = 0; c < k; c++) s += J[r * k + c] * src[(idx + c) * n + i];
            t[r] = s;
        }
        for (int r = 0; r < k; r++) P[(idx + r) * n + i] = t[r];
    }
}
static void cols_apply(double *P, int n, int idx, int k, const double *J /* right-multiplied by J^T */)
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

/* ---- update_iterated_dyn_share_modified (esekfom.hpp:1619-1928) around a measurement callback of the reference's shape:
 * `typedef void measurementModel_dyn_share(state &, dyn_share_datastruct<scalar_type> &)` (esekfom.hpp:129), registered by
 * init_dyn_share (:238-254) and invoked once per iteration at :1636.  The callback sees the state and the in/out flags `valid`
 * (reset to true before every call, :1635) and `converge`, and hands back h_x (rows x 12, row-major) and h (rows) -- the members
 * of dyn_share_datastruct the updater reads (:1641,:1712,:1781,:1801,:1806).  `valid == false` skips the iteration (:1649-1652). */
int orc_ikfom_update_dyn_share(orc_state23 *x_, double *P_, double R, int maximum_iter, const double *limit,
                               orc_h_dyn_share_fn h_dyn_share, void *h_ctx, orc_ikfom_out *out)
{
    static const int SO3_idx[2] = {3, 6};
    const int S2_idx = 21;

    int converge = 1, valid = 1, t = 0, status = 0, iters = 0, neff = 0;
    orc_state23 x_propagated = *x_;
    double P_propagated[N23 * N23], L_[N23 * N23];
    memcpy(P_propagated, P_, sizeof P_propagated);
    double K_h[N23], K_x[N23 * N23], dx_new[N23], dx_[N23], HTH[144], HTh[12];
    memset(dx_new, 0, sizeof dx_new); memset(dx_, 0, sizeof dx_);
    memset(HTH, 0, sizeof HTH); memset(HTh, 0, sizeof HTh);
    int finished = 0;

    for (int i = -1; i < maximum_iter && !finished; i++) {
        const double *h_x = NULL, *h = NULL;
        valid = 1;
        h_dyn_share(h_ctx, x_, &valid, &converge, &neff, &h_x, &h);
        iters++;
        const int dof_Measurement = neff;
        double dx[N23];
        orc_state23_boxminus(x_, &x_propagated, dx);
        memcpy(dx_new, dx, sizeof dx);
        if (!valid) continue;                                  /* esekfom.hpp:1649-1652 */
        memcpy(P_, P_propagated, sizeof P_propagated);

        for (int b = 0; b < 2; b++) {
            int idx = SO3_idx[b];
            double A[9], J[9], tv[3];
            A_matrix(dx + idx, A);
            m3_tr(A, J);
            m3_vec(J, dx_new + idx, tv);
            dx_new[idx] = tv[0]; dx_new[idx + 1] = tv[1]; dx_new[idx + 2] = tv[2];
            rows_apply(P_, N23, idx, 3, J, P_);
            cols_apply(P_, N23, idx, 3, J);
        }
        {
            double Nx[6], Mx[6], J2[4], tv[2];
            s2_Nx_yy(x_->grav, Nx);
            s2_Mx(x_propagated.grav, dx + S2_idx, Mx);
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 2; c++) J2[r * 2 + c] = Nx[r * 3] * Mx[0 * 2 + c] + Nx[r * 3 + 1] * Mx[1 * 2 + c] + Nx[r * 3 + 2] * Mx[2 * 2 + c];
            tv[0] = J2[0] * dx_new[S2_idx] + J2[1] * dx_new[S2_idx + 1];
            tv[1] = J2[2] * dx_new[S2_idx] + J2[3] * dx_new[S2_idx + 1];
            dx_new[S2_idx] = tv[0]; dx_new[S2_idx + 1] = tv[1];
            rows_apply(P_, N23, S2_idx, 2, J2, P_);
            cols_apply(P_, N23, S2_idx, 2, J2);
        }

        if (N23 > dof_Measurement) {
            /* esekfom.hpp:1712-1741 : K_ = P H^T (H P H^T / R + I)^-1 / R */
            const int m = dof_Measurement;
            memset(K_h, 0, sizeof K_h); memset(K_x, 0, sizeof K_x);
            if (m > 0) {
                double *PHt = (double *)malloc(sizeof(double) * N23 * m);   /* 23 x m */
                double *S = (double *)malloc(sizeof(double) * m * m), *Si = (double *)malloc(sizeof(double) * m * m);
                double *K_ = (double *)malloc(sizeof(double) * N23 * m);
                for (int r = 0; r < N23; r++)
                    for (int c = 0; c < m; c++) {
                        double s = 0.0;
                        for (int k = 0; k < 12; k++) s += P_[r * N23 + k] * h_x[c * 12 + k];
                        PHt[r * m + c] = s;
                    }
                /* `h_x_cur * P_ * h_x_cur.transpose() / R` groups as ((H P) H^T) / R (until round 4 this computed H (P H^T): the same to
                 * rounding, found by running the updater's own text, oracle/ref_eigen ikf).  Columns 12..22 of h_x_cur are zero. */
                double *HP = (double *)malloc(sizeof(double) * m * N23);    /* m x 23 */
                for (int r = 0; r < m; r++)
                    for (int c = 0; c < N23; c++) {
                        double s = 0.0;
                        for (int k = 0; k < 12; k++) s += h_x[r * 12 + k] * P_[k * N23 + c];
                        HP[r * N23 + c] = s;
                    }
                for (int r = 0; r < m; r++)
                    for (int c = 0; c < m; c++) {
                        double s = 0.0;
                        for (int k = 0; k < 12; k++) s += HP[r * N23 + k] * h_x[c * 12 + k];
                        S[r * m + c] = s / R + ((r == c) ? 1.0 : 0.0);
                    }
                free(HP);
                status |= orc_inverse(m, S, Si);
                for (int r = 0; r < N23; r++)
                    for (int c = 0; c < m; c++) {
                        double s = 0.0;
                        for (int k = 0; k < m; k++) s += PHt[r * m + k] * Si[k * m + c];
                        K_[r * m + c] = s / R;
                    }
                for (int r = 0; r < N23; r++) {
                    double s = 0.0;
                    for (int k = 0; k < m; k++) s += K_[r * m + k] * h[k];
                    K_h[r] = s;
                    for (int c = 0; c < 12; c++) {
                        double s2 = 0.0;
                        for (int k = 0; k < m; k++) s2 += K_[r * m + k] * h_x[k * 12 + c];
                        K_x[r * N23 + c] = s2;
                    }
                }
                free(PHt); free(S); free(Si); free(K_);
            }
            for (int a = 0; a < 12; a++) {
                for (int b = 0; b < 12; b++) { double s = 0.0; for (int k = 0; k < m; k++) s += h_x[k * 12 + a] * h_x[k * 12 + b]; HTH[a * 12 + b] = s; }
                double s = 0.0; for (int k = 0; k < m; k++) s += h_x[k * 12 + a] * h[k]; HTh[a] = s;
            }
        } else {
            /* esekfom.hpp:1779-1806 */
            double Pt[N23 * N23], P_temp[N23 * N23], P_inv[N23 * N23];
            for (int k = 0; k < N23 * N23; k++) Pt[k] = P_[k] / R;
            status |= orc_inverse(N23, Pt, P_temp);
            for (int a = 0; a < 12; a++) {
                for (int b = 0; b < 12; b++) {
                    double s = 0.0;
                    for (int k = 0; k < neff; k++) s += h_x[k * 12 + a] * h_x[k * 12 + b];
                    HTH[a * 12 + b] = s;
                    P_temp[a * N23 + b] += s;
                }
            }
            status |= orc_inverse(N23, P_temp, P_inv);
            /* K_h = (P_inv[:,0:12] * h_x^T) * h  -- evaluated left to right like Eigen */
            double *T = (double *)malloc(sizeof(double) * N23 * (size_t)neff);
            for (int r = 0; r < N23; r++)
                for (int k = 0; k < neff; k++) {
                    double s = 0.0;
                    for (int c = 0; c < 12; c++) s += P_inv[r * N23 + c] * h_x[k * 12 + c];
                    T[(size_t)r * neff + k] = s;
                }
            for (int r = 0; r < N23; r++) {
                double s = 0.0;
                for (int k = 0; k < neff; k++) s += T[(size_t)r * neff + k] * h[k];
                K_h[r] = s;
            }
            free(T);
            for (int a = 0; a < 12; a++) { double s = 0.0; for (int k = 0; k < neff; k++) s += h_x[k * 12 + a] * h[k]; HTh[a] = s; }
            memset(K_x, 0, sizeof K_x);
            for (int r = 0; r < N23; r++)
                for (int c = 0; c < 12; c++) {
                    double s = 0.0;
                    for (int k = 0; k < 12; k++) s += P_inv[r * N23 + k] * HTH[k * 12 + c];
                    K_x[r * N23 + c] = s;
                }
        }

        /* dx_ = K_h + (K_x - I) * dx_new */
        for (int r = 0; r < N23; r++) {
            double s = 0.0;
            for (int c = 0; c < N23; c++) s += (K_x[r * N23 + c] - ((r == c) ? 1.0 : 0.0)) * dx_new[c];
            dx_[r] = K_h[r] + s;
        }
        orc_state23_boxplus(x_, dx_);
        converge = 1;
        for (int k = 0; k < N23; k++)
            if (fabs(dx_[k]) > limit[k]) { converge = 0; break; }
        if (converge) t++;
        if (!t && i == maximum_iter - 2) converge = 1;

        if (t > 1 || i == maximum_iter - 1) {
            memcpy(L_, P_, sizeof L_);
            for (int b = 0; b < 2; b++) {
                int idx = SO3_idx[b];
                double A[9], J[9];
                A_matrix(dx_ + idx, A);
                m3_tr(A, J);
                rows_apply(L_, N23, idx, 3, J, P_);           /* L_ rows <- J * P_ rows */
                for (int c = 0; c < 12; c++) {                /* K_x rows, first 12 columns */
                    double tv[3];
                    for (int r = 0; r < 3; r++) tv[r] = J[r * 3] * K_x[idx * N23 + c] + J[r * 3 + 1] * K_x[(idx + 1) * N23 + c] + J[r * 3 + 2] * K_x[(idx + 2) * N23 + c];
                    for (int r = 0; r < 3; r++) K_x[(idx + r) * N23 + c] = tv[r];
                }
                cols_apply(L_, N23, idx, 3, J);
                cols_apply(P_, N23, idx, 3, J);
            }
            {
                double Nx[6], Mx[6], J2[4];
                s2_Nx_yy(x_->grav, Nx);
                s2_Mx(x_propagated.grav, dx_ + S2_idx, Mx);
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 2; c++) J2[r * 2 + c] = Nx[r * 3] * Mx[0 * 2 + c] + Nx[r * 3 + 1] * Mx[1 * 2 + c] + Nx[r * 3 + 2] * Mx[2 * 2 + c];
                rows_apply(L_, N23, S2_idx, 2, J2, P_);
                for (int c = 0; c < 12; c++) {
                    double t0 = J2[0] * K_x[S2_idx * N23 + c] + J2[1] * K_x[(S2_idx + 1) * N23 + c];
                    double t1 = J2[2] * K_x[S2_idx * N23 + c] + J2[3] * K_x[(S2_idx + 1) * N23 + c];
                    K_x[S2_idx * N23 + c] = t0; K_x[(S2_idx + 1) * N23 + c] = t1;
                }
                cols_apply(L_, N23, S2_idx, 2, J2);
                cols_apply(P_, N23, S2_idx, 2, J2);
            }
            /* P_ = L_ - K_x[:,0:12] * P_[0:12,:] */
            double Pn[N23 * N23];
            for (int r = 0; r < N23; r++)
                for (int c = 0; c < N23; c++) {
                    double s = 0.0;
                    for (int k = 0; k < 12; k++) s += K_x[r * N23 + k] * P_[k * N23 + c];
                    Pn[r * N23 + c] = L_[r * N23 + c] - s;
                }
            memcpy(P_, Pn, sizeof Pn);
            finished = 1;
        }
    }
    for (int k = 0; k < N23; k++)
        if (!isfinite(dx_[k])) status |= 2;
    if (out) {
        memcpy(out->HTH, HTH, sizeof HTH);
        memcpy(out->HTh, HTh, sizeof HTh);
        memcpy(out->dx, dx_, sizeof dx_);
        out->iterations = iters;
        out->searches = 0;
        out->effct_feat_num = neff;
        out->status = status;
    }
    return status;
}

/* ---- the reference's own callback, h_share_model (laserMapping.cpp:961-1093), as the h_dyn_share of the update above */
typedef struct {
    const float *body_xyz;
    int n, nthreads, searches;
    orc_knn_fn knn;
    void *knn_ctx;
    float *world, *nbr, *normvec;
    uint8_t *valid, *sel;
    double *res_last, *h_x, *h;
} hshare_ctx;
static void hshare_cb(void *vctx, orc_state23 *x, int *valid, int *converge, int *rows, const double **h_x, const double **h)
{
    hshare_ctx *c = (hshare_ctx *)vctx;
    (void)valid;
    if (*converge) {                                            /* laserMapping.cpp:994-1013 */
        for (int k = 0; k < c->n; k++) world_point23(x, c->body_xyz + (size_t)k * 3, c->world + (size_t)k * 3);
        c->knn(c->knn_ctx, c->world, c->n, c->nbr, c->valid);
        memcpy(c->sel, c->valid, (size_t)(c->n > 0 ? c->n : 1));
        c->searches++;
    }
    double tr;
    *rows = orc_h_share_model(x, c->body_xyz, c->nbr, c->sel, c->n, c->nthreads, NULL, c->normvec, c->res_last, c->h_x, c->h, &tr);
    *h_x = c->h_x; *h = c->h;
}

int orc_ikfom_update_iterated(orc_state23 *x_, double *P_, const float *body_xyz, int n, double R,
                              int maximum_iter, const double *limit, orc_knn_fn knn, void *knn_ctx,
                              int nthreads, uint8_t *sel_out, float *normvec_out, orc_ikfom_out *out)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    hshare_ctx c;
    c.body_xyz = body_xyz; c.n = n; c.nthreads = nthreads; c.searches = 0; c.knn = knn; c.knn_ctx = knn_ctx;
    c.world = (float *)malloc(sizeof(float) * 3 * nn);
    c.nbr = (float *)calloc(15 * nn, sizeof(float));
    c.valid = (uint8_t *)calloc(nn, 1);
    c.sel = (uint8_t *)calloc(nn, 1);
    c.normvec = (float *)calloc(4 * nn, sizeof(float));
    c.res_last = (double *)calloc(nn, sizeof(double));
    c.h_x = (double *)malloc(sizeof(double) * 12 * nn);
    c.h = (double *)malloc(sizeof(double) * nn);
    const int status = orc_ikfom_update_dyn_share(x_, P_, R, maximum_iter, limit, hshare_cb, &c, out);
    if (out) out->searches = c.searches;
    if (sel_out) memcpy(sel_out, c.sel, nn);
    if (normvec_out) memcpy(normvec_out, c.normvec, sizeof(float) * 4 * nn);
    free(c.world); free(c.nbr); free(c.valid); free(c.sel); free(c.normvec); free(c.res_last); free(c.h_x); free(c.h);
    return status;
}
