// oracle/ref_eigen/eigen_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// C entry points around the REFERENCE's own headers (compiled from /root/reference/include where they lie, see the Makefile):
//   ref_esti_plane        -> esti_plane<float>                       include/common_lib.h:448-493
//   ref_state18_plus      -> StatesGroup::operator+=                 include/common_lib.h:343-352
//   ref_state18_minus     -> StatesGroup::operator-                  include/common_lib.h:354-365
//   ref_so3_exp / _exp_dt -> Exp(v1, v2, v3) / Exp(ang_vel, dt)      include/so3_math.h:54-72 / :33-52
//   ref_so3_log           -> Log                                     include/so3_math.h:76-81
// and, with -DREF_HAVE_MTK (Boost found):
//   ref_state23_boxplus / _boxminus -> state_ikfom::boxplus / boxminus   include/use-ikfom.hpp:12-21, mtk/build_manifold.hpp:192-200
//   ref_A_matrix, ref_S2_Bx / _Nx_yy / _Mx                               mtk/src/mtkmath.hpp:236-247, mtk/types/S2.hpp:215-231,259-280
//   ref_ikfom_update_dyn_share      -> esekf::update_iterated_dyn_share_modified   IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928
// The entry points take the same flat layouts as the oracle's (oracle/fastlivo_oracle.h: orc_state18 fields, orc_state23 = 26
// doubles, row-major matrices) so that tests/test_ref_eigen_cpu.py can hold oracle/orc_*.c to them value for value.
//
// Status (round 4): no Eigen exists in the build container or on the GPU box.  The recipe therefore falls back to shim/ (this
// repository's own small dense-matrix library behind the part of Eigen's API these sources use -- NOT Eigen, see shim/Eigen/Core):
// the part of this file outside REF_HAVE_MTK, the reference's common_lib.h / so3_math.h and the text units of ref_text.sh compile
// and RUN against it here (ref_linalg_kind() == 1), which pins the reference's own logic but not Eigen's arithmetic.  Against a real
// Eigen this file has never been compiled; the REF_HAVE_MTK part additionally needs Boost.Preprocessor and has not met a compiler.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <omp.h>

#include <common_lib.h>          // the reference's (which includes its so3_math.h)
#ifdef REF_HAVE_MTK
#include <use-ikfom.hpp>         // the reference's (IKFoM_toolkit/esekfom/esekfom.hpp and the MTK types)
#endif

// common_lib.h:71-74 declares these extern; the reference defines them in its node sources
M3D Eye3d(M3D::Identity());
M3F Eye3f(M3F::Identity());
V3D Zero3d(0, 0, 0);
V3F Zero3f(0, 0, 0);

namespace {

void load18(StatesGroup &s, const double *rot9, const double *v15)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) s.rot_end(i, j) = rot9[i * 3 + j];
    for (int i = 0; i < 3; i++) {
        s.pos_end(i) = v15[i];
        s.vel_end(i) = v15[3 + i];
        s.bias_g(i) = v15[6 + i];
        s.bias_a(i) = v15[9 + i];
        s.gravity(i) = v15[12 + i];
    }
}

void store18(const StatesGroup &s, double *rot9, double *v15)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) rot9[i * 3 + j] = s.rot_end(i, j);
    for (int i = 0; i < 3; i++) {
        v15[i] = s.pos_end(i);
        v15[3 + i] = s.vel_end(i);
        v15[6 + i] = s.bias_g(i);
        v15[9 + i] = s.bias_a(i);
        v15[12 + i] = s.gravity(i);
    }
}

}  // namespace

extern "C" {

int ref_have_mtk(void)
{
#ifdef REF_HAVE_MTK
    return 1;
#else
    return 0;
#endif
}

// 0 = compiled against a real Eigen, 1 = against oracle/ref_eigen/shim (this repository's stand-in behind Eigen's API: the
// reference's text runs, Eigen's own arithmetic does not)
int ref_linalg_kind(void)
{
#ifdef REF_LINALG_SHIM
    return 1;
#else
    return 0;
#endif
}

const char *ref_eigen_version(void)
{
    static char v[32];
    std::snprintf(v, sizeof v, "%d.%d.%d", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
    return v;
}

/* near: 5 x 3 floats; pabcd: 4 floats (written also when the fit is rejected, like the reference's pca_result); returns the bool */
int ref_esti_plane(const float *near, float threshold, float *pabcd)
{
    PointVector pts(NUM_MATCH_POINTS);
    for (int j = 0; j < NUM_MATCH_POINTS; j++) {
        pts[j].x = near[3 * j];
        pts[j].y = near[3 * j + 1];
        pts[j].z = near[3 * j + 2];
    }
    Matrix<float, 4, 1> r;
    r.setZero();
    const float thr = threshold;
    const bool ok = esti_plane<float>(r, pts, thr);
    for (int k = 0; k < 4; k++) pabcd[k] = r(k);
    return ok ? 1 : 0;
}

void ref_state18_plus(double *rot9, double *v15, const double *d18)
{
    StatesGroup s;
    load18(s, rot9, v15);
    Matrix<double, DIM_STATE, 1> d;
    for (int i = 0; i < DIM_STATE; i++) d(i) = d18[i];
    s += d;
    store18(s, rot9, v15);
}

void ref_state18_minus(const double *rot_a, const double *v15_a, const double *rot_b, const double *v15_b, double *out18)
{
    StatesGroup a, b;
    load18(a, rot_a, v15_a);
    load18(b, rot_b, v15_b);
    Matrix<double, DIM_STATE, 1> d = a - b;
    for (int i = 0; i < DIM_STATE; i++) out18[i] = d(i);
}

void ref_so3_exp(const double *v, double *R9)
{
    const double v1 = v[0], v2 = v[1], v3 = v[2];
    M3D R = Exp<double>(v1, v2, v3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R9[i * 3 + j] = R(i, j);
}

void ref_so3_exp_dt(const double *ang_vel, double dt, double *R9)
{
    const V3D w(ang_vel[0], ang_vel[1], ang_vel[2]);
    M3D R = Exp<double, double>(w, dt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R9[i * 3 + j] = R(i, j);
}

void ref_so3_log(const double *R9, double *out3)
{
    M3D R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R(i, j) = R9[i * 3 + j];
    V3D l = Log<double>(R);
    for (int i = 0; i < 3; i++) out3[i] = l(i);
}

#ifdef REF_HAVE_MTK
}  // extern "C"

namespace {

/* 26 doubles: pos 3, rot (x y z w) 4, offset_R_L_I (x y z w) 4, offset_T_L_I 3, vel 3, bg 3, ba 3, grav 3  (= orc_state23) */
void load23(state_ikfom &s, const double *p)
{
    for (int i = 0; i < 3; i++) s.pos[i] = p[i];
    s.rot = SO3(Eigen::Quaterniond(p[6], p[3], p[4], p[5]));                 // (w, x, y, z); SO3(const base&) does not normalise
    s.offset_R_L_I = SO3(Eigen::Quaterniond(p[10], p[7], p[8], p[9]));
    for (int i = 0; i < 3; i++) {
        s.offset_T_L_I[i] = p[11 + i];
        s.vel[i] = p[14 + i];
        s.bg[i] = p[17 + i];
        s.ba[i] = p[20 + i];
        s.grav.vec[i] = p[23 + i];                                            // raw: the S2 constructors would renormalise
    }
}

void store23(const state_ikfom &s, double *p)
{
    for (int i = 0; i < 3; i++) p[i] = s.pos[i];
    p[3] = s.rot.x(); p[4] = s.rot.y(); p[5] = s.rot.z(); p[6] = s.rot.w();
    p[7] = s.offset_R_L_I.x(); p[8] = s.offset_R_L_I.y(); p[9] = s.offset_R_L_I.z(); p[10] = s.offset_R_L_I.w();
    for (int i = 0; i < 3; i++) {
        p[11 + i] = s.offset_T_L_I[i];
        p[14 + i] = s.vel[i];
        p[17 + i] = s.bg[i];
        p[20 + i] = s.ba[i];
        p[23 + i] = s.grav.vec[i];
    }
}

typedef void (*ref_h_fn)(void *ctx, double *state26, int *valid, int *converge, int *rows, const double **h_x, const double **h);
ref_h_fn g_h_fn = nullptr;
void *g_h_ctx = nullptr;
int g_calls = 0;

/* measurementModel_dyn_share (esekfom.hpp:129): plain function, so the C callback rides in globals (one update at a time) */
void h_trampoline(state_ikfom &s, esekfom::dyn_share_datastruct<double> &d)
{
    double st[26];
    store23(s, st);
    int valid = d.valid ? 1 : 0, converge = d.converge ? 1 : 0, rows = 0;
    const double *hx = nullptr, *h = nullptr;
    g_h_fn(g_h_ctx, st, &valid, &converge, &rows, &hx, &h);
    g_calls++;
    d.valid = valid != 0;
    d.h_x = Eigen::MatrixXd::Zero(rows, 12);
    d.h.resize(rows);
    for (int i = 0; i < rows; i++) {
        for (int j = 0; j < 12; j++) d.h_x(i, j) = hx[(size_t)i * 12 + j];
        d.h(i) = h[i];
    }
}

}  // namespace

extern "C" {

void ref_state23_boxplus(double *state26, const double *dx23)
{
    state_ikfom s;
    load23(s, state26);
    Eigen::Matrix<double, 23, 1> d;
    for (int i = 0; i < 23; i++) d(i) = dx23[i];
    s.boxplus(d);
    store23(s, state26);
}

void ref_state23_boxminus(const double *state26, const double *other26, double *dx23)
{
    state_ikfom a, b;
    load23(a, state26);
    load23(b, other26);
    Eigen::Matrix<double, 23, 1> d;
    a.boxminus(d, b);
    for (int i = 0; i < 23; i++) dx23[i] = d(i);
}

void ref_A_matrix(const double *v, double *A9)
{
    vect3 w;                                   // MTK::vect: A_matrix wants Base::scalar (mtkmath.hpp:236)
    for (int i = 0; i < 3; i++) w[i] = v[i];
    Eigen::Matrix3d A = MTK::A_matrix(w);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A9[i * 3 + j] = A(i, j);
}

void ref_S2_Bx(const double *grav, double *Bx6 /* 3 x 2 row-major */)
{
    S2 g;
    for (int i = 0; i < 3; i++) g.vec[i] = grav[i];
    Eigen::Matrix<double, 3, 2> B;
    g.S2_Bx(B);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) Bx6[i * 2 + j] = B(i, j);
}

void ref_S2_Nx_yy(const double *grav, double *Nx6 /* 2 x 3 row-major */)
{
    S2 g;
    for (int i = 0; i < 3; i++) g.vec[i] = grav[i];
    Eigen::Matrix<double, 2, 3> Nx;
    g.S2_Nx_yy(Nx);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) Nx6[i * 3 + j] = Nx(i, j);
}

void ref_S2_Mx(const double *grav, const double *delta2, double *Mx6 /* 3 x 2 row-major */)
{
    S2 g;
    for (int i = 0; i < 3; i++) g.vec[i] = grav[i];
    MTK::vect<2, double> d;                    // the type esekfom.hpp:1675 passes (seg_S2)
    d[0] = delta2[0];
    d[1] = delta2[1];
    Eigen::Matrix<double, 3, 2> M;
    g.S2_Mx(M, d);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) Mx6[i * 2 + j] = M(i, j);
}

/* The unmodified updater around a C measurement callback.  state26 / P (23 x 23 row-major) in and out; returns the number of
 * callback invocations. */
int ref_ikfom_update_dyn_share(double *state26, double *P, double R, int maximum_iter, const double *limit23, ref_h_fn h_fn, void *h_ctx)
{
    typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
    kf_t kf;
    double limit[23];
    for (int i = 0; i < 23; i++) limit[i] = limit23[i];
    g_h_fn = h_fn;
    g_h_ctx = h_ctx;
    g_calls = 0;
    kf.init_dyn_share(get_f, df_dx, df_dw, h_trampoline, maximum_iter, limit);
    state_ikfom x;
    load23(x, state26);
    kf.change_x(x);
    kf_t::cov Pm;
    for (int i = 0; i < 23; i++)
        for (int j = 0; j < 23; j++) Pm(i, j) = P[i * 23 + j];
    kf.change_P(Pm);
    double solve_time = 0.0;
    kf.update_iterated_dyn_share_modified(R, solve_time);
    store23(kf.get_x(), state26);
    const kf_t::cov &Po = kf.get_P();
    for (int i = 0; i < 23; i++)
        for (int j = 0; j < 23; j++) P[i * 23 + j] = Po(i, j);
    return g_calls;
}
#endif  /* REF_HAVE_MTK */

}  // extern "C"
