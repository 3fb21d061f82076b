/*
 * oracle/orc_lio18.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Mode-18 LiDAR ESKF iteration: the loop the shipped FAST-LIVO binary runs (USE_IKFOM is
 * commented out, /root/reference/include/common_lib.h:28).
 *
 * Reference lines restated (file:line under /root/reference):
 *   per-point residual loop [A]        src/laserMapping.cpp:1516-1586
 *   pointBodyToWorld                   src/laserMapping.cpp:272-286
 *   compaction [B]                     src/laserMapping.cpp:1588-1602
 *   H rows [C]                         src/laserMapping.cpp:1608-1629
 *   gain solve [D]                     src/laserMapping.cpp:1664-1695
 *   rematch / stop / covariance [E]    src/laserMapping.cpp:1700-1731
 *   StatesGroup += / -                 include/common_lib.h:343-365
 * No reference tests exist.  Held to the reference's own text of that loop since round 4 (oracle/ref_eigen, tests/test_ref_eigen_cpu.py:
 * bit for bit over a stand-in for Eigen); Eigen's own arithmetic stays unpinned -- see fastlivo_oracle.h.
 * OpenMP is used exactly where the reference uses it: over points in [A] only.
 */
#include "fastlivo_oracle.h"
#include "orc_lio_common.h"
#include "orc_math.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* StatesGroup::operator- (common_lib.h:354-365): a = this - b */
static void state18_minus(const orc_state18 *a, const orc_state18 *b, double *out)
{
    double bt[9], rotd[9];
    m3_tr(b->rot, bt);
    m3_mul(bt, a->rot, rotd);
    so3_Log(rotd, out);
    for (int i = 0; i < 3; i++) {
        out[3 + i] = a->pos[i] - b->pos[i];
        out[6 + i] = a->vel[i] - b->vel[i];
        out[9 + i] = a->bg[i] - b->bg[i];
        out[12 + i] = a->ba[i] - b->ba[i];
        out[15 + i] = a->grav[i] - b->grav[i];
    }
}
/* StatesGroup::operator+= (common_lib.h:343-352) */
static void state18_plus(orc_state18 *x, const double *d)
{
    double E[9];
    so3_Exp(d[0], d[1], d[2], E);
    m3_mul(x->rot, E, x->rot);
    for (int i = 0; i < 3; i++) {
        x->pos[i] += d[3 + i];
        x->vel[i] += d[6 + i];
        x->bg[i] += d[9 + i];
        x->ba[i] += d[12 + i];
        x->grav[i] += d[15 + i];
    }
}

/* The 18x18 solve shared by LIO (sign=+1) and VIO (sign=-1):
 *   laserMapping.cpp:1664-1683 / lidar_selection.cpp:871-879. */
int orc_solve18(orc_state18 *x, const orc_state18 *x_prop, const double *HTH6, const double *HTz6,
                double meas_cov, double sign, double *G, double *solution)
{
    enum { N = 18 };
    double H_T_H[N * N], A[N * N], Ainv[N * N], M[N * N], K1[N * N];
    int st = 0;
    memset(H_T_H, 0, sizeof H_T_H);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) H_T_H[i * N + j] = HTH6[i * 6 + j];
    for (int i = 0; i < N * N; i++) A[i] = x->cov[i] / meas_cov;
    st |= orc_inverse(N, A, Ainv);
    for (int i = 0; i < N * N; i++) M[i] = H_T_H[i] + Ainv[i];
    st |= orc_inverse(N, M, K1);
    /* G.block<18,6>(0,0) = K_1.block<18,6>(0,0) * H_T_H.block<6,6>(0,0) */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += K1[i * N + k] * HTH6[k * 6 + j];
            G[i * N + j] = s;
        }
    double vec[N];
    state18_minus(x_prop, x, vec);
    for (int i = 0; i < N; i++) {
        double kh = 0.0, gv = 0.0;
        for (int k = 0; k < 6; k++) kh += K1[i * N + k] * HTz6[k];
        for (int k = 0; k < 6; k++) gv += G[i * N + k] * vec[k];
        solution[i] = sign * kh + vec[i] - gv;
    }
    state18_plus(x, solution);
    for (int i = 0; i < N; i++)
        if (!isfinite(solution[i])) return 2;
    return st;
}

int orc_lio18_iterate(orc_state18 *x, const orc_state18 *x_prop, const float *body_xyz,
                      const float *nbr_xyz, uint8_t *sel, int n, const double *R_LI,
                      const double *t_LI, double laser_point_cov, int nthreads, float *world_xyz,
                      float *normvec, double *res_last, double *G, orc_lio18_iter_out *out)
{
    /* [A] laserMapping.cpp:1516-1586 */
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for
#endif
    for (int i = 0; i < n; i++) {
        double p_body[3] = {(double)body_xyz[i * 3], (double)body_xyz[i * 3 + 1], (double)body_xyz[i * 3 + 2]};
        double p_i[3], p_g[3];
        m3_vec(R_LI, p_body, p_i);
        p_i[0] += t_LI[0]; p_i[1] += t_LI[1]; p_i[2] += t_LI[2];
        m3_vec(x->rot, p_i, p_g);
        float pw[3] = {(float)(p_g[0] + x->pos[0]), (float)(p_g[1] + x->pos[1]), (float)(p_g[2] + x->pos[2])};
        if (world_xyz) { world_xyz[i * 3] = pw[0]; world_xyz[i * 3 + 1] = pw[1]; world_xyz[i * 3 + 2] = pw[2]; }
        if (!sel[i]) continue;
        sel[i] = (uint8_t)orc_point_residual(nbr_xyz + (size_t)i * 15, pw, p_body, normvec + (size_t)i * 4, res_last + i);
    }

    /* [B] laserMapping.cpp:1588-1602 (serial, order preserving) */
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int neff = 0;
    double total_residual = 0.0;
    for (int i = 0; i < n; i++) {
        if (sel[i] && res_last[i] <= 2.0) {
            idx[neff++] = i;
            total_residual += res_last[i];
        }
    }

    /* [C] laserMapping.cpp:1608-1629 : Hsub (neff x 6) and meas_vec */
    double *Hsub = (double *)malloc(sizeof(double) * 6 * (size_t)(neff > 0 ? neff : 1));
    double *meas = (double *)malloc(sizeof(double) * (size_t)(neff > 0 ? neff : 1));
    double Rt[9];
    m3_tr(x->rot, Rt);
    for (int k = 0; k < neff; k++) {
        int i = idx[k];
        double pt[3] = {(double)body_xyz[i * 3], (double)body_xyz[i * 3 + 1], (double)body_xyz[i * 3 + 2]};
        double p_this[3];
        m3_vec(R_LI, pt, p_this);
        p_this[0] += t_LI[0]; p_this[1] += t_LI[1]; p_this[2] += t_LI[2];
        double cm[9], nv[3] = {(double)normvec[i * 4], (double)normvec[i * 4 + 1], (double)normvec[i * 4 + 2]};
        skew3(p_this, cm);
        double M1[9], Av[3];
        m3_mul(cm, Rt, M1);  /* point_crossmat * state.rot_end.transpose() */
        m3_vec(M1, nv, Av);  /* ... * norm_vec */
        Hsub[k * 6 + 0] = Av[0]; Hsub[k * 6 + 1] = Av[1]; Hsub[k * 6 + 2] = Av[2];
        Hsub[k * 6 + 3] = nv[0]; Hsub[k * 6 + 4] = nv[1]; Hsub[k * 6 + 5] = nv[2];
        meas[k] = -(double)normvec[i * 4 + 3];
    }

    /* [D] laserMapping.cpp:1664-1695 */
    double HTH[36], HTz[6];
    for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) {
            double s = 0.0;
            for (int k = 0; k < neff; k++) s += Hsub[k * 6 + a] * Hsub[k * 6 + b];
            HTH[a * 6 + b] = s;
        }
        double s = 0.0;
        for (int k = 0; k < neff; k++) s += Hsub[k * 6 + a] * meas[k];
        HTz[a] = s;
    }
    double solution[18];
    int st = orc_solve18(x, x_prop, HTH, HTz, laser_point_cov, 1.0, G, solution);
    double rn = sqrt(solution[0] * solution[0] + solution[1] * solution[1] + solution[2] * solution[2]);
    double tn = sqrt(solution[3] * solution[3] + solution[4] * solution[4] + solution[5] * solution[5]);
    int converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);

    if (out) {
        memcpy(out->HTH, HTH, sizeof HTH);
        memcpy(out->HTz, HTz, sizeof HTz);
        memcpy(out->solution, solution, sizeof solution);
        out->total_residual = total_residual;
        out->effct_feat_num = neff;
        out->converged = converged;
        out->status = st;
        out->pad = 0;
    }
    free(Hsub); free(meas); free(idx);
    return st;
}

int orc_lio18_frame(orc_state18 *x, const float *body_xyz, int n, const double *R_LI,
                    const double *t_LI, double laser_point_cov, int max_iterations, orc_knn_fn knn,
                    void *knn_ctx, int nthreads, uint8_t *sel_out, float *normvec_out,
                    orc_lio_frame_out *out)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    orc_state18 prop = *x;                      /* state_propagat = state (laserMapping.cpp:1292) */
    float *world = (float *)malloc(sizeof(float) * 3 * nn);
    float *nbr = (float *)calloc(15 * nn, sizeof(float));
    uint8_t *valid = (uint8_t *)calloc(nn, 1);
    uint8_t *sel = (uint8_t *)calloc(nn, 1);
    float *normvec = (float *)calloc(4 * nn, sizeof(float));
    double *res_last = (double *)calloc(nn, sizeof(double));
    double G[18 * 18];
    memset(G, 0, sizeof G);                     /* laserMapping.cpp:1227 */
    int rematch_num = 0, nearest_search_en = 1; /* laserMapping.cpp:1472-1473 */
    int iters = 0, searches = 0, st = 0;
    orc_lio18_iter_out io;
    memset(&io, 0, sizeof io);

    for (int iterCount = -1; iterCount < max_iterations; iterCount++) {
        if (nearest_search_en) {
            for (int i = 0; i < n; i++) {       /* pointBodyToWorld at the current state */
                double pb[3] = {(double)body_xyz[i * 3], (double)body_xyz[i * 3 + 1], (double)body_xyz[i * 3 + 2]};
                double pi[3], pg[3];
                m3_vec(R_LI, pb, pi);
                pi[0] += t_LI[0]; pi[1] += t_LI[1]; pi[2] += t_LI[2];
                m3_vec(x->rot, pi, pg);
                world[i * 3] = (float)(pg[0] + x->pos[0]);
                world[i * 3 + 1] = (float)(pg[1] + x->pos[1]);
                world[i * 3 + 2] = (float)(pg[2] + x->pos[2]);
            }
            knn(knn_ctx, world, n, nbr, valid);
            memcpy(sel, valid, nn);             /* point_selected_surf[i] = sq[4] <= 5 (:1549) */
            searches++;
        }
        st |= orc_lio18_iterate(x, &prop, body_xyz, nbr, sel, n, R_LI, t_LI, laser_point_cov, nthreads,
                                NULL, normvec, res_last, G, &io);
        iters++;
        int EKF_stop_flg = 0;
        /* Rematch judgement, laserMapping.cpp:1700-1705 */
        nearest_search_en = 0;
        if (io.converged || ((rematch_num == 0) && (iterCount == (max_iterations - 2)))) {
            nearest_search_en = 1;
            rematch_num++;
        }
        /* Convergence judgement and covariance update, :1708-1728 */
        if (!EKF_stop_flg && (rematch_num >= 2 || (iterCount == max_iterations - 1))) {
            double IG[18 * 18], Pn[18 * 18];
            for (int i = 0; i < 18; i++)
                for (int j = 0; j < 18; j++) IG[i * 18 + j] = ((i == j) ? 1.0 : 0.0) - G[i * 18 + j];
            for (int i = 0; i < 18; i++)
                for (int j = 0; j < 18; j++) {
                    double s = 0.0;
                    for (int k = 0; k < 18; k++) s += IG[i * 18 + k] * x->cov[k * 18 + j];
                    Pn[i * 18 + j] = s;
                }
            memcpy(x->cov, Pn, sizeof Pn);
            EKF_stop_flg = 1;
        }
        if (EKF_stop_flg) break;
    }
    if (sel_out) memcpy(sel_out, sel, nn);
    if (normvec_out) memcpy(normvec_out, normvec, sizeof(float) * 4 * nn);
    if (out) {
        out->iterations = iters;
        out->searches = searches;
        out->effct_feat_num = io.effct_feat_num;
        out->converged_last = io.converged;
        out->total_residual = io.total_residual;
    }
    free(world); free(nbr); free(valid); free(sel); free(normvec); free(res_last);
    return st;
}


/* Sensitivity of the plane fit and of the gates behind it to the association order of the QR's short float reductions (orc_sum_terms,
 * orc_lio_common.h): per point the plane under `mode`, the planarity verdict, and the first-pass gate results at the given float world
 * points (laserMapping.cpp:1571-1584,1593).  Report only -- nothing in the product or in the parity tests uses a mode other than 0. */
int orc_plane_sensitivity(const float *nbr_xyz /* n x 5 x 3 */, const float *world_xyz /* n x 3 */, const float *body_xyz /* n x 3 */, int n,
                          int mode, float *plane_out /* n x 4 */, uint8_t *planar_out, uint8_t *sel_out, uint8_t *eff_out)
{
    for (int i = 0; i < n; i++) {
        float pabcd[4] = {0.f, 0.f, 0.f, 0.f};
        const int ok = orc_esti_plane_mode(nbr_xyz + (size_t)i * 15, 0.1f, pabcd, mode);
        for (int k = 0; k < 4; k++) plane_out[(size_t)i * 4 + k] = pabcd[k];
        planar_out[i] = (uint8_t)ok;
        const float *pw = world_xyz + (size_t)i * 3;
        const double b0 = body_xyz[i * 3], b1 = body_xyz[i * 3 + 1], b2 = body_xyz[i * 3 + 2];
        const float pd2 = pabcd[0] * pw[0] + pabcd[1] * pw[1] + pabcd[2] * pw[2] + pabcd[3];
        const float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(sqrt(b0 * b0 + b1 * b1 + b2 * b2)));
        const int sel = ok && ((double)s > 0.9);
        sel_out[i] = (uint8_t)sel;
        eff_out[i] = (uint8_t)(sel && ((double)fabsf(pd2) <= 2.0));
    }
    return 0;
}

/* ---- unit entry points for tests/test_cross_oracle_cpu.py and tests/test_ref_eigen_cpu.py (oracle/ref_eigen pins these four to the
 * reference's own headers wherever Eigen exists): the restatements above, one call each. */
int orc_unit_esti_plane(const float *near /*5x3*/, float threshold, float *pabcd /*4*/) { return orc_esti_plane(near, threshold, pabcd); }
void orc_unit_state18_plus(orc_state18 *x, const double *d /*18*/) { state18_plus(x, d); }
void orc_unit_state18_minus(const orc_state18 *a, const orc_state18 *b, double *out /*18*/) { state18_minus(a, b, out); }
void orc_unit_so3_exp(const double *v /*3*/, double *R /*9 row-major*/) { so3_Exp(v[0], v[1], v[2], R); }
void orc_unit_so3_log(const double *R /*9 row-major*/, double *out /*3*/) { so3_Log(R, out); }
