/*
 * oracle/orc_lio_common.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Per-point LiDAR measurement model shared by the Mode-18 and Mode-23 restatements:
 * plane fit from 5 map neighbours and the point-to-plane residual with its gates.
 *
 * Reference lines restated:
 *   esti_plane<float>            /root/reference/include/common_lib.h:448-493
 *   A.colPivHouseholderQr().solve(b)   Eigen 3.3 ColPivHouseholderQR (third party, unpinned
 *                                >=3.3.4, README.md:52): computeInPlace() + _solve_impl(),
 *                                Householder.h makeHouseholder()/applyHouseholderOnTheLeft().
 *                                Restated from the published algorithm; reductions evaluated
 *                                sequentially (Eigen's SIMD reduction order cannot be pinned here).
 *   residual + gates             /root/reference/src/laserMapping.cpp:1573-1584 (= :1023-1034)
 */
#ifndef ORC_LIO_COMMON_H
#define ORC_LIO_COMMON_H

#include <float.h>
#include <math.h>
#include <stdint.h>

#define ORC_NUM_MATCH_POINTS 5 /* common_lib.h:39 */

/* How a reduction over <= 5 float terms is associated. 0 = left to right (what every parity test and the device use). 1 and 2 model
 * what an x86 Eigen build may do instead for these short fixed-size reductions -- one SSE packet of four reduced horizontally, then the
 * remaining terms: 1 = ((t0 + t2) + (t1 + t3)) + rest (movehl-style predux), 2 = ((t0 + t1) + (t2 + t3)) + rest (hadd-style). Used ONLY by
 * the sensitivity report (orc_plane_sensitivity, tests/test_qr_sensitivity_cpu.py): Eigen is not available here, so which one a real
 * build takes cannot be pinned -- the report bounds what that uncertainty can change. */
static inline float orc_sum_terms(const float *t, int n, int mode)
{
    if (mode == 0 || n < 4) {
        float s = 0.f;
        for (int i = 0; i < n; i++) s += t[i];
        return s;
    }
    float s = (mode == 1) ? ((t[0] + t[2]) + (t[1] + t[3])) : ((t[0] + t[1]) + (t[2] + t[3]));
    for (int i = 4; i < n; i++) s += t[i];
    return s;
}

/* Eigen ColPivHouseholderQR<Matrix<float,5,3>>::solve(b) with b = -1 (common_lib.h:451-463). */
static inline void orc_colpiv_qr_solve_5x3_mode(const float *near /*5x3 row-major*/, float *x /*3*/, int mode)
{
    enum { ROWS = 5, COLS = 3 };
    float qr[ROWS][COLS];
    float hc[COLS];
    float nrmU[COLS], nrmD[COLS];
    int trans[COLS];
    for (int r = 0; r < ROWS; r++)
        for (int c = 0; c < COLS; c++) qr[r][c] = near[r * 3 + c];

    for (int k = 0; k < COLS; k++) {
        float tt[ROWS];
        for (int r = 0; r < ROWS; r++) tt[r] = qr[r][k] * qr[r][k];
        const float s = orc_sum_terms(tt, ROWS, mode);
        nrmD[k] = sqrtf(s);
        nrmU[k] = nrmD[k];
    }
    float mx = nrmU[0];
    for (int k = 1; k < COLS; k++) if (nrmU[k] > mx) mx = nrmU[k];
    const float th = mx * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)ROWS;
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int nonzero_pivots = COLS;

    for (int k = 0; k < COLS; k++) {
        /* column with the largest remaining norm (first maximum) */
        int big = k;
        float bigv = nrmU[k];
        for (int j = k + 1; j < COLS; j++) if (nrmU[j] > bigv) { bigv = nrmU[j]; big = j; }
        float big_sq = bigv * bigv;
        if (nonzero_pivots == COLS && big_sq < threshold_helper * (float)(ROWS - k)) nonzero_pivots = k;
        trans[k] = big;
        if (k != big) {
            for (int r = 0; r < ROWS; r++) { float t = qr[r][k]; qr[r][k] = qr[r][big]; qr[r][big] = t; }
            float t = nrmU[k]; nrmU[k] = nrmU[big]; nrmU[big] = t;
            t = nrmD[k]; nrmD[k] = nrmD[big]; nrmD[big] = t;
        }
        /* makeHouseholderInPlace on qr[k..,k] */
        float c0 = qr[k][k];
        float tq[ROWS];
        for (int r = k + 1; r < ROWS; r++) tq[r - k - 1] = qr[r][k] * qr[r][k];
        float tailSq = orc_sum_terms(tq, ROWS - k - 1, mode);
        float tau, beta;
        if (tailSq <= FLT_MIN) {
            tau = 0.f; beta = c0;
            for (int r = k + 1; r < ROWS; r++) qr[r][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
            for (int r = k + 1; r < ROWS; r++) qr[r][k] = qr[r][k] / den;
            tau = (beta - c0) / beta;
        }
        qr[k][k] = beta;
        hc[k] = tau;
        /* apply H_k to the trailing columns */
        if (tau != 0.f) {
            for (int j = k + 1; j < COLS; j++) {
                float td[ROWS];
                for (int r = k + 1; r < ROWS; r++) td[r - k - 1] = qr[r][k] * qr[r][j];
                float tmp = orc_sum_terms(td, ROWS - k - 1, mode);
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
                for (int r = k + 1; r < ROWS; r++) qr[r][j] -= (tau * qr[r][k]) * tmp;
            }
        }
        /* LAPACK WN176 column-norm downdate */
        for (int j = k + 1; j < COLS; j++) {
            if (nrmU[j] != 0.f) {
                float temp = fabsf(qr[k][j]) / nrmU[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float ratio = nrmU[j] / nrmD[j];
                float temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    float tn[ROWS];
                    for (int r = k + 1; r < ROWS; r++) tn[r - k - 1] = qr[r][j] * qr[r][j];
                    const float s = orc_sum_terms(tn, ROWS - k - 1, mode);
                    nrmD[j] = sqrtf(s);
                    nrmU[j] = nrmD[j];
                } else {
                    nrmU[j] *= sqrtf(temp);
                }
            }
        }
    }

    /* _solve_impl: c = Q^T b, back-substitute on R, undo the column permutation */
    float c[ROWS];
    for (int r = 0; r < ROWS; r++) c[r] = -1.0f;
    x[0] = x[1] = x[2] = 0.f;
    if (nonzero_pivots == 0) return;
    for (int k = 0; k < nonzero_pivots; k++) {
        float tau = hc[k];
        if (tau != 0.f) {
            float tc[ROWS];
            for (int r = k + 1; r < ROWS; r++) tc[r - k - 1] = qr[r][k] * c[r];
            float tmp = orc_sum_terms(tc, ROWS - k - 1, mode);
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int r = k + 1; r < ROWS; r++) c[r] -= (tau * qr[r][k]) * tmp;
        }
    }
    for (int i = nonzero_pivots - 1; i >= 0; i--) {
        c[i] = c[i] / qr[i][i];
        for (int r = 0; r < i; r++) c[r] -= c[i] * qr[r][i];
    }
    int perm[COLS] = {0, 1, 2};
    for (int k = 0; k < COLS; k++) { int t = perm[k]; perm[k] = perm[trans[k]]; perm[trans[k]] = t; }
    for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = c[i];
}

static inline void orc_colpiv_qr_solve_5x3(const float *near, float *x) { orc_colpiv_qr_solve_5x3_mode(near, x, 0); }

/* common_lib.h:448-493 with threshold as passed (0.1f at both call sites). */
static inline int orc_esti_plane_mode(const float *near /*5x3*/, float threshold, float *pabcd /*4*/, int mode)
{
    float nv[3];
    orc_colpiv_qr_solve_5x3_mode(near, nv, mode);
    float n = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    pabcd[0] = nv[0] / n;
    pabcd[1] = nv[1] / n;
    pabcd[2] = nv[2] / n;
    pabcd[3] = (float)(1.0 / (double)n);
    for (int j = 0; j < ORC_NUM_MATCH_POINTS; j++) {
        float v = pabcd[0] * near[j * 3 + 0] + pabcd[1] * near[j * 3 + 1] + pabcd[2] * near[j * 3 + 2] + pabcd[3];
        if (fabsf(v) > threshold) return 0;
    }
    return 1;
}
static inline int orc_esti_plane(const float *near, float threshold, float *pabcd) { return orc_esti_plane_mode(near, threshold, pabcd, 0); }

/* laserMapping.cpp:1569-1585: on entry *sel is point_selected_surf[i] (already AND-ed with the
 * kNN validity on search passes). Returns the new point_selected_surf[i]; on selection fills
 * normvec[0..3] = (n, pd2) and *res = |pd2|. p_w is the FLOAT world point (PointType store). */
static inline int orc_point_residual(const float *near, const float *p_w, const double *p_b,
                                     float *normvec, double *res)
{
    float pabcd[4];
    if (!orc_esti_plane(near, 0.1f, pabcd)) return 0;
    float pd2 = pabcd[0] * p_w[0] + pabcd[1] * p_w[1] + pabcd[2] * p_w[2] + pabcd[3];
    double pbn = sqrt(p_b[0] * p_b[0] + p_b[1] * p_b[1] + p_b[2] * p_b[2]);
    float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(pbn));
    if ((double)s > 0.9) {
        normvec[0] = pabcd[0]; normvec[1] = pabcd[1]; normvec[2] = pabcd[2]; normvec[3] = pd2;
        *res = (double)fabsf(pd2);
        return 1;
    }
    return 0;
}

#endif /* ORC_LIO_COMMON_H */
