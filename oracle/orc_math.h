/*
 * oracle/orc_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Small fixed-size linear algebra used by the CPU restatement of FAST-LIVO's ESKF hot path.
 * The reference uses Eigen for all of this; Eigen is not available in this container, so the
 * operations it performs are restated here in plain C with a fixed, documented evaluation
 * order (left-to-right dot products, unblocked partial-pivot LU). The reference's own lines (Exp, Log) are held to their text
 * since round 4 (oracle/ref_eigen); Eigen's own arithmetic is unpinned: the
 * reference has no tests/golden vectors and cannot be built here.
 *
 * Reference lines restated:
 *   Exp(v1,v2,v3)            /root/reference/include/so3_math.h:54-72
 *   Log(R)                   /root/reference/include/so3_math.h:75-81
 *   SKEW_SYM_MATRX           /root/reference/include/so3_math.h:9
 *   Matrix::inverse() (N>4)  Eigen PartialPivLU (third party, unpinned >=3.3.4, README.md:52)
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <string.h>

/* ---- 3-vectors / 3x3 (row-major) in double ------------------------------------------------ */
static inline void m3_mul(const double *A, const double *B, double *C) /* C = A*B */
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, T, sizeof T);
}
static inline void m3_tr(const double *A, double *At)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
    memcpy(At, T, sizeof T);
}
static inline void m3_vec(const double *A, const double *v, double *o) /* o = A*v */
{
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static inline void m3t_vec(const double *A, const double *v, double *o) /* o = A^T*v */
{
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static inline void skew3(const double *v, double *K) /* so3_math.h:9 */
{
    K[0] = 0.0;   K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2];  K[4] = 0.0;   K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0];  K[8] = 0.0;
}
static inline double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* so3_math.h:54-72 : Rodrigues with threshold 1e-5 on the norm.  The reference writes `Eye3 + std::sin(norm) * K + (1.0 - std::cos(norm)) * K * K`,
 * which C++ groups as (Eye3 + s*K) + ((c*K)*K): the scalar goes into the LEFT factor before the product (for 3x3 fixed-size operands
 * Eigen evaluates exactly that, coefficient by coefficient).  Until round 4 this function computed c*(K*K) instead -- one unit in the last place
 * away in 14 % of random rotations, found by running the reference's own text (oracle/ref_eigen, tests/test_ref_eigen_cpu.py). */
static inline void so3_Exp(double v1, double v2, double v3, double *R)
{
    double nrm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (nrm > 0.00001) {
        double r[3] = {v1 / nrm, v2 / nrm, v3 / nrm};
        double K[9], cK[9], cKK[9];
        skew3(r, K);
        double s = sin(nrm), c = 1.0 - cos(nrm);
        for (int i = 0; i < 9; i++) cK[i] = c * K[i];
        m3_mul(cK, K, cKK);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + cKK[i];
    }
}
/* so3_math.h:75-81 */
static inline void so3_Log(const double *R, double *out)
{
    double tr = R[0] + R[4] + R[8];
    double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
    double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double f = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
    out[0] = f * K[0]; out[1] = f * K[1]; out[2] = f * K[2];
}

/* ---- dense n x n (row-major, n <= ORC_MAXN) ------------------------------------------------- */
#define ORC_MAXN 32

/* Inverse through partial-pivot LU then solving for the identity, the algorithm behind Eigen's
 * Matrix::inverse() for N>4 (PartialPivLU::inverse). Unblocked right-looking elimination.
 * Returns 0 on success, 1 if a zero pivot was met (Eigen would silently produce inf/nan). */
static inline int orc_inverse(int n, const double *A, double *Ainv)
{
    double LU[ORC_MAXN * ORC_MAXN];
    int perm[ORC_MAXN];
    int status = 0;
    memcpy(LU, A, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(LU[k * n + k]);
        for (int r = k + 1; r < n; r++) {
            double v = fabs(LU[r * n + k]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0.0) { status = 1; continue; }
        if (p != k) {
            for (int c = 0; c < n; c++) { double t = LU[k * n + c]; LU[k * n + c] = LU[p * n + c]; LU[p * n + c] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        double piv = LU[k * n + k];
        for (int r = k + 1; r < n; r++) LU[r * n + k] /= piv;
        for (int r = k + 1; r < n; r++) {
            double l = LU[r * n + k];
            for (int c = k + 1; c < n; c++) LU[r * n + c] -= l * LU[k * n + c];
        }
    }
    for (int j = 0; j < n; j++) {
        double y[ORC_MAXN];
        for (int i = 0; i < n; i++) {           /* L y = P e_j */
            double s = (perm[i] == j) ? 1.0 : 0.0;
            for (int c = 0; c < i; c++) s -= LU[i * n + c] * y[c];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {      /* U x = y */
            double s = y[i];
            for (int c = i + 1; c < n; c++) s -= LU[i * n + c] * y[c];
            y[i] = s / LU[i * n + i];
        }
        for (int i = 0; i < n; i++) Ainv[i * n + j] = y[i];
    }
    return status;
}

#endif /* ORC_MATH_H */
