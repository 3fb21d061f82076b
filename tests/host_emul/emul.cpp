// tests/host_emul/emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the product's per-measurement arithmetic (fast-livo_amd/csrc/fl_math.h) for the host so
// the CPU test-suite (-m "not gpu") can unit-test it against the oracle without a GPU.  This
// object is never linked into libfastlivo_hip.so and is not a fallback path.
#include "../../fast-livo_amd/csrc/fl_math.h"
#include <string.h>

extern "C" {

int emul_fit_planes(const float *nbr, int n, float *plane, unsigned char *ok)
{
    for (int i = 0; i < n; i++) ok[i] = (unsigned char)fl_esti_plane(nbr + (size_t)i * 15, plane + (size_t)i * 4);
    return 0;
}

// one pass over the points: updates sel, accumulates the 32-double record sequentially
int emul_lio18_accumulate(const float *body, const float *plane, unsigned char *sel, int n, const double *x,
                          const double *R_LI, const double *t_LI, double *sums, float *normvec)
{
    for (int k = 0; k < 32; k++) sums[k] = 0.0;
    for (int i = 0; i < n; i++) {
        if (!sel[i]) continue;
        double p_i[3]; float pw[3], pd2; int eff;
        int s = fl_point_gates(body + (size_t)i * 3, plane + (size_t)i * 4, x, x + 9, R_LI, t_LI, p_i, pw, &pd2, &eff);
        sel[i] = (unsigned char)s;
        if (s && normvec) { normvec[i*4] = plane[i*4]; normvec[i*4+1] = plane[i*4+1]; normvec[i*4+2] = plane[i*4+2]; normvec[i*4+3] = pd2; }
        if (!eff) continue;
        double row[6], z;
        fl_row18(p_i, plane + (size_t)i * 4, pd2, x, row, &z);
        fl_accum6(sums, row, z);
        sums[27] += 1.0;
        sums[28] += (double)fabsf(pd2);
        sums[29] += (double)pd2 * (double)pd2;
    }
    return 0;
}

int emul_solve18(double *x, const double *xprop, const double *P, double meas_cov, const double *sums, double sign,
                 double *G6, double *delta)
{
    return fl_solve18_serial(x, xprop, P, meas_cov, sums, sign, G6, delta);
}

// per-frame prepare + fast per-iteration solve + gain (the forms the device kernels run)
int emul_solve18_fast(double *x, const double *xprop, const double *P, double meas_cov, const double *sums, double sign,
                      double *G6, double *delta)
{
    double Q[36], T[108];
    int st = fl_prepare18(P, meas_cov, Q, T);
    st |= fl_solve18_fast(x, xprop, Q, T, sums, sign, delta);
    st |= fl_gain18(Q, T, sums, G6);
    return st;
}
}
