// tests/host_emul/emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the product's per-measurement arithmetic (fast-livo_amd/csrc/fl_math.h) for the host so
// the CPU test-suite (-m "not gpu") can unit-test it against the oracle without a GPU.  This
// object is never linked into libfastlivo_hip.so and is not a fallback path.
#include "../../fast-livo_amd/csrc/fl_math.h"
#include <string.h>
#include <stdlib.h>

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

// the per-point gate threshold (fl_gate_threshold) against the reference's expression evaluated directly:
// for every point, the threshold T and the number of probes a around T (T-3ulp .. T+3ulp, plus a few far ones) for which
// `a <= T` and the expression disagree
int emul_gate_thresholds(const float *body, int n, float *T_out, int *mismatches)
{
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const float *pb = body + (size_t)i * 3;
        const float T = fl_gate_threshold(pb);
        T_out[i] = T;
        const double b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const double q = sqrt(sqrt(b0 * b0 + b1 * b1 + b2 * b2));
        float probes[16];
        int np = 0;
        probes[np++] = 0.0f; probes[np++] = 1e-30f; probes[np++] = 1.0f; probes[np++] = 2.0f; probes[np++] = 3.4e38f;
        if (T >= 0.0f) {
            const unsigned u = fl_float_bits(T);
            for (int d = -3; d <= 3; d++) {
                const long long v = (long long)u + d;
                if (v >= 0 && v <= 0x7F7FFFFFll) probes[np++] = fl_bits_float((unsigned)v);
            }
        }
        for (int k = 0; k < np; k++)
            if ((probes[k] <= T) != (fl_gate_expr(probes[k], q) != 0)) bad++;
    }
    *mismatches = bad;
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

// ------------------------------------------------------------------------------------ Mode-23
#include "../../fast-livo_amd/csrc/fl_ikfom_math.h"
extern "C" {
// h_share_model reduced to sums at state x (26 doubles); planes fitted beforehand
int emul_h_share_sums(const double *x, const float *body, const float *plane, unsigned char *sel, int n, double *sums, float *normvec)
{
    for (int k = 0; k < 96; k++) sums[k] = 0.0;
    for (int i = 0; i < n; i++) {
        if (!sel[i]) continue;
        double p_i[3]; float pw[3], pd2; int eff;
        fl_world_point23(x, body + (size_t)i * 3, p_i, pw);
        int s = fl_gates_from_pw(body + (size_t)i * 3, plane + (size_t)i * 4, pw, &pd2, &eff);
        sel[i] = (unsigned char)s;
        if (s && normvec) { normvec[i*4] = plane[i*4]; normvec[i*4+1] = plane[i*4+1]; normvec[i*4+2] = plane[i*4+2]; normvec[i*4+3] = pd2; }
        if (!eff) continue;
        double row[12], z;
        fl_row23(x, body + (size_t)i * 3, p_i, plane + (size_t)i * 4, pd2, row, &z);
        fl_accum12(sums, row, z);
        sums[FL_S23_NEFF] += 1.0;
        sums[FL_S23_RES] += (double)fabsf(pd2);
        sums[FL_S23_RES2] += (double)pd2 * (double)pd2;
    }
    return 0;
}
void emul_world_points23(const double *x, const float *body, int n, float *world)
{
    for (int i = 0; i < n; i++) { double p_i[3]; fl_world_point23(x, body + (size_t)i * 3, p_i, world + (size_t)i * 3); }
}
// ctl: iter_i, t_count, converge, finished, max_iter, status
int emul_ikfom_iterate(double *x, const double *xprop, const double *Pprop, double *P, const double *limit, double R,
                       const double *sums, int *ctl6, double *dx_out)
{
    FlIkfomCtl c; c.iter_i = ctl6[0]; c.t_count = ctl6[1]; c.converge = ctl6[2]; c.finished = ctl6[3]; c.max_iter = ctl6[4]; c.status = ctl6[5];
    static double work[FL_IKFOM_WORK];
    fl_ikfom_iterate(x, xprop, Pprop, P, limit, R, sums, &c, dx_out, work);
    ctl6[0] = c.iter_i; ctl6[1] = c.t_count; ctl6[2] = c.converge; ctl6[3] = c.finished; ctl6[4] = c.max_iter; ctl6[5] = c.status;
    return c.status;
}
void emul_x23_boxplus(double *x, const double *dx) { fl_x23_boxplus(x, dx); }
void emul_x23_boxminus(const double *x, const double *o, double *dx) { fl_x23_boxminus(x, o, dx); }
}

// ---- exact_chain.h: the wave-parallel float running sum, its 64 lanes run one after the other per phase
#include "../../fast-livo_amd/csrc/exact_chain.h"
extern "C" {
float emul_chain_f32(const float *in, int cnt, float init, int *steps_out)
{
    float *scr = (float *)calloc((size_t)cnt + FL_CHAIN_STEP, sizeof(float));     // zero-padded like the LDS staging buffer
    bool bad = false;
    for (int i = 0; i < cnt; i++) { scr[i] = in[i]; if (!(in[i] >= 0.0f)) bad = true; }
    float s = init;
    int k = 0, steps = 0;
    const int lead = bad ? cnt : (cnt < FL_CHAIN_LEAD ? cnt : FL_CHAIN_LEAD);
    for (; k < lead; k++) s = s + scr[k];
    while (k < cnt) {
        if (fl_chain_plain_only(s)) break;
        int S, Eb;
        fl_chain_split(s, &S, &Eb);
        bool crossed = false;
        do {
            FlChainLane L[64];
            int any_tie = 0;
            for (int l = 0; l < 64; l++) any_tie |= fl_chain_phase1(L[l], scr, k + FL_CHAIN_EPL * l, Eb);
            if (any_tie) {
                unsigned long long Cm = 0, Xm = 0;
                for (int l = 0; l < 64; l++) {
                    int isc, xr;
                    fl_chain_parity_map(L[l], &isc, &xr);
                    Cm |= (unsigned long long)(isc & 1) << l;
                    Xm |= (unsigned long long)(xr & 1) << l;
                }
                for (int l = 0; l < 64; l++) fl_chain_ties(L[l], fl_chain_parity_in(Cm, Xm, l, S), Eb);
            }
            steps++;
            int inc[64], run = 0;
            for (int l = 0; l < 64; l++) { fl_chain_sum(L[l]); run += L[l].Qc; inc[l] = run; }
            if (S + inc[63] < FL_CHAIN_LIMIT) {
                S += inc[63];
                k += FL_CHAIN_STEP;
            } else {
                int Lc = -1;
                for (int l = 0; l < 64; l++) {
                    fl_chain_crossing(L[l], l, S + inc[l] - L[l].Qc);
                    if (Lc < 0 && L[l].cidx >= 0) Lc = l;
                }
                s = fl_chain_from_units(L[Lc].sprev, Eb) + L[Lc].ec;
                k += L[Lc].cidx + 1;
                crossed = true;
            }
        } while (!crossed && k < cnt);
        if (!crossed) s = fl_chain_from_units(S, Eb);
    }
    for (; k < cnt; k++) s = s + scr[k];
    free(scr);
    if (steps_out) *steps_out = steps;
    return s;
}
// the workgroup form (fl_chain_f32_block): the product's per-element / per-event functions, the 256 threads and their scans run one
// after the other. *fellback: a check failed (the device then runs the wavefront form); *nev: events walked.
float emul_chain_f32_spec(const float *in, int cnt, float init, int *fellback, int *nev_out)
{
    *fellback = 0; if (nev_out) *nev_out = 0;
    bool bad = false;
    for (int i = 0; i < cnt; i++) if (!(in[i] >= 0.0f)) bad = true;
    if (bad || cnt > 256 * FL_SPEC_EPT || fl_chain_plain_only(init)) { *fellback = 1; return emul_chain_f32(in, cnt, init, nullptr); }
    static double evA[FL_SPEC_MAX_EVENTS + 1]; static unsigned evE[FL_SPEC_MAX_EVENTS + 1]; static int evB[FL_SPEC_MAX_EVENTS + 1];
    const int lead = cnt < FL_SPEC_LEAD ? cnt : FL_SPEC_LEAD;
    volatile float init2 = init;
    for (int k = 0; k < lead; k++) init2 = init2 + in[k];
    int nev = 0, fail = fl_chain_plain_only(init2) ? 1 : 0;
    double incl_prev = (double)init2, A = 0.0;
    for (int t = 0; t < 256; t++) {
        float e[FL_SPEC_EPT]; double d[FL_SPEC_EPT]; double run = 0.0;
        for (int j = 0; j < FL_SPEC_EPT; j++) { const int k = FL_SPEC_EPT * t + j; e[j] = (k >= lead && k < cnt) ? in[k] : 0.0f; run += (double)e[j]; d[j] = run; }
        const double excl = incl_prev, incl = incl_prev + run;
        int Eb[FL_SPEC_EPT + 1];
        Eb[0] = fl_spec_binade(excl);
        for (int j = 1; j < FL_SPEC_EPT; j++) Eb[j] = fl_spec_binade(excl + d[j - 1]);
        Eb[FL_SPEC_EPT] = fl_spec_binade(incl);
        for (int j = 0; j < FL_SPEC_EPT; j++) {
            int q, tf;
            const int kind = fl_spec_elem(e[j], Eb[j], Eb[j + 1], &q, &tf);
            if (kind) {
                union { float f; unsigned u; } b; b.f = e[j];
                if (nev < FL_SPEC_MAX_EVENTS) { evA[nev] = A; evE[nev] = kind == 2 ? (unsigned)tf : b.u; evB[nev] = kind == 2 ? FL_SPEC_TIE : Eb[j + 1]; } else fail = 1;
                nev++;
            }
            A += (double)q;
        }
        incl_prev = incl;
    }
    if (nev_out) *nev_out = nev;
    if (!fail) {
        evA[nev] = A; evE[nev] = 0u; evB[nev] = FL_SPEC_TAIL;
        int S, Eb;
        fl_chain_split(init2, &S, &Eb);
        double Ap = 0.0;
        for (int i = 0; i <= nev; i++) {
            const double dq = evA[i] - Ap;
            const int Q = dq < 16777216.0 ? (int)dq : FL_CHAIN_LIMIT;
            fail |= fl_spec_event(&S, &Eb, Q, evE[i], evB[i]);
            Ap = evA[i];
        }
        if (!fail) return fl_chain_from_units(S, Eb);
    }
    *fellback = 1;
    return emul_chain_f32(in, cnt, init, nullptr);
}
float emul_chain_f32_plain(const float *scr, int cnt, float init)
{
    volatile float s = init;
    for (int k = 0; k < cnt; k++) s = s + scr[k];
    return s;
}
}
