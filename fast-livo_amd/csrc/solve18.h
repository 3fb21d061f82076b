// solve18.h -- the dedicated solver workgroup of an 18-state ESKF pass (see handoff.h), also used
// alone by eskf18_solve_kernel for the sharded (multi-GPU) path.
//
// Math (fl_math.h, fast form):  C = Q + S (SPD 6x6),  z = C^-1 (sign*HTz - S vec6),
//       delta = T z + vec,  x (+)= delta, then the rematch/stop judgement (LIO,
//       laserMapping.cpp:1688-1728) or the accept/revert bookkeeping (VIO, lidar_selection.cpp:857-899).
// Latency is what matters (one workgroup on the critical path of every pass), so
//   * everything that depends only on the incoming state is done BEFORE the records arrive, while
//     the producers are still working: Q, T, x, x_prop staged in LDS, vec = x_prop (-) x including
//     the SO(3) Log;
//   * after the gather the independent chains run on different wavefronts:
//       wave 0: LDL^T of C in registers, rhs, triangular solves; lanes 0..17: delta_r = vec_r + T_r.z
//       then wave 0 lanes 0..8: one element each of R*Exp(delta_rot) || wave 1: additive states ||
//       wave 2: judgement.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "handoff.h"

struct FlSolveLds {
    double Q[36];
    double T[108];
    double x[24];
    double xp[24];
    double vec[18];
    double delta[18];
    double xn[12];      // rotation (9) and position (3) after the pass: input of the VIO derived pose
    double xadd[15];    // the additive states (pos, vel, bg, ba, grav) after the pass (multi-pass kernels restage from LDS)
    int ctrl;           // multi-pass kernels: bit0 stop, bit1 search wanted (written by the judging lane)
    int fragile;        // VIO: sticky FL_NUM_FRAGILE (16) once an accept test was decided within float-rounding distance
    // VIO exact accept test (see eskf18_solve_block): mirrors of the FlDev18 fields, kept across the passes of a multi-pass launch
    int need_exact, acc_buf, last_exact_valid, exact_timeout;
    unsigned acc_epoch;
    float last_exact, exact_cur;
    // loop counters of the judgement, staged with the solve inputs so that the judging lane does not wait for global loads
    int rematch, iterCount, max_iter, iters_run, accepted;
    float last_error;
    int accept;
    int st;
    int pad;
};

enum { FL_EPI_LIO = 0, FL_EPI_VIO = 1 };

// Per-patch errors of one pass for the exact VIO accept test (see eskf18_solve_block)
struct FlVioExact {
    const unsigned long long *words;   // [2][cap]
    int m, cap;
    unsigned epoch;                    // of the current pass = tag of its words
    float *scratch;                    // LDS, FL_EXACT_CHUNK floats
    int enabled;                       // 0: forced passes (FL_ITER_FORCE, benchmark/diagnostic mode) keep the fast test only
};
#define FL_EXACT_CHUNK 2048
// The reference's `error += patch_error` over patches 0..m-1 as one chain of float additions (no contraction). All threads of the
// workgroup stage the words (polling until their tag says they belong to pass `tag`); thread 0 adds. Result valid in thread 0.
__device__ __forceinline__ float vio_exact_sum(const unsigned long long *w, int m, unsigned tag, float *scr, int *timeout_flag)
{
#pragma clang fp contract(off)
    const int tid = threadIdx.x, nt = blockDim.x;
    float f = 0.0f;
    for (int base = 0; base < m; base += FL_EXACT_CHUNK) {
        const int cnt = min(FL_EXACT_CHUNK, m - base);
        for (int k = tid; k < cnt; k += nt) {
            unsigned long long v = 0ull;
            int spin = 0;
            do {
                v = __hip_atomic_load(w + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)v != tag && ++spin < 4096);
            if ((unsigned)v != tag) *timeout_flag = 1;
            scr[k] = __uint_as_float((unsigned)(v >> 32));
        }
        __syncthreads();
        if (tid == 0) {
            // one dependent float addition per patch; the operands come from LDS sixteen at a time so that the chain waits for
            // the adder, not for LDS
            int k = 0;
            for (; k + 16 <= cnt; k += 16) {
                const float4 a = *reinterpret_cast<const float4 *>(scr + k), b = *reinterpret_cast<const float4 *>(scr + k + 4);
                const float4 c = *reinterpret_cast<const float4 *>(scr + k + 8), d = *reinterpret_cast<const float4 *>(scr + k + 12);
                f = f + a.x; f = f + a.y; f = f + a.z; f = f + a.w;
                f = f + b.x; f = f + b.y; f = f + b.z; f = f + b.w;
                f = f + c.x; f = f + c.y; f = f + c.z; f = f + c.w;
                f = f + d.x; f = f + d.y; f = f + d.z; f = f + d.w;
            }
            for (; k < cnt; k++) f = f + scr[k];
        }
        __syncthreads();
    }
    return f;
}

// Stage 0 (before the gather): stage the solve inputs in LDS and form vec = x_prop (-) x.
// Split in two so that the loads can be issued before the kernel's control-word round trip.
__device__ __forceinline__ double eskf18_prefetch_issue(const FlDev18 *__restrict__ D)
{
    const int tid = threadIdx.x;
    double v = 0.0;
    if (tid < 36) v = D->Q[tid];
    else if (tid < 144) v = D->T[tid - 36];
    else if (tid < 168) v = D->x[tid - 144];
    else if (tid < 192) v = D->xprop[tid - 168];
    else if (tid == 192) v = (double)D->last_error;
    else if (tid == 193) v = (double)D->rematch_num;
    else if (tid == 194) v = (double)D->iterCount;
    else if (tid == 195) v = (double)D->max_iter;
    else if (tid == 196) v = (double)D->iters_run;
    else if (tid == 197) v = (double)D->accepted;
    else if (tid == 198) v = (double)(D->status & 16);
    else if (tid == 199) v = (double)D->err_acc_buf;
    else if (tid == 200) v = (double)D->err_acc_epoch;
    else if (tid == 201) v = (double)D->last_exact_valid;
    else if (tid == 202) v = (double)D->last_exact;
    return v;
}
__device__ __forceinline__ void eskf18_prefetch_commit(double v, FlSolveLds &L)
{
    const int tid = threadIdx.x;
    if (tid < 36) L.Q[tid] = v;
    else if (tid < 144) L.T[tid - 36] = v;
    else if (tid < 168) L.x[tid - 144] = v;
    else if (tid < 192) L.xp[tid - 168] = v;
    else if (tid == 192) L.last_error = (float)v;
    else if (tid == 193) L.rematch = (int)v;
    else if (tid == 194) L.iterCount = (int)v;
    else if (tid == 195) L.max_iter = (int)v;
    else if (tid == 196) L.iters_run = (int)v;
    else if (tid == 197) L.accepted = (int)v;
    else if (tid == 198) L.fragile = (int)v;
    else if (tid == 199) L.acc_buf = (int)v;
    else if (tid == 200) L.acc_epoch = (unsigned)v;
    else if (tid == 201) L.last_exact_valid = (int)v;
    else if (tid == 202) L.last_exact = (float)v;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if (wave == 1) {
        if (lane < 15) L.vec[3 + lane] = L.xp[9 + lane] - L.x[9 + lane];
    } else if (wave == 2 && lane == 0) {
        // Log(R^T R_prop), common_lib.h:354-358, so3_math.h:75-81
        double rd[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) rd[i * 3 + j] = L.x[0 * 3 + i] * L.xp[0 * 3 + j] + L.x[1 * 3 + i] * L.xp[1 * 3 + j] + L.x[2 * 3 + i] * L.xp[2 * 3 + j];
        const double tr = rd[0] + rd[4] + rd[8];
        const double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
        const double fk = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
        L.vec[0] = fk * (rd[7] - rd[5]);
        L.vec[1] = fk * (rd[2] - rd[6]);
        L.vec[2] = fk * (rd[3] - rd[1]);
    }
    // visibility of L.vec to wave 0 is ensured by the __syncthreads inside the gather / before the solve
}
__device__ __forceinline__ void eskf18_prefetch(const FlDev18 *__restrict__ D, FlSolveLds &L)
{
    eskf18_prefetch_commit(eskf18_prefetch_issue(D), L);
}

// Solve + state update + judgement. All threads of the workgroup call it (NT >= 256); s_sums in LDS.
// FMA contraction is allowed here (compared to the oracle by tolerance, never bitwise).
// bcast != nullptr (multi-pass kernels): the lanes that form the new pose publish it themselves the moment it exists
// (self-validating words of handoff.h, tagged bepoch), the judging lane publishes the control word.
__device__ __forceinline__ void fl_bcast_store(unsigned long long *bcast, int idx, double v, unsigned bepoch)
{
    __hip_atomic_store(bcast + 2 * idx, ((unsigned long long)f64_lo(v) << 32) | (unsigned long long)bepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bcast + 2 * idx + 1, ((unsigned long long)f64_hi(v) << 32) | (unsigned long long)bepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int KIND>
__device__ __forceinline__ void eskf18_solve_block(FlDev18 *__restrict__ D, const double *s_sums, FlSolveLds &L, int gather_status,
                                                   unsigned long long *bcast = nullptr, unsigned bepoch = 0u, const FlVioExact ex = FlVioExact{})
{
#pragma clang fp contract(fast)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const double sign = (KIND == FL_EPI_VIO) ? -1.0 : 1.0;

    if (KIND == FL_EPI_VIO) {
        // error = sum(res^2) / n_meas ; accept iff error <= last_error (lidar_selection.cpp:857-861).
        // The reference forms `error` as a FLOAT running sum: per patch `patch_error += res*res` over its 64 pixels, then
        // `error += patch_error` over the patches in order (:849-857). Its last digits depend on that order, which no tree reduction
        // reproduces, and near convergence the test is decided inside that noise. So: the producers leave every patch's
        // `patch_error` exactly as the reference rounds it (vio_produce), the fast test uses the fp64-reduced sum, and whenever it is
        // closer than (m + 8) * 2^-24 relative (the worst-case distance between the two sums) the workgroup replays the reference's running sum
        // over those per-patch floats -- m dependent float additions by one lane, ~5 us, only on such passes -- for this pass and, if
        // not known yet, for the last accepted one, and decides on the reference's own float values (status bit 16 reports that the
        // slow path ran). Without the per-patch words (sharded solve kernel: `ex` == nullptr) bit 16 means "may differ".
        const bool can_replay = ex.words != nullptr && ex.enabled;     // (by value: a nullable pointer to it kept the struct in scratch)
        auto decide = [&](bool exact) {          // thread 0 only
            const float error = exact ? L.exact_cur : (float)s_sums[FL_S_RES] / (float)s_sums[FL_S_NEFF];
            const float last = exact ? L.last_exact : L.last_error;
            const int acc = (error <= last) ? 1 : 0;
            if (L.need_exact) L.fragile = 16;
            L.accept = acc;
            D->error = error;
            if (acc) {
                D->last_error = error; L.last_error = error;
                L.acc_buf = L.iters_run & 1; L.acc_epoch = ex.epoch;
                L.last_exact_valid = exact ? 1 : 0; L.last_exact = error;
                D->err_acc_buf = L.acc_buf; D->err_acc_epoch = L.acc_epoch; D->last_exact_valid = L.last_exact_valid; D->last_exact = error;
            }
        };
        if (tid == 0) {
            const float n_meas = (float)s_sums[FL_S_NEFF];
            const float error = (float)s_sums[FL_S_RES] / n_meas;
            const float last = L.last_error;
            // worst-case distance between this fp64-reduced value and the reference's float running sum, for both operands:
            // m additions of at most half an ulp each, plus the casts and the division
            const float thr = (n_meas * (1.0f / 64.0f) + 8.0f) * 5.9604645e-8f;
            L.need_exact = (last < 1e9f && fabsf(error - last) <= thr * fabsf(error)) ? 1 : 0;
            L.exact_timeout = 0;
            if (!(L.need_exact && can_replay)) decide(false);       // the common case: one barrier, as before
        }
        __syncthreads();
        if (L.need_exact && can_replay) {
            const int cur_buf = L.iters_run & 1;
            const float fc = vio_exact_sum(ex.words + (size_t)cur_buf * ex.cap, ex.m, ex.epoch, ex.scratch, &L.exact_timeout);
            if (tid == 0) L.exact_cur = fc / (float)(64 * ex.m);
            if (!L.last_exact_valid && L.acc_buf != cur_buf) {   // (same half: only when forced passes ran on after a rejection)
                const float fl = vio_exact_sum(ex.words + (size_t)L.acc_buf * ex.cap, ex.m, L.acc_epoch, ex.scratch, &L.exact_timeout);
                if (tid == 0) { L.last_exact = fl / (float)(64 * ex.m); L.last_exact_valid = 1; }
            }
            __syncthreads();
            if (tid == 0) decide(!L.exact_timeout && L.last_exact_valid);
            __syncthreads();
        }
        if (!L.accept) {   // revert: state = old_state ; EKF_end (:888-892)
            if (tid < 24) {
                const double xo = D->xold[tid];
                D->x[tid] = xo;
                if (tid < 12) L.xn[tid] = xo;
                if (tid >= 9) L.xadd[tid - 9] = xo;
            }
            if (tid == 64) {
                L.iters_run = L.iters_run + 1;
                D->iters_run = L.iters_run;
                D->stop = 1;
                L.ctrl = 1;
                D->converged = 1;
                D->neff = (int)s_sums[FL_S_NEFF];
                D->total_residual = (double)L.last_error;
                D->status = gather_status | L.fragile;
            }
            if (wave == 3 && lane < FL_SUMS18) D->sums[lane] = s_sums[lane];
            return;
        }
    }

    // ---- wave 0: factorise, solve, delta.  Other waves: side outputs.
    if (wave == 0) {
        FlLdl6 f;
        double S[6][6], C[6][6], z[6];
        fl_unpack_S(s_sums, S);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) C[i][j] = L.Q[i * 6 + j] + S[i][j];
        const int bad = fl_ldl6(C, f);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double b = sign * s_sums[FL_S_HTZ + i];
#pragma unroll
            for (int k = 0; k < 6; k++) b -= S[i][k] * L.vec[k];
            z[i] = b;
        }
        fl_ldl6_solve(f, z);
        if (lane < 18) {
            double dl = L.vec[lane];
#pragma unroll
            for (int c = 0; c < 6; c++) dl += L.T[lane * 6 + c] * z[c];
            L.delta[lane] = dl;
            D->solution[lane] = dl;
        }
        if (lane == 0) L.st = bad | gather_status;
    } else if (wave == 3) {
        if (lane < FL_SUMS18) {
            const double sv = s_sums[lane];
            D->sums[lane] = sv;
            D->sums_acc[lane] = sv;   // LIO: last executed pass ; VIO: last accepted pass (accept path)
        }
    } else if (KIND == FL_EPI_VIO && wave == 1) {
        if (lane < 24) D->xold[lane] = L.x[lane];   // old_state = *state (:863)
    }
    __syncthreads();

    // ---- state update and judgement on separate waves
    if (wave == 0) {
        if (lane < 9) {
            // R <- R * Exp(d0,d1,d2): lane (i,j) forms its element (so3_math.h:54-72, common_lib.h:345)
            const int i = lane / 3, j = lane % 3;
            const double d0 = L.delta[0], d1 = L.delta[1], d2 = L.delta[2];
            const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (nrm > 0.00001) {
                const double r0 = d0 / nrm, r1 = d1 / nrm, r2 = d2 / nrm;
                const double K[9] = {0.0, -r2, r1, r2, 0.0, -r0, -r1, r0, 0.0};
                double s, c;
                fl_sin_omc(nrm, &s, &c);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double e[3];   // row k of E = I + s K + c K K
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const double kk = K[k * 3 + 0] * K[0 * 3 + q] + K[k * 3 + 1] * K[1 * 3 + q] + K[k * 3 + 2] * K[2 * 3 + q];
                        e[q] = ((k == q) ? 1.0 : 0.0) + s * K[k * 3 + q] + c * kk;
                    }
                    const double ekj = (j == 0) ? e[0] : ((j == 1) ? e[1] : e[2]);
                    acc += L.x[i * 3 + k] * ekj;
                }
                D->x[lane] = acc;
                L.xn[lane] = acc;
                if (KIND == FL_EPI_LIO && bcast) fl_bcast_store(bcast, lane, acc, bepoch);
            } else {
                L.xn[lane] = L.x[lane];
                if (KIND == FL_EPI_LIO && bcast) fl_bcast_store(bcast, lane, L.x[lane], bepoch);
            }
        }
    } else if (wave == 1) {
        if (lane < 15) {
            const double nv = L.x[9 + lane] + L.delta[3 + lane];
            D->x[9 + lane] = nv;
            L.xadd[lane] = nv;
            if (lane < 3) {
                L.xn[9 + lane] = nv;
                if (KIND == FL_EPI_LIO && bcast) fl_bcast_store(bcast, 9 + lane, nv, bepoch);
            }
        }
    } else if (wave == 2) {
        if (lane == 0) {
            const double rn = sqrt(L.delta[0] * L.delta[0] + L.delta[1] * L.delta[1] + L.delta[2] * L.delta[2]);
            const double tn = sqrt(L.delta[3] * L.delta[3] + L.delta[4] * L.delta[4] + L.delta[5] * L.delta[5]);
            int st = L.st;
#pragma unroll
            for (int r = 0; r < 18; r++)
                if (!(fabs(L.delta[r]) <= DBL_MAX)) st |= 2;
            if (KIND == FL_EPI_LIO) {
                // laserMapping.cpp:1688-1728
                const int converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
                int rematch = L.rematch, need_search = 0, stop = 0;
                const int it = L.iterCount;
                if (converged || ((rematch == 0) && (it == (L.max_iter - 2)))) { need_search = 1; rematch++; }
                if (rematch >= 2 || (it == L.max_iter - 1)) stop = 1;
                L.rematch = rematch; L.iterCount = it + 1; L.iters_run = L.iters_run + 1;
                D->converged = converged;
                D->rematch_num = rematch;
                D->need_search = need_search;
                D->stop = stop;
                D->iterCount = it + 1;
                D->iters_run = L.iters_run;
                D->neff = (int)s_sums[FL_S_NEFF];
                D->total_residual = s_sums[FL_S_RES];
                D->status = st | ((s_sums[FL_S_NEFF] < 1.0) ? 4 : 0);
                L.ctrl = (stop ? 1 : 0) | (need_search ? 2 : 0) | (gather_status ? 4 : 0);
                if (bcast)
                    __hip_atomic_store(bcast + 24, ((unsigned long long)(unsigned)L.ctrl << 32) | (unsigned long long)bepoch, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // lidar_selection.cpp:883-899
                int stop = ((rn * 57.3f < 0.001f) && (tn * 100.0f < 0.001f)) ? 1 : 0;
                D->converged = stop;
                L.accepted = L.accepted + 1;
                D->accepted = L.accepted;
                const int it = L.iters_run + 1;
                L.iters_run = it;
                D->iters_run = it;
                if (it >= L.max_iter) stop = 1;
                D->stop = stop;
                D->neff = (int)s_sums[FL_S_NEFF];
                D->total_residual = (double)L.last_error;
                D->status = st | L.fragile;
                L.ctrl = stop ? 1 : 0;
            }
        }
    }
}

// Multi-pass kernels: after eskf18_solve_block (and a __syncthreads) make the new state the solve input of the next
// pass without a global round trip: x <- x (+) delta from LDS, vec = x_prop (-) x as in eskf18_prefetch_commit.
__device__ __forceinline__ void eskf18_restage(FlSolveLds &L)
{
    const int tid = threadIdx.x;
    if (tid < 9) L.x[tid] = L.xn[tid];
    else if (tid < 24) L.x[tid] = L.xadd[tid - 9];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if (wave == 1) {
        if (lane < 15) L.vec[3 + lane] = L.xp[9 + lane] - L.x[9 + lane];
    } else if (wave == 2 && lane == 0) {
        double rd[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) rd[i * 3 + j] = L.x[0 * 3 + i] * L.xp[0 * 3 + j] + L.x[1 * 3 + i] * L.xp[1 * 3 + j] + L.x[2 * 3 + i] * L.xp[2 * 3 + j];
        const double tr = rd[0] + rd[4] + rd[8];
        const double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
        const double fk = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
        L.vec[0] = fk * (rd[7] - rd[5]);
        L.vec[1] = fk * (rd[2] - rd[6]);
        L.vec[2] = fk * (rd[3] - rd[1]);
    }
}
