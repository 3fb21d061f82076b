// ikfom_kernels.h -- Mode-23 (IKFoM) device block and kernels: the dormant USE_IKFOM path of the
// reference (h_share_model, src/laserMapping.cpp:961-1093, driven by
// esekf::update_iterated_dyn_share_modified, include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928).
//
//  ikfom_pass_kernel   one pass of the iterated update in ONE launch, same producer / solver
//                      structure as the Mode-18 pass (handoff.h): producers form the 1x12 rows
//                      [n, A, B, C] and reduce the 96-double record (78 unique h_x^T h_x + 12 h_x^T h
//                      + 3 scalars); the solver workgroup gathers it and runs the manifold update
//                      (fl_ikfom_math.h) with the 23x23 covariance staged in LDS.
//                      MODE 1: accumulate only (the "sum-compat" body of h_share_model and the sharded form).
//  ikfom_rows_kernel   the "row-compat" body of h_share_model: per-point rows + mask for the host to compact.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "fl_ikfom_math.h"
#include "handoff.h"

struct FlDev23 {
    double x[FL_X23_LEN];     // state_ikfom, see fl_ikfom_math.h
    double xprop[FL_X23_LEN]; // x_propagated
    double P[529];            // P_ (projected while iterating; final P_ after the finishing pass)
    double Pprop[529];        // P_propagated
    double limit[23];
    double solution[23];      // last dx_
    double sums[FL_SUMS23];
    double total_residual;
    double meas_cov;          // R
    int32_t iter_i;           // loop index i of esekfom.hpp:1633 (starts at -1)
    int32_t t_count;          // t
    int32_t need_search;      // dyn_share.converge: the next h_share_model call re-runs the kNN
    int32_t stop;             // the final covariance block has run (or max iterations reached)
    int32_t converged;
    int32_t neff;
    int32_t status;
    int32_t iters_run;
    int32_t max_iter;
    int32_t searched_at;      // device k-NN: value of iters_run the last search was made for (-1: none)
    int32_t resume_count;     // passes an abandoned launch chain left undone (FL_NUM_TIMEOUT; see solve18.h)
    int32_t pad23;
    // sharded form with the exchange inside the pass kernels (api_p2p.inc), as FlDev18
    unsigned long long *xchg_peer[8];
    unsigned *xchg_epoch;
    int32_t xchg_rank, xchg_world;
    // result mailbox of fl_ikfom_update_iterated_dev (as FlDev18::pub_flag, fl_device.h): the update's last launch copies the block into
    // the page-locked mirror and raises a host word the calling thread polls -- no copy command, no stream synchronisation
    unsigned long long *pub_flag;
    void *pub_dst;
    unsigned long long pub_seq;
};

#define FL_IK_PUBLISH 0x100            /* launch flag (internal): this launch is the update's last one -- serve the mailbox at its end */

// see fl_publish_state (fl_device.h): all threads of ONE workgroup, as the last action of the update's last kernel; one-shot
__device__ __forceinline__ void fl_publish_state23(FlDev23 *__restrict__ D)
{
    __syncthreads();
    unsigned long long *flag = D->pub_flag;
    if (!flag) return;                                   // (uniform)
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(D);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(D->pub_dst);
    const unsigned long long seq = D->pub_seq;
#ifdef FL_PUB_WORDS
    constexpr int WORDS = FL_PUB_WORDS;
#else
    constexpr int WORDS = (int)(sizeof(FlDev23) / 8);
#endif
    fl_publish_copy<WORDS>(dst, src);
    // every wavefront waits for its own mirror stores (explicit s_waitcnt vmcnt(0), see fl_publish_state), then ONE wavefront pays the
    // system-scope release (an L2 write-back, ~3 us whoever issues it: with all four wavefronts issuing their own it took 12.8 us)
    fl_wait_own_stores();
    __syncthreads();
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __threadfence_system();
        if (threadIdx.x == 0) {
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            D->pub_flag = nullptr;
        }
    }
}

#include "ikfom_solve_block.h"

#ifndef FL_IK_NT
#define FL_IK_NT 256
#endif

// no-op launch? (abandoned chain: count what is skipped; stopped / waiting for a search) -- as fl_pass_skipped of lio_kernels.h
__device__ __forceinline__ bool ik_pass_skipped(FlDev23 *__restrict__ D, int flags, int passes, bool counter_thread)
{
    const int status = D->status, stop = D->stop, need = D->need_search, searched_at = D->searched_at, iters_run = D->iters_run;      // (in flight together)
    if (status & FL_NUM_TIMEOUT) {
        if (counter_thread) D->resume_count += passes;
        return true;
    }
    return !(flags & FL_ITER_FORCE) && (stop || (need && searched_at != iters_run));
}

// one producer workgroup's share of a pass at state x: rows [n, A, B, C] reduced to the 96-double record and published.
// The scan arrives as (x, y, z, T) with the selection threshold T of fl_math.h (fl_gate_threshold) and a NaN normal marks a point
// that is not selected (lio_fit_planes_kernel): 32 bytes and two loads per point, no sqrt / division in the gate.
// The first point's inputs do not depend on the state: ikfom_prefetch_first requests them BEFORE the wait for the state (multi-pass
// kernel) or the state's own loads; inside the loop the next point's inputs are requested before the current point's arithmetic (the
// rolling prefetch of lio18_pass_kernel). Until round 6 the loop read plane[i], waited, tested the NaN mark and only then read body4[i]:
// two L2 round trips in a row per point at the head of every pass (tools/isa_chains.py).
struct FlIkFirst { float4 plq, bq; };
__device__ __forceinline__ FlIkFirst ikfom_prefetch_first(const float4 *__restrict__ body4, const float4 *__restrict__ plane, int n)
{
    FlIkFirst f;
    f.plq = make_float4(0.f, 0.f, 0.f, 0.f); f.bq = f.plq;
    const int i = blockIdx.x * FL_IK_NT + threadIdx.x;
    if (i < n) { f.plq = plane[i]; f.bq = body4[i]; }
    return f;
}
__device__ __forceinline__ void ikfom_produce(const float4 *__restrict__ body4, float4 *__restrict__ plane, uint8_t *__restrict__ sel,
                                              float4 *__restrict__ normvec, int n, const double (&x)[FL_X23_LEN], int nprod, int flags,
                                              double *s_red, unsigned epoch, void *__restrict__ records, FlIkFirst pf)
{
    constexpr int NT = FL_IK_NT;
    double v[FL_SUMS23I];
#pragma unroll
    for (int k = 0; k < FL_SUMS23I; k++) v[k] = 0.0;
#ifdef FL_IK_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_fl_stamps[41] = (long long)wall_clock64();
#endif
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += nprod * NT) {
        const float4 plq = pf.plq, bq = pf.bq;
        const int inext = i + nprod * NT;
        if (inext < n) { pf.plq = plane[inext]; pf.bq = body4[inext]; }
        if (!(plq.x == plq.x)) continue;
        const float pb[3] = {bq.x, bq.y, bq.z};
        const float pl[4] = {plq.x, plq.y, plq.z, plq.w};
        double p_i[3];
        float pw[3];
        fl_world_point23(x, pb, p_i, pw);
        const float pd2 = pl[0] * pw[0] + pl[1] * pw[1] + pl[2] * pw[2] + pl[3];
        const float a = fabsf(pd2);
        const int s = (a <= bq.w) ? 1 : 0;
        const int eff = (s && (a <= 2.0f)) ? 1 : 0;
        if (!s) { sel[i] = 0; plane[i].x = __builtin_nanf(""); }
        if ((flags & FL_ITER_KEEP_NORMVEC) && s) normvec[i] = make_float4(pl[0], pl[1], pl[2], pd2);
        if (eff) {
            double row[12], z;
            fl_row23(x, pb, p_i, pl, pd2, row, &z);
            fl_accum9(v, row, z);                       // the [n, A, B] block only: the C block follows from it (fl_ikfom_math.h)
            v[FL_S23I_NEFF] += 1.0;
            v[FL_S23I_RES] += (double)fabsf(pd2);
            v[FL_S23I_RES2] += (double)pd2 * (double)pd2;
        }
    }
#ifdef FL_IK_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) { asm volatile("" ::"v"(v[0] + v[63])); g_fl_stamps[42] = (long long)wall_clock64(); }
#endif
    const double mine = block_reduce_record<NT, FL_SUMS23I>(v, s_red);
    publish_record<FL_SUMS23I>(mine, epoch, records);
#ifdef FL_IK_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_fl_stamps[43] = (long long)wall_clock64();
#endif
}

template <int MODE>
__global__ __launch_bounds__(FL_IK_NT) void ikfom_pass_kernel(const float4 *__restrict__ body4, float4 *__restrict__ plane,
                                                             uint8_t *__restrict__ sel, float4 *__restrict__ normvec, int n,
                                                             FlDev23 *__restrict__ D, void *__restrict__ records,
                                                             unsigned *__restrict__ epoch_ptr, double *__restrict__ sums_out,
                                                             int flags)
{
    constexpr int NT = FL_IK_NT;
    const int nprod = gridDim.x - 1;
    if (ik_pass_skipped(D, flags, 1, blockIdx.x == nprod && threadIdx.x == 0)) return;     // (counted by the solver workgroup only)
    const unsigned epoch = *epoch_ptr;

    if (blockIdx.x == nprod) {
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS23I];                  // the internal 64-double record (fl_ikfom_math.h)
        __shared__ FlIkLds s_ik;
#ifdef FL_IK_STAMPS
        if (threadIdx.x == 0) g_fl_stamps[32] = (long long)wall_clock64();
#endif
        if (MODE == 0) { ikfom_stage_once(D, s_ik);
#ifdef FL_IK_STAMPS
        if (threadIdx.x == 0) g_fl_stamps[33] = (long long)wall_clock64();
#endif
            ikfom_pre(s_ik); }       // while the producers work
#ifdef FL_IK_STAMPS
        if (threadIdx.x == 0) g_fl_stamps[34] = (long long)wall_clock64();
#endif
        int gst = gather_records<NT, FL_SUMS23I>(records, nprod, epoch, s_fin, s_sums);
#ifdef FL_IK_STAMPS
        if (threadIdx.x == 0) g_fl_stamps[35] = (long long)wall_clock64();
#endif
        if (threadIdx.x == 0) *epoch_ptr = epoch + 1u;
        if (MODE == 0 && D->xchg_world > 1) {               // sharded form: totals over the ranks (handoff.h)
            __shared__ double s_xchg[FL_MAX_PEERS * 32];
            const FlPeerView PV = fl_peer_view(D);
            const unsigned xe = *D->xchg_epoch;
            gst |= peer_allreduce64(PV, xe, s_sums, s_xchg);
            if (threadIdx.x == 0) *D->xchg_epoch = xe + 2u;
        }
        if (MODE == 0) {
            ikfom_post(D, s_sums, s_ik, gst);
#ifdef FL_IK_STAMPS
            if (threadIdx.x == 0) g_fl_stamps[40] = (long long)wall_clock64();
#endif
        } else {
            // accumulate only: the PUBLIC 96-double record (78 + 12 + 3) leaves the kernel, the C block rebuilt from the state's rotation
            __shared__ double s_Rm[9];
            if (threadIdx.x == 0) { double q[4] = {D->x[FL_X23_ROT], D->x[FL_X23_ROT + 1], D->x[FL_X23_ROT + 2], D->x[FL_X23_ROT + 3]}, Rm[9]; flq_to_R(q, Rm); for (int k = 0; k < 9; k++) s_Rm[k] = Rm[k]; }
            __syncthreads();
            if (threadIdx.x < FL_SUMS23) sums_out[threadIdx.x] = ikfom_public_value(s_sums, s_Rm, (int)threadIdx.x);
        }
        return;
    }

    __shared__ double s_red[(NT / 64) * FL_SUMS23I];
    const FlIkFirst pf = ikfom_prefetch_first(body4, plane, n);
    double x[FL_X23_LEN];
#pragma unroll
    for (int i = 0; i < FL_X23_LEN; i++) x[i] = D->x[i];
    ikfom_produce(body4, plane, sel, normvec, n, x, nprod, flags, s_red, epoch, records, pf);
}

// Up to `count` passes of update_iterated_dyn_share_modified in ONE launch (as lio18_multipass_kernel): the solver broadcasts the
// whole state_ikfom (26 doubles: the producers need pos, rot, offset_R_L_I, offset_T_L_I) plus the stop / search-wanted bits, stages
// the next pass's state-only work (ikfom_pre) while the producers compute, and runs the final covariance block in the pass that
// finishes. Needs every workgroup resident (host: admission check); all waits are bounded, a time-out abandons the pass.
#define FL_IK_BCAST_WORDS (2 * FL_X23_LEN + 1)
__global__ __launch_bounds__(FL_IK_NT) void ikfom_multipass_kernel(const float4 *__restrict__ body4, float4 *__restrict__ plane,
                                                                  uint8_t *__restrict__ sel, float4 *__restrict__ normvec, int n,
                                                                  FlDev23 *__restrict__ D, void *__restrict__ records,
                                                                  unsigned *__restrict__ epoch_ptr, unsigned long long *__restrict__ bcast,
                                                                  int count, int flags, unsigned *__restrict__ done_word, unsigned done_seq)
{
    constexpr int NT = FL_IK_NT;
    const int nprod = gridDim.x - 1;
    const bool force = (flags & FL_ITER_FORCE) != 0;
    if (ik_pass_skipped(D, flags, count, blockIdx.x == nprod && threadIdx.x == 0)) {
        if ((flags & FL_IK_PUBLISH) && blockIdx.x == nprod) fl_publish_state23(D);      // (the update ended in an earlier launch)
        fl_mp_done(done_word, done_seq, blockIdx.x == nprod);
        return;
    }
    const unsigned epoch0 = *epoch_ptr;

    if (blockIdx.x == nprod) {
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS23I];                  // the internal 64-double record (fl_ikfom_math.h)
        __shared__ FlIkLds s_ik;
        ikfom_stage_once(D, s_ik);
        __shared__ double s_xchg[FL_MAX_PEERS * 32];
        __shared__ unsigned long long *s_peers[FL_MAX_PEERS];
        const FlPeerView PV = fl_peer_view_lds(D, s_peers);
        const unsigned xe0 = PV.world > 1 ? *D->xchg_epoch : 0u;
        int done = 0;
        bool wrote_P = false;
        for (int p = 0; p < count; p++) {
            const unsigned epoch = epoch0 + (unsigned)p;
#ifdef FL_IK_STAMPS
#define FL_IK_MP_STAMP(slot) do { if (p == 2 && threadIdx.x == 0) g_fl_stamps[slot] = (long long)wall_clock64(); } while (0)
#else
#define FL_IK_MP_STAMP(slot) do { } while (0)
#endif
            FL_IK_MP_STAMP(20);
            ikfom_pre(s_ik);                                      // state-only half of the iteration, while the producers work
            FL_IK_MP_STAMP(21);
            int gst = gather_records<NT, FL_SUMS23I>(records, nprod, epoch, s_fin, s_sums);
            FL_IK_MP_STAMP(22);
            if (PV.world > 1) gst |= peer_allreduce64(PV, xe0 + 2u * (unsigned)p, s_sums, s_xchg);      // sharded form: totals over the ranks
            ikfom_post(D, s_sums, s_ik, gst, bcast, epoch + 1u, false);
            FL_IK_MP_STAMP(23);
            __syncthreads();
            done = p + 1;
            if (s_ik.ctl[4]) {                                    // abandoned
                if (threadIdx.x == 0) D->resume_count = count - p;
                break;
            }
            wrote_P = s_ik.ctl[2] != 0;                           // the finishing pass wrote the final P_
            if (!force && (s_ik.ctl[5] || s_ik.ctl[1])) break;    // stop, or the next pass wants a search first
        }
        if (!wrote_P) {                                           // a reader of the device block finds the projected P_ of the last pass
            for (int e = threadIdx.x; e < FL_N23 * FL_N23; e += NT) D->P[e] = s_ik.P[e];
        }
        if (threadIdx.x == 0) {
            *epoch_ptr = epoch0 + (unsigned)done;
            if (PV.world > 1) *D->xchg_epoch = xe0 + 2u * (unsigned)done;
        }
        if (flags & FL_IK_PUBLISH) { __threadfence(); fl_publish_state23(D); }      // (the barrier inside orders this workgroup's stores to D)
        fl_mp_done(done_word, done_seq, true);
        return;
    }

    const int spin_limit = D->xchg_world > 1 ? FL_XCHG_SPIN_LIMIT : FL_GATHER_SPIN_LIMIT;   // the solver may be waiting for another process
    __shared__ double s_red[(NT / 64) * FL_SUMS23I];
    __shared__ double s_state[FL_X23_LEN];
    __shared__ int s_ctrl;
    double x[FL_X23_LEN];
#pragma unroll
    for (int i = 0; i < FL_X23_LEN; i++) x[i] = D->x[i];
    for (int ps = 0; ps < count; ps++) {
        const unsigned epoch = epoch0 + (unsigned)ps;
        const FlIkFirst pf = ikfom_prefetch_first(body4, plane, n);      // (behind this thread's own store of the last pass's NaN mark: program order)
#ifdef FL_IK_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0 && (ps == 2 || ps == 3)) g_fl_stamps[24 + 4 * (ps - 2)] = (long long)wall_clock64();
#endif
        if (ps > 0) {
            bcast_wait<FL_IK_BCAST_WORDS>(bcast, epoch, s_state, &s_ctrl, spin_limit);
            __syncthreads();
#ifdef FL_IK_STAMPS
            if (blockIdx.x == 0 && threadIdx.x == 0 && (ps == 2 || ps == 3)) g_fl_stamps[25 + 4 * (ps - 2)] = (long long)wall_clock64();
#endif
            if (s_ctrl & 4) break;
            if (!force && (s_ctrl & 3)) break;
#pragma unroll
            for (int i = 0; i < FL_X23_LEN; i++) x[i] = s_state[i];
        }
        ikfom_produce(body4, plane, sel, normvec, n, x, nprod, flags, s_red, epoch, records, pf);
#ifdef FL_IK_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0 && (ps == 2 || ps == 3)) g_fl_stamps[26 + 4 * (ps - 2)] = (long long)wall_clock64();
#endif
        __syncthreads();
    }
}

// Solve from an externally reduced record (sharded form).
__global__ __launch_bounds__(FL_IK_NT) void ikfom_solve_kernel(FlDev23 *__restrict__ D, const double *__restrict__ sums_in, int flags)
{
    if (D->status & FL_NUM_TIMEOUT) return;
    if (!(flags & FL_ITER_FORCE) && (D->stop || (D->need_search && D->searched_at != D->iters_run))) return;
    __shared__ double s_sums[FL_SUMS23];
    __shared__ FlIkLds s_ik;
    if (threadIdx.x < FL_SUMS23) s_sums[threadIdx.x] = sums_in[threadIdx.x];
    ikfom_stage_once(D, s_ik);
    ikfom_pre(s_ik);
    ikfom_post<false>(D, s_sums, s_ik, 0);          // (the record arrives in the public 96-double form)
}

__global__ void ikfom_resume_kernel(FlDev23 *__restrict__ D)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { D->status &= ~FL_NUM_TIMEOUT; D->resume_count = 0; D->pub_flag = nullptr; }    // (a resumed update is read back with a copy)
}

// world points at the current state_ikfom (laserMapping.cpp:980-984) for the host kNN
__global__ __launch_bounds__(FL_BLOCK) void ikfom_world_points_kernel(const float *__restrict__ body, float *__restrict__ world, int n,
                                                                     const FlDev23 *__restrict__ D)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    double x[FL_X23_LEN], p_i[3];
#pragma unroll
    for (int k = 0; k < FL_X23_LEN; k++) x[k] = D->x[k];
    float pw[3];
    const float pb[3] = {body[i * 3], body[i * 3 + 1], body[i * 3 + 2]};
    fl_world_point23(x, pb, p_i, pw);
    world[i * 3] = pw[0]; world[i * 3 + 1] = pw[1]; world[i * 3 + 2] = pw[2];
}

// "row-compat" h_share_model: every point writes its 12-wide row, h and an effective flag; the host
// compacts them in ascending point order exactly like laserMapping.cpp:1041-1052.
__global__ __launch_bounds__(FL_BLOCK) void ikfom_rows_kernel(const float *__restrict__ body, const float4 *__restrict__ plane,
                                                             uint8_t *__restrict__ sel, int n, const FlDev23 *__restrict__ D,
                                                             double *__restrict__ rows /* n x 13 */, uint8_t *__restrict__ eff_out)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    eff_out[i] = 0;
    if (!sel[i]) return;
    double x[FL_X23_LEN], p_i[3];
#pragma unroll
    for (int k = 0; k < FL_X23_LEN; k++) x[k] = D->x[k];
    const float pb[3] = {body[i * 3], body[i * 3 + 1], body[i * 3 + 2]};
    const float4 plq = plane[i];
    const float pl[4] = {plq.x, plq.y, plq.z, plq.w};
    float pw[3], pd2;
    int eff;
    fl_world_point23(x, pb, p_i, pw);
    const int s = fl_gates_from_pw(pb, pl, pw, &pd2, &eff);
    if (!s) sel[i] = 0;
    if (eff) {
        double row[12], z;
        fl_row23(x, pb, p_i, pl, pd2, row, &z);
        for (int k = 0; k < 12; k++) rows[(size_t)i * 13 + k] = row[k];
        rows[(size_t)i * 13 + 12] = z;
        eff_out[i] = 1;
    }
}
