// lio_kernels.h -- gfx950 kernels for the LiDAR point-to-plane ESKF iteration (Mode-18).
//
//  K0  lio_fit_planes_kernel   once per neighbour staging: 5-NN -> plane (n,d) + selection flag.
//                              The plane depends only on the neighbours, not on the state, so the
//                              reference's per-iteration esti_plane (laserMapping.cpp:1571) is
//                              hoisted out of the iteration loop with bit-identical results.
//  K1  lio18_pass_kernel       one ESKF pass in ONE launch: producer workgroups do the per-point
//                              residual + gates + 1x6 row and reduce to one 32-double record each
//                              (fp64, transposing wave butterfly -> LDS -> tagged write-through
//                              record); the last workgroup of the grid is the solver (handoff.h,
//                              solve18.h): gathers the records in block order, runs the gain solve,
//                              updates the state in HBM and makes the rematch/stop judgement.
//                              MODE 1 (sharded form) stops after the gather: sums -> sums_out.
//  K3  eskf18_solve_kernel     sharded form: gain solve from an (all-reduced) record.
//  K4  eskf18_cov_update_kernel  P <- (I - G) P   (laserMapping.cpp:1715)
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "handoff.h"
#include "solve18.h"

#define FL_ITER_FORCE 1
#define FL_ITER_KEEP_NORMVEC 2
#ifndef FL_LIO_NT
#define FL_LIO_NT 256
#endif

// -------------------------------------------------------------------------------------------- K0
__global__ __launch_bounds__(FL_BLOCK) void lio_fit_planes_kernel(const float *__restrict__ nbr, const uint8_t *__restrict__ valid,
                                                                 float4 *__restrict__ plane, uint8_t *__restrict__ sel, int n)
{
    // Coalesced staging of this block's 256 x 15 floats through LDS, then one point per lane.
    __shared__ float s_nb[FL_BLOCK * 15];
    const int base = blockIdx.x * FL_BLOCK;
    const int cnt = min(FL_BLOCK, n - base);
    const float *src = nbr + (size_t)base * 15;
    for (int i = threadIdx.x; i < cnt * 15; i += FL_BLOCK) s_nb[i] = src[i];
    __syncthreads();
    const int i = base + threadIdx.x;
    if (threadIdx.x >= cnt) return;
    float nb[15];
#pragma unroll
    for (int k = 0; k < 15; k++) nb[k] = s_nb[threadIdx.x * 15 + k];   // stride 15 dwords: conflict-free
    float pl[4];
    const int ok = fl_esti_plane(nb, pl);
    const bool keep = (valid[i] != 0) && ok && (pl[0] == pl[0]);   // (a NaN normal from degenerate neighbours fails every gate of the first pass anyway)
    // a point that is not selected carries a NaN normal: the pass kernels then need no separate read of the selection byte
    // (n.x = NaN -> pd2 = NaN -> fails every gate), 32 instead of 33 bytes per point and pass
    plane[i] = keep ? make_float4(pl[0], pl[1], pl[2], pl[3]) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
    sel[i] = (uint8_t)keep;
}

// Per-frame, per-point gate threshold (fl_math.h: fl_gate_threshold): depends on the body point only.
// Written as (x, y, z, T): a pass then fetches a point with ONE 16-byte load instead of three 4-byte loads at stride 12.
__global__ __launch_bounds__(FL_BLOCK) void lio_gate_kernel(const float *__restrict__ body, float4 *__restrict__ body4, int n)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float pb[3] = {body[i * 3], body[i * 3 + 1], body[i * 3 + 2]};
    body4[i] = make_float4(pb[0], pb[1], pb[2], fl_gate_threshold(pb));
}

// Per-frame prepare: Q and T of fl_math.h (depends on P and the measurement covariance only). One workgroup of
// 128 threads; the arithmetic per output element is fl_prepare18's (threads 0..5 each eliminate one column of the
// 6x6 inverse -- the pivoting depends on the matrix only, so a column solved alone is bit-identical to the same
// column solved with the others; threads 0..107 each form one element of T).
__device__ __forceinline__ void eskf18_prepare_body(FlDev18 *__restrict__ D)
{
    __shared__ double s_B[36], s_Q[36];
    __shared__ int s_st[6];
    const int t = (int)threadIdx.x;
    const double meas_cov = D->meas_cov;
    if (t < 6) {
        double M[6][6], B[6][1];
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = 0; j < 6; j++) M[i][j] = 0.5 * (D->P[i * 18 + j] + D->P[j * 18 + i]) / meas_cov;   // symmetrised A66
            B[i][0] = (i == t) ? 1.0 : 0.0;
        }
        s_st[t] = fl_gauss_solve<6, 1>(M, B);
#pragma unroll
        for (int i = 0; i < 6; i++) s_B[i * 6 + t] = B[i][0];
    }
    __syncthreads();
    if (t < 36) {
        const int i = t / 6, j = t % 6;
        const double q = 0.5 * (s_B[i * 6 + j] + s_B[j * 6 + i]);
        s_Q[t] = q;
        D->Q[t] = q;
    }
    __syncthreads();
    if (t < 108) {
        const int r = t / 6, c = t % 6;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += (D->P[r * 18 + k] / meas_cov) * s_Q[k * 6 + c];
        D->T[t] = s;
    }
    if (t == 0) D->status = s_st[0] | s_st[1] | s_st[2] | s_st[3] | s_st[4] | s_st[5];
}
__global__ __launch_bounds__(128) void eskf18_prepare_kernel(FlDev18 *__restrict__ D) { eskf18_prepare_body(D); }

// A multi-pass launch tells the host that its workgroups are gone (admission check of fastlivo_hip.hip): one system-scope
// store of the launch's sequence number into a pinned host word, by the solver workgroup as its last action.
__device__ __forceinline__ void fl_mp_done(unsigned *done_word, unsigned seq, bool solver_wg)
{
    if (solver_wg && threadIdx.x == 0 && done_word) __hip_atomic_store(done_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Is this launch a no-op?  (a) an earlier pass of the enqueued chain was ABANDONED (hand-off time-out, solve18.h): nothing runs
// until the host has resumed the frame, and the launch adds the passes it would have run to D->resume_count so that the host
// knows how much is left -- by thread 0 of its SOLVER workgroup, the only writer of resume_count (a producer workgroup that becomes
// resident after the same launch's solver has timed out sees the bit too and must not count); (b) the filter stopped or waits for a search (not under FL_ITER_FORCE). Uniform over the grid.
__device__ __forceinline__ bool fl_pass_skipped(FlDev18 *__restrict__ D, int flags, int passes, bool counter_thread)
{
    // (all five words in flight at once: as a short-circuit chain they were up to four L2 round trips in a row at the head of every launch)
    const int status = D->status, stop = D->stop, need = D->need_search, searched_at = D->searched_at, iters_run = D->iters_run;
    if (status & FL_NUM_TIMEOUT) {
        if (counter_thread) D->resume_count += passes;
        return true;
    }
    return !(flags & FL_ITER_FORCE) && (stop || (need && searched_at != iters_run));
}

// -------------------------------------------------------------------------------------------- K1
// grid = producers + 1 ; MODE 0: fused pass ; MODE 1: accumulate only (sums -> sums_out)
// FL_LIO_PASS_WAVES (tuning aid): 4 = a 128-VGPR budget, 4 instead of 3 wavefronts per SIMD. Measured: 32 M points 217 -> 207 us per
// pass, 8 M points 45 -> 48 us, 50 k points 8.2 -> 8.5 us (12 bytes of scratch): not adopted. (Round 6, checked before choosing it by size as
// the round-5 review suggested: the kernel has since come down to 116 VGPRs, i.e. it runs four wavefronts per SIMD under EITHER launch bound --
// the two instantiations compile to the same resources; there is no 5 % left to take.)
#ifndef FL_LIO_PASS_WAVES
#define FL_LIO_PASS_WAVES 1
#endif
template <int MODE>
__global__ __launch_bounds__(FL_LIO_NT, FL_LIO_PASS_WAVES) void lio18_pass_kernel(const float4 *__restrict__ body4,
                                                              float4 *__restrict__ plane,
                                                              uint8_t *__restrict__ sel, float4 *__restrict__ normvec, int n,
                                                              FlDev18 *__restrict__ D, void *__restrict__ records,
                                                              unsigned *__restrict__ epoch_ptr, double *__restrict__ sums_out,
                                                              int flags)
{
    constexpr int NT = FL_LIO_NT;
    FL_INSTR(if ((flags & FL_ITER_STAMP) && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[blockIdx.x] = (long long)wall_clock64();)
    const int nprod = gridDim.x - 1;
    // Software prefetch: the first point's inputs do not depend on the state, so their (vector)
    // loads are issued BEFORE the scalar loads of the control words and the state -- one memory
    // round trip instead of two on the producers' critical path (harmless if the pass is a no-op).
    const int i_first = blockIdx.x * NT + threadIdx.x;
    const bool have_first = (blockIdx.x != nprod) && (i_first < n);
    float4 pf_b = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf_pl = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have_first) {
        pf_b = body4[i_first];
        pf_pl = plane[i_first];
    }
    double pf_solver = 0.0;
    if (MODE == 0 && blockIdx.x == nprod) pf_solver = eskf18_prefetch_issue(D);
    if (fl_pass_skipped(D, flags, 1, blockIdx.x == nprod && threadIdx.x == 0)) return;
    const unsigned epoch = *epoch_ptr;

    if (blockIdx.x == nprod) {
        // ------------------------------------------------------------------ solver workgroup
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS18];
        __shared__ FlSolveLds s_solve;
        FL_INSTR(fl_stamp(flags, 8);)
        FlSolveRegs G;
        if (MODE == 0) { eskf18_prefetch_commit(pf_solver, s_solve); eskf18_load_regs(s_solve, G); }
        FL_INSTR(fl_stamp(flags, 9);)
        int gst = gather_records<NT, FL_SUMS18>(records, nprod, epoch, s_fin, s_sums);
        FL_INSTR(fl_stamp(flags, 10);)
        if (threadIdx.x == 0) *epoch_ptr = epoch + 1u;
        if (MODE == 0 && D->xchg_world > 1) {               // sharded form: totals over the ranks (handoff.h)
            __shared__ double s_xchg[FL_MAX_PEERS * 32];
            const FlPeerView PV = fl_peer_view(D);
            const unsigned xe = *D->xchg_epoch;
            gst |= peer_allreduce32(PV, xe, s_sums, s_xchg);
            if (threadIdx.x == 0) *D->xchg_epoch = xe + 1u;
        }
        if (MODE == 0) {
            eskf18_solve_block<FL_EPI_LIO>(D, s_sums, s_solve, G, gst);
        } else {
            if (threadIdx.x < FL_SUMS18) sums_out[threadIdx.x] = s_sums[threadIdx.x];
        }
        FL_INSTR(fl_stamp(flags, 11);)
        return;
    }

    // -------------------------------------------------------------------- producer workgroups
    __shared__ double s_red[(NT / 64) * FL_SUMS18];
    double R[9], p[3], RLI[9], tLI[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = D->x[i]; RLI[i] = D->R_LI[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { p[i] = D->x[9 + i]; tLI[i] = D->t_LI[i]; }

    double v[FL_SUMS18];
#pragma unroll
    for (int k = 0; k < FL_SUMS18; k++) v[k] = 0.0;
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 0);)

    // rolling software prefetch: the next point's inputs (same lane, one grid stride ahead) are requested before the current
    // point's ~150 fp64 instructions, so that with several points per lane (n > 65 k) the loop is bound by issue/HBM, not by
    // one memory round trip per point
    // (a second prefetch stage -- two points in flight per lane -- was tried again in round 2 with the lighter loop: 55 vs 45 us at 8 M
    // points, 261 vs 215 at 32 M: the extra registers cost more occupancy than the extra loads in flight gain)
    for (int i = i_first; i < n; i += nprod * NT) {
        const float pb[3] = {pf_b.x, pf_b.y, pf_b.z};
        const float c_gate = pf_b.w;
        const float4 plq = pf_pl;
        const int inext = i + nprod * NT;
        if (inext < n) {
            pf_b = body4[inext];
            pf_pl = plane[inext];
        }
        if (!(plq.x == plq.x)) continue;          // not selected (NaN normal, see lio_fit_planes_kernel)
        const float pl[4] = {plq.x, plq.y, plq.z, plq.w};
        double p_i[3];
        float pw[3], pd2;
        int eff;
        const int s = fl_point_gates_T(pb, c_gate, pl, R, p, RLI, tLI, p_i, pw, &pd2, &eff);
        if (!s) { sel[i] = 0; plane[i].x = __builtin_nanf(""); }
        if ((flags & FL_ITER_KEEP_NORMVEC) && s) normvec[i] = make_float4(pl[0], pl[1], pl[2], pd2);
        if (eff) {
            double row[6], z;
            fl_row18(p_i, pl, pd2, R, row, &z);
            fl_accum6(v, row, z);
            v[FL_S_NEFF] += 1.0;
            v[FL_S_RES] += (double)fabsf(pd2);
            v[FL_S_RES2] += (double)pd2 * (double)pd2;
        }
    }
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 1);)
    const double mine = block_reduce_record<NT, FL_SUMS18>(v, s_red);
    publish_record<FL_SUMS18>(mine, epoch, records);
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 2);)
    FL_INSTR(if ((flags & FL_ITER_STAMP) && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[1024 + blockIdx.x] = (long long)wall_clock64();)
}

#define FL_LIO_DO_COV 1
// producers of a scan of n points: lio_grid() of fastlivo_hip.hip minus the solver (host and device share this one definition)
__host__ __device__ __forceinline__ int fl_lio_producers(int n, int max_prod)
{
    int b = (n + FL_LIO_NT - 1) / FL_LIO_NT;
    if (b < 1) b = 1;
    int cap = (n <= 130000) ? 160 : ((n <= 300000) ? 255 : ((n <= 2000000) ? 511 : (FL_MAX_BLOCKS - 1)));
    if (max_prod > 0 && cap > max_prod) cap = max_prod;
    if (b > cap) b = cap;
    return b;
}
__device__ __attribute__((noinline)) void eskf18_cov_outofline(FlDev18 *D);      // (out of line: its LDS and registers stay out of the pass loop)
// -------------------------------------------------------------------------------------------- K1m
// Up to `count` passes in ONE launch (the LIO block of a frame between two searches): the kernel boundary (~1.5 us), the
// launch prologue and the no-op launches after stop/need_search disappear. Per pass the producers wait for the pose of the
// previous solve (bcast_wait, handoff.h), the solver gathers the records, solves, and broadcasts the new pose together with
// the stop / search-wanted bits, so every workgroup leaves the loop at the same pass. The points' loads do not depend on the
// pose and are issued before the wait. Requires every workgroup co-resident: the host uses it only for grids <= 256
// workgroups (<= 1 per CU) and falls back to one launch per pass otherwise; every spin is bounded.
// Arithmetic per pass is that of lio18_pass_kernel<0>: states are bit-identical.
__global__ __launch_bounds__(FL_LIO_NT, 2) void lio18_multipass_kernel(const float4 *__restrict__ body4,
                                                                   float4 *__restrict__ plane,
                                                                   uint8_t *__restrict__ sel, float4 *__restrict__ normvec, int n,
                                                                   FlDev18 *__restrict__ D, void *__restrict__ records,
                                                                   unsigned *__restrict__ epoch_ptr, unsigned long long *__restrict__ bcast,
                                                                   int count, int flags, unsigned *__restrict__ done_word, unsigned done_seq,
                                                                   int extra, const int *__restrict__ n_dev = nullptr, int max_prod = 0)
{
    // extra & FL_LIO_DO_COV (the LAST pass launch of fl_lio_frame18_dev): the solver workgroup ends with the covariance update
    // P <- (I - G) P of eskf18_cov_update_kernel -- also when the launch has no pass left to run (the filter stopped in the first
    // segment), not when the chain was abandoned (the host resumes and enqueues the kernel).
    constexpr int NT = FL_LIO_NT;
    const int solver_block = gridDim.x - 1;
    int nprod = solver_block;
    // the words of the prologue in flight together (one L2 round trip, not one per branch): the epoch, the device-side scan size, the
    // five words of fl_pass_skipped
    const unsigned epoch0 = *epoch_ptr;
    const int n_dev_now = *(n_dev ? n_dev : reinterpret_cast<const int *>(epoch_ptr));
    const bool skipped = fl_pass_skipped(D, flags, count, (int)blockIdx.x == solver_block && threadIdx.x == 0);
    // n_dev (fl_lidar_front): the scan's size is known to the device only; the grid was sized for the raw scan. The producers that
    // lio_grid() would give a scan of the real size take the points (same partition, same record order: same bits as the staged
    // calls); the workgroups beyond them mark their record slot "never written" (tag 0 -- a later, larger launch must not meet an old
    // tag there, see records_for) and leave.
    if (n_dev) {
        n = n_dev_now;
        const int want = fl_lio_producers(n, max_prod);
        if ((int)blockIdx.x < solver_block && (int)blockIdx.x >= want) {
            if (threadIdx.x < FL_SUMS18)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(records) + (size_t)blockIdx.x * FL_SUMS18 + threadIdx.x, 0ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        nprod = min(nprod, want);
    }
    const bool force = (flags & FL_ITER_FORCE) != 0;
    if (skipped) {
        if ((extra & FL_LIO_DO_COV) && (int)blockIdx.x == solver_block) eskf18_cov_outofline(D);
        fl_mp_done(done_word, done_seq, (int)blockIdx.x == solver_block);
        return;
    }

    if ((int)blockIdx.x == solver_block) {
        // ------------------------------------------------------------------ solver workgroup
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS18];
        __shared__ FlSolveLds s_solve;
        __shared__ double s_xchg[FL_MAX_PEERS * 32];
        eskf18_prefetch(D, s_solve);
        __shared__ unsigned long long *s_peers[FL_MAX_PEERS];
        const FlPeerView PV = fl_peer_view_lds(D, s_peers);
        const unsigned xe0 = PV.world > 1 ? *D->xchg_epoch : 0u;
        int done = 0;
        for (int p = 0; p < count; p++) {
            const unsigned epoch = epoch0 + (unsigned)p;
            FlSolveRegs G;
            eskf18_load_regs(s_solve, G);                        // solve operands into wave 0's registers while the producers work
            FL_INSTR(if (p == 5) fl_stamp(flags, 16);)
            int gst = gather_records<NT, FL_SUMS18>(records, nprod, epoch, s_fin, s_sums);
            if (PV.world > 1) gst |= peer_allreduce32(PV, xe0 + (unsigned)p, s_sums, s_xchg);      // sharded form: totals over the ranks
            FL_INSTR(if (p == 5) fl_stamp(flags, 17);)
            eskf18_solve_block<FL_EPI_LIO>(D, s_sums, s_solve, G, gst, bcast, epoch + 1u, FlVioExact{}, nullptr, (p == 5) ? (flags & FL_ITER_STAMP) : 0);   // wave 0 publishes pose + control word
            FL_INSTR(if (p == 5) fl_stamp(flags, 35);)
            __syncthreads();
            FL_INSTR(if (p == 5) fl_stamp(flags, 18);)
            done = p + 1;
            const int ctrl = s_solve.ctrl;                       // bit 2: a hand-off time-out abandoned the pass and ends the launch
            if (ctrl & 4) {
                if (threadIdx.x == 0) D->resume_count = count - p;      // this pass and the ones behind it are still to do
                break;
            }
            if (!force && (ctrl & 3)) break;
            if (p + 1 < count) eskf18_restage(s_solve);
        }
        if (threadIdx.x == 0) {
            *epoch_ptr = epoch0 + (unsigned)done;
            if (PV.world > 1) *D->xchg_epoch = xe0 + (unsigned)done;
        }
        if (extra & FL_LIO_DO_COV) {
            __syncthreads();
            eskf18_cov_outofline(D);
        }
        fl_mp_done(done_word, done_seq, true);     // the solver leaves last: every producer has left its last wait by now
        return;
    }

    // -------------------------------------------------------------------- producer workgroups
    const int spin_limit = D->xchg_world > 1 ? FL_XCHG_SPIN_LIMIT : FL_GATHER_SPIN_LIMIT;   // the solver may be waiting for another process
    __shared__ double s_red[(NT / 64) * FL_SUMS18];
    __shared__ double s_pose[12];
    __shared__ int s_ctrl;
    double R[9], p[3], RLI[9], tLI[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = D->x[i]; RLI[i] = D->R_LI[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { p[i] = D->x[9 + i]; tLI[i] = D->t_LI[i]; }
    const int i_first = blockIdx.x * NT + threadIdx.x;

    for (int ps = 0; ps < count; ps++) {
        const unsigned epoch = epoch0 + (unsigned)ps;
        // this pass's first point: loads issued before the wait for the pose
            float4 pf_b = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 pf_pl = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i_first < n) {
                pf_b = body4[i_first];
            pf_pl = plane[i_first];
        }
        if (ps > 0) {
            FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 20 + 4 * (ps - 5));)
            bcast_wait(bcast, epoch, s_pose, &s_ctrl, spin_limit);
            __syncthreads();
            FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 21 + 4 * (ps - 5));)
            if (!force && (s_ctrl & 3)) break;
            if (s_ctrl & 4) break;
#pragma unroll
            for (int i = 0; i < 9; i++) R[i] = s_pose[i];
#pragma unroll
            for (int i = 0; i < 3; i++) p[i] = s_pose[9 + i];
        }
        double v[FL_SUMS18];
#pragma unroll
        for (int k = 0; k < FL_SUMS18; k++) v[k] = 0.0;
        // rolling software prefetch as in lio18_pass_kernel
        for (int i = i_first; i < n; i += nprod * NT) {
                const float pb[3] = {pf_b.x, pf_b.y, pf_b.z};
            const float c_gate = pf_b.w;
            const float4 plq = pf_pl;
            const int inext = i + nprod * NT;
            if (inext < n) {
                    pf_b = body4[inext];
                pf_pl = plane[inext];
            }
            if (!(plq.x == plq.x)) continue;          // not selected (NaN normal, see lio_fit_planes_kernel)
            const float pl[4] = {plq.x, plq.y, plq.z, plq.w};
            double p_i[3];
            float pw[3], pd2;
            int eff;
            const int s = fl_point_gates_T(pb, c_gate, pl, R, p, RLI, tLI, p_i, pw, &pd2, &eff);
            if (!s) { sel[i] = 0; plane[i].x = __builtin_nanf(""); }
            if ((flags & FL_ITER_KEEP_NORMVEC) && s) normvec[i] = make_float4(pl[0], pl[1], pl[2], pd2);
            if (eff) {
                double row[6], z;
                fl_row18(p_i, pl, pd2, R, row, &z);
                fl_accum6(v, row, z);
                v[FL_S_NEFF] += 1.0;
                v[FL_S_RES] += (double)fabsf(pd2);
                v[FL_S_RES2] += (double)pd2 * (double)pd2;
            }
        }
        FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 22 + 4 * (ps - 5));)
        const double mine = block_reduce_record<NT, FL_SUMS18>(v, s_red);
        publish_record<FL_SUMS18>(mine, epoch, records);
        __syncthreads();          // s_red / s_pose are reused by the next pass
        FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 23 + 4 * (ps - 5));)
        FL_INSTR(if ((flags & FL_ITER_STAMP) && ps == 5 && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[blockIdx.x] = (long long)wall_clock64();)   // every producer's publish time
    }
}

// -------------------------------------------------------------------------------------------- K3
// Solve from an externally reduced record (sharded form). vio != 0 selects the VIO epilogue.
// flat != nullptr (VIO): the ranks' all-gathered per-patch floats (FlVioExact::flat) -- the accept test is then decided on the
// reference's float running sum over all patches, as in the fused and in-kernel-exchange forms.
__global__ __launch_bounds__(FL_BLOCK) void eskf18_solve_kernel(FlDev18 *__restrict__ D, const double *__restrict__ sums_in,
                                                               int vio, int flags, const FlVioConst *__restrict__ VC,
                                                               const float *__restrict__ flat = nullptr, int flat_stride = 0, int flat_world = 0)
{
    if (D->status & FL_NUM_TIMEOUT) return;
    if (!(flags & FL_ITER_FORCE) && (D->stop || (!vio && D->need_search && D->searched_at != D->iters_run))) return;
    __shared__ double s_sums[FL_SUMS18];
    __shared__ FlSolveLds s_solve;
    if (threadIdx.x < FL_SUMS18) s_sums[threadIdx.x] = sums_in[threadIdx.x];
    eskf18_prefetch(D, s_solve);
    FlSolveRegs G;
    eskf18_load_regs(s_solve, G, vio ? VC : nullptr);
    if (vio) {
        __shared__ __attribute__((aligned(16))) float s_ex[FL_EXACT_LDS];
        FlVioExact ex{};
        ex.words = nullptr; ex.scratch = s_ex; ex.enabled = !(flags & FL_ITER_FORCE);
        ex.flat = flat; ex.flat_stride = flat_stride; ex.world = flat_world;
        eskf18_solve_block<FL_EPI_VIO>(D, s_sums, s_solve, G, 0, nullptr, 0u, ex, VC);   // incl. the derived camera pose
    } else eskf18_solve_block<FL_EPI_LIO>(D, s_sums, s_solve, G, 0);
}

// -------------------------------------------------------------------------------------------- K4
// P <- P - G6 * P[0:6,:]  (== (I - G) P with G's columns 6..17 zero); G from the last executed pass.
// G[:,0:6] = T (Q+S)^-1 S of the last executed/accepted pass into LDS and D->G6, by the calling workgroup (>= 108 threads, barriers
// inside): six threads solve one column of (Q+S)^-1 S each, 108 threads form one element of G each (fl_math.h fl_gain18, in
// parallel -- one thread doing all of it was half the kernel).
__device__ __forceinline__ void eskf18_gain_block(FlDev18 *__restrict__ D, double *sG /*108, LDS*/)
{
    __shared__ double sX[36];
    const int t = threadIdx.x;
    if (t < 6) {
        double xc[6];
        fl_gain18_column(D->Q, D->sums_acc, t, xc);
#pragma unroll
        for (int i = 0; i < 6; i++) sX[i * 6 + t] = xc[i];
    }
    __syncthreads();
    if (t < 108) {
        const int r = t / 6, c = t % 6;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += D->T[r * 6 + k] * sX[k * 6 + c];
        sG[t] = s;
        D->G6[t] = s;
    }
    __syncthreads();
}

// any workgroup of >= 128 threads, barriers inside; ends with the frame's result mailbox (fl_publish_state: a no-op unless a frame
// driver asked for it)
__device__ __forceinline__ void eskf18_cov_update_body(FlDev18 *__restrict__ D)
{
    __shared__ double sP[324];
    __shared__ double sG[108];
    const int t = threadIdx.x, nt = blockDim.x;
    if (!(D->status & FL_NUM_TIMEOUT)) {             // abandoned frame: the host resumes it and enqueues this kernel again (uniform)
        for (int e = t; e < 324; e += nt) sP[e] = D->P[e];
        eskf18_gain_block(D, sG);
        for (int e = t; e < 324; e += nt) {
            const int r = e / 18, c = e % 18;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += sG[r * 6 + k] * sP[k * 18 + c];
            D->P[e] = sP[e] - s;
        }
    }
    fl_publish_state(D);
}
__global__ __launch_bounds__(384) void eskf18_cov_update_kernel(FlDev18 *__restrict__ D) { eskf18_cov_update_body(D); }
__device__ __attribute__((noinline)) void eskf18_cov_outofline(FlDev18 *D) { eskf18_cov_update_body(D); }

// pointBodyToWorld (laserMapping.cpp:695-698) of one body point under the device state: double arithmetic, stored as float
__device__ __forceinline__ void fl_world_point(const FlDev18 *__restrict__ D, float bx, float by, float bz, float &wx, float &wy, float &wz)
{
    double R[9], p[3], RLI[9], tLI[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { R[k] = D->x[k]; RLI[k] = D->R_LI[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { p[k] = D->x[9 + k]; tLI[k] = D->t_LI[k]; }
    const double b0 = (double)bx, b1 = (double)by, b2 = (double)bz;
    const double q0 = (RLI[0] * b0 + RLI[1] * b1 + RLI[2] * b2) + tLI[0];
    const double q1 = (RLI[3] * b0 + RLI[4] * b1 + RLI[5] * b2) + tLI[1];
    const double q2 = (RLI[6] * b0 + RLI[7] * b1 + RLI[8] * b2) + tLI[2];
    wx = (float)((R[0] * q0 + R[1] * q1 + R[2] * q2) + p[0]);
    wy = (float)((R[3] * q0 + R[4] * q1 + R[5] * q2) + p[1]);
    wz = (float)((R[6] * q0 + R[7] * q1 + R[8] * q2) + p[2]);
}
// world points at the current device state (pointBodyToWorld) for the host kNN on rematch passes
__global__ __launch_bounds__(FL_BLOCK) void lio_world_points_kernel(const float *__restrict__ body, float *__restrict__ world, int n,
                                                                   const FlDev18 *__restrict__ D)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float wx, wy, wz;
    fl_world_point(D, body[i * 3], body[i * 3 + 1], body[i * 3 + 2], wx, wy, wz);
    world[i * 3 + 0] = wx; world[i * 3 + 1] = wy; world[i * 3 + 2] = wz;
}
