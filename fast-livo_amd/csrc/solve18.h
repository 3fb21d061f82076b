// solve18.h -- the dedicated solver workgroup of an 18-state ESKF pass (see handoff.h), also used
// alone by eskf18_solve_kernel for the sharded (multi-GPU) path.
//
// Math (fl_math.h, fast form):  C = Q + S (SPD 6x6),  z = C^-1 (sign*HTz - S vec6),
//       delta = T z + vec,  x (+)= delta, then the rematch/stop judgement (LIO,
//       laserMapping.cpp:1688-1728) or the accept/revert bookkeeping (VIO, lidar_selection.cpp:857-899).
// Latency is what matters (one workgroup on the critical path of every pass). Round-2 form:
//   * everything that depends only on the incoming state is done BEFORE the records arrive, while the producers are
//     still working: Q, T, x, x_prop staged in LDS, vec = x_prop (-) x including the SO(3) Log, and then -- new --
//     the per-lane operands of the solve are pulled into the REGISTERS of wavefront 0 (FlSolveRegs: row r of T / vec_r /
//     x_r in lane r);
//   * after the gather, wavefront 0 alone runs the whole chain without a single workgroup barrier: C = Q + S and the
//     right-hand side as a 7th row, right-looking LDL^T with reciprocal + two Newton steps instead of six IEEE
//     divisions (the forward substitution falls out of the same elimination), back substitution, delta_r in lane r,
//     rotation update R * Exp(delta) from the power series of sin(t)/t and (1-cos t)/t^2 in t^2 (no sqrt, no division,
//     no sin/cos), additive states, judgement, the pose words of the multi-pass broadcast -- every lane that forms a
//     result stores/publishes it itself. The other wavefronts only write side outputs (sums). The previous form spent
//     ~2 us here (three barriers, 6 divisions, sqrt + sin + cos, LDS round trips): DESIGN.md section 4.1.
//   * a pass whose hand-off timed out (gather or peer exchange) is ABANDONED: the state does not move, FL_NUM_TIMEOUT
//     becomes sticky in D->status, every later kernel of the enqueued chain sees it and does nothing, and the host re-runs the remaining passes
//     (fastlivo_hip.hip: resume_after_timeout). Status bits are sticky on the device between fl_*_begin and the read-back.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "handoff.h"
#include "exact_chain.h"

// one pyramid level's result of ComputeJ (UpdateState's outputs), behind the state block (api_vio.inc)
struct FlVioLevelInfo {
    double solution[18];
    float error;
    int32_t iterations, n_meas, accepted, status, converged;
};
struct FlSolveLds {
    double Q[36];
    double T[108];
    double x[24];
    double xp[24];
    double vec[18];
    double delta[18];
    double xn[12];      // rotation (9) and position (3) after the pass: input of the VIO derived pose
    double xadd[15];    // the additive states (pos, vel, bg, ba, grav) after the pass (multi-pass kernels restage from LDS)
    double cam[12];     // VIO: derived camera pose (Rcw, Pcw) of the state after the pass
    int ctrl;           // bit0 stop, bit1 search wanted, bit2 abandoned (written by wavefront 0)
    int fragile;        // VIO: sticky FL_NUM_FRAGILE (16) once an accept test was decided within float-rounding distance
    int sticky;         // status bits accumulated since fl_*_begin (mirror of D->status)
    // VIO exact accept test (see eskf18_solve_block): mirrors of the FlDev18 fields, kept across the passes of a multi-pass launch
    int need_exact, acc_buf, last_exact_valid, exact_timeout;
    unsigned acc_epoch;
    float last_exact, exact_cur;
    // loop counters of the judgement, staged with the solve inputs so that the judging lane does not wait for global loads
    int rematch, iterCount, max_iter, iters_run, accepted;
    float last_error;
    int accept;
    int st;
    int audited;
    // hand-off of the judgement to wavefront 1 (eskf18_solve_block): passes judged so far / the flag wavefront 0 raises when delta is in LDS
    int jpass, jflag;
    // VIO, round 5 -- a FRAGILE ACCEPT that went ahead on the fp64 decision (vio_spec_confirm below): the pass whose float-chain
    // verdict is still out, what a rejection of it leaves, and what its acceptance still has to write
    int spec_pending, spec_buf, spec_accepted, spec_iters, spec_acc_buf;
    unsigned spec_epoch, spec_acc_epoch;
    float spec_nall, spec_last_error, spec_last_exact;
    double def_xold[24], def_sums[FL_SUMS18], def_sol[18];      // old_state / sums_acc / solution of that pass, written to the device block once it is confirmed
    // VIO, round 6 -- all pyramid levels of ComputeJ in one launch (vio_multipass_kernel under FL_VIO_LEVELS): `levels` says so; `pb` =
    // the half of the per-patch word buffer the level's first pass writes (0 in a launch of one level; across levels the halves go on
    // alternating, so that the first pass of a level never overwrites the words of the last pass of the level before).
    // xl_*: a fragile accept that ENDED a level and went ahead (the next level has begun on the speculated state): what the finished
    // level's result block gets when the float chain confirms it, and the old_state a rejection goes back to (vio_spec_confirm /
    // vio_spec_rollback); xl_restart: the rejection happened -- the level that had begun starts again from the reverted state.
    int levels, pb;
    int xl_pending, xl_restart, xl_level, xl_iters, xl_accepted, xl_status, xl_converged, xl_neff;
    float xl_err;
    double xl_xold[24];
    FlVioLevelInfo *li_base;
};

enum { FL_EPI_LIO = 0, FL_EPI_VIO = 1 };

#define FL_AUDIT_RING 16               /* slots of the auditor's ring of per-pass totals (vio_kernels.h vio_audit_pass) */
#define FL_AUDIT_WORDS (FL_AUDIT_RING + 8) /* + word FL_AUDIT_RING: the epoch the auditor is working on (its progress) */
// Per-patch errors of one pass for the exact VIO accept test (see eskf18_solve_block)
struct FlVioExact {
    const unsigned long long *words;   // [2][cap]
    int m, cap;
    unsigned epoch;                    // of the current pass = tag of its words
    float *scratch;                    // LDS, FL_EXACT_LDS floats
    int enabled;                       // 0: forced passes (FL_ITER_FORCE, benchmark/diagnostic mode) keep the fast test only
    // sharded form (in-kernel peer exchange): the reference's running sum `error += patch_error` runs over ALL patches in order, i.e.
    // through the ranks' contiguous patch ranges one after the other: rank r starts from the float rank r-1 ended with, the last rank
    // sends the total back to everybody. One hop per rank, on the rare fragile passes only.
    unsigned long long *own;           // this rank's exchange buffer (nullptr: single rank)
    unsigned long long *const *peer;   // everybody's buffers
    int rank, world;
    unsigned xe;                       // exchange epoch of this pass = tag of the mail
    // RCCL / torch.distributed form (fl_vio_iterate_sharded, fl_vio_solve_exact): the ranks have all-gathered their per-patch floats,
    // `flat` holds `world` chunks of `flat_stride` floats in rank order, chunk = [patch count of the rank (int bits), its
    // patch_error floats in patch order, zero padding]. Every rank then runs the reference's chain over ALL patches itself, on every
    // pass -- the accept test is the reference's float comparison in this form too.
    const float *flat = nullptr;
    int flat_stride = 0;
};
// single rank: the auditor workgroup's ring of per-pass totals sits behind the words (vio_kernels.h vio_audit_pass), slot = epoch & 15
// total of pass `tag` out of the auditor's ring (thread 0); false when it does not arrive within `spins` polls or the auditor gave
// up on that pass (payload FL_AUDIT_NONE)
#define FL_AUDIT_NONE 0xffffffffu
#ifdef FL_AUDIT_STAMPS                      /* debug build (tools/vio_audit_stamps.py): entry / staged / added, per pass tag */
#define FL_CHAIN_STAMP(tag, j) do { if (threadIdx.x == 0) g_fl_wall[1024 + 4 * ((tag) & 63u) + (j)] = (long long)wall_clock64(); } while (0)
#else
#define FL_CHAIN_STAMP(tag, j) do { } while (0)
#endif
__device__ __forceinline__ bool vio_audit_read(const unsigned long long *ring, unsigned tag, int spins, float *out)
{
    unsigned long long v = 0ull;
    int spin = 0;
    do { v = __hip_atomic_load(ring + (tag & (FL_AUDIT_RING - 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)v != tag && ++spin < spins);
    *out = __uint_as_float((unsigned)(v >> 32));
    return (unsigned)v == tag && (unsigned)(v >> 32) != FL_AUDIT_NONE;
}
#ifndef FL_EXACT_CHUNK
#define FL_EXACT_CHUNK 2048
#endif
#define FL_EXACT_LDS (FL_EXACT_CHUNK + FL_CHAIN_STEP)      /* staging buffer of vio_exact_sum: a chunk + the chain's zero padding */
// The reference's `error += patch_error` over patches 0..m-1 as one chain of float additions (no contraction), bit for bit. All
// threads of the workgroup stage the words (polling until their tag says they belong to pass `tag`); wavefront 0 adds them up
// binade-wise over its 64 lanes (exact_chain.h: ~3 us for 2 k patches instead of ~20 us for one lane adding one by one). Result
// valid in thread 0. A word of a NEWER pass (the double-buffered half was reused: the caller is two passes late) or one that never
// arrives sets *timeout_flag.
__device__ __forceinline__ float vio_exact_sum_inl(const unsigned long long *w, int m, unsigned tag, float *scr /* FL_EXACT_LDS floats */, int *timeout_flag,
                                                   float init = 0.0f)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    float f = init;
    FL_CHAIN_STAMP(tag, 0);
    for (int base = 0; base < m; base += FL_EXACT_CHUNK) {
        const int cnt = min(FL_EXACT_CHUNK, m - base);
        constexpr int PER = 8;                               // first loads of a thread's words all in flight at once
        unsigned long long v[PER];
        bool bad = false;                                    // a negative / NaN element: the plain chain (exact_chain.h)
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int k = tid + j * nt;
            v[j] = (k < cnt) ? __hip_atomic_load(w + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int k = tid + j * nt;
            if (k < cnt) {
                unsigned long long x = v[j];
                int spin = 0;
                while ((unsigned)x != tag && (int)((unsigned)x - tag) < 0 && ++spin < 4096)
                    x = __hip_atomic_load(w + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)x != tag) *timeout_flag = 1;
                const float e = __uint_as_float((unsigned)(x >> 32));
                bad |= !(e >= 0.0f);
                scr[k] = e;
            }
        }
        for (int k = tid + PER * nt; k < cnt; k += nt) {      // (workgroups of fewer than 256 threads)
            unsigned long long x = 0ull;
            int spin = 0;
            do { x = __hip_atomic_load(w + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            while ((unsigned)x != tag && (int)((unsigned)x - tag) < 0 && ++spin < 4096);
            if ((unsigned)x != tag) *timeout_flag = 1;
            const float e = __uint_as_float((unsigned)(x >> 32));
            bad |= !(e >= 0.0f);
            scr[k] = e;
        }
        for (int k = cnt + tid; k < cnt + FL_CHAIN_STEP; k += nt) scr[k] = 0.0f;       // the zero padding the chain's steps read into
        const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
        FL_CHAIN_STAMP(tag, 1);
        f = fl_chain_f32_block(scr, cnt, f, any_bad);       // (ends with a barrier; the result is in every thread)
        FL_CHAIN_STAMP(tag, 2);
    }
    return f;
}

// Out of line for the solver workgroup's own (rare) replay: its registers and code stay out of the pass loop (the auditor, itself out
// of line, inlines vio_exact_sum_inl so that its staging buffer is known to be LDS).
__device__ __attribute__((noinline)) float vio_exact_sum(const unsigned long long *w, int m, unsigned tag, float *scr, int *timeout_flag, float init = 0.0f)
{
    return vio_exact_sum_inl(w, m, tag, scr, timeout_flag, init);
}

// The running sum over the all-gathered per-patch floats of all ranks (FlVioExact::flat), rank after rank in patch order: the m
// dependent float additions of lidar_selection.cpp:849-857, bit for bit (exact_chain.h). Whole workgroup (256 threads), barriers
// inside, the result in every thread. Not inlined: its staging and the chain stay out of the solve's register allocation.
__device__ __attribute__((noinline)) float vio_exact_flat_sum(const float *__restrict__ flat, int stride, int world, float *scr /* LDS, FL_EXACT_LDS */)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    float f = 0.0f;
    for (int r = 0; r < world; r++) {
        const float *seg = flat + (size_t)r * stride;
        int m = __float_as_int(seg[0]);
        m = m < 0 ? 0 : (m > stride - 1 ? stride - 1 : m);
        for (int base = 0; base < m; base += FL_EXACT_CHUNK) {
            const int cnt = min(FL_EXACT_CHUNK, m - base);
            bool bad = false;
            for (int k = tid; k < cnt; k += nt) {
                const float e = seg[1 + base + k];
                bad |= !(e >= 0.0f);
                scr[k] = e;
            }
            for (int k = cnt + tid; k < cnt + FL_CHAIN_STEP; k += nt) scr[k] = 0.0f;
            const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
            f = fl_chain_f32_block(scr, cnt, f, any_bad);       // (ends with a barrier; the result is in every thread)
        }
    }
    return f;
}

// The running sum over the patches of ALL ranks (see FlVioExact): thread 0 waits for the carry of rank-1, the workgroup adds this
// rank's patches, thread 0 passes the result on / collects the total. slot 0/1 (current pass) or 2/3 (last accepted pass).
// Returns the total (valid in thread 0).
__device__ __forceinline__ float vio_exact_chain(const FlVioExact &ex, const unsigned long long *w, unsigned wtag, int slot, float *carry_lds,
                                                 int *timeout_flag)
{
    const size_t mail = 2 * (size_t)ex.world * FL_XCHG_WORDS;
    if (threadIdx.x == 0) {
        float c = 0.0f;
        if (ex.world > 1 && ex.rank > 0) {
            unsigned long long v = 0ull;
            int spin = 0;
            do { v = __hip_atomic_load(ex.own + mail + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if ((unsigned)v != ex.xe) __builtin_amdgcn_s_sleep(2); }
            while ((unsigned)v != ex.xe && ++spin < FL_XCHG_SPIN_LIMIT);
            if ((unsigned)v != ex.xe) *timeout_flag = 1;
            c = __uint_as_float((unsigned)(v >> 32));
        }
        *carry_lds = c;
    }
    __syncthreads();
    float f = vio_exact_sum(w, ex.m, wtag, ex.scratch, timeout_flag, *carry_lds);
    if (ex.world > 1 && threadIdx.x == 0) {
        if (ex.rank < ex.world - 1) {
            __hip_atomic_store(ex.peer[ex.rank + 1] + mail + slot, ((unsigned long long)__float_as_uint(f) << 32) | (unsigned long long)ex.xe, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned long long v = 0ull;
            int spin = 0;
            do { v = __hip_atomic_load(ex.own + mail + slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if ((unsigned)v != ex.xe) __builtin_amdgcn_s_sleep(2); }
            while ((unsigned)v != ex.xe && ++spin < FL_XCHG_SPIN_LIMIT);
            if ((unsigned)v != ex.xe) *timeout_flag = 1;
            f = __uint_as_float((unsigned)(v >> 32));
        } else {
            for (int r = 0; r < ex.world - 1; r++)
                __hip_atomic_store(ex.peer[r] + mail + slot + 1, ((unsigned long long)__float_as_uint(f) << 32) | (unsigned long long)ex.xe, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return f;
}

// vec = x_prop (-) x: additive part by wave 1, Log(R^T R_prop) (common_lib.h:354-358, so3_math.h:75-81) by one lane of wave 2.
__device__ __forceinline__ void eskf18_form_vec(FlSolveLds &L)
{
    const int tid = threadIdx.x;
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

// Stage 0 (before the gather): stage the solve inputs in LDS and form vec = x_prop (-) x.
// Split in two so that the loads can be issued before the kernel's control-word round trip.
__device__ __forceinline__ double eskf18_prefetch_issue(const FlDev18 *__restrict__ D)
{
    // ONE 8-byte load for the 192 doubles and ONE 4-byte load for the eleven scalars, each lane at its own offset. (As an if / else chain
    // over the fields -- the form of rounds 1-5 -- the compiler emitted one masked load + s_waitcnt per field: fifteen round trips to the L2
    // one after the other, ~6 us at the head of every launch, longer than the producers' first pass beside it.)
    const int tid = threadIdx.x;
    const char *base = reinterpret_cast<const char *>(D);
    double v = 0.0;
    if (tid < 192) {
        size_t off = offsetof(FlDev18, Q) + 8 * (size_t)tid;
        off = tid >= 36 ? offsetof(FlDev18, T) + 8 * (size_t)(tid - 36) : off;
        off = tid >= 144 ? offsetof(FlDev18, x) + 8 * (size_t)(tid - 144) : off;
        off = tid >= 168 ? offsetof(FlDev18, xprop) + 8 * (size_t)(tid - 168) : off;
        v = *reinterpret_cast<const double *>(base + off);
    } else if (tid < 203) {
        const int k = tid - 192;
        size_t off = offsetof(FlDev18, last_error);
        off = k == 1 ? offsetof(FlDev18, rematch_num) : off;
        off = k == 2 ? offsetof(FlDev18, iterCount) : off;
        off = k == 3 ? offsetof(FlDev18, max_iter) : off;
        off = k == 4 ? offsetof(FlDev18, iters_run) : off;
        off = k == 5 ? offsetof(FlDev18, accepted) : off;
        off = k == 6 ? offsetof(FlDev18, status) : off;
        off = k == 7 ? offsetof(FlDev18, err_acc_buf) : off;
        off = k == 8 ? offsetof(FlDev18, err_acc_epoch) : off;
        off = k == 9 ? offsetof(FlDev18, last_exact_valid) : off;
        off = k == 10 ? offsetof(FlDev18, last_exact) : off;
        const unsigned raw = *reinterpret_cast<const unsigned *>(base + off);
        const double as_int = (double)(int)raw, as_uint = (double)raw, as_float = (double)__uint_as_float(raw);
        v = (k == 0 || k == 10) ? as_float : (k == 8 ? as_uint : as_int);      // (float: last_error, last_exact; unsigned: err_acc_epoch)
    }
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
    else if (tid == 198) { L.sticky = (int)v; L.fragile = (int)v & 16; }
    else if (tid == 199) L.acc_buf = (int)v;
    else if (tid == 200) L.acc_epoch = (unsigned)v;
    else if (tid == 201) L.last_exact_valid = (int)v;
    else if (tid == 202) L.last_exact = (float)v;
    else if (tid == 203) { L.jpass = 0; L.jflag = 0; L.spec_pending = 0; L.levels = 0; L.pb = 0; L.xl_pending = 0; L.xl_restart = 0; L.li_base = nullptr; }
    __syncthreads();
    eskf18_form_vec(L);
    __syncthreads();
}
__device__ __forceinline__ void eskf18_prefetch(const FlDev18 *__restrict__ D, FlSolveLds &L)
{
    eskf18_prefetch_commit(eskf18_prefetch_issue(D), L);
}

// Operands of the solve in the registers of wavefront 0 (other wavefronts load them too -- cheap -- and never use them).
// Loaded after eskf18_prefetch_commit / eskf18_restage, i.e. BEFORE the gather, and live across it.
struct FlSolveRegs {
    double trow[6];     // lane r < 18: row r of T
    double vecr;        // lane r < 18: vec[r]
    double xl;          // lane l < 24: x[l]
    double rrow[3];     // lane l < 9 : row (l / 3) of R
    float last_error;   // VIO: last_error as it stands BEFORE this pass
    double rci[3], pci; // VIO, lane t < 12: the row of Rci (and the element of Pci) its camera-pose element needs (vio_cam_element)
};
__device__ __forceinline__ constexpr int fl_tri(int i, int j) { return i * (i + 1) / 2 + j; }   // j <= i
__device__ __forceinline__ void eskf18_load_regs(const FlSolveLds &L, FlSolveRegs &G, const FlVioConst *__restrict__ VC = nullptr)
{
    const int lane = threadIdx.x & 63;
    G.rci[0] = G.rci[1] = G.rci[2] = 0.0; G.pci = 0.0;
    if (VC) {
        const int t = lane < 12 ? lane : 0;
        const int i = t < 9 ? t / 3 : t - 9;
#pragma unroll
        for (int k = 0; k < 3; k++) G.rci[k] = VC->Rci[i * 3 + k];
        G.pci = VC->Pci[i];
    }
    const int r = lane < 18 ? lane : 0;
#pragma unroll
    for (int c = 0; c < 6; c++) G.trow[c] = L.T[r * 6 + c];
    G.vecr = L.vec[r];
    G.xl = L.x[lane < 24 ? lane : 0];
    const int ri = lane < 9 ? lane / 3 : 0;
#pragma unroll
    for (int k = 0; k < 3; k++) G.rrow[k] = L.x[ri * 3 + k];
    G.last_error = L.last_error;
}

__device__ __forceinline__ void fl_bcast_store(unsigned long long *bcast, int idx, double v, unsigned bepoch)
{
    const unsigned long long lo = ((unsigned long long)f64_lo(v) << 32) | (unsigned long long)bepoch;
    const unsigned long long hi = ((unsigned long long)f64_hi(v) << 32) | (unsigned long long)bepoch;
#pragma unroll
    for (int r = 0; r < FL_BCAST_REPL; r++) {          // every copy (handoff.h): workgroup b polls copy b % FL_BCAST_REPL
        __hip_atomic_store(bcast + (size_t)r * FL_BCAST_STRIDE + 2 * idx, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(bcast + (size_t)r * FL_BCAST_STRIDE + 2 * idx + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void fl_bcast_ctrl_at(unsigned long long *bcast, int word, int ctrl, unsigned bepoch)
{
#pragma unroll
    for (int r = 0; r < FL_BCAST_REPL; r++)
        __hip_atomic_store(bcast + (size_t)r * FL_BCAST_STRIDE + word, ((unsigned long long)(unsigned)ctrl << 32) | (unsigned long long)bepoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fl_bcast_ctrl(unsigned long long *bcast, int ctrl, unsigned bepoch) { fl_bcast_ctrl_at(bcast, 24, ctrl, bepoch); }

// Reciprocal for the pivots: v_rcp_f64 + two Newton steps (relative error ~1e-16; the solve is compared by tolerance).
__device__ __forceinline__ double fl_rcp_nr(double d)
{
#pragma clang fp contract(fast)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}
// value of a wave-0 lane in every lane
__device__ __forceinline__ double fl_lane_bcast(double v, int src)
{
    const unsigned lo = __builtin_amdgcn_readlane(f64_lo(v), src);
    const unsigned hi = __builtin_amdgcn_readlane(f64_hi(v), src);
    return f64_make(lo, hi);
}
// `sqrt(t2) * k < thr` decided on t2 (no sqrt) unless t2 is within rounding distance of the boundary, where the exact expression runs
#define FL_NORM_BELOW(t2, k, thr) fl_norm_below_impl((t2), (k), (thr), ((thr) / (k)) * ((thr) / (k)), 1e-12 * (((thr) / (k)) * ((thr) / (k))))
__device__ __forceinline__ bool fl_norm_below_impl(double t2, double k, double thr, const double b /* compile-time */, const double band)
{
#pragma clang fp contract(off)
    if (fabs(t2 - b) > band) return t2 < b;
    asm volatile("" : "+v"(t2));          // keeps the compiler from speculating the (long) sqrt expansion into the common path
    return sqrt(t2) * k < thr;
}

// camera pose element `t` (0..8: Rcw, 9..11: Pcw) of the state xn = {rot(9), pos(3)}: Rcw = Rci Rwi^T, Pcw = -Rci Rwi^T Pwi + Pci
// (lidar_selection.cpp:780-784) in exactly the reference's operation order (it feeds the float sub-pixel weights).
__device__ __forceinline__ double vio_cam_element(int t, const double *xn, const double (&rci)[3], double pci)
{
#pragma clang fp contract(off)
    if (t < 9) {
        const int j = t % 3;              // Rcw[i][j] = sum_k Rci[i][k] * Rwi[j][k]
        return rci[0] * xn[j * 3 + 0] + rci[1] * xn[j * 3 + 1] + rci[2] * xn[j * 3 + 2];
    }
    double T[3];                          // T = (-Rci) Rwi^T ; Pcw = T Pwi + Pci
#pragma unroll
    for (int j = 0; j < 3; j++)
        T[j] = (-rci[0]) * xn[j * 3 + 0] + (-rci[1]) * xn[j * 3 + 1] + (-rci[2]) * xn[j * 3 + 2];
    return (T[0] * xn[9] + T[1] * xn[10] + T[2] * xn[11]) + pci;
}

// Solve + state update + judgement. All threads of the workgroup call it (NT >= 256); s_sums in LDS (complete: the gather
// ends with a barrier). G: eskf18_load_regs of THIS pass. FMA contraction is allowed in the solve (compared to the oracle by
// tolerance, never bitwise). bcast != nullptr (multi-pass kernels): wavefront 0 publishes the pose for the producers' next pass
// (LIO: R, p; VIO: the derived Rcw, Pcw) and the control word as self-validating words tagged bepoch.
// On return (after the CALLER's __syncthreads) L.ctrl, L.xn, L.xadd, L.cam are valid for everybody.
// The slow path of the VIO accept test (see eskf18_solve_block): the reference's float values of this pass's error and of the last
// accepted one into L.exact_cur / L.last_exact. Called by the whole workgroup on the rare fragile passes; NOT inlined -- its
// registers (staging, the chain) stay out of the pass loop's allocation, where they cost spills on every pass.
__device__ __forceinline__ void vio_exact_decide(const FlVioExact ex, FlSolveLds *Lp, float n_all /* 64 x the patches of ALL ranks */)
{
    FlSolveLds &L = *Lp;
    const int tid = threadIdx.x;
    __syncthreads();
    const int cur_buf = (L.iters_run + L.pb) & 1;
    bool audited = false;
    const unsigned long long *audit = ex.words + 2 * (size_t)ex.cap;
    if (ex.world <= 1) {     // single rank: the auditor workgroup has been adding this pass's chain up since its words arrived
        if (tid == 0) {
            float fc = 0.f, fl = 0.f;
            // An auditor that is on this pass or finishing the previous one is waited for: its remaining chain (<= ~7 us, resp. two of
            // them) is shorter than the two chains the workgroup would replay itself (~20 us: round 3 time line of ComputeJ -- on the
            // coarse pyramid levels the auditor was one pass late on every fragile pass and never waited for). One that is further
            // behind (many patches: its chain takes longer than a pass) is not.
            const unsigned at = (unsigned)__hip_atomic_load(audit + FL_AUDIT_RING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = (int)(at + 1u - ex.epoch) >= 0 && vio_audit_read(audit, ex.epoch, 1 << 12, &fc);
            if (ok && !L.last_exact_valid) {
                ok = vio_audit_read(audit, L.acc_epoch, 64, &fl);     // (a pass long finished: there, or never audited)
                if (ok) { L.last_exact = fl / n_all; L.last_exact_valid = 1; }
            }
            L.exact_cur = fc / n_all;
            L.audited = ok ? 1 : 0;
        }
        __syncthreads();
        audited = L.audited != 0;
    }
    if (!audited) {          // no auditor (sharded form: the chain runs through the ranks) or it gave up: the workgroup replays
        const float fc = vio_exact_chain(ex, ex.words + (size_t)cur_buf * ex.cap, ex.epoch, 0, &L.exact_cur, &L.exact_timeout);
        __syncthreads();
        if (tid == 0) L.exact_cur = fc / n_all;
        if (!L.last_exact_valid && L.acc_buf != cur_buf) {   // (same half: only when forced passes ran on after a rejection)
            const float fl = vio_exact_chain(ex, ex.words + (size_t)L.acc_buf * ex.cap, L.acc_epoch, 2, &L.last_exact, &L.exact_timeout);
            __syncthreads();
            if (tid == 0) { L.last_exact = fl / n_all; L.last_exact_valid = 1; }
        }
    }
    __syncthreads();
}

// The judgement of a pass (wavefront 1 of the solver workgroup, see eskf18_solve_block): delta and the solve's status bits come out
// of LDS. Uniform arithmetic; lane 0 publishes the control word and writes the bookkeeping.
template <int KIND>
__device__ __forceinline__ void eskf18_judge(FlDev18 *__restrict__ D, const double *s_sums, FlSolveLds &L, unsigned long long *bcast, unsigned bepoch)
{
    const int lane = threadIdx.x & 63;
    const double d0 = L.delta[0], d1 = L.delta[1], d2 = L.delta[2], d3 = L.delta[3], d4 = L.delta[4], d5 = L.delta[5];
    const double t2 = d0 * d0 + d1 * d1 + d2 * d2;
    const double p2 = d3 * d3 + d4 * d4 + d5 * d5;
    const int st = L.st;
    if (KIND == FL_EPI_LIO) {
        // laserMapping.cpp:1688-1728
        const int converged = (FL_NORM_BELOW(t2, 57.3, 0.01) && FL_NORM_BELOW(p2, 100.0, 0.015)) ? 1 : 0;
        int rematch = L.rematch, need_search = 0, stop = 0;
        const int it = L.iterCount, iters = L.iters_run + 1;
        if (converged || ((rematch == 0) && (it == (L.max_iter - 2)))) { need_search = 1; rematch++; }
        if (rematch >= 2 || (it == L.max_iter - 1)) stop = 1;
        const int ctrl = (stop ? 1 : 0) | (need_search ? 2 : 0);
        const int neff_lt1 = (s_sums[FL_S_NEFF] < 1.0) ? 4 : 0;
        if (lane == 0) {
            if (bcast) fl_bcast_ctrl(bcast, ctrl, bepoch);
            L.rematch = rematch; L.iterCount = it + 1; L.iters_run = iters;
            L.ctrl = ctrl;
            D->converged = converged;
            D->rematch_num = rematch;
            D->need_search = need_search;
            D->stop = stop;
            D->iterCount = it + 1;
            D->iters_run = iters;
            D->neff = (int)s_sums[FL_S_NEFF];
            D->total_residual = s_sums[FL_S_RES];
            L.sticky |= st | neff_lt1;
            D->status = L.sticky;
        }
    } else {
        // lidar_selection.cpp:883-899
        int stop = (FL_NORM_BELOW(t2, (double)57.3f, (double)0.001f) && FL_NORM_BELOW(p2, (double)100.0f, (double)0.001f)) ? 1 : 0;
        const int converged = stop;
        const int it = L.iters_run + 1;
        if (it >= L.max_iter) stop = 1;
        const int accepted = L.accepted + 1;
        if (lane == 0) {
            if (bcast) fl_bcast_ctrl(bcast, stop ? 1 : 0, bepoch);
            L.accepted = accepted; L.iters_run = it;
            L.ctrl = stop ? 1 : 0;
            D->converged = converged;
            D->accepted = accepted;
            D->iters_run = it;
            D->stop = stop;
            D->neff = (int)s_sums[FL_S_NEFF];
            D->total_residual = (double)L.last_error;
            L.sticky |= st | L.fragile;
            D->status = L.sticky;
        }
    }
}

// SPEC (VIO, vio_multipass_kernel<1, 1> only): a fragile ACCEPT may go ahead on the fp64 decision (see `spec` below). A template
// parameter, not a run-time flag: the forced passes of the headline and every other caller keep the code they had (a handful of extra
// uniform values in this function cost the VIO multi-pass kernel 0.9 us per pass -- it lives at its SGPR limit).
template <int KIND, int SPEC = 0>
__device__ __forceinline__ void eskf18_solve_block(FlDev18 *__restrict__ D, const double *s_sums, FlSolveLds &L, const FlSolveRegs &G_in,
                                                   int gather_status, unsigned long long *bcast = nullptr, unsigned bepoch = 0u,
                                                   const FlVioExact ex = FlVioExact{}, const FlVioConst *__restrict__ VC = nullptr, int dbg = 0)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    FlSolveRegs G = G_in;

    // ---- a hand-off timed out: abandon the pass. Nothing of the state moves; the launch chain ends (stop), the host resumes.
    if (gather_status) {
        if (wave == 0) {
            if (lane < 12) {
                L.xn[lane] = L.x[lane];
                if (bcast) fl_bcast_store(bcast, lane, (KIND == FL_EPI_VIO) ? ((lane < 9) ? D->Rcw[lane] : D->Pcw[lane - 9]) : L.x[lane], bepoch);
            }
            if (lane >= 9 && lane < 24) L.xadd[lane - 9] = L.x[lane];
            if (lane == 0) {
                L.sticky |= FL_NUM_TIMEOUT;
                D->status = L.sticky;                 // every kernel of the chain checks this bit first: nothing else runs (D->stop stays as it is)
                L.ctrl = 1 | 4;
                if (bcast) fl_bcast_ctrl(bcast, 1 | 4, bepoch);
                else D->resume_count = 1;             // one launch per pass: THIS pass is still to do (the multi-pass loops write count - p)
            }
        }
        return;
    }

    if (wave == 3 && lane < FL_SUMS18) D->sums[lane] = s_sums[lane];

    bool slow = false, need_exact = false, spec = false;
    if (KIND == FL_EPI_VIO) {
        // error = sum(res^2) / n_meas ; accept iff error <= last_error (lidar_selection.cpp:857-861).
        // The reference forms `error` as a FLOAT running sum: per patch `patch_error += res*res` over its 64 pixels, then
        // `error += patch_error` over the patches in order (:849-857). Its last digits depend on that order, which no tree reduction
        // reproduces, and near convergence the test is decided inside that noise. So: the producers leave every patch's
        // `patch_error` exactly as the reference rounds it (vio_produce), the fast test uses the fp64-reduced sum, and whenever it is
        // closer than (m + 8) * 2^-24 relative (the worst-case distance between the two sums) the decision is taken on the reference's
        // running sum over those per-patch floats, for this pass and for the last accepted one (status bit 16 reports that the slow
        // path ran). Single rank: the auditor workgroup of the pass kernels has been adding that chain up beside the pass
        // (vio_kernels.h vio_audit_pass; exact_chain.h: bit-identical to the m dependent additions, lane-parallel) and the solver reads
        // its ring; if the auditor is not there yet or gave up, the workgroup replays the chain itself (vio_exact_decide).
        // With the patches spread over ranks (in-kernel exchange) the chain runs through the ranks (vio_exact_chain); without
        // the per-patch words AND without the all-gathered floats (`ex.flat`, the RCCL / torch form's exact path: vio_exact_flat_sum on every
        // pass) -- i.e. only fl_vio_solve without the gather -- bit 16 means "may differ".
        // Every thread evaluates the trigger itself from pass-invariant inputs (G.last_error was read before the gather), so the
        // common case needs no barrier.
        bool can_replay = ex.words != nullptr && ex.enabled;     // (by value: a nullable pointer to it kept the struct in scratch)
#ifdef FL_AB_NO_EXACT
        can_replay = false;                                            // A/B build: what the replays cost (tools/computej_breakdown.py)
#endif
        const float n_meas = (float)s_sums[FL_S_NEFF];
        const float error = (float)s_sums[FL_S_RES] / n_meas;
        const float last = G.last_error;
        // worst-case distance between this fp64-reduced value and the reference's float running sum, for both operands:
        // m additions of at most half an ulp each, plus the casts and the division
        const float thr = (n_meas * (1.0f / 64.0f) + 8.0f) * 5.9604645e-8f;
        need_exact = (last < 1e9f && fabsf(error - last) <= thr * fabsf(error));
        slow = need_exact && can_replay;
        // Round 5: a fragile pass the fp64 test ACCEPTS does not wait for the float chain (the auditor's total arrives ~6 us after the
        // records): it solves and publishes the pose at once, the producers start the next pass, and the chain's verdict is taken when
        // it is there -- after the NEXT gather, or at the end of the launch (vio_spec_confirm). Until then the pass's writes that a
        // rejection would have to undo (old_state, sums_acc, solution, the error bookkeeping) stay in LDS. Needs the exact value of the
        // last accepted error (it always is there: every confirmed pass leaves it). A fragile REJECT still waits -- it ends the level.
        if constexpr (SPEC != 0) {
            const bool cand = slow && error <= last;
            if (cand && !L.last_exact_valid) {      // (uniform) the exact value of the last accepted error: out of the auditor's ring, as vio_exact_decide takes it
                if (tid == 0) {
                    float fl = 0.f;
                    if (vio_audit_read(ex.words + 2 * (size_t)ex.cap, L.acc_epoch, 64, &fl)) { L.last_exact = fl / n_meas; L.last_exact_valid = 1; }
                }
                __syncthreads();
            }
            spec = cand && L.last_exact_valid;
        }
        if (spec) {
            slow = false;
            FL_INSTR(if (tid == 0) g_fl_wall[2040]++;)        // (debug build: speculated / confirmed / rolled back, tools/fuzz_vio_spec.py)
            if (tid == 0) {
                L.spec_pending = 1; L.spec_epoch = ex.epoch; L.spec_buf = (L.iters_run + L.pb) & 1; L.spec_nall = n_meas;
                L.spec_last_error = L.last_error; L.spec_last_exact = L.last_exact; L.spec_accepted = L.accepted; L.spec_iters = L.iters_run;
                L.spec_acc_buf = L.acc_buf; L.spec_acc_epoch = L.acc_epoch;
            }
        }
        if (slow && tid == 0) L.exact_timeout = 0;
        if (slow) {      // uniform over the workgroup
            vio_exact_decide(ex, &L, (float)s_sums[FL_S_NEFF]);
            eskf18_load_regs(L, G, VC);      // the operands come back from LDS: nothing of them has to survive the call in registers
        }
        if (ex.flat != nullptr && ex.enabled) {      // all-gathered per-patch floats: the reference's chain on EVERY pass (uniform)
            __syncthreads();
            const float fc = vio_exact_flat_sum(ex.flat, ex.flat_stride, ex.world, ex.scratch);
            if (tid == 0) { L.exact_cur = fc / n_meas; L.exact_timeout = 0; }
            __syncthreads();
            slow = true;                     // every accepted pass stores an exact last_error: last_exact_valid stays 1 from the begin on
            eskf18_load_regs(L, G, VC);
        }
    }
    if (wave >= 2) return;
    // Wavefront 0 solves; WAVEFRONT 1 judges beside it (round 3): the rematch / stop decision and the control word the producers of
    // the next pass wait for need |delta_rot|, |delta_pos| and the loop counters only, and on wavefront 0 they sat between delta and
    // the pose (0.3 us of every pass). Wavefront 0 drops delta and its status bits into LDS and raises L.jflag (LDS operations of a
    // wavefront are performed in order); wavefront 1 spins on the flag, judges, publishes the control word and does the bookkeeping
    // that follows from the judgement, while wavefront 0 forms R Exp(delta) and publishes the pose (A/B against judging on wavefront 0
    // behind the pose: 6.08 vs 6.15 us per LIO pass, 7.15 vs 7.27 VIO). Nobody writes L.jpass before
    // wavefront 0 has raised the flag, so both read the same tag.
    const int jtag = L.jpass + 1;
    if (wave == 1) {
        int f = 0;
        for (int spin = 0; spin < (1 << 22); spin++) {
            f = __hip_atomic_load(&L.jflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (f == jtag || f == -jtag) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (f == jtag) eskf18_judge<KIND>(D, s_sums, L, bcast, bepoch);
        if (lane == 0) L.jpass = jtag;
        return;
    }

    // =========================================================================== wavefront 0 from here on, no workgroup barrier
    int accept = 1, vio_exact = 0;
    float vio_error = 0.f;
    if (KIND == FL_EPI_VIO) {
        const float n_meas = (float)s_sums[FL_S_NEFF];
        const float e_fast = (float)s_sums[FL_S_RES] / n_meas;
        const bool exact = slow && !L.exact_timeout && L.last_exact_valid;
        const float error = exact ? L.exact_cur : e_fast;
        const float last = exact ? L.last_exact : G.last_error;
        accept = (error <= last) ? 1 : 0;
        __builtin_amdgcn_wave_barrier();           // every lane has read the fields lane 0 rewrites below
        vio_error = error; vio_exact = exact ? 1 : 0;
        if (lane == 0) {       // (the LDS mirrors now -- the judgement reads them; the device block behind the pose, see the tail)
            if (need_exact) L.fragile = 16;
            L.accept = accept;
            if (accept) {
                L.last_error = error;
                L.acc_buf = (L.iters_run + L.pb) & 1; L.acc_epoch = ex.epoch;
                L.last_exact_valid = vio_exact; L.last_exact = error;
            }
        }
        if (!accept) {   // revert: state = old_state ; EKF_end (:888-892)
            if (lane == 0) D->error = error;
            double xo = 0.0;
            if (lane < 24) {
                xo = D->xold[lane];
                D->x[lane] = xo;
                if (lane < 12) L.xn[lane] = xo;
                if (lane >= 9) L.xadd[lane - 9] = xo;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (VC && lane < 12) {
                const double ce = vio_cam_element(lane, L.xn, G.rci, G.pci);
                if (lane < 9) D->Rcw[lane] = ce; else D->Pcw[lane - 9] = ce;
                L.cam[lane] = ce;
                if (bcast) fl_bcast_store(bcast, lane, ce, bepoch);
            }
            if (lane == 0) {
                __hip_atomic_store(&L.jflag, -jtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // wavefront 1: nothing to judge
                const int it = L.iters_run + 1;
                L.iters_run = it;
                D->iters_run = it;
                D->stop = 1;
                L.ctrl = 1;
                D->converged = 1;
                D->neff = (int)s_sums[FL_S_NEFF];
                D->total_residual = (double)G.last_error;
                L.sticky |= L.fragile;
                D->status = L.sticky;
                if (bcast) fl_bcast_ctrl(bcast, 1, bepoch);
            }
            return;
        }
    }

    double dl, z[6];
    int bad = 0;
    FL_INSTR(fl_stamp(dbg, 30);)
    {
#pragma clang fp contract(fast)
        const double sign = (KIND == FL_EPI_VIO) ? -1.0 : 1.0;
        double s[27];
#pragma unroll
        for (int k = 0; k < 27; k++) s[k] = s_sums[k];             // LDS broadcast reads
        // S(i,j), i <= j, row-major upper triangle: index i*6 - i*(i-1)/2 + (j-i)
#define FL_SIDX(i, j) ((i) <= (j) ? ((i) * 6 - (i) * ((i) - 1) / 2 + ((j) - (i))) : ((j) * 6 - (j) * ((j) - 1) / 2 + ((i) - (j))))
        double c[6][6];        // lower triangle of C = Q + S
        double w[6];           // right-hand side, eliminated along with C as a 7th row
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) c[i][j] = L.Q[i * 6 + j] + s[FL_SIDX(i, j)];      // (Q, vec6: LDS broadcast reads too --
            double b = sign * s[FL_S_HTZ + i];                                              //  held in registers across the gather they
#pragma unroll                                                                              //  cost the kernel half its occupancy)
            for (int k = 0; k < 6; k++) b = fma(-s[FL_SIDX(i, k)], L.vec[k], b);
            w[i] = b;
        }
#undef FL_SIDX
        FL_INSTR(if (dbg) { double keep = w[5] + c[5][0]; asm volatile("" ::"v"(keep)); fl_stamp(dbg, 31); })
        // right-looking LDL^T on the augmented lower triangle: after column j, c[i][j] = L_ij and w carries D^-1 L^-1 b
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double dj = c[j][j];
            if (!(dj > 0.0)) bad = 1;
            const double inv = fl_rcp_nr(dj);
            double u[6];
#pragma unroll
            for (int i = j + 1; i < 6; i++) { u[i] = c[i][j]; c[i][j] = u[i] * inv; }
            const double wj = w[j] * inv;
            w[j] = wj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
#pragma unroll
                for (int k = j + 1; k < 6; k++)
                    if (k <= i) c[i][k] = fma(-c[i][j], u[k], c[i][k]);
                w[i] = fma(-wj, u[i], w[i]);
            }
        }
        FL_INSTR(if (dbg) { double keep = w[5] + c[5][4]; asm volatile("" ::"v"(keep)); fl_stamp(dbg, 32); })
        // back substitution L^T z = w
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            double zi = w[i];
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k > i) zi = fma(-c[k][i], z[k], zi);
            z[i] = zi;
        }
        dl = G.vecr;
#pragma unroll
        for (int cc = 0; cc < 6; cc++) dl = fma(G.trow[cc], z[cc], dl);
    }
    FL_INSTR(if (dbg) { asm volatile("" ::"v"(dl)); fl_stamp(dbg, 33); })

    // ---- delta and the status bits of the solve to wavefront 1 (the judgement), then the state update
    {
        int st0 = bad;
        if (__ballot(lane < 18 && !(fabs(dl) <= DBL_MAX)) != 0ull) st0 |= 2;
        if (lane < 18) L.delta[lane] = dl;
        if (lane == 18) L.st = st0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (compiler order only: the LDS performs a wavefront's operations in order)
        if (lane == 0) __hip_atomic_store(&L.jflag, jtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // lane l < 9 forms R(l/3, l%3) of R * Exp(d0,d1,d2) (so3_math.h:54-72, common_lib.h:345), lanes 9..23 add
    const double d0 = fl_lane_bcast(dl, 0), d1 = fl_lane_bcast(dl, 1), d2 = fl_lane_bcast(dl, 2);
    const double t2 = d0 * d0 + d1 * d1 + d2 * d2;

    double xnew = G.xl;
    {
#pragma clang fp contract(fast)
        // the reference leaves R alone unless |d| > 1e-5
        bool rotate;
        if (fabs(t2 - 1e-10) > 1e-22) rotate = t2 > 1e-10;
        else { double tt = t2; asm volatile("" : "+v"(tt)); rotate = sqrt(tt) > 0.00001; }   // (asm: no speculation of the sqrt)
        if (lane < 9) {
            if (rotate) {
                double a, b;       // a = sin(t)/t, b = (1 - cos t)/t^2
                if (t2 <= 0.25) {
                    double pa = 1.0 / 1307674368000.0;
                    pa = fma(pa, -t2, 1.0 / 6227020800.0);
                    pa = fma(pa, -t2, 1.0 / 39916800.0);
                    pa = fma(pa, -t2, 1.0 / 362880.0);
                    pa = fma(pa, -t2, 1.0 / 5040.0);
                    pa = fma(pa, -t2, 1.0 / 120.0);
                    pa = fma(pa, -t2, 1.0 / 6.0);
                    a = fma(pa, -t2, 1.0);
                    double pb = 1.0 / 20922789888000.0;
                    pb = fma(pb, -t2, 1.0 / 87178291200.0);
                    pb = fma(pb, -t2, 1.0 / 479001600.0);
                    pb = fma(pb, -t2, 1.0 / 3628800.0);
                    pb = fma(pb, -t2, 1.0 / 40320.0);
                    pb = fma(pb, -t2, 1.0 / 720.0);
                    pb = fma(pb, -t2, 1.0 / 24.0);
                    b = fma(pb, -t2, 0.5);
                } else {
                    const double t = sqrt(t2);
                    a = sin(t) / t;
                    b = (1.0 - cos(t)) / t2;
                }
                // E = I + a K + b K^2, K = skew(d): K^2 = d d^T - t2 I.  Lane (i, j): sum_k R(i,k) E(k,j)
                const int j = lane % 3;
                const double dj = (j == 0) ? d0 : ((j == 1) ? d1 : d2);
                const double diag = 1.0 - b * t2;
                // column j of E
                double e0 = b * d0 * dj, e1 = b * d1 * dj, e2 = b * d2 * dj;
                if (j == 0) { e0 += diag; e1 += a * d2; e2 -= a * d1; }
                else if (j == 1) { e0 -= a * d2; e1 += diag; e2 += a * d0; }
                else { e0 += a * d1; e1 -= a * d0; e2 += diag; }
                xnew = G.rrow[0] * e0 + G.rrow[1] * e1 + G.rrow[2] * e2;
            }
        }
    }
    // additive states: lane 9 + k adds delta[3 + k]; delta lives in lane 3 + k, six lanes further down
    {
        const double dsh = __shfl(dl, (lane >= 6) ? lane - 6 : 0, FL_WAVE);
        if (lane >= 9 && lane < 24) xnew = G.xl + dsh;
    }
    FL_INSTR(if (dbg) { asm volatile("" ::"v"(xnew)); fl_stamp(dbg, 34); })
    if (lane < 24) {
        if (lane < 12) {
            if (KIND == FL_EPI_LIO && bcast) fl_bcast_store(bcast, lane, xnew, bepoch);     // the pose words first: the producers wait for them
            L.xn[lane] = xnew;
        }
        D->x[lane] = xnew;
        if (lane >= 9) L.xadd[lane - 9] = xnew;
    }
    if (KIND == FL_EPI_VIO) {
        // the derived camera pose of the new state for the next pass's producers
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (VC && lane < 12) {
            const double ce = vio_cam_element(lane, L.xn, G.rci, G.pci);
            if (bcast) fl_bcast_store(bcast, lane, ce, bepoch);
            if (lane < 9) D->Rcw[lane] = ce; else D->Pcw[lane - 9] = ce;
            L.cam[lane] = ce;
        }
    }
    FL_INSTR(fl_stamp(dbg, 36);)
    // ---- off the hand-off's critical path: what the host, the next pass's solver and the finish kernels read
    if (KIND == FL_EPI_VIO && SPEC != 0 && spec) {      // (uniform) an unconfirmed accept: these wait in LDS for the chain's verdict (vio_spec_confirm)
        if (lane < 18) L.def_sol[lane] = dl;
        if (lane < FL_SUMS18) L.def_sums[lane] = s_sums[lane];
        if (lane < 24) L.def_xold[lane] = G.xl;
        return;
    }
    if (lane < 18) D->solution[lane] = dl;
    if (lane < FL_SUMS18) D->sums_acc[lane] = s_sums[lane];   // LIO: last executed pass ; VIO: last accepted pass
    if (KIND == FL_EPI_VIO && lane < 24) D->xold[lane] = G.xl;   // old_state = *state (:863)
    if (KIND == FL_EPI_VIO && lane == 0) {                         // the accepted pass's error bookkeeping (mirrors set before the solve)
        D->error = vio_error; D->last_error = vio_error;
        D->err_acc_buf = L.acc_buf; D->err_acc_epoch = ex.epoch; D->last_exact_valid = vio_exact; D->last_exact = vio_error;
    }
}

// The float chain's verdict on a fragile accept that went ahead (eskf18_solve_block, `spec`). Whole workgroup, barriers inside; called
// by vio_multipass_kernel's solver after the NEXT pass's gather (bcast != nullptr: the producers of that pass wait for a control word)
// or when the launch ends right behind the speculative pass (bcast == nullptr). Confirmed (the usual case): the deferred writes
// land, with the chain's value as last_error, exactly what the waiting form leaves. Rejected: the state goes back to old_state and the
// level ends as lidar_selection.cpp:888-892 ends it -- the iteration count, the error, the per-patch errors (restored from the pass's own
// half of the word buffer when a later pass has overwritten the array) are those of the rejecting pass. Returns 1 if it rolled back.
// the two rare branches of vio_spec_confirm, out of line and with scalar arguments only (a by-value FlVioExact went to scratch at the
// call site and cost the kernel its second wavefront per SIMD; everything inlined cost every pass 1.8 us of register shuffling)
__device__ __attribute__((noinline)) void vio_spec_replay(FlSolveLds *Lp, const unsigned long long *words, int cap, int m, float *scratch)
{
    FlSolveLds &L = *Lp;
    FlVioExact ex{};
    ex.words = words; ex.m = m; ex.cap = cap; ex.epoch = L.spec_epoch; ex.scratch = scratch; ex.enabled = 1; ex.own = nullptr; ex.peer = nullptr;
    ex.rank = 0; ex.world = 1; ex.xe = 0u;
    const float fc = vio_exact_chain(ex, words + (size_t)L.spec_buf * cap, L.spec_epoch, 0, &L.exact_cur, &L.exact_timeout);
    __syncthreads();
    if (threadIdx.x == 0) L.exact_cur = fc;
    __syncthreads();
}
__device__ __attribute__((noinline)) void vio_spec_rollback(FlDev18 *D, FlSolveLds *Lp, const unsigned long long *words, int cap, int m, float *errors,
                                                           const FlVioConst *VC, unsigned long long *bcast, unsigned bepoch)
{
    FlSolveLds &L = *Lp;
    const int tid = threadIdx.x, lane = tid & 63;
    const float exact = L.exact_cur;
    FlSolveRegs G;
    eskf18_load_regs(L, G, VC);
    if (L.xl_pending) {      // (uniform) the rejected pass had ENDED its pyramid level and the next level has begun on its state
        // The finished level ends as lidar_selection.cpp:888-892 ends it at that pass -- its result block says so --, the state goes back
        // to that level's old_state, and the level that had begun begins again from there: old_state = *state (:747), the counters are
        // still those of its prologue (its first pass was gathered, never solved). The producers wait for that pass's broadcast: they
        // get the reverted pose and "start the level again" (ctrl bit 4).
        FlVioLevelInfo *li = L.li_base + L.xl_level;
        FL_INSTR(if (tid == 0) g_fl_wall[2043]++;)           // (debug build: cross-level roll-backs, tools/fuzz_vio_spec.py)
        if (tid < 24) {
            const double xo = L.xl_xold[tid];
            D->x[tid] = xo; D->xold[tid] = xo;
            if (tid < 12) L.xn[tid] = xo;
            if (tid >= 9) L.xadd[tid - 9] = xo;
        } else if (tid >= 64 && tid < 82) li->solution[tid - 64] = D->solution[tid - 64];      // (of the level's last CONFIRMED accept)
        __syncthreads();
        if (VC && tid < 12) {
            const double ce = vio_cam_element(lane, L.xn, G.rci, G.pci);
            if (tid < 9) D->Rcw[tid] = ce; else D->Pcw[tid - 9] = ce;
            L.cam[tid] = ce;
            if (bcast) fl_bcast_store(bcast, tid, ce, bepoch);
        }
        // (The per-patch error array is NOT put back to the rejecting pass's values here, as the same-level roll-back below does: the level
        // that starts again rewrites every entry with its first pass -- and plain stores from this workgroup would sit in this XCD's L2
        // beside the producers' later ones in theirs, two dirty copies of a line whose write-back order nobody defines. Found the hard
        // way: 7 of 8 workgroups' entries came back with the restored values.)
        if (tid == 0) {
            li->error = L.spec_last_error; li->iterations = L.spec_iters + 1; li->n_meas = L.xl_neff;
            li->accepted = L.spec_accepted; li->status = L.xl_status; li->converged = 1;
            L.spec_pending = 0; L.xl_pending = 0; L.xl_restart = 1;
        }
        __syncthreads();
        if (bcast && tid == 0) fl_bcast_ctrl(bcast, 16, bepoch);
        return;
    }
    if (tid < 24) {
        const double xo = D->xold[tid];              // old_state of the rejected pass: its deferred write never happened
        D->x[tid] = xo;
        if (tid < 12) L.xn[tid] = xo;
        if (tid >= 9) L.xadd[tid - 9] = xo;
    }
    __syncthreads();
    if (VC && tid < 12) {
        const double ce = vio_cam_element(lane, L.xn, G.rci, G.pci);
        if (tid < 9) D->Rcw[tid] = ce; else D->Pcw[tid - 9] = ce;
        L.cam[tid] = ce;
        if (bcast) fl_bcast_store(bcast, tid, ce, bepoch);
    }
    // bcast != nullptr: a later pass has run and the per-patch error array holds ITS values; it has to hold the rejecting pass's again. The
    // PRODUCERS put them back, each workgroup its own patches out of that pass's half of the word buffer, when they read the control word
    // below (bit 5, bit 6 = the half; vio_multipass_kernel). (Until round 6 this workgroup did it with plain stores: behind the
    // producers' own late stores of the dropped pass in time, perhaps, and in another XCD's L2 than theirs -- two dirty copies of a line.)
    if (tid == 0) {
        const int it = L.spec_iters + 1;
        D->error = exact;
        D->iters_run = it; L.iters_run = it;
        D->accepted = L.spec_accepted; L.accepted = L.spec_accepted;
        D->stop = 1; D->converged = 1;
        D->total_residual = (double)L.spec_last_error;
        L.last_error = L.spec_last_error; L.last_exact = L.spec_last_exact; L.last_exact_valid = 1;
        L.acc_buf = L.spec_acc_buf; L.acc_epoch = L.spec_acc_epoch;
        L.sticky |= L.fragile;
        D->status = L.sticky;
        L.ctrl = 1;
        L.spec_pending = 0;
        if (bcast) fl_bcast_ctrl(bcast, 1 | 32 | (L.spec_buf ? 64 : 0), bepoch);
    }
    __syncthreads();
}
__device__ __forceinline__ int vio_spec_confirm(FlDev18 *D, FlSolveLds *Lp, const unsigned long long *words, int cap, int m, float *scratch, float *errors,
                                                const FlVioConst *VC, unsigned long long *bcast, unsigned bepoch)
{
    FlSolveLds &L = *Lp;
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long *audit = words + 2 * (size_t)cap;
        const unsigned sepoch = L.spec_epoch;
        float fc = 0.f;
        const unsigned at = (unsigned)__hip_atomic_load(audit + FL_AUDIT_RING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = (int)(at + 1u - sepoch) >= 0 && vio_audit_read(audit, sepoch, 1 << 12, &fc);
        L.exact_cur = fc;
        L.audited = ok ? 1 : 0;
        L.exact_timeout = 0;
    }
    __syncthreads();
    if (!L.audited) vio_spec_replay(Lp, words, cap, m, scratch);      // the auditor is not there (or gave up on that pass): the workgroup adds the chain up itself
    const float exact = L.exact_cur / L.spec_nall;
    const bool accept = L.exact_timeout ? true : (exact <= L.spec_last_exact);      // (a chain that timed out: the fp64 decision stands, as in the waiting form)
    FL_INSTR(if (tid == 0) g_fl_wall[accept ? 2041 : 2042]++;)
    if (!accept) {       // the chain says the error went UP: revert (lidar_selection.cpp:888-892), the level ends at that pass --
        if (tid == 0) L.exact_cur = exact;      // the caller leaves its pass loop and calls vio_spec_rollback BEHIND it (a call site inside the loop costs every pass)
        __syncthreads();
        return 1;
    }
    if (L.xl_pending) {      // (uniform) the confirmed pass had ended its pyramid level: what that level's result block still lacked; old_state,
        FlVioLevelInfo *li = L.li_base + L.xl_level;      // last_error and the counters are the NEXT level's by now (its prologue ran)
        if (tid < 18) li->solution[tid] = L.def_sol[tid];
        else if (tid >= 64 && tid < 64 + FL_SUMS18) D->sums_acc[tid - 64] = L.def_sums[tid - 64];
        else if (tid >= 128 && tid < 146) D->solution[tid - 128] = L.def_sol[tid - 128];
        if (tid == 0) {
            li->error = L.exact_timeout ? L.xl_err : exact; li->iterations = L.xl_iters; li->n_meas = L.xl_neff;
            li->accepted = L.xl_accepted; li->status = L.xl_status; li->converged = L.xl_converged;
            L.spec_pending = 0; L.xl_pending = 0;
        }
        __syncthreads();
        return 0;
    }
    if (tid < 24) D->xold[tid] = L.def_xold[tid];
    else if (tid >= 64 && tid < 64 + FL_SUMS18) D->sums_acc[tid - 64] = L.def_sums[tid - 64];
    else if (tid >= 128 && tid < 146) D->solution[tid - 128] = L.def_sol[tid - 128];
    if (tid == 0) {
        const bool have = !L.exact_timeout;
        const float e = have ? exact : L.last_error;
        // (only if no later pass has been accepted since: there is none -- the verdict is taken before the next decision)
        D->error = e; D->last_error = e; D->err_acc_buf = L.acc_buf; D->err_acc_epoch = L.spec_epoch;
        D->last_exact_valid = have ? 1 : 0; D->last_exact = e; D->total_residual = (double)e;
        L.last_error = e; L.last_exact = e; L.last_exact_valid = have ? 1 : 0;
        L.spec_pending = 0;
    }
    __syncthreads();
    return 0;
}

// Multi-pass kernels: after eskf18_solve_block (and a __syncthreads) make the new state the solve input of the next
// pass without a global round trip: x <- x (+) delta from LDS, vec = x_prop (-) x as in eskf18_prefetch_commit.
// All threads; ends with a barrier (the registers of the next pass are loaded right after).
__device__ __forceinline__ void eskf18_restage(FlSolveLds &L)
{
    const int tid = threadIdx.x;
    if (tid < 9) L.x[tid] = L.xn[tid];
    else if (tid < 24) L.x[tid] = L.xadd[tid - 9];
    __syncthreads();
    eskf18_form_vec(L);
    __syncthreads();
}
