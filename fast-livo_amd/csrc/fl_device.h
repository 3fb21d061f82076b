// fl_device.h -- device-side data layout and small math for the gfx950 ESKF kernels.
//
// All code in csrc/ is compiled with -ffp-contract=off: the float part of the measurement model
// (world point, plane fit, pd2, gates) must round exactly like the reference's x86-64 build, which
// has no FMA (CMakeLists.txt:8 has no -march), because the selection gates are discontinuous.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define FL_WAVE 64
#define FL_BLOCK 256
#define FL_SUMS18 32
#define FL_SUMS23 96          /* the PUBLIC Mode-23 record (fl_ikfom_accumulate / fl_ikfom_solve, fl_h_share_model_sums) */
#define FL_SUMS23I 64         /* what the pass kernels hand from the producers to the solver: see fl_ikfom_math.h */

// Reduction record layout (Mode-18 and VIO), FL_SUMS18 doubles:
//   [0..20]  upper triangle of the 6x6 H^T H, row-major (i<=j)
//   [21..26] H^T z
//   [27]     number of effective measurements
//   [28]     LIO: sum |pd2|            VIO: sum res^2
//   [29]     LIO: sum pd2^2            VIO: unused
//   [30..31] zero
#define FL_S_HTZ 21
#define FL_S_NEFF 27
#define FL_S_RES 28
#define FL_S_RES2 29

// Device-resident block for the 18-state filter (one per handle, in HBM, ~6 KB).
struct FlDev18 {
    double x[24];       // rot(9, row-major) pos vel bg ba grav : current state
    double xprop[24];   // state_propagat
    double xold[24];    // VIO old_state (lidar_selection.cpp:747,863)
    double P[324];      // state.cov (constant during the iterations of a frame)
    double G6[108];     // G.block<18,6>(0,0), 18x6 row-major (formed by the finish kernels)
    double Q[36];       // (P66/meas_cov)^-1, per-frame prepare (fl_math.h fl_prepare18)
    double T[108];      // (P[:,0:6]/meas_cov) Q, 18x6 row-major
    double sums_acc[FL_SUMS18]; // record of the last executed (LIO) / accepted (VIO) pass: source of G
    double Rcw[9];      // VIO: camera pose derived from x, refreshed by whoever writes x (vio_derive_pose)
    double Pcw[3];
    double R_LI[9];
    double t_LI[3];
    double solution[24];
    double sums[FL_SUMS18];
    double total_residual;
    double meas_cov;    // LASER_POINT_COV or IMG_POINT_COV
    float last_error;   // VIO
    float error;        // VIO
    int32_t iterCount;  // LIO loop counter (starts at -1)
    int32_t rematch_num;
    int32_t need_search;
    int32_t stop;
    int32_t converged;
    int32_t neff;
    int32_t status;
    int32_t iters_run;
    int32_t accepted;
    int32_t max_iter;
    int32_t level;      // VIO pyramid level of the current UpdateState
    int32_t searched_at; // device k-NN: value of iters_run the last search was made for (-1: none)
    // VIO: exact emulation of the reference's float error sum (solve18.h). Per-patch errors of the passes of a level, ping-pong by
    // pass parity, each an 8-byte self-validating word (float bits << 32 | epoch of the pass that wrote it).
    unsigned long long *err_words;   // [2][err_cap], set by the host
    int32_t err_cap;
    int32_t err_acc_buf;             // half that holds the per-patch errors of the last ACCEPTED pass
    uint32_t err_acc_epoch;          // ... and the epoch they are tagged with
    int32_t last_exact_valid;        // last_exact is the reference's float value of last_error (not the fp64-reduced one)
    float last_exact;
    int32_t resume_count;            // passes an abandoned launch chain left undone (FL_NUM_TIMEOUT; solve18.h, fl_pass_skipped)
    // peer exchange of the sharded form (handoff.h peer_allreduce32): every rank's exchange buffer as this device addresses it
    unsigned long long *xchg_peer[8];
    unsigned *xchg_epoch;            // device word: epoch of the next exchange (same value on every rank)
    int32_t xchg_rank, xchg_world;   // world <= 1: no exchange
    // result mailbox of the frame drivers (fl_publish_state below): the frame's last kernel copies this block (+ its tail) into the
    // page-locked host mirror and then writes pub_seq to the host word pub_flag -- the host polls that word instead of enqueueing a
    // copy and synchronising the stream. nullptr: nobody polls (every other call reads the block back with a copy).
    unsigned long long *pub_flag;
    void *pub_dst;
    unsigned long long pub_seq;
    // fl_lidar_front (round 5): the scan's size is decided on the device (the voxel filter's count); the first search of the frame
    // leaves it here for the host (0: the host knew it)
    int32_t n_scan;
    // fl_vio_detect, fused form (api_vmap.inc): the number of selected patches is decided on the device (the selection's last kernel
    // writes it here); vio_multipass_kernel launched with FL_VIO_M_DEV sizes its passes from it instead of from its argument
    int32_t m_dev;
};

// What fl_lidar_front's kernels leave for the host in the tail behind FlDev18 (FL_DEV18_TAIL bytes, travels with the result mailbox)
// ... and what fl_vio_detect's fused form leaves there BEHIND the three FlVioLevelInfo of ComputeJ (offset FL_DETECT_TAIL_OFF)
struct FlDetectTail { int32_t n_cand, n_selected, n_added, n_observed, n_down, vox_cells_short, pad0, pad1; long long vox_cells; };
#define FL_DETECT_TAIL_OFF 512
struct FlFrontTail {
    double acc_s_last[3], angvel_last[3];      // ImuProcess members the next frame starts from (IMU_Processing.cpp:731-732)
    int32_t n_poses;
    int32_t vox_count, vox_leaf_too_small, vox_cells_short, vox_nfinite;
    int32_t unsorted;                          // undistort_sorted_kernel met a point earlier than its predecessor (or a NaN time): the frame is run again with the general kernels
    long long vox_cells;
};

// VIO constants (lidar_selection.cpp:35-59 + camera), computed on the host once per handle.
struct FlVioConst {
    double Rci[9], Pci[3], Jdphi_dR[9], Jdp_dR[9];
    double fx_abs, fy_abs;          // errorMultiplier2(), errorMultiplier()/(4 fx)
    double fx, fy, cx, cy, d[5];    // projection (world2cam)
    int32_t width, height, stride, distort;
};

// ---- 3x3 helpers (row-major, double) --------------------------------------------------------
__device__ __forceinline__ void m3_vec(const double *A, const double *v, double *o)
{
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void m3t_vec(const double *A, const double *v, double *o)
{
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void m3_mul(const double *A, const double *B, double *C)
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            T[i * 3 + j] = A[i * 3 + 0] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = T[i];
}
__device__ __forceinline__ void m3_tr(const double *A, double *T)
{
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[j * 3 + i] = A[i * 3 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = t[i];
}
__device__ __forceinline__ void skew3(const double *v, double *K)
{
    K[0] = 0.0;   K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2];  K[4] = 0.0;   K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0];  K[8] = 0.0;
}
// Exp(v1,v2,v3), so3_math.h:54-72
__device__ __forceinline__ void so3_Exp(double v1, double v2, double v3, double *R)
{
    double nrm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (nrm > 0.00001) {
        double r[3] = {v1 / nrm, v2 / nrm, v3 / nrm};
        double K[9], cK[9], cKK[9];
        skew3(r, K);
        double s = sin(nrm), c = 1.0 - cos(nrm);
        // `(1.0 - cos) * K * K` groups as ((1 - cos) K) K (so3_math.h:66)
#pragma unroll
        for (int i = 0; i < 9; i++) cK[i] = c * K[i];
        m3_mul(cK, K, cKK);
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = R[i] + s * K[i] + cKK[i];
    }
}
// Log(R), so3_math.h:75-81
__device__ __forceinline__ void so3_Log(const double *R, double *out)
{
    double tr = R[0] + R[4] + R[8];
    double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
    double K0 = R[7] - R[5], K1 = R[2] - R[6], K2 = R[3] - R[1];
    double f = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
    out[0] = f * K0; out[1] = f * K1; out[2] = f * K2;
}

// sin(x) and 1 - cos(x) of a rotation increment. The increments of an ESKF pass / an IMU interval are tiny (|x| << 0.5): a Taylor/Horner form
// (truncation < 1e-17 relative for |x| <= 0.5, and 1 - cos without cancellation) replaces two library calls of several hundred
// dependent cycles each on the solver workgroup's critical path; larger angles use the library. The solve is compared to the
// oracle by tolerance (1e-9), never bitwise.
__device__ __forceinline__ void fl_sin_omc(double x, double *s, double *omc)
{
    if (fabs(x) <= 0.5) {
        const double x2 = x * x;
        double ps = -1.0 / 1307674368000.0;
        ps = ps * x2 + 1.0 / 6227020800.0;
        ps = ps * x2 - 1.0 / 39916800.0;
        ps = ps * x2 + 1.0 / 362880.0;
        ps = ps * x2 - 1.0 / 5040.0;
        ps = ps * x2 + 1.0 / 120.0;
        ps = ps * x2 - 1.0 / 6.0;
        *s = x + x * (x2 * ps);
        double pc = 1.0 / 87178291200.0;
        pc = pc * x2 - 1.0 / 479001600.0;
        pc = pc * x2 + 1.0 / 3628800.0;
        pc = pc * x2 - 1.0 / 40320.0;
        pc = pc * x2 + 1.0 / 720.0;
        pc = pc * x2 - 1.0 / 24.0;
        pc = pc * x2 + 0.5;
        *omc = x2 * pc;
    } else {
        *s = sin(x);
        *omc = 1.0 - cos(x);
    }
}

// s_waitcnt vmcnt(0): this wavefront's outstanding vector-memory operations (loads AND stores on gfx9) have been acknowledged. The
// explicit wait in front of a barrier behind which ANOTHER wavefront raises a flag for stores of this one (the result mailboxes).
__device__ __forceinline__ void fl_wait_own_stores()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define FL_MAX_BLOCKS 1024                 /* largest pass grid (records buffer; fastlivo_hip.hip repeats the definition) */
#define FL_DEV18_TAIL 576                  /* bytes behind FlDev18 in its device and pinned-host allocations (fastlivo_hip.hip) */

// The result mailbox (FlDev18::pub_flag): called by ALL threads of ONE workgroup as the last action of a frame's last kernel. The
// block was last written by this workgroup or by earlier kernels; it travels as 8-byte words over the host link (~7 KB: well under a
// microsecond), every thread makes its words visible to the system, then one thread raises the flag. The host (read_info18) sees the
// flag ~1 us after the kernel's last store; against a device-to-host copy command plus a stream synchronisation that is 4-5 us less
// per frame driver call (tools/mailbox_ab.py).
// WORDS 8-byte words from device memory (agent-scope loads: what the workgroup itself just wrote) into the page-locked mirror, by a
// workgroup of >= 128 threads: every load in flight before the first store (as the plain word loop it was, each round of blockDim.x
// words waited for its loads before the next round's were issued -- three L2 round trips in a row behind every frame's last kernel).
template <int WORDS>
__device__ __forceinline__ void fl_publish_copy(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src)
{
    constexpr int PER = (WORDS + 127) / 128;
    unsigned long long v[PER];
    const int nt = (int)blockDim.x;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = (int)threadIdx.x + k * nt;
        v[k] = (i < WORDS) ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = (int)threadIdx.x + k * nt;
        if (i < WORDS) dst[i] = v[k];
    }
}
__device__ __forceinline__ void fl_publish_state(FlDev18 *__restrict__ D)
{
    __syncthreads();
    unsigned long long *flag = D->pub_flag;
    if (!flag) return;                                   // (uniform)
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(D);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(D->pub_dst);
    const unsigned long long seq = D->pub_seq;
    constexpr int WORDS = (int)((sizeof(FlDev18) + FL_DEV18_TAIL) / 8);
    fl_publish_copy<WORDS>(dst, src);
    // Every wavefront waits for ITS OWN mirror stores to be acknowledged (s_waitcnt vmcnt(0): on gfx9 the counter covers stores; a
    // workgroup-scope release fence compiles to lgkmcnt(0) only, which is not that wait), then ONE wavefront pays the system-scope
    // release -- an L2 write-back of a few microseconds whoever issues it: with every wavefront of the workgroup issuing its own the
    // publishing kernel ran 2.5-4.5 us longer (round 4, rocprofv3 time line of the Mode-23 update).
    fl_wait_own_stores();
    __syncthreads();
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __threadfence_system();
    if (threadIdx.x == 0) {
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // one-shot: a later kernel over this block that no begin preceded (fl_lio_finish18 behind a frame driver) must not publish
        // again -- the host may be filling the mirror for the next frame by then (the copy above already carries the live value)
        D->pub_flag = nullptr;
    }
}

// ---- instrumentation: compiled ONLY into the -DFL_INSTRUMENT build (libfastlivo_hip_debug.so, include/fastlivo_hip_debug.h).
// The release library carries no stamp, no debug global, no run-time branch on FL_ITER_STAMP: every site is FL_INSTR(...).
// The finer-grained stamp sets (-DFL_GATHER_STAMPS, -DFL_IK_STAMPS, -DFL_AUDIT_STAMPS) and the A/B switches (-DFL_AB_*) of tools/
// are further options of that build.
#ifdef FL_INSTRUMENT
#define FL_INSTR(...) __VA_ARGS__
__device__ long long g_fl_stamps[64];
__device__ long long g_fl_wall[2048];   // per-workgroup start/end wall clock (100 MHz)
__device__ unsigned g_fl_fault_epoch;   // fault injection (fl_debug_drop_record): producer workgroup 0 drops its record of this epoch (0: never, epochs start at 1)
#else
#define FL_INSTR(...)
#if defined(FL_GATHER_STAMPS) || defined(FL_IK_STAMPS) || defined(FL_AUDIT_STAMPS) || defined(FL_AB_NO_EXACT) || defined(FL_AB_NO_AUDITOR)
#error "stamp sets and A/B switches are options of the -DFL_INSTRUMENT build"
#endif
#endif

// ---- wavefront reduction --------------------------------------------------------------------
// Transposing butterfly: every lane enters with V partial sums; on exit lane L holds in v[0] the
// wave-wide total of value (L >> 1) (both lanes of a pair hold the same total). V must be 32.
// 32 exchanges instead of the 32 x 6 of an independent butterfly per value.
// 64-bit helpers over the 32-bit cross-lane instructions.
__device__ __forceinline__ unsigned f64_lo(double x) { return (unsigned)__double2loint(x); }
__device__ __forceinline__ unsigned f64_hi(double x) { return (unsigned)__double2hiint(x); }
__device__ __forceinline__ double f64_make(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b (gfx950). After the swap a+b holds, in the
// lower half-wave, a's total over the lane pair (l, l+32) and, in the upper half-wave, b's total.
__device__ __forceinline__ void swap32_f64(double &a, double &b)
{
    auto lo = __builtin_amdgcn_permlane32_swap(f64_lo(a), f64_lo(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(f64_hi(a), f64_hi(b), false, false);
    a = f64_make(lo[0], hi[0]);
    b = f64_make(lo[1], hi[1]);
}
// v_permlane16_swap: odd 16-lane rows of a <-> even rows of b (gfx950).
__device__ __forceinline__ void swap16_f64(double &a, double &b)
{
    auto lo = __builtin_amdgcn_permlane16_swap(f64_lo(a), f64_lo(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(f64_hi(a), f64_hi(b), false, false);
    a = f64_make(lo[0], hi[0]);
    b = f64_make(lo[1], hi[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, f64_lo(x), CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, f64_hi(x), CTRL, 0xf, 0xf, false);
    return f64_make(lo, hi);
}
#define FL_DPP_ROW_ROR8 0x128       // lane ^ 8 inside a 16-lane row
#define FL_DPP_QUAD_XOR2 0x4E       // quad_perm [2,3,0,1]
#define FL_DPP_QUAD_XOR1 0xB1       // quad_perm [1,0,3,2]

// Transposing butterfly: every lane enters with 32 partial sums; on exit lane L holds in v[0] the
// wave-wide total of value (L >> 1) (both lanes of a pair hold the same total).
// Exchanges: 16 + 8 lane-swaps (no selects), then 4+2+1+1 DPP/bpermute steps, instead of 32 x 6.
__device__ __forceinline__ void wave_transpose_reduce32(double (&v)[32], int lane)
{
#pragma unroll
    for (int i = 0; i < 16; i++) { swap32_f64(v[i], v[i + 16]); v[i] = v[i] + v[i + 16]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { swap16_f64(v[i], v[i + 8]); v[i] = v[i] + v[i + 8]; }
    {
        const bool upper = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double send = upper ? v[i] : v[i + 4];
            const double keep = upper ? v[i + 4] : v[i];
            v[i] = keep + dpp_f64<FL_DPP_ROW_ROR8>(send);
        }
    }
    {
        const bool upper = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const double send = upper ? v[i] : v[i + 2];
            const double keep = upper ? v[i + 2] : v[i];
            v[i] = keep + __shfl_xor(send, 4, FL_WAVE);
        }
    }
    {
        const bool upper = (lane & 2) != 0;
        const double send = upper ? v[0] : v[1];
        const double keep = upper ? v[1] : v[0];
        v[0] = keep + dpp_f64<FL_DPP_QUAD_XOR2>(send);
    }
    v[0] = v[0] + dpp_f64<FL_DPP_QUAD_XOR1>(v[0]);
}

// Wave totals of 6 values (w[6], w[7] must be 0): transposing steps for masks 32/16/8, plain butterfly
// for 4/2/1, then the 6 totals are read back with v_readlane (they end up wave-uniform, in SGPRs).
__device__ __forceinline__ void wave_sum6(double (&w)[8], int lane, double (&T)[6])
{
#pragma unroll
    for (int i = 0; i < 4; i++) { swap32_f64(w[i], w[i + 4]); w[i] = w[i] + w[i + 4]; }
#pragma unroll
    for (int i = 0; i < 2; i++) { swap16_f64(w[i], w[i + 2]); w[i] = w[i] + w[i + 2]; }
    {
        const bool upper = (lane & 8) != 0;
        const double send = upper ? w[0] : w[1];
        const double keep = upper ? w[1] : w[0];
        w[0] = keep + dpp_f64<FL_DPP_ROW_ROR8>(send);
    }
    w[0] = w[0] + __shfl_xor(w[0], 4, FL_WAVE);
    w[0] = w[0] + dpp_f64<FL_DPP_QUAD_XOR2>(w[0]);
    w[0] = w[0] + dpp_f64<FL_DPP_QUAD_XOR1>(w[0]);
    // lane L now holds the total of value ((L>>5)&1)*4 + ((L>>4)&1)*2 + ((L>>3)&1)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int src = ((k >> 2) & 1) * 32 + ((k >> 1) & 1) * 16 + (k & 1) * 8;
        const unsigned lo = __builtin_amdgcn_readlane(f64_lo(w[0]), src);
        const unsigned hi = __builtin_amdgcn_readlane(f64_hi(w[0]), src);
        T[k] = f64_make(lo, hi);
    }
}

// Same for the two 32-lane halves of a wave independently (one patch per half-wave): totals of the 6
// values over the lanes of the caller's half, returned in every lane of that half.
__device__ __forceinline__ void half_sum6(double (&w)[8], int lane, double (&T)[6])
{
#pragma unroll
    for (int i = 0; i < 4; i++) { swap16_f64(w[i], w[i + 4]); w[i] = w[i] + w[i + 4]; }
    {
        const bool upper = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const double send = upper ? w[i] : w[i + 2];
            const double keep = upper ? w[i + 2] : w[i];
            w[i] = keep + dpp_f64<FL_DPP_ROW_ROR8>(send);
        }
    }
    {
        const bool upper = (lane & 4) != 0;
        const double send = upper ? w[0] : w[1];
        const double keep = upper ? w[1] : w[0];
        w[0] = keep + __shfl_xor(send, 4, FL_WAVE);
    }
    w[0] = w[0] + dpp_f64<FL_DPP_QUAD_XOR2>(w[0]);
    w[0] = w[0] + dpp_f64<FL_DPP_QUAD_XOR1>(w[0]);
    // lane L holds the total (over its half) of value ((L>>4)&1)*4 + ((L>>3)&1)*2 + ((L>>2)&1)
    const int base = lane & 32;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int src = base + ((k >> 2) & 1) * 16 + ((k >> 1) & 1) * 8 + (k & 1) * 4;
        T[k] = __shfl(w[0], src, FL_WAVE);
    }
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, FL_WAVE);
    return x;
}

// Write-through (sc1) 8-byte store / L1-bypassing load: the cross-workgroup hand-off form of
// cdna_hip_programming.md Guideline 16 (R1) -- no release/acquire fence needed.
__device__ __forceinline__ void store_wt(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_wt(const double *p)
{
    unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}

typedef unsigned int fl_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int fl_u2 __attribute__((ext_vector_type(2)));
// the same vectors at DWORD alignment: the VIO tap rows are fetched with 8- / 12- / 16-byte loads from addresses rounded down to 4
// bytes (global_load_dwordxN needs no more); loading through the naturally aligned types would be undefined behaviour
typedef unsigned int fl_u2_dw __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int fl_u4_dw __attribute__((ext_vector_type(4), aligned(4)));

// A block of BYTES (a multiple of 8) fetched from page-locked host memory by NT threads: every load of the block is in flight before
// the first store. (Written as `for (w = tid; w < words; w += NT) dst[w] = load(src + w)` the compiler issued load, wait, store per
// round -- 6 trips over the host link, one after the other, for the 5 KB state block: 10 us where one trip takes 2. Round 6.)
template <int NT, int BYTES>
struct FlPull {
    static constexpr int WORDS = BYTES / 8, PER = (WORDS + NT - 1) / NT;
    static_assert(BYTES % 8 == 0, "word copy");
    unsigned long long v[PER];
    __device__ __forceinline__ void load(const void *src_)
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(src_);
#pragma unroll
        for (int k = 0; k < PER; k++) { const int w = (int)threadIdx.x + NT * k; v[k] = w < WORDS ? __builtin_nontemporal_load(src + w) : 0ull; }
    }
    __device__ __forceinline__ void store(void *dst_) const
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(dst_);
#pragma unroll
        for (int k = 0; k < PER; k++) { const int w = (int)threadIdx.x + NT * k; if (w < WORDS) dst[w] = v[k]; }
    }
};

// Phase timestamps for tools/kstamps.py (FL_INSTRUMENT build only): slot i of workgroup 0 and of the last workgroup, under the
// FL_ITER_STAMP flag. The release build ignores the flag.
#define FL_ITER_STAMP 4
#define FL_PUBLIC_ITER_FLAGS 7           /* FL_ITER_FORCE | FL_ITER_KEEP_NORMVEC | FL_ITER_STAMP: what a caller may pass */
#ifdef FL_INSTRUMENT
__device__ __forceinline__ void fl_stamp(int flags, int slot)
{
    if ((flags & FL_ITER_STAMP) && threadIdx.x == 0) g_fl_stamps[slot] = (long long)wall_clock64();
}
#endif
