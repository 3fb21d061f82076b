// handoff.h -- in-launch reduction of per-workgroup records into ONE dedicated solver workgroup.
//
// Every pass kernel is launched with (producers + 1) workgroups. Producers reduce their points /
// patches to one NV-double record and publish it; the LAST workgroup of the grid owns no points: it
// prefetches what the gain solve needs while the producers work, then gathers the records, sums them
// in block-index order (results never depend on arrival order) and runs the solve. No atomics, no
// fences, no ticket:
//
//   publish : each double travels as one 16-byte write-through (sc1) store {epoch, lo, hi, epoch}.
//             A 16-byte store can tear only into its two 8-byte halves and each half carries the
//             tag, so "both tags == epoch" proves both payload words belong to this launch
//             (data-tagged granules, cdna_hip_programming.md Guideline 16 recipe R2). The producer
//             never waits: the stores drain while the workgroup exits.
//   gather  : the solver workgroup re-reads the records with 16-byte sc1 buffer loads (L1-bypass, a
//             whole sweep in flight together). A half-wave reads one whole record, so one load
//             instruction of a wave covers two records; each WAVE polls its own records with a
//             wave-uniform "still missing" mask (built from ballots): later sweeps skip complete
//             pairs with scalar branches. One workgroup barrier at the end, none per sweep. Spins are
//             bounded: a timeout sets status bit 8 instead of hanging the GPU.
//   epoch   : a device word, read by every workgroup at kernel start and incremented by the solver at
//             the end of the pass (graph-replay safe: not a kernel argument). Never 0; the record
//             buffer is zeroed at fl_create.
//
// Measured alternatives on MI355X (tools/kwall.py, 50k points, 196 producers), per pass:
//   last-arriver ticket (write-through store, vmcnt(0) drain, returning atomic, re-read)  8.9 us
//     -- the atomics serialise at ~12 ns per arrival and every step is a full memory round trip;
//   this scheme                                                                         8.0-8.5 us
//   raw 8-byte words + one XOR checksum per record (half the bytes)                      11.4 us
//     -- the gather is latency-, not byte-bound, and the checksum costs the producers an LDS pass.
#pragma once

#include "fl_device.h"

#define FL_GATHER_SPIN_LIMIT (1 << 15)
#define FL_NUM_TIMEOUT 8

// Block-level reduction: on return thread tid < NV holds the workgroup total of value tid.
template <int NT, int NV>
__device__ __forceinline__ double block_reduce_record(double (&v)[NV], double *lds /* >= (NT/64)*NV */)
{
    static_assert(NV % 32 == 0 && NT % 64 == 0 && NT >= NV, "shape");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int g = 0; g < NV / 32; g++) {
        double w[32];
#pragma unroll
        for (int i = 0; i < 32; i++) w[i] = v[g * 32 + i];
        wave_transpose_reduce32(w, lane);
        if ((lane & 1) == 0) lds[wave * NV + g * 32 + (lane >> 1)] = w[0];
    }
    __syncthreads();
    double s = 0.0;
    if (tid < NV) {
        s = lds[tid];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) s += lds[w * NV + tid];
    }
    return s;
}

// Fire-and-forget publication of this workgroup's record (thread tid < NV holds value tid).
template <int NV>
__device__ __forceinline__ void publish_record(double mine, unsigned epoch, void *records, int nrecords)
{
    const int tid = threadIdx.x;
    if (tid < NV) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(records, 0, nrecords * NV * 16, 0x00020000);
        fl_u4 g;
        g.x = epoch; g.y = f64_lo(mine); g.z = f64_hi(mine); g.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, (blockIdx.x * NV + tid) * 16, 0, 16 /* sc1: write-through */);
    }
}

// Solver workgroup: gather nprod records and leave the totals in out_lds[NV]. Returns 0, or
// FL_NUM_TIMEOUT if some record never showed up (all threads agree).
template <int NT, int NV>
__device__ __forceinline__ int gather_records(const void *records, int nprod, unsigned epoch, double *lds /* >= NT */,
                                              double *out_lds /* NV */)
{
    constexpr int GROUPS = NT / 32;       // records covered by one load instruction of the workgroup
    constexpr int BATCH = 32;             // up to GROUPS*32 producers per sweep (256 @ NT=256)
    const int tid = threadIdx.x;
    const int kk = tid & 31, grp = tid >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)records, 0, nprod * NV * 16, 0x00020000);
    int timeout = 0;
#pragma unroll
    for (int g = 0; g < NV / 32; g++) {
        double s = 0.0;
        for (int base = 0; base < nprod; base += GROUPS * BATCH) {   // trip count uniform over the workgroup
            const int b0 = base + grp;
            fl_u4 t[BATCH];
            unsigned need = 0u;                                       // wave-uniform
#pragma unroll
            for (int j = 0; j < BATCH; j++) need |= ((base + 2 * wave_u + j * GROUPS) < nprod) ? (1u << j) : 0u;
#pragma unroll
            for (int j = 0; j < BATCH; j++) { t[j].x = 0u; t[j].y = 0u; t[j].z = 0u; t[j].w = 0u; }
            for (int spin = 0; need != 0u; spin++) {
#pragma unroll
                for (int j = 0; j < BATCH; j++) {
                    if (need & (1u << j)) {
                        const int b = b0 + j * GROUPS;
                        t[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (b * NV + g * 32 + kk) * 16, 0, 16 /* sc1 */);
                    }
                }
#pragma unroll
                for (int j = 0; j < BATCH; j++) {
                    if (need & (1u << j)) {
                        const int b = b0 + j * GROUPS;
                        const bool ok = (b >= nprod) || (t[j].x == epoch && t[j].w == epoch);
                        if (__ballot(ok) == ~0ull) need &= ~(1u << j);
                    }
                }
                if (need != 0u) {
                    if (spin >= FL_GATHER_SPIN_LIMIT) { timeout = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                const int b = b0 + j * GROUPS;
                s += (b < nprod) ? f64_make(t[j].y, t[j].z) : 0.0;
            }
        }
        lds[tid] = s;
        __syncthreads();
        if (tid < 32) {
            double t2 = lds[tid];
#pragma unroll
            for (int j = 1; j < GROUPS; j++) t2 += lds[j * 32 + tid];
            out_lds[g * 32 + tid] = t2;
        }
        __syncthreads();
    }
    return __syncthreads_or(timeout) ? FL_NUM_TIMEOUT : 0;
}
