// handoff.h -- in-launch reduction of per-workgroup records into ONE dedicated solver workgroup.
//
// Every pass kernel is launched with (producers + 1) workgroups. Producers reduce their points /
// patches to one NV-double record and publish it; the LAST workgroup of the grid owns no points: it
// prefetches what the gain solve needs while the producers work, then gathers the records, sums them
// in block-index order (results never depend on arrival order) and runs the solve. No atomics, no
// fences, no ticket:
//
//   publish : a record is NV doubles, 8 bytes each, stored write-through (sc1), fire-and-forget (the
//             stores drain while the workgroup exits). Every 8-byte word validates itself: its 6
//             lowest mantissa bits carry tag = 1 + epoch % 63 (never 0, the value of never-written
//             memory). An aligned 8-byte store is single-copy atomic, so a word whose tag matches
//             belongs to this launch (cdna_hip_programming.md Guideline 16, recipe R2 "the data IS the flag").
//             Cost: the partial sums keep 46 of 52 mantissa bits (relative 1.4e-14, two orders below
//             the 1e-12 the parity tests allow for fp64 re-ordering; the inputs are float32).
//             Stale words are always from the previous launch (every launch rewrites every record
//             of its grid; the host zeroes the buffer whenever the grid size changes, and zeroes what
//             the previous launch did not cover before a larger launch of another kernel reads it:
//             records_for in fastlivo_hip.hip), i.e. tag-1 or 0.
//   gather  : the solver workgroup re-reads the records with 16-byte sc1 buffer loads (L1-bypass, a
//             whole sweep in flight together). A 16-lane row reads one 32-value group of a record, so
//             one load instruction of a wave covers four records; each WAVE polls its own records
//             with a wave-uniform "still missing" mask built from ballots: later sweeps skip complete
//             records with scalar branches. Spins are bounded: a timeout sets status bit 8 instead of
//             hanging the GPU.
//   epoch   : a device word, read by every workgroup at kernel start and incremented by the solver at
//             the end of the pass (graph-replay safe: not a kernel argument).
//
// Why 8-byte self-tagged words: tools/gather_bench.hip shows one workgroup reads freshly written
// data at 0.8 us for <= 32 KB and ~0.4 us per further 32 KB (in-flight limit of one CU, ~50-80 GB/s),
// so a sweep over 196 records costs 2.0 us at 512 B/record ({epoch,lo,hi,epoch} per double, the first
// version of this scheme) and 1.25 us at 256 B/record. Measured per LIO pass (50k points): arrival
// ticket + atomics 8.9 us; 16-byte tagged granules 8.5 us; XOR-checksummed raw words 11.4 us (the
// producers pay an LDS pass); this form: see DESIGN.md section 4.1.
#pragma once

#include "fl_device.h"

#define FL_GATHER_SPIN_LIMIT (1 << 15)
#define FL_NUM_TIMEOUT 8
#define FL_TAG_MASK 0x3Fu

__device__ __forceinline__ unsigned fl_epoch_tag(unsigned epoch) { return 1u + epoch % 63u; }

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
__device__ __forceinline__ void publish_record(double mine, unsigned epoch, void *records)
{
    const int tid = threadIdx.x;
    FL_INSTR(if (blockIdx.x == 0 && epoch == g_fl_fault_epoch) return;)      // fault injection (debug build): this record never shows up
    if (tid < NV) {
        // The tag takes the low 6 bits of the mantissa: the partial travels with 46 of its 52 bits. Rounded to nearest (+ half of the
        // dropped field, carries run into the exponent like any rounding), not truncated -- truncation biased every partial toward
        // zero by up to 2^-46 and the bias of ~1000 like-signed partials (the diagonal of H^T H at millions of points) adds up
        // instead of averaging out. inf and NaN (exponent all ones) are NOT rounded: the carry of an all-ones NaN mantissa would run
        // through the exponent into the sign and come out finite; untouched they still fail the finiteness check of the solve (a NaN
        // whose only mantissa bits sit in the dropped field becomes inf -- still not finite).
        unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        const bool nonfinite = (bits & 0x7FF0000000000000ull) == 0x7FF0000000000000ull;
        bits += nonfinite ? 0ull : (unsigned long long)((FL_TAG_MASK + 1u) >> 1);
        bits = (bits & ~(unsigned long long)FL_TAG_MASK) | fl_epoch_tag(epoch);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(records) + (size_t)blockIdx.x * NV + tid, bits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);   // sc1 write-through
    }
}

__device__ __forceinline__ double fl_untag(unsigned lo, unsigned hi) { return f64_make(lo & ~FL_TAG_MASK, hi); }

// Solver workgroup: gather nprod records and leave the totals in out_lds[NV]. Returns 0, or
// FL_NUM_TIMEOUT if some record never showed up (all threads agree).
// All NV / 32 groups of a record are polled in the SAME sweep (slots (group, batch), 16 loads per lane in flight at most): one
// sweep covers 16 * NT / 16 / (NV / 32) records -- 256 of the 32-double records, 128 of Mode-23's 64-double ones.
template <int NT, int NV>
__device__ __forceinline__ int gather_records(const void *records, int nprod, unsigned epoch, double *lds /* >= 2*NT */,
                                              double *out_lds /* NV */)
{
    constexpr int ROWS = NT / 16;         // records covered by one load instruction of the workgroup
    constexpr int G = NV / 32;            // 32-value groups of a record
    constexpr int BATCH = 16 / G;         // batches of ROWS records per sweep
    static_assert(G == 1 || G == 2, "16 load slots per lane");
    const int tid = threadIdx.x;
    const int kp = tid & 15, row = tid >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tag = fl_epoch_tag(epoch);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)records, 0, nprod * NV * 8, 0x00020000);
    int timeout = 0;
    // (tried in round 2 and rejected: waiting for the records with a light one-word-per-record poll and only then doing ONE full
    // sweep -- 7.5 vs 6.9 us per LIO pass, 10.1 vs 9.0 per VIO pass: the speculative sweeps below pick the early records up while the
    // late ones are still in flight, and the last sweep re-reads only the missing batches.
    // Tried in round 3 and removed: every wavefront first polling ONE load instruction's worth -- its four records of batch 0 -- and
    // starting the full sweep when those have arrived. Per-wave stamps: polling done 1.84-2.04 us after workgroup 0's publication
    // against 1.72-1.84 us with the speculative sweeps: the sentinel's detection round trip costs what the wasted first sweep cost.
    // Note for anyone who retries it: a loop around ONE buffer load with a loop-invariant address needs a compiler barrier, the load
    // is hoisted otherwise.)
    double s0[G], s1[G];
#pragma unroll
    for (int g = 0; g < G; g++) { s0[g] = 0.0; s1[g] = 0.0; }
    for (int base = 0; base < nprod; base += ROWS * BATCH) {   // trip count uniform over the workgroup
        const int b0 = base + row;
        fl_u4 t[G][BATCH];
        unsigned need = 0u;                                     // wave-uniform; bit g * BATCH + j
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                need |= ((base + 4 * wave_u + j * ROWS) < nprod) ? (1u << (g * BATCH + j)) : 0u;
                t[g][j].x = 0u; t[g][j].y = 0u; t[g][j].z = 0u; t[g][j].w = 0u;
            }
        // The byte offsets of the sweep's loads, one register each and opaque to the compiler: left to itself under register pressure
        // (vio_multipass_kernel<1, 1>, round 6) it re-derived each offset from the first one INTO the destination registers of that
        // load, which costs an s_waitcnt vmcnt(0) in front of every load -- the sixteen polls of a sweep went out one round trip after
        // the other (1.5 us of every pass).
        int off[G][BATCH];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                off[g][j] = ((b0 + j * ROWS) * NV + g * 32 + kp * 2) * 8;
                asm volatile("" : "+v"(off[g][j]));
            }
        for (int spin = 0; need != 0u; spin++) {
#ifdef FL_GATHER_STAMPS
            if (tid == 0 && spin < 8) { g_fl_stamps[56 + spin] = (long long)wall_clock64(); g_fl_wall[2040 + spin] = (long long)__builtin_popcount(need); }
#endif
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int j = 0; j < BATCH; j++) {
                    if (need & (1u << (g * BATCH + j))) {
                        t[g][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[g][j], 0, 16 /* sc1 */);
                    }
                }
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int j = 0; j < BATCH; j++) {
                    if (need & (1u << (g * BATCH + j))) {
                        const int b = b0 + j * ROWS;
                        const bool ok = (b >= nprod) || (((t[g][j].x & FL_TAG_MASK) == tag) && ((t[g][j].z & FL_TAG_MASK) == tag));
                        if (__ballot(ok) == ~0ull) need &= ~(1u << (g * BATCH + j));
                    }
                }
#ifdef FL_GATHER_STAMPS
            if (tid == 0 && spin < 8) { g_fl_stamps[48 + spin] = (long long)wall_clock64(); g_fl_stamps[47] = spin + 1; g_fl_stamps[46] = (long long)__builtin_popcount(need); }
#endif
            if (need != 0u) {
                if (spin >= FL_GATHER_SPIN_LIMIT) { timeout = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
#ifdef FL_GATHER_STAMPS
        if ((tid & 63) == 0) g_fl_wall[2024 + wave_u] = (long long)wall_clock64();
#endif
        // this lane's records in ascending order, straight-line: batches beyond the grid were never loaded (t = 0) and a row beyond
        // the grid inside a live batch read zeros (buffer bounds) -- both untag to +0.0. (This tail runs on ONE wavefront per SIMD:
        // a compare + branch per record cost 0.7 us of every pass, a scalar branch per batch still 0.1 us.)
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                s0[g] += fl_untag(t[g][j].x, t[g][j].y);
                s1[g] += fl_untag(t[g][j].z, t[g][j].w);
            }
        if (timeout) break;
    }
    // the four 16-lane rows of a wavefront hold four different records' share of value pair kp: rows first (two lane swaps, no
    // LDS), then the NT/64 wavefronts through LDS. Fixed order: (row 0 + row 1) + (row 2 + row 3), then wavefronts ascending.
#pragma unroll
    for (int g = 0; g < G; g++) {
        double a = s0[g], b = s1[g];
        swap16_f64(a, b);
        double c = a + b;                                           // even rows: value 2 kp, odd rows: value 2 kp + 1
        double c2 = c;
        swap32_f64(c, c2);
        c = c + c2;
        if ((tid & 63) < 32) lds[(g * (NT / 64) + wave_u) * 32 + 2 * kp + ((tid & 63) >> 4)] = c;
    }
    // ONE barrier: the time-out flags ride through it (__syncthreads_or is a workgroup reduction of its own), and behind it EVERY
    // wavefront adds the NT/64 partials up itself (same order, same bits) and writes all of out_lds -- identical values from every
    // wavefront, so whoever reads out_lds next reads what its own wavefront wrote and no second barrier is needed.
    int *lds_to = reinterpret_cast<int *>(lds + G * (NT / 64) * 32);
    if ((tid & 63) == 32) lds_to[wave_u] = timeout;
    __syncthreads();
    timeout = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) timeout |= lds_to[w];
    if ((tid & 63) < 32) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            double t2 = lds[(g * (NT / 64)) * 32 + (tid & 63)];
#pragma unroll
            for (int w = 1; w < NT / 64; w++) t2 += lds[(g * (NT / 64) + w) * 32 + (tid & 63)];
            out_lds[g * 32 + (tid & 63)] = t2;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef FL_GATHER_STAMPS
    if (tid == 0) g_fl_wall[2036] = (long long)wall_clock64();
#endif
    return timeout ? FL_NUM_TIMEOUT : 0;
}


// ---- state broadcast of the multi-pass kernels: solver workgroup -> every producer workgroup, inside one launch.
// The new pose (12 doubles) + a control word travel as 25 self-validating 8-byte words: high 32 bits = payload (one
// half of a double, or the control bits), low 32 bits = the pass epoch the word belongs to. An aligned 8-byte store
// is single-copy atomic and written through (sc1); a reader that sees the expected epoch in a word has that word's
// payload -- no fence, no flag ordering (Guideline 16 "the data IS the flag"), and the doubles arrive bit-exact.
#define FL_BCAST_WORDS 25
// The words exist in FL_BCAST_REPL copies, FL_BCAST_STRIDE words apart (different memory channels): ~200 producer workgroups
// polling ONE 200-byte region made that region's channel the hot spot of the whole hand-off; workgroup b polls copy b % REPL.
#ifndef FL_BCAST_REPL
#define FL_BCAST_REPL 4
#endif
#define FL_BCAST_STRIDE 544                 /* 4352 bytes */
#ifndef FL_BCAST_PHASES
#define FL_BCAST_PHASES 2
#endif
__device__ __forceinline__ void bcast_publish(unsigned long long *words, const double *x12 /* LDS */, int ctrl, unsigned epoch)
{
    const int tid = threadIdx.x;
    if (tid < FL_BCAST_WORDS) {
        unsigned payload;
        if (tid < 24) {
            const double v = x12[tid >> 1];
            payload = (tid & 1) ? f64_hi(v) : f64_lo(v);
        } else {
            payload = (unsigned)ctrl;
        }
#pragma unroll
        for (int r = 0; r < FL_BCAST_REPL; r++)
            __hip_atomic_store(words + (size_t)r * FL_BCAST_STRIDE + tid, ((unsigned long long)payload << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Wave 0 of a producer workgroup polls (measured: letting all four waves poll with staggered phases costs more in extra
// traffic than it gains in detection latency: 7.2-7.8 vs 7.1 us per pass). After the caller's __syncthreads
// out12[0..11] = pose, *ctrl_out = control bits (bit 2 set on a timeout: bounded spin).
template <int NWORDS = FL_BCAST_WORDS>      // NWORDS - 1 payload words (halves of doubles) + the control word, <= 64
__device__ __forceinline__ void bcast_wait(const unsigned long long *words, unsigned epoch, double *out12 /* LDS, (NWORDS-1)/2 */, int *ctrl_out /* LDS */,
                                           int spin_limit = FL_GATHER_SPIN_LIMIT)
{
    static_assert(NWORDS <= 64 && (NWORDS & 1) == 1, "one wavefront polls the words");
    const int tid = threadIdx.x;
    if (tid >= 64) return;
    words += (size_t)(blockIdx.x % FL_BCAST_REPL) * FL_BCAST_STRIDE;
    const bool mine = tid < NWORDS;
    const unsigned long long *src = words + (mine ? tid : 0);
    unsigned long long w = 0ull;
    bool timeout = false;
#ifdef FL_BCAST_PRESLEEP
    __builtin_amdgcn_s_sleep(FL_BCAST_PRESLEEP);
#endif
#if FL_BCAST_PHASES == 2
    // two polls in flight, half a round trip apart: a poll answers one memory round trip (~0.6 us) after it was issued, so a
    // single poll loop notices the words up to one round trip late; two interleaved loops halve that
    unsigned long long wa = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(10);
    for (int spin = 0; ; spin++) {
        unsigned long long wb = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(!mine || (unsigned)wa == epoch) == ~0ull) { w = wa; break; }
        wa = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(!mine || (unsigned)wb == epoch) == ~0ull) { w = wb; break; }
        if (spin > spin_limit) { timeout = true; break; }
        if (spin > 32) __builtin_amdgcn_s_sleep(8);      // a long wait (a peer rank, a delayed solver): back off instead of hammering the words
    }
#else
    bool ok = !mine;
    for (int spin = 0; ; spin++) {
        if (!ok) {
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ((unsigned)w == epoch);
        }
        if (__ballot(ok) == ~0ull) break;
        if (spin > spin_limit) { timeout = true; break; }
    }
#endif
    const unsigned payload = (unsigned)(w >> 32);
    const unsigned other = (unsigned)__shfl_xor((int)payload, 1, 64);
    if (tid < NWORDS - 1 && (tid & 1) == 0) out12[tid >> 1] = f64_make(payload, other);
    if (tid == NWORDS - 1) *ctrl_out = timeout ? 5 : (int)payload;
}


// ---- exchange between the ranks of the sharded form ("one-shot P2P all-gather: each rank writes its record into the
// peers' mapped buffers + flag; fixed-order local sum"), INSIDE the pass kernel: the solver workgroup of every rank publishes the
// 32 sums of its shard to every peer and adds up what the peers sent, in rank order -- so every rank holds bitwise the same
// totals, solves redundantly and needs no broadcast, and a frame's passes stay one launch per rank (multi-pass kernels) instead of
// accumulate -> ncclAllReduce -> solve per pass. Same "the data IS the flag" words as the pose broadcast: 2 words per double,
// high half = payload, low half = exchange epoch, system-scope stores into fine-grained peer memory (over xGMI between GPUs), polled
// with system-scope loads in the receiver's own memory. Layout of a rank's buffer: [epoch parity][sender rank][64 words]; two
// parities suffice because a rank can be at most one exchange ahead of a peer (it needs the peer's words of exchange k to leave
// exchange k). Ranks start their kernels at different times (separate processes): the wait is long (seconds) but bounded.
#define FL_MAX_PEERS 8
#define FL_XCHG_WORDS 64
#define FL_XCHG_SPIN_LIMIT (1 << 22)
// behind the record words of a rank's buffer: four 8-byte mail slots of the sharded VIO accept replay (solve18.h): float payload in
// the high half, exchange epoch of the pass in the low half.  0 / 2: carry from rank-1 (current / last-accepted chain), 1 / 3: the
// final sum from the last rank.
#define FL_XCHG_REPLAY_SLOTS 4
#define FL_XCHG_TOTAL_WORDS(world) (2 * (size_t)(world) * FL_XCHG_WORDS + FL_XCHG_REPLAY_SLOTS)
struct FlPeerView {
    unsigned long long *own;                   // this rank's buffer
    unsigned long long *const *peer;           // D->xchg_peer (read with constant indices: a register array indexed at run time would live in scratch)
    int rank, world;
};
template <typename DEV>
__device__ __forceinline__ FlPeerView fl_peer_view(const DEV *D)
{
    FlPeerView P;
    P.rank = D->xchg_rank; P.world = D->xchg_world;
    P.peer = D->xchg_peer;
    P.own = (P.world > 1) ? D->xchg_peer[P.rank] : nullptr;
    return P;
}
// The same with the peers' addresses staged in LDS once per launch (multi-pass kernels: the state block is rewritten by the solver
// every pass, so the compiler would re-load the eight pointers from memory in every exchange). All threads call it; ends with a barrier.
template <typename DEV>
__device__ __forceinline__ FlPeerView fl_peer_view_lds(const DEV *D, unsigned long long **lds8)
{
    if (threadIdx.x < FL_MAX_PEERS) lds8[threadIdx.x] = D->xchg_peer[threadIdx.x];
    FlPeerView P;
    P.rank = D->xchg_rank; P.world = D->xchg_world;
    P.own = (P.world > 1) ? D->xchg_peer[P.rank] : nullptr;
    __syncthreads();
    P.peer = lds8;
    return P;
}
// All threads of the (>= 256-thread) solver workgroup call it; sums = LDS[32], replaced by the total over the ranks;
// tmp = LDS[FL_MAX_PEERS * 32]. Returns 0 or FL_NUM_TIMEOUT (all threads agree).
__device__ __forceinline__ int peer_allreduce32(const FlPeerView &P, unsigned xe, double *sums, double *tmp)
{
    const int tid = threadIdx.x;
    const size_t half = (size_t)(xe & 1u) * (size_t)P.world * FL_XCHG_WORDS;
    if (tid < FL_XCHG_WORDS) {
        const double v = sums[tid >> 1];
        const unsigned payload = (tid & 1) ? f64_hi(v) : f64_lo(v);
        const unsigned long long w = ((unsigned long long)payload << 32) | (unsigned long long)xe;
#pragma unroll
        for (int r = 0; r < FL_MAX_PEERS; r++)
            if (r < P.world && r != P.rank)
                __hip_atomic_store(P.peer[r] + half + (size_t)P.rank * FL_XCHG_WORDS + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int timeout = 0;
    const int per_round = (int)blockDim.x / FL_XCHG_WORDS;           // senders polled at a time: one wave each
    for (int s0 = 0; s0 < P.world; s0 += per_round) {
        const int s = s0 + (tid >> 6), k = tid & 63;
        const bool active = s < P.world && s != P.rank;
        const unsigned long long *src = P.own + half + (size_t)(active ? s : 0) * FL_XCHG_WORDS + k;
        unsigned long long w = 0ull;
        bool ok = !active;
        for (int spin = 0; ; spin++) {
            if (!ok) {
                w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = ((unsigned)w == xe);
            }
            if (__ballot(ok) == ~0ull) break;
            if (spin > FL_XCHG_SPIN_LIMIT) { timeout = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        const unsigned payload = (unsigned)(w >> 32);
        const unsigned other = (unsigned)__shfl_xor((int)payload, 1, 64);
        if (active && (k & 1) == 0) tmp[s * 32 + (k >> 1)] = f64_make(payload, other);
    }
    __syncthreads();
    double tot = 0.0;
    if (tid < 32) {
        for (int r = 0; r < P.world; r++) {
            const double v = (r == P.rank) ? sums[tid] : tmp[r * 32 + tid];
            tot = (r == 0) ? v : tot + v;
        }
    }
    __syncthreads();
    if (tid < 32) sums[tid] = tot;
    return __syncthreads_or(timeout) ? FL_NUM_TIMEOUT : 0;
}

// The 64-double record of Mode-23 as two exchanges of 32 (epochs xe, xe + 1: the buffer keeps two parities, and an exchange is left
// only when every peer's words of it have arrived, so consecutive exchanges cannot overtake each other).
__device__ __forceinline__ int peer_allreduce64(const FlPeerView &P, unsigned xe, double *sums64, double *tmp)
{
    int st = 0;
#pragma unroll
    for (int g = 0; g < 2; g++) st |= peer_allreduce32(P, xe + (unsigned)g, sums64 + 32 * g, tmp);
    return st;
}
