// exact_chain.h -- the reference's float running sum `error += patch_error` (lidar_selection.cpp:849-857), bit for bit, without
// doing its m additions one after the other.
//
// A dependent v_add_f32 costs ~18 cycles on gfx950 (one wavefront on a SIMD: nothing to hide the pipeline behind), so one lane
// replaying 2 k additions takes ~20 us -- two whole VIO passes. The chain is sequential only in appearance:
//
//   while the running sum s stays inside one binade [2^E, 2^(E+1)) it is an integer multiple S of u = 2^(E-23), S < 2^24, and
//   fl(s + e) = (S + q) u with q = e/u rounded to the nearest integer -- INDEPENDENT of S, except on an exact tie (frac(e/u) = 1/2),
//   where round-half-even makes the RESULT even: q = floor(e/u) + (parity(S) ^ parity(floor)), and the parity after a tie is 0
//   whatever it was before. So the integer increments of a block of elements are a function of one bit of state that composes
//   associatively (x -> x ^ c, or x -> 0): 256 elements per step (4 per lane) need, when the block holds a tie at all, two ballots
//   for the parities, and one integer prefix sum for the S values. The first element whose exact sum reaches 2^24 u leaves the
//   binade: the adder itself does that one addition (its rounding unit is 2u or more and does depend on S), and the next step
//   starts from the new binade.
//
// ~8 steps for 2 k patches plus one per binade the sum climbs through (the first 16 additions, which climb fastest, are done
// plainly). A step is ~70 fp32/integer instructions when nothing special happens in it (no tie, no crossing). Elements are >= 0 by
// construction (sums of squares); a negative/NaN element (`bad`, found while staging) or a non-finite running sum falls back to the
// plain chain.
//
// The per-lane phases are plain functions (FL_HD): tests/host_emul runs them lane by lane on the CPU against the plain loop
// (tests/test_exact_chain_cpu.py); fl_debug_chain runs the device driver against the plain loop on the GPU.
#pragma once
#include "fl_math.h"

#define FL_CHAIN_EPL 4                       /* elements per lane and step */
#define FL_CHAIN_STEP (64 * FL_CHAIN_EPL)    /* elements per step; the array is zero-padded by this much behind its end */
#define FL_CHAIN_LEAD 16                     /* leading elements added plainly */
#define FL_CHAIN_LIMIT (1 << 24)             /* S of the next binade, in units u */

struct FlChainLane {
    float e[FL_CHAIN_EPL];
    int q[FL_CHAIN_EPL];                     // integer increment (of a tie: set by fl_chain_ties)
    int tie;                                 // bit i: element i is an exact tie
    int Qc;                                  // sum of the increments, capped at 2^24 (a lane that large has left the binade anyway)
    int cidx, sprev;                         // first element (index within the step) whose sum reaches 2^24, S before it
    float ec;
};

FL_HD unsigned fl_chain_bits(float s) { union { float f; unsigned u; } b; b.f = s; return b.u; }
FL_HD bool fl_chain_plain_only(float s)      // inf / NaN / negative running sum
{
    const unsigned b = fl_chain_bits(s);
    return ((b >> 23) & 0xffu) == 255u || (b >> 31) != 0u;
}
// finite s >= 0 as S u with u = 2^(Eb - 150): Eb = max(biased exponent, 1) (subnormals share u = 2^-149 with the first normal binade)
FL_HD void fl_chain_split(float s, int *S, int *Eb)
{
    const unsigned b = fl_chain_bits(s);
    const unsigned eb = b >> 23;
    *S = (int)(eb ? ((b & 0x7fffffu) | 0x800000u) : b);
    *Eb = (int)(eb ? eb : 1u);
}
FL_HD float fl_chain_from_units(int S, int Eb) { return ldexpf((float)S, Eb - 150); }     // S <= 2^24: exact

FL_HD float fl_chain_fract(float x)          // x >= 0, exact
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(x);
#else
    return x - floorf(x);
#endif
}

// phase 1: this lane's elements as increments in units u = 2^(Eb - 150). Everything is exact in fp32: a power-of-two scaling, a
// fraction of a number below 2^24. Returns the tie mask.
FL_HD int fl_chain_phase1(FlChainLane &L, const float *scr, int first, int Eb)
{
    L.tie = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        const float e = scr[first + i];                  // (zero-padded behind the end)
        L.e[i] = e;
        float x = ldexpf(e, 150 - Eb);                   // e / u (inf when far outside: clamped; an underflow is far below 1/2)
        x = x < 16777216.0f ? x : 16777216.0f;
        // nearest integer, ties to even ON x -- right for every non-tie; a tie (fraction exactly 1/2) is marked and gets its floor
        // and its parity-dependent increment in fl_chain_ties
        L.q[i] = (int)rintf(x);
        L.tie |= ((fl_chain_fract(x) == 0.5f) ? 1 : 0) << i;
    }
    return L.tie;
}
FL_HD void fl_chain_sum(FlChainLane &L)
{
    int Q = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) Q += L.q[i];
    L.Qc = Q < FL_CHAIN_LIMIT ? Q : FL_CHAIN_LIMIT;
}

// ---- only when the step holds a tie
// the lane's parity map: out = isc ? xr : in ^ xr
FL_HD void fl_chain_parity_map(const FlChainLane &L, int *isc, int *xr)
{
    int c = 0, x = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        if ((L.tie >> i) & 1) { c = 1; x = 0; } else x ^= L.q[i] & 1;     // (a tie's q is not looked at here)
    }
    *isc = c; *xr = x;
}
// parity of S before this lane's first element, from the ballots of (isc, xr) over the lanes and the parity of the step's S
FL_HD int fl_chain_parity_in(unsigned long long Cm, unsigned long long Xm, int lane, int S)
{
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long cb = Cm & below;
    if (cb) {
        const int t = 63 - __builtin_clzll(cb);
        return (int)(__builtin_popcountll(Xm & below & ~((1ull << t) - 1ull)) & 1);       // lane t's bit is its constant
    }
    return (int)(((unsigned)S + (unsigned)__builtin_popcountll(Xm & below)) & 1u);
}
// ties decided: round half to even, i.e. the sum becomes even
FL_HD void fl_chain_ties(FlChainLane &L, int parity_in, int Eb)
{
    int p = parity_in;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        if ((L.tie >> i) & 1) {
            const int fl = (int)(ldexpf(L.e[i], 150 - Eb) - 0.5f);     // floor of n + 1/2 (exact; a tie is never clamped)
            L.q[i] = fl + ((p ^ fl) & 1);
            p = 0;
        } else p ^= L.q[i] & 1;
    }
}

// ---- only when the step leaves the binade: first element of the lane whose sum reaches 2^24; s_before = S before the lane's
// first element
FL_HD void fl_chain_crossing(FlChainLane &L, int lane, int s_before)
{
    int t = s_before;
    L.cidx = -1; L.sprev = 0; L.ec = 0.0f;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        const int tn = t + L.q[i];
        if (L.cidx < 0 && tn >= FL_CHAIN_LIMIT) { L.cidx = FL_CHAIN_EPL * lane + i; L.sprev = t; L.ec = L.e[i]; }
        t = tn;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The WORKGROUP form (fl_chain_f32_block below): all of a chunk's elements at once instead of 256 per step and a step per binade.
//
// Which binade the running sum is in when element k is added is not known without the chain -- but it can be GUESSED from a double
// prefix sum P (any order of additions: 256 threads scan it in a few hundred cycles), and a guess can be CHECKED. With the guessed
// binade of every element,
//   * an element whose guessed binade differs from its successor's (the sum is about to leave the binade), or which is an exact tie
//     under the guessed unit, is an EVENT: the float adder itself will do that addition (right whatever the state is);
//   * every other element has an integer increment q = rne(e/u) that does not depend on the state (see above), so a SEGMENT between
//     two events is one integer: the difference of a prefix sum over q.
// What is left of the chain is a walk over the events (2 k patch errors: ~12 binades + ~8 ties), ~17 scalar instructions each:
// add the segment's integer (checking the sum stays below 2^24: every addition of the segment then was inside the binade), let the
// adder add the event's element, take the result apart, and check that its binade is the one guessed for the elements that follow.
// Every check passed => every element was added under the unit the true running sum had at that point, i.e. the result is the
// plain chain's, bit for bit. A failed check (the double prefix and the float chain disagree about a binade next to a power of two:
// ~1 % of 2 k-element chains) falls back to fl_chain_f32_wave. Nothing is ever accepted unchecked.
#define FL_SPEC_LEAD 32                       /* leading elements added plainly by wavefront 0 (all events anyway: a binade each) */
#define FL_SPEC_EPT 8                         /* elements per thread (256 threads: a chunk of 2048) */
#define FL_SPEC_MAX_EVENTS 63                 /* (+ the tail segment in slot nev) */
#define FL_SPEC_CNT_SCALE 68719476736.0       /* 2^36: event count and increment sum packed into one double for one scan */

// binade (as fl_chain_split's Eb) of a float running sum whose value is about P: anything out of range fails the checks later
FL_HD int fl_spec_binade(double P)
{
    union { double d; unsigned long long u; } b; b.d = P;
    const int e = (int)((b.u >> 52) & 0x7ffull) - 896;     // 1023 - 127
    return e < 1 ? 1 : e;
}
// element e under the guessed binades Eb (when it is added) and Ebn (its successor's): 0 = plain element with increment *q ;
// 1 = event for the adder (the binade is about to change) ; 2 = exact tie inside the binade, *tfloor = floor(e/u) (its increment
// depends on the parity of the sum before it: the walk adds it as an integer). *q = 0 for events.
FL_HD int fl_spec_elem(float e, int Eb, int Ebn, int *q, int *tfloor)
{
    float x = ldexpf(e, 150 - Eb);
    x = x < 16777216.0f ? x : 16777216.0f;
    const bool tie = fl_chain_fract(x) == 0.5f;
    const int kind = (Ebn != Eb) ? 1 : (tie ? 2 : 0);
    *q = kind ? 0 : (int)rintf(x);
    *tfloor = (int)(x - 0.5f);                             // (exact for a tie: n + 1/2 - 1/2)
    return kind;
}
#define FL_SPEC_TAIL (-1)                     /* event slot codes besides a successor's binade (>= 1) */
#define FL_SPEC_TIE (-2)
// one event of the walk: segment increment Q (capped at 2^24), then the event itself -- code >= 1: its element (ebits: the float's
// bits) by the adder, the result must be in binade `code` ; FL_SPEC_TIE: round-half-even as an integer (ebits = floor(e/u)) ;
// FL_SPEC_TAIL: nothing. State (S, Eb) in / out; returns nonzero when a check fails (the state is then meaningless but bounded).
// Two arms only (a tie is integer work on the scalar unit, a fifth of the adder's arm): the walk is ONE wavefront's dependent
// chain, every instruction of it costs 5-6 cycles and a taken branch ~20.
FL_HD int fl_spec_event(int *S, int *Eb, int Q, unsigned ebits, int code)
{
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
    const int Sb = *S + Q;
    int fail = Sb >= FL_CHAIN_LIMIT ? 1 : 0;
    if (code == FL_SPEC_TIE) {       // the sum becomes even (a tie that leaves the binade: rare, the wavefront form does it)
        const int fl = (int)ebits;
        const int St = Sb + fl + ((Sb ^ fl) & 1);
        fail |= St >= FL_CHAIN_LIMIT ? 1 : 0;
        *S = St < FL_CHAIN_LIMIT ? St : FL_CHAIN_LIMIT;
        return fail;
    }
    // the adder's arm; the tail (code -1) runs through it with e = +0: the sum comes back as it went in, no binade to check
    union { unsigned u; float f; } eb; eb.u = ebits;
    const float s = fl_chain_from_units(Sb < FL_CHAIN_LIMIT ? Sb : FL_CHAIN_LIMIT, *Eb) + eb.f;
    int Ea;
    fl_chain_split(s, S, &Ea);
    fail |= (fl_chain_plain_only(s) ? 1 : 0) | ((code != FL_SPEC_TAIL && Ea != code) ? 1 : 0);
    *Eb = (code == FL_SPEC_TAIL) ? *Eb : Ea;
    return fail;
}

#if defined(__HIPCC__)
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the 16-lane DPP rows (row_shr 1, 2, 4, 8; lanes shifted in from
// outside the row read 0), then the three row totals through scalar registers
__device__ __forceinline__ int fl_wave_prefix_i32(int v, int lane)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}

// init + scr[0] + scr[1] + ... + scr[cnt-1] as ONE chain of float additions, computed by the 64 lanes of the calling wavefront
// (arguments uniform over the wavefront; scr in LDS with scr[cnt .. cnt + FL_CHAIN_STEP) == 0; bad: an element is negative or NaN).
// The result is in every lane.
__device__ __forceinline__ float fl_chain_f32_wave(const float *scr, int cnt, float init, bool bad)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    float s = init;
    int k = 0;
    const int lead = bad ? cnt : (cnt < FL_CHAIN_LEAD ? cnt : FL_CHAIN_LEAD);
    for (; k < lead; k++) s = s + scr[k];
    while (k < cnt) {
        if (fl_chain_plain_only(s)) break;
        int S, Eb;
        fl_chain_split(s, &S, &Eb);
        bool crossed = false;
        do {                                             // steps inside the binade: S stays an integer, no conversions
            FlChainLane L;
            const int tie = fl_chain_phase1(L, scr, k + FL_CHAIN_EPL * lane, Eb);
            if (__ballot(tie != 0) != 0ull) {
                int isc, xr;
                fl_chain_parity_map(L, &isc, &xr);
                const unsigned long long Cm = __ballot(isc), Xm = __ballot(xr & 1);
                fl_chain_ties(L, fl_chain_parity_in(Cm, Xm, lane, S), Eb);
            }
            fl_chain_sum(L);
            const int inc = fl_wave_prefix_i32(L.Qc, lane);
            const int total = __builtin_amdgcn_readlane(inc, 63);
            if (S + total < FL_CHAIN_LIMIT) {
                S += total;
                k += FL_CHAIN_STEP;
            } else {
                fl_chain_crossing(L, lane, S + inc - L.Qc);
                const unsigned long long xm = __ballot(L.cidx >= 0);
                const int Lc = __builtin_amdgcn_readfirstlane(__ffsll((long long)xm) - 1);       // lowest lane = earliest element
                const int c = __builtin_amdgcn_readlane(L.cidx, Lc);
                const int sp = __builtin_amdgcn_readlane(L.sprev, Lc);
                const float ec = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(L.ec), Lc));
                s = fl_chain_from_units(sp, Eb) + ec;    // the addition that leaves the binade: the adder rounds it
                k += c + 1;
                crossed = true;
            }
        } while (!crossed && k < cnt);
        if (!crossed) s = fl_chain_from_units(S, Eb);
    }
    for (; k < cnt; k++) s = s + scr[k];                 // (only after the `break` above)
    return s;
}

// inclusive prefix sum of doubles over the 64 lanes (same scheme as fl_wave_prefix_i32)
__device__ __forceinline__ double fl_dpp_row_shr_f64(double v, const int ctrl_sel)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false); break;
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fl_readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double fl_wave_prefix_f64(double v, int lane)
{
    v += fl_dpp_row_shr_f64(v, 1);
    v += fl_dpp_row_shr_f64(v, 2);
    v += fl_dpp_row_shr_f64(v, 4);
    v += fl_dpp_row_shr_f64(v, 8);
    const double r0 = fl_readlane_f64(v, 15), r1 = fl_readlane_f64(v, 31), r2 = fl_readlane_f64(v, 47);
    const int row = lane >> 4;
    return ((v + (row > 0 ? r0 : 0.0)) + (row > 1 ? r1 : 0.0)) + (row > 2 ? r2 : 0.0);
}

// The workgroup form (see above): init + scr[0] + ... + scr[cnt-1] as ONE chain of float additions by the 256 threads of the calling
// workgroup (arguments uniform over the workgroup, cnt <= 256 * FL_SPEC_EPT, scr in LDS and readable up to 256 * FL_SPEC_EPT). The
// result is in every thread. *fellback (optional, uniform): 1 when a check failed and the wavefront form did the chain.
// The first FL_SPEC_LEAD elements -- where the sum climbs a binade per element or two, i.e. all events -- are added plainly by
// wavefront 0 while the others load.
__device__ __forceinline__ float fl_chain_f32_block(const float *scr, int cnt, float init, bool bad, int *fellback = nullptr, long long *prof = nullptr)
{
#define FL_SPEC_PROF(i) do { if (prof && threadIdx.x == 0) prof[i] = (long long)clock64(); } while (0)
#pragma clang fp contract(off)
    __shared__ double w_tot[8];                           // [0..3] wavefront totals of the double prefix, [4..7] of the packed scan
    __shared__ double w_evA[FL_SPEC_MAX_EVENTS + 2];      // increments before event i (slot nev: all of them; the last slot: dummy)
    __shared__ unsigned w_evE[FL_SPEC_MAX_EVENTS + 2];    // the event's element (float bits) / a tie's floor
    __shared__ int w_evB[FL_SPEC_MAX_EVENTS + 2];         // the binade guessed for its successor / FL_SPEC_TIE / FL_SPEC_TAIL
    __shared__ int w_nev;
    __shared__ float w_res;
    __shared__ int w_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (fellback) *fellback = 0;
    if (bad || blockDim.x != 256 || cnt > 256 * FL_SPEC_EPT || fl_chain_plain_only(init)) {
        float f = 0.0f;
        if (tid < 64) f = fl_chain_f32_wave(scr, cnt, init, bad);
        if (tid == 0) w_res = f;
        __syncthreads();
        f = w_res;
        __syncthreads();
        if (fellback) *fellback = 1;
        return f;
    }
    FL_SPEC_PROF(0);
    // ---- the thread's elements and their running double sums
    const int lead = cnt < FL_SPEC_LEAD ? cnt : FL_SPEC_LEAD;
    float e[FL_SPEC_EPT];
    double d[FL_SPEC_EPT];
    {
        const float4 a = *reinterpret_cast<const float4 *>(scr + FL_SPEC_EPT * tid), b = *reinterpret_cast<const float4 *>(scr + FL_SPEC_EPT * tid + 4);
        e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
    }
    if (wave == 0) {                                       // (uniform addresses: every lane adds the same floats; x + 0 = x)
        float4 ld[FL_SPEC_LEAD / 4];
#pragma unroll
        for (int k = 0; k < FL_SPEC_LEAD / 4; k++) ld[k] = *reinterpret_cast<const float4 *>(scr + 4 * k);
        float s = init;
#pragma unroll
        for (int k = 0; k < FL_SPEC_LEAD / 4; k++) {
            s = s + (4 * k + 0 < cnt ? ld[k].x : 0.0f);
            s = s + (4 * k + 1 < cnt ? ld[k].y : 0.0f);
            s = s + (4 * k + 2 < cnt ? ld[k].z : 0.0f);
            s = s + (4 * k + 3 < cnt ? ld[k].w : 0.0f);
        }
        if (lane == 0) { w_res = s; w_nev = 0; w_fail = 0; }
    }
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < FL_SPEC_EPT; j++) {
        const int k = FL_SPEC_EPT * tid + j;
        e[j] = (k >= lead && k < cnt) ? e[j] : 0.0f;
        run += (double)e[j];
        d[j] = run;
    }
    const double pin = fl_wave_prefix_f64(run, lane);     // inclusive over the wavefront
    if (lane == 63) w_tot[wave] = pin;
    FL_SPEC_PROF(1);
    __syncthreads();
    FL_SPEC_PROF(2);
    const float init2 = w_res;                            // init + the leading elements
    // boundaries between elements: B[0] before the thread's first element ... B[8] behind its last. Neighbouring threads must see the
    // SAME double at their common boundary (an event is "the binade differs across this boundary"): B[8] is the thread's inclusive
    // prefix, B[0] the previous thread's, fetched from it (the first lane of a wavefront: the previous wavefront's last value is this
    // wavefront's offset, formed by the same additions).
    const double t0 = w_tot[0], t1 = w_tot[1], t2 = w_tot[2];
    double off = (double)init2;
    off = wave > 0 ? off + t0 : off;
    off = wave > 1 ? off + t1 : off;
    off = wave > 2 ? off + t2 : off;
    const double incl = off + pin;
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = off;
    int Eb[FL_SPEC_EPT + 1];
    Eb[0] = fl_spec_binade(excl);
#pragma unroll
    for (int j = 1; j < FL_SPEC_EPT; j++) Eb[j] = fl_spec_binade(excl + d[j - 1]);
    Eb[FL_SPEC_EPT] = fl_spec_binade(incl);
    int q[FL_SPEC_EPT], evm = 0, tim = 0, qsum = 0;
#pragma unroll
    for (int j = 0; j < FL_SPEC_EPT; j++) {
        int tf;                                            // (a tie's floor is formed again where the event is written: few)
        const int kind = fl_spec_elem(e[j], Eb[j], Eb[j + 1], &q[j], &tf);
        evm |= (kind != 0 ? 1 : 0) << j;
        tim |= (kind == 2 ? 1 : 0) << j;
        qsum += q[j];
    }
    const int nev_t = __builtin_popcount(evm);
    // ---- one scan for the increments and the event counts (integers in a double: exact)
    const double packed = (double)qsum + (double)nev_t * FL_SPEC_CNT_SCALE;
    const double sin_ = fl_wave_prefix_f64(packed, lane);
    if (lane == 63) w_tot[4 + wave] = sin_;
    FL_SPEC_PROF(3);
    __syncthreads();
    FL_SPEC_PROF(4);
    const double u0 = w_tot[4], u1 = w_tot[5], u2 = w_tot[6];
    const double soff = (wave > 0 ? u0 : 0.0) + (wave > 1 ? u1 : 0.0) + (wave > 2 ? u2 : 0.0);      // (integers: any order)
    const double sincl = soff + sin_, sexcl = sincl - packed;
    if (evm) {                                             // (few threads; stores to a dummy slot instead of a branch per element)
        const double cnt_excl = floor(sexcl * (1.0 / FL_SPEC_CNT_SCALE));
        double A = sexcl - cnt_excl * FL_SPEC_CNT_SCALE;    // increments of all elements before this thread's
        const int slot0 = (int)cnt_excl;
#pragma unroll
        for (int j = 0; j < FL_SPEC_EPT; j++) {
            const bool ev = (evm >> j) & 1, tie = (tim >> j) & 1;
            const int slot = slot0 + __builtin_popcount(evm & ((1 << j) - 1));
            const int at = (ev && slot < FL_SPEC_MAX_EVENTS) ? slot : FL_SPEC_MAX_EVENTS + 1;
            w_evA[at] = A; w_evE[at] = tie ? (unsigned)(int)(ldexpf(e[j], 150 - Eb[j]) - 0.5f) : __float_as_uint(e[j]); w_evB[at] = tie ? FL_SPEC_TIE : Eb[j + 1];
            if (ev && slot >= FL_SPEC_MAX_EVENTS) w_fail = 1;
            A += (double)q[j];
        }
    }
    if (tid == 255) {
        const double cnt_incl = floor(sincl * (1.0 / FL_SPEC_CNT_SCALE));
        const int nev = (int)cnt_incl;
        w_nev = nev;
        if (nev <= FL_SPEC_MAX_EVENTS) { w_evA[nev] = sincl - cnt_incl * FL_SPEC_CNT_SCALE; w_evE[nev] = 0u; w_evB[nev] = FL_SPEC_TAIL; }
    }
    FL_SPEC_PROF(5);
    __syncthreads();
    FL_SPEC_PROF(6);
    // ---- the walk over the events (wavefront 0; lane i holds event i, the loop reads them lane by lane)
    if (wave == 0) {
        const int nev = w_nev;
        int fail = w_fail | (nev > FL_SPEC_MAX_EVENTS ? 1 : 0) | (fl_chain_plain_only(init2) ? 1 : 0);
        float res = 0.0f;
        if (!fail) {
            const double Ai = (lane <= nev) ? w_evA[lane] : 0.0, Ap = (lane > 0 && lane <= nev) ? w_evA[lane - 1] : 0.0;
            const double dq = Ai - Ap;
            const int Qv = (dq < 16777216.0) ? (int)dq : FL_CHAIN_LIMIT;
            const int Ev = (lane <= nev) ? (int)w_evE[lane] : 0;
            const int Bv = (lane <= nev) ? w_evB[lane] : FL_SPEC_TAIL;
            int S, Ebc;
            fl_chain_split(init2, &S, &Ebc);
            for (int i = 0; i <= nev; i++) {                 // (no early exit: a failed walk just runs to its end)
                const int Q = __builtin_amdgcn_readlane(Qv, i), B = __builtin_amdgcn_readlane(Bv, i);
                const unsigned ee = (unsigned)__builtin_amdgcn_readlane(Ev, i);
                fail |= fl_spec_event(&S, &Ebc, Q, ee, B);
            }
            fail = __builtin_amdgcn_readfirstlane(fail);
            res = fl_chain_from_units(S, Ebc);
        }
        if (fail) res = fl_chain_f32_wave(scr, cnt, init, false);
        if (lane == 0) { w_res = res; w_fail = fail; }
        FL_SPEC_PROF(7);
    }
    __syncthreads();
    FL_SPEC_PROF(8);
    const float f = w_res;
    if (fellback) *fellback = w_fail;
    __syncthreads();                                      // (w_res / w_fail are rewritten by the next call)
    FL_SPEC_PROF(9);
    return f;
#undef FL_SPEC_PROF
}
#endif
