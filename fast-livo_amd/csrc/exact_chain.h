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
#endif
