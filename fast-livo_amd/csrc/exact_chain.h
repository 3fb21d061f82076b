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
//   associatively (x -> x ^ c, or x -> 0): 256 elements per step (4 per lane) need two ballots for the parities and one integer
//   prefix sum for the S values. The first element whose exact sum reaches 2^24 u leaves the binade: the adder itself does that one
//   addition (its rounding unit is 2u or more and does depend on S), and the next step starts from the new binade.
//
// ~8 steps for 2 k patches plus one per binade the sum climbs through (the first 16 additions, which climb fastest, are done
// plainly): ~3 us instead of ~20. Elements are >= 0 by construction (sums of squares); a negative/NaN element or a non-finite
// running sum falls back to the plain chain for the rest.
//
// The per-lane phases are plain functions (FL_HD): tests/host_emul runs them lane by lane on the CPU against the plain loop
// (tests/test_exact_chain_cpu.py); fl_debug_chain runs the device driver against the plain loop on the GPU.
#pragma once
#include "fl_math.h"

#define FL_CHAIN_EPL 4                       /* elements per lane and step */
#define FL_CHAIN_LEAD 16                     /* leading elements added plainly */
#define FL_CHAIN_LIMIT (1 << 24)             /* S of the next binade, in units u */

struct FlChainLane {
    float e[FL_CHAIN_EPL];
    int q[FL_CHAIN_EPL];                     // integer increment (floor for a tie until phase 2 decides)
    int tie;                                 // bit i: element i is an exact tie
    int isc, xr;                             // the lane's parity map: out = isc ? xr : in ^ xr
    int bad;                                 // a negative or NaN element
    int Q, Qc;                               // sum of the increments, capped at 2^24 (a lane that large has left the binade anyway)
    int cidx, sprev;                         // first element (index within the step) whose sum reaches 2^24, S before it
    float ec;
};

// binade of a finite s >= 0: E = max(exponent, -126) (subnormals share u = 2^-149 with the first normal binade)
FL_HD int fl_chain_binade(float s)
{
    union { float f; unsigned u; } b; b.f = s;
    const int ex = (int)((b.u >> 23) & 0xffu) - 127;
    return ex < -126 ? -126 : ex;
}
FL_HD bool fl_chain_plain_only(float s)      // inf / NaN / negative running sum
{
    union { float f; unsigned u; } b; b.f = s;
    return ((b.u >> 23) & 0xffu) == 255u || (b.u >> 31) != 0u;
}
FL_HD double fl_chain_scale(int E)           // 2^(23 - E) = 1 / u
{
    union { double d; unsigned long long u; } b;
    b.u = (unsigned long long)(1023 + 23 - E) << 52;
    return b.d;
}
FL_HD float fl_chain_from_units(int S, int E) { return ldexpf((float)S, E - 23); }     // S < 2^24: exact

// phase 1: this lane's elements as increments + the lane's parity map
FL_HD void fl_chain_phase1(FlChainLane &L, const float *scr, int cnt, int first, double scale)
{
    L.tie = 0; L.isc = 0; L.xr = 0; L.bad = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        const int idx = first + i;
        const float e = idx < cnt ? scr[idx] : 0.0f;
        L.e[i] = e;
        if (!(e >= 0.0f)) L.bad = 1;
        double x = (double)e * scale;                    // exact: a power of two
        x = x < 16777216.0 ? x : 16777216.0;             // (also inf; NaN is `bad`)
        const double f = floor(x), fr = x - f;           // exact
        const int t = (fr == 0.5) ? 1 : 0;
        L.q[i] = (int)f + ((fr > 0.5) ? 1 : 0);
        L.tie |= t << i;
        if (t) { L.isc = 1; L.xr = 0; } else L.xr ^= L.q[i] & 1;
    }
}

// parity of S before this lane's first element, from the ballots of (isc, xr) over the lanes and the parity of the step's S
FL_HD int fl_chain_parity_in(unsigned long long Cm, unsigned long long Xm, int lane, int S)
{
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long cb = Cm & below;
    if (cb) {
        int t = 63;
        while (!((cb >> t) & 1ull)) t--;                 // (device: 63 - clz)
        return (int)(__builtin_popcountll(Xm & below & ~((1ull << t) - 1ull)) & 1);       // lane t's bit is its constant
    }
    return (int)(((unsigned)S + (unsigned)__builtin_popcountll(Xm & below)) & 1u);
}

// phase 2: ties decided, the lane's total increment
FL_HD void fl_chain_phase2(FlChainLane &L, int parity_in)
{
    int p = parity_in, Q = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < FL_CHAIN_EPL; i++) {
        if ((L.tie >> i) & 1) { L.q[i] += (p ^ L.q[i]) & 1; p = 0; }      // round half to even: the sum becomes even
        else p ^= L.q[i] & 1;
        Q += L.q[i];
    }
    L.Q = Q;
    L.Qc = Q < FL_CHAIN_LIMIT ? Q : FL_CHAIN_LIMIT;
}

// phase 3: first element of the lane whose sum reaches the next binade; s_before = S before the lane's first element
FL_HD void fl_chain_phase3(FlChainLane &L, int lane, int s_before)
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
// (arguments uniform over the wavefront, scr in LDS). The result is in every lane.
__device__ __forceinline__ float fl_chain_f32_wave(const float *scr, int cnt, float init)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    float s = init;
    int k = 0;
    const int lead = cnt < FL_CHAIN_LEAD ? cnt : FL_CHAIN_LEAD;
    for (; k < lead; k++) s = s + scr[k];
    while (k < cnt) {
        if (fl_chain_plain_only(s)) break;
        const int E = fl_chain_binade(s);
        const double scale = fl_chain_scale(E);
        const int S = (int)((double)s * scale);
        FlChainLane L;
        fl_chain_phase1(L, scr, cnt, k + FL_CHAIN_EPL * lane, scale);
        if (__ballot(L.bad) != 0ull) break;
        const unsigned long long Cm = __ballot(L.isc), Xm = __ballot(L.xr & 1);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const unsigned long long cb = Cm & below;
        int pin;
        if (cb) {
            const int t = 63 - __clzll((long long)cb);
            pin = __popcll(Xm & below & ~((1ull << t) - 1ull)) & 1;
        } else {
            pin = (S + __popcll(Xm & below)) & 1;
        }
        fl_chain_phase2(L, pin);
        const int inc = fl_wave_prefix_i32(L.Qc, lane);
        fl_chain_phase3(L, lane, S + inc - L.Qc);
        const unsigned long long xm = __ballot(L.cidx >= 0);
        if (xm) {                                        // lowest lane = earliest element
            const int Lc = __builtin_amdgcn_readfirstlane(__ffsll((long long)xm) - 1);
            const int c = __builtin_amdgcn_readlane(L.cidx, Lc);
            const int sp = __builtin_amdgcn_readlane(L.sprev, Lc);
            const float ec = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(L.ec), Lc));
            s = fl_chain_from_units(sp, E) + ec;         // the addition that leaves the binade: the adder rounds it
            k += c + 1;
        } else {
            s = fl_chain_from_units(S + __builtin_amdgcn_readlane(inc, 63), E);
            k += 64 * FL_CHAIN_EPL;
        }
    }
    for (; k < cnt; k++) s = s + scr[k];                 // (only after a `break` above)
    return s;
}
#endif
