// objective_chunk.h -- what a wave does with one 512-point chunk for a group of peaks: Lorentzians over a common
// denominator (general and scaled pair form, batch inversion), Gaussians directly and by recurrence.
#pragma once
#include "objective_math.h"

namespace nmrfit {
namespace {

// ---- per-chunk building blocks ---------------------------------------------------------------
// Per-(particle, peak) constants in LDS, two arrays per wave: PeakLor (32 B: read in the main
// loop as one broadcast ds_read_b128 + one ds_read_b64) and PeakWin (16 B: Gaussian window).

// Lorentzians of G peaks over one common denominator.  With s_k = 1 + t_k^2 >= 1, a pair is
//   AL0/s0 + AL1/s1 = (AL0 s1 + AL1 s0) / (s0 s1)
// and (numerator, denominator) pairs combine the same way up a binary tree:
//   (n0, d0) + (n1, d1) = (n0 d1 + n1 d0, d0 d1)            3 FMA-class ops per combine
// -> ONE reciprocal per G (point, peak) units: 2G + 3(G-1) + 3 FMA-class ops + v_rcp_f64,
// against G x (5 + v_rcp_f64) done one by one (G = 8: 5.0 ops + 1/8 rcp per unit).  All
// products are of factors >= 1 and |t| is capped at 1e18 when the record is staged, so the
// denominator of 8 peaks stays below 1e289.
template <int G, int LO, int HI>
__device__ __forceinline__ void lorentz_tree(const double (&a)[G], const double (&s)[G], double &n, double &d)
{
    if constexpr (HI - LO == 1) {
        n = a[LO];
        d = s[LO];
    } else {
        constexpr int MID = LO + (HI - LO + 1) / 2;
        double n0, d0, n1, d1;
        lorentz_tree<G, LO, MID>(a, s, n0, d0);
        lorentz_tree<G, MID, HI>(a, s, n1, d1);
        d = d0 * d1;
        n = __builtin_fma(n0, d1, n1 * d0);
    }
}

template <int G>
__device__ __forceinline__ void lorentz_group(const PeakLor *r, const double (&wv)[kPointsPerLane],
                                              double (&acc)[kPointsPerLane])
{
    double ih[G], c[G], a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        ih[g] = r[g].ihw;
        c[g] = r[g].c;
        a[g] = r[g].al;
    }
    // a scheduling fence every kInterleave points bounds how many points the scheduler may
    // interleave (register pressure); interleaved A/B on one device (tools/ab.py) shows no
    // difference between 1, 2, 4 and 8 on C3 (within +-0.4 %)
#pragma unroll
    for (int q = 0; q < kPointsPerLane; ++q) {
        double s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double t = __builtin_fma(wv[q], ih[g], c[g]);
            s[g] = __builtin_fma(t, t, 1.0);
        }
        double num, den;
        lorentz_tree<G, 0, G>(a, s, num, den);
        acc[q] = __builtin_fma(num, rcp64(den), acc[q]);
        if ((q + 1) % kInterleave == 0) __builtin_amdgcn_sched_barrier(0);
    }
}

// The same group when every amplitude is positive: al/(1+t^2) = 1/s', s' = ia + t'^2 with
// t' = t/sqrt(al), ia = 1/al (scaled constants staged beside the plain ones).  A pair of plain
// reciprocals combines in TWO operations, 1/s0 + 1/s1 = (s0 + s1)/(s0 s1), instead of three,
// so a group of 8 costs 16 + 8 + 6 + 3 + 4 = 37 operations per point instead of 41.  Staging
// marks the groups for which this is safe (PeakFast::ok); the others take lorentz_group.
template <int G>
__device__ __forceinline__ void lorentz_group_fast(const PeakFast *r, const double (&wv)[kPointsPerLane],
                                                   double (&acc)[kPointsPerLane])
{
    static_assert(G % 2 == 0, "pairs");
    double ih[G], c[G], ia[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        ih[g] = r[g].ihs;
        c[g] = r[g].cs;
        ia[g] = r[g].ia;
    }
    // One reciprocal serves kBatchInv points (batch inversion): r = 1/(d0 d1), 1/d0 = r d1,
    // 1/d1 = r d0 -- a multiply is ~4 cycles, v_rcp_f64 16.  The staging bound on the group's
    // denominator is divided by kBatchInv accordingly.
    constexpr int B = kBatchInv;
    static_assert(kPointsPerLane % B == 0, "batch");
    auto point = [&](const double w, double &num, double &den) {
        double pn[G / 2], pd[G / 2];
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const double t0 = __builtin_fma(w, ih[g], c[g]);
            const double t1 = __builtin_fma(w, ih[g + 1], c[g + 1]);
            const double s0 = __builtin_fma(t0, t0, ia[g]);
            const double s1 = __builtin_fma(t1, t1, ia[g + 1]);
            pn[g / 2] = s0 + s1;
            pd[g / 2] = s0 * s1;
        }
        lorentz_tree<G / 2, 0, G / 2>(pn, pd, num, den);
    };
    static_assert(B == 4, "four points per reciprocal");
    // Four points per reciprocal, folded pair by pair: once two points' (numerator, denominator) are known they
    // become (n0 d1, n1 d0, d0 d1) -- three values instead of four held while the other pair is worked out (the
    // same 13 operations + one reciprocal per batch as holding two numerator / denominator pairs; values move by one rounding)
#pragma unroll
    for (int q0 = 0; q0 < kPointsPerLane; q0 += 4) {
        double n0, d0, n1, d1;
        point(wv[q0], n0, d0);
        point(wv[q0 + 1], n1, d1);
        const double p01 = d0 * d1, a0 = n0 * d1, a1 = n1 * d0;
        __builtin_amdgcn_sched_barrier(0);
        double n2, d2, n3, d3;
        point(wv[q0 + 2], n2, d2);
        point(wv[q0 + 3], n3, d3);
        const double p23 = d2 * d3, a2 = n2 * d3, a3 = n3 * d2;
        const double r = rcp64(p01 * p23);
        const double r01 = r * p23, r23 = r * p01;
        acc[q0] = __builtin_fma(a0, r01, acc[q0]);
        acc[q0 + 1] = __builtin_fma(a1, r01, acc[q0 + 1]);
        acc[q0 + 2] = __builtin_fma(a2, r23, acc[q0 + 2]);
        acc[q0 + 3] = __builtin_fma(a3, r23, acc[q0 + 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A single peak in the scaled form (the odd one out of a short tail group): 1/s' per point, one
// reciprocal per four points.
__device__ __forceinline__ void lorentz_one_fast(const PeakFast *r, const double (&wv)[kPointsPerLane],
                                                 double (&acc)[kPointsPerLane])
{
    const double ih = r->ihs, c = r->cs, ia = r->ia;
#pragma unroll
    for (int q0 = 0; q0 < kPointsPerLane; q0 += 4) {
        double s[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double t = __builtin_fma(wv[q0 + b], ih, c);
            s[b] = __builtin_fma(t, t, ia);
        }
        const double p01 = s[0] * s[1], p23 = s[2] * s[3];
        const double rr = rcp64(p01 * p23);
        const double a01 = rr * p23, a23 = rr * p01;
        acc[q0] = __builtin_fma(a01, s[1], acc[q0]);
        acc[q0 + 1] = __builtin_fma(a01, s[0], acc[q0 + 1]);
        acc[q0 + 2] = __builtin_fma(a23, s[3], acc[q0 + 2]);
        acc[q0 + 3] = __builtin_fma(a23, s[2], acc[q0 + 3]);
    }
}

// The short tail group (1..7 peaks) in the scaled form: an even-sized group, then the odd peak.
__device__ __forceinline__ void lorentz_tail_fast(int n, const PeakFast *r, const double (&wv)[kPointsPerLane],
                                                  double (&acc)[kPointsPerLane])
{
    const int even = n & ~1;
    if (even == 6)
        lorentz_group_fast<6>(r, wv, acc);
    else if (even == 4)
        lorentz_group_fast<4>(r, wv, acc);
    else if (even == 2)
        lorentz_group_fast<2>(r, wv, acc);
    if (n & 1) lorentz_one_fast(r + even, wv, acc);
}

// One peak over the lane's points with one reciprocal per four points (batch inversion; with
// s >= 1 and |t| <= 1e18 the product of four stays below 1e145): the near peaks of FARFIELD.
__device__ __forceinline__ void lorentz_one(const PeakLor *r, const double (&wv)[kPointsPerLane],
                                            double (&acc)[kPointsPerLane])
{
    const double ih = r->ihw, c = r->c, al = r->al;
#pragma unroll
    for (int q0 = 0; q0 < kPointsPerLane; q0 += 4) {
        double s[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double t = __builtin_fma(wv[q0 + b], ih, c);
            s[b] = __builtin_fma(t, t, 1.0);
        }
        const double p01 = s[0] * s[1], p23 = s[2] * s[3];
        const double rr = rcp64(p01 * p23);
        const double a01 = al * (rr * p23), a23 = al * (rr * p01);
        acc[q0] = __builtin_fma(a01, s[1], acc[q0]);
        acc[q0 + 1] = __builtin_fma(a01, s[0], acc[q0 + 1]);
        acc[q0 + 2] = __builtin_fma(a23, s[3], acc[q0 + 2]);
        acc[q0 + 3] = __builtin_fma(a23, s[2], acc[q0 + 3]);
    }
}

// group of a run-time size 1..GMAX-1 (tail of a pass)
template <int GMAX>
__device__ __forceinline__ void lorentz_tail(int n, const PeakLor *r, const double (&wv)[kPointsPerLane],
                                             double (&acc)[kPointsPerLane])
{
    if constexpr (GMAX > 1) {
        if (n == GMAX - 1)
            lorentz_group<GMAX - 1>(r, wv, acc);
        else
            lorentz_tail<GMAX - 1>(n, r, wv, acc);
    }
}

// Gaussian of one peak: acc += AG2 * 2^-(1 + t^2)   (t recomputed: cheaper than keeping s live)
// ROWS (round 5): a chunk that touches the peak's window is evaluated ROW by row -- a row is the 64 consecutive grid
// points the lanes hold at one q -- and a row none of whose points lies inside the window (|t| <= sqrt(63): beyond,
// the term is below 2^-64 of its amplitude, the same criterion as the chunk-level skip) costs one FMA and one
// wave-uniform compare instead of the 21 operations of an exp2.  On a coarse grid -- the reference's own 4096-point
// spectra: a chunk spans 1/8 of the spectrum, a line's window a third of that -- two rows in three are skipped, and
// the recurrence (whose premise is a fine grid) never applies there.  NaN counts as inside: it propagates as before.
constexpr double kGaussWindowT = 7.9372539331937721;   // sqrt(63), in half-widths (kGaussWindow is the same in widths)
template <bool ROWS>
__device__ __forceinline__ void gauss_add(const PeakLor *r, const double (&wv)[kPointsPerLane],
                                          double (&acc)[kPointsPerLane])
{
    const double ihw = r->ihw, c = r->c, ag2 = r->ag2;
    // which rows have a point inside the window: one scalar bit per row (each t dies at its compare; it is recomputed
    // below for the rows that are evaluated -- one FMA, against eight values held across the rows' branches)
    unsigned rows = 0xffu;
    if (ROWS) {
        rows = 0u;
#pragma unroll
        for (int q = 0; q < kPointsPerLane; ++q) {
            const double t = __builtin_fma(wv[q], ihw, c);
            rows |= (__ballot(!(fabs(t) > kGaussWindowT)) != 0ull) ? (1u << q) : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < kPointsPerLane; ++q) {
        // (the accumulator is updated unconditionally, with a zero for a skipped row: a conditional update makes the
        // compiler carry two copies of all eight accumulators through the peak loop -- +20 VGPRs, a wave per SIMD)
        double e = 0.0;
        if (!ROWS || (rows & (1u << q)) != 0u) {
            const double t = __builtin_fma(wv[q], ihw, c);
            const double s = __builtin_fma(t, t, 1.0);
            e = ROWS ? exp2_neg_sc(-s) : exp2_neg(-s);   // (the same operations: bit-identical values)
        }
        acc[q] = __builtin_fma(ag2, e, acc[q]);
    }
}

// The same over one FULL chunk of a uniformly spaced grid, by recurrence from the lane's first
// point: with t[q] = t[0] + q*d (d = 64 grid steps in half-widths) the ratio of successive values
// is R[q] = 2^-(2 t[q] d + d^2) and the ratio of successive ratios is the constant C = 2^-(2 d^2),
// so seven of the eight exp2 become two multiplies each.  Valid while nothing leaves the fp64
// range: |d| <= 2 bounds |t| of every lane of a chunk that touches the window by 8 + 16, i.e.
// 2^-577 <= 2^-s and R <= 2^100.  The grid's departure from uniform spacing (`devk`, scaled so
// that |ihw|*devk <= 1 means <= 1e-10 relative on the in-window values; a linspace grid gives
// ~4e-12) decides when the constants are staged: (d, C) sit in LDS beside the other per-peak
// records, and ONE flag per particle says whether every peak qualifies -- the recurrence and the
// direct form then run as two separate loops (a branch per peak would make the compiler copy the
// eight accumulators on every arm).
__device__ __forceinline__ void gauss_add_rec(const PeakLor *r, const double2 *rec, const double (&wv)[kPointsPerLane],
                                              double (&acc)[kPointsPerLane])
{
    const double2 dc = *rec;     // (d, C)
    const double ihw = r->ihw, c = r->c, ag2 = r->ag2;
    const double t0 = __builtin_fma(wv[0], ihw, c);
    double g = ag2 * exp2_neg(-__builtin_fma(t0, t0, 1.0));
    double ratio = exp2_neg(-__builtin_fma(t0 + t0, dc.x, dc.x * dc.x));
    acc[0] += g;
#pragma unroll
    for (int q = 1; q < kPointsPerLane; ++q) {
        g *= ratio;
        acc[q] += g;
        if (q + 1 < kPointsPerLane) ratio *= dc.y;
    }
}

}  // namespace
}  // namespace nmrfit
