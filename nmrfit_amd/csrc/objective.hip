// objective.hip -- the hot path: batched nmrfit objective / residual for gfx950 (CDNA4).
//
// What it computes (reference: nmrfit/equations.py:152-212 `objective`, :115-149 `voigt`,
// nmrfit/proc_autophase.py:9-36 `ps2`), for every particle i of a swarm X[S, 4+3P]:
//
//     phi_j   = p0 + (p1*j)/N                                    proc_autophase.py:31
//     Vd_j    = cos(phi_j)*u_j - sin(phi_j)*v_j                  proc_autophase.py:35 (real part)
//     Vf_j    = sum_k [ yoff + a_k*( r*L_k(w_j) + (1-r)*G_k(w_j) ) ]   equations.py:141-147,195
//     f_i     = sqrt( mean_j ( weights_j*(Vd_j - Vf_j) )^2 )     equations.py:202
//
// MI355X mapping (no MFMA: there is no contraction here; the kernel is bound by the fp64
// vector-ALU issue rate):
//   * one WAVE owns one (particle, grid-segment); its 64 lanes stride the grid points so
//     the w/u/v/weights reads are coalesced 512-B rows that stay L2-resident (the four
//     arrays are shared by every particle: <= 2 MiB at N = 65536);
//   * each lane register-blocks 8 grid points, so the per-peak constants are fetched once
//     per 8 points.  They are wave-uniform and live in LDS (48 B per peak; the main loop
//     reads 24 B of it with broadcast ds_reads), staged once per wave from the particle's
//     row of X;
//   * algebra (derived from equations.py:141-147, exact in real arithmetic): with
//     t = (w-loc)*(2/width), s = 1 + t^2:  L = (2/(pi*width))/s  and
//     exp(-((w-loc)/(width/(2 sqrt(ln2))))^2) = 2^(-t^2) = 2*2^(-s), so one fma chain per
//     point and peak: t = fma(wc, ihw, c); s = fma(t, t, 1); acc += AL*rcp(s) + AG2*exp2(-s);
//   * the Lorentzians of eight peaks share one reciprocal (common denominator, combined up
//     a binary tree of (numerator, denominator) pairs); rcp: v_rcp_f64 + one Newton step
//     (relative error 2.2e-15);
//     exp2(-s): round-to-nearest split + degree-11 polynomial + v_ldexp_f64 (<= 3e-16);
//   * the Gaussian term is < 2^-64 of its amplitude once |w-loc| > 3.97*width; a wave
//     skips it for a whole 512-point chunk when the chunk's [min,max] of w (precomputed
//     at context creation) misses that window -- a wave-uniform branch, exact to fp64
//     rounding, and the common case (a line is ~100x narrower than the spectrum);
//   * the phase ramp is a complex rotation recurrence z <- z*rho (4 fp64 ops per point),
//     re-seeded at the start of each of <= 16 blocks of the grid;
//   * sum of squares: per-lane fp64 accumulation over a block, then a wave64 shuffle tree.
//     With one segment per particle the wave writes f directly; otherwise a tiny second
//     kernel adds the per-block sums in grid order (deterministic, no atomics, and the same
//     order whatever the segmentation: f does not depend on launch geometry or on sharding).
#include "nmrfit_internal.h"
#include "pso_update.h"

#define NMRFIT_DAWSON_QUAL __device__ const
#include "dawson_coeffs.h"

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace nmrfit {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr double kInvPi = 0.31830988618379067154;
constexpr double kSqrtLn2OverPi = 0.46971863934982566689;   // sqrt(ln2/pi)
#ifndef NMRFIT_INTERLEAVE
#define NMRFIT_INTERLEAVE 4     // tuning knobs, A/B-tested with tools/ab.py
#endif
#ifndef NMRFIT_GROUP
#define NMRFIT_GROUP 8
#endif
#ifndef NMRFIT_BATCHINV
#define NMRFIT_BATCHINV 4     // points sharing one reciprocal in the pair-form groups (1, 2 or 4)
#endif
#ifndef NMRFIT_FASTPAIR
#define NMRFIT_FASTPAIR 1     // two-operation pair form for groups of positive Lorentzian amplitudes
#endif
#ifndef NMRFIT_MIN_WAVES
#define NMRFIT_MIN_WAVES 3
#endif
#ifndef NMRFIT_PAIRFOLD
#define NMRFIT_PAIRFOLD 1
#endif
#ifndef NMRFIT_SADDR
#define NMRFIT_SADDR 1
#endif
#ifndef NMRFIT_PREFETCH_W
#define NMRFIT_PREFETCH_W 0   // w of the next chunk requested in the epilogue of the current one: measured +1 % (C3) ... +5 %
#endif                        // (204 x 4096 x 6) -- 16 register copies per chunk and a fuller epilogue; A/B knob
#ifndef NMRFIT_DISP_INTERLEAVE
#define NMRFIT_DISP_INTERLEAVE 4   // fit_im=True: points of the last peak's dispersion line in flight together
#endif
constexpr int kBatchInv = NMRFIT_BATCHINV;
// objective_kernel's static LDS: block sums (x2 with the imaginary channel); f; then what the end of a fused swarm
// generation needs, parked by the first instructions of the kernel and by its prologue: [+1] personal bests on,
// [+2] p, [+3] S, [+4] the row's LDS offset, [+5] this particle's fp, [+6] fg, [+7] completed generations
constexpr int kWsumsCount = 2 * kMaxBlocks + 8;
#ifndef NMRFIT_DIAG_ABLATE
#define NMRFIT_DIAG_ABLATE 0   // diagnostic builds (wrong values on purpose): 1 no expansions, 2 no near peaks / Gaussians,
#endif                         // 4 no Horner, 8 no epilogue arithmetic -- what each phase of the far-field chunk costs
constexpr int kAblate = NMRFIT_DIAG_ABLATE;
#ifndef NMRFIT_FF_PIPE
#define NMRFIT_FF_PIPE 0   // FARFIELD, P <= 32: the NEXT pair's expansions started in the odd chunk before it (measured: +2 %; A/B knob)
#endif
constexpr bool kFarPipe = NMRFIT_FF_PIPE != 0;
#ifndef NMRFIT_FF_HORNER_FIRST
#define NMRFIT_FF_HORNER_FIRST 1   // FARFIELD, P <= 32: the shared polynomial before the near peaks and Gaussians (A/B knob)
#endif
constexpr bool kHornerFirst = NMRFIT_FF_HORNER_FIRST != 0;
#ifdef NMRFIT_DIAG_REMAP
constexpr bool kOneWorkgroupParticle = false;
#else
constexpr bool kOneWorkgroupParticle = true;
#endif
constexpr int kFarTerms = 16;      // Taylor terms of the far-field expansion (rho <= 0.1 -> 1e-16)
constexpr int kFarPad = 68;        // row stride (doubles) of the per-wave coefficient scratch in LDS: lane l writes column
                                   // l + l/16 of 16 rows, then reads 16 consecutive doubles of row l/4 from column 17*(l%4) --
                                   // for ds_read_b64 / ds_read2_b64 (32- and 16-lane groups) every lane of a group then hits
                                   // its own bank.  SQ_LDS_BANK_CONFLICT of the far-field kernel is 3.2e6 cycles per C3 launch
                                   // (DEFAULT: 5e4) all the same: 12 cycles per chunk PAIR, from the per-lane reads of the
                                   // 32-byte peak records (lanes i and i + 8 of a ds_read_b128 group share banks) -- 0.3 % of a
                                   // pair's ~4500 cycles, not worth a padded record (profiles/r04/farfield_c3_pmc_summary.json)
constexpr size_t kSharedPrologueBytes = ((2 + 2 * kWave) * sizeof(double) + 16 * sizeof(int) + 15) & ~(size_t)15;
constexpr double kGaussWindow = 3.9686269665968861;          // 0.5*sqrt(63): 2^-(1+t^2) < 2^-64 beyond

// ---- fp64 helpers (coefficients: tools/gen_poly.py) ---------------------------------------

// 1/s: v_rcp_f64 (measured 4.6e-8 relative on gfx950) + one Newton step -> 2.2e-15, full fp64
// range.  Same issue cost as an f32 seed (16 cycles vs cvt + v_rcp_f32 + cvt) and more accurate.
__device__ __forceinline__ double rcp64(double s)
{
    const double r0 = __builtin_amdgcn_rcp(s);
    const double e = __builtin_fma(-s, r0, 1.0);
    return __builtin_fma(r0, e, r0);
}

// 2^x for x <= 0.  n = rint(x), f = x - n in [-1/2, 1/2], degree-11 interpolant of 2^f
// (max relative error 2.2e-16 in float64 Horner form), scaled by v_ldexp_f64.
__device__ __forceinline__ double exp2_neg(double x)
{
    x = fmax(x, -1100.0);   // 2^-1100 == 0 in fp64; keeps n inside int range
    const double n = __builtin_rint(x);
    const double f = x - n;
    double p = 4.455817908336064493e-10;
    p = __builtin_fma(p, f, 7.0741942972885210056e-9);
    p = __builtin_fma(p, f, 1.0178057087733941105e-7);
    p = __builtin_fma(p, f, 1.3215432535912376166e-6);
    p = __builtin_fma(p, f, 1.5252733841556772589e-5);
    p = __builtin_fma(p, f, 1.5403530463724354209e-4);
    p = __builtin_fma(p, f, 1.3333558146406470697e-3);
    p = __builtin_fma(p, f, 9.6181291075872566681e-3);
    p = __builtin_fma(p, f, 5.5504108664821627039e-2);
    p = __builtin_fma(p, f, 2.4022650695910159567e-1);
    p = __builtin_fma(p, f, 6.9314718055994530925e-1);
    p = __builtin_fma(p, f, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// sin and cos of phi by a 3-term Cody-Waite reduction by pi/2 (FMA form: each step is exact
// before its single rounding, so the reduced angle stays accurate to ~|k| * 1e-26 + 1e-16)
// + polynomials on [-pi/4, pi/4] (<= 3e-16).  Branch-free; good to ~1e-14 up to |phi| ~ 1e12.
__device__ __forceinline__ void sincos_cw(double phi, double *s_out, double *c_out)
{
    const double k = __builtin_rint(phi * 0.6366197723675814);
    double r = __builtin_fma(-k, 1.5707963267341256, phi);
    r = __builtin_fma(-k, 6.077100506303966e-11, r);
    r = __builtin_fma(-k, 2.0222662487959506e-21, r);
    const double y = r * r;
    double ps = 1.5894736651849095259e-10;
    ps = __builtin_fma(ps, y, -2.5050716974102745028e-8);
    ps = __builtin_fma(ps, y, 2.7557313376400129128e-6);
    ps = __builtin_fma(ps, y, -1.9841269828650300013e-4);
    ps = __builtin_fma(ps, y, 8.3333333333203624567e-3);
    ps = __builtin_fma(ps, y, -1.6666666666666616666e-1);
    const double sn = __builtin_fma(r * y, ps, r);
    double pc = -1.1353379638297574126e-11;
    pc = __builtin_fma(pc, y, 2.0875582380663953044e-9);
    pc = __builtin_fma(pc, y, -2.7557313097790086271e-7);
    pc = __builtin_fma(pc, y, 2.4801587283881153004e-5);
    pc = __builtin_fma(pc, y, -1.3888888888861094596e-3);
    pc = __builtin_fma(pc, y, 4.1666666666666452389e-2);
    pc = __builtin_fma(pc, y, -0.5);
    const double cs = __builtin_fma(pc, y, 1.0);
    const int q = (int)(k - 4.0 * __builtin_floor(k * 0.25));   // k mod 4 in {0,1,2,3}, any |k| < 2^52
    const double s1 = (q & 1) ? cs : sn;
    const double c1 = (q & 1) ? sn : cs;
    *s_out = (q & 2) ? -s1 : s1;
    *c_out = ((q + 1) & 2) ? -c1 : c1;
}

// the same with the libdevice routine (Payne-Hanek) for absurd arguments; not used inside
// the chunk loop (its register footprint would spill)
__device__ __forceinline__ void sincos_fast(double phi, double *s_out, double *c_out)
{
    if (!(fabs(phi) < 1.0e12)) {
        sincos(phi, s_out, c_out);
        return;
    }
    sincos_cw(phi, s_out, c_out);
}

// A value that is the same in every lane (computed from the particle's globals), moved to
// scalar registers: frees VGPRs in the chunk loop (VALU ops take one SGPR operand each).
__device__ __forceinline__ double wave_uniform(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// Lanes of ONE wave handing data to each other through LDS.  The LDS executes a wave's instructions in issue order,
// so a read issued after a write of the same wave sees it -- no s_waitcnt is needed between them (the compiler waits
// by itself before a read's RESULT is used).  What must not happen is the COMPILER moving one across the other: to it
// they are accesses of one thread to different addresses.  Hence a compiler-only fence.  Round 3 had
// `s_waitcnt lgkmcnt(0)` here: four drained LDS round trips per chunk pair in the far-field expansions with nothing
// else for the wave to issue (-DNMRFIT_LDS_WAITS restores it for A/B runs).
__device__ __forceinline__ void wave_lds_fence()
{
#ifdef NMRFIT_LDS_WAITS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#endif
}

// Diagnostic builds (-DNMRFIT_DIAG_STAMPS): shader-clock stamps of wave 0 of every workgroup at the phases of a
// one-launch swarm generation, read back with nmrfit_diag_read_stamps (tools/generation_phases.py).
__device__ __forceinline__ void phase_stamp(unsigned long long *clk, int i)
{
#ifdef NMRFIT_DIAG_STAMPS
    if (clk && threadIdx.x == 0 && blockIdx.x < 1024) clk[4 + 16 * blockIdx.x + i] = __builtin_amdgcn_s_memtime();
#else
    (void)clk;
    (void)i;
#endif
}

// A wave-uniform pointer / value moved to VECTOR registers once, opaquely: what the swarm-generation prologue does with
// the ~20 pointers and constants of PsoFused.  Left to itself the compiler keeps all of them in scalar registers from the
// kernel's first instruction, runs out, and parks the grid-array pointers of the CHUNK LOOP in VGPR lanes instead -- a
// v_readlane per pointer per chunk (+1.9 % VALU instructions in every launch, swarm generation or not; measured).
template <class T>
__device__ __forceinline__ const T __attribute__((address_space(1))) *vector_ptr(const T *p)
{
    unsigned lo = (unsigned)(uintptr_t)p, hi = (unsigned)((uintptr_t)p >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return reinterpret_cast<const T __attribute__((address_space(1))) *>(((uintptr_t)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T __attribute__((address_space(1))) *vector_ptr_rw(const T *p)
{
    unsigned lo = (unsigned)(uintptr_t)p, hi = (unsigned)((uintptr_t)p >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return reinterpret_cast<T __attribute__((address_space(1))) *>(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ double vector_f64(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Address of a lane's 16-byte pair in a chunk: wave-uniform base + 16 * lane, with the lane part made opaque at the
// point of use -- otherwise the compiler hoists `array + lane` out of the chunk loop as a 64-bit per-lane pointer for
// each of the four arrays (8 VGPRs held across the loop, a v_lshl_add_u64 per array per chunk) instead of using the
// scalar-base + 32-bit-offset form of global_load.
__device__ __forceinline__ const double2 *lane_ptr(const double *uniform_base, int lane)
{
#if NMRFIT_SADDR
    unsigned off = (unsigned)lane * 16u;
    asm volatile("" : "+v"(off));
    return reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(uniform_base) + (size_t)off);
#else
    return reinterpret_cast<const double2 *>(uniform_base) + lane;
#endif
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
    return x;
}

// Dawson's integral D(x) = exp(-x^2) int_0^x exp(t^2) dt, |error| <= 4.1e-16 relative
// (piecewise polynomials generated by tools/gen_dawson.py).  The Hilbert transform of the
// Gaussian line exp(-x^2) is (2/sqrt(pi)) D(x): the closed form of what the reference's
// Kramers-Kronig quadrature computes point by point (nmrfit/equations.py:9-80).
__device__ __forceinline__ double dawson(double x)
{
    const double ax = fabs(x);
    double r;
    if (ax < 1.0) {
        const double y = x * x;
        double p = dawson::kNear[14];
#pragma unroll
        for (int i = 13; i >= 0; --i) p = __builtin_fma(p, y, dawson::kNear[i]);
        return x * p;
    } else if (ax < 7.0) {
        const int k = (int)ax;                 // 1..6
        const double t = 2.0 * (ax - (double)k) - 1.0;
        const double *q = dawson::kMid[k - 1];
        double p = q[18];
#pragma unroll
        for (int i = 17; i >= 0; --i) p = __builtin_fma(p, t, q[i]);
        r = p;
    } else {
        const double inv = rcp64(ax);         // NaN/inf propagate: D(inf) = 0
        const double s2 = 49.0 * inv * inv;
        double p = dawson::kFar[11];
#pragma unroll
        for (int i = 10; i >= 0; --i) p = __builtin_fma(p, s2, dawson::kFar[i]);
        r = 0.5 * p * inv;
    }
    return copysign(r, x);
}

constexpr double kSqrtLn2 = 0.83255461115769775635;     // sqrt(ln 2)
constexpr double kInvSqrtPi = 0.56418958354775628695;   // 1/sqrt(pi)

// Imaginary (dispersive) partner of one peak at one point: the Hilbert transform of
// a*(r*L + (1-r)*G) -- yoff drops out of the transform (equations.py:43-48: V2 - V1).
//   L -> AL * t/(1+t^2),   G -> (AG2/2) * (2/sqrt(pi)) * D(sqrt(ln2) t)
__device__ __forceinline__ double dispersion(double wcj, const PeakLor &r)
{
    const double t = __builtin_fma(wcj, r.ihw, r.c);
    const double s = __builtin_fma(t, t, 1.0);
    return __builtin_fma(r.al * t, rcp64(s), (r.ag2 * kInvSqrtPi) * dawson(kSqrtLn2 * t));
}

// Dawson's integral for the objective's imaginary channel: the same piecewise fits, but one
// degree-18 polynomial for EVERY unit interval [k, k+1), k = 0..15, gathered from a 2.4 KiB table
// in LDS by a per-lane index -- no divergent branches (the lanes of a wave sit in two or three
// different intervals), 19 FMAs + 19 broadcast-friendly LDS reads.  Beyond 16 the asymptotic form
// (a branch almost no wave takes: such peaks are summed through the far-field expansion).
constexpr int kDawTabFar = 16 * 19, kDawTabCount = 16 * 19 + 12;   // kTab[16][19], then kFar[12]
__device__ __forceinline__ double dawson_tab(double x, const double *tab)
{
    const double ax = fabs(x);
    const int k = (int)fmin(ax, 15.0);                 // NaN -> 15
    const double t = __builtin_fma(2.0, ax - (double)k, -1.0);
    const double *q = tab + k * 19;
    double p = q[18];
#pragma unroll
    for (int i = 17; i >= 0; --i) p = __builtin_fma(p, t, q[i]);
    if (!(ax < 16.0)) {
        const double inv = rcp64(ax);                  // NaN/inf propagate: D(inf) = 0
        const double s2 = 49.0 * inv * inv;
        double g = tab[kDawTabFar + 11];
#pragma unroll
        for (int i = 10; i >= 0; --i) g = __builtin_fma(g, s2, tab[kDawTabFar + i]);
        p = 0.5 * g * inv;
    }
    return copysign(p, x);
}

// dispersion() with the gathered Dawson table
__device__ __forceinline__ double dispersion_tab(double wcj, const PeakLor &r, const double *tab)
{
    const double t = __builtin_fma(wcj, r.ihw, r.c);
    const double s = __builtin_fma(t, t, 1.0);
    return __builtin_fma(r.al * t, rcp64(s), (r.ag2 * kInvSqrtPi) * dawson_tab(kSqrtLn2 * t, tab));
}

// The reference's fit_im=True compares the imaginary channel with the LAST peak's dispersion line only
// (equations.py:199 assigns I_fit instead of accumulating): that one line at the lane's points of a chunk, all
// points together.  In almost every chunk the peak is far away (|sqrt(ln2) t| >= 16 at every point of the wave: a
// wave-uniform test), where Dawson's integral is its asymptotic series -- 12 coefficients read ONCE per chunk,
// straight-line code over the eight points; otherwise the gathered table.  Round 3 evaluated point after point
// with a three-way branch whose Horner steps each waited for their own LDS read (a chain of ~18 LDS round trips
// per point at two or three waves per SIMD).
__device__ __forceinline__ void dispersion_points(const double (&wv)[kPointsPerLane], const PeakLor &r, const double *tab,
                                                  double (&out)[kPointsPerLane])
{
    double t[kPointsPerLane];
    bool far = true;
#pragma unroll
    for (int q = 0; q < kPointsPerLane; ++q) {
        t[q] = __builtin_fma(wv[q], r.ihw, r.c);
        far = far && (fabs(kSqrtLn2 * t[q]) >= 16.0);   // false for NaN
    }
    const double agd = r.ag2 * kInvSqrtPi;
    if (__ballot(!far) == 0ull) {
        double cfar[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) cfar[i] = tab[kDawTabFar + i];
#pragma unroll
        for (int q = 0; q < kPointsPerLane; ++q) {
            const double x = kSqrtLn2 * t[q];
            const double inv = rcp64(fabs(x));
            const double s2 = 49.0 * inv * inv;
            double g = cfar[11];
#pragma unroll
            for (int i = 10; i >= 0; --i) g = __builtin_fma(g, s2, cfar[i]);
            const double d = copysign(0.5 * g * inv, x);
            out[q] = __builtin_fma(r.al * t[q], rcp64(__builtin_fma(t[q], t[q], 1.0)), agd * d);
            // (scheduling fence: 2, 4 or 8 points in flight together time within 0.5 % of each other, and none brings the
            // far-field kernel under 168 VGPRs -- it is its scalar registers that run out)
            if ((q + 1) % NMRFIT_DISP_INTERLEAVE == 0) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < kPointsPerLane; ++q)
            out[q] = __builtin_fma(r.al * t[q], rcp64(__builtin_fma(t[q], t[q], 1.0)), agd * dawson_tab(kSqrtLn2 * t[q], tab));
    }
}

constexpr double binom_d(int n, int k)
{
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (double)(n - k + i) / (double)i;
    return r;
}
constexpr double pow49_half(int j)
{
    double r = 0.5;
    for (int i = 0; i < j; ++i) r *= 49.0;
    return r;
}
constexpr int kDawFarTerms = 12;        // terms of the asymptotic series of D kept in the far-field expansion (kFar)
constexpr double kDawFarX = 7.0;        // ... which is valid from |x| = 7 on

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
    constexpr int kInterleave = NMRFIT_INTERLEAVE;
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
    constexpr int kInterleave = NMRFIT_INTERLEAVE;
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
#if NMRFIT_PAIRFOLD
    if constexpr (B == 4) {
        // Four points per reciprocal, folded pair by pair: once two points' (numerator, denominator) are known they
        // become (n0 d1, n1 d0, d0 d1) -- three values instead of four held while the other pair is worked out (the
        // same 13 operations + one reciprocal per batch as the unfolded form below; values move by one rounding)
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
        return;
    }
#endif
#pragma unroll
    for (int q0 = 0; q0 < kPointsPerLane; q0 += B) {
        double num[B], den[B];
#pragma unroll
        for (int b = 0; b < B; ++b) point(wv[q0 + b], num[b], den[b]);
        if constexpr (B == 1) {
            acc[q0] = __builtin_fma(num[0], rcp64(den[0]), acc[q0]);
        } else if constexpr (B == 2) {
            const double r = rcp64(den[0] * den[1]);
            acc[q0] = __builtin_fma(num[0], r * den[1], acc[q0]);
            acc[q0 + 1] = __builtin_fma(num[1], r * den[0], acc[q0 + 1]);
        } else {
            static_assert(B == 4, "1, 2 or 4");
            const double p01 = den[0] * den[1], p23 = den[2] * den[3];
            const double r = rcp64(p01 * p23);
            const double r01 = r * p23, r23 = r * p01;
            acc[q0] = __builtin_fma(num[0], r01 * den[1], acc[q0]);
            acc[q0 + 1] = __builtin_fma(num[1], r01 * den[0], acc[q0 + 1]);
            acc[q0 + 2] = __builtin_fma(num[2], r23 * den[3], acc[q0 + 2]);
            acc[q0 + 3] = __builtin_fma(num[3], r23 * den[2], acc[q0 + 3]);
        }
        if ((q0 + B) % kInterleave == 0 || B > kInterleave) __builtin_amdgcn_sched_barrier(0);
    }
}

// A single peak in the scaled form (the odd one out of a short tail group): 1/s' per point, one
// reciprocal per four points.
__device__ __forceinline__ void lorentz_one_fast(const PeakFast *r, const double (&wv)[kPointsPerLane],
                                                 double (&acc)[kPointsPerLane])
{
    const double ih = r->ihs, c = r->cs, ia = r->ia;
    if constexpr (kPointsPerLane % 4 != 0) {   // NMRFIT_POINTS=2 (A/B builds): two points per reciprocal
#pragma unroll
        for (int q0 = 0; q0 < kPointsPerLane; q0 += 2) {
            const double t0 = __builtin_fma(wv[q0], ih, c), t1 = __builtin_fma(wv[q0 + 1], ih, c);
            const double s0 = __builtin_fma(t0, t0, ia), s1 = __builtin_fma(t1, t1, ia);
            const double rr = rcp64(s0 * s1);
            acc[q0] = __builtin_fma(rr, s1, acc[q0]);
            acc[q0 + 1] = __builtin_fma(rr, s0, acc[q0 + 1]);
        }
    } else
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
    if constexpr (kPointsPerLane % 4 != 0) {   // NMRFIT_POINTS=2 (A/B builds): two points per reciprocal
#pragma unroll
        for (int q0 = 0; q0 < kPointsPerLane; q0 += 2) {
            const double t0 = __builtin_fma(wv[q0], ih, c), t1 = __builtin_fma(wv[q0 + 1], ih, c);
            const double s0 = __builtin_fma(t0, t0, 1.0), s1 = __builtin_fma(t1, t1, 1.0);
            const double rr = al * rcp64(s0 * s1);
            acc[q0] = __builtin_fma(rr, s1, acc[q0]);
            acc[q0 + 1] = __builtin_fma(rr, s0, acc[q0 + 1]);
        }
    } else
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
__device__ __forceinline__ void gauss_add(const PeakLor *r, const double (&wv)[kPointsPerLane],
                                          double (&acc)[kPointsPerLane])
{
    const double ihw = r->ihw, c = r->c, ag2 = r->ag2;
#pragma unroll
    for (int q = 0; q < kPointsPerLane; ++q) {
        const double t = __builtin_fma(wv[q], ihw, c);
        const double s = __builtin_fma(t, t, 1.0);
        acc[q] = __builtin_fma(ag2, exp2_neg(-s), acc[q]);
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

// ---- the kernel ----------------------------------------------------------------------------
// VARIANT: NMRFIT_VARIANT_DEFAULT  8 Lorentzians per reciprocal + Gaussian window skip (+ on uniform
//                                  grids the Gaussian recurrence, objective launches only)
//          NMRFIT_VARIANT_NOREC    the same without the recurrence
//          NMRFIT_VARIANT_STAGED   the same + u/v/weights of each chunk prefetched into LDS by
//                                  global_load_lds (LDS-DMA) and w of the next chunk into
//                                  registers: hides the load latency when there are very few
//                                  peaks (P = 1: 0.54 -> 0.43 ms), neutral to -5 % otherwise
//          NMRFIT_VARIANT_BASELINE IEEE divide and libdevice exp2 per unit, no skip: the
//                                  obviously-right form the tuned ones are A/B-checked against
//          NMRFIT_VARIANT_NOSKIP   8 per reciprocal, Gaussian evaluated everywhere
//          NMRFIT_VARIANT_SINGLE   one reciprocal per unit + Gaussian window skip
//          NMRFIT_VARIANT_QUAD     4 per reciprocal + Gaussian window skip
//          NMRFIT_VARIANT_FARFIELD Lorentzian tails of distant peaks through a shared Taylor
//                                  expansion per chunk (opt-in; see the chunk loop)
// Wave g = blockIdx.x*4 + wave  ->  particle g / nseg, segment g % nseg;
// a segment is seg_len (multiple of 512) consecutive grid points.
// FIT_IM: 0 real part only (reference default); 1 reference-compatible fit_im=True -- the
// imaginary model is the LAST peak's dispersion only, because equations.py:199 assigns
// instead of accumulating; 2 the imaginary model is the sum over all peaks.
// The 8-peak group keeps 24 per-peak constants live next to the 8-point register block:
// ~152 VGPRs, i.e. 3 waves per SIMD (measured faster than 4 peaks per reciprocal at 4 waves).
// FIT_IM == 1 evaluates the last peak's dispersion line at the chunk's points in the epilogue
// (dispersion_points, Dawson coefficients from LDS): the direct kernels keep three waves per SIMD, the
// far-field one takes two rather than spilling; FIT_IM == 2 holds eight more accumulators and the
// far-field sums: two waves.
#ifndef NMRFIT_FARFIELD_TRUE_WAVES
#define NMRFIT_FARFIELD_TRUE_WAVES 2
#endif
#ifndef NMRFIT_FARFIELD_IM_WAVES
#define NMRFIT_FARFIELD_IM_WAVES 2
#endif
#ifndef NMRFIT_SUM_WAVES
#define NMRFIT_SUM_WAVES 2     // launch bound of the direct kernels with the imaginary sum.  Round 4: they fit in 168 VGPRs,
                               // i.e. run at THREE waves per SIMD (3.5 -> 2.57 ms at C3) -- with the bound left at two: asked
                               // for three the compiler stops at 160 registers and schedules worse (2.86 ms, measured)
#endif
#define NMRFIT_OBJECTIVE_MIN_WAVES(VARIANT, FIT_IM)                                                                    \
    (((FIT_IM) == 1 && (VARIANT) == NMRFIT_VARIANT_FARFIELD) ? NMRFIT_FARFIELD_TRUE_WAVES                             \
     : ((FIT_IM) == 2 && (VARIANT) == NMRFIT_VARIANT_FARFIELD) ? NMRFIT_FARFIELD_IM_WAVES                             \
     : ((FIT_IM) == 2) ? NMRFIT_SUM_WAVES                                                                             \
                   : ((VARIANT) == NMRFIT_VARIANT_DEFAULT || (VARIANT) == NMRFIT_VARIANT_NOSKIP ||                     \
                      (VARIANT) == NMRFIT_VARIANT_STAGED || (VARIANT) == NMRFIT_VARIANT_FARFIELD ||                    \
                      (VARIANT) == NMRFIT_VARIANT_NOREC)                                                               \
                         ? NMRFIT_MIN_WAVES                                                                            \
                         : 4)
template <int VARIANT, bool WRITE_R, int FIT_IM, int WPB>
__device__ __forceinline__ void objective_body(
    unsigned char *lds_raw, const int64_t g,
    const double *__restrict__ wc, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ wt, const double2 *__restrict__ chunk_minmax,
    const double *__restrict__ X, int64_t S, int P, int64_t N, double w0, double wspan, int nseg,
    int64_t seg_len, int blk_chunks, double lane_step, double rec_devk,
    double *__restrict__ out,       // nseg == 1: f[S];  else per-block sums [S * n_blocks] (x2 with FIT_IM)
    double *__restrict__ R_out,     // WRITE_R: residual rows [S*N]
    unsigned long long *__restrict__ clk,   // profiling only (else null): shader / reference clock of workgroup 0
    const PsoFused &upd,            // swarm generations: advance the particle first (x_in != null), X is then unused
    const unsigned aux_off,         // FIT_IM != 0: byte offset of the Dawson table in dynamic LDS
    double *wsums)                  // [2 * kMaxBlocks] in LDS: the particle's block sums when one workgroup owns it
{
    // WPB: waves per workgroup = LDS slices.  Four (one per SIMD of a CU) everywhere except for particles cut into
    // EIGHT segments (small swarms on short grids: the reference's default 204 particles x 4096 points), where an
    // eight-wave workgroup holds the whole particle: one prologue, f and the personal best finished in this launch.
    const int lane = threadIdx.x & (kWave - 1);
    // (the wave index through v_readfirstlane: the compiler then KNOWS that everything derived from it -- particle,
    // segment, chunk bases, the chunk table's address -- is wave-uniform, keeps it in scalar registers and fetches the
    // chunk table with scalar loads instead of a vector load on the critical path at the top of every chunk)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // When every wave of the workgroup evaluates a segment of the SAME particle (nseg a multiple of
    // the waves per workgroup) the particle's prologue is done once per workgroup instead of once
    // per wave: one copy of the per-peak records (slice 0), the position update by wave 0, the
    // per-peak constants by the waves in turn (64 peaks a pass), the phase seeds by the last wave --
    // and a workgroup barrier.  For a short grid the prologue is as long as a chunk or two, so this
    // is what makes four or eight segments per particle affordable (C2: 17.7 -> see DESIGN.md).
#ifdef NMRFIT_DIAG_REMAP   // diagnostic builds only (tools/ab.py): the waves of a workgroup = the SAME segment of four particles
    const bool shared = false;
#else
    const bool shared = (nseg % WPB == 0);
#endif
    const int slice = shared ? 0 : wave;
    // (one copy of every per-peak record per workgroup when its waves share a particle, else one per wave: the
    // dynamic LDS is sized accordingly by resolve_variant -- at C3 that is what lets a fourth workgroup onto a CU)
    const int nslices = shared ? 1 : WPB;
    PeakLor *lor = reinterpret_cast<PeakLor *>(lds_raw) + (size_t)slice * P;
    PeakWin *win = reinterpret_cast<PeakWin *>(lds_raw + (size_t)nslices * P * sizeof(PeakLor)) +
                   (size_t)slice * P;
    constexpr bool kStage = (VARIANT == NMRFIT_VARIANT_STAGED);
    // STAGED (and, as an A/B knob, -DNMRFIT_PREFETCH_W=1 for the selectable kernels): w of the NEXT chunk is requested
    // in the epilogue of the current one, the first chunk's before the prologue's barrier.  Round 4 measured it on
    // its own: slower everywhere (+1 % at C3, +5 % on the reference's default swarm)
    // (not with the imaginary sum: its 16 registers are what keeps that kernel at three waves per SIMD)
    constexpr bool kPrefW = kStage || (NMRFIT_PREFETCH_W != 0 && FIT_IM != 2 &&
                                       (VARIANT == NMRFIT_VARIANT_DEFAULT || VARIANT == NMRFIT_VARIANT_FARFIELD ||
                                        VARIANT == NMRFIT_VARIANT_NOREC));
    // per-wave table of block seeds (<= 16 blocks per grid); the shared-prologue area (rotation step,
    // per-lane phase seeds, flags); then (kStage) the per-wave staging area for one chunk of u, v,
    // weights (3 x 512 doubles = 12 KiB)
    unsigned char *lds_tail = lds_raw + (((size_t)nslices * P * (sizeof(PeakLor) + sizeof(PeakWin)) + 15) & ~(size_t)15);
    double2 *seeds = reinterpret_cast<double2 *>(lds_tail) + (size_t)wave * kMaxBlocks;
    double *shr = reinterpret_cast<double *>(lds_tail + (size_t)WPB * kMaxBlocks * sizeof(double2));   // rho, L_lane[64] re / im
    int *sflag = reinterpret_cast<int *>(shr + 2 + 2 * kWave);                                          // one per wave
    // per-lane phase seeds L_lane: in `shr` when the workgroup is one particle, else one copy per wave right behind
    // it; read back at the start of every block instead of living in four VGPRs across the chunk loop
    double *lseed = shared ? shr + 2
                           : reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(shr) + kSharedPrologueBytes) + (size_t)wave * (2 * kWave);
    unsigned char *lds_tail2 = reinterpret_cast<unsigned char *>(shr) + kSharedPrologueBytes +
                               (shared ? 0 : (size_t)WPB * 2 * kWave * sizeof(double));
    double *stage = reinterpret_cast<double *>(lds_tail2) + (size_t)wave * (3 * kChunk);
    // FARFIELD: per-wave scratch [kFarTerms][kFarPad] for the cross-peak coefficient sums
    // (shares the offset of `stage`; the two variants are exclusive)
    double *ffs = reinterpret_cast<double *>(lds_tail2) + (size_t)wave * (kFarTerms * kFarPad);

    // objective launches of DEFAULT / FARFIELD: per-peak (d, C) of the Gaussian recurrence, after
    // everything else (residual rows are evaluated point by point: they feed finite differences)
    constexpr bool kRec = !WRITE_R && (VARIANT == NMRFIT_VARIANT_DEFAULT || VARIANT == NMRFIT_VARIANT_FARFIELD);
    unsigned char *grec_base = lds_tail2 +
                        (kStage ? (size_t)WPB * 3 * kChunk * sizeof(double)
                                : (VARIANT == NMRFIT_VARIANT_FARFIELD || FIT_IM == 2) ? (size_t)WPB * kFarTerms * kFarPad * sizeof(double) : 0);
    double2 *grec = reinterpret_cast<double2 *>(grec_base) + (size_t)slice * P;

    // DEFAULT: scaled Lorentzian constants for the two-operation pair form, after grec
    constexpr bool kFast = (NMRFIT_FASTPAIR != 0) && (VARIANT == NMRFIT_VARIANT_DEFAULT) && (NMRFIT_GROUP == 8);
    PeakFast *lorf = reinterpret_cast<PeakFast *>(grec_base + (kRec ? (size_t)nslices * P * sizeof(double2) : 0)) +
                     (size_t)slice * P;

    // FIT_IM != 0: Dawson table (16 intervals x 19 coefficients for the gathered evaluation, then the 12 of the
    // asymptotic series), one copy per workgroup; the barrier after the staging below makes it visible
    double *dtab = reinterpret_cast<double *>(lds_raw + aux_off);
    if constexpr (FIT_IM != 0)   // kTab[16][19], then kFar[12]
        for (int i = threadIdx.x; i < kDawTabCount; i += WPB * kWave)
            dtab[i] = (i < kDawTabFar) ? (&dawson::kTab[0][0])[i] : dawson::kFar[i - kDawTabFar];
#if defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD == 3
    if constexpr (FIT_IM == 0)   // (the region the diagnostic loads read: zeros; a barrier follows the staging below)
        for (int i = threadIdx.x; i < 4 * kChunk; i += WPB * kWave) reinterpret_cast<double *>(lds_raw + aux_off)[i] = 0.0;
#endif
    phase_stamp(clk, 0);
    if (clk && g == 0 && lane == 0) {   // nmrfit_prof_*: ticks of the core clock and of the 100 MHz reference
        clk[0] = __builtin_amdgcn_s_memtime();
        clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    const bool active = g < S * nseg;
    const int64_t particle = active ? g / nseg : 0;
    const int seg = active ? (int)(g % nseg) : 0;
    const int64_t D = 4 + 3 * (int64_t)P;
    double wnext[kPointsPerLane];
    if (kPrefW && active) {   // the first chunk's w: on its way while the prologue runs
        const int64_t ja = (int64_t)seg * seg_len;
        const int64_t je = (ja + seg_len < N) ? ja + seg_len : N;
        if (ja + kChunk <= je) {
            const double2 *wp = reinterpret_cast<const double2 *>(wc + ja) + lane;
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const double2 d = wp[m * kWave];
                wnext[2 * m] = d.x;
                wnext[2 * m + 1] = d.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q)
                wnext[q] = (ja + lane + q * kWave < je) ? wc[ja + (q >> 1) * (2 * kWave) + 2 * lane + (q & 1)] : 0.0;
        }
    }
    double p0, p1, r, yoff;
    // stage this particle's per-peak constants in the wave's LDS slices (x: the particle's row,
    // in global memory or -- fused swarm update -- in this wave's LDS copy)
    bool fast_bad = false, rec_bad = false;
    auto stage_peaks = [&](const double *x, const int first_pass, const int pass_stride) {
    p0 = x[0], p1 = x[1], r = x[2], yoff = x[3];   // equations.py:177
    for (int kb0 = first_pass * kWave; kb0 < P; kb0 += pass_stride * kWave) {   // every lane iterates (the group sums below shuffle)
        const int k = kb0 + lane;
        const bool have = k < P;
        const int kx = have ? k : 0;
        const double width = x[4 + 3 * kx], loc = x[5 + 3 * kx], a = x[6 + 3 * kx];
        const double ihw = 2.0 / width;
        const double locc = loc - w0;
        // |t| <= 1e18 keeps the grouped denominators finite; the cap only engages for widths
        // below 2e-18 of the spectral span, where L and G are 0 to 1e-36 either way
        const double lim = 1.0e18 / (wspan + fabs(locc));
        const double it = (fabs(ihw) > lim) ? copysign(lim, ihw) : ihw;
        PeakLor rec;
        rec.ihw = it;
        rec.c = -locc * it;
        rec.al = a * r * ihw * kInvPi;                            // a*r*(2/(pi*width))
        rec.ag2 = 2.0 * a * (1.0 - r) * ihw * kSqrtLn2OverPi;     // 2 * a*(1-r)*(2/width)*sqrt(ln2/pi)
        if (have) lor[k] = rec;
        // window bounds in f32, rounded outwards (a slightly wider window is still exact)
        const double gw = kGaussWindow * fabs(width);
        const double wlo = locc - gw, whi = locc + gw;
        if (have) win[k] = PeakWin{(float)(wlo - fabs(wlo) * 1.2e-7 - 1e-37), (float)(whi + fabs(whi) * 1.2e-7 + 1e-37)};
        if (kFast) {
            // exponent budget of the group's denominator: s' <= (1 + tmax^2)/al, s' >= 1/al
            const double al = rec.al;
            const double tmax = fabs(it) * (wspan + fabs(locc));
            const bool pos = al > 0.0 && al < 1.0e300;                  // false for NaN
            const double ia = pos ? 1.0 / al : 1.0;
            const double rs = pos ? sqrt(ia) : 1.0;
            int ehi = pos ? ilogb(__builtin_fma(tmax, tmax, 1.0) * ia) + 2 : 100000;
            int elo = pos ? ilogb(ia) : -100000;
            if (ehi > 100000) ehi = 100000;                              // inf / overflow
            if (!have) ehi = elo = 0;                                    // beyond the last peak: no factor
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                ehi += __shfl_xor(ehi, m, kWave);
                elo += __shfl_xor(elo, m, kWave);
            }
            // every group of (up to) 8 peaks must have positive amplitudes and a denominator that
            // stays within 2^+-1000 for kBatchInv points; ONE flag per particle -- all groups
            // qualify or none does -- keeps the chunk loop free of a per-group branch (whose two
            // arms cost 16 register copies per group in phi moves: measured, it ate the gain)
            if (have && !(ehi < 1000 / kBatchInv && elo > -1000 / kBatchInv)) fast_bad = true;
            // cs from the ROUNDED ihs (one rounding, like c from ihw): the zero of t' then sits at
            // loc to the same accuracy as the zero of t
            const double ihs = it * rs;
            if (have) lorf[k] = PeakFast{ihs, -locc * ihs, ia, 0.0};
        }
        if (kRec) {
            const double d = lane_step * it;
            const bool ok = (lane_step != 0.0) && (fabs(d) <= 2.0) && (fabs(it) * rec_devk <= 1.0);
            if (have && !ok) rec_bad = true;
            if (have) grec[k] = make_double2(d, ok ? exp2_neg(-2.0 * d * d) : 0.0);
        }
    }
    };
    bool fused = false;
    if constexpr (!WRITE_R) fused = upd.x_in != nullptr;
    // (the row is staged from global memory or from LDS by two separate calls: one pointer that may
    // be either makes this compiler's address-space inference crash, and would cost flat loads)
    double *const xrow = reinterpret_cast<double *>(lds_raw + upd.xrow_off) + (size_t)slice * D;
    auto stage_row = [&](const int first_pass, const int pass_stride) {
        if (fused)
            stage_peaks(xrow, first_pass, pass_stride);
        else
            stage_peaks(X + particle * D, first_pass, pass_stride);
    };
    if (fused) {
        // (see vector_ptr: the swarm's pointers and constants live in vector registers for the length of this block)
        const auto gx_in = vector_ptr(upd.x_in), gv_in = vector_ptr(upd.v_in), gp = vector_ptr(upd.p);
        const auto gbest = vector_ptr(upd.best), glb = vector_ptr(upd.lb), gub = vector_ptr(upd.ub);
        const auto gflags = vector_ptr(upd.flags);
        const auto gx_out = vector_ptr_rw(upd.x_out), gv_out = vector_ptr_rw(upd.v_out), gp_rw = vector_ptr_rw(upd.p);
        const auto gbest_rw = vector_ptr_rw(upd.best), gcand = vector_ptr_rw(upd.cand);
        const auto gflags_rw = vector_ptr_rw(upd.flags);
        const double q_omega = vector_f64(upd.omega), q_phip = vector_f64(upd.phip), q_phig = vector_f64(upd.phig);
        const double q_minstep = vector_f64(upd.minstep), q_minfunc = vector_f64(upd.minfunc);
        // Swarm generation: the velocity / position update of this particle happens HERE, in the
        // prologue of the kernel that evaluates it (one launch fewer per generation).  Every wave of
        // the particle computes the same new row into its own LDS slice; the wave of segment 0 also
        // writes it (and the velocity) to the swarm's other state buffer -- never the one being
        // read, so the segments of a particle cannot race.  After a stop every launch is a no-op:
        // the row is carried over unchanged and the kernel returns.
        const bool deferred = upd.tail != 0u;   // (the host asks for it only when the workgroup is the particle: `shared`)
        const bool updater = !shared || wave == 0;   // the wave that moves the particle
        bool stopped = false;
        long long gen_done = 0, stop_code = 0;
        double *const grow = xrow + D, *const crow = xrow + 2 * D;   // (tail != 0: rows 1 and 2 of the row area)
        // Everything whose address is known goes out NOW, in one round trip: the flags and, into registers, the first 64
        // entries (all of them up to 20 peaks) of the particle's state, of the bounds and of g.
        const bool have0 = updater && lane < D;
        const int64_t idx0 = particle * D + lane;
        double x0 = 0.0, v0 = 0.0, pold0 = 0.0, lo0 = 0.0, hi0 = 0.0, g0 = 0.0, fg = 0.0;
        if (updater || !deferred) {   // (deferred form: wave 0 tells the workgroup what the fold said, through LDS)
            gen_done = gflags[0];
            stop_code = gflags[1];
        }
        if (updater) {
            if (deferred) fg = gbest[0];
            if (have0) {
                x0 = gx_in[idx0];
                v0 = gv_in[idx0];
                pold0 = gp[idx0];
                lo0 = glb[lane];
                hi0 = gub[lane];
                g0 = gbest[2 + lane];
            }
        }
        double rp0 = 0.0, rg0 = 0.0;   // the first entry's uniforms (deferred form: drawn while the winner's row is on its way)
        bool drawn0 = false;
        if (deferred) {
            // ---- deferred fold (PsoFused): the previous launch left the personal bests of its generation; before this
            // particle moves, its workgroup works out what the swarm's best is NOW -- as every other workgroup does,
            // from the same memory with the same operations (pso_update.h apply_wave, pso.hip argmin_block).
            if (upd.pending != 0u) {   // every wave: first index of the minimum over its share of fp
                const auto fpb = gp + S * D;
                const int kper = (int)((S + WPB * kWave - 1) / (WPB * kWave));   // <= kDeferredPerLane (launch_objective)
                const int64_t base = (int64_t)wave * kper * kWave + lane;
                double vv[kDeferredPerLane];
#pragma unroll
                for (int k = 0; k < kDeferredPerLane; ++k) {   // all loads of a lane in flight together
                    const int64_t i = base + (int64_t)k * kWave;
                    vv[k] = (k < kper && i < S) ? fpb[i] : INFINITY;
                }
                double best = INFINITY;
                long long bi = 0x7fffffffffffffffLL;
#pragma unroll
                for (int k = 0; k < kDeferredPerLane; ++k)
                    if (vv[k] < best) {
                        best = vv[k];
                        bi = base + (long long)k * kWave;
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ob = __shfl_down(best, off, kWave);
                    const long long oi = __shfl_down(bi, off, kWave);
                    if (lex_less(ob, oi, best, bi)) {
                        best = ob;
                        bi = oi;
                    }
                }
                if (lane == 0) {   // (the block-sum slots are free until the chunk loop ends)
                    wsums[wave] = best;
                    wsums[kMaxBlocks + wave] = __longlong_as_double(bi);
                }
            }
            __syncthreads();
            phase_stamp(clk, 10);   // every wave's share of the argmin over fp is in LDS
            if (wave == 0) {
                if (upd.pending != 0u) {
                    double fc = wsums[0];
                    long long bi = __double_as_longlong(wsums[kMaxBlocks]);
#pragma unroll
                    for (int w2 = 1; w2 < WPB; ++w2) {
                        const double ob = wsums[w2];
                        const long long oi = __double_as_longlong(wsums[kMaxBlocks + w2]);
                        if (lex_less(ob, oi, fc, bi)) {
                            fc = ob;
                            bi = oi;
                        }
                    }
                    if (bi >= S) bi = 0;   // every fp is +inf: np.argmin -> 0, and the row is x[0] (pso.hip, argmin_block)
                    const auto src = (fc < INFINITY) ? gp + bi * D : gx_in;
                    const double c0 = have0 ? src[lane] : 0.0;   // the second (and last) round trip of the prologue
                    if (stop_code == 0) gen_done += 1;   // (after a stop nothing folds and nothing counts: pso_apply_kernel)
                    if (have0 && stop_code == 0) {       // meanwhile: this generation's uniforms of entry `lane`
                        uniform2(upd.seed, (uint32_t)(gen_done + 1), (uint32_t)lane, (uint64_t)(upd.offset + particle), &rp0, &rg0);
                        drawn0 = true;
                    }
                    int code = 0;   // 0: not better, 1: stop minfunc, 2: stop minstep, 3: accept
                    {
#pragma clang fp contract(off)
                        double acc = 0.0;
                        if (have0) {
                            crow[lane] = c0;
                            grow[lane] = g0;
                            const double df = g0 - c0;
                            acc += df * df;
                        }
                        for (int64_t d = lane + kWave; d < D; d += kWave) {
                            const double c = src[d], gd = gbest[2 + d];
                            crow[d] = c;
                            grow[d] = gd;
                            const double df = gd - c;
                            acc += df * df;
                        }
                        if (stop_code == 0 && fc < fg) {
                            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                            acc = __shfl(acc, 0, 64);
                            const double stepsize = sqrt(acc);
                            if (fabs(fg - fc) <= q_minfunc)
                                code = 1;
                            else if (stepsize <= q_minstep)
                                code = 2;
                            else
                                code = 3;
                        }
                        if (code == 1 || code == 2) stop_code = code;
                    }
                    wave_lds_fence();
                    if (particle == 0) {   // ONE writer of the other state block (nobody reads it in this launch)
                        auto bo = gbest_rw + upd.flip;
                        auto fo = gflags_rw + upd.flip;
                        for (int64_t d = lane; d < D; d += kWave) {
                            const double c = crow[d];
                            bo[2 + d] = (code == 3) ? c : grow[d];
                            bo[2 + D + d] = (code != 0) ? c : gbest[2 + D + d];
                            gcand[1 + d] = c;
                        }
                        if (lane == 0) {
                            bo[0] = (code == 3) ? fc : fg;
                            bo[1] = (code != 0) ? fc : gbest[1];
                            fo[0] = gen_done;
                            fo[1] = stop_code;
                            gcand[0] = fc;
                        }
                    }
                    if (code == 3) {
                        g0 = c0;
                        for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = crow[d];
                    }
                    wave_lds_fence();
                } else {
                    for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = gbest[2 + d];
                    wave_lds_fence();
                }
                if (lane == 0) wsums[2 * kMaxBlocks + 6] = (stop_code != 0) ? 1.0 : 0.0;
                phase_stamp(clk, 11);   // folded
            }
        }
        stopped = stop_code != 0;
        const uint32_t gen = (uint32_t)(gen_done + 1);

        // every entry of the row: draw, move, clip (pso_update.h); the new row goes to LDS (what this launch evaluates)
        // and, by the wave of segment 0, with the velocity to the swarm's other state buffer.  The first entry of a
        // lane comes from the registers loaded above.
        if (updater)
        for (int64_t d = lane; d < D; d += kWave) {
            const int64_t idx = particle * D + d;
            const bool first = d < kWave;
            double xn = first ? x0 : gx_in[idx], vn = first ? v0 : gv_in[idx];
            const double pold = first ? pold0 : gp[idx];
            // (two loads and a select of VALUES: a select between an LDS and a global address crashes this compiler)
            double g_lds = 0.0, g_mem = 0.0;
            if (!first && deferred) g_lds = grow[d];
            if (!first && !deferred) g_mem = gbest[2 + d];
            const double gd = first ? g0 : deferred ? g_lds : g_mem;
            const double lo = first ? lo0 : glb[d], hi = first ? hi0 : gub[d];
            if (deferred) {   // the personal best as it stands: for the kernel's end (row 2 is free again), or carried over now
                crow[d] = pold;
                if (stopped) gp_rw[upd.pflip + idx] = pold;
            }
            if (!stopped) {
                double rp = rp0, rg = rg0;
                if (!(first && drawn0)) uniform2(upd.seed, gen, (uint32_t)d, (uint64_t)(upd.offset + particle), &rp, &rg);
                xn = update_value(xn, vn, pold, gd, lo, hi, rp, rg, q_omega, q_phip, q_phig, &vn);
            }
            xrow[d] = xn;
            if (active && seg == 0) {
                gx_out[idx] = xn;
                gv_out[idx] = vn;
            }
        }
        if (deferred && stopped && wave == 0 && lane == 0)   // (after a stop: the value is carried over like the rows)
            gp_rw[upd.pflip + S * D + particle] = gp[S * D + particle];
        if (!deferred && stopped) return;   // the same for every wave of the grid
        phase_stamp(clk, 1);   // position update done
        if (shared) __syncthreads();   // wave 0's row is every wave's input
        if (deferred && wsums[2 * kMaxBlocks + 6] != 0.0) return;   // (wave 0 told the workgroup: the same in every workgroup of the grid)
        wave_lds_fence();   // same-wave LDS write -> read
    }
    double rr = 1.0, ri = 0.0;   // rotation step exp(i p1 64/N) (the lane seeds exp(i (p0 + p1 lane/N)): lseed, in LDS)
    const double invN = 1.0 / (double)N;
    bool fast_all, rec_all;
    if (shared) {
        stage_row(wave, WPB);                          // wave w: peaks 64w..64w+63, 64(w+WPB).., usually wave 0 alone
        if (wave == WPB - 1) {                         // meanwhile the last wave makes the phase seeds
            double sr, si, tr, ti;
            sincos_fast((p1 * 64.0) * invN, &si, &sr);
            sincos_fast(p0 + (p1 * (double)lane) * invN, &ti, &tr);
            if (lane == 0) {
                shr[0] = sr;
                shr[1] = si;
            }
            shr[2 + lane] = tr;
            shr[2 + kWave + lane] = ti;
        }
        const int fl = (__ballot(fast_bad) != 0ull ? 1 : 0) | (__ballot(rec_bad) != 0ull ? 2 : 0);
        if (lane == 0) sflag[wave] = fl;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int w2 = 0; w2 < WPB; ++w2) all |= sflag[w2];
        fast_all = kFast && !(all & 1);
        rec_all = kRec && !(all & 2);
        rr = wave_uniform(shr[0]);
        ri = wave_uniform(shr[1]);
    } else {
        stage_row(0, 1);
        // wave-uniform: every group of this particle may take the two-operation pair form
        fast_all = kFast && (__ballot(fast_bad) == 0ull);
        rec_all = kRec && (__ballot(rec_bad) == 0ull);   // every peak may take the Gaussian recurrence
        __syncthreads();
    }
    if (!active) return;
    phase_stamp(clk, 2);   // per-peak constants and phase seeds staged

    const int64_t j0 = (int64_t)seg * seg_len;
    const int64_t j1 = (j0 + seg_len < N) ? j0 + seg_len : N;
    const int64_t n_chunks = (N + kChunk - 1) / kChunk;

    // phase ramp: z = exp(i*phi_j) for this lane's current point, rho = exp(i*p1*64/N)
    // z is re-seeded at the start of every block of blk_chunks chunks as E_b * L_lane with
    // E_b = exp(i*p1*(b*blk_len)/N) (wave-uniform, tabulated in LDS for this segment's blocks)
    // and L_lane = exp(i*(p0 + p1*lane/N)): both depend on the GLOBAL block index and the lane
    // only, never on where the segment starts.
    double zr = 1.0, zi = 0.0;
    const int64_t blk_len = (int64_t)blk_chunks * kChunk;
    if (!shared) {   // (shared prologue: made once per workgroup above)
        sincos_fast((p1 * 64.0) * invN, &ri, &rr);
        rr = wave_uniform(rr);
        ri = wave_uniform(ri);
        double lr, li;
        sincos_fast(p0 + (p1 * (double)lane) * invN, &li, &lr);
        lseed[lane] = lr;
        lseed[kWave + lane] = li;
    }
    {
        const int64_t b0 = j0 / blk_len;
        const int64_t nb = (j1 - j0 + blk_len - 1) / blk_len;
        if (lane < nb) {
            double er, ei;
            sincos_fast((p1 * (double)((b0 + lane) * blk_len)) * invN, &ei, &er);
            seeds[lane] = make_double2(er, ei);
        }
        wave_lds_fence();   // same-wave LDS write -> read
    }
    const int64_t n_blocks = (n_chunks + blk_chunks - 1) / blk_chunks;
    double bs = 0.0, bs_im = 0.0;             // per-lane sums of squares of the current block
    const int64_t blk0 = j0 / blk_len;         // global index of this segment's first block
    int cib = 0, bidx = 0;                     // chunk within block, block within segment
    const double base = wave_uniform((double)P * yoff);   // yoff is added once per peak (equations.py:147,195)
    double ss = 0.0, ss_im = 0.0;
    constexpr bool kSkip = (VARIANT != NMRFIT_VARIANT_NOSKIP && VARIANT != NMRFIT_VARIANT_BASELINE);
    constexpr bool kFar = (VARIANT == NMRFIT_VARIANT_FARFIELD);
    constexpr int kGroup = (VARIANT == NMRFIT_VARIANT_SINGLE) ? 1 : (VARIANT == NMRFIT_VARIANT_QUAD) ? 4 : NMRFIT_GROUP;

    unsigned even_near = 0, even_hits = 0;     // FARFIELD, P <= 32: near-peak and Gaussian-window masks of the even ...
    unsigned pend_near = 0, pend_hits = 0;     // ... and of the odd chunk of the current pair (expand_pair)

    // The chunk loop exists twice, once per Lorentzian group form, chosen ONCE per wave: inside one
    // copy the accumulators never meet the other form's registers (a merge of the two forms per
    // chunk costs the compiler 8-16 register copies per chunk).
    auto chunk_loop = [&](auto fast_tag) {
    constexpr bool kFastLoop = decltype(fast_tag)::value;
    // ... and the chunk body twice more, for full chunks and for the one ragged chunk at the end of
    // the grid: `full` is a compile-time constant inside, so the predicated and the unpredicated
    // loads never merge (each merge is eight register copies).
    // FARFIELD, P <= 32: the far-field expansions of a PAIR of chunks -- lanes 0..31 those of chunk jbE, lanes 32..63
    // those of the chunk after it -- summed over peaks into slots 0..15 / 16..31 of the wave's scratch, with the
    // near-peak and Gaussian-window masks of the two chunks in scalar registers.  Either half runs the same
    // operations in the same order, so a chunk's coefficients do not depend on which half made them, nor on when.
    auto expand_pair = [&](const int64_t jbE) {
        const double2 mm = chunk_minmax[jbE / kChunk];
        const bool has_next = jbE + kChunk < j1;                  // wave-uniform
        double2 mn = mm;
        if (has_next) mn = chunk_minmax[jbE / kChunk + 1];
        const bool upper = lane >= 32;
        const int k = lane & 31;
        const double lo_w = upper ? mn.x : mm.x, hi_w = upper ? mn.y : mm.y;
        const bool act = (k < P) && (!upper || has_next);
        bool far = false, ghit = false;
        double a2 = 0.0, b2 = 0.0, y0 = 0.0, y1 = 0.0;
        if (act) {
            const PeakLor rec = lor[k];
            const PeakWin wn = win[k];
            ghit = (hi_w >= (double)wn.lo) && (lo_w <= (double)wn.hi);
            const double tc = __builtin_fma(0.5 * (lo_w + hi_w), rec.ihw, rec.c);
            const double hk = (0.5 * (hi_w - lo_w)) * rec.ihw;
            const double den = __builtin_fma(tc, tc, 1.0);
            far = den >= 100.0 * hk * hk;            // rho^2 <= 0.01 (false for NaN)
            if (far) {
                const double rq = rcp64(den);
                const double qr = tc * rq;            // q = (tc + i)/(tc^2 + 1)
                const double mr = -hk * qr, mi = -hk * rq;   // m = -hk q
                a2 = mr + mr;
                b2 = -__builtin_fma(mr, mr, mi * mi);
                y0 = rec.al * rq;
                y1 = rec.al * __builtin_fma(qr, mi, rq * mr);
            }
        }
        const unsigned long long nearmask = __ballot(act && !far);
        const unsigned long long hits = __ballot(ghit);
        // order n carries al * Im(q m^n); both roots of the real recurrence y[n+1] = 2 Re(m) y[n] - |m|^2 y[n-1]
        // have modulus |m| (stable), two operations a term; lanes without a far peak carry exact zeros (no branch
        // on "any far peak at all": a pair of chunks without one is the rare case, and the branch would cut the
        // straight-line code the scheduler interleaves with the chunk's other work)
        double *dst = ffs + lane + (lane >> 4);
#pragma unroll
        for (int n = 0; n < kFarTerms; ++n) {
            dst[n * kFarPad] = y0;
            const double y2 = __builtin_fma(a2, y1, b2 * y0);
            y0 = y1;
            y1 = y2;
        }
        even_near = (unsigned)nearmask;
        even_hits = (unsigned)hits;
        pend_near = (unsigned)(nearmask >> 32);
        pend_hits = (unsigned)(hits >> 32);
        wave_lds_fence();   // same-wave LDS write -> read (expand_sums)
    };
    // ... second half: lane l sums order l>>2 over 16 peaks (quarter rows padded to 17 doubles), lanes l, l^1 hold
    // the halves of one chunk; sums of the pair's first chunk -> slots 0..15, of its second -> slots 16..31
    auto expand_sums = [&]() {
        const double *row = ffs + (lane >> 2) * kFarPad + (lane & 3) * 17;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) part += row[j];
        part += __shfl_xor(part, 1, kWave);
        wave_lds_fence();   // reads issued before the sums overwrite row 0
        if ((lane & 1) == 0) ffs[((lane & 2) << 3) + (lane >> 2)] = part;
        wave_lds_fence();
    };
    auto chunk = [&](const int64_t jb, auto full_tag, auto odd_tag) {
        // FARFIELD, P <= 32: an odd chunk's expansion was made by the even chunk before it
        constexpr bool ff_odd = decltype(odd_tag)::value;
        // Full chunks (all but possibly the last of a segment) take unpredicated loads at
        // constant offsets from one pointer; the ragged tail is predicated per point.
        constexpr bool full = decltype(full_tag)::value;
        const int64_t jl = jb + lane;
        // A block = blk_chunks consecutive chunks, a function of N only; segments are whole
        // blocks.  Everything that carries state from point to point restarts at block
        // boundaries -- the phase recurrence is re-seeded here, the sums of squares are reduced
        // at the block's end -- so every value, and hence f, is bit-identical for any
        // segmentation of the grid and any sharding of the swarm.
        if (cib == 0) {   // first chunk of a block (segments start on block boundaries)
            const double2 e = seeds[bidx];
            const double lr = lseed[lane], li = lseed[kWave + lane];
            zr = __builtin_fma(e.x, lr, -(e.y * li));
            zi = __builtin_fma(e.x, li, e.y * lr);
        }
        double wv[kPointsPerLane], acc[kPointsPerLane];
        double uq[kPointsPerLane], vq[kPointsPerLane], tq[kPointsPerLane];
        if (kPrefW) {
            // w of this chunk was prefetched into registers during the previous epilogue
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) {
                wv[q] = wnext[q];
                if (kStage) asm volatile("" : "+v"(wv[q]));   // consume the load before any LDS-DMA is in flight
            }
            if (kStage && full) {
                // LDS-DMA: u, v, weights of this chunk -> the wave's staging area, 16 B per lane
                // per instruction, no VGPRs held; they land while the peak loop runs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier reads of the area are done
#pragma unroll
                for (int i = 0; i < kChunk / 128; ++i) {
                    const int64_t js = jb + i * 128 + lane * 2;
                    __builtin_amdgcn_global_load_lds((gbl_void *)(u + js), (lds_void *)(stage + i * 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gbl_void *)(v + js), (lds_void *)(stage + kChunk + i * 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gbl_void *)(wt + js), (lds_void *)(stage + 2 * kChunk + i * 128), 16, 0, 0);
                }
            }
        } else if (full) {
#if defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD == 3   // diagnostic: the loads come from LDS (garbage) instead of memory
            const double2 *wp = reinterpret_cast<const double2 *>(lds_raw + aux_off) + lane;
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const double2 d = wp[m * kWave];
                wv[2 * m] = d.x + (double)(jl + 2 * m * kWave) * (lane_step * (1.0 / 64.0)) - wspan;
                wv[2 * m + 1] = d.y + (double)(jl + (2 * m + 1) * kWave) * (lane_step * (1.0 / 64.0)) - wspan;
            }
#elif defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD >= 2   // diagnostic: no grid loads (values are wrong on purpose)
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) wv[q] = (double)(jl + q * kWave) * (lane_step * (1.0 / 64.0)) - wspan;
#else
            // (grid_slot order: the lane's points 2m, 2m+1 are one 16-byte pair -> global_load_dwordx4)
            const double2 *wp = lane_ptr(wc + jb, lane);
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const double2 d = wp[m * kWave];
                wv[2 * m] = d.x;
                wv[2 * m + 1] = d.y;
            }
#endif
        } else {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q)
                wv[q] = (jl + q * kWave < j1) ? wc[jb + (q >> 1) * (2 * kWave) + 2 * lane + (q & 1)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kPointsPerLane; ++q) acc[q] = base;

        if (VARIANT == NMRFIT_VARIANT_BASELINE) {
            for (int k = 0; k < P; ++k) {
                const PeakLor rec = lor[k];
#pragma unroll
                for (int q = 0; q < kPointsPerLane; ++q) {
                    const double t = __builtin_fma(wv[q], rec.ihw, rec.c);
                    const double s = __builtin_fma(t, t, 1.0);
                    acc[q] = __builtin_fma(rec.al, 1.0 / s, acc[q]);
                    acc[q] = __builtin_fma(rec.ag2, exp2(-s), acc[q]);
                }
            }
        } else {
            double2 mm = make_double2(0.0, 0.0);
            if (kSkip) mm = chunk_minmax[jb / kChunk];
            if constexpr (kFar) {
                // ---- far-field form --------------------------------------------------------
                // For a peak whose centre is far from this chunk (rho = chunk half-span /
                // |distance to the pole of 1/(1+t^2)| <= 0.1) the Lorentzian is summed through a
                // Taylor expansion about the chunk centre: 1/(1+(tc+tau)^2) = Im sum_n
                // (-tau)^n q^(n+1), q = 1/(tc - i).  The expansions of ALL far peaks share one
                // set of kFarTerms coefficients in u = (w - centre)/half-span, so their cost per
                // point is one degree-15 Horner instead of ~6 ops per peak; truncation
                // <= 0.1^16 of each peak's term.  Near peaks are evaluated directly.
                const double wcen = wave_uniform(0.5 * (mm.x + mm.y));
                const double hw = wave_uniform(0.5 * (mm.y - mm.x));
                double cf[kFarTerms];
                bool horner_done = false;
                if (P <= 32 && FIT_IM != 2) {   // (the all-peak imaginary pass below reuses the scratch that parks the odd chunk's sums)
                    // Half a wave of peaks: the even chunks of a segment work out the expansions
                    // of TWO chunks at once -- lanes 0..31 for this chunk, lanes 32..63 for the
                    // next -- and park the second set (sums in LDS, masks in SGPRs) for the odd
                    // chunk that follows.  Either half runs the same operations in the same
                    // order, so a chunk's coefficients do not depend on which half made them.
                    unsigned near_c, hits_c;
                    if constexpr ((kAblate & 1) != 0) {
                        near_c = hits_c = 0u;
                    } else {
                        if constexpr (!kFarPipe)
                            if (!ff_odd) {
                                expand_pair(jb);
                                expand_sums();
                            }
                        near_c = ff_odd ? pend_near : even_near;
                        hits_c = ff_odd ? pend_hits : even_hits;
                    }
                    wave_lds_fence();
                    const double *src = ffs + (ff_odd ? kFarTerms : 0);
                    if constexpr ((kAblate & 1) != 0) {
#pragma unroll
                        for (int n = 0; n < kFarTerms; ++n) cf[n] = base * (double)(n + 1);
                    } else {
#pragma unroll
                    for (int n = 0; n < kFarTerms; ++n) cf[n] = src[n];
                    }
                    if constexpr ((kAblate & 2) != 0) near_c = hits_c = 0u;
                    if constexpr (kHornerFirst && (kAblate & 4) == 0) {
                        // The shared polynomial FIRST, straight into the accumulators (the offset P*yoff rides in its constant
                        // term): its 16 coefficients are dead before the near peaks and Gaussians need their registers,
                        // and the accumulators need neither initialising nor a separate add per point.
                        const double ihw1 = (hw > 0.0) ? rcp64(hw) : 0.0;
                        const double c0 = cf[0] + base;
#pragma unroll
                        for (int q = 0; q < kPointsPerLane; ++q) {
                            const double uu = (wv[q] - wcen) * ihw1;
                            double pz = cf[kFarTerms - 1];
#pragma unroll
                            for (int n = kFarTerms - 2; n >= 1; --n) pz = __builtin_fma(pz, uu, cf[n]);
                            acc[q] = __builtin_fma(pz, uu, c0);
                        }
                        horner_done = true;
                    }
                    for (unsigned m = near_c; m; m &= m - 1) lorentz_one(lor + __builtin_ctz(m), wv, acc);
                    if (kRec && full && rec_all) {
                        for (unsigned m = hits_c; m; m &= m - 1) gauss_add_rec(lor + __builtin_ctz(m), grec + __builtin_ctz(m), wv, acc);
                    } else {
                        for (unsigned m = hits_c; m; m &= m - 1) gauss_add(lor + __builtin_ctz(m), wv, acc);
                    }
                } else {
                double csum = 0.0;    // lane l: coefficient of order l >> 2 (all 4 lanes of a quad)
                for (int kb = 0; kb < P; kb += kWave) {
                    const int k = kb + lane;
                    const bool act = k < P;
                    bool far = false, ghit = false;
                    double zr = 0.0, zi = 0.0, mr = 0.0, mi = 0.0, al = 0.0;
                    if (act) {
                        const PeakLor rec = lor[k];
                        const PeakWin wn = win[k];
                        ghit = (mm.y >= (double)wn.lo) && (mm.x <= (double)wn.hi);
                        const double tc = __builtin_fma(wcen, rec.ihw, rec.c);
                        const double hk = hw * rec.ihw;
                        const double den = __builtin_fma(tc, tc, 1.0);
                        far = den >= 100.0 * hk * hk;            // rho^2 <= 0.01 (false for NaN)
                        const double rq = rcp64(den);
                        zr = tc * rq;                             // q = (tc + i)/(tc^2 + 1)
                        zi = rq;
                        mr = -hk * zr;                            // multiplier -hk*q per order
                        mi = -hk * zi;
                        al = rec.al;
                    }
                    const unsigned long long farmask = __ballot(far);
                    const unsigned long long nearmask = __ballot(act && !far);
                    const unsigned long long hits = __ballot(ghit);
                    if (farmask) {
                        // order n carries al * Im(q m^n), m = -hk q.  Both roots of the real
                        // recurrence y[n+1] = 2 Re(m) y[n] - |m|^2 y[n-1] have modulus |m|, so
                        // it is as stable as the complex product and costs two operations a term.
                        const double a2 = far ? mr + mr : 0.0;       // lanes without a far peak carry exact zeros
                        const double b2 = far ? -__builtin_fma(mr, mr, mi * mi) : 0.0;
                        double y0 = far ? al * zi : 0.0;
                        double y1 = far ? al * __builtin_fma(zr, mi, zi * mr) : 0.0;
                        double *dst = ffs + lane + (lane >> 4);
#pragma unroll
                        for (int n = 0; n < kFarTerms; ++n) {
                            dst[n * kFarPad] = y0;
                            const double y2 = __builtin_fma(a2, y1, b2 * y0);
                            y0 = y1;
                            y1 = y2;
                        }
                        wave_lds_fence();   // same-wave LDS write -> read
                        // lane l sums order l>>2 over 16 peaks of this pass (quarters padded to
                        // 17: each lane of a read group its own bank), then the quad combines
                        double part = 0.0;
                        const double *row = ffs + (lane >> 2) * kFarPad + (lane & 3) * 17;
#pragma unroll
                        for (int j = 0; j < 16; ++j) part += row[j];
                        part += __shfl_xor(part, 1, kWave);
                        part += __shfl_xor(part, 2, kWave);
                        csum += part;
                        wave_lds_fence();   // reads done before the next pass overwrites
                    }
                    for (unsigned long long m = nearmask; m; m &= m - 1)
                        lorentz_one(lor + kb + __builtin_ctzll(m), wv, acc);
                    for (unsigned long long m = hits; m; m &= m - 1) {
                        const int k1 = kb + __builtin_ctzll(m);
                        if (kRec && full && rec_all)
                            gauss_add_rec(lor + k1, grec + k1, wv, acc);
                        else
                            gauss_add(lor + k1, wv, acc);
                    }
                }
                // broadcast the kFarTerms sums through LDS and evaluate them at the lane's points
                if ((lane & 3) == 0) ffs[lane >> 2] = csum;
                wave_lds_fence();
#pragma unroll
                for (int n = 0; n < kFarTerms; ++n) cf[n] = ffs[n];
                }
                if constexpr (kFarPipe && ff_odd && full && FIT_IM != 2 && (kAblate & 1) == 0) {
                    // Software pipeline (round 4): the expansions of the NEXT pair of chunks are started here, in the
                    // odd chunk of the current pair -- whose own sums are in registers (cf) and whose masks are spent
                    // -- so that their LDS writes complete under the Horner below instead of standing at the head of
                    // the next even chunk with nothing else to issue; the sums over peaks follow in the epilogue.
                    if (P <= 32 && jb + kChunk < j1) expand_pair(jb + kChunk);
                }
                if (!horner_done) {
                const double ihwc = (hw > 0.0) ? rcp64(hw) : 0.0;
                if constexpr ((kAblate & 4) != 0) {
#pragma unroll
                    for (int q = 0; q < kPointsPerLane; ++q) acc[q] += cf[q] + cf[q + 8] * wv[q];
                } else
#pragma unroll
                for (int q = 0; q < kPointsPerLane; ++q) {
                    const double uu = (wv[q] - wcen) * ihwc;
                    double pz = cf[kFarTerms - 1];
#pragma unroll
                    for (int n = kFarTerms - 2; n >= 0; --n) pz = __builtin_fma(pz, uu, cf[n]);
                    acc[q] += pz;
                }
                }
                wave_lds_fence();
            } else
            for (int kb = 0; kb < P; kb += kWave) {
                const int kend = (P < kb + kWave) ? P : kb + kWave;
                // which of peaks kb..kb+63 have their Gaussian window inside this chunk's
                // [min,max] of w: lane i tests peak kb+i, the ballot is a scalar bit mask
                unsigned long long hits = ~0ull;
                if (kSkip) {
                    bool h = false;
                    if (kb + lane < P) {
                        const PeakWin wn = win[kb + lane];
                        h = (mm.y >= (double)wn.lo) && (mm.x <= (double)wn.hi);
                    }
                    hits = __ballot(h);
                }
                // Lorentzians first, in straight-line groups; then the (few) Gaussians whose
                // window touches this chunk, one scalar loop over the set bits of the mask
                int k = kb;
                if constexpr (kFastLoop) {
                    for (; k + kGroup <= kend; k += kGroup) lorentz_group_fast<kGroup>(lorf + k, wv, acc);
                    if (k < kend) lorentz_tail_fast(kend - k, lorf + k, wv, acc);
                } else {
                    for (; k + kGroup <= kend; k += kGroup) lorentz_group<kGroup>(lor + k, wv, acc);
                    if (k < kend) lorentz_tail<kGroup>(kend - k, lor + k, wv, acc);   // one smaller group
                }
                if (kend - kb < kWave) hits &= (1ull << (kend - kb)) - 1ull;
                if (kRec && full && rec_all) {
                    for (unsigned long long m = hits; m; m &= m - 1) {
                        const int k1 = kb + __builtin_ctzll(m);
                        gauss_add_rec(lor + k1, grec + k1, wv, acc);
                    }
                } else
                for (unsigned long long m = hits; m; m &= m - 1) gauss_add(lor + kb + __builtin_ctzll(m), wv, acc);
            }
        }

        // ---- imaginary model, all peaks (FIT_IM == 2: what generate_result builds, utils.py:271-277) ----
        // I(w) = sum_k [ al_k t/(1+t^2) + (ag2_k/sqrt(pi)) D(sqrt(ln2) t) ]: the Hilbert partner of the
        // pseudo-Voigt sum.  It decays only like 1/t, so there is no window to skip; instead every peak
        // that is FAR from this chunk (the chunk spans <= 0.1 of its distance to the pole, and Dawson's
        // asymptotic series holds over all of it) goes through ONE shared degree-15 polynomial per
        // chunk: t/(1+t^2) is the real part of the same series 1/(t - i) = sum_n q m^n u^n whose
        // imaginary part FARFIELD sums, and x^-(2j+1) of D's series expands binomially about the
        // chunk centre (all terms of one sign: no cancellation; truncation <= 1e-16 of each peak's
        // term).  Near peaks are evaluated point by point with the gathered Dawson table.
        double iacc[kPointsPerLane];
        if constexpr (FIT_IM == 2) {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) iacc[q] = 0.0;
            const double2 mi2 = chunk_minmax[jb / kChunk];
            const double icen = wave_uniform(0.5 * (mi2.x + mi2.y));
            const double ihalf = wave_uniform(0.5 * (mi2.y - mi2.x));
            double isum = 0.0;     // lane l: coefficient of order l >> 2 (all 4 lanes of a quad)
            bool anyfar = false;
            for (int kb = 0; kb < P; kb += kWave) {
                const int k = kb + lane;
                const bool act = k < P;
                bool farim = false;
                double tc = 0.0, hk = 0.0, rq = 0.0, al = 0.0, agd = 0.0;
                if (act) {
                    const PeakLor rec = lor[k];
                    tc = __builtin_fma(icen, rec.ihw, rec.c);
                    hk = ihalf * rec.ihw;
                    const double den = __builtin_fma(tc, tc, 1.0);
                    farim = (den >= 100.0 * hk * hk) && ((fabs(tc) - fabs(hk)) * kSqrtLn2 >= kDawFarX);   // false for NaN
                    rq = rcp64(den);
                    al = rec.al;
                    agd = rec.ag2 * kInvSqrtPi;
                }
                const unsigned long long farmask = __ballot(farim);
                const unsigned long long nearmask = __ballot(act && !farim);
                if (farmask) {
                    anyfar = true;
                    // Lorentzian dispersion: al * Re(q m^n), q = (tc + i)/(tc^2 + 1), m = -hk q, by the
                    // real two-term recurrence (both roots of modulus |m|)
                    const double qr = tc * rq, qi = rq;
                    const double mr = -hk * qr, mi = -hk * qi;
                    const double a2 = farim ? mr + mr : 0.0;
                    const double b2 = farim ? -__builtin_fma(mr, mr, mi * mi) : 0.0;
                    double y0 = farim ? al * qr : 0.0;
                    double y1 = farim ? al * __builtin_fma(qr, mr, -(qi * mi)) : 0.0;
                    // Gaussian dispersion: agd * sum_j A_j x^-(2j+1), x = xc (1 - eps u), eps = -hk/tc:
                    // coefficient of u^n = agd eps^n sum_j B_j binom(2j + n, n), B_j = A_j xc^-(2j+1)
                    double B[kDawFarTerms];
                    {
                        const double xc = farim ? kSqrtLn2 * tc : 1.0;
                        const double inv = rcp64(xc), inv2 = inv * inv;
                        double pw = farim ? agd * inv : 0.0;
#pragma unroll
                        for (int j = 0; j < kDawFarTerms; ++j) {
                            B[j] = (dawson::kFar[j] * pow49_half(j)) * pw;
                            pw *= inv2;
                        }
                    }
                    const double eps = farim ? -hk * rcp64(tc) : 0.0;
                    double en = 1.0;
                    double *dst = ffs + lane + (lane >> 4);
#pragma unroll
                    for (int n = 0; n < kFarTerms; ++n) {
                        double sg = 0.0;
#pragma unroll
                        for (int j = kDawFarTerms - 1; j >= 0; --j) sg = __builtin_fma(B[j], binom_d(2 * j + n, n), sg);
                        dst[n * kFarPad] = __builtin_fma(sg, en, y0);
                        en *= eps;
                        const double y2 = __builtin_fma(a2, y1, b2 * y0);
                        y0 = y1;
                        y1 = y2;
                    }
                    wave_lds_fence();   // same-wave LDS write -> read
                    double part = 0.0;
                    const double *row = ffs + (lane >> 2) * kFarPad + (lane & 3) * 17;
#pragma unroll
                    for (int j = 0; j < 16; ++j) part += row[j];
                    part += __shfl_xor(part, 1, kWave);
                    part += __shfl_xor(part, 2, kWave);
                    isum += part;
                    wave_lds_fence();   // reads done before the next pass overwrites
                }
                for (unsigned long long m = nearmask; m; m &= m - 1) {
                    const PeakLor rec = lor[kb + __builtin_ctzll(m)];
#pragma unroll
                    for (int q = 0; q < kPointsPerLane; ++q) iacc[q] += dispersion_tab(wv[q], rec, dtab);
                }
            }
            if (anyfar) {   // wave-uniform
                if ((lane & 3) == 0) ffs[lane >> 2] = isum;
                wave_lds_fence();
                double cfi[kFarTerms];
#pragma unroll
                for (int n = 0; n < kFarTerms; ++n) cfi[n] = ffs[n];
                const double ihc = (ihalf > 0.0) ? rcp64(ihalf) : 0.0;
#pragma unroll
                for (int q = 0; q < kPointsPerLane; ++q) {
                    double uu = (wv[q] - icen) * ihc;
                    if (!full) uu = fmin(fmax(uu, -1.0), 1.0);   // padding points of the ragged chunk (weight 0)
                    double pz = cfi[kFarTerms - 1];
#pragma unroll
                    for (int n = kFarTerms - 2; n >= 0; --n) pz = __builtin_fma(pz, uu, cfi[n]);
                    iacc[q] += pz;
                }
                wave_lds_fence();
            }
        }

        if constexpr (FIT_IM == 1) {   // equations.py:197-199: the last peak's line only (before the data loads: 48 VGPRs fewer are live)
            __builtin_amdgcn_sched_barrier(0);   // (not interleaved with the far-field Horner above: its coefficients are dead first)
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) iacc[q] = 0.0;
            if (P > 0) dispersion_points(wv, lor[P - 1], dtab, iacc);
        }
        // keep the u/v/weights loads below the peak loop: hoisted, they would hold 48 VGPRs
        // across it
        asm volatile("" ::: "memory");
        if (kPrefW) {
            if (kStage && full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of this chunk has landed
            // prefetch w of the next chunk into registers (the per-peak constants are dead here)
            const int64_t jn = jl + kChunk, jnb = jb + kChunk;
            if (jb + 2 * kChunk <= j1) {
                const double2 *wp = reinterpret_cast<const double2 *>(wc + jnb) + lane;
#pragma unroll
                for (int m = 0; m < kPointsPerLane / 2; ++m) {
                    const double2 d = wp[m * kWave];
                    wnext[2 * m] = d.x;
                    wnext[2 * m + 1] = d.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < kPointsPerLane; ++q)
                    wnext[q] = (jn + q * kWave < j1) ? wc[jnb + (q >> 1) * (2 * kWave) + 2 * lane + (q & 1)] : 0.0;
            }
        }
        if (kStage && full) {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) {
                const int o = (q >> 1) * (2 * kWave) + 2 * lane + (q & 1);   // the staged copy keeps the grid_slot order
                uq[q] = stage[o];
                vq[q] = stage[kChunk + o];
                tq[q] = stage[2 * kChunk + o];
            }
        } else if (full) {
#if defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD == 3   // diagnostic: the loads come from LDS (garbage) instead of memory
            const double2 *up = reinterpret_cast<const double2 *>(lds_raw + aux_off) + kChunk / 2 + lane, *vp = up + kChunk / 2, *tp = vp + kChunk / 2;
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const double2 du = up[m * kWave], dv = vp[m * kWave], dt = tp[m * kWave];
                uq[2 * m] = du.x + (double)(lane + 2 * m) * 1.0e-3;
                uq[2 * m + 1] = du.y + (double)(lane + 2 * m + 1) * 1.0e-3;
                vq[2 * m] = dv.x + (double)(lane - 2 * m) * 1.0e-3;
                vq[2 * m + 1] = dv.y + (double)(lane - 2 * m - 1) * 1.0e-3;
                tq[2 * m] = dt.x + 1.0;
                tq[2 * m + 1] = dt.y + 1.125;
            }
#elif defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD >= 1   // diagnostic: no data loads (values are wrong on purpose)
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) {
                uq[q] = (double)(lane + q) * 1.0e-3;
                vq[q] = (double)(lane - q) * 1.0e-3;
                tq[q] = 1.0 + (double)q * 0.125;
            }
#else
            const double2 *up = lane_ptr(u + jb, lane), *vp = lane_ptr(v + jb, lane), *tp = lane_ptr(wt + jb, lane);
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const double2 du = up[m * kWave], dv = vp[m * kWave], dt = tp[m * kWave];
                uq[2 * m] = du.x;
                uq[2 * m + 1] = du.y;
                vq[2 * m] = dv.x;
                vq[2 * m + 1] = dv.y;
                tq[2 * m] = dt.x;
                tq[2 * m + 1] = dt.y;
            }
#endif
        } else {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) {
                const bool ok = jl + q * kWave < j1;
                const int64_t js = jb + (q >> 1) * (2 * kWave) + 2 * lane + (q & 1);   // grid_slot order
                uq[q] = ok ? u[js] : 0.0;
                vq[q] = ok ? v[js] : 0.0;
                tq[q] = ok ? wt[js] : 0.0;   // weight 0: the point contributes nothing
            }
        }
        if constexpr (kFar && kFarPipe && ff_odd && full && FIT_IM != 2 && (kAblate & 1) == 0) {
            // ... and their sums over peaks here, while this chunk's u / v / weights are on their way: the LDS round
            // trip (16 reads, the sums, one write) overlaps with a wait the wave has anyway.
            if (P <= 32 && jb + kChunk < j1) expand_sums();
        }
        if constexpr ((kAblate & 8) != 0) {
#pragma unroll
            for (int q = 0; q < kPointsPerLane; ++q) bs = __builtin_fma(acc[q], tq[q] + uq[q] * vq[q], bs);
        } else
#pragma unroll
        for (int q = 0; q < kPointsPerLane; ++q) {
            const double vd = __builtin_fma(zr, uq[q], -(zi * vq[q]));   // Re((zr + i zi)(u + i v))
            const double e = tq[q] * (vd - acc[q]);                       // equations.py:202
            bs = __builtin_fma(e, e, bs);
            if (FIT_IM != 0) {                                            // equations.py:197-199,205-206
                const double id = __builtin_fma(zr, vq[q], zi * uq[q]);  // Im((zr + i zi)(u + i v))
                const double ei = tq[q] * (id - iacc[q]);
                bs_im = __builtin_fma(ei, ei, bs_im);
            }
            if (WRITE_R && (full || jl + q * kWave < j1)) R_out[particle * N + jl + q * kWave] = e;
            const double nzr = __builtin_fma(zr, rr, -(zi * ri));         // z *= rho
            zi = __builtin_fma(zr, ri, zi * rr);
            zr = nzr;
        }
        // Canonical summation order: lane sums over its points of the block, wave tree over
        // lanes, then block sums are added one after another in grid order -- by this wave if
        // it owns the whole grid, else by finalize_kernel.
        if (++cib == blk_chunks || jb + kChunk >= j1) {
            const double cs = wave_sum(bs);
            const double cs_im = (FIT_IM != 0) ? wave_sum(bs_im) : 0.0;
            bs = 0.0;
            bs_im = 0.0;
            cib = 0;
            if (nseg == 1) {
                ss += cs;
                ss_im += cs_im;
            } else if (kOneWorkgroupParticle && nseg == WPB) {   // the waves of THIS workgroup (four, or eight) hold the whole particle: sums meet in LDS
                if (lane == 0) {
                    wsums[blk0 + bidx] = cs;
                    if (FIT_IM != 0) wsums[kMaxBlocks + blk0 + bidx] = cs_im;
                }
            } else if (lane == 0) {
                const int64_t slot = particle * n_blocks + blk0 + bidx;
                if (FIT_IM == 0) {
                    out[slot] = cs;
                } else {
                    out[2 * slot] = cs;
                    out[2 * slot + 1] = cs_im;
                }
            }
            ++bidx;
        }
    };
    int64_t jb = j0;
    if constexpr (kFar && kFarPipe && FIT_IM != 2 && (kAblate & 1) == 0)
        if (P <= 32 && j0 < j1) {   // the first pair's expansions; every later pair's: in the odd chunk before it
            expand_pair(j0);
            expand_sums();
        }
    if constexpr (VARIANT == NMRFIT_VARIANT_FARFIELD) {   // chunks alternate even / odd from the segment start
        for (; jb + 2 * kChunk <= j1; jb += 2 * kChunk) {
            chunk(jb, std::true_type{}, std::false_type{});
            chunk(jb + kChunk, std::true_type{}, std::true_type{});
        }
        if (jb + kChunk <= j1) {
            chunk(jb, std::true_type{}, std::false_type{});
            jb += kChunk;
            if (jb < j1) chunk(jb, std::false_type{}, std::true_type{});
        } else if (jb < j1) {
            chunk(jb, std::false_type{}, std::false_type{});
        }
    } else {
        for (; jb + kChunk <= j1; jb += kChunk) chunk(jb, std::true_type{}, std::false_type{});
        if (jb < j1) chunk(jb, std::false_type{}, std::false_type{});
    }
    };
    if (kFast && fast_all)
        chunk_loop(std::integral_constant<bool, kFast>{});
    else
        chunk_loop(std::false_type{});

    phase_stamp(clk, 3);   // chunk loop done
    if (clk && g == 0 && lane == 0) {
        clk[2] = __builtin_amdgcn_s_memtime();
        clk[3] = __builtin_amdgcn_s_memrealtime();
    }
    // Personal best of this particle, when this wave / workgroup holds all of it and the swarm asked for it
    // (fused generations only: the updated row sits in LDS): pyswarm's `i_update = fx < fp; p[i_update] =
    // x[i_update]; fp[i_update] = fx[i_update]`.  Particle-local: nobody else reads or writes this row in this launch.
    auto personal_best = [&](const double f) {
        if constexpr (!WRITE_R) {
            if (wsums[2 * kMaxBlocks + 1] != 0.0) {
                // What this needs -- p, S, the row's place in LDS -- was parked in LDS by the prologue (wsums[..+2..4])
                // and is read back here: kept in scalar registers across the chunk loop those few values tipped the
                // headline kernel, which has neither a scalar nor a vector register to spare, into scratch memory.
                // fp[S] sits right behind p[S x D] (PsoFused).
                const int64_t D2 = 4 + 3 * (int64_t)P;
                // (as many segments as waves per workgroup: workgroup = particle -- no need for the prologue's 64-bit division result)
                const int64_t part = (int64_t)blockIdx.x;
                double *pb = reinterpret_cast<double *>((uintptr_t)__double_as_longlong(wsums[2 * kMaxBlocks + 2]));
                double *fpb = pb + __double_as_longlong(wsums[2 * kMaxBlocks + 3]) * D2;
                const double *row = reinterpret_cast<const double *>(lds_raw + (unsigned)__double_as_longlong(wsums[2 * kMaxBlocks + 4]));
                const int ln = threadIdx.x & (kWave - 1);
                const double fp_old = wsums[2 * kMaxBlocks + 5];   // (requested by the kernel's first instructions)
                const long long pflip = __double_as_longlong(wsums[2 * kMaxBlocks + 7]);   // (0: no deferred fold)
                if (pflip != 0) {
                    // deferred form: the other (p, fp) buffer gets this particle's row and value whether it improved
                    // or not (PsoFused::pflip; the old row was parked in row 2 of the LDS area by the prologue)
                    const bool better = f < fp_old;
                    const double *keep = row + 2 * D2;
                    for (int64_t d = ln; d < D2; d += kWave) pb[pflip + part * D2 + d] = better ? row[d] : keep[d];
                    if (ln == 0) fpb[pflip + part] = better ? f : fp_old;
                    phase_stamp(clk, 5);   // personal best on its way to memory
                } else if (f < fp_old) {
                    for (int64_t d = ln; d < D2; d += kWave) pb[part * D2 + d] = row[d];
                    if (ln == 0) fpb[part] = f;
                }
            }
        }
    };
    if (nseg == 1) {
        double f = 0.0;
        if (FIT_IM == 0)
            f = sqrt(ss / (double)N);
        else   // (rmse_real + rmse_imag) / 2, equations.py:205-209
            f = 0.5 * (sqrt(ss / (double)N) + sqrt(ss_im / (double)N));
        if (lane == 0) out[particle] = f;
        // (no fused personal best here: one wave per particle means >= 16384 particles, where the swarm's own
        // select kernels are noise next to the objective -- and the call cost the headline kernel 12 bytes of scratch)
    }
    if (kOneWorkgroupParticle && nseg == WPB && nseg > 1) {
        // One workgroup = one particle (segment = wave): the block sums are added here, in grid order
        // like finalize_value does -- the same canonical order, bit-identical f -- and the launch needs
        // neither the partial-sum buffer nor a finalize pass after it.  (All its waves get here: a
        // workgroup is active or inactive as a whole, and a stopped swarm returned before the loop.)
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0, ti = 0.0;
            for (int64_t c = 0; c < n_blocks; ++c) {
                t += wsums[c];
                if (FIT_IM != 0) ti += wsums[kMaxBlocks + c];
            }
            const double f = (FIT_IM == 0) ? sqrt(t / (double)N) : 0.5 * (sqrt(t / (double)N) + sqrt(ti / (double)N));
            out[particle] = f;
            wsums[2 * kMaxBlocks] = f;
        }
        if constexpr (!WRITE_R) {
            // (wave 0 alone goes on: f travels from its lane 0 through the same LDS word, no second workgroup barrier)
            if ((threadIdx.x >> 6) == 0) {
                wave_lds_fence();
                phase_stamp(clk, 4);   // f known
                personal_best(wsums[2 * kMaxBlocks]);
            }
        }
    }
}

template <int VARIANT, bool WRITE_R, int FIT_IM, int WPB = kWavesPerBlock>
__global__ __launch_bounds__(kWave *WPB, (WPB == kWavesPerBlock) ? NMRFIT_OBJECTIVE_MIN_WAVES(VARIANT, FIT_IM) : 2) void objective_kernel(
    const double *__restrict__ wc, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ wt, const double2 *__restrict__ chunk_minmax, const double *__restrict__ X, int64_t S,
    int P, int64_t N, double w0, double wspan, int nseg, int64_t seg_len, int blk_chunks, double lane_step,
    double rec_devk, double *__restrict__ out, double *__restrict__ R_out, unsigned long long *__restrict__ clk,
    const PsoFused upd, const unsigned aux_off)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // block sums (x2 with the imaginary channel); then f; then what the fused personal-best step needs at the very
    // end of the kernel (flag, p, S, the row's LDS offset), parked here by the prologue
    __shared__ double wsums[kWsumsCount];
    if (threadIdx.x == 0) {   // first thing in the kernel, while nothing else is live (a barrier follows the staging)
        const bool pbest = !WRITE_R && upd.x_in != nullptr && upd.pbest != 0u;
        wsums[2 * kMaxBlocks + 1] = pbest ? 1.0 : 0.0;
        wsums[2 * kMaxBlocks + 2] = __longlong_as_double((long long)(uintptr_t)upd.p);
        wsums[2 * kMaxBlocks + 3] = __longlong_as_double((long long)S);
        wsums[2 * kMaxBlocks + 4] = __longlong_as_double((long long)upd.xrow_off);
        // this particle's personal-best value, requested NOW: a memory round trip off the end of the kernel's
        // critical path (nobody else writes it in this launch)
        if (pbest && nseg == WPB) wsums[2 * kMaxBlocks + 5] = upd.p[S * (4 + 3 * (int64_t)P) + blockIdx.x];
        // deferred fold: where the other (p, fp) buffer is (never 0 then) -- read back by the personal-best step
        wsums[2 * kMaxBlocks + 7] = __longlong_as_double((pbest && upd.tail != 0u) ? (long long)upd.pflip : 0LL);
    }
#ifdef NMRFIT_DIAG_REMAP
    // block b, wave w -> particle 4*(b / nseg) + w, segment b % nseg (S a multiple of 4)
    const int64_t g = ((int64_t)(blockIdx.x / nseg) * WPB + (threadIdx.x >> 6)) * nseg + (blockIdx.x % nseg);
#elif defined(NMRFIT_DIAG_VECTOR_G)   // A/B: the wave index as the compiler sees it without help (per-lane)
    const int64_t g = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
#else
    const int64_t g = (int64_t)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
    objective_body<VARIANT, WRITE_R, FIT_IM, WPB>(lds_raw, g, wc, u, v, wt, chunk_minmax, X, S, P, N, w0, wspan, nseg, seg_len,
                                             blk_chunks, lane_step, rec_devk, out, R_out, clk, upd, aux_off, wsums);
}

// f[i] = sqrt( (sum of the particle's per-block sums, in grid order) / N ); with the imaginary
// part: the mean of the real and imaginary RMSE (two sums per block)
__global__ void finalize_kernel(const double *__restrict__ partial, int64_t S, int64_t n_chunks, int64_t N,
                                int fit_im, double *__restrict__ f)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    f[i] = finalize_value(partial + i * n_chunks * (fit_im ? 2 : 1), n_chunks, N, fit_im);
}

// Per-peak real and imaginary contributions on an output grid (FitUtility.generate_result,
// nmrfit/utils.py:262-281): real[k, j] = voigt(w_j; r, yoff, peak k) (equations.py:141-147),
// imag[k, j] = its Kramers-Kronig partner in closed form.  One thread per (peak, point).
__global__ void contributions_kernel(const double *__restrict__ wc_out, int64_t Nout, const double *__restrict__ x,
                                     int P, double w0, double wspan, double *__restrict__ real_out,
                                     double *__restrict__ imag_out, int grid_order)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)P * Nout) return;
    const int k = (int)(idx / Nout);
    const int64_t j = idx - (int64_t)k * Nout;
    const double r = x[2], yoff = x[3];
    const double width = x[4 + 3 * k], loc = x[5 + 3 * k], a = x[6 + 3 * k];
    const double ihw = 2.0 / width;
    const double locc = loc - w0;
    const double lim = 1.0e18 / (wspan + fabs(locc));
    PeakLor rec;
    rec.ihw = (fabs(ihw) > lim) ? copysign(lim, ihw) : ihw;
    rec.c = -locc * rec.ihw;
    rec.al = a * r * ihw * kInvPi;
    rec.ag2 = 2.0 * a * (1.0 - r) * ihw * kSqrtLn2OverPi;
    const double wj = wc_out[grid_order ? grid_slot(j) : j];
    const double t = __builtin_fma(wj, rec.ihw, rec.c);
    const double s = __builtin_fma(t, t, 1.0);
    real_out[idx] = yoff + __builtin_fma(rec.al, rcp64(s), rec.ag2 * exp2_neg(-s));
    imag_out[idx] = dispersion(wj, rec);
}

// per-chunk (min, max) of the centred grid: one wave per chunk
__global__ void chunk_minmax_kernel(const double *__restrict__ wc, int64_t N, int64_t n_chunks,
                                    double2 *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int q = 0; q < kPointsPerLane; ++q) {
        const int64_t j = c * kChunk + q * kWave + lane;
        if (j < N) {
            const double x = wc[grid_slot(j)];
            lo = fmin(lo, x);
            hi = fmax(hi, x);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, kWave));
        hi = fmax(hi, __shfl_down(hi, off, kWave));
    }
    if (lane == 0) out[c] = make_double2(lo, hi);
}

__global__ void centre_kernel(const double *__restrict__ w, int64_t N, double w0, double *__restrict__ wc)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) wc[grid_slot(j)] = w[j] - w0;
}

__global__ void scatter_grid_kernel(const double *__restrict__ src, int64_t N, double *__restrict__ dst)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) dst[grid_slot(j)] = src[j];
}

// Which instantiations exist with eight-wave workgroups (one workgroup = one particle cut into eight segments):
// the objective launches without the imaginary channel of the three kernels fit() can select.
constexpr bool has_eight_wave_form(int variant)
{
    return variant == NMRFIT_VARIANT_DEFAULT || variant == NMRFIT_VARIANT_FARFIELD || variant == NMRFIT_VARIANT_NOREC;
}
constexpr int kWideWaves = 8;

template <int VARIANT>
int launch_variant(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *out, double *dR,
                   int nseg, int64_t seg_len, int blk_chunks, int64_t blocks, size_t lds, int fit_im,
                   const PsoFused &upd, unsigned aux_off, int wpb)
{
#define NMRFIT_LAUNCH_W(WR, FI, W)                                                                              \
    hipLaunchKernelGGL((objective_kernel<VARIANT, WR, FI, W>), dim3((unsigned)blocks), dim3(kWave *(W)), lds,  \
                       ctx->stream, ctx->d_wc, ctx->d_u, ctx->d_v, ctx->d_wt, ctx->d_chunk, dX, S, (int)P,     \
                       ctx->N, ctx->w0, ctx->wspan, nseg, seg_len, blk_chunks, ctx->lane_step,                 \
                       ctx->grid_dev * 11.0e10, out, dR, clk, upd, aux_off)
#define NMRFIT_LAUNCH(WR, FI) NMRFIT_LAUNCH_W(WR, FI, kWavesPerBlock)
    // nmrfit_prof_enable: HIP events on the launch stream around this kernel alone
    const bool prof = ctx->prof_cap > 0 && ctx->prof_nk < ctx->prof_cap;
    unsigned long long *clk = prof ? ctx->d_clk : nullptr;
    if (prof) NMRFIT_HIP(hipEventRecord(ctx->prof_k0[(size_t)ctx->prof_nk], ctx->stream));
    if (dR) {
        NMRFIT_LAUNCH(true, 0);
    } else if (fit_im == 0) {
        if constexpr (has_eight_wave_form(VARIANT)) {
            if (wpb == kWideWaves)
                NMRFIT_LAUNCH_W(false, 0, kWideWaves);
            else
                NMRFIT_LAUNCH(false, 0);
        } else {
            NMRFIT_LAUNCH(false, 0);
        }
    } else if constexpr (VARIANT == NMRFIT_VARIANT_DEFAULT || VARIANT == NMRFIT_VARIANT_FARFIELD || VARIANT == NMRFIT_VARIANT_NOREC) {
        if (fit_im == 1)
            NMRFIT_LAUNCH(false, 1);
        else
            NMRFIT_LAUNCH(false, 2);
    } else {
        set_error("fit_im is implemented for the DEFAULT, NOREC and FARFIELD kernel variants (STAGED runs DEFAULT) only");
        return NMRFIT_E_UNSUPPORTED;
    }
#undef NMRFIT_LAUNCH
#undef NMRFIT_LAUNCH_W
    NMRFIT_HIP(hipGetLastError());
    if (prof) {
        NMRFIT_HIP(hipEventRecord(ctx->prof_k1[(size_t)ctx->prof_nk], ctx->stream));
        ++ctx->prof_nk;
    }
    return NMRFIT_OK;
}

}  // namespace

// Builds the derived device arrays of a context (centred grid, chunk table).
int prepare_grid(nmrfit_ctx *ctx, const double *d_w_raw)
{
    const int64_t N = ctx->N;
    hipLaunchKernelGGL(centre_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_w_raw, N,
                       ctx->w0, ctx->d_wc);
    NMRFIT_HIP(hipGetLastError());
    const int waves_per_block = 4;
    hipLaunchKernelGGL(chunk_minmax_kernel, dim3((unsigned)((ctx->n_chunks + waves_per_block - 1) / waves_per_block)),
                       dim3(kWave * waves_per_block), 0, ctx->stream, ctx->d_wc, N, ctx->n_chunks, ctx->d_chunk);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

int scatter_grid(nmrfit_ctx *ctx, const double *d_src, double *d_dst)
{
    const int64_t N = ctx->N;
    hipLaunchKernelGGL(scatter_grid_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_src, N, d_dst);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

// The kernel variant a launch actually runs (the requested one may not fit in LDS, or may not
// implement the imaginary part) and the dynamic LDS its per-wave records need.
constexpr size_t kStaticLds = (size_t)kWsumsCount * sizeof(double) + 64;   // objective_kernel's own __shared__ (wsums) + alignment slack
// `slices`: copies of the per-peak records in a workgroup (1 when its waves are segments of one particle, else wpb);
// `rows`: copies of the updated row kept for a fused swarm generation (0: none)
static size_t resolve_variant(const nmrfit_ctx *ctx, int32_t P, bool residual, int fit_im, int *variant_out,
                              unsigned *aux_off, int wpb, int slices, int rows)
{
    const size_t np = (size_t)std::max(P, 1);
    // what a workgroup may take of a CU's 160 KiB for the records sized here: everything but the kernel's static
    // LDS and (swarm generations) the per-wave copies of the updated rows that launch_objective appends
    const size_t room = 160 * 1024 - kStaticLds -
                        (rows ? (size_t)rows * (size_t)(4 + 3 * (int64_t)P) * sizeof(double) + 16 : 0);
    const size_t lds_recs = (((size_t)slices * np * (sizeof(PeakLor) + sizeof(PeakWin)) + 15) & ~(size_t)15) +
                            (size_t)wpb * kMaxBlocks * sizeof(double2) + kSharedPrologueBytes +
                            (slices == 1 ? 0 : (size_t)wpb * 2 * kWave * sizeof(double));   // per-wave lane phase seeds
    const size_t lds_stage = (size_t)wpb * 3 * kChunk * sizeof(double);
    const size_t lds_far = (size_t)wpb * kFarTerms * kFarPad * sizeof(double);
    // Gaussian recurrence constants (d, C) per peak: objective launches of DEFAULT / FARFIELD
    const size_t lds_rec = residual ? 0 : (size_t)slices * np * sizeof(double2);
    const size_t lds_fast = (NMRFIT_FASTPAIR != 0 && NMRFIT_GROUP == 8) ? (size_t)slices * np * sizeof(PeakFast) : 0;
    // the all-peak imaginary model sums far peaks through the far-field scratch and evaluates Dawson's
    // integral from a table in LDS
    const size_t lds_im = (fit_im == 2) ? lds_far : 0;
    // (fit_im == 1 reads the same table: gathered intervals near the last peak, the asymptotic series elsewhere)
    const size_t lds_tab = (fit_im != 0) ? (size_t)kDawTabCount * sizeof(double) + 16 : 0;
    int variant = ctx->variant;
    // the imaginary channel exists in DEFAULT, NOREC and FARFIELD; the A/B variants fail in launch_variant
    if (variant == NMRFIT_VARIANT_STAGED && fit_im != 0) variant = NMRFIT_VARIANT_DEFAULT;
    // STAGED needs three workgroups to still fit in a CU's 160 KiB (P <= 27); beyond that it
    // runs the unstaged kernel.
    if (variant == NMRFIT_VARIANT_STAGED && 3 * (lds_recs + lds_stage + kStaticLds) > 160 * 1024) variant = NMRFIT_VARIANT_DEFAULT;
    if (variant == NMRFIT_VARIANT_FARFIELD && lds_recs + lds_far + lds_rec + lds_tab + 16 > room)
        variant = NMRFIT_VARIANT_DEFAULT;   // P > ~600
    if (variant == NMRFIT_VARIANT_DEFAULT && lds_recs + lds_im + lds_rec + lds_fast + lds_tab + 16 > room)
        variant = NMRFIT_VARIANT_NOREC;     // P > ~450: no room for the recurrence / scaled records
    size_t lds = lds_recs + (variant == NMRFIT_VARIANT_STAGED ? lds_stage : 0) +
                 (variant == NMRFIT_VARIANT_FARFIELD ? lds_far : lds_im) +
                 ((variant == NMRFIT_VARIANT_FARFIELD || variant == NMRFIT_VARIANT_DEFAULT) ? lds_rec : 0) +
                 (variant == NMRFIT_VARIANT_DEFAULT ? lds_fast : 0);
    *aux_off = 0;
#if defined(NMRFIT_DIAG_NOLOAD) && NMRFIT_DIAG_NOLOAD == 3   // diagnostic: one chunk of the four arrays' worth of LDS to read from
    if (!lds_tab) {
        lds = (lds + 15) & ~(size_t)15;
        *aux_off = (unsigned)lds;
        lds += 4 * kChunk * sizeof(double);
    }
#endif
    if (lds_tab) {
        lds = (lds + 15) & ~(size_t)15;
        *aux_off = (unsigned)lds;
        lds += lds_tab;
    }
    *variant_out = variant;
    return lds;
}

int launch_objective(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df, double *dR,
                     ObjectiveDeferred *defer, const PsoFused *fused)
{
    if (defer) *defer = ObjectiveDeferred{};
    const int fit_im = dR ? 0 : ctx->fit_im;
    if (S == 0) return NMRFIT_OK;
    const int64_t N = ctx->N;
    // Segmenting: a wave is one (particle, segment) task; a segment is a whole number of blocks.  Swarms that already
    // supply enough waves get nseg = 1 and the wave writes f directly.  (Rounds 1-3 aimed at ~16 tasks per SIMD so that
    // the hardware dispatcher load-balances -- C3: 4096 one-per-particle waves 1.81 ms, 16384 waves 1.74 ms; the
    // round-4 rule below replaces it unless NMRFIT_SEG_RULE=3 or an explicit target is set.)
    int64_t target_waves = (int64_t)ctx->compute_units * 4 * 16;
    if (ctx->target_waves > 0) target_waves = ctx->target_waves;
    // Blocks: the unit of the canonical summation / phase re-seeding, a function of N only
    // (at most 16 per grid), so that results do not depend on S or on sharding.
    const int64_t n_chunks = (N + kChunk - 1) / kChunk;
    const int blk_chunks = (int)((n_chunks + kMaxBlocks - 1) / kMaxBlocks);
    const int64_t n_blocks = (n_chunks + blk_chunks - 1) / blk_chunks;
    const int64_t blk_len = (int64_t)blk_chunks * kChunk;
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(n_blocks, (target_waves + S - 1) / S));
    static const bool seg_rule_r4 = [] {
        const char *e = getenv("NMRFIT_SEG_RULE");   // A/B knob: 3 = the round-3 rule (~16 tasks per SIMD)
        return !(e && atoi(e) == 3);
    }();
    if (ctx->target_waves == 0 && seg_rule_r4) {
        // Round 4 (the selectable kernels now run four waves per SIMD): as few segments as fill the chip ONCE -- every
        // further segment repeats a wave's prologue and cuts its chunk loop shorter (1024 x 16384 x 12: 16 segments
        // 69 us, 4 segments 60 us; 4096 x 4096 x 6: 4 segments 56 us, 1 segment 49 us) -- but four where a wave then
        // still has 16 chunks or more: the prologue no longer counts there, and four segments make the workgroup
        // the particle (f and the personal best finished in the launch)
        const int64_t slots = (int64_t)ctx->compute_units * 4 * 4;
        nseg = std::max<int64_t>(1, std::min<int64_t>(n_blocks, (slots + S - 1) / S));
        if (n_chunks / 4 >= 16) nseg = std::max<int64_t>(nseg, std::min<int64_t>(4, n_blocks));
    }
    // Short grids: a wave's own prologue (block seeds, pointers, the barrier of the shared part) still
    // costs a good fraction of a chunk, so prefer >= 2 chunks per wave as long as two waves per SIMD
    // remain (measured on C2, S=1024 N=4096, with the per-workgroup prologue: 8 segments 25.0 us,
    // 4 segments 22.7 us, 2 segments 23.3 us; round 1, prologue per wave: 8 -> 26.2, 2 -> 22.2).
    if (ctx->target_waves == 0) {
        const int64_t simds = (int64_t)ctx->compute_units * 4;
        while (nseg > 1 && n_chunks / nseg < 2 && S * ((nseg + 1) / 2) >= 2 * simds) nseg = (nseg + 1) / 2;
        // a small swarm whose waves just miss fitting on the chip at once (three per SIMD) while half
        // as many would leave it well filled: take the half (204 x 16384 x 12: 16 segments 30.4 us per
        // generation, 8 segments 28.5; tools/archive/nseg_generation_ab.py)
        if (nseg > 1 && S * nseg > 3 * simds && S * (nseg / 2) < 2 * simds && n_chunks / nseg <= 2) nseg /= 2;
    }
    int64_t seg_len = ((n_blocks + nseg - 1) / nseg) * blk_len;
    nseg = (N + seg_len - 1) / seg_len;
    const int64_t waves = S * nseg;
    int variant = NMRFIT_VARIANT_DEFAULT;
    unsigned aux_off = 0;
    const bool fused_rows = fused && fused->x_in;
    // LDS copies: per-peak records once per workgroup when its waves are segments of ONE particle (nseg a multiple of
    // the waves per workgroup), else once per wave; the updated row of a fused swarm generation likewise, plus two
    // more rows (g, the winner's row) when the workgroup may finish the generation (objective.hip, personal_best)
    // swarms the two whole-generation forms of a fused launch take (PsoFused::tail), by waves per workgroup
    auto tail_fits = [&](unsigned tail, int w) {
        return tail != 0u && S <= (int64_t)kDeferredPerLane * kWave * w;
    };
    auto copies = [&](int w, int *slices, int *rows) {
        const bool one_particle = kOneWorkgroupParticle && (nseg % w == 0);   // (objective_body: `shared`)
        *slices = one_particle ? 1 : w;
        *rows = !fused_rows ? 0 : !one_particle ? w : (nseg == w && tail_fits(fused->tail, w)) ? 3 : 1;
    };
    int wpb = kWavesPerBlock, slices = 0, rows = 0;
    copies(wpb, &slices, &rows);
    size_t lds = resolve_variant(ctx, P, dR != nullptr, fit_im, &variant, &aux_off, wpb, slices, rows);
    // Eight segments per particle (small swarms on short grids -- the reference's default 204 x 4096): an EIGHT-wave
    // workgroup is the particle, as the four-wave workgroup is for four segments: one prologue per particle, block
    // sums through LDS, f (and, in a swarm generation, the personal best) finished in this launch.
    const size_t xrow_bytes = fused_rows ? (size_t)(4 + 3 * (int64_t)P) * sizeof(double) : 0;
    if (nseg == kWideWaves && !dR && fit_im == 0 && has_eight_wave_form(variant) && ctx->wide_workgroups) {
        int v8 = variant, s8 = 0, r8 = 0;
        unsigned aux8 = 0;
        copies(kWideWaves, &s8, &r8);
        const size_t lds8 = resolve_variant(ctx, P, false, fit_im, &v8, &aux8, kWideWaves, s8, r8);
        if (v8 == variant && lds8 + kStaticLds + 16 + (size_t)r8 * xrow_bytes <= 160 * 1024) {
            wpb = kWideWaves;
            slices = s8;
            rows = r8;
            lds = lds8;
            aux_off = aux8;
        }
    }
    const int64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks > 0x7fffffffLL) {
        set_error("swarm too large for one launch");
        return NMRFIT_E_INVALID;
    }
    if (lds + kStaticLds + 16 + (size_t)rows * xrow_bytes > 160 * 1024) {
        set_error("too many peaks for the kernel's LDS records (with the imaginary model / the fused swarm update)");
        return NMRFIT_E_UNSUPPORTED;
    }
    // fused swarm update: one copy of the particle's updated row per wave, after everything else
    PsoFused upd{};
    if (fused && fused->x_in) {
        if (dR || (4 + 3 * (int64_t)P) > kFusedMaxD) {
            set_error("fused swarm update: objective launches with D <= kFusedMaxD only");
            return NMRFIT_E_INVALID;
        }
        upd = *fused;
        lds = (lds + 15) & ~(size_t)15;
        upd.xrow_off = (unsigned)lds;
        lds += (size_t)rows * (size_t)(4 + 3 * (int64_t)P) * sizeof(double);
    }
    // nseg == 1: the wave writes f; nseg == 4 (or 8, wide form): the waves of a workgroup are the particle's
    // segments and the workgroup writes f (block sums through LDS); otherwise per-block sums go to a
    // buffer and finalize_kernel (or the swarm's select kernel) adds them.  Same summation order in all.
    const bool direct_f = (nseg == 1 || (kOneWorkgroupParticle && nseg == wpb));
    if (!kOneWorkgroupParticle || nseg != wpb) upd.pbest = 0u;   // (only when ONE workgroup holds the whole particle; else the caller's kernel)
    if (upd.pbest == 0u || !tail_fits(upd.tail, wpb)) upd.tail = 0u;   // (the whole generation in this launch: only on top of the personal bests)
    if (fused && fused->tail != 0u && fused->pending != 0u && upd.tail == 0u) {
        // a generation waits to be folded and this launch cannot do it (the variant or fit_im changed between two
        // generations): nothing is launched, the caller folds in a launch of its own and comes back
        if (!defer) {
            set_error("launch_objective: pending fold without a deferred-launch record");
            return NMRFIT_E_STATE;
        }
        defer->need_flush = true;
        return NMRFIT_OK;
    }
    double *out = df;
    if (!direct_f) {
        int rc = ensure(ctx, &ctx->d_partial, &ctx->cap_partial, S * n_blocks * (fit_im ? 2 : 1));
        if (rc != NMRFIT_OK) return rc;
        out = ctx->d_partial;
    }
    int rc;
    switch (variant) {
        case NMRFIT_VARIANT_BASELINE:
            rc = launch_variant<NMRFIT_VARIANT_BASELINE>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_NOSKIP:
            rc = launch_variant<NMRFIT_VARIANT_NOSKIP>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_SINGLE:
            rc = launch_variant<NMRFIT_VARIANT_SINGLE>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_QUAD:
            rc = launch_variant<NMRFIT_VARIANT_QUAD>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_FARFIELD:
            rc = launch_variant<NMRFIT_VARIANT_FARFIELD>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_NOREC:
            rc = launch_variant<NMRFIT_VARIANT_NOREC>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        case NMRFIT_VARIANT_STAGED:
            rc = launch_variant<NMRFIT_VARIANT_STAGED>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
        default:
            rc = launch_variant<NMRFIT_VARIANT_DEFAULT>(ctx, S, P, dX, out, dR, (int)nseg, seg_len, blk_chunks, blocks, lds, fit_im, upd, aux_off, wpb);
            break;
    }
    if (rc != NMRFIT_OK) return rc;
    if (defer) defer->pbest_done = upd.x_in != nullptr && upd.pbest != 0u;
    if (defer) defer->tail_done = defer->pbest_done && upd.tail != 0u;
    if (!direct_f && defer) {   // the caller's own kernel adds the per-block sums (pso_tail_kernel)
        defer->needed = true;
        defer->partial = ctx->d_partial;
        defer->n_blocks = n_blocks;
        defer->fit_im = fit_im;
    } else if (!direct_f) {
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->d_partial, S, n_blocks, N, fit_im, df);
        NMRFIT_HIP(hipGetLastError());
    }
    ctx->last.waves = waves;
    ctx->last.waves_per_workgroup = wpb;
    ctx->last.nseg = (int32_t)nseg;
    ctx->last.seg_len = seg_len;
    return NMRFIT_OK;
}

int launch_contributions(nmrfit_ctx *ctx, int32_t P, const double *dx, int64_t Nout, const double *d_wc_out,
                         double *d_real, double *d_imag, bool grid_order)
{
    const int64_t n = (int64_t)P * Nout;
    if (n == 0) return NMRFIT_OK;
    hipLaunchKernelGGL(contributions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_wc_out,
                       Nout, dx, (int)P, ctx->w0, ctx->wspan, d_real, d_imag, grid_order ? 1 : 0);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace nmrfit
