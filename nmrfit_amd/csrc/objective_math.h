// objective_math.h -- fp64 building blocks of the objective kernels (gfx950): reciprocal, exp2, sincos, Dawson's
// integral and the dispersion line shapes, wave-level helpers, and the constants of the kernels' LDS layout.
// Included by the kernel translation units (objective_kernel.h) and by objective.hip (host launch + small kernels).
#pragma once
#include "nmrfit_internal.h"
#include "pso_update.h"

#define NMRFIT_DAWSON_QUAL __device__ const
#include "dawson_coeffs.h"

#include <type_traits>

namespace nmrfit {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr double kInvPi = 0.31830988618379067154;
constexpr double kSqrtLn2OverPi = 0.46971863934982566689;   // sqrt(ln2/pi)
// Settled tuning constants (each was an A/B knob in rounds 1-4; the measurements are in profiles/r05/DESIGN_r05.md 4.4 and git history)
constexpr int kInterleave = 4;       // points the scheduler may interleave inside a Lorentzian group (1, 2, 4, 8: +-0.4 % at C3)
constexpr int kGroupSize = 8;        // Lorentzians sharing one reciprocal (6 or 12: +1.2 % / +1.3 %)
constexpr int kBatchInv = 4;         // points sharing one reciprocal in the scaled pair form
constexpr int kMinWaves = 3;         // launch bound of the four-wave objective kernels (they reach four waves per SIMD by themselves)
constexpr int kDispInterleave = 4;   // fit_im=True: points of the last peak's dispersion line in flight together
// objective_kernel's static LDS: block sums (x2 with the imaginary channel); f; then what the end of a fused swarm
// generation needs, parked by the first instructions of the kernel and by its prologue: [+1] personal bests on,
// [+2] p, [+3] S, [+4] the row's LDS offset, [+5] this particle's fp, [+6] fg, [+7] completed generations
constexpr int kWsumsCount = 2 * kMaxBlocks + 8;
constexpr int kFarTerms = 16;      // Taylor terms of the far-field expansion (rho <= 0.1 -> 1e-16)
constexpr int kFarPad = 68;        // row stride (doubles) of the per-wave coefficient scratch in LDS: lane l writes column
                                   // l + l/16 of 16 rows, then reads 16 consecutive doubles of row l/4 from column 17*(l%4) --
                                   // for ds_read_b64 / ds_read2_b64 (32- and 16-lane groups) every lane of a group then hits
                                   // its own bank.  SQ_LDS_BANK_CONFLICT of the far-field kernel is 3.2e6 cycles per C3 launch
                                   // (DEFAULT: 5e4) all the same: 12 cycles per chunk PAIR, from the per-lane reads of the
                                   // 32-byte peak records (lanes i and i + 8 of a ds_read_b128 group share banks) -- 0.3 % of a
                                   // pair's ~4500 cycles, not worth a padded record (profiles/r04/farfield_c3_pmc_summary.json)
// doubles of far-field scratch per wave: the rows -- and, with the all-peak imaginary model, two more rows where the even chunk
// of a pair parks the odd chunk's coefficient sums, real and imaginary (the imaginary pass uses every row of the scratch itself)
constexpr int far_stride(int fit_im) { return kFarTerms * kFarPad + (fit_im == 2 ? 2 * kFarTerms : 0); }
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

// The same, with every coefficient pinned to SCALAR registers at the point of use (s_mov pairs, issued by the scalar
// unit beside the vector work).  For code that sits inside a conditional block of a loop: left to itself the compiler
// hoists the eleven coefficients out of the loop into VECTOR registers -- 22 VGPRs held across the chunk loop (a wave
// per SIMD for the kernels at the 128-register line) and a v_mov + v_fmac pair per Horner step instead of one v_fma
// with a scalar operand.
__device__ __forceinline__ double scalar_const(double x)
{
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ double exp2_neg_sc(double x)
{
    x = fmax(x, -1100.0);
    const double n = __builtin_rint(x);
    const double f = x - n;
    double p = scalar_const(4.455817908336064493e-10);
    p = __builtin_fma(p, f, scalar_const(7.0741942972885210056e-9));
    p = __builtin_fma(p, f, scalar_const(1.0178057087733941105e-7));
    p = __builtin_fma(p, f, scalar_const(1.3215432535912376166e-6));
    p = __builtin_fma(p, f, scalar_const(1.5252733841556772589e-5));
    p = __builtin_fma(p, f, scalar_const(1.5403530463724354209e-4));
    p = __builtin_fma(p, f, scalar_const(1.3333558146406470697e-3));
    p = __builtin_fma(p, f, scalar_const(9.6181291075872566681e-3));
    p = __builtin_fma(p, f, scalar_const(5.5504108664821627039e-2));
    p = __builtin_fma(p, f, scalar_const(2.4022650695910159567e-1));
    p = __builtin_fma(p, f, scalar_const(6.9314718055994530925e-1));
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
// else for the wave to issue.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
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
// The result is typed as what it is, device global memory: for a base that came out of a descriptor table (the batched
// kernel) the compiler cannot tell and would emit flat_load through a 64-bit per-lane address (a v_lshl_add_u64 per
// array and chunk, and a wait on the LDS counter with every load).
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef const f64x2 __attribute__((address_space(1))) *global_pairs;
__device__ __forceinline__ global_pairs lane_ptr(const double *uniform_base, int lane)
{
    unsigned off = (unsigned)lane * 16u;
    asm volatile("" : "+v"(off));
    return reinterpret_cast<global_pairs>(reinterpret_cast<const char __attribute__((address_space(1))) *>((uintptr_t)uniform_base) + (size_t)off);
}
// ... and a wave-uniform table in global memory (the chunks' [min, max] of w): scalar loads
__device__ __forceinline__ double2 global_table(const double2 *p, int64_t i)
{
    const f64x2 d = reinterpret_cast<global_pairs>((uintptr_t)p)[i];
    return make_double2(d.x, d.y);
}

// Sum over the wave's lanes, in lane 0 (the other lanes hold partial sums nobody reads): the tree x[l] += x[l + off],
// off = 32, 16, ..., 1.  Below 32 the partner comes through ds_swizzle (lane l reads lane l ^ off: for the lanes that
// still count, l < off, that IS lane l + off) -- no address arithmetic, where __shfl_down costs a compare, a select
// and a shift-add per step, once per block of the grid (every chunk of a short grid).  Same operands, same order:
// the same bits as the plain shuffle tree.
template <int XOR>
__device__ __forceinline__ double swizzle_xor(double x)
{
    constexpr int pattern = (XOR << 10) | 0x1f;   // bit-mask mode: and 0x1f, or 0, xor XOR (within 32 lanes)
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), pattern);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), pattern);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x)
{
    x += __shfl_down(x, 32, kWave);
    x += swizzle_xor<16>(x);
    x += swizzle_xor<8>(x);
    x += swizzle_xor<4>(x);
    x += swizzle_xor<2>(x);
    x += swizzle_xor<1>(x);
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

// Dawson's integral for the objective's imaginary channel: the same kind of piecewise fits, but one polynomial for
// EVERY quarter interval [k/4, (k+1)/4) of |x| < 16, gathered from a 5.6 KiB table in LDS by a per-lane index -- no
// divergent branches (the lanes of a wave sit in a few different intervals).  Round 6: quarter intervals of degree 10
// (11 FMAs + 11 LDS reads per point, |error| <= 3.3e-16 absolute) instead of unit intervals of degree 18 (19 + 19):
// these evaluations are the largest item of the all-peak imaginary model's cost.  Beyond 16 the asymptotic form (a
// branch almost no wave takes: such peaks are summed through the far-field expansion).
constexpr int kDawTabFar = dawson::kTabIntervals * dawson::kTabCoeffs, kDawTabCount = kDawTabFar + 12;   // kTab[64][11], then kFar[12]
__device__ __forceinline__ double dawson_tab(double x, const double *tab)
{
    const double ax = fabs(x);
    const int k = (int)fmin(4.0 * ax, (double)(dawson::kTabIntervals - 1));   // NaN -> the last interval
    const double t = __builtin_fma(8.0, ax, -(double)(2 * k + 1));
    const double *q = tab + k * dawson::kTabCoeffs;
    double p = q[dawson::kTabCoeffs - 1];
#pragma unroll
    for (int i = dawson::kTabCoeffs - 2; i >= 0; --i) p = __builtin_fma(p, t, q[i]);
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
            if ((q + 1) % kDispInterleave == 0) __builtin_amdgcn_sched_barrier(0);
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

}  // namespace
}  // namespace nmrfit
