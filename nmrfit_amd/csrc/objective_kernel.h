// objective_kernel.h -- the hot path: the batched nmrfit objective / residual kernel for gfx950 (CDNA4), a template over
// (kernel variant, residual rows, imaginary-channel mode, waves per workgroup).  Instantiated by objective_default.hip /
// objective_farfield.hip / objective_norec.hip -- the variants nmrfit_amd.fit() can select -- and, in -DNMRFIT_AB_BUILD
// builds only, by objective_ab.hip (the A/B forms); launched through objective.hip (launch_objective).
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
#pragma once
#include "objective_chunk.h"
#include "objective_launch.h"
#include "swarm_prologue.h"

namespace nmrfit {
namespace {

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
// Registers and occupancy of the selectable kernels (tools/kernel_resources.py, checked by
// tests/test_kernel_resources_cpu.py; figures of the round-6 build): the headline objective kernel <DEFAULT, real part
// only, four waves per workgroup> 127 VGPRs -- FOUR waves per SIMD, no scratch -- with the 8-peak group's 24 constants,
// the 8-point register block and the batch inversion's intermediates live together; FARFIELD 119.
// FIT_IM == 1 evaluates the last peak's dispersion line at the chunk's points in the epilogue (dispersion_points, Dawson
// coefficients from LDS): three waves per SIMD (DEFAULT 131 VGPRs, FARFIELD 142).  FIT_IM == 2 holds eight more
// accumulators and the imaginary far-field sums: the direct kernel 156 VGPRs, the far-field one 162 -- three waves per
// SIMD both, no scratch (round 6: up to round 5 the far-field kernel took the general expansion path with the imaginary
// sum, 190 VGPRs and two waves; it now makes the expansions of two chunks at once there too, real and imaginary).
// Launch bounds (waves per SIMD the compiler must leave room for), by variant and imaginary-channel mode: the direct
// kernels with the imaginary sum run at three waves per SIMD with the bound left at two -- asked for three the compiler
// stops at 160 registers and schedules worse (2.86 against 2.57 ms at C3, round 4); the far-field kernels with the
// imaginary channel keep the bound at two as well (they reach three by themselves: asked for three, the all-peak form
// is 2.32 instead of 2.04 ms).
#ifndef NMRFIT_IM2_FAR_WAVES
#define NMRFIT_IM2_FAR_WAVES 2   // (A/B knob)
#endif
constexpr int objective_min_waves(int variant, int fit_im)
{
    const bool tuned = variant == NMRFIT_VARIANT_DEFAULT || variant == NMRFIT_VARIANT_NOSKIP || variant == NMRFIT_VARIANT_STAGED ||
                       is_farfield(variant) || variant == NMRFIT_VARIANT_NOREC;
    return (fit_im == 2 && is_farfield(variant)) ? NMRFIT_IM2_FAR_WAVES : (fit_im != 0 && is_farfield(variant)) ? 2 : (fit_im == 2) ? 2 : tuned ? kMinWaves : 4;
}
// WAVE_SWARM (device-batched fits, objective_batch.hip): every WAVE holds a whole particle (one segment) and does
// the particle's whole swarm step by itself -- deferred fold, update, evaluation, personal best (swarm_prologue.h).
// pblock: the particle of this workgroup when its waves are the particle's segments (blockIdx.x in a plain launch).
template <int VARIANT, bool WRITE_R, int FIT_IM, int WPB, bool WAVE_SWARM = false>
__device__ __forceinline__ void objective_body(
    unsigned char *lds_raw, const int64_t g, const int64_t pblock,
    const double *__restrict__ wc, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ wt, const double2 *__restrict__ chunk_minmax,
    const double *__restrict__ X, int64_t S, int P, int64_t N, double w0, double wspan, int nseg,
    int64_t seg_len, int blk_chunks, int seg_blocks /* blocks per segment */, int n_blocks_i /* blocks per grid */,
    double lane_step, double rec_devk,
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
    const bool shared = (nseg % WPB == 0);
    const int slice = shared ? 0 : wave;
    // (one copy of every per-peak record per workgroup when its waves share a particle, else one per wave: the
    // dynamic LDS is sized accordingly by resolve_variant -- at C3 that is what lets a fourth workgroup onto a CU)
    const int nslices = shared ? 1 : WPB;
    PeakLor *lor = reinterpret_cast<PeakLor *>(lds_raw) + (size_t)slice * P;
    PeakWin *win = reinterpret_cast<PeakWin *>(lds_raw + (size_t)nslices * P * sizeof(PeakLor)) +
                   (size_t)slice * P;
    constexpr bool kStage = (VARIANT == NMRFIT_VARIANT_STAGED);
    // STAGED: w of the NEXT chunk is requested in the epilogue of the current one, the first chunk's before the
    // prologue's barrier (measured on its own for the selectable kernels in round 4: slower everywhere, not kept)
    // (round 5, the same prefetch alone in the one-particle-per-wave kernels of the device batches, where a chunk of six
    // peaks is short: the eight values cost the direct kernel 56 bytes of scratch per lane at its 128 registers, -5 %)
    constexpr bool kPrefW = kStage;
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
    double *ffs = reinterpret_cast<double *>(lds_tail2) + (size_t)wave * far_stride(FIT_IM);
    // FIT_IM == 2: where the even chunk of a pair parks the odd chunk's 16 coefficient sums (expand_sums).  Without the
    // imaginary sum they wait in slots 16..31 of the scratch's first row; the all-peak imaginary pass uses every row of
    // the scratch for its own expansions, so there they wait in a row of their own behind it.
    // (at ffs + kFarTerms * kFarPad)

    // objective launches of DEFAULT / FARFIELD: per-peak (d, C) of the Gaussian recurrence, after
    // everything else (residual rows are evaluated point by point: they feed finite differences)
    constexpr bool kRec = !WRITE_R && (VARIANT == NMRFIT_VARIANT_DEFAULT || is_farfield(VARIANT));
    unsigned char *grec_base = lds_tail2 +
                        (kStage ? (size_t)WPB * 3 * kChunk * sizeof(double)
                                : (is_farfield(VARIANT) || FIT_IM == 2) ? (size_t)WPB * far_stride(FIT_IM) * sizeof(double) : 0);
    double2 *grec = reinterpret_cast<double2 *>(grec_base) + (size_t)slice * P;

    // DEFAULT: scaled Lorentzian constants for the two-operation pair form, after grec
    constexpr bool kFast = (VARIANT == NMRFIT_VARIANT_DEFAULT);
    PeakFast *lorf = reinterpret_cast<PeakFast *>(grec_base + (kRec ? (size_t)nslices * P * sizeof(double2) : 0)) +
                     (size_t)slice * P;

    // FIT_IM != 0: Dawson table (64 quarter intervals x 11 coefficients for the gathered evaluation, then the 12 of the
    // asymptotic series), one copy per workgroup; the barrier after the staging below makes it visible
    double *dtab = reinterpret_cast<double *>(lds_raw + aux_off);
    if constexpr (FIT_IM != 0)   // kTab[64][11], then kFar[12]
        for (int i = threadIdx.x; i < kDawTabCount; i += WPB * kWave)
            dtab[i] = (i < kDawTabFar) ? (&dawson::kTab[0][0])[i] : dawson::kFar[i - kDawTabFar];
    phase_stamp(clk, 0);
    if (clk && g == 0 && lane == 0) {   // nmrfit_prof_*: ticks of the core clock and of the 100 MHz reference
        clk[0] = __builtin_amdgcn_s_memtime();
        clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    // (no 64-bit division on the way: the GPU has none, a software one is ~80 instructions on the critical path of a
    // wave that may have a single chunk to work on -- the two geometries swarm generations use need none, and the
    // block arithmetic below works from host-computed counts, seg_blocks and n_blocks)
    const bool active = g < S * nseg;
    int64_t particle = 0;
    int seg = 0;
    if (active) {
        if (nseg == 1) {
            particle = g;
        } else if (nseg == WPB) {   // the workgroup is the particle, its waves the segments
            particle = pblock;
            seg = wave;
        } else {
            particle = g / nseg;
            seg = (int)(g - particle * nseg);
        }
    }
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
            // (sums over the aligned groups of eight lanes: ds_swizzle, lane l reads lane l ^ m)
            ehi += __builtin_amdgcn_ds_swizzle(ehi, (1 << 10) | 0x1f);
            elo += __builtin_amdgcn_ds_swizzle(elo, (1 << 10) | 0x1f);
            ehi += __builtin_amdgcn_ds_swizzle(ehi, (2 << 10) | 0x1f);
            elo += __builtin_amdgcn_ds_swizzle(elo, (2 << 10) | 0x1f);
            ehi += __builtin_amdgcn_ds_swizzle(ehi, (4 << 10) | 0x1f);
            elo += __builtin_amdgcn_ds_swizzle(elo, (4 << 10) | 0x1f);
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
    // (WAVE_SWARM: three rows per wave -- the new position, g, the old personal best -- and the old fp behind them)
    double *const xrow = reinterpret_cast<double *>(lds_raw + upd.xrow_off) + (size_t)slice * (WAVE_SWARM ? 3 * D + 2 : D);
    auto stage_row = [&](const int first_pass, const int pass_stride) {
        if (fused)
            stage_peaks(xrow, first_pass, pass_stride);
        else
            stage_peaks(X + particle * D, first_pass, pass_stride);
    };
    if (fused) {
        // swarm generation: the particle moves HERE, in the prologue of the kernel that evaluates it (swarm_prologue.h)
        if constexpr (WAVE_SWARM) {
            if (swarm_prologue_wave(upd, xrow, D, S, particle, active, lane)) return;
        } else {
            if (swarm_prologue<WPB>(upd, xrow, D, S, particle, seg, active, shared, wave, lane, wsums, clk)) return;
        }
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
    if (!shared) {   // (shared prologue: made once per workgroup above; the rotation step comes with the block seeds below)
        double lr, li;
        sincos_fast(p0 + (p1 * (double)lane) * invN, &li, &lr);
        lseed[lane] = lr;
        lseed[kWave + lane] = li;
    }
    {
        const int64_t n_blocks = n_blocks_i;
        const int64_t b0 = (int64_t)seg * seg_blocks;
        const int64_t nb = (seg_blocks < n_blocks - b0) ? seg_blocks : n_blocks - b0;   // blocks of [j0, j1)
        if (shared) {
            if (lane < nb) {
                double er, ei;
                sincos_fast((p1 * (double)((b0 + lane) * blk_len)) * invN, &ei, &er);
                seeds[lane] = make_double2(er, ei);
            }
        } else {
            // ... and, one particle per wave, the rotation step exp(i p1 64/N) in the same call: lane nb (<= 16) takes it.
            // The same function of the same argument as a call of its own: the same bits.
            const double m = (lane < nb) ? (double)((b0 + lane) * blk_len) : 64.0;
            double er, ei;
            sincos_fast((p1 * m) * invN, &ei, &er);
            if (lane < nb) seeds[lane] = make_double2(er, ei);
            const int rl = (nb > 0) ? (int)nb : 0;   // (a segment without blocks: every lane took the step)
            rr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(er), rl), __builtin_amdgcn_readlane(__double2loint(er), rl));
            ri = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ei), rl), __builtin_amdgcn_readlane(__double2loint(ei), rl));
        }
        wave_lds_fence();   // same-wave LDS write -> read
    }
    const int64_t n_blocks = n_blocks_i;
    double bs = 0.0, bs_im = 0.0;             // per-lane sums of squares of the current block
    const int64_t blk0 = (int64_t)seg * seg_blocks;   // global index of this segment's first block
    int cib = 0, bidx = 0;                     // chunk within block, block within segment
    const double base = wave_uniform((double)P * yoff);   // yoff is added once per peak (equations.py:147,195)
    double ss = 0.0, ss_im = 0.0;
    constexpr bool kSkip = (VARIANT != NMRFIT_VARIANT_NOSKIP && VARIANT != NMRFIT_VARIANT_BASELINE);
    constexpr bool kFar = is_farfield(VARIANT);
    constexpr int kGroup = (VARIANT == NMRFIT_VARIANT_SINGLE) ? 1 : (VARIANT == NMRFIT_VARIANT_QUAD) ? 4 : kGroupSize;

    unsigned even_near = 0, even_hits = 0;     // FARFIELD, P <= 32: near-peak and Gaussian-window masks of the even ...
    unsigned pend_near = 0, pend_hits = 0;     // ... and of the odd chunk of the current pair (expand_pair)
    unsigned even_near_im = 0, pend_near_im = 0, pair_far_im = 0;   // FIT_IM == 2, P <= 32: the same for the imaginary model (expand_pair_im)

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
        const double2 mm = global_table(chunk_minmax, jbE / kChunk);
        const bool has_next = jbE + kChunk < j1;                  // wave-uniform
        double2 mn = mm;
        if (has_next) mn = global_table(chunk_minmax, jbE / kChunk + 1);
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
    auto expand_sums = [&](const int park) {   // park: where the second chunk's sums go (doubles from ffs; FIT_IM == 2 only)
        const double *row = ffs + (lane >> 2) * kFarPad + (lane & 3) * 17;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) part += row[j];
        part += __shfl_xor(part, 1, kWave);
        wave_lds_fence();   // reads issued before the sums overwrite row 0
        if constexpr (FIT_IM == 2) {
            if ((lane & 1) == 0) ffs[((lane & 2) ? park : 0) + (lane >> 2)] = part;
        } else {
            if ((lane & 1) == 0) ffs[((lane & 2) << 3) + (lane >> 2)] = part;
        }
        wave_lds_fence();
    };
    // FIT_IM == 2: one lane's 16 coefficients of the far-field form of ITS peak's imaginary line about a chunk centre --
    // al Re(q m^n) by the real two-term recurrence + the Gaussian's asymptotic series expanded binomially (see the
    // imaginary block of the chunk below) -- written to the lane's column of the wave's scratch.  Lanes without a far
    // peak write exact zeros.
    auto im_rows = [&](const bool farim, const double tc, const double hk, const double rq, const double al, const double agd) {
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
    };
    // ... and, far-field kernel with P <= 32, for a PAIR of chunks at once like expand_pair: lanes 0..31 the peaks against
    // chunk jbE, lanes 32..63 against the chunk after it; the sums of the first chunk land in slots 0..15 of the scratch,
    // those of the second in a parking row of their own (behind the real part's), the near-peak masks in scalar registers.
    auto expand_pair_im = [&](const int64_t jbE) {
        const double2 mm = global_table(chunk_minmax, jbE / kChunk);
        const bool has_next = jbE + kChunk < j1;                  // wave-uniform
        double2 mn = mm;
        if (has_next) mn = global_table(chunk_minmax, jbE / kChunk + 1);
        const bool upper = lane >= 32;
        const int k = lane & 31;
        const double lo_w = upper ? mn.x : mm.x, hi_w = upper ? mn.y : mm.y;
        const bool act = (k < P) && (!upper || has_next);
        bool farim = false;
        double tc = 0.0, hk = 0.0, rq = 0.0, al = 0.0, agd = 0.0;
        if (act) {
            const PeakLor rec = lor[k];
            tc = __builtin_fma(0.5 * (lo_w + hi_w), rec.ihw, rec.c);
            hk = (0.5 * (hi_w - lo_w)) * rec.ihw;
            const double den = __builtin_fma(tc, tc, 1.0);
            farim = (den >= 100.0 * hk * hk) && ((fabs(tc) - fabs(hk)) * kSqrtLn2 >= kDawFarX);   // false for NaN
            rq = rcp64(den);
            al = rec.al;
            agd = rec.ag2 * kInvSqrtPi;
        }
        const unsigned long long farmask = __ballot(farim);
        const unsigned long long nearmask = __ballot(act && !farim);
        even_near_im = (unsigned)nearmask;
        pend_near_im = (unsigned)(nearmask >> 32);
        pair_far_im = ((unsigned)farmask != 0u ? 1u : 0u) | ((unsigned)(farmask >> 32) != 0u ? 2u : 0u);
        if (farmask) {   // wave-uniform
            im_rows(farim, tc, hk, rq, al, agd);
            wave_lds_fence();   // same-wave LDS write -> read
            expand_sums(kFarTerms * kFarPad + kFarTerms);
        }
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
            // (grid_slot order: the lane's points 2m, 2m+1 are one 16-byte pair -> global_load_dwordx4)
            const global_pairs wp = lane_ptr(wc + jb, lane);
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const f64x2 d = wp[m * kWave];
                wv[2 * m] = d.x;
                wv[2 * m + 1] = d.y;
            }
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
            if (kSkip) mm = global_table(chunk_minmax, jb / kChunk);
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
                if (P <= 32) {
                    // Half a wave of peaks: the even chunks of a segment work out the expansions
                    // of TWO chunks at once -- lanes 0..31 for this chunk, lanes 32..63 for the
                    // next -- and park the second set (sums in LDS, masks in SGPRs) for the odd
                    // chunk that follows.  Either half runs the same operations in the same
                    // order, so a chunk's coefficients do not depend on which half made them.
                    if (!ff_odd) {
                        expand_pair(jb);
                        expand_sums(kFarTerms * kFarPad);
                    }
                    const unsigned near_c = ff_odd ? pend_near : even_near;
                    const unsigned hits_c = ff_odd ? pend_hits : even_hits;
                    wave_lds_fence();
                    const double *src = ffs + (ff_odd ? (FIT_IM == 2 ? kFarTerms * kFarPad : kFarTerms) : 0);
#pragma unroll
                    for (int n = 0; n < kFarTerms; ++n) cf[n] = src[n];
                    {
                        // The shared polynomial FIRST, straight into the accumulators (the offset P*yoff rides in its constant
                        // term): its 16 coefficients are dead before the near peaks and Gaussians need their registers,
                        // and the accumulators need neither initialising nor a separate add per point.
                        const double ihw1 = (hw > 0.0) ? rcp64(hw) : 0.0;
                        const double c0 = cf[0] + base;
                        if constexpr (VARIANT == NMRFIT_VARIANT_FARFIELD32) {
                            // Mixed precision, opt-in: orders 1..15 of the shared polynomial in PACKED
                            // fp32 -- v_pk_fma_f32, two points per instruction -- the constant term and the last step in
                            // fp64.  What is rounded to fp32 is the VARIATION of the far peaks' tails across the chunk
                            // (rho <= 0.1: first order <= a tenth of their sum), never a near peak, a Gaussian or the
                            // data: f moves by <= 5e-12 relative on the reference-generated goldens and on dense spectra
                            // (profiles/r05/farfield_f32_horner.txt), -5.5 % kernel time at C3.
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            float cff[kFarTerms];
#pragma unroll
                            for (int n = 1; n < kFarTerms; ++n) cff[n] = (float)cf[n];
#pragma unroll
                            for (int q = 0; q < kPointsPerLane; q += 2) {
                                const double u0 = (wv[q] - wcen) * ihw1, u1 = (wv[q + 1] - wcen) * ihw1;
                                const f32x2 uf = {(float)u0, (float)u1};
                                f32x2 pz = {cff[kFarTerms - 1], cff[kFarTerms - 1]};
#pragma unroll
                                for (int n = kFarTerms - 2; n >= 1; --n) {
                                    const f32x2 cn = {cff[n], cff[n]};
                                    pz = __builtin_elementwise_fma(pz, uf, cn);
                                }
                                acc[q] = __builtin_fma((double)pz.x, u0, c0);         // the constant term and the last step: fp64
                                acc[q + 1] = __builtin_fma((double)pz.y, u1, c0);
                            }
                        } else
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
                        for (unsigned m = hits_c; m; m &= m - 1) gauss_add<true>(lor + __builtin_ctz(m), wv, acc);
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
                            gauss_add<true>(lor + k1, wv, acc);
                    }
                }
                // broadcast the kFarTerms sums through LDS and evaluate them at the lane's points
                if ((lane & 3) == 0) ffs[lane >> 2] = csum;
                wave_lds_fence();
#pragma unroll
                for (int n = 0; n < kFarTerms; ++n) cf[n] = ffs[n];
                }
                if (!horner_done) {
                const double ihwc = (hw > 0.0) ? rcp64(hw) : 0.0;
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
                for (unsigned long long m = hits; m; m &= m - 1) gauss_add<kSkip>(lor + kb + __builtin_ctzll(m), wv, acc);
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
            const double2 mi2 = global_table(chunk_minmax, jb / kChunk);
            const double icen = wave_uniform(0.5 * (mi2.x + mi2.y));
            const double ihalf = wave_uniform(0.5 * (mi2.y - mi2.x));
            double isum = 0.0;     // lane l: coefficient of order l >> 2 (all 4 lanes of a quad)
            bool anyfar = false;
            const bool pair_im = kFar && P <= 32;   // (wave-uniform; a compile-time false outside the far-field kernel)
            if (pair_im) {
                // the expansions of this chunk and the next were made together by the even chunk (expand_pair_im)
                if (!ff_odd) expand_pair_im(jb);
                anyfar = (pair_far_im & (ff_odd ? 2u : 1u)) != 0u;
                for (unsigned m = ff_odd ? pend_near_im : even_near_im; m; m &= m - 1) {
                    const PeakLor rec = lor[__builtin_ctz(m)];
#pragma unroll
                    for (int q = 0; q < kPointsPerLane; ++q) iacc[q] += dispersion_tab(wv[q], rec, dtab);
                }
            } else
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
                    im_rows(farim, tc, hk, rq, al, agd);
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
                if (!pair_im) {
                    if ((lane & 3) == 0) ffs[lane >> 2] = isum;
                    wave_lds_fence();
                }
                const double *srci = ffs + ((pair_im && ff_odd) ? kFarTerms * kFarPad + kFarTerms : 0);
                double cfi[kFarTerms];
#pragma unroll
                for (int n = 0; n < kFarTerms; ++n) cfi[n] = srci[n];
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
            const global_pairs up = lane_ptr(u + jb, lane), vp = lane_ptr(v + jb, lane), tp = lane_ptr(wt + jb, lane);
#pragma unroll
            for (int m = 0; m < kPointsPerLane / 2; ++m) {
                const f64x2 du = up[m * kWave], dv = vp[m * kWave], dt = tp[m * kWave];
                uq[2 * m] = du.x;
                uq[2 * m + 1] = du.y;
                vq[2 * m] = dv.x;
                vq[2 * m + 1] = dv.y;
                tq[2 * m] = dt.x;
                tq[2 * m + 1] = dt.y;
            }
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
            } else if (nseg == WPB) {   // the waves of THIS workgroup (four, or eight) hold the whole particle: sums meet in LDS
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
    if constexpr (is_farfield(VARIANT)) {   // chunks alternate even / odd from the segment start
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
                const int64_t part = pblock;
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
        // (no fused personal best here in a plain launch: one wave per particle means >= 16384 particles, where the swarm's
        // own select kernels are noise next to the objective -- and the call cost the headline kernel 12 bytes of scratch)
        if constexpr (WAVE_SWARM && !WRITE_R) {   // device-batched fits: the wave finishes its particle's step
            if (fused) personal_best_wave(upd, xrow, D, S, particle, lane, wave_uniform(f));
        }
    }
    if (nseg == WPB && nseg > 1) {
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
__global__ __launch_bounds__(kWave *WPB, (WPB == kWavesPerBlock) ? objective_min_waves(VARIANT, FIT_IM) : 2) void objective_kernel(
    const double *__restrict__ wc, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ wt, const double2 *__restrict__ chunk_minmax, const double *__restrict__ X, int64_t S,
    int P, int64_t N, double w0, double wspan, int nseg, int64_t seg_len, int blk_chunks, int seg_blocks, int n_blocks,
    double lane_step, double rec_devk, double *__restrict__ out, double *__restrict__ R_out,
    unsigned long long *__restrict__ clk, const PsoFused upd, const unsigned aux_off)
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
    const int64_t g = (int64_t)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    objective_body<VARIANT, WRITE_R, FIT_IM, WPB>(lds_raw, g, (int64_t)blockIdx.x, wc, u, v, wt, chunk_minmax, X, S, P, N, w0, wspan, nseg, seg_len,
                                             blk_chunks, seg_blocks, n_blocks, lane_step, rec_devk, out, R_out, clk, upd, aux_off, wsums);
}

template <int VARIANT>
int launch_variant(const ObjectiveLaunch &a)
{
    nmrfit_ctx *const ctx = a.ctx;
    const int64_t S = a.S, seg_len = a.seg_len, blocks = a.blocks;
    const int32_t P = a.P;
    const double *const dX = a.dX;
    double *const out = a.out, *const dR = a.dR;
    const int nseg = a.nseg, blk_chunks = a.blk_chunks, fit_im = a.fit_im, wpb = a.wpb;
    const size_t lds = a.lds;
    const PsoFused &upd = a.upd;
    const unsigned aux_off = a.aux_off;
#define NMRFIT_LAUNCH_W(WR, FI, W)                                                                              \
    hipLaunchKernelGGL((objective_kernel<VARIANT, WR, FI, W>), dim3((unsigned)blocks), dim3(kWave *(W)), lds,  \
                       ctx->stream, ctx->d_wc, ctx->d_u, ctx->d_v, ctx->d_wt, ctx->d_chunk, dX, S, (int)P,     \
                       ctx->N, ctx->w0, ctx->wspan, nseg, seg_len, blk_chunks, a.seg_blocks, a.n_blocks,      \
                       ctx->lane_step, ctx->grid_dev * 11.0e10, out, dR, clk, upd, aux_off)
#define NMRFIT_LAUNCH(WR, FI) NMRFIT_LAUNCH_W(WR, FI, kWavesPerBlock)
    // nmrfit_prof_enable: HIP events on the launch stream around this kernel alone
    const bool prof = ctx->prof_cap > 0 && ctx->prof_nk < ctx->prof_cap;
    unsigned long long *clk = prof ? ctx->d_clk : nullptr;
    if (prof) NMRFIT_HIP(hipEventRecord(ctx->prof_k0[(size_t)ctx->prof_nk], ctx->stream));
    if constexpr (VARIANT == NMRFIT_VARIANT_FARFIELD32) {   // (objective launches without the imaginary channel only:
        if (dR || fit_im != 0) {                             // launch_objective sends everything else to FARFIELD)
            set_error("internal: FARFIELD32 launch with residual rows or the imaginary channel");
            return NMRFIT_E_STATE;
        }
        if (wpb == kWideWaves)
            NMRFIT_LAUNCH_W(false, 0, kWideWaves);
        else
            NMRFIT_LAUNCH(false, 0);
    } else if (dR) {
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
}  // namespace nmrfit
