// objective_batch_kernel.h -- the objective kernel over K independent fits (see objective_batch.hip), as a template over
// (kernel variant, waves per workgroup, wave = particle, imaginary-channel mode); instantiated by objective_batch.hip
// (fit_im = 0) and objective_batch_im.hip (fit_im = 1, 2).
#pragma once
#include "batch_internal.h"
#include "objective_kernel.h"

namespace nmrfit {
namespace {

// Launch bound: FOUR waves per SIMD for the four-wave forms.  The direct kernel then parks one accumulator and one
// prologue value in scratch (20 bytes per lane: one 8-byte reload + store per chunk, the other once per wave) -- the
// descriptor indirection costs it the four registers the lone kernel has to spare at 127 -- and is 4 % FASTER that
// way than at three waves without scratch (K = 40 ... 100 default fits: 2.36 / 2.27 against 2.45 / 2.38 us per fit
// and generation; profiles/r05/batch_fits_waves_ab.txt): a generation is a few lock-step rounds of short waves, and
// a round of four hides its memory round trips better than a round of three.
#ifndef NMRFIT_BATCH_MIN_WAVES
#define NMRFIT_BATCH_MIN_WAVES 4   // (A/B knob: tools/batch_fits.py with NMRFIT_LIB)
#endif
template <int VARIANT, int WPB, bool WAVE_SWARM, int FIT_IM = 0>
__global__ __launch_bounds__(kWave *WPB, (WPB != kWavesPerBlock) ? 2 : (FIT_IM == 0) ? NMRFIT_BATCH_MIN_WAVES : objective_min_waves(VARIANT, FIT_IM)) void objective_batch_kernel(
    const BatchFit *__restrict__ fits, int64_t S, int blocks_per_fit, int64_t N, int nseg, int64_t seg_len, int blk_chunks,
    int seg_blocks, int n_blocks, const unsigned aux_off)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double wsums[kWsumsCount];
    // (blockIdx-derived: wave-uniform, the record's fields come through scalar loads)
    const int fit = (int)(blockIdx.x / (unsigned)blocks_per_fit);
    const int64_t lblock = (int64_t)(blockIdx.x - (unsigned)fit * (unsigned)blocks_per_fit);
    const BatchFit &d = fits[fit];
    const int P = d.P;
    if constexpr (!WAVE_SWARM) {
        if (threadIdx.x == 0) {   // what the end of a fused generation needs, parked like objective_kernel does
            const bool pbest = d.upd.x_in != nullptr && d.upd.pbest != 0u;
            wsums[2 * kMaxBlocks + 1] = pbest ? 1.0 : 0.0;
            wsums[2 * kMaxBlocks + 2] = __longlong_as_double((long long)(uintptr_t)d.upd.p);
            wsums[2 * kMaxBlocks + 3] = __longlong_as_double((long long)S);
            wsums[2 * kMaxBlocks + 4] = __longlong_as_double((long long)d.upd.xrow_off);
            if (pbest && nseg == WPB) wsums[2 * kMaxBlocks + 5] = d.upd.p[S * (4 + 3 * (int64_t)P) + lblock];
            wsums[2 * kMaxBlocks + 7] = __longlong_as_double((pbest && d.upd.tail != 0u) ? (long long)d.upd.pflip : 0LL);
        }
    }
    const int64_t g = lblock * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (WAVE_SWARM) {
        // one particle per wave, one segment: the grid's length and block structure are the FIT's (scalar loads from its
        // record) -- the fits of a batch may differ in length
        // (... and in swarm size: a fit smaller than the largest leaves the workgroups beyond its last particle idle)
        objective_body<VARIANT, false, FIT_IM, WPB, true>(lds_raw, g, lblock, d.wc, d.u, d.v, d.wt, d.chunk, d.X, d.S, P, d.N, d.w0,
                                                          d.wspan, 1, d.seg_len, d.blk_chunks, d.n_blocks, d.n_blocks, d.lane_step,
                                                          d.rec_devk, d.fx, nullptr, nullptr, d.upd, aux_off, wsums);
    } else {
        objective_body<VARIANT, false, FIT_IM, WPB, false>(lds_raw, g, lblock, d.wc, d.u, d.v, d.wt, d.chunk, d.X, S, P, N, d.w0,
                                                           d.wspan, nseg, seg_len, blk_chunks, seg_blocks, n_blocks, d.lane_step,
                                                           d.rec_devk, d.fx, nullptr, nullptr, d.upd, aux_off, wsums);
    }
}

}  // namespace
}  // namespace nmrfit
