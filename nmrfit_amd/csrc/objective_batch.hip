// objective_batch.hip -- the objective kernel launched over K INDEPENDENT fits at once (device-batched fits, batch.hip).
//
// The reference's users fit spectrum after spectrum (`nmrfit.fit` per spectrum, nmrfit/core.py:64, README.md:64-66) with
// 204 particles each (nmrfit/utils.py:177): one such swarm fills a fraction of an MI355X, and its generation is bound
// by one wave's critical path, not by throughput.  Here workgroup b of ONE launch belongs to fit b / blocks_per_fit:
// it reads that fit's record (BatchFit: the spectrum's arrays, its peak count, its swarm) and runs the same
// objective_body as a lone fit's launch -- same operations, same canonical summation order, so every fit's trajectory is
// bit-identical to the one `nmrfit_amd.fit` takes alone.  Two geometries:
//   workgroup = particle  its 4 or 8 waves are the particle's grid segments (what a lone default fit uses): few fits
//   wave = particle       one segment, the wave does the particle's whole step (swarm_prologue_wave): many fits -- one
//                         prologue per 8 chunks instead of per 1, no idle waves while wave 0 folds and moves
// Fits of one batch share N, the swarm size, the kernel variant and fit_im = 0; peak counts may differ (the dynamic
// LDS is sized for the largest).
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
template <int VARIANT, int WPB, bool WAVE_SWARM>
__global__ __launch_bounds__(kWave *WPB, (WPB == kWavesPerBlock) ? NMRFIT_BATCH_MIN_WAVES : 2) void objective_batch_kernel(
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
    objective_body<VARIANT, false, 0, WPB, WAVE_SWARM>(lds_raw, g, lblock, d.wc, d.u, d.v, d.wt, d.chunk, d.X, S, P, N, d.w0,
                                                      d.wspan, nseg, seg_len, blk_chunks, seg_blocks, n_blocks, d.lane_step, d.rec_devk, d.fx,
                                                      nullptr, nullptr, d.upd, aux_off, wsums);
}

template <int VARIANT>
int launch_batch_variant(const BatchLaunch &a)
{
    const dim3 grid((unsigned)(a.blocks_per_fit * a.K));
#define NMRFIT_BATCH_LAUNCH(W, WS)                                                                                      \
    hipLaunchKernelGGL((objective_batch_kernel<VARIANT, W, WS>), grid, dim3(kWave *(W)), a.lds, a.stream, a.fits, a.S,    \
                       (int)a.blocks_per_fit, a.N, a.nseg, a.seg_len, a.blk_chunks, a.seg_blocks, a.n_blocks, a.aux_off)
    if (a.wave_swarm)
        NMRFIT_BATCH_LAUNCH(kWavesPerBlock, true);
    else if (a.wpb == kWideWaves)
        NMRFIT_BATCH_LAUNCH(kWideWaves, false);
    else
        NMRFIT_BATCH_LAUNCH(kWavesPerBlock, false);
#undef NMRFIT_BATCH_LAUNCH
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace

int launch_objective_batch(const BatchLaunch &a)
{
    if (a.K <= 0 || a.S <= 0) return NMRFIT_OK;
    if (a.blocks_per_fit * (int64_t)a.K > 0x7fffffffLL) {
        set_error("batch too large for one launch");
        return NMRFIT_E_INVALID;
    }
    switch (a.variant) {
        case NMRFIT_VARIANT_DEFAULT: return launch_batch_variant<NMRFIT_VARIANT_DEFAULT>(a);
        case NMRFIT_VARIANT_FARFIELD: return launch_batch_variant<NMRFIT_VARIANT_FARFIELD>(a);
        default: break;
    }
    set_error("device-batched fits run the DEFAULT and FARFIELD kernels (what nmrfit_amd.fit selects by problem size)");
    return NMRFIT_E_UNSUPPORTED;
}

}  // namespace nmrfit
