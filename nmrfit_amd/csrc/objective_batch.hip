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
// Fits of one batch share the kernel variant and the imaginary-channel mode; peak counts may differ (the dynamic LDS is
// sized for the largest) and, in the wave = particle geometry, so may grid lengths and swarm sizes (round 6).  This unit holds the real-part-only forms, objective_batch_im.hip the others.
#include "objective_batch_kernel.h"

namespace nmrfit {
namespace {

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
