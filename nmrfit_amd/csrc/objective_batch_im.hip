// objective_batch_im.hip -- the batched objective kernel with the imaginary channel (fit_im = True: the reference's
// last-peak-only term, nmrfit/equations.py:197-209; "sum": every peak), in the wave = particle geometry (a batch is
// worth making from two fits on, and from three on that form is the faster one anyway).  DEFAULT and FARFIELD for both modes
// (what nmrfit_amd.fit selects: utils.default_variant).  A translation unit of its own: these are the slowest kernels
// to compile (the all-peak sum has a unit of its own, objective_batch_im2.hip).
#include "objective_batch_kernel.h"

namespace nmrfit {
namespace {

template <int VARIANT, int FIT_IM>
int launch_batch_im(const BatchLaunch &a)
{
    const dim3 grid((unsigned)(a.blocks_per_fit * a.K));
    hipLaunchKernelGGL((objective_batch_kernel<VARIANT, kWavesPerBlock, true, FIT_IM>), grid, dim3(kWave * kWavesPerBlock), a.lds,
                       a.stream, a.fits, a.S, (int)a.blocks_per_fit, a.N, a.nseg, a.seg_len, a.blk_chunks, a.seg_blocks,
                       a.n_blocks, a.aux_off);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace

int launch_objective_batch_im(const BatchLaunch &a)
{
    if (a.K <= 0 || a.S <= 0) return NMRFIT_OK;
    if (a.blocks_per_fit * (int64_t)a.K > 0x7fffffffLL || !a.wave_swarm) {
        set_error("batch too large for one launch, or not the wave = particle geometry");
        return NMRFIT_E_INVALID;
    }
    if (a.variant == NMRFIT_VARIANT_DEFAULT && a.fit_im == NMRFIT_FIT_IM_REFERENCE) return launch_batch_im<NMRFIT_VARIANT_DEFAULT, 1>(a);
    if (a.variant == NMRFIT_VARIANT_DEFAULT && a.fit_im == NMRFIT_FIT_IM_SUM) return launch_objective_batch_im2(a);   // objective_batch_im2.hip
    if (a.variant == NMRFIT_VARIANT_FARFIELD && a.fit_im == NMRFIT_FIT_IM_REFERENCE) return launch_batch_im<NMRFIT_VARIANT_FARFIELD, 1>(a);
    if (a.variant == NMRFIT_VARIANT_FARFIELD && a.fit_im == NMRFIT_FIT_IM_SUM) return launch_objective_batch_im2f(a);   // objective_batch_im2f.hip
    set_error("device-batched fits with the imaginary channel: DEFAULT and FARFIELD -- what nmrfit_amd.fit selects");
    return NMRFIT_E_UNSUPPORTED;
}

}  // namespace nmrfit
