// objective_batch_im2f.hip -- the batched objective kernel with the all-peak imaginary model (fit_im = "sum") in the
// FAR-FIELD form, wave = particle geometry: what nmrfit_amd.fit selects from grid x peaks = 1e5 on since round 6
// (utils.default_variant).  A translation unit of its own: it compiles beside objective_batch_im2.hip.
#include "objective_batch_kernel.h"

namespace nmrfit {

int launch_objective_batch_im2f(const BatchLaunch &a)
{
    const dim3 grid((unsigned)(a.blocks_per_fit * a.K));
    hipLaunchKernelGGL((objective_batch_kernel<NMRFIT_VARIANT_FARFIELD, kWavesPerBlock, true, 2>), grid, dim3(kWave * kWavesPerBlock),
                       a.lds, a.stream, a.fits, a.S, (int)a.blocks_per_fit, a.N, a.nseg, a.seg_len, a.blk_chunks, a.seg_blocks,
                       a.n_blocks, a.aux_off);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace nmrfit
