// objective_batch_im2.hip -- the batched objective kernel with the all-peak imaginary model (fit_im = "sum"), DEFAULT
// kernel, wave = particle geometry: split from objective_batch_im.hip so that the two compile side by side.
#include "objective_batch_kernel.h"

namespace nmrfit {

int launch_objective_batch_im2(const BatchLaunch &a)
{
    const dim3 grid((unsigned)(a.blocks_per_fit * a.K));
    hipLaunchKernelGGL((objective_batch_kernel<NMRFIT_VARIANT_DEFAULT, kWavesPerBlock, true, 2>), grid, dim3(kWave * kWavesPerBlock),
                       a.lds, a.stream, a.fits, a.S, (int)a.blocks_per_fit, a.N, a.nseg, a.seg_len, a.blk_chunks, a.seg_blocks,
                       a.n_blocks, a.aux_off);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace nmrfit
