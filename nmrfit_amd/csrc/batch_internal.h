// batch_internal.h -- shared by batch.hip (host side of the device-batched fits) and objective_batch.hip (the kernel
// instantiations): the per-fit descriptor the batched objective kernel reads, and the launch record.
#pragma once
#include "objective_launch.h"

namespace nmrfit {

// One fit of a batch, as the kernel sees it: its spectrum (the `args=(w, u, v, weights)` tuple of nmrfit/utils.py:176,
// prepared like a context's arrays: centred, padded, grid_slot order), its peak count and its swarm.  Lives in device
// memory, one table of K records per buffer phase (batch.hip): the kernel finds its record from blockIdx alone.
struct BatchFit {
    const double *wc, *u, *v, *wt;
    const double2 *chunk;
    double w0, wspan, lane_step, rec_devk;
    const double *X;     // positions to evaluate when the launch does not move the swarm (generation 0)
    double *fx;          // objective values of this launch [S]
    int32_t P, pad;
    // the fit's own grid (round 6: the fits of a batch may differ in length -- spectra cropped per dataset,
    // nmrfit/containers.py:112-130 -- in the wave = particle geometry, whose kernel reads these instead of its launch
    // arguments): length, the one segment's length (whole blocks), chunks per block, blocks per grid; and where the fit's
    // points start in the planes of the upload
    int64_t N, seg_len;
    int32_t blk_chunks, n_blocks;
    int64_t raw_off;
    int64_t S;           // the fit's swarm size (round 6: may differ between the fits of a batch, wave = particle geometry)
    PsoFused upd;        // x_in == null: plain evaluation of X
};

struct BatchLaunch {
    hipStream_t stream;
    const BatchFit *fits;   // device, K records
    int32_t K;
    int64_t S;              // particles per fit (the LARGEST swarm of the batch: the wave form reads BatchFit::S)
    int64_t N;              // (the workgroup = particle form: equal for every fit; the wave form reads BatchFit::N)
    int nseg;
    int64_t seg_len;
    int blk_chunks;
    int seg_blocks, n_blocks;   // blocks per segment, per grid
    int wpb;                // 4 or 8 (workgroup = particle), or 4 with one particle per WAVE (wave_swarm)
    bool wave_swarm;
    int64_t blocks_per_fit;
    size_t lds;
    unsigned aux_off;
    int variant;            // NMRFIT_VARIANT_DEFAULT or NMRFIT_VARIANT_FARFIELD
    int fit_im;             // NMRFIT_FIT_IM_* (the imaginary channel: wave = particle form only)
};

int launch_objective_batch(const BatchLaunch &a);      // objective_batch.hip (fit_im = 0)
int launch_objective_batch_im(const BatchLaunch &a);   // objective_batch_im.hip (fit_im = 1; forwards 2)
int launch_objective_batch_im2(const BatchLaunch &a);  // objective_batch_im2.hip (fit_im = 2, DEFAULT)
int launch_objective_batch_im2f(const BatchLaunch &a); // objective_batch_im2f.hip (fit_im = 2, FARFIELD)

}  // namespace nmrfit
