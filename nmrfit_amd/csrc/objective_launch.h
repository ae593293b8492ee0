// objective_launch.h -- what objective.hip (host side: launch geometry, LDS budget) hands to the translation units
// that hold the kernel instantiations, one per selectable variant so that they compile in parallel.
#pragma once
#include "nmrfit_internal.h"
#include "pso_update.h"

namespace nmrfit {

struct ObjectiveLaunch {
    nmrfit_ctx *ctx;
    int64_t S;
    int32_t P;
    const double *dX;
    double *out;           // f[S], or per-block sums when a particle is spread over several workgroups
    double *dR;            // residual rows (else null)
    int nseg;
    int64_t seg_len;
    int blk_chunks;
    int seg_blocks;        // blocks per segment (seg_len / block length)
    int n_blocks;          // blocks per grid
    int64_t blocks;        // workgroups
    size_t lds;            // dynamic LDS per workgroup
    int fit_im;
    PsoFused upd;
    unsigned aux_off;
    int wpb;               // waves per workgroup (4, or 8: has_eight_wave_form)
};

// Which instantiations exist with eight-wave workgroups (one workgroup = one particle cut into eight segments):
// the objective launches without the imaginary channel of the three kernels fit() can select.
constexpr bool is_farfield(int variant) { return variant == NMRFIT_VARIANT_FARFIELD || variant == NMRFIT_VARIANT_FARFIELD32; }
constexpr bool has_eight_wave_form(int variant)
{
    return variant == NMRFIT_VARIANT_DEFAULT || is_farfield(variant) || variant == NMRFIT_VARIANT_NOREC;
}
constexpr int kWideWaves = 8;
// objective_kernel's own __shared__ block (wsums: 2 x kMaxBlocks block sums + 8 parked values) + alignment slack
constexpr size_t kObjectiveStaticLds = (size_t)(2 * kMaxBlocks + 8) * sizeof(double) + 64;

// The kernel variant a launch actually runs (the requested one may not fit in LDS, or may not implement the imaginary
// part) and the dynamic LDS its records need (objective.hip).  `slices`: copies of the per-peak records in a workgroup
// (1 when its waves are segments of one particle, else wpb); `rows`: row copies kept for a fused swarm generation.
size_t objective_lds(int variant, int32_t P, bool residual, int fit_im, int *variant_out, unsigned *aux_off, int wpb,
                     int slices, int rows);

int launch_objective_default(const ObjectiveLaunch &a);    // objective_default.hip
int launch_objective_farfield(const ObjectiveLaunch &a);   // objective_farfield.hip
int launch_objective_norec(const ObjectiveLaunch &a);      // objective_norec.hip
int launch_objective_farfield32(const ObjectiveLaunch &a); // objective_farfield32.hip (objective launches, fit_im = 0)
#ifdef NMRFIT_AB_BUILD
int launch_objective_ab(int variant, const ObjectiveLaunch &a);   // objective_ab.hip: BASELINE, NOSKIP, SINGLE, QUAD, STAGED
#endif

}  // namespace nmrfit
