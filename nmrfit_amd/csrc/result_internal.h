// result_internal.h -- the post-fit reconstruction (FitUtility.generate_result, nmrfit/utils.py:226-295) as one kernel
// over a table of jobs: shared by cabi.hip (one fit: nmrfit_contributions, nmrfit_generate_result) and batch.hip (every
// fit of a device batch in one launch: nmrfit_batch_contributions).
#pragma once
#include "nmrfit_internal.h"

namespace nmrfit {

// One fit's reconstruction.  Every pointer is device memory; an output that is null is not produced.
struct ResultJob {
    const double *wc;        // the fit's own centred grid, grid_slot order (w_plain == null), N points
    const double *w_plain;   // or an output grid in plain order, NOT centred (the upsampled np.linspace of utils.py:236), Nout points
    const double *x;         // the parameter vector the lines are built from, 4 + 3 P doubles (utils.py:247-248)
    const double *u, *v;     // the spectrum, grid_slot order, N points (only read for `data`)
    double w0, wspan;        // centring offset of the fit's grid and its span (the clamp of 2/width)
    int64_t Nout, N;
    int32_t P, pad;
    double *real, *imag;     // [P][Nout]: voigt per peak and its Kramers-Kronig partner (utils.py:262-274)
    double *fit;             // [4][Nout]: V_fit, I_fit (utils.py:276-277), u_fit, v_fit (utils.py:284)
    double *data;            // [2][N]: V, I = ps2(u, v, p0, p1), what data.shift_phase(method='manual') stores (utils.py:251)
};

constexpr int kResultThreads = 256;
// `d_jobs`: njobs records in device memory; `max_points`: the largest max(Nout, data ? N : 0) of the table
int launch_result_jobs(hipStream_t stream, const ResultJob *d_jobs, int32_t njobs, int64_t max_points, int32_t Pmax);
int launch_result_one(hipStream_t stream, const ResultJob &job);

}  // namespace nmrfit
