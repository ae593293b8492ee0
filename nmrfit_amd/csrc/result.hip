// result.hip -- the reconstruction that follows a fit (FitUtility.generate_result, nmrfit/utils.py:226-295), on the GPU:
// per-peak real lines (equations.voigt, nmrfit/equations.py:115-149) and their Kramers-Kronig partners in closed form
// (the reference integrates each point numerically, equations.py:9-80), their running sums V_fit / I_fit
// (utils.py:276-277), the fit rotated back to the (u, v) frame (utils.py:284: ps2(V_fit, I_fit, inv=True)) and the
// spectrum rotated by the fitted phase (utils.py:251: data.shift_phase(method='manual'), nmrfit/containers.py:68-78).
//
// One thread per output point walks the peaks of its fit; a workgroup first stages the fit's per-peak constants in LDS
// (one fp64 division per peak and workgroup instead of per point).  The kernel is bound by what it writes -- (2 P + 6)
// doubles per point -- and is launched once for a whole device batch (blockIdx.y = the fit): K x (2 P + 6) x N x 8 B,
// 38 MB for 64 default-size fits, ~10 us of HBM time; the D2H copy that follows is what costs.
#include "objective_math.h"
#include "result_internal.h"

#include <algorithm>

namespace nmrfit {
namespace {

// the per-peak constants the lines are evaluated from (the same as the objective kernels': nmrfit_internal.h, PeakLor)
__device__ __forceinline__ PeakLor peak_record(const double *__restrict__ x, int k, double w0, double wspan)
{
    const double r = x[2];
    const double width = x[4 + 3 * k], loc = x[5 + 3 * k], a = x[6 + 3 * k];
    const double ihw = 2.0 / width;
    const double locc = loc - w0;
    const double lim = 1.0e18 / (wspan + fabs(locc));
    PeakLor rec;
    rec.ihw = (fabs(ihw) > lim) ? copysign(lim, ihw) : ihw;
    rec.c = -locc * rec.ihw;
    rec.al = a * r * ihw * kInvPi;
    rec.ag2 = 2.0 * a * (1.0 - r) * ihw * kSqrtLn2OverPi;
    return rec;
}

__device__ __forceinline__ void result_body(const ResultJob &job, PeakLor *__restrict__ recs)
{
    const int P = job.P;
    const int64_t first = (int64_t)blockIdx.x * kResultThreads;
    const int64_t n_data = job.data ? job.N : 0;
    if (first >= job.Nout && first >= n_data) return;   // (a table's fits may differ in length: whole workgroups leave)
    const double *__restrict__ x = job.x;
    for (int k = threadIdx.x; k < P; k += kResultThreads) recs[k] = peak_record(x, k, job.w0, job.wspan);
    __syncthreads();
    const int64_t j = first + threadIdx.x;
    const double p0 = x[0], p1 = x[1], yoff = x[3];
    if (j < job.Nout) {
        const int64_t Nout = job.Nout;
        const double wj = job.w_plain ? job.w_plain[j] - job.w0 : job.wc[grid_slot(j)];
        // utils.py:262-277: real, imag per peak; V_fit = V_fit + real, I_fit = I_fit + imag, peak after peak from zero
        double V = 0.0, I = 0.0;
        for (int k = 0; k < P; ++k) {
            const PeakLor rec = recs[k];
            const double t = __builtin_fma(wj, rec.ihw, rec.c);
            const double s = __builtin_fma(t, t, 1.0);
            const double re = yoff + __builtin_fma(rec.al, rcp64(s), rec.ag2 * exp2_neg(-s));
            const double im = dispersion(wj, rec);
            if (job.real) job.real[(int64_t)k * Nout + j] = re;
            if (job.imag) job.imag[(int64_t)k * Nout + j] = im;
            V = V + re;
            I = I + im;
        }
        if (job.fit) {
            // proc_autophase.ps2(V_fit, I_fit, inv=True, p0, p1) (proc_autophase.py:29-36): the ramp (p1 * j) / size over
            // the OUTPUT grid's index, rotation by 1 / exp(i phi)
            const double phi = p0 + (p1 * (double)j) / (double)Nout;
            double sn, cs;
            sincos_fast(phi, &sn, &cs);
            job.fit[j] = V;
            job.fit[Nout + j] = I;
            job.fit[2 * Nout + j] = cs * V + sn * I;
            job.fit[3 * Nout + j] = cs * I - sn * V;
        }
    }
    if (j < n_data) {
        // ps2(u, v, p0, p1): V = cos phi u - sin phi v, I = sin phi u + cos phi v, phi_j = p0 + (p1 j) / N
        const int64_t N = job.N;
        const double phi = p0 + (p1 * (double)j) / (double)N;
        double sn, cs;
        sincos_fast(phi, &sn, &cs);
        const int64_t slot = grid_slot(j);
        const double uj = job.u[slot], vj = job.v[slot];
        job.data[j] = cs * uj - sn * vj;
        job.data[N + j] = sn * uj + cs * vj;
    }
}

__global__ __launch_bounds__(kResultThreads) void result_jobs_kernel(const ResultJob *__restrict__ jobs)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    result_body(jobs[blockIdx.y], reinterpret_cast<PeakLor *>(lds_raw));
}

__global__ __launch_bounds__(kResultThreads) void result_one_kernel(const ResultJob job)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    result_body(job, reinterpret_cast<PeakLor *>(lds_raw));
}

}  // namespace

int launch_result_jobs(hipStream_t stream, const ResultJob *d_jobs, int32_t njobs, int64_t max_points, int32_t Pmax)
{
    if (njobs <= 0 || max_points <= 0) return NMRFIT_OK;
    const int64_t tiles = (max_points + kResultThreads - 1) / kResultThreads;
    if (njobs > 65535 || tiles > 0x7fffffffLL) {
        set_error("reconstruction launch: at most 65535 fits and 2^31 tiles");
        return NMRFIT_E_INVALID;
    }
    const size_t lds = (size_t)std::max(Pmax, 1) * sizeof(PeakLor);
    hipLaunchKernelGGL(result_jobs_kernel, dim3((unsigned)tiles, (unsigned)njobs), dim3(kResultThreads), lds, stream, d_jobs);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

int launch_result_one(hipStream_t stream, const ResultJob &job)
{
    const int64_t points = std::max<int64_t>(job.Nout, job.data ? job.N : 0);
    if (points <= 0) return NMRFIT_OK;
    const int64_t tiles = (points + kResultThreads - 1) / kResultThreads;
    if (tiles > 0x7fffffffLL) {
        set_error("reconstruction launch: grid too long");
        return NMRFIT_E_INVALID;
    }
    const size_t lds = (size_t)std::max(job.P, 1) * sizeof(PeakLor);
    hipLaunchKernelGGL(result_one_kernel, dim3((unsigned)tiles), dim3(kResultThreads), lds, stream, job);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

}  // namespace nmrfit
