/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product, never linked into
 * libnmrfit_amd.so.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg load the shared object built from this file (oracle/Makefile ->
 * oracle/liboracle.so).
 *
 * Plain-C restatement of the nmrfit objective hot path, scalar loops, float64, with the
 * operation ORDER of the reference's numpy expressions (paths relative to
 * /root/reference):
 *   nmrfit/proc_autophase.py:29-36   ps2      phi_j = p0 + (p1*j)/N ; (u+iv)*exp(i phi)
 *   nmrfit/equations.py:141-147      voigt    L, G, yoff + a*(r*L + (1-r)*G)
 *   nmrfit/equations.py:177-202      objective  sqrt(mean((weights*(V_data - V_fit))^2))
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks it against
 * tests/golden/ vectors produced by the reference itself (oracle/make_golden.py).
 * The reference is Python, so there is no oracle/_ref build.
 *
 * Build: make -C oracle      (gcc -O2 -fopenmp, no -ffast-math: keep IEEE evaluation)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* one particle; R_out may be NULL.  Returns the RMSE. */
static double oracle_one(int64_t N, const double *w, const double *u, const double *v,
                         const double *wt, int P, const double *x, double *R_out)
{
    const double p0 = x[0], p1 = x[1], r = x[2], yoff = x[3];   /* equations.py:177 */
    const double ln2 = log(2.0);
    double ss = 0.0;
    for (int64_t j = 0; j < N; ++j) {
        /* proc_autophase.py:31  p0 + (p1*arange(size)/size) */
        const double phi = p0 + (p1 * (double)j) / (double)N;
        const double c = cos(phi), s = sin(phi);
        /* real part of (c + i s)*(u + i v) */
        const double Vd = c * u[j] - s * v[j];
        double Vf = 0.0;                                        /* equations.py:181 */
        for (int k = 0; k < P; ++k) {                           /* equations.py:188-195 */
            const double width = x[4 + 3 * k], loc = x[5 + 3 * k], a = x[6 + 3 * k];
            const double q = (w[j] - loc) / (0.5 * width);
            const double L = (2.0 / (M_PI * width)) * 1.0 / (1.0 + q * q);          /* :141 */
            const double g = (w[j] - loc) / (width / (2.0 * sqrt(ln2)));
            const double G = (2.0 / width) * sqrt(ln2 / M_PI) * exp(-(g * g));      /* :144 */
            Vf = Vf + (yoff + a * (r * L + (1.0 - r) * G));                         /* :147,:195 */
        }
        const double e = wt[j] * (Vd - Vf);                     /* equations.py:202 */
        if (R_out) R_out[j] = e;
        ss += e * e;
    }
    return sqrt(ss / (double)N);
}

/* f_out[S]; X row-major S x (4+3P).  threads<=1 -> serial. */
int oracle_objective_batch(int64_t N, const double *w, const double *u, const double *v,
                           const double *wt, int64_t S, int P, const double *X,
                           double *f_out, int threads)
{
    const int64_t D = 4 + 3 * (int64_t)P;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < S; ++i)
        f_out[i] = oracle_one(N, w, u, v, wt, P, X + i * D, NULL);
    (void)threads;
    return 0;
}

/* R_out row-major B x N, f_out[B] (may be NULL). */
int oracle_residual_batch(int64_t N, const double *w, const double *u, const double *v,
                          const double *wt, int64_t B, int P, const double *X,
                          double *R_out, double *f_out, int threads)
{
    const int64_t D = 4 + 3 * (int64_t)P;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < B; ++i) {
        double f = oracle_one(N, w, u, v, wt, P, X + i * D, R_out + i * N);
        if (f_out) f_out[i] = f;
    }
    (void)threads;
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
