/*
 * c_abi_client.c -- a plain-C caller of libnmrfit_amd.so: no Python, no HIP headers.
 *
 *   gcc -std=c99 -O2 -I include examples/c_abi_client.c -L nmrfit_amd/lib -lnmrfit_amd \
 *       -Wl,-rpath,$PWD/nmrfit_amd/lib -lm -o c_abi_client
 *   ./c_abi_client            full run on GPU 0 (exit 0 = every check passed)
 *   ./c_abi_client --abi      only load the library and print the ABI version (no GPU needed)
 *
 * It builds a three-line spectrum, evaluates a small swarm through nmrfit_objective_batch and
 * nmrfit_residual_batch, checks them against the textbook formulas written out below
 * (reference: nmrfit/equations.py:115-149 voigt, :152-212 objective; nmrfit/proc_autophase.py:9-37
 * ps2), then lets the device-resident swarm (nmrfit_pso_run) fit the spectrum, fits three spectra as one device batch,
 * reconstructs the three fits in one launch (nmrfit_batch_contributions) and runs a batch of spectra of different lengths.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nmrfit_amd.h"
#include "nmrfit_amd_diag.h" /* A/B knobs and diagnostics exercised below; a binding needs nmrfit_amd.h only */

#define N 3000
#define P 3
#define D (4 + 3 * P)
#define S 16

static const double PI = 3.14159265358979323846;

static double voigt(double w, double r, double yoff, double width, double loc, double a)
{
    const double d = w - loc;
    const double L = (2.0 / (PI * width)) / (1.0 + (d / (0.5 * width)) * (d / (0.5 * width)));
    const double G = (2.0 / width) * sqrt(log(2.0) / PI) * exp(-(d / (0.5 * width)) * (d / (0.5 * width)) * log(2.0));
    return yoff + a * (r * L + (1.0 - r) * G);
}

/* objective of one parameter vector, straight from the definitions; residual row optional */
static double objective(const double *x, const double *w, const double *u, const double *v, const double *wt,
                        double *row)
{
    double ss = 0.0;
    for (int j = 0; j < N; ++j) {
        const double phi = x[0] + x[1] * (double)j / (double)N;
        const double vd = u[j] * cos(phi) - v[j] * sin(phi);
        double vf = 0.0;
        for (int k = 0; k < P; ++k) vf += voigt(w[j], x[2], x[3], x[4 + 3 * k], x[5 + 3 * k], x[6 + 3 * k]);
        const double e = wt[j] * (vd - vf);
        if (row) row[j] = e;
        ss += e * e;
    }
    return sqrt(ss / (double)N);
}

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != NMRFIT_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nmrfit_last_error());           \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv)
{
    printf("libnmrfit_amd ABI version %d (header %d)\n", nmrfit_abi_version(), NMRFIT_ABI_VERSION);
    if (nmrfit_abi_version() != NMRFIT_ABI_VERSION) return 1;
    if (argc > 1 && strcmp(argv[1], "--abi") == 0) return 0;

    static double w[N], u[N], v[N], wt[N], X[S * D], f[S], R[2 * N], row[N], lo[D], hi[D];
    const double truth[D] = {0.25, -0.4, 0.6, 0.002, 0.02, 3.2, 0.5, 0.03, 3.5, 0.8, 0.025, 3.75, 0.3};
    /* spectrum = model at `truth`, rotated back by its phase (ps2 with inv=True) */
    for (int j = 0; j < N; ++j) {
        w[j] = 3.0 + (double)j / (double)(N - 1);
        double vf = 0.0;
        for (int k = 0; k < P; ++k) vf += voigt(w[j], truth[2], truth[3], truth[4 + 3 * k], truth[5 + 3 * k], truth[6 + 3 * k]);
        const double phi = truth[0] + truth[1] * (double)j / (double)N;
        u[j] = vf * cos(phi);      /* V = vf, I = 0 rotated by -phi */
        v[j] = -vf * sin(phi);
        wt[j] = 1.0 + 0.5 * sin(0.01 * j);
    }
    for (int d = 0; d < D; ++d) {
        /* the box nmrfit builds around picked peaks (containers.py:195-215): phases, then per peak
         * width and area within a factor, the centre within a fraction of the width */
        const int is_loc = (d >= 4) && ((d - 4) % 3 == 1);
        const double span = (d < 2) ? 0.5 : (d == 3) ? 0.004 : is_loc ? 0.2 * truth[d - 1] : 0.5 * fabs(truth[d]);
        lo[d] = truth[d] - span;
        hi[d] = truth[d] + span;
    }
    lo[2] = 0.0;
    hi[2] = 1.0;
    unsigned long long lcg = 12345;
    for (int i = 0; i < S; ++i)
        for (int d = 0; d < D; ++d) {
            lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
            const double t = (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
            X[i * D + d] = (i == 0) ? truth[d] : lo[d] + t * (hi[d] - lo[d]);
        }

    int ndev = 0;
    CHECK(nmrfit_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no HIP device\n");
        return 3;
    }
    nmrfit_ctx *ctx = NULL;
    CHECK(nmrfit_ctx_create(0, N, w, u, v, wt, &ctx));
    CHECK(nmrfit_objective_batch(ctx, S, P, X, NMRFIT_FIT_IM_OFF, f));
    double worst = 0.0, scale = 0.0;
    for (int j = 0; j < N; ++j) scale = fmax(scale, fabs(u[j]));
    for (int i = 0; i < S; ++i) {   /* relative to f; the truth row (f ~ 0) only carries rounding of the spectrum */
        const double want = objective(X + i * D, w, u, v, wt, NULL);
        const double err = (fabs(f[i] - want) - 1e-13 * scale) / fmax(fabs(want), 1e-300);
        if (err > worst) worst = err;
    }
    printf("objective_batch: %d particles, max relative difference from the formulas %.2e (f[truth] = %.2e)\n", S,
           worst, f[0]);
    if (!(worst <= 1e-9) || !(f[0] < 1e-12)) return 4;

    double f2[2];
    CHECK(nmrfit_residual_batch(ctx, 2, P, X + D, R, f2));
    double worst_r = 0.0;
    for (int b = 0; b < 2; ++b) {
        objective(X + (1 + b) * D, w, u, v, wt, row);
        for (int j = 0; j < N; ++j) worst_r = fmax(worst_r, fabs(R[b * N + j] - row[j]));
        if (fabs(f2[b] - f[1 + b]) > 1e-13 * f[1 + b]) return 5;   /* the rows' own RMS, next to the objective's */
    }
    printf("residual_batch: max absolute difference %.2e\n", worst_r);
    if (!(worst_r <= 1e-12)) return 5;

    /* the swarm, on the device: omega / phip / phig as nmrfit passes them (utils.py:179-181);
     * pyswarm's minstep / minfunc = 1e-8 rule often stops on a tiny early improvement, so the
     * demonstration disables it (negative thresholds never trigger) and runs 1000 generations */
    nmrfit_pso_params prm = {-0.2134, -0.3344, 2.3259, -1.0, -1.0, 7};
    nmrfit_pso *pso = NULL;
    CHECK(nmrfit_pso_create(ctx, 204, 204, 0, P, lo, hi, &prm, &pso));
    CHECK(nmrfit_pso_run(pso, 1000, 100));
    long long it = 0;
    int stop = 0;
    double fg = 0.0, xb[D], fb = 0.0;
    CHECK(nmrfit_pso_status(pso, (int64_t *)&it, &stop, &fg));
    CHECK(nmrfit_pso_best(pso, xb, &fb));
    printf("pso_run: %lld generations, stop code %d, best f = %.3e\n", it, stop, fb);
    const double f0 = objective(xb, w, u, v, wt, NULL);
    if (fabs(f0 - fb) > 1e-9 * fmax(fb, 1e-6) || !(fb < 0.02 * scale)) return 6;
    CHECK(nmrfit_pso_destroy(pso));

    /* the same swarm with the reduction's cross-workgroup hand-over in its other two forms (release /
     * acquire fences; the reduction as its own launch): bit-identical answers, it is an A/B knob */
    for (int mode = NMRFIT_HANDOVER_FENCED; mode <= NMRFIT_HANDOVER_TWO_LAUNCH; ++mode) {
        double xm[D], fm = 0.0;
        CHECK(nmrfit_pso_create(ctx, 204, 204, 0, P, lo, hi, &prm, &pso));
        CHECK(nmrfit_pso_set_handover(pso, mode));
        CHECK(nmrfit_pso_run(pso, 1000, 100));
        CHECK(nmrfit_pso_best(pso, xm, &fm));
        if (fm != fb || memcmp(xm, xb, sizeof xb) != 0) return 9;
        CHECK(nmrfit_pso_destroy(pso));
    }
    if (nmrfit_pso_set_handover(NULL, 0) != NMRFIT_E_INVALID) return 9;
    if (nmrfit_pso_set_fused_pbest(NULL, 0) != NMRFIT_E_INVALID) return 9;
    /* 1024 particles on this 4096-point grid: four segments per particle, so the objective launch also does the
     * personal bests (the default) -- or, switched off, the separate kernel does: the same swarm either way */
    {
        double xa[D], xb1[D], fa = 0.0, fb1 = 0.0;
        for (int fusedpb = 1; fusedpb >= 0; --fusedpb) {
            CHECK(nmrfit_pso_create(ctx, 1024, 1024, 0, P, lo, hi, &prm, &pso));
            CHECK(nmrfit_pso_set_fused_pbest(pso, fusedpb));
            CHECK(nmrfit_pso_run(pso, 200, 100));
            CHECK(nmrfit_pso_best(pso, fusedpb ? xa : xb1, fusedpb ? &fa : &fb1));
            CHECK(nmrfit_pso_destroy(pso));
        }
        printf("1024 particles, personal bests inside / outside the objective launch: best f = %.6e / %.6e\n", fa, fb1);
        if (fa != fb1 || memcmp(xa, xb1, sizeof xa) != 0) return 9;
    }

    /* the multi-GPU form of the same loop, as far as one GPU goes: an RCCL communicator of ONE rank
     * (rank 0 makes the 128-byte id; with more ranks it would travel to them by any means), attached
     * to the swarm, so that every nmrfit_pso_step includes the all-gather of the candidate records.
     * Same seed => the same answer as the plain run above, bit for bit. */
    {
        unsigned char uid[NMRFIT_UNIQUE_ID_BYTES];
        nmrfit_comm *comm = NULL;
        nmrfit_pso *sw = NULL;
        double xb2[D], fb2 = 0.0;
        int32_t rank = -1, nranks = -1, version = 0;
        char what[256], pci[64];
        CHECK(nmrfit_comm_available());            /* every rank asks this first, before anything collective */
        CHECK(nmrfit_device_pci_bus_id(0, pci, (int)sizeof pci));
        CHECK(nmrfit_comm_unique_id(uid));
        CHECK(nmrfit_comm_create(ctx, 0, 1, uid, &comm));
        CHECK(nmrfit_comm_info(comm, &rank, &nranks, &version));
        CHECK(nmrfit_comm_describe(comm, what, (int)sizeof what));
        printf("communicator: %s\n", what);
        if (!strstr(what, pci)) return 8;
        CHECK(nmrfit_comm_barrier(comm));
        CHECK(nmrfit_pso_create(ctx, 204, 204, 0, P, lo, hi, &prm, &sw));
        CHECK(nmrfit_pso_set_comm(sw, comm));
        CHECK(nmrfit_pso_run(sw, 1000, 100));
        CHECK(nmrfit_pso_best(sw, xb2, &fb2));
        printf("pso_run over RCCL %d (rank %d of %d): best f = %.3e\n", version, rank, nranks, fb2);
        if (fb2 != fb || memcmp(xb2, xb, sizeof xb) != 0) return 8;
        /* a communicator in use cannot be destroyed under the swarm's feet */
        if (nmrfit_comm_destroy(comm) != NMRFIT_E_STATE) return 8;
        CHECK(nmrfit_pso_set_comm(sw, NULL));
        CHECK(nmrfit_pso_destroy(sw));
        CHECK(nmrfit_comm_destroy(comm));
    }

    /* Device-batched fits (nmrfit_batch_*): the same spectrum three times with three seeds, ONE launch per generation
     * for all three swarms.  Fit 0 has the lone swarm's seed: its answer is the lone swarm's, bit for bit. */
    {
        enum { K = 3 };
        static double wK[K * N], uK[K * N], vK[K * N], wtK[K * N];
        double loK[K * D], hiK[K * D], xK[K * D], fK[K];
        int32_t PK[K], stopK[K];
        int64_t itK[K];
        nmrfit_pso_params prmK[K];
        for (int k = 0; k < K; ++k) {
            memcpy(wK + k * N, w, sizeof w);
            memcpy(uK + k * N, u, sizeof u);
            memcpy(vK + k * N, v, sizeof v);
            memcpy(wtK + k * N, wt, sizeof wt);
            memcpy(loK + k * D, lo, sizeof lo);
            memcpy(hiK + k * D, hi, sizeof hi);
            PK[k] = P;
            prmK[k] = prm;
            prmK[k].seed = prm.seed + (uint64_t)k;
        }
        nmrfit_batch *batch = NULL;
        /* the kernel variant of the lone swarm above: DEFAULT, unless the test knob NMRFIT_DEFAULT_VARIANT makes the
         * far-field kernel every context's default (the batch names its variant itself) */
        int variant = NMRFIT_VARIANT_DEFAULT;
        const char *knob = getenv("NMRFIT_DEFAULT_VARIANT");
        if (knob && atoi(knob) == NMRFIT_VARIANT_FARFIELD) variant = NMRFIT_VARIANT_FARFIELD;
        CHECK(nmrfit_batch_create(0, K, N, wK, uK, vK, wtK, PK, loK, hiK, 204, prmK, variant, NMRFIT_FIT_IM_OFF, &batch));
        CHECK(nmrfit_batch_run(batch, 1000, 100));
        CHECK(nmrfit_batch_status(batch, itK, stopK, NULL));
        CHECK(nmrfit_batch_best(batch, xK, fK));
        printf("batch of %d fits: %lld generations each, best f = %.3e %.3e %.3e\n", K, (long long)itK[0], fK[0], fK[1], fK[2]);
        if (fK[0] != fb || memcmp(xK, xb, sizeof xb) != 0) return 10;

        /* The reconstruction that follows a fit (FitUtility.generate_result, nmrfit/utils.py:226-295) for all three fits
         * in ONE launch from the batch's resident spectra: per-peak lines, their sums, the fit rotated back to (u, v) and
         * the spectrum rotated by the fitted phase -- against the formulas above and against the one-fit call. */
        static double realK[K * P * N], imagK[K * P * N], fitK[K * 4 * N], dataK[K * 2 * N];
        static double real1[P * N], imag1[P * N], fit1[4 * N], data1[2 * N];
        CHECK(nmrfit_batch_contributions(batch, NULL, NULL, realK, imagK, fitK, dataK));
        CHECK(nmrfit_generate_result(ctx, P, xK, 0, NULL, real1, imag1, fit1, data1));
        if (memcmp(realK, real1, sizeof real1) || memcmp(imagK, imag1, sizeof imag1) || memcmp(fitK, fit1, sizeof fit1) ||
            memcmp(dataK, data1, sizeof data1)) {
            fprintf(stderr, "batched reconstruction differs from the one-fit call\n");
            return 11;
        }
        double worst_r = 0.0, scale = 0.0;
        for (int k = 0; k < K; ++k) {
            const double *xk = xK + k * D;
            for (int j = 0; j < N; j += 7) {
                double vsum = 0.0;
                for (int q = 0; q < P; ++q) {
                    const double want = voigt(w[j], xk[2], xk[3], xk[4 + 3 * q], xk[5 + 3 * q], xk[6 + 3 * q]);
                    worst_r = fmax(worst_r, fabs(realK[((size_t)k * P + q) * N + j] - want));
                    scale = fmax(scale, fabs(want));
                    vsum += realK[((size_t)k * P + q) * N + j];
                }
                if (vsum != fitK[(size_t)k * 4 * N + j]) return 11;                      /* V_fit: the same additions */
                const double phi = xk[0] + xk[1] * (double)j / (double)N;
                const double vd = u[j] * cos(phi) - v[j] * sin(phi);
                worst_r = fmax(worst_r, fabs(dataK[(size_t)k * 2 * N + j] - vd));
                const double vf = fitK[(size_t)k * 4 * N + j], iif = fitK[(size_t)k * 4 * N + N + j];
                worst_r = fmax(worst_r, fabs(fitK[(size_t)k * 4 * N + 2 * N + j] - (vf * cos(phi) + iif * sin(phi))));
            }
        }
        printf("reconstruction of %d fits in one launch: max abs deviation from the formulas %.2e (scale %.2e)\n", K, worst_r, scale);
        if (!(worst_r < 1e-12 * scale)) return 11;
        if (nmrfit_batch_contributions(batch, NULL, w, realK, imagK, NULL, NULL) != NMRFIT_E_INVALID) return 11;   /* grids without lengths */
        CHECK(nmrfit_batch_destroy(batch));
        if (nmrfit_batch_run(NULL, 1, 1) != NMRFIT_E_INVALID) return 10;

        /* Spectra of DIFFERENT lengths in one batch (every dataset is cropped to its own region,
         * nmrfit/containers.py:112-130): the first 3000, 2200 and 1000 points of the spectrum, concatenated.  Fit 0 is
         * again the lone swarm. */
        const int64_t NK[K] = {N, 2200, 1000};
        int64_t at = 0;
        for (int k = 0; k < K; ++k) {
            memcpy(wK + at, w, (size_t)NK[k] * sizeof(double));
            memcpy(uK + at, u, (size_t)NK[k] * sizeof(double));
            memcpy(vK + at, v, (size_t)NK[k] * sizeof(double));
            memcpy(wtK + at, wt, (size_t)NK[k] * sizeof(double));
            at += NK[k];
        }
        const int64_t SK[K] = {204, 96, 51};   /* ... and swarms of different sizes (options['swarmsize'], nmrfit/utils.py:177) */
        CHECK(nmrfit_batch_create_ragged(0, K, NK, wK, uK, vK, wtK, PK, loK, hiK, SK, prmK, variant, NMRFIT_FIT_IM_OFF, &batch));
        CHECK(nmrfit_batch_run(batch, 1000, 100));
        CHECK(nmrfit_batch_best(batch, xK, fK));
        printf("ragged batch (%lld, %lld, %lld points): best f = %.3e %.3e %.3e\n", (long long)NK[0], (long long)NK[1],
               (long long)NK[2], fK[0], fK[1], fK[2]);
        if (fK[0] != fb || memcmp(xK, xb, sizeof xb) != 0) return 12;
        CHECK(nmrfit_batch_destroy(batch));
    }

    /* errors come back as codes, never as crashes */
    if (nmrfit_objective_batch(ctx, S, 1001, X, 0, f) != NMRFIT_E_INVALID) return 7;
    if (nmrfit_objective_batch(NULL, S, P, X, 0, f) != NMRFIT_E_INVALID) return 7;
    CHECK(nmrfit_ctx_destroy(ctx));
    printf("ok\n");
    return 0;
}
