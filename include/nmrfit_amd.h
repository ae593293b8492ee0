/*
 * nmrfit_amd.h -- C-ABI of libnmrfit_amd.so: the MI355X (gfx950) batched evaluator for
 * nmrfit's objective function and the swarm loop that drives it.
 *
 * The reference (pnnl/nmrfit) is pure Python and has no FFI of its own; the entry points
 * below are what a ctypes binding for the hot path binds (INTEGRATION.md shows the stub a
 * maintainer would add to the reference).  Each entry point cites the reference interface
 * it replaces, as file:line relative to the reference repository root.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No exceptions cross the boundary.
 *   - Every function returns an int status: NMRFIT_OK (0) or a negative NMRFIT_E_* code;
 *     nmrfit_last_error() returns a thread-local message for the last failure.
 *   - float64 everywhere ("double"), arrays contiguous.  Parameter vectors are laid out as
 *     the reference lays them out (nmrfit/equations.py:177,188-192;
 *     nmrfit/containers.py:193-217):
 *         x = [p0, p1, r, yoff, width_1, loc_1, area_1, ..., width_P, loc_P, area_P]
 *     so a swarm is a row-major S x (4 + 3P) matrix.
 *   - Host-pointer calls are synchronous (results are in the output buffer on return).
 *     "_dev" calls take device pointers, enqueue on the context's HIP stream and return
 *     immediately; nmrfit_ctx_synchronize() waits for them.
 *   - A context is bound to one GPU and is NOT thread-safe; different contexts may be
 *     driven from different host threads.  One process per GPU is the intended use.
 *   - There is no CPU fallback: every call fails with NMRFIT_E_NO_DEVICE when no gfx950
 *     device is usable.
 */
#ifndef NMRFIT_AMD_H
#define NMRFIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMRFIT_ABI_VERSION 6

enum {
    NMRFIT_OK = 0,
    NMRFIT_E_INVALID = -1,      /* bad argument (NULL pointer, negative size, P out of range) */
    NMRFIT_E_NO_DEVICE = -2,    /* no usable HIP device / device index out of range         */
    NMRFIT_E_HIP = -3,          /* a HIP runtime call failed; see nmrfit_last_error()       */
    NMRFIT_E_UNSUPPORTED = -4,  /* combination not implemented (e.g. fit_im with an A/B variant) */
    NMRFIT_E_STATE = -5,        /* call sequence error (e.g. pso step before init)          */
    NMRFIT_E_COMM = -6          /* an RCCL call failed; see nmrfit_last_error()             */
};

/* Kernel variants (numerics identical to <= 1e-12 relative).  DEFAULT, FARFIELD and NOREC are in every build of the
 * library; BASELINE, NOSKIP, SINGLE, QUAD and STAGED are A/B forms the tuned kernels are measured and checked
 * against: they exist in libnmrfit_amd_ab.so only (nmrfit_amd/csrc/build.sh --ab), and nmrfit_ctx_set_variant
 * answers NMRFIT_E_UNSUPPORTED for them in the product library. */
enum {
    NMRFIT_VARIANT_DEFAULT = 0,   /* tuned fp64: 8 Lorentzians x 4 points per reciprocal (scaled pair form
                                     for positive amplitudes) + Gaussian window skip (+ Gaussian
                                     recurrence on uniform grids, see NOREC)                          */
    NMRFIT_VARIANT_BASELINE = 1,  /* plain fp64: IEEE divide + libdevice exp2 per unit, no skipping  */
    NMRFIT_VARIANT_NOSKIP = 2,    /* tuned arithmetic, Gaussian evaluated everywhere                 */
    NMRFIT_VARIANT_SINGLE = 3,    /* one reciprocal per unit + Gaussian window skip                  */
    NMRFIT_VARIANT_QUAD = 4,      /* 4 Lorentzians per reciprocal + Gaussian window skip             */
    NMRFIT_VARIANT_STAGED = 5,    /* DEFAULT + LDS-DMA staging of u/v/weights (pays only for P <= 2)  */
    NMRFIT_VARIANT_FARFIELD = 6,  /* opt-in: Lorentzian tails of distant peaks through one shared
                                     Taylor expansion per 512-point chunk (truncation <= 1e-16 of each
                                     term); not the default because it changes the per-unit work      */
    NMRFIT_VARIANT_FARFIELD32 = 8, /* opt-in MIXED PRECISION (ABI 5): FARFIELD with orders 1..15 of its shared polynomial in
                                     packed fp32 (v_pk_fma_f32); everything else -- near peaks, Gaussians, data, the constant
                                     term, every sum -- stays fp64.  f within 5e-12 relative of FARFIELD on the
                                     reference-generated goldens and on dense spectra, -5.5 % kernel time at
                                     4096 x 65536 x 24.  Never selected automatically (narrower than the reference's
                                     float64); objective launches without the imaginary channel only -- residual rows
                                     and fit_im run FARFIELD                                                        */
    NMRFIT_VARIANT_NOREC = 7      /* the general form throughout: 8 Lorentzians per reciprocal, one
                                     reciprocal per point, one exp2 for every in-window Gaussian.
                                     (DEFAULT and FARFIELD objective launches on a uniformly spaced
                                     grid run the in-window Gaussians of a lane's 8 points as a
                                     two-multiply recurrence from one seed; f moves by <= 5e-15.
                                     Residual rows are always evaluated point by point.)                */
};

/* What the objective compares besides the real part (nmrfit/equations.py:197-209).
 * The reference evaluates the imaginary line shape by a Kramers-Kronig quadrature per grid
 * point (equations.py:9-80); this library uses its closed form (Lorentzian dispersion +
 * Dawson's integral), which agrees with the quadrature to the quadrature's tolerance. */
enum {
    NMRFIT_FIT_IM_OFF = 0,     /* fit_im=False: real part only (reference default)                */
    NMRFIT_FIT_IM_REFERENCE = 1, /* fit_im=True exactly as the reference computes it: the imaginary
                                  model is the LAST peak's dispersion only, because
                                  equations.py:199 assigns I_fit instead of accumulating it       */
    NMRFIT_FIT_IM_SUM = 2      /* imaginary model = sum over all peaks (what generate_result,
                                  utils.py:271-276, builds)                                      */
};

typedef struct nmrfit_ctx nmrfit_ctx;
typedef struct nmrfit_pso nmrfit_pso;
typedef struct nmrfit_comm nmrfit_comm;

/* ---- library ------------------------------------------------------------------------- */
int nmrfit_abi_version(void);
const char *nmrfit_last_error(void);
int nmrfit_device_count(int *count);
/* name (<= len-1 chars), compute units, and gcnArchName of a device */
int nmrfit_device_info(int device, char *name, int name_len, int *compute_units, char *arch, int arch_len);

/* ---- context: the per-fit constant arrays ---------------------------------------------
 * Replaces the `args=(data.w, data.u, data.v, weights, fit_im)` tuple that
 * FitUtility.fit hands to pyswarm for every objective call (nmrfit/utils.py:176): the four
 * length-N arrays are copied to the GPU once.  The caller keeps ownership of its host
 * arrays. */
int nmrfit_ctx_create(int device, int64_t N, const double *w, const double *u, const double *v,
                      const double *weights, nmrfit_ctx **out);
int nmrfit_ctx_destroy(nmrfit_ctx *ctx);
/* new weights, same N (FitUtility.fit: weights = ones when dynamic_weighting is False,
 * nmrfit/utils.py:171-173) */
int nmrfit_ctx_set_weights(nmrfit_ctx *ctx, const double *weights);
int nmrfit_ctx_synchronize(nmrfit_ctx *ctx);
int nmrfit_ctx_set_variant(nmrfit_ctx *ctx, int variant);
/* imaginary-part mode (NMRFIT_FIT_IM_*) for the device-pointer objective calls and the swarm */
int nmrfit_ctx_set_fit_im(nmrfit_ctx *ctx, int fit_im);
int nmrfit_ctx_n(const nmrfit_ctx *ctx, int64_t *N);

/* ---- the hot path -----------------------------------------------------------------------
 * nmrfit_objective_batch replaces the per-particle loop
 *     fx[i] = equations.objective(x[i, :], w, u, v, weights, fit_im)
 * (nmrfit/equations.py:152-212, called by pyswarm from nmrfit/utils.py:176-182) by one
 * batched launch: f_out[i] = sqrt(mean_j (weights_j * (V_data_ij - V_fit_ij))^2).
 * fit_im is one of NMRFIT_FIT_IM_*; with it the value is (rmse_real + rmse_imag)/2
 * (equations.py:205-209).  S == 0 is a no-op.  P is limited to 960 peaks (a workgroup keeps its
 * particle's per-peak records in the CU's 160 KiB of LDS); more is NMRFIT_E_INVALID.     */
int nmrfit_objective_batch(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *X, int fit_im,
                           double *f_out);
/* R_out[b*N + j] = weights_j * (V_data_bj - V_fit_bj): the vector inside the mean of
 * nmrfit/equations.py:202.  f_out (may be NULL) receives the matching objective values. */
int nmrfit_residual_batch(nmrfit_ctx *ctx, int64_t B, int32_t P, const double *X, double *R_out,
                          double *f_out);

/* Per-peak contributions of one parameter vector on an output grid -- the building block of
 * FitUtility.generate_result (nmrfit/utils.py:226-295): real_out[k*Nout + j] =
 * voigt(w_out[j]; r, yoff, peak k) (equations.py:115-149, yoff included per peak) and
 * imag_out[k*Nout + j] = its Kramers-Kronig partner (equations.py:52-80) in closed form.
 * w_out == NULL evaluates on the context's own grid (Nout is then ignored and taken as N). */
int nmrfit_contributions(nmrfit_ctx *ctx, int32_t P, const double *x, int64_t Nout, const double *w_out,
                         double *real_out, double *imag_out);
/* The whole of FitUtility.generate_result for one parameter vector (ABI 6; nmrfit/utils.py:226-295), one launch.  Besides
 * the per-peak contributions above (real_out / imag_out: both or neither), any of which may be NULL:
 *   fit_out   4 x Nout: V_fit and I_fit, the contributions summed peak after peak from zero (utils.py:276-277), then
 *             u_fit and v_fit = ps2(V_fit, I_fit, inv=True, p0, p1) (utils.py:284; nmrfit/proc_autophase.py:9-36: the
 *             phase ramp p0 + (p1 j)/Nout runs over the index of the OUTPUT grid)
 *   data_out  2 x N: V, I = ps2(u, v, p0, p1) of the context's spectrum -- what data.shift_phase(method='manual', p0, p1)
 *             stores (utils.py:251; nmrfit/containers.py:68-78)
 * with p0, p1 = x[0], x[1]. */
int nmrfit_generate_result(nmrfit_ctx *ctx, int32_t P, const double *x, int64_t Nout, const double *w_out,
                           double *real_out, double *imag_out, double *fit_out, double *data_out);

/* device-pointer forms (asynchronous on the context's stream) */
int nmrfit_objective_batch_dev(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df_out);
int nmrfit_residual_batch_dev(nmrfit_ctx *ctx, int64_t B, int32_t P, const double *dX, double *dR_out,
                              double *df_out);

/* ---- device memory helpers (so callers need no HIP binding of their own) --------------- */
int nmrfit_dev_alloc(nmrfit_ctx *ctx, int64_t bytes, void **dptr);
int nmrfit_dev_free(nmrfit_ctx *ctx, void *dptr);
int nmrfit_memcpy_h2d(nmrfit_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes);
int nmrfit_memcpy_d2h(nmrfit_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);

/* ---- swarm loop ---------------------------------------------------------------------------
 * Replaces pyswarm.pso as FitUtility.fit calls it (nmrfit/utils.py:176-182): the swarm
 * state (x, v, personal bests) lives on the GPU, one generation is
 * [velocity/position update -> nmrfit_objective_batch_dev -> personal-best update ->
 * local argmin], and only the (f_best, x_best[D]) candidate leaves the device.  The swarm
 * axis shards across ranks: this rank owns particles [offset, offset + S_local) of
 * S_global; random numbers are a counter-based function of (seed, generation, GLOBAL
 * particle index, dimension), so any sharding produces the same swarm. */
typedef struct nmrfit_pso_params {
    double omega, phip, phig;     /* utils.py:179-181 defaults -0.2134, -0.3344, 2.3259 */
    double minstep, minfunc;      /* pyswarm defaults 1e-8, 1e-8                        */
    uint64_t seed;
} nmrfit_pso_params;

int nmrfit_pso_create(nmrfit_ctx *ctx, int64_t S_local, int64_t S_global, int64_t offset, int32_t P,
                      const double *lower, const double *upper, const nmrfit_pso_params *params,
                      nmrfit_pso **out);
int nmrfit_pso_destroy(nmrfit_pso *pso);
/* generation 0: x ~ U(lb,ub), evaluate, personal bests, v ~ U(-|ub-lb|,|ub-lb|);
 * leaves this rank's candidate in the candidate buffer. */
int nmrfit_pso_init(nmrfit_pso *pso);
/* status: iteration counter, stop code (0 running, 1 minfunc, 2 minstep), current fg */
int nmrfit_pso_status(nmrfit_pso *pso, int64_t *iteration, int32_t *stop_code, double *fg);
/* best position (D doubles) and value; after a stop these are pyswarm's return values */
int nmrfit_pso_best(nmrfit_pso *pso, double *x_best, double *f_best);
/* init (if needed) + up to maxiter generations, polling the stop flag every `check_every`
 * generations (generations after a stop are no-ops on the GPU); with a communicator attached
 * every rank makes the same call. */
int nmrfit_pso_run(nmrfit_pso *pso, int64_t maxiter, int32_t check_every);

/* ---- multi-GPU: the candidate exchange (RCCL over xGMI) -------------------------------------
 * Replaces the reference's only parallel mode, `processes=self.processes` handed to pyswarm,
 * which maps the particles of a generation over a multiprocessing.Pool (nmrfit/utils.py:182).
 * One process per GPU; rank q owns particles [offset, offset + S_local) of the swarm and the
 * four grid arrays are replicated.  Per generation ONE collective crosses ranks: an
 * ncclAllGather of each rank's (D+1)-double candidate record on the context's stream,
 * followed by the same deterministic fold on every rank.  librccl is dlopen'ed on first use;
 * no PyTorch is involved.  Bootstrap: rank 0 calls nmrfit_comm_unique_id and hands the 128
 * bytes to the other ranks by any means (the Python side: stdlib sockets,
 * nmrfit_amd/rendezvous.py); then EVERY rank calls nmrfit_comm_create (collective). */
#define NMRFIT_UNIQUE_ID_BYTES 128
/* NMRFIT_OK if RCCL can be loaded (dlopen + every symbol used), NMRFIT_E_UNSUPPORTED with the reason
 * in nmrfit_last_error() otherwise.  Creates nothing and touches no GPU: every rank calls it and the
 * ranks compare notes BEFORE anyone enters the collective nmrfit_comm_create, so that RCCL missing
 * on one node is an error on every rank instead of a hang on the others. */
int nmrfit_comm_available(void);
int nmrfit_comm_unique_id(void *out128);
int nmrfit_comm_create(nmrfit_ctx *ctx, int32_t rank, int32_t nranks, const void *unique_id128,
                       nmrfit_comm **out);
/* NMRFIT_E_STATE while a swarm still has the communicator attached (detach or destroy it first) */
int nmrfit_comm_destroy(nmrfit_comm *comm);
/* rank, size and the RCCL version code (any pointer may be NULL) */
int nmrfit_comm_info(const nmrfit_comm *comm, int32_t *rank, int32_t *nranks, int32_t *rccl_version);
/* broadcast of up to 512 bytes of HOST memory from `root` (the seed of a multi-rank fit: every rank must run the
 * same swarm); synchronous, every rank calls it */
int nmrfit_comm_broadcast_host(nmrfit_comm *comm, void *buf, int64_t bytes, int32_t root);
/* Attach a communicator to a sharded swarm (NULL detaches).  With one attached,
 * nmrfit_pso_step / nmrfit_pso_run include the exchange: every rank of the communicator must
 * make the same calls.  The communicator may have been created on any context of the swarm's DEVICE
 * (NMRFIT_E_INVALID otherwise): the all-gather is enqueued on the swarm's own context's stream, so one
 * ncclCommInitRank serves fit after fit, each with a context of its own -- one swarm at a time. */
int nmrfit_pso_set_comm(nmrfit_pso *pso, nmrfit_comm *comm);
/* One whole generation in one call, no Python in the loop: position update -> objective ->
 * personal bests -> local candidate -> [all-gather over the attached communicator] -> fold with
 * pyswarm's stopping rule.  The first call after nmrfit_pso_init only folds generation 0.
 * Asynchronous on the context's stream. */
int nmrfit_pso_step(nmrfit_pso *pso);


/* ---- device-batched fits: K independent fits, ONE launch per swarm generation for all of them (ABI 5) ---------
 * The reference's users fit spectrum after spectrum -- `nmrfit.fit(data, lb, ub)` per spectrum (nmrfit/core.py:64,
 * README.md:64-66), each a 204-particle swarm (nmrfit/utils.py:177-178) -- and a lone swarm of that size fills a
 * fraction of an MI355X.  A batch holds K spectra (each one the `args=(w, u, v, weights)` tuple of nmrfit/utils.py:176;
 * nmrfit_batch_create: of equal length N, nmrfit_batch_create_ragged: of any lengths) and K swarms (of one size, or of
 * any sizes); peak counts, bounds, seeds and swarm constants are per fit.  Every
 * generation of every swarm that has not stopped is one launch of the objective kernel whose workgroups look their
 * fit up by blockIdx; each fit follows exactly the trajectory nmrfit_pso_run gives it alone (bit-identical params and
 * error for the same seed) and stops by its own pyswarm rule.
 *   w, u, v, weights   K x N, row-major (fit k's arrays at offset k*N)
 *   P                  K peak counts; lower / upper: the K boxes concatenated, sum_k (4 + 3 P[k]) doubles each
 *   params             K records (omega, phip, phig, minstep, minfunc, seed)
 *   variant, fit_im    NMRFIT_VARIANT_DEFAULT or NMRFIT_VARIANT_FARFIELD and NMRFIT_FIT_IM_* for the whole batch (the
 *                      kernels nmrfit_amd.fit selects)
 * nmrfit_batch_run: generation 0 (if needed) + up to maxiter generations, polling the K stop flags every
 * `check_every`; returns when every fit has stopped or maxiter is reached.  nmrfit_batch_status / _best: K values
 * each (any pointer may be NULL); x_best receives the K best positions concatenated like `lower`. */
typedef struct nmrfit_batch nmrfit_batch;
int nmrfit_batch_create(int device, int32_t K, int64_t N, const double *w, const double *u, const double *v,
                        const double *weights, const int32_t *P, const double *lower, const double *upper,
                        int64_t swarmsize, const nmrfit_pso_params *params, int variant, int fit_im,
                        nmrfit_batch **out);
/* The same for spectra of DIFFERENT lengths and swarms of different sizes (ABI 6): the reference's users crop every dataset
 * to its own region (Data.select_bounds, nmrfit/containers.py:112-130), so the spectra of one study rarely share N; and
 * `swarmsize` is an option of every fit (nmrfit/utils.py:177).  N: K grid lengths; w, u, v, weights: the K spectra one
 * after the other (fit k's N[k] points at offset N[0] + ... + N[k-1]); swarmsize: K swarm sizes.  Fits that differ in
 * length or swarm size run in the launch geometry that gives every particle one wave (the wave reads its fit's length,
 * block structure and swarm size from the fit's record; a launch has room for the largest swarm, smaller ones leave
 * their spare workgroups idle); results stay bit-identical to each fit run alone.  nmrfit_batch_create is this call with
 * K equal lengths and sizes. */
int nmrfit_batch_create_ragged(int device, int32_t K, const int64_t *N, const double *w, const double *u, const double *v,
                               const double *weights, const int32_t *P, const double *lower, const double *upper,
                               const int64_t *swarmsize, const nmrfit_pso_params *params, int variant, int fit_im,
                               nmrfit_batch **out);
int nmrfit_batch_destroy(nmrfit_batch *batch);
int nmrfit_batch_run(nmrfit_batch *batch, int64_t maxiter, int32_t check_every);
int nmrfit_batch_status(nmrfit_batch *batch, int64_t *iteration, int32_t *stop_code, double *fg);
int nmrfit_batch_best(nmrfit_batch *batch, double *x_best, double *f_best);
/* FitUtility.generate_result (nmrfit/utils.py:226-295; README.md:64-72: fit -> generate_result) for EVERY fit of the
 * batch at its best position, ONE launch over the batch's resident grids and best rows (ABI 6) -- no context, no upload
 * of spectra per fit.  Outputs as nmrfit_generate_result's, fit after fit; with n_k the output length of fit k:
 *   Nout, w_out   both NULL: every fit on its own grid (n_k = N[k]); else K output lengths and the K output grids one
 *             after the other (the reference's np.linspace(w.min(), w.max(), int(scale * N)), utils.py:236)
 *   real_out, imag_out   for fit 0, then fit 1, ...: P[k] rows of n_k doubles (both or neither)
 *   fit_out   for each fit 4 x n_k (V_fit, I_fit, u_fit, v_fit);  data_out  for each fit 2 x N[k] (V, I of its spectrum)
 * Every value is bit-identical to what nmrfit_generate_result returns for that fit alone. */
int nmrfit_batch_contributions(nmrfit_batch *batch, const int64_t *Nout, const double *w_out, double *real_out,
                               double *imag_out, double *fit_out, double *data_out);

#ifdef __cplusplus
}
#endif
#endif /* NMRFIT_AMD_H */
