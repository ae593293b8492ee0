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

#define NMRFIT_ABI_VERSION 4

enum {
    NMRFIT_OK = 0,
    NMRFIT_E_INVALID = -1,      /* bad argument (NULL pointer, negative size, P out of range) */
    NMRFIT_E_NO_DEVICE = -2,    /* no usable HIP device / device index out of range         */
    NMRFIT_E_HIP = -3,          /* a HIP runtime call failed; see nmrfit_last_error()       */
    NMRFIT_E_UNSUPPORTED = -4,  /* combination not implemented (e.g. fit_im with an A/B variant) */
    NMRFIT_E_STATE = -5,        /* call sequence error (e.g. pso step before init)          */
    NMRFIT_E_COMM = -6          /* an RCCL call failed; see nmrfit_last_error()             */
};

/* Kernel variants (numerics identical to <= 1e-12 relative; for A/B measurement). */
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
/* PCI bus id ("0000:c1:00.0") of a device: what a multi-GPU launch prints per rank so that a
 * failed first contact can be traced to a card (len >= 16) */
int nmrfit_device_pci_bus_id(int device, char *buf, int len);

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
/* Run this context's launches on an externally owned HIP stream (a hipStream_t passed as
 * void*; NULL restores the context's own stream).  Lets the caller order the swarm kernels
 * with an RCCL collective on the same stream with no host synchronisation. */
int nmrfit_ctx_set_stream(nmrfit_ctx *ctx, void *hip_stream);
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

/* device-pointer forms (asynchronous on the context's stream) */
int nmrfit_objective_batch_dev(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df_out);
int nmrfit_residual_batch_dev(nmrfit_ctx *ctx, int64_t B, int32_t P, const double *dX, double *dR_out,
                              double *df_out);

/* ---- device memory + timing helpers (so callers need no HIP binding of their own) ------ */
int nmrfit_dev_alloc(nmrfit_ctx *ctx, int64_t bytes, void **dptr);
int nmrfit_dev_free(nmrfit_ctx *ctx, void *dptr);
int nmrfit_memcpy_h2d(nmrfit_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes);
int nmrfit_memcpy_d2h(nmrfit_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);
/* HIP events recorded on the context's stream; elapsed_ms covers everything enqueued
 * between begin and end (end synchronizes). */
int nmrfit_timer_begin(nmrfit_ctx *ctx);
int nmrfit_timer_end(nmrfit_ctx *ctx, double *elapsed_ms);
/* launch geometry the last objective/residual launch used (for reports) */
int nmrfit_last_launch(const nmrfit_ctx *ctx, int64_t *waves, int32_t *segments, int64_t *segment_len);
/* ... and how many waves its workgroups had (ABI 4): 4, or 8 when a particle cut into eight segments was one
 * eight-wave workgroup (small swarms on short grids -- the reference's default 204 particles, nmrfit/utils.py:177) */
int nmrfit_last_launch_workgroup(const nmrfit_ctx *ctx, int32_t *waves_per_workgroup);

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
/* one generation on this rank's shard; leaves the local candidate in the buffer */
int nmrfit_pso_step_local(nmrfit_pso *pso);
/* device pointer to this rank's candidate record: (D+1) doubles = [f_best, x_best[0..D)].  The record is
 * the LAST generation's once the work queued on the context's stream up to this call has run (a single-rank
 * swarm folds a generation in the next one's launch, nmrfit_pso_set_fused_tail: this call -- like every entry
 * point that shows or continues the swarm's state -- first enqueues the fold that is still waiting). */
int nmrfit_pso_candidate_dev(nmrfit_pso *pso, double **dptr);
/* make the swarm write its candidate record into caller-owned device memory ((D+1) doubles,
 * e.g. the send buffer of an all-gather); NULL restores the internal buffer */
int nmrfit_pso_set_candidate_dev(nmrfit_pso *pso, double *dptr);
/* fold `nranks` gathered candidate records (device pointer, nranks x (D+1) doubles, rank
 * order) into the global best with pyswarm's rule (lowest rank wins ties) and evaluate
 * the minfunc / minstep stopping tests.  Single-rank callers pass their own candidate. */
int nmrfit_pso_apply_global_dev(nmrfit_pso *pso, const double *d_candidates, int32_t nranks);
/* status: iteration counter, stop code (0 running, 1 minfunc, 2 minstep), current fg */
int nmrfit_pso_status(nmrfit_pso *pso, int64_t *iteration, int32_t *stop_code, double *fg);
/* best position (D doubles) and value; after a stop these are pyswarm's return values */
int nmrfit_pso_best(nmrfit_pso *pso, double *x_best, double *f_best);
/* init (if needed) + up to maxiter generations, polling the stop flag every `check_every`
 * generations (generations after a stop are no-ops on the GPU); with a communicator attached
 * every rank makes the same call. */
int nmrfit_pso_run(nmrfit_pso *pso, int64_t maxiter, int32_t check_every);
/* How the workgroups of the personal-best / argmin kernel hand their results to the workgroup that
 * finishes the reduction (swarms of up to 1024 particles do it inside ONE launch; the swarm loop
 * replacing nmrfit/utils.py:176-182).  Results are bit-identical in every mode.
 *   FAST (default)  agent-scope write-through atomic stores ordered by s_waitcnt vmcnt(0): no L2
 *                   write-back per workgroup (6.7 us instead of 8.8 us per select at 51 workgroups)
 *   FENCED          release / acquire fences at agent scope: the textbook form, kept as the A/B
 *                   reference for FAST (tools/handover_stress.py); NMRFIT_SAFE_HANDOVER=1 in the
 *                   environment makes it the default of every swarm created afterwards
 *   TWO_LAUNCH      no hand-over inside a launch: the reduction is its own launch (what larger
 *                   swarms use anyway)
 * Never switched automatically. */
enum { NMRFIT_HANDOVER_FAST = 0, NMRFIT_HANDOVER_FENCED = 1, NMRFIT_HANDOVER_TWO_LAUNCH = 2 };
int nmrfit_pso_set_handover(nmrfit_pso *pso, int mode);
/* When the launch geometry puts a whole particle into one workgroup (four grid segments per particle: e.g.
 * 512 or 1024 particles x 4096 points, 4096 x 65536), the objective launch also updates the personal bests
 * (pyswarm: where fx < fp: p = x, fp = fx) and a single workgroup finishes the generation (argmin over fp,
 * candidate record, fold) -- no many-workgroup personal-best / argmin kernel at all, nothing handed over
 * inside a launch: 512 x 4096 x 6: 19.5 -> 17.0 us per generation, 1024 x 4096 x 6: 30.2 -> 26.8 us.  On by
 * default; enable = 0 restores the separate kernel (an A/B knob: results are bit-identical either way). */
int nmrfit_pso_set_fused_pbest(nmrfit_pso *pso, int enable);
/* ABI 4.  On top of that, a single-rank swarm of up to 1024 particles (2048 where the workgroup has eight waves) runs a
 * whole generation as ONE launch (what pyswarm.pso's loop body, nmrfit/utils.py:176-182, becomes): the objective launch
 * ends with the personal bests, and the rest -- argmin over fp, candidate record, pyswarm's acceptance / stopping rule --
 * is deferred into the NEXT launch's prologue, where every workgroup works it out for itself before it moves its
 * particle (nothing is handed over inside a launch; the state blocks are double-buffered).  Every entry point that
 * shows or continues the swarm's state first folds the waiting generation in a launch of its own, so the deferral is
 * not observable through this interface (204 x 4096 x 6: 13.4 -> 11.7 us per generation).  On by
 * default; enable = 0 restores the separate one-workgroup launch (an A/B knob: results are bit-identical either
 * way).  nmrfit_pso_last_launches: how many kernel launches the evaluate-and-select part of the last generation
 * took (1, 2 or 3). */
int nmrfit_pso_set_fused_tail(nmrfit_pso *pso, int enable);
int nmrfit_pso_last_launches(const nmrfit_pso *pso, int32_t *launches);
/* copy swarm state to host for inspection/tests (any pointer may be NULL):
 * x, v, p are S_local x D; fx, fp are S_local */
int nmrfit_pso_get_state(nmrfit_pso *pso, double *x, double *v, double *p, double *fx, double *fp);

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
/* one line for logs: "rank R of N, HIP device D, PCI 0000:xx:00.0, RCCL V" */
int nmrfit_comm_describe(const nmrfit_comm *comm, char *buf, int len);
/* all-gather of n doubles per rank between device buffers, asynchronous on the context's stream */
int nmrfit_comm_all_gather_dev(nmrfit_comm *comm, const double *d_send, double *d_recv, int64_t n);
/* bookkeeping collectives on HOST values (1..64 doubles; op 0 sum, 1 max, 2 min), a broadcast
 * of up to 512 bytes from `root`, and a barrier; synchronous, every rank calls them */
int nmrfit_comm_all_reduce_host(nmrfit_comm *comm, double *inout, int32_t n, int32_t op);
int nmrfit_comm_broadcast_host(nmrfit_comm *comm, void *buf, int64_t bytes, int32_t root);
int nmrfit_comm_barrier(nmrfit_comm *comm);
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

/* ---- in-run timing of the hot kernel ----------------------------------------------------------
 * With profiling enabled every objective/residual kernel launch of the context is bracketed by
 * HIP events on the context's stream, and nmrfit_prof_mark records a step boundary; reading
 * synchronizes and returns the per-launch kernel durations and the durations between
 * consecutive marks, both in milliseconds and in launch order.  capacity = the number of
 * launches / marks to keep (0 disables and frees the events).  clock_mhz (may be NULL) is the
 * shader clock the first workgroup of the last profiled objective kernel saw while the chip
 * was loaded (s_memtime ticks per s_memrealtime tick x 100 MHz), 0 if unknown. */
int nmrfit_prof_enable(nmrfit_ctx *ctx, int64_t capacity);
int nmrfit_prof_mark(nmrfit_ctx *ctx);
int nmrfit_prof_read(nmrfit_ctx *ctx, double *kernel_ms, int64_t kernel_cap, int64_t *n_kernel,
                     double *step_ms, int64_t step_cap, int64_t *n_step, double *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* NMRFIT_AMD_H */
