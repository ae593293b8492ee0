/*
 * nmrfit_amd_diag.h -- the rest of what libnmrfit_amd.so exports: diagnostics, measurement helpers, A/B knobs and the
 * composable phases of a generation that tests and host-staged exchanges drive one by one.  A binding for the hot
 * path needs none of these (include/nmrfit_amd.h is the product interface); bench.py, tools/ and tests/ do.
 * Same conventions as nmrfit_amd.h.  Results never depend on any knob here: every form is bit-identical.
 */
#ifndef NMRFIT_AMD_DIAG_H
#define NMRFIT_AMD_DIAG_H

#include "nmrfit_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 in libnmrfit_amd_ab.so (built with -DNMRFIT_AB_BUILD: the A/B kernel variants exist), 0 in the product library */
int nmrfit_diag_ab_build(void);

/* ---- devices, contexts -------------------------------------------------------------------------------------- */
/* PCI bus id ("0000:c1:00.0") of a device: what a multi-GPU launch prints per rank so that a
 * failed first contact can be traced to a card (len >= 16) */
int nmrfit_device_pci_bus_id(int device, char *buf, int len);

/* Run this context's launches on an externally owned HIP stream (a hipStream_t passed as
 * void*; NULL restores the context's own stream).  Lets the caller order the swarm kernels
 * with an RCCL collective on the same stream with no host synchronisation. */
int nmrfit_ctx_set_stream(nmrfit_ctx *ctx, void *hip_stream);

/* ---- timing and launch geometry -------------------------------------------------------------------------------- */
/* HIP events recorded on the context's stream; elapsed_ms covers everything enqueued
 * between begin and end (end synchronizes). */
int nmrfit_timer_begin(nmrfit_ctx *ctx);
int nmrfit_timer_end(nmrfit_ctx *ctx, double *elapsed_ms);
/* launch geometry the last objective/residual launch used (for reports) */
int nmrfit_last_launch(const nmrfit_ctx *ctx, int64_t *waves, int32_t *segments, int64_t *segment_len);
/* ... and how many waves its workgroups had (ABI 4): 4, or 8 when a particle cut into eight segments was one
 * eight-wave workgroup (small swarms on short grids -- the reference's default 204 particles, nmrfit/utils.py:177) */
int nmrfit_last_launch_workgroup(const nmrfit_ctx *ctx, int32_t *waves_per_workgroup);

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

/* ---- the phases of a generation, one by one (tests; exchanges staged through the host) --------------------------- */
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

/* How the workgroups of the personal-best / argmin kernel hand their results to the workgroup that finishes the
 * reduction (the swarm loop replacing nmrfit/utils.py:176-182), where a generation is NOT finished inside the objective
 * launch (swarms whose launch geometry does not make a workgroup the particle: the imaginary channel on a small swarm,
 * A/B settings).  Results are bit-identical in every mode.
 *   TWO_LAUNCH (default, round 5)  no hand-over inside a launch: the reduction is its own launch, the kernel boundary
 *                   orders everything -- nothing in the default configuration rests on memory-ordering behaviour
 *                   outside the AMDGPU memory model.  +1.8-2.1 us per generation at 204 particles against FAST, none
 *                   at 1024 (profiles/r05/handover_cost.txt)
 *   FAST            one launch: agent-scope write-through atomic stores ordered by s_waitcnt vmcnt(0), no L2 write-back
 *                   per workgroup; rests on measured gfx950 behaviour (tools/handover_stress.py)
 *   FENCED          one launch: release / acquire fences at agent scope, the textbook form (a fence is an L2
 *                   write-back on this 8-XCD part: 9 us more per generation at 1024 particles)
 * NMRFIT_HANDOVER=fast|fenced in the environment changes the default of swarms created afterwards. */
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

/* ---- communicator bookkeeping ------------------------------------------------------------------------------------- */
/* one line for logs: "rank R of N, HIP device D, PCI 0000:xx:00.0, RCCL V" */
int nmrfit_comm_describe(const nmrfit_comm *comm, char *buf, int len);
/* all-gather of n doubles per rank between device buffers, asynchronous on the context's stream */
int nmrfit_comm_all_gather_dev(nmrfit_comm *comm, const double *d_send, double *d_recv, int64_t n);

/* bookkeeping collectives on HOST values (1..64 doubles; op 0 sum, 1 max, 2 min) and a barrier: what bench.py takes its
 * maximum over ranks with; synchronous, every rank calls them */
int nmrfit_comm_all_reduce_host(nmrfit_comm *comm, double *inout, int32_t n, int32_t op);
int nmrfit_comm_barrier(nmrfit_comm *comm);

/* ---- device-batched fits (nmrfit_batch_*): one generation at a time, geometry A/B, state for tests ------------- */
/* generation 0 on the first call, one generation of every running swarm on each later one (asynchronous) */
int nmrfit_batch_step(nmrfit_batch *batch);
int nmrfit_batch_synchronize(nmrfit_batch *batch);
/* launch geometry: 0 workgroup = particle (4 or 8 waves: the particle's grid segments, what a lone default fit uses),
 * 1 wave = particle (one segment; the wave does the particle's whole step).  Chosen at creation from K x swarmsize;
 * NMRFIT_E_UNSUPPORTED where the shape does not allow the form.  Results are bit-identical either way. */
int nmrfit_batch_set_geometry(nmrfit_batch *batch, int mode);
int nmrfit_batch_geometry(const nmrfit_batch *batch, int32_t *mode, int32_t *waves_per_workgroup, int32_t *segments,
                          int64_t *workgroups);
/* swarm state of fit k (any pointer may be NULL): x, v, p are S x (4 + 3 P[k]); fx, fp are S */
int nmrfit_batch_get_state(nmrfit_batch *batch, int32_t k, double *x, double *v, double *p, double *fx, double *fp);

#ifdef __cplusplus
}
#endif
#endif /* NMRFIT_AMD_DIAG_H */
