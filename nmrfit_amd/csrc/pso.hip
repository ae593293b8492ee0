// pso.hip -- device-resident particle-swarm loop around the batched objective.
//
// Replaces pyswarm.pso as nmrfit calls it (nmrfit/utils.py:176-182; pyswarm is a third-party
// dependency that is not vendored in the reference, github.com/tisimst/pyswarm master,
// pso.py -- its published algorithm is restated here).  pyswarm evaluates one particle per
// Python call; here the whole swarm state stays in HBM and a generation is these phases:
//
//   update   v = omega*v + phip*rp*(p-x) + phig*rg*(g-x);  x = clip(x+v, lb, ub)
//   evaluate fx = objective_batch(x)                        (objective.hip)
//   pbest    where fx < fp:  p = x, fp = fx
//   argmin   candidate = (min fp, p[argmin])  of THIS rank's shard   -> (D+1) doubles
//   apply    fold the gathered candidates of all ranks into (g, fg) with pyswarm's
//            minfunc/minstep stopping rule; every rank computes the same answer
//
// Only the candidate record crosses ranks (one RCCL all-gather per generation, done by the
// caller on the candidate buffer).  Random numbers are Philox4x32-10 keyed by the seed with
// counter (generation, dimension, GLOBAL particle index): the swarm's trajectory does not
// depend on how it is sharded.  After a stop is flagged on the device every later launch is
// a no-op, so the host may poll the flag every k generations without changing the result.
//
// Every launch costs ~4.8 us on this part however little it does, so the phases are packed
// into as few launches as the data dependences allow:
//   one workgroup = one particle (its four or eight grid segments are the workgroup's waves: the reference's
//       default 204 x 4096 x 6, 1024 x 4096 x 6, 4096 x 65536 x 24): the objective launch does the particle's
//       WHOLE step -- update, evaluate, personal best.  Single rank and up to 1024 (eight-wave workgroups: 2048)
//       particles, the rest of the generation (argmin over fp, candidate record, fold) is DEFERRED into the next
//       objective launch's prologue, every workgroup for itself (pso_update.h, PsoFused::tail)  -> 1 launch
//       otherwise ONE workgroup finishes in its own launch (pso_tail_kernel)       -> 2 launches
//   S*D <= 2560 in any other geometry: objective + ONE single-workgroup kernel for everything else
//       (pso_tail_kernel)                                                          -> 2 launches
//   otherwise, S <= 1024: objective (with the position update in its prologue), pso_select_kernel (the
//       objective's block sums + pbest + argmin [+ apply] in a many-workgroup kernel finished by
//       its last-ticket workgroup)                                                -> 2 launches
//   larger:  the same with the final reduction as its own single-workgroup launch -> 3 launches
// Multi-rank generations run apply as its own launch after the all-gather.
#include "nmrfit_internal.h"
#include "nmrfit_amd_diag.h"
#include "pso_update.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

struct nmrfit_pso {
    nmrfit_ctx *ctx = nullptr;
    int64_t S = 0, S_global = 0, offset = 0;
    int32_t P = 0;
    int64_t D = 0;
    nmrfit_pso_params prm{};
    void *d_block = nullptr;           // the one allocation behind every pointer below
    double *d_lb = nullptr, *d_ub = nullptr;
    double *d_x = nullptr, *d_v = nullptr, *d_p = nullptr;
    double *d_x2 = nullptr, *d_v2 = nullptr;   // the other half of the x / v ping-pong (fused update: the objective
                                               // kernel reads one pair and writes the other, then they swap)
    double *d_fx = nullptr, *d_fp = nullptr;
    double *d_cand = nullptr;          // (D+1): f_best, x_best (own buffer or caller's)
    double *d_cand_own = nullptr;
    long long *d_flags = nullptr;      // [0] completed generations, [1] stop code
    double *d_best = nullptr;          // [0] fg, [1] best_f, [2..2+D) g, [2+D..2+2D) best_x
    double *d_part_val = nullptr;      // pso_select_kernel: per-workgroup (min fp, index) posts
    long long *d_part_idx = nullptr;
    unsigned *d_ticket = nullptr;
    int handover = NMRFIT_HANDOVER_TWO_LAUNCH;   // nmrfit_pso_set_handover: how the select kernel's workgroups hand over
                                                 // (round 5 default: not inside a launch at all -- see the comment below)
    bool fused_pbest = true;               // nmrfit_pso_set_fused_pbest: personal bests inside the objective launch
    bool fused_tail = true;                // ... and the rest of a single-rank generation, as a fold deferred into the next
                                           // objective launch's prologue (PsoFused::tail; NMRFIT_NO_FUSED_TAIL: A/B knob)
    double *d_best_alt = nullptr;          // the other (g, fg, best | flags) block of the deferred form's double buffer
    double *d_p_alt = nullptr;             // ... and the other (p, fp) buffer
    long long *d_flags_alt = nullptr;
    bool fold_pending = false;             // deferred form: the last launch's personal bests have not been folded yet
    int last_launches = 0;                 // kernel launches of the last generation's evaluate-and-select (diagnostics)
    nmrfit_comm *comm = nullptr;       // attached communicator (sharded swarm): the exchange runs inside nmrfit_pso_step
    bool initialized = false;    // nmrfit_pso_init has run
    bool seeded = false;         // the generation-0 candidates have been folded into (g, fg)
};

namespace nmrfit {
namespace {

// The swarm arithmetic is written without fused multiply-add so that it is bit-identical to
// the numpy mirror in nmrfit_amd/pso.py (IEEE mul/add in numpy's evaluation order).
#pragma clang fp contract(off)

__global__ void pso_init_kernel(int64_t S, int64_t D, int64_t offset, uint64_t seed, const double *__restrict__ lb,
                                const double *__restrict__ ub, double *__restrict__ x, double *__restrict__ v,
                                double *__restrict__ p, double *__restrict__ fp)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    const int64_t i = idx / D;
    const int d = (int)(idx - i * D);
    double r0, r1;
    uniform2(seed, 0u, (uint32_t)d, (uint64_t)(offset + i), &r0, &r1);
    const double lo = lb[d], hi = ub[d];
    const double vhigh = fabs(hi - lo), vlow = -vhigh;
    x[idx] = lo + r0 * (hi - lo);
    v[idx] = vlow + r1 * (vhigh - vlow);
    p[idx] = 0.0;
    if (d == 0) fp[i] = INFINITY;
}

// ---- device bodies (shared by the stand-alone kernels and the fused tail kernel) -----------

// velocity / position update of element idx of the S x D swarm for generation gen
__device__ __forceinline__ void update_element(int64_t idx, int64_t D, int64_t offset, uint64_t seed, uint32_t gen,
                                               double omega, double phip, double phig, const double *best,
                                               const double *lb, const double *ub, const double *p, double *x,
                                               double *v)
{
    const int64_t i = idx / D;
    const int d = (int)(idx - i * D);
    double rp, rg, vn;
    uniform2(seed, gen, (uint32_t)d, (uint64_t)(offset + i), &rp, &rg);
    const double xn = update_value(x[idx], v[idx], p[idx], best[2 + d], lb[d], ub[d], rp, rg, omega, phip, phig, &vn);
    v[idx] = vn;
    x[idx] = xn;
}

// (pbest_particle, argmin_block: pso_update.h -- shared with the device-batched fits, batch.hip)

// ---- stand-alone kernels (large swarms: one launch per phase, many workgroups) ------------

__global__ void pso_update_kernel(int64_t S, int64_t D, int64_t offset, uint64_t seed, double omega, double phip,
                                  double phig, const long long *__restrict__ flags, const double *__restrict__ best,
                                  const double *__restrict__ lb, const double *__restrict__ ub,
                                  const double *__restrict__ p, double *__restrict__ x, double *__restrict__ v)
{
    if (flags[1] != 0) return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    update_element(idx, D, offset, seed, (uint32_t)(flags[0] + 1), omega, phip, phig, best, lb, ub, p, x, v);
}

__global__ __launch_bounds__(1024) void pso_argmin_kernel(int64_t S, int64_t D, const long long *__restrict__ flags,
                                                          const double *__restrict__ fp, const double *__restrict__ p,
                                                          const double *__restrict__ x, double *__restrict__ cand)
{
    if (flags[1] != 0) return;
    __shared__ double s_val[16];
    __shared__ long long s_idx[16];
    argmin_block(S, D, fp, p, x, cand, s_val, s_idx);
}

__global__ void pso_apply_kernel(int64_t D, int nranks, int is_init, double minstep, double minfunc,
                                 const double *__restrict__ cands, long long *__restrict__ flags,
                                 double *__restrict__ best)
{
    if (flags[1] != 0) return;
    apply_wave(threadIdx.x, D, nranks, is_init, minstep, minfunc, cands, flags, best);
}

// ---- fused tail for small swarms (S*D <= tail_max_elems()): ONE workgroup runs every phase that
// follows the objective launch, separated by workgroup barriers (all of it is tiny: a
// 204 x 22 swarm is 4488 elements).  For a launch-bound small swarm -- the reference's default
// is 204 particles -- this turns six launches per generation into two.
enum { kTailFinalize = 1, kTailPbest = 2, kTailArgmin = 4, kTailApply = 8, kTailUpdate = 16 };

struct TailArgs {
    int64_t S, D, offset, N, n_blocks;
    uint64_t seed;
    double omega, phip, phig, minstep, minfunc;
    int phases, fit_im, nranks, is_init;
    int fenced;                  // last-ticket select: release / acquire fences instead of the fence-free hand-over
    const double *partial;       // per-block sums of squares from the objective launch (kTailFinalize)
    const double *lb, *ub;
    const double *cands;         // candidate records to fold (kTailApply)
    double *x, *v, *p, *fx, *fp, *cand, *best;
    long long *flags;
};

__global__ __launch_bounds__(1024) void pso_tail_kernel(TailArgs a)
{
    if (a.flags[1] != 0) return;
    __shared__ double s_val[16];
    __shared__ long long s_idx[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nw = blockDim.x / kWave;
    if (a.phases & kTailFinalize) {   // same arithmetic and order as finalize_kernel
        for (int64_t i = threadIdx.x; i < a.S; i += blockDim.x)
            a.fx[i] = finalize_value(a.partial + i * a.n_blocks * (a.fit_im ? 2 : 1), a.n_blocks, a.N, a.fit_im);
        __syncthreads();
    }
    if (a.phases & kTailPbest) {
        for (int64_t i = wave; i < a.S; i += nw) pbest_particle(i, lane, a.D, a.x, a.fx, a.p, a.fp);
        __syncthreads();
    }
    if (a.phases & kTailArgmin) {
        argmin_block(a.S, a.D, a.fp, a.p, a.x, a.cand, s_val, s_idx);
        __syncthreads();
    }
    if (a.phases & kTailApply) {
        if (wave == 0) apply_wave(lane, a.D, a.nranks, a.is_init, a.minstep, a.minfunc, a.cands, a.flags, a.best);
        __syncthreads();
    }
    if ((a.phases & kTailUpdate) && a.flags[1] == 0) {
        const uint32_t gen = (uint32_t)(a.flags[0] + 1);
        for (int64_t idx = threadIdx.x; idx < a.S * a.D; idx += blockDim.x)
            update_element(idx, a.D, a.offset, a.seed, gen, a.omega, a.phip, a.phig, a.best, a.lb, a.ub, a.p, a.x,
                           a.v);
    }
}


// ---- large swarms: everything between the objective launch and the next position update in
// ONE many-workgroup kernel.  Every launch costs ~4.8 us on this part however little it does
// (measured: profiles/r01/small_swarm_kernel_trace.txt), so finalize + pbest + argmin (+ apply)
// as four launches cost more than a 1024 x 4096 x 6 objective.  One wave per particle adds the
// block sums of its objective (finalize), updates the personal best, and the workgroup posts
// its (fp, index) minimum; the workgroup that draws the last ticket -- all others have fenced
// their writes by then -- reduces the posted minima, writes the candidate record and, for a
// single-rank run, folds it into (g, fg) with the stopping rule.
constexpr int kSelectWaves = 4;          // particles per workgroup pass
constexpr int kSelectTicketBlocks = 256;  // up to this many workgroups the last-ticket form (one launch) beats two launches
constexpr int kSelectMaxPosts = 65536;    // posts buffer; larger swarms: the workgroups stride over the particles

// How the workgroups of the last-ticket form hand their posts and personal-best rows to the
// workgroup that finishes (nmrfit_pso_set_handover, include/nmrfit_amd.h):
//   FAST    agent-scope relaxed atomic stores (write-through: coherent across the 8 XCDs by
//           themselves), each group of stores completed with s_waitcnt vmcnt(0) before the next is
//           issued (value before tag/ticket); the reader uses agent-scope atomic loads.  No release
//           fence -- on this part a fence is an L2 write-back, and the fences of one launch's
//           workgroups serialise (tools/archive/barrier_probe.hip: 8.0 us per exchange at 51 workgroups
//           against 2.1).  Compiler ordering is pinned on both sides (the asm carries a memory
//           clobber; __atomic_signal_fence around the ticket); the hardware ordering rests on
//           measured gfx950 behaviour and is what tools/handover_stress.py is for.
//   FENCED  the textbook form: plain stores, every writing wave issues an agent-scope fence, the
//           ticket is an acq_rel read-modify-write, the finishing workgroup fences before it reads.
//           A/B reference for FAST, never chosen automatically.
//   TWO_LAUNCH  no hand-over inside a launch at all: posts, then pso_select_final_kernel as its own
//           launch (the kernel boundary orders everything).  What swarms above 1024 particles use -- and, from
//           round 5, the DEFAULT everywhere: the shapes nmrfit_amd.fit() spends its time in finish a generation
//           inside the objective launch (deferred fold) and never come here; what still does (the imaginary
//           channel on a small swarm, personal bests switched off) pays 1.8-2.1 us per generation for an
//           ordering that rests on the kernel boundary instead of on measured hardware behaviour
//           (204 x 4096 x 6 with fit_im: 21.5 -> 23.3 us; 1024 particles: the same or faster;
//           profiles/r05/handover_cost.txt).  FAST and FENCED remain as opt-in A/B forms.

// Reduce the nb posted (min fp, index) pairs, write the candidate record and (kTailApply) fold
// it.  Called by every thread of ONE workgroup; posts and rows written by other workgroups are
// read through volatile (device-coherent) loads.
// `shared`: the posts and rows were written by other workgroups of THIS launch (last-ticket form):
// they were stored with agent-scope atomic stores (write-through) and are read the same way, which
// is coherent across the 8 XCDs without any fence.  A release fence per workgroup is an L2
// write-back on this part: 8 us per 51-workgroup launch against 1.8 us this way
// (tools/archive/barrier_probe.hip).
template <bool SHARED>
__device__ __forceinline__ double load_f64(const double *p)
{
    if (SHARED) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <bool SHARED>
__device__ __forceinline__ void select_final(const TailArgs &a, const double *part_val, const long long *part_idx,
                                             unsigned nb, double *s_val, long long *s_idx)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nw = blockDim.x / kWave;
    double best = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) {
        const double v = load_f64<SHARED>(part_val + b);
        const long long ix = SHARED ? __hip_atomic_load(part_idx + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : part_idx[b];
        if (lex_less(v, ix, best, bi)) {
            best = v;
            bi = ix;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, kWave);
        const long long oi = __shfl_down(bi, off, kWave);
        if (lex_less(ob, oi, best, bi)) {
            best = ob;
            bi = oi;
        }
    }
    __syncthreads();   // s_val / s_idx are free
    if (lane == 0) {
        s_val[wave] = best;
        s_idx[wave] = bi;
    }
    __syncthreads();
    best = s_val[0];
    bi = s_idx[0];
    for (int w = 1; w < nw; ++w)
        if (lex_less(s_val[w], s_idx[w], best, bi)) {
            best = s_val[w];
            bi = s_idx[w];
        }
    if (bi >= a.S) bi = 0;   // np.argmin of an all-inf array
    // (`best` is fp[bi] as posted; +inf: nothing finite yet, the record carries x[0] -- see argmin_block; x was
    // written by an earlier launch, plain loads)
    if (threadIdx.x == 0) a.cand[0] = load_f64<SHARED>(a.fp + bi);
    if (best < INFINITY) {
        for (int64_t d = threadIdx.x; d < a.D; d += blockDim.x) a.cand[1 + d] = load_f64<SHARED>(a.p + bi * a.D + d);
    } else {
        for (int64_t d = threadIdx.x; d < a.D; d += blockDim.x) a.cand[1 + d] = a.x[bi * a.D + d];
    }
    if (a.phases & kTailApply) {
        __syncthreads();
        if (wave == 0) apply_wave(lane, a.D, a.nranks, a.is_init, a.minstep, a.minfunc, a.cands, a.flags, a.best);
    }
}

// ticket != nullptr: the workgroup that draws the last ticket runs select_final (one launch);
// ticket == nullptr: posts only, pso_select_final_kernel follows (two launches, no fences).
__global__ __launch_bounds__(kWave *kSelectWaves) void pso_select_kernel(TailArgs a, double *part_val,
                                                                          long long *part_idx, unsigned *ticket)
{
    // Latency first: this kernel is a chain of dependent round trips to memory and nothing else, so
    // everything whose address is known is requested at once -- the stop flag, the block sums, the
    // personal best and (speculatively) the particle's row -- and only then looked at.
    const long long stopped = a.flags[1];
    __shared__ double s_val[kSelectWaves];
    __shared__ long long s_idx[kSelectWaves];
    __shared__ int s_last;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    double mine = INFINITY;
    long long mi = 0x7fffffffffffffffLL;
    for (int64_t i = (int64_t)blockIdx.x * kSelectWaves + wave; i < a.S; i += (int64_t)gridDim.x * kSelectWaves) {
        const double fp_old = a.fp[i];
        const double x_head = (lane < a.D) ? a.x[i * a.D + lane] : 0.0;   // the first 64 entries of the row, in case it improves
        double f;
        if (a.phases & kTailFinalize)   // same arithmetic and order as finalize_kernel
            f = finalize_value(a.partial + i * a.n_blocks * (a.fit_im ? 2 : 1), a.n_blocks, a.N, a.fit_im);
        else
            f = a.fx[i];
        if (stopped != 0) return;   // (the same in every wave of the grid; nothing has been written)
        if ((a.phases & kTailFinalize) && lane == 0) a.fx[i] = f;
        double cur = fp_old;
        if (f < cur) {   // pyswarm: i_update = fx < fp
            if (ticket && !a.fenced) {   // rows another workgroup of this launch may read: write-through stores
                if (lane < a.D) __hip_atomic_store(a.p + i * a.D + lane, x_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int64_t d = lane + kWave; d < a.D; d += kWave)
                    __hip_atomic_store(a.p + i * a.D + d, a.x[i * a.D + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_store(a.fp + i, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                global_stores_done();   // complete in memory before this workgroup draws its ticket
            } else {
                if (lane < a.D) a.p[i * a.D + lane] = x_head;
                for (int64_t d = lane + kWave; d < a.D; d += kWave) a.p[i * a.D + d] = a.x[i * a.D + d];
                if (lane == 0) a.fp[i] = f;
                if (ticket) __threadfence();   // FENCED: this wave's row and value are released at agent scope
            }
            cur = f;
        }
        if (lex_less(cur, i, mine, mi)) {
            mine = cur;
            mi = i;
        }
    }
    if (stopped != 0) return;   // (waves that had no particle to look at)
    if (lane == 0) {
        s_val[wave] = mine;
        s_idx[wave] = mi;
    }
    __syncthreads();   // (every wave has waited for its own p / fp stores above: a barrier alone does not)
    if (threadIdx.x == 0) {
        double b = s_val[0];
        long long bi = s_idx[0];
        for (int w = 1; w < kSelectWaves; ++w)
            if (lex_less(s_val[w], s_idx[w], b, bi)) {
                b = s_val[w];
                bi = s_idx[w];
            }
        s_last = 0;
        if (ticket && a.fenced) {
            part_val[blockIdx.x] = b;
            part_idx[blockIdx.x] = bi;
            // release: the post (and, through the barrier above and the writers' own fences, the rows);
            // acquire: everything released before the tickets drawn earlier
            s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u);
        } else if (ticket) {
            __hip_atomic_store(part_val + blockIdx.x, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_idx + blockIdx.x, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            global_stores_done();   // the post has completed (asm with a memory clobber: the compiler
                                    // cannot move the ticket above it either)
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);   // ... nor any later load above the ticket
        } else {
            part_val[blockIdx.x] = b;
            part_idx[blockIdx.x] = bi;
        }
    }
    __syncthreads();
    if (!s_last) return;
    // every other workgroup's stores had completed before it drew its ticket
    if (a.fenced)
        __threadfence();   // FENCED: every reading wave acquires at agent scope
    else
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    select_final<true>(a, part_val, part_idx, gridDim.x, s_val, s_idx);
}

__global__ __launch_bounds__(1024) void pso_select_final_kernel(TailArgs a, const double *part_val,
                                                                const long long *part_idx, unsigned nb)
{
    if (a.flags[1] != 0) return;
    __shared__ double s_val[16];
    __shared__ long long s_idx[16];
    select_final<false>(a, part_val, part_idx, nb, s_val, s_idx);
}


int bind_pso(const nmrfit_pso *pso)
{
    if (!pso || !pso->ctx) {
        set_error("null swarm handle");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipSetDevice(pso->ctx->device));
    return NMRFIT_OK;
}

// One workgroup handles the whole tail only while that is cheaper than the two launches of the
// many-workgroup form (select + update, ~4.8 us each): the single-workgroup tail takes
// ~5.5 us + 1.55 us per 1000 swarm elements (7.2 us at 50 x 22, 12.5 us at 204 x 22).
int64_t tail_max_elems()
{
    static const int64_t v = [] {
        const char *e = getenv("NMRFIT_TAIL_MAX_ELEMS");   // tuning knob
        return e ? (int64_t)atoll(e) : (int64_t)2560;
    }();
    return v;
}

bool small_swarm(const nmrfit_pso *pso) { return pso->S > 0 && pso->S * pso->D <= tail_max_elems(); }

TailArgs tail_args(nmrfit_pso *pso, const ObjectiveDeferred &def, int phases)
{
    TailArgs a{};
    a.S = pso->S;
    a.D = pso->D;
    a.offset = pso->offset;
    a.N = pso->ctx->N;
    a.n_blocks = def.n_blocks;
    a.seed = pso->prm.seed;
    a.omega = pso->prm.omega;
    a.phip = pso->prm.phip;
    a.phig = pso->prm.phig;
    a.minstep = pso->prm.minstep;
    a.minfunc = pso->prm.minfunc;
    a.phases = phases;
    a.fit_im = def.fit_im;
    a.nranks = 1;
    a.is_init = 0;
    a.fenced = (pso->handover == NMRFIT_HANDOVER_FENCED) ? 1 : 0;
    a.partial = def.partial;
    a.lb = pso->d_lb;
    a.ub = pso->d_ub;
    a.cands = pso->d_cand;
    a.x = pso->d_x;
    a.v = pso->d_v;
    a.p = pso->d_p;
    a.fx = pso->d_fx;
    a.fp = pso->d_fp;
    a.cand = pso->d_cand;
    a.best = pso->d_best;
    a.flags = pso->d_flags;
    return a;
}

int launch_tail(nmrfit_pso *pso, const TailArgs &a)
{
    hipLaunchKernelGGL(pso_tail_kernel, dim3(1), dim3(1024), 0, pso->ctx->stream, a);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

// The position update rides in the objective kernel's prologue (objective.hip) unless the
// parameter vector is too long for its LDS copy or NMRFIT_NO_FUSED_UPDATE is set (A/B knob).
bool fuse_update(const nmrfit_pso *pso)
{
    static const bool off = getenv("NMRFIT_NO_FUSED_UPDATE") != nullptr;
    return !off && pso->S > 0 && pso->D <= kFusedMaxD;
}

int launch_update(nmrfit_pso *pso)
{
    const int64_t n = pso->S * pso->D;
    if (n == 0) return NMRFIT_OK;
    hipLaunchKernelGGL(pso_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pso->ctx->stream, pso->S,
                       pso->D, pso->offset, pso->prm.seed, pso->prm.omega, pso->prm.phip, pso->prm.phig, pso->d_flags,
                       pso->d_best, pso->d_lb, pso->d_ub, pso->d_p, pso->d_x, pso->d_v);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

// [position update +] objective + personal bests + local candidate [+ fold].
//   advance: move the swarm one generation first (fused into the objective launch when possible)
//   more:    kTailApply folds this rank's own candidate in the same launch (single-rank runs)
// Deferred form: fold the generation whose personal bests are still waiting (argmin over fp, candidate record,
// pyswarm's acceptance / stopping rule) in ONE single-workgroup launch, in place on the current state block.
// Every entry point that shows the swarm's state to the outside, or continues it in another way, comes here first.
int flush_fold(nmrfit_pso *pso)
{
    if (!pso->fold_pending) return NMRFIT_OK;
    TailArgs a = tail_args(pso, ObjectiveDeferred{}, kTailArgmin | kTailApply);
    const int rc = launch_tail(pso, a);
    if (rc == NMRFIT_OK) pso->fold_pending = false;   // (a failed launch leaves the generation waiting, as evaluate_and_select does)
    return rc;
}

int evaluate_and_select(nmrfit_pso *pso, bool advance, int more = 0, int is_init = 0)
{
    nmrfit_ctx *ctx = pso->ctx;
    const int64_t S = pso->S, D = pso->D;
    // the whole generation in the objective launch (single rank: this rank's candidate is the swarm's), its fold deferred
    // into the next launch's prologue -- if the launch geometry allows, which launch_objective decides
    const bool want_deferred = advance && fuse_update(pso) && pso->fused_pbest && pso->fused_tail && (more & kTailApply) && !is_init;
    int rc;
    if (!want_deferred && (rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    if (S == 0) {
        // an empty shard still posts its (+inf, zeros) candidate
        hipLaunchKernelGGL(pso_argmin_kernel, dim3(1), dim3(1024), 0, ctx->stream, S, D, pso->d_flags, pso->d_fp,
                           pso->d_p, pso->d_x, pso->d_cand);
        NMRFIT_HIP(hipGetLastError());
        return NMRFIT_OK;
    }
    ObjectiveDeferred def;
    if (advance && fuse_update(pso)) {
        PsoFused f;
        f.x_in = pso->d_x;
        f.v_in = pso->d_v;
        f.x_out = pso->d_x2;
        f.v_out = pso->d_v2;
        f.p = pso->d_p;
        f.best = pso->d_best;
        f.lb = pso->d_lb;
        f.ub = pso->d_ub;
        f.flags = pso->d_flags;
        f.seed = pso->prm.seed;
        f.offset = pso->offset;
        f.omega = pso->prm.omega;
        f.phip = pso->prm.phip;
        f.phig = pso->prm.phig;
        // personal bests in the same launch when one workgroup holds a whole particle (the launch decides from
        // its geometry and says so in def.pbest_done); d_fp == d_p + S*D, see nmrfit_pso_create
        if (pso->fused_pbest) f.pbest = 1u;
        // ... and, single rank, the rest of the generation too (argmin over fp, candidate record, fold), deferred into the
        // NEXT launch's prologue: ONE launch per generation
        if (want_deferred) {
            f.tail = 1u;
            f.cand = pso->d_cand;
            f.minstep = pso->prm.minstep;
            f.minfunc = pso->prm.minfunc;
            f.pending = pso->fold_pending ? 1u : 0u;   // this launch first folds the generation before it, if one is waiting
            f.flip = (int)(pso->d_best_alt - pso->d_best);
            f.pflip = (int)(pso->d_p_alt - pso->d_p);
        }
        rc = launch_objective(ctx, S, pso->P, pso->d_x2, pso->d_fx, nullptr, &def, &f);
        if (rc != NMRFIT_OK) return rc;
        if (def.need_flush) {   // (nothing was launched: the geometry changed under a waiting fold)
            if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
            f.pending = 0u;
            rc = launch_objective(ctx, S, pso->P, pso->d_x2, pso->d_fx, nullptr, &def, &f);
            if (rc != NMRFIT_OK) return rc;
        }
        if (def.tail_done) {
            std::swap(pso->d_p, pso->d_p_alt);   // every particle's personal best went to the other buffer
            pso->d_fp = pso->d_p + S * D;
            if (f.pending != 0u) {   // workgroup 0 wrote the folded state into the other block
                std::swap(pso->d_best, pso->d_best_alt);
                std::swap(pso->d_flags, pso->d_flags_alt);
            }
            pso->fold_pending = true;   // this launch's own generation waits for the next launch (or flush_fold)
        } else if (pso->fold_pending) {
            // (cannot happen: a launch that does not fold a waiting generation reports need_flush above)
            set_error("internal: deferred fold lost");
            return NMRFIT_E_STATE;
        }
        std::swap(pso->d_x, pso->d_x2);   // d_x / d_v: the state the kernel has just written
        std::swap(pso->d_v, pso->d_v2);
    } else {
        if (advance && (rc = launch_update(pso)) != NMRFIT_OK) return rc;
        rc = launch_objective(ctx, S, pso->P, pso->d_x, pso->d_fx, nullptr, &def);
        if (rc != NMRFIT_OK) return rc;
    }
    pso->last_launches = def.tail_done ? 1 : 2;
    if (def.tail_done) return NMRFIT_OK;   // the objective launch was the whole generation
    if (def.pbest_done) {
        // The particle's whole step (update, evaluate, personal best) happened in the objective launch: what is
        // left is the argmin over fp, the candidate record and -- single rank -- the fold, none of which needs
        // more than ONE workgroup and none of which hands anything over inside a launch.
        TailArgs a = tail_args(pso, def, kTailArgmin | (more & kTailApply));
        a.is_init = is_init;
        return launch_tail(pso, a);
    }
    TailArgs a = tail_args(pso, def, (def.needed ? kTailFinalize : 0) | kTailPbest | kTailArgmin | (more & kTailApply));
    a.is_init = is_init;
    if (small_swarm(pso)) return launch_tail(pso, a);
    // up to 256 workgroups (1024 particles): one launch, the last-ticket workgroup finishes; larger
    // (or NMRFIT_HANDOVER_TWO_LAUNCH): posts, then a single-workgroup launch
    const int64_t nb64 = (S + kSelectWaves - 1) / kSelectWaves;
    const unsigned nb = (unsigned)std::min<int64_t>(nb64, kSelectMaxPosts);
    static const unsigned ticket_blocks = [] {
        const char *e = getenv("NMRFIT_TICKET_BLOCKS");   // tuning knob
        return e ? (unsigned)atoi(e) : (unsigned)kSelectTicketBlocks;
    }();
    const bool ticket = nb <= ticket_blocks && pso->handover != NMRFIT_HANDOVER_TWO_LAUNCH;
    hipLaunchKernelGGL(pso_select_kernel, dim3(nb), dim3(kWave * kSelectWaves), 0, ctx->stream, a, pso->d_part_val,
                       pso->d_part_idx, ticket ? pso->d_ticket : nullptr);
    NMRFIT_HIP(hipGetLastError());
    if (!ticket) {
        pso->last_launches = 3;
        hipLaunchKernelGGL(pso_select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, a, pso->d_part_val,
                           pso->d_part_idx, nb);
        NMRFIT_HIP(hipGetLastError());
    }
    return NMRFIT_OK;
}

}  // namespace
}  // namespace nmrfit

using namespace nmrfit;

#pragma GCC visibility push(default)   // the C-ABI: the only symbols the library exports (build.sh: -fvisibility=hidden)
extern "C" {

int nmrfit_pso_create(nmrfit_ctx *ctx, int64_t S_local, int64_t S_global, int64_t offset, int32_t P,
                      const double *lower, const double *upper, const nmrfit_pso_params *params, nmrfit_pso **out)
{
    if (!out) {
        set_error("null out pointer");
        return NMRFIT_E_INVALID;
    }
    *out = nullptr;
    if (!ctx || !lower || !upper || !params || S_local < 0 || S_global <= 0 || offset < 0 ||
        offset + S_local > S_global || P < 0 || P > kMaxPeaks) {
        set_error("nmrfit_pso_create: bad arguments");
        return NMRFIT_E_INVALID;
    }
    const int64_t D = 4 + 3 * (int64_t)P;
    for (int64_t d = 0; d < D; ++d) {
        if (!(upper[d] > lower[d])) {   // pyswarm: assert np.all(ub > lb)
            set_error("All upper-bound values must be greater than lower-bound values");
            return NMRFIT_E_INVALID;
        }
    }
    NMRFIT_HIP(hipSetDevice(ctx->device));
    nmrfit_pso *pso = new (std::nothrow) nmrfit_pso();
    if (!pso) {
        set_error("out of host memory");
        return NMRFIT_E_INVALID;
    }
    pso->ctx = ctx;
    pso->S = S_local;
    pso->S_global = S_global;
    pso->offset = offset;
    pso->P = P;
    pso->D = D;
    pso->prm = *params;
    if (const char *e = getenv("NMRFIT_HANDOVER")) {   // A/B knob: every new swarm's hand-over form
        if (!strcmp(e, "fast")) pso->handover = NMRFIT_HANDOVER_FAST;
        if (!strcmp(e, "fenced")) pso->handover = NMRFIT_HANDOVER_FENCED;
    }
    if (getenv("NMRFIT_NO_FUSED_PBEST")) pso->fused_pbest = false;   // A/B knob
    if (getenv("NMRFIT_NO_FUSED_TAIL")) pso->fused_tail = false;     // A/B knob
    const size_t sd = (size_t)std::max<int64_t>(S_local * D, 1) * sizeof(double);
    const size_t s1 = (size_t)std::max<int64_t>(S_local, 1) * sizeof(double);
#define PSO_HIP(call)                                                   \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) {                                         \
            int _rc = hip_fail(_e, #call, __FILE__, __LINE__);          \
            nmrfit_pso_destroy(pso);                                    \
            return _rc;                                                 \
        }                                                               \
    } while (0)
    // ONE allocation for the whole swarm state (every hipMalloc / hipFree is a synchronising call; a default fit
    // is 30 ms).  Layout: 256-byte aligned pieces; fp[S] right behind p[S x D] (the objective kernel's fused
    // personal-best step finds fp from p without another pointer argument, objective.hip).
    {
        const size_t nposts = (size_t)kSelectMaxPosts;
        // (fg, best_f, g[D], best_x[D] | generations, stop code): two copies, 256-byte aligned, the flags right behind
        // the doubles -- the deferred fold of a fused launch reads one and writes the other (PsoFused::flip)
        const size_t state_bytes = (((size_t)(2 + 2 * D) * sizeof(double) + 2 * sizeof(long long)) + 255) & ~(size_t)255;
        const size_t sizes[] = {(size_t)D * sizeof(double), (size_t)D * sizeof(double), sd, sd, sd, sd,
                                (size_t)std::max<int64_t>(S_local * D, 0) * sizeof(double) + s1, s1,
                                (size_t)(D + 1) * sizeof(double), 2 * state_bytes,
                                (size_t)std::max<int64_t>(S_local * D, 0) * sizeof(double) + s1,
                                nposts * sizeof(double), nposts * sizeof(long long), 256};
        size_t off[sizeof(sizes) / sizeof(sizes[0])], total = 0;
        for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
            off[i] = total;
            total += (sizes[i] + 255) & ~(size_t)255;
        }
        PSO_HIP(hipMalloc(&pso->d_block, total));
        unsigned char *base = reinterpret_cast<unsigned char *>(pso->d_block);
        pso->d_lb = reinterpret_cast<double *>(base + off[0]);
        pso->d_ub = reinterpret_cast<double *>(base + off[1]);
        pso->d_x = reinterpret_cast<double *>(base + off[2]);
        pso->d_v = reinterpret_cast<double *>(base + off[3]);
        pso->d_x2 = reinterpret_cast<double *>(base + off[4]);
        pso->d_v2 = reinterpret_cast<double *>(base + off[5]);
        pso->d_p = reinterpret_cast<double *>(base + off[6]);
        pso->d_fp = pso->d_p + S_local * D;
        pso->d_fx = reinterpret_cast<double *>(base + off[7]);
        pso->d_cand_own = reinterpret_cast<double *>(base + off[8]);
        pso->d_cand = pso->d_cand_own;
        pso->d_best = reinterpret_cast<double *>(base + off[9]);
        pso->d_flags = reinterpret_cast<long long *>(pso->d_best + 2 + 2 * D);
        pso->d_p_alt = reinterpret_cast<double *>(base + off[10]);
        pso->d_best_alt = reinterpret_cast<double *>(base + off[9] + state_bytes);
        pso->d_flags_alt = reinterpret_cast<long long *>(pso->d_best_alt + 2 + 2 * D);
        PSO_HIP(hipMemsetAsync(pso->d_best, 0, 2 * state_bytes, ctx->stream));
        pso->d_part_val = reinterpret_cast<double *>(base + off[11]);
        pso->d_part_idx = reinterpret_cast<long long *>(base + off[12]);
        pso->d_ticket = reinterpret_cast<unsigned *>(base + off[13]);
        // (posts beyond the workgroups a swarm of this size launches are never read)
        const size_t used_posts = (size_t)std::min<int64_t>((S_local + kSelectWaves - 1) / kSelectWaves + 1, kSelectMaxPosts);
        PSO_HIP(hipMemsetAsync(pso->d_part_idx, 0, used_posts * sizeof(long long), ctx->stream));
        PSO_HIP(hipMemsetAsync(pso->d_ticket, 0, sizeof(unsigned), ctx->stream));
    }
    PSO_HIP(hipMemcpyAsync(pso->d_lb, lower, (size_t)D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PSO_HIP(hipMemcpyAsync(pso->d_ub, upper, (size_t)D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PSO_HIP(hipStreamSynchronize(ctx->stream));
#undef PSO_HIP
    *out = pso;
    return NMRFIT_OK;
}

int nmrfit_pso_destroy(nmrfit_pso *pso)
{
    if (!pso) return NMRFIT_OK;
    if (pso->comm) comm_detach(pso->comm);
    pso->comm = nullptr;
    if (pso->ctx) {
        (void)hipSetDevice(pso->ctx->device);
        (void)hipStreamSynchronize(pso->ctx->stream);
    }
    if (pso->d_block) (void)hipFree(pso->d_block);
    delete pso;
    return NMRFIT_OK;
}

int nmrfit_pso_init(nmrfit_pso *pso)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    nmrfit_ctx *ctx = pso->ctx;
    const int64_t n = pso->S * pso->D;
    pso->fold_pending = false;   // (a new swarm: whatever generation was waiting to be folded is gone with the old one)
    NMRFIT_HIP(hipMemsetAsync(pso->d_flags, 0, 2 * sizeof(long long), ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(pso_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, pso->S, pso->D,
                           pso->offset, pso->prm.seed, pso->d_lb, pso->d_ub, pso->d_x, pso->d_v, pso->d_p, pso->d_fp);
        NMRFIT_HIP(hipGetLastError());
    }
    rc = evaluate_and_select(pso, /*advance=*/false);
    if (rc != NMRFIT_OK) return rc;
    pso->initialized = true;
    pso->seeded = false;
    return NMRFIT_OK;
}

int nmrfit_pso_step_local(nmrfit_pso *pso)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if (!pso->initialized) {
        set_error("nmrfit_pso_step_local before nmrfit_pso_init");
        return NMRFIT_E_STATE;
    }
    return evaluate_and_select(pso, /*advance=*/true);
}

int nmrfit_pso_candidate_dev(nmrfit_pso *pso, double **dptr)
{
    if (!pso || !dptr) {
        set_error("null argument");
        return NMRFIT_E_INVALID;
    }
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;   // (the record of the LAST generation, on the stream before any copy)
    *dptr = pso->d_cand;
    return NMRFIT_OK;
}

int nmrfit_pso_set_candidate_dev(nmrfit_pso *pso, double *dptr)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(pso->ctx->stream));
    double *next = dptr ? dptr : pso->d_cand_own;
    if (next != pso->d_cand) {   // carry the current record over
        NMRFIT_HIP(hipMemcpy(next, pso->d_cand, (size_t)(pso->D + 1) * sizeof(double), hipMemcpyDeviceToDevice));
        pso->d_cand = next;
    }
    return NMRFIT_OK;
}

int nmrfit_pso_apply_global_dev(nmrfit_pso *pso, const double *d_candidates, int32_t nranks)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if (!d_candidates || nranks < 1) {
        set_error("nmrfit_pso_apply_global_dev: bad arguments");
        return NMRFIT_E_INVALID;
    }
    if (!pso->initialized) {
        set_error("nmrfit_pso_apply_global_dev before nmrfit_pso_init");
        return NMRFIT_E_STATE;
    }
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    const int is_init = pso->seeded ? 0 : 1;   // first fold after init sets (g, fg) with no stop test
    hipLaunchKernelGGL(pso_apply_kernel, dim3(1), dim3(kWave), 0, pso->ctx->stream, pso->D, (int)nranks, is_init,
                       pso->prm.minstep, pso->prm.minfunc, d_candidates, pso->d_flags, pso->d_best);
    NMRFIT_HIP(hipGetLastError());
    pso->seeded = true;
    return NMRFIT_OK;
}

int nmrfit_pso_set_comm(nmrfit_pso *pso, nmrfit_comm *comm)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    // the all-gather is enqueued on THIS swarm's context's stream, between its select and fold kernels; the
    // communicator may come from another context of the same device (one ncclCommInitRank serves fit after fit)
    if (comm && (!comm_ctx(comm) || comm_ctx(comm)->device != pso->ctx->device)) {
        set_error("nmrfit_pso_set_comm: the communicator was created on another device than the swarm's context");
        return NMRFIT_E_INVALID;
    }
    if (comm == pso->comm) return NMRFIT_OK;
    // ONE swarm at a time per communicator: two swarms' all-gathers -- issued from two host threads on two streams, in
    // whatever order the threads happen to run -- would pair up differently on different ranks (a hang or mixed-up
    // records), and both would write the communicator's one gather buffer
    if (comm && !comm_attach(comm)) {   // (nmrfit_comm_destroy refuses while a swarm still points at it)
        set_error("nmrfit_pso_set_comm: another swarm is attached to this communicator (one swarm at a time: detach or "
                  "destroy it first)");
        return NMRFIT_E_STATE;
    }
    if ((rc = flush_fold(pso)) == NMRFIT_OK) {
        hipError_t e = hipStreamSynchronize(pso->ctx->stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    }
    if (rc != NMRFIT_OK) {
        if (comm) comm_detach(comm);
        return rc;
    }
    if (pso->comm) comm_detach(pso->comm);
    pso->comm = comm;
    return NMRFIT_OK;
}

int nmrfit_pso_set_fused_pbest(nmrfit_pso *pso, int enable)
{
    if (!pso) {
        set_error("null swarm handle");
        return NMRFIT_E_INVALID;
    }
    pso->fused_pbest = enable != 0;
    return NMRFIT_OK;
}

int nmrfit_pso_set_fused_tail(nmrfit_pso *pso, int enable)
{
    if (!pso) {
        set_error("null swarm handle");
        return NMRFIT_E_INVALID;
    }
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;   // (like its sibling entry points: nothing waits across a change of form)
    pso->fused_tail = enable != 0;
    return NMRFIT_OK;
}

int nmrfit_pso_last_launches(const nmrfit_pso *pso, int32_t *launches)
{
    if (!pso || !launches) {
        set_error("null argument");
        return NMRFIT_E_INVALID;
    }
    *launches = pso->last_launches;
    return NMRFIT_OK;
}

int nmrfit_pso_set_handover(nmrfit_pso *pso, int mode)
{
    if (!pso) {
        set_error("null swarm handle");
        return NMRFIT_E_INVALID;
    }
    if (mode != NMRFIT_HANDOVER_FAST && mode != NMRFIT_HANDOVER_FENCED && mode != NMRFIT_HANDOVER_TWO_LAUNCH) {
        set_error("nmrfit_pso_set_handover: mode must be one of NMRFIT_HANDOVER_*");
        return NMRFIT_E_INVALID;
    }
    pso->handover = mode;
    return NMRFIT_OK;
}

// candidate exchange (one all-gather over the attached communicator) + fold
static int exchange_and_fold(nmrfit_pso *pso)
{
    if (!pso->comm) return nmrfit_pso_apply_global_dev(pso, pso->d_cand, 1);
    const double *d_all = nullptr;
    int32_t nranks = 1;
    int rc = comm_all_gather(pso->comm, pso->ctx->stream, pso->d_cand, pso->D + 1, &d_all);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = nmrfit_comm_info(pso->comm, nullptr, &nranks, nullptr)) != NMRFIT_OK) return rc;
    return nmrfit_pso_apply_global_dev(pso, d_all, nranks);
}

int nmrfit_pso_step(nmrfit_pso *pso)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if (!pso->initialized) {
        set_error("nmrfit_pso_step before nmrfit_pso_init");
        return NMRFIT_E_STATE;
    }
    if (!pso->seeded) return exchange_and_fold(pso);   // generation 0: (g, fg) from the initial candidates
    if (pso->comm || pso->S == 0) {
        if ((rc = nmrfit_pso_step_local(pso)) != NMRFIT_OK) return rc;
        return exchange_and_fold(pso);
    }
    // single rank: the fold and the stopping rule ride in the select launch
    return evaluate_and_select(pso, /*advance=*/true, kTailApply);
}

int nmrfit_pso_status(nmrfit_pso *pso, int64_t *iteration, int32_t *stop_code, double *fg)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    long long flags[2];
    double head[2];
    hipStream_t st = pso->ctx->stream;
    NMRFIT_HIP(hipMemcpyAsync(flags, pso->d_flags, sizeof flags, hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipMemcpyAsync(head, pso->d_best, sizeof head, hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipStreamSynchronize(st));
    if (iteration) *iteration = flags[0];
    if (stop_code) *stop_code = (int32_t)flags[1];
    if (fg) *fg = head[0];
    return NMRFIT_OK;
}

int nmrfit_pso_best(nmrfit_pso *pso, double *x_best, double *f_best)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    hipStream_t st = pso->ctx->stream;
    if (x_best)
        NMRFIT_HIP(hipMemcpyAsync(x_best, pso->d_best + 2 + pso->D, (size_t)pso->D * sizeof(double), hipMemcpyDeviceToHost, st));
    if (f_best) NMRFIT_HIP(hipMemcpyAsync(f_best, pso->d_best + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipStreamSynchronize(st));
    return NMRFIT_OK;
}

int nmrfit_pso_run(nmrfit_pso *pso, int64_t maxiter, int32_t check_every)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if (maxiter < 0 || check_every < 1) {
        set_error("nmrfit_pso_run: maxiter must be >= 0 and check_every >= 1");
        return NMRFIT_E_INVALID;
    }
    if (!pso->initialized) {
        if ((rc = nmrfit_pso_init(pso)) != NMRFIT_OK) return rc;
    }
    if (!pso->seeded) {
        if ((rc = exchange_and_fold(pso)) != NMRFIT_OK) return rc;
    }
    // A generation is the objective launch (which also advances the swarm) and the select launch
    // (which, single rank, also folds the candidate and applies the stopping rule).  Sharded swarm:
    // every rank runs the same generations and folds the same gathered records, so every rank reads
    // the same stop flag at the same poll and leaves the loop together.
    for (int64_t it = 1; it <= maxiter; ++it) {
        if ((rc = nmrfit_pso_step(pso)) != NMRFIT_OK) return rc;
        if (it % check_every == 0 || it == maxiter) {
            int32_t stop = 0;
            if ((rc = nmrfit_pso_status(pso, nullptr, &stop, nullptr)) != NMRFIT_OK) return rc;
            if (stop) break;
        }
    }
    return NMRFIT_OK;
}

int nmrfit_pso_get_state(nmrfit_pso *pso, double *x, double *v, double *p, double *fx, double *fp)
{
    int rc = bind_pso(pso);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = flush_fold(pso)) != NMRFIT_OK) return rc;
    hipStream_t st = pso->ctx->stream;
    const size_t sd = (size_t)(pso->S * pso->D) * sizeof(double), s1 = (size_t)pso->S * sizeof(double);
    if (sd) {
        if (x) NMRFIT_HIP(hipMemcpyAsync(x, pso->d_x, sd, hipMemcpyDeviceToHost, st));
        if (v) NMRFIT_HIP(hipMemcpyAsync(v, pso->d_v, sd, hipMemcpyDeviceToHost, st));
        if (p) NMRFIT_HIP(hipMemcpyAsync(p, pso->d_p, sd, hipMemcpyDeviceToHost, st));
        if (fx) NMRFIT_HIP(hipMemcpyAsync(fx, pso->d_fx, s1, hipMemcpyDeviceToHost, st));
        if (fp) NMRFIT_HIP(hipMemcpyAsync(fp, pso->d_fp, s1, hipMemcpyDeviceToHost, st));
    }
    NMRFIT_HIP(hipStreamSynchronize(st));
    return NMRFIT_OK;
}

}  // extern "C"
#pragma GCC visibility pop
